"""MI355X-native ASR hot path behind the OpenEAT module API.

``openeat_amd.{models,modules,utils}`` mirror ``openeat.{models,modules,utils}``
of TongtongSong/OpenEAT (same class names, constructor kwargs, forward
signatures and state-dict keys); the arithmetic runs in hand-written gfx950
kernels reached through the C ABI of ``lib/libopeneat_hip.so``
(``include/openeat_hip.h``).  There is no CPU fallback.
"""
__version__ = "0.1.0"
