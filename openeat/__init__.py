"""Drop-in alias: ``import openeat.models.asr_model`` etc. resolve to ``openeat_amd``.

Lets code written against TongtongSong/OpenEAT (``from openeat.models.asr_model import ASRModel``,
``from openeat.utils.executor import Executor`` ...) run on the MI355X-native path unmodified.
Only the hot-path modules exist (see DESIGN.md section 7 for what is out of scope)."""
import importlib
import sys

_SUBMODULES = [
    "models", "models.asr_model", "models.language_model",
    "modules", "modules.attention", "modules.cmvn", "modules.convolution", "modules.ctc", "modules.decoder",
    "modules.decoder_layer", "modules.embedding", "modules.encoder", "modules.encoder_layer",
    "modules.label_smoothing_loss", "modules.positionwise_feed_forward", "modules.subsampling", "modules.swish",
    "dataset", "dataset.audio_processor", "dataset.dataset", "dataset.sampler",
    "utils", "utils.checkpoint", "utils.cmvn", "utils.common", "utils.executor", "utils.mask", "utils.scheduler",
]
for _name in _SUBMODULES:
    _mod = importlib.import_module("openeat_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    if "." not in _name:
        setattr(sys.modules[__name__], _name, _mod)
