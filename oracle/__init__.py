"""CPU oracle for the fbank -> Conformer encoder -> CTC / attention-decoder path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker.  The product
(``openeat_amd``) never imports this package and fails loudly when its HIP
library is missing.

The oracle is a functional, state-dict driven restatement of the reference's
PyTorch-CPU algorithm (every function cites the reference file:line it
follows).  It is pinned against golden vectors generated from the reference
itself (``tests/golden/make_fixtures.py``; vectors under ``tests/golden``).
The fbank front end is the exception: its arithmetic lives in torchaudio,
which is absent from the reference tree and from this image -> that part is
"parity unpinned" (see ``oracle/fbank.py``).
"""
