"""GPU: a captured HIP graph and eager code never share pre-split operands (openeat_amd.planes.capture_scope).

Round 3's GPUTEST went red in test_batched_rescoring_from_cached_graphs_equals_eager[bf16x6-planes-forced]: the first REPLAY of
the decode stage-1 graph returned empty hypotheses.  Cause: under OE_PLANES=all the registry of pre-split operands is keyed by
device address; the position table slice that RelPositionMultiHeadedAttention hands to linear_pos (attention.py:166-209,
embedding.py:75-88) has the same address at every call, so the capture FOUND the planes an earlier eager call had made and
baked their address into its launches - planes owned by a 64-entry FIFO, freed a few registrations later and overwritten by
whatever the allocator handed that block to next.  Whether the replay then read garbage depended on the allocator's history,
i.e. on the box.  The tests below make the overwrite deterministic: after the capture everything the eager registry and the
weight cache own is released and the freed blocks are re-filled with NaN bit patterns before the replay."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden, load_golden_json  # noqa: E402
from openeat_amd.models.asr_model import ASRModel  # noqa: E402

DEV = "cuda"


@pytest.fixture
def planes_everywhere():
    from openeat_amd import hip, planes
    old = (hip.GEMM_PRECISION, planes.MIN_SPLIT_ELEMS, planes.POLICY)
    hip.GEMM_PRECISION, planes.MIN_SPLIT_ELEMS, planes.POLICY = 6, 0, "all"
    hip.lib().oe_gemm_pl_config(0, -1, -1, -1)
    planes.clear_all()
    yield planes
    hip.GEMM_PRECISION, planes.MIN_SPLIT_ELEMS, planes.POLICY = old
    hip.lib().oe_gemm_pl_config(96, 0, 0, 8)
    planes.clear_all()


def _model():
    g = load_golden("f12_tiny_conformer")
    meta = load_golden_json("f12_tiny_conformer")
    model = ASRModel(80, meta["V"], **meta["kwargs"])
    model.load_state_dict(g["sd"])
    return model.to(DEV).eval()


def _case(seed, lens):
    torch.manual_seed(seed)
    feats = torch.randn(len(lens), 97, 80, device=DEV)
    for b, n in enumerate(lens):
        feats[b, n:] = 0.0
    return feats, torch.tensor(lens, dtype=torch.int32, device=DEV)


def release_and_poison(planes):
    """Drop every planes buffer eager code owns and fill the blocks they occupied with 0xFF bytes (bf16 NaN): same-size
    requests get the just-freed blocks back from the caching allocator.  Returns the poison tensors (keep them alive)."""
    sizes = {}
    for e in list(planes._REG.values()) + list(planes._WCACHE.values()):
        n = e[0].t.numel() * e[0].t.element_size()
        sizes[n] = sizes.get(n, 0) + 1
    planes.clear_all()
    torch.cuda.synchronize()
    poison = []
    for n, cnt in sizes.items():
        for _ in range(cnt + 8):
            poison.append(torch.full((n,), 0xFF, dtype=torch.uint8, device=DEV))
    torch.cuda.synchronize()
    return poison


def test_replay_survives_the_release_of_every_eager_planes_buffer(planes_everywhere):
    planes = planes_everywhere
    model = _model()
    a, b = _case(41, [97, 83, 64, 41, 23]), _case(42, [97, 90, 97, 60, 97])
    kw = dict(ctc_weight=0.5, reverse_weight=0.3)
    with torch.no_grad():
        want_a = model.attention_rescoring_batch(*a, 4, use_graphs=False, **kw)       # eager: the registry now holds planes
        want_b = model.attention_rescoring_batch(*b, 4, use_graphs=False, **kw)
        assert len(planes._REG) > 0 and len(planes._WCACHE) > 0                      # (the hazard's precondition is really there)
        got_a = model.attention_rescoring_batch(*a, 4, use_graphs=True, **kw)         # eager + capture
        assert got_a == want_a
        poison = release_and_poison(planes)
        got_b = model.attention_rescoring_batch(*b, 4, use_graphs=True, **kw)         # replay of stage 1 (and of stage 2 if lengths agree)
        again_a = model.attention_rescoring_batch(*a, 4, use_graphs=True, **kw)
    assert poison
    assert any(len(h) > 0 for h in want_b)
    assert got_b == want_b and again_a == want_a
    assert any(k[0] == "s1" and v is not None for k, v in model._decode_graphs.items())


def test_eager_code_never_finds_a_captures_planes(planes_everywhere):
    """The other direction: what a capture registers points into the graph's private pool, which holds nothing until the first
    replay.  After a capture the registry must be what it was before, and an eager pass right behind the capture (no replay
    yet) must give the eager result."""
    planes = planes_everywhere
    model = _model()
    a = _case(43, [50, 97, 30, 88, 61])
    kw = dict(ctc_weight=0.5, reverse_weight=0.3)
    with torch.no_grad():
        want = model.attention_rescoring_batch(*a, 4, use_graphs=False, **kw)
        reg = planes._REG
        got = model.attention_rescoring_batch(*a, 4, use_graphs=True, **kw)          # eager + capture of both stages
        assert planes._REG is reg and planes._CAPTURE_DEPTH == 0                      # the outer registry is back
        eager_again = model.attention_rescoring_batch(*a, 4, use_graphs=False, **kw)  # right behind the capture, before any replay
    assert got == want and eager_again == want


def test_weights_outside_an_arena_follow_their_values_through_a_replay(planes_everywhere):
    """A cached weight split handed to a capture would be read by every replay even after the weights changed (in place:
    load_state_dict, an averaged checkpoint).  Inside capture_scope the split is a captured launch: the replay follows."""
    planes = planes_everywhere
    model = _model()
    a = _case(44, [97, 97, 80, 97, 70])
    kw = dict(ctc_weight=0.5, reverse_weight=0.3)
    with torch.no_grad():
        model.attention_rescoring_batch(*a, 4, use_graphs=True, **kw)                 # eager + capture
        first = model.attention_rescoring_batch(*a, 4, use_graphs=True, **kw)         # replay
        for p in model.encoder.parameters():
            if p.dim() == 2:
                p.mul_(0.5)
        planes.clear_all()
        want = model.attention_rescoring_batch(*a, 4, use_graphs=False, **kw)
        got = model.attention_rescoring_batch(*a, 4, use_graphs=True, **kw)           # replay with the new weights
    assert got == want and isinstance(first, list)


def test_arena_planes_are_never_allocated_inside_a_capture():
    from openeat_amd import arena as A, hip, planes
    old = (hip.GEMM_PRECISION, planes.HYB_MIN_ROWS)
    hip.GEMM_PRECISION, planes.HYB_MIN_ROWS = 6, 0
    lin = torch.nn.Linear(64, 64).to(DEV)
    ar = A.ParamArena(lin).activate()
    try:
        assert ar.planes is None
        g = torch.cuda.CUDAGraph()
        with planes.capture_scope():
            assert ar.planes is not None                       # allocated ahead of the capture, from the ordinary pool
            with torch.cuda.graph(g):
                pl = planes.arena_weight(lin.weight)           # the split is a captured launch
                assert pl is not None and ar._planes_fresh
        assert not ar._planes_fresh                            # only recorded: the next eager reader splits for real
        ar2 = A.ParamArena(torch.nn.Linear(64, 64).to(DEV))
        g2 = torch.cuda.CUDAGraph()
        with pytest.raises(RuntimeError, match="before a graph capture"):
            with torch.cuda.graph(g2):
                _ = torch.ones(8, device=DEV) * 2.0            # (a capture with one node, whatever happens next)
                ar2.alloc_planes()
    finally:
        hip.GEMM_PRECISION, planes.HYB_MIN_ROWS = old
        ar.deactivate()
