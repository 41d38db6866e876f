"""Stress probe of the LayerNorm-backward epilogue of oe_rowgemm6 (lne): N launches against oe_rowgemm6 + oe_layernorm_bwd_dx.  (GPU box.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openeat_amd import hip, ops
DEV = "cuda"
hip.GEMM_PRECISION = 6
torch.manual_seed(97)
rows, d, act = 7936, 256, int(os.environ.get("LNE_ACT", "2"))
gq = torch.randn(rows, d, device=DEV)
w = torch.nn.Parameter(torch.randn(d, d, device=DEV) / 16)
yc = torch.randn(rows, d, device=DEV) * 1.5 + 0.4
gamma, beta = torch.randn(d, device=DEV) * 0.2 + 1.0, torch.randn(d, device=DEV) * 0.1
stats = torch.stack([yc.mean(1), 1.0 / torch.sqrt(yc.var(1, unbiased=False) + 1e-5)], 1).contiguous()
nws = hip.lib().oe_layernorm_bwd_workspace_floats(rows, d)
with torch.no_grad():
    dz = ops.gemm_nn(gq, w)
    dx0, ws0 = torch.empty_like(yc), torch.zeros(nws, device=DEV)
    hip.call("oe_layernorm_bwd_dx", dz, yc, gamma, beta, act, stats, rows, d, None, None, dx0, ws0)
    worst = 0.0
    for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
        dx1, ws1 = torch.full_like(yc, float("nan")), torch.zeros(nws, device=DEV)
        epi = dict(x=yc, stats=stats, gamma=gamma, beta=beta, act=act, dx=dx1, ws=ws1, done=False)
        ops.gemm_nn(gq, w, ln_epi=epi)
        torch.cuda.synchronize()
        err = float((dx1 - dx0).abs().max())
        worst = max(worst, err)
        if err > 1e-4:
            bad = ((dx1 - dx0).abs() > 1e-4).any(1).nonzero().flatten()
            print("trial", trial, "max err", err, "bad rows", bad[:16].tolist())
    print("worst error over the trials:", worst)
