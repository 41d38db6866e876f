import os, sys, torch
sys.path.insert(0, os.getcwd())
from openeat_amd import hip
B, T, d, K = 32, 248, 256, 15
a, dy = torch.randn(B, T, 2 * d, device="cuda"), torch.randn(B, T, d, device="cuda")
w = torch.randn(d, K, device="cuda")
da = torch.empty_like(a)
dw, db = torch.zeros(d, K, device="cuda"), torch.zeros(d, device="cuda")
ws = torch.empty(hip.lib().oe_dwconv_glu_bwd_workspace_floats(B, T, d, K), device="cuda")
for _ in range(20):
    hip.call("oe_dwconv_glu_bwd", a, dy, w, None, B, T, d, K, 0, da, dw, db, None, ws)
torch.cuda.synchronize()
