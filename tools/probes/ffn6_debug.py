"""Debug aid for csrc/ffn6.hip: per 32 x 32 tile error map of y / pre / act against float64 on small shapes."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openeat_amd import hip  # noqa: E402

DEV = "cuda"


def run(rows, d, ff, act=0, p_in=0.0, p_out=0.0):
    torch.manual_seed(1)
    x = torch.randn(rows, d)
    w1, b1 = torch.randn(ff, d) / math.sqrt(d), torch.randn(ff) * 0.1
    w2, b2 = torch.randn(d, ff) / math.sqrt(ff), torch.randn(d) * 0.1
    nb = hip.lib().oe_ffn_packed_bytes(d, ff, 6)
    w1p, w2p = torch.empty(nb, dtype=torch.uint8, device=DEV), torch.empty(nb, dtype=torch.uint8, device=DEV)
    xd, w1d, w2d, b1d, b2d = [t.to(DEV) for t in (x, w1, w2, b1, b2)]
    hip.call("oe_ffn_pack_weights", w1d, w2d, d, ff, 6, w1p, w2p)
    pre = torch.full((rows, ff), float("nan"), device=DEV)
    aout = torch.full((rows, ff), float("nan"), device=DEV)
    y = torch.full((rows, d), float("nan"), device=DEV)
    hip.ffn_fwd(xd, w1p, b1d, w2p, b2d, rows, d, ff, act, drop_in=p_in, seed_in=1, drop_out=p_out, seed_out=2, pre_out=pre, act_out=aout, y=y,
                precision=6)
    torch.cuda.synchronize()
    h = x.double() @ w1.double().t() + b1.double()
    a = h * torch.sigmoid(h) if act == 2 else h.clamp(min=0) if act == 1 else h
    want = a @ w2.double().t() + b2.double()
    for name, got, ref in (("pre", pre, h), ("act", aout, a), ("y", y, want)):
        e = (got.cpu().double() - ref).abs()
        e[torch.isnan(e)] = 1e9
        R, Cc = (rows + 31) // 32, ref.shape[1] // 32
        m = torch.zeros(R, Cc)
        for i in range(R):
            for j in range(Cc):
                m[i, j] = e[32 * i: 32 * i + 32, 32 * j: 32 * j + 32].max()
        bad = (m > 1e-3).nonzero().tolist()
        print(f"rows {rows} d {d} ff {ff} {name}: max err {float(e.max()):.3g}; bad tiles (row tile, col tile): {bad[:24]}{' ...' if len(bad) > 24 else ''} of {R} x {Cc}", flush=True)


if __name__ == "__main__":
    run(64, 256, 128)
    run(64, 256, 256)
    run(64, 256, 1024)
    run(128, 256, 1024)
    run(640, 256, 1024, act=2)
    run(64, 128, 128)
    run(32, 512, 256)
