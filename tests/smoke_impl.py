"""Body of __graft_entry__.smoke(): a tiny hot-path invocation on cuda:0 checked
against the CPU oracle."""
import torch
import torch.nn.functional as F


def run():
    from openeat_amd import hip
    torch.manual_seed(0)
    B, T, V, Lmax = 3, 40, 50, 6
    logits = torch.randn(B, T, V)
    hl = torch.tensor([40, 31, 17], dtype=torch.int32)
    yl = torch.tensor([6, 4, 2], dtype=torch.int32)
    ys = torch.randint(1, V, (B, Lmax), dtype=torch.int32)
    per = F.ctc_loss(logits.transpose(0, 1).log_softmax(2), ys, hl, yl, reduction="none", zero_infinity=True)
    L = hip.lib()
    dev = "cuda:0"
    buf = logits.to(dev).contiguous()
    ws = torch.empty(L.oe_ctc_workspace_floats(B, T, Lmax), device=dev)
    nll = torch.empty(B, device=dev)
    hl_d, ys_d, yl_d = hl.to(dev), ys.to(dev), yl.to(dev)
    hip.check(L.oe_ctc_loss_fused(hip.ptr(buf), V, B, T, V, hip.ptr(hl_d), hip.ptr(ys_d), Lmax,
                                  hip.ptr(yl_d), 1.0, hip.ptr(nll), None, None, hip.ptr(ws), hip.stream()), "ctc")
    torch.cuda.synchronize()
    torch.testing.assert_close(nll.cpu(), per, rtol=1e-4, atol=1e-3)
    print("smoke OK: ctc nll", nll.cpu().tolist())
