"""The entry point that allocates nothing (mmb_loss_fwd_bwd_noalloc): through the C ABI, against the existing
mmb_loss_fwd_bwd (caller-provided workspace) on the same inputs -- bit-identical losses and head gradients -- on two
calls in a row, the second with a smaller input (the scratch keeps the first call's partial sums)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    d = lambda a: a.cuda().contiguous()
    mask = (torch.rand(B, N, generator=g) < 0.4).float()
    mask[0, 0] = 1.0
    return dict(hc=d(torch.randn(B, N, 3, generator=g)), hd=d(torch.randn(B, N, 8, generator=g)),
                t=d(torch.rand(B, generator=g) * 0.9 + 0.05), xt=d(torch.randn(B, N, 3, generator=g)),
                x0=d(torch.randn(B, N, 3, generator=g)), x1=d(torch.randn(B, N, 3, generator=g)),
                k1=d(torch.randint(0, 8, (B, N), generator=g)), maskf=d(mask),
                total=d(mask.sum().to(torch.int32).reshape(1)), coef=d(torch.tensor([0.7, 1.3])),
                freq=d(torch.tensor([0.304, 0.0847, 0.2986, 0.3078, 0.0013, 0.0014, 0.0011, 0.001])))


def _run(noalloc, B, N, x):
    import mmb_hip
    from mmb_hip import ops
    losses = torch.full((2,), float("nan"), device="cuda")
    dc, dd = torch.full_like(x["hc"], float("nan")), torch.full_like(x["hd"], float("nan"))
    p = lambda a: C.c_void_p(a.data_ptr())
    head = (B, N, 3, 8, p(x["hc"]), p(x["hd"]), p(x["t"]), p(x["xt"]), p(x["x0"]), p(x["x1"]), 1, p(x["k1"]),
            p(x["maskf"]), p(x["total"]), p(x["freq"]), p(x["coef"]))
    tail = (p(losses), p(dc), p(dd), ops.stream())
    if noalloc:
        mmb_hip.call("mmb_loss_fwd_bwd_noalloc", *head, *tail)
    else:
        ws = torch.empty(ops.LOSS_WS, device="cuda")
        mmb_hip.call("mmb_loss_fwd_bwd", *head, p(ws), *tail)
    torch.cuda.synchronize()
    return losses.cpu(), dc.cpu(), dd.cpu()


def test_noalloc_matches_workspace_entry_point_twice():
    for B, N, seed in ((4096, 150, 1), (3, 17, 2)):          # 614 400 particles (2048 blocks, the cap), then 51
        x = _inputs(B, N, seed)
        want = _run(False, B, N, x)
        got = _run(True, B, N, x)
        assert torch.isfinite(want[0]).all()
        for a, b in zip(got, want):
            assert torch.equal(a, b)                         # same kernels, same order: bit-identical


def test_noalloc_rejects_oversize_and_bad_arguments():
    import mmb_hip
    x = _inputs(2, 4, 3)
    p = lambda a: C.c_void_p(a.data_ptr())
    losses = torch.zeros(2, device="cuda")
    with pytest.raises(mmb_hip.MMBError, match="beyond the maximum"):
        mmb_hip.call("mmb_loss_fwd_bwd_noalloc", 1 << 16, 1 << 15, 3, 8, p(x["hc"]), p(x["hd"]), p(x["t"]), p(x["xt"]),
                     p(x["x0"]), p(x["x1"]), 0, p(x["k1"]), p(x["maskf"]), p(x["total"]), None, p(x["coef"]),
                     p(losses), None, None, None)
    with pytest.raises(mmb_hip.MMBError):
        mmb_hip.call("mmb_loss_fwd_bwd_noalloc", 2, 4, 3, 8, None, None, None, None, None, None, 0, None,
                     p(x["maskf"]), p(x["total"]), None, None, p(losses), None, None, None)
