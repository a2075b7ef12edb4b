"""Achieved HBM GB/s of the fused bridge / solver / loss kernels at P >= 1M particles."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-bridges_amd"))
from mmb_hip import ops

B, N, Dc, V = int(os.environ.get("B", 65536)), 128, 3, 8
P = B * N
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x1 = torch.randn(B, N, Dc, device=dev, generator=g)
x0 = torch.randn(B, N, Dc, device=dev, generator=g)
z = torch.randn(B, N, Dc, device=dev, generator=g)
v = torch.randn(B, N, Dc, device=dev, generator=g)
k1 = torch.randint(0, V, (B, N, 1), device=dev, generator=g)
k0 = torch.randint(0, V, (B, N, 1), device=dev, generator=g)
u = torch.rand(B, N, device=dev, generator=g)
tu = torch.rand(B, device=dev, generator=g)
logits = torch.randn(B, N, V, device=dev, generator=g)
hc = torch.randn(B, N, Dc, device=dev, generator=g)
mask = (torch.rand(B, N, 1, device=dev, generator=g) > 0.6).long()
mi = ops.mask_prepare(mask)
coef = torch.tensor([0.5, 0.5], device=dev)
dn = torch.poisson(torch.full((B, N, V), 0.05, device=dev), generator=g)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


t_out = torch.empty(B, device=dev); xt = torch.empty_like(x1); kt = torch.empty_like(k1)
x0o = torch.empty_like(x1); k0o = torch.empty_like(k1)
from mmb_hip import call
from mmb_hip.ops import f32, i64, i32, stream, _fl
cases = {
    "K-A bridge_sample_rng (60 B/particle)": (60, lambda: call("mmb_bridge_sample_rng", B, N, Dc, V, _fl(1e-4), f32(x1), i64(k1), 0, _fl(1e-4), 0, _fl(0.125), 1, 2, f32(t_out), f32(x0o), i64(k0o), f32(xt), i64(kt), stream())),
    "K-A bridge_sample injected (88 B/particle)": (88, lambda: call("mmb_bridge_sample", B, N, Dc, V, f32(tu), _fl(1e-4), f32(x0), f32(x1), f32(z), i64(k0), i64(k1), f32(u), 0, _fl(1e-4), 0, _fl(0.125), f32(t_out), f32(xt), i64(kt), stream())),
    "K-B euler_step (36 B/particle)": (36, lambda: ops.euler_step_(x0, v, 0.01)),
    "K-B euler_maruyama philox (36 B/particle)": (36, lambda: ops.euler_maruyama_step_(x0, v, 0.01, None, 0.5, 0.1, seed=1, offset=2)),
    "K-C rate_tauleap philox (48 B/particle)": (48, lambda: ops.rate_tauleap_(logits, k0, None, 0.5, 0.01, 0, 0.125, seed=1, offset=3)),
    "K-C rate_tauleap injected (80 B/particle)": (80, lambda: ops.rate_tauleap_(logits, k0, None, 0.5, 0.01, 0, 0.125, delta_n=dn)),
    "K-F loss fwd+bwd (72+44 B/particle)": (116, lambda: ops.loss_fwd_bwd(hc, logits, tu, xt, x0, x1, 0, k1, mi, None, coef)),
    "mask_prepare (12 B/particle)": (12, lambda: ops.mask_prepare(mask)),
}
res = {}
for name, (bpp, fn) in cases.items():
    dt = timeit(fn)
    gbs = bpp * P / dt / 1e9
    res[name] = {"us": dt * 1e6, "GB/s": gbs, "frac_of_8TBs": gbs / 8000}
    print(f"{name:48s} {dt*1e6:9.1f} us  {gbs:8.1f} GB/s  {gbs/8000*100:5.1f}% of 8 TB/s")
print(json.dumps({"P": P, "results": res}))
