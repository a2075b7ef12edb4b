"""developer check: where do idle-tile generations differ from all-tile ones"""
import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "tests"), os.path.join(R, "multimodal-bridges_amd"), R]
import conftest
from test_hip_model import build
from tensorclass import TensorMultiModal
from datamodules.datasets import DataCoupling
name = sys.argv[1]
g = conftest.Golden(name)
model = build(g, num_timesteps=int(sys.argv[2]) if len(sys.argv) > 2 else 6)
model.use_fused = True
N = g.t["mask"].shape[1]
gen = torch.Generator().manual_seed(5)
ns = [1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, N - 1, N]
ns = [min(n, N) for n in ns]
B = len(ns) + 2
mask = torch.zeros(B, N, 1, dtype=torch.int64)
for b, n in enumerate(ns):
    mask[b, :n] = 1
mask[-2, N // 3:N // 3 + 4] = 1
mask[-1, ::37] = 1
ns = ns + [-2, -1]
k0 = torch.randint(0, 8, (B, N, 1), generator=gen) * mask
x0 = torch.randn(B, N, 3, generator=gen) * mask
def run(bits):
    os.environ["MMB_GEN_TILES"] = str(bits)
    model.manual_seed(11)
    src = TensorMultiModal(None, x0.cuda(), k0.cuda(), mask.cuda())
    state = TensorMultiModal(torch.zeros_like(src.mask), src.continuous.clone(), src.discrete.clone(), src.mask)
    p = model.simulate_dynamics(state, DataCoupling(src, TensorMultiModal(None, None, None, src.mask), None))
    return p.continuous[1].clone().cpu(), p.discrete[1].clone().cpu()
xa, ka = run(0)
xb, kb = run(1)
for b, n in enumerate(ns):
    d = (xa[b] - xb[b]).abs().amax(-1)
    bad = torch.nonzero(d > 0).flatten().tolist()
    bits = torch.nonzero((xa[b].view(torch.int32) != xb[b].view(torch.int32)).any(-1)).flatten().tolist()
    print(f"n={n}: {len(bad)} slots differ numerically (max {float(d.max()):.3e}), {len(bits)} bitwise; first {bits[:6]}; k differ {(ka[b]!=kb[b]).sum().item()}")
    if bits:
        i = bits[0]
        print("   ", xa[b, i].tolist(), xb[b, i].tolist())
