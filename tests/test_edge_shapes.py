"""Odd batch / cloud shapes (B = 1: quirk Q3 breaks the reference, the build must handle it; N = 1, odd N, N = 128 with
half the slots padded) through the whole hot path: the fused training step must agree with the layered one and the
generation loop must return finite states."""
import copy

import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["mm_mid", "uni_mid"])
@pytest.mark.parametrize("B,N", [(1, 5), (3, 7), (2, 128), (5, 64), (1, 1)])
def test_odd_shapes(golden, name, B, N):
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    g = golden(name)
    cfg = copy.deepcopy(g.cfg)
    cfg["data"]["target_path"] = GOLDEN
    cfg["model"]["num_timesteps"] = 5
    cfg["encoder"]["dropout"] = 0.1
    model = MultiModalBridgeMatching(ExperimentConfigs(cfg))
    model.load_state_dict(g.params, strict=True)
    model = model.cuda()
    multi = name.startswith("mm")
    gen = torch.Generator().manual_seed(B * 100 + N)
    mask = torch.ones(B, N, 1, dtype=torch.int64)
    if N > 2:
        mask[:, N // 2:] = 0
    x1 = (torch.randn(B, N, 3, generator=gen) * mask).cuda() if multi else None
    k1 = (torch.randint(0, 8, (B, N, 1), generator=gen) * mask).cuda()
    mask = mask.cuda()
    res = {}
    for fused in (True, False):
        model.train()
        model.use_fused_training = model.use_fused_backward = fused
        model.manual_seed(4)
        out = model.training_step(DataCoupling(TensorMultiModal(), TensorMultiModal(None, x1, k1, mask), None), 0)
        model.zero_grad(set_to_none=True)
        out["loss"].backward()
        gn = sum(float((p.grad.double() ** 2).sum()) for p in model.parameters() if p.grad is not None) ** 0.5
        res[fused] = (float(out["loss"].detach()), gn)
    assert abs(res[True][0] - res[False][0]) <= 1e-5 * max(1.0, abs(res[False][0])), res
    assert abs(res[True][1] - res[False][1]) <= 2e-3 * res[False][1] + 1e-7, res
    model.eval()
    model.manual_seed(9)
    paths, _ = model.predict_step(DataCoupling(TensorMultiModal(), TensorMultiModal(None, x1, k1, mask), None), 0)
    assert paths.discrete.shape == (2, B, N, 1) and int(paths.discrete.min()) >= 0 and int(paths.discrete.max()) < 8
    if multi:
        assert torch.isfinite(paths.continuous).all()
