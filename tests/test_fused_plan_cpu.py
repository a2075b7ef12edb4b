"""CPU: the slot tables of the fused training kernels (encoders/fused_train.py) against the counts the C-ABI entry
points insist on (csrc/fused_epic.hip, fused_epic_bwd.hip: `expect`), for several depths, with and without the skip
connection, for both model families.  No kernel runs: the planner only allocates and records."""
import copy

import pytest
import torch

from conftest import GOLDEN


def _model(golden, name, blocks, skip):
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    cfg = copy.deepcopy(golden(name).cfg)
    cfg["data"]["target_path"] = GOLDEN
    cfg["encoder"]["num_blocks"] = blocks
    cfg["encoder"]["skip_connection"] = skip
    return MultiModalBridgeMatching(ExperimentConfigs(cfg))


@pytest.mark.parametrize("blocks,skip", [([1, 1, 2], False), ([3, 3, 1], True), ([2, 2, 4], True), ([1, 3, 2], True),
                                         ([2, 1, 3], False)])
def test_multi_plan_counts(golden, blocks, skip):
    from encoders import fused_train as ft
    model = _model(golden, "mm_mid", blocks, skip)
    e = model.config.encoder
    B, N = 3, 10
    M = B * N
    pl = ft._Plan(B, N, torch.device("cpu"), [], 0.1, 0)
    temb, cont, disc = torch.empty(B, e.dim_emb_time), torch.empty(M, e.dim_emb_continuous), torch.empty(M, e.dim_emb_discrete)
    ft._plan_forward(model, pl, temb, cont, disc, torch.empty(M, 3), torch.empty(M, 8))
    nb0, nb1 = model.encoder.continuous_encoder.num_blocks, model.encoder.discrete_encoder.num_blocks
    nbf = model.encoder.fused_encoder.num_blocks
    assert (nb0, nb1) == (blocks[0], blocks[1])                    # the two branch depths may differ
    assert nbf == blocks[2] - 1                                   # K2: one constructed layer never runs
    assert len(pl.slots) == 2 * 10 + 8 * (nb0 + nb1) + 1 + 8 * nbf + 1 + 4
    for ptr, ld, n in pl.slots:
        assert ptr % 16 == 0 and ld > 0 and n >= ld
    # dropout offsets are handed out in execution order, two per executed layer
    offs = [en[12] for en in pl.tape if en[0] == "linear" and en[11] > 0.0]
    assert offs == list(range(1, 2 * (nb0 + nb1 + nbf) + 1))
    bp = ft._plan_backward(pl, (torch.empty(M, 4), torch.empty(M, 8)), cont.shape[1])
    assert bp["n"] == 4 + 8 + (1 + 9 * nbf) + 2 * (1 + 9) + 9 * (nb0 + nb1) + 1
    # one weight-gradient job per executed weight-normed Linear
    n_lin = sum(1 for en in pl.tape if en[0] == "linear")
    assert len(bp["jobs"]) == n_lin
    assert len({id(r["layer"]) for r, _, _ in bp["jobs"]}) == n_lin


@pytest.mark.parametrize("blocks,skip", [(1, False), (3, True)])
def test_uni_plan_counts(golden, blocks, skip):
    from encoders import fused_train as ft
    model = _model(golden, "uni_mid", blocks, skip)
    e = model.config.encoder
    B, N = 2, 6
    M = B * N
    pl = ft._Plan(B, N, torch.device("cpu"), [], 0.0, 0)
    ft._plan_forward(model, pl, torch.empty(B, e.dim_emb_time), None, torch.empty(M, e.dim_emb_discrete), None, torch.empty(M, 8))
    nb = model.encoder.epic.num_blocks
    assert pl.kind == "uni" and len(pl.slots) == 9 + 8 * nb
    # the projection's first layer concatenates the time embedding twice (multimodal_epic.py:47-54 + epic.py:155)
    first = next(en for en in pl.tape if en[0] == "linear")
    assert [s[1] for s in first[3]].count(first[3][0][1]) == 2 and len(first[3]) == 3
    bp = ft._plan_backward(pl, (torch.empty(M, 8),), 0)
    assert bp["n"] == 1 + 2 + 9 * nb + 9
    assert len(bp["jobs"]) == sum(1 for en in pl.tape if en[0] == "linear")
