"""GPU parity of the drop-in classes (same names / signatures as the reference) against the
golden vectors of the imported reference: bridge sampling, embedder, encoder heads, losses,
parameter gradients and the full generation loop under injected noise.
Bar (north_star): jump-state indices bit-exact; continuous features / losses rtol 1e-4 in fp32."""
import copy

import pytest
import torch

from conftest import GOLDEN, MODEL_GOLDENS, EPIC_GOLDENS, TRANSFORMER_GOLDENS, rel_err, grads_close, GRAD_TOL, GRAD_NORM_TOL

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-4, atol=1e-5):
    torch.testing.assert_close(a.detach().cpu().float(), b.detach().cpu().float(), rtol=rtol, atol=atol)


def build(g, **model_updates):
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    cfg = copy.deepcopy(g.cfg)
    cfg["data"]["target_path"] = GOLDEN            # metadata.json (multiplicity pmf) lives there
    cfg["model"].update(model_updates)
    config = ExperimentConfigs(cfg)
    model = MultiModalBridgeMatching(config)
    missing = model.load_state_dict(g.params, strict=True)       # reference parameter names (Appendix B)
    return model.cuda().eval()


def batch_of(g, with_source=True):
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    t = g.t
    c = lambda a: None if a is None else a.cuda()
    tgt = TensorMultiModal(None, c(t.get("x1")), c(t["k1"]), c(t["mask"]))
    src = TensorMultiModal(None, c(t.get("x0")), c(t["k0"]), torch.ones_like(c(t["mask"]))) if with_source else TensorMultiModal()
    return DataCoupling(src, tgt, None)


@pytest.mark.parametrize("name", MODEL_GOLDENS)
def test_sample_bridges_injected(golden, name):
    g = golden(name)
    model = build(g)
    t = g.t
    noise = {"t_u": t["t_u"].cuda(), "u_k": t["u_k"].cuda()}
    if "z" in t:
        noise["z"] = t["z"].cuda()
    st = model.sample_bridges(batch_of(g), noise=noise)
    assert st.time.shape == t["time"].shape and st.time.dim() == 3
    close(st.time, t["time"], rtol=1e-6, atol=1e-7)
    if "xt" in t:
        close(st.continuous, t["xt"], rtol=1e-5, atol=1e-6)
    ok = t["u_k_margin"] > 1e-5
    assert st.discrete.shape == t["kt"].shape and st.discrete.dtype == torch.int64
    assert torch.equal(st.discrete.cpu().squeeze(-1)[ok], t["kt"].squeeze(-1)[ok])


@pytest.mark.parametrize("name", MODEL_GOLDENS)
def test_embedder_encoder_heads(golden, name, margins):
    from tensorclass import TensorMultiModal
    g = golden(name)
    model = build(g)
    t = g.t
    c = lambda a: None if a is None else a.cuda()
    state = TensorMultiModal(c(t["time"]), c(t.get("xt")), c(t["kt"]), c(t["mask"]))
    loc, glob = model.embedder(state, None)
    close(glob.time, t["cap:emb.glob_time"], rtol=1e-4, atol=2e-5)
    close(loc.time, t["cap:emb.time"], rtol=1e-4, atol=2e-5)
    if "cap:emb.continuous" in t:
        close(loc.continuous, t["cap:emb.continuous"])
    assert loc.discrete.dtype == torch.int64                                  # quirk Q1
    assert torch.equal(loc.discrete.cpu(), t["cap:emb.discrete"])
    heads = model(state, None)
    if "head_continuous" in t:
        margins(name, "inference", "head:continuous", rel_err(heads.continuous, t["head_continuous"]))
        close(heads.continuous, t["head_continuous"], rtol=1e-4, atol=2e-5)
    margins(name, "inference", "head:discrete", rel_err(heads.discrete, t["head_discrete"]))
    close(heads.discrete, t["head_discrete"], rtol=1e-4, atol=2e-5)
    # reference-style call with materialised (B,N,*) tensors goes through the generic segment path
    plain = TensorMultiModal(loc.time.clone(), None if loc.continuous is None else loc.continuous.clone(),
                             loc.discrete.clone(), loc.mask)
    heads2 = model.encoder(plain, glob)
    close(heads2.discrete, t["head_discrete"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name", MODEL_GOLDENS)
def test_loss_values(golden, name, margins):
    from tensorclass import TensorMultiModal
    g = golden(name)
    model = build(g)
    t = g.t
    c = lambda a: None if a is None else a.cuda()
    state = TensorMultiModal(c(t["time"]), c(t.get("xt")), c(t["kt"]), c(t["mask"]))
    batch = batch_of(g)
    heads = model(state, batch)
    lc, ld = model.loss_fn(heads, state, batch)
    margins(name, "inference", "loss:continuous", rel_err(lc, t["loss_continuous"]))
    margins(name, "inference", "loss:discrete", rel_err(ld, t["loss_discrete"]))
    close(lc, t["loss_continuous"], rtol=1e-4, atol=1e-6)
    close(ld, t["loss_discrete"], rtol=1e-4, atol=1e-6)
    close(model.loss_multimode([lc, ld]), t["loss"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", MODEL_GOLDENS)
def test_generation_injected(golden, name):
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    g = golden(name)
    t = g.t
    tags = sorted({k.split(":")[0] for k in t if k.startswith("gen")})
    for tag in tags:
        steps = int(tag[3:])
        model = build(g, num_timesteps=steps)
        c = lambda a: None if a is None else a.cuda()
        src = TensorMultiModal(None, c(t.get(tag + ":src_x")), c(t[tag + ":src_k"]), c(t[tag + ":src_mask"]))
        batch = DataCoupling(src, TensorMultiModal(None, None, None, src.mask), None)
        state = TensorMultiModal(torch.zeros_like(src.mask), None if src.continuous is None else src.continuous.clone(),
                                 src.discrete.clone(), src.mask)
        noise = {}
        if tag + ":delta_n" in t:
            noise["delta_n"] = t[tag + ":delta_n"].cuda()
        if tag + ":u_jump" in t:
            noise["u_jump"] = t[tag + ":u_jump"].cuda()
        paths = model.simulate_dynamics(state, batch, noise=noise)
        assert paths.discrete.shape[0] == 2 and paths.time.shape == (2,) + tuple(src.mask.shape)
        assert torch.equal(paths.discrete[0].cpu(), t[tag + ":src_k"])
        if tag + ":final_x" in t:
            close(paths.continuous[1], t[tag + ":final_x"], rtol=1e-4, atol=2e-5)
        agree = (paths.discrete[1].cpu() == t[tag + ":final_k"]).float().mean().item()
        assert agree == 1.0, f"{name}/{tag}: final-k agreement {agree}"


@pytest.mark.parametrize("path", ["fused", "layered"])
@pytest.mark.parametrize("name", MODEL_GOLDENS)
def test_training_step_gradients(golden, name, path, margins):
    """Taped forward + hand-written backward vs the reference's autograd gradients, through the fused per-jet training
    kernels (where the width set has them) and through the layered path."""
    from tensorclass import TensorMultiModal
    g = golden(name)
    model = build(g).train()
    if path == "layered":
        model.use_fused_training = False
    elif not model.fused_training_available():
        pytest.skip("no fused training kernels for this encoder / width set")
    t = g.t
    c = lambda a: None if a is None else a.cuda()
    state = TensorMultiModal(c(t["time"]), c(t.get("xt")), c(t["kt"]), c(t["mask"]))
    batch = batch_of(g)
    out = model.step_from_state(state, batch, "train", True)
    margins(name, path, "loss:total", rel_err(out["loss"], t["loss"]))
    close(out["loss"], t["loss"], rtol=1e-4, atol=1e-6)
    close(out["train_loss_continuous"], t["loss_continuous"], rtol=1e-4, atol=1e-6)
    close(out["train_loss_discrete"], t["loss_discrete"], rtol=1e-4, atol=1e-6)
    model.zero_grad(set_to_none=True)
    out["loss"].backward()
    params = dict(model.named_parameters())
    gn2 = 0.0
    for n, p in params.items():
        if p.grad is not None:
            gn2 += float((p.grad.double() ** 2).sum())
    gn = float(t["grad_norm"])
    margins(name, path, "gradnorm:total", abs(gn2 ** 0.5 - gn) / gn)
    assert abs(gn2 ** 0.5 - gn) <= GRAD_NORM_TOL * gn + 1e-6, (gn2 ** 0.5, gn)
    bad = []
    for n, gref in g.grads().items():
        if n.endswith("time_embedding.0.W"):
            continue
        got = params[n].grad
        if got is None:
            assert float(gref.abs().max()) == 0.0, f"{n}: no gradient produced but the reference has one"
            continue
        e = rel_err(got, gref)
        margins(name, path, "grad:" + n, e)
        if e > GRAD_TOL:
            bad.append((n, e))
    assert not bad, f"{name}/{path}: gradients beyond {GRAD_TOL}: {bad[:5]} ({len(bad)})"
    if g.grad_summaries():          # paper-size fixtures: norm + probe projection of EVERY parameter's gradient
        worst = g.check_grad_summaries(lambda n: params[n].grad, rtol=GRAD_TOL, atol_scale=GRAD_NORM_TOL)
        margins(name, path, "gsum:norm_or_projection", worst)
    # quirks: the truncated embedding (Q1) and the never-executed fused layer (K2) get no gradient
    assert params["embedder.discrete_embedding.weight_v"].grad is None


def test_predict_step_end_to_end(golden):
    """Philox source + in-kernel noise: shapes, dtypes, ranges and determinism."""
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    g = golden("mm_mid")
    model = build(g, num_timesteps=20)
    B, N = 64, 16
    mask = torch.ones(B, N, 1, dtype=torch.int64, device="cuda")
    tgt = TensorMultiModal(None, torch.zeros(B, N, 3, device="cuda"), torch.zeros(B, N, 1, dtype=torch.int64, device="cuda"), mask)
    outs = []
    for _ in range(2):
        model.manual_seed(123)
        paths, target = model.predict_step(DataCoupling(TensorMultiModal(), tgt, None), 0)
        outs.append(paths)
    p = outs[0]
    assert p.continuous.shape == (2, B, N, 3) and p.discrete.shape == (2, B, N, 1) and p.discrete.dtype == torch.int64
    assert not p.continuous.is_cuda
    assert int(p.discrete.min()) >= 0 and int(p.discrete.max()) < 8
    assert torch.isfinite(p.continuous).all()
    assert torch.equal(outs[0].discrete, outs[1].discrete) and torch.equal(outs[0].continuous, outs[1].continuous)


@pytest.mark.parametrize("name", ["mm_mid", "S3", "S3_n150"])
def test_fused_matches_layered(golden, name):
    """The per-jet fused kernels (registers-resident EPiC) against the layered HIP path and the goldens."""
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    from mmb_hip import fused
    g = golden(name)
    model = build(g)
    assert fused.eligible(model) is not None
    t = g.t
    c = lambda a: None if a is None else a.cuda()
    state = TensorMultiModal(c(t["time"]), c(t["xt"]), c(t["kt"]), c(t["mask"]))
    model.use_fused = True
    hf = model(state, None)
    model.use_fused = False
    hl = model(state, None)
    close(hf.continuous, t["head_continuous"], rtol=1e-4, atol=2e-5)
    close(hf.discrete, t["head_discrete"], rtol=1e-4, atol=2e-5)
    close(hf.continuous, hl.continuous, rtol=1e-4, atol=2e-5)
    close(hf.discrete, hl.discrete, rtol=1e-4, atol=2e-5)
    # single-head variants
    model.use_fused = True
    close(model.forward_heads(state, None, which="continuous").continuous, t["head_continuous"], rtol=1e-4, atol=2e-5)
    close(model.forward_heads(state, None, which="discrete").discrete, t["head_discrete"], rtol=1e-4, atol=2e-5)
    # Philox generation: fused and layered paths must be statistically alike and each deterministic
    B, N = 256, t["mask"].shape[1]
    mask = torch.ones(B, N, 1, dtype=torch.int64, device="cuda")
    tgt = TensorMultiModal(None, torch.zeros(B, N, 3, device="cuda"), torch.zeros(B, N, 1, dtype=torch.int64, device="cuda"), mask)
    model.config.model.num_timesteps = 12
    outs = {}
    for fz in (True, False):
        model.use_fused = fz
        model.manual_seed(7)
        p1, _ = model.predict_step(DataCoupling(TensorMultiModal(), tgt, None), 0)
        model.manual_seed(7)
        p2, _ = model.predict_step(DataCoupling(TensorMultiModal(), tgt, None), 0)
        assert torch.equal(p1.discrete, p2.discrete) and torch.equal(p1.continuous, p2.continuous)
        outs[fz] = p1
    # same source (same seed) -> identical x trajectories up to fp32 noise wherever k paths agree;
    # compare distributions of the final state
    a, b = outs[True], outs[False]
    assert torch.equal(a.mask[0], b.mask[0])
    ha = torch.bincount(a.discrete[1].reshape(-1), minlength=8).float() / a.discrete[1].numel()
    hb = torch.bincount(b.discrete[1].reshape(-1), minlength=8).float() / b.discrete[1].numel()
    assert (ha - hb).abs().max() < 0.05
    assert (a.continuous[1].mean() - b.continuous[1].mean()).abs() < 0.05
    assert (a.continuous[1].std() - b.continuous[1].std()).abs() < 0.05 * float(b.continuous[1].std()) + 0.02


@pytest.mark.parametrize("fused", [True, False])
def test_euler_maruyama_generation_vs_oracle(golden, fused):
    """BASELINE config 5's solver pair (Euler-Maruyama + tau-leap, SchrodingerBridge).  The reference
    crashes here at HEAD, so this is pinned against the Q5-REPAIRED oracle only -- a
    documented deviation, not reference parity: x += dt v + sigma sqrt(t(1-t)) dW, no sqrt(dt)."""
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    from oracle import model as OM
    g = golden("mm_mid")
    steps = 10
    upd = dict(bridge_continuous="SchrodingerBridge", sigma=0.05, solver_continuous="euler_maruyama", num_timesteps=steps)
    model = build(g, **upd)
    model.use_fused = fused
    t = g.t
    c = lambda a: a.cuda()
    src = TensorMultiModal(None, c(t["gen10:src_x"]), c(t["gen10:src_k"]), c(t["gen10:src_mask"]))
    state = TensorMultiModal(torch.zeros_like(src.mask), src.continuous.clone(), src.discrete.clone(), src.mask)
    batch = DataCoupling(src, TensorMultiModal(None, None, None, src.mask), None)
    B, N = src.mask.shape[:2]
    gen = torch.Generator().manual_seed(31)
    dw = torch.randn(steps, B, N, 3, generator=gen)
    dn = torch.poisson(torch.full((steps, B, N, 8), 0.08), generator=gen)
    paths = model.simulate_dynamics(state, batch, noise={"delta_n": c(dn), "delta_w": c(dw)})
    want = OM.simulate_dynamics(g.params, dict(g.flat, **upd),
                                {"continuous": t["gen10:src_x"], "discrete": t["gen10:src_k"], "mask": t["gen10:src_mask"]},
                                {"delta_n": dn, "delta_w": dw})
    close(paths.continuous[1], want["continuous"], rtol=1e-4, atol=2e-5)
    assert torch.equal(paths.discrete[1].cpu(), want["discrete"])
    # the noise term really is in play: the plain-Euler result differs
    model2 = build(g, num_timesteps=steps)
    state2 = TensorMultiModal(torch.zeros_like(src.mask), src.continuous.clone(), src.discrete.clone(), src.mask)
    p2 = model2.simulate_dynamics(state2, batch, noise={"delta_n": c(dn)})
    assert (p2.continuous[1] - paths.continuous[1]).abs().max() > 1e-3


@pytest.mark.parametrize("name", ["uni_mid", "S2", "S2_n150"])
def test_fused_uni_matches_layered(golden, name):
    """Discrete-only UniModalEPiC (BASELINE configs[1]) through the per-jet fused kernels: logits against the
    golden and the layered path; Philox generation deterministic and statistically like the layered one."""
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    from mmb_hip import fused
    g = golden(name)
    model = build(g)
    assert fused.eligible_uni(model) is not None
    t = g.t
    c = lambda a: None if a is None else a.cuda()
    state = TensorMultiModal(c(t["time"]), None, c(t["kt"]), c(t["mask"]))
    model.use_fused = True
    hf = model(state, None)
    model.use_fused = False
    hl = model(state, None)
    assert hf.continuous is None
    close(hf.discrete, t["head_discrete"], rtol=1e-4, atol=2e-5)
    close(hf.discrete, hl.discrete, rtol=1e-4, atol=2e-5)
    B, N = 256, t["mask"].shape[1]
    mask = torch.ones(B, N, 1, dtype=torch.int64, device="cuda")
    tgt = TensorMultiModal(None, None, torch.zeros(B, N, 1, dtype=torch.int64, device="cuda"), mask)
    model.config.model.num_timesteps = 12
    outs = {}
    for fz in (True, False):
        model.use_fused = fz
        model.manual_seed(7)
        p1, _ = model.predict_step(DataCoupling(TensorMultiModal(), tgt, None), 0)
        model.manual_seed(7)
        p2, _ = model.predict_step(DataCoupling(TensorMultiModal(), tgt, None), 0)
        assert torch.equal(p1.discrete, p2.discrete)
        assert int(p1.discrete.min()) >= 0 and int(p1.discrete.max()) < 8
        outs[fz] = p1
    a, b = outs[True], outs[False]
    assert torch.equal(a.mask[0], b.mask[0]) and torch.equal(a.discrete[0], b.discrete[0])
    m = a.mask[0] > 0
    ha = torch.bincount(a.discrete[1][m].flatten(), minlength=8).float()
    hb = torch.bincount(b.discrete[1][m].flatten(), minlength=8).float()
    assert (ha / ha.sum() - hb / hb.sum()).abs().max() < 0.03


@pytest.mark.parametrize("name", ["mm_mid", "S3", "uni_mid", "S2", "S3_n150", "S2_n150"])
@pytest.mark.parametrize("drop", [0.0, 0.2])
def test_fused_training_matches_layered(golden, name, drop):
    """Training step through the fused per-jet kernels (ONE forward launch that saves the tape, ONE backward launch
    that leaves every layer's dZ) against the layered taped forward + layered reverse pass: same Philox dropout
    masks -> same activations, heads, loss and parameter gradients."""
    from tensorclass import TensorMultiModal
    g = golden(name)
    model = build(g).train()
    model.config.encoder.dropout = drop
    t = g.t
    c = lambda a: None if a is None else a.cuda()
    res = {}
    # fused: persistent workspace + multi-problem launches; fused_step: per-step buffers, fused reverse pass;
    # fused_forward: fused forward, layered reverse pass
    for mode in ("fused", "fused_step", "fused_forward", "layered"):
        model.use_fused_training = mode != "layered"
        model.use_fused_backward = False
        model.manual_seed(11)
        state = TensorMultiModal(c(t["time"]), c(t.get("xt")), c(t["kt"]), c(t["mask"]))
        model._tape = []
        heads = model.forward(state, None)
        tape, model._tape = model._tape, None
        model.use_fused_backward = mode == "fused"
        model.manual_seed(11)
        out = model.step_from_state(state, batch_of(g), "train", True)
        model.use_fused_backward = mode in ("fused", "fused_step")
        model.zero_grad(set_to_none=True)
        out["loss"].backward()
        res[mode] = (heads, out["loss"].detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()
                                                            if p.grad is not None}, tape)
    hl, ll, gl, tl = res["layered"]
    lin_l = [e for e in tl if e[0] == "linear"]
    # Rows of padded slots: the fused kernels leave tiles without a real particle idle (their rows of the tape stay
    # unwritten, their heads are zero) -- the masked losses and the exactly-zero gradient rows never look at them
    real = (t["mask"].reshape(-1) > 0).cuda()
    rows = lambda a: a[real] if a.shape[0] == real.numel() else a
    for mode in ("fused", "fused_step", "fused_forward"):
        hf, lf, gf, tf = res[mode]
        assert any(e[0] == "fused_plan" for e in tf), "the fused training forward did not run"
        lin_f = [e for e in tf if e[0] == "linear"]
        assert len(lin_f) == len(lin_l)
        for ef, el in zip(lin_f, lin_l):               # every activation of the tape (same order in both tapes)
            assert ef[1] is el[1] and ef[4] == el[4] and ef[6] == el[6] and ef[11] == el[11] and ef[12] == el[12]
            for i in (8, 9, 10):
                assert (ef[i] is None) == (el[i] is None), (ef[1], i)
                if ef[i] is not None:
                    close(rows(ef[i]), rows(el[i]), rtol=1e-4, atol=2e-5)
        m3 = c(t["mask"]) > 0
        if hl.continuous is not None:
            close(hf.continuous * m3, hl.continuous * m3, rtol=1e-4, atol=2e-5)
        close(hf.discrete * m3, hl.discrete * m3, rtol=1e-4, atol=2e-5)
        close(lf, ll, rtol=1e-5, atol=1e-6)
        grads_close(gf, gl, name, f"{mode}_vs_layered", what=f"drop {drop}: ")
    if drop == 0.0:
        close(res["fused"][0].discrete.cpu() * (t["mask"] > 0), t["head_discrete"] * (t["mask"] > 0), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name", ["mm_mid", "uni_mid"])
def test_fused_repack_tracks_parameter_updates(golden, name):
    """After an optimizer step the packed buffer is refreshed through the gather map (one cat + index_select);
    it must equal a from-scratch packing."""
    from mmb_hip import fused
    g = golden(name)
    model = build(g)
    p0 = model._fused_pack()
    assert "gather" in p0
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01 * torch.randn_like(p))
    p1 = model._fused_pack()
    assert p1 is p0                                       # refreshed in place
    fresh = fused.pack_model(model)
    assert torch.equal(p1["packed"], fresh["packed"])
    assert p1["time_parts"] is None                       # dropped by the repack, rebuilt on demand
    for a, b in zip(fused.time_parts(p1), fused.time_parts(fresh)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name,blocks,skip", [("mm_mid", [1, 1, 2], False), ("mm_mid", [3, 3, 1], True),
                                              ("uni_mid", 1, False), ("uni_mid", 3, True)])
def test_fused_training_other_depths(golden, name, blocks, skip):
    """Fused training step against the layered one on freshly initialised models of other depths and without the
    skip connection (the slot tables of both kernels change with either)."""
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    from tensorclass import TensorMultiModal
    g = golden(name)
    cfg = copy.deepcopy(g.cfg)
    cfg["data"]["target_path"] = GOLDEN
    cfg["encoder"]["num_blocks"] = blocks
    cfg["encoder"]["skip_connection"] = skip
    cfg["encoder"]["dropout"] = 0.1
    torch.manual_seed(17)
    model = MultiModalBridgeMatching(ExperimentConfigs(cfg)).cuda().train()
    t = g.t
    c = lambda a: None if a is None else a.cuda()
    res = {}
    for fused in (True, False):
        model.use_fused_training = model.use_fused_backward = fused
        model.manual_seed(3)
        state = TensorMultiModal(c(t["time"]), c(t.get("xt")), c(t["kt"]), c(t["mask"]))
        out = model.step_from_state(state, batch_of(g), "train", True)
        model.zero_grad(set_to_none=True)
        out["loss"].backward()
        res[fused] = (out["loss"].detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()
                                                     if p.grad is not None})
    assert model._train_ws is not None, "the fused training step did not run"
    close(res[True][0], res[False][0], rtol=1e-5, atol=1e-6)
    grads_close(res[True][1], res[False][1], name, "fused_vs_layered", what=f"blocks {blocks} skip {skip}: ")


@pytest.mark.parametrize("name", ["mm_mid", "uni_mid"])
def test_generation_after_parameter_update(golden, name):
    """The packed buffers are refreshed IN PLACE after an optimizer step; everything cached beside them (the per-grid
    time-embedding / time-vector tables of the generation loop) must be rebuilt too: generating, updating the
    parameters and generating again must equal a freshly built model with the updated parameters."""
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    g = golden(name)
    model = build(g, num_timesteps=6)
    multi = name.startswith("mm")
    B, N = 32, 16
    mask = torch.ones(B, N, 1, dtype=torch.int64, device="cuda")
    tgt = TensorMultiModal(None, torch.zeros(B, N, 3, device="cuda") if multi else None,
                           torch.zeros(B, N, 1, dtype=torch.int64, device="cuda"), mask)
    model.manual_seed(5)
    model.predict_step(DataCoupling(TensorMultiModal(), tgt, None), 0)            # fills the caches
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=gen).cuda())
    model.manual_seed(5)
    got, _ = model.predict_step(DataCoupling(TensorMultiModal(), tgt, None), 0)
    fresh = build(g, num_timesteps=6)
    fresh.load_state_dict(model.state_dict())
    fresh.manual_seed(5)
    want, _ = fresh.predict_step(DataCoupling(TensorMultiModal(), tgt, None), 0)
    assert torch.equal(got.discrete, want.discrete)
    if multi:
        assert torch.equal(got.continuous, want.continuous)


@pytest.mark.parametrize("name", MODEL_GOLDENS)
def test_fused_forward_is_fp32_accurate(golden, name):
    """The fused kernels form every fp32 product from six bf16 partial products of an exact three-plane split
    (DESIGN §2).  Measured against the oracle evaluated in FLOAT64 on the same parameters and inputs, their error
    must be of the size of the fp32-MFMA layered kernels' own (both are fp32 results with different summation
    orders) -- a bf16 or truncated-plane GEMM would be 100-1000x off this bar."""
    from tensorclass import TensorMultiModal
    from oracle import encoder as OE
    g = golden(name)
    model = build(g)
    if model._fused_pack() is None:
        pytest.skip("architecture not served by the fused kernels")
    t = g.t
    c = lambda a: None if a is None else a.cuda()
    state = TensorMultiModal(c(t["time"]), c(t.get("xt")), c(t["kt"]), c(t["mask"]))
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in g.params.items()}
    d = lambda a: None if a is None else a.double()
    ref_c, ref_d = OE.forward(p64, d(t["time"]), d(t.get("xt")), t["kt"], t["mask"],
                              encoder_name=g.cfg["encoder"]["name"], skip=g.cfg["encoder"].get("skip_connection", True))
    errs = {}
    for fused in (True, False):
        model.use_fused = fused
        h = model(state, None)
        e = 0.0
        for got, ref in ((h.continuous, ref_c), (h.discrete, ref_d)):
            if ref is not None:
                scale = ref.abs().max().item()
                e = max(e, ((got.detach().cpu().double() - ref).abs().max().item()) / scale)
        errs[fused] = e
    # the reference's own fp32 CPU goldens sit 1.5e-6 .. 4.5e-6 from the float64 evaluation (fp32 roundings over
    # ~20 layers); bf16 operands would give ~1e-3
    print(name, "max error / max |head| vs float64: fused %.3g, layered %.3g" % (errs[True], errs[False]))
    assert errs[True] < 1e-5, errs
    assert errs[True] <= 3 * errs[False] + 5e-7, errs


# ---------------------------------------------------------------------------------------------------------------
# round 2: the production instantiations of the headline kernels, the remaining thermostat, sub-encoder captures,
# gradient accumulation
def _gen_inputs(g, tag):
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    t = g.t
    c = lambda a: None if a is None else a.cuda()
    src = TensorMultiModal(None, c(t.get(tag + ":src_x")), c(t[tag + ":src_k"]), c(t[tag + ":src_mask"]))
    state = TensorMultiModal(torch.zeros_like(src.mask), None if src.continuous is None else src.continuous.clone(),
                             src.discrete.clone(), src.mask)
    batch = DataCoupling(src, TensorMultiModal(None, None, None, src.mask), None)
    cpu = {"continuous": t.get(tag + ":src_x"), "discrete": t[tag + ":src_k"], "mask": t[tag + ":src_mask"]}
    return state, batch, cpu


@pytest.mark.parametrize("steps", [100, 200])
def test_s5_instantiation_em_tauleap_at_s3_widths(golden, steps):
    """BASELINE configs[4] runs fused_generate_kernel<DimsS3> with em = 1 (Euler-Maruyama + SchrodingerBridge +
    tau-leap) for 200 grid points: exactly that instantiation, at 100 and 200 steps, with injected dW and dn, against
    the Q5-repaired oracle (the reference crashes here, SURVEY Q5: pinned to the oracle only, a documented deviation)."""
    from oracle import model as OM
    from mmb_hip import fused
    g = golden("S3")
    upd = dict(bridge_continuous="SchrodingerBridge", sigma=1e-4, solver_continuous="euler_maruyama", num_timesteps=steps)
    model = build(g, **upd)
    assert fused.eligible(model) == 0, "not the S3-width fused variant"
    state, batch, cpu = _gen_inputs(g, "gen5")
    B, N = state.mask.shape[:2]
    gen = torch.Generator().manual_seed(41 + steps)
    dw = torch.randn(steps, B, N, 3, generator=gen)
    dn = torch.poisson(torch.full((steps, B, N, 8), 0.04), generator=gen)
    import mmb_hip
    mmb_hip.PROFILE = []
    paths = model.simulate_dynamics(state, batch, noise={"delta_n": dn.cuda(), "delta_w": dw.cuda()})
    torch.cuda.synchronize()
    names = [r[0] for r in mmb_hip.PROFILE]
    mmb_hip.PROFILE = None
    assert names.count("mmb_fused_epic_generate") == 1, names           # ONE launch, the fused kernel
    with torch.no_grad():
        want = OM.simulate_dynamics(g.params, dict(g.flat, **upd), cpu, {"delta_n": dn, "delta_w": dw})
    close(paths.continuous[1], want["continuous"], rtol=1e-4, atol=2e-5)
    assert torch.equal(paths.discrete[1].cpu(), want["discrete"])


def test_em_philox_noise_moments(golden):
    """The in-kernel Philox + Box-Muller branch of the fused kernel (delta_w == NULL): with the drift head zeroed
    x(T) - x(0) = sum_i sigma sqrt(t_i (1 - t_i)) dW_i exactly, so its mean is 0, its variance sigma^2 sum t(1-t),
    it is Gaussian (kurtosis 3), and the draws differ across jets, particles, coordinates and calls."""
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    g = golden("S3")
    steps, sigma = 50, 0.3
    model = build(g, bridge_continuous="SchrodingerBridge", sigma=sigma, solver_continuous="euler_maruyama",
                  num_timesteps=steps)
    with torch.no_grad():
        model.encoder.continuous_head["2"].weight_g.zero_()
        model.encoder.continuous_head["2"].bias.zero_()
    B, N = 256, 128
    gen = torch.Generator().manual_seed(3)
    mask = (torch.rand(B, N, 1, generator=gen) < 0.4).long().cuda()
    x0 = torch.randn(B, N, 3, generator=gen).cuda()
    k0 = torch.randint(0, 8, (B, N, 1), generator=gen).cuda()
    outs = []
    for call in range(2):
        src = TensorMultiModal(None, x0.clone(), k0.clone(), mask)
        state = TensorMultiModal(torch.zeros_like(mask), x0.clone(), k0.clone(), mask)
        paths = model.simulate_dynamics(state, DataCoupling(src, TensorMultiModal(None, None, None, mask), None))
        outs.append((paths.continuous[1] - x0).double().cpu())
    d = outs[0]
    grid = torch.linspace(1e-4, 1 - 1e-4, steps).double()
    var = sigma ** 2 * float((grid * (1 - grid)).sum())
    n = d.numel()
    assert abs(float(d.mean())) < 5 * (var / n) ** 0.5
    assert abs(float(d.var()) / var - 1.0) < 0.02                      # 98 304 samples: sd of the ratio ~ 0.45 %
    assert abs(float((d ** 4).mean()) / var ** 2 - 3.0) < 0.15
    flat = d.reshape(B, -1)
    cj = float((flat[0::2] * flat[1::2]).mean()) / var                 # neighbouring jets
    cp = float((d[:, 0::2] * d[:, 1::2]).mean()) / var                 # neighbouring particles
    cc = float((d[..., 0] * d[..., 1]).mean()) / var                   # coordinates
    assert max(abs(cj), abs(cp), abs(cc)) < 0.03, (cj, cp, cc)
    assert float((outs[0] - outs[1]).abs().max()) > 0.1                # a new call draws new noise
    cs = float((outs[0] * outs[1]).mean()) / var
    assert abs(cs) < 0.03


@pytest.mark.parametrize("fused", [True, False])
def test_inverse_thermostat_generation(golden, fused):
    """InverseThermostat (kind 1: w = exp(-V gamma log(t1/t0))) through the generation loop, fused and layered,
    against the oracle (whose w_ts is pinned by the reference's thermostat golden) under injected dn."""
    from oracle import model as OM
    g = golden("mm_mid")
    upd = dict(thermostat_fn="InverseThermostat", num_timesteps=10)
    model = build(g, **upd)
    model.use_fused = fused
    state, batch, cpu = _gen_inputs(g, "gen10")
    dn = g.t["gen10:delta_n"]
    paths = model.simulate_dynamics(state, batch, noise={"delta_n": dn.cuda()})
    with torch.no_grad():
        want = OM.simulate_dynamics(g.params, dict(g.flat, **upd), cpu, {"delta_n": dn})
    close(paths.continuous[1], want["continuous"], rtol=1e-4, atol=2e-5)
    assert torch.equal(paths.discrete[1].cpu(), want["discrete"])


@pytest.mark.parametrize("name", ["mm_tiny", "mm_tiny_masked", "mm_mid", "mm_odd", "S3"] + TRANSFORMER_GOLDENS)
def test_sub_encoder_captures(golden, name):
    """The reference's intermediate outputs (forward hooks on its sub-encoders) against the HIP modules of the same
    names called the reference's way, with materialised (B,N,*) tensors."""
    g = golden(name)
    model = build(g)
    t = g.t
    c = lambda a: a.cuda()
    lt, lc, ld, gt, mask = c(t["cap:emb.time"]), c(t["cap:emb.continuous"]), c(t["cap:emb.discrete"]), \
        c(t["cap:emb.glob_time"]), c(t["mask"])
    enc = model.encoder
    if name in TRANSFORMER_GOLDENS:
        h1 = enc.continuous_encoder(lt, lc, mask)
        close(h1, t["cap:continuous_encoder"], rtol=1e-4, atol=2e-5)
        h2 = enc.discrete_encoder(lt, ld, mask)
        close(h2, t["cap:discrete_encoder"], rtol=1e-4, atol=2e-5)
        if "cap:fused_encoder" in t:
            f = enc.fused_encoder(lt, torch.cat([h1, h2], -1), mask)
            close(f, t["cap:fused_encoder"], rtol=1e-4, atol=2e-5)
        return
    h1, g1 = enc.continuous_encoder(lt, lc, gt, mask)
    close(h1, t["cap:continuous_encoder.0"], rtol=1e-4, atol=2e-5)
    close(g1, t["cap:continuous_encoder.1"], rtol=1e-4, atol=2e-5)
    h2, g2 = enc.discrete_encoder(lt, ld, gt, mask)
    close(h2, t["cap:discrete_encoder.0"], rtol=1e-4, atol=2e-5)
    pl, pg = enc.continuous_encoder.epic_proj(lt, lc, gt, mask)
    close(pl, t["cap:continuous_encoder.epic_proj.0"], rtol=1e-4, atol=2e-5)
    close(pg, t["cap:continuous_encoder.epic_proj.1"], rtol=1e-4, atol=2e-5)
    f, gf = enc.fused_encoder(lt, torch.cat([h1, h2], -1), torch.cat([g1, g2], -1), mask)
    close(f, t["cap:fused_encoder.0"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name,fused", [("mm_mid", True), ("mm_mid", False), ("uni_mid", True), ("tr_tiny", False)])
def test_gradients_accumulate_across_backward_calls(golden, name, fused):
    """two training_step(...)["loss"].backward() without zeroing must leave the SUM of the two steps'
    gradients in .grad (autograd semantics; micro-batch accumulation), on the fused and the layered path."""
    from tensorclass import TensorMultiModal
    g = golden(name)
    model = build(g).train()
    model.use_fused_training = model.use_fused_backward = fused
    t = g.t
    c = lambda a: None if a is None else a.cuda()

    def step(scale):
        state = TensorMultiModal(c(t["time"]), c(t.get("xt")), c(t["kt"]), c(t["mask"]))
        out = model.step_from_state(state, batch_of(g), "train", True)
        (out["loss"] * scale).backward()

    model.zero_grad(set_to_none=True)
    step(1.0)
    g1 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    step(0.5)
    g2 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    step(1.0)
    step(0.5)
    assert g1.keys() == g2.keys()
    for n, p in model.named_parameters():
        if n in g1:
            want = g1[n] + g2[n]
            close(p.grad, want, rtol=1e-4, atol=1e-5 * float(want.abs().max()) + 1e-8)
            assert float(g2[n].abs().max()) > 0 or float(g1[n].abs().max()) == 0


@pytest.mark.parametrize("name,entry", [("S3_n150", "mmb_fused_epic_generate"), ("S2_n150", "mmb_fused_uni_generate")])
def test_n150_generation_runs_the_fused_five_wave_kernels(golden, name, entry):
    """AOJ-sized clouds (128 < N <= 160) stay on the single-launch generation path (five waves per jet,
    csrc/fused_epic_w5.hip): one fused launch per generation and none of the layered path's per-step kernels."""
    import mmb_hip
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    g = golden(name)
    model = build(g, num_timesteps=7)
    B, N = 6, 150
    tgt = TensorMultiModal(None, None if name.startswith("S2") else torch.zeros(B, N, 3, device="cuda"),
                           torch.zeros(B, N, 1, dtype=torch.int64, device="cuda"),
                           torch.ones(B, N, 1, dtype=torch.int64, device="cuda"))
    model.manual_seed(3)
    mmb_hip.PROFILE = []
    try:
        paths, _ = model.predict_step(DataCoupling(TensorMultiModal(), tgt, None), 0, to_cpu=False)
        torch.cuda.synchronize()
        names = [r[0] for r in mmb_hip.PROFILE]
    finally:
        mmb_hip.PROFILE = None
    # (mmb_linear_cat appears a few times: the per-grid-point time-vector table is built with it, once per weight version)
    layered = {"mmb_meansum_pool", "mmb_euler_step", "mmb_rate_tauleap"} & set(names)
    assert names.count(entry) == 1 and not layered, (layered, names.count(entry))
    assert int(paths.discrete[1].min()) >= 0 and int(paths.discrete[1].max()) < 8
    if paths.continuous is not None:
        assert torch.isfinite(paths.continuous[1]).all()
    assert model.fused_training_available()                  # the training kernels have the five-wave build too


@pytest.mark.parametrize("name", ["mm_mid", "S3", "S3_n150", "uni_mid", "S2", "S2_n150"])
def test_idle_tiles_leave_generation_bit_identical(golden, name, monkeypatch):
    """Tiles with nothing but padded slots stay idle in the generation kernels and take drift and logits from one
    representative padded slot (csrc/fused_epic.hip: TilePlan).  Every slot of the result -- padded ones move too,
    K3 -- must be bit-identical to the kernels that compute all tiles, whatever the multiplicity: tile edges, no padded
    slot next to a real one, full jets, holes in the mask."""
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    g = golden(name)
    model = build(g, num_timesteps=6)
    model.use_fused = True
    N = g.t["mask"].shape[1]
    uni = "xt" not in g.t
    gen = torch.Generator().manual_seed(5)
    ns = [min(n, N) for n in (1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, N - 1, N)] + torch.randint(1, N + 1, (51,), generator=gen).tolist()
    B = len(ns) + 2
    mask = torch.zeros(B, N, 1, dtype=torch.int64)
    for b, n in enumerate(ns):
        mask[b, :n] = 1
    mask[-2, N // 3:N // 3 + 4] = 1          # not a prefix: real particles in the second tile only (four-tile clouds)
    mask[-1, ::37] = 1                       # one real particle in every tile
    k0 = torch.randint(0, 8, (B, N, 1), generator=gen) * mask
    x0 = None if uni else torch.randn(B, N, 3, generator=gen) * mask

    def run(bits):
        monkeypatch.setenv("MMB_GEN_TILES", str(bits))
        model.manual_seed(11)
        src = TensorMultiModal(None, None if uni else x0.cuda(), k0.cuda(), mask.cuda())
        state = TensorMultiModal(torch.zeros_like(src.mask), None if uni else src.continuous.clone(), src.discrete.clone(),
                                 src.mask)
        paths = model.simulate_dynamics(state, DataCoupling(src, TensorMultiModal(None, None, None, src.mask), None))
        return (None if uni else paths.continuous[1].clone()), paths.discrete[1].clone()

    x_all, k_all = run(0)
    if N > 128:
        # 128 < N <= 160: jets of fewer than 128 particles run on four waves with shadow slots, the others in the
        # five-wave kernel on a side stream -- with the list of long jets from the caller, and without it
        monkeypatch.setenv("MMB_GEN_LONG_UNKNOWN", "1")
        xs, ks = run(3)
        monkeypatch.delenv("MMB_GEN_LONG_UNKNOWN")
        x3, k3 = run(3)
        assert torch.equal(ks, k3) and (uni or torch.equal(xs, x3))
        monkeypatch.setenv("MMB_GEN_SPLIT", "0")             # every jet in the five-wave kernel
        x5, k5 = run(3)
        monkeypatch.delenv("MMB_GEN_SPLIT")
        assert torch.equal(k5, k3) and (uni or torch.equal(x5, x3))
    prefix = torch.arange(B) < len(ns)                      # a prefix mask keeps its slot order: nothing may move at all
    m = mask.cuda() > 0
    for bits in (1, 3):
        xs, ks = run(bits)
        assert torch.equal(ks[prefix], k_all[prefix]), f"{name}: k differs with MMB_GEN_TILES={bits}"
        assert (ks == k_all).float().mean() > 0.999
        if not uni:
            assert torch.equal(xs[prefix], x_all[prefix]), f"{name}: x differs with MMB_GEN_TILES={bits}"
            close(xs, x_all, rtol=1e-4, atol=2e-5)           # reordered slots: pooled sums in another order
            assert (x_all[~m.expand_as(x_all)] != 0).any()  # padded slots moved (K3)


@pytest.mark.parametrize("chunk", [0, 1, 256])
def test_jet_schedule_is_a_permutation_heaviest_with_lightest(chunk, monkeypatch):
    """mmb_jet_schedule: the launch order of a generation call -- every jet once; rounds of `chunk` jets alternate
    between the heavy and the light end of the jets sorted by particle count (chunk 0: plain longest first)."""
    from mmb_hip import ops, fused
    monkeypatch.setenv("MMB_GEN_PAIR", str(chunk))
    gen = torch.Generator().manual_seed(3)
    for B in (1, 7, 511, 1024, 5000):
        mask = (torch.rand(B, 150, 1, generator=gen) < torch.rand(B, 1, 1, generator=gen)).long().cuda()
        mi = ops.mask_prepare(mask)
        order = fused.jet_schedule(mi).long()
        assert torch.equal(torch.sort(order).values, torch.arange(B, device="cuda"))
        c = mi.counts[order].cpu()
        srt = torch.sort(mi.counts.cpu(), descending=True).values
        if chunk == 0:
            assert torch.equal(c, srt)
        else:
            cp = (B // (2 * chunk)) * chunk
            for q in range(cp // chunk):
                assert torch.equal(c[2 * q * chunk:(2 * q + 1) * chunk], srt[q * chunk:(q + 1) * chunk])
                assert torch.equal(c[(2 * q + 1) * chunk:(2 * q + 2) * chunk], srt.flip(0)[q * chunk:(q + 1) * chunk])
            assert torch.equal(c[2 * cp:], srt[cp:B - cp])
        assert fused.jet_schedule(mi) is mi._order               # computed once per mask


def test_mask_rows_lists_the_real_particles_in_order():
    """mmb_mask_rows: the flat rows of the real particles, ascending, whatever slots the masks mark (N not a multiple of
    64, empty and full jets included)."""
    from mmb_hip import ops
    gen = torch.Generator().manual_seed(9)
    for B, N in ((1, 16), (7, 128), (1024, 150), (33, 160)):
        mask = (torch.rand(B, N, 1, generator=gen) < torch.rand(B, 1, 1, generator=gen)).long()
        mask[0] = 1
        if B > 2:
            mask[1] = 0
        mi = ops.mask_prepare(mask.cuda())
        rows = torch.full((B * N,), -1, dtype=torch.int32, device="cuda")
        ops.mask_rows(mi, rows)
        want = torch.nonzero(mask.reshape(-1)).flatten().int()
        assert int(mi.total) == want.numel()
        assert torch.equal(rows[:want.numel()].cpu(), want)
        assert bool((rows[want.numel():] == -1).all())


@pytest.mark.parametrize("name", ["S3", "S2", "S3_n150"])
def test_weight_gradients_over_real_rows_match_all_rows(golden, name, monkeypatch):
    """The multi-problem weight-gradient launch reduces over the rows of real particles only (GwProb.rows); the rows it
    leaves out carry gradient rows that are exactly zero, so every parameter gradient must match the all-rows launch to
    summation order."""
    res = {}
    for rows_on in ("1", "0"):
        monkeypatch.setenv("MMB_GW_ROWS", rows_on)
        g = golden(name)
        model = build(g).train()
        model.use_fused_training, model.use_fused_backward = True, True
        model.manual_seed(11)
        from tensorclass import TensorMultiModal
        c = lambda a: None if a is None else a.cuda()
        t = g.t
        state = TensorMultiModal(c(t["time"]), c(t.get("xt")), c(t["kt"]), c(t["mask"]))
        out = model.step_from_state(state, batch_of(g), "train", True)
        out["loss"].backward()
        res[rows_on] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    grads_close(res["1"], res["0"], name, "fused_rows_vs_all_rows")


@pytest.mark.parametrize("name", ["S3", "S2", "S3_n150"])
def test_unwritten_rows_of_idle_tiles_leave_no_trace(golden, name, monkeypatch):
    """With the weight gradients reduced over the rows of real particles, the reverse kernel's idle tiles leave their rows
    of the per-particle dZ slots UNWRITTEN (bit 1 of use_skip): whatever an earlier step with other masks left there
    must never be read.  Step A, step B (the masks of A in reverse jet order: other real rows, non-zero gradients in
    rows that A pads), step A again on the same persistent workspace -> the gradients of A, to the rounding of the
    atomics; and the same against the build that writes every row (MMB_BWD_DEAD_ROWS=0)."""
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    from conftest import GRAD_TOL_RERUN
    g = golden(name)
    N = g.t["mask"].shape[1]
    uni = "xt" not in g.t
    gen = torch.Generator().manual_seed(41)
    B = 96
    n = torch.randint(1, N + 1, (B,), generator=gen)
    n[:8] = torch.tensor([1, 31, 32, 33, 64, 96, N - 1, N])
    rank = torch.rand(B, N, generator=gen).argsort(dim=1).argsort(dim=1)
    mask = (rank < n[:, None]).long().unsqueeze(-1).cuda()
    rnd = lambda *s: torch.randn(*s, generator=gen).cuda()
    ri = lambda *s: torch.randint(0, 8, s, generator=gen).cuda()
    xt, x1, kt, k1 = rnd(B, N, 3), rnd(B, N, 3), ri(B, N, 1), ri(B, N, 1)
    t = (torch.rand(B, generator=gen) * 0.98 + 0.01).cuda().reshape(B, 1, 1)

    def run(dead):
        monkeypatch.setenv("MMB_BWD_DEAD_ROWS", dead)
        model = build(g).train()
        model.use_fused_training = model.use_fused_backward = True

        def step(flip):
            f = (lambda a: a.flip(0)) if flip else (lambda a: a)
            m = f(mask)
            model.manual_seed(7)
            st = TensorMultiModal(f(t), None if uni else f(xt) * m, f(kt) * m, m)
            tgt = TensorMultiModal(None, None if uni else f(x1) * m, f(k1) * m, m)
            src = TensorMultiModal(None, None if uni else torch.zeros_like(xt), torch.zeros_like(kt), torch.ones_like(m))
            out = model.step_from_state(st, DataCoupling(src, tgt, None), "train", True)
            model.zero_grad(set_to_none=True)
            out["loss"].backward()
            return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

        a = step(False)
        step(True)
        return a, step(False)

    a1, a1_again = run("1")
    grads_close(a1_again, a1, name, "dead_rows_rerun", tol=GRAD_TOL_RERUN)
    a0, _ = run("0")
    grads_close(a1, a0, name, "dead_rows_vs_written_rows", tol=GRAD_TOL_RERUN)


@pytest.mark.parametrize("name", ["S3", "S2", "S3_n150"])
def test_generation_does_not_depend_on_the_jet_schedule(golden, name, monkeypatch):
    """jet_order = NULL (natural launch order) against the scheduled order (mmb_jet_schedule): a jet's result does not
    depend on the workgroup that computes it or on its neighbours on the CU -- bit-identical, padded slots included."""
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    from mmb_hip import fused
    g = golden(name)
    model = build(g, num_timesteps=5)
    model.use_fused = True
    N = g.t["mask"].shape[1]
    uni = "xt" not in g.t
    gen = torch.Generator().manual_seed(23)
    B = 96
    n = torch.randint(1, N + 1, (B,), generator=gen)
    rank = torch.rand(B, N, generator=gen).argsort(dim=1).argsort(dim=1)
    mask = (rank < n[:, None]).long().unsqueeze(-1)
    k0 = torch.randint(0, 8, (B, N, 1), generator=gen) * mask
    x0 = None if uni else torch.randn(B, N, 3, generator=gen) * mask

    def run():
        model.manual_seed(11)
        src = TensorMultiModal(None, None if uni else x0.cuda(), k0.cuda(), mask.cuda())
        state = TensorMultiModal(torch.zeros_like(src.mask), None if uni else src.continuous.clone(), src.discrete.clone(),
                                 src.mask)
        paths = model.simulate_dynamics(state, DataCoupling(src, TensorMultiModal(None, None, None, src.mask), None))
        return (None if uni else paths.continuous[1].clone()), paths.discrete[1].clone()

    xs, ks = run()
    monkeypatch.setattr(fused, "jet_schedule", lambda mi: None)
    xn, kn = run()
    assert torch.equal(ks, kn)
    if not uni:
        assert torch.equal(xs, xn)
