"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes), against
the golden vectors of the imported reference and against the CPU oracle on seeded inputs.
Bar: jump-state indices bit-exact (near-tie categorical draws, |u*S - cdf| < 1e-5, excluded as
the noise-injection protocol prescribes); fp32 values rtol 1e-4 (north_star), usually 1e-6."""
import numpy as np
import pytest
import torch

from conftest import load_npz

pytestmark = pytest.mark.gpu

RT, AT = 1e-4, 1e-6


def T(a):
    return torch.from_numpy(np.array(a))


def close(a, b, rtol=RT, atol=AT):
    torch.testing.assert_close(a.cpu(), b.cpu(), rtol=rtol, atol=atol)


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test collected without a GPU")
    from mmb_hip import ops as o
    return o


def dev(t):
    return t.cuda().contiguous()


def test_continuous_bridges(ops):
    z = load_npz("bridges")
    t, x0, x1 = T(z["t"]), T(z["x0"]), T(z["x1"])
    for sigma in (1e-4, 0.3):
        for name, kind in (("UniformFlow", 0), ("SchrodingerBridge", 1)):
            tag = f"{name}_{sigma}"
            t_out, xt, _ = ops.bridge_sample(dev(t), 0.0, dev(x0), dev(x1), dev(T(z[tag + "_z"])), None, None, None,
                                             kind, sigma, 0, 0.0, 8)
            close(t_out, t)
            close(xt, T(z[tag + "_xt"]))
            d = ops.bridge_drift(dev(t), dev(T(z[tag + "_xt"])), dev(x0), dev(x1), kind)
            close(d, T(z[tag + "_drift"]), rtol=1e-4, atol=2e-3 if kind == 1 else 1e-6)


def test_telegraph_bridge(ops):
    z = load_npz("bridges")
    t, k0, k1 = T(z["t"]), T(z["k0"]), T(z["k1"])
    for kname, kind in (("ConstantThermostat", 0), ("LinearThermostat", 2), ("SigmoidThermostat", 4)):
        for gamma in (0.075, 0.125):
            tag = f"Telegraph_{kname}_{gamma}"
            p = ops.telegraph_probs(dev(t), dev(k0), dev(k1), kind, gamma, 8)
            close(p, T(z[tag + "_probs"]), rtol=1e-4, atol=1e-7)
            _, _, kt = ops.bridge_sample(dev(t), 0.0, None, None, None, dev(k0), dev(k1), dev(T(z[tag + "_u"])),
                                         0, 0.0, kind, gamma, 8)
            ok = T(z[tag + "_margin"]) > 1e-5
            assert torch.equal(kt.cpu().squeeze(-1)[ok], T(z[tag + "_kt"]).squeeze(-1)[ok])


def test_rate_and_jump_solvers(ops):
    z = load_npz("solvers")
    kinds = {"ConstantThermostat": 0, "InverseThermostat": 1, "InverseSquareThermostat": 3}
    for kname, gamma in (("ConstantThermostat", 0.125), ("ConstantThermostat", 0.075), ("InverseSquareThermostat", 0.125),
                         ("InverseThermostat", 0.125)):
        tag = f"{kname}_{gamma}"
        kind = kinds[kname]
        t, logits, k, dt = T(z[tag + "_t"]), T(z[tag + "_logits"]), T(z[tag + "_k"]), float(z[tag + "_dt"])
        rates = ops.telegraph_rate(dev(logits), dev(k), dev(t), 0.0, kind, gamma)
        close(rates, T(z[tag + "_rates"]), rtol=1e-4, atol=1e-5)
        for ov in ("wrap", "clamp"):
            for src in ("dn_" + ov, "dn_handmade"):
                kk = dev(k.clone())
                ro = torch.empty_like(dev(logits))
                ops.rate_tauleap_(dev(logits), kk, dev(t), 0.0, dt, kind, gamma, delta_n=dev(T(z[tag + "_" + src])),
                                  overflow=ov, rates_out=ro)
                want = z[tag + ("_k_tauleap_" + ov if src.startswith("dn_" + ov) else "_k_handmade_" + ov)]
                assert torch.equal(kk.cpu(), T(want)), (tag, ov, src)
                close(ro, T(z[tag + "_rates"]), rtol=1e-4, atol=1e-5)
        kk = dev(k.clone())
        ops.rate_euler_jump_(dev(logits), kk, dev(t), 0.0, dt, kind, gamma, u=dev(T(z[tag + "_u_euler"])))
        ok = T(z[tag + "_margin_euler"]) > 1e-5
        assert torch.equal(kk.cpu().squeeze(-1)[ok], T(z[tag + "_k_euler"]).squeeze(-1)[ok])
        # final-step argmax (multimodal_bridge_matching.py:253-255)
        kk = dev(k.clone())
        ops.rate_tauleap_(dev(logits), kk, dev(t), 0.0, dt, kind, gamma, delta_n=dev(T(z[tag + "_dn_wrap"])),
                          final_argmax=True)
        assert torch.equal(kk.cpu().squeeze(-1), T(z[tag + "_rates"]).max(dim=2)[1])
        # scalar-time form used by the generation loop
        tt = float(t[3])
        lg, k3 = logits[3:4].contiguous(), k[3:4].contiguous()
        r = ops.telegraph_rate(dev(lg), dev(k3), None, tt, kind, gamma)
        close(r, T(z[tag + "_rates"])[3:4], rtol=1e-4, atol=1e-5)


def test_generic_vocab_path(ops):
    """V != 8 takes the non-vectorised kernel instantiation."""
    from oracle import bridges as OB, solvers as OS
    g = torch.Generator().manual_seed(5)
    B, N, V = 3, 7, 5
    logits = torch.randn(B, N, V, generator=g)
    k = torch.randint(0, V, (B, N, 1), generator=g)
    t = torch.rand(B, generator=g) * 0.9 + 0.05
    rates = OB.telegraph_rate(t, k, logits, 0.1, V)
    close(ops.telegraph_rate(dev(logits), dev(k), dev(t), 0.0, 0, 0.1), rates, rtol=1e-4, atol=1e-5)
    dn = torch.poisson(rates * 0.02, generator=g)
    kk = dev(k.clone())
    ops.rate_tauleap_(dev(logits), kk, dev(t), 0.0, 0.02, 0, 0.1, delta_n=dev(dn))
    assert torch.equal(kk.cpu(), OS.tauleap_step(k, rates, dn, V))


def test_euler_steps(ops):
    from oracle import solvers as OS, bridges as OB
    g = torch.Generator().manual_seed(6)
    for shape in ((4, 16, 3), (3, 7, 3), (5, 128, 3)):
        x = torch.randn(*shape, generator=g)
        v = torch.randn(*shape, generator=g)
        xd = dev(x.clone())
        ops.euler_step_(xd, dev(v), 0.0101)
        close(xd, OS.euler_step(x, v, torch.tensor(0.0101)), rtol=1e-6, atol=1e-7)
        dW = torch.randn(*shape, generator=g)
        t = torch.rand(shape[0], generator=g) * 0.9 + 0.05
        sig = OB.schrodinger_diffusion(t.reshape(-1, 1, 1), 0.3)
        xd = dev(x.clone())
        ops.euler_maruyama_step_(xd, dev(v), 0.0101, dev(t), 0.0, 0.3, dW=dev(dW))
        close(xd, OS.euler_maruyama_step(x, v, torch.tensor(0.0101), sig, dW), rtol=1e-5, atol=1e-6)


def test_loss_fwd_bwd(ops):
    from oracle import model as OM, bridges as OB
    g = torch.Generator().manual_seed(7)
    for bridge_kind, freq in ((0, None), (1, [0.304, 0.0847, 0.2986, 0.3078, 0.0013, 0.0014, 0.0011, 0.001])):
        B, N, Dc, V = 5, 19, 3, 8
        hc = torch.randn(B, N, Dc, generator=g, requires_grad=True)
        hd = torch.randn(B, N, V, generator=g, requires_grad=True)
        t = torch.rand(B, generator=g) * 0.9 + 0.05
        xt, x0, x1 = (torch.randn(B, N, Dc, generator=g) for _ in range(3))
        k1 = torch.randint(0, V, (B, N, 1), generator=g)
        mask = (torch.rand(B, N, 1, generator=g) > 0.4).long()
        mask[:, 0] = 1
        if bridge_kind == 0:
            tgt = OB.uniform_flow_drift(xt, x0, x1)
        else:
            tgt = OB.schrodinger_drift(t.reshape(-1, 1, 1), xt, x0, x1)
        lc, ld = OM.loss_fn(hc, hd, tgt, k1, mask, freq)
        s = torch.tensor([0.3, -0.2], requires_grad=True)
        loss = OM.multimode_loss([lc, ld], "learnable", s)
        loss.backward()
        mi = ops.mask_prepare(dev(mask))
        assert int(mi.total.cpu()) == int(mask.sum())
        coef = ops.multimode_coef(dev(s.detach()), None, "cuda")
        losses, dc, dd = ops.loss_fwd_bwd(dev(hc.detach()), dev(hd.detach()), dev(t), dev(xt), dev(x0), dev(x1),
                                          bridge_kind, dev(k1), mi, dev(torch.tensor(freq)) if freq else None, coef)
        close(losses[0], lc.detach(), rtol=1e-5)
        close(losses[1], ld.detach(), rtol=1e-5)
        close(dc, hc.grad, rtol=1e-4, atol=1e-7)
        close(dd, hd.grad, rtol=1e-4, atol=1e-7)
        tot, d_s = ops.multimode_loss(losses, dev(s.detach()), None)
        close(tot[0], loss.detach(), rtol=1e-5)
        close(d_s, s.grad, rtol=1e-5)
        tot, _ = ops.multimode_loss(losses, None, [1.0, 1.0])
        close(tot[0], (lc + ld).detach(), rtol=1e-5)


@pytest.mark.parametrize("M,N,widths,kinds,Nout,act", [
    (4 * 16, 16, [16, 32, 16], [2, 0, 1], 64, "lrelu"),          # aligned BK=32
    (4 * 16, 16, [8, 24, 8], [2, 0, 1], 40, "gelu"),             # aligned BK=16, ragged Nout
    (3 * 11, 11, [6, 5, 7], [2, 0, 1], 12, None),                # generic path
    (1024 * 8, 128, [32, 128, 32], [2, 0, 1], 128, "lrelu"),     # 4-wave blocks, NT=4
    (5000, 50, [48], [0], 3, None),                              # head-like, ragged M
    (64, 1, [160, 16], [0, 0], 64, "gelu"),                      # per-jet rows
    (1024, 1, [144], [0], 64, "lrelu"),                          # per-jet split-K kernel (fc_glob1, S3 branch)
    (1000, 1, [288], [0], 128, "lrelu"),                         # split-K, ragged M, widest K
    (300, 1, [20], [0], 8, None),                                # split-K, K below one slice, ragged Nout
    (1024, 1, [384], [0], 128, "gelu"),                          # split-K upper limit (S2 projection)
    (512, 1, [400], [0], 16, None),                              # beyond the split-K limit: layered small-M path
    (40 * 150, 150, [64], [0], 192, "gelu"),                     # S4 in_proj: three column blocks of 64
    (40 * 150, 150, [256], [0], 64, None),                       # S4 linear2
    (4100, 1, [272], [0], 100, "lrelu"),                         # ragged everything, K % 32 = 16
    (4200, 100, [1280, 32], [0, 1], 40, None),                   # K too wide for a resident weight block: tiled kernel
])
@pytest.mark.parametrize("path", ["x6", "stream", "tiled"])
def test_linear_cat(ops, M, N, widths, kinds, Nout, act, path, monkeypatch):
    """path: M > 4096 rows with 16-aligned segments run a row-streaming kernel (weight block resident in LDS): x6 = the
    default, six bf16-plane products per fp32 product; stream = the same on the fp32 MFMA; tiled = the block-tiled
    fp32-MFMA kernel.  Every other shape runs the same kernel under all three values."""
    monkeypatch.setenv("MMB_LINEAR_PATH", path)
    g = torch.Generator().manual_seed(M + Nout)
    B = M // N
    segs_cpu, segs = [], []
    mask = (torch.rand(M, generator=g) > 0.3).float()
    cols = []
    for w, kd in zip(widths, kinds):
        if kd == 0:
            a = torch.randn(M, w, generator=g)
            cols.append(a)
        else:
            a = torch.randn(B, w, generator=g)
            full = a.repeat_interleave(N, dim=0)
            cols.append(full * mask[:, None] if kd == 2 else full)
        segs.append((dev(a), kd))
    K = sum(widths)
    W = torch.randn(Nout, K, generator=g) / K ** 0.5
    b = torch.randn(Nout, generator=g)
    res = torch.randn(M, Nout, generator=g)
    z = torch.nn.functional.linear(torch.cat(cols, 1), W, b)
    y = {None: z, "gelu": torch.nn.functional.gelu(z), "lrelu": torch.nn.functional.leaky_relu(z)}[act] + res
    out, pre = ops.linear_cat(segs, dev(W), dev(b), N, maskf=dev(mask), act=act, residual=dev(res), want_pre=True)
    close(pre, z, rtol=1e-4, atol=2e-5)
    close(out, y, rtol=1e-4, atol=2e-5)
    # second-stage epilogue: y2 = lrelu(y1) + res2, written into a column slice of a wider buffer
    res2 = torch.randn(M, Nout, generator=g)
    wide = torch.zeros(M, Nout + 8, device="cuda")
    y1, y2 = ops.linear_cat(segs, dev(W), dev(b), N, maskf=dev(mask), act=act, residual=dev(res), post="lrelu",
                            res2=dev(res2), out2=wide[:, 8:])
    close(y1, y, rtol=1e-4, atol=2e-5)
    close(y2, torch.nn.functional.leaky_relu(y) + res2, rtol=1e-4, atol=2e-5)
    assert float(wide[:, :8].abs().sum()) == 0.0
    # dropout keep-rate and scaling
    _, yd = ops.linear_cat(segs, dev(W), dev(b), N, maskf=dev(mask), act=act, residual=dev(res), post="lrelu",
                           drop_p=0.25, seed=3, offset=9, want_y1=False)
    ref2 = torch.nn.functional.leaky_relu(y)
    kept = yd.cpu() != 0
    if M * Nout > 2000:
        assert abs(kept.float().mean().item() - 0.75) < 0.05
    close(yd.cpu()[kept], (ref2 / 0.75)[kept], rtol=1e-4, atol=3e-5)


def test_pool_embed_weightnorm(ops, golden):
    from oracle import encoder as OE
    g = golden("mm_mid")
    p, t = g.params, g.t
    mask = t["mask"]
    mi = ops.mask_prepare(dev(mask))
    B, N = mask.shape[:2]
    lt, lc, ld, gt = OE.embedder(p, t["time"], t["xt"], t["kt"], mask)
    table = OE.wn_weight(p, "embedder.discrete_embedding")
    temb, cont, disc = ops.embed(dev(t["time"].reshape(-1)), 0.0, dev(t["xt"]), dev(t["kt"]), mi,
                                 dev(p["embedder.time_embedding.0.W"]), dev(p["embedder.time_embedding.1.weight"]),
                                 dev(p["embedder.time_embedding.1.bias"]), dev(p["embedder.continuous_embedding.weight"]),
                                 dev(p["embedder.continuous_embedding.bias"]), dev(table))
    close(temb, gt, rtol=1e-4, atol=2e-5)
    close(cont.reshape(B, N, -1), lc, rtol=1e-4, atol=1e-6)
    assert torch.equal(disc.reshape(B, N, -1).cpu().long(), ld)
    x = torch.randn(B * N, 24)
    pooled = ops.meansum_pool(dev(x), mi)
    want = OE.meansum_pool(mask, x.reshape(B, N, -1), torch.zeros(B, 0))
    close(pooled, want, rtol=1e-5, atol=1e-6)
    # weight norm over a packed buffer of two layers
    v1, v2 = torch.randn(6, 10), torch.randn(3, 7)
    g1, g2 = torch.rand(6, 1) + 0.5, torch.rand(3, 1) + 0.5
    vflat = torch.cat([v1.reshape(-1), v2.reshape(-1)])
    gflat = torch.cat([g1.reshape(-1), g2.reshape(-1)])
    off = torch.tensor([i * 10 for i in range(6)] + [60 + i * 7 for i in range(3)], dtype=torch.int32)
    ln = torch.tensor([10] * 6 + [7] * 3, dtype=torch.int32)
    Wf, inv = ops.weight_norm(dev(off), dev(ln), dev(vflat), dev(gflat), want_inv=True)
    close(Wf[:60].reshape(6, 10), torch._weight_norm(v1, g1, 0), rtol=1e-5)
    close(Wf[60:].reshape(3, 7), torch._weight_norm(v2, g2, 0), rtol=1e-5)
    v1r, g1r = v1.clone().requires_grad_(True), g1.clone().requires_grad_(True)
    dW = torch.randn(6, 10)
    (torch._weight_norm(v1r, g1r, 0) * dW).sum().backward()
    dWf = torch.cat([dW.reshape(-1), torch.zeros(21)])
    dv, dg = ops.weight_norm_bwd(dev(off), dev(ln), dev(vflat), dev(gflat), inv, dev(dWf))
    close(dv[:60].reshape(6, 10), v1r.grad, rtol=1e-4, atol=1e-6)
    close(dg[:6], g1r.grad.reshape(-1), rtol=1e-4, atol=1e-6)


def test_source_uniform_statistics(ops):
    from oracle.synth import load_pmf
    pmf = torch.tensor(load_pmf(), dtype=torch.float32)
    B, N = 4096, 128
    x0, k0, mask, maskf, counts = ops.source_uniform(B, N, 3, 8, "cuda", pmf=dev(pmf), sample_masks=True, seed=7, offset=1)
    x0, k0, mask, counts = x0.cpu(), k0.cpu(), mask.cpu(), counts.cpu()
    assert torch.equal(mask.sum((1, 2)).int(), counts)
    assert torch.equal(maskf.cpu(), mask.squeeze(-1).float())
    n_ref = torch.multinomial(pmf / pmf.sum(), 200000, True).clamp(max=N).float()
    assert abs(counts.float().mean() - n_ref.mean()) < 1.0
    assert abs(counts.float().std() - n_ref.std()) < 1.0
    # slots uniformly occupied
    occ = mask.squeeze(-1).float().mean(0)
    assert (occ - occ.mean()).abs().max() < 0.05
    assert abs(x0.mean()) < 5e-3 and abs(x0.std() - 1) < 5e-3
    hist = torch.bincount(k0.reshape(-1), minlength=8).float() / k0.numel()
    assert (hist - 0.125).abs().max() < 3e-3
    # reproducible
    again = ops.source_uniform(B, N, 3, 8, "cuda", pmf=dev(pmf), sample_masks=True, seed=7, offset=1)
    assert torch.equal(again[0].cpu(), x0) and torch.equal(again[2].cpu(), mask)
    x2, _, m2, _, c2 = ops.source_uniform(8, 16, 3, 8, "cuda", sample_masks=False)
    assert int(m2.sum()) == 8 * 16 and int(c2.sum()) == 8 * 16


def test_bridge_sample_rng_statistics(ops):
    B, N, V = 512, 128, 8
    g = torch.Generator().manual_seed(3)
    x1 = torch.randn(B, N, 3, generator=g)
    k1 = torch.randint(0, V, (B, N, 1), generator=g)
    t, x0, k0, xt, kt = ops.bridge_sample_rng(1e-4, dev(x1), dev(k1), 0, 1e-4, 0, 0.125, V, seed=11, offset=5)
    t, x0, k0, xt, kt = (a.cpu() for a in (t, x0, k0, xt, kt))
    assert t.min() >= 1e-4 and t.max() < 1.0 and abs(t.mean() - 0.5) < 0.05
    assert abs(x0.mean()) < 1e-2 and abs(x0.std() - 1) < 1e-2
    tt = t.reshape(-1, 1, 1)
    assert (xt - (tt * x1 + (1 - tt) * x0)).abs().max() < 1e-3          # sigma = 1e-4
    # k_t must be k0 or k1 most of the time and move from k0 to k1 as t grows
    late = tt.expand(-1, N, 1) > 0.9
    assert (kt[late] == k1[late]).float().mean() > 0.9
    early = tt.expand(-1, N, 1) < 0.1
    assert (kt[early] == k0[early]).float().mean() > 0.9
    # empirical law of k_t against the exact posterior, pooled
    from oracle import bridges as OB
    p = OB.telegraph_transition_probability(tt, k0, k1, 0.125, V)
    p = p / p.sum(-1, keepdim=True)
    exp_hit = p.gather(2, kt).mean()
    assert abs(exp_hit - (p * p).sum(-1).mean()) < 0.01


def test_invalid_arguments_raise(ops):
    from mmb_hip import MMBError
    with pytest.raises(MMBError):
        ops.euler_step_(torch.zeros(4), torch.zeros(4), 0.1)            # CPU tensors: no fallback
    with pytest.raises(MMBError):
        ops.telegraph_rate(torch.zeros(1, 1, 64, device="cuda"), torch.zeros(1, 1, 1, dtype=torch.int64, device="cuda"),
                           None, 0.5, 0, 0.1)                           # V > 32
    with pytest.raises(MMBError):
        ops.telegraph_rate(torch.zeros(1, 1, 8, device="cuda"), torch.zeros(1, 1, 1, dtype=torch.int64, device="cuda"),
                           None, 0.5, 9, 0.1)                           # bad thermostat


@pytest.mark.parametrize("M,K,Nout,x_pad", [
    (131, 64, 64, 0), (4096, 64, 64, 8), (1024, 288, 128, 0), (777, 16, 3, 0), (2049, 128, 128, 32),
    (64, 37, 70, 3), (5000, 144, 64, 0), (300, 128, 8, 0), (70000, 80, 96, 0), (66000, 256, 128, 8)])
@pytest.mark.parametrize("gw_path", ["x6", "f32"])
def test_grad_weight_rows(ops, M, K, Nout, x_pad, gw_path, monkeypatch):
    """(gw_path: six bf16-plane products per fp32 product, the default, or the fp32 MFMA: MMB_GW_PATH.  The two long
    streams take the 4-tile / 1024-row blocks with two chunks in flight and ragged K / output windows.)
    Row-streaming weight gradient against a plain torch fp32 matmul (ragged rows/columns, strided X,
    dW written into a column slice of a wider gradient, atomic accumulation on top of existing values)."""
    monkeypatch.setenv("MMB_GW_PATH", gw_path)
    g = torch.Generator().manual_seed(M + K + Nout)
    Xfull = torch.randn(M, K + x_pad, generator=g).cuda()
    X = Xfull[:, x_pad:] if x_pad else Xfull
    dZ = torch.randn(M, Nout, generator=g).cuda()
    wide = torch.zeros(Nout, K + 5, device="cuda")
    wide[:, :] = 0.25
    dW = wide[:, 2:2 + K]
    db = torch.full((Nout,), -1.0, device="cuda")
    ops.grad_weight_rows(X, dZ, dW, db)
    want = dZ.double().t() @ X.double()
    scale = float(want.abs().max())
    close(dW.double() - 0.25, want, rtol=1e-4, atol=2e-5 * scale)
    close(db.double() + 1.0, dZ.double().sum(0), rtol=1e-4, atol=2e-5 * float(dZ.abs().sum(0).max()))
    assert float((wide[:, :2] - 0.25).abs().max()) == 0.0 and float((wide[:, 2 + K:] - 0.25).abs().max()) == 0.0


POST_CASES = [(tag, rank, tr) for tag in ("tokens3", "onehot11", "onehot8") for rank in (3, 4)
              for tr in ("none", "standardize", "normalize", "log_pt")]


@pytest.mark.parametrize("tag,rank,tr", POST_CASES)
def test_postprocess(ops, tag, rank, tr):
    """Device post-processing against the reference's own _postprocess outputs (tests/golden/postprocess.npz:
    callbacks_generator.py:99-128 run on synthetic samples) -- four transforms, (J,N,D) and stacked (T,J,N,D) input
    (upstream's `[:, :, 0]` quirk), tokens and both onehot shapes: tokens bit-exact, x rtol 1e-6."""
    import json
    from conftest import load_npz
    from pipeline.postprocess import postprocess
    from tensorclass import TensorMultiModal
    z = load_npz("postprocess")
    stats = json.loads(str(z[f"{tag}:stats_json"]))
    feats = "onehot" if tag.startswith("onehot") else "tokens"
    x = torch.from_numpy(z[f"{tag}:r{rank}:x"])
    k = torch.from_numpy(z[f"{tag}:r{rank}:k"]) if feats == "tokens" else None
    mask = torch.from_numpy(z[f"{tag}:r{rank}:mask"])
    paths = TensorMultiModal(None, dev(x), None if k is None else dev(k), dev(mask))
    out = postprocess(paths, None if tr == "none" else tr, {"target": stats}, feats, 8)
    key = f"{tag}:r{rank}:{tr}:continuous"
    if key in z:
        close(out.continuous, torch.from_numpy(z[key]), rtol=1e-6, atol=1e-6)
    else:
        assert out.continuous is None
    want_k = torch.from_numpy(z[f"{tag}:r{rank}:{tr}:discrete"])
    assert out.discrete.dtype == torch.int64 and torch.equal(out.discrete.cpu(), want_k)


def test_postprocess_discrete_only(ops):
    from pipeline.postprocess import postprocess
    from tensorclass import TensorMultiModal
    g = torch.Generator().manual_seed(5)
    k = torch.randint(0, 8, (37, 50, 1), generator=g)
    mask = (torch.rand(37, 50, 1, generator=g) > 0.4).long()
    out = postprocess(TensorMultiModal(None, None, dev(k), dev(mask)), None, None)
    assert torch.equal(out.discrete.cpu(), (k * mask).long())


def test_grad_weight_rows_multi(ops):
    """Many weight-gradient problems per launch (device table, one launch per tile class) against torch matmuls:
    mixed row counts (per-particle and per-jet), replicas, strided operands, column slices of wider gradients."""
    g = torch.Generator().manual_seed(77)
    shapes = [(20000, 64, 64, 8), (1000, 64, 64, 1), (20000, 128, 128, 8), (513, 32, 64, 1), (20000, 64, 128, 8),
              (700, 144, 64, 1), (20000, 16, 64, 8), (333, 64, 16, 1), (900, 256, 128, 1), (20000, 128, 32, 8),
              (20000, 32, 128, 8)]
    probs, keep = {}, []
    for M, K, Nout, R in shapes:
        Xw = torch.randn(M, K + 4, generator=g).cuda()
        X = Xw[:, 4:]                                    # strided, still 16-byte aligned
        dZ = torch.randn(M, Nout, generator=g).cuda()
        dWr = torch.zeros(R, Nout * (K + 8), device="cuda")
        dbr = torch.zeros(R, Nout, device="cuda")
        dWs = dWr[0].view(Nout, K + 8)[:, 4:4 + K]
        assert ops.gw_fast(X, dZ)
        cls = ops.gw_class(K, Nout)
        lst = probs.setdefault(cls, [[], 0])
        rec, n = ops.gw_record(X, dZ, dWs, dbr[0], R, dWr.stride(0), dbr.stride(0), lst[1])
        lst[0].append(rec)
        lst[1] += n
        keep.append((X, dZ, dWr, dbr, K, Nout))
    assert len(probs) >= 4
    for cls, (recs, nblk) in sorted(probs.items()):
        ops.grad_weight_rows_multi(cls, torch.tensor(recs, dtype=torch.int64).cuda(), len(recs), nblk)
    for X, dZ, dWr, dbr, K, Nout in keep:
        want = dZ.double().t() @ X.double()
        got = dWr.double().sum(0).view(Nout, K + 8)
        close(got[:, 4:4 + K], want, rtol=1e-4, atol=2e-5 * float(want.abs().max()))
        assert float(got[:, :4].abs().max()) == 0.0 and float(got[:, 4 + K:].abs().max()) == 0.0
        close(dbr.double().sum(0), dZ.double().sum(0), rtol=1e-4, atol=2e-5 * float(dZ.abs().sum(0).max()))


def test_weight_norm_multi_and_backward(ops):
    """All layers' weight norms (and their reverse) in one launch each against torch autograd of g*v/||v||."""
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 48), (16, 144), (128, 192), (3, 64)]
    R = 4
    vs = [torch.randn(O, K, generator=g).cuda().requires_grad_() for O, K in shapes]
    gs = [(torch.rand(O, 1, generator=g) + 0.5).cuda().requires_grad_() for O, K in shapes]
    bias = torch.randn(40, generator=g).cuda()
    n = sum(O * K for O, K in shapes) + 40
    flat = torch.full((n,), -7.0, device="cuda")
    inv = torch.zeros(sum(O for O, _ in shapes), device="cuda")
    ar = lambda m: torch.arange(m, dtype=torch.int64)
    rows, off, ioff = [], 0, 0
    for v, gg in zip(vs, gs):
        O, K = v.shape
        r = torch.empty(O, 5, dtype=torch.int64)
        r[:, 0], r[:, 1] = v.data_ptr() + 4 * K * ar(O), gg.data_ptr() + 4 * ar(O)
        r[:, 2], r[:, 3], r[:, 4] = flat.data_ptr() + 4 * (off + K * ar(O)), inv.data_ptr() + 4 * (ioff + ar(O)), K
        rows.append(r)
        off += O * K
        ioff += O
    rows.append(torch.tensor([[bias.data_ptr(), 0, flat.data_ptr() + 4 * off, 0, 40]], dtype=torch.int64))
    table = torch.cat(rows).cuda()
    ops.weight_norm_multi(table, table.shape[0])
    off = 0
    Ws = []
    for v, gg in zip(vs, gs):
        O, K = v.shape
        W = gg * v / v.norm(dim=1, keepdim=True)
        close(flat[off:off + O * K].view(O, K), W, rtol=1e-5, atol=1e-6)
        Ws.append(W)
        off += O * K
    assert torch.equal(flat[off:], bias)
    # reverse: dW, db arrive as R replicas
    brows, outs, ioff = [], [], 0
    for v, gg, W in zip(vs, gs, Ws):
        O, K = v.shape
        dWr = torch.randn(R, O * K + 4, generator=g).cuda()
        dbr = torch.randn(R, O, generator=g).cuda()
        dv, dg, db = torch.zeros(O, K, device="cuda"), torch.zeros(O, device="cuda"), torch.zeros(O, device="cuda")
        r = torch.empty(O, 12, dtype=torch.int64)
        r[:, 0], r[:, 1], r[:, 2] = v.data_ptr() + 4 * K * ar(O), gg.data_ptr() + 4 * ar(O), inv.data_ptr() + 4 * (ioff + ar(O))
        r[:, 3], r[:, 4] = dWr.data_ptr() + 4 * K * ar(O), dbr.data_ptr() + 4 * ar(O)
        r[:, 5], r[:, 6], r[:, 7] = dv.data_ptr() + 4 * K * ar(O), dg.data_ptr() + 4 * ar(O), db.data_ptr() + 4 * ar(O)
        r[:, 8], r[:, 9], r[:, 10], r[:, 11] = K, R, dWr.stride(0), dbr.stride(0)
        brows.append(r)
        outs.append((v, gg, W, dWr, dbr, dv, dg, db))
        ioff += O
    table = torch.cat(brows).cuda()
    ops.weight_norm_bwd_multi(table, table.shape[0])
    for v, gg, W, dWr, dbr, dv, dg, db in outs:
        O, K = v.shape
        dW = dWr[:, :O * K].sum(0).view(O, K)
        gv, ggr = torch.autograd.grad(W, (v, gg), dW, retain_graph=True)
        close(dv, gv, rtol=1e-4, atol=1e-5)
        close(dg, ggr.reshape(-1), rtol=1e-4, atol=1e-5)
        close(db, dbr.sum(0), rtol=1e-5, atol=1e-5)


def test_jet_observables_and_w1(ops):
    """Device jet observables (one wave per jet) against the CPU restatement of JetFeatures, and the device
    Wasserstein-1 against scipy."""
    import scipy.stats
    from oracle.observables import jet_observables as ref_obs
    g = torch.Generator().manual_seed(8)
    B, N = 203, 128
    mask = (torch.rand(B, N, 1, generator=g) < 0.4).long()
    x = torch.randn(B, N, 3, generator=g) * torch.tensor([0.5, 0.3, 0.3])
    x[..., 0] = x[..., 0].abs() + 0.01
    x = x * mask                                          # a post-processed sample: zeros in the padded slots
    k = torch.randint(0, 8, (B, N, 1), generator=g) * mask
    want = ref_obs(x, k, mask)
    got = ops.jet_observables(dev(x), dev(k), dev(mask)).cpu()
    for i, name in enumerate(ops.OBSERVABLES):
        close(got[:, i], want[name], rtol=2e-4, atol=2e-5)
    # discrete-only sample
    got_d = ops.jet_observables(None, dev(k), dev(mask)).cpu()
    for i, name in enumerate(ops.OBSERVABLES):
        if name.startswith("num") or name == "charge":
            close(got_d[:, i], want[name], rtol=0, atol=0)
    a, b = torch.randn(1000, generator=g), torch.randn(777, generator=g) * 1.3 + 0.2
    w = float(ops.wasserstein_1d(dev(a), dev(b)))
    assert abs(w - scipy.stats.wasserstein_distance(a.numpy(), b.numpy())) < 1e-6


@pytest.mark.parametrize("M,K,Nout", [(131072, 3, 16), (1000, 3, 8), (77, 4, 32), (5000, 1, 5)])
def test_grad_weight_smallk(ops, M, K, Nout):
    g = torch.Generator().manual_seed(M + K)
    X = torch.randn(M, K, generator=g).cuda()
    dZ = torch.randn(M, Nout + 2, generator=g).cuda()[:, 1:1 + Nout]
    dW = torch.full((Nout, K), 0.5, device="cuda")
    db = torch.full((Nout,), -2.0, device="cuda")
    ops.grad_weight_smallk(X, dZ, dW, db)
    want = dZ.double().t() @ X.double()
    close(dW.double() - 0.5, want, rtol=1e-4, atol=2e-5 * float(want.abs().max()))
    close(db.double() + 2.0, dZ.double().sum(0), rtol=1e-4, atol=2e-5 * float(dZ.abs().sum(0).max()))


def test_pack_params_matches_torch_split():
    """mmb_pack_params (one launch: gathers + exact three-plane bf16 split) against the torch restatement, bit for bit"""
    from mmb_hip import fused
    g = torch.Generator().manual_seed(5)
    n_flat, n_main, F = 5000, 1236, 7
    flat = torch.randn(n_flat + 1, generator=g) * 10.0 ** torch.empty(n_flat + 1).uniform_(-4, 3, generator=g)
    flat[-1] = 0.0
    gather = torch.randint(0, n_flat + 1, (n_main,), generator=g)
    gather2 = torch.where(torch.rand(n_main, generator=g) < 0.3, torch.randint(0, n_flat, (n_main,), generator=g),
                          torch.full((n_main,), n_flat))
    gather_x = torch.randint(0, n_flat + 1, (F * 512,), generator=g)
    for g2 in (gather2, None):
        want = torch.empty(n_main + F * 768)
        fused._fill_packed(want, n_main, flat, gather, g2, gather_x)                    # host tensors: torch lines
        got = torch.zeros(n_main + F * 768, device="cuda")
        fused._fill_packed(got, n_main, flat.cuda(), gather.cuda(), None if g2 is None else g2.cuda(), gather_x.cuda())
        assert torch.equal(got.cpu().view(torch.int32), want.view(torch.int32))


@pytest.mark.parametrize("M,K,Nout", [(6000, 64, 256), (300, 128, 256), (5000, 48, 40), (64, 37, 20)])
@pytest.mark.parametrize("act", ["relu", "lrelu", "gelu", "selu"])
def test_linear_cat_gate(ops, M, K, Nout, act):
    """mmb_linear_cat_gate = (x W^T) * act'(gate) on every kernel family of mmb_linear_cat (row streaming, split-K,
    tiled, generic), against fp64 autograd of the activation."""
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(Nout, K, generator=g) / K ** 0.5
    gate = torch.randn(M, Nout, generator=g)
    gate[::7] = 0.0                                     # relu output exactly 0: no gradient
    z = gate.double().requires_grad_(True)
    fn = {"relu": torch.relu, "lrelu": torch.nn.functional.leaky_relu, "gelu": torch.nn.functional.gelu,
          "selu": torch.nn.functional.selu}[act]
    y = fn(z)
    y.sum().backward()                                  # z.grad = act'(z)
    ref = y.detach().float() if act == "selu" else gate   # SELU takes its derivative from the OUTPUT
    want = torch.nn.functional.linear(x.double(), W.double()) * z.grad
    out = torch.full((M, Nout), 7.0, device="cuda")
    ops.linear_cat_gate([(dev(x), ops.SEG_ROW)], dev(W), 1, dev(ref), out, act=act)
    close(out.double(), want, rtol=1e-4, atol=2e-5)
