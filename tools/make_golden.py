#!/usr/bin/env python3
"""Golden-vector harness: runs the REFERENCE hot path (imported from /root/reference) under
the noise-injection protocol of SURVEY.md 8c and writes small fixtures to tests/golden/.

Runs ONLY in the authoring container (it needs /root/reference).  It contains harness code
only -- no reference code -- and its outputs are data (inputs + expected outputs).

    python tools/make_golden.py            # regenerate every fixture

Stubs registered here (not in the reference): `h5py` (only HDF5 load/save touch it) and a
minimal `lightning` whose LightningModule is nn.Module + a cpu `device` property.
Harness repairs: config.topk=None, target_path -> the reference's
tests/resources (metadata.json), K1-compliant encoder widths, explicit solver names.
"""
import copy
import json
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.synth import synth_param, aoj_like_batch  # noqa: E402  (no reference code)

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _install_stubs():
    sys.modules["h5py"] = types.ModuleType("h5py")
    L = types.ModuleType("lightning")

    class LM(torch.nn.Module):
        @property
        def device(self):
            return torch.device("cpu")

        def save_hyperparameters(self, *a, **k):
            pass

    L.LightningModule = LM
    lp = types.ModuleType("lightning.pytorch")
    lp.LightningDataModule = object
    L.pytorch = lp
    sys.modules["lightning"] = L
    sys.modules["lightning.pytorch"] = lp
    sys.path.insert(0, os.path.join(REF, "multimodal_bridges"))


# ------------------------------------------------------------------ noise tape --
class Tape:
    """Wraps torch RNG entry points while the reference runs; records every draw.
    Categorical draws over V classes go through the injected-u inverse CDF."""

    FUNCS = ("rand", "randn", "randint", "randn_like", "poisson", "randperm", "multinomial")

    def __init__(self, vocab_size):
        self.V = vocab_size
        self.rec = {f: [] for f in self.FUNCS}
        self.rec["u_cat"] = []
        self.rec["cat_margin"] = []
        self._orig = {}

    def __enter__(self):
        for f in self.FUNCS:
            self._orig[f] = getattr(torch, f)
            setattr(torch, f, self._wrap(f))
        return self

    def __exit__(self, *a):
        for f, o in self._orig.items():
            setattr(torch, f, o)

    def _wrap(self, name):
        orig = self._orig[name]

        def fn(*a, **k):
            if name == "multinomial":
                probs, ns = a[0], a[1]
                if probs.dim() == 2 and probs.shape[-1] == self.V and ns == 1:
                    u = self._orig["rand"](probs.shape[0])
                    cdf = torch.empty_like(probs)
                    acc = torch.zeros(probs.shape[0])
                    for v in range(self.V):
                        acc = acc + probs[:, v]
                        cdf[:, v] = acc
                    thr = u * cdf[:, -1]
                    kk = (cdf <= thr[:, None]).sum(-1).clamp(max=self.V - 1)
                    self.rec["u_cat"].append(u.clone())
                    self.rec["cat_margin"].append((cdf - thr[:, None]).abs().min(-1)[0])
                    return kk.long().unsqueeze(-1)
            out = orig(*a, **k)
            self.rec[name].append(out.clone())
            return out

        return fn


# ------------------------------------------------------------------- configs --
def base_cfg(modality="multi-modal"):
    cfg = {
        "trainer": {"max_epochs": 1, "optimizer_name": "AdamW",
                    "optimizer_params": {"lr": 1e-3, "weight_decay": 5e-5}},
        "data": {"modality": modality, "batch_size": 4, "target_name": "AspenOpenJets",
                 "target_path": REF + "/tests/resources/", "source_name": "SourceUniform",
                 "min_num_particles": 0, "max_num_particles": 16, "vocab_size": 8,
                 "discrete_features": "tokens"},
        "model": {"bridge_discrete": "TelegraphBridge", "gamma": 0.125,
                  "thermostat_fn": "ConstantThermostat", "num_timesteps": 10, "time_eps": 1e-4,
                  "solver_discrete": "tauleap", "loss_weights": "fixed"},
        "encoder": {"name": "MultiModalEPiC", "num_blocks": [1, 1, 2], "dim_hidden_local": [6, 2, 8],
                    "dim_hidden_glob": [6, 2, 8], "skip_connection": True, "dropout": 0.0,
                    "dim_emb_time": 4, "dim_emb_continuous": 4, "dim_emb_discrete": 4},
        "comet_logger": {"project_name": "golden"},
        "checkpoints": {},
    }
    if modality in ("multi-modal", "continuous"):
        cfg["data"]["continuous_features"] = ["pt", "eta_rel", "phi_rel"]
        cfg["model"].update(bridge_continuous="UniformFlow", sigma=1e-4, solver_continuous="euler")
    if modality == "discrete":
        cfg["encoder"].update(name="UniModalEPiC", num_blocks=2, dim_hidden_local=8, dim_hidden_glob=4,
                              dim_emb_continuous=0)
        cfg["model"].update(gamma=0.075)
    return cfg


def variants():
    v = {}
    c = base_cfg()
    v["mm_tiny"] = c
    c = base_cfg()
    c["encoder"].update(num_blocks=[2, 2, 3], dim_hidden_local=[32, 32, 64], dim_hidden_glob=[8, 8, 16],
                        dim_emb_time=16, dim_emb_continuous=8, dim_emb_discrete=8)
    c["model"].update(loss_weights="learnable")
    v["mm_mid"] = c
    c = base_cfg()
    c["encoder"].update(num_blocks=[1, 1, 2], dim_hidden_local=[12, 20, 32], dim_hidden_glob=[5, 3, 8],
                        dim_emb_time=6, dim_emb_continuous=5, dim_emb_discrete=7, skip_connection=False)
    c["model"].update(bridge_continuous="SchrodingerBridge", sigma=0.05, thermostat_fn="LinearThermostat",
                      solver_discrete="euler")
    c["data"]["vocab_freq"] = [0.304, 0.0847, 0.2986, 0.3078, 0.0013, 0.0014, 0.0011, 0.001]
    v["mm_odd"] = c
    c = base_cfg("discrete")
    v["uni_tiny"] = c
    c = base_cfg("discrete")
    c["encoder"].update(num_blocks=3, dim_hidden_local=32, dim_hidden_glob=8, dim_emb_time=16, dim_emb_discrete=16)
    v["uni_mid"] = c
    # paper-size canonical configs; weights are synthesised by name, not stored
    c = base_cfg()
    c["data"].update(max_num_particles=128)
    c["encoder"].update(num_blocks=[4, 4, 4], dim_hidden_local=[64, 64, 128], dim_hidden_glob=[16, 16, 32],
                        dim_emb_time=32, dim_emb_continuous=16, dim_emb_discrete=16)
    c["model"].update(loss_weights="learnable")
    v["S3"] = c
    c = base_cfg("discrete")
    c["data"].update(max_num_particles=128)
    c["encoder"].update(num_blocks=10, dim_hidden_local=128, dim_hidden_glob=16, dim_emb_time=128, dim_emb_discrete=128)
    v["S2"] = c
    # transformer set encoder (particle_transformer.py:52-167)
    c = base_cfg()
    c["encoder"].update(name="MultiModalParticleTransformer", num_blocks=[2, 1, 2], dim_hidden_local=[8, 8, 16],
                        num_heads=2, dim_emb_time=8, dim_emb_continuous=4, dim_emb_discrete=4)
    c["encoder"].pop("dim_hidden_glob")
    v["tr_tiny"] = c
    c = base_cfg()                     # no fusion stage, time width != model width (time_proj is a Linear), head dim 6/4
    c["encoder"].update(name="MultiModalParticleTransformer", num_blocks=[1, 2, 0], dim_hidden_local=[12, 8, 0],
                        num_heads=2, dim_emb_time=6, dim_emb_continuous=5, dim_emb_discrete=7)
    c["encoder"].pop("dim_hidden_glob")
    c["model"].update(loss_weights="learnable")
    v["tr_odd"] = c
    c = base_cfg()                     # SURVEY 8d S4 widths
    c["data"].update(max_num_particles=150)
    c["encoder"].update(name="MultiModalParticleTransformer", num_blocks=[2, 2, 2], dim_hidden_local=[64, 64, 128],
                        num_heads=4, dim_emb_time=64, dim_emb_continuous=64, dim_emb_discrete=64)
    c["encoder"].pop("dim_hidden_glob")
    c["model"].update(loss_weights="learnable")
    v["S4"] = c
    return v


def flat_cfg(cfg):
    """The plain-dict view the oracle uses."""
    d = {}
    d.update(cfg["model"])
    d["vocab_size"] = cfg["data"]["vocab_size"]
    d["vocab_freq"] = cfg["data"].get("vocab_freq")
    d["modality"] = cfg["data"]["modality"]
    d["encoder_name"] = cfg["encoder"]["name"]
    d["skip_connection"] = cfg["encoder"]["skip_connection"]
    d["num_heads"] = cfg["encoder"].get("num_heads")
    return d


def build_model(cfg, seed):
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    config = ExperimentConfigs(copy.deepcopy(cfg))
    config.topk = None                                    # Q2
    model = MultiModalBridgeMatching(config)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(synth_param(n, p.shape, seed))
    return model.eval()


def npz_save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  {os.path.getsize(path)/1024:.1f} KiB")


def make_mask(B, N, kind, g):
    if kind == "ones":
        return torch.ones(B, N, 1, dtype=torch.int64)
    n = torch.randint(1, N + 1, (B,), generator=g)
    if kind == "random_full":             # jet 0 full (Q4: every key ignored), jet 1 with ONE padded slot
        n[0], n[1] = N, N - 1
    if kind == "random_notfull":          # every jet keeps a padded slot, jet 0 exactly one
        n = n.clamp(max=N - 1)
        n[0] = N - 1
    r = torch.rand(B, N, generator=g)
    rank = r.argsort(1).argsort(1)
    return (rank < n[:, None]).long().unsqueeze(-1)


# ------------------------------------------------------------ elementwise goldens --
def golden_thermostats():
    from model import thermostats as T
    t0 = torch.tensor([1e-4, 0.013, 0.25, 0.5, 0.77, 0.9, 0.9999], dtype=torch.float32)
    t1 = torch.tensor([0.2, 0.5, 0.3, 0.9, 1.0, 0.95, 1.0], dtype=torch.float32)
    out = {"t0": t0, "t1": t1}
    for kind in ("ConstantThermostat", "InverseThermostat", "LinearThermostat",
                 "InverseSquareThermostat", "SigmoidThermostat"):
        for gamma in (0.075, 0.125):
            th = getattr(T, kind)(gamma, 8)
            out[f"{kind}_{gamma}"] = th.w_ts(t0, t1)
            if kind != "SigmoidThermostat":       # the reference's tanh(float) raises there
                out[f"{kind}_{gamma}_to1"] = th.w_ts(t0, 1.0)
    npz_save("thermostats", **out)


def golden_bridges():
    from model import thermostats as T
    from model.bridges import UniformFlow, SchrodingerBridge, TelegraphBridge
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    g = torch.Generator().manual_seed(11)
    B, N, Dc, V = 6, 16, 3, 8
    t = 1e-4 + (1 - 1e-4) * torch.rand(B, generator=g)
    t[0] = 1e-4
    t[1] = 0.9999
    time = t.reshape(-1, 1, 1)
    x0, x1 = torch.randn(B, N, Dc, generator=g), torch.randn(B, N, Dc, generator=g)
    k0, k1 = torch.randint(0, V, (B, N, 1), generator=g), torch.randint(0, V, (B, N, 1), generator=g)
    # make sure every (k0,k1) pair occurs
    pairs = [(a, b) for a in range(V) for b in range(V)]
    for i, (a, b) in enumerate(pairs[: B * N]):
        k0.view(-1)[i], k1.view(-1)[i] = a, b
    mask = torch.ones(B, N, 1, dtype=torch.int64)
    batch = DataCoupling(TensorMultiModal(None, x0, k0, mask), TensorMultiModal(None, x1, k1, mask), None)
    out = dict(t=t, x0=x0, x1=x1, k0=k0, k1=k1)
    for sigma in (1e-4, 0.3):
        for cls in (UniformFlow, SchrodingerBridge):
            br = cls(sigma)
            with Tape(V) as tape:
                xt = br.sample(time, batch)
            z = tape.rec["randn_like"][0]
            st = TensorMultiModal(time.clone(), xt, None, mask)
            tag = f"{cls.__name__}_{sigma}"
            out[tag + "_z"], out[tag + "_xt"] = z, xt
            out[tag + "_drift"] = br.drift(st, batch)
            if cls is SchrodingerBridge:
                st2 = TensorMultiModal(t.reshape(-1, 1).clone(), xt, None, mask)
                out[tag + "_diffusion"] = br.diffusion(st2)          # (B,N,1) after broadcast_time
    for kind in ("ConstantThermostat", "LinearThermostat", "SigmoidThermostat"):
        for gamma in (0.075, 0.125):
            th = getattr(T, kind)(gamma, V)
            tb = TelegraphBridge(gamma, V, th)
            tag = f"Telegraph_{kind}_{gamma}"
            out[tag + "_probs"] = tb.transition_probability(time, k0, k1)
            with Tape(V) as tape:
                kt = tb.sample(time, batch)
            out[tag + "_u"] = tape.rec["u_cat"][0].reshape(B, N)
            out[tag + "_margin"] = tape.rec["cat_margin"][0].reshape(B, N)
            out[tag + "_kt"] = kt
    npz_save("bridges", **out)


def golden_solvers():
    from model import thermostats as T
    from model.bridges import TelegraphBridge
    from model.solvers import DiscreteSolver
    from tensorclass import TensorMultiModal
    g = torch.Generator().manual_seed(12)
    B, N, V = 6, 16, 8
    out = {}
    for kind, gamma in (("ConstantThermostat", 0.125), ("ConstantThermostat", 0.075), ("InverseSquareThermostat", 0.125),
                        ("InverseThermostat", 0.125)):
        tb = TelegraphBridge(gamma, V, getattr(T, kind)(gamma, V))
        t = torch.rand(B, generator=g) * 0.98 + 0.01
        t[0], t[1] = 1e-4, 0.9999
        logits = 3 * torch.randn(B, N, V, generator=g)
        k = torch.randint(0, V, (B, N, 1), generator=g)
        dt = torch.tensor(0.0101)
        heads = TensorMultiModal(None, None, logits, None)

        class FakeModel:                        # stands in for the LightningModule inside the solver
            bridge_discrete = tb

            def __call__(self, state, batch):
                return heads

        cfgobj = types.SimpleNamespace(model=types.SimpleNamespace(solver_discrete="tauleap"),
                                       data=types.SimpleNamespace(vocab_size=V), topk=None)
        solver = DiscreteSolver(cfgobj, FakeModel())
        tag = f"{kind}_{gamma}"
        st = TensorMultiModal(t.reshape(-1, 1).clone(), None, k.clone(), None)
        out[tag + "_t"], out[tag + "_logits"], out[tag + "_k"], out[tag + "_dt"] = t, logits, k, dt
        out[tag + "_rates"] = tb.rate(st, heads)
        for ov in ("wrap", "clamp"):
            st = TensorMultiModal(t.reshape(-1, 1).clone(), None, k.clone(), None)
            with Tape(V) as tape:
                st, rates = solver.tauleap_step(st, None, dt, overflow=ov)
            out[tag + f"_dn_{ov}"] = tape.rec["poisson"][0]
            out[tag + f"_k_tauleap_{ov}"] = st.discrete
        # hand-made delta_n covering sum in {0,1,2+}
        dn = torch.zeros(B, N, V)
        idx = torch.randint(0, V, (B, N), generator=g)
        cnt = torch.randint(0, 3, (B, N), generator=g).float()
        dn.scatter_(2, idx[:, :, None], cnt[:, :, None])
        dn[:, ::5, 3] += 1
        orig = torch.poisson
        torch.poisson = lambda *_a, **_k: dn.clone()
        try:
            for ov in ("wrap", "clamp"):
                st = TensorMultiModal(t.reshape(-1, 1).clone(), None, k.clone(), None)
                st, _ = solver.tauleap_step(st, None, dt, overflow=ov)
                out[tag + f"_k_handmade_{ov}"] = st.discrete
        finally:
            torch.poisson = orig
        out[tag + "_dn_handmade"] = dn
        st = TensorMultiModal(t.reshape(-1, 1).clone(), None, k.clone(), None)
        with Tape(V) as tape:
            st, _ = solver.euler_step(st, None, dt)
        out[tag + "_u_euler"] = tape.rec["u_cat"][0].reshape(B, N)
        out[tag + "_margin_euler"] = tape.rec["cat_margin"][0].reshape(B, N)
        out[tag + "_k_euler"] = st.discrete
    npz_save("solvers", **out)


# ---------------------------------------------------------------- model goldens --
def capture_encoder(model, state, batch):
    """Forward with hooks on the sub-encoders; returns dict of intermediate tensors."""
    cap = {}
    hooks = []

    def hook(name):
        def fn(mod, inp, outp):
            if isinstance(outp, tuple):
                for i, o in enumerate(outp):
                    cap[f"{name}.{i}"] = o.detach().clone()
            else:
                cap[name] = outp.detach().clone()
        return fn

    for name, mod in model.encoder.named_modules():
        if name in ("continuous_encoder", "discrete_encoder", "fused_encoder", "epic",
                    "continuous_encoder.epic_proj", "continuous_encoder.epic_layers.0",
                    "epic.epic_proj", "epic.epic_layers.0", "fused_encoder.epic_layers.0",
                    "continuous_encoder.transformer.layers.0", "discrete_encoder.transformer.layers.0",
                    "fused_encoder.transformer.layers.0"):
            hooks.append(mod.register_forward_hook(hook(name)))
    loc, glob = model.embedder(state, batch)
    cap["emb.time"] = loc.time.detach().clone()
    if loc.continuous is not None:
        cap["emb.continuous"] = loc.continuous.detach().clone()
    if loc.discrete is not None:
        cap["emb.discrete"] = loc.discrete.detach().clone()
    cap["emb.glob_time"] = glob.time.detach().clone()
    heads = model.encoder(loc, glob)
    for h in hooks:
        h.remove()
    return heads, cap


def golden_model(name, cfg, B, N, mask_kind, store_params, seed=3, steps_list=(10,), explicit_source=False):
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    print(f"[{name}] B={B} N={N} mask={mask_kind}")
    cfg = copy.deepcopy(cfg)
    cfg["data"]["max_num_particles"] = N
    model = build_model(cfg, seed)
    modality = cfg["data"]["modality"]
    V = cfg["data"]["vocab_size"]
    g = torch.Generator().manual_seed(100 + seed)
    mask = make_mask(B, N, mask_kind, g)
    x1 = torch.randn(B, N, 3, generator=g) * mask if modality != "discrete" else None
    k1 = torch.randint(0, V, (B, N, 1), generator=g) * mask
    out = {"cfg_json": json.dumps(cfg), "flat_cfg_json": json.dumps(flat_cfg(cfg)), "seed": seed,
           "param_names": json.dumps([n for n, _ in model.named_parameters()]),
           "param_shapes": json.dumps([list(p.shape) for _, p in model.named_parameters()]),
           "mask": mask, "k1": k1}
    if x1 is not None:
        out["x1"] = x1
    if store_params:
        for n, p in model.named_parameters():
            out["param:" + n] = p.detach()

    # ---- training step (golden 5,6,7) under the tape
    target = TensorMultiModal(None, None if x1 is None else x1.clone(), k1.clone(), mask.clone())
    batch = DataCoupling(TensorMultiModal(), target, None)
    torch.manual_seed(555 + seed)
    with Tape(V) as tape:
        state = model.sample_bridges(batch)
    out["t_u"] = tape.rec["rand"][0]
    if modality != "discrete":
        out["x0"] = tape.rec["randn"][0]
        out["z"] = tape.rec["randn_like"][0]
        out["xt"] = state.continuous
    out["k0"] = tape.rec["randint"][0]
    out["u_k"] = tape.rec["u_cat"][0].reshape(B, N)
    out["u_k_margin"] = tape.rec["cat_margin"][0].reshape(B, N)
    out["kt"] = state.discrete
    out["time"] = state.time
    st_clone = state.clone()
    heads, cap = capture_encoder(model, st_clone, batch)
    for kname, val in cap.items():
        out["cap:" + kname] = val
    model.zero_grad()
    heads = model.forward(state, batch)
    lc, ld = model.loss_fn(heads, state, batch)
    loss = model.loss_multimode([lc, ld])
    loss.backward()
    if heads.continuous is not None:
        out["head_continuous"] = heads.continuous
    out["head_discrete"] = heads.discrete
    out["loss"], out["loss_continuous"], out["loss_discrete"] = loss, lc, ld
    gnorm2 = 0.0
    for n, p in model.named_parameters():
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        gnorm2 += float((gr.double() ** 2).sum())
        if store_params:
            out["grad:" + n] = gr
        else:                                     # paper-size: keep a few representative grads
            if n.endswith("output_layer.weight_g") or n.endswith("head.2.bias") or n.endswith("loss_weights") \
                    or n.endswith("epic_layers.0.fc_glob1.weight_v") or n.endswith("time_embedding.1.bias") \
                    or n.endswith("layers.1.self_attn.in_proj_bias") or n.endswith("layers.0.norm1.weight"):
                out["grad:" + n] = gr
            # ... and pin EVERY parameter's gradient through two scalars: its norm and its projection on a
            # deterministic probe vector
            probe = synth_param(n + "#probe", gr.shape, 99).double()
            out["gsum:" + n] = np.array([float(gr.double().norm()), float((gr.double() * probe).sum())])
    out["grad_norm"] = np.float64(gnorm2 ** 0.5)

    # ---- generation (golden 8): predict_step under the tape
    for steps in steps_list:
        model.config.model.num_timesteps = steps
        target = TensorMultiModal(None, None if x1 is None else x1.clone(), k1.clone(), mask.clone())
        src = TensorMultiModal()
        if explicit_source:
            # given source (no multiplicity draw from the pmf: at N = 16 nearly every jet would be full, and under
            # no_grad torch's fused encoder-layer path returns NaN for a jet without a padded slot -- every key is
            # ignored, Q4 -- so the reference cannot generate such a jet at all; the training path gives zeros there)
            gs = torch.Generator().manual_seed(4242 + seed + steps)
            src = TensorMultiModal(None, None if x1 is None else torch.randn(B, N, 3, generator=gs),
                                   torch.randint(0, V, (B, N, 1), generator=gs),
                                   make_mask(B, N, "random_notfull", gs))
        batch = DataCoupling(src, target, None)
        torch.manual_seed(777 + seed + steps)
        with Tape(V) as tape, torch.no_grad():
            paths, _ = model.predict_step(batch, 0)
        tag = f"gen{steps}"
        out[tag + ":src_mask"] = paths.mask[0]
        out[tag + ":src_k"] = paths.discrete[0]
        out[tag + ":final_k"] = paths.discrete[1]
        if modality != "discrete":
            out[tag + ":src_x"] = paths.continuous[0]
            out[tag + ":final_x"] = paths.continuous[1]
        if cfg["model"]["solver_discrete"] == "tauleap":
            dn = torch.stack(tape.rec["poisson"], 0)
            assert dn.max() < 65535
            out[tag + ":delta_n"] = dn.numpy().astype(np.uint16)
        else:
            out[tag + ":u_jump"] = torch.stack(tape.rec["u_cat"], 0).reshape(steps, B, N)
            out[tag + ":u_jump_margin"] = torch.stack(tape.rec["cat_margin"], 0).reshape(steps, B, N)
    npz_save("model_" + name, **out)


def golden_toy(name="toy", B=48, seed=21, steps_list=(10, 100)):
    """S1 (BASELINE configs[0]): the toy notebook's MultiModalMLP -- its class definition is executed from
    notebooks/data_toys.ipynb cell 3 where it lies -- inside the reference's MultiModalBridgeMatching (bridges, loss_fn,
    MultiModeLoss, solvers, simulate_dynamics) at N = 1, Dc = 2, V = 8.  Harness glue only: an identity embedder (the
    notebook network takes the raw state) and an adapter that gives the MLP the (B,1,1) time it squeezes and returns
    (B,1,.) heads.  Source / target points come from oracle.toys (toys.py needs torchdyn, absent here)."""
    import contextlib
    import io
    from torch import nn
    from tensorclass import TensorMultiModal
    from datamodules.datasets import DataCoupling
    from oracle import toys as OT
    print(f"[{name}] B={B}")
    with open(REF + "/notebooks/data_toys.ipynb") as fh:
        nb = json.load(fh)
    ns = {"TensorMultiModal": TensorMultiModal}
    with contextlib.redirect_stdout(io.StringIO()):
        exec(compile("".join(nb["cells"][3]["source"]), "data_toys.ipynb[3]", "exec"), ns)
    MLP = ns["MultiModalMLP"]

    class Adapter(MLP):
        def forward(self, state_local, state_global=None):
            Bn = state_local.discrete.shape[0]
            st = TensorMultiModal(state_local.time.reshape(Bn, 1, 1), state_local.continuous, state_local.discrete,
                                  state_local.mask)
            with contextlib.redirect_stdout(io.StringIO()):          # the notebook class prints shapes
                h = MLP.forward(self, st)
            return TensorMultiModal(None, h.continuous.unsqueeze(1), h.discrete.unsqueeze(1), state_local.mask)

    class Ident(nn.Module):
        def forward(self, state, batch):
            return state, None

    ref_cfg = base_cfg()
    ref_cfg["data"].update(max_num_particles=1, continuous_features=["x", "y"])
    ref_cfg["model"].update(gamma=0.075, loss_weights="fixed")
    cfg = copy.deepcopy(ref_cfg)                  # what THIS repo's model is built from
    cfg["encoder"] = {"name": "MultiModalMLP", "num_blocks": 4, "dim_hidden_local": 64, "skip_connection": False,
                      "dropout": 0.0, "dim_emb_time": 1}
    cfg["data"].update(target_name="TwoMoons", source_name="NGaussians")
    model = build_model(ref_cfg, seed)
    model.embedder = Ident()
    model.encoder = Adapter(dim=2, vocab_size=8, dim_hidden=64)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(synth_param(n, p.shape, seed))
    model.eval()
    V = 8
    g = torch.Generator().manual_seed(100 + seed)
    x1, k1 = OT.two_moons(B, generator=g)
    x0, k0 = OT.n_gaussians(B, generator=g)
    mask = torch.ones(B, 1, 1, dtype=torch.int64)
    out = {"cfg_json": json.dumps(cfg), "flat_cfg_json": json.dumps(flat_cfg(cfg)), "seed": seed,
           "param_names": json.dumps([n for n, _ in model.named_parameters()]),
           "param_shapes": json.dumps([list(p.shape) for _, p in model.named_parameters()]),
           "mask": mask, "k1": k1, "x1": x1, "x0": x0, "k0": k0}
    for n, p in model.named_parameters():
        out["param:" + n] = p.detach()
    mk = lambda x, k: TensorMultiModal(None, x.clone(), k.clone(), mask.clone())
    batch = DataCoupling(mk(x0, k0), mk(x1, k1), None)
    torch.manual_seed(555 + seed)
    with Tape(V) as tape:
        state = model.sample_bridges(batch)
    out["t_u"] = tape.rec["rand"][0]
    out["z"] = tape.rec["randn_like"][0]
    out["xt"], out["kt"], out["time"] = state.continuous, state.discrete, state.time
    out["u_k"] = tape.rec["u_cat"][0].reshape(B, 1)
    out["u_k_margin"] = tape.rec["cat_margin"][0].reshape(B, 1)
    model.zero_grad()
    heads = model.forward(state, batch)
    lc, ld = model.loss_fn(heads, state, batch)
    loss = model.loss_multimode([lc, ld])
    loss.backward()
    out["head_continuous"], out["head_discrete"] = heads.continuous, heads.discrete
    out["loss"], out["loss_continuous"], out["loss_discrete"] = loss, lc, ld
    gnorm2 = 0.0
    for n, p in model.named_parameters():
        out["grad:" + n] = p.grad
        gnorm2 += float((p.grad.double() ** 2).sum())
    out["grad_norm"] = np.float64(gnorm2 ** 0.5)
    for steps in steps_list:
        model.config.model.num_timesteps = steps
        gs = torch.Generator().manual_seed(4242 + seed + steps)
        sx, sk = OT.n_gaussians(B, generator=gs)
        batch = DataCoupling(mk(sx, sk), mk(x1, k1), None)
        torch.manual_seed(777 + seed + steps)
        with Tape(V) as tape, torch.no_grad():
            paths, _ = model.predict_step(batch, 0)
        tag = f"gen{steps}"
        out[tag + ":src_mask"], out[tag + ":src_k"], out[tag + ":src_x"] = paths.mask[0], paths.discrete[0], paths.continuous[0]
        out[tag + ":final_k"], out[tag + ":final_x"] = paths.discrete[1], paths.continuous[1]
        dn = torch.stack(tape.rec["poisson"], 0)
        assert dn.max() < 65535
        out[tag + ":delta_n"] = dn.numpy().astype(np.uint16)
    npz_save("model_" + name, **out)


def synth_pfcands(J, M, F=10, seed=31):
    """AOJ-shaped raw PFCands block (J,M,F) f32: columns px,py,pz,E,d0,d0Err,dz,dzErr,pid,puppi; a jet-like
    collimated spray around a random axis, pids from the AOJ flavour mix plus a few the reference filters out
    (|pid| < 11), unknown ones (pid 310) and zero padding."""
    r = np.random.default_rng(seed)
    pf = np.zeros((J, M, F), dtype=np.float32)
    pids = np.array([22, 130, -211, 211, -11, 11, -13, 13, 310, 1, 2], dtype=np.float32)
    prob = np.array([0.30, 0.08, 0.29, 0.29, 0.005, 0.005, 0.005, 0.005, 0.01, 0.005, 0.005])
    prob = prob / prob.sum()
    for j in range(J):
        n = int(r.integers(1, M + 1)) if j else M                   # jet 0 is full
        jeta, jphi = r.uniform(-2, 2), r.uniform(-np.pi, np.pi)
        pt = r.exponential(20.0, n).astype(np.float32) + 0.5
        eta = (jeta + 0.3 * r.standard_normal(n)).astype(np.float32)
        phi = (jphi + 0.3 * r.standard_normal(n)).astype(np.float32)
        pf[j, :n, 0] = pt * np.cos(phi)
        pf[j, :n, 1] = pt * np.sin(phi)
        pf[j, :n, 2] = pt * np.sinh(eta)
        pf[j, :n, 3] = pt * np.cosh(eta)
        pf[j, :n, 4:8] = r.standard_normal((n, 4)).astype(np.float32)
        pf[j, :n, 8] = r.choice(pids, n, p=prob)
        pf[j, :n, 9] = 1.0
    pf[1, 0, :4] = 0.0                                              # a candidate with pt = 0 and E = 0 but a valid pid
    pf[2, 1, 0:2] = 0.0                                             # pt = 0, pz and E kept: the `where=pt != 0` branch
    return pf


def golden_aoj():
    """the reference's AspenOpenJets arithmetic (datamodules/aoj.py) on a synthetic PFCands block --
    everything of _read_aoj_file after the h5py read, _preprocess for the three transforms, _extract_metadata."""
    from datamodules.aoj import AspenOpenJets
    print("[aoj]")
    aoj = AspenOpenJets(data_dir="/nonexistent", data_files=None)
    J, M = 24, 30
    raw = synth_pfcands(J, M)
    out = {"pfcands": raw}
    pf = aoj._filter_particles(raw.copy())
    feats, mask = aoj._compute_continuous_coordinates(pf)
    tokens = aoj._map_pid_to_tokens(pf[:, :, -2])[:, :, None]
    for k in ("pt", "eta_rel", "phi_rel"):
        out["feat:" + k] = feats[k].astype(np.float32)
    out["tokens"] = tokens.astype(np.int64)
    out["mask"] = mask.astype(np.int64)
    cont = torch.cat([torch.tensor(feats[k], dtype=torch.float32) for k in ("pt", "eta_rel", "phi_rel")], dim=-1)
    disc = torch.tensor(tokens, dtype=torch.float32)
    mk = torch.tensor(mask[:, :, None], dtype=torch.long)
    for tr in ("none", "standardize", "normalize", "log_pt"):
        torch.manual_seed(90)
        with Tape(8) as tape:
            c, d, m, md = aoj._preprocess(cont.clone(), disc.clone(), mk.clone(), None if tr == "none" else tr)
        out[f"pre:{tr}:perm"] = tape.rec["randperm"][0]
        out[f"pre:{tr}:continuous"], out[f"pre:{tr}:discrete"], out[f"pre:{tr}:mask"] = c, d, m
        out[f"pre:{tr}:metadata_json"] = json.dumps(md)
    npz_save("aoj", **out)


def golden_postprocess():
    """the reference's own JetGeneratorCallback._postprocess (pipeline/callbacks_generator.py:99-128) on
    synthetic samples: the four transforms x {(J,N,D) sample, stacked (T,J,N,D) paths} x {tokens, onehot (Dc = 3 and
    Dc = 0)}.  The module imports Lightning's Callback / rank_zero_only, pandas, matplotlib and datamodules.utils
    (fastjet, awkward, vector: absent); the method touches none of them, so the harness registers empty stand-ins
    for the absent ones (here, never in the reference) before the import."""
    import tempfile
    print("[postprocess]")
    cb = types.ModuleType("lightning.pytorch.callbacks")
    cb.Callback = object
    ut = types.ModuleType("lightning.pytorch.utilities")
    ut.rank_zero_only = lambda f: f
    sys.modules["lightning.pytorch.callbacks"] = cb
    sys.modules["lightning.pytorch.utilities"] = ut
    sys.modules["lightning.pytorch"].callbacks = cb
    sys.modules["lightning.pytorch"].utilities = ut
    du = types.ModuleType("datamodules.utils")
    du.JetFeatures = object
    import datamodules
    had = sys.modules.get("datamodules.utils")
    sys.modules["datamodules.utils"] = du
    try:
        from pipeline.callbacks_generator import JetGeneratorCallback
    finally:
        if had is not None:
            sys.modules["datamodules.utils"] = had
        else:
            del sys.modules["datamodules.utils"]
    from tensorclass import TensorMultiModal
    g = torch.Generator().manual_seed(41)
    J, N, V = 7, 9, 8
    out = {}
    for D, feats in ((3, "tokens"), (3 + V, "onehot"), (V, "onehot")):
        stats = {"mean": (torch.randn(D, generator=g) * 0.5).tolist(), "std": (torch.rand(D, generator=g) + 0.3).tolist(),
                 "min": (-1.0 - torch.rand(D, generator=g)).tolist(), "max": (1.0 + torch.rand(D, generator=g)).tolist()}
        tag = f"{feats}{D}"
        out[f"{tag}:stats_json"] = json.dumps(stats)
        for rank in (3, 4):
            shape = (J, N) if rank == 3 else (2, J, N)
            x = torch.randn(*shape, D, generator=g)
            k = torch.randint(0, V, (*shape, 1), generator=g)
            mask = (torch.rand(*shape, 1, generator=g) > 0.35).long()
            out[f"{tag}:r{rank}:x"], out[f"{tag}:r{rank}:k"], out[f"{tag}:r{rank}:mask"] = x, k, mask
            for tr in ("none", "standardize", "normalize", "log_pt"):
                with tempfile.TemporaryDirectory() as tmp:
                    with open(os.path.join(tmp, "metadata.json"), "w") as fh:
                        json.dump({"target": stats}, fh)
                    cfg = types.SimpleNamespace(path=tmp, data=types.SimpleNamespace(
                        transform=None if tr == "none" else tr, discrete_features=feats, vocab_size=V))
                    obj = JetGeneratorCallback(cfg)
                    paths = TensorMultiModal(None, x.clone(), k.clone() if feats == "tokens" else None, mask.clone())
                    res = obj._postprocess(paths, transform=cfg.data.transform)
                if res.continuous is not None:
                    out[f"{tag}:r{rank}:{tr}:continuous"] = res.continuous
                out[f"{tag}:r{rank}:{tr}:discrete"] = res.discrete
    npz_save("postprocess", **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    _install_stubs()
    torch.set_num_threads(8)
    # the one data artefact we reuse from the reference: the AOJ multiplicity pmf
    with open(REF + "/tests/resources/metadata.json") as f:
        meta = json.load(f)
    with open(os.path.join(OUT, "metadata.json"), "w") as f:
        json.dump({"source": "reference tests/resources/metadata.json 'categorical_probs'",
                   "max_num_particles_per_jet": meta["max_num_particles_per_jet"],
                   "categorical_probs": meta["categorical_probs"]}, f)
    golden_thermostats()
    golden_bridges()
    golden_solvers()
    V = variants()
    golden_model("mm_tiny", V["mm_tiny"], 4, 16, "ones", True, steps_list=(10, 100))
    golden_model("mm_tiny_masked", V["mm_tiny"], 5, 8, "random", True, seed=4)
    golden_model("mm_mid", V["mm_mid"], 4, 16, "random", True, seed=5, steps_list=(10, 100))
    golden_model("mm_odd", V["mm_odd"], 3, 11, "random", True, seed=6)
    golden_model("uni_tiny", V["uni_tiny"], 4, 16, "random", True, seed=7, steps_list=(10, 100))
    golden_model("uni_mid", V["uni_mid"], 4, 16, "random", True, seed=8, steps_list=(10, 100))
    golden_model("S3", V["S3"], 2, 128, "random", False, seed=9, steps_list=(5, 100))
    golden_model("S2", V["S2"], 2, 128, "random", False, seed=10, steps_list=(5, 100))
    golden_model("S3_n150", V["S3"], 2, 150, "random", False, seed=14, steps_list=(5, 100))      # AOJ's 150-particle clouds
    golden_model("S2_n150", V["S2"], 2, 150, "random", False, seed=15, steps_list=(5, 100))
    golden_model("tr_tiny", V["tr_tiny"], 4, 16, "random_full", True, seed=11, steps_list=(10, 100), explicit_source=True)
    golden_model("tr_odd", V["tr_odd"], 3, 11, "random_full", True, seed=12, explicit_source=True)
    golden_model("S4", V["S4"], 2, 150, "random_full", False, seed=13, steps_list=(5, 100), explicit_source=True)
    golden_toy()
    golden_aoj()
    golden_postprocess()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "toy":       # regenerate one fixture without touching the others
        _install_stubs()
        torch.set_num_threads(8)
        golden_toy()
    elif len(sys.argv) > 1 and sys.argv[1] == "aoj":
        _install_stubs()
        golden_aoj()
    elif len(sys.argv) > 1 and sys.argv[1] == "transformers":        # tr_tiny / S4 with their 100-step generation pins
        _install_stubs()
        torch.set_num_threads(8)
        V = variants()
        golden_model("tr_tiny", V["tr_tiny"], 4, 16, "random_full", True, seed=11, steps_list=(10, 100), explicit_source=True)
        golden_model("S4", V["S4"], 2, 150, "random_full", False, seed=13, steps_list=(5, 100), explicit_source=True)
    elif len(sys.argv) > 1 and sys.argv[1] == "n150":                # the EPiC models at AOJ's 150 particles per jet
        _install_stubs()
        torch.set_num_threads(8)
        V = variants()
        golden_model("S3_n150", V["S3"], 2, 150, "random", False, seed=14, steps_list=(5, 100))
        golden_model("S2_n150", V["S2"], 2, 150, "random", False, seed=15, steps_list=(5, 100))
    elif len(sys.argv) > 1 and sys.argv[1] == "postprocess":
        _install_stubs()
        golden_postprocess()
    else:
        main()
