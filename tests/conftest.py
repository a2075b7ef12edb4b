import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-bridges_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Parity margins (profiles/rN/parity_margins.json): tests record how far the HIP path really is from the reference
# values; written at session end when MMB_MARGINS_OUT names a file.
_MARGINS = {}


def record_margin(golden_name, path, key, value):
    slot = _MARGINS.setdefault(golden_name, {}).setdefault(path, {})
    slot[key] = max(float(value), slot.get(key, 0.0))


def rel_err(got, ref):
    """max |got - ref| / max |ref| (0/0 -> 0)"""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    scale = float(ref.abs().max())
    d = float((got - ref).abs().max())
    return d / scale if scale > 0 else d


# Gradient bar shared by the gradient tests: max |g - g_ref| <= GRAD_TOL * max |g_ref| per parameter (g_ref: the
# reference's autograd gradient, or the layered HIP path where a test compares the fused path with it).
# profiles/r3/parity_margins.json holds the measured values; the tolerances are <= 5 x the worst of them.
GRAD_TOL = 1.4e-5          # measured worst 2.83e-6 (S2 output_layer.weight_g, fused path vs the reference)
GRAD_NORM_TOL = 5.5e-6     # measured worst 1.13e-6 (global gradient norm)
GRAD_TOL_FULL_SIZE = 3.7e-4   # fused vs layered HIP step at B=1024, N=128, dropout 0.2: measured 7.5e-5
GRAD_TOL_RERUN = 2.5e-6    # the same fused step twice (atomics order only): measured 3.5e-7 .. 5.0e-7


def grads_close(got, ref, golden_name, path, tol=None, what=""):
    """Every parameter gradient of `got` within tol of `ref` in the max-norm of that parameter; margins recorded."""
    tol = GRAD_TOL if tol is None else tol
    assert got.keys() == ref.keys(), set(got) ^ set(ref)
    bad = []
    for n in got:
        e = rel_err(got[n], ref[n])
        record_margin(golden_name, path, "grad:" + n, e)
        if e > tol:
            bad.append((n, e))
    assert not bad, f"{what}{golden_name}/{path}: gradients beyond {tol}: {bad[:5]} ({len(bad)})"


@pytest.fixture(scope="session")
def margins():
    return record_margin


def pytest_sessionfinish(session, exitstatus):
    out = os.environ.get("MMB_MARGINS_OUT")
    if out and _MARGINS:
        summary = {}
        for gname, paths in _MARGINS.items():
            for path, vals in paths.items():
                for k, v in vals.items():
                    kind = k.split(":")[0]
                    key = f"{path}/{kind}"
                    summary[key] = max(summary.get(key, 0.0), v)
        with open(out, "w") as fh:
            json.dump({"definition": "max |hip - reference| / max |reference| per tensor (losses: relative error); "
                                     "reference = golden vectors of the imported reference (tools/make_golden.py)",
                       "worst_by_path_and_kind": summary, "per_golden": _MARGINS}, fh, indent=1, sort_keys=True)


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


class Golden:
    """A model_* fixture: tensors + params (stored, or re-synthesised by name) + flat cfg."""

    def __init__(self, name):
        from oracle.synth import synth_param
        z = load_npz("model_" + name)
        self.name = name
        self.cfg = json.loads(str(z["cfg_json"]))
        self.flat = json.loads(str(z["flat_cfg_json"]))
        names = json.loads(str(z["param_names"]))
        shapes = json.loads(str(z["param_shapes"]))
        seed = int(z["seed"])
        self.seed = seed
        self.params = {}
        for n, s in zip(names, shapes):
            key = "param:" + n
            self.params[n] = torch.from_numpy(z[key].copy()) if key in z else synth_param(n, s, seed)
        self.t = {}
        for k, v in z.items():
            if k.endswith("_json") or k in ("param_names", "param_shapes", "seed") or k.startswith("param:"):
                continue
            a = v
            if a.dtype == np.uint16:
                a = a.astype(np.float32)
            self.t[k] = torch.from_numpy(np.array(a))

    def grads(self):
        return {k[5:]: v for k, v in self.t.items() if k.startswith("grad:")}

    def grad_summaries(self):
        """{name: (norm, projection on the deterministic probe vector)} for EVERY parameter (paper-size fixtures)."""
        return {k[5:]: v for k, v in self.t.items() if k.startswith("gsum:")}

    def check_grad_summaries(self, grad_of, rtol, atol_scale):
        """grad_of(name) -> gradient tensor | None.  Compares norm and probe projection of every parameter; returns the
        worst |got - ref| / (largest gradient norm of the model) seen."""
        from oracle.synth import synth_param
        bad = []
        worst = 0.0
        gmax = max(float(v[0]) for v in self.grad_summaries().values())
        for n, ref in self.grad_summaries().items():
            gr = grad_of(n)
            gr = torch.zeros(self.params[n].shape) if gr is None else gr.detach().cpu()
            probe = synth_param(n + "#probe", gr.shape, 99).double()
            got = (float(gr.double().norm()), float((gr.double() * probe).sum()))
            tol = rtol * float(ref[0]) + atol_scale * gmax
            # the projection of a length-n vector on a unit-variance probe has the scale of the norm
            pscale = max(1.0, probe.norm().item() / max(gr.numel(), 1) ** 0.5)
            worst = max(worst, abs(got[0] - float(ref[0])) / gmax, abs(got[1] - float(ref[1])) / (gmax * pscale))
            if abs(got[0] - float(ref[0])) > tol or abs(got[1] - float(ref[1])) > tol * pscale:
                bad.append((n, got, (float(ref[0]), float(ref[1]))))
        assert not bad, f"{self.name}: gradient summaries differ: {bad[:4]} ({len(bad)} parameters)"
        return worst


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


MODEL_GOLDENS = ["mm_tiny", "mm_tiny_masked", "mm_mid", "mm_odd", "uni_tiny", "uni_mid", "S3", "S2", "S3_n150", "S2_n150",
                 "tr_tiny", "tr_odd", "S4"]
EPIC_GOLDENS = [n for n in MODEL_GOLDENS if not (n.startswith("tr_") or n == "S4")]
TRANSFORMER_GOLDENS = ["tr_tiny", "tr_odd", "S4"]
