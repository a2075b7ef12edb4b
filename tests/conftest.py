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
        """grad_of(name) -> gradient tensor | None.  Compares norm and probe projection of every parameter."""
        from oracle.synth import synth_param
        bad = []
        gmax = max(float(v[0]) for v in self.grad_summaries().values())
        for n, ref in self.grad_summaries().items():
            gr = grad_of(n)
            gr = torch.zeros(self.params[n].shape) if gr is None else gr.detach().cpu()
            probe = synth_param(n + "#probe", gr.shape, 99).double()
            got = (float(gr.double().norm()), float((gr.double() * probe).sum()))
            tol = rtol * float(ref[0]) + atol_scale * gmax
            # the projection of a length-n vector on a unit-variance probe has the scale of the norm
            if abs(got[0] - float(ref[0])) > tol or abs(got[1] - float(ref[1])) > tol * max(1.0, probe.norm().item() / max(gr.numel(), 1) ** 0.5):
                bad.append((n, got, (float(ref[0]), float(ref[1]))))
        assert not bad, f"{self.name}: gradient summaries differ: {bad[:4]} ({len(bad)} parameters)"


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


MODEL_GOLDENS = ["mm_tiny", "mm_tiny_masked", "mm_mid", "mm_odd", "uni_tiny", "uni_mid", "S3", "S2",
                 "tr_tiny", "tr_odd", "S4"]
EPIC_GOLDENS = [n for n in MODEL_GOLDENS if not (n.startswith("tr_") or n == "S4")]
TRANSFORMER_GOLDENS = ["tr_tiny", "tr_odd", "S4"]
