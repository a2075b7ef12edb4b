"""CPU: the C-ABI library loads and exports every symbol include/mmb_hip.h declares, the
ctypes table covers them all, and the product path refuses to run without a HIP device."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "mmb_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mmb_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    import mmb_hip
    syms = declared_symbols()
    assert len(syms) >= 20
    assert os.path.exists(mmb_hip.LIB_PATH), "libmmb_hip.so not built (python __graft_entry__.py build)"
    h = ctypes.CDLL(mmb_hip.LIB_PATH)
    for s in syms:
        assert hasattr(h, s), f"{s} declared in mmb_hip.h but not exported"
        if s != "mmb_last_error":
            assert s in mmb_hip.SIGNATURES, f"{s} has no ctypes signature"
    for s in mmb_hip.SIGNATURES:
        assert s in syms, f"{s} bound in ctypes but not declared in the header"
    assert mmb_hip.lib().mmb_version() >= 100


def test_invalid_arguments_do_not_launch():
    """Argument validation happens on the host before any launch, so it is testable without a GPU."""
    import mmb_hip
    h = mmb_hip.lib()
    rc = h.mmb_euler_step(0, None, None, ctypes.c_float(0.1), None)
    assert rc == -1 and b"mmb_euler_step" in h.mmb_last_error()
    rc = h.mmb_mask_prepare(0, 4, None, 1, None, None, None, None)
    assert rc == -1


def test_no_cpu_fallback():
    from mmb_hip import ops, MMBError
    with pytest.raises(MMBError):
        ops.euler_step_(torch.zeros(8), torch.zeros(8), 0.1)


def test_algorithmic_flops_match_survey():
    """The per-particle GEMM flop counts the roofline uses."""
    import yaml
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    from mmb_hip.fused import algorithmic_flops_per_particle, eligible
    want = {"configs/S3_multimodal.yaml": 574944, "configs/S2_discrete.yaml": 1159424}
    for path, flops in want.items():
        cfg = yaml.safe_load(open(os.path.join(ROOT, path)))
        cfg["data"]["target_path"] = os.path.join(ROOT, "tests", "golden")
        model = MultiModalBridgeMatching(ExperimentConfigs(cfg))
        assert algorithmic_flops_per_particle(model) == flops
    assert eligible(model) is None            # S2 (UniModalEPiC) runs the layered path
