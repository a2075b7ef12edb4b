"""Evaluation reductions on the device (reference: JetGeneratorCallback metrics, pipeline/callbacks_generator.py:143-183,
built on datamodules/utils.py JetFeatures + Wassertein1D): jet-level observables of two post-processed samples
(one wave per jet, mmb_jet_observables) and their 1-D Wasserstein distances."""
from mmb_hip import ops


def jet_features(sample):
    """TensorMultiModal (post-processed) -> dict name -> (B,) device tensor"""
    obs = ops.jet_observables(sample.continuous, sample.discrete, sample.mask)
    names = ops.OBSERVABLES
    if sample.continuous is None:
        names = [n for n in names if n.startswith("num") or n == "charge"]
    if sample.discrete is None:
        names = [n for n in names if not n.startswith("num") or n == "numParticles"]
        names = [n for n in names if n not in ("charge", "jet_charge")]
    return {n: obs[:, ops.OBSERVABLES.index(n)] for n in names}


def wasserstein_report(generated, reference, features=None):
    """W1 between the jet-level distributions of a generated and a reference sample -> dict name -> float"""
    a, b = jet_features(generated), jet_features(reference)
    keys = [k for k in a if k in b and (features is None or k in features)]
    vals = {k: ops.wasserstein_1d(a[k], b[k]) for k in keys}
    return {k: float(v) for k, v in vals.items()}
