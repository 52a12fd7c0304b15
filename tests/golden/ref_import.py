"""Import the UNMODIFIED reference hot-path modules from /root/reference under the NumPy-backed
jax/numpyro stand-ins in ./refstub (build container only -- /root/reference does not exist on the
GPU box and nothing in the product imports this file).

Usage:  from ref_import import load_reference; ref = load_reference()
        ref.interpolation.LogXLogYBSpline(...), ref.analysis.hierarchical_likelihood(...), ...
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("GWI_REFERENCE_ROOT", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "gwinferno"))


def load_reference():
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    stub = os.path.join(_HERE, "refstub")
    if stub not in sys.path:
        sys.path.insert(0, stub)
    # Bare package objects: skip gwinferno/__init__.py (it pulls arviz/xarray/h5py through
    # pipeline.utils / preprocess / postprocess, none of which are on the hot path).
    for name, sub in (("gwinferno", "gwinferno"), ("gwinferno.pipeline", "gwinferno/pipeline")):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.__path__ = [os.path.join(REFERENCE_ROOT, sub)]
            sys.modules[name] = mod
    ns = types.SimpleNamespace()
    ns.cosmology = importlib.import_module("gwinferno.cosmology")
    ns.interpolation = importlib.import_module("gwinferno.interpolation")
    ns.distributions = importlib.import_module("gwinferno.distributions")
    ns.parametric = importlib.import_module("gwinferno.models.parametric.parametric")
    ns.single = importlib.import_module("gwinferno.models.bsplines.single")
    ns.separable = importlib.import_module("gwinferno.models.bsplines.separable")
    ns.smoothing = importlib.import_module("gwinferno.models.bsplines.smoothing")
    ns.spline_perturbation = importlib.import_module("gwinferno.models.spline_perturbation")
    ns.analysis = importlib.import_module("gwinferno.pipeline.analysis")
    ns.numpyro_distributions = importlib.import_module("gwinferno.numpyro_distributions")
    ns.numpyro = importlib.import_module("numpyro")
    ns.jnp = importlib.import_module("jax.numpy")
    return ns


def load_postprocess():
    """postprocess/calculations.py (posterior-predictive curves); pure jax.numpy + tqdm."""
    load_reference()
    if "gwinferno.postprocess" not in sys.modules:
        mod = types.ModuleType("gwinferno.postprocess")
        mod.__path__ = [os.path.join(REFERENCE_ROOT, "gwinferno/postprocess")]
        sys.modules["gwinferno.postprocess"] = mod
    return importlib.import_module("gwinferno.postprocess.calculations")


def load_preprocess():
    """The reference's injection-selection and PE-prior helpers (preprocess/selection.py,
    preprocess/data_collection.py) under in-memory h5py / xarray / arviz stand-ins (refstub/)."""
    load_reference()
    if "gwinferno.preprocess" not in sys.modules:
        mod = types.ModuleType("gwinferno.preprocess")
        mod.__path__ = [os.path.join(REFERENCE_ROOT, "gwinferno/preprocess")]
        sys.modules["gwinferno.preprocess"] = mod
    ns = types.SimpleNamespace()
    ns.h5py = importlib.import_module("h5py")
    ns.selection = importlib.import_module("gwinferno.preprocess.selection")
    ns.data_collection = importlib.import_module("gwinferno.preprocess.data_collection")
    return ns
