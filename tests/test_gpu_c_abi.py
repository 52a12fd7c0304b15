"""GPU (-m gpu): the C ABI used from plain C (examples/c_abi_example.c) -- no Python, no torch in the process:
gcc compiles the example against include/gwi_engine.h, links libgwi_engine.so and the program checks the
engine's value, per-event log Bayes factors and gradient against a double loop of its own."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_program_through_the_abi(tmp_path):
    libdir = os.path.join(ROOT, "gwinferno_amd", "_lib")
    assert os.path.exists(os.path.join(libdir, "libgwi_engine.so")), "build the engine first (__graft_entry__.build())"
    exe = str(tmp_path / "c_abi_example")
    cc = subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_example.c"), "-o", exe,
                         "-L" + libdir, "-lgwi_engine", "-Wl,-rpath," + libdir, "-lm"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.strip().endswith("OK"), run.stdout + run.stderr


@pytest.mark.gpu
def test_measured_hbm_bandwidth_is_plausible():
    """gwi_hbm_bandwidth (the measured figure reported next to the vendor peak): a read-only sweep and a STREAM
    triad over arrays beyond the Infinity Cache land between a quarter of and just above the 8 TB/s the roofline is
    normalised against; bad arguments are refused."""
    from gwinferno_amd import _native as N
    from gwinferno_amd.engine import hbm_bandwidth

    read, triad = hbm_bandwidth(0, n_doubles=1 << 26, iters=5)
    assert 2000.0 < read < 9000.0 and 2000.0 < triad < 9000.0
    with pytest.raises(N.NativeEngineError):
        hbm_bandwidth(0, n_doubles=8)


def test_creating_and_destroying_engines_leaves_device_memory_where_it_was():
    """gwi_create / gwi_destroy in a loop (host-setup and device-setup paths, a spline model with private knot-coordinate
    copies): the device's free memory after 40 engines is what it was after the first -- every column is ONE allocation
    holding both sample sets, freed once."""
    import numpy as np
    import torch

    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    import gc

    from gwinferno_amd import likelihood

    likelihood.clear_engine_cache()
    gc.collect()  # engines of earlier tests that are still awaiting collection would free their memory in the middle of the loop
    pe, inj, total = make_catalog(12, 2000, 20000, seed=9)
    free = []
    for it in range(41):
        comp = COMPOSITIONS["bspline_iid" if it % 2 else "plpeak"](pe, inj)
        eng = comp.engine(device_setup=bool(it & 2))
        th = comp.theta(draw_params("bspline_iid" if it % 2 else "plpeak", np.random.default_rng(it)))
        assert np.isfinite(eng.evaluate(th, total, min_neff_cut=False).log_likelihood)
        eng.close()
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info()[0])
    assert abs(free[-1] - free[4]) <= 8 << 20, (free[4], free[-1])  # (the runtime's own pools settle in the first few)
