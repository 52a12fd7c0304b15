"""CPU: sanitizer builds.  The library's C++ sampler (gwinferno_amd/csrc/gwi_sampler.cpp) compiled on its own with g++ under
AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer (GPU sanitizers are not available on the
MI355X pool; the sampler is pure host code).  tests/native/sampler_driver.cpp stands in for the engine with a Gaussian
log-likelihood and runs the callback entry and the threaded multi-chain entry with every target feature switched on."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags", ["-fsanitize=address,undefined -fno-sanitize-recover=all", "-fsanitize=thread"])
def test_sampler_under_sanitizers(tmp_path, flags):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "sampler_driver")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", *flags.split(), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "gwinferno_amd", "csrc", "gwi_sampler.cpp"), os.path.join(ROOT, "tests", "native", "sampler_driver.cpp"), "-o", exe, "-lpthread"]
    cc = subprocess.run(cmd, capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0 and run.stdout.strip().endswith("OK"), run.stdout[-2000:] + run.stderr[-4000:]


def test_c_oracle_under_address_and_ub_sanitizers(tmp_path):
    """the CPU restatement (oracle/gwpop_oracle.c) under -fsanitize=address,undefined: every golden
    composition family evaluated through the instrumented build (value, gradient, sites; 1 and 4 OpenMP threads) gives
    the bits of the regular build, with no report."""
    import sys

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    libubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(libasan) and os.path.exists(libasan)):
        pytest.skip("libasan not available")
    so = str(tmp_path / "libgwpop_oracle_asan.so")
    cc = subprocess.run(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fopenmp", "-fPIC", "-std=gnu11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-shared",
                         "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "oracle", "gwpop_oracle.c"), "-o", so, "-lm"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    code = """
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from golden_util import GoldenCase
from gwinferno_amd.compositions import COMPOSITIONS
from gwinferno_amd.engine import bind
from oracle.c_oracle import COracle
for name in ("plpeak_full", "bspline_full", "bspline_chieff", "plpeak_smooth", "chm_powerlaw", "chm_bspline"):
    case = GoldenCase(name)
    comp = COMPOSITIONS[case.composition](case.pe, case.inj, mmin=case.meta["mmin"], mmax=case.meta["mmax"])
    p0 = comp.placeholder()
    bm = bind(comp.weights(p0, True), comp.weights(p0, False), comp.hypervolume(p0))
    class E: bound = bm
    comp._engine = E()
    orc = COracle(bm)
    for i in range(case.n_points):
        th = comp.theta(case.point(i))
        a = orc.evaluate(th, case.total_inj, min_neff_cut=False, n_threads=1)
        b = orc.evaluate(th, case.total_inj, min_neff_cut=False, marginalize_selection=True, n_threads=4)
        print(name, i, repr(a["log_likelihood"]), repr(float(np.sum(a["grad"]))), repr(b["log_likelihood"]))
print("DONE")
""" % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    plain = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert plain.returncode == 0 and plain.stdout.strip().endswith("DONE"), plain.stderr[-3000:]
    env_san = dict(env, LD_PRELOAD=libasan + (":" + libubsan if os.path.exists(libubsan) else ""), GWPOP_ORACLE_LIB=so)
    san = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env_san)
    assert san.returncode == 0 and san.stdout.strip().endswith("DONE"), san.stdout[-1000:] + san.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in san.stderr and "runtime error" not in san.stderr, san.stderr[-4000:]
    # -O1 instrumented vs -O3 -march=native regular build: same algorithm, contraction may differ in the last bits
    for la, lb in zip(plain.stdout.splitlines()[:-1], san.stdout.splitlines()[:-1]):
        fa, fb = la.split(), lb.split()
        assert fa[:2] == fb[:2]
        for xa, xb in zip(fa[2:], fb[2:]):
            xa, xb = float(xa), float(xb)
            assert xa == xb or abs(xa - xb) <= 1e-11 * max(1.0, abs(xa)), (la, lb)
