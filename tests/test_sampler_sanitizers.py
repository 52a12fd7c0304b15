"""CPU: the library's C++ sampler (gwinferno_amd/csrc/gwi_sampler.cpp) compiled on its own with g++ under
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
