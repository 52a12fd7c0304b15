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
