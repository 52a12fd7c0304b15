"""CPU: the C-ABI library builds, loads, and exports every symbol include/gwi_engine.h declares.
No compute entry point is called here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from gwinferno_amd import _native

    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return _native.load_library()


def test_header_symbols_are_exported(lib):
    from gwinferno_amd import _native

    hdr = open(os.path.join(ROOT, "include", "gwi_engine.h")).read() + open(os.path.join(ROOT, "include", "gwi_sampler.h")).read()
    declared = set(re.findall(r"^(?:const )?[a-z_0-9]+\**\s+\**(gwi_[a-z_]+)\s*\(", hdr, flags=re.M))
    assert declared == set(_native.EXPORTED_SYMBOLS), declared ^ set(_native.EXPORTED_SYMBOLS)
    raw = ctypes.CDLL(_native.LIB_PATH)
    for sym in declared:
        assert hasattr(raw, sym), sym


def test_struct_sizes_match_header(lib):
    """ctypes mirrors vs the C layout (sizes computed by hand from include/gwi_engine.h)."""
    from gwinferno_amd import _native as N

    assert ctypes.sizeof(N.GwiTerm) == 4 * 12 + 8 * 4
    assert ctypes.sizeof(N.GwiNorm) == 4 * 6 + 8 * 3 + 8 * 4
    assert ctypes.sizeof(N.GwiSpec) == 4 * 8 + ctypes.sizeof(N.GwiTerm) * N.GWI_MAX_TERMS + ctypes.sizeof(N.GwiNorm) * N.GWI_MAX_NORMS
    assert ctypes.sizeof(N.GwiOptions) == 32
    assert ctypes.sizeof(N.GwiSummary) == 16 * 8
    # include/gwi_sampler.h
    assert ctypes.sizeof(N.GwiNutsOptions) == 4 * 4 + 8 + 8
    assert ctypes.sizeof(N.GwiNutsResult) == 8 + 8 + 8 + 4 + 4
    assert ctypes.sizeof(N.GwiParamPrior) == 4 * 2 + 8 * 3
    assert ctypes.sizeof(N.GwiSmoothingPenalty) == 4 * 4 + 8


def test_version_and_variants(lib):
    assert lib.gwi_abi_version() == 3  # 2: GWI_TERM_PLPEAK takes one column (log x); 3: twelve normalisers, 32 columns
    names = [lib.gwi_kernel_variant_name(i).decode() for i in range(lib.gwi_kernel_variants())]
    for needed in ("plpeak+plq+plz", "plpeak+plq+beta2+tilt2+plz", "plq+plz+spline5", "plz+spline7", "plz+spline3", "pl+plq+plz"):
        assert needed in names
    assert lib.gwi_kernel_variant_name(10_000) is None


def test_no_silent_cpu_fallback(lib):
    """Without a GPU gwi_create must fail with GWI_ERR_NO_DEVICE; the Python layer must raise."""
    import numpy as np

    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    from gwinferno_amd import _native as N
    from gwinferno_amd.compositions import COMPOSITIONS
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, _ = make_catalog(2, 8, 16)
    comp = COMPOSITIONS["plpeak"](pe, inj)
    with pytest.raises(N.NativeEngineError):
        comp.engine()


def test_raw_code_object_for_the_aql_path(lib):
    """build() also produces the device code as a RAW gfx950 code object (what the engine's own AQL queue loads through
    the HSA runtime, gwinferno_amd/csrc/gwi_aql.h): an ELF holding a kernel descriptor for every scan variant and the
    two tail kernels, none of them with scratch or implicit arguments (the AQL path supplies neither)."""
    import shutil
    import subprocess

    from gwinferno_amd import _native

    path = os.path.join(os.path.dirname(_native.LIB_PATH), "gwi_kernels.hsaco")
    if not os.path.exists(path):
        import __graft_entry__ as g

        g.build()
    blob = open(path, "rb").read(4)
    assert blob == b"\x7fELF"
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        readelf = shutil.which("llvm-readelf")
    if not readelf:
        pytest.skip("llvm-readelf not available")
    notes = subprocess.run([readelf, "--notes", path], capture_output=True, text=True).stdout
    names = re.findall(r"\.name:\s+(\S+)", notes)
    n_variants = lib.gwi_kernel_variants()
    scans = [n for n in names if "scan_kernel" in n]
    safe = [n for n in scans if "scan_kernelILb0ELb0ELb1E" in n]  # the two-pass / replay instantiation of the spline term sequences
    # value / log-weight / batched instantiation per compiled term sequence, + the SAFE instantiation of the spline sequences,
    # + the generic (run-time term loop) chain: one SAFE instantiation for every role and its log-weight variant
    generic = [n for n in scans if re.search(r"JLi0EEEEv.*NS_5KArgsE$", n)]  # (scalar head arguments, then the argument block)
    assert len(generic) == 2 and sum(n in safe for n in generic) == 1
    assert len(scans) == 3 * n_variants + (len(safe) - 1) + 2
    assert 0 < len(safe) - 1 < n_variants
    assert any("combine_kernel" in n for n in names) and any("final_kernel" in n for n in names)
    assert "hidden_" not in notes  # no implicit kernel arguments anywhere
    # no scratch in anything the AQL path dispatches: the value-and-gradient scan of every variant and the tail kernels
    per_kernel = dict(zip(names, re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)))
    assert len(per_kernel) == len(names)
    for n, scratch in per_kernel.items():
        if "scan_kernelILb0ELb0E" in n or "combine_kernel" in n or "final_kernel" in n:
            assert scratch == "0", (n, scratch)


def _check_kernarg_warm(path):
    import importlib.util

    spec = importlib.util.spec_from_file_location("check_kernarg_warm", os.path.join(ROOT, "tools", "check_kernarg_warm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.check(path)


def test_hand_issued_argument_loads_are_settled_on_every_path():
    """The scan kernels request the lines of their argument block with s_load_dwords written as asm, which the compiler does not
    track (gwi_device.h: KernargWarm): until the s_waitcnt that settles them nothing may touch a destination register, on ANY
    path.  Round 6: the normaliser workgroups branched away before the wait, the compiler reused a destination for the high half
    of a completion stamp, and now and then a load landed in it -- a ten-second time-out on some boxes.  The disassembly of every
    ahead-of-time kernel is walked (tools/check_kernarg_warm.py)."""
    n, problems = _check_kernarg_warm(os.path.join(os.path.dirname(N_LIB_PATH()), "gwi_kernels.hsaco"))
    assert n > 100 and not problems, problems[:5]


def N_LIB_PATH():
    from gwinferno_amd import _native as N

    return N.LIB_PATH


def test_a_scan_chain_compiles_at_run_time_without_a_gpu(lib, tmp_path):
    """gwi_jit_compile (gwinferno_amd/csrc/gwi_jit.h): the scan template instantiated by hipRTC for a term-kind sequence the
    library has no ahead-of-time kernel for -- cross-compiled for gfx950 like hipcc does, so it runs in the CPU suite.  The
    cache file holds a raw code object with one kernel per role, none with scratch or implicit arguments (the engine's AQL
    queue supplies neither); a second PROCESS finds it in the disk cache; and the chain of a sequence that does have an
    ahead-of-time kernel comes out under the same names with a comparable register budget (the compiler is the process's
    hipRTC -- PyTorch's bundled ROCm when torch was imported first -- not necessarily the hipcc of the build)."""
    import shutil
    import subprocess
    import sys

    from gwinferno_amd import _native as N

    os.environ["GWI_JIT_CACHE"] = str(tmp_path)
    try:
        names = [lib.gwi_kernel_variant_name(i).decode() for i in range(lib.gwi_kernel_variants())]
        assert not any(n.startswith("user:2,3,4,6,8") for n in names)
        got = N.jit_compile([2, 3, 4, 6, 8], 2)  # PL+Peak x PL q x Beta x PL z x truncated normal: parametric -> scan, logw, batch, pbatch
        assert got["path"].startswith(str(tmp_path)) and os.path.exists(got["path"]) and not got["from_cache"] and got["compile_seconds"] > 0
        spl = N.jit_compile([1, 5, 7, 9], 1)    # with spline terms: scan, logw, batch, safe
        with pytest.raises(N.NativeEngineError):
            N.jit_compile([3, 2], 2)             # not ascending
        with pytest.raises(N.NativeEngineError):
            N.jit_compile([2, 99], 2)            # not a term kind
        # a fresh process: the disk cache answers
        code = ("import sys; sys.path.insert(0, %r)\nfrom gwinferno_amd import _native as N\nr = N.jit_compile([2, 3, 4, 6, 8], 2)\n"
                "assert r['from_cache'] and r['compile_seconds'] == 0.0, r\nprint(r['path'])" % ROOT)
        again = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, GWI_JIT_CACHE=str(tmp_path)))
        assert again.returncode == 0 and again.stdout.strip() == got["path"], again.stderr[-2000:]
        aot_like = N.jit_compile([2, 3, 6], 2)   # BASELINE config 2's chain once more, by hipRTC
        mm = N.jit_compile([5, 5, 6, 107, 107], 0)  # the batched matrix-core kernel of a spline model: kind + 100 x gradient tiles
        assert os.path.basename(mm["path"]).startswith("mfma_5-5-6-107-107")
    finally:
        os.environ.pop("GWI_JIT_CACHE", None)
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        readelf = shutil.which("llvm-readelf")
    if not readelf:
        pytest.skip("llvm-readelf not available")

    def kernels(path_or_blob, is_cache_file):
        p = path_or_blob
        if is_cache_file:
            blob = open(path_or_blob, "rb").read()
            head, _, _ = blob.partition(b"\x7fELF")
            lowered = head.decode().splitlines()[1:6]
            p = str(tmp_path / (os.path.basename(path_or_blob) + ".hsaco"))
            open(p, "wb").write(blob[len(head):])
        else:
            lowered = None
        notes = subprocess.run([readelf, "--notes", p], capture_output=True, text=True).stdout
        rows = {}
        for blockm in re.finditer(r"\.name:\s+(\S+)(.*?)\.wavefront_size", notes, flags=re.S):
            rows[blockm.group(1)] = {"scratch": int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blockm.group(0)).group(1)) if re.search(r"\.private_segment_fixed_size", blockm.group(0)) else None}
        per_name = dict(zip(re.findall(r"\.name:\s+(\S+)", notes), zip(re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes), re.findall(r"\.vgpr_count:\s+(\d+)", notes),
                                                                   re.findall(r"\.sgpr_count:\s+(\d+)", notes))))
        return lowered, per_name, notes

    for made in (got, spl, aot_like):  # chains compiled at run time come from the same header: the same check on their code objects
        blob = open(made["path"], "rb").read()
        co = str(tmp_path / (os.path.basename(made["path"]) + ".co"))
        open(co, "wb").write(blob[blob.index(b"\x7fELF"):])
        n_warm, problems = _check_kernarg_warm(co)
        assert n_warm >= 2 and not problems, problems[:5]
    low, per, notes = kernels(got["path"], True)
    assert low[0] and low[1] and low[2] and not low[3] and low[4]          # parametric: no SAFE instantiation, a pbatch one
    assert "scan_pbatch_kernel" in low[4] and "hidden_" not in notes
    for role in (0, 2, 4):
        assert per[low[role]][0] == "0", (low[role], per[low[role]])        # no scratch in what the AQL queue dispatches
    low_m, per_m, _ = kernels(mm["path"], True)
    assert "scan_mfma_kernel" in low_m[0] and not any(low_m[1:]) and per_m[low_m[0]][0] == "0"  # one kernel, no scratch
    low_s, per_s, _ = kernels(spl["path"], True)
    assert low_s[3] and not low_s[4] and per_s[low_s[0]][0] == "0" and per_s[low_s[3]][0] == "0"
    # the same chain from hipcc (ahead of time) and from hipRTC: the same kernels by name, no scratch, registers within a third
    low_c2, per_c2, _ = kernels(aot_like["path"], True)
    _, per_aot, _ = kernels(os.path.join(os.path.dirname(N.LIB_PATH), "gwi_kernels.hsaco"), False)
    for role in (0, 1, 2, 4):
        assert low_c2[role] in per_aot, low_c2[role]
        jit_k, aot_k = per_c2[low_c2[role]], per_aot[low_c2[role]]
        assert jit_k[0] == aot_k[0] == "0" or role == 1, (low_c2[role], jit_k, aot_k)
        assert abs(int(jit_k[1]) - int(aot_k[1])) <= max(16, int(aot_k[1]) // 3), (low_c2[role], jit_k, aot_k)
