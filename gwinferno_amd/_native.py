"""ctypes binding of include/gwi_engine.h (the C ABI of the HIP engine).

The shared library is built in-tree by ``__graft_entry__.build()`` (``hipcc --offload-arch=gfx950``)
to ``gwinferno_amd/_lib/libgwi_engine.so``.  There is no CPU fallback anywhere in this package:
if the library is missing, or no MI355X is visible, the calls below raise ``NativeEngineError``.
"""
import ctypes as C
import os

import numpy as np

GWI_ABI_VERSION = 3
GWI_MAX_TERMS = 12
GWI_MAX_THETA = 256
GWI_MAX_NORMS = 12
GWI_MAX_COLS = 32

# term kinds (include/gwi_engine.h)
TERM_POWERLAW = 1
TERM_PLPEAK = 2
TERM_POWERLAW_RATIO = 3
TERM_BETA = 4
TERM_TILT_MIXTURE = 5
TERM_POWERLAW_REDSHIFT = 6
TERM_EXP_SPLINE = 7
TERM_TRUNCNORM = 8
TERM_LINEAR_SPLINE = 9
TERM_TILT_JOINT = 10
TERM_SMOOTH = 11
TERM_PLPEAK_SMOOTH = 12
TERM_POWERLAW_BOUNDS = 13
TERM_EXP_SPLINE_LERP = 14

SPLINE_OUTSIDE_ZERO_EXPONENT = 1
RATIO_LOGM_FROM_SPLINE = 8
POWERLAW_UNNORMALISED = 2
NORM_LINEAR_SPLINE = 4
DEVICE_CURRENT = -1
DEVICE_HOST_ONLY = -2

STATUS_NAMES = {0: "GWI_OK", -1: "GWI_ERR_INVALID", -2: "GWI_ERR_NO_DEVICE", -3: "GWI_ERR_HIP", -4: "GWI_ERR_UNSUPPORTED", -5: "GWI_ERR_TIMEOUT"}

_DP = C.POINTER(C.c_double)


class NativeEngineError(RuntimeError):
    pass


class GwiTerm(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("cols", C.c_int32 * 2),
        ("theta", C.c_int32 * 4),
        ("n_basis", C.c_int32),
        ("coef_off", C.c_int32),
        ("flags", C.c_int32),
        ("norm", C.c_int32),
        ("reserved", C.c_int32),
        ("p", C.c_double * 4),
    ]


class GwiNorm(C.Structure):
    _fields_ = [
        ("n_pts", C.c_int32),
        ("expo_theta", C.c_int32),
        ("n_basis", C.c_int32),
        ("coef_off", C.c_int32),
        ("spline_flags", C.c_int32),
        ("reserved", C.c_int32),
        ("expo_add", C.c_double),
        ("lo", C.c_double),
        ("hi", C.c_double),
        ("tw", _DP),
        ("lb", _DP),
        ("l1", _DP),
        ("us", _DP),
    ]


class GwiSpec(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_cols", C.c_int32),
        ("kappa_col", C.c_int32),
        ("n_theta", C.c_int32),
        ("n_terms", C.c_int32),
        ("n_norms", C.c_int32),
        ("vt_norm", C.c_int32),
        ("reserved", C.c_int32),
        ("terms", GwiTerm * GWI_MAX_TERMS),
        ("norms", GwiNorm * GWI_MAX_NORMS),
    ]


class GwiOptions(C.Structure):
    _fields_ = [
        ("n_obs", C.c_double),
        ("total_inj", C.c_double),
        ("marginalize_selection", C.c_int32),
        ("min_neff_cut", C.c_int32),
        ("max_variance_cut", C.c_int32),
        ("reserved", C.c_int32),
    ]


class GwiSummary(C.Structure):
    _fields_ = [
        ("log_likelihood", C.c_double),
        ("log_l", C.c_double),
        ("sum_logBFs", C.c_double),
        ("selection_factor", C.c_double),
        ("log_det_eff", C.c_double),
        ("log_nEff_inj", C.c_double),
        ("variance_log_detection_efficiency", C.c_double),
        ("variance_log_likelihood", C.c_double),
        ("min_log_nEff", C.c_double),
        ("surveyed_hypervolume_norm", C.c_double),
        ("log_norm_const", C.c_double),
        ("reserved", C.c_double * 5),
    ]


class GwiIngestOp(C.Structure):
    _fields_ = [("op", C.c_int32), ("dst", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32), ("reserved", C.c_int32), ("k", C.c_double)]


class GwiIngestProgram(C.Structure):
    _fields_ = [
        ("n_ops", C.c_int32),
        ("n_regs", C.c_int32),
        ("n_sources", C.c_int32),
        ("n_tables", C.c_int32),
        ("ops", C.POINTER(GwiIngestOp)),
        ("sources", C.POINTER(C.c_void_p)),
        ("source_dtype", C.POINTER(C.c_int32)),
        ("tables", C.POINTER(C.POINTER(C.c_double))),
        ("table_len", C.POINTER(C.c_int64)),
    ]


GWI_INGEST_MAX_REGS, GWI_INGEST_MAX_SOURCES, GWI_INGEST_MAX_TABLES = 64, 32, 16
GWI_DTYPE_F64, GWI_DTYPE_F32 = 0, 1


def ingest_program(prog):
    """``gwi_ingest_program`` of a compiled :class:`gwinferno_amd.expr.Program`.  Returns ``(struct, keep)``: ``keep`` holds
    every buffer the struct points to (sources that had to be made contiguous / converted included) and must outlive the call."""
    if len(prog.sources) > GWI_INGEST_MAX_SOURCES or len(prog.tables) > GWI_INGEST_MAX_TABLES or prog.n_regs > GWI_INGEST_MAX_REGS:
        raise ValueError(f"setup program too large for the device evaluator ({len(prog.sources)} sources, {len(prog.tables)} tables, {prog.n_regs} registers)")
    ops = (GwiIngestOp * max(len(prog.ops), 1))()
    for i, (op, dst, a, b, c, k) in enumerate(prog.ops):
        o = ops[i]
        o.op, o.dst, o.a, o.b, o.c, o.k = op, dst, a, b, c, k
    srcs, dtypes = [], []
    for a in prog.sources:
        a = np.asarray(a)
        if a.dtype not in (np.float64, np.float32):  # bool masks, integer columns: converted once
            a = a.astype(np.float64)
        srcs.append(np.ascontiguousarray(a))
        dtypes.append(GWI_DTYPE_F32 if a.dtype == np.float32 else GWI_DTYPE_F64)
    tabs = [np.ascontiguousarray(t, dtype=np.float64) for t in prog.tables]
    src_ptrs = (C.c_void_p * max(len(srcs), 1))(*[a.ctypes.data for a in srcs])
    dt = (C.c_int32 * max(len(srcs), 1))(*dtypes)
    tab_ptrs = (_DP * max(len(tabs), 1))(*[as_dp(t) for t in tabs])
    tab_len = (C.c_int64 * max(len(tabs), 1))(*[t.size for t in tabs])
    st = GwiIngestProgram()
    st.n_ops, st.n_regs, st.n_sources, st.n_tables = len(prog.ops), prog.n_regs, len(srcs), len(tabs)
    st.ops = ops
    st.sources = src_ptrs
    st.source_dtype = dt
    st.tables = tab_ptrs
    st.table_len = tab_len
    return st, (ops, srcs, tabs, src_ptrs, dt, tab_ptrs, tab_len)


class GwiNutsOptions(C.Structure):
    _fields_ = [
        ("n_warmup", C.c_int32),
        ("n_samples", C.c_int32),
        ("max_tree_depth", C.c_int32),
        ("reserved", C.c_int32),
        ("target_accept", C.c_double),
        ("seed", C.c_uint64),
    ]


class GwiNutsResult(C.Structure):
    _fields_ = [
        ("accept_rate", C.c_double),
        ("step_size", C.c_double),
        ("n_evals", C.c_int64),
        ("n_divergent", C.c_int32),
        ("reserved", C.c_int32),
    ]


class GwiParamPrior(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("lo", C.c_double), ("hi", C.c_double), ("sigma", C.c_double)]


class GwiSmoothingPenalty(C.Structure):
    _fields_ = [("offset", C.c_int32), ("count", C.c_int32), ("degree", C.c_int32), ("reserved", C.c_int32), ("tau", C.c_double)]


# include/gwi_sampler.h: int32 (*)(void* user, const double* x, double* log_prob, double* grad)
GWI_TARGET_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
GWI_BATCH_TARGET_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))

LIB_PATH = os.environ.get("GWI_ENGINE_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libgwi_engine.so")

# every symbol include/gwi_engine.h declares
EXPORTED_SYMBOLS = [
    "gwi_create",
    "gwi_create_ingest",
    "gwi_ingest_columns",
    "gwi_read_column",
    "gwi_eval",
    "gwi_eval_begin",
    "gwi_eval_end",
    "gwi_eval_batch",
    "gwi_eval_batch_begin",
    "gwi_eval_batch_end",
    "gwi_eval_sequence",
    "gwi_log_weights",
    "gwi_partial_len",
    "gwi_eval_partial",
    "gwi_prepare_combine",
    "gwi_combine",
    "gwi_comm_unique_id",
    "gwi_comm_init",
    "gwi_eval_sharded",
    "gwi_shm_comm_init",
    "gwi_shm_comm_unlink",
    "gwi_shm_exchange",
    "gwi_eval_latencies",
    "gwi_two_pass_repeats",
    "gwi_batch_path",
    "gwi_batch_calibration",
    "gwi_batch_kernel_note",
    "gwi_selftime",
    "gwi_last_kernel_ms",
    "gwi_set_timing",
    "gwi_launch_geometry",
    "gwi_dispatch_info",
    "gwi_pin_thread_to_engine",
    "gwi_pin_thread_to_device",
    "gwi_hbm_bandwidth",
    "gwi_last_error",
    "gwi_destroy",
    "gwi_abi_version",
    "gwi_kernel_variants",
    "gwi_kernel_variant_name",
    "gwi_scan_kernel_name",
    "gwi_jit_compile",
    "gwi_jit_info",
]
# ... and include/gwi_sampler.h
EXPORTED_SYMBOLS += ["gwi_nuts_run", "gwi_nuts_engine", "gwi_nuts_run_lockstep", "gwi_nuts_engine_lockstep", "gwi_nuts_lockstep_stats", "gwi_nuts_run_queue", "gwi_nuts_engine_queue"]

_lib = None


def load_library():
    """dlopen the engine; raises NativeEngineError (never silently degrades) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64 (same soname as the
    # system one).  If torch is importable, load it FIRST so that the engine binds to the runtime
    # torch.distributed / RCCL will use; otherwise a later `import torch` would bring in a second,
    # disjoint runtime and the in-engine RCCL exchange (gwi_comm_init) could not see our buffers.
    # Kernel arguments in device memory: the scan kernel reads ~3 KB of them with scalar loads at wave start;
    # with host-resident kernargs (HIP_FORCE_DEV_KERNARG=0) the config-2 scan takes 13-15 us instead of 8
    # (measured).  Device placement is the runtime's default on this ROCm; make it explicit, before HIP starts.
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for single-GPU use
        pass
    if not os.path.exists(LIB_PATH):
        raise NativeEngineError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). gwinferno_amd has no CPU fallback."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the machine
        raise NativeEngineError(f"cannot load {LIB_PATH}: {exc}") from exc
    vp = C.c_void_p
    lib.gwi_create.restype = C.c_int32
    lib.gwi_create.argtypes = [C.POINTER(GwiSpec), C.POINTER(_DP), C.c_int64, C.c_int64, C.POINTER(_DP), C.c_int64, C.c_int32, C.POINTER(vp)]
    if hasattr(lib, "gwi_create_ingest"):  # absent from older builds loaded through GWI_ENGINE_LIB for A/B timing
        lib.gwi_create_ingest.restype = C.c_int32
        lib.gwi_create_ingest.argtypes = [C.POINTER(GwiSpec), C.POINTER(GwiIngestProgram), C.c_int64, C.c_int64, C.POINTER(GwiIngestProgram), C.c_int64, C.c_int32, C.POINTER(vp)]
        lib.gwi_ingest_columns.restype = C.c_int32
        lib.gwi_ingest_columns.argtypes = [C.POINTER(GwiIngestProgram), C.c_int64, C.c_int32, C.POINTER(_DP), C.c_int32]
        lib.gwi_read_column.restype = C.c_int32
        lib.gwi_read_column.argtypes = [vp, C.c_int32, C.c_int32, _DP]
    lib.gwi_eval.restype = C.c_int32
    lib.gwi_eval.argtypes = [vp, _DP, C.POINTER(GwiOptions), C.POINTER(GwiSummary), _DP, _DP, _DP, _DP, _DP]
    lib.gwi_eval_begin.restype = C.c_int32
    lib.gwi_eval_begin.argtypes = [vp, _DP, C.POINTER(GwiOptions), C.c_int32]
    lib.gwi_eval_end.restype = C.c_int32
    lib.gwi_eval_end.argtypes = [vp, C.POINTER(GwiSummary), _DP, _DP, _DP, _DP, _DP]
    lib.gwi_eval_batch.restype = C.c_int32
    lib.gwi_eval_batch.argtypes = [vp, _DP, C.c_int32, C.POINTER(GwiOptions), C.POINTER(GwiSummary), _DP, _DP, _DP, _DP, _DP]
    if hasattr(lib, "gwi_eval_batch_begin"):
        lib.gwi_eval_batch_begin.restype = C.c_int32
        lib.gwi_eval_batch_begin.argtypes = [vp, _DP, C.c_int32, C.POINTER(GwiOptions), C.c_int32, C.c_int32]
        lib.gwi_eval_batch_end.restype = C.c_int32
        lib.gwi_eval_batch_end.argtypes = [vp, C.POINTER(GwiSummary), _DP, _DP, _DP, _DP, _DP]
    lib.gwi_eval_sequence.restype = C.c_int32
    lib.gwi_eval_sequence.argtypes = [vp, _DP, C.c_int32, C.POINTER(GwiOptions), _DP, _DP, C.c_int32, C.POINTER(C.c_float)]
    lib.gwi_log_weights.restype = C.c_int32
    lib.gwi_log_weights.argtypes = [vp, _DP, _DP, _DP]
    lib.gwi_partial_len.restype = C.c_int64
    lib.gwi_partial_len.argtypes = [vp]
    lib.gwi_eval_partial.restype = C.c_int32
    lib.gwi_eval_partial.argtypes = [vp, _DP, _DP, _DP, _DP, _DP]
    lib.gwi_prepare_combine.restype = C.c_int32
    lib.gwi_prepare_combine.argtypes = [vp, _DP]
    lib.gwi_combine.restype = C.c_int32
    lib.gwi_combine.argtypes = [vp, _DP, C.c_int32, C.POINTER(GwiOptions), C.POINTER(GwiSummary), _DP, _DP]
    lib.gwi_comm_unique_id.restype = C.c_int32
    lib.gwi_comm_unique_id.argtypes = [C.c_char_p, C.c_void_p]
    lib.gwi_comm_init.restype = C.c_int32
    lib.gwi_comm_init.argtypes = [vp, C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
    lib.gwi_eval_sharded.restype = C.c_int32
    lib.gwi_eval_sharded.argtypes = [vp, _DP, C.POINTER(GwiOptions), C.POINTER(GwiSummary), _DP, _DP, _DP, _DP, _DP]
    lib.gwi_shm_comm_init.restype = C.c_int32
    lib.gwi_shm_comm_init.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32]
    lib.gwi_shm_comm_unlink.restype = C.c_int32
    lib.gwi_shm_comm_unlink.argtypes = [C.c_char_p]
    lib.gwi_shm_exchange.restype = C.c_int32
    lib.gwi_shm_exchange.argtypes = [vp, _DP, _DP]
    lib.gwi_eval_latencies.restype = C.c_int32
    lib.gwi_eval_latencies.argtypes = [vp, _DP, C.c_int32, C.POINTER(GwiOptions), _DP]
    lib.gwi_batch_path.restype = C.c_char_p
    lib.gwi_batch_path.argtypes = [vp, C.c_int32]
    lib.gwi_two_pass_repeats.restype = C.c_int64
    lib.gwi_two_pass_repeats.argtypes = [vp]
    lib.gwi_selftime.restype = C.c_int32
    lib.gwi_selftime.argtypes = [vp, _DP, C.POINTER(GwiOptions), C.c_int32, _DP]
    lib.gwi_last_kernel_ms.restype = C.c_int32
    lib.gwi_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    if hasattr(lib, "gwi_launch_geometry"):
        lib.gwi_launch_geometry.restype = C.c_int32
        lib.gwi_launch_geometry.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.gwi_set_timing.restype = C.c_int32
    lib.gwi_set_timing.argtypes = [vp, C.c_int32]
    lib.gwi_pin_thread_to_engine.restype = C.c_int32
    lib.gwi_pin_thread_to_engine.argtypes = [vp]
    lib.gwi_pin_thread_to_device.restype = C.c_int32
    lib.gwi_pin_thread_to_device.argtypes = [C.c_int32]
    lib.gwi_hbm_bandwidth.restype = C.c_int32
    lib.gwi_hbm_bandwidth.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.gwi_dispatch_info.restype = C.c_char_p
    lib.gwi_dispatch_info.argtypes = [vp]
    lib.gwi_last_error.restype = C.c_char_p
    lib.gwi_last_error.argtypes = [vp]
    lib.gwi_destroy.restype = None
    lib.gwi_destroy.argtypes = [vp]
    lib.gwi_abi_version.restype = C.c_int32
    lib.gwi_kernel_variants.restype = C.c_int32
    lib.gwi_kernel_variant_name.restype = C.c_char_p
    lib.gwi_kernel_variant_name.argtypes = [C.c_int32]
    if hasattr(lib, "gwi_scan_kernel_name"):  # absent from older builds loaded through GWI_ENGINE_LIB for A/B timing
        lib.gwi_scan_kernel_name.restype = C.c_char_p
        lib.gwi_scan_kernel_name.argtypes = [vp]
    _IP = C.POINTER(C.c_int32)
    if hasattr(lib, "gwi_batch_calibration"):
        lib.gwi_batch_calibration.restype = C.c_int32
        lib.gwi_batch_calibration.argtypes = [vp, _IP, _DP, _DP]
    if hasattr(lib, "gwi_batch_kernel_note"):
        lib.gwi_batch_kernel_note.restype = C.c_char_p
        lib.gwi_batch_kernel_note.argtypes = [vp]
    if hasattr(lib, "gwi_jit_compile"):  # absent from older builds loaded through GWI_ENGINE_LIB for A/B timing
        lib.gwi_jit_compile.restype = C.c_int32
        lib.gwi_jit_compile.argtypes = [_IP, C.c_int32, C.c_int32, C.c_char_p, C.c_int64, _DP, _IP]
        lib.gwi_jit_info.restype = C.c_int32
        lib.gwi_jit_info.argtypes = [vp, _IP, _DP, _IP, C.POINTER(C.c_char_p)]
    lib.gwi_nuts_run.restype = C.c_int32
    lib.gwi_nuts_run.argtypes = [GWI_TARGET_FN, vp, C.c_int32, _DP, C.POINTER(GwiNutsOptions), _DP, _DP, _IP, C.POINTER(GwiNutsResult)]
    lib.gwi_nuts_engine.restype = C.c_int32
    lib.gwi_nuts_engine.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.POINTER(GwiOptions), C.POINTER(GwiParamPrior), C.POINTER(GwiSmoothingPenalty), C.c_int32, _DP,
                                    C.POINTER(GwiNutsOptions), _DP, _DP, _IP, C.POINTER(GwiNutsResult)]
    if hasattr(lib, "gwi_nuts_run_lockstep"):
        lib.gwi_nuts_run_lockstep.restype = C.c_int32
        lib.gwi_nuts_run_lockstep.argtypes = [GWI_BATCH_TARGET_FN, vp, C.c_int32, C.c_int32, _DP, C.POINTER(GwiNutsOptions), _DP, _DP, _IP, C.POINTER(GwiNutsResult)]
        lib.gwi_nuts_lockstep_stats.restype = None
        lib.gwi_nuts_lockstep_stats.argtypes = [_DP]
        lib.gwi_nuts_engine_lockstep.restype = C.c_int32
        lib.gwi_nuts_engine_lockstep.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int32, C.POINTER(GwiOptions), C.POINTER(GwiParamPrior), C.POINTER(GwiSmoothingPenalty),
                                                 C.c_int32, _DP, C.POINTER(GwiNutsOptions), _DP, _DP, _IP, C.POINTER(GwiNutsResult)]
    if hasattr(lib, "gwi_nuts_run_queue"):
        lib.gwi_nuts_run_queue.restype = C.c_int32
        lib.gwi_nuts_run_queue.argtypes = [GWI_BATCH_TARGET_FN, vp, C.c_int32, C.c_int32, C.c_int32, _DP, C.POINTER(GwiNutsOptions), _DP, _DP, _IP, C.POINTER(GwiNutsResult)]
        lib.gwi_nuts_engine_queue.restype = C.c_int32
        lib.gwi_nuts_engine_queue.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(GwiOptions), C.POINTER(GwiParamPrior), C.POINTER(GwiSmoothingPenalty),
                                              C.c_int32, _DP, C.POINTER(GwiNutsOptions), _DP, _DP, _IP, C.POINTER(GwiNutsResult)]
    if lib.gwi_abi_version() != GWI_ABI_VERSION:
        raise NativeEngineError(f"ABI mismatch: library {lib.gwi_abi_version()} vs binding {GWI_ABI_VERSION}")
    _lib = lib
    return lib


def as_dp(arr):
    return arr.ctypes.data_as(_DP) if arr is not None else None


def f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def jit_compile(kinds, samples_per_lane=2):
    """Compile (or find in the disk cache) the scan chain of a term-kind sequence with hipRTC -- no GPU needed
    (include/gwi_engine.h: gwi_jit_compile).  Returns ``{"path", "compile_seconds", "from_cache"}``; raises
    ``NativeEngineError`` with hipRTC's reason when the chain cannot be built."""
    lib = load_library()
    arr = (C.c_int32 * len(kinds))(*[int(k) for k in kinds])
    buf = C.create_string_buffer(4096)
    sec, hit = C.c_double(0.0), C.c_int32(0)
    st = lib.gwi_jit_compile(arr, len(kinds), int(samples_per_lane), buf, len(buf), C.byref(sec), C.byref(hit))
    if st != 0:
        raise NativeEngineError(f"gwi_jit_compile({list(kinds)}): {STATUS_NAMES.get(st, st)}: {buf.value.decode(errors='replace')}")
    return {"path": buf.value.decode(), "compile_seconds": sec.value, "from_cache": bool(hit.value)}
