#!/usr/bin/env python3
"""bench.py -- log-likelihood evaluations per second of the hierarchical population likelihood.

One *step* = one value-and-gradient evaluation of ``hierarchical_likelihood`` (log_l, d log_l/d theta
and every diagnostic site) for one hyper-parameter point, end to end from a host ``theta`` to host
results, with the catalog already resident in HBM -- what one NUTS leapfrog costs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c1|c3|c5|b50k|def50k] [--full] [--detail PATH]

The LAST line of stdout is ONE compact, strict JSON object (<= 4 KB, `compact_line`): the contract's keys, `roofline`,
`cpu_baseline`, a short entry per further configuration under `configs` and, for N > 1, a short `multi_gpu` block.  Everything
else this script measures goes to the side file (`--detail`, default ./bench_detail.json), never to stdout.  The default run
holds the core legs only (timed K steps, latency distribution, kernel durations, blocking K = 16 batches, bounded CPU
baseline); `--full` adds the secondary legs (sets in flight, the other batched kernels, chains, the library's NUTS, ...).

N = 1: a single engine.  N > 1: one rank per GPU (events and injections sharded across ranks, each rank scans
its shard, ONE exchange of the ~1 KiB partial records -- an ncclAllGather (RCCL over xGMI) inside the engine, with the
shared-memory exchange as the secondary figure and the fallback -- every rank assembles the same result; "strong"
scaling: the BASELINE catalog size is fixed).  Launched plainly (`python bench.py --gpus N`) the script starts its own
ranks -- a fresh `python -m torch.distributed.run` child, before this process has touched a GPU -- and relays
the child's JSON line; under torch.distributed.run it is a rank.  Prints ONE JSON line on rank 0.

The headline block (`value`, `ms_per_step`, `roofline`, `cpu_baseline`) is BASELINE config 2, the configuration
the metric is quoted on; `configs` carries the same measurements for the B-spline configs 3 and 5.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# name -> (composition, catalog, SURVEY 8(d) algorithmic fp64 scalars per sample, description)
CONFIGS = {
    "c1": ("plpeak_full", "c1", 10, "C1: PL+Peak m1 x PL q x Beta spins x iso+aligned tilts x PL z, 10 ev x 1000 PE x 5k inj"),
    "c2": ("plpeak", "c2", 4, "C2: PL+Peak m1 x PL q x PL z, 69 ev x 5000 PE x 50k inj"),
    "c3": ("bspline_iid", "c3", 8, "C3: B-spline m1(30) x PL q x IID spin mag(16) x IID tilt(16) x PL z, 69 ev x 5000 PE x 100k inj"),
    "c5": ("bspline_full", "c5", 9, "C5: B-spline m1(30) q(14) a1,a2(12) ct1,ct2(12) x PL z x spline z(12), 200 ev x 10000 PE x 500k inj"),
    # the size north_star's last sentence names (69 x 5000 x 50k: the catalog of config 2) under B-spline models: config 3's composition, and the
    # reference's DEFAULT spline counts (pipeline/utils.py:29-39, 104-155: m1 50, q 30, a 16+16, tilt 16+16, z 20 = 164 coefficients + lamb)
    "b50k": ("bspline_iid", "c2", 8, "C3's model (B-spline m1(30) x PL q x IID spin mag(16), tilt(16) x PL z) at 69 ev x 5000 PE x 50k inj"),
    "def50k": ("bspline_defaults", "c2", 9, "reference default spline counts: m1(50) q(30) a1,a2(16) ct1,ct2(16) x PL z x spline z(20), 69 ev x 5000 PE x 50k inj"),
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VECTOR_PEAK_TFLOPS = 78.6


# ----------------------------------------------------------------------------------------------------
# --gpus N launched plainly: start the ranks ourselves
# ----------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n, argv):
    """Spawn `python -m torch.distributed.run --nproc-per-node n bench.py ...` as a CHILD (never exec: this process may
    not replace itself once anything has initialised the GPU, and a child keeps that rule trivially true) and relay its
    JSON line.  With fewer than n GPUs visible the ranks share what there is (round robin) and exchange over gloo --
    a logic run of the multi-rank path, flagged in the output."""
    import torch  # counting devices does not initialise the GPU

    n_vis = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n_vis < n:
        env["GWI_BENCH_BACKEND"] = "gloo"
        env["GWI_BENCH_SHARE_DEVICES"] = str(max(n_vis, 1))
        print(f"[bench] {n_vis} GPU(s) visible for --gpus {n}: ranks share them (gloo rendezvous)", file=sys.stderr)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + argv
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
    for ln in child.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return child.returncode if lines or child.returncode else 1


# ----------------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------------
def pmc_table(config):
    """Row of profiles/<latest round>/traffic.json for this config (or {}): HBM bytes and fp64 flops per scan launch from
    separate rocprofv3 --pmc passes (tools/profile_round.sh + tools/summarize_profiles.py; FETCH_SIZE doubled per the
    gfx950 correction of MI355X_MICROARCH.md)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", "traffic.json")))
    for f in reversed(files):
        try:
            row = json.load(open(f)).get(config)
            if row:
                return dict(row, source=os.path.relpath(f, ROOT))
        except Exception:
            pass
    return {}


def read_clocks(device=None):
    """Current shader / memory clock from sysfs (the `*` line of pp_dpm_sclk / pp_dpm_mclk) of the amdgpu card at the
    PCI address of HIP device `device` (the box may show more cards than this process may use), or of every card when
    the address is unknown."""
    want = None
    try:
        import torch

        p = torch.cuda.get_device_properties(device if device is not None else 0)
        want = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}."
    except Exception:
        pass
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        addr = os.path.basename(os.path.realpath(d))
        if want and not addr.startswith(want):
            continue
        row = {"pci": addr}
        for key, fn in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk")):
            try:
                for ln in open(os.path.join(d, fn)):
                    if "*" in ln:
                        row[key] = ln.split(":")[1].replace("*", "").strip()
            except Exception:
                pass
        if len(row) > 1:
            out.append(row)
    return out or None


def device_identity(torch, index):
    """PCI address of HIP device `index` ("domain:bus:device"): what tells two ranks apart that believe they hold different GPUs."""
    p = torch.cuda.get_device_properties(index)
    return f"{getattr(p, 'pci_domain_id', 0):04x}:{getattr(p, 'pci_bus_id', index):02x}:{getattr(p, 'pci_device_id', 0):02x}"


def host_cores():
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 CPU quota of the container, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            avail = max(1, min(avail, int(float(quota) / float(period))))
    except Exception:
        pass
    return avail


def cpu_baseline(comp, comp_name, pe, inj, pool, thetas, total, budget_s=6.0, numpy_reference=False):
    """The C/OpenMP restatement (oracle/gwpop_oracle.c: value + gradient + sites, same flat model description as the GPU
    engine receives) timed on the host cores, plus the NumPy restatement of the REFERENCE formulation (dense Cox-de Boor
    design matrices contracted per evaluation, value only, one core).  Checker code, never the product path."""
    from oracle.c_oracle import COracle

    orc = COracle(comp.engine().bound)
    avail = host_cores()

    def rate(nt, repeats=2):
        """evaluations per second at nt threads: best of `repeats` single evaluations after one untimed"""
        orc.evaluate(thetas[0], total, min_neff_cut=False, n_threads=nt)
        best = None
        for _ in range(repeats):
            t0 = time.perf_counter()
            orc.evaluate(thetas[0], total, min_neff_cut=False, n_threads=nt)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return 1.0 / best

    # thread count = the CPUs this process may use (one OpenMP region over (event, sample-block) pairs and injection
    # blocks keeps them all busy), unless a 3-repeat probe shows that fewer threads are faster; both are reported
    # (`avail` honours the container's CPU quota; the thread count the scheduler affinity alone would allow is probed too, so
    # that the line shows what asking for every visible CPU gives on a box whose quota is smaller)
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = avail
    probe = {nt: rate(nt) for nt in sorted({avail, max(1, avail // 2), min(avail, 16), min(affinity, 256)})}
    # ... and the probe's two best counts are then run SUSTAINED (half the budget each, the smaller count first): a burst of a few
    # evaluations on every visible CPU can beat the cgroup's CPU quota that a run of seconds is throttled to (a 16-CPU quota
    # on a 128-CPU box: 356 evals/s in the probe, 8 sustained)
    sustained = {}
    for nt in sorted(sorted(probe, key=probe.get, reverse=True)[:2]):
        k, t_nt = 0, 0.0
        while t_nt < budget_s / 2.0:
            t0 = time.perf_counter()
            orc.evaluate(thetas[k % len(thetas)], total, min_neff_cut=False, n_threads=nt)
            t_nt += time.perf_counter() - t0
            k += 1
        sustained[nt] = (k, t_nt)
    cores = max(sustained, key=lambda nt: sustained[nt][0] / sustained[nt][1])
    n, t_used = sustained[cores]
    t0 = time.perf_counter()
    orc.evaluate(thetas[0], total, min_neff_cut=False, n_threads=1)
    t_single = time.perf_counter() - t0
    out = {
        "value": n / t_used,
        "unit": "evals/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n} value+gradient evals of the full catalog by the C/OpenMP oracle on {cores} threads ({t_used:.1f}s)",
        "threads_probe_evals_per_s": {str(k): v for k, v in probe.items()},
        "sustained_evals_per_s": {str(nt): k / t for nt, (k, t) in sustained.items()},
        "at_all_usable_cpus": {"cores": avail, "evals_per_s": probe[avail], "what": "CPUs this process may use: scheduler affinity capped by the cgroup CPU quota"},
        "at_affinity_count": {"cores": min(affinity, 256), "evals_per_s": probe[min(affinity, 256)]},
        "parallel_units": "blocks of 512 samples: (event, block) pairs and injection blocks in ONE OpenMP region, merged in block order (oracle/gwpop_oracle.c)",
        "single_thread_evals_per_s": 1.0 / t_single,
        "host": host_description(),
    }
    if numpy_reference:
        # reference formulation (interpolation.py:304 einsum over a dense (N_basis, N) matrix), NumPy, one core, value only
        try:
            from oracle import numpy_oracle as O

            t0 = time.perf_counter()
            ref = O.COMPOSITIONS[comp_name](pe, inj)  # builds the dense design matrices once (setup, not timed per eval)
            t_setup = time.perf_counter() - t0
            ref.evaluate(pool[0], total, min_neff_cut=False)
            k, t_np = 0, 0.0
            while t_np < 3.0 and k < 50:
                t0 = time.perf_counter()
                ref.evaluate(pool[k % len(pool)], total, min_neff_cut=False)
                t_np += time.perf_counter() - t0
                k += 1
            out["numpy_reference_formulation"] = {"evals_per_s": k / t_np, "cores": 1, "value_only": True, "setup_s": t_setup,
                                                 "sample": f"{k} evals by oracle/numpy_oracle.py (dense design matrices, einsum per eval)"}
        except Exception as exc:  # the dense matrices of config 5 need ~2 GB
            out["numpy_reference_formulation"] = {"error": repr(exc)}
    return out


def host_description():
    """CPU model, logical CPUs the box shows and CPUs this process may use (nproc and model next to the baseline)."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = None
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "usable_cpus": usable, "cgroup_cpu_quota": quota}


_MEASURED_PEAK = {}


def measured_peak(dev):
    """{"read_gbs", "triad_gbs"} of GPU ``dev``, measured once per process (gwi_hbm_bandwidth); None if it fails."""
    if dev not in _MEASURED_PEAK:
        try:
            from gwinferno_amd.engine import hbm_bandwidth

            r, t = hbm_bandwidth(dev)
            _MEASURED_PEAK[dev] = {"read_gbs": r, "triad_gbs": t, "what": "read-only sweep / STREAM triad over 1 GiB arrays, best of 10 launches, HIP events"}
        except Exception as exc:
            print(f"[bench] HBM bandwidth probe failed: {exc}", file=sys.stderr)
            _MEASURED_PEAK[dev] = None
    return _MEASURED_PEAK[dev]


def percentiles(seconds):
    ms = 1e3 * np.asarray(seconds)
    return {"median_ms_per_step": float(np.median(ms)), "p5_ms": float(np.percentile(ms, 5)), "p95_ms": float(np.percentile(ms, 95)), "n_evals": int(ms.size)}


LINE_LIMIT = 4096  # bytes: the driver reads the last line out of a bounded tail of stdout (round 5's 23 KB line was cut and not parsed)


def strict(x, digits=None):
    """A copy of x that `json.dumps(..., allow_nan=False)` accepts: NaN / +-Infinity -> None, NumPy scalars and arrays -> Python;
    floats rounded to `digits` significant digits when given."""
    if isinstance(x, dict):
        return {str(k): strict(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [strict(v, digits) for v in x]
    if isinstance(x, np.ndarray):
        return strict(x.tolist(), digits)
    if isinstance(x, (bool, np.bool_)):
        return bool(x)
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if not np.isfinite(x):
            return None
        return float(f"{x:.{digits}g}") if digits else x
    return x


def _pick(d, *path):
    for k in path:
        if not isinstance(d, dict) or d.get(k) is None:
            return None
        d = d[k]
    return d


def _roofline_entry(blk):
    r = blk.get("roofline") or {}
    return {"bound": r.get("bound", "hbm"), "kernel": _pick(blk, "scan_chain", "name") or r.get("kernel"), "achieved": r.get("achieved"), "peak": r.get("peak"),
            "unit": r.get("unit", "GB/s"), "frac": r.get("frac"), "traffic": r.get("traffic"), "algorithmic_bytes_per_launch": r.get("algorithmic_bytes_per_launch"),
            "avg_kernel_us": r.get("avg_kernel_us"), "timed_launches": r.get("timed_launches"), "dispatch": r.get("dispatch"),
            "measured_read_peak_gbs": _pick(r, "measured_peak", "read_gbs")}


def _cpu_entry(blk):
    c = blk.get("cpu_baseline")
    if not c:
        return None
    return {"value": c.get("value"), "unit": c.get("unit", "evals/s"), "cores": c.get("cores"), "kind": c.get("kind"), "sample": c.get("sample"),
            "single_thread_evals_per_s": c.get("single_thread_evals_per_s"), "cpu_model": _pick(c, "host", "cpu_model"), "cgroup_cpu_quota": _pick(c, "host", "cgroup_cpu_quota")}


def _batched_entry(blk):
    b = blk.get("batched")
    if not b:
        return None
    out = {"k": b.get("k_batch"), "evals_per_s": b.get("evals_per_s"), "path": b.get("path"), "scan_us": _pick(b, "avg_kernel_us", "scan")}
    flight = {t: _pick(b, "sets_in_flight", f"{t}_one_thread", "evals_per_s") for t in (2, 3)}
    if any(v is not None for v in flight.values()):  # T sets of k points in flight from ONE host thread (gwi_eval_batch_begin / _end)
        out["sets_in_flight_one_thread_evals_per_s"] = {str(t): v for t, v in flight.items() if v is not None}
    if _pick(b, "mfma", "frac_of_78.6") is not None:
        out["mfma_issued_frac"] = _pick(b, "mfma", "frac_of_78.6")
        out["mfma_useful_frac"] = _pick(b, "mfma", "useful_frac_of_78.6")
    return out


def _multi_gpu_entry(blk):
    m = blk.get("multi_gpu")
    if not m:
        return None
    ex = m.get("exchanges") or {}
    out = {"ranks": m.get("ranks"), "rccl_ranks": m.get("rccl_ranks"), "exchange": (m.get("exchange") or "")[:80], "rendezvous_backend": m.get("rendezvous_backend"),
           "devices_shared_between_ranks": m.get("devices_shared_between_ranks"), "sharded_vs_single_gpu": m.get("sharded_vs_single_gpu"),
           "evals_per_s": {k: v.get("evals_per_s") for k, v in ex.items() if isinstance(v, dict) and v.get("evals_per_s") is not None},
           "per_rank_scan_us": [_pick(r, "avg_kernel_us", "scan") for r in (m.get("per_rank") or [])][:8],
           "per_rank_events": [r.get("n_ev") for r in (m.get("per_rank") or [])][:8]}
    probe = ex.get("rccl_allgather_probe") or (ex.get("rccl_allgather") if "child_exit_code" in (ex.get("rccl_allgather") or {}) or "skipped" in (ex.get("rccl_allgather") or {}) else None)
    if probe:
        out["rccl_probe"] = {k: (v[:300] if isinstance(v, str) else v) for k, v in probe.items()
                             if k in ("child_exit_code", "error", "skipped", "probe_clean_on_every_rank", "identical_on_all_ranks", "evals_per_s", "rccl_debug_tail", "unique_devices")}
    return out


def compact_line(detail, detail_path=None):
    """The record the driver keeps: the contract's keys, `roofline` and `cpu_baseline` of the headline configuration, one short
    entry per further configuration, a short `multi_gpu` block for N > 1 -- picked out of the DETAIL dict (what `measure`
    returns per configuration, merged in main()).  Pure: tests/test_bench_line_cpu.py runs it on canned detail."""
    line = {k: detail.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = detail.get("config") or {}
    line["config"] = {k: cfg.get(k) for k in ("workload", "n_events", "n_pe", "n_inj", "n_theta", "parallelism")}
    line["median_ms_per_step"] = detail.get("median_ms_per_step")
    line["roofline"] = _roofline_entry(detail)
    line["cpu_baseline"] = _cpu_entry(detail)
    line["batched"] = _batched_entry(detail)
    cfgs = {}
    for name, blk in (detail.get("configs") or {}).items():
        if not blk:
            continue
        r = blk.get("roofline") or {}
        c = blk.get("cpu_baseline") or {}
        b = _batched_entry(blk) or {}
        cfgs[name] = {"workload": _pick(blk, "config", "workload"), "n_theta": _pick(blk, "config", "n_theta"), "value": blk.get("value"), "ms_per_step": blk.get("ms_per_step"),
                      "steps": blk.get("steps"), "kernel": _pick(blk, "scan_chain", "name"), "scan_us": _pick(r, "avg_kernel_us", "scan"), "frac": r.get("frac"),
                      "algorithmic_bytes_per_launch": r.get("algorithmic_bytes_per_launch"), "traffic": r.get("traffic"), "cpu": c.get("value"), "cpu_cores": c.get("cores"),
                      "batched_k16_evals_per_s": b.get("evals_per_s"), "batched_path": b.get("path"), "batched_scan_us": b.get("scan_us")}
        if b.get("mfma_issued_frac") is not None:
            cfgs[name]["mfma_issued_frac"], cfgs[name]["mfma_useful_frac"] = b["mfma_issued_frac"], b["mfma_useful_frac"]
        mg = _multi_gpu_entry(blk)
        if mg:
            cfgs[name]["multi_gpu"] = {k: mg[k] for k in ("rccl_ranks", "sharded_vs_single_gpu", "evals_per_s")}
    if cfgs:
        line["configs"] = cfgs
    mg = _multi_gpu_entry(detail)
    if mg:
        line["multi_gpu"] = mg
    line["detail"] = detail_path
    line = strict(line, digits=6)
    for name, blk in (line.get("configs") or {}).items():
        line["configs"][name] = {k: v for k, v in blk.items() if v is not None}  # a leg that did not run is absent, not null
    return line


def dump_line(line):
    """Strict JSON on one line; sheds the least important blocks (never the contract's keys, `roofline` or `cpu_baseline`) should it
    ever exceed LINE_LIMIT, and says so."""
    line = dict(line)
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    for victim in ("batched", "configs", "multi_gpu"):
        if len(text.encode()) <= LINE_LIMIT:
            break
        line[victim] = f"dropped: line over {LINE_LIMIT} bytes; see the detail file"
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert "\n" not in text and len(text.encode()) <= LINE_LIMIT
    return text


class Run:
    """Process-wide context: rank, world, device and the torch.distributed group (if any)."""

    def __init__(self, args):
        import torch

        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.shared_devices = None
        force_sharded = os.environ.get("GWI_FORCE_SHARDED") == "1"  # exercise the N>1 code path in one process
        if self.world > 1 or force_sharded:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            # GWI_BENCH_BACKEND=gloo with GWI_BENCH_DEVICE=d (all ranks on device d) or GWI_BENCH_SHARE_DEVICES=n (round
            # robin over n devices): several ranks on fewer GPUs with a CPU-side rendezvous -- a logic run of the multi-rank
            # path on a small box
            backend = os.environ.get("GWI_BENCH_BACKEND", "nccl")
            if "GWI_BENCH_DEVICE" in os.environ:
                self.local_rank = int(os.environ["GWI_BENCH_DEVICE"])
                self.shared_devices = 1
            elif "GWI_BENCH_SHARE_DEVICES" in os.environ:
                self.shared_devices = int(os.environ["GWI_BENCH_SHARE_DEVICES"])
                self.local_rank = self.local_rank % self.shared_devices
            elif torch.cuda.is_available() and 0 < torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", self.world)):
                # more ranks than GPUs on this node (a launcher asked for N ranks on a smaller box): share the devices round robin
                # with a CPU-side rendezvous instead of failing on an invalid device ordinal -- a logic run, reported as such
                self.shared_devices = torch.cuda.device_count()
                self.local_rank = self.local_rank % self.shared_devices
                backend = "gloo"
            torch.cuda.set_device(self.local_rank)
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend)
            self.dist = dist
            self.backend = backend
            # one process per GPU means ONE GPU per process: with a GPU per rank every rank must sit on a device of its own (a launcher
            # that hands two ranks the same LOCAL_RANK, or a HIP_VISIBLE_DEVICES that maps two ordinals onto one card, would otherwise
            # time two shards on one GPU and call it scaling)
            self.devices = [None] * self.world
            dist.all_gather_object(self.devices, device_identity(torch, self.local_rank))
            if self.shared_devices is None and len(set(self.devices)) != self.world:
                raise SystemExit(f"[rank {self.rank}] ranks do not hold distinct GPUs: {self.devices}")
        # Bring torch's device context (and, for N > 1, the barrier's communicator) up NOW: their lazy start-up inside the
        # fence just ahead of the timed region would leave the GPU idle for tens to hundreds of milliseconds after the warm-up,
        # long enough for its clocks to drop, and the first timed steps would run at them.
        if torch.cuda.is_available():
            if self.dist is not None:
                self.dist.barrier()
            torch.cuda.synchronize()

    def fence(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_agree(self, ok):
        if self.dist is None:
            return bool(ok)
        t = self.torch.tensor([1 if ok else 0], device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item()) == 1

    def gather_rows(self, row):
        """Every rank's small list of floats -> list over ranks (rank 0's view is what gets printed)."""
        if self.dist is None:
            return [row]
        box = [None] * self.world
        self.dist.all_gather_object(box, row)
        return box


def measure(run, cfg, steps, warmup, timing_every, spin_s=0.6, headline=False, with_cpu=True, k_batch=16, chains=4, nuts_chains=0, prefer_rccl=False, full=False,
            cpu_budget_s=6.0):
    """One configuration: W warm-up + exactly K timed steps (barrier + synchronise on both sides, max over ranks), then
    the latency distribution of >= 1000 further evaluations, kernel durations of >= 20 timed launches, blocking K-point batches
    and the bounded CPU baseline; with `full` also the secondary throughput figures.  Returns the DETAIL dict of this
    configuration (rank 0; `compact_line` picks the record's fields from it) or None."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.engine import pin_thread_to_device
    from gwinferno_amd.synthetic import make_config_catalog

    torch, dist, rank, world, dev = run.torch, run.dist, run.rank, run.world, run.local_rank
    comp_name, cat_name, c_alg, desc = CONFIGS[cfg]
    pe, inj, total = make_config_catalog(cat_name)
    n_ev, n_pe = pe["mass_1"].shape
    n_inj = inj["mass_1"].shape[0]
    comp = COMPOSITIONS[comp_name](pe, inj)
    # the driving thread next to its GPU (numactl-style host placement), before anything pinned is allocated
    pinned = pin_thread_to_device(dev) if os.environ.get("GWI_BENCH_PIN", "1") != "0" and torch.cuda.is_available() else False
    if headline and torch.cuda.is_available():  # HIP start-up, code-object load and the first queue are not catalog setup: pay them on a throw-away engine
        COMPOSITIONS[comp_name](pe, inj).engine(device=dev, rank=rank, world=world).close()
    t_setup = time.perf_counter()
    eng = comp.engine(device=dev, rank=rank, world=world)
    setup = {"device_s": time.perf_counter() - t_setup,
             "what": "model objects + gwi_create_ingest: raw catalog columns up, masks / logs / dVc/dz / kappa by the ingest kernel (SURVEY 8f rank 1)"}
    if full and rank == 0 and world == 1 and torch.cuda.is_available():  # the host path beside it: NumPy evaluates the same setup expressions, gwi_create uploads
        t_host = time.perf_counter()
        host_comp = COMPOSITIONS[comp_name](pe, inj)
        host_comp.engine(device=dev, device_setup=False).close()
        setup["host_s"] = time.perf_counter() - t_host
        del host_comp
    rng = np.random.default_rng(1234)
    pool = [draw_params(comp_name, rng) for _ in range(64)]
    thetas = [comp.theta(p) for p in pool]

    exchange, sharded = None, None
    if dist is not None:
        # hot loop inside the engine (no Python/torch in the data path).  The headline exchange is the one BASELINE.json's
        # north_star names -- ONE ncclAllGather (RCCL over xGMI) of the ~1 KiB partial records on the engine's own stream --
        # whenever the probe of exactly that exchange (a child process per rank with a time limit, run before this process
        # touched a GPU: run_rccl_leg) came back clean on every rank; otherwise, and as the secondary figure next to it, the
        # node's shared memory (publish + poll between host cores, no collective launch).  GWI_BENCH_EXCHANGE=shm|rccl|torch
        # overrides the choice.
        from gwinferno_amd.distributed import ShardedLikelihood, init_engine_communicator, init_shared_memory_exchange

        want = os.environ.get("GWI_BENCH_EXCHANGE", "rccl" if (prefer_rccl and run.backend == "nccl") else "shm")
        if want == "rccl" and run.backend == "nccl":
            try:
                init_engine_communicator(eng)
                ok = True
            except Exception as exc:
                print(f"[rank {rank}] in-engine RCCL communicator unavailable ({exc})", file=sys.stderr)
                ok = False
            if run.all_agree(ok):
                exchange = "ncclAllGather (RCCL over xGMI) on the engine's own stream"
            else:
                want = "shm"
        if exchange is None and want == "shm":
            try:
                init_shared_memory_exchange(eng)
                ok = True
            except Exception as exc:
                print(f"[rank {rank}] shared-memory exchange unavailable ({exc})", file=sys.stderr)
                ok = False
            if run.all_agree(ok):
                exchange = "host shared-memory segment: every rank publishes its record and polls the others' stamps (no collective launch)"
        if exchange is None:
            exchange = "torch.distributed all_gather_into_tensor"
            sharded = ShardedLikelihood(eng, total, device=torch.device("cuda", dev) if run.backend == "nccl" else None)

    if sharded is not None:
        def step(i):
            r = sharded.evaluate(thetas[i % len(thetas)], min_neff_cut=False)
            return r.log_likelihood, r.grad
    else:
        vg = eng.configure(total, min_neff_cut=False)  # value_and_grad(theta) -> (log_likelihood, grad buffer); sharded when a communicator is attached

        def step(i):
            return vg(thetas[i % len(thetas)])

    # N > 1: before timing, rank 0 checks the sharded evaluation against an unsharded engine over the whole catalog on
    # its own GPU (same theta)
    sharded_check = None
    if dist is not None:
        ll_sh, g_sh = step(0)
        g_sh = np.array(g_sh)
        if rank == 0:
            whole = COMPOSITIONS[comp_name](pe, inj)
            eng_full = whole.engine(device=dev)
            r_full = eng_full.evaluate(thetas[0], total, min_neff_cut=False)
            scale = max(1.0, float(np.max(np.abs(r_full.grad))))
            sharded_check = {"log_likelihood_rel_err": abs(ll_sh - r_full.log_likelihood) / max(1e-300, abs(r_full.log_likelihood)),
                             "grad_max_err_over_scale": float(np.max(np.abs(g_sh - r_full.grad))) / scale}
            eng_full.close()
            if sharded_check["log_likelihood_rel_err"] > 1e-9 or sharded_check["grad_max_err_over_scale"] > 1e-8:
                print(f"[bench] WARNING: sharded result differs from the single-GPU result: {sharded_check}", file=sys.stderr)

    in_library = sharded is None and os.environ.get("GWI_BENCH_PYTHON_LOOP") != "1"
    seq = np.stack([thetas[i % len(thetas)] for i in range(max(steps, warmup, 64))])
    clocks_before = read_clocks(dev) if rank == 0 else None

    # ---- clocks up: >= spin_s seconds of evaluations whatever --steps / --warmup say (a 20-step run lasts 0.4 ms, during
    # which the GPU would still be at its idle clocks).  Rank 0 decides the count so that all ranks issue the same number.
    n_spin = 0
    if spin_s > 0:
        t0 = time.perf_counter()
        for i in range(32):
            step(i)
        per = (time.perf_counter() - t0) / 32
        n_spin = int(run.max_over_ranks(spin_s / max(per, 1e-7)))
        n_spin = max(0, min(n_spin, 200000))
        if in_library:
            for i in range(0, n_spin, len(seq)):
                eng.evaluate_sequence(seq[: min(len(seq), n_spin - i)], total, min_neff_cut=False)
        else:
            for i in range(n_spin):
                step(i)

    # ---- W untimed warm-up steps, then EXACTLY K timed steps bracketed by barrier + synchronise -----------------------
    # The K steps are K sequential, blocking evaluations at K given points.  The reference runs that loop inside one XLA
    # program (NUTS under jit, examples/utils.py:63-85): no host-language binding between two evaluations.  So does the
    # timed region here: one gwi_eval_sequence call (a C loop of gwi_eval / gwi_eval_sharded, every result back on the
    # host before the next point starts).
    scan_ms, comb_ms, fin_ms = [], [], []
    if in_library:
        if warmup:
            eng.evaluate_sequence(seq[:warmup], total, min_neff_cut=False)
        timed = eng.configure_sequence(seq[:steps], total, min_neff_cut=False)  # buffers and marshalling outside the timed region
        run.fence()
        t0 = time.perf_counter()
        ll_seq, _ = timed()  # ---- the timed region: exactly K steps, one library call ----
        run.fence()
        elapsed = time.perf_counter() - t0
        last_ll = float(ll_seq[-1])
    else:
        for i in range(warmup):
            step(i)
        run.fence()
        t0 = time.perf_counter()
        for i in range(steps):
            res = step(i)
        run.fence()
        elapsed = time.perf_counter() - t0
        last_ll = float(res[0])
    elapsed = run.max_over_ranks(elapsed)

    # ---- latency distribution: >= 1000 evaluations timed one by one inside the library (median, p5/p95)
    n_lat = max(1000, steps)
    lat_seq = np.stack([thetas[i % len(thetas)] for i in range(n_lat)])
    if in_library:
        lat = eng.evaluate_latencies(lat_seq, total, min_neff_cut=False)
    else:
        lat = np.empty(n_lat)
        for i in range(n_lat):
            t0 = time.perf_counter()
            step(i)
            lat[i] = time.perf_counter() - t0
    # ---- kernel durations: every timing_every-th of a further block (>= 20 timed launches), begin/end of each launch
    n_timed = max(20, min(200, steps // max(timing_every, 1)))
    if in_library and timing_every > 0:
        blk = np.stack([thetas[i % len(thetas)] for i in range(n_timed * timing_every)])
        _, _, kms = eng.evaluate_sequence(blk, total, min_neff_cut=False, timing_every=timing_every)
        sel = kms[:, 0] >= 0
        scan_ms, comb_ms, fin_ms = list(kms[sel, 0]), list(kms[sel, 1]), list(kms[sel, 2])
    elif timing_every > 0:
        for i in range(n_timed):
            eng.set_timing(True)
            step(i)
            ms = eng.last_kernel_ms()
            scan_ms.append(ms[0]), comb_ms.append(ms[1]), fin_ms.append(ms[2])
            eng.set_timing(False)
    clocks_after = read_clocks(dev) if rank == 0 else None
    per_rank = run.gather_rows({"rank": rank, "device": dev, "n_ev": int(eng.n_ev), "n_inj": int(eng.n_inj),
                                "avg_kernel_us": {"scan": 1e3 * float(np.mean(scan_ms)) if scan_ms else None, "combine": 1e3 * float(np.mean(comb_ms)) if comb_ms else None,
                                                  "final": 1e3 * float(np.mean(fin_ms)) if fin_ms else None},
                                "median_ms_per_step": float(np.median(lat) * 1e3)})

    # ---- the same loop driven from Python through the allocation-free closure (what a NumPy sampler pays per step)
    python_driven = None
    if full:
        n_py = min(max(steps, 200), 2000)
        for i in range(min(50, n_py)):
            step(i)
        run.fence()
        t0p = time.perf_counter()
        for i in range(n_py):
            step(i)
        run.fence()
        python_driven = n_py / run.max_over_ranks(time.perf_counter() - t0p)

    # ---- N > 1, secondary figures: (a) one independent chain per GPU over the WHOLE catalog (numpyro
    # chain_method="parallel"): no exchange, per-GPU work fixed (weak scaling); (b) the same sharded evaluation with the
    # in-engine ncclAllGather instead of the shared-memory exchange
    shm_side = None
    if dist is not None and exchange.startswith("ncclAllGather") and headline:
        # the same sharded evaluation with the records exchanged through the node's shared memory instead: same procedure
        # (warm-up, K steps between barriers, max over ranks) on an engine of its own
        try:
            comp_shm = COMPOSITIONS[comp_name](pe, inj)
            eng_shm = comp_shm.engine(device=dev, rank=rank, world=world)
            init_shared_memory_exchange(eng_shm)
            ok = True
        except Exception as exc:
            print(f"[rank {rank}] shared-memory exchange unavailable ({exc})", file=sys.stderr)
            ok = False
        if run.all_agree(ok):
            n_shm = max(200, min(steps, 2000))
            blk = np.stack([thetas[i % len(thetas)] for i in range(n_shm)])
            eng_shm.evaluate_sequence(blk[:100], total, min_neff_cut=False)
            timed_shm = eng_shm.configure_sequence(blk, total, min_neff_cut=False)
            run.fence()
            t0s = time.perf_counter()
            ll_shm, _ = timed_shm()
            run.fence()
            ts = run.max_over_ranks(time.perf_counter() - t0s)
            shm_side = {"evals_per_s": n_shm / ts, "ms_per_step": 1e3 * ts / n_shm, "steps": n_shm, "last_log_likelihood": float(ll_shm[-1]),
                        "exchange": "host shared-memory segment: every rank publishes its record and polls the others' stamps (no collective launch)"}
        if ok:
            eng_shm.close()

    replicas = None
    if dist is not None and full:
        rep = COMPOSITIONS[comp_name](pe, inj)
        eng_rep = rep.engine(device=dev)
        n_rep = max(200, steps // 2)
        eng_rep.evaluate_sequence(seq[:50], total, min_neff_cut=False)
        run.fence()
        t0r = time.perf_counter()
        eng_rep.evaluate_sequence(np.stack([thetas[i % len(thetas)] for i in range(n_rep)]), total, min_neff_cut=False)
        run.fence()
        tr = run.max_over_ranks(time.perf_counter() - t0r)
        replicas = {"evals_per_s": world * n_rep / tr, "scaling": "weak", "what": "one independent chain per GPU over the whole catalog, no exchange"}
        eng_rep.close()

    out = None
    if rank == 0:
        scan_us = 1e3 * float(np.mean(scan_ms)) if scan_ms else float("nan")
        alg_bytes = 8.0 * c_alg * (n_ev * n_pe + n_inj) / world  # per launch on one GPU
        achieved = alg_bytes / (scan_us * 1e-6) / 1e9 if scan_ms else float("nan")
        pmc = pmc_table(cfg) if world == 1 else {}
        flop = pmc.get("fp64_flop_per_launch")
        out = {
            "value": steps / elapsed,
            "unit": "evals/s",
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / steps,
            **percentiles(lat),
            "warm_spin": {"seconds_requested": spin_s, "evaluations": n_spin},
            "step_loop": "gwi_eval_sequence: K blocking evaluations in a C loop inside the library" if in_library else "Python loop over the configured closure",
            "python_driven_evals_per_s": python_driven,
            "config": {
                "workload": desc,
                "composition": comp_name,
                "n_events": int(n_ev),
                "n_pe": int(n_pe),
                "n_inj": int(n_inj),
                "n_theta": int(eng.n_theta),
                "flags": "min_neff_cut=False (tests/inference_test.py:185)",
                "parallelism": f"events+injections sharded over {world} rank(s), one exchange of partial records per eval" if dist is not None else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "scan_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # the box's own HBM bandwidth next to the vendor figure: read-only sweep and STREAM triad over
                # 1 GiB arrays (gwi_hbm_bandwidth), and the fraction against the measured read bandwidth
                "measured_peak": measured_peak(dev) if headline else None,
                "traffic": pmc.get("hbm_bytes_per_launch"),
                "traffic_source": pmc.get("source"),
                "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_read_per_launch": float(eng.bytes_per_sample) * (eng.n_ev * eng.n_pe + eng.n_inj),
                "avg_kernel_us": {"scan": scan_us, "combine": 1e3 * float(np.mean(comb_ms)) if comb_ms else None, "final": 1e3 * float(np.mean(fin_ms)) if fin_ms else None},
                "median_scan_us": 1e3 * float(np.median(scan_ms)) if scan_ms else None,
                "timing": (f"kernel begin/end of every {timing_every}th step of a block after the timed region: dispatch timestamps of the engine's AQL queue "
                           "(hsa_amd_profiling_get_dispatch_time, what rocprofv3's kernel trace reports)" if eng.dispatch_info() == "aql: active" else
                           f"HIP start/stop events attached to each launch (hipExtLaunchKernelGGL) on the engine's stream, every {timing_every}th step of a block after the timed region"),
                "dispatch": eng.dispatch_info(),
                "host_thread_pinned_to_gpu_numa_node": bool(pinned),
                "timed_launches": len(scan_ms),
                # second view: the scan is fp64-issue/latency bound, not HBM bound
                "fp64_vector": {"peak_tflops": FP64_VECTOR_PEAK_TFLOPS, "flop_per_launch_pmc": flop,
                                "achieved_tflops": (flop / (scan_us * 1e-6) / 1e12) if (flop and scan_ms) else None},
            },
            "clocks": {"before": clocks_before, "after": clocks_after},
            "last_log_likelihood": last_ll,
            "setup": setup,
            "two_pass_repeats": eng.two_pass_repeats(),
            # which scan kernel ran: an ahead-of-time chain by name, a chain compiled at gwi_create by hipRTC ("jit:..."; the time
            # that cost this process, 0 when the disk cache supplied the code object) or the generic kernel
            "scan_chain": {"name": eng.scan_kernel_name(), **eng.jit_info()},
        }
        if dist is None and headline and full:
            # What the NumPyro seam adds on the host (likelihood._host_callback: the function jax.pure_callback calls once per leapfrog):
            # theta array in -> engine -> packed (summary, per-event sites, gradient) arrays out, without JAX's own dispatch and
            # device synchronisation around the callback, which no image here can measure.
            try:
                from gwinferno_amd import likelihood as L

                flags_cb = dict(marginalize_selection=False, min_neff_cut=False, max_variance_cut=False)
                host_cb = L._host_callback(eng, float(total), float(n_ev), flags_cb)
                for i in range(50):
                    host_cb(thetas[i % len(thetas)])
                n_cb = 1000
                t0c = time.perf_counter()
                for i in range(n_cb):
                    host_cb(thetas[i % len(thetas)])
                dtc = time.perf_counter() - t0c
                tbk = np.stack(thetas[:16])
                for _ in range(5):
                    host_cb(tbk)
                t0c = time.perf_counter()
                for _ in range(60):
                    host_cb(tbk)
                dtb = time.perf_counter() - t0c
                out["numpyro_seam"] = {"host_callback_us_per_eval": 1e6 * dtc / n_cb, "host_callback_evals_per_s": n_cb / dtc,
                                       "host_callback_vectorized_16_chains_us_per_eval": 1e6 * dtb / (60 * 16), "host_callback_vectorized_16_chains_evals_per_s": 60 * 16 / dtb,
                                       "what": ("the host function behind jax.pure_callback (one call per leapfrog; 16 points per call under vmap / chain_method='vectorized'), "
                                                "timed without JAX: an upper bound on what NumPyro NUTS can reach through the seam -- JAX's dispatch and synchronisation "
                                                "around the callback come on top and are unmeasured here (python -m gwinferno_amd.jax_check --nuts N on a box with JAX)")}
            except Exception as exc:
                out["numpyro_seam"] = {"error": repr(exc)}
        if dist is not None:
            key = "shm" if exchange.startswith("host shared-memory") else ("rccl_allgather" if exchange.startswith("ncclAllGather") else "torch_all_gather")
            rccl_ranks = world if (key == "rccl_allgather" or (key == "torch_all_gather" and run.backend == "nccl")) else 0
            out["multi_gpu"] = {"ranks": world, "rccl_ranks": rccl_ranks, "headline_exchange_key": key, "rendezvous_backend": run.backend, "exchange": exchange,
                                "devices_shared_between_ranks": run.shared_devices, "per_rank": per_rank, "sharded_vs_single_gpu": sharded_check,
                                "independent_chains": replicas}
            if shm_side is not None:
                out["multi_gpu"]["shm_side"] = shm_side
        else:
            out["c_loop_us_per_eval"] = 1e6 * eng.selftime(thetas[0], total, n_iter=min(max(steps, 200), 2000), min_neff_cut=False)
        if dist is None and k_batch > 1:
            # secondary number: vectorised chains -- K hyper-points per launch (not the headline `value`, which is one
            # sequential chain)
            K = k_batch
            tb = np.stack(thetas[:K] if len(thetas) >= K else (thetas * K)[:K])
            vgb = eng.configure_batch(K, total, min_neff_cut=False)  # values_and_grads(thetas[K]) -> (log_l[K], grad[K, n_theta])
            for _ in range(10):
                vgb(tb)
            n_b = max(20, min(200, steps // (4 * K)))
            t0 = time.perf_counter()
            for _ in range(n_b):
                vgb(tb)
            dt = time.perf_counter() - t0
            out["batched"] = {"k_batch": K, "evals_per_s": n_b * K / dt, "us_per_eval": 1e6 * dt / (n_b * K), "launch_sets": n_b, "path": eng.batch_path(K)}
            cal = eng.batch_calibration()
            if cal["measured"]:  # spline models: the engine timed both of its batched kernels on its first batched launch and kept the faster
                out["batched"]["path_choice"] = {"how": "measured by the engine on its first batched launch: three evaluation sets of either kernel, host theta -> host results, best of each",
                                                 "mfma_us_per_set": cal["mfma_us"], "taps_us_per_set": cal["taps_us"]}

            # kernel durations of the batched launches (start/stop of each launch, a few timed batches after the loop)
            eng.set_timing(True)
            bk = []
            for _ in range(8):
                vgb(tb)
                bk.append(eng.last_kernel_ms())
            eng.set_timing(False)
            bk = np.asarray(bk)
            out["batched"]["avg_kernel_us"] = {"scan": 1e3 * float(np.mean(bk[:, 0])), "combine": 1e3 * float(np.mean(bk[:, 1])), "final": 1e3 * float(np.mean(bk[:, 2]))}
            if out["batched"]["path"] == "mfma":
                # matrix-core utilisation of the batched scan (north_star: "MFMA utilisation reported against gfx950 peak"):
                # v_mfma_f64_16x16x4 instructions per launch from the SQ_INSTS_MFMA counter pass (profiles/<round>/traffic.json,
                # tools/profile_round.sh; the count is a property of the kernel and the catalog, not of the run) x 2048 flop,
                # over the LIVE duration of the batched scan launch, against the 78.6 TFLOP/s fp64 matrix peak
                row = (pmc_table("batched_k16") or {}).get(f"{cfg}_mfma") if K == 16 else None
                insts = row.get("mfma_instructions_per_launch") if row else None
                scan_s = 1e-3 * float(np.mean(bk[:, 0]))
                tfl = (2048.0 * insts / scan_s / 1e12) if (insts and scan_s > 0) else None
                out["batched"]["mfma"] = {"instruction": "v_mfma_f64_16x16x4_f64", "insts_per_launch": insts, "flop_per_launch": 2048.0 * insts if insts else None, "tflops": tfl,
                                          "peak_tflops": FP64_VECTOR_PEAK_TFLOPS, "frac_of_78.6": (tfl / FP64_VECTOR_PEAK_TFLOPS) if tfl else None,
                                          # ISSUED flops: the A operand is a 16 x 4 slab with 4 non-zero taps per column, so 12 of every 16 products multiply zeros
                                          "useful_frac_of_78.6": (0.25 * tfl / FP64_VECTOR_PEAK_TFLOPS) if tfl else None,
                                          "useful_note": "frac_of_78.6 counts issued matrix-core flops; a quarter of them (4 of 16 rows per column) carry taps",
                                          "source": (pmc_table("batched_k16") or {}).get("source"),
                                          "what": "the design-matrix contraction sum_s w_s B_p(x_s) for 16 hyper-parameter points per wavefront (gwi_mfma.h); A = 16 bases x 4 samples with 4 non-zero taps per column"}
            # several sets in flight (full: from T host threads and from one; default run: the one-thread form for the headline only)
            if full or headline:
                try:
                    out["batched"]["sets_in_flight"] = sets_in_flight(eng, comp_name, pe, inj, total, tb, K, steps, dev, threaded=full)
                except Exception as exc:  # a secondary figure: never the reason a line is missing
                    out["batched"]["sets_in_flight"] = {"error": repr(exc)}
            # the other batched kernels on the same batch, where the model has them (spline models): the 4-tap kernel (one
            # grid row per point, LDS atomics) and the LDS-row variant of the 16-points-per-wavefront kernel
            alts = {}
            for alt_name, env_kv in ((("taps", {"GWI_BATCH_MFMA": "0"}), ("mfma", {"GWI_BATCH_MFMA": "1"}), ("rows", {"GWI_BATCH_ROWS": "1"})) if full else ()):
                saved = {k_: os.environ.get(k_) for k_ in env_kv}
                os.environ.update(env_kv)
                try:
                    alt = COMPOSITIONS[comp_name](pe, inj).engine(device=dev)
                    if alt.batch_path(K) == alt_name and alt_name != out["batched"]["path"]:
                        vgm = alt.configure_batch(K, total, min_neff_cut=False)
                        for _ in range(10):
                            vgm(tb)
                        t0 = time.perf_counter()
                        for _ in range(n_b):
                            vgm(tb)
                        dtm = time.perf_counter() - t0
                        alts[alt_name] = {"evals_per_s": n_b * K / dtm, "us_per_eval": 1e6 * dtm / (n_b * K)}
                    alt.close()
                finally:
                    for k_, v_ in saved.items():
                        if v_ is None:
                            os.environ.pop(k_, None)
                        else:
                            os.environ[k_] = v_
            if alts:
                out["batched"]["other_paths"] = alts
                out["batched"]["paths"] = ("taps: one grid row per point, 4-tap gradient into LDS rows (scan_kernel BATCH); mfma: 16 points per wavefront, gradient as "
                                           "v_mfma_f64_16x16x4 tiles (gwi_mfma.h); rows: the same kernel with the gradient in conflict-free LDS rows")
        if dist is None and headline and chains > 1 and full:
            out.update(multi_chain(eng, comp, comp_name, pe, inj, total, thetas, chains, steps, dev))
        elif dist is None and nuts_chains > 1 and full:
            # configs 3 / 5: the sampler figure under the reference's priors, on engines of its own
            comps = [comp] + [COMPOSITIONS[comp_name](pe, inj) for _ in range(nuts_chains - 1)]
            engs = [eng] + [c.engine(device=dev) for c in comps[1:]]
            out["native_nuts"] = native_nuts(engs, comp_name, comp, total, thetas, **({"n_warmup": 300, "n_samples": 100, "convergence_fields": False} if cfg == "c5" else {}))
            try:  # the same sampler with its chains in lock step on the batched kernels (a secondary figure)
                out["native_nuts_lockstep"] = (native_nuts_lockstep(engs[:1], 16, comp_name, comp, total, thetas, 30, 10) if cfg == "c5" else
                                               native_nuts_lockstep(engs[:2], 16, comp_name, comp, total, thetas, 60, 30))
            except Exception as exc:
                out["native_nuts_lockstep"] = {"error": repr(exc)}
            for e in engs[1:]:
                e.close()
        if with_cpu and world == 1:  # reported at N = 1 only (rank 0), as the contract asks
            out["cpu_baseline"] = cpu_baseline(comp, comp_name, pe, inj, pool, thetas, total, budget_s=cpu_budget_s, numpy_reference=full and cfg != "c5")
    if dist is not None:
        dist.barrier()
    eng.close()
    return out


def sets_in_flight(eng, comp_name, pe, inj, total, tb, K, steps, dev, threaded):
    """A blocking batch leaves the GPU to its combine / final launches and the host for a third of its time; with several sets of
    K points in flight the other sets' scans fill it.  `threaded`: T host threads, each driving an engine of its own with blocking
    sets; always: T sets in flight from ONE host thread (gwi_eval_batch_begin on set i + 1 before gwi_eval_batch_end on set i, an
    engine per set).  The blocking figure stays the one quoted as `batched.evals_per_s`."""
    import threading

    from gwinferno_amd.compositions import COMPOSITIONS

    extra_comps = [COMPOSITIONS[comp_name](pe, inj) for _ in range(2)]
    set_engs = [eng] + [c.engine(device=dev) for c in extra_comps]
    n_o = max(40, min(400, steps // (2 * K)))
    over = {}
    try:
        if threaded:
            fns = [e.configure_batch(K, total, min_neff_cut=False) for e in set_engs]
            for T in (2, 3):
                gate = threading.Barrier(T + 1)

                def drive(f):
                    for _ in range(10):
                        f(tb)
                    gate.wait()
                    for _ in range(n_o):
                        f(tb)
                    gate.wait()

                ws = [threading.Thread(target=drive, args=(fns[i],)) for i in range(T)]
                for w in ws:
                    w.start()
                gate.wait()
                t0o = time.perf_counter()
                gate.wait()
                dto = time.perf_counter() - t0o
                for w in ws:
                    w.join()
                over[str(T)] = {"evals_per_s": T * n_o * K / dto, "us_per_eval": 1e6 * dto / (T * n_o * K)}
        for T in (2, 3):
            hv = [e.configure_batch_async(K, total, min_neff_cut=False) for e in set_engs[:T]]
            for rep in range(2):  # first lap untimed
                n_1 = 30 if rep == 0 else n_o
                t0 = time.perf_counter()
                for j in range(T - 1):
                    hv[j][0](tb)
                for it in range(n_1 * T):
                    hv[(it + T - 1) % T][0](tb)
                    hv[it % T][1]()
                for j in range(T - 1):
                    hv[(n_1 * T + j) % T][1]()
                dto = time.perf_counter() - t0
            n_sets = n_o * T + T - 1
            over[f"{T}_one_thread"] = {"evals_per_s": n_sets * K / dto, "us_per_eval": 1e6 * dto / (n_sets * K)}
    finally:
        for e in set_engs[1:]:
            e.close()
    return dict(over, what=f"T blocking sets of {K} points side by side, one host thread and one engine per set (same catalog, same GPU); 'T_one_thread': T sets in flight "
                           "from ONE thread through gwi_eval_batch_begin / gwi_eval_batch_end")


def multi_chain(eng, eng_comp, comp_name, pe, inj, total, thetas, C, steps, dev):
    """Secondary numbers (N = 1): C independent chains on the one GPU, each with its own engine."""
    import threading

    from gwinferno_amd.compositions import COMPOSITIONS

    out = {}
    extra = [COMPOSITIONS[comp_name](pe, inj) for _ in range(C - 1)]
    # (a) evaluations in flight together from ONE host thread (gwi_eval_begin / gwi_eval_end); no lock step, every chain
    # follows its own theta sequence
    pairs = [e.configure_async(total, min_neff_cut=False) for e in [eng] + [c.engine(device=dev) for c in extra]]
    n_c = max(200, min(2000, steps // 2))
    for rep in range(2):
        for b, _ in pairs:
            b(thetas[0])
        t0 = time.perf_counter()
        for i in range(n_c):
            for c, (b, e) in enumerate(pairs):
                e()
                b(thetas[(i + c) % len(thetas)])
        for _, e in pairs:
            e()
        dt = time.perf_counter() - t0
    out["interleaved_chains"] = {"chains": C, "evals_per_s": C * n_c / dt, "us_per_eval": 1e6 * dt / (C * n_c), "host_threads": 1}
    # (b) the same C engines, one HOST THREAD each running blocking evaluations in a C loop (the GIL is released inside
    # the library): launch costs spread over cores, the GPU interleaves the chains' kernels
    all_engines = [eng] + [c.engine() for c in extra]
    gate = threading.Barrier(C + 1)

    def chain(e, th):
        e.selftime(th, total, n_iter=50, min_neff_cut=False)
        gate.wait()
        e.selftime(th, total, n_iter=n_c, min_neff_cut=False)
        gate.wait()

    workers = [threading.Thread(target=chain, args=(e, thetas[i])) for i, e in enumerate(all_engines)]
    for w in workers:
        w.start()
    gate.wait()
    t0 = time.perf_counter()
    gate.wait()
    dt = time.perf_counter() - t0
    for w in workers:
        w.join()
    out["threaded_chains"] = {"chains": C, "host_threads": C, "evals_per_s": C * n_c / dt, "us_per_eval": 1e6 * dt / (C * n_c)}
    # ... and with twice as many chains: how much of the GPU one chain's kernels leave idle (a single scan keeps three waves per
    # SIMD busy; the kernels of other chains fill the gaps until vector issue saturates)
    more = [COMPOSITIONS[comp_name](pe, inj) for _ in range(C)]
    engines2 = all_engines + [c.engine() for c in more]
    gate2 = threading.Barrier(2 * C + 1)

    def chain2(e, th):
        e.selftime(th, total, n_iter=50, min_neff_cut=False)
        gate2.wait()
        e.selftime(th, total, n_iter=n_c, min_neff_cut=False)
        gate2.wait()

    workers = [threading.Thread(target=chain2, args=(e, thetas[i])) for i, e in enumerate(engines2)]
    for w in workers:
        w.start()
    gate2.wait()
    t0 = time.perf_counter()
    gate2.wait()
    dt2 = time.perf_counter() - t0
    for w in workers:
        w.join()
    out["threaded_chains"]["with_twice_the_chains"] = {"chains": 2 * C, "host_threads": 2 * C, "evals_per_s": 2 * C * n_c / dt2, "us_per_eval": 1e6 * dt2 / (2 * C * n_c)}
    for c in more:
        c.engine().close()
    out["native_nuts"] = native_nuts(all_engines, comp_name, eng_comp, total, thetas)
    try:  # the same sampler with its chains in lock step on the batched kernel: a queue of 64 chains over two groups of 16 slots, one host thread
        out["native_nuts_lockstep"] = native_nuts_lockstep(all_engines[:2], 16, comp_name, eng_comp, total, thetas, 100, 50, n_chains=64)
    except Exception as exc:
        out["native_nuts_lockstep"] = {"error": repr(exc)}
    for c in extra:
        c.engine().close()
    return out


def param_slices(comp):
    """{parameter name: slice of the flat theta} of a composition (contiguous by construction of the layout)."""
    idx = {}
    for slot, (name, _) in enumerate(comp._theta_map()):
        if name is not None:
            idx.setdefault(name, []).append(slot)
    return {name: slice(min(v), max(v) + 1) for name, v in idx.items()}


def reference_priors(comp_name, comp, n_theta):
    """The priors a reference run puts on this composition's hyper-parameters, for the library's sampler
    (``GaussianSmoothingPrior`` + ``Bijector``).  B-spline models: Normal(0, 15) on the mass coefficients, Normal(0, 5) on
    mass-ratio and spin coefficients, Normal(0, 1) on the redshift coefficients (the first pinned to 0), each with its
    P-spline difference penalty (pipeline/utils.py:163-216 with the tau of examples/simple_bspline_example.py:47-56:
    m 1, q 1, spins 25, z 1), lamb ~ Normal(0, 3) (:56); beta ~ Normal(0, 5) (examples/simple_powerlaw_peak_example.py:52).
    Parametric models: wide Normals (the bounded sites of that example need its bijectors; flat enough not to matter)."""
    from gwinferno_amd.pipeline_utils import bspline_example_prior
    from gwinferno_amd.sampling import Bijector, GaussianSmoothingPrior

    sl = param_slices(comp)
    if comp_name in ("bspline_full", "bspline_defaults"):
        prior, bij = bspline_example_prior({"m1": sl["m1_coefs"], "q": sl["q_coefs"], "a1": sl["a1_coefs"], "a2": sl["a2_coefs"], "tilt1": sl["t1_coefs"], "tilt2": sl["t2_coefs"],
                                            "redshift": sl["z_coefs"], "lamb": sl["lamb"]})
        return prior, bij, "pipeline/utils.py:163-216 as used at examples/simple_bspline_example.py:47-56: N(0,15) m1, N(0,5) q / spins, N(0,1) z, P-spline penalties tau 1/1/25/25/1, lamb N(0,3)"
    if comp_name == "bspline_iid":
        prior, bij = GaussianSmoothingPrior(n_theta), Bijector(n_theta)
        prior.normal(sl["m1_coefs"], 15.0).smoothing(sl["m1_coefs"], 1.0, 1)   # bspline_mass_prior(m_nsplines, m_tau=1)
        for key in ("a_coefs", "t_coefs"):                                       # bspline_spin_prior(IID=True, a_tau=25, ct_tau=25)
            prior.normal(sl[key], 5.0).smoothing(sl[key], 25.0, 2)
        prior.normal(sl["beta"], 5.0).normal(sl["lamb"], 3.0)
        return prior, bij, "pipeline/utils.py:163-208 (IID spins): N(0,15) m1 tau 1, N(0,5) spins tau 25 degree 2; beta N(0,5), lamb N(0,3)"
    if comp_name == "plpeak":
        # examples/simple_powerlaw_peak_example.py:52-56, :77: beta, alpha ~ Normal(0, 5); mu_peak ~ Uniform(mmin, mmax);
        # sig_peak ~ HalfNormal(10); lambda_m ~ Uniform(0, 1); lamb ~ Normal(0, 5) -- sampled through the bijections numpyro's
        # biject_to(support) applies to those sites (logistic map onto the interval, exp onto the positive axis)
        prior, bij = GaussianSmoothingPrior(n_theta), Bijector(n_theta)
        prior.normal(sl["alpha"], 5.0).normal(sl["beta"], 5.0).normal(sl["sigpp"], 10.0).normal(sl["lamb"], 5.0)
        bij.interval(sl["mpp"], float(comp.mmin), float(comp.mmax)).positive(sl["sigpp"]).interval(sl["lam"], 0.0, 1.0)
        return prior, bij, ("examples/simple_powerlaw_peak_example.py:52-56, :77: alpha, beta ~ N(0,5), mu_peak ~ U(mmin, mmax), sig_peak ~ HalfNormal(10), "
                            "lambda_m ~ U(0,1), lamb ~ N(0,5), through numpyro's support bijections")
    return GaussianSmoothingPrior(n_theta).normal(slice(0, n_theta), 10.0), None, "Normal(0, 10) on every parameter"


def native_nuts(engines, comp_name, comp, total, thetas, n_warmup=300, n_samples=200, convergence_fields=True):
    """The engines inside a sampler: the library's C++ NUTS (gwi_nuts_engine, include/gwi_sampler.h), one chain per engine
    and host thread, under the priors a reference run uses (reference_priors), with the reference's tree depth
    (numpyro's default max_tree_depth = 10: up to 1023 leapfrogs per iteration) and a warm-up long enough for the step size
    and the diagonal mass matrix to adapt.  The wall time is bounded through the iteration count (worst case
    (n_warmup + n_samples) x 1023 evaluations per chain), not through the depth.  Reported: likelihood evaluations per
    second as a sampler sees them, how many the engine had to repeat (gwi_two_pass_repeats), and the sampler's own
    diagnostics -- mean tree depth, acceptance rate, adapted step size, divergences after warm-up, and the smallest
    bulk effective sample size over the free parameters per second of sampling (post-warm-up draws of all chains)."""
    from gwinferno_amd.sampling import effective_sample_size, nuts_engine, split_rhat

    C = len(engines)
    prior, bij, what = reference_priors(comp_name, comp, engines[0].n_theta)
    starts = np.stack(thetas[:C])
    if bij is not None:
        for k in np.flatnonzero(bij.kind == 3):  # pinned entries take their fixed value
            starts[:, k] = bij.lo[k]
    kw = dict(max_tree_depth=10, seed=1, min_neff_cut=False)
    nuts_engine(engines, total, prior, bij, starts, n_warmup=3, n_samples=3, **dict(kw, max_tree_depth=5))  # threads, buffers, code paths warm
    repeats0 = sum(e.two_pass_repeats() for e in engines)
    t0 = time.perf_counter()
    res = nuts_engine(engines, total, prior, bij, starts, n_warmup=n_warmup, n_samples=n_samples, **kw)
    dt = time.perf_counter() - t0
    n_lf = sum(r["n_evals"] for r in res)
    reps = sum(e.two_pass_repeats() for e in engines) - repeats0
    draws = np.stack([r["samples"] for r in res])
    ess = effective_sample_size(draws)
    free = np.isfinite(ess)
    within = np.stack([effective_sample_size(draws[c]) for c in range(C)])  # each chain by itself
    rhat = split_rhat(draws)
    logp_chain = [float(np.mean(r["log_prob"])) for r in res]
    depth = np.concatenate([r["tree_depth"] for r in res])
    # share of the wall time spent after warm-up ~ share of the evaluations made there (2^depth - 1 per iteration)
    n_lf_sampling = float(np.sum(2.0 ** depth - 1.0))
    t_sampling = dt * min(1.0, n_lf_sampling / max(n_lf, 1))
    if not convergence_fields:
        # config 5: 300 warm-up iterations (the wall time of the default run is bounded through the iteration count) are fewer than the
        # reference's 1 000 (pipeline/utils.py:29-39), and after them the chains' mean log posteriors still lie thousands apart on this
        # catalog: effective sample sizes and R-hat of such draws are evidence of nothing and are not reported.  The throughput is.
        return {"chains": C, "host_threads": C, "warmup_iterations": n_warmup, "sampling_iterations": n_samples, "max_tree_depth": 10, "evals": n_lf, "evals_per_s": n_lf / dt,
                "us_per_leapfrog": 1e6 * dt / n_lf, "wall_s": dt, "priors": what, "two_pass_repeats": reps, "repeat_fraction": reps / max(n_lf, 1),
                "mean_tree_depth": float(np.mean(depth)), "max_tree_depth_reached": int(np.max(depth)), "fraction_at_max_depth": float(np.mean(depth >= 10)),
                "accept_prob": float(np.mean([r["accept_rate"] for r in res])), "step_size": [float(r["step_size"]) for r in res],
                "divergences": int(sum(r["n_divergent"] for r in res)), "mean_log_prob_per_chain": logp_chain,
                "convergence": ("not assessed: the warm-up is shorter than the reference's 1 000 iterations (pipeline/utils.py:29-39) and the chains have not "
                                "reached a common region (see mean_log_prob_per_chain); a throughput figure, not a sampling result")}
    return {"chains": C, "host_threads": C, "warmup_iterations": n_warmup, "sampling_iterations": n_samples, "max_tree_depth": 10, "evals": n_lf, "evals_per_s": n_lf / dt,
            "us_per_leapfrog": 1e6 * dt / n_lf, "wall_s": dt, "priors": what, "two_pass_repeats": reps, "repeat_fraction": reps / max(n_lf, 1),
            "mean_tree_depth": float(np.mean(depth)), "max_tree_depth_reached": int(np.max(depth)), "fraction_at_max_depth": float(np.mean(depth >= 10)),
            "accept_prob": float(np.mean([r["accept_rate"] for r in res])), "step_size": [float(r["step_size"]) for r in res],
            "divergences": int(sum(r["n_divergent"] for r in res)),
            "min_ess": float(np.min(ess[free])) if free.any() else None, "median_ess": float(np.median(ess[free])) if free.any() else None,
            "min_ess_per_s": float(np.min(ess[free]) / max(t_sampling, 1e-9)) if free.any() else None,
            "within_chain_min_ess": [float(np.min(w[free])) for w in within] if free.any() else None,
            "within_chain_median_ess": [float(np.median(w[free])) for w in within] if free.any() else None,
            "max_split_rhat": float(np.nanmax(rhat[free])) if free.any() else None, "mean_log_prob_per_chain": logp_chain,
            "ess_note": ("bulk ESS (multi-chain, Geyer) of the post-warm-up draws of all chains, per second of the sampling phase; within_chain_*: the same "
                         "estimator on each chain alone.  A multi-chain ESS near the chain count with healthy within-chain ESS and a large split R-hat means the "
                         "chains sit in different modes of this synthetic catalog's posterior (compare mean_log_prob_per_chain), not that they mix slowly")}


def native_nuts_lockstep(engines, chains_per_engine, comp_name, comp, total, thetas, n_warmup, n_samples, common_start=None, n_chains=None):
    """Vectorised chains inside the library (gwi_nuts_engine_lockstep): len(engines) groups of chains_per_engine chains, every
    leapfrog step of a group ONE batched launch, the groups alternating on one host thread -- numpyro's
    chain_method="vectorized" (examples/utils.py:63-85).  Same priors and tree depth as native_nuts; a THROUGHPUT figure
    (likelihood evaluations per second as the sampler sees them), bounded through the iteration count."""
    from gwinferno_amd.sampling import lockstep_stats, nuts_engine_lockstep

    G, K = len(engines), int(chains_per_engine)
    C = int(n_chains or G * K)  # more chains than slots: a queue (gwi_nuts_engine_queue) -- a chain that ends hands its slot to the next one waiting
    prior, bij, what = reference_priors(comp_name, comp, engines[0].n_theta)
    starts = np.stack([thetas[c % len(thetas)] if common_start is None else thetas[common_start] for c in range(C)])
    if bij is not None:
        for k in np.flatnonzero(bij.kind == 3):
            starts[:, k] = bij.lo[k]
    kw = dict(max_tree_depth=10, seed=1, min_neff_cut=False)
    nuts_engine_lockstep(engines, K, total, prior, bij, starts[: G * K], n_warmup=2, n_samples=2, **dict(kw, max_tree_depth=4))  # code paths warm
    repeats0 = sum(e.two_pass_repeats() for e in engines)
    t0 = time.perf_counter()
    res = nuts_engine_lockstep(engines, K, total, prior, bij, starts, n_warmup=n_warmup, n_samples=n_samples, **kw)
    dt = time.perf_counter() - t0
    st = lockstep_stats()
    n_lf = sum(r["n_evals"] for r in res)
    depth = np.concatenate([r["tree_depth"] for r in res])
    per_chain = [int(r["n_evals"]) for r in res]
    return {"mean_points_per_batch": st["mean_points_per_batch"], "evals_per_chain_min_max": [min(per_chain), max(per_chain)],
            "per_batch_us": {"collect": st["collect_us_per_batch"], "chains": st["chains_us_per_batch"], "issue": st["issue_us_per_batch"]},
            "starts": "one prior draw per chain (as native_nuts)" if common_start is None else f"every chain from prior draw {common_start}, seeds differ",
            "note": ("chains that need fewer evaluations finish earlier and the batches shrink: mean_points_per_batch of chains_per_group is what the batched "
                     "kernels get to work with; chains started from different prior draws on this synthetic catalog differ by up to 15x in evaluations"),
            "chains": C, "groups": G, "chains_per_group": K, "queued": C > G * K, "host_threads": 1, "batch_path": engines[0].batch_path(K), "warmup_iterations": n_warmup,
            "sampling_iterations": n_samples, "max_tree_depth": 10, "evals": n_lf, "evals_per_s": n_lf / dt, "us_per_leapfrog": 1e6 * dt / n_lf, "wall_s": dt,
            "two_pass_repeats": sum(e.two_pass_repeats() for e in engines) - repeats0, "mean_tree_depth": float(np.mean(depth)),
            "accept_prob": float(np.mean([r["accept_rate"] for r in res])), "divergences": int(sum(r["n_divergent"] for r in res)),
            "what": "gwi_nuts_engine_lockstep: every leapfrog step of a group of chains is one gwi_eval_batch_begin / _end; groups alternate on one host thread"}


RCCL_LEG_FLAG = "--rccl-leg"


def rccl_leg_main(argv):
    """Child process of one rank (`bench.py --rccl-leg ...`): the headline configuration sharded over the ranks with the
    partial records exchanged by ONE ncclAllGather (RCCL over xGMI) on each engine's own stream -- the exchange
    BASELINE.json's north_star names.  A process of its own, started by the rank before that touches a GPU, because this
    communicator has only ever run with one rank in the build environment (1-GPU boxes): if it hangs, the rank kills this
    child after a time limit and reports a non-zero exit code; the main measurement (shared-memory exchange) is unaffected."""
    import torch
    import torch.distributed as dist

    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.distributed import init_engine_communicator
    from gwinferno_amd.synthetic import make_config_catalog

    ap = argparse.ArgumentParser()
    ap.add_argument(RCCL_LEG_FLAG, action="store_true")
    ap.add_argument("--config", default="c2")
    ap.add_argument("--steps", type=int, default=400)
    args = ap.parse_args(argv)
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    comp_name, cat_name, _, desc = CONFIGS[args.config]
    pe, inj, total = make_config_catalog(cat_name)
    comp = COMPOSITIONS[comp_name](pe, inj)
    eng = comp.engine(device=local, rank=rank, world=world)
    devices = [None] * world
    dist.all_gather_object(devices, device_identity(torch, local))
    if len(set(devices)) != world:
        raise SystemExit(f"[rccl leg, rank {rank}] ranks do not hold distinct GPUs: {devices}")
    init_engine_communicator(eng)  # ncclCommInitRank; the unique id travels through torch.distributed
    rng = np.random.default_rng(1234)
    thetas = [comp.theta(draw_params(comp_name, rng)) for _ in range(64)]
    n = max(200, args.steps)
    blk = np.stack([thetas[i % len(thetas)] for i in range(n)])
    # before timing: the sharded evaluation (records through the all-gather) against an UNSHARDED engine over the whole
    # catalog on rank 0's GPU, value and whole gradient (what the shared-memory path reports as sharded_vs_single_gpu)
    r_sh = eng.evaluate_sharded(thetas[0], total, min_neff_cut=False)
    sharded_check = None
    if rank == 0:
        eng_full = COMPOSITIONS[comp_name](pe, inj).engine(device=local)
        r_full = eng_full.evaluate(thetas[0], total, min_neff_cut=False)
        scale = max(1.0, float(np.max(np.abs(r_full.grad))))
        sharded_check = {"log_likelihood_rel_err": abs(r_sh.log_likelihood - r_full.log_likelihood) / max(1e-300, abs(r_full.log_likelihood)),
                         "grad_max_err_over_scale": float(np.max(np.abs(r_sh.grad - r_full.grad))) / scale}
        sharded_check["within_tolerance"] = bool(sharded_check["log_likelihood_rel_err"] <= 1e-9 and sharded_check["grad_max_err_over_scale"] <= 1e-8)
        eng_full.close()
    eng.evaluate_sequence(blk[:100], total, min_neff_cut=False)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ll, _ = eng.evaluate_sequence(blk, total, min_neff_cut=False)
    dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    lls = torch.tensor([float(ll[-1])], dtype=torch.float64, device="cuda")
    box = [torch.zeros_like(lls) for _ in range(world)]
    dist.all_gather(box, lls)
    same = all(float(b.item()) == float(lls.item()) for b in box)
    if rank == 0:
        print(json.dumps({"evals_per_s": n / float(t.item()), "ms_per_step": 1e3 * float(t.item()) / n, "steps": n, "rccl_ranks": world, "workload": desc,
                          "exchange": "one ncclAllGather of the ~1 KiB partial records per evaluation, on the engine's own stream (gwi_eval_sharded)",
                          "last_log_likelihood": float(ll[-1]), "identical_on_all_ranks": bool(same), "sharded_vs_single_gpu": sharded_check,
                          "two_pass_repeats": eng.two_pass_repeats(), "unique_devices": len(set(devices)) == world, "devices": devices}), flush=True)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()
    return 0


def run_rccl_leg(args):
    """Called by every rank BEFORE it touches a GPU: start this rank's `--rccl-leg` child (same RANK / WORLD_SIZE /
    LOCAL_RANK, rendezvous on the port after the run's own), wait for it with a time limit, and return rank 0's parsed line
    plus the child's exit code.  A child that does not finish is killed by its PID and reported with exit code 124."""
    env = dict(os.environ)
    env["MASTER_ADDR"] = env.get("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29517")) + 1)
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
        env.setdefault(k, v)
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)  # the child makes a plain env:// rendezvous of its own (rank 0 hosts the store)
    world = int(env.get("WORLD_SIZE", "1"))
    env.setdefault("NCCL_DEBUG", "WARN")  # a failed first contact should say why: the warnings' tail goes into multi_gpu.rccl_probe
    # communicator start-up grows with the ranks (a ring over xGMI per channel, one bootstrap connection per peer), and so do N
    # processes importing torch and loading the code objects side by side: 120 s + 15 s per rank unless told otherwise
    limit = float(os.environ.get("GWI_BENCH_RCCL_TIMEOUT", str(120 + 15 * world)))
    cmd = [sys.executable, os.path.abspath(__file__), RCCL_LEG_FLAG, "--config", args.config, "--steps", str(max(200, min(args.steps, 2000)))]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        out, err = child.communicate(timeout=limit)
        code = child.returncode
    except subprocess.TimeoutExpired:
        child.kill()  # this exact PID
        out, err = child.communicate()
        code = 124
    res = {"child_exit_code": code, "time_limit_s": limit}
    warn = [ln for ln in (err or "").splitlines() if "NCCL WARN" in ln or "RCCL" in ln and "error" in ln.lower()]
    if code != 0:
        res["rccl_debug_tail"] = (" | ".join(warn) or (err or "").strip())[-300:]
    lines = [ln for ln in (out or "").splitlines() if ln.startswith("{")]
    if lines and code == 0:
        try:
            res.update(json.loads(lines[-1]))
        except ValueError:
            res["error"] = "unparsable line from the RCCL leg"
    elif code == 124:
        res["error"] = f"the RCCL leg did not finish within {limit:.0f} s; its process was killed"
    elif code != 0:
        res["error"] = "the RCCL leg failed: " + (err or "").strip().splitlines()[-1][:300] if (err or "").strip() else "the RCCL leg failed"
    return res


def main():
    if RCCL_LEG_FLAG in sys.argv[1:]:
        sys.exit(rccl_leg_main(sys.argv[1:]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS), help="headline configuration (default: BASELINE config 2, the one the metric is quoted on)")
    ap.add_argument("--also", default=None, help="comma-separated further configs reported under `configs` (default with headline c2: c3,c5,b50k,def50k at N = 1, c3,c5 at N > 1; 'none' disables)")
    ap.add_argument("--full", action="store_true", help="also run the secondary legs (sets in flight, the other batched kernels, chains, the library's NUTS, host-side setup, NumPy reference formulation); they go to the detail file")
    ap.add_argument("--detail", default="bench_detail.json", help="side file for everything that is not in the compact line ('' disables)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--k-batch", type=int, default=16, help="also time batched evaluation (K hyper-points per launch); 0 disables")
    ap.add_argument("--chains", type=int, default=4, help="also time this many independent chains interleaved on one GPU (begin/end); <= 1 disables")
    ap.add_argument("--timing-every", type=int, default=10, help="kernel begin/end timing on every n-th step of the kernel-timing block")
    ap.add_argument("--spin", type=float, default=0.6, help="seconds of untimed evaluations before the timed region (GPU clocks up)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and os.environ.get("GWI_FORCE_SHARDED") != "1":
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    # N > 1 on one GPU per rank: the in-engine RCCL exchange runs FIRST, in a child process of every rank, before this
    # process touches a GPU (run_rccl_leg); the main measurement below uses the shared-memory exchange unless told otherwise
    rccl_leg = None
    forced = os.environ.get("GWI_FORCE_SHARDED") == "1"  # the N > 1 code path with a world of one rank (tests)
    if (((args.gpus > 1 and "RANK" in os.environ) or forced) and os.environ.get("GWI_BENCH_BACKEND", "nccl") == "nccl" and "GWI_BENCH_DEVICE" not in os.environ
            and "GWI_BENCH_SHARE_DEVICES" not in os.environ and os.environ.get("GWI_BENCH_RCCL_VARIANT", "1") != "0"):
        import torch  # counting devices does not initialise the GPU

        if 0 < torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", args.gpus)):
            rccl_leg = {"skipped": "fewer GPUs than ranks on this node: RCCL cannot place two ranks of one communicator on one device"}
        else:
            rccl_leg = run_rccl_leg(args)

    run = Run(args)
    if run.world != args.gpus and not (run.world == 1 and os.environ.get("GWI_FORCE_SHARDED") == "1"):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={run.world}")
    # the north_star exchange carries the headline when its probe came back clean on EVERY rank (exit code 0; on rank 0 also
    # the comparison with the unsharded engine within tolerance and identical results on all ranks)
    prefer_rccl = False
    if rccl_leg is not None and run.dist is not None and "skipped" not in rccl_leg:
        mine = rccl_leg.get("child_exit_code") == 0
        if run.rank == 0:
            chk = rccl_leg.get("sharded_vs_single_gpu") or {}
            mine = mine and bool(chk.get("within_tolerance")) and bool(rccl_leg.get("identical_on_all_ranks")) and rccl_leg.get("rccl_ranks") == run.world
        prefer_rccl = run.all_agree(mine)
        # what the OTHER ranks' children said when they failed (rank 0's own child may merely have timed out waiting for them)
        fails = [r for r in run.gather_rows({"rank": run.rank, "child_exit_code": rccl_leg.get("child_exit_code"), "tail": rccl_leg.get("rccl_debug_tail")}) if r["child_exit_code"] != 0]
        if run.rank == 0:
            rccl_leg["probe_clean_on_every_rank"] = bool(prefer_rccl)
            if fails:
                rccl_leg["failed_ranks"] = fails[:8]
                if not rccl_leg.get("rccl_debug_tail"):
                    rccl_leg["rccl_debug_tail"] = f"rank {fails[0]['rank']}: {fails[0]['tail']}"[-300:]
    also = args.also
    if also is None:
        also = ("c3,c5,b50k,def50k" if args.gpus == 1 else "c3,c5") if args.config == "c2" else "none"
    extra_cfgs = [c for c in also.split(",") if c and c != "none" and c != args.config]

    head = measure(run, args.config, args.steps, args.warmup, args.timing_every, spin_s=args.spin, headline=True, with_cpu=not args.no_cpu_baseline, k_batch=args.k_batch,
                   chains=args.chains, prefer_rccl=prefer_rccl, full=args.full, cpu_budget_s=6.0)
    blocks = {}
    for cfg in extra_cfgs:
        # the other BASELINE configurations with the same procedure (their own warm-up, K and latency blocks: sized so that
        # the default run stays within minutes)
        k = {"c5": 300, "c3": 600}.get(cfg, 600)
        blocks[cfg] = measure(run, cfg, k, 50, args.timing_every, spin_s=min(args.spin, 0.3), headline=False, with_cpu=not args.no_cpu_baseline, k_batch=args.k_batch, chains=0,
                              nuts_chains=args.chains, prefer_rccl=prefer_rccl, full=args.full, cpu_budget_s=2.0)

    if run.rank == 0:
        out = {
            "metric": "log-likelihood evals/sec (value + gradient + diagnostic sites, host theta -> host results)",
            "value": head["value"],
            "unit": "evals/s",
            "n_gpus": run.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
        }
        for k, v in head.items():
            out.setdefault(k, v)
        if blocks:
            out["configs"] = blocks
    if run.rank == 0 and run.dist is not None:
        # the exchanges side by side; `rccl_ranks` says how many ranks the communicator that carried the HEADLINE's records had
        mg = out["multi_gpu"]
        head_key = mg.pop("headline_exchange_key")
        mg["exchanges"] = {head_key: {"ms_per_step": out["ms_per_step"], "evals_per_s": out["value"], "headline": True}}
        if "shm_side" in mg:
            mg["exchanges"]["shm"] = mg.pop("shm_side")
        if rccl_leg is not None:  # the probe of the RCCL exchange in a child process per rank (time-limited), always reported
            mg["exchanges"]["rccl_allgather_probe" if head_key == "rccl_allgather" else "rccl_allgather"] = rccl_leg
        for blk in out.get("configs", {}).values():
            bm = blk.get("multi_gpu")
            if bm:
                bm["exchanges"] = {bm.pop("headline_exchange_key"): {"ms_per_step": blk["ms_per_step"], "evals_per_s": blk["value"]}}
    if run.rank == 0:
        # RCCL prints a version banner through C stdio; flush it first so the JSON line is the last
        import ctypes

        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        detail_path = None
        if args.detail:
            try:
                with open(args.detail, "w") as fh:
                    json.dump(strict(out), fh, allow_nan=False, indent=1)
                detail_path = args.detail
            except OSError as exc:
                print(f"[bench] could not write {args.detail}: {exc}", file=sys.stderr)
        print(dump_line(compact_line(out, detail_path)), flush=True)  # the ONE JSON line, last thing on stdout
    if run.dist is not None:
        # the line is out: a rank that does not come back from the shutdown (a peer lost in the RCCL variant) must not
        # hold the run
        import threading

        dog = threading.Timer(60.0, lambda: os._exit(3))  # the line is out; a shutdown that hangs is still a failure of this rank
        dog.daemon = True
        dog.start()
        run.dist.barrier()
        run.dist.destroy_process_group()
        dog.cancel()


if __name__ == "__main__":
    main()
