#!/usr/bin/env python3
"""bench.py -- log-likelihood evaluations per second of the hierarchical population likelihood.

One *step* = one value-and-gradient evaluation of ``hierarchical_likelihood`` (log_l, d log_l/d theta
and every diagnostic site) for one hyper-parameter point, end to end from a host ``theta`` to host
results, with the catalog already resident in HBM -- what one NUTS leapfrog costs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c1|c3|c5]

N = 1: a single engine.  N > 1 (launched by torch.distributed.run, one rank per GPU): events and
injections are sharded across ranks, each rank scans its shard, ONE RCCL all-gather of the small
partial records is the only exchange, and every rank assembles the same result ("strong" scaling:
the BASELINE catalog size is fixed).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def pmc_table(config):
    """Row of profiles/<latest round>/traffic.json for this config (or {})."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", "traffic.json")))
    try:
        return json.load(open(files[-1])).get(config, {}) if files else {}
    except Exception:
        return {}


def pmc_traffic(config):
    """HBM bytes per scan launch from the committed rocprofv3 PMC passes of the latest round
    (profiles/<round>/traffic.json, produced by tools/profile_round.sh + tools/summarize_profiles.py:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950 correction)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", "traffic.json")))
    if not files:
        return None, None
    try:
        table = json.load(open(files[-1]))
        return table[config]["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def cpu_baseline(comp, thetas, total, budget_s=10.0):
    """The C/OpenMP restatement (oracle/gwpop_oracle.c: value + gradient + sites, same flat model
    description as the GPU engine receives) timed on all host cores, plus the NumPy restatement of the
    reference formulation (value only, 1 core) for scale.  Checker code, never the product path."""
    from oracle.c_oracle import COracle

    orc = COracle(comp.engine().bound)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 CPU quota of the container, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            avail = max(1, min(avail, int(float(quota) / float(period))))
    except Exception:
        pass
    # the visible CPU count can exceed what the container may really use: probe a few thread counts
    # (one evaluation each) and keep the fastest
    best, cores = None, 1
    for nt in sorted({1, 2, 4, 8, 16, 32, 64, 128, avail}):
        if nt > avail:
            continue
        orc.evaluate(thetas[0], total, min_neff_cut=False, n_threads=nt)
        t0 = time.perf_counter()
        orc.evaluate(thetas[0], total, min_neff_cut=False, n_threads=nt)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, nt
    n, t_used = 0, 0.0
    while t_used < budget_s:
        t0 = time.perf_counter()
        orc.evaluate(thetas[n % len(thetas)], total, min_neff_cut=False, n_threads=cores)
        t_used += time.perf_counter() - t0
        n += 1
    t0 = time.perf_counter()
    orc.evaluate(thetas[0], total, min_neff_cut=False, n_threads=1)
    t_single = time.perf_counter() - t0
    return {
        "value": n / t_used,
        "unit": "evals/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n} value+gradient evals of the full catalog by the C/OpenMP oracle on {cores} threads ({t_used:.1f}s)",
        "single_thread_evals_per_s": 1.0 / t_single,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--k-batch", type=int, default=16, help="also time batched evaluation (K hyper-points per launch); 0 disables")
    ap.add_argument("--chains", type=int, default=4, help="also time this many independent chains interleaved on one GPU (begin/end); <= 1 disables")
    ap.add_argument("--timing-every", type=int, default=50, help="kernel begin/end timing on every n-th timed step")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    dist = None
    collective = None
    force_sharded = os.environ.get("GWI_FORCE_SHARDED") == "1"  # exercise the N>1 code path on one GPU
    if world > 1 or force_sharded:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # GWI_BENCH_BACKEND=gloo + GWI_BENCH_DEVICE=0: several ranks sharing ONE GPU with a CPU-side exchange --
        # a logic test of the multi-rank path on a single-GPU box (implies the torch collective)
        backend = os.environ.get("GWI_BENCH_BACKEND", "nccl")
        if "GWI_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["GWI_BENCH_DEVICE"])
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
            os.environ["GWI_TORCH_COLLECTIVE"] = "1"

    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_config_catalog

    comp_name, cat_name, c_alg, desc = CONFIGS[args.config]
    pe, inj, total = make_config_catalog(cat_name)
    n_ev, n_pe = pe["mass_1"].shape
    n_inj = inj["mass_1"].shape[0]
    comp = COMPOSITIONS[comp_name](pe, inj)
    # the driving thread next to its GPU (numactl-style host placement), before anything pinned is allocated
    from gwinferno_amd.engine import pin_thread_to_device

    pinned = pin_thread_to_device(local_rank) if os.environ.get("GWI_BENCH_PIN", "1") != "0" and torch.cuda.is_available() else False
    eng = comp.engine(device=local_rank, rank=rank, world=world)
    rng = np.random.default_rng(1234)
    pool = [draw_params(comp_name, rng) for _ in range(64)]
    thetas = [comp.theta(p) for p in pool]

    if dist is not None:
        # hot loop: scan + ncclAllGather + assembly inside the engine (no Python/torch in the data path);
        # GWI_TORCH_COLLECTIVE=1 selects the torch.distributed all_gather_into_tensor variant instead
        from gwinferno_amd.distributed import ShardedLikelihood, init_engine_communicator

        use_torch = os.environ.get("GWI_TORCH_COLLECTIVE") == "1"
        if not use_torch:
            try:
                init_engine_communicator(eng)
                ok = 1
            except Exception as exc:  # keep the run alive on the torch path; say so
                print(f"[rank {rank}] in-engine RCCL communicator unavailable ({exc}); using torch.distributed all_gather", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # all ranks must agree on the path
            use_torch = int(flag.item()) == 0
        collective = "torch.distributed all_gather_into_tensor" if use_torch else "ncclAllGather inside the engine"
        if use_torch:
            sharded = ShardedLikelihood(eng, total, device=torch.device("cuda", local_rank) if dist.get_backend() == "nccl" else None)

            def step(i):
                return sharded.evaluate(thetas[i % len(thetas)], min_neff_cut=False)
        else:
            vg = eng.configure(total, min_neff_cut=False)

            def step(i):
                return vg(thetas[i % len(thetas)])
    else:
        vg = eng.configure(total, min_neff_cut=False)  # value_and_grad(theta) -> (log_likelihood, grad buffer)

        def step(i):
            return vg(thetas[i % len(thetas)])

    # N > 1: before timing, rank 0 checks the sharded evaluation against an unsharded engine over the whole
    # catalog on its own GPU (same theta): the multi-GPU path is otherwise only covered by world-2 gloo tests
    sharded_check = None
    if dist is not None:
        r_sh = step(0)
        ll_sh, g_sh = (r_sh[0], np.array(r_sh[1])) if isinstance(r_sh, tuple) else (r_sh.log_likelihood, np.array(r_sh.grad))
        if rank == 0:
            full = COMPOSITIONS[comp_name](pe, inj)
            eng_full = full.engine(device=local_rank)
            r_full = eng_full.evaluate(thetas[0], total, min_neff_cut=False)
            scale = max(1.0, float(np.max(np.abs(r_full.grad))))
            sharded_check = {"log_likelihood_rel_err": abs(ll_sh - r_full.log_likelihood) / max(1e-300, abs(r_full.log_likelihood)),
                             "grad_max_err_over_scale": float(np.max(np.abs(g_sh - r_full.grad))) / scale}
            eng_full.close()
            if sharded_check["log_likelihood_rel_err"] > 1e-9 or sharded_check["grad_max_err_over_scale"] > 1e-8:
                print(f"[bench] WARNING: sharded result differs from the single-GPU result: {sharded_check}", file=sys.stderr)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The K steps are K sequential, blocking evaluations at K given points.  The reference runs that loop inside one
    # XLA program (NUTS under jit, examples/utils.py:63-85): no host-language binding between two evaluations.  So does
    # the timed region here: one gwi_eval_sequence call (a C loop of gwi_eval / gwi_eval_sharded, every result back on
    # the host before the next point starts).  The same loop driven from Python is reported beside it.
    in_library = not (dist is not None and use_torch) and os.environ.get("GWI_BENCH_PYTHON_LOOP") != "1"
    seq = np.stack([thetas[i % len(thetas)] for i in range(max(args.steps, args.warmup, 1))])
    scan_ms, comb_ms, fin_ms = [], [], []
    if in_library:
        if args.warmup:
            eng.evaluate_sequence(seq[: args.warmup], total, min_neff_cut=False)
        fence()
        t0 = time.perf_counter()
        # ---- timed region: exactly K steps -------------------------------------------------------
        ll_seq, g_seq, kms = eng.evaluate_sequence(seq[: args.steps], total, min_neff_cut=False, timing_every=max(args.timing_every, 0) or args.steps + 1)
        fence()
        elapsed = time.perf_counter() - t0
        if args.timing_every > 0:
            sel = kms[:, 0] >= 0
            scan_ms, comb_ms, fin_ms = list(kms[sel, 0]), list(kms[sel, 1]), list(kms[sel, 2])
    else:
        for i in range(args.warmup):
            step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            timed = args.timing_every > 0 and (i % args.timing_every == 0)
            if timed:
                eng.set_timing(True)
            res = step(i)
            if timed:
                ms = eng.last_kernel_ms()
                scan_ms.append(ms[0])
                comb_ms.append(ms[1])
                fin_ms.append(ms[2])
                eng.set_timing(False)
        fence()
        elapsed = time.perf_counter() - t0
    # the same loop driven from Python through the allocation-free closure (what a NumPy sampler pays per step)
    n_py = min(args.steps, 2000)
    for i in range(min(50, n_py)):
        step(i)
    fence()
    t0p = time.perf_counter()
    for i in range(n_py):
        res = step(i)
    fence()
    python_driven = n_py / (time.perf_counter() - t0p)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N > 1, secondary figure: the other way N GPUs are used for this workload -- one independent chain per GPU,
    # each over the WHOLE catalog (numpyro chain_method="parallel"); no collective, per-GPU work fixed (weak scaling)
    replicas = None
    if dist is not None:
        rep = COMPOSITIONS[comp_name](pe, inj)
        eng_rep = rep.engine(device=local_rank)
        vg_rep = eng_rep.configure(total, min_neff_cut=False)
        n_rep = max(200, args.steps // 2)
        for i in range(50):
            vg_rep(thetas[i % len(thetas)])
        fence()
        t0r = time.perf_counter()
        for i in range(n_rep):
            vg_rep(thetas[i % len(thetas)])
        fence()
        tr = torch.tensor([time.perf_counter() - t0r], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        replicas = {"evals_per_s": world * n_rep / float(tr.item()), "scaling": "weak", "what": "one independent chain per GPU over the whole catalog, no collective"}
        eng_rep.close()

    out = None
    if rank == 0:
        evals_per_s = args.steps / elapsed
        scan_us = 1e3 * float(np.mean(scan_ms)) if scan_ms else float("nan")
        alg_bytes = 8.0 * c_alg * (n_ev * n_pe + n_inj) / world  # per launch on one GPU
        achieved = alg_bytes / (scan_us * 1e-6) / 1e9 if scan_ms else float("nan")
        out = {
            "metric": "log-likelihood evals/sec (value + gradient + diagnostic sites, host theta -> host results)",
            "value": evals_per_s,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "step_loop": "gwi_eval_sequence: K blocking evaluations in a C loop inside the library" if in_library else "Python loop over the configured closure",
            "python_driven_evals_per_s": python_driven,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": desc,
                "composition": comp_name,
                "n_events": int(n_ev),
                "n_pe": int(n_pe),
                "n_inj": int(n_inj),
                "n_theta": int(eng.n_theta),
                "flags": "min_neff_cut=False (tests/inference_test.py:185)",
                "parallelism": f"events+injections sharded over {world} GPU(s), one all-gather of partial records per eval ({collective})" if dist is not None else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "scan_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args.config)[0] if world == 1 else None,
                "traffic_source": pmc_traffic(args.config)[1] if world == 1 else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_read_per_launch": float(eng.bytes_per_sample) * (eng.n_ev * eng.n_pe + eng.n_inj),
                "avg_kernel_us": {"scan": scan_us, "combine": 1e3 * float(np.mean(comb_ms)) if comb_ms else None, "final": 1e3 * float(np.mean(fin_ms)) if fin_ms else None},
                "median_scan_us": 1e3 * float(np.median(scan_ms)) if scan_ms else None,
                "timing": (f"kernel begin/end of every {args.timing_every}th timed step: dispatch timestamps of the engine's AQL queue "
                           "(hsa_amd_profiling_get_dispatch_time, what rocprofv3's kernel trace reports)" if eng.dispatch_info() == "aql: active" else
                           f"HIP start/stop events attached to each launch (hipExtLaunchKernelGGL) on the engine's stream, every {args.timing_every}th timed step"),
                "dispatch": eng.dispatch_info(),
                "host_thread_pinned_to_gpu_numa_node": bool(pinned),
                "timed_launches": len(scan_ms),
                # second view: the scan is fp64-issue/latency bound, not HBM bound
                "fp64_vector": {
                    "peak_tflops": 78.6,
                    "flop_per_launch_pmc": pmc_table(args.config).get("fp64_flop_per_launch") if world == 1 else None,
                    "achieved_tflops": (pmc_table(args.config).get("fp64_flop_per_launch", 0.0) / (scan_us * 1e-6) / 1e12) if (world == 1 and scan_ms and pmc_table(args.config).get("fp64_flop_per_launch")) else None,
                },
            },
            "last_log_likelihood": float(res[0]) if isinstance(res, tuple) else float(res.log_likelihood),
            "sharded_vs_single_gpu": sharded_check,
            "independent_chains": replicas,
            "c_loop_us_per_eval": (1e6 * eng.selftime(thetas[0], total, n_iter=min(args.steps, 2000), min_neff_cut=False)) if dist is None else None,
        }
        if dist is None and args.k_batch > 1:
            # secondary number: vectorised chains -- K hyper-points per launch (not the headline `value`,
            # which is one sequential chain)
            K = args.k_batch
            tb = np.stack(thetas[:K] if len(thetas) >= K else (thetas * K)[:K])
            vgb = eng.configure_batch(K, total, min_neff_cut=False)  # values_and_grads(thetas[K]) -> (log_l[K], grad[K, n_theta])
            for _ in range(30):
                vgb(tb)
            n_b = max(20, args.steps // (4 * K))
            t0 = time.perf_counter()
            for _ in range(n_b):
                vgb(tb)
            dt = time.perf_counter() - t0
            out["batched"] = {"k_batch": K, "evals_per_s": n_b * K / dt, "us_per_eval": 1e6 * dt / (n_b * K), "launch_sets": n_b}
        if dist is None and args.chains > 1:
            # secondary number: C independent chains on this one GPU, each with its own engine, their evaluations in
            # flight together (gwi_eval_begin / gwi_eval_end); no lock step, every chain follows its own theta sequence
            C = args.chains
            extra = [COMPOSITIONS[comp_name](pe, inj) for _ in range(C - 1)]
            pairs = [e.configure_async(total, min_neff_cut=False) for e in [eng] + [c.engine(device=local_rank) for c in extra]]
            n_c = max(200, args.steps // 2)
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
            # the same C engines, one HOST THREAD each running blocking evaluations in a C loop (the GIL is released inside
            # the library): launch costs spread over cores, the GPU interleaves the chains' kernels
            import threading

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
            # the same engines inside a sampler: the library's C++ NUTS (gwi_nuts_engine, include/gwi_sampler.h), one chain
            # per engine and host thread, flat priors wide enough not to matter; trees capped at 2^6 leapfrogs so that the
            # line stays within seconds whatever the synthetic posterior looks like.  Likelihood evaluations per second
            # as a sampler sees them (every leapfrog = one value + gradient)
            from gwinferno_amd.sampling import GaussianSmoothingPrior, nuts_engine

            prior = GaussianSmoothingPrior(eng.n_theta).normal(slice(0, eng.n_theta), 10.0)
            starts = np.stack(thetas[:C])
            kw = dict(max_tree_depth=6, seed=1, min_neff_cut=False)
            nuts_engine(all_engines, total, prior, None, starts, n_warmup=5, n_samples=5, **kw)
            t0 = time.perf_counter()
            res = nuts_engine(all_engines, total, prior, None, starts, n_warmup=60, n_samples=60, **kw)
            dt = time.perf_counter() - t0
            n_lf = sum(r["n_evals"] for r in res)
            out["native_nuts"] = {"chains": C, "host_threads": C, "iterations_per_chain": 120, "evals": n_lf, "evals_per_s": n_lf / dt, "us_per_leapfrog": 1e6 * dt / n_lf}
            for c in extra:
                c.engine().close()
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only (rank 0), as the contract asks
            out["cpu_baseline"] = cpu_baseline(comp, thetas, total)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush it first so the JSON line is the last
        import ctypes

        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)  # the ONE JSON line, last thing on stdout


if __name__ == "__main__":
    main()
