#!/usr/bin/env python3
"""Summarise gpurun_out/prof/<config>/ (written by tools/profile_round.sh on the GPU box) into
profiles/<round>/: per-config rocprofv3 kernel stats (copied verbatim), a PMC traffic table, and
traffic.json, which bench.py reads to fill roofline.traffic.

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and
WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B, i.e.
reports exactly half of a coalesced streaming read, so reads = 2 x FETCH_SIZE x 1024.  (Calibrated
here: the scan kernel's loads are 8 B/lane coalesced; 2 x FETCH_SIZE reproduces the known column
bytes of every config to within 3 %.)"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counter(path_glob, kernel_substr, counter):
    files = glob.glob(path_glob)
    if not files:
        return None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0])) if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals) if vals else None


def main(round_name, dst=None):
    src = os.path.join(ROOT, "gpurun_out", "prof")
    dst = dst or os.path.join(ROOT, "profiles", round_name)
    os.makedirs(dst, exist_ok=True)
    traffic = {}
    lines = ["| config | kernel | calls | avg us (rocprofv3) | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes/launch (2*F+W)*1024 |", "|---|---|---|---|---|---|---|"]
    for cfg in sorted(c for c in os.listdir(src) if not c.startswith(("batch_", "chain_")) and os.path.isdir(os.path.join(src, c))):
        stats = glob.glob(os.path.join(src, cfg, "trace", "*", "*_kernel_stats.csv"))
        if not stats:
            continue
        shutil.copy(stats[0], os.path.join(dst, f"{cfg}_kernel_stats.csv"))
        for name in ("bench.json", "bench_under_trace.json", "bench_detail.json"):
            p = os.path.join(src, cfg, name)
            if os.path.exists(p) and os.path.getsize(p):
                shutil.copy(p, os.path.join(dst, f"{cfg}_{name}"))
        rows = list(csv.DictReader(open(stats[0])))
        for r in rows:
            if "gwi::" not in r["Name"]:
                continue
            short = r["Name"].split("(")[0].replace("void ", "")
            key = "scan_kernel" if "scan_kernel" in short else short.replace("gwi::", "")
            f = mean_counter(os.path.join(src, cfg, "fetch", "*", "*_counter_collection.csv"), key, "FETCH_SIZE")
            w = mean_counter(os.path.join(src, cfg, "write", "*", "*_counter_collection.csv"), key, "WRITE_SIZE")
            hbm = (2 * f + w) * 1024 if f is not None and w is not None else None
            lines.append(f"| {cfg} | {short} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {f:.1f} | {w:.1f} | {hbm:.0f} |" if hbm else f"| {cfg} | {short} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | - | - | - |")
            if key == "scan_kernel":
                traffic[cfg] = {"scan_avg_us_rocprof": float(r["AverageNs"]) / 1e3, "fetch_size_kib": f, "write_size_kib": w, "hbm_bytes_per_launch": hbm}
    # ---- SQ counters of the scan kernel: instruction mix, fp64 rate, where wave time goes
    sq_lines = ["| config | waves | VALU/wave | SALU/wave | LDS/wave | fp64 GFLOP per launch | fp64 TFLOP/s (of 78.6 peak) | wave time waiting (s_waitcnt/barrier) | issuing | LDS bank-conflict share |", "|---|---|---|---|---|---|---|---|---|---|"]
    for cfg in sorted(c for c in os.listdir(src) if not c.startswith(("batch_", "chain_")) and os.path.isdir(os.path.join(src, c))):
        def c(name, part):
            return mean_counter(os.path.join(src, cfg, part, "*", "*_counter_collection.csv"), "scan_kernel", name)
        w = c("SQ_WAVES", "sq_a")
        if not w or cfg not in traffic:
            continue
        valu, salu, lds = c("SQ_INSTS_VALU", "sq_a"), c("SQ_INSTS_SALU", "sq_a"), c("SQ_INSTS_LDS", "sq_a")
        add, mul, fma, tr = c("SQ_INSTS_VALU_ADD_F64", "sq_a"), c("SQ_INSTS_VALU_MUL_F64", "sq_a"), c("SQ_INSTS_VALU_FMA_F64", "sq_a"), c("SQ_INSTS_VALU_TRANS_F64", "sq_a")
        flop = 64.0 * ((add or 0) + (mul or 0) + 2 * (fma or 0) + (tr or 0))
        t_us = traffic[cfg]["scan_avg_us_rocprof"]
        wc, wa, ai = c("SQ_WAVE_CYCLES", "sq_b"), c("SQ_WAIT_ANY", "sq_b"), c("SQ_ACTIVE_INST_ANY", "sq_b")
        bc, ia = c("SQ_LDS_BANK_CONFLICT", "sq_b"), c("SQ_LDS_IDX_ACTIVE", "sq_b")
        traffic[cfg].update(fp64_flop_per_launch=flop, fp64_tflops=flop / (t_us * 1e-6) / 1e12)
        if any(ln.startswith(f"| {cfg} |") for ln in sq_lines):
            continue
        sq_lines.append(f"| {cfg} | {w:.0f} | {valu / w:.0f} | {salu / w:.0f} | {lds / w:.0f} | {flop / 1e9:.2f} | {flop / (t_us * 1e-6) / 1e12:.2f} | {wa / wc:.0%} | {ai / wc:.0%} | {(bc / ia if ia else 0):.0%} |")
    # ---- LDS pipe of the scan kernel (spline-gradient atomics): address / bank conflicts and time waiting on LDS
    lds_lines = ["| config | SQ_LDS_ADDR_CONFLICT / SQ_LDS_IDX_ACTIVE | SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE | SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES | SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES |", "|---|---|---|---|---|"]
    for cfg in sorted(c for c in os.listdir(src) if not c.startswith(("batch_", "chain_")) and os.path.isdir(os.path.join(src, c))):
        def c(name):
            return mean_counter(os.path.join(src, cfg, "lds", "*", "*_counter_collection.csv"), "scan_kernel", name)
        ia, wc = c("SQ_LDS_IDX_ACTIVE"), c("SQ_WAVE_CYCLES")
        if not ia or not wc:
            continue
        ac, bc, wl, al = c("SQ_LDS_ADDR_CONFLICT") or 0.0, c("SQ_LDS_BANK_CONFLICT") or 0.0, c("SQ_WAIT_INST_LDS") or 0.0, c("SQ_ACTIVE_INST_LDS") or 0.0
        lds_lines.append(f"| {cfg} | {ac / ia:.1%} | {bc / ia:.1%} | {wl / wc:.1%} | {al / wc:.1%} |")
        if cfg in traffic:
            traffic[cfg].update(lds_addr_conflict_share=ac / ia, lds_bank_conflict_share=bc / ia, wait_inst_lds_share=wl / wc)
    # ---- batched kernels (K = 16 points per launch): 4-tap gradient vs the MFMA gradient GEMM
    b_lines = ["| config | path | kernel | calls | avg us (rocprofv3) | us per evaluation (untraced loop) | MFMA instr / launch | fp64 MFMA GFLOP / launch | MFMA TFLOP/s | of the 78.6 TFLOP/s matrix peak | VALU instr / launch | LDS instr / launch |",
               "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    batched = {}
    p_lines = ["| config | path | kernel | scan avg us (rocprofv3) | us per evaluation (untraced loop) | FETCH_SIZE KiB | WRITE_SIZE KiB | bytes beyond L2 per launch, MB ((2F+W)*1024) | algorithmic bytes of ONE pass over the catalog, MB | ratio |",
               "|---|---|---|---|---|---|---|---|---|---|"]
    for name in sorted(c for c in os.listdir(src) if c.startswith("batch_")):
        _, cfg, path = name.split("_")
        stats = glob.glob(os.path.join(src, name, "trace", "*", "*_kernel_stats.csv"))
        if not stats:
            continue
        shutil.copy(stats[0], os.path.join(dst, f"{name}_kernel_stats.csv"))
        run = {}
        try:
            run = json.loads(open(os.path.join(src, name, "run.json")).read().strip().splitlines()[-1])
        except Exception:
            pass
        for r in csv.DictReader(open(stats[0])):
            if "gwi::scan_" not in r["Name"]:
                continue
            short = r["Name"].split("(")[0].replace("void ", "")
            key = "scan_mfma_kernel" if "scan_mfma_kernel" in short else ("scan_rows_kernel" if "scan_rows_kernel" in short else ("scan_pbatch_kernel" if "scan_pbatch_kernel" in short else "scan_kernel"))
            def c(cn):
                return mean_counter(os.path.join(src, name, "mfma", "*", "*_counter_collection.csv"), key, cn)
            n_mfma, mops = c("SQ_INSTS_MFMA"), c("SQ_INSTS_VALU_MFMA_MOPS_F64")
            n_valu, n_lds = c("SQ_INSTS_VALU"), c("SQ_INSTS_LDS")
            t_us = float(r["AverageNs"]) / 1e3
            flop = 2048.0 * n_mfma if n_mfma else 0.0  # v_mfma_f64_16x16x4: 16 x 16 x 4 x 2 flop per wave instruction
            tf = flop / (t_us * 1e-6) / 1e12 if flop else 0.0
            b_lines.append(f"| {cfg} | {path} | {short[:60]} | {r['Calls']} | {t_us:.1f} | {run.get('us_per_eval', float('nan')):.2f} | {n_mfma or 0:.0f} | {flop / 1e9:.2f} | {tf:.2f} | {tf / 78.6:.1%} | "
                           f"{n_valu or 0:.0f} | {n_lds or 0:.0f} |")
            batched[f"{cfg}_{path}"] = {"scan_avg_us_rocprof": t_us, "us_per_eval": run.get("us_per_eval"), "mfma_instructions_per_launch": n_mfma, "mfma_mops_f64_counter": mops,
                                        "mfma_tflops": tf, "mfma_utilisation_of_78.6": tf / 78.6, "valu_instructions_per_launch": n_valu, "lds_instructions_per_launch": n_lds}
            # what the batched scan moves from beyond L2 (parametric config 2: one load per sample against one grid row per point)
            fsz = mean_counter(os.path.join(src, name, "fetch", "*", "*_counter_collection.csv"), key, "FETCH_SIZE")
            wsz = mean_counter(os.path.join(src, name, "write", "*", "*_counter_collection.csv"), key, "WRITE_SIZE")
            if fsz is not None and wsz is not None:
                alg = 8.0 * 4 * run.get("samples", 0)  # config 2: four fp64 scalars per sample
                hbm = (2 * fsz + wsz) * 1024
                batched[f"{cfg}_{path}"].update(fetch_size_kib=fsz, write_size_kib=wsz, bytes_beyond_l2_per_launch=hbm, algorithmic_bytes_one_pass=alg,
                                                traffic_over_one_pass=(hbm / alg) if alg else None)
                p_lines.append(f"| {cfg} | {path} | {short[:70]} | {t_us:.1f} | {run.get('us_per_eval', float('nan')):.2f} | {fsz:.0f} | {wsz:.0f} | {hbm / 1e6:.1f} | {alg / 1e6:.2f} | {hbm / alg if alg else float('nan'):.2f} |")
    if batched:
        traffic["batched_k16"] = batched
    lines += ["", "LDS pipe of the scan kernel (`--pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES`):", ""] + lds_lines
    lines += ["", "Batched launches, K = 16 hyper-parameter points (tools/batch_run.py; paths: `mfma` = the default for spline models, gwi_mfma.h; `taps` = GWI_BATCH_MFMA=0; `rows` = GWI_BATCH_ROWS=1). MFMA FLOP = 2048 x SQ_INSTS_MFMA; utilisation = that rate over the 78.6 TFLOP/s fp64 matrix peak (SQ_INSTS_VALU includes the matrix instructions):", ""] + b_lines
    lines += ["", "Scan-kernel SQ counters (separate `--pmc` passes; fp64 FLOP = 64 x (ADD + MUL + 2 FMA + TRANS) wave-instructions):", ""] + sq_lines
    if len(p_lines) > 2:
        lines += ["", "Batched launches of the parametric config 2, K = 16 points: `pbatch` = scan_pbatch_kernel (GWI_PBATCH=1), a tile's samples loaded once for the run of points a workgroup draws (balanced mode: (tile, point) units dealt out evenly to one round of workgroups), "
                  "`rows4` = the same kernel with round 5's four grid rows of four points, `rowsperpoint` = one grid row per point (scan_kernel BATCH: the default since round 6).  Bytes from beyond L2 per LAUNCH against the algorithmic bytes of one pass over the catalog:", ""] + p_lines
    # ---- chains compiled at gwi_create (hipRTC) and the generic kernel on the same configurations
    c_lines = ["| config | scan kernel | how | calls | scan avg us (rocprofv3) | algorithmic GB/s | frac of 8 TB/s | evals/s (untraced) |", "|---|---|---|---|---|---|---|---|"]
    alg_scalars = {"c1": 10, "c2": 4, "c3": 8, "c5": 9, "b50k": 8, "def50k": 9}
    for name in sorted(c for c in os.listdir(src) if c.startswith("chain_")):
        _, cfg, mode = name.split("_")
        stats = glob.glob(os.path.join(src, name, "trace", "*", "*_kernel_stats.csv"))
        if not stats:
            continue
        shutil.copy(stats[0], os.path.join(dst, f"{name}_kernel_stats.csv"))
        bench = {}
        try:
            bench = json.loads(open(os.path.join(src, name, "bench.json")).read().strip().splitlines()[-1])
            shutil.copy(os.path.join(src, name, "bench.json"), os.path.join(dst, f"{name}_bench.json"))
        except Exception:
            pass
        n_samples = bench.get("config", {}).get("n_events", 0) * bench.get("config", {}).get("n_pe", 0) + bench.get("config", {}).get("n_inj", 0)
        for r in csv.DictReader(open(stats[0])):
            if "scan_kernel" not in r["Name"]:
                continue
            t_us = float(r["AverageNs"]) / 1e3
            gbs = 8.0 * alg_scalars.get(cfg, 0) * n_samples / (t_us * 1e-6) / 1e9 if n_samples else float("nan")
            how = "compiled at gwi_create (hipRTC, GWI_FORCE_JIT=1)" if mode == "jit" else "generic kernel: run-time term loop (GWI_FORCE_GENERIC=1)"
            c_lines.append(f"| {cfg} | {r['Name'].split('(')[0].replace('void ', '')[:70]} | {how} | {r['Calls']} | {t_us:.2f} | {gbs:.0f} | {gbs / 8000.0:.3f} | {bench.get('value', float('nan')):.0f} |")
            traffic.setdefault("other_chains", {})[f"{cfg}_{mode}"] = {"scan_avg_us_rocprof": t_us, "algorithmic_gbs": gbs, "frac_of_8tbs": gbs / 8000.0, "evals_per_s": bench.get("value"),
                                                                       "scan_chain": bench.get("scan_chain") or bench.get("roofline", {}).get("kernel")}
    if len(c_lines) > 2:
        lines += ["", "The same single evaluations when the model's term sequence has NO ahead-of-time chain: the chain hipRTC compiles at gwi_create, and the generic kernel that runs where hipRTC is missing:", ""] + c_lines
    with open(os.path.join(dst, "SUMMARY.md"), "w") as fh:
        fh.write(f"# rocprofv3 summary, {round_name}\n\nCommands: tools/profile_round.sh (kernel trace: `rocprofv3 --kernel-trace --stats`; counters: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes).  The traced launches are AQL dispatches from the engine's own queues (gwinferno_amd/csrc/gwi_aql.h); `<config>_bench.json` holds the untraced run of the same command, whose live kernel durations come from the same dispatch timestamps.\n\n")
        fh.write("\n".join(lines) + "\n")
    with open(os.path.join(dst, "traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "round1", sys.argv[2] if len(sys.argv) > 2 else None)
