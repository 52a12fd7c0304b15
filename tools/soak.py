#!/usr/bin/env python3
"""Diagnostic (GPU box): soak test of the completion-stamp protocol -- many sequential evaluations, single and
batched, checking every result against the first (same theta => same bits for values)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
comp = COMPOSITIONS[comp_name](pe, inj)
eng = comp.engine()
rng = np.random.default_rng(0)
ths = np.stack([comp.theta(draw_params(comp_name, rng)) for _ in range(16)])
vg = eng.configure(total, min_neff_cut=False)
ref = [vg(t)[0] for t in ths]
t0 = time.perf_counter()
bad = 0
for i in range(n):
    v, _ = vg(ths[i & 15])
    bad += v != ref[i & 15]
dt = time.perf_counter() - t0
print(f"{cfg}: {n} sequential evaluations in {dt:.1f} s ({n / dt:.0f}/s), value mismatches: {bad}")
vgb = eng.configure_batch(16, total, min_neff_cut=False)
refb = vgb(ths)[0].copy()
t0 = time.perf_counter()
badb = 0
for i in range(n // 32):
    v, _ = vgb(ths)
    badb += int(np.sum(v != refb))
dt = time.perf_counter() - t0
print(f"{cfg}: {n // 32} batches of 16 in {dt:.1f} s ({(n // 32) * 16 / dt:.0f} evals/s), value mismatches: {badb}; single vs batched max rel diff {np.max(np.abs(refb - np.array(ref)) / np.abs(np.array(ref))):.2e}")

# threaded: one engine per host thread, every thread its own in-library sequence; all results against the single-thread ones
import threading  # noqa: E402

C_THREADS = int(os.environ.get("SOAK_THREADS", "6"))
comps = [COMPOSITIONS[comp_name](pe, inj) for _ in range(C_THREADS)]
engines = [c.engine() for c in comps]
seq = np.concatenate([ths] * 64)  # 1024 points per call
want = np.array(ref * 64)
bad_t = [0] * C_THREADS
calls = max(1, n // (1024 * 4))


def work(k):
    for _ in range(calls):
        ll, g = engines[k].evaluate_sequence(seq, total, min_neff_cut=False)
        bad_t[k] += int(np.sum(ll != want))


t0 = time.perf_counter()
workers = [threading.Thread(target=work, args=(k,)) for k in range(C_THREADS)]
for w in workers:
    w.start()
for w in workers:
    w.join()
dt = time.perf_counter() - t0
print(f"{cfg}: {C_THREADS} threads x {calls * 1024} evaluations in {dt:.1f} s ({C_THREADS * calls * 1024 / dt:.0f} evals/s aggregate), value mismatches per thread: {bad_t}")

# batches in two halves, three engines driven alternately from this thread: every set against the blocking result
halves = [e.configure_batch_async(16, total, min_neff_cut=False) for e in engines[:3]]
n_sets = max(30, n // 32)
bad_h = 0
t0 = time.perf_counter()
halves[0][0](ths)
halves[1][0](ths)
for i in range(n_sets):
    halves[(i + 2) % 3][0](ths)
    v, _ = halves[i % 3][1]()
    bad_h += int(np.sum(v != refb))
for j in range(2):
    v, _ = halves[(n_sets + j) % 3][1]()
    bad_h += int(np.sum(v != refb))
dt = time.perf_counter() - t0
print(f"{cfg}: {n_sets + 2} sets of 16 in two halves, three in flight from one thread, in {dt:.1f} s ({(n_sets + 2) * 16 / dt:.0f} evals/s), value mismatches: {bad_h}")
