#!/usr/bin/env python3
"""Diagnostic (GPU box): per-wave phase timeline of the scan kernel from s_memrealtime stamps.
Build the stamped library first:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -mllvm -amdgpu-kernarg-preload-count=16 -DGWI_STAMPS -Iinclude \
        gwinferno_amd/csrc/gwi_engine.hip -o gwinferno_amd/_lib/libgwi_engine_stamps.so -ldl
  GWI_ENGINE_LIB=gwinferno_amd/_lib/libgwi_engine_stamps.so python tools/stamp_phases.py c2
Stamps: 0 wave entry, 7 first trip's column loads issued, 1 the argument block's lines arrived, 2 first trip's loads landed,
3 loop done, 6 the waves' sums in place, 4 record written (the stamps stay in registers until then)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
comp = COMPOSITIONS[comp_name](pe, inj)
eng = comp.engine()
th = comp.theta(draw_params(comp_name, np.random.default_rng(0)))
for _ in range(20):
    eng.evaluate(th, total, min_neff_cut=False)
n_words = 1 << 22
buf = (C.c_uint64 * n_words)()
eng.lib.gwi_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
assert eng.lib.gwi_debug_stamps(eng.handle, buf, n_words) == 0
raw = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8)
st_all = raw[:, :5].astype(np.int64)
hw_all = raw[:, 5].copy()
blk = np.arange(len(st_all)) // 4
ok = (st_all[:, 0] > 0) & (st_all[:, 4] > 0)
st, blk, hw = st_all[ok], blk[ok], hw_all[ok]
# blocks are dealt round-robin over the 8 XCDs (block b and b+8 share one); each XCD has its own
# realtime counter phase, so the origin is taken per XCD group
xcd = blk % 8
us = np.zeros(st.shape)
for x in range(8):
    sel = xcd == x
    if sel.any():
        print(f"  xcd-group {x}: first entry tick offset vs global min {(st[sel, 0].min() - st[:, 0].min()) / 100.0:7.2f} us, waves {sel.sum()}")
        us[sel] = (st[sel] - st[sel, 0].min()) / 100.0
print(f"{cfg}: {len(st)} waves; kernel span (first entry -> last record) {us[:, 4].max():.2f} us")
names = ["entry (dispatch skew)", "column loads issued", "argument block arrived", "first loads land (+ theta staging)", "evaluate + accumulate", "record"]
order = [0, 7, 1, 2, 3, 4]
raw_ok = raw[ok].astype(np.int64)
tt = np.zeros((len(st), 6))
for x in range(8):
    sel = xcd == x
    if sel.any():
        tt[sel] = (raw_ok[sel][:, order] - st[sel, 0].min()) / 100.0
for k in range(6):
    d = tt[:, k] - (tt[:, k - 1] if k else 0.0)
    print(f"  {names[k]:36s} median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us   (absolute median end {np.median(tt[:, k]):6.2f})")
# inside the record phase: stamp 6 = the waves' sums in place (after the workgroup barrier); a -DGWI_AB_OLD_RECORD build also
# has 7 = scalar sums reduced (there 6 = the waves' references exchanged)
ext = raw[:, 6:8].astype(np.int64)[ok]
if (ext[:, 0] > 0).all():
    e6 = (ext[:, 0] - st[:, 3]) / 100.0
    if (ext[:, 1] > 0).all() and os.environ.get('GWI_OLD_RECORD_STAMPS'):
        e7 = (ext[:, 1] - ext[:, 0]) / 100.0
        e8 = (st[:, 4] - ext[:, 1]) / 100.0
        print(f"  record phase split (old record): reference exchange {np.median(e6):.2f} (p90 {np.percentile(e6, 90):.2f}), sums through LDS {np.median(e7):.2f} (p90 {np.percentile(e7, 90):.2f}), "
              f"row readout + record stores {np.median(e8):.2f} (p90 {np.percentile(e8, 90):.2f}) us")
    else:
        e8 = (st[:, 4] - ext[:, 0]) / 100.0
        print(f"  record phase split: own sums + wait for the workgroup {np.median(e6):.2f} (p90 {np.percentile(e6, 90):.2f}, min {e6.min():.2f}), cross-wave sum + readout + stores {np.median(e8):.2f} "
              f"(p90 {np.percentile(e8, 90):.2f}) us")
# how long after the last wave's record does the kernel end?  (the dispatch's own end stamp is not visible here; the span is)
print(f"  last wave entry at {us[:, 0].max():.2f} us; last 'loop done' at {us[:, 3].max():.2f}; per-wave lifetime median {np.median(us[:, 4] - us[:, 0]):.2f} us")

# ---- placement: which SIMD of which CU every wave ran on (HW_ID: SIMD_ID bits 5:4, CU_ID 11:8, SH_ID 12, SE_ID 15:13; XCC_ID low bits)
simd = (hw >> np.uint64(4)) & np.uint64(3)
cu = (hw >> np.uint64(8)) & np.uint64(15)
sh = (hw >> np.uint64(12)) & np.uint64(1)
se = (hw >> np.uint64(13)) & np.uint64(7)
xcc = (hw >> np.uint64(32)) & np.uint64(15)
cu_key = (((xcc * np.uint64(8) + se) * np.uint64(2) + sh) * np.uint64(16) + cu).astype(np.int64)
simd_key = cu_key * 4 + simd.astype(np.int64)
wave_in_wg = np.arange(len(st_all))[ok] % 4
dur = us[:, 3] - us[:, 2]  # evaluate + accumulate
print(f"placement: {len(np.unique(cu_key))} CUs, {len(np.unique(simd_key))} SIMDs hold waves; xcc values {sorted(set(xcc.tolist()))}")
cnt = np.bincount(np.unique(simd_key, return_inverse=True)[1])
print("  waves per SIMD: " + ", ".join(f"{k}: {int((cnt == k).sum())} SIMDs" for k in sorted(set(cnt.tolist()))))
for w in range(4):
    sel = wave_in_wg == w
    print(f"  wave {w} of its workgroup: SIMD histogram {np.bincount(simd[sel].astype(np.int64), minlength=4).tolist()}, evaluate median {np.median(dur[sel]):.2f} us")
blocks_of_cu = {}
for b_, c_ in zip(blk, cu_key):
    blocks_of_cu.setdefault(int(c_), set()).add(int(b_))
some = sorted(blocks_of_cu.items())[:6]
for c_, bs in some:
    print(f"  CU {c_}: workgroups {sorted(bs)}")
load = np.zeros(simd_key.max() + 1)
np.add.at(load, simd_key, dur)
print(f"  sum of evaluate time per SIMD: median {np.median(load[load > 0]):.2f}, max {load.max():.2f} us")
# ---- evaluate time along the block index (PE tiles come first, event by event; the injection tiles are the last blocks).  Blocks
# b, b + 256, b + 512, ... share a CU and a SIMD's arbiter prefers its oldest wave: the phase grows from one 256-block group to the next
nb = int(blk.max()) + 1
edges = np.linspace(0, nb, 11).astype(int)
print("  evaluate median by block-index decile: " + " ".join(f"{np.median(dur[(blk >= a) & (blk < b)]):.1f}" for a, b in zip(edges[:-1], edges[1:])))
