#!/usr/bin/env python3
"""Diagnostic (GPU box): read bandwidth of gwi_hbm_bandwidth's sweep as a function of the array size -- from arrays that stay in
the 256 MB Infinity Cache between launches to arrays far beyond it (what rate can a streaming kernel hope for at the sizes of the
BASELINE catalogs: 12.6 / 28.5 / 180 MB)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gwinferno_amd.engine import hbm_bandwidth  # noqa: E402

for mb in (12, 28, 64, 128, 180, 256, 512, 1024):
    n = mb * (1 << 20) // 8
    r, t = hbm_bandwidth(0, n_doubles=n, iters=20)
    print(f"{mb:5d} MB per array: read sweep {r:8.1f} GB/s   triad (3 arrays) {t:8.1f} GB/s", flush=True)
