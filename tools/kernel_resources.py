#!/usr/bin/env python3
"""Registers / scratch / static LDS of the compiled kernels, from the code object's metadata
(llvm-readelf --notes gwinferno_amd/_lib/gwi_kernels.hsaco).  Usage: python tools/kernel_resources.py [substring ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", os.path.join(ROOT, "gwinferno_amd", "_lib", "gwi_kernels.hsaco")], capture_output=True, text=True).stdout
rows = []
for k in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
    g = lambda key: re.search(rf"\.{key}:\s+(\S+)", k).group(1)  # noqa: E731
    rows.append((g("name"), int(g("vgpr_count")), int(g("sgpr_count")), int(g("private_segment_fixed_size")), int(g("group_segment_fixed_size"))))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
want = sys.argv[1:]
print(f"{'vgpr':>5} {'sgpr':>5} {'scratch':>8} {'lds':>6}  kernel")
for (raw, v, s, sc, lds), name in zip(rows, names):
    name = name.replace("void gwi::", "").replace("(gwi::KArgs)", "").replace("(gwi::TailArgs)", "")
    if not want or any(w in name for w in want):
        print(f"{v:5d} {s:5d} {sc:8d} {lds:6d}  {name}")
