#!/usr/bin/env python3
"""Diagnostic (GPU box): does the evaluation latency depend on which NUMA node the calling thread runs on?  Pins the
process to the CPUs of each node in turn (sysfs) and times the library's loop; prints the GPU's own node."""
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402


def cpus_of(text):
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


import torch  # noqa: E402

bdf = torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None
print("allowed cpus:", len(os.sched_getaffinity(0)), "| device pci bus id:", bdf)
for p in sorted(glob.glob("/sys/bus/pci/devices/*/numa_node")):
    dev = os.path.dirname(p)
    try:
        if open(os.path.join(dev, "vendor")).read().strip() == "0x1002" and open(os.path.join(dev, "class")).read().startswith(("0x0302", "0x0380", "0x0300")):
            print("AMD display/accelerator", os.path.basename(dev), "numa_node", open(p).read().strip(), "local_cpulist", open(os.path.join(dev, "local_cpulist")).read().strip())
    except OSError:
        pass
comp_name, cat, _, _ = CONFIGS["c2"]
pe, inj, total = make_config_catalog(cat)
comp = COMPOSITIONS[comp_name](pe, inj)
eng = comp.engine()
th = comp.theta(draw_params(comp_name, np.random.default_rng(0)))
allowed = os.sched_getaffinity(0)
for node in sorted(glob.glob("/sys/devices/system/node/node*")):
    cpus = cpus_of(open(os.path.join(node, "cpulist")).read()) & allowed
    if not cpus:
        continue
    os.sched_setaffinity(0, cpus)
    eng.selftime(th, total, n_iter=300, min_neff_cut=False)
    t = [1e6 * eng.selftime(th, total, n_iter=3000, min_neff_cut=False) for _ in range(3)]
    print(os.path.basename(node), f"{len(cpus)} cpus: us/eval", np.round(t, 2))
os.sched_setaffinity(0, allowed)
