import sys, time, numpy as np
sys.path.insert(0, '.')
from bench import CONFIGS
from gwinferno_amd.compositions import COMPOSITIONS, draw_params
from gwinferno_amd.synthetic import make_config_catalog
cfg = sys.argv[1]
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
comp = COMPOSITIONS[comp_name](pe, inj)
eng = comp.engine()
rng = np.random.default_rng(0)
ths = np.stack([comp.theta(draw_params(comp_name, rng)) for _ in range(16)])
for rep in range(2):
    for K in (1, 2, 4, 8, 16):
        for _ in range(10): eng.evaluate_batch(ths[:K], total, min_neff_cut=False)
        n = 200
        t0 = time.perf_counter()
        for _ in range(n): eng.evaluate_batch(ths[:K], total, min_neff_cut=False)
        dt = time.perf_counter() - t0
        print(cfg, "K", K, "us/batch %.1f  us/eval %.2f" % (1e6*dt/n, 1e6*dt/n/K), flush=True)
# where a K = 16 batch spends its time: kernel durations (launch-attached events) vs the whole call
eng.set_timing(True)
ks = []
for _ in range(50):
    eng.evaluate_batch(ths, total, min_neff_cut=False)
    ks.append(eng.last_kernel_ms())
print(cfg, "K 16 kernel us [scan, combine, final] (median):", np.round(1e3 * np.median(np.array(ks), axis=0), 1), flush=True)
