import os,sys,time,numpy as np
sys.path.insert(0,'.')
from gwinferno_amd.compositions import COMPOSITIONS, draw_params
from gwinferno_amd.synthetic import make_config_catalog
pe,inj,total=make_config_catalog("c3")
for env in ({"GWI_BATCH_MFMA":"1"},{"GWI_BATCH_ROWS":"1"},{"GWI_BATCH_MFMA":"0"}):
    for k in ("GWI_BATCH_MFMA","GWI_BATCH_ROWS"): os.environ.pop(k,None)
    os.environ.update(env)
    comp=COMPOSITIONS["bspline_defaults"](pe,inj); eng=comp.engine()
    rng=np.random.default_rng(1)
    tb=np.stack([comp.theta(draw_params("bspline_defaults",rng)) for _ in range(16)])
    vg=eng.configure_batch(16,total,min_neff_cut=False)
    for _ in range(5): vg(tb)
    t0=time.perf_counter()
    for _ in range(40): vg(tb)
    dt=time.perf_counter()-t0
    print(env, eng.batch_path(16), "us/eval", 1e6*dt/(40*16)); eng.close()
