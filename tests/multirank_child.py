"""Child process of tests/test_gpu_multirank.py (and of nothing else): ONE rank of an R-rank sharded evaluation with the
partial records exchanged by the engine's own ncclAllGather (RCCL over xGMI, ``gwi_comm_init`` + ``gwi_eval_sharded``) --
the exchange BASELINE.json's north_star names (reference contract: pipeline/analysis.py:78-86 reduces along axis 1 only,
:126-134 is an associative sum).

    python tests/multirank_child.py RANK WORLD PORT OUT_PREFIX CONFIGS

Started by the parent BEFORE the parent has touched a GPU; the rendezvous (and the 128-byte ncclUniqueId) travels over a
gloo process group on 127.0.0.1.  Every rank writes OUT_PREFIX.<rank>.npz with, per configuration: the sharded result
(value, whole gradient, this rank's per-event sites, the number of two-pass repeats), the result of an UNSHARDED engine over
the whole catalog on the same GPU, and -- rank 0 -- the C oracle's value and gradient.  The parent does the asserting.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {  # name -> (composition, catalog, evaluations)
    "c3": ("bspline_iid", "c3", 3),
    "c5": ("bspline_full", "c5", 2),
    "c2": ("plpeak", "c2", 3),
    "small": ("bspline_test", None, 3),  # 7 events x 96 PE x 1001 injections: ragged shards (4 + 3 events, 501 + 500 injections at R = 2)
}


def main(argv):
    rank, world, port, out, configs = int(argv[0]), int(argv[1]), int(argv[2]), argv[3], argv[4].split(",")
    import torch
    import torch.distributed as dist

    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.distributed import init_engine_communicator
    from gwinferno_amd.synthetic import make_catalog, make_config_catalog

    n_dev = torch.cuda.device_count()  # counting devices does not initialise the GPU
    if n_dev < world:
        raise SystemExit(f"{world} ranks need {world} GPUs; {n_dev} visible (RCCL cannot place two ranks of one communicator on one device)")
    dev = rank
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    res = {}
    for cfg in configs:
        comp_name, cat, n_eval = CASES[cfg]
        pe, inj, total = make_config_catalog(cat) if cat else make_catalog(7, 96, 1001, seed=31)
        comp = COMPOSITIONS[comp_name](pe, inj)  # model objects from the GLOBAL arrays (zmin / zmax: parametric.py:114-115)
        eng = comp.engine(device=dev, rank=rank, world=world)
        init_engine_communicator(eng)  # ncclCommInitRank inside the engine; the unique id travels through the gloo group
        full = COMPOSITIONS[comp_name](pe, inj).engine(device=dev)
        rng = np.random.default_rng(17)
        thetas = np.stack([comp.theta(draw_params(comp_name, rng)) for _ in range(n_eval)])
        res[f"{cfg}/theta"] = thetas
        res[f"{cfg}/events"] = np.array(eng.event_range)
        res[f"{cfg}/injections"] = np.array(eng.inj_range)
        for i, th in enumerate(thetas):
            marg = i == n_eval - 1  # the last point also runs the second exchange (squared-weight records of marginalize_selection)
            kw = dict(min_neff_cut=False, marginalize_selection=marg)
            r = eng.evaluate_sharded(th, total, **kw)
            f = full.evaluate(th, total, **kw)
            e0, e1 = eng.event_range
            res[f"{cfg}/{i}/sharded_ll"] = np.array(r.log_likelihood)
            res[f"{cfg}/{i}/sharded_grad"] = np.array(r.grad)
            res[f"{cfg}/{i}/sharded_log_bfs"] = np.array(r.log_bfs)
            res[f"{cfg}/{i}/sharded_log_mu"] = np.array(r.summary.log_det_eff)
            res[f"{cfg}/{i}/sharded_neff_inj"] = np.array(r.summary.log_nEff_inj)
            res[f"{cfg}/{i}/full_ll"] = np.array(f.log_likelihood)
            res[f"{cfg}/{i}/full_grad"] = np.array(f.grad)
            res[f"{cfg}/{i}/full_log_bfs"] = np.array(f.log_bfs[e0:e1])
            res[f"{cfg}/{i}/full_log_mu"] = np.array(f.summary.log_det_eff)
            if rank == 0:
                from oracle.c_oracle import COracle  # the checker (test infrastructure)

                o = COracle(full.bound).evaluate(th, total, **kw)
                res[f"{cfg}/{i}/oracle_ll"] = np.array(o["log_likelihood"])
                res[f"{cfg}/{i}/oracle_grad"] = np.array(o["grad"])
        # the C loop of sequential sharded evaluations (what bench.py times): same values as one by one
        ll_seq, g_seq = eng.evaluate_sequence(thetas[:2], total, min_neff_cut=False)[:2]
        res[f"{cfg}/seq_ll"] = np.array(ll_seq)
        res[f"{cfg}/repeats"] = np.array(eng.two_pass_repeats())
        dist.barrier()
        eng.close()
        full.close()
    np.savez(f"{out}.{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
