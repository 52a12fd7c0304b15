"""CPU, world_size 2, gloo: the N>1 path -- shard bounds, partial-record layout, the single
all-gather, and the C++ assembly (gwi_combine through a host-only handle) -- against the oracles'
unsharded evaluation (value: NumPy oracle; gradient: C oracle).  The per-rank scan itself (HIP) is replaced by the NumPy BoundModel evaluator;
on the GPU box test_gpu_parity.py::test_partial_records_combine_like_single_device covers the
same path with real device records."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dlogw_dtheta(bm, theta, h=1e-3):
    """d log w / d theta_p of every sample, [n_theta][...]: fourth-order central differences of the NumPy statement of the
    bound model (truncation ~ h^4 f^(5) / 30, rounding ~ 1e-16 / h: both far below the 1e-8 the test asks of the assembled
    gradient).  Excluded samples (log w = -inf) come out as NaN and are given weight 0 by the caller."""
    from bound_eval import log_weights

    d_pe, d_inj = [], []
    for p in range(len(theta)):
        acc_pe = acc_inj = 0.0
        for k, c in ((-2, 1.0), (-1, -8.0), (1, 8.0), (2, -1.0)):
            th = np.array(theta, dtype=np.float64)
            th[p] += k * h
            with np.errstate(all="ignore"):
                lpe, linj, _ = log_weights(bm, th, include_consts=False)
                acc_pe = acc_pe + c * lpe
                acc_inj = acc_inj + c * linj
        d_pe.append(acc_pe / (12.0 * h))
        d_inj.append(acc_inj / (12.0 * h))
    return np.array(d_pe), np.array(d_inj)


def _numpy_partial_record(eng, bm, theta, want_grad=False):
    """What gwi_eval_partial would publish for this rank's shard: the value part and, with ``want_grad``, the gradient
    numerators (record layout of gwi_engine.hip: sum_i G_ip / S1_i over the rank's events, then sum_j w_j e^{-M} dl_j over
    its injections)."""
    from bound_eval import log_weights
    from scipy.special import logsumexp

    lpe, linj, norms = log_weights(bm, theta, include_consts=False)
    e0, e1 = eng.event_range
    j0, j1 = eng.inj_range
    lpe, linj = lpe[e0:e1], linj[j0:j1]
    with np.errstate(all="ignore"):
        lse = logsumexp(lpe, axis=1)
        log_neff = 2 * lse - logsumexp(2 * lpe, axis=1)
        var = 1 / np.exp(log_neff) - 1 / lpe.shape[1]
        M = np.max(linj) if linj.size else -np.inf
        w = np.exp(linj - M) if np.isfinite(M) else np.zeros_like(linj)
    rec = np.zeros(eng.partial_len)
    rec[1] = lse.sum()
    rec[2] = var.sum()
    rec[3] = np.min(np.nan_to_num(log_neff)) if log_neff.size else np.inf
    rec[4], rec[5], rec[6] = M, w.sum(), (w * w).sum()
    rec[7] = e1 - e0
    rec[8 : 8 + len(norms)] = norms
    if want_grad:
        n_theta = len(theta)
        d_pe, d_inj = _dlogw_dtheta(bm, theta)
        with np.errstate(all="ignore"):
            soft = np.exp(lpe - lse[:, None])  # softmax over the samples of an event
            g_pe = np.array([np.sum(np.where(soft > 0, soft * d_pe[p][e0:e1], 0.0)) for p in range(n_theta)])
            g_inj = np.array([np.sum(np.where(w > 0, w * d_inj[p][j0:j1], 0.0)) for p in range(n_theta)])
        off = 8 + len(norms)
        rec[off : off + n_theta] = g_pe
        rec[off + n_theta : off + 2 * n_theta] = g_inj
    return rec, lse, log_neff, var


def _worker(rank, world, port, comp_name, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from gwinferno_amd import _native as N
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.distributed import ShardedLikelihood
    from gwinferno_amd.engine import NativePopulationLikelihood
    from gwinferno_amd.synthetic import make_catalog

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    pe, inj, total = make_catalog(7, 96, 1001, seed=31)  # 7 events -> 4 + 3; 1001 injections -> 501 + 500
    comp = COMPOSITIONS[comp_name](pe, inj)
    p = draw_params(comp_name, np.random.default_rng(9))
    eng = NativePopulationLikelihood(comp.weights(p, True), comp.weights(p, False), comp.hypervolume(p), device=N.DEVICE_HOST_ONLY, rank=rank, world=world)
    theta = eng.bound.theta_of(comp.weights(p, True))

    class _Eng:  # stand-in for the device scan: same record, computed with NumPy
        def __getattr__(self, name):
            return getattr(eng, name)

        def eval_partial(self, th):
            eng.prepare_combine(th)
            return _numpy_partial_record(eng, eng.bound, th, want_grad=True)

    sh = ShardedLikelihood(_Eng(), total)
    res = sh.evaluate(theta, min_neff_cut=False, want_grad=True)
    np.savez(f"{out_path}.{rank}", log_l=res.log_likelihood, grad=res.grad, theta=theta, log_bfs=res.log_bfs, ev=np.array(eng.event_range), log_mu=res.summary.log_det_eff,
             neff_inj=res.summary.log_nEff_inj, var=res.summary.variance_log_likelihood, vt=res.summary.surveyed_hypervolume_norm)
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


@pytest.mark.parametrize("comp_name", ["plpeak", "bspline_test"])
def test_world2_gloo_matches_oracle(tmp_path, comp_name):
    import torch.multiprocessing as mp

    from gwinferno_amd.compositions import draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle import numpy_oracle as O

    out = str(tmp_path / "r")
    mp.spawn(_worker, args=(2, _free_port(), comp_name, out), nprocs=2, join=True)
    pe, inj, total = make_catalog(7, 96, 1001, seed=31)
    p = draw_params(comp_name, np.random.default_rng(9))
    ref = O.COMPOSITIONS[comp_name](pe, inj).evaluate(p, total, min_neff_cut=False)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    for r in (r0, r1):  # every rank assembles the same global result
        assert abs(float(r["log_l"]) - float(ref["log_likelihood"])) < 1e-10 * abs(float(ref["log_likelihood"]))
        assert abs(np.exp(float(r["log_mu"])) / float(ref["detection_efficiency"]) - 1) < 1e-10
        assert abs(float(r["neff_inj"]) / float(ref["log_nEff_inj"]) - 1) < 1e-9
        assert abs(float(r["var"]) / float(ref["variance_log_likelihood"]) - 1) < 1e-8
    assert float(r0["log_l"]) == float(r1["log_l"])  # bit-identical across ranks
    # the gradient half of the records crossed the process boundary too: the assembled d log_l / d theta of both ranks
    # against the C oracle's analytic gradient of the unsharded catalog
    from gwinferno_amd.compositions import COMPOSITIONS
    from oracle.c_oracle import COracle

    comp = COMPOSITIONS[comp_name](pe, inj)
    orc = COracle(comp.engine(device=-2).bound)
    want = orc.evaluate(r0["theta"], total, min_neff_cut=False)
    scale = max(1.0, float(np.max(np.abs(want["grad"]))))
    assert np.any(r0["grad"] != 0.0)
    assert np.max(np.abs(r0["grad"] - want["grad"])) < 1e-8 * scale
    assert np.array_equal(r0["grad"], r1["grad"])
    assert list(r0["ev"]) == [0, 4] and list(r1["ev"]) == [4, 7]
    got = np.concatenate([r0["log_bfs"], r1["log_bfs"]])
    assert np.max(np.abs(got - ref["logBFs"])) < 1e-10


def _shm_worker(rank, world, port, out_path, rounds):
    """Host-only handles, records made with NumPy: exercises gwi_shm_comm_init / gwi_shm_exchange / gwi_combine
    -- the node-local exchange gwi_eval_sharded uses instead of a collective launch -- without a GPU."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from gwinferno_amd import _native as N
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.distributed import init_shared_memory_exchange
    from gwinferno_amd.engine import NativePopulationLikelihood
    from gwinferno_amd.synthetic import make_catalog

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    pe, inj, total = make_catalog(7, 96, 1001, seed=31)
    comp = COMPOSITIONS["plpeak"](pe, inj)
    eng = NativePopulationLikelihood(comp.weights(draw_params("plpeak", np.random.default_rng(9)), True), comp.weights(draw_params("plpeak", np.random.default_rng(9)), False),
                                     None, device=N.DEVICE_HOST_ONLY, rank=rank, world=world)
    name = init_shared_memory_exchange(eng)
    dist.barrier()
    assert not os.path.exists("/dev/shm" + name)  # unlinked once every rank is attached
    rng = np.random.default_rng(5)
    vals = []
    for i in range(rounds):
        p = draw_params("plpeak", rng)  # same stream on every rank
        theta = eng.bound.theta_of(comp.weights(p, True))
        eng.prepare_combine(theta)
        rec, lse, _, _ = _numpy_partial_record(eng, eng.bound, theta)
        if rank == 1 and i % 7 == 3:
            import time

            time.sleep(0.01)  # ranks drift apart; the stamps keep them in step
        gathered = eng.shm_exchange(rec)
        assert np.array_equal(gathered[rank], rec)
        res = eng.combine(gathered, total, nobs=eng.n_ev_global, min_neff_cut=False, want_grad=False)
        vals.append(res.log_likelihood)
    np.save(f"{out_path}.{rank}.npy", np.array(vals))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


@pytest.mark.parametrize("world", [2, 3])
def test_shared_memory_exchange_matches_oracle_and_agrees_across_ranks(tmp_path, world):
    import torch.multiprocessing as mp

    from gwinferno_amd.compositions import draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle import numpy_oracle as O

    out, rounds = str(tmp_path / "s"), 40
    mp.spawn(_shm_worker, args=(world, _free_port(), out, rounds), nprocs=world, join=True)
    got = [np.load(f"{out}.{r}.npy") for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(got[0], got[r])  # every rank assembles identical bits, round after round
    pe, inj, total = make_catalog(7, 96, 1001, seed=31)
    orc = O.COMPOSITIONS["plpeak"](pe, inj)
    rng = np.random.default_rng(5)
    for i in range(rounds):
        ref = float(orc.evaluate(draw_params("plpeak", rng), total, min_neff_cut=False)["log_likelihood"])
        assert abs(got[0][i] - ref) < 1e-10 * abs(ref)
