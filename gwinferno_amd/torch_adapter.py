"""``torch.autograd`` face of the engine: the log-likelihood as a differentiable function
of a float64 theta tensor, for samplers and optimisers written against PyTorch (Pyro's HMC/NUTS, L-BFGS for a MAP
point, ...).  The reference's caller is ``jit(value_and_grad(potential_fn))`` (tests/inference_test.py:320-326); this
is the same contract in torch terms: forward runs one engine evaluation (value + analytic gradient, one set of
launches), backward multiplies the stored gradient by the incoming cotangent -- no second evaluation.

    ll = log_likelihood(engine, theta, total_inj, min_neff_cut=False)     # 0-d float64 tensor
    ll.backward()                                                          # theta.grad == engine gradient

``theta`` lives on the CPU (the engine takes host hyper-parameters and returns host results; the catalog is what
stays in HBM).  A cut (analysis.py:272-317) returns the reference's ``nan_to_num(-inf)`` value with a zero gradient.
"""
import numpy as np
import torch


class _EngineLogLikelihood(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, vg):
        value, grad = vg(np.ascontiguousarray(theta.detach().cpu().numpy(), dtype=np.float64))
        ctx.save_for_backward(torch.from_numpy(np.array(grad, dtype=np.float64, copy=True)).reshape(theta.shape))
        ctx.theta_meta = (theta.dtype, theta.device)
        return torch.tensor(float(value), dtype=torch.float64)

    @staticmethod
    def backward(ctx, cotangent):
        (grad,) = ctx.saved_tensors
        dtype, device = ctx.theta_meta
        return (cotangent.to(torch.float64) * grad).to(dtype=dtype, device=device), None


_CONFIGURED = {}


def log_likelihood(engine, theta, total_inj, **likelihood_flags):
    """One engine evaluation as a differentiable torch scalar; keyword flags as ``hierarchical_likelihood``'s
    (``nobs``, ``marginalize_selection``, ``min_neff_cut``, ``max_variance_cut``)."""
    if theta.numel() != engine.n_theta:
        raise ValueError(f"theta must have {engine.n_theta} elements")
    key = (id(engine), float(total_inj), tuple(sorted(likelihood_flags.items())))
    vg = _CONFIGURED.get(key)
    if vg is None:
        if len(_CONFIGURED) > 64:
            _CONFIGURED.clear()
        vg = _CONFIGURED[key] = engine.configure(total_inj, **likelihood_flags)
    return _EngineLogLikelihood.apply(theta, vg)
