"""P-spline smoothing prior (gwinferno/models/bsplines/smoothing.py:8-28).  O(N_basis) arithmetic on
hyper-parameters only -- it never touches sample data, so it stays in the user's Python model; this
implementation works on NumPy arrays and, unchanged, on JAX arrays."""


def apply_difference_prior(coefs, inv_var, degree=1):
    """``-0.5 * inv_var * ||Delta^degree coefs||^2``."""
    d = coefs
    for _ in range(degree):
        d = d[1:] - d[:-1]
    return -0.5 * inv_var * (d * d).sum()
