import numpy as _np


def fori_loop(lower, upper, body_fun, init_val):
    val = init_val
    for i in range(int(lower), int(upper)):
        val = body_fun(i, val)
    return val


def broadcast_shapes(*shapes):
    return _np.broadcast_shapes(*shapes)
