class _C:
    def __init__(self, *a, **k):
        self.a = a

    def __call__(self, *a, **k):
        return _C(*a, **k)


real = _C()
positive = _C()
real_vector = _C()
unit_interval = _C()


def interval(lo, hi):
    return _C(lo, hi)


def dependent_property(*a, **k):
    def deco(fn):
        return property(fn)

    if len(a) == 1 and callable(a[0]) and not k:
        return property(a[0])
    return deco
