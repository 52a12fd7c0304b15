"""Compile the scan chain of a model ahead of its first use -- without a GPU.

The scan kernel is a compile-time chain of terms, one instantiation per sorted sequence of term kinds.  The library ships the
chains of the BASELINE configurations and of every model in the reference's tests and examples; any other product of densities
gets its chain from hipRTC when ``gwi_create`` first meets it (gwinferno_amd/csrc/gwi_jit.h; a second or two, then cached under
``$GWI_JIT_CACHE``, default ``~/.cache/gwinferno_amd``).  This module does the same compilation on request, e.g. in an image
build, so that the first engine of a production process finds the code object in the cache -- or so that a machine WITHOUT
libhiprtc can be handed a cache directory filled elsewhere (same library build, same hipRTC: the cache key covers both):

    python -m gwinferno_amd.precompile 2 3 6 7 7 7 7           # kinds as gwi_create prints them (GWI_TERM_* numbers, ascending)
    python -m gwinferno_amd.precompile --samples-per-lane 1 3 6 7 7 7 7 7 7
    python -m gwinferno_amd.precompile --mfma 5 5 6 107 107    # the batched matrix-core kernel: kind + 100 x 16-basis gradient tiles

The cache directory must belong to the user and be writable by nobody else (a directory others could have prepared is ignored).
"""
import argparse

SPLINE_KINDS = (7, 9, 14)
MAX_TERMS = 12
KNOWN_KINDS = range(1, 15)


def default_samples_per_lane(kinds):
    """The engine's own rule: two samples per lane, one from six spline terms on (the register budget of BASELINE config 5)."""
    return 1 if sum(k % 100 in SPLINE_KINDS for k in kinds) >= 6 else 2


def check_kinds(kinds, mfma=False):
    kinds = [int(k) for k in kinds]
    if not 1 <= len(kinds) <= MAX_TERMS:
        raise ValueError(f"1 to {MAX_TERMS} terms")
    base = [k % 100 if mfma else k for k in kinds]
    if any(k not in KNOWN_KINDS for k in base):
        raise ValueError("term kinds are the GWI_TERM_* numbers 1..14 of include/gwi_engine.h")
    if base != sorted(base):
        raise ValueError("kinds must be in ascending order (the host sorts a model's terms by kind)")
    return kinds


def main(argv=None):
    from . import _native as N

    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("kinds", nargs="+", type=int)
    ap.add_argument("--samples-per-lane", type=int, default=None, choices=(1, 2))
    ap.add_argument("--mfma", action="store_true", help="the batched matrix-core kernel of a spline model (kinds carry 100 x gradient tiles)")
    a = ap.parse_args(argv)
    kinds = check_kinds(a.kinds, a.mfma)
    got = N.jit_compile(kinds, 0 if a.mfma else (a.samples_per_lane or default_samples_per_lane(kinds)))
    where = got["path"] or "(no trusted cache directory: compiled for this process only)"
    print(f"{'found in the cache' if got['from_cache'] else 'compiled in %.2f s' % got['compile_seconds']}: {where}")
    return got


if __name__ == "__main__":
    main()
