"""Add a compiled kernel for a model whose term-kind sequence the library does not have yet.

The scan kernel is a compile-time chain of terms (one instantiation per sorted sequence of term kinds: the BASELINE
configurations, every model of the reference's tests, ...).  A product of population models that is not among them runs on
the generic scan kernel (term kinds read at run time: several times slower) and ``gwi_create`` prints the sequence once;
this appends that sequence to
``gwinferno_amd/csrc/gwi_user_variants.inc`` and rebuilds ``libgwi_engine.so`` + ``gwi_kernels.hsaco`` (hipcc, about a
minute; cross-compiles without a GPU):

    python -m gwinferno_amd.add_variant 2 3 6 7 7 7 7        # kinds as printed by the error (GWI_TERM_* numbers, ascending)
    python -m gwinferno_amd.add_variant --samples-per-lane 1 2 3 6 7 7 7 7 7 7

``--samples-per-lane`` picks the unroll (default: 2, or 1 from six spline terms on -- the register budget of the
BASELINE config-5 kernel)."""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
INC = os.path.join(HERE, "csrc", "gwi_user_variants.inc")
SPLINE_KINDS = (7, 9, 14)
MAX_TERMS = 12
KNOWN_KINDS = range(1, 15)


def variant_line(kinds, samples_per_lane=None):
    kinds = [int(k) for k in kinds]
    if not 1 <= len(kinds) <= MAX_TERMS:
        raise ValueError(f"1 to {MAX_TERMS} terms")
    if any(k not in KNOWN_KINDS for k in kinds):
        raise ValueError("term kinds are the GWI_TERM_* numbers 1..14 of include/gwi_engine.h")
    if kinds != sorted(kinds):
        raise ValueError("kinds must be in ascending order (the host sorts a model's terms by kind)")
    u = samples_per_lane or (1 if sum(k in SPLINE_KINDS for k in kinds) >= 6 else 2)
    if u not in (1, 2):
        raise ValueError("samples per lane: 1 or 2")
    name = "user:" + ",".join(map(str, kinds))
    return f'    GWI_VARIANT_U("{name}", {u}, {", ".join(map(str, kinds))}),\n'


def add(kinds, samples_per_lane=None, rebuild=True):
    line = variant_line(kinds, samples_per_lane)
    text = open(INC).read()
    tag = '"user:' + ",".join(str(int(k)) for k in kinds) + '"'
    if tag in text:
        print(f"{tag} is already in {INC}")
    else:
        with open(INC, "a") as fh:
            fh.write(line)
        print(f"appended to {INC}:\n{line}", end="")
    if rebuild:
        sys.path.insert(0, os.path.dirname(HERE))
        import __graft_entry__ as g

        g.build(force=True)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("kinds", nargs="+", type=int)
    ap.add_argument("--samples-per-lane", type=int, default=None)
    ap.add_argument("--no-build", action="store_true")
    a = ap.parse_args(argv)
    add(a.kinds, a.samples_per_lane, rebuild=not a.no_build)


if __name__ == "__main__":
    main()
