#!/usr/bin/env python3
"""Static check of the scan kernels' hand-issued argument-block loads (gwi_device.h: KernargWarm).

The scan requests the lines of its argument block with `s_load_dword sN, s[a:b], 0x0` written as asm; the compiler does not
know these loads are in flight, so nothing may READ or WRITE a destination register -- and no branch may leave the straight
line -- until the `s_waitcnt lgkmcnt(0)` that settles them.  (Round 6: the normaliser workgroups branched away before the wait
and a reused destination register corrupted a completion stamp now and then.)  This walks the disassembly of a code object and
reports every violation.      python tools/check_kernarg_warm.py [gwinferno_amd/_lib/gwi_kernels.hsaco]"""
import os
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
WARM = re.compile(r"^\s*s_load_dword (s\d+), s\[\d+:\d+\], 0x0\b")
REG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def sregs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(path):
    """Walk every path from a kernel's first hand-issued load until the `s_waitcnt lgkmcnt(0)` that settles them (or the end
    of the program): no instruction on the way may read or write a destination register of a load still in flight."""
    asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
    kernels, name = {}, None
    for raw in asm.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", raw)
        if m:
            name = m.group(2)
            kernels[name] = {"base": int(m.group(1), 16), "ins": []}
            continue
        if name is None or "//" not in raw:
            continue
        text, comment = raw.split("//", 1)
        text = text.strip()
        am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        if not text or not am:
            continue
        tm = re.search(r"<\S+\+0x([0-9a-fA-F]+)>", comment)
        kernels[name]["ins"].append((int(am.group(1), 16), text, int(tm.group(1), 16) if tm else None))
    problems, with_warm = [], 0
    for kname, k in kernels.items():
        ins = k["ins"]
        index_of = {addr: n for n, (addr, _, _) in enumerate(ins)}
        first = next((n for n, (_, t, _) in enumerate(ins) if WARM.match(t)), None)
        if first is None:
            continue
        with_warm += 1
        seen, stack, bad = set(), [(first, frozenset())], set()
        while stack:
            n, flight = stack.pop()
            while n < len(ins):
                key = (n, flight)
                if key in seen:
                    break
                seen.add(key)
                _, text, target = ins[n]
                op = text.split()[0]
                w = WARM.match(text)
                if w:
                    flight = flight | {int(w.group(1)[1:])}
                    n += 1
                    continue
                if op == "s_waitcnt" and "lgkmcnt(0)" in text:
                    break  # settled on this path
                if op == "s_endpgm":
                    break
                touched = sregs(text.split(None, 1)[1] if " " in text else "") & flight
                if touched:
                    bad.add(f"`{text}` touches s{sorted(touched)} while their hand-issued loads are in flight")
                if op.startswith("s_cbranch") and target is not None and k["base"] + target in index_of:
                    stack.append((index_of[k["base"] + target], flight))
                if op == "s_branch" and target is not None and k["base"] + target in index_of:
                    n = index_of[k["base"] + target]
                    continue
                n += 1
        problems += [f"{kname}: {b}" for b in sorted(bad)]
    return with_warm, problems


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, problems = check(sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gwinferno_amd", "_lib", "gwi_kernels.hsaco"))
    print(f"{n} kernels issue argument-block loads by hand; {len(problems)} problem(s)")
    for p in problems[:20]:
        print("  " + p)
    sys.exit(1 if problems else 0)
