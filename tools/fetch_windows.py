#!/usr/bin/env python3
"""Which conditional branches of a decode kernel's byte loop break the fetch-window rule (DESIGN 2.1 "Round 4, second half",
zh_cm_fast.h rule 2): on gfx950 instruction fetch does not run ahead across a conditional branch, and a NOT-TAKEN branch
whose next two instructions do not lie completely inside the branch's own 32-byte window costs a lone wave ~19 cycles.

  python tools/fetch_windows.py [--lib zpaqsharp_amd/libzpaqhip.so] [--list]        (no GPU needed: llvm-objdump)

Per kernel: the byte loop as tools/count_instr.py finds it, its conditional branches in layout order (forward ones: the hot
path falls through them; the loop's own backward branch is taken and not counted), and how many of them have their next
one / next two instructions outside their window.  For the hand-laid loop of zh_cm_fast.h the answer is pinned by
tests/test_abi_and_framing.py; for the compiled chain kernels it is what the compiler happened to produce (shifting the
whole kernel moves mid / min by under 1 %: profiles/r04/ab_notes.txt call 29) and tells where a hand-written bit would
have to place its blocks."""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import count_instr  # noqa: E402


def disassemble_sized(lib):
    """{symbol: [(addr, mnemonic, operands, bytes)]} — as count_instr.disassemble, with the encoding's length."""
    tmp = tempfile.mkdtemp(prefix="zh_fw_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([count_instr.OBJDUMP, "--offloading", so], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        funcs = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([count_instr.OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in txt.split("\n"):
                m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
                if m:
                    cur = funcs.setdefault(m.group(2), [])
                    continue
                m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*((?:[0-9A-Fa-f]{8}\s*)+)", line)
                if m and cur is not None:
                    cur.append((int(m.group(3), 16), m.group(1), m.group(2), 4 * len(m.group(4).split())))
        return funcs
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "zpaqsharp_amd", "libzpaqhip.so"))
    ap.add_argument("--list", action="store_true", help="print every offending branch")
    a = ap.parse_args()
    funcs = disassemble_sized(a.lib)
    for tag, (sym, nsteps, nbytes) in count_instr.KERNELS.items():
        ins = funcs.get(sym)
        if not ins:
            print(f"{sym}: not in the library")
            continue
        found = count_instr.byte_loop([(x[0], x[1], x[2]) for x in ins], nsteps)
        if not found:
            print(f"{sym}: no byte loop found")
            continue
        lo, hi, _ = found
        body = ins[lo:hi + 1]
        n = one = two = 0
        for i, x in enumerate(body[:-2]):
            if not x[1].startswith("s_cbranch"):
                continue
            n += 1
            off = x[0] % 32
            a1, a2 = body[i + 1], body[i + 2]
            bad1 = off + 4 + a1[3] > 32
            bad2 = off + 4 + a1[3] + a2[3] > 32
            one += bad1
            two += bad2
            if a.list and bad2:
                print(f"  {x[0]:#x} (+{off:2d}) {x[1]} -> {a1[1]} ({a1[3]} B), {a2[1]} ({a2[3]} B){'   next instruction already outside' if bad1 else ''}")
        print(f"{sym}: {n // nbytes} conditional branches per byte in the loop's range; next instruction outside the branch's 32-byte window: "
              f"{one / nbytes:.1f} per byte, one of the next two outside: {two / nbytes:.1f} per byte")


if __name__ == "__main__":
    main()
