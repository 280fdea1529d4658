#!/usr/bin/env python3
"""Static instruction counts of the decode kernels' hot loops, from the gfx950 code objects inside libzpaqhip.so.

  python tools/count_instr.py [--out profiles/<round>]      (needs no GPU: llvm-objdump on the built library)

For every decode kernel: the code objects are taken out of the library (llvm-objdump --offloading), disassembled
(llvm-objdump -d), and the BYTE LOOP is located as the innermost loop (a backward branch and its target) that contains the
arithmetic-decoder steps of a byte — the EOS flag and 8 bits, each one `s_mul_hi_u32` of Decoder.decode's range
split (Decoder.cs:136-158; zh_dev.h ZH_DEC_STEP, zh_cm_fast.h ZH_FAST_STEP; the hand-written loop of zh_cm_fast.h
decodes the EOS flag, whose probability is 0, with a compare: 8 multiplies there).  Counted: the instructions of that address range
in layout order, without `s_nop`.  LLVM lays blocks it thinks unlikely out of line, behind the loop, so the range is
the hot path plus the (few) cold blocks the compiler left inside; loops nested inside the range (renormalisation, polls)
count once.  The number is an upper bound of what a byte executes on the common path and a lower bound of nothing: it
is reported as `instr_per_byte_static`, with the per-step distances between consecutive decoder steps beside it.

Writes profiles/<round>/instr_<kernel>.json with the hash of the kernel sources (bench.source_hash): bench.py quotes a
file only when that hash is the one of the sources it runs with."""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# kernel symbol, decoder steps with a multiply inside the byte loop (zh_cm_fast.h decodes the EOS flag, p = 0, with a compare)
# ... and the bytes one pass of the loop decodes (zh_cm_fast.h lays its body out twice)
# (min / mid: the assembly loop of nb_fast<model> in zh_nibble.hip — a device function, found by its mangled name; its EOS flag is
# a compare like zh_cm_fast.h's: 8 multiplies per byte)
KERNELS = {"l1": ("zh_decode_cm", 16, 2, None), "min": ("nb_fast_min", 8, 1, "nb_fastINS_5C2MinELb0E"), "mid": ("nb_fast_mid", 8, 1, "nb_fastINS_5C2MidELb0E"),
           "max": ("zh_decode_c2_max", 9, 1, None)}


def disassemble(lib):
    """{symbol: [(addr, mnemonic, operands)]} over all gfx950 code objects of the library."""
    tmp = tempfile.mkdtemp(prefix="zh_instr_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, "--offloading", so], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        funcs = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in txt.split("\n"):
                m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
                if m:
                    cur = funcs.setdefault(m.group(2), [])
                    continue
                m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
                if m and cur is not None:
                    cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
        return funcs
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def byte_loop(ins, nsteps=9):
    """(first, last) instruction indices of the innermost loop around `nsteps` consecutive decoder steps, and the steps."""
    steps = [i for i, (_, mn, _) in enumerate(ins) if mn == "s_mul_hi_u32"]
    addr_ix = {a: i for i, (a, _, _) in enumerate(ins)}
    loops = []
    for i, (a, mn, ops) in enumerate(ins):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            mt = re.match(r"^(\d+)", ops)               # raw simm16: dwords relative to the next instruction
            if not mt:
                continue
            rel = int(mt.group(1))
            rel = rel - 65536 if rel >= 32768 else rel
            tgt = a + 4 + 4 * rel
            if tgt in addr_ix and addr_ix[tgt] <= i:
                loops.append((addr_ix[tgt], i))
    # the byte loop = the loop closed by the FIRST backward branch behind the last step that encloses all the steps and
    # no other decoder step (backward branches of polling / retry paths laid out behind the loop close larger ranges)
    best = None
    for w in range(len(steps) - nsteps + 1):
        grp = steps[w:w + nsteps]
        mine = None
        for lo, hi in loops:
            if lo <= grp[0] and grp[-1] <= hi:
                n_in = sum(1 for s_ in steps if lo <= s_ <= hi)
                if n_in == nsteps and (mine is None or hi < mine[1] or (hi == mine[1] and lo > mine[0])):
                    mine = (lo, hi, grp)
        # several groups (the hand-written loop of zh_cm_fast.h and the C++ body of the same loop): the tightest loop is the hot one
        if mine and (best is None or mine[1] - mine[0] < best[1] - best[0]):
            best = mine
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "zpaqsharp_amd", "libzpaqhip.so"))
    ap.add_argument("--out", default=None, help="directory for instr_<kernel>.json (nothing is written without it)")
    a = ap.parse_args()
    import bench
    funcs = disassemble(a.lib)
    if a.out:
        os.makedirs(a.out, exist_ok=True)
    for tag, (sym, nsteps, nbytes, mangled) in KERNELS.items():
        ins = funcs.get(sym) if not mangled else next((v for k, v in funcs.items() if mangled in k), None)
        if not ins:
            print(f"{sym}: not in the library")
            continue
        found = byte_loop(ins, nsteps)
        if not found:
            print(f"{sym}: no loop around {nsteps} decoder steps found ({sum(1 for x in ins if x[1] == 's_mul_hi_u32')} s_mul_hi_u32)")
            continue
        lo, hi, grp = found
        body = [x for x in ins[lo:hi + 1] if x[1] != "s_nop"]
        per_step = [sum(1 for x in ins[grp[k]:grp[k + 1]] if x[1] != "s_nop") for k in range(nsteps - 1)]
        kinds = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "smem": 0, "branch": 0, "waitcnt": 0}
        for _, mn, _ in body:
            k = ("waitcnt" if mn.startswith("s_waitcnt") else "branch" if mn.startswith("s_cbranch") or mn == "s_branch" else
                 "smem" if mn.startswith("s_load") or mn.startswith("s_buffer") or mn.startswith("s_memtime") else
                 "salu" if mn.startswith("s_") else "lds" if mn.startswith("ds_") else
                 "vmem" if mn.startswith(("buffer_", "global_", "flat_", "scratch_")) else "valu")
            kinds[k] += 1
        rec = {"kernel": sym, "model": tag, "src_hash": bench.source_hash(tag), "instr_per_byte_static": len(body) // nbytes, "bytes_per_loop_pass": nbytes,
               "instr_between_decoder_steps": per_step, "loop_bytes": ins[hi][0] - ins[lo][0], "mix": kinds,
               "how": f"tools/count_instr.py: innermost loop around the {nsteps} s_mul_hi_u32 decoder steps of a byte, layout order, s_nop excluded"}
        if a.out:
            with open(os.path.join(a.out, f"instr_{sym}.json"), "w") as f:
                json.dump(rec, f, indent=1)
        print(f"{sym}: {len(body) // nbytes} instructions per byte (static), between decoder steps {per_step}, {kinds}")


if __name__ == "__main__":
    main()
