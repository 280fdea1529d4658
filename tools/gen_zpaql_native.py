#!/usr/bin/env python3
"""Ahead-of-time translation of known ZPAQL programs to HIP device functions.

The reference's default build does not interpret ZPAQL: `ZPAQL.assemble()` and
`Predictor.assemble_p()` emit x86 code for the block's HCOMP at run time
(ZPAQL.cs:353-1008, "JIT").  A GPU cannot run an x86 JIT, and interpreting ZPAQL on a
single wavefront costs hundreds of cycles per instruction, so the programs this repo
knows (zpaqsharp_amd/models.py: the HCOMP of min / mid / max and the E8E9 PCOMP) are
translated here, once, at build time, into straight C++ that hipcc compiles to native
gfx950 code.  Any other program still runs on the interpreter (zh_core.h vm_run); the
host picks the native routine only when the program bytes match exactly.

Semantics follow ZPAQL.cs:1028-1251 instruction by instruction (one statement per
ZPAQL instruction, labels at every instruction start, jumps as `goto`).

Output: zpaqsharp_amd/csrc/zh_zpaql_native.h   (regenerate: python tools/gen_zpaql_native.py)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from zpaqsharp_amd import models, zpaql  # noqa: E402

# Every value of a ZPAQL machine is wave-uniform on the device; ZH_UNI (v_readfirstlane) tells the compiler so at the
# places where it cannot see it (function entry, memory reads), and the whole program then runs on the scalar unit.
SRC = ["a", "b", "c", "d", "zh_uni<MP>((uint32_t)M[b & mmask])", "zh_uni<MP>((uint32_t)M[c & mmask])", "zh_uni<MP>(H[d & hmask])"]
ALU = ["a += {s};", "a -= {s};", "a *= {s};", "{{ uint32_t s_ = {s}; a = s_ ? a / s_ : 0; }}",
       "{{ uint32_t s_ = {s}; a = s_ ? a % s_ : 0; }}", "a &= {s};", "a &= ~({s});", "a |= {s};", "a ^= {s};",
       "a <<= (({s}) & 31);", "a >>= (({s}) & 31);", "f = a == ({s});", "f = a < ({s});", "f = a > ({s});"]


def store(ddd: int, val: str) -> str:
    if ddd < 4:
        return f"{'abcd'[ddd]} = {val};"
    if ddd == 4:
        return f"M[b & mmask] = (uint8_t)({val});"
    if ddd == 5:
        return f"M[c & mmask] = (uint8_t)({val});"
    return f"H[d & hmask] = {val};"


def reachable(code: bytes):
    """Decode positions control can reach from pc 0 (a jump may land inside a 2-byte instruction, which ZPAQL defines
    as decoding from there), in address order."""
    n = len(code)

    def length(pc):
        op = code[pc]
        return 3 if op == 255 else 2 if op & 7 == 7 else 1

    valid, work = set(), [0]
    while work:
        pc = work.pop()
        if pc in valid or not 0 <= pc < n:
            continue
        valid.add(pc)
        op = code[pc]
        arg = code[pc + 1] if pc + 1 < n else 0
        nxt = pc + length(pc)
        if op in (39, 47):
            work += [nxt, nxt + ((arg + 128) & 255) - 128]
        elif op == 63:
            work.append(nxt + ((arg + 128) & 255) - 128)
        elif op == 255:
            work.append(arg + 256 * (code[pc + 2] if pc + 2 < n else 0))
        elif op == 56 or zpaql.is_error_op(op):
            pass
        else:
            work.append(nxt)
    return sorted(valid)


def free_immediates(code: bytes):
    """pcs of the reachable instructions whose second byte is a plain numeric operand (`a= N`, `a+= N`, `a> N` ...: not a
    jump distance, not an R index) and is not itself a decode position: the bytes that vary between the programs the
    reference generates from one template (LibZPAQ.cs:427-826) without changing their structure."""
    valid = set(reachable(code))
    return [pc for pc in sorted(valid) if code[pc] >= 64 and code[pc] != 255 and code[pc] & 7 == 7 and pc + 1 < len(code)
            and pc + 1 not in valid]


def translate(code: bytes, name: str, free=None) -> str:
    """code = program bytes including the trailing END 0.  free: pcs whose numeric operand is read from the program at
    run time (imm, one lane per operand) instead of being folded into the code."""
    n = len(code)
    free_ix = {pc: k for k, pc in enumerate(free or [])}

    def length(pc):
        op = code[pc]
        return 3 if op == 255 else 2 if op & 7 == 7 else 1

    def targets(pc):
        """pcs control can reach from the instruction at pc (inside the program only)."""
        op = code[pc]
        arg = code[pc + 1] if pc + 1 < n else 0
        nxt = pc + length(pc)
        if op in (39, 47):
            return [nxt, nxt + ((arg + 128) & 255) - 128]
        if op == 63:
            return [nxt + ((arg + 128) & 255) - 128]
        if op == 255:
            return [arg + 256 * (code[pc + 2] if pc + 2 < n else 0)]
        if op == 56 or zpaql.is_error_op(op):
            return []
        return [nxt]

    # every decode position control can reach (a jump may land inside a 2-byte instruction,
    # which ZPAQL defines as decoding from there), in address order
    valid, work = set(), [0]
    while work:
        pc = work.pop()
        if pc in valid or not 0 <= pc < n:
            continue
        valid.add(pc)
        work.extend(targets(pc))
    starts = sorted(valid)

    def jump(target: int, back_from: int) -> str:
        if not 0 <= target < n:
            return "return ZH_E_ZPAQL;"            # lands in the zero padding: opcode 0 = error
        pre = "if (budget-- == 0) return ZH_E_BUDGET; " if target <= back_from else ""
        return f"{pre}goto L{target};"

    # R registers the program names (constant indices): in the post-processor form they live in locals for the run —
    # read at entry, the written ones stored back at every exit — instead of an LDS round trip per `a=r N` / `r=a N`.
    r_read = sorted({code[pc + 1] for pc in starts if code[pc] < 64 and code[pc] & 7 == 7 and (code[pc] >> 3) < 4 and pc + 1 < n})
    r_written = sorted({code[pc + 1] for pc in starts if code[pc] < 64 and code[pc] & 7 == 7 and (code[pc] >> 3) == 6 and pc + 1 < n})
    r_local = free is not None
    out = [f"// {name}: {' '.join(zpaql.disassemble_code(code[:-1]))}",
           "template <class MP, class HP>",
           f"ZH_HD inline __attribute__((always_inline)) int zh_native_{name}(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d, uint32_t &f, uint32_t input,",
           "    MP M, uint32_t mmask, HP H, uint32_t hmask, uint32_t *R, zhcore::Sink *out, uint64_t budget"
           + (", const ZhImm &imm) {" if free is not None else ") {"),
           "  a = zh_uni<MP>(input); b = zh_uni<MP>(b); c = zh_uni<MP>(c); d = zh_uni<MP>(d); f = zh_uni<MP>(f);",
           "  (void)R; (void)out; (void)budget; (void)f;"]
    if r_local:
        for k in sorted(set(r_read) | set(r_written)):
            out.append(f"  uint32_t R{k} = zh_uni<MP>(R[{k}]);")
        out.append("  int rc_ = 0;")
        out.append("  zhcore::Sink sk_ = *out;                        // the Writer's cursor in registers for the run (out is never null for a PCOMP)")
    for pc in starts:
        op = code[pc]
        arg = code[pc + 1] if pc + 1 < n else 0
        nxt = pc + length(pc)
        st = None
        if op < 64:
            ddd, x = op >> 3, op & 7
            if x == 7:
                off = ((arg + 128) & 255) - 128
                if ddd < 4:
                    st = f"{'abcd'[ddd]} = R{arg};" if r_local else f"{'abcd'[ddd]} = zh_uni<MP>(R[{arg}]);"
                elif ddd == 4:
                    st = f"if (f) {{ {jump(nxt + off, pc)} }}"
                elif ddd == 5:
                    st = f"if (!f) {{ {jump(nxt + off, pc)} }}"
                elif ddd == 6:
                    st = f"R{arg} = a;" if r_local else f"R[{arg}] = a;"
                else:
                    st = jump(nxt + off, pc)
            elif ddd == 7:
                st = {0: "return 0;", 1: "if (out) zhcore::sink_put(*out, a & 255);",
                      3: "a = (a + zh_uni<MP>((uint32_t)M[b & mmask]) + 512u) * 773u;",
                      4: "H[d & hmask] = (zh_uni<MP>(H[d & hmask]) + a + 512u) * 773u;"}.get(x, "return ZH_E_ZPAQL;")
            elif x > 4 or op == 0:
                st = "return ZH_E_ZPAQL;"
            else:
                v = SRC[ddd]
                if x == 0:                           # <>a ; *b / *c swap the low byte only
                    if ddd in (4, 5):
                        st = f"{{ uint32_t t_ = {v}; {store(ddd, 'a')} a = (a & ~255u) | t_; }}"
                    else:
                        st = f"{{ uint32_t t_ = {v}; {store(ddd, 'a')} a = t_; }}"
                else:
                    expr = {1: f"{v} + 1", 2: f"{v} - 1", 3: f"~{v}", 4: "0u"}[x]
                    st = store(ddd, expr)
        elif op == 255:
            tgt = code[pc + 1] + 256 * (code[pc + 2] if pc + 2 < n else 0)
            st = "return ZH_E_ZPAQL;" if tgt >= n else jump(tgt, pc)
        else:
            s = SRC[op & 7] if op & 7 < 7 else (f"zh_imm_get(imm, {free_ix[pc]}u)" if pc in free_ix else f"{arg}u")
            if op < 128:
                ddd = (op >> 3) & 7
                st = "return ZH_E_ZPAQL;" if ddd == 7 else store(ddd, s)
            else:
                k = (op >> 3) & 15
                st = ALU[k].format(s=s) if k < 14 else "return ZH_E_ZPAQL;"
        fall = "" if (st.startswith("return") or st.startswith("goto") or st.startswith("if (budget")) else f" {jump(nxt, -1)}"
        out.append(f"  L{pc}: {st}{fall}  // {zpaql.OPCODES[op] or 'error'}")
    if r_local:
        import re
        out = [re.sub(r"return ([A-Za-z0-9_]+);", r"{ rc_ = \1; goto Lexit; }", ln) if ln.startswith("  L") else ln for ln in out]
        out = [ln.replace("if (out) zhcore::sink_put(*out, a & 255);", "zhcore::sink_put(sk_, a & 255);") for ln in out]
        out.append("  Lexit:")
        out.append("  out->len = sk_.len;")
        for k in r_written:
            out.append(f"  R[{k}] = R{k};")
        out.append("  return rc_;")
    out.append("}")
    return "\n".join(out)


def render_native() -> str:
    progs = [("hcomp_min", models.get("min").header), ("hcomp_mid", models.get("mid").header),
             ("hcomp_max", models.get("max").header)]
    items = []
    for name, header in progs:
        items.append((name, zpaql.parse_header(header)[5]))
    items.append(("pcomp_e8e9", models.get("max+e8e9").pcomp))
    # HCOMP of the reference's level-4 and BWT level-3 method strings (LibZPAQ.makeConfig for `ci1,1,1,1,2am` and `ci1`:
    # the text does not depend on the block-size argument).  zh_nibble.hip runs them on its helper wave.
    from tools import methods
    items.append(("hcomp_m4", zpaql.parse_header(methods.model_of("x0,0ci1,1,1,1,2am")[0].header)[5]))
    items.append(("hcomp_m3", zpaql.parse_header(methods.model_of("x0,3ci1")[0].header)[5]))
    items.append(("hcomp_m4w", zpaql.parse_header(methods.model_of("x0,0ci1,1,1,1,2awm")[0].header)[5]))
    # level 3's LZ77 + CM model (`...,1c0,0,511i2`): the parse state lives in R1 / R2; one immediate follows the PCOMP's length
    # (111 without, 168 with the E8E9 pass) — two exact programs, whatever the block-size argument
    items.append(("hcomp_m2", zpaql.parse_header(methods.model_of("x0,2,12,0,7,21,1c0,0,511i2")[0].header)[5]))
    items.append(("hcomp_m2e", zpaql.parse_header(methods.model_of("x0,6,12,0,7,21,1c0,0,511i2")[0].header)[5]))
    # ... and level 4's form for barely compressible data (`...,5,0,7,..,1c0,0,511`): the same program without the ISSE's context
    items.append(("hcomp_m2s", zpaql.parse_header(methods.model_of("x0,2,5,0,7,21,1c0,0,511")[0].header)[5]))
    items.append(("hcomp_m2se", zpaql.parse_header(methods.model_of("x0,6,5,0,7,21,1c0,0,511")[0].header)[5]))
    lines = ["// zh_zpaql_native.h — GENERATED by tools/gen_zpaql_native.py from zpaqsharp_amd/models.py; do not edit.",
             "// Native (ahead-of-time translated) forms of the ZPAQL programs this repo knows; see the generator.",
             "#pragma once", "#include <stdint.h>", "#include <string.h>", "", '#include "zh_core.h"', "",
             "#if defined(__HIPCC__)", "#pragma clang diagnostic push", '#pragma clang diagnostic ignored "-Wunused-label"', "#endif",
             "#if defined(__HIP_DEVICE_COMPILE__)", "#define ZH_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))", "#else",
             "#define ZH_UNI(x) ((uint32_t)(x))", "#endif",
             "// The M accessor type decides whether the machine's values are wave-uniform (the decoder's own run: pinned to the",
             "// scalar unit) or per-lane (zh_chain2.hip runs a program for 16 candidate input bytes at once, one per lane).",
             "template <class MP> struct ZhUniform { static constexpr bool value = true; };",
             "template <class MP> ZH_HD inline __attribute__((always_inline)) uint32_t zh_uni(uint32_t x) { return ZhUniform<MP>::value ? ZH_UNI(x) : x; }", ""]
    for name, code in items:
        lines.append(translate(code, name))
        lines.append("")
    lines.append("#if defined(__HIPCC__)\n#pragma clang diagnostic pop\n#endif\n")
    lines.append("// ids stored in ZhModel.kind (HCOMP) / matched on the device (PCOMP); 0 = interpret")
    for i, (name, code) in enumerate(items):
        lines.append(f"#define ZH_NATIVE_{name.upper()} {i + 2}u")
    lines.append("")
    for name, code in items:
        lines.append(f"ZH_HD inline __attribute__((always_inline)) bool zh_native_is_{name}(const uint8_t *p, uint32_t len) {{")
        lines.append(f"  if (len != {len(code)}u) return false;")
        chunks = [" && ".join(f"p[{i}] == {code[i]}" for i in range(k, min(k + 8, len(code)))) for k in range(0, len(code), 8)]
        lines.append("  return " + " &&\n         ".join(chunks) + ";")
        lines.append("}")
    lines.append("// Exact-match lookup (host or device).  Returns 0 when the program is not a known one.")
    lines.append("ZH_HD inline __attribute__((always_inline)) uint32_t zh_native_lookup(const uint8_t *prog, uint32_t len) {")
    for name, code in items:
        lines.append(f"  if (zh_native_is_{name}(prog, len)) return ZH_NATIVE_{name.upper()};")
    lines.append("  return 0;")
    lines.append("}")
    lines.append("// ... among the post-processors only (what a kernel asks when a block's PCOMP has arrived: the comparisons against the")
    lines.append("// HCOMP programs have no business in its code — zh_chain2.hip's max kernel grew from 2 833 to 3 908 instructions per byte")
    lines.append("// when two more HCOMP programs joined the lookup above, round 5)")
    lines.append("ZH_HD inline __attribute__((always_inline)) uint32_t zh_native_pcomp_lookup(const uint8_t *prog, uint32_t len) {")
    for name, code in items:
        if name.startswith("pcomp_"):
            lines.append(f"  if (zh_native_is_{name}(prog, len)) return ZH_NATIVE_{name.upper()};")
    lines.append("  return 0;")
    lines.append("}")
    return "\n".join(lines) + "\n"


# Method strings that cover the structural variants of the reference's generated post-processors (LibZPAQ.cs:427-826):
# lazy2 with / without the separate low offset bits (blocks > 16 MiB), lzpre, bwtrle for blocks up to / above 16 MiB,
# each with and without the E8E9 pass, and E8E9 alone.  Other arguments only change numeric operands.
PCOMP_SAMPLES = ["x0,1,4,0,3,16", "x6,1,4,0,3,24", "x0,5,4,0,3,16", "x6,5,4,0,3,24", "x0,2,12,0,7,16",
                 "x0,6,5,0,3,16c0,0,511", "x0,3ci1", "x0,7ci1", "x5,3ci1", "x5,7ci1", "x0,4ci1"]


def render_pcomp() -> str:
    """zh_zpaql_pcomp.h: the post-processors, matched by STRUCTURE (opcodes, jumps, R indices, length) with their numeric
    operands read from the program the block carries."""
    from tools import methods
    skel = []                                         # (name, code, free pcs, sample method)
    seen = set()
    for mt in PCOMP_SAMPLES:
        code = methods.model_of(mt)[0].pcomp
        free = free_immediates(code)
        assert len(free) <= 64, (mt, len(free))
        fixed = tuple(None if (i - 1) in free else code[i] for i in range(len(code)))
        if fixed in seen:
            continue
        seen.add(fixed)
        cmd = methods.make_config(mt)[0].split("pcomp", 1)[1].split()[0]
        skel.append((f"pcomp_{cmd}_{len(code)}", code, free, mt))
    L = ["// zh_zpaql_pcomp.h — GENERATED by tools/gen_zpaql_native.py from tools/methods.py; do not edit.",
         "// Ahead-of-time translations of the reference's generated post-processors (lazy2, lzpre, bwtrle, e8e9:",
         "// LibZPAQ.cs:427-826).  A block's PCOMP is matched by structure; its numeric operands (`a= N`, `a> N`, ...) are",
         "// taken from the program it carries (ZhImm, filled when the program has been read).  Device only.",
         "#pragma once", "#include <stdint.h>", "", '#include "zh_zpaql_native.h"', "",
         "#pragma clang diagnostic push", '#pragma clang diagnostic ignored "-Wunused-label"', "",
         "// The operands: wave-uniform values kept in vector registers (the kernels have hundreds to spare, and few scalar",
         "// ones), made scalar where they are used.",
         "struct ZhImm { uint32_t v[64]; };",
         "__device__ inline __attribute__((always_inline)) uint32_t zh_imm_get(const ZhImm &m, uint32_t k) {",
         "  return (uint32_t)__builtin_amdgcn_readfirstlane((int)m.v[k]);", "}",
         "__device__ inline __attribute__((always_inline)) void zh_imm_set(ZhImm &m, uint32_t k, uint32_t val) { m.v[k] = val; }", ""]
    for name, code, free, mt in skel:
        L.append(f"// ---- {name}: e.g. method {mt}; {len(free)} operands")
        L.append(translate(code, name, free).replace("ZH_HD inline", "__device__ inline"))
        L.append("")
    L.append("#pragma clang diagnostic pop")
    L.append("")
    for i, (name, code, free, mt) in enumerate(skel):
        L.append(f"#define ZH_{name.upper()} {i + 1}u")
    L.append("")
    L.append("// Structure match.  Returns 0 when the program is not one of the above (it then runs on the interpreter).")
    L.append("__device__ inline uint32_t zh_pcomp_lookup(const uint8_t *p, uint32_t len) {")
    for name, code, free, mt in skel:
        fr = set(pc + 1 for pc in free)
        conds = [f"p[{i}] == {code[i]}" for i in range(len(code)) if i not in fr]
        chunks = [" && ".join(conds[k:k + 8]) for k in range(0, len(conds), 8)]
        L.append(f"  if (len == {len(code)}u &&\n      " + " &&\n      ".join(chunks) + f") return ZH_{name.upper()};")
    L.append("  return 0;")
    L.append("}")
    L.append("// The operands of a matched program, written once (by the wave that has just read the program) to 64 words of LDS.")
    L.append("__device__ inline void zh_pcomp_operands(uint32_t id, const uint8_t *p, uint32_t *words) {")
    L.append("  for (int k = 0; k < 64; ++k) words[k] = 0;")
    L.append("  switch (id) {")
    for name, code, free, mt in skel:
        L.append(f"    case ZH_{name.upper()}:")
        for k, pc in enumerate(free):
            L.append(f"      words[{k}] = p[{pc + 1}];")
        L.append("      break;")
    L.append("    default: break;")
    L.append("  }")
    L.append("}")
    L.append("// One run of a matched program (PostProcessor.write in state 5, PostProcessor.cs:80-83).  A real call: the")
    L.append("// translated programs stay out of the decoder loops' register allocation.  The machine registers travel by value.")
    L.append("struct ZhPcRegs { uint32_t a, b, c, d, f; int rc; };")
    L.append("__device__ __attribute__((noinline)) static ZhPcRegs zh_pcomp_call(uint32_t id, ZhPcRegs r, uint32_t input, uint8_t *M, uint32_t mmask,")
    L.append("    uint32_t *H, uint32_t hmask, uint32_t *R, zhcore::Sink *out, uint64_t budget, const uint32_t *words) {")
    L.append("  ZhImm imm;")
    L.append("#pragma unroll")
    L.append("  for (int k = 0; k < 16; ++k) {                   // (all 64 words: the programs' operand counts differ little)")
    L.append("    const uint4 q = reinterpret_cast<const uint4 *>(words)[k];")
    L.append("    imm.v[4 * k] = q.x; imm.v[4 * k + 1] = q.y; imm.v[4 * k + 2] = q.z; imm.v[4 * k + 3] = q.w;")
    L.append("  }")
    L.append("  switch (id) {")
    for name, code, free, mt in skel:
        L.append(f"    case ZH_{name.upper()}: r.rc = zh_native_{name}(r.a, r.b, r.c, r.d, r.f, input, M, mmask, H, hmask, R, out, budget, imm); break;")
    L.append("    default: r.rc = ZH_E_ZPAQL; break;")
    L.append("  }")
    L.append("  return r;")
    L.append("}")
    return "\n".join(L) + "\n"


OUTPUTS = {"zh_zpaql_native.h": render_native, "zh_zpaql_pcomp.h": render_pcomp}


def main():
    for name, fn in OUTPUTS.items():
        path = os.path.join(ROOT, "zpaqsharp_amd", "csrc", name)
        with open(path, "w") as f:
            f.write(fn())
        print("wrote", path)


if __name__ == "__main__":
    main()
