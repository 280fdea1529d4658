#!/usr/bin/env python3
"""Generates model-specialised level code for the lane-per-component kernel.

The reference's default build specialises the predictor per model at run time: an x86
emitter unrolls the component list of the block header into straight-line code
(Predictor.assemble_p, Predictor.cs:579-1356).  The GPU analogue is done ahead of time,
here, for the component lists this repo knows (zpaqsharp_amd/models.py: min / mid / max):
the dependent part of predict() — ISSE / AVG / MIX2 / SSE / MIX evaluated level by level —
is emitted as straight-line code with compile-time lane numbers, so operands move with DPP
row shifts or constant-lane v_readlane instead of a run-time level/descriptor loop.
Any other header runs the generic loop of zh_chain.hip; results are identical either way
(tests/test_gpu_parity.py compares both with the oracle).

Output: zpaqsharp_amd/csrc/zh_chain_spec.h     (regenerate: python tools/gen_chain_spec.py)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from zpaqsharp_amd import models, zpaql  # noqa: E402

T = {n: i for i, n in enumerate(zpaql.COMP_NAMES)}


def levels(comps):
    lv = []
    for i, c in enumerate(comps):
        t = c[0]
        ins = []
        if t == T["avg"]:
            ins = [c[1], c[2]]
        elif t == T["mix2"]:
            ins = [c[2], c[3]]
        elif t == T["mix"]:
            ins = list(range(c[2], c[2] + c[3]))
        elif t in (T["isse"], T["sse"]):
            ins = [c[2]]
        lv.append(1 + max(lv[j] for j in ins) if ins else 0)
    return lv


def operand(var, src, dst):
    """int `var` = p of lane `src`, as seen from lane `dst` (only lane dst uses it)."""
    if 1 <= dst - src <= 15 and src // 16 == dst // 16:
        # same 16-lane row: DPP row_shr:(dst-src) delivers lane src's value in lane dst
        return f"const int {var} = __builtin_amdgcn_update_dpp(0, me.p, 0x{0x110 + dst - src:x}, 0xf, 0xf, false);"
    return f"const int {var} = (int)zhdev::rdlane((uint32_t)me.p, {src}u);"


def gen(name, spec_id, header):
    hh, hm, ph, pm, comps, hcomp = zpaql.parse_header(header)
    n = len(comps)
    lv = levels(comps)
    mixers = [i for i, c in enumerate(comps) if c[0] == T["mix"]]
    types = 0
    for c in comps:
        types |= 1 << c[0]
    out = [f"// ---- {name}: " + "; ".join(f"{i} {zpaql.COMP_NAMES[c[0]]} " + " ".join(map(str, c[1:])) + f" (L{lv[i]})" for i, c in enumerate(comps)),
           f"struct ZhSpec_{name} {{",
           f"  static constexpr uint32_t id = {spec_id}u, n = {n}u, types = 0x{types:x}u, nmix = {len(mixers)}u, depth = {max(lv)}u;",
           "};",
           "template <class LaneT>",
           f"__device__ __forceinline__ void zh_spec_levels_{name}(LaneT &me, uint32_t lane, uint32_t c8, const int16_t *stretch, const uint8_t *arena, uint32_t *sserow) {{",
           "  (void)c8; (void)stretch; (void)arena; (void)sserow;"]
    # An SSE interpolates between two neighbours of the 32-entry row (h + c8) * 32 of its table; which two depends on its
    # input, known only at its level — but the row is known now.  All lanes request it here (lane & 31 = entry), so that
    # the HBM/L2 round trip runs under the levels before it; the entries are parked in LDS just before the level.
    sses = [i for i, c in enumerate(comps) if c[0] == T["sse"]]
    for k, i in enumerate(sses):
        out.append(f"  const uint32_t sse_rb{i} = zhdev::rdlane(((me.h + c8) * 32u) & me.cm_mask, {i}u), sse_of{i} = zhdev::rdlane(me.cmo, {i}u);")
        out.append(f"  const uint32_t sse_pre{i} = reinterpret_cast<const uint32_t *>(arena + sse_of{i})[sse_rb{i} + (lane & 31u)];")
    for level in range(1, max(lv) + 1):
        out.append(f"  // level {level}")
        for i, c in enumerate(comps):
            if lv[i] != level or c[0] == T["mix"]:
                continue
            t = c[0]
            out.append("  {")
            if t == T["isse"]:
                out.append("    " + operand("pj", c[2], i))
                out.append("    const int v = zhcore::clamp2k((__mul24(me.w0, pj) + me.w1 * 64) >> 16);")
                out.append(f"    const bool mine = lane == {i}u;")
                out.append("    me.p = mine ? v : me.p; me.pj = mine ? pj : me.pj;")
            elif t == T["mix2"]:
                out.append("    " + operand("pj", c[2], i))
                out.append("    " + operand("pk", c[3], i))
                out.append("    const int v = (__mul24(me.w0, pj) + __mul24(65536 - me.w0, pk)) >> 16;")
                out.append(f"    const bool mine = lane == {i}u;")
                out.append("    me.p = mine ? v : me.p; me.pj = mine ? pj : me.pj; me.pk = mine ? pk : me.pk;")
            elif t == T["avg"]:
                out.append("    " + operand("pj", c[1], i))
                out.append("    " + operand("pk", c[2], i))
                out.append(f"    const int v = (pj * {c[3]} + pk * {256 - c[3]}) >> 8;")
                out.append(f"    me.p = lane == {i}u ? v : me.p;")
            elif t == T["sse"]:
                out.append("    " + operand("pj", c[2], i))
                out.append(f"    sserow[{sses.index(i) * 32}u + (lane & 31u)] = sse_pre{i};            // the row requested at the top (every entry twice, same value)")
                out.append(f"    if (lane == {i}u) {{                          // Predictor.cs:327-340")
                out.append("      me.pj = pj;")
                out.append("      me.cxt = (me.h + c8) * 32u;")
                out.append("      int pq = pj + 992; pq = pq < 0 ? 0 : pq > 1983 ? 1983 : pq;")
                out.append("      const int wt = pq & 63; pq >>= 6;")
                out.append("      me.cxt += (uint32_t)pq;")
                out.append(f"      const uint32_t e0 = sserow[{sses.index(i) * 32} + pq], e1 = sserow[{sses.index(i) * 32} + pq + 1];")
                out.append("      me.p = stretch[((e0 >> 10) * (uint32_t)(64 - wt) + (e1 >> 10) * (uint32_t)wt) >> 13];")
                out.append("      me.cxt += (uint32_t)(wt >> 5);")
                out.append("      me.w0 = (int)((wt >> 5) ? e1 : e0);")
                out.append("    }")
            out.append("  }")
        for q, mi in enumerate(mixers):
            if lv[mi] != level:
                continue
            j0, m = comps[mi][2], comps[mi][3]
            out.append(f"  {{  // MIX {mi}: inputs {j0}..{j0 + m - 1}, weights me.mw[{q}]")
            out.append(f"    const int term = (lane >= {j0}u && lane < {j0 + m}u) ? __mul24(me.mw[{q}] >> 8, me.p) : 0;")
            out.append("    const int sum = zhdev::wave_sum(term);")
            out.append(f"    me.p = lane == {mi}u ? zhcore::clamp2k(sum >> 8) : me.p;")
            out.append("  }")
    out.append("}")
    return "\n".join(out), comps


def main():
    lines = ["// zh_chain_spec.h — GENERATED by tools/gen_chain_spec.py from zpaqsharp_amd/models.py; do not edit.",
             "// Model-specialised level code for zh_chain.hip (see the generator).", "#pragma once", "#include <stdint.h>", "#include <string.h>", ""]
    specs = []
    lines.append("#if defined(ZH_CHAIN_SPEC_DEVICE)   // device code: only zh_chain.hip asks for it")
    for sid, name in enumerate(("min", "mid", "max"), start=1):
        text, comps = gen(name, sid, models.get(name).header)
        lines += [text, ""]
        specs.append((sid, name, models.get(name).header))
    lines.append("#endif  // ZH_CHAIN_SPEC_DEVICE")
    lines.append("")
    lines.append("// Host side: which specialisation (0 = none) matches a block header's COMP section.")
    lines.append("inline uint32_t zh_spec_lookup(const uint8_t *hdr, size_t len) {")
    for sid, name, header in specs:
        hh, hm, ph, pm, comps, hcomp = zpaql.parse_header(header)
        comp_bytes = header[6:len(header) - len(hcomp)]          # n, COMP..., 0
        arr = ", ".join(str(b) for b in comp_bytes)
        lines.append(f"  {{ static const uint8_t k[] = {{{arr}}};")
        lines.append(f"    if (len >= 6 + sizeof k && memcmp(hdr + 6, k, sizeof k) == 0) return {sid}u; }}")
    lines.append("  return 0;")
    lines.append("}")
    path = os.path.join(ROOT, "zpaqsharp_amd", "csrc", "zh_chain_spec.h")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
