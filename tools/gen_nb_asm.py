#!/usr/bin/env python3
"""Generator of zpaqsharp_amd/csrc/zh_nb_fast.h: the steady-state byte loop of the nibble-at-a-time chain kernel
(zh_nibble.hip, nb_fast) for the built-in min model as hand-laid gfx950 assembly.

    python tools/gen_nb_asm.py            (writes the header; tests/test_abi_and_framing.py checks that it is current)

What the loop is: one iteration = one byte of Decoder.decompress (Decoder.cs:32-56) with the model of Compressor.cs:49-50
(`icm 16; isse 19 0`), the algorithm of nb_decode_byte / nb_boundary in zh_nibble.hip statement for statement — those
functions stay the specification and the general path; GPU tests compare both with the oracle.  The loop runs only when
nothing unusual can happen inside a byte (>= 40 coded bytes in the register-held chunk, output room, a primed coder whose
EOS flag decodes as 0); otherwise it leaves before changing anything and the C++ form takes that byte.

Why by hand: per byte the compiler's rendering of the same source carries ~75 instructions of loop-carried copies and
SGPR spill traffic at the back edge, waits for a store's acknowledgement in front of loads issued before it, and walks
the exec mask for every uniform test whose operands it keeps in vector registers (profiles/r05/nb_stage_notes.txt).
Here every scalar lives in a fixed SGPR, every rare path is out of line, the vector work of a level is ordered around
its two LDS round trips, and stores are issued behind the loads the same stretch of code still has to take.

Register plan (fixed registers are clobbers of the asm statement; operands the compiler allocates are %[name]):
  s64-s101 scalars and masks, v128-v255 constants (loaded from S.fxk), per-lane state (S.fxv) and temporaries."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "zpaqsharp_amd", "csrc", "zh_nb_fast.h")

# ---- per-lane constants the C++ side writes to S.fxk[i][lane] (order = index) ----
KNAMES = ["tab", "slot", "wr2", "wr3", "wr4", "sh2", "sh3", "sh4", "ey1", "ey2", "ey3", "ys1", "ys2", "ys3", "hto", "htm15", "sb2",
          "cshift", "issem", "c8off", "g", "selrow", "seloff", "hspec", "rowst", "slotoff", "koob", "c2047", "c512k", "rnd", "c10000",
          "evo", "mb"]
# ---- per-lane state exchanged through S.fxv[i][lane] ----
VNAMES = ["rx", "rq1", "rq2", "rq3", "rowoff", "hv", "ob0", "ob1", "ob2", "ob3", "oboff", "park", "cur"]


class Regs:
    pass


R = Regs()
_v = 128
for n in KNAMES:
    setattr(R, "k_" + n, f"v{_v}")
    _v += 1
assert _v <= 164
# state (row regs as an aligned tuple)
R.rx, R.rq1, R.rq2, R.rq3 = "v164", "v165", "v166", "v167"
R.row4 = "v[164:167]"
R.rowoff, R.hv, R.oboff, R.o1off = "v168", "v169", "v170", "v171"
R.ob = ["v172", "v173", "v174", "v175"]; R.ob4 = "v[172:175]"
R.o1 = ["v176", "v177", "v178", "v179"]; R.o14 = "v[176:179]"
R.park, R.cur = "v180", "v181"
R.ea = [None, "v182", "v183", "v184", "v185"]
R.etA = [None, "v186", "v188", "v190", "v192"]
R.etB = [None, "v187", "v189", "v191", "v193"]
R.et2 = [None, "v[186:187]", "v[188:189]", "v[190:191]", "v[192:193]"]
R.nsp = [None, "v194", "v195", "v196", "v197"]
R.nA = [None, "v198", "v200", "v202", "v204"]
R.nB = [None, "v199", "v201", "v203", "v205"]
R.n2 = [None, "v[198:199]", "v[200:201]", "v[202:203]", "v[204:205]"]
R.nsb = [None, "v206", "v207", "v208", "v209"]
R.p, R.sq, R.psv, R.pj, R.e = "v210", "v211", "v212", "v213", "v214"
R.t = ["v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223"]
R.eA, R.eB = "v224", "v225"
R.u = ["v226", "v227"]
# candidate probes of the second nibble: [k][probe] -> 4 registers
R.c = [[["v228", "v229", "v230", "v231"], ["v232", "v233", "v234", "v235"], ["v236", "v237", "v238", "v239"]],
       [["v240", "v241", "v242", "v243"], ["v244", "v245", "v246", "v247"], ["v248", "v249", "v250", "v251"]]]
R.c4 = [["v[228:231]", "v[232:235]", "v[236:239]"], ["v[240:243]", "v[244:247]", "v[248:251]"]]
R.ch0 = ["v252", "v254"]; R.cchk = ["v253", "v255"]

# scalars
S = Regs()
S.mb1, S.misse, S.mact, S.mcanon, S.mk, S.sav = "s[64:65]", "s[66:67]", "s[68:69]", "s[70:71]", "s[72:73]", "s[74:75]"
S.nv, S.v1, S.lsel, S.bad, S.fail = "s76", "s77", "s78", "s79", "s80"
S.c24, S.m2048, S.m512k, S.sqb, S.nsbase = "s81", "s82", "s83", "s84", "s85"
S.r, S.off, S.mid, S.m1, S.x, S.ps = "s86", "s87", "s88", "s89", "s90", "s91"
S.t0, S.t1, S.t2, S.t3 = "s92", "s93", "s94", "s95"
S.spin, S.ey4, S.ys4, S.c = "s96", "s97", "s98", "s99"
S.M0, S.M1, S.M2 = "s[86:87]", "s[88:89]", "s[90:91]"      # FIND (the decoder temporaries are free then)
S.win = "s[100:101]"

L = []          # output lines
ICM_ONLY = False  # the MIN1 variant: the model is ONE ICM (level 4's form for barely compressible data); both lanes of a group are
                  # that ICM (the second an exact replica: same constants, same reads, the same values written to the same places), so
                  # no lane takes the ISSE half of the update — the only difference of the loop
PROF = False    # the *_PROF variant: s_memtime stamps, cycles per stage summed in v100.. (lane-uniform), written out at exit


def stamp(i):
    if PROF:
        o(f"""
      s_waitcnt lgkmcnt(0)
      s_memtime s[62:63]
      s_waitcnt lgkmcnt(0)
      s_sub_u32 s61, s62, s60
      s_mov_b32 s60, s62
      v_add_u32_e32 v{100 + i}, s61, v{100 + i}""")


def o(s=""):
    for ln in s.strip("\n").split("\n"):
        ln = ln.strip()
        if ln:
            L.append(ln)


def label(name):
    L.append(f".L{name}_%=:")


# ---- one decoder step (Decoder.cs:136-158) on the scalar unit; `after` = instructions placed behind the v_readlane (its shadow)
def dec_step(tag, shadow=""):
    o(f"v_readlane_b32 {S.ps}, {R.psv}, {S.lsel}")
    o(shadow)
    o(f"""
      s_sub_u32 {S.r}, %[high], %[low]
      s_mul_hi_u32 {S.off}, {S.r}, {S.ps}
      s_add_u32 {S.mid}, %[low], {S.off}
      s_add_u32 {S.m1}, {S.mid}, 1
      s_cmp_le_u32 %[curr], {S.mid}
      s_cselect_b32 %[high], {S.mid}, %[high]
      s_cselect_b32 %[low], %[low], {S.m1}
      s_addc_u32 {S.nv}, {S.nv}, {S.nv}
      s_xor_b32 {S.x}, %[high], %[low]
      s_cmp_lt_u32 {S.x}, {S.c24}
      s_cbranch_scc1 .Lrn{tag}_%=""")
    label(f"bk{tag}")


def renorm_block(tag, last=False):
    """out of line: shift coded bytes in while the top bytes of low and high agree, then the range test the next decode()
    would make (not after the byte's last bit: the next EOS step makes it)"""
    o(".p2align 5")
    label(f"rn{tag}")
    label(f"rl{tag}")
    o(f"""
      s_lshl_b32 %[high], %[high], 8
      s_or_b32 %[high], %[high], 0xff
      s_lshl_b32 %[low], %[low], 8
      s_max_u32 %[low], %[low], 1
      s_lshr_b32 {S.t0}, %[k], 2
      v_readlane_b32 {S.t1}, {R.cur}, {S.t0}
      s_lshl_b32 {S.t0}, %[k], 3
      s_lshr_b32 {S.t1}, {S.t1}, {S.t0}
      s_and_b32 {S.t1}, {S.t1}, 0xff
      s_lshl_b32 %[curr], %[curr], 8
      s_or_b32 %[curr], %[curr], {S.t1}
      s_add_u32 %[k], %[k], 1
      s_xor_b32 {S.x}, %[high], %[low]
      s_cmp_lt_u32 {S.x}, {S.c24}
      s_cbranch_scc1 .Lrl{tag}_%=""")
    if not last:
        o(f"""
      s_cmp_lt_u32 %[curr], %[low]
      s_cselect_b32 {S.bad}, 1, {S.bad}
      s_cmp_gt_u32 %[curr], %[high]
      s_cselect_b32 {S.bad}, 1, {S.bad}""")
    o(f"s_branch .Lbk{tag}_%=")


def nibble(n):
    """four levels of one nibble (nb_decode_byte's inner loops), commit included; n = 0 / 1"""
    T = R.t
    # ---- setup: states of the four nodes of every group's path, their entries and next-state pairs (Predictor.cs:269-272)
    o(f"""
      v_bfe_u32 {R.ea[1]}, {R.rx}, 8, 8
      v_bfe_u32 {R.ea[2]}, {R.rx}, {R.k_sh2}, 8
      v_bfe_u32 {R.ea[3]}, {R.rq1}, {R.k_sh3}, 8
      v_cndmask_b32_e64 {T[0]}, {R.rq2}, {R.rq3}, {S.mb1}
      s_mov_b32 {S.nv}, 0
      v_bfe_u32 {R.ea[4]}, {T[0]}, {R.k_sh4}, 8""")
    for d in range(1, 5):
        o(f"""
      v_lshl_add_u32 {T[d]}, {R.ea[d]}, 2, {S.nsbase}
      v_lshl_add_u32 {R.ea[d]}, {R.ea[d]}, 3, {R.k_tab}""")
    for d in range(1, 5):
        o(f"ds_read_b64 {R.et2[d]}, {R.ea[d]}")
    for d in range(1, 5):
        o(f"ds_read_u16 {R.nsp[d]}, {T[d]}")
    o(f"s_mov_b32 {S.lsel}, 1")
    for d in range(1, 5):
        tag = f"{n}{d}"
        # ---- the entry: from the table, or from an earlier level of this path that trained the same entry (later wins)
        if d == 1:
            o("s_waitcnt lgkmcnt(7)")
            eA, eB = R.etA[1], R.etB[1]
        else:
            o(f"s_waitcnt lgkmcnt({8 - d})" if d < 4 else "s_waitcnt lgkmcnt(4)")
            eA, eB = R.eA, R.eB
            srcA, srcB = R.etA[d], R.etB[d]
            for k in range(1, d):
                o(f"""
      v_cmp_eq_u32_e32 vcc, {R.ea[d]}, {R.ea[k]}
      v_cndmask_b32_e32 {R.eA}, {srcA}, {R.nA[k]}, vcc
      v_cndmask_b32_e32 {R.eB}, {srcB}, {R.nB[k]}, vcc""")
                srcA, srcB = R.eA, R.eB
        # ---- predict (Predictor.cs:267-272, 317-326): ICM lanes p = stretch(cm >> 8) (kept beside the entry), ISSE lanes
        # clamp2k((w0 * p[lane - 1] + w1 * 64) >> 16) in one systolic step (lanes that are no ISSE reproduce their value)
        o(f"""
      v_and_b32_e32 {T[5]}, {eA}, {R.k_issem}
      v_lshlrev_b32_e32 {T[6]}, {R.k_cshift}, {eB}
      v_lshrrev_b32_e32 {T[7]}, 8, {eA}
      v_mov_b32_dpp {T[0]}, {eB} row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
      v_mad_i32_i24 {T[0]}, {T[0]}, {T[5]}, {T[6]}
      v_ashrrev_i32_e32 {T[0]}, 16, {T[0]}
      v_med3_i32 {R.p}, {T[0]}, {S.m2048}, {R.k_c2047}
      v_lshl_add_u32 {T[0]}, {R.p}, 1, {S.sqb}
      ds_read_u16 {R.sq}, {T[0]}""")
        # ---- while squash(p) travels: the ICM half of the update, which needs only the bit (levels 1-3: the group's own)
        if d < 4:
            o(f"""
      v_sub_u32_e32 {T[7]}, {getattr(R, 'k_ey%d' % d)}, {T[7]}
      v_ashrrev_i32_e32 {T[7]}, 2, {T[7]}
      v_add_u32_e32 {T[7]}, {T[7]}, {eA}
      v_lshrrev_b32_e32 {T[5]}, 7, {T[7]}
      v_and_b32_e32 {T[5]}, 0x1fffe, {T[5]}
      ds_read_i16 {T[6]}, {T[5]}
      v_mov_b32_dpp {R.pj}, {R.p} row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
      s_waitcnt lgkmcnt(1)
      v_bfe_u32 {R.nsb[d]}, {R.nsp[d]}, {getattr(R, 'k_ys%d' % d)}, 8""")
        else:
            # level 4: the bit is not the group's own.  The ICM half for BOTH values, so that its stretch look-ups travel under
            # the decode instead of behind it
            o(f"""
      v_sub_u32_e32 {T[5]}, 0, {T[7]}
      v_sub_u32_e32 {T[6]}, 0x7fff, {T[7]}
      v_ashrrev_i32_e32 {T[5]}, 2, {T[5]}
      v_ashrrev_i32_e32 {T[6]}, 2, {T[6]}
      v_add_u32_e32 {T[5]}, {T[5]}, {eA}
      v_add_u32_e32 {T[6]}, {T[6]}, {eA}
      v_lshrrev_b32_e32 {R.u[0]}, 7, {T[5]}
      v_lshrrev_b32_e32 {R.u[1]}, 7, {T[6]}
      v_and_b32_e32 {R.u[0]}, 0x1fffe, {R.u[0]}
      v_and_b32_e32 {R.u[1]}, 0x1fffe, {R.u[1]}
      ds_read_i16 {T[2]}, {R.u[0]}
      ds_read_i16 {T[3]}, {R.u[1]}
      v_mov_b32_dpp {R.pj}, {R.p} row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
      s_waitcnt lgkmcnt(2)""")
        o(f"""
      v_lshl_or_b32 {R.psv}, {R.sq}, 17, {R.k_c10000}
      v_sub_u32_e32 {R.e}, {getattr(R, 'k_ey%d' % d) if d < 4 else R.sq}, {R.sq}""" if d < 4 else f"""
      v_lshl_or_b32 {R.psv}, {R.sq}, 17, {R.k_c10000}
      s_nop 0""")
        if d < 4:
            # ---- decode, and in its shadow the ISSE half of the update (Predictor.cs:440-449) with the group's own bit
            shadow = f"""
      v_mad_i32_i24 {T[0]}, {R.e}, {R.pj}, {R.k_rnd}
      v_add_u32_e32 {T[1]}, 16, {R.e}
      v_ashrrev_i32_e32 {T[0]}, 13, {T[0]}
      v_ashrrev_i32_e32 {T[1]}, 5, {T[1]}"""
            dec_step(tag, shadow)
            o(f"""
      v_add_u32_e32 {T[0]}, {T[0]}, {eA}
      v_add_u32_e32 {T[1]}, {T[1]}, {eB}
      v_med3_i32 {T[0]}, {T[0]}, {S.m512k}, {R.k_c512k}
      v_med3_i32 {T[1]}, {T[1]}, {S.m512k}, {R.k_c512k}
      s_lshl_b32 {S.lsel}, {S.nv}, {4 - d}
      v_cndmask_b32_e64 {R.nA[d]}, {T[7]}, {T[0]}, {S.misse}
      s_or_b32 {S.lsel}, {S.lsel}, 1
      s_waitcnt lgkmcnt(0)
      v_cndmask_b32_e64 {R.nB[d]}, {T[6]}, {T[1]}, {S.misse}""")
            if n == 0 and d == 3:
                # three bits of the byte are known: the helper wave starts on the next byte's 32 candidates.  vmcnt(6): at most the
                # six candidate-row loads of this byte are still out, so every store issued before them (the row written back at
                # the last byte boundary) has been acknowledged — vector memory completes in issue order
                o(f"""
      s_waitcnt vmcnt(6)
      s_lshl_b32 {S.t0}, %[bseq], 8
      s_or_b32 {S.t0}, {S.t0}, {S.nv}
      v_mov_b32_e32 {R.u[0]}, {S.t0}
      s_mov_b64 exec, 1
      ds_write_b32 {R.k_mb}, {R.u[0]}
      s_mov_b64 exec, -1""")
        else:
            # ---- level 4: decode first, then the whole update with the decoded bit
            dec_step(tag, "")
            o(f"""
      s_and_b32 {S.t0}, {S.nv}, 1
      s_cmp_lg_u32 {S.t0}, 0
      s_cselect_b32 {S.ey4}, 0x7fff, 0
      s_cselect_b64 {S.mk}, -1, 0
      s_lshl_b32 {S.ys4}, {S.t0}, 3
      v_cndmask_b32_e64 {T[7]}, {T[5]}, {T[6]}, {S.mk}
      v_sub_u32_e32 {R.e}, {S.ey4}, {R.sq}
      v_bfe_u32 {R.nsb[4]}, {R.nsp[4]}, {S.ys4}, 8
      v_mad_i32_i24 {T[0]}, {R.e}, {R.pj}, {R.k_rnd}
      v_add_u32_e32 {T[1]}, 16, {R.e}
      v_ashrrev_i32_e32 {T[0]}, 13, {T[0]}
      v_ashrrev_i32_e32 {T[1]}, 5, {T[1]}
      v_add_u32_e32 {T[0]}, {T[0]}, {eA}
      v_add_u32_e32 {T[1]}, {T[1]}, {eB}
      v_med3_i32 {T[0]}, {T[0]}, {S.m512k}, {R.k_c512k}
      v_med3_i32 {T[1]}, {T[1]}, {S.m512k}, {R.k_c512k}
      s_lshr_b32 {S.t1}, {S.nv}, 1
      v_cndmask_b32_e64 {R.nA[4]}, {T[7]}, {T[0]}, {S.misse}
      v_cmp_eq_u32_e32 vcc, {S.t1}, {R.k_g}
      s_and_b64 {S.win}, vcc, {S.mact}
      s_waitcnt lgkmcnt(0)
      v_cndmask_b32_e64 {T[6]}, {T[2]}, {T[3]}, {S.mk}
      v_cndmask_b32_e64 {R.nB[4]}, {T[6]}, {T[1]}, {S.misse}""")
    stamp(1 if n == 0 else 4)
    # ---- commit: the group the first three bits name writes its four entries (in path order) and its four bit histories
    o(f"s_and_saveexec_b64 {S.sav}, {S.win}")
    for d in range(1, 5):
        o(f"ds_write_b64 {R.ea[d]}, {R.n2[d]}")
    o(f"""
      ds_write_b8 {R.k_slot}, {R.nsb[1]} offset:1
      ds_write_b8 {R.k_wr2}, {R.nsb[2]}
      ds_write_b8 {R.k_wr3}, {R.nsb[3]}
      ds_write_b8 {R.k_wr4}, {R.nsb[4]}
      s_mov_b64 exec, {S.sav}""")


def find(pr, h0, chk, patches, row, sel, tag):
    """Predictor.find (Predictor.cs:550-567) on three probes pr[0..2] (4 registers each) of the bucket at h0 with check byte
    chk; patches = [(off_reg, row_regs)]: rows this wave wrote back after the probes' loads may have been issued.  Result in
    row[0..3], sel.  Temporaries: R.t, R.u, S.M0-M2."""
    T = R.t
    h1, h2, p0, p1, p2, m, vic, x = T[0], T[1], T[2], T[3], T[4], T[5], T[6], T[7]
    o(f"""
      v_xor_b32_e32 {h1}, 16, {h0}
      v_xor_b32_e32 {h2}, 32, {h0}""")
    # one test for the wave: does a patch row lie in this bucket at all (rare)
    first = True
    for off, _ in patches:
        o(f"""
      v_xor_b32_e32 {x}, {off}, {h0}
      v_and_b32_e32 {x}, 0xffffffcf, {x}
      v_cmp_eq_u32_e32 vcc, 0, {x}""")
        if first:
            o(f"s_mov_b64 {S.M0}, vcc")
            first = False
        else:
            o(f"s_or_b64 {S.M0}, {S.M0}, vcc")
    o(f"""
      s_and_b64 {S.M0}, {S.M0}, exec
      s_cbranch_scc1 .Lpatch{tag}_%=""")
    label(f"patched{tag}")
    o(f"""
      v_and_b32_e32 {x}, 0xff, {pr[0][0]}
      v_cmp_eq_u32_e64 {S.M0}, {x}, {chk}
      v_and_b32_e32 {x}, 0xff, {pr[1][0]}
      v_cmp_eq_u32_e64 {S.M1}, {x}, {chk}
      v_and_b32_e32 {x}, 0xff, {pr[2][0]}
      v_cmp_eq_u32_e64 {S.M2}, {x}, {chk}
      v_bfe_u32 {p0}, {pr[0][0]}, 8, 8
      v_bfe_u32 {p1}, {pr[1][0]}, 8, 8
      v_bfe_u32 {p2}, {pr[2][0]}, 8, 8
      v_cmp_lt_u32_e32 vcc, {p1}, {p2}
      v_min_u32_e32 {m}, {p1}, {p2}
      v_cndmask_b32_e32 {vic}, {h2}, {h1}, vcc
      v_cmp_le_u32_e32 vcc, {p0}, {m}
      v_cndmask_b32_e32 {vic}, {vic}, {h0}, vcc
      v_cndmask_b32_e64 {sel}, {vic}, {h2}, {S.M2}
      v_cndmask_b32_e64 {row[0]}, {chk}, {pr[2][0]}, {S.M2}
      v_cndmask_b32_e64 {row[1]}, 0, {pr[2][1]}, {S.M2}
      v_cndmask_b32_e64 {row[2]}, 0, {pr[2][2]}, {S.M2}
      v_cndmask_b32_e64 {row[3]}, 0, {pr[2][3]}, {S.M2}
      v_cndmask_b32_e64 {sel}, {sel}, {h1}, {S.M1}""")
    for i in range(4):
        o(f"v_cndmask_b32_e64 {row[i]}, {row[i]}, {pr[1][i]}, {S.M1}")
    o(f"v_cndmask_b32_e64 {sel}, {sel}, {h0}, {S.M0}")
    for i in range(4):
        o(f"v_cndmask_b32_e64 {row[i]}, {row[i]}, {pr[0][i]}, {S.M0}")
    # the patch block, out of line
    cold = []
    save = list(L)
    del L[:]
    o(".p2align 5")
    label(f"patch{tag}")
    for off, prow in patches:
        for h, r in ((h0, pr[0]), (h1, pr[1]), (h2, pr[2])):
            o(f"v_cmp_eq_u32_e32 vcc, {off}, {h}")
            for i in range(4):
                o(f"v_cndmask_b32_e32 {r[i]}, {r[i]}, {prow[i]}, vcc")
    o(f"s_branch .Lpatched{tag}_%=")
    cold = list(L)
    del L[:]
    L.extend(save)
    return cold


def gen():
    T = R.t
    cold = []
    # ======== entry: constants and state from LDS, masks, scalar constants
    for i, n in enumerate(KNAMES):
        o(f"ds_read_b32 {getattr(R, 'k_' + n)}, %[kb] offset:{i * 256}")
    vregs = [R.rx, R.rq1, R.rq2, R.rq3, R.rowoff, R.hv, R.ob[0], R.ob[1], R.ob[2], R.ob[3], R.oboff, R.park, R.cur]
    for i, r in enumerate(vregs):
        o(f"ds_read_b32 {r}, %[vb] offset:{i * 256}")
    o(f"""
      s_mov_b32 s64, 0xff00ff00
      s_mov_b32 s65, 0xff00ff00
      s_mov_b32 s66, {'0' if ICM_ONLY else '0xaaaaaaaa'}
      s_mov_b32 s67, {'0' if ICM_ONLY else '0xaaaaaaaa'}
      s_mov_b32 s68, 0xffff
      s_mov_b32 s69, 0
      s_mov_b32 s70, 3
      s_mov_b32 s71, 0
      s_mov_b32 {S.c24}, 0x1000000
      s_mov_b32 {S.m2048}, 0xfffff800
      s_mov_b32 {S.m512k}, 0xfff80000
      s_mov_b32 {S.sqb}, %[sqb]
      s_mov_b32 {S.nsbase}, %[nsb]
      s_mov_b32 {S.bad}, 0
      s_mov_b32 {S.fail}, 0
      s_mov_b32 %[why], 0
      s_mov_b32 %[m0s], m0""")
    if PROF:
        for i in range(12):
            o(f"v_mov_b32_e32 v{100 + i}, 0")
        o("""
      s_memtime s[62:63]
      s_waitcnt lgkmcnt(0)
      s_mov_b32 s60, s62""")
    o(f"""
      s_waitcnt lgkmcnt(0)
      s_branch .Lbyte_%=
      .p2align 6""")
    # ======== one byte
    label("byte")
    o(f"""
      s_cmp_gt_u32 %[k], %[klim]
      s_cbranch_scc1 .Lexit_%=
      s_cmp_eq_u32 %[room], 0
      s_cbranch_scc1 .Lexit_%=
      s_or_b32 {S.t2}, {S.bad}, {S.fail}
      s_cmp_lg_u32 {S.t2}, 0
      s_cbranch_scc1 .Lerr_%=
      s_sub_u32 {S.r}, %[high], %[low]
      s_sub_u32 {S.t0}, %[curr], %[low]
      s_cmp_gt_u32 {S.t0}, {S.r}
      s_cbranch_scc1 .Lexit_%=
      s_cmp_eq_u32 {S.t0}, 0
      s_cbranch_scc1 .Lexit_%=""")
    # EOS flag decoded as 0 (Decoder.cs:41-45 with p = 0: mid = low): low = mid + 1
    o(f"""
      s_add_u32 %[low], %[low], 1
      s_xor_b32 {S.x}, %[high], %[low]
      s_cmp_lt_u32 {S.x}, {S.c24}
      s_cbranch_scc1 .Lrn00_%=""")
    label("bk00")
    # ---- the second nibble's candidate rows: this group's two values of the first nibble (Predictor.find's three probes each)
    for k in range(2):
        cx = T[0]
        o(f"""
      v_add_u32_e32 {cx}, {R.hv}, {R.k_c8off}""" + (f"""
      v_add_u32_e32 {cx}, 16, {cx}""" if k else "") + f"""
      v_lshrrev_b32_e32 {R.cchk[k]}, {R.k_sb2}, {cx}
      v_lshlrev_b32_e32 {R.ch0[k]}, 4, {cx}
      v_and_b32_e32 {R.cchk[k]}, 0xff, {R.cchk[k]}
      v_and_b32_e32 {R.ch0[k]}, {R.ch0[k]}, {R.k_htm15}
      v_add_u32_e32 {T[1]}, {R.ch0[k]}, {R.k_hto}
      v_cndmask_b32_e64 {T[1]}, {R.k_koob}, {T[1]}, {S.mact}
      v_xor_b32_e32 {T[2]}, 16, {T[1]}
      v_xor_b32_e32 {T[3]}, 32, {T[1]}
      buffer_load_dwordx4 {R.c4[k][0]}, {T[1]}, %[rs], 0 offen
      buffer_load_dwordx4 {R.c4[k][1]}, {T[2]}, %[rs], 0 offen
      buffer_load_dwordx4 {R.c4[k][2]}, {T[3]}, %[rs], 0 offen""")
    stamp(0)
    # ======== first nibble
    nibble(0)
    stamp(2)
    # ======== nibble switch (Predictor.cs:267-270: c8 & 0xf0 == 16): the row of the finished nibble, the candidate the
    # decoded value names, find, the row to every group through LDS, then the write-back (behind the loads it might wait for)
    o(f"""
      s_mov_b32 {S.v1}, {S.nv}
      ds_read_b128 {R.o14}, {R.k_slot}
      v_mov_b32_e32 {R.o1off}, {R.rowoff}
      s_and_b32 {S.t0}, {S.nv}, 1
      s_cmp_lg_u32 {S.t0}, 0
      s_cselect_b64 {S.mk}, -1, 0
      s_waitcnt vmcnt(0)""")
    for p in range(3):
        for i in range(4):
            o(f"v_cndmask_b32_e64 {R.c[0][p][i]}, {R.c[0][p][i]}, {R.c[1][p][i]}, {S.mk}")
    o(f"""
      v_cndmask_b32_e64 {R.ch0[0]}, {R.ch0[0]}, {R.ch0[1]}, {S.mk}
      v_cndmask_b32_e64 {R.cchk[0]}, {R.cchk[0]}, {R.cchk[1]}, {S.mk}
      s_waitcnt lgkmcnt(0)""")
    row = [R.nA[1], R.nB[1], R.nA[2], R.nB[2]]          # v198..v201 (a 4-aligned tuple)
    sel = R.nA[3]
    cold += find(R.c[0], R.ch0[0], R.cchk[0], [(R.oboff, R.ob), (R.o1off, R.o1)], row, sel, "s")
    o(f"""
      v_add_u32_e32 {T[0]}, {R.k_evo}, {R.o1off}
      buffer_store_dwordx4 {R.o14}, {T[0]}, %[rs], 0 offen
      s_and_saveexec_b64 {S.sav}, {S.win}
      ds_write_b128 {R.k_slot}, v[198:201]
      ds_write_b32 {R.k_slotoff}, {sel}
      s_mov_b64 exec, {S.sav}
      ds_read_b128 {R.row4}, {R.k_slot}
      ds_read_b32 {R.rowoff}, {R.k_slotoff}
      s_waitcnt lgkmcnt(0)""")
    stamp(3)
    # ======== second nibble
    nibble(1)
    stamp(5)
    # ======== byte boundary: the byte to the helper wave, its staging for this value (h[], the row Predictor.find settles on)
    o(f"""
      s_lshl_b32 {S.c}, {S.v1}, 4
      s_or_b32 {S.c}, {S.c}, {S.nv}
      s_lshl_b32 {S.t0}, %[bseq], 8
      s_or_b32 {S.t0}, {S.t0}, {S.c}
      v_mov_b32_e32 {T[0]}, {S.t0}
      s_mov_b64 exec, 1
      ds_write_b32 {R.k_mb}, {T[0]} offset:4
      s_mov_b64 exec, -1
      ds_read_b128 {R.ob4}, {R.k_slot}
      v_mov_b32_e32 {R.oboff}, {R.rowoff}
      s_and_b32 {S.t1}, {S.c}, 31
      s_lshl_b32 {S.t2}, {S.t1}, 4
      s_lshl_b32 {S.t1}, {S.t1}, 2
      v_add_u32_e32 {T[1]}, {S.t2}, {R.k_selrow}
      v_add_u32_e32 {T[2]}, {S.t1}, {R.k_seloff}
      v_add_u32_e32 {T[3]}, {S.t1}, {R.k_hspec}""")
    label("staged")
    o(f"""
      ds_read_b32 {T[0]}, {R.k_mb} offset:8
      ds_read_b32 {R.hv}, {T[3]}
      ds_read_b128 v[198:201], {T[1]}
      ds_read_b32 {sel}, {T[2]}
      s_waitcnt lgkmcnt(3)
      v_readfirstlane_b32 {S.t0}, {T[0]}
      s_cmp_lg_u32 {S.t0}, %[bseq]
      s_cbranch_scc1 .Lspin_%=
      s_waitcnt lgkmcnt(0)""")
    stamp(6)
    # write-back of the second nibble's row; is one of the two rows written late in the bucket the helper probed?
    o(f"""
      v_add_u32_e32 {T[0]}, {R.k_evo}, {R.oboff}
      buffer_store_dwordx4 {R.ob4}, {T[0]}, %[rs], 0 offen
      v_add_u32_e32 {T[4]}, 16, {R.hv}
      v_lshlrev_b32_e32 {T[5]}, 4, {T[4]}
      v_and_b32_e32 {T[5]}, {T[5]}, {R.k_htm15}
      v_xor_b32_e32 {T[6]}, {R.o1off}, {T[5]}
      v_xor_b32_e32 {T[7]}, {R.oboff}, {T[5]}
      v_and_b32_e32 {T[6]}, 0xffffffcf, {T[6]}
      v_and_b32_e32 {T[7]}, 0xffffffcf, {T[7]}
      v_cmp_eq_u32_e32 vcc, 0, {T[6]}
      v_cmp_eq_u32_e64 {S.M1}, 0, {T[7]}
      s_or_b64 vcc, vcc, {S.M1}
      s_cbranch_vccnz .Lnear_%=""")
    label("taken")
    o(f"""
      s_mov_b64 exec, {S.mcanon}
      ds_write_b128 {R.k_slot}, v[198:201]
      s_mov_b64 exec, -1
      v_mov_b32_e32 {R.rx}, v198
      v_mov_b32_e32 {R.rq1}, v199
      v_mov_b32_e32 {R.rq2}, v200
      v_mov_b32_e32 {R.rq3}, v201
      v_mov_b32_e32 {R.rowoff}, {sel}
      s_add_u32 %[bseq], %[bseq], 1""")
    stamp(7)
    # ======== PostProcessor.write in PASS state (ZPAQL.outc, ZPAQL.cs:201-207): the dword assembled on the scalar unit
    o(f"""
      s_add_u32 {S.t0}, %[vlo], %[nput]
      s_add_u32 %[nput], %[nput], 1
      s_sub_u32 %[room], %[room], 1
      s_and_b32 {S.t1}, {S.t0}, 3
      s_lshl_b32 {S.t2}, {S.t1}, 3
      s_lshl_b32 {S.t3}, {S.c}, {S.t2}
      s_cmp_eq_u32 {S.t1}, 0
      s_cselect_b32 %[word], 0, %[word]
      s_or_b32 %[word], %[word], {S.t3}
      s_cmp_lg_u32 {S.t1}, 3
      s_cbranch_scc1 .Lnext_%=
      s_bfe_u32 {S.t1}, {S.t0}, 0x60002
      s_mov_b32 m0, {S.t1}
      s_and_b32 {S.t2}, {S.t0}, 0xff
      v_writelane_b32 {R.park}, %[word], m0
      s_cmp_lg_u32 {S.t2}, 0xff
      s_cbranch_scc1 .Lnext_%=
      s_mov_b32 %[why], 3
      s_branch .Lexit_%=""")
    label("next")
    stamp(8)
    o("s_branch .Lbyte_%=")
    # ======== out of line
    renorm_block("00")
    for n in range(2):
        for d in range(1, 5):
            renorm_block(f"{n}{d}", last=(n == 1 and d == 4))
    L.extend(cold)
    # the helper wave is not ready: poll its word (bounded), then read the staging again
    o(".p2align 5")
    label("spin")
    o(f"s_mov_b32 {S.spin}, 0x4000000")
    label("spin1")
    o(f"""
      ds_read_b32 {T[0]}, {R.k_mb} offset:8
      s_waitcnt lgkmcnt(0)
      v_readfirstlane_b32 {S.t0}, {T[0]}
      s_cmp_eq_u32 {S.t0}, %[bseq]
      s_cbranch_scc1 .Lstaged_%=
      s_sub_u32 {S.spin}, {S.spin}, 1
      s_cmp_lg_u32 {S.spin}, 0
      s_cbranch_scc1 .Lspin1_%=
      s_mov_b32 {S.fail}, 1
      s_waitcnt lgkmcnt(0)
      s_branch .Ltaken_%=""")
    # a row written late lies in the probed bucket: the helper's three probes, patched, and find
    o(".p2align 5")
    label("near")
    # A context that does not change from byte to byte (the order-0 ICM of the method models, h = 0) reads the very row this byte's
    # first nibble used.  Predictor.find returns probe 0's place when the row there carries the context's check byte — and what
    # lies there now is the row this wave wrote back (old1 at the nibble switch, oldb just now: the later store wins).  Lanes for
    # which that holds take the row from the registers it was written from; only a lane whose late row lies at another place of
    # the bucket, or under another check byte, needs the three probes patched and searched.
    o(f"""
      s_mov_b64 {S.M2}, vcc
      v_lshrrev_b32_e32 {T[6]}, {R.k_sb2}, {T[4]}
      v_and_b32_e32 {T[0]}, 0xff, {R.o1[0]}
      v_and_b32_e32 {T[6]}, 0xff, {T[6]}
      v_and_b32_e32 {T[2]}, 0xff, {R.ob[0]}
      v_cmp_eq_u32_e32 vcc, {T[6]}, {T[0]}
      v_cmp_eq_u32_e64 {S.M1}, {R.o1off}, {T[5]}
      s_and_b64 {S.M1}, {S.M1}, vcc
      v_cmp_eq_u32_e32 vcc, {T[6]}, {T[2]}
      v_cmp_eq_u32_e64 {S.M0}, {R.oboff}, {T[5]}
      s_and_b64 {S.M0}, {S.M0}, vcc
      s_or_b64 vcc, {S.M0}, {S.M1}
      s_andn2_b64 vcc, {S.M2}, vcc
      s_cbranch_vccnz .Lnearfull_%=
      v_cndmask_b32_e64 {row[0]}, {row[0]}, {R.o1[0]}, {S.M1}
      v_cndmask_b32_e64 {row[1]}, {row[1]}, {R.o1[1]}, {S.M1}
      v_cndmask_b32_e64 {row[2]}, {row[2]}, {R.o1[2]}, {S.M1}
      v_cndmask_b32_e64 {row[3]}, {row[3]}, {R.o1[3]}, {S.M1}
      v_cndmask_b32_e64 {sel}, {sel}, {T[5]}, {S.M1}
      v_cndmask_b32_e64 {row[0]}, {row[0]}, {R.ob[0]}, {S.M0}
      v_cndmask_b32_e64 {row[1]}, {row[1]}, {R.ob[1]}, {S.M0}
      v_cndmask_b32_e64 {row[2]}, {row[2]}, {R.ob[2]}, {S.M0}
      v_cndmask_b32_e64 {row[3]}, {row[3]}, {R.ob[3]}, {S.M0}
      v_cndmask_b32_e64 {sel}, {sel}, {T[5]}, {S.M0}
      s_branch .Ltaken_%=""")
    label("nearfull")
    o(f"""
      v_lshrrev_b32_e32 {T[6]}, {R.k_sb2}, {T[4]}
      v_and_b32_e32 {T[6]}, 0xff, {T[6]}
      v_mov_b32_e32 {R.u[0]}, {T[5]}
      v_mov_b32_e32 {R.u[1]}, {T[6]}
      v_add_u32_e32 {T[1]}, {S.t2}, {R.k_rowst}
      ds_read_b128 {R.c4[0][0]}, {T[1]}
      ds_read_b128 {R.c4[0][1]}, {T[1]} offset:512
      ds_read_b128 {R.c4[0][2]}, {T[1]} offset:1024
      s_waitcnt lgkmcnt(0)""")
    cold2 = find(R.c[0], R.u[0], R.u[1], [(R.o1off, R.o1), (R.oboff, R.ob)], row, sel, "b")
    o("s_branch .Ltaken_%=")
    L.extend(cold2)
    label("err")
    o(f"""
      s_mov_b32 %[why], 2
      s_branch .Lexit_%=""")
    label("exit")
    vregs = [R.rx, R.rq1, R.rq2, R.rq3, R.rowoff, R.hv, R.ob[0], R.ob[1], R.ob[2], R.ob[3], R.oboff, R.park]
    for i, r in enumerate(vregs):
        o(f"ds_write_b32 %[vb], {r} offset:{i * 256}")
    if PROF:
        for i in range(12):
            o(f"ds_write_b32 %[vb], v{100 + i} offset:{(len(VNAMES) + i) * 256}")
    o(f"""
      s_mov_b32 m0, %[m0s]
      s_mov_b32 %[obad], {S.bad}
      s_mov_b32 %[ofail], {S.fail}
      s_waitcnt lgkmcnt(0)""")


def emit(name, prof, icm_only=False):
    global PROF, ICM_ONLY
    PROF = prof
    ICM_ONLY = icm_only
    del L[:]
    gen()
    clob = ["memory", "scc", "vcc"] + [f"s{i}" for i in range(60 if prof else 64, 102)] + [f"v{i}" for i in range(100 if prof else 128, 256)]
    text = f"""#define {name}(low_, high_, curr_, k_, bseq_, nput_, room_, word_, why_, obad_, ofail_, m0s_, klim_, vlo_, kb_, vb_, rs_, sqb_, nsb_) \\
  asm volatile( \\
"""
    for ln in L:
        text += '  "' + ln.replace('"', '\\"') + '\\n\\t" \\\n'
    text += """  : [low] "+s"(low_), [high] "+s"(high_), [curr] "+s"(curr_), [k] "+s"(k_), [bseq] "+s"(bseq_), [nput] "+s"(nput_), [room] "+s"(room_), \\
    [word] "+s"(word_), [why] "=&s"(why_), [obad] "=&s"(obad_), [ofail] "=&s"(ofail_), [m0s] "=&s"(m0s_) \\
  : [klim] "s"(klim_), [vlo] "s"(vlo_), [kb] "v"(kb_), [vb] "v"(vb_), [rs] "s"(rs_), [sqb] "s"(sqb_), [nsb] "s"(nsb_) \\
  : """ + ", ".join('"' + c + '"' for c in clob) + ")\n"
    return text, len(L)


def main():
    t0, n0 = emit("ZH_NB_FAST_MIN_LOOP", False)
    t1, _ = emit("ZH_NB_FAST_MIN_LOOP_PROF", True)
    t2, _ = emit("ZH_NB_FAST_MIN1_LOOP", False, True)
    t3, _ = emit("ZH_NB_FAST_MIN1_LOOP_PROF", True, True)
    head = f"""// zh_nb_fast.h — GENERATED by tools/gen_nb_asm.py (do not edit: edit the generator and run it).
// The steady-state byte loop of nb_fast (zh_nibble.hip) for the built-in min model, hand-laid gfx950 assembly.
// ZH_NB_FAST_MIN_LOOP_PROF is the same loop with s_memtime stamps (cycles per stage, kNbS_count + i of the state area);
// ZH_NB_FAST_MIN1_LOOP[_PROF] the loop for a model of one ICM (both lanes of a group are that ICM).
#pragma once
#define ZH_NB_FAST_MIN 1
enum : int {{ {", ".join("kNbK_" + n + (" = 0" if i == 0 else "") for i, n in enumerate(KNAMES))}, kNbK_count }};
enum : int {{ {", ".join("kNbS_" + n + (" = 0" if i == 0 else "") for i, n in enumerate(VNAMES))}, kNbS_count }};
// clang-format off
"""
    text = head + t0 + t1 + t2 + t3 + "// clang-format on\n"
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"{OUT}: {n0} lines of assembly")


def _old_main():
    clob = []
    head = f"""// zh_nb_fast.h — GENERATED by tools/gen_nb_asm.py (do not edit: edit the generator and run it).
// The steady-state byte loop of nb_fast (zh_nibble.hip) for the built-in min model, hand-laid gfx950 assembly.
#pragma once
#define ZH_NB_FAST_MIN 1
enum : int {{ {", ".join("kNbK_" + n + (" = 0" if i == 0 else "") for i, n in enumerate(KNAMES))}, kNbK_count }};
enum : int {{ {", ".join("kNbS_" + n + (" = 0" if i == 0 else "") for i, n in enumerate(VNAMES))}, kNbS_count }};
// clang-format off
#define ZH_NB_FAST_MIN_LOOP(low_, high_, curr_, k_, bseq_, nput_, room_, word_, why_, obad_, ofail_, m0s_, klim_, vlo_, kb_, vb_, rs_, sqb_, nsb_) \\
  asm volatile( \\
"""
    body = ""
    for ln in L:
        body += '  "' + ln.replace('"', '\\"') + '\\n\\t" \\\n'
    tail = """  : [low] "+s"(low_), [high] "+s"(high_), [curr] "+s"(curr_), [k] "+s"(k_), [bseq] "+s"(bseq_), [nput] "+s"(nput_), [room] "+s"(room_), \\
    [word] "+s"(word_), [why] "=&s"(why_), [obad] "=&s"(obad_), [ofail] "=&s"(ofail_), [m0s] "=&s"(m0s_) \\
  : [klim] "s"(klim_), [vlo] "s"(vlo_), [kb] "v"(kb_), [vb] "v"(vb_), [rs] "s"(rs_), [sqb] "s"(sqb_), [nsb] "s"(nsb_) \\
  : """ + ", ".join('"' + c + '"' for c in clob) + ")\n// clang-format on\n"
    text = head + body + tail
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"{OUT}: {len(L)} lines of assembly")


if __name__ == "__main__":
    main()
