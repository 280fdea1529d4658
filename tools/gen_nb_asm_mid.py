#!/usr/bin/env python3
"""Generator of zpaqsharp_amd/csrc/zh_nb_fast_mid.h: the steady-state byte loop of nb_fast (zh_nibble.hip) for the built-in
mid model (Compressor.cs:53-57: icm 5; isse 13 0; isse 17 1; isse 18 2; isse 18 3; isse 19 4; match 22 24; mix 16 0 7 24 255)
as hand-laid gfx950 assembly.  Same plan as tools/gen_nb_asm.py (whose helpers it uses): eight groups of eight lanes, one per
3-bit path prefix of the nibble, lane (g, c) = component c on group g's path; nb_decode_byte / nb_boundary of zh_nibble.hip
are the specification, statement for statement.

    python tools/gen_nb_asm_mid.py [--check]

What the mid model adds to the min loop:
  * the ISSE chain is five systolic steps (v_mov_dpp row_shr:1, v_mad_i32_i24, v_ashr, v_med3); a DPP read needs two wait
    states behind the write of its source — every gap carries two instructions of the ICM half of the update (which needs only
    the entry and the group's own bit) instead of an s_nop;
  * MATCH: the nibble the match predicts and which groups' paths have left it are per-lane arithmetic on the group id —
    no scalar hand-over;
  * the mixer: weights of the four rows a group's path walks are requested when the nibble starts, the dot product is a
    butterfly over the group's eight lanes (every lane ends with the sum: no broadcast of the mixer lane's output), its
    training rides behind the decoder step; the trained weights are stored by the winning group BEHIND the next nibble's
    weight loads (vector memory completes in issue order: a load issued behind a store waits for the store's acknowledgement)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_nb_asm as G
from gen_nb_asm import R, S, L, o, label, dec_step, renorm_block, find

ROOT = G.ROOT
OUT = os.path.join(ROOT, "zpaqsharp_amd", "csrc", "zh_nb_fast_mid.h")

KX = ["rslot", "rate", "vomix", "cm0", "row1_1", "row1_2", "row1_3", "row1_4", "pre2", "pre3", "pre4", "lane1", "htmask", "cmo", "cmmask",
      "mixst", "mixb", "evm"]
KNAMES = G.KNAMES + KX
VX = ["m_len", "m_ptr", "m_limit", "m_byte", "pm0", "pm1", "cm_pre", "va_pre", "vb_pre", "mbn_pre", "mbc_pre", "mx_rb", "w1_new",
      "mwl1", "mwl2", "mwl3", "mwl4", "mra1", "mra2", "mra3", "mra4"]
VNAMES = G.VNAMES + VX

_v = 64
for n in KX:
    setattr(R, "k_" + n, f"v{_v}")
    _v += 1
assert _v <= 82
for i, n in enumerate(VX[:13]):
    setattr(R, n, f"v{82 + i}")
R.cw0, R.cw1, R.x, R.pmv, R.ex, R.mism = "v95", "v96", "v97", "v98", "v99", "v48"
R.mwl = [None, "v100", "v101", "v102", "v103"]
R.mrA = [None, "v108", "v109", "v110", "v111"]
R.mrB = [None, "v112", "v113", "v114", "v115"]
R.nmw = [None, "v116", "v117", "v118", "v119"]
R.cmw = ["v120", "v121"]
R.sqm, R.pmx, R.em, R.term, R.mw8 = "v122", "v123", "v124", "v125", "v126"
R.pl1, R.sql1, R.mwl1s = "v60", "v61", "v62"
R.sa1, R.sa2, R.sgmw, R.rbold = "v56", "v57", "v58", "v59"
R.sv2 = "v49"

# more scalars (s40-s59)
S.mii, S.mmatch, S.mfeed = "s[40:41]", "s[42:43]", "s[44:45]"
S.mmsk, S.mbase, S.mxbase, S.mxsize1 = "s48", "s49", "s50", "s51"
S.u = [None, "s52", "s53", "s54", "s55"]
S.w0, S.w1, S.w2 = "s56", "s57", "s58"
S.y1 = "s59"
S.mw = "s[46:47]"       # a mask temporary

MIX_M = 7          # mixer inputs: 7 (mid: lane 7 is the MIX, whose context is h[7]) or 8 (the level-4 text model: lane 7 is an ICM, the
                   # MIX has no lane and its context h[8] is 0)
PROF = False
PROF_FINE = int(os.environ.get('NB_PROF_FINE', '0')) == 1      # diagnostic: three more stamps inside every level (9: chain done, 10: squash back, 11: decoded)
PROF_BND = int(os.environ.get('NB_PROF_FINE', '0')) == 2       # ... or inside the byte boundary (9: next byte's mixer rows requested, 10: this byte's stores issued, 11: MATCH settled)


def stamp(i):
    if PROF:
        o(f"""
      s_waitcnt lgkmcnt(0)
      s_memtime s[62:63]
      s_waitcnt lgkmcnt(0)
      s_sub_u32 s61, s62, s60
      s_mov_b32 s60, s62
      v_add_u32_e32 v{32 + i}, s61, v{32 + i}""")


DPP1 = "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"


def pmv_part1(d):
    """the match's prediction at level d, before the `left its path` mask: pmv = bit (4-d) of ex ? pm1 : pm0"""
    return [f"v_bfe_u32 {R.t[4]}, {R.ex}, {4 - d}, 1",
            f"v_cmp_ne_u32_e32 vcc, 0, {R.t[4]}",
            f"v_cndmask_b32_e32 {R.pmv}, {R.pm0}, {R.pm1}, vcc"]


def pmv_part2(d):
    m = {2: 4, 3: 6, 4: 7}[d]
    return [f"v_and_b32_e32 {R.t[4]}, {m}, {R.mism}",
            f"v_cmp_eq_u32_e32 vcc, 0, {R.t[4]}",
            f"v_cndmask_b32_e32 {R.pmv}, 0, {R.pmv}, vcc"]


def nibble(n, mr):
    """four levels of one nibble; mr = the register set holding this nibble's mixer row offsets"""
    T = R.t
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
    # MATCH (Predictor.cs:273-287): the nibble the match predicts; which groups' paths leave it, and where
    o(f"""
      v_bfe_u32 {R.ex}, {R.m_byte}, {0 if n else 4}, 4
      s_mov_b32 {S.lsel}, 7
      v_lshrrev_b32_e32 {T[0]}, 1, {R.ex}
      v_xor_b32_e32 {T[0]}, {T[0]}, {R.k_g}
      v_and_b32_e32 {R.mism}, 7, {T[0]}""")
    o("\n".join(pmv_part1(1)))
    for d in range(1, 5):
        tag = f"{n}{d}"
        ey = getattr(R, "k_ey%d" % d) if d < 4 else None
        if d == 1:
            o("s_waitcnt lgkmcnt(7)")
            eA, eB = R.etA[1], R.etB[1]
        else:
            eA, eB = R.eA, R.eB
            srcA, srcB = R.etA[d], R.etB[d]
            for k in range(1, d):
                o(f"""
      v_cmp_eq_u32_e32 vcc, {R.ea[d]}, {R.ea[k]}
      v_cndmask_b32_e32 {R.eA}, {srcA}, {R.nA[k]}, vcc
      v_cndmask_b32_e32 {R.eB}, {srcB}, {R.nB[k]}, vcc""")
                srcA, srcB = R.eA, R.eB
        # ---- predict: x = ICM / ISSE lanes the entry's second word, the MATCH lane its prediction, the mixer lane 0
        o(f"""
      v_cndmask_b32_e64 {R.x}, {R.pmv}, {eB}, {S.mii}
      v_and_b32_e32 {R.cw0}, {eA}, {R.k_issem}
      v_lshrrev_b32_e32 {T[7]}, 8, {eA}
      v_lshlrev_b32_e32 {R.cw1}, {R.k_cshift}, {R.x}""")
        # five systolic ISSE steps; the gaps between them (two wait states in front of each DPP read) carry the ICM half of
        # the update: ncm = cm + ((y * 32767 - (cm >> 8)) >> 2) and its stretch look-up (levels 1-3: the group's own bit;
        # level 4: both values of the bit)
        # (the weights of levels 2-4 were requested at the last nibble boundary, IN FRONT of that boundary's stores: vmcnt(N) with N =
        # the vector memory operations issued behind the load, so that no store's acknowledgement is waited for)
        vm = [] if d == 1 else [f"s_waitcnt vmcnt({(18 if n == 0 else 11) - (d - 2)})"]
        if d < 4:
            gaps = [[f"v_sub_u32_e32 {T[7]}, {ey}, {T[7]}", f"v_ashrrev_i32_e32 {T[7]}, 2, {T[7]}"],
                    [f"v_add_u32_e32 {T[7]}, {T[7]}, {eA}"] + vm + [f"v_ashrrev_i32_e32 {R.mw8}, 8, {R.mwl[d]}"],
                    [f"v_lshrrev_b32_e32 {T[5]}, 7, {T[7]}", pmv_part1(d + 1)[0]],
                    [f"v_and_b32_e32 {T[5]}, 0x1fffe, {T[5]}", f"ds_read_i16 {T[6]}, {T[5]}"]]
        else:
            gaps = [[f"v_sub_u32_e32 {T[5]}, 0, {T[7]}", f"v_sub_u32_e32 {T[6]}, 0x7fff, {T[7]}"],
                    [f"v_ashrrev_i32_e32 {T[5]}, 2, {T[5]}", f"v_ashrrev_i32_e32 {T[6]}, 2, {T[6]}"],
                    [f"v_add_u32_e32 {T[5]}, {T[5]}, {eA}", f"v_add_u32_e32 {T[6]}, {T[6]}, {eA}"],
                    [f"v_lshrrev_b32_e32 {R.u[0]}, 7, {T[5]}"] + vm + [f"v_ashrrev_i32_e32 {R.mw8}, 8, {R.mwl[d]}"]]
        src = R.x
        for t in range(5):
            o(f"""
      v_mov_b32_dpp {T[0]}, {src} {DPP1}
      v_mad_i32_i24 {T[0]}, {T[0]}, {R.cw0}, {R.cw1}
      v_ashrrev_i32_e32 {T[0]}, 16, {T[0]}
      v_med3_i32 {R.p}, {T[0]}, {S.m2048}, {R.k_c2047}""")
            src = R.p
            if t < 4:
                o("\n".join(gaps[t]))
        if PROF_FINE:
            stamp(9)
        # ---- the mixer (Predictor.cs:302-316): dot product over the group's eight lanes, every lane ends with the sum
        fill2 = pmv_part1(d + 1)[1:2] if d < 4 else [f"v_lshrrev_b32_e32 {R.u[1]}, 7, {T[6]}"]
        fill3 = (pmv_part1(d + 1)[2:] + ["s_nop 0"]) if d < 4 else [f"v_and_b32_e32 {R.u[0]}, 0x1fffe, {R.u[0]}", f"v_and_b32_e32 {R.u[1]}, 0x1fffe, {R.u[1]}"]
        o(f"""
      v_mul_i32_i24_e32 {R.term}, {R.mw8}, {R.p}
      v_lshl_add_u32 {T[0]}, {R.p}, 1, {S.sqb}
      ds_read_u16 {R.sq}, {T[0]}
      v_add_u32_dpp {R.term}, {R.term}, {R.term} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf
      v_mov_b32_dpp {R.pj}, {R.p} {DPP1}""")
        o("\n".join(fill2))
        o(f"v_add_u32_dpp {R.term}, {R.term}, {R.term} quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        o("\n".join(fill3))
        o(f"""
      v_add_u32_dpp {R.term}, {R.term}, {R.term} row_half_mirror row_mask:0xf bank_mask:0xf
      v_ashrrev_i32_e32 {R.term}, 8, {R.term}
      v_med3_i32 {R.pmx}, {R.term}, {S.m2048}, {R.k_c2047}
      v_lshl_add_u32 {T[0]}, {R.pmx}, 1, {S.sqb}
      ds_read_u16 {R.sqm}, {T[0]}""")
        if d < 4:
            o("\n".join(pmv_part2(d + 1)))
            o("s_waitcnt lgkmcnt(0)")
        else:
            o(f"""
      ds_read_i16 {T[2]}, {R.u[0]}
      ds_read_i16 {T[3]}, {R.u[1]}
      s_waitcnt lgkmcnt(2)""")
        if PROF_FINE:
            stamp(10)
        o(f"""
      v_lshl_or_b32 {R.psv}, {R.sqm}, 17, {R.k_c10000}""")
        if d < 4:
            o(f"v_sub_u32_e32 {R.e}, {ey}, {R.sq}")
            o(f"v_bfe_u32 {R.nsb[d]}, {R.nsp[d]}, {getattr(R, 'k_ys%d' % d)}, 8")
            shadow = f"""
      v_sub_u32_e32 {R.em}, {ey}, {R.sqm}
      v_mad_i32_i24 {T[0]}, {R.e}, {R.pj}, {R.k_rnd}
      v_add_u32_e32 {T[1]}, 16, {R.e}
      v_mul_i32_i24_e32 {R.em}, {R.em}, {R.k_rate}"""
            dec_step(tag, shadow)
            if PROF_FINE:
                stamp(11)
            o(f"""
      v_ashrrev_i32_e32 {T[0]}, 13, {T[0]}
      v_ashrrev_i32_e32 {T[1]}, 5, {T[1]}
      v_ashrrev_i32_e32 {R.em}, 4, {R.em}
      v_add_u32_e32 {T[0]}, {T[0]}, {eA}
      v_add_u32_e32 {T[1]}, {T[1]}, {eB}
      v_mad_i32_i24 {R.em}, {R.em}, {R.p}, {R.k_rnd}
      v_med3_i32 {T[0]}, {T[0]}, {S.m512k}, {R.k_c512k}
      v_med3_i32 {T[1]}, {T[1]}, {S.m512k}, {R.k_c512k}
      v_ashrrev_i32_e32 {R.em}, 13, {R.em}
      s_lshl_b32 {S.lsel}, {S.nv}, {6 - d}
      v_cndmask_b32_e64 {R.nA[d]}, {T[7]}, {T[0]}, {S.misse}
      v_add_u32_e32 {R.em}, {R.em}, {R.mwl[d]}
      s_or_b32 {S.lsel}, {S.lsel}, 7
      v_cndmask_b32_e64 {R.nB[d]}, {T[6]}, {T[1]}, {S.misse}
      v_med3_i32 {R.nmw[d]}, {R.em}, {S.m512k}, {R.k_c512k}""")
            if n == 0 and d == 1:
                o(f"""
      v_mov_b32_e32 {R.pl1}, {R.p}
      v_mov_b32_e32 {R.sql1}, {R.sqm}
      v_mov_b32_e32 {R.mwl1s}, {R.mwl[1]}""")
            if n == 0 and d == 3:
                # three bits known: the helper starts.  vmcnt(9): the byte's 8 candidate loads and the match index load of the last
                # boundary may still be out; every store issued before them has been acknowledged
                o(f"""
      s_waitcnt vmcnt(9)
      s_lshl_b32 {S.t0}, %[bseq], 8
      s_or_b32 {S.t0}, {S.t0}, {S.nv}
      v_mov_b32_e32 {R.u[0]}, {S.t0}
      s_mov_b64 exec, 1
      ds_write_b32 {R.k_mb}, {R.u[0]}
      s_mov_b64 exec, -1""")
        else:
            o("s_nop 0")
            dec_step(tag, "")
            o(f"""
      s_and_b32 {S.t0}, {S.nv}, 1
      s_cmp_lg_u32 {S.t0}, 0
      s_cselect_b32 {S.ey4}, 0x7fff, 0
      s_cselect_b64 {S.mk}, -1, 0
      s_lshl_b32 {S.ys4}, {S.t0}, 3
      v_cndmask_b32_e64 {T[7]}, {T[5]}, {T[6]}, {S.mk}
      v_sub_u32_e32 {R.e}, {S.ey4}, {R.sq}
      v_sub_u32_e32 {R.em}, {S.ey4}, {R.sqm}
      v_bfe_u32 {R.nsb[4]}, {R.nsp[4]}, {S.ys4}, 8
      v_mad_i32_i24 {T[0]}, {R.e}, {R.pj}, {R.k_rnd}
      v_add_u32_e32 {T[1]}, 16, {R.e}
      v_mul_i32_i24_e32 {R.em}, {R.em}, {R.k_rate}
      v_ashrrev_i32_e32 {T[0]}, 13, {T[0]}
      v_ashrrev_i32_e32 {T[1]}, 5, {T[1]}
      v_ashrrev_i32_e32 {R.em}, 4, {R.em}
      v_add_u32_e32 {T[0]}, {T[0]}, {eA}
      v_add_u32_e32 {T[1]}, {T[1]}, {eB}
      v_mad_i32_i24 {R.em}, {R.em}, {R.p}, {R.k_rnd}
      v_med3_i32 {T[0]}, {T[0]}, {S.m512k}, {R.k_c512k}
      v_med3_i32 {T[1]}, {T[1]}, {S.m512k}, {R.k_c512k}
      v_ashrrev_i32_e32 {R.em}, 13, {R.em}
      s_lshr_b32 {S.t1}, {S.nv}, 1
      v_cndmask_b32_e64 {R.nA[4]}, {T[7]}, {T[0]}, {S.misse}
      v_add_u32_e32 {R.em}, {R.em}, {R.mwl[4]}
      v_cmp_eq_u32_e64 {S.win}, {S.t1}, {R.k_g}
      s_waitcnt lgkmcnt(0)
      v_cndmask_b32_e64 {T[6]}, {T[2]}, {T[3]}, {S.mk}
      v_med3_i32 {R.nmw[4]}, {R.em}, {S.m512k}, {R.k_c512k}
      v_cndmask_b32_e64 {R.nB[4]}, {T[6]}, {T[1]}, {S.misse}""")
    stamp(1 if n == 0 else 4)
    o(f"s_and_saveexec_b64 {S.sav}, {S.win}")
    for d in range(1, 5):
        o(f"ds_write_b64 {R.ea[d]}, {R.n2[d]}")
    o(f"""
      ds_write_b8 {R.k_slot}, {R.nsb[1]} offset:1
      ds_write_b8 {R.k_wr2}, {R.nsb[2]}
      ds_write_b8 {R.k_wr3}, {R.nsb[3]}
      ds_write_b8 {R.k_wr4}, {R.nsb[4]}
      s_mov_b64 exec, {S.sav}""")
    # MATCH (Predictor.cs:383-384): a miss ends the match
    o(f"""
      v_cmp_ne_u32_e32 vcc, {S.nv}, {R.ex}
      v_cndmask_b32_e64 {R.m_len}, {R.m_len}, 0, vcc
      v_cndmask_b32_e64 {R.pm0}, {R.pm0}, 0, vcc
      v_cndmask_b32_e64 {R.pm1}, {R.pm1}, 0, vcc""")


def mixer_stores(mr):
    o(f"s_and_saveexec_b64 {S.sav}, {S.win}")
    for d in range(1, 5):
        o(f"buffer_store_dword {R.nmw[d]}, {mr[d]}, %[rs], 0 offen")
    o(f"s_mov_b64 exec, {S.sav}")


def gen():
    T = R.t
    cold = []
    # ======== entry
    for i, n in enumerate(KNAMES):
        o(f"ds_read_b32 {getattr(R, 'k_' + n)}, %[kb] offset:{i * 256}")
    vregs = [R.rx, R.rq1, R.rq2, R.rq3, R.rowoff, R.hv, R.ob[0], R.ob[1], R.ob[2], R.ob[3], R.oboff, R.park, R.cur] + \
            [getattr(R, n) for n in VX[:13]] + [R.mwl[1], R.mwl[2], R.mwl[3], R.mwl[4], R.mrA[1], R.mrA[2], R.mrA[3], R.mrA[4]]
    for i, r in enumerate(vregs):
        o(f"ds_read_b32 {r}, %[vb] offset:{i * 256}")
    o(f"""
      s_mov_b32 s64, 0
      s_mov_b32 s65, 0xffffffff
      s_mov_b32 s66, 0x3e3e3e3e
      s_mov_b32 s67, 0x3e3e3e3e
      s_mov_b32 s68, 0xffffffff
      s_mov_b32 s69, 0xffffffff
      s_mov_b32 s70, 0xff
      s_mov_b32 s71, 0
      s_mov_b32 s40, {'0x3f3f3f3f' if MIX_M == 7 else '0xbfbfbfbf'}
      s_mov_b32 s41, {'0x3f3f3f3f' if MIX_M == 7 else '0xbfbfbfbf'}
      s_mov_b32 s42, 0x40404040
      s_mov_b32 s43, 0x40404040
      s_mov_b32 s44, {'0x7f7f7f7f' if MIX_M == 7 else '0xffffffff'}
      s_mov_b32 s45, {'0x7f7f7f7f' if MIX_M == 7 else '0xffffffff'}
      s_mov_b32 {S.c24}, 0x1000000
      s_mov_b32 {S.m2048}, 0xfffff800
      s_mov_b32 {S.m512k}, 0xfff80000
      s_mov_b32 {S.sqb}, %[sqb]
      s_mov_b32 {S.nsbase}, %[nsb]
      s_mov_b32 {S.mxbase}, %[mxb]
      s_mov_b32 {S.mxsize1}, %[mxs]
      s_mov_b32 {S.bad}, 0
      s_mov_b32 {S.fail}, 0
      s_mov_b32 %[why], 0
      s_mov_b32 %[m0s], m0""")
    if PROF:
        for i in range(12):
            o(f"v_mov_b32_e32 v{32 + i}, 0")
        o("""
      s_memtime s[62:63]
      s_waitcnt lgkmcnt(0)
      s_mov_b32 s60, s62""")
    o(f"""
      s_waitcnt vmcnt(0) lgkmcnt(0)
      v_readlane_b32 {S.mmsk}, {R.k_htmask}, 6
      v_readlane_b32 {S.mbase}, {R.k_hto}, 6
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
      s_cbranch_scc1 .Lexit_%=
      s_add_u32 %[low], %[low], 1
      s_xor_b32 {S.x}, %[high], %[low]
      s_cmp_lt_u32 {S.x}, {S.c24}
      s_cbranch_scc1 .Lrn00_%=""")
    label("bk00")
    # ---- the second nibble's candidate rows and the mixer weights of the rows they start with
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
      v_cndmask_b32_e64 {T[1]}, {R.k_koob}, {T[1]}, {S.mii}
      v_xor_b32_e32 {T[2]}, 16, {T[1]}
      v_xor_b32_e32 {T[3]}, 32, {T[1]}
      buffer_load_dwordx4 {R.c4[k][0]}, {T[1]}, %[rs], 0 offen
      buffer_load_dwordx4 {R.c4[k][1]}, {T[2]}, %[rs], 0 offen
      buffer_load_dwordx4 {R.c4[k][2]}, {T[3]}, %[rs], 0 offen""")
    o(f"""
      v_add_u32_e32 {T[4]}, {R.mx_rb}, {R.k_cm0}
      buffer_load_dword {R.cmw[0]}, {T[4]}, %[rs], 0 offen
      buffer_load_dword {R.cmw[1]}, {T[4]}, %[rs], 0 offen offset:{4 * MIX_M}""")
    stamp(0)
    nibble(0, R.mrA)
    stamp(2)
    # ======== nibble switch
    o(f"""
      s_mov_b32 {S.v1}, {S.nv}
      s_bfe_u32 {S.t0}, {S.nv}, 0x10003
      s_cmp_lg_u32 {S.t0}, 0
      s_cselect_b32 {S.y1}, 0x7fff, 0
      v_sub_u32_e32 {T[0]}, {S.y1}, {R.sql1}
      v_mul_i32_i24_e32 {T[0]}, {T[0]}, {R.k_rate}
      s_add_u32 {S.t1}, {S.nv}, 16
      v_ashrrev_i32_e32 {T[0]}, 4, {T[0]}
      s_mul_i32 {S.u[1]}, {S.t1}, {4 * MIX_M}
      v_mad_i32_i24 {T[0]}, {T[0]}, {R.pl1}, {R.k_rnd}
      s_lshl_b32 {S.u[2]}, {S.u[1]}, 1
      v_ashrrev_i32_e32 {T[0]}, 13, {T[0]}
      s_lshl_b32 {S.u[3]}, {S.u[1]}, 2
      v_add_u32_e32 {T[0]}, {T[0]}, {R.mwl1s}
      s_lshl_b32 {S.u[4]}, {S.u[1]}, 3
      v_med3_i32 {R.w1_new}, {T[0]}, {S.m512k}, {R.k_c512k}
      v_add_u32_e32 {R.mrB[1]}, {S.u[1]}, {R.mx_rb}
      v_add3_u32 {R.mrB[2]}, {R.mx_rb}, {S.u[2]}, {R.k_pre2}
      v_add3_u32 {R.mrB[3]}, {R.mx_rb}, {S.u[3]}, {R.k_pre3}
      v_add3_u32 {R.mrB[4]}, {R.mx_rb}, {S.u[4]}, {R.k_pre4}
      buffer_load_dword {R.mwl[2]}, {R.mrB[2]}, %[rs], 0 offen
      buffer_load_dword {R.mwl[3]}, {R.mrB[3]}, %[rs], 0 offen
      buffer_load_dword {R.mwl[4]}, {R.mrB[4]}, %[rs], 0 offen
      ds_read_b128 {R.o14}, {R.k_rslot}
      v_mov_b32_e32 {R.o1off}, {R.rowoff}
      s_and_b32 {S.t0}, {S.nv}, 1
      s_cmp_lg_u32 {S.t0}, 0
      s_cselect_b64 {S.mk}, -1, 0
      s_waitcnt vmcnt(3)""")
    for p in range(3):
        for i in range(4):
            o(f"v_cndmask_b32_e64 {R.c[0][p][i]}, {R.c[0][p][i]}, {R.c[1][p][i]}, {S.mk}")
    o(f"""
      v_cndmask_b32_e64 {R.ch0[0]}, {R.ch0[0]}, {R.ch0[1]}, {S.mk}
      v_cndmask_b32_e64 {R.cchk[0]}, {R.cchk[0]}, {R.cchk[1]}, {S.mk}
      v_cndmask_b32_e64 {R.cmw[0]}, {R.cmw[0]}, {R.cmw[1]}, {S.mk}
      s_waitcnt lgkmcnt(0)""")
    row = [R.nA[1], R.nB[1], R.nA[2], R.nB[2]]
    sel = R.nA[3]
    cold += find(R.c[0], R.ch0[0], R.cchk[0], [(R.oboff, R.ob), (R.o1off, R.o1)], row, sel, "s")
    o(f"""
      s_and_b64 {S.mw}, {S.win}, {S.mii}
      s_and_saveexec_b64 {S.sav}, {S.mw}
      ds_write_b128 {R.k_slot}, v[198:201]
      s_mov_b64 exec, {S.win}
      ds_write_b32 {R.k_slotoff}, {sel}
      ds_write_b32 {R.k_mixb}, {R.cmw[0]}
      s_mov_b64 exec, {S.sav}
      ds_read_b128 {R.row4}, {R.k_rslot}
      ds_read_b32 {R.rowoff}, {R.k_slotoff}
      ds_read_b32 {R.mwl[1]}, {R.k_mixb}""")
    mixer_stores(R.mrA)
    o(f"""
      v_add_u32_e32 {T[0]}, {R.k_evo}, {R.o1off}
      buffer_store_dwordx4 {R.o14}, {T[0]}, %[rs], 0 offen""")
    # match_prefetch (zh_chain2.hip): what the byte boundary will want from the history — the first 64 byte pairs of the
    # candidate's verification, the byte the candidate / the running match predicts
    o(f"""
      v_readlane_b32 {S.w0}, {R.m_limit}, 6
      v_readlane_b32 {S.w1}, {R.cm_pre}, 6
      s_add_u32 {S.w0}, {S.w0}, 1
      s_and_b32 {S.w0}, {S.w0}, {S.mmsk}
      s_sub_u32 {S.w1}, {S.w0}, {S.w1}
      v_sub_u32_e32 {T[0]}, {S.w0}, {R.k_lane1}
      v_subrev_u32_e32 {T[1]}, {S.w1}, {T[0]}
      v_and_b32_e32 {T[0]}, {S.mmsk}, {T[0]}
      v_and_b32_e32 {T[1]}, {S.mmsk}, {T[1]}
      s_sub_u32 {S.w2}, {S.w0}, {S.w1}
      v_add_u32_e32 {T[0]}, {S.mbase}, {T[0]}
      v_add_u32_e32 {T[1]}, {S.mbase}, {T[1]}
      s_and_b32 {S.w2}, {S.w2}, {S.mmsk}
      v_sub_u32_e32 {T[2]}, {S.w0}, {R.m_ptr}
      s_add_u32 {S.w2}, {S.w2}, {S.mbase}
      v_and_b32_e32 {T[2]}, {S.mmsk}, {T[2]}
      buffer_load_ubyte {R.va_pre}, {T[0]}, %[rs], 0 offen
      v_add_u32_e32 {T[2]}, {S.mbase}, {T[2]}
      buffer_load_ubyte {R.vb_pre}, {T[1]}, %[rs], 0 offen
      buffer_load_ubyte {R.mbn_pre}, off, %[rs], {S.w2}
      buffer_load_ubyte {R.mbc_pre}, {T[2]}, %[rs], 0 offen
      s_waitcnt lgkmcnt(0)""")
    stamp(3)
    nibble(1, R.mrB)
    stamp(5)
    # ======== byte boundary
    o(f"""
      s_lshl_b32 {S.c}, {S.v1}, 4
      s_or_b32 {S.c}, {S.c}, {S.nv}
      v_mov_b32_e32 {R.sv2}, {S.c}
      v_and_b32_e32 {T[0]}, {R.m_limit}, {R.k_htmask}
      v_and_b32_e32 {T[1]}, {R.hv}, {R.k_cmmask}
      v_add_u32_e32 {T[2]}, 1, {R.m_limit}
      v_add3_u32 {R.sa1}, {T[0]}, {R.k_hto}, {R.k_evm}
      v_lshl_add_u32 {T[1]}, {T[1]}, 2, {R.k_cmo}
      v_and_b32_e32 {T[2]}, {T[2]}, {R.k_htmask}
      v_add_u32_e32 {R.sa2}, {T[1]}, {R.k_evm}
      v_cndmask_b32_e64 {R.m_limit}, {R.m_limit}, {T[2]}, {S.mmatch}
      s_lshl_b32 {S.t0}, %[bseq], 8
      s_or_b32 {S.t0}, {S.t0}, {S.c}
      v_mov_b32_e32 {T[0]}, {S.t0}
      s_mov_b64 exec, 1
      ds_write_b32 {R.k_mb}, {T[0]} offset:4
      s_mov_b64 exec, -1
      ds_read_b128 {R.ob4}, {R.k_rslot}
      v_mov_b32_e32 {R.oboff}, {R.rowoff}
      s_and_b32 {S.t1}, {S.c}, 31
      s_lshl_b32 {S.t2}, {S.t1}, 4
      s_lshl_b32 {S.t3}, {S.t1}, 5
      s_lshl_b32 {S.t1}, {S.t1}, 2
      v_add_u32_e32 {T[1]}, {S.t2}, {R.k_selrow}
      v_add_u32_e32 {T[2]}, {S.t1}, {R.k_seloff}
      v_add_u32_e32 {T[3]}, {S.t1}, {R.k_hspec}
      v_add_u32_e32 {T[4]}, {S.t3}, {R.k_mixst}""")
    label("staged")
    o(f"""
      ds_read_b32 {T[0]}, {R.k_mb} offset:8
      ds_read_b32 {R.hv}, {T[3]}
      ds_read_b128 v[198:201], {T[1]}
      ds_read_b32 {sel}, {T[2]}
      ds_read_b32 {R.sgmw}, {T[4]}
      s_waitcnt lgkmcnt(4)
      v_readfirstlane_b32 {S.t0}, {T[0]}
      s_cmp_lg_u32 {S.t0}, %[bseq]
      s_cbranch_scc1 .Lspin_%=
      s_waitcnt lgkmcnt(0)""")
    stamp(6)
    # the mixer's block of rows for the next byte (Predictor.cs:307: its context is h[7]); the first row's weights come staged —
    # unless the block is the one this byte used: then the helper's copy of row c8 = 1 may predate this byte's training of it
    o(f"""
      {'v_readlane_b32 ' + S.w0 + ', ' + R.hv + ', 7' if MIX_M == 7 else 's_mov_b32 ' + S.w0 + ', 0'}
      v_mov_b32_e32 {R.rbold}, {R.mx_rb}
      s_and_b32 {S.w0}, {S.w0}, {S.mxsize1}
      s_andn2_b32 {S.w0}, {S.w0}, 0xff
      s_mul_i32 {S.w0}, {S.w0}, {4 * MIX_M}
      s_add_u32 {S.w0}, {S.w0}, {S.mxbase}
      v_add_u32_e32 {R.mx_rb}, {S.w0}, {R.k_vomix}
      v_cmp_eq_u32_e32 vcc, {R.mx_rb}, {R.rbold}
      v_add_u32_e32 {R.mrA[1]}, {R.mx_rb}, {R.k_row1_1}
      v_add_u32_e32 {R.mrA[2]}, {R.mx_rb}, {R.k_row1_2}
      v_add_u32_e32 {R.mrA[3]}, {R.mx_rb}, {R.k_row1_3}
      v_add_u32_e32 {R.mrA[4]}, {R.mx_rb}, {R.k_row1_4}
      v_cndmask_b32_e32 {R.mwl[1]}, {R.sgmw}, {R.w1_new}, vcc
      buffer_load_dword {R.mwl[2]}, {R.mrA[2]}, %[rs], 0 offen
      buffer_load_dword {R.mwl[3]}, {R.mrA[3]}, %[rs], 0 offen
      buffer_load_dword {R.mwl[4]}, {R.mrA[4]}, %[rs], 0 offen
      v_cndmask_b32_e64 {R.mwl[1]}, 0, {R.mwl[1]}, {S.mfeed}""")
    if PROF_BND:
        stamp(9)
    # ---- the stores of this byte, behind those loads: MATCH's history byte and hash index (Predictor.cs:386-410 with the h[] of
    # the byte just coded), the second nibble's row, its trained mixer weights
    o(f"""
      buffer_store_byte {R.sv2}, {R.sa1}, %[rs], 0 offen
      buffer_store_dword {R.m_limit}, {R.sa2}, %[rs], 0 offen
      v_add_u32_e32 {T[0]}, {R.k_evo}, {R.oboff}
      buffer_store_dwordx4 {R.ob4}, {T[0]}, %[rs], 0 offen""")
    mixer_stores(R.mrB)
    if PROF_BND:
        stamp(10)
    # ---- match_boundary (Predictor.cs:391-410); the history bytes it compares were requested at the nibble switch
    o(f"""
      v_sub_u32_e32 {T[0]}, {R.m_limit}, {R.cm_pre}
      v_cmp_eq_u32_e64 {S.M0}, 0, {R.m_len}
      v_and_b32_e32 {T[1]}, {T[0]}, {R.k_htmask}
      s_and_b64 {S.M1}, {S.M0}, {S.mmatch}
      v_cmp_ne_u32_e32 vcc, 0, {T[1]}
      v_add_u32_e32 {T[2]}, 1, {R.m_len}
      v_cndmask_b32_e64 {R.m_ptr}, {R.m_ptr}, {T[0]}, {S.M1}
      v_min_u32_e32 {T[2]}, 0xff, {T[2]}
      s_and_b64 {S.M1}, {S.M1}, vcc
      v_cndmask_b32_e64 {R.m_len}, {T[2]}, {R.m_len}, {S.M0}
      s_cmp_lg_u64 {S.M1}, 0
      s_cbranch_scc1 .Lmverify_%=
      v_add_u32_e32 {T[0]}, -1, {R.m_ptr}
      s_waitcnt vmcnt(10)
      v_and_b32_e32 {T[1]}, 0xff, {R.mbc_pre}
      v_and_b32_e32 {T[0]}, {T[0]}, {R.k_htmask}
      v_cmp_ne_u32_e64 {S.M0}, 0, {R.m_len}
      v_cmp_eq_u32_e32 vcc, 0, {T[0]}
      s_and_b64 {S.M0}, {S.M0}, {S.mmatch}
      v_cndmask_b32_e32 {T[1]}, {T[1]}, {R.sv2}, vcc
      v_cndmask_b32_e64 {R.m_byte}, {R.m_byte}, {T[1]}, {S.M0}""")
    label("mdone")
    o(f"""
      v_lshl_add_u32 {T[0]}, {R.m_len}, 2, %[pmb]
      ds_read_b32 {T[1]}, {T[0]}
      v_and_b32_e32 {T[2]}, {R.hv}, {R.k_cmmask}
      v_lshl_add_u32 {T[2]}, {T[2]}, 2, {R.k_cmo}
      v_cndmask_b32_e64 {T[2]}, {R.k_koob}, {T[2]}, {S.mmatch}
      buffer_load_dword {R.cm_pre}, {T[2]}, %[rs], 0 offen""")
    if PROF_BND:
        stamp(11)
    # ---- is one of the two rows written late in the bucket the helper probed?
    o(f"""
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
      s_and_b64 vcc, vcc, {S.mii}
      s_waitcnt lgkmcnt(0)
      v_bfe_i32 {R.pm0}, {T[1]}, 0, 16
      v_ashrrev_i32_e32 {R.pm1}, 16, {T[1]}
      s_cbranch_vccnz .Lnear_%=""")
    label("taken")
    o(f"""
      s_and_b64 exec, {S.mcanon}, {S.mii}
      ds_write_b128 {R.k_slot}, v[198:201]
      s_mov_b64 exec, -1
      v_cndmask_b32_e64 {R.rx}, 0, v198, {S.mii}
      v_cndmask_b32_e64 {R.rq1}, 0, v199, {S.mii}
      v_cndmask_b32_e64 {R.rq2}, 0, v200, {S.mii}
      v_cndmask_b32_e64 {R.rq3}, 0, v201, {S.mii}
      v_mov_b32_e32 {R.rowoff}, {sel}
      s_add_u32 %[bseq], %[bseq], 1""")
    stamp(7)
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
      s_mov_b32 %[why], 2
      s_waitcnt lgkmcnt(0)
      s_branch .Lexit_%=""")
    # ---- MATCH: a candidate has to be verified (Predictor.cs:403-405: count equal bytes backwards, at most 255)
    o(".p2align 5")
    label("mverify")
    o(f"""
      s_waitcnt vmcnt(10)
      v_readlane_b32 {S.w0}, {R.m_limit}, 6
      v_readlane_b32 {S.w1}, {R.m_ptr}, 6
      v_and_b32_e32 {T[0]}, 0xff, {R.va_pre}
      v_and_b32_e32 {T[1]}, 0xff, {R.vb_pre}
      v_add_u32_e32 {T[2]}, -1, {R.k_lane1}
      v_cmp_eq_u32_e32 vcc, 0, {T[2]}
      v_add_u32_e32 {T[2]}, {S.w1}, {T[2]}
      v_cndmask_b32_e32 {T[0]}, {T[0]}, {R.sv2}, vcc
      v_and_b32_e32 {T[2]}, {S.mmsk}, {T[2]}
      v_cmp_eq_u32_e32 vcc, 0, {T[2]}
      s_nop 0
      v_cndmask_b32_e32 {T[1]}, {T[1]}, {R.sv2}, vcc
      v_cmp_ne_u32_e32 vcc, {T[0]}, {T[1]}
      s_cmp_lg_u64 vcc, 0
      s_cbranch_scc0 .Lmlong_%=
      s_ff1_i32_b64 {S.w2}, vcc""")
    label("mlen")
    o(f"""
      s_min_u32 {S.w2}, {S.w2}, 0xff
      s_sub_u32 {S.w1}, {S.w1}, 1
      s_and_b32 {S.w1}, {S.w1}, {S.mmsk}
      v_and_b32_e32 {T[0]}, 0xff, {R.mbn_pre}
      s_cmp_eq_u32 {S.w1}, 0
      v_mov_b32_e32 {T[1]}, {S.w2}
      s_cselect_b64 vcc, -1, 0
      v_cndmask_b32_e64 {R.m_len}, {R.m_len}, {T[1]}, {S.mmatch}
      v_cndmask_b32_e32 {T[0]}, {T[0]}, {R.sv2}, vcc
      v_cndmask_b32_e64 {R.m_byte}, {R.m_byte}, {T[0]}, {S.mmatch}
      s_branch .Lmdone_%=""")
    # all 64 pairs equal: bytes 64..254 from memory, 64 at a time
    label("mlong")
    o(f"""
      s_mov_b32 {S.w2}, 64
      s_mov_b32 {S.spin}, 64""")
    label("mlong1")
    o(f"""
      v_add_u32_e32 {T[2]}, {S.spin}, {R.k_lane1}
      v_sub_u32_e32 {T[0]}, {S.w0}, {T[2]}
      v_subrev_u32_e32 {T[1]}, {S.w1}, {T[0]}
      v_and_b32_e32 {T[0]}, {S.mmsk}, {T[0]}
      v_and_b32_e32 {T[1]}, {S.mmsk}, {T[1]}
      v_add_u32_e32 {T[0]}, {S.mbase}, {T[0]}
      v_add_u32_e32 {T[1]}, {S.mbase}, {T[1]}
      buffer_load_ubyte {T[3]}, {T[0]}, %[rs], 0 offen
      buffer_load_ubyte {T[4]}, {T[1]}, %[rs], 0 offen
      v_cmp_gt_u32_e32 vcc, 0x100, {T[2]}
      s_mov_b64 {S.M2}, vcc
      s_waitcnt vmcnt(0)
      v_cmp_eq_u32_e32 vcc, {T[3]}, {T[4]}
      s_and_b64 vcc, vcc, {S.M2}
      s_not_b64 vcc, vcc
      s_cmp_lg_u64 vcc, 0
      s_cbranch_scc0 .Lmlong2_%=
      s_ff1_i32_b64 {S.t0}, vcc
      s_add_u32 {S.w2}, {S.w2}, {S.t0}
      s_branch .Lmlen_%=""")
    label("mlong2")
    o(f"""
      s_add_u32 {S.w2}, {S.w2}, 64
      s_add_u32 {S.spin}, {S.spin}, 64
      s_cmp_lt_u32 {S.spin}, 256
      s_cbranch_scc1 .Lmlong1_%=
      s_branch .Lmlen_%=""")
    # ---- a row written late lies in the probed bucket: the helper's three probes, patched, and find
    o(".p2align 5")
    label("near")
    # The common reason to be here is the ICM of order 0 (h = 0): the next byte's first nibble reads the very row this byte's first
    # nibble used.  Predictor.find returns probe 0's place when the row there carries the context's check byte — and what lies
    # there now is the row this wave wrote back (old1 at the nibble switch, oldb just now: the later store wins).  Lanes for
    # which that holds take the row from the registers it was written from; only a lane whose late row lies at another place
    # of the bucket, or under another check byte, needs the three probes patched and searched.
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
      s_and_b64 {S.M1}, {S.M1}, {S.mii}
      s_and_b64 {S.M0}, {S.M0}, {S.mii}
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
    vregs = [R.rx, R.rq1, R.rq2, R.rq3, R.rowoff, R.hv, R.ob[0], R.ob[1], R.ob[2], R.ob[3], R.oboff, R.park, None] + \
            [getattr(R, n) for n in VX[:13]] + [R.mwl[1], R.mwl[2], R.mwl[3], R.mwl[4], R.mrA[1], R.mrA[2], R.mrA[3], R.mrA[4]]
    o("s_waitcnt vmcnt(0)")
    for i, r in enumerate(vregs):
        if r:
            o(f"ds_write_b32 %[vb], {r} offset:{i * 256}")
    if PROF:
        for i in range(12):
            o(f"ds_write_b32 %[vb], v{32 + i} offset:{(len(VNAMES) + i) * 256}")
    o(f"""
      s_mov_b32 m0, %[m0s]
      s_mov_b32 %[obad], {S.bad}
      s_mov_b32 %[ofail], {S.fail}
      s_waitcnt lgkmcnt(0)""")


def emit(name, prof, mix_m=7):
    global PROF, MIX_M
    PROF = prof
    MIX_M = mix_m
    G.PROF = False
    del L[:]
    gen()
    clob = ["memory", "scc", "vcc"] + [f"s{i}" for i in range(40, 102)] + [f"v{i}" for i in range(32 if prof else 48, 256)]
    text = f"""#define {name}(low_, high_, curr_, k_, bseq_, nput_, room_, word_, why_, obad_, ofail_, m0s_, klim_, vlo_, kb_, vb_, rs_, sqb_, nsb_, mxb_, mxs_, pmb_) \\
  asm volatile( \\
"""
    for ln in L:
        text += '  "' + ln.replace('"', '\\"') + '\\n\\t" \\\n'
    text += """  : [low] "+s"(low_), [high] "+s"(high_), [curr] "+s"(curr_), [k] "+s"(k_), [bseq] "+s"(bseq_), [nput] "+s"(nput_), [room] "+s"(room_), \\
    [word] "+s"(word_), [why] "=&s"(why_), [obad] "=&s"(obad_), [ofail] "=&s"(ofail_), [m0s] "=&s"(m0s_) \\
  : [klim] "s"(klim_), [vlo] "s"(vlo_), [kb] "v"(kb_), [vb] "v"(vb_), [rs] "s"(rs_), [sqb] "s"(sqb_), [nsb] "s"(nsb_), [mxb] "s"(mxb_), [mxs] "s"(mxs_), [pmb] "s"(pmb_) \\
  : """ + ", ".join('"' + c + '"' for c in clob) + ")\n"
    return text, len(L)


def main():
    t0, n0 = emit("ZH_NB_FAST_MID_LOOP", False)
    t1, _ = emit("ZH_NB_FAST_MID_LOOP_PROF", True)
    t2, _ = emit("ZH_NB_FAST_MID8_LOOP", False, 8)
    t3, _ = emit("ZH_NB_FAST_MID8_LOOP_PROF", True, 8)
    head = f"""// zh_nb_fast_mid.h — GENERATED by tools/gen_nb_asm_mid.py (do not edit: edit the generator and run it).
// The steady-state byte loop of nb_fast (zh_nibble.hip) for the built-in mid model, hand-laid gfx950 assembly.
#pragma once
#define ZH_NB_FAST_MID 1
enum : int {{ {", ".join("kNmK_" + n + (" = 0" if i == 0 else "") for i, n in enumerate(KNAMES))}, kNmK_count }};
enum : int {{ {", ".join("kNmS_" + n + (" = 0" if i == 0 else "") for i, n in enumerate(VNAMES))}, kNmS_count }};
// clang-format off
"""
    text = head + t0 + t1 + t2 + t3 + "// clang-format on\n"
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"{OUT}: {n0} lines of assembly")


if __name__ == "__main__":
    main()
