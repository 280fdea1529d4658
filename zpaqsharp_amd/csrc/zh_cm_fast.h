// zh_cm_fast.h — the steady-state byte loop of wave A (zh_cm.hip) written directly in gfx950 assembly.
//
// One iteration = one byte of Decoder.decompress (Decoder.cs:32-56) for a block whose model is a
// single direct CM and whose HCOMP is "a<<= K  *d=a  halt" with K >= 9 (so the low 9 bits of the
// context hash are zero: first-nibble group 0, second-nibble groups 16..31 of the window).
//
// The loop only runs when nothing unusual can happen inside the byte.  It checks, before it
// changes any state, that
//   * at least 40 coded bytes are in the register-held chunk (a byte consumes at most 9 x 4), so
//     a renormalisation never has to refill or can hit EOF (the test is part of the lag compare, see below);
//   * the context's window is resident (otherwise the swap wave brings it in: .Lzh_miss, served without leaving the loop);
//   * wave B has finished every earlier byte that touched that window and the ring has room;
//   * the EOS flag decodes as 0 (Decoder.cs:138; the state is in range by the invariant stated at the
//     loop); an unprimed coder (curr == 0 < low) fails this test too, so priming needs no test of its own.
// If any test fails it leaves with code 0 and the C++ body of the loop in zh_cm.hip handles that
// byte (it is the same algorithm, including the rare cases), then re-enters.  Code 1 = the range
// check after a renormalisation failed ("archive corrupted").
//
// Per decoded bit: v_readlane (probability of tree node j), 8 SALU instructions for the split
// (Decoder.cs:140-147; mulhi against p16 << 16 replaces the 64-bit multiply and shift; y = curr <= mid
// as the reference words it), and xor + compare + branch for the renormalisation, which is out of line.
//
// Register use: operands are allocated by the compiler; temporaries are the fixed registers
// s76-s77, s80-s94 and v237-v245, v249-v252 (declared as clobbers).  exec is all ones on entry and exit.
#pragma once

// clang-format off
// One bit: the probability of tree node IDX comes out of lane IDX of PV, SHADOW is vector work that rides in the v_readlane's
// shadow (see below), then the split and the renormalisation test of THIS split (Decoder.cs:148-156 runs right after the
// split: nothing between the two reads low / high / curr).
// (Measured on one GPU box and not kept, round 3: the test moved behind the next step's v_readlane — 554 against 565 MB/s;
// the test replaced by `high - low < 2^24`, which the step computes anyway — 531 MB/s: ranges below 2^24 whose top bytes
// differ are common, and each one sends the wave out of line for nothing.)
#define ZH_FAST_STEP_(PV, IDX, N, ADDC, SHADOW)                       \
  "v_readlane_b32 s94, " PV ", " IDX "\n\t"                           \
  SHADOW                                                              \
  "s_sub_u32 s84, %[high], %[low]\n\t"                                \
  "s_mul_hi_u32 s86, s84, s94\n\t"                                    \
  "s_add_u32 s87, %[low], s86\n\t"                                    \
  "s_add_u32 s88, s87, 1\n\t"                                         \
  "s_cmp_le_u32 %[curr], s87\n\t"                                     \
  "s_cselect_b32 %[high], s87, %[high]\n\t"                           \
  "s_cselect_b32 %[low], %[low], s88\n\t"                             \
  ADDC "\n\t"                                                         \
  "s_xor_b32 s84, %[high], %[low]\n\t"                                \
  "s_cmp_lt_u32 s84, %[c24]\n\t"                                      \
  "s_cbranch_scc1 .Lzh_rn" #N "_%=\n"                                 \
  ".Lzh_bk" #N "_%=:\n\t"
// second nibble: the next step's lane (s89 + j2) is computed right behind the s_addc, three instructions ahead of the
// v_readlane that takes it as its lane select (a lane select the SALU has only just written costs the v_readlane extra)
#define ZH_FAST_ADDC_IDX(J) "s_addc_u32 " J ", " J ", " J "\n\ts_add_u32 s80, s89, " J
#define ZH_FAST_ADDC1_IDX(J) "s_addc_u32 " J ", 1, 1\n\ts_add_u32 s80, s89, " J

// out-of-line blocks start on a 32-byte fetch window (each is entered by a taken branch and left by one: the padding is never run)
#ifndef ZH_L1_COLD_ALIGN
#define ZH_L1_COLD_ALIGN 1
#endif
#if ZH_L1_COLD_ALIGN
#define ZH_FAST_COLD_ALIGN ".p2align 5\n"
#else
#define ZH_FAST_COLD_ALIGN ""
#endif
// Renormalisation (Decoder.cs:148-156): shift a coded byte in while the top bytes of low and
// high agree; then the range test the next decode() would make (after the byte's last bit: ZH_FAST_CHK8).
#define ZH_FAST_SHIFT_IN(N)                                           \
  ".Lzh_rl" #N "_%=:\n\t"                                             \
  "s_lshl_b32 %[high], %[high], 8\n\t"                                \
  "s_or_b32 %[high], %[high], 0xff\n\t"                               \
  "s_lshl_b32 %[low], %[low], 8\n\t"                                  \
  "s_max_u32 %[low], %[low], 1\n\t"                                   \
  "s_lshr_b32 s84, %[k], 2\n\t"                                       \
  "v_readlane_b32 s85, %[cur], s84\n\t"                               \
  "s_lshl_b32 s84, %[k], 3\n\t"                                       \
  "s_lshr_b32 s85, s85, s84\n\t"                                      \
  "s_and_b32 s85, s85, 0xff\n\t"                                      \
  "s_lshl_b32 %[curr], %[curr], 8\n\t"                                \
  "s_or_b32 %[curr], %[curr], s85\n\t"                                \
  "s_add_u32 %[k], %[k], 1\n\t"                                       \
  "s_xor_b32 s84, %[high], %[low]\n\t"                                \
  "s_cmp_lt_u32 s84, 0x1000000\n\t"                                   \
  "s_cbranch_scc1 .Lzh_rl" #N "_%=\n\t"
// ... after the EOS flag (N = 0) or bit N (N = 1..7)
#define ZH_FAST_RENORM(N)                                             \
  ZH_FAST_COLD_ALIGN                                                  \
  ".Lzh_rn" #N "_%=:\n\t"                                             \
  ZH_FAST_SHIFT_IN(N)                                                 \
  ZH_FAST_CHK                                                         \
  "s_branch .Lzh_bk" #N "_%=\n\t"
// ... after the byte's last bit
#define ZH_FAST_RENORM_LAST(N, S)                                     \
  ZH_FAST_COLD_ALIGN                                                  \
  ".Lzh_rn" #N "_%=:\n\t"                                             \
  ZH_FAST_SHIFT_IN(N)                                                 \
  ZH_FAST_CHK8(S)                                                     \
  "s_branch .Lzh_bk" #N "_%=\n\t"

#define ZH_FAST_CHK                                                   \
  "s_cmp_lt_u32 %[curr], %[low]\n\t"                                  \
  "s_cbranch_scc1 .Lzh_corrupt_%=\n\t"                                \
  "s_cmp_gt_u32 %[curr], %[high]\n\t"                                 \
  "s_cbranch_scc1 .Lzh_corrupt_%=\n\t"

// The last bit's renormalisation: a state out of range is not an error of THIS byte (Decoder.cs:138 raises it at the
// next decode() call), so the byte is still published and the loop is left; the C++ body's EOS test reports it.
#define ZH_FAST_CHK8(S)                                               \
  "s_cmp_lt_u32 %[curr], %[low]\n\t"                                  \
  "s_cbranch_scc1 .Lzh_oor" #S "_%=\n\t"                              \
  "s_cmp_gt_u32 %[curr], %[high]\n\t"                                 \
  "s_cbranch_scc1 .Lzh_oor" #S "_%=\n\t"

// byte = (j << 4) + j2 - 272;  message to wave B: tag(t) << 25 | byte << 15 | slot, written by lane 0 to the ring slot
// of message t (the other lanes of `vr` hold the addresses of their own dummy words: no exec switch around the write);
// then the ring address of lane 0 steps on (v_bfi keeps the other lanes), t, the slot's last-use stamp (vcc still is
// the one-hot lane mask of the window lookup: nothing in the loop writes vcc after it), h[0]
// (round 4) Two rules of a lone wave's instruction stream shape this loop (profiles/r04/ab_notes.txt, calls 14-26):
//
// 1. The first SALU instruction behind a VALU instruction that writes an SGPR (v_readlane, v_cmp) issues ~21 cycles after it,
//    whatever it reads; three to four VALU instructions in between are free (tools/ubench/sgprw_bench,
//    profiles/r04/ubench_sgprw.txt).  So everything of a byte's bookkeeping that is vector work, or can be made vector work,
//    sits in the SHADOW of a step's v_readlane:
//      step 1: wave B's progress counter, read from LDS with the byte's probabilities, is taken over (a second SGPR-writing
//              VALU instruction back to back costs one issue slot, not a second stall): the NEXT byte's lag test works on a
//              counter one byte old instead of one refreshed only after a failed test;  tag | slot of the message begins;
//              v238 = t + 1 - 13 (t lives in a VGPR, v241, inside the loop: nothing scalar on the hot path needs it)
//      step 2: the window's last-use stamp t + 1 (vcc still is the one-hot lane mask of the window lookup: nothing in the loop
//              writes vcc after it)
//      step 3: the message without the byte, tag(t) << 25 | slot
//      steps 5, 6: the next ring address of lane 0 (v_bfi keeps the other lanes, which point at dummy words: no exec switch
//              around the write; v251 is free since the nibble switch)
//      steps 7, 8: v242 = max(last use, t + 1 - 13, (k - klim) | 2^30) — the NEXT byte's lag test and its chunk test as ONE
//              compare against wave B's counter: wave B may be at most min(messages since the window's last use, 13) behind,
//              t - bdone <= min(t - lu, 13)  <=>  max(lu, t - 13) <= bdone  (signed: t - 13 is negative at first; lanes that
//              stand for no slot carry -1 and are masked by the lookup; zh_cm.hip keeps t below 2^30 - 2^25), and k - klim is
//              negative while enough coded bytes are ahead.  k is read in step 7, before the renormalisations of the last two
//              steps (at most 4 bytes each): klim = avail - 48 leaves the 40 a byte can consume.  The out-of-line wait tells
//              the two reasons apart.
//    The epilogue then is: byte = (j << 4) + j2 - 272, message = byte << 15 | (tag | slot) by one v_lshl_or, the write, h[0].
// 2. Instruction fetch does not run ahead across a conditional branch: a NOT-TAKEN branch whose next two instructions do not
//    lie in the branch's own 32-byte window costs ~19 cycles (the same loop shifted in 4-byte steps: 580 .. 664 MB/s, period
//    32 bytes).  Every bit step therefore is exactly 64 bytes (the constants of the renormalisation tests in SGPRs, s_addk
//    for the -272, v_nop in the shadows as free filler), the nibble switch 32, the byte 672, and ZH_L1_PAD puts a step's
//    v_readlane 16 bytes into a window: its branch sits at 12, the v_readlane and the first shadow instruction behind it.
//    ANY edit of the hot path has to keep this (llvm-objdump -d with addresses; then sweep ZH_L1_PAD).
// Measured and not kept: the step with the next bit's probability fetched both ways (pairs of children per lane, two v_readlane
// while the split is computed, one s_cselect by the bit: the chain s_addc -> v_readlane -> s_mul_hi becomes s_cselect ->
// s_mul_hi) — 522 against 595 MB/s: the ~21 cycles behind a v_readlane are not a dependency a schedule can cover.
#define ZH_FAST_READ_BSEQ "ds_read_b32 v245, %[bsa]\n\t"
#define ZH_FAST_SHADOW1 "v_readfirstlane_b32 %[bdone], v245\n\tv_mov_b32_e32 v243, s82\n\tv_add_u32_e32 v238, -12, v241\n\t"
#define ZH_FAST_SHADOW2 "v_add_u32_e32 v252, 1, v241\n\tv_cndmask_b32_e32 %[lu], %[lu], v252, vcc\n\tv_nop\n\t"
#define ZH_FAST_SHADOW3 "v_lshl_or_b32 v243, v241, 25, v243\n\tv_nop\n\t"
#define ZH_FAST_SHADOW4 "v_nop\n\tv_nop\n\tv_nop\n\t"
#define ZH_FAST_SHADOW5 "v_add_u32_e32 v251, 4, %[vr]\n\tv_nop\n\t"
#define ZH_FAST_SHADOW6 "v_bfi_b32 v251, %[vm], v251, %[vr]\n\t"
#define ZH_FAST_SHADOW7 "v_sub_u32_e32 v239, %[k], v240\n\tv_or_b32_e32 v239, v237, v239\n\t"
#define ZH_FAST_SHADOW8 "v_max3_i32 v242, v238, %[lu], v239\n\tv_add_u32_e32 v241, 1, v241\n\t"
// v242 for THIS byte: at loop entry, and after a miss has re-stamped a slot
#define ZH_FAST_LAGV "v_add_u32_e32 v244, -13, v241\n\tv_sub_u32_e32 v239, %[k], v240\n\tv_or_b32_e32 v239, v237, v239\n\tv_max3_i32 v242, v244, %[lu], v239\n\t"
// loop entry: t, the chunk limit and 2^30 into their VGPRs;  ZH_FAST_T: t back into its SGPR (out-of-line paths, exit)
#define ZH_FAST_ENTRY "v_mov_b32_e32 v241, %[t]\n\tv_mov_b32_e32 v240, %[klim]\n\tv_mov_b32_e32 v237, 0x40000000\n\t"
#define ZH_FAST_T "v_readfirstlane_b32 %[t], v241\n\t"
#define ZH_FAST_EPILOGUE                                              \
  "s_lshl4_add_u32 s92, s90, s91\n\t"                                 \
  "s_addk_i32 s92, 0xfef0\n\t"                                        \
  "v_lshl_or_b32 v250, s92, 15, v243\n\t"                             \
  "ds_write_b32 %[vr], v250\n\t"                                      \
  "v_mov_b32_e32 %[vr], v251\n\t"                                     \
  "s_lshl_b32 %[h0], s92, %[hs]\n\t"
// window lookup (lane s of `tag` holds the window id cached in slot s) and lag test; the out-of-line wait computes the
// allowance min(t - lu[slot], 13) itself
#define ZH_FAST_LOOKUP(S)                                             \
  "s_bfe_u32 s81, %[h0], %[bfe]\n\t"                                  \
  "v_cmp_eq_u32_e32 vcc, s81, %[tag]\n\t"                             \
  "v_cmp_gt_i32_e64 s[76:77], v242, %[bdone]\n\t"                     \
  "s_cbranch_vccz .Lzh_miss_%=\n\t"                                   \
  "s_ff1_i32_b64 s82, vcc\n\t"                                        \
  "s_and_b64 s[76:77], s[76:77], vcc\n\t"                             \
  "s_cbranch_scc1 .Lzh_fresh" #S "_%=\n"
#define ZH_FAST_FRESH_IN ZH_FAST_T "s_cmp_ge_u32 %[k], %[klim]\n\ts_cbranch_scc1 .Lzh_slow_%=\n\tv_readlane_b32 s83, %[lu], s82\n\ts_sub_u32 s83, %[t], s83\n\ts_min_u32 s83, s83, 13\n\t"
// the eight bit steps: first nibble, node j in lane j of v249 (high half); second nibble, group n1 = quad (n1 & 3), element
// n1 >> 2 of the four candidates a lane holds in v[250:251]
#define ZH_FAST_BITS(N1, N2, N3, N4, N5, N6, N7, N8)                  \
  ZH_FAST_STEP_("v249", "1", N1, "s_addc_u32 s90, 1, 1", ZH_FAST_SHADOW1)\
  ZH_FAST_STEP_("v249", "s90", N2, "s_addc_u32 s90, s90, s90", ZH_FAST_SHADOW2)\
  ZH_FAST_STEP_("v249", "s90", N3, "s_addc_u32 s90, s90, s90", ZH_FAST_SHADOW3)\
  ZH_FAST_STEP_("v249", "s90", N4, "s_addc_u32 s90, s90, s90", ZH_FAST_SHADOW4)    \
  "s_and_b32 s89, s90, 3\n\t"                                         \
  "s_lshl_b32 s89, s89, 4\n\t"                                        \
  "s_lshl_b32 s83, s90, 2\n\t"                                        \
  "s_and_b32 s83, s83, 48\n\t"                                        \
  "s_add_u32 s80, s89, 1\n\t"                                         \
  "v_lshrrev_b64 v[250:251], s83, v[250:251]\n\t"                     \
  "v_lshlrev_b32_e32 v250, 16, v250\n\t"                              \
  ZH_FAST_STEP_("v250", "s80", N5, ZH_FAST_ADDC1_IDX("s91"), ZH_FAST_SHADOW5)      \
  ZH_FAST_STEP_("v250", "s80", N6, ZH_FAST_ADDC_IDX("s91"), ZH_FAST_SHADOW6)\
  ZH_FAST_STEP_("v250", "s80", N7, ZH_FAST_ADDC_IDX("s91"), ZH_FAST_SHADOW7)\
  ZH_FAST_STEP_("v250", "s80", N8, "s_addc_u32 s91, s91, s91", ZH_FAST_SHADOW8)
// s76:s77 lag-test mask   s80 lane of the next second-nibble step / scratch   s81 window id   s82 slot
// s83 allowance of the out-of-line wait / shift of the nibble switch   s84-s88 step scratch   s89 first lane of the group's quad
// s90 j (16|n1) s91 j2 (16|n2)  s92 byte   s94 probability
// v237 2^30   v238 t + 1 - 13   v239 chunk term   v240 klim   v241 t   v242 lag-test operand   v243 message without the byte
// v244 scratch   v245 wave B's counter as read from LDS
// v250:v251 four second-nibble probabilities / selected one (v251 later: next ring address)   v252 scratch
// v249 first-nibble probabilities << 16 (loaded into the high half: ds_read_u16_d16_hi; its low half stays zero)
//
// Invariant on entry and at every .Lzh_byte: low <= curr <= high unless the coder is unprimed (curr == 0 < low) — every
// renormalisation inside the loop re-checks it (ZH_FAST_CHK / ZH_FAST_CHK8), a split keeps it — so the EOS flag (p = 0:
// y = curr <= low) needs one compare.  zh_cm.hip enters the loop only with the state in range.
// One byte of the loop (hot part) and its out-of-line parts; S = copy suffix, E = label number of the renormalisation after
// the EOS flag, N1..N8 = label numbers of the eight bit steps.  The loop body is laid out TWICE (a taken s_branch costs a lone
// wave ~21 cycles, tools/ubench/salu_bench: the second copy falls through from the first and only it branches back; same-box
// A/B: one copy 564.6, two 571.6, four 568.7 MB/s).
#define ZH_CM_FAST_BYTE(S, E, N1, N2, N3, N4, N5, N6, N7, N8, H0, H1, H2, H3) \
  H0                                                                  \
  ZH_FAST_LOOKUP(S)                                                   \
  ".Lzh_ok" #S "_%=:\n\t"                                             \
  H1                                                                  \
  /* cached probabilities: lane j <- node j of the first nibble; lane (q, j) <- node j of groups q, q+4, q+8, q+12. */ \
  /* Issued before the remaining tests so that their latency is covered; a slow exit waits for them. */ \
  "v_lshl_add_u32 v252, s82, 5, %[la]\n\t"                            \
  "v_lshl_add_u32 v250, s82, 9, %[lb]\n\t"                            \
  "ds_read_u16_d16_hi v249, v252\n\t"                                 \
  "ds_read_b64 v[250:251], v250\n\t"                                  \
  ZH_FAST_READ_BSEQ                                                   \
  /* (enough coded bytes in the chunk, k < klim: part of the lag compare, ZH_FAST_SHADOW7 / 8) */ \
  /* EOS flag (p = 0): y = curr <= low; leave when y = 1 (or the coder is unprimed) */ \
  "s_cmp_le_u32 %[curr], %[low]\n\t"                                  \
  "s_cbranch_scc1 .Lzh_slow_%=\n\t"                                   \
  "s_add_u32 %[low], %[low], 1\n\t"                                   \
  /* the state was normalised (every split is followed by its renormalisation): the + 1 matters only when it carries into the top byte */ \
  "s_and_b32 s84, %[low], %[c24m]\n\t"                                \
  "s_cbranch_scc0 .Lzh_e0" #S "_%=\n"                                 \
  ".Lzh_bk" #E "_%=:\n\t"                                             \
  "s_waitcnt lgkmcnt(0)\n\t"                                          \
  H2                                                                  \
  ZH_FAST_BITS(N1, N2, N3, N4, N5, N6, N7, N8)                        \
  H3                                                                  \
  ZH_FAST_EPILOGUE
// SPIN_IN: nothing in the product build, a time stamp in the diagnostic one; SPIN_OK(S): where the spin leaves to when B has caught up —
// straight back into the byte (product), or through a block that stamps the time wave A spent waiting (diagnostic)
#define ZH_FAST_OK_DIRECT(S) ".Lzh_ok" #S "_%="
#define ZH_FAST_OK_STAMPED(S) ".Lzh_okp" #S "_%="
#define ZH_CM_FAST_COLD(S, E, N1, N2, N3, N4, N5, N6, N7, N8, SPIN_IN, SPIN_OK) \
  /* wave B is behind: re-read its progress counter a bounded number of times, then give up */ \
  ZH_FAST_COLD_ALIGN                                                  \
  ".Lzh_fresh" #S "_%=:\n\t"                                          \
  ZH_FAST_FRESH_IN                                                    \
  SPIN_IN                                                             \
  "s_mov_b32 s80, 48\n"                                               \
  ".Lzh_spin" #S "_%=:\n\t"                                           \
  "ds_read_b32 v252, %[bsa]\n\t"                                      \
  "s_waitcnt lgkmcnt(0)\n\t"                                          \
  "v_readfirstlane_b32 %[bdone], v252\n\t"                            \
  "s_sub_u32 s89, %[t], %[bdone]\n\t"                                 \
  "s_cmp_gt_u32 s89, s83\n\t"                                         \
  "s_cbranch_scc0 " SPIN_OK(S) "\n\t"                                 \
  "s_sub_u32 s80, s80, 1\n\t"                                         \
  "s_cmp_lg_u32 s80, 0\n\t"                                           \
  "s_cbranch_scc1 .Lzh_spin" #S "_%=\n\t"                             \
  "s_branch .Lzh_slow_%=\n\t"                                         \
  ZH_FAST_COLD_ALIGN                                                  \
  ".Lzh_e0" #S "_%=:\n\t"                                             \
  "s_xor_b32 s84, %[high], %[low]\n\t"                                \
  "s_cmp_lt_u32 s84, %[c24]\n\t"                                      \
  "s_cbranch_scc0 .Lzh_bk" #E "_%=\n\t"                               \
  "s_branch .Lzh_rn" #E "_%=\n\t"                                     \
  ZH_FAST_RENORM(E)                                                   \
  ZH_FAST_RENORM(N1)                                                  \
  ZH_FAST_RENORM(N2)                                                  \
  ZH_FAST_RENORM(N3)                                                  \
  ZH_FAST_RENORM(N4)                                                  \
  ZH_FAST_RENORM(N5)                                                  \
  ZH_FAST_RENORM(N6)                                                  \
  ZH_FAST_RENORM(N7)                                                  \
  ZH_FAST_RENORM_LAST(N8, S)                                          \
  ".Lzh_oor" #S "_%=:\n\t"                                            \
  ZH_FAST_EPILOGUE                                                    \
  "s_branch .Lzh_slow_%=\n"

// Where the loop lies against its 256-byte alignment (dwords of padding behind the .p2align, jumped over at entry):
// -DZH_L1_PAD=n shifts the unchanged body by 4n bytes (profiles/r04/ab_notes.txt, calls 22, 26)
#ifndef ZH_L1_PAD
#define ZH_L1_PAD 4
#endif
#define ZH_STR_(x) #x
#define ZH_STR(x) ZH_STR_(x)
#define ZH_FAST_PAD ".fill " ZH_STR(ZH_L1_PAD) ", 4, 0xbf800000\n"
// The window miss, served inside the loop: ONE definition for the product loop and the stamped diagnostic loop below (a fix in
// one copy used to diverge the stage tables from the product)
#define ZH_FAST_MISS_BLOCK                                            \
  /* ---- window miss, served without leaving the loop (round 4: the stage table of a miss, profiles/r04/stages_l1_miss_before.txt, \
     showed ~1 000 of its ~3 000 cycles between leaving this loop and re-entering it, and most of the rest waiting for wave B to \
     finish the byte before).  Victim = the first slot not used since message `thr` (empty slots carry 0, lanes that stand for no \
     slot ~0); the directory is updated here and the swap wave (wave C, zh_cm.hip) gets victim, window and slot through three LDS \
     words — every lane writes the same value to the same word: no exec switch —; the wave spins on C's answer and starts the \
     byte over: the lookup then hits.  The fresh window is stamped as if used 13 messages ago: nothing of it is outstanding with \
     wave B, 13 is what the ring can hold.  No candidate, or a window id beyond 16 bits: the C++ body serves it. */ \
  ZH_FAST_COLD_ALIGN                                                  \
  ".Lzh_miss_%=:\n\t"                                                 \
  ZH_FAST_T                                                           \
  "s_cmp_ge_u32 s81, 0x10000\n\t"                                     \
  "s_cbranch_scc1 .Lzh_slow_%=\n\t"                                   \
  "v_cmp_gt_u32_e32 vcc, %[thr], %[lu]\n\t"                           \
  "s_cbranch_vccnz .Lzh_mvic_%=\n\t"                                  \
  /* no slot that old: the threshold moves up to kb messages ago (pick_victim in zh_cm.hip is the same rule) */ \
  "s_sub_u32 %[thr], %[t], %[kb]\n\t"                                 \
  "s_max_i32 %[thr], %[thr], 1\n\t"                                   \
  "v_cmp_gt_u32_e32 vcc, %[thr], %[lu]\n\t"                           \
  "s_cbranch_vccz .Lzh_slow_%=\n"                                     \
  ".Lzh_mvic_%=:\n\t"                                                 \
  "s_ff1_i32_b64 s82, vcc\n\t"                                        \
  "s_mov_b32 s84, m0\n\t"                                             \
  "s_mov_b32 m0, s82\n\t"                                             \
  "s_add_u32 %[nm], %[nm], 1\n\t"                                     \
  "s_and_b32 s85, %[nm], 0xff\n\t"                                    \
  "s_lshl_b32 s80, s85, 23\n\t"                                       \
  "s_lshl_b32 s86, s81, 6\n\t"                                        \
  "s_or_b32 s80, s80, s86\n\t"                                        \
  "s_or_b32 s80, s80, s82\n\t"                                        \
  "v_readlane_b32 s83, %[tag], m0\n\t"                                \
  "v_writelane_b32 %[tag], s81, m0\n\t"                               \
  "v_mov_b32_e32 v250, s83\n\t"                                       \
  "ds_write_b32 %[mqa], v250 offset:4\n\t"                            \
  "v_mov_b32_e32 v250, s80\n\t"                                       \
  "ds_write_b32 %[mqa], v250\n\t"                                     \
  "s_sub_u32 s87, %[t], 13\n\t"                                       \
  "s_max_i32 s87, s87, 1\n\t"                                         \
  "v_writelane_b32 %[lu], s87, m0\n\t"                                \
  ZH_FAST_LAGV                                                        \
  "s_mov_b32 m0, s84\n\t"                                             \
  "s_mov_b32 s80, 0x4000\n"                                           \
  ".Lzh_mspin_%=:\n\t"                                                \
  "ds_read_b32 v252, %[mqa] offset:8\n\t"                             \
  "s_waitcnt lgkmcnt(0)\n\t"                                          \
  "v_readfirstlane_b32 s86, v252\n\t"                                 \
  "s_cmp_eq_u32 s86, s85\n\t"                                         \
  "s_cbranch_scc1 .Lzh_byte_%=\n\t"                                   \
  "s_sub_u32 s80, s80, 1\n\t"                                         \
  "s_cmp_lg_u32 s80, 0\n\t"                                           \
  "s_cbranch_scc1 .Lzh_mspin_%=\n\t"                                  \
  "s_branch .Lzh_slow_%=\n"
#define ZH_CM_FAST_LOOP(low_, high_, curr_, k_, t_, h0_, bdone_, lu_, code_, klim_, bfe_, hs_, vr_, vm_, bsa_, cur_, tag_, la_, lb_, thr_, nm_, mqa_, kb_) \
  asm volatile(                                                       \
  "v_mov_b32_e32 v249, 0\n\t"                                         \
  ZH_FAST_ENTRY                                                       \
  ZH_FAST_LAGV                                                        \
  "s_branch .Lzh_byte_%=\n\t"   /* over the alignment padding: up to 63 s_nop, ~150 cycles per entry on average, and a window miss enters anew */ \
  ".p2align 8\n"                                                      \
  ZH_FAST_PAD                                                         \
  ".Lzh_byte_%=:\n\t"                                                 \
  ZH_CM_FAST_BYTE(a, 0, 1, 2, 3, 4, 5, 6, 7, 8, "", "", "", "")       \
  ZH_CM_FAST_BYTE(b, 10, 11, 12, 13, 14, 15, 16, 17, 18, "", "", "", "") \
  "s_branch .Lzh_byte_%=\n"                                           \
  /* ---- out of line ---- */                                         \
  ZH_CM_FAST_COLD(a, 0, 1, 2, 3, 4, 5, 6, 7, 8, "", ZH_FAST_OK_DIRECT)               \
  ZH_CM_FAST_COLD(b, 10, 11, 12, 13, 14, 15, 16, 17, 18, "", ZH_FAST_OK_DIRECT)      \
  ZH_FAST_MISS_BLOCK                                                  \
  ".Lzh_corrupt_%=:\n\t"                                              \
  "s_mov_b32 %[code], 1\n\t"                                          \
  "s_branch .Lzh_end_%=\n"                                            \
  ".Lzh_slow_%=:\n\t"                                                 \
  "s_waitcnt lgkmcnt(0)\n\t"                                          \
  "s_mov_b32 %[code], 0\n"                                            \
  ".Lzh_end_%=:\n\t"                                                  \
  ZH_FAST_T                                                           \
  : [low] "+s"(low_), [high] "+s"(high_), [curr] "+s"(curr_), [k] "+s"(k_), [t] "+s"(t_), [h0] "+s"(h0_), \
    [bdone] "+s"(bdone_), [lu] "+v"(lu_), [vr] "+v"(vr_), [tag] "+v"(tag_), [nm] "+s"(nm_), [thr] "+s"(thr_), [code] "=s"(code_)          \
  : [klim] "s"(klim_), [bfe] "s"(bfe_), [hs] "s"(hs_), [vm] "v"(vm_), [bsa] "v"(bsa_),             \
    [cur] "v"(cur_), [la] "v"(la_), [lb] "v"(lb_), [mqa] "v"(mqa_), [kb] "s"(kb_), [c24] "s"(0x1000000u), [c24m] "s"(0xffffffu)                      \
  : "memory", "scc", "vcc", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90",   \
    "s91", "s92", "s94", "v249", "v250", "v251", "v252", "v245", "s76", "s77", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244")
// Diagnostic build (zh_decode_cm_prof): the same loop with the cycles spent in the spin (wave A waiting for wave B: a window
// being swapped in, or B behind by more than the window's lag allowance) summed into spin_ (s96-s101 are scratch here).
// -DZH_L1_STAGE=n (n = 1..4) turns the diagnostic loop into a STAGE build instead: the cycles between two points of the byte
// (1: byte start -> lag test passed; 2: -> probabilities back from LDS, i.e. address arithmetic, the two reads, the chunk and
// EOS tests and the wait; 3: -> the eight bit steps; 4: -> epilogue and loop branch, up to the next byte's start), two
// s_memtime per byte without a wait of their own (each is read behind a wait the byte has anyway) and four scalar
// instructions to add the difference up: ~28 cycles per byte of distortion, one stage per build so that it stays that small.
#ifndef ZH_L1_STAGE
#define ZH_L1_STAGE 0
#endif
#define ZH_STG_A "s_memtime s[96:97]\n\t"
#define ZH_STG_B "s_memtime s[98:99]\n\t"
#define ZH_STG_ACC "s_sub_u32 s98, s98, s96\n\ts_subb_u32 s99, s99, s97\n\ts_add_u32 s100, s100, s98\n\ts_addc_u32 s101, s101, s99\n\t"
#if ZH_L1_STAGE == 1
#define ZH_STG_H0 ZH_STG_A
#define ZH_STG_H1 ZH_STG_B
#define ZH_STG_H2 ZH_STG_ACC
#define ZH_STG_H3 ""
#elif ZH_L1_STAGE == 2
#define ZH_STG_H0 ""
#define ZH_STG_H1 ZH_STG_A
#define ZH_STG_H2 ZH_STG_B
#define ZH_STG_H3 "s_waitcnt lgkmcnt(0)\n\t" ZH_STG_ACC
#elif ZH_L1_STAGE == 3
#define ZH_STG_H0 ""
#define ZH_STG_H1 ""
#define ZH_STG_H2 ZH_STG_ACC ZH_STG_A
#define ZH_STG_H3 ZH_STG_B
#elif ZH_L1_STAGE == 4
#define ZH_STG_H0 ZH_STG_B
#define ZH_STG_H1 ""
#define ZH_STG_H2 ZH_STG_ACC
#define ZH_STG_H3 ZH_STG_A
#else
#define ZH_STG_H0 ""
#define ZH_STG_H1 ""
#define ZH_STG_H2 ""
#define ZH_STG_H3 ""
#endif
#define ZH_FAST_SPIN_IN "s_memtime s[96:97]\n\t"
#define ZH_FAST_SPIN_OK(S) ".Lzh_okp" #S "_%=:\n\ts_memtime s[98:99]\n\ts_waitcnt lgkmcnt(0)\n\ts_sub_u32 s98, s98, s96\n\ts_subb_u32 s99, s99, s97\n\ts_add_u32 s100, s100, s98\n\ts_addc_u32 s101, s101, s99\n\ts_add_u32 %[nspin], %[nspin], 1\n\ts_branch .Lzh_ok" #S "_%=\n\t"
#if ZH_L1_STAGE
#define ZH_PROF_SPIN_IN ""
#define ZH_PROF_SPIN_OKL ZH_FAST_OK_DIRECT
#define ZH_PROF_SPIN_OKB(S) ""
#else
#define ZH_PROF_SPIN_IN ZH_FAST_SPIN_IN
#define ZH_PROF_SPIN_OKL ZH_FAST_OK_STAMPED
#define ZH_PROF_SPIN_OKB(S) ZH_FAST_SPIN_OK(S)
#endif
#define ZH_CM_FAST_LOOP_PROF(low_, high_, curr_, k_, t_, h0_, bdone_, lu_, code_, klim_, bfe_, hs_, vr_, vm_, bsa_, cur_, tag_, la_, lb_, thr_, nm_, mqa_, kb_, spin_lo_, spin_hi_, nspin_) \
  asm volatile(                                                       \
  "v_mov_b32_e32 v249, 0\n\t"                                         \
  ZH_FAST_ENTRY                                                       \
  ZH_FAST_LAGV                                                        \
  "s_mov_b32 s100, 0\n\t"                                             \
  "s_mov_b32 s101, 0\n\t"                                             \
  "s_memtime s[96:97]\n\t"                                            \
  "s_waitcnt lgkmcnt(0)\n\t"                                          \
  "s_mov_b32 s98, s96\n\t"                                            \
  "s_mov_b32 s99, s97\n\t"                                            \
  "s_branch .Lzh_byte_%=\n\t"                                         \
  ".p2align 8\n"                                                      \
  ZH_FAST_PAD                                                         \
  ".Lzh_byte_%=:\n\t"                                                 \
  ZH_CM_FAST_BYTE(a, 0, 1, 2, 3, 4, 5, 6, 7, 8, ZH_STG_H0, ZH_STG_H1, ZH_STG_H2, ZH_STG_H3)       \
  ZH_CM_FAST_BYTE(b, 10, 11, 12, 13, 14, 15, 16, 17, 18, ZH_STG_H0, ZH_STG_H1, ZH_STG_H2, ZH_STG_H3) \
  "s_branch .Lzh_byte_%=\n"                                           \
  ZH_CM_FAST_COLD(a, 0, 1, 2, 3, 4, 5, 6, 7, 8, ZH_PROF_SPIN_IN, ZH_PROF_SPIN_OKL)              \
  ZH_CM_FAST_COLD(b, 10, 11, 12, 13, 14, 15, 16, 17, 18, ZH_PROF_SPIN_IN, ZH_PROF_SPIN_OKL)     \
  ZH_PROF_SPIN_OKB(a)                                                 \
  ZH_PROF_SPIN_OKB(b)                                                 \
  ZH_FAST_MISS_BLOCK                                                  \
  ".Lzh_corrupt_%=:\n\t"                                              \
  "s_mov_b32 %[code], 1\n\t"                                          \
  "s_branch .Lzh_end_%=\n"                                            \
  ".Lzh_slow_%=:\n\t"                                                 \
  "s_waitcnt lgkmcnt(0)\n\t"                                          \
  "s_mov_b32 %[code], 0\n"                                            \
  ".Lzh_end_%=:\n\t"                                                  \
  ZH_FAST_T                                                           \
  "s_mov_b32 %[splo], s100\n\t"                                       \
  "s_mov_b32 %[sphi], s101\n\t"                                       \
  : [low] "+s"(low_), [high] "+s"(high_), [curr] "+s"(curr_), [k] "+s"(k_), [t] "+s"(t_), [h0] "+s"(h0_), \
    [bdone] "+s"(bdone_), [lu] "+v"(lu_), [vr] "+v"(vr_), [tag] "+v"(tag_), [nm] "+s"(nm_), [thr] "+s"(thr_), [code] "=s"(code_), [splo] "=s"(spin_lo_), [sphi] "=s"(spin_hi_), [nspin] "+s"(nspin_) \
  : [klim] "s"(klim_), [bfe] "s"(bfe_), [hs] "s"(hs_), [vm] "v"(vm_), [bsa] "v"(bsa_),             \
    [cur] "v"(cur_), [la] "v"(la_), [lb] "v"(lb_), [mqa] "v"(mqa_), [kb] "s"(kb_), [c24] "s"(0x1000000u), [c24m] "s"(0xffffffu)                      \
  : "memory", "scc", "vcc", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90",   \
    "s91", "s92", "s94", "s96", "s97", "s98", "s99", "s100", "s101", "v249", "v250", "v251", "v252", "v245", "s76", "s77", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244")
// clang-format on
