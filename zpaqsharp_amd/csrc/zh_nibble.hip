// zh_nibble.hip — the built-in min and mid models (Compressor.cs:48-57) decoded a NIBBLE at a time: the vector side of the
// decoder wave never waits for a decoded bit.
//
// zh_chain2.hip walks a byte bit by bit: predict -> squash -> decode -> y -> update -> select the next node's entries ->
// predict ...  Every vector instruction of bit k+1 sits behind the scalar decoder step of bit k, and what bit k+1 might
// read is fetched "both ways" and selected by y.  Here the 64 lanes are EIGHT GROUPS of NC lanes (NC = components rounded
// up to a power of two: 2 for min, 8 for mid), one group per 3-bit path prefix (b1 b2 b3) of the nibble being decoded
// (Predictor.cs:463-474: hmap4's low nibble is the node 1, 1y, 1yy, 1yyy of the bit-history row).  Group g walks ITS path:
//   level d (1..4): node n_d(g) = 1, 2+b1, 4+2b1+b2, 8+4b1+2b2+b3; predict there (Predictor.cs:245-350), and for d < 3
//   train with the group's own bit b_d (Predictor.cs:353-461) — no select, no hand-over of y to the vector side.
// Groups that share a prefix compute the same thing; after 4 levels group (y1 y2 y3) holds what the decoded path needs.
// The decoder step of level d (Decoder.cs:136-158, scalar unit) reads ONE value: the split factor of group
// (y1 .. y_{d-1}) — a v_readlane with a scalar lane select, as in zh_cm_fast.h.  When the nibble is known the winning
// group trains level 4 with the real bit and COMMITS: its four entries go back to the LDS tables in path order (a later
// node that shares an entry with an earlier one has taken the earlier one's new value from registers: `fwd`), its four new
// bit histories into the row, its mixer weights to HBM.  Everything a nibble reads from LDS or HBM is requested when the
// nibble starts (all four levels' states are bytes of the row held in registers).
//
// What this removes from a bit of zh_chain2.hip (mid: 127 instructions): the both-ways fetch and its selects (25), y's
// hand-over (5), the dependent LDS walk state -> entry at every bit, and every wait of the vector side for the decoder.
// The helper wavefront (nb_helper below: HCOMP for the 32 values the byte can still take once three of its bits are known,
// and the hash rows / mixer row the next byte starts with) is zh_c2_common.h's with twice the candidates.  Results are
// bit-exact with zh_chain2.hip and the oracle (tests/).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zh_core.h"
#include "zh_dev.h"
#include "zh_model.h"
#include "zh_zpaql_native.h"
#include "zh_zpaql_pcomp.h"
#include "zh_ibwt.h"
#include "zh_e8e9.h"

using namespace zhcore;
using namespace zhdev;

#pragma clang diagnostic ignored "-Wint-to-pointer-cast"

#define C2_FINDB 1
#include "zh_c2_common.h"
#include "zh_nb_fast.h"
#include "zh_nb_fast_mid.h"
#ifndef ZH_NB_ASM
#define ZH_NB_ASM 1                           /* 0: nb_fast runs the C++ form of its loop (A/B runs, the *_prof kernels) */
#endif

namespace {

// ---- the models this file decodes.  Shape 1: ICM + ISSE (min); shape 2: ICM, five ISSE, MATCH, MIX (mid).  Table sizes are
// run-time values (ZhComp), so the models LibZPAQ.makeConfig writes for the method strings `ci1` (BWT, level 3) and
// `ci1,1,1,1,2am` (level 4) are the same two shapes with another HCOMP program: zh_framing.cpp gives them the family of
// min / mid with that program's id in ZhModel.kind, and the helper wave runs the program the block's model names.
// Those programs (zh_native_hcomp_m3 / _m4) keep the last bytes in a 64 KiB ring M (hm = 16: `c-- *c=a`) and read
// M[c] / M[c .. c+5] only: the helper wave keeps the ring's last 4 / 8 bytes (index & 3 / & 7, the size of min's / mid's M
// — a slot is overwritten four / eight bytes after it was written, and both arrays start as zeros).  Their first line
// stores C into H[(a + 255) & 511], a word no component reads (hh = 9, components read H[0 .. n-1]): the H view below
// drops stores outside the words it stages.
// The text variant of the level-4 model (`ci1,1,1,1,2awm`) has a word-model ICM as an eighth mixer input: mid's shape with
// lane 7 an ICM and the MIX (component 8) without a lane — the mixer's sum is formed in every lane of a group anyway.  Its
// context H[8] is a word the program (zh_native_hcomp_m4w) never writes: 0.
struct NbMid8 {                                 // 0 icm ; 1-5 isse ; 6 match ; 7 icm ; 8 mix N 0 8 24 255
  static constexpr uint32_t id = 6, n = 9, depth = 5, final_lane = 7, nmix = 1, hh = 3, hm = 3;
  static constexpr uint64_t icm = 0x81, isse = 0x3e;
  static constexpr int helper = 1;
  static constexpr bool smem_ps = true, guard_rows = false, has_tail = false;
  static constexpr int match_lane = 6;
  static constexpr uint32_t mix_lane[2] = {8, 0}, mix_j0[2] = {0, 0}, mix_m[2] = {8, 0};
};
// Level 4's model for barely compressible data (`...,5,0,7,..,1c0,0,511`) is ONE ICM over the LZ77 parse state.  It runs on min's
// loop with both lanes of a group being that ICM: lane 1 has lane 0's constants, reads what lane 0 reads and writes the same
// values to the same places (its entry table is a second copy that stays equal), so the lane the decoder reads (lane 1 of a
// group) holds the ICM's prediction; no lane takes the ISSE half of the update (ZH_NB_FAST_MIN1_LOOP: an empty ISSE mask).
struct NbMin1 {                                 // 0 icm N
  static constexpr uint32_t id = 7, n = 1, depth = 1, final_lane = 1, nmix = 0, hh = 0, hm = 2;
  static constexpr uint64_t icm = 0x3, isse = 0;
  static constexpr int helper = 1;
  static constexpr bool smem_ps = false, guard_rows = true, has_tail = false;
  static constexpr int match_lane = -1;
  static constexpr uint32_t mix_lane[2] = {0, 0}, mix_j0[2] = {0, 0}, mix_m[2] = {0, 0};
};
template <class SP> struct NbT { static constexpr uint32_t shape = SP::id; };
template <> struct NbT<NbMid8> { static constexpr uint32_t shape = 2; };
template <> struct NbT<NbMin1> { static constexpr uint32_t shape = 1; };
// component a lane / unit index stands for (a replica lane stands for the last real component)
template <class SP> __device__ constexpr uint32_t nb_comp_of(uint32_t ci) { return ci < SP::n ? ci : SP::n - 1u; }
template <int NH>
struct NbSpecH {                                // SpecH (zh_c2_common.h) that ignores words >= NH
  lds_u32_p base;
  uint32_t *hs, *wmask;
  struct Ref {
    const NbSpecH *h; uint32_t d;
    __device__ __forceinline__ operator uint32_t() const { return d >= (uint32_t)NH ? 0u : ((*h->wmask >> d) & 1u) ? h->hs[d] : h->base[d]; }
    __device__ __forceinline__ const Ref &operator=(uint32_t x) const {
      if (d < (uint32_t)NH) { h->hs[d] = x; *h->wmask |= 1u << d; }
      return *this;
    }
  };
  __device__ __forceinline__ Ref operator[](uint32_t d) const { return Ref{this, d}; }
};

// The helper wave starts when THREE bits of a byte are known and prepares the next byte for the 32 values it can still take
// (zh_chain2.hip: four bits, 16 values).  The nibble-at-a-time decoder gets through a nibble in about the time a hash row takes
// to arrive from HBM: with the later start it waited ~670 cycles per byte for the helper (profiles/r05/nb_stage_notes.txt).
constexpr uint32_t kNbCand = 32;

template <uint32_t NU>
struct alignas(16) NbLds {
  static constexpr bool kMixLds = false;
  int16_t stretch[32768];                     // at LDS offset 0
  uint16_t squash[4096];
  uint32_t pm01[256];                         // MATCH: stretch(dt2k[len]) | stretch(-dt2k[len]) << 16 (Predictor.cs:273-287), [0] = 0
  uint8_t ns[1024];
  v2u_ ent[NU][256];                          // {A, B}: ISSE {w0, w1} (Predictor.cs:148-152), ICM {cm, stretch(cm >> 8)}
  v4u_ slot[8];                               // per component: the hash row of the current nibble
  v4u_ zrow;                                  // all-zero row read by components without a hash table
  v2u_ lent[8];                               // entry cell of those components
  uint32_t lsink[8];                          // sink for their bit-history writes
  uint32_t slotoff[8];                        // place of slot[c] in the component's hash table
  uint32_t mixb[8];                           // the mixer weights of the second nibble's first row, from the group that fetched them
  // helper wave: what it prepares for the NEXT byte, for each of the 16 values the current byte can still take
  uint32_t hspec[kSpecH][kNbCand];
  v4u_ rowst[NU][3][kNbCand];
  uint32_t mixst[kNbCand][8];
  v4u_ selrow[NU][kNbCand];
  uint32_t seloff[NU][kNbCand];
  uint32_t mb_nib, mb_byte, mb_ready;
  uint32_t mb_cmd, mb_ack, mb_model;
  alignas(16) uint32_t pimm[64];              // operands of a structurally matched PCOMP (zh_zpaql_pcomp.h)
  uint32_t fxs[96];                           // nb_fast: scalars in and out (kFx*)
  uint32_t fxv[64][64];                       // ... and per-lane words (NbV, the coded chunk, the parked output)
  uint32_t fxk[kNmK_count][64];               // zh_nb_fast*.h: per-lane constants of the assembly loop
  uint32_t fxa[kNmS_count + 12][64];          // ... and the per-lane state it loads and stores (+ the stamped variant's 12 sums)
  uint32_t hreg[kHWords];
  uint8_t mreg[kMBytes];
  uint32_t r[256], pr[256];
  uint8_t code[kCodeBytes];
  uint32_t phreg[kPHWords];
  uint8_t pmreg[kPMBytes];
  Vm hz, pz;
  Sink sink;
};

#define NB_STAMP(i)                                                                                  \
  do {                                                                                               \
    if (PROF) {                                                                                      \
      uint64_t now_;                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      prof[i] += now_ - tprev;                                                                       \
      tprev = now_;                                                                                  \
    }                                                                                                \
  } while (0)

__device__ __forceinline__ void nb_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
__device__ __forceinline__ int nb_mul24_sv(int sc, int vec) {
  int r;
  asm("v_mul_i32_i24_e32 %0, %1, %2" : "=v"(r) : "s"(sc), "v"(vec));
  return r;
}
// butterfly sum over each group of 8 lanes: every lane ends with the group's total (quad_perm xor 1, xor 2, row_half_mirror)
__device__ __forceinline__ int sum8(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
  return v;
}

// ---- the helper wave of the nibble kernels: c2_helper (zh_c2_common.h) for 32 candidates.  While the decoder wave finishes a
// byte whose first three bits it has published, this wave runs HCOMP (the translated program, one candidate byte per lane)
// for the 32 values the byte can still take and brings what the next byte starts with into LDS for each of them: h[], the
// row Predictor.find settles on among its three probes (and the probes themselves, for the decoder's patch path), the
// mixer row for c8 = 1.  When the decoder knows the byte it takes column `byte & 31`; this wave commits that candidate's
// machine state.  It never decides anything: a late helper only makes the decoder wait.
template <class SP, class LDS, bool PROF>
__device__ void nb_helper(const ZhLaunch &L, LDS &S, uint32_t lane, uint32_t wg_block_slot) {
  uint64_t hb_busy = 0, hb_slack = 0, hb_t0 = 0, hb_t1 = 0;   // PROF: start seen -> staging complete; staging complete -> byte seen
  constexpr uint32_t NU = c2_units<SP>(), NH = 1u << SP::hh, RN = (NU + 1u) / 2u;
  static_assert(NH <= (uint32_t)kSpecH && NU <= 8u, "staging size");
  uint8_t *slot_mem = L.arena + (uint64_t)wg_block_slot * L.arena_stride;
  const uint32_t cand = lane & (kNbCand - 1u), grp = lane >> 5;
  uint32_t seen_cmd = 0;
  for (;;) {
    uint32_t cmd, sp = 0;
    while ((cmd = c2_ld(&S.mb_cmd)) == seen_cmd) { __builtin_amdgcn_s_sleep(4); if (++sp > kC2Spin) return; }
    seen_cmd = cmd;
    if ((cmd & 3u) == kC2Exit) return;
    if ((cmd & 3u) != kC2New) {
      if (PROF && lane == 0 && L.debug) {
        atomicAdd((unsigned long long *)&L.debug[14], (unsigned long long)hb_busy);
        atomicAdd((unsigned long long *)&L.debug[15], (unsigned long long)hb_slack);
      }
      hb_busy = 0; hb_slack = 0;
      c2_put0(&S.mb_ack, cmd); continue;
    }   // End: acknowledged once this wave has left the block (its last commit is in LDS)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const ZhModel *M = &L.models[uni(c2_ld(&S.mb_model))];
    const uint32_t arena_bytes = uni((uint32_t)M->arena_bytes);
    const uint32_t hprog = uni((M->kind >> 8) & 255u);      // which translated HCOMP program this block's model carries
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(slot_mem, 0, (int)arena_bytes, 0x00020000);
    // this lane's units: u = grp, grp + 2, ... (rows of unit u for candidate `cand`)
    uint32_t u_hto[RN], u_mask[RN], u_comp[RN], u_sb2[RN];
    bool u_on[RN];
#pragma unroll
    for (uint32_t r = 0; r < RN; ++r) {
      const uint32_t u = grp + 2u * r;
      u_on[r] = u < NU;
      uint32_t ci = 0;
#pragma unroll
      for (uint32_t k = 0; k < NU; ++k) if (k == u) ci = nb_comp_of<SP>(c2_unit_comp<SP>(k));
      const ZhComp *cp = &M->comp[ci];
      u_comp[r] = ci; u_hto[r] = (uint32_t)cp->ht_off; u_mask[r] = cp->ht_mask; u_sb2[r] = (uint32_t)cp->arg[0] + 2u;
    }
    uint32_t mx_base = 0, mx_size1 = 0, mx_c1 = 0;
    if (SP::nmix) {
      const ZhComp &mc = M->comp[SP::mix_lane[0]];
      mx_base = uni((uint32_t)mc.cm_off); mx_size1 = uni(mc.cm_mask); mx_c1 = 1u & (uint32_t)mc.arg[4];
    }
    uint32_t hb = 0, hc = 0, hd = 0, hf = 0;              // committed HCOMP registers (A is the input at every run; M and H: S.mreg / S.hreg, zeroed by the decoder)
    uint32_t hr1 = 0, hr2 = 0;                            // ... and R1 / R2 (the only R registers a program of these models uses)
    c2_put0(&S.mb_ack, cmd);
    uint32_t seq = 1;
    bool alive = true;
    while (alive) {
      // ---- the first three bits of byte #seq
      uint32_t v;
      sp = 0;
      while ((((v = c2_ld(&S.mb_nib)) >> 8) ^ seq) & 0xFFFFFFu) {   // (this one is on the clock; 24-bit sequence numbers)
        if (c2_ld(&S.mb_cmd) != seen_cmd || ++sp > kC2Spin) { alive = false; break; }
      }
      if (!alive) break;
      if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hb_t0)::"memory"); }
      const uint32_t x = (v & 7u) << 5 | cand;
      // ---- HCOMP for the candidates (both halves of the wave run it)
      uint32_t sa = x, sb = hb, sc = hc, sd = hd, sf = hf;
      uint32_t wi = 0, wv = 0, wn = 0, hs[NH], wmask = 0;
#pragma unroll
      for (uint32_t d = 0; d < NH; ++d) hs[d] = 0;
      const SpecM sm{(lds_u8_p)lds_off(S.mreg), &wi, &wv, &wn};
      const NbSpecH<NH> sh{(lds_u32_p)lds_off(S.hreg), hs, &wmask};
      constexpr uint32_t mmask_ = (1u << SP::hm) - 1u;
      uint32_t rl[3] = {0u, hr1, hr2};                  // R1 / R2 of the candidate's run (the LZ77 + CM model keeps its parse state there)
      if constexpr (SP::id == 7) {
        if (hprog == ZH_NATIVE_HCOMP_M2SE) (void)zh_native_hcomp_m2se(sa, sb, sc, sd, sf, x, sm, mmask_, sh, 511u, rl, (Sink *)nullptr, L.budget);
        else (void)zh_native_hcomp_m2s(sa, sb, sc, sd, sf, x, sm, mmask_, sh, 511u, rl, (Sink *)nullptr, L.budget);
      } else if constexpr (SP::id == 1) {
        if (hprog == ZH_NATIVE_HCOMP_M3) (void)zh_native_hcomp_m3(sa, sb, sc, sd, sf, x, sm, mmask_, sh, 511u, S.r, (Sink *)nullptr, L.budget);
        else if (hprog == ZH_NATIVE_HCOMP_M2) (void)zh_native_hcomp_m2(sa, sb, sc, sd, sf, x, sm, mmask_, sh, 511u, rl, (Sink *)nullptr, L.budget);
        else if (hprog == ZH_NATIVE_HCOMP_M2E) (void)zh_native_hcomp_m2e(sa, sb, sc, sd, sf, x, sm, mmask_, sh, 511u, rl, (Sink *)nullptr, L.budget);
        else (void)zh_native_hcomp_min(sa, sb, sc, sd, sf, x, sm, mmask_, sh, NH - 1u, S.r, (Sink *)nullptr, L.budget);
      } else if constexpr (SP::id == 6) {
        (void)zh_native_hcomp_m4w(sa, sb, sc, sd, sf, x, sm, mmask_, sh, 511u, S.r, (Sink *)nullptr, L.budget);
      } else {
        if (hprog == ZH_NATIVE_HCOMP_M4) (void)zh_native_hcomp_m4(sa, sb, sc, sd, sf, x, sm, mmask_, sh, 511u, S.r, (Sink *)nullptr, L.budget);
        else (void)zh_native_hcomp_mid(sa, sb, sc, sd, sf, x, sm, mmask_, sh, NH - 1u, S.r, (Sink *)nullptr, L.budget);
      }
      if (grp == 0) {
#pragma unroll
        for (uint32_t d = 0; d < NH; ++d) S.hspec[d][cand] = (uint32_t)sh[d];
      }
      // ---- rows of the first nibble of the next byte (c8 = 1): Predictor.find's three candidates per component
      v4u rr[RN][3];
      uint32_t cxts[RN];
#pragma unroll
      for (uint32_t r = 0; r < RN; ++r) {
        uint32_t hval = 0;
#pragma unroll
        for (uint32_t d = 0; d < NH; ++d) if ((u_comp[r] & (NH - 1u)) == d) hval = (uint32_t)sh[d];
        const uint32_t cxt = hval + 16u;
        cxts[r] = cxt;
        const uint32_t h0 = (cxt * 16u) & (u_mask[r] - 15u);
        const uint32_t vo = u_on[r] ? u_hto[r] + h0 : kOob;
        rr[r][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, 0, 0);
        rr[r][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo ^ 16u, 0, 0);
        rr[r][2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo ^ 32u, 0, 0);
      }
      // ---- mixer row for c8 = 1: weights grp, grp + 2, grp + 4, grp + 6 of the row of candidate `cand`
      uint32_t mwv[4] = {0, 0, 0, 0};
      if (SP::nmix) {
        uint32_t hq = 0;
#pragma unroll
        for (uint32_t d = 0; d < NH; ++d) if (SP::mix_lane[0] == d) hq = (uint32_t)sh[d];      // (a MIX beyond the staged words: its context is 0, see NbMid8)
        const uint32_t row = mx_base + ((hq + mx_c1) & mx_size1) * (SP::mix_m[0] * 4u);
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) {
          const uint32_t jj = grp + 2u * t;
          mwv[t] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, jj < SP::mix_m[0] ? row + jj * 4u : kOob, 0, 0);
        }
      }
#pragma unroll
      for (uint32_t r = 0; r < RN; ++r) {
        if (u_on[r]) {
          const uint32_t u = grp + 2u * r;
#pragma unroll
          for (uint32_t k = 0; k < 3; ++k) *(lds_u4_p)lds_off(&S.rowst[u][k][cand]) = rr[r][k];
          // Predictor.find (Predictor.cs:550-567) on the three probes, here instead of at the decoder wave's byte boundary:
          // check compare, then the lowest-priority row as the victim (ties as the reference breaks them)
          const uint32_t chk = (cxts[r] >> u_sb2[r]) & 255u;
          const uint32_t h0 = (cxts[r] * 16u) & (u_mask[r] - 15u);
          const v4u &r0 = rr[r][0], &r1 = rr[r][1], &r2 = rr[r][2];
          const bool m0 = (r0.x & 255u) == chk, m1 = (r1.x & 255u) == chk, m2 = (r2.x & 255u) == chk;
          const uint32_t p0 = (r0.x >> 8) & 255u, p1 = (r1.x >> 8) & 255u, p2 = (r2.x >> 8) & 255u;
          const uint32_t victim = (p0 <= p1 && p0 <= p2) ? h0 : p1 < p2 ? (h0 ^ 16u) : (h0 ^ 32u);
          const uint32_t sel = m0 ? h0 : m1 ? (h0 ^ 16u) : m2 ? (h0 ^ 32u) : victim;
          const v4u fresh = {chk, 0, 0, 0};
          const v4u row = m0 ? r0 : m1 ? r1 : m2 ? r2 : fresh;
          *(lds_u4_p)lds_off(&S.selrow[u][cand]) = row;
          S.seloff[u][cand] = sel;
        }
      }
      if (SP::nmix) {
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) S.mixst[cand][grp + 2u * t] = mwv[t];
      }
      asm volatile("" ::: "memory");
      c2_put0(&S.mb_ready, seq);
      if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hb_t1)::"memory"); hb_busy += hb_t1 - hb_t0; }
      // ---- the byte: commit its candidate
      sp = 0;
      while ((((v = c2_ld(&S.mb_byte)) >> 8) ^ seq) & 0xFFFFFFu) {
        if (c2_ld(&S.mb_cmd) != seen_cmd || ++sp > kC2Spin) { alive = false; break; }
        __builtin_amdgcn_s_sleep(1);                     // (nothing to do until the byte is known: poll gently)
      }
      if (!alive) break;
      if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hb_t0)::"memory"); hb_slack += hb_t0 - hb_t1; }
      const uint32_t lo = v & (kNbCand - 1u);
      hb = rdlane(sb, lo); hc = rdlane(sc, lo); hd = rdlane(sd, lo); hf = rdlane(sf, lo);
      hr1 = rdlane(rl[1], lo); hr2 = rdlane(rl[2], lo);
      const uint32_t cwi = rdlane(wi, lo), cwv = rdlane(wv, lo), cwn = rdlane(wn, lo);
      if (cwn && lane == 0) S.mreg[cwi] = (uint8_t)cwv;
      if (lane < NH) { const uint32_t hv_ = S.hspec[lane][lo]; S.hreg[lane] = hv_; }
      ++seq;
    }
  }
}

// ---- per-lane constants: the component this lane stands for and the path its group walks ----------------------------
template <class SP>
struct NbK {
  static constexpr uint32_t NC = NbT<SP>::shape == 1 ? 2u : 8u;          // lanes of a group
  static constexpr uint32_t NG = 8u;                               // groups: the 3-bit prefixes of a nibble's path
  static constexpr uint64_t kII = SP::icm | SP::isse;
  static constexpr uint32_t mx_m4 = SP::mix_m[0] * 4u;
  uint32_t lane, ci, g, b1, b2, b3;
  bool act, canon, l_isse, l_ii, l_match, l_feed;
  uint32_t sh2, sh3, sh4, node[5], pre[5], ybit[4];
  uint32_t hto, ht_mask, cmo, cm_mask, sizebits2, tab, wrow, wrow_mask, unit, un_;
  int isse_m;
  uint32_t cshift;
  int pself, mx_rate;
  uint32_t vo_mix, mx_base, mx_size1;
  __amdgpu_buffer_rsrc_t rsrc;
  uint8_t *slot_mem;
};
// ---- per-lane state carried from nibble to nibble and from byte to byte (the same in every group) ---------------------
struct NbV {
  uint32_t hv;                                        // h[component] (Predictor.cs:469)
  uint32_t rowoff;                                    // place of the hash row of the current nibble (in S.slot[ci] and below)
  uint32_t row_x, row_q1, row_q2, row_q3;             // the row as it was when the nibble began (zero for components without a table)
  uint32_t rowvalid;
  int mwl[5];                                         // mixer: this lane's weight in the row of level d of the current nibble
  uint32_t mrowl[5];                                  // ... and its buffer offset
  uint32_t mx_rb;                                     // this lane's buffer offset in row 0 of the byte's block of mixer rows
  uint32_t m_len, m_ptr, m_limit, m_byte;             // MATCH (Predictor.cs:273-287, 382-411): the Component fields
  int pm0, pm1;                                       // stretch of -+dt2k[len] for this byte; 0 once the match has failed
  uint32_t cm_pre, va_pre, vb_pre, mbn_pre, mbc_pre;  // see zh_chain2.hip (match_prefetch)
  v4u oldb; uint32_t oldb_off, oldb_valid;            // the row written back at the last byte boundary
  v4u old1; uint32_t old1_off, old1_valid;            // the first nibble's row as it was evicted
  int w1_new;                                         // mixer: row c8 = 1 of the byte's block after the first nibble
};
constexpr int kNbVWords = sizeof(NbV) / 4;
static_assert(sizeof(NbV) % 4 == 0 && kNbVWords + 2 <= 64, "nb_fast exchange area");
struct NbProf { uint64_t prof[16]; uint64_t tprev; };

#undef NB_STAMP
#define NB_STAMP(i)                                                                                  \
  do {                                                                                               \
    if (PROF) {                                                                                      \
      uint64_t now_;                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      P.prof[i] += now_ - P.tprev;                                                                   \
      P.tprev = now_;                                                                                \
    }                                                                                                \
  } while (0)

template <class SP, class LDS>
__device__ __forceinline__ void nb_setup(NbK<SP> &K, LDS &S, const ZhModel *M, uint8_t *slot_mem, uint32_t lane) {
  constexpr uint32_t NC = NbK<SP>::NC, NG = NbK<SP>::NG;
  K.lane = lane; K.ci = lane & (NC - 1u); K.g = (lane / NC) & (NG - 1u);
  K.act = lane < NC * NG; K.canon = lane < NC;            // min: lanes 16-63 repeat lanes 0-15 and never write
  K.b1 = (K.g >> 2) & 1u; K.b2 = (K.g >> 1) & 1u; K.b3 = K.g & 1u;
  K.l_isse = (SP::isse >> K.ci) & 1; K.l_ii = (NbK<SP>::kII >> K.ci) & 1;
  K.l_match = SP::match_lane >= 0 && K.ci == (uint32_t)SP::match_lane;
  K.sh2 = 16u + 8u * K.b1;                               // node 2 + b1: byte 2 / 3 of row dword 0
  K.sh3 = 8u * (2u * K.b1 + K.b2);                       // node 4 + 2 b1 + b2: a byte of dword 1
  K.sh4 = 8u * (2u * K.b2 + K.b3);                       // node 8 + 4 b1 + 2 b2 + b3: a byte of dword 2 (b1 = 0) / 3
  K.node[0] = 0; K.node[1] = 1u; K.node[2] = 2u + K.b1; K.node[3] = 4u + 2u * K.b1 + K.b2; K.node[4] = 8u + 4u * K.b1 + 2u * K.b2 + K.b3;
  K.pre[0] = 0; K.pre[1] = 0; K.pre[2] = K.b1; K.pre[3] = 2u * K.b1 + K.b2; K.pre[4] = 4u * K.b1 + 2u * K.b2 + K.b3;   // c8 of level d = (c8 of the nibble << (d-1)) + pre[d]
  K.ybit[0] = 0; K.ybit[1] = K.b1; K.ybit[2] = K.b2; K.ybit[3] = K.b3;     // the bit this group assumes at level d (d = 1..3)
  K.unit = (uint32_t)__builtin_popcountll(NbK<SP>::kII & ((1ull << K.ci) - 1));
  K.un_ = K.unit < c2_units<SP>() ? K.unit : 0u;
  const ZhComp *mycp = &M->comp[nb_comp_of<SP>(K.ci)];
  K.hto = K.l_ii || K.l_match ? (uint32_t)mycp->ht_off : 0u; K.ht_mask = mycp->ht_mask;
  K.cmo = (uint32_t)mycp->cm_off; K.cm_mask = mycp->cm_mask;
  K.sizebits2 = (uint32_t)mycp->arg[0] + 2;
  K.tab = K.l_ii ? lds_off(&S.ent[K.unit][0]) : lds_off(&S.lent[K.ci]);      // entry table of this lane's component
  K.wrow = K.l_ii ? lds_off(&S.slot[K.ci]) : lds_off(&S.lsink[K.ci]);        // where its bit histories are written (+ node)
  K.wrow_mask = K.l_ii ? 15u : 0u;
  K.isse_m = K.l_isse ? -1 : 0;
  K.cshift = K.l_isse ? 6u : 16u;
  K.pself = 0;                                           // prediction of a lane that is neither ICM nor ISSE (MATCH, CONST)
  if (K.ci < SP::n && mycp->type == ZH_CONS) K.pself = ((int)mycp->arg[0] - 128) * 4;
  K.vo_mix = kOob; K.mx_base = 0; K.mx_size1 = 0; K.mx_rate = 0;
  if (SP::nmix) {                                        // the mixer kept in HBM: lane (g, k) owns weight k of the rows its group reads
    const ZhComp &mc = M->comp[SP::mix_lane[0]];
    K.mx_base = uni((uint32_t)mc.cm_off);
    K.mx_size1 = uni(mc.cm_mask);
    K.mx_rate = (int)uni((uint32_t)mc.arg[3]);
    asm volatile("" : "+v"(K.mx_rate));
    if (K.ci >= SP::mix_j0[0] && K.ci < SP::mix_j0[0] + SP::mix_m[0]) K.vo_mix = (K.ci - SP::mix_j0[0]) * 4u;
  }
  K.l_feed = K.vo_mix != kOob;
  K.slot_mem = slot_mem;
  K.rsrc = __builtin_amdgcn_make_buffer_rsrc(slot_mem, 0, (int)uni((uint32_t)M->arena_bytes), 0x00020000);
}

// Hash rows of a nibble (c8 == 1 or 16 <= c8 < 32), Predictor.find (Predictor.cs:550-567): see zh_chain2.hip
struct NbProbe { v4u r0, r1, r2; uint32_t h0, chk; };
template <class SP>
__device__ __forceinline__ void nb_rows_issue(const NbK<SP> &K, const NbV &V, uint32_t c8, NbProbe &pr, bool on) {
  const uint32_t cxt = V.hv + 16u * c8;
  pr.chk = (cxt >> K.sizebits2) & 255;
  pr.h0 = (cxt * 16u) & (K.ht_mask - 15u);
  const uint32_t vo = (K.l_ii && on) ? K.hto + pr.h0 : kOob;
  pr.r0 = __builtin_amdgcn_raw_buffer_load_b128(K.rsrc, vo, 0, 0);
  pr.r1 = __builtin_amdgcn_raw_buffer_load_b128(K.rsrc, vo ^ 16u, 0, 0);
  pr.r2 = __builtin_amdgcn_raw_buffer_load_b128(K.rsrc, vo ^ 32u, 0, 0);
}
// find on the three probes; rows this wave evicted after the probes' loads may have been issued (olda, old) are taken
// from the copies.  Result: the row and its place (nothing is written).
__device__ __forceinline__ void nb_rows_pick(const NbProbe &pr, const v4u &olda, uint32_t olda_off, bool olda_valid, const v4u &old, uint32_t old_off,
                                             bool old_valid, v4u &row, uint32_t &sel) {
  const uint32_t h0 = pr.h0, h1 = h0 ^ 16u, h2 = h0 ^ 32u;
  v4u r0 = pr.r0, r1 = pr.r1, r2 = pr.r2;
  if (olda_valid && olda_off == h0) r0 = olda;
  if (olda_valid && olda_off == h1) r1 = olda;
  if (olda_valid && olda_off == h2) r2 = olda;
  if (old_valid && old_off == h0) r0 = old;
  if (old_valid && old_off == h1) r1 = old;
  if (old_valid && old_off == h2) r2 = old;
  const uint32_t chk = pr.chk;
  const bool m0 = (r0.x & 255) == chk, m1 = (r1.x & 255) == chk, m2 = (r2.x & 255) == chk;
  const uint32_t p0 = (r0.x >> 8) & 255, p1 = (r1.x >> 8) & 255, p2 = (r2.x >> 8) & 255;
  const uint32_t victim = (p0 <= p1 && p0 <= p2) ? h0 : p1 < p2 ? h1 : h2;
  sel = m0 ? h0 : m1 ? h1 : m2 ? h2 : victim;
  const v4u fresh = {chk, 0, 0, 0};
  row = m0 ? r0 : m1 ? r1 : m2 ? r2 : fresh;
}
template <class SP>
__device__ __forceinline__ void nb_row_take(const NbK<SP> &K, NbV &V, const v4u &row, uint32_t sel) {     // every lane holds the row of its component
  V.rowoff = sel; V.rowvalid = 1u;
  V.row_x = K.l_ii ? row.x : 0u; V.row_q1 = K.l_ii ? row.y : 0u; V.row_q2 = K.l_ii ? row.z : 0u; V.row_q3 = K.l_ii ? row.w : 0u;
}
// the row of the finished nibble: its content for the caller, and (row_store) its write-back, fire and forget.  The store is
// issued BEHIND whatever loads the caller still has to take: vector memory completes in issue order (one vmcnt for loads and
// stores), so a wait for anything issued after a store also waits for that store's acknowledgement
template <class SP, class LDS>
__device__ __forceinline__ void nb_row_old(const NbK<SP> &K, const NbV &V, LDS &S, v4u &old, uint32_t &old_off, uint32_t &old_valid) {
  old = *(lds_u4_p)lds_off(&S.slot[K.ci]);
  old_off = V.rowoff; old_valid = (V.rowvalid && K.l_ii) ? 1u : 0u;
}
template <class SP>
__device__ __forceinline__ void nb_row_store(const NbK<SP> &K, const v4u &old, uint32_t old_off, uint32_t old_valid) {
  __builtin_amdgcn_raw_buffer_store_b128(old, K.rsrc, (old_valid && K.canon) ? K.hto + old_off : kOob, 0, 0);
}
template <class SP>
__device__ __forceinline__ void nb_mix_set(const NbK<SP> &K, NbV &V, uint32_t hq) {
  V.mx_rb = K.vo_mix + (K.mx_base + __umul24(uni(hq) & K.mx_size1 & ~255u, NbK<SP>::mx_m4));
}
// rows of levels 2..4 of a nibble whose first row is c8n (1, or 16 + first nibble); level 1 comes staged / fetched ahead
template <class SP>
__device__ __forceinline__ void nb_mix_rows(const NbK<SP> &K, NbV &V, uint32_t c8n) {
#pragma unroll
  for (int dd = 1; dd <= 4; ++dd) V.mrowl[dd] = V.mx_rb + __umul24((c8n << (dd - 1)) + K.pre[dd], NbK<SP>::mx_m4);
#pragma unroll
  for (int dd = 2; dd <= 4; ++dd) V.mwl[dd] = (int)__builtin_amdgcn_raw_buffer_load_b32(K.rsrc, V.mrowl[dd], 0, 0);
}
template <class SP>
__device__ __forceinline__ void nb_match_prefetch(const NbK<SP> &K, NbV &V) {
  const uint32_t ml = (uint32_t)(SP::match_lane >= 0 ? SP::match_lane : 0);
  const uint32_t msk = rdlane(K.ht_mask, ml), base = rdlane(K.hto, ml);
  const uint32_t lim = (rdlane(V.m_limit, ml) + 1u) & msk;                 // m_limit once this byte is stored
  const uint32_t off = lim - rdlane(V.cm_pre, ml);                        // the candidate's distance, should the byte end unmatched
  V.va_pre = __builtin_amdgcn_raw_buffer_load_b8(K.rsrc, base + ((lim - K.lane - 1u) & msk), 0, 0);
  V.vb_pre = __builtin_amdgcn_raw_buffer_load_b8(K.rsrc, base + ((lim - K.lane - off - 1u) & msk), 0, 0);
  V.mbn_pre = __builtin_amdgcn_raw_buffer_load_b8(K.rsrc, K.l_match ? base + ((lim - off) & msk) : kOob, 0, 0);
  V.mbc_pre = __builtin_amdgcn_raw_buffer_load_b8(K.rsrc, K.l_match ? base + ((lim - V.m_ptr) & msk) : kOob, 0, 0);
}
// Predictor.update's MATCH part at the byte boundary (Predictor.cs:391-410); c is already in the history and the
// hash index, m_limit advanced
template <class SP, class LDS>
__device__ __forceinline__ void nb_match_boundary(const NbK<SP> &K, NbV &V, LDS &S, uint32_t cb) {
  const uint32_t lane = K.lane;
  const bool zero = V.m_len == 0;
  const uint32_t nptr = V.m_limit - V.cm_pre;
  const bool need = K.l_match && zero && (nptr & K.ht_mask) != 0;
  V.m_ptr = (K.l_match && zero) ? nptr : V.m_ptr;
  V.m_len = (K.l_match && !zero && V.m_len < 255) ? V.m_len + 1 : V.m_len;
  if (__ballot(need) != 0) {                         // verify the candidate with the whole wave (Predictor.cs:403-405)
    const uint32_t ml = (uint32_t)SP::match_lane;
    const uint32_t lim = rdlane(V.m_limit, ml), off = rdlane(V.m_ptr, ml), msk = rdlane(K.ht_mask, ml);
    const uint32_t a = lane == 0 ? cb : (V.va_pre & 255u);
    const uint32_t b = ((lane + off) & msk) == 0 ? cb : (V.vb_pre & 255u);
    uint64_t mism = __ballot(a != b);
    uint32_t len = 64;
    if (LIKELY(mism != 0)) len = (uint32_t)__builtin_ctzll(mism);
    else {
      const uint8_t *hp = K.slot_mem + rdlane(K.hto, ml);
      for (uint32_t base = 64; base < 256; base += 64) {
        const uint32_t t = base + lane;
        const bool eq = t < 255 && hp[(lim - t - 1) & msk] == hp[(lim - t - off - 1) & msk];
        mism = __ballot(!eq);
        if (mism) { len += (uint32_t)__builtin_ctzll(mism); break; }
        len += 64;
      }
    }
    const uint32_t nl = len > 255 ? 255 : len;
    V.m_len = K.l_match ? nl : V.m_len;
    V.m_byte = K.l_match ? (((off - 1u) & msk) == 0 ? cb : (V.mbn_pre & 255u)) : V.m_byte;
  } else {
    const uint32_t cont = ((V.m_ptr - 1u) & K.ht_mask) == 0 ? cb : (V.mbc_pre & 255u);
    V.m_byte = (K.l_match && V.m_len) ? cont : V.m_byte;
  }
  const uint32_t pw = *(lds_u32_p)(lds_off(S.pm01) + V.m_len * 4u);      // m_len stays 0 in the other lanes
  V.pm0 = (int)(int16_t)(pw & 0xffffu); V.pm1 = (int)pw >> 16;
}

// Renormalisation when the coded bytes it can need are in the register-held chunk (the fast loop checks that before a byte):
// Decoder.cs:148-156, scalar throughout.
__device__ __forceinline__ void nb_renorm_fast(Dec &d, uint32_t cur, uint32_t &k) {
  uint32_t low = d.low, high = d.high, curr = d.curr, kk = k;
  do {
    high = high << 8 | 255;
    low = low << 8;
    low = low ? low : 1u;
    const uint32_t c = (rdlane(cur, kk >> 2) >> ((kk & 3) * 8)) & 255u;
    ++kk;
    curr = curr << 8 | c;
  } while ((high ^ low) < 0x1000000u);
  d.low = uni(low); d.high = uni(high); d.curr = uni(curr); k = uni(kk);
}

// ---- the eight bits of a byte (Decoder.cs:48-55 around Predictor.predict / update); j, bad, err as in zh_chain2.hip.
// FAST: renormalisations take their bytes from the chunk in registers without the refill test (the caller made sure)
template <class SP, bool PROF, bool FAST, class LDS>
__device__ __forceinline__ uint32_t nb_decode_byte(const NbK<SP> &K, NbV &V, LDS &S, Dec &d, InBuf &in, uint32_t &j, uint32_t &bad, uint32_t &err,
                                                   uint32_t bseq, NbProf &P) {
  constexpr uint32_t NC = NbK<SP>::NC;
  constexpr uint32_t mx_m4 = NbK<SP>::mx_m4;
  const uint32_t lane = K.lane, ci = K.ci, g = K.g;
  const lds_i16_p lds_stretch = (lds_i16_p)lds_off(S.stretch);
  const lds_u16_p lds_squash = (lds_u16_p)lds_off(S.squash);
  const uint32_t ns_off = lds_off(S.ns);
  NB_STAMP(10);
  int rnd12 = 1 << 12;                           // the weight updates' rounding addend, in a VGPR (zh_chain2.hip, C2V 64)
  asm volatile("" : "+v"(rnd12));
  // ---- the second nibble's hash rows, for the two values of the first nibble this group's prefix leaves open
  // (16 candidates over the 8 groups), and the mixer weights of the rows they start with
  NbProbe cand[2];
  int cmw[2] = {0, 0};
#pragma unroll
  for (uint32_t k = 0; k < 2; ++k) {
    nb_rows_issue(K, V, 16u + 2u * g + k, cand[k], K.act);
    if (SP::nmix) cmw[k] = (int)__builtin_amdgcn_raw_buffer_load_b32(K.rsrc, K.act ? V.mx_rb + __umul24(16u + 2u * g + k, mx_m4) : kOob, 0, 0);
  }
  uint32_t cbyte = 0;
  int p_l1 = 0, sqm_l1 = 0, mw_l1 = 0;            // mixer: level 1 of the first nibble (the same in every group)
#pragma unroll
  for (int nib = 0; nib < 2; ++nib) {
    // ================= one nibble: four levels, every group on its own path =================
    uint32_t st[5], ea[5], nsp[5], nA[5], nB[5], nsb[5];
    v2u et[5];
    int nmw[5] = {0, 0, 0, 0, 0};
    st[1] = __builtin_amdgcn_ubfe(V.row_x, 8u, 8u);
    st[2] = __builtin_amdgcn_ubfe(V.row_x, K.sh2, 8u);
    st[3] = __builtin_amdgcn_ubfe(V.row_q1, K.sh3, 8u);
    st[4] = __builtin_amdgcn_ubfe(K.b1 ? V.row_q3 : V.row_q2, K.sh4, 8u);
#pragma unroll
    for (int dd = 1; dd <= 4; ++dd) {
      ea[dd] = K.tab + st[dd] * 8u;
      et[dd] = *(lds_u2_p)ea[dd];
      nsp[dd] = *(lds_u16_p)(ns_off + st[dd] * 4u);      // next(state, 0) | next(state, 1) << 8
    }
    // MATCH: the nibble the match predicts; a group whose path left it predicts 0 from there on
    uint32_t ex = 0, mism = 0;
    if (SP::match_lane >= 0) {
      ex = (V.m_byte >> (nib ? 0u : 4u)) & 15u;
      mism = (g ^ (ex >> 1)) & 7u;
    }
    uint32_t nv = 0;                               // the nibble's bits decoded so far (scalar)
    int p = 0, sqp = 0, sqm = 0, pj = 0;
    uint32_t eA = 0, eB = 0;
#pragma unroll
    for (int dd = 1; dd <= 4; ++dd) {
      // ---- the entry of this level: from the table, or from a level of this path that trained the same entry
      eA = et[dd].x; eB = et[dd].y;
#pragma unroll
      for (int k = 1; k < dd; ++k) {
        const bool same = ea[dd] == ea[k];
        eA = same ? nA[k] : eA;
        eB = same ? nB[k] : eB;
      }
      // ---- predict (Predictor.cs:259-343)
      int xs = K.pself;
      if (SP::match_lane >= 0) {
        const uint32_t cbit = (ex >> (4 - dd)) & 1u;
        int pmv = cbit ? V.pm1 : V.pm0;
        const uint32_t left = dd == 1 ? 0u : dd == 2 ? (mism & 4u) : dd == 3 ? (mism & 6u) : mism;
        pmv = left ? 0 : pmv;
        xs = K.l_match ? pmv : K.pself;
      }
      const int x = K.l_ii ? (int)eB : xs;
      const int cw0 = (int)eA & K.isse_m;
      const int cw1m = (int)((uint32_t)x << K.cshift);
      p = x;
#pragma unroll
      for (uint32_t t = 0; t < SP::depth; ++t) p = med3i((__mul24(shr1(p), cw0) + cw1m) >> 16, -2048, 2047);
      uint32_t psv;
      if (SP::nmix) {
        const int term = sum8(__mul24(V.mwl[dd] >> 8, p));     // lanes that do not feed the mixer hold weight 0
        const int pmx = med3i(term >> 8, -2048, 2047);
        if (ci == SP::mix_lane[0]) p = pmx;
        sqm = (int)*(lds_u16_p)((uint32_t)(uintptr_t)lds_squash + (uint32_t)(pmx + 2048) * 2u);
      }
      sqp = (int)*(lds_u16_p)((uint32_t)(uintptr_t)lds_squash + (uint32_t)(p + 2048) * 2u);
      // ---- the part of the update that needs neither the squashed prediction nor the decoded bit: the ICM's entry
      // (levels 1-3: the group's own bit; Predictor.cs:375-381) — its stretch look-up travels under the squash look-up
      const uint32_t yl_pre = dd < 4 ? K.ybit[dd] : 0u;
      const int ey_pre = yl_pre ? 32767 : 0;
      uint32_t ncm = eA + (uint32_t)((int)(ey_pre - (int)(eA >> 8)) >> 2);
      int npst = 0;
      if (dd < 4) npst = *(lds_i16_p)((uint32_t)(uintptr_t)lds_stretch + ((ncm >> 7) & 0x1fffeu));
      psv = ((uint32_t)(SP::nmix ? sqm : sqp) << 17) | 0x10000u;
      if (SP::nmix && nib == 0 && dd == 1) { p_l1 = p; sqm_l1 = sqm; mw_l1 = V.mwl[1]; }
      asm("" : "+v"(psv));
      pj = shr1(p);                              // ISSE update: the prediction of the component before
      // ---- decode (Decoder.cs:136-158): the split factor of the group the decoded bits lead to
      const uint32_t lsel = (nv << (4 - dd)) * NC + SP::final_lane;
      const uint32_t ps = rdlane(psv, lsel);
      uint32_t jb = uni(j), xr;
      d.low = uni(d.low); d.high = uni(d.high); d.curr = uni(d.curr);   // (already scalar: says so to the compiler)
      ZH_DEC_STEP_LITE(d, ps, jb, xr);
      j = jb;
      if (UNLIKELY(xr < 0x1000000u)) {
        if (FAST) {
          nb_renorm_fast(d, in.cur, in.k);
          const uint32_t oor = (uint32_t)(d.curr < d.low) | (uint32_t)(d.curr > d.high);
          if (!(nib == 1 && dd == 4)) bad = uni(bad | oor);    // after the byte's last bit the next EOS step re-checks by itself
        } else {
          const uint32_t was = bad;
          uint32_t later = 0;
          if (dec_renorm_chk(d, in, lane, (nib == 1 && dd == 4) ? later : bad) && !err) err = was ? (uint32_t)-ZH_E_CORRUPT : (uint32_t)-ZH_E_EOF;
        }
      }
      const uint32_t y = uni(j & 1u);
      nv = nv * 2u + y;
      // ---- update (Predictor.cs:363-461): levels 1-3 with the group's own bit, level 4 with the decoded one
      const uint32_t yl = dd < 4 ? K.ybit[dd] : y;
      const int ey = yl ? 32767 : 0;
      const int e = ey - sqp;
      nsb[dd] = __builtin_amdgcn_ubfe(nsp[dd], yl * 8u, 8u);
      if (dd == 4) {
        ncm = eA + (uint32_t)((int)(ey - (int)(eA >> 8)) >> 2);
        npst = *(lds_i16_p)((uint32_t)(uintptr_t)lds_stretch + ((ncm >> 7) & 0x1fffeu));
      }
      const int nw0 = med3i((int)eA + ((__mul24(e, pj) + rnd12) >> 13), -(1 << 19), (1 << 19) - 1);
      const int nw1 = med3i((int)eB + ((e + 16) >> 5), -(1 << 19), (1 << 19) - 1);
      nA[dd] = K.l_isse ? (uint32_t)nw0 : ncm;
      nB[dd] = K.l_isse ? (uint32_t)nw1 : (uint32_t)npst;
      if (SP::nmix) {                            // MIX (Predictor.cs:427-439)
        const int eq = __mul24(ey - sqm, K.mx_rate) >> 4;
        nmw[dd] = med3i(V.mwl[dd] + ((__mul24(eq, p) + rnd12) >> 13), -(1 << 19), (1 << 19) - 1);
      }
      if (nib == 0 && dd == 3) {
        // Three bits of the byte are known: the helper wave starts on the next byte's 32 candidates.  What its loads must see of
        // this wave's stores (the rows written back at the last byte boundary, the mixer rows of the byte before) has reached
        // memory: vector memory completes in issue order, and everything this wave has issued so far is waited for here (the
        // youngest are the second nibble's candidate rows, requested a thousand cycles ago).  The rows written back later — at
        // the nibble switch and at the byte boundary — are patched in from the copies kept here when the staging is taken.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        c2_put0(&S.mb_nib, bseq << 8 | (nv & 7u));
      }
    }
    NB_STAMP(nib);
    // ---- commit: the group the nibble's first three bits name
    const uint32_t gw = nv >> 1;
    if (K.act && g == gw) {
#pragma unroll
      for (int dd = 1; dd <= 4; ++dd) *(lds_u2_p)ea[dd] = v2u{nA[dd], nB[dd]};
#pragma unroll
      for (int dd = 1; dd <= 4; ++dd) *(lds_u8_p)(K.wrow + (K.node[dd] & K.wrow_mask)) = (uint8_t)nsb[dd];
    }
    if (SP::nmix) {
#pragma unroll
      for (int dd = 1; dd <= 4; ++dd) __builtin_amdgcn_raw_buffer_store_b32((uint32_t)nmw[dd], K.rsrc, (K.act && g == gw) ? V.mrowl[dd] : kOob, 0, 0);
    }
    if (SP::nmix && nib == 0) {
      // row c8 = 1 of this byte's block as it is now, in every group (level 1 is the same everywhere; y1 is known): should
      // the next byte have the same mixer context, the helper's copy of that row may predate the store above
      const int eq1 = __mul24(((nv & 8u) ? 32767 : 0) - sqm_l1, K.mx_rate) >> 4;
      V.w1_new = med3i(mw_l1 + ((__mul24(eq1, p_l1) + rnd12) >> 13), -(1 << 19), (1 << 19) - 1);
    }
    if (SP::match_lane >= 0) {                    // MATCH (Predictor.cs:383-384): a miss ends the match
      const bool miss = nv != ex;
      V.m_len = miss ? 0u : V.m_len; V.pm0 = miss ? 0 : V.pm0; V.pm1 = miss ? 0 : V.pm1;
    }
    NB_STAMP(2 + nib);
    if (nib == 0) {
      // ---- second nibble (Predictor.cs:267-270: c8 & 0xf0 == 16): its rows were requested when the byte began
      cbyte = nv;
      v4u old; uint32_t old_off, old_valid;
      nb_row_old(K, V, S, old, old_off, old_valid);
      V.old1 = old; V.old1_off = old_off; V.old1_valid = old_valid;
      const bool k1 = (nv & 1u) != 0;
      NbProbe pr;
      pr.h0 = k1 ? cand[1].h0 : cand[0].h0; pr.chk = k1 ? cand[1].chk : cand[0].chk;
      pr.r0 = k1 ? cand[1].r0 : cand[0].r0; pr.r1 = k1 ? cand[1].r1 : cand[0].r1; pr.r2 = k1 ? cand[1].r2 : cand[0].r2;
      v4u row; uint32_t sel;
      nb_rows_pick(pr, V.oldb, V.oldb_off, V.oldb_valid != 0, old, old_off, old_valid != 0, row, sel);   // (oldb: written back just before the candidates were requested)
      asm volatile("" :: "v"(row.x), "v"(row.y), "v"(row.z), "v"(row.w), "v"(sel));       // (the candidates' loads are consumed: now the store)
      nb_row_store(K, old, old_off, old_valid);
      if (K.act && g == gw) {
        *(lds_u4_p)lds_off(&S.slot[ci]) = row;
        S.slotoff[ci] = sel;
        if (SP::nmix) S.mixb[ci] = (uint32_t)(k1 ? cmw[1] : cmw[0]);
      }
      asm volatile("" ::: "memory");
      {
        const v4u rowb = *(lds_u4_p)lds_off(&S.slot[ci]);
        const uint32_t selb = S.slotoff[ci];
        nb_row_take(K, V, rowb, selb);
        if (SP::nmix) { V.mwl[1] = K.l_feed ? (int)S.mixb[ci] : 0; nb_mix_rows(K, V, 16u + nv); }
      }
      if (SP::match_lane >= 0) nb_match_prefetch(K, V);
      NB_STAMP(4);
    } else cbyte = cbyte * 16u + nv;
  }
  return cbyte;
}

// ---- byte boundary: MATCH (Predictor.cs:391-410), h[] and the rows of the next byte from the helper wave.
// false: the helper wavefront stopped answering (cannot happen by design)
template <class SP, bool PROF, class LDS>
__device__ __forceinline__ bool nb_boundary(const NbK<SP> &K, NbV &V, LDS &S, int c, uint32_t &bseq, bool &helper_ok, NbProf &P) {
  const uint32_t ci = K.ci;
  v4u sg_row = {0, 0, 0, 0}; uint32_t sg_sel = 0; int sg_mw = 0;
  if (SP::match_lane >= 0) {                   // still with the h[i] of the byte just coded (update0 runs before z.run)
    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)c, K.rsrc, (K.l_match && K.canon) ? K.hto + (V.m_limit & K.ht_mask) : kOob, 0, 0);
    V.m_limit = K.l_match ? (V.m_limit + 1) & K.ht_mask : V.m_limit;
    __builtin_amdgcn_raw_buffer_store_b32(V.m_limit, K.rsrc, (K.l_match && K.canon) ? K.cmo + (V.hv & K.cm_mask) * 4u : kOob, 0, 0);   // (its old value: cm_pre)
  }
  c2_put0(&S.mb_byte, bseq << 8 | (uint32_t)c);
  NB_STAMP(5);
  const uint32_t lo = (uint32_t)c & (kNbCand - 1u), un_ = K.un_;
  auto read_staged = [&]() __attribute__((always_inline)) {
    V.hv = S.hspec[ci & ((1u << SP::hh) - 1u)][lo];
    sg_row = *(lds_u4_p)lds_off(&S.selrow[un_][lo]); sg_sel = S.seloff[un_][lo];
    if (SP::nmix) { const uint32_t jj = ci - SP::mix_j0[0]; sg_mw = (int)S.mixst[lo][jj & 7u]; }
  };
  {
    const uint32_t rdy_v = __hip_atomic_load(&S.mb_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");             // (the flag first: what is read behind it is what the flag vouches for)
    read_staged();
    if (UNLIKELY(uni(rdy_v) != uni(bseq))) {   // not yet: wait, then read again
      if (helper_ok) helper_ok = c2_wait(&S.mb_ready, bseq);
      asm volatile("" ::: "memory");
      read_staged();
    }
  }
  if (!helper_ok) return false;
  ++bseq;
  NB_STAMP(6);
  v4u old; uint32_t old_off, old_valid;
  nb_row_old(K, V, S, old, old_off, old_valid);
  nb_row_store(K, old, old_off, old_valid);
  NbProbe pr;
  {
    const uint32_t cxt = V.hv + 16u;
    pr.chk = (cxt >> K.sizebits2) & 255;
    pr.h0 = (cxt * 16u) & (K.ht_mask - 15u);
  }
  if (SP::nmix) {
    const uint32_t rb_was = V.mx_rb;
    nb_mix_set(K, V, SP::mix_lane[0] < NbK<SP>::NC ? rdlane(V.hv, SP::mix_lane[0] & 63u) : 0u);
    V.mwl[1] = K.l_feed ? (V.mx_rb == rb_was ? V.w1_new : sg_mw) : 0;
    nb_mix_rows(K, V, 1u);
  }
  if (SP::match_lane >= 0) {
    nb_match_boundary(K, V, S, (uint32_t)c);
    V.cm_pre = __builtin_amdgcn_raw_buffer_load_b32(K.rsrc, K.l_match ? K.cmo + (V.hv & K.cm_mask) * 4u : kOob, 0, 0);
  }
  const bool near = (V.old1_valid && ((V.old1_off ^ pr.h0) & ~48u) == 0) || (old_valid && ((old_off ^ pr.h0) & ~48u) == 0);
  v4u row = sg_row; uint32_t sel = sg_sel;
  if (UNLIKELY(__ballot(near) != 0)) {         // something this wave wrote late lies in a probed bucket: the probes, patched
    pr.r0 = *(lds_u4_p)lds_off(&S.rowst[un_][0][lo]);
    pr.r1 = *(lds_u4_p)lds_off(&S.rowst[un_][1][lo]);
    pr.r2 = *(lds_u4_p)lds_off(&S.rowst[un_][2][lo]);
    nb_rows_pick(pr, V.old1, V.old1_off, V.old1_valid != 0, old, old_off, old_valid != 0, row, sel);
  }
  if (K.canon) *(lds_u4_p)lds_off(&S.slot[ci]) = row;
  nb_row_take(K, V, row, sel);
  V.oldb = old; V.oldb_off = old_off; V.oldb_valid = old_valid;
  asm volatile("" ::: "memory");
  NB_STAMP(7);
  return true;
}

// ---- The common case as a function of its own: post-processor in PASS state, nothing unusual in the byte.  It is a real call
// (noinline) so that its loop is compiled for itself: inside decode_nibble_body the general form's many live values and exits
// made the compiler carry the decoder state in vector registers and run the loop's control flow through exec masks.
// State crosses in LDS (S.fxs: scalars, S.fxv: per-lane words).  Leaves when a byte cannot start here: fxs[kFxWhy]
//   0  nothing done for the next byte (priming, < 40 coded bytes in the chunk, output capacity reached)
//   1  the EOS flag's step is done and something about it is unusual (fxs[kFxJ], kFxBad, kFxRn hold its results)
//   2  error (fxs[kFxStatus])
enum : int { kFxLow = 0, kFxHigh, kFxCurr, kFxK, kFxAvail, kFxBseq, kFxHelper, kFxWhy, kFxJ, kFxBad, kFxRn, kFxStatus, kFxModel, kFxWord, kFxRoom,
             kFxLenLo, kFxLenHi, kFxStoredLo, kFxStoredHi, kFxCapLo, kFxCapHi, kFxBaseLo, kFxBaseHi,
             kFxPmode, kFxPnative, kFxPskel, kFxPbwt, kFxPlz, kFxPa, kFxPb, kFxPc, kFxPd, kFxPf, kFxProf, kFxN = kFxProf + 36 };
static_assert(kFxN <= 96, "S.fxs");

// ---- PostProcessor.write in state 5 (PostProcessor.cs:80-83) for the bytes nb_fast has decoded: the post-processor only
// consumes the decoded bytes, in order, so the loop parks them exactly as it parks the output of the PASS state and this
// function feeds the parked chunk's bytes [from, to) (virtual positions, at most 256) to the block's PCOMP — the translated
// E8E9 (H, M in LDS), a structurally matched program of zh_zpaql_pcomp.h, or the interpreter.  The machine's registers
// travel in S.fxs.  A real call: none of this belongs to nb_fast's register allocation.
template <class LDS>
__device__ __attribute__((noinline)) int nb_pcomp_drain(const ZhLaunch *Lp_, LDS *Sp_, uint32_t park, uint32_t from_, uint32_t to_) {
  typedef __attribute__((address_space(3))) LDS *lds_S_p;
  LDS &S = *(LDS *)(lds_S_p)(uintptr_t)uni((uint32_t)(uintptr_t)Sp_);
  const ZhLaunch &L = *reinterpret_cast<const ZhLaunch *>(uni64((uint64_t)(uintptr_t)Lp_));
  const uint32_t from = uni(from_), to = uni(to_);
  const uint32_t pnative = uni(S.fxs[kFxPnative]), pskel = uni(S.fxs[kFxPskel]);
  uint32_t pa = uni(S.fxs[kFxPa]), pb = uni(S.fxs[kFxPb]), pc_ = uni(S.fxs[kFxPc]), pd = uni(S.fxs[kFxPd]), pf = uni(S.fxs[kFxPf]);
  Vm &pz = S.pz;
  Sink &sink = S.sink;
  int rc = 0;
  if (uni(S.fxs[kFxPbwt]) && (uint64_t)pb + (to - from) <= (uint64_t)uni(pz.mmask) + 1u) {
    // bwtrle collects the segment in M (`a> 255 ifnot *b=a b++`): the chunk's bytes go there at once
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t *Mw = reinterpret_cast<uint8_t *>(uni64((uint64_t)(uintptr_t)pz.m));
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
      const uint32_t q = ((from & ~255u) + 4u * lane + i) - from;        // this lane's byte i, counted from `from`
      if (q < to - from) Mw[pb + q] = (uint8_t)(park >> (8u * i));
    }
    if (to != from) { pa = (rdlane(park, ((to - 1u) >> 2) & 63u) >> (((to - 1u) & 3u) * 8u)) & 255u; pb += to - from; pf = 0; }
  } else if (uni(S.fxs[kFxPlz])) {
    // The reference's lzpre (LibZPAQ.cs:575-639; zh_native_pcomp_lzpre_108 is its instruction-for-instruction translation) as the
    // state machine it is — D: 0 code byte / 1 literals / 3..5 offset bytes to come / 2 last offset byte; R1: literals or match
    // bytes left; R2: the offset so far; B: the write position in M — with a match copied by the wave: lane i takes byte i of a
    // chunk of up to 64 (the program: one dependent load per byte on one lane).  A match that overlaps its source (distance <
    // length) is periodic with the distance: a chunk reads at a multiple of the distance that the bytes written so far cover.
    // M is written as the program writes it (later matches and the program's own runs — the general path's bytes, the end of a
    // segment — find it as they expect); loads of M go to L2 (glc): the bytes may have been written by other lanes a moment ago.
    const uint32_t lane = threadIdx.x & 63u, minlen = uni(S.fxs[kFxPlz]) - 1u, mm = uni(pz.mmask);
    uint8_t *Mw = reinterpret_cast<uint8_t *>(uni64((uint64_t)(uintptr_t)pz.m));
    uint32_t r1 = uni(S.pr[1]), r2 = uni(S.pr[2]);
    OutBuf o;
    o.base = reinterpret_cast<uint8_t *>(uni64((uint64_t)(uintptr_t)sink.out)); o.cap = uni64(sink.cap);
    o.len = o.stored = uni64(sink.len); o.word = 0; o.park = 0;
    out_room(o);
    for (uint32_t p = from; p != to; ++p) {
      const uint32_t w = rdlane(park, (p >> 2) & 63u);
      const uint32_t x = (w >> ((p & 3u) * 8u)) & 255u;
      if (pd == 0u) {
        pd = (x >> 6) + 1u;
        if (pd == 1u) r1 = x + 1u; else { ++pd; r1 = (x & 63u) + minlen; }
        r2 = 0;
      } else if (pd == 1u) {
        if (lane == 0) Mw[pb & mm] = (uint8_t)x;
        ++pb;
        out_put(o, x, lane);
        if (--r1 == 0u) pd = 0;
      } else if (pd > 2u) {
        r2 = r2 << 8 | x;
        --pd;
      } else {
        r2 = r2 << 8 | x;
        const uint32_t n = r1, dist = r2 + 1u;
        out_flush(o, lane);                              // the literals parked so far
        __builtin_amdgcn_s_waitcnt(0);                   // (their stores to M, too)
        uint32_t done = 0, deff = dist;
        while (done < n) {
          uint32_t m = n - done;
          m = m < 64u ? m : 64u;
          m = m < deff ? m : deff;
          if (lane < m) {
            const uint8_t v = __hip_atomic_load(&Mw[(pb + done + lane - deff) & mm], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            Mw[(pb + done + lane) & mm] = v;
            if (o.len + done + lane < o.cap) o.base[o.len + done + lane] = v;
          }
          done += m;
          __builtin_amdgcn_s_waitcnt(0);
          if (deff < 64u && 2u * deff <= done + dist) deff *= 2u;
        }
        pb += n;
        o.len += n; o.stored = o.len; o.word = 0;
        out_room(o);
        pd = 0;
      }
    }
    out_flush(o, lane);
    if (lane == 0) { sink.len = o.len; S.pr[1] = r1; S.pr[2] = r2; }
  } else if (pnative == ZH_NATIVE_PCOMP_E8E9 && uni(pz.mmask) == 0u) {
    // E8E9 as the scalar operations it amounts to (zh_e8e9.h); its output parked and written like the decoder's own
    const uint32_t lane = threadIdx.x & 63u;
    OutBuf o;
    o.base = reinterpret_cast<uint8_t *>(uni64((uint64_t)(uintptr_t)sink.out)); o.cap = uni64(sink.cap);
    o.len = o.stored = uni64(sink.len); o.word = 0; o.park = 0;
    out_room(o);
    for (uint32_t p = from; p != to; ++p) {
      const uint32_t w = rdlane(park, (p >> 2) & 63u);
      uint32_t ob_;
      if (zh_e8e9_step(pb, pc_, (w >> ((p & 3u) * 8u)) & 255u, ob_)) out_put(o, ob_, lane);
    }
    out_flush(o, lane);
    if (lane == 0) sink.len = o.len;
  } else
  for (uint32_t p = from; p != to; ++p) {
    const uint32_t w = rdlane(park, (p >> 2) & 63u);
    const uint32_t c = (w >> ((p & 3u) * 8u)) & 255u;
    if (pnative == ZH_NATIVE_PCOMP_E8E9)
      rc = zh_native_pcomp_e8e9(pa, pb, pc_, pd, pf, c, (lds_u8_p)lds_off(S.pmreg), uni(pz.mmask), (lds_u32_p)lds_off(S.phreg), uni(pz.hmask), S.pr, &sink, L.budget);
    else if (pskel) {
      ZhPcRegs r{pa, pb, pc_, pd, pf, 0};
      r = zh_pcomp_call(pskel, r, c, pz.m, pz.mmask, pz.h, pz.hmask, S.pr, &sink, L.budget, S.pimm);
      pa = r.a; pb = r.b; pc_ = r.c; pd = r.d; pf = r.f;
      rc = r.rc;
    } else rc = vm_run(pz, c, &sink, L.budget);
    rc = (int)uni((uint32_t)rc);
    if (rc) break;
  }
  if ((threadIdx.x & 63u) == 0) { S.fxs[kFxPa] = pa; S.fxs[kFxPb] = pb; S.fxs[kFxPc] = pc_; S.fxs[kFxPd] = pd; S.fxs[kFxPf] = pf; }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  return rc;
}
template <class SP, bool PROF, class LDS>
__device__ __attribute__((noinline)) void nb_fast(const ZhLaunch *Lp_, LDS *Sp_) {
  // (arguments of a real call arrive in vector registers: say that they are the same in every lane, or every buffer access
  // below is wrapped in a loop over the distinct resources the lanes might hold)
  // (the block's LDS struct: the low half of the generic pointer is its LDS address; going through an LDS-typed pointer lets the
  // compiler see that every access below is an LDS access)
  typedef __attribute__((address_space(3))) LDS *lds_S_p;
  LDS &S = *(LDS *)(lds_S_p)(uintptr_t)uni((uint32_t)(uintptr_t)Sp_);
  const ZhLaunch &L = *reinterpret_cast<const ZhLaunch *>(uni64((uint64_t)(uintptr_t)Lp_));
  const uint32_t lane = threadIdx.x & 63u;
  NbK<SP> K;
  uint8_t *slot_mem = reinterpret_cast<uint8_t *>(uni64((uint64_t)(uintptr_t)(L.arena + (uint64_t)blockIdx.x * L.arena_stride)));
  nb_setup(K, S, &L.models[uni(S.fxs[kFxModel])], slot_mem, lane);
  NbV V;
  {
    uint32_t *w = reinterpret_cast<uint32_t *>(&V);
#pragma unroll
    for (int i = 0; i < kNbVWords; ++i) w[i] = S.fxv[i][lane];
  }
  NbProf P;
  if (PROF) { for (int i = 0; i < 16; ++i) P.prof[i] = 0; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(P.tprev)::"memory"); }
  Dec d;
  d.low = uni(S.fxs[kFxLow]); d.high = uni(S.fxs[kFxHigh]); d.curr = uni(S.fxs[kFxCurr]);
  InBuf in;
  in.stream = L.in; in.total = L.in_total; in.cbase = 0;
  in.k = uni(S.fxs[kFxK]); in.avail = uni(S.fxs[kFxAvail]); in.cur = S.fxv[kNbVWords][lane];
  OutBuf ob;
  ob.base = reinterpret_cast<uint8_t *>((uint64_t)uni(S.fxs[kFxBaseLo]) | (uint64_t)uni(S.fxs[kFxBaseHi]) << 32);
  ob.cap = (uint64_t)uni(S.fxs[kFxCapLo]) | (uint64_t)uni(S.fxs[kFxCapHi]) << 32;
  ob.len = (uint64_t)uni(S.fxs[kFxLenLo]) | (uint64_t)uni(S.fxs[kFxLenHi]) << 32;
  ob.stored = (uint64_t)uni(S.fxs[kFxStoredLo]) | (uint64_t)uni(S.fxs[kFxStoredHi]) << 32;
  ob.word = uni(S.fxs[kFxWord]); ob.room = uni(S.fxs[kFxRoom]); ob.park = S.fxv[kNbVWords + 1][lane];
  uint32_t bseq = uni(S.fxs[kFxBseq]);
  const uint32_t pmode = uni(S.fxs[kFxPmode]);          // 1: post-processor program loaded — `ob` is a staging cursor, chunks go to nb_pcomp_drain
  bool helper_ok = true;
  uint32_t why = 0, j = 0, bad = 0, rn = 0, status = 0;
  const uint32_t klim = in.avail >= 40u ? in.avail - 40u : 0u;
  uint32_t vlo = (uint32_t)(uintptr_t)ob.base + (uint32_t)ob.len;     // low bits of the virtual output position
  uint32_t nput = 0, room = ob.room, word = ob.word;
  if constexpr (ZH_NB_ASM != 0) {
    // ---- the loop in assembly (zh_nb_fast.h, tools/gen_nb_asm.py): constants and state through LDS
    if (lds_off(S.stretch) == 0 && V.rowvalid) {
      const uint32_t hmask_ = (1u << SP::hh) - 1u;
      constexpr int kNK = NbT<SP>::shape == 2 ? (int)kNmK_count : (int)kNbK_count;
      uint32_t kc[kNK];
      kc[kNbK_tab] = K.tab; kc[kNbK_slot] = K.wrow; kc[kNbK_wr2] = K.wrow + K.node[2]; kc[kNbK_wr3] = K.wrow + K.node[3]; kc[kNbK_wr4] = K.wrow + K.node[4];
      kc[kNbK_sh2] = K.sh2; kc[kNbK_sh3] = K.sh3; kc[kNbK_sh4] = K.sh4;
      kc[kNbK_ey1] = K.ybit[1] ? 32767u : 0u; kc[kNbK_ey2] = K.ybit[2] ? 32767u : 0u; kc[kNbK_ey3] = K.ybit[3] ? 32767u : 0u;
      kc[kNbK_ys1] = K.ybit[1] * 8u; kc[kNbK_ys2] = K.ybit[2] * 8u; kc[kNbK_ys3] = K.ybit[3] * 8u;
      kc[kNbK_hto] = K.hto; kc[kNbK_htm15] = K.ht_mask - 15u; kc[kNbK_sb2] = K.sizebits2; kc[kNbK_cshift] = K.cshift; kc[kNbK_issem] = (uint32_t)K.isse_m;
      kc[kNbK_c8off] = 256u + 32u * K.g; kc[kNbK_g] = K.g;
      kc[kNbK_selrow] = lds_off(&S.selrow[K.un_][0]); kc[kNbK_seloff] = lds_off(&S.seloff[K.un_][0]);
      kc[kNbK_hspec] = lds_off(&S.hspec[K.ci & hmask_][0]); kc[kNbK_rowst] = lds_off(&S.rowst[K.un_][0][0]); kc[kNbK_slotoff] = lds_off(&S.slotoff[K.ci]);
      kc[kNbK_koob] = kOob; kc[kNbK_c2047] = 2047u; kc[kNbK_c512k] = (1u << 19) - 1u; kc[kNbK_rnd] = 1u << 12; kc[kNbK_c10000] = 0x10000u;
      kc[kNbK_evo] = (K.canon && K.l_ii) ? K.hto : kOob; kc[kNbK_mb] = lds_off(&S.mb_nib);
      if constexpr (NbT<SP>::shape == 2) {
        constexpr uint32_t m4 = NbK<SP>::mx_m4;
        kc[kNmK_rslot] = K.l_ii ? lds_off(&S.slot[K.ci]) : lds_off(&S.zrow);
        kc[kNmK_rate] = (uint32_t)K.mx_rate; kc[kNmK_vomix] = K.vo_mix; kc[kNmK_cm0] = (16u + 2u * K.g) * m4;
        kc[kNmK_row1_1] = (1u + K.pre[1]) * m4; kc[kNmK_row1_2] = (2u + K.pre[2]) * m4; kc[kNmK_row1_3] = (4u + K.pre[3]) * m4; kc[kNmK_row1_4] = (8u + K.pre[4]) * m4;
        kc[kNmK_pre2] = K.pre[2] * m4; kc[kNmK_pre3] = K.pre[3] * m4; kc[kNmK_pre4] = K.pre[4] * m4;
        kc[kNmK_lane1] = lane + 1u; kc[kNmK_htmask] = K.ht_mask; kc[kNmK_cmo] = K.cmo; kc[kNmK_cmmask] = K.cm_mask;
        kc[kNmK_mixst] = lds_off(&S.mixst[0][K.ci & 7u]); kc[kNmK_mixb] = lds_off(&S.mixb[K.ci]);
        kc[kNmK_evm] = (K.canon && K.l_match) ? 0u : kOob;
      }
#pragma unroll
      for (int i = 0; i < kNK; ++i) S.fxk[i][lane] = kc[i];
      S.fxa[kNbS_rx][lane] = V.row_x; S.fxa[kNbS_rq1][lane] = V.row_q1; S.fxa[kNbS_rq2][lane] = V.row_q2; S.fxa[kNbS_rq3][lane] = V.row_q3;
      S.fxa[kNbS_rowoff][lane] = V.rowoff; S.fxa[kNbS_hv][lane] = V.hv;
      S.fxa[kNbS_ob0][lane] = V.oldb.x; S.fxa[kNbS_ob1][lane] = V.oldb.y; S.fxa[kNbS_ob2][lane] = V.oldb.z; S.fxa[kNbS_ob3][lane] = V.oldb.w;
      S.fxa[kNbS_oboff][lane] = V.oldb_valid ? V.oldb_off : 0xFFFFFFFFu;       // (no row written back yet: a place no bucket has)
      S.fxa[kNbS_cur][lane] = in.cur;
      if constexpr (NbT<SP>::shape == 2) {
        S.fxa[kNmS_m_len][lane] = V.m_len; S.fxa[kNmS_m_ptr][lane] = V.m_ptr; S.fxa[kNmS_m_limit][lane] = V.m_limit; S.fxa[kNmS_m_byte][lane] = V.m_byte;
        S.fxa[kNmS_pm0][lane] = (uint32_t)V.pm0; S.fxa[kNmS_pm1][lane] = (uint32_t)V.pm1; S.fxa[kNmS_cm_pre][lane] = V.cm_pre;
        S.fxa[kNmS_va_pre][lane] = V.va_pre; S.fxa[kNmS_vb_pre][lane] = V.vb_pre; S.fxa[kNmS_mbn_pre][lane] = V.mbn_pre; S.fxa[kNmS_mbc_pre][lane] = V.mbc_pre;
        S.fxa[kNmS_mx_rb][lane] = V.mx_rb; S.fxa[kNmS_w1_new][lane] = (uint32_t)V.w1_new;
#pragma unroll
        for (int dd = 1; dd <= 4; ++dd) { S.fxa[kNmS_mwl1 + dd - 1][lane] = (uint32_t)V.mwl[dd]; S.fxa[kNmS_mra1 + dd - 1][lane] = V.mrowl[dd]; }
      }
      const uint64_t sm = (uint64_t)(uintptr_t)slot_mem;
      const v4u rs = {uni((uint32_t)sm), uni((uint32_t)(sm >> 32) & 0xffffu), uni((uint32_t)L.models[uni(S.fxs[kFxModel])].arena_bytes), 0x00020000u};
      const uint32_t kb = lds_off(&S.fxk[0][lane]), vb = lds_off(&S.fxa[0][lane]);
      const uint32_t sqb = uni(lds_off(S.squash) + 4096u), nsb = uni(lds_off(S.ns));
      uint32_t lo_ = uni(d.low), hi_ = uni(d.high), cu_ = uni(d.curr), k_ = uni(in.k), asm_why = 0, obad = 0, ofail = 0, m0s = 0;
      const uint32_t klim_s = uni(klim);
      bseq = uni(bseq); nput = uni(nput); room = uni(room); word = uni(word);
      for (;;) {
        S.fxa[kNbS_park][lane] = ob.park;
        const uint32_t vlo_s = uni(vlo);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (NbT<SP>::shape == 2) {
          const uint32_t mxb = uni(K.mx_base), mxs = uni(K.mx_size1), pmb = uni(lds_off(S.pm01));
          if constexpr (PROF) {
            if constexpr (SP::mix_m[0] == 8) { ZH_NB_FAST_MID8_LOOP_PROF(lo_, hi_, cu_, k_, bseq, nput, room, word, asm_why, obad, ofail, m0s, klim_s, vlo_s, kb, vb, rs, sqb, nsb, mxb, mxs, pmb); }
            else { ZH_NB_FAST_MID_LOOP_PROF(lo_, hi_, cu_, k_, bseq, nput, room, word, asm_why, obad, ofail, m0s, klim_s, vlo_s, kb, vb, rs, sqb, nsb, mxb, mxs, pmb); }
            for (int i = 0; i < 12; ++i) P.prof[i] += S.fxa[kNmS_count + i][0];
          } else if constexpr (SP::mix_m[0] == 8) {
            ZH_NB_FAST_MID8_LOOP(lo_, hi_, cu_, k_, bseq, nput, room, word, asm_why, obad, ofail, m0s, klim_s, vlo_s, kb, vb, rs, sqb, nsb, mxb, mxs, pmb);
          } else {
            ZH_NB_FAST_MID_LOOP(lo_, hi_, cu_, k_, bseq, nput, room, word, asm_why, obad, ofail, m0s, klim_s, vlo_s, kb, vb, rs, sqb, nsb, mxb, mxs, pmb);
          }
        } else if constexpr (PROF) {
          if constexpr (SP::isse == 0) { ZH_NB_FAST_MIN1_LOOP_PROF(lo_, hi_, cu_, k_, bseq, nput, room, word, asm_why, obad, ofail, m0s, klim_s, vlo_s, kb, vb, rs, sqb, nsb); }
          else { ZH_NB_FAST_MIN_LOOP_PROF(lo_, hi_, cu_, k_, bseq, nput, room, word, asm_why, obad, ofail, m0s, klim_s, vlo_s, kb, vb, rs, sqb, nsb); }
          for (int i = 0; i < 12; ++i) P.prof[i] += S.fxa[kNbS_count + i][0];
        } else if constexpr (SP::isse == 0) {
          ZH_NB_FAST_MIN1_LOOP(lo_, hi_, cu_, k_, bseq, nput, room, word, asm_why, obad, ofail, m0s, klim_s, vlo_s, kb, vb, rs, sqb, nsb);
        } else {
          ZH_NB_FAST_MIN_LOOP(lo_, hi_, cu_, k_, bseq, nput, room, word, asm_why, obad, ofail, m0s, klim_s, vlo_s, kb, vb, rs, sqb, nsb);
        }
        ob.park = S.fxa[kNbS_park][lane];
        if (asm_why != 3) break;
        ob.len += nput; ob.word = word; ob.room = room; vlo += nput; nput = 0;      // the parked 256-byte chunk is complete
        if (pmode) {
          const int prc = (int)uni((uint32_t)nb_pcomp_drain<LDS>(Lp_, Sp_, ob.park, (uint32_t)ob.stored, (uint32_t)ob.len));
          ob.stored = ob.len; out_room(ob);
          if (UNLIKELY(prc != 0)) { asm_why = 4; status = (uint32_t)prc; break; }
        } else out_flush(ob, lane);
        room = uni(ob.room);
      }
      d.low = uni(lo_); d.high = uni(hi_); d.curr = uni(cu_); in.k = uni(k_);
      V.row_x = S.fxa[kNbS_rx][lane]; V.row_q1 = S.fxa[kNbS_rq1][lane]; V.row_q2 = S.fxa[kNbS_rq2][lane]; V.row_q3 = S.fxa[kNbS_rq3][lane];
      V.rowoff = S.fxa[kNbS_rowoff][lane]; V.hv = S.fxa[kNbS_hv][lane];
      V.oldb.x = S.fxa[kNbS_ob0][lane]; V.oldb.y = S.fxa[kNbS_ob1][lane]; V.oldb.z = S.fxa[kNbS_ob2][lane]; V.oldb.w = S.fxa[kNbS_ob3][lane];
      { const uint32_t oo = S.fxa[kNbS_oboff][lane]; V.oldb_valid = oo != 0xFFFFFFFFu; V.oldb_off = oo; }
      if constexpr (NbT<SP>::shape == 2) {
        V.m_len = S.fxa[kNmS_m_len][lane]; V.m_ptr = S.fxa[kNmS_m_ptr][lane]; V.m_limit = S.fxa[kNmS_m_limit][lane]; V.m_byte = S.fxa[kNmS_m_byte][lane];
        V.pm0 = (int)S.fxa[kNmS_pm0][lane]; V.pm1 = (int)S.fxa[kNmS_pm1][lane]; V.cm_pre = S.fxa[kNmS_cm_pre][lane];
        V.va_pre = S.fxa[kNmS_va_pre][lane]; V.vb_pre = S.fxa[kNmS_vb_pre][lane]; V.mbn_pre = S.fxa[kNmS_mbn_pre][lane]; V.mbc_pre = S.fxa[kNmS_mbc_pre][lane];
        V.mx_rb = S.fxa[kNmS_mx_rb][lane]; V.w1_new = (int)S.fxa[kNmS_w1_new][lane];
#pragma unroll
        for (int dd = 1; dd <= 4; ++dd) { V.mwl[dd] = (int)S.fxa[kNmS_mwl1 + dd - 1][lane]; V.mrowl[dd] = S.fxa[kNmS_mra1 + dd - 1][lane]; }
      }
      if (asm_why == 2) { why = 2; status = ofail ? (uint32_t)-24 : (uint32_t)ZH_E_CORRUPT; }
      if (asm_why == 4) why = 2;                           // the post-processor failed: status holds its code
    }
  } else
  for (;;) {
    NB_STAMP(11);
    d.low = uni(d.low); d.high = uni(d.high); d.curr = uni(d.curr); in.k = uni(in.k);
    if (UNLIKELY(d.curr == 0 || in.k > klim || room == 0 || in.avail < 40u)) break;
    bad = 0; j = 0;
    NB_STAMP(12);
    ZH_DEC_STEP(d, 0u, j, bad, rn);              // EOS flag: p = 0
    if (UNLIKELY((bad | rn | j) != 0)) { why = 1; break; }
    uint32_t err = 0;
    const uint32_t cb = uni(nb_decode_byte<SP, PROF, true>(K, V, S, d, in, j, bad, err, bseq, P));
    if (UNLIKELY(bad != 0)) { why = 2; status = (uint32_t)ZH_E_CORRUPT; break; }
    if (UNLIKELY((!nb_boundary<SP, PROF>(K, V, S, (int)cb, bseq, helper_ok, P)))) { why = 2; status = (uint32_t)-24; break; }     // ZPAQHIP_E_HIP
    // ---- PostProcessor.write in PASS state: ZPAQL.outc (ZPAQL.cs:201-207), the dword assembled on the scalar unit
    {
      const uint32_t v = vlo + nput;
      ++nput; --room;
      const uint32_t sh = (v & 3u) * 8u;
      word = sh ? (word | cb << sh) : cb;
      if ((v & 3u) == 3u) {
        ob.park = wrlane(word, (v >> 2) & 63u, ob.park);
        if (UNLIKELY((v & 255u) == 255u)) {
          ob.len += nput; ob.word = word; vlo += nput; nput = 0;
          if (pmode) {
            const int prc = (int)uni((uint32_t)nb_pcomp_drain<LDS>(Lp_, Sp_, ob.park, (uint32_t)ob.stored, (uint32_t)ob.len));
            ob.stored = ob.len; out_room(ob);
            if (UNLIKELY(prc != 0)) { why = 2; status = (uint32_t)prc; break; }
          } else out_flush(ob, lane);
          room = ob.room;
        }
      }
    }
    NB_STAMP(9);
  }
  ob.len += nput; ob.word = word; ob.room = room;
  if (pmode && ob.len != ob.stored) {                    // the bytes of the unfinished chunk: the general path's next byte goes to the program directly
    uint32_t park = ob.park;
    if ((uint32_t)ob.len & 3u) park = wrlane(ob.word, ((uint32_t)ob.len >> 2) & 63u, park);
    const int prc = (int)uni((uint32_t)nb_pcomp_drain<LDS>(Lp_, Sp_, park, (uint32_t)ob.stored, (uint32_t)ob.len));
    ob.stored = ob.len;
    if (prc != 0 && why != 2) { why = 2; status = (uint32_t)prc; }
  }
  {
    uint32_t *w = reinterpret_cast<uint32_t *>(&V);
#pragma unroll
    for (int i = 0; i < kNbVWords; ++i) S.fxv[i][lane] = w[i];
    S.fxv[kNbVWords][lane] = in.cur;
    S.fxv[kNbVWords + 1][lane] = ob.park;
  }
  if (lane == 0) {
    S.fxs[kFxLow] = d.low; S.fxs[kFxHigh] = d.high; S.fxs[kFxCurr] = d.curr; S.fxs[kFxK] = in.k; S.fxs[kFxBseq] = bseq;
    S.fxs[kFxWhy] = why; S.fxs[kFxJ] = j; S.fxs[kFxBad] = bad; S.fxs[kFxRn] = rn; S.fxs[kFxStatus] = status;
    S.fxs[kFxWord] = ob.word; S.fxs[kFxRoom] = ob.room;
    S.fxs[kFxLenLo] = (uint32_t)ob.len; S.fxs[kFxLenHi] = (uint32_t)(ob.len >> 32);
    S.fxs[kFxStoredLo] = (uint32_t)ob.stored; S.fxs[kFxStoredHi] = (uint32_t)(ob.stored >> 32);
    if (PROF) for (int i = 0; i < 16; ++i) { S.fxs[kFxProf + 2 * i] = (uint32_t)P.prof[i]; S.fxs[kFxProf + 2 * i + 1] = (uint32_t)(P.prof[i] >> 32); }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <class LDS>
__device__ __attribute__((noinline)) void nb_load_tables(const ZhLaunch &L, LDS &S, uint32_t lane) {
  const ZhTables *T = L.tables;
  for (uint32_t i = lane; i < 32768 / 8; i += 64) reinterpret_cast<uint4 *>(S.stretch)[i] = reinterpret_cast<const uint4 *>(T->stretch)[i];
  for (uint32_t i = lane; i < 4096 / 8; i += 64) reinterpret_cast<uint4 *>(S.squash)[i] = reinterpret_cast<const uint4 *>(T->squash)[i];
  for (uint32_t i = lane; i < 1024 / 16; i += 64) reinterpret_cast<uint4 *>(S.ns)[i] = reinterpret_cast<const uint4 *>(T->ns)[i];
}
template <class LDS>
__device__ __attribute__((noinline)) void nb_load_pm01(const ZhLaunch &L, LDS &S, uint32_t lane) {
  for (uint32_t i = lane; i < 256; i += 64) {            // the two predictions of a match of length i
    const int dk = L.tables->dt2k[i];
    const uint32_t lo = (uint16_t)S.stretch[dk & 32767], hi = (uint16_t)S.stretch[(-dk) & 32767];
    S.pm01[i] = i ? lo | hi << 16 : 0u;
  }
}

// The inverse BWT of the reference's bwtrle program at the end of a block's only segment, wave-wide (zh_ibwt.h) instead of the
// program's one dependent load per byte.  Its tables overlay this block's stretch / squash / ns / pm01 / entry tables — the
// model is finished (one segment) and the helper wave only polls its mailbox, which lies behind them — and are loaded again.
typedef BwtLdsT<1024u> NbBwtLds;
template <class LDS>
__device__ __attribute__((noinline)) bool nb_ibwt(const ZhLaunch &L, LDS &S, uint32_t lane, uint32_t n_in, uint32_t *produced, bool *touched) {
  static_assert(sizeof(NbBwtLds) <= offsetof(LDS, slot), "the inverse BWT's tables overlay the model's tables only");
  Vm &pz = S.pz;
  Sink &sink = S.sink;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  __builtin_amdgcn_s_waitcnt(0);
  const uint64_t have = uni64(sink.len), cap = uni64(sink.cap);
  const bool ok = ibwt_block<1024u>(*reinterpret_cast<NbBwtLds *>(&S), pz.m, pz.h, n_in, (uint64_t)uni(pz.mmask) + 1u, (uint64_t)uni(pz.hmask) + 1u,
                                    sink.out + have, cap > have ? cap - have : 0u, produced, lane, touched);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  nb_load_tables(L, S, lane);
  nb_wave_sync();
  nb_load_pm01(L, S, lane);
  nb_wave_sync();
  return ok;
}

template <class SP, bool PROF, class LDS>
__device__ __forceinline__ void decode_nibble_body(const ZhLaunch &L, LDS &S) {
  constexpr uint64_t kII = NbK<SP>::kII;
  static_assert(SP::n <= NbK<SP>::NC + (SP::nmix && SP::mix_lane[0] >= NbK<SP>::NC ? 1u : 0u) && NbK<SP>::NC * NbK<SP>::NG <= 64, "lane budget");
  NbProf P;
  for (int i = 0; i < 16; ++i) P.prof[i] = 0;
  P.tprev = 0;
  const uint32_t lane = threadIdx.x & 63u;
  const bool wave_a = (threadIdx.x >> 6) == 0;

  if (wave_a) {  // model-independent tables -> LDS
    nb_load_tables(L, S, lane);
    if (lane == 0) { S.zrow = v4u_{0, 0, 0, 0}; S.mb_cmd = 0; S.mb_ack = 0; S.mb_nib = 0; S.mb_byte = 0; S.mb_ready = 0; }
  }
  __syncthreads();                                       // the only workgroup barrier of the kernel
  if (!wave_a) { nb_helper<SP, LDS, PROF>(L, S, lane, blockIdx.x); return; }
  uint32_t cmd_seq = 0;
  nb_load_pm01(L, S, lane);
  nb_wave_sync();

  uint8_t *slot_mem = L.arena + (uint64_t)blockIdx.x * L.arena_stride;

  for (;;) {
    uint32_t bi = 0;
    if (lane == 0) bi = atomicAdd(L.queue, 1u);
    bi = uni((uint32_t)__shfl((int)bi, 0));
    if (bi >= L.n_blocks) break;                       // every wave reaches this exit

    const ZhBlockDesc *bdp = &L.blocks[bi];
    const uint32_t model_i = uni(bdp->model);
    const uint32_t first_seg = uni(bdp->first_seg), n_seg = uni(bdp->n_seg);
    const uint64_t b_out_off = uni64(bdp->out_off), b_out_cap = uni64(bdp->out_cap);
    const ZhModel *M = &L.models[model_i];
    const uint32_t hh = uni(M->hh), hmb = uni(M->hm);

    // ---- Predictor.init (Predictor.cs:82-171): the arena tables this kernel keeps in HBM
    for (uint32_t i = 0; i < SP::n; ++i) {
      const ZhComp &cp = M->comp[i];
      const uint32_t type = uni(cp.type);
      uint8_t *cm = slot_mem + uni64(cp.cm_off), *ht = slot_mem + uni64(cp.ht_off);
      const uint64_t cmb = uni64(cp.cm_bytes), htb = uni64(cp.ht_bytes);
      uint4 pat = make_uint4(0, 0, 0, 0);
      bool fill_cm = false;
      if (type == ZH_MATCH) fill_cm = true;
      else if (type == ZH_MIX) { const uint32_t w = 65536u / uni(cp.arg[2]); pat = make_uint4(w, w, w, w); fill_cm = true; }
      if (fill_cm) { uint4 *q = reinterpret_cast<uint4 *>(cm); for (uint64_t k = lane; k < cmb / 16; k += 64) q[k] = pat; }
      if (type == ZH_ICM || type == ZH_ISSE || type == ZH_MATCH) {
        uint4 *q = reinterpret_cast<uint4 *>(ht);
        for (uint64_t k = lane; k < htb / 16; k += 64) q[k] = make_uint4(0, 0, 0, 0);
      }
    }
    {  // VM memories: arena tail zeroed; LDS copies zeroed
      const uint64_t h_off = uni64(M->h_off), tail = uni64(M->arena_bytes) - h_off;
      uint4 *z = reinterpret_cast<uint4 *>(slot_mem + h_off);
      for (uint64_t i = lane; i < tail / 16; i += 64) z[i] = make_uint4(0, 0, 0, 0);
      for (uint32_t i = lane; i < 256; i += 64) { S.r[i] = 0; S.pr[i] = 0; S.hreg[i] = 0; }
      for (uint32_t i = lane; i < kMBytes / 4; i += 64) reinterpret_cast<uint32_t *>(S.mreg)[i] = 0;
      for (uint32_t i = lane; i < kPHWords; i += 64) S.phreg[i] = 0;
      for (uint32_t i = lane; i < kPMBytes / 4; i += 64) reinterpret_cast<uint32_t *>(S.pmreg)[i] = 0;
    }
    // ---- ICM / ISSE entry tables in LDS.  Unit u of S.ent belongs to the u-th ICM/ISSE component.
    {
      for (uint32_t j = lane; j < 256; j += 64) {
        const uint32_t n0 = S.ns[j * 4 + 2], n1 = S.ns[j * 4 + 3];
        const uint32_t cinit = ((n1 * 2 + 1) << 22) / (n0 + n1 + 1);                // StateTable.cminit
        const int stv = S.stretch[cinit >> 8];
        const v2u e_icm = {cinit, (uint32_t)stv};
        const v2u e_isse = {1u << 15, (uint32_t)clamp512k(stv * 1024)};
        uint32_t u = 0;
        for (uint32_t i = 0; i < NbK<SP>::NC; ++i) {     // (lanes beyond the model's components are replicas: NbMin1)
          if (!((kII >> i) & 1)) continue;
          S.ent[u][j] = ((SP::icm >> i) & 1) ? e_icm : e_isse;
          ++u;
        }
      }
      if (lane < 8) { S.slot[lane] = v4u{0, 0, 0, 0}; S.lent[lane] = v2u{0, 0}; S.lsink[lane] = 0; S.slotoff[lane] = 0; S.mixb[lane] = 0; }
    }
    nb_wave_sync();

    NbK<SP> K;
    nb_setup(K, S, M, slot_mem, lane);
    if (K.l_match && lane == (uint32_t)SP::match_lane) (slot_mem + K.hto)[0] = 1;      // Predictor.cs:121 ht(0)=1 ... overwritten like the reference
    const uint32_t ci = K.ci;

    // HCOMP machine (ZPAQL.cs:1010-1026): H and M in LDS; the helper wave runs the program
    Vm &hz = S.hz;
    hz.a = hz.b = hz.c = hz.d = hz.f = 0;
    hz.len = uni(M->hcomp_len);
    hz.hmask = (uint32_t)((1ull << hh) - 1); hz.mmask = (uint32_t)((1ull << hmb) - 1);
    hz.h = S.hreg; hz.m = S.mreg; hz.r = S.r;

    int pp_state = 0, pp_hsize = 0;                    // PostProcessor (PostProcessor.cs:12-16)
    uint32_t pp_len = 0;
    Vm &pz = S.pz;
    pz.a = pz.b = pz.c = pz.d = pz.f = 0;
    pz.prog = nullptr; pz.len = 0;
    const uint32_t phb = uni(M->ph), pmb = uni(M->pm);
    pz.mmask = (uint32_t)((1ull << pmb) - 1); pz.hmask = (uint32_t)((1ull << phb) - 1);
    pz.m = pmb < 31 && (1u << pmb) <= (uint32_t)kPMBytes ? S.pmreg : slot_mem + uni64(M->pm_off);
    pz.h = phb < 31 && (1u << phb) <= (uint32_t)kPHWords ? S.phreg : reinterpret_cast<uint32_t *>(slot_mem + uni64(M->ph_off));
    pz.r = S.pr;
    const bool p_lds = pz.m == S.pmreg && pz.h == S.phreg;
    uint32_t pnative = 0, pskel = 0;                    // the loaded program is the translated E8E9 / has the structure of one of zh_zpaql_pcomp.h's
    uint32_t pbwt = 0;                                  // ... is the reference's bwtrle, operand for operand, in a block of one segment: M collects, nb_ibwt inverts
    uint32_t plz = 0;                                   // ... is the reference's lzpre, operand for operand (1 + its minimum match length): nb_pcomp_drain's state machine
    uint32_t pa = 0, pb = 0, pc_ = 0, pd = 0, pf = 0;
    uint8_t *pzbuf = slot_mem + uni64(M->pz_off) + ZH_CODE_PAD;
    OutBuf sb;                                          // state 5: the cursor nb_fast parks decoded bytes under on their way to the program
    sb.base = nullptr; sb.cap = ~0ull; sb.len = 0; sb.stored = 0; sb.word = 0; sb.park = 0;
    out_room(sb);

    Dec d;
    d.low = 1; d.high = 0xFFFFFFFFu; d.curr = 0;
    OutBuf ob;
    ob.base = L.out + b_out_off; ob.cap = b_out_cap; ob.len = 0; ob.stored = 0; ob.word = 0; ob.park = 0;
    out_room(ob);
    Sink &sink = S.sink;
    sink.out = ob.base; sink.cap = ob.cap; sink.len = 0;
    nb_wave_sync();
    uint32_t bseq = 1;                                  // bytes of this block decoded so far + 1 (the helper wave's clock)
    bool helper_ok = true;
    {                                                   // wake the helper wave for this block (tables and VM memories are ready)
      if (lane == 0) { S.mb_model = model_i; S.mb_nib = 0; S.mb_byte = 0; S.mb_ready = 0; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      ++cmd_seq;
      c2_put0(&S.mb_cmd, cmd_seq << 2 | kC2New);
      helper_ok = c2_wait(&S.mb_ack, cmd_seq << 2 | kC2New);
    }
    if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(P.tprev)::"memory"); }
    InBuf in;
    in.stream = L.in; in.total = L.in_total; in.cbase = 0; in.k = 0; in.avail = 0; in.cur = 0;

    NbV V;
    {
      uint32_t *w = reinterpret_cast<uint32_t *>(&V);
#pragma unroll
      for (int i = 0; i < kNbVWords; ++i) w[i] = 0;
      V.mx_rb = kOob;
    }

    int failed = 0;
    for (uint32_t s = 0; s < n_seg; ++s) {
      const uint32_t si = first_seg + s;
      const uint64_t produced0 = pp_state == 5 ? uni64(sink.len) : ob.len;
      int status = 0;
      if (failed) {
        if (lane == 0) {
          ZhSegResult res;
          res.status = ZH_E_SKIPPED; res.pp_state = (uint32_t)pp_state; res.out_off = b_out_off + produced0; res.out_len = 0;
          res.in_used = 0;
          L.results[si] = res;
        }
        continue;
      }
      const uint64_t seg_off = uni64(L.segs[si].in_off);
      in_seek(in, seg_off, lane);
      if (s == 0) {                                      // first nibble of the block (h[] = 0)
        NbProbe pr;
        nb_rows_issue(K, V, 1u, pr, true);
        v4u row; uint32_t sel;
        nb_rows_pick(pr, v4u{0, 0, 0, 0}, 0u, false, v4u{0, 0, 0, 0}, 0u, false, row, sel);
        if (K.canon) *(lds_u4_p)lds_off(&S.slot[ci]) = row;
        nb_row_take(K, V, row, sel);
        if (SP::nmix) {
          nb_mix_set(K, V, 0u);
          nb_mix_rows(K, V, 1u);
          V.mwl[1] = (int)__builtin_amdgcn_raw_buffer_load_b32(K.rsrc, V.mrowl[1], 0, 0);
        }
      }

      for (;;) {                                       // one decoded byte per iteration
        uint32_t bad = 0, rn = 0, j = 0, err = 0;
        bool after_eos = false;
        // ---- the common case runs in nb_fast
        if (LIKELY((pp_state == 1 || pp_state == 5) && helper_ok)) {
          OutBuf &fo = pp_state == 5 ? sb : ob;
          if (in.k + 40u > in.avail && in.cbase + in.avail < in.total) in_seek(in, in_pos(in), lane);   // nb_fast wants >= 40 coded bytes in the chunk: re-base it at the cursor
          if (d.curr != 0 && in.avail >= 40u && in.k + 40u <= in.avail && fo.room != 0) {
            {
              const uint32_t *w = reinterpret_cast<const uint32_t *>(&V);
#pragma unroll
              for (int i = 0; i < kNbVWords; ++i) S.fxv[i][lane] = w[i];
              S.fxv[kNbVWords][lane] = in.cur;
              S.fxv[kNbVWords + 1][lane] = fo.park;
            }
            if (lane == 0) {
              S.fxs[kFxLow] = d.low; S.fxs[kFxHigh] = d.high; S.fxs[kFxCurr] = d.curr; S.fxs[kFxK] = in.k; S.fxs[kFxAvail] = in.avail;
              S.fxs[kFxBseq] = bseq; S.fxs[kFxModel] = model_i; S.fxs[kFxWord] = fo.word; S.fxs[kFxRoom] = fo.room;
              S.fxs[kFxLenLo] = (uint32_t)fo.len; S.fxs[kFxLenHi] = (uint32_t)(fo.len >> 32);
              S.fxs[kFxStoredLo] = (uint32_t)fo.stored; S.fxs[kFxStoredHi] = (uint32_t)(fo.stored >> 32);
              S.fxs[kFxCapLo] = (uint32_t)fo.cap; S.fxs[kFxCapHi] = (uint32_t)(fo.cap >> 32);
              S.fxs[kFxBaseLo] = (uint32_t)(uintptr_t)fo.base; S.fxs[kFxBaseHi] = (uint32_t)((uintptr_t)fo.base >> 32);
              S.fxs[kFxPmode] = pp_state == 5; S.fxs[kFxPnative] = pnative; S.fxs[kFxPskel] = pskel; S.fxs[kFxPbwt] = pbwt; S.fxs[kFxPlz] = plz;
              S.fxs[kFxPa] = pa; S.fxs[kFxPb] = pb; S.fxs[kFxPc] = pc_; S.fxs[kFxPd] = pd; S.fxs[kFxPf] = pf;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            nb_fast<SP, PROF, LDS>(&L, &S);
            asm volatile("" ::: "memory");
            {
              uint32_t *w = reinterpret_cast<uint32_t *>(&V);
#pragma unroll
              for (int i = 0; i < kNbVWords; ++i) w[i] = S.fxv[i][lane];
              fo.park = S.fxv[kNbVWords + 1][lane];
            }
            d.low = uni(S.fxs[kFxLow]); d.high = uni(S.fxs[kFxHigh]); d.curr = uni(S.fxs[kFxCurr]); in.k = uni(S.fxs[kFxK]);
            bseq = uni(S.fxs[kFxBseq]);
            fo.word = uni(S.fxs[kFxWord]); fo.room = uni(S.fxs[kFxRoom]);
            fo.len = (uint64_t)uni(S.fxs[kFxLenLo]) | (uint64_t)uni(S.fxs[kFxLenHi]) << 32;
            fo.stored = (uint64_t)uni(S.fxs[kFxStoredLo]) | (uint64_t)uni(S.fxs[kFxStoredHi]) << 32;
            if (pp_state == 5) { pa = uni(S.fxs[kFxPa]); pb = uni(S.fxs[kFxPb]); pc_ = uni(S.fxs[kFxPc]); pd = uni(S.fxs[kFxPd]); pf = uni(S.fxs[kFxPf]); }
            const uint32_t why = uni(S.fxs[kFxWhy]);
            if (PROF) for (int i = 0; i < 16; ++i) P.prof[i] += (uint64_t)S.fxs[kFxProf + 2 * i] | (uint64_t)S.fxs[kFxProf + 2 * i + 1] << 32;
            if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(P.tprev)::"memory"); }
            if (why == 2) { status = (int)uni(S.fxs[kFxStatus]); if (status == -24) helper_ok = false; break; }
            if (why == 1) { after_eos = true; j = uni(S.fxs[kFxJ]); bad = uni(S.fxs[kFxBad]); rn = uni(S.fxs[kFxRn]); }
          }
        }
        // ---- Decoder.decompress prologue (Decoder.cs:36-45)
        if (!after_eos) {
          if (UNLIKELY(d.curr == 0)) {
            uint32_t cu = 0;
            for (int i = 0; i < 4; ++i) cu = cu << 8 | (uint32_t)in_get(in, lane);
            d.curr = uni(cu);
          }
          bad = 0; j = 0;
          d.low = uni(d.low); d.high = uni(d.high); d.curr = uni(d.curr);
          ZH_DEC_STEP(d, 0u, j, bad, rn);              // EOS flag: p = 0
        }
        bad = uni(bad); rn = uni(rn); j = uni(j);
        if (UNLIKELY(bad)) { status = ZH_E_CORRUPT; break; }
        if (UNLIKELY(rn)) { if (dec_renorm_chk(d, in, lane, bad)) { status = ZH_E_EOF; break; } }
        int c;
        if (UNLIKELY(j)) {
          if (d.curr != 0) { status = ZH_E_EOS; break; }
          c = -1;
        } else {
          err = 0;
          c = (int)nb_decode_byte<SP, PROF, false>(K, V, S, d, in, j, bad, err, bseq, P);
          if (UNLIKELY(err | bad)) { status = err ? -(int)err : ZH_E_CORRUPT; break; }
          if (UNLIKELY((!nb_boundary<SP, PROF>(K, V, S, c, bseq, helper_ok, P)))) { status = -24; break; }             // ZPAQHIP_E_HIP
        }

        // ---- PostProcessor.write(c) (PostProcessor.cs:37-86)
        c = (int)uni((uint32_t)c);
        if (LIKELY(pp_state == 1)) {
          if (LIKELY(c >= 0)) out_put(ob, (uint32_t)c, lane);
        } else if (pp_state == 5) {
          int rc;
          bool done = false;
          if (pbwt && c < 0) {                          // end of the block's only segment: the inverse BWT, wave-wide
            uint32_t produced_b = 0;
            bool touched = false;
            if (nb_ibwt(L, S, lane, pb, &produced_b, &touched)) {
              if (lane == 0) sink.len += uni(produced_b);
              nb_wave_sync();
              done = true;
            } else if (touched) {                       // not a BWT after all: the program's own run wants its H as it left it (zeros)
              uint32_t *Hz = reinterpret_cast<uint32_t *>(uni64((uint64_t)(uintptr_t)pz.h));
              const uint64_t hw = (uint64_t)uni(pz.hmask) + 1u;
              for (uint64_t i = lane; i < hw; i += 64) Hz[i] = 0;
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
              __builtin_amdgcn_s_waitcnt(0);
            }
          }
          if (done) rc = 0;
          else if (pnative == ZH_NATIVE_PCOMP_E8E9)
            rc = zh_native_pcomp_e8e9(pa, pb, pc_, pd, pf, (uint32_t)c, (lds_u8_p)lds_off(S.pmreg), pz.mmask, (lds_u32_p)lds_off(S.phreg), pz.hmask, S.pr, &sink, L.budget);
          else if (pskel) {
            ZhPcRegs r{pa, pb, pc_, pd, pf, 0};
            r = zh_pcomp_call(pskel, r, (uint32_t)c, pz.m, pz.mmask, pz.h, pz.hmask, S.pr, &sink, L.budget, S.pimm);
            pa = r.a; pb = r.b; pc_ = r.c; pd = r.d; pf = r.f;
            rc = r.rc;
          }
          else rc = vm_run(pz, (uint32_t)c, &sink, L.budget);
          rc = (int)uni((uint32_t)rc);
          if (rc) { status = rc; break; }
        } else if (pp_state == 0) {
          if (c < 0) { status = ZH_E_PP_EOS; break; }
          pp_state = c + 1;
          if (pp_state > 2) { status = ZH_E_PP_TYPE; break; }
        } else if (pp_state == 2) {
          if (c < 0) { status = ZH_E_PP_EOS; break; }
          pp_hsize = c; pp_state = 3;
        } else if (pp_state == 3) {
          if (c < 0) { status = ZH_E_PP_EOS; break; }
          pp_hsize += c * 256;
          if (pp_hsize < 1) { status = ZH_E_PP_EMPTY; break; }
          pp_len = 0; pp_state = 4;
        } else {                                        // state 4: PCOMP bytes
          if (c < 0) { status = ZH_E_PP_EOS; break; }
          pzbuf[pp_len] = (uint8_t)c;
          if ((int)++pp_len == pp_hsize) {
            nb_wave_sync();
            pz.prog = pzbuf; pz.len = pp_len;
            pz.a = pz.b = pz.c = pz.d = pz.f = 0;
            pnative = p_lds ? uni(zh_native_pcomp_lookup(pzbuf, pp_len)) : 0;
            pskel = pnative ? 0u : uni(zh_pcomp_lookup(pzbuf, pp_len));
            if (lane == 0) zh_pcomp_operands(pskel, pzbuf, S.pimm);
            nb_wave_sync();
            if ((pskel == ZH_PCOMP_BWTRLE_123 || pskel == ZH_PCOMP_BWTRLE_106) && n_seg == 1u && !p_lds) {
              const int nk = pskel == ZH_PCOMP_BWTRLE_123 ? 11 : 9;
              bool same = true;
              for (int k = 0; k < nk; ++k) same = same && uni(S.pimm[k]) == (pskel == ZH_PCOMP_BWTRLE_123 ? kBwt123[k] : kBwt106[k]);
              pbwt = same ? 1u : 0u;
            }
            if (pskel == ZH_PCOMP_LZPRE_108 && !p_lds && pz.m != S.pmreg) {
              bool same = true;
              for (int k = 0; k < 12; ++k) same = same && (k == 5 || uni(S.pimm[k]) == kLzpre108[k]);
              const uint32_t minlen = uni(S.pimm[5]);
              plz = (same && minlen >= 1u) ? minlen + 1u : 0u;
            }
            pp_state = 5;
          }
        }
        { NbProf &P_ = P; (void)P_; }
        if (c < 0) break;
      }

      if (pp_state != 5) out_flush(ob, lane);
      const uint64_t produced = pp_state == 5 ? uni64(sink.len) : ob.len;
      if (!status && produced > b_out_cap) status = ZH_E_OUTPUT_FULL;
      if (status && status != ZH_E_OUTPUT_FULL) failed = 1;
      if (lane == 0) {
        ZhSegResult res;
        res.status = status; res.pp_state = (uint32_t)pp_state | (uint32_t)pp_hsize << 8;
        res.out_off = b_out_off + produced0; res.out_len = produced - produced0;
        res.in_used = in_pos(in) - seg_off;
        L.results[si] = res;
      }
    }
    if (PROF && lane == 0 && L.debug)
      for (int i = 0; i < 16; ++i) atomicAdd((unsigned long long *)&L.debug[i], (unsigned long long)P.prof[i]);
    {                                                   // the helper wave leaves the block; its late commit of the last byte
      ++cmd_seq;                                        // (S.mreg / S.hreg) must be in LDS before this wave zeroes them again
      c2_put0(&S.mb_cmd, cmd_seq << 2 | kC2End);
      (void)c2_wait(&S.mb_ack, cmd_seq << 2 | kC2End);
    }
    nb_wave_sync();
  }
  ++cmd_seq; c2_put0(&S.mb_cmd, cmd_seq << 2 | kC2Exit);
}

}  // namespace

#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "zh_nibble.hip: the helper-wave protocol is written for gfx9-family CUs (shared vector L1, in-order vmcnt)"
#endif
#define ZH_NIBBLE_KERNEL(name, spec, units, prof)                                      \
  extern "C" __global__ __launch_bounds__(128) void name(ZhLaunch L) {                 \
    typedef NbLds<units> Lds;                                                          \
    __shared__ Lds S;                                                                  \
    decode_nibble_body<spec, prof, Lds>(L, S);                                         \
  }
ZH_NIBBLE_KERNEL(zh_decode_nb_min, C2Min, 2, false)
ZH_NIBBLE_KERNEL(zh_decode_nb_mid, C2Mid, 6, false)
ZH_NIBBLE_KERNEL(zh_decode_nb_min_prof, C2Min, 2, true)
ZH_NIBBLE_KERNEL(zh_decode_nb_mid_prof, C2Mid, 6, true)
ZH_NIBBLE_KERNEL(zh_decode_nb_mid8, NbMid8, 7, false)
ZH_NIBBLE_KERNEL(zh_decode_nb_mid8_prof, NbMid8, 7, true)
ZH_NIBBLE_KERNEL(zh_decode_nb_min1, NbMin1, 2, false)
ZH_NIBBLE_KERNEL(zh_decode_nb_min1_prof, NbMin1, 2, true)

// spec: 1 min, 2 mid (zh_chain_spec.h ids; zh_framing.cpp also files the method models of their shapes under them), 5: mid's
// shape with eight mixer inputs (ZH_FAM_CHAIN_MID8), 6: one ICM on min's loop (ZH_FAM_CHAIN_MIN1)
extern "C" hipError_t zh_launch_nibble(const ZhLaunch *L, uint32_t grid, hipStream_t stream, uint32_t spec, int prof) {
  void (*k)(ZhLaunch) = spec == 1 ? (prof ? zh_decode_nb_min_prof : zh_decode_nb_min)
                        : spec == 2 ? (prof ? zh_decode_nb_mid_prof : zh_decode_nb_mid)
                        : spec == 5 ? (prof ? zh_decode_nb_mid8_prof : zh_decode_nb_mid8)
                        : spec == 6 ? (prof ? zh_decode_nb_min1_prof : zh_decode_nb_min1) : nullptr;
  if (!k) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k, dim3(grid), dim3(128), 0, stream, *L);     // decoder wave + helper wave
  return hipGetLastError();
}
extern "C" int zh_nibble_has(uint32_t spec) { return spec == 1 || spec == 2; }
