// zh_c2_common.h — what zh_chain2.hip (decoder wave + helper wave) and zh_chain3.hip (decoder wave, model wave, helper
// wave) share: the compile-time description of the reference's three built-in models (Compressor.cs:48-74), the LDS
// layout of a block, the mailbox primitives of the multi-wave protocols and the helper wavefront (speculative HCOMP
// over the 16 values the current byte can still take).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zh_core.h"
#include "zh_dev.h"
#include "zh_model.h"
#include "zh_zpaql_native.h"

using namespace zhcore;
using namespace zhdev;

#pragma clang diagnostic ignored "-Wint-to-pointer-cast"

namespace {

// ---- compile-time description of the three models (lane = component index) --------------------------------------
struct C2Min {                               // 0 icm 16 ; 1 isse 19 0
  static constexpr uint32_t id = 1, n = 2, depth = 1, final_lane = 1, nmix = 0, hh = 1, hm = 2;
  static constexpr uint64_t icm = 0x1, isse = 0x2;
  static constexpr int helper = 1;            // helper wave: HCOMP, hash rows and mixer rows of the next byte
  static constexpr bool smem_ps = false;
  static constexpr bool guard_rows = true;    // boundary: patch candidate rows behind one wave-level test (same-box A/B: +1.6 %)
  static constexpr int match_lane = -1;
  static constexpr uint32_t mix_lane[2] = {0, 0}, mix_j0[2] = {0, 0}, mix_m[2] = {0, 0};
  static constexpr bool has_tail = false;
};
struct C2Mid {                               // 0 icm ; 1-5 isse ; 6 match ; 7 mix 16 0 7 24 255
  static constexpr uint32_t id = 2, n = 8, depth = 5, final_lane = 7, nmix = 1, hh = 3, hm = 3;
  static constexpr uint64_t icm = 0x01, isse = 0x3e;
  static constexpr int helper = 1;            // helper wave: HCOMP, hash rows and mixer rows of the next byte
  static constexpr bool smem_ps = true;
  static constexpr bool guard_rows = false;   // ... here the test's taken branch costs more than the selects it skips (-1.5 %)
  static constexpr int match_lane = 6;
  static constexpr uint32_t mix_lane[2] = {7, 0}, mix_j0[2] = {0, 0}, mix_m[2] = {7, 0};
  static constexpr bool has_tail = false;
};

typedef uint32_t v2u_ __attribute__((ext_vector_type(2)));
typedef uint32_t v4u_ __attribute__((ext_vector_type(4)));
struct C2Max {                               // Compressor.cs:60-72: 0 const; 1 icm; 2-7 isse; 8 match; 9 icm; 10 isse; 11-14 icm;
                                             // 15 mix 16 0 15 24 255; 16 mix 8 0 16 10 255; 17 mix2 0 15 16 24 0; 18 sse 8 17 32 255;
                                             // 19 mix2 8 17 18 16 255; 20 sse 16 19 32 255; 21 mix2 0 19 20 16 0
  static constexpr uint32_t id = 3, n = 22, depth = 6, final_lane = 21, nmix = 2, hh = 5, hm = 9;
  static constexpr int helper = 2;            // helper wave: HCOMP, and the lines of the next byte's rows touched (no LDS left to stage them)
  static constexpr bool smem_ps = false;
  static constexpr bool guard_rows = false;
  static constexpr uint64_t icm = (1u << 1) | (1u << 9) | (1u << 11) | (1u << 12) | (1u << 13) | (1u << 14), isse = 0xfcu | (1u << 10);
  static constexpr int match_lane = 8;
  static constexpr uint32_t mix_lane[2] = {15, 16}, mix_j0[2] = {0, 0}, mix_m[2] = {15, 16};
  static constexpr bool has_tail = true;
  // tail constants (component arguments of the exact header this kernel is selected for)
  static constexpr int rate17 = 24, rate19 = 16, rate21 = 16;
  static constexpr uint32_t sse_limit = 255 * 4, sse_start = 32;
};

constexpr int kHWords = 256, kMBytes = 4096, kCodeBytes = 2048, kPHWords = 256, kPMBytes = 1024;
constexpr int kEntUnits = 15;                 // ICM / ISSE entry tables of 256 x 8 bytes

constexpr int kSpecUnits = 8, kSpecH = 8, kSpecHMax = 32;      // helper wave staging: ICM/ISSE components and H words per candidate (min / mid)
template <bool TAIL, int HELP, bool MIXLDS = false>
struct alignas(16) C2LdsT {
  static constexpr bool kMixLds = MIXLDS;
  int16_t stretch[32768];                     // at LDS offset 0 of this struct: see lds_stretch()
  uint16_t squash[4096];
  int32_t dt[1024];
  uint32_t pm01[256];                         // MATCH: stretch(dt2k[len]) | stretch(-dt2k[len]) << 16 (Predictor.cs:273-287), [0] = 0
  uint8_t ns[1024];
  v2u_ ent[kEntUnits][256];                  // {A, B}: ISSE {w0, w1} (Predictor.cs:148-152), ICM {cm, stretch(cm >> 8)}
  v4u_ slot[64];                             // per-lane hash row of the current nibble
  v4u_ zrow;                                 // all-zero row read by lanes without a hash table
  v2u_ lent[64];                             // per-lane entry cell of those lanes
  uint32_t lsink[64];                         // per-lane sink for their bit-history writes
  uint32_t sse18[TAIL ? 256 * 32 : 1];        // max: the table of `sse 8 17` (h = 0: row = c8), Predictor.cs:163-164
  uint16_t a19[TAIL ? 256 : 2];               // max: the weights of `mix2 8 17 18`
  // helper wave (HELP): what it prepares for the NEXT byte, for each of the 16 values the current byte can still take
  uint32_t hspec[HELP == 2 ? kSpecHMax : HELP == 1 ? kSpecH : 1][16];      // H[d] after HCOMP(candidate)
  v4u_ rowst[HELP == 1 ? kSpecUnits : 1][3][16];   // the three candidate hash rows of every ICM / ISSE for c8 = 1
  uint32_t mixst[HELP == 1 ? 2 : 1][16][16];       // the mixer rows for c8 = 1
  v4u_ selrow[HELP == 1 ? kSpecUnits : 1][16];     // C2_FINDB: the row Predictor.find settles on for each unit and candidate ...
  uint32_t seloff[HELP == 1 ? kSpecUnits : 1][16]; // ... and its place in the hash table (the helper wave ran the three compares and the victim choice)
  uint32_t mb_nib, mb_byte, mb_ready;         // A -> B: seq << 8 | first nibble / byte;  B -> A: seq whose staging is complete
  uint32_t mb_cmd, mb_ack, mb_model;          // A -> B: block start / end / exit
#ifdef ZH_WITH_CHAIN3
  // three-wave form (zh_chain3.hip): model wave <-> decoder wave
  uint32_t mb_ack2, mb_block;                 // the model wave's acknowledgement; block index of the command
  uint32_t yv;                                // decoder -> model: bit sequence number << 8 | the last 8 decoded bits
  uint32_t pv[2][2][16];                      // model -> decoder: [seq & 1][hypothesis for the bit before][component] = seq << 12 | p & 0xfff
  v4u_ xfer[128];                             // model wave: state of the winning hypothesis -> all lanes ([lane], [64 + lane])
  // MIXLDS (zh_chain3.hip, mid): the mixer's weights come to the decoder wave through LDS, brought by the helper wave.
  // A byte's 255 rows are one block of 256 x m weights (row = context + c8); mixblk holds the block of the byte being
  // decoded (and, in its other half, the one before); mixrow8 holds rows 0-7 of the block each candidate byte leads to
  uint32_t mixrow8[MIXLDS ? 16 : 1][64];
  uint32_t mixblk[MIXLDS ? 2 : 1][MIXLDS ? 1792 : 1];
  uint32_t mb_blk;                            // helper -> decoder: byte sequence number << 1 | half of mixblk the next byte's block is in
#endif
  uint32_t hreg[kHWords];
  uint8_t mreg[kMBytes];
  uint32_t r[256], pr[256];
  uint8_t code[kCodeBytes];
  uint32_t phreg[kPHWords];
  uint8_t pmreg[kPMBytes];
  Vm hz, pz;
  Sink sink;
};
static_assert(sizeof(C2LdsT<true, 2>) <= 163840 && sizeof(C2LdsT<false, 1>) <= 163840 && sizeof(C2LdsT<false, 1, true>) <= 163840, "LDS budget");

typedef __attribute__((address_space(3))) uint8_t *lds_u8_p;
typedef __attribute__((address_space(3))) uint16_t *lds_u16_p;
typedef __attribute__((address_space(3))) int16_t *lds_i16_p;
typedef __attribute__((address_space(3))) uint32_t *lds_u32_p;
typedef uint32_t v2u __attribute__((ext_vector_type(2)));   // native vectors: usable through address_space(3) pointers
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v2u *lds_u2_p;
typedef __attribute__((address_space(3))) v4u *lds_u4_p;
__device__ __forceinline__ uint32_t lds_off(const void *p) { return (uint32_t)(uintptr_t)p; }

__device__ __forceinline__ int med3i(int x, int lo, int hi) { return x < lo ? lo : x > hi ? hi : x; }
__device__ __forceinline__ int shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); }   // row_shr:1, 0 shifted in
__device__ __forceinline__ int dpp_shr(int v, int n) {
  switch (n) {
    case 1: return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    case 2: return __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    case 4: return __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    default: return __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  }
}

// park[LANE] = val (wave-uniform)
template <int LANE>
__device__ __forceinline__ int wrlane_c(int park, int val) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(park) : "s"((int)uni((uint32_t)val)), "n"(LANE));
  return park;
}

// HCOMP's M array in registers (the three models have 2^hm <= 512 bytes): byte i is byte (i & 3) of lane (i >> 2) & 63
// of v[i >> 8].  A `hash` instruction (a = (a + M[b] + 512) * 773) then costs a v_readlane and a bit-field extract
// instead of a dependent LDS round trip; seven of them in a row are mid's whole program.
template <int NREG>
struct MRegs { uint32_t v[NREG]; };
template <int NREG>
struct MByteRef {
  MRegs<NREG> *m;
  uint32_t i;
  __device__ __forceinline__ operator uint32_t() const {
    const uint32_t ln = (i >> 2) & 63u;
    uint32_t w = rdlane(m->v[0], ln);
    if (NREG > 1) { const uint32_t w1 = rdlane(m->v[NREG > 1 ? 1 : 0], ln); w = (i >> 8) & 1u ? w1 : w; }
    return (w >> ((i & 3u) * 8u)) & 255u;
  }
  __device__ __forceinline__ MByteRef &operator=(uint32_t x) {
    const uint32_t ln = (i >> 2) & 63u, sh = (i & 3u) * 8u;
    if (NREG > 1 && ((i >> 8) & 1u)) {
      const uint32_t w = rdlane(m->v[NREG > 1 ? 1 : 0], ln);
      m->v[NREG > 1 ? 1 : 0] = wrlane((w & ~(255u << sh)) | (x & 255u) << sh, ln, m->v[NREG > 1 ? 1 : 0]);
    } else {
      const uint32_t w = rdlane(m->v[0], ln);
      m->v[0] = wrlane((w & ~(255u << sh)) | (x & 255u) << sh, ln, m->v[0]);
    }
    return *this;
  }
};
template <int NREG>
struct MView {
  MRegs<NREG> *m;
  __device__ __forceinline__ MByteRef<NREG> operator[](uint32_t i) const { return MByteRef<NREG>{m, i}; }
};

constexpr uint32_t kOob = 0x80000000u;        // buffer offset beyond every arena slot of this family (< 2 GiB): dropped


// ---- speculative HCOMP (helper wave): the translated program runs once for 16 candidate input bytes, one per lane --------
// M: the committed bytes (LDS) under ONE shadow write per run (the three built-in programs store the input byte once);
// H: local array, written entries shadow the committed words (LDS).  All of a, b, c, d, f are per-lane copies.
struct SpecM {
  lds_u8_p base;
  uint32_t *wi, *wv, *wn;
  struct Ref {
    const SpecM *m; uint32_t i;
    __device__ __forceinline__ operator uint32_t() const { return (*m->wn && *m->wi == i) ? *m->wv : (uint32_t)m->base[i]; }
    __device__ __forceinline__ const Ref &operator=(uint32_t x) const { *m->wi = i; *m->wv = x & 255u; *m->wn = 1u; return *this; }
  };
  __device__ __forceinline__ Ref operator[](uint32_t i) const { return Ref{this, i}; }
};
template <int NH>
struct SpecH {
  lds_u32_p base;
  uint32_t *hs, *wmask;
  struct Ref {
    const SpecH *h; uint32_t d;
    __device__ __forceinline__ operator uint32_t() const { return ((*h->wmask >> d) & 1u) ? h->hs[d] : h->base[d]; }
    __device__ __forceinline__ const Ref &operator=(uint32_t x) const { h->hs[d] = x; *h->wmask |= 1u << d; return *this; }
  };
  __device__ __forceinline__ Ref operator[](uint32_t d) const { return Ref{this, d}; }
};
}  // namespace
template <> struct ZhUniform<SpecM> { static constexpr bool value = false; };
namespace {

__device__ __forceinline__ uint32_t c2_ld(const uint32_t *p) {      // a mailbox word: one value for the wave (scalar control flow)
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// one dword into LDS from lane 0 (all lanes enabled around it).  The LDS unit serves a CU's requests in arrival order and a
// wave issues them in program order, so data written before this flag is seen by whoever sees the flag (as in zh_cm.hip).
__device__ __forceinline__ void c2_put0(const uint32_t *where, uint32_t val) {
  const uint32_t addr = (uint32_t)(uintptr_t)where;
  asm volatile("s_mov_b64 exec, 1\n\tds_write_b32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(addr), "v"(val) : "memory");
}
// Spins (on the scalar unit) until the mailbox word equals `want`; false after kC2Spin polls: nothing may hang the GPU.
__device__ __forceinline__ bool c2_wait(const uint32_t *where, uint32_t want) {
  const uint32_t addr = (uint32_t)(uintptr_t)where;
  uint32_t left = 1u << 26, got, tmp;
  want = (uint32_t)__builtin_amdgcn_readfirstlane((int)want);
  asm volatile(
      ".Lc2w_%=:\n\t"
      "ds_read_b32 %[t], %[a]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_readfirstlane_b32 %[g], %[t]\n\t"
      "s_cmp_eq_u32 %[g], %[w]\n\t"
      "s_cbranch_scc1 .Lc2d_%=\n\t"
      "s_sub_u32 %[l], %[l], 1\n\t"
      "s_cmp_lg_u32 %[l], 0\n\t"
      "s_cbranch_scc1 .Lc2w_%=\n"
      ".Lc2d_%=:"
      : [t] "=&v"(tmp), [g] "=&s"(got), [l] "+s"(left)
      : [a] "v"(addr), [w] "s"(want)
      : "memory", "scc");
  return left != 0;
}
enum : uint32_t { kC2New = 1, kC2End = 2, kC2Exit = 3 };
constexpr uint32_t kC2Spin = 1u << 26;            // bounded waits: nothing may hang the GPU

// k-th ICM / ISSE component of a model (compile-time)
template <class SP>
__device__ constexpr uint32_t c2_unit_comp(uint32_t k) {
  uint32_t seen = 0;
  for (uint32_t i = 0; i < 64; ++i)
    if (((SP::icm | SP::isse) >> i) & 1) { if (seen == k) return i; ++seen; }
  return 0;
}
template <class SP>
__device__ constexpr uint32_t c2_units() {
  uint32_t n = 0;
  for (uint32_t i = 0; i < 64; ++i) n += ((SP::icm | SP::isse) >> i) & 1;
  return n;
}

// Wave B of a two-wave block: while wave A decodes the second nibble of byte s, B runs HCOMP for the 16 bytes s can still
// become, and brings what byte s+1 will start with — h[], the three candidate hash rows of every ICM / ISSE for c8 = 1,
// the mixer row — into LDS for each of them.  When A knows byte s it takes the matching column; B commits that
// candidate's machine state.  B never decides anything: a late B only makes A wait.
template <class SP, class LDS, bool PROF = false>
__device__ void c2_helper(const ZhLaunch &L, LDS &S, uint32_t lane, uint32_t wg_block_slot) {
  uint64_t hb_busy = 0, hb_slack = 0, hb_t0 = 0, hb_t1 = 0;   // PROF: nibble seen -> staging complete; staging complete -> byte seen
  constexpr bool ROWS = SP::helper == 1;                 // min / mid: rows and mixer weights staged in LDS too
  constexpr bool TOUCH = SP::helper == 2;                // max (no LDS left): HCOMP, and the lines of those rows pulled towards L2
  constexpr uint32_t NU = c2_units<SP>(), NH = 1u << SP::hh, RN = ROWS ? 2u : (NU + 3u) / 4u;
  static_assert((!ROWS || NU <= (uint32_t)kSpecUnits) && NU <= 16u && NH <= (uint32_t)(ROWS ? kSpecH : kSpecHMax), "staging size");
  uint8_t *slot_mem = L.arena + (uint64_t)wg_block_slot * L.arena_stride;
  const uint32_t cand = lane & 15u, grp = lane >> 4;
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
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(slot_mem, 0, (int)arena_bytes, 0x00020000);
    // this lane's units: u = grp, grp + 4 (rows of unit u for candidate `cand`)
    uint32_t u_hto[RN], u_mask[RN], u_comp[RN], u_sb2[RN];
    bool u_on[RN];
#pragma unroll
    for (uint32_t r = 0; r < RN; ++r) {
      const uint32_t u = grp + 4u * r;
      u_on[r] = u < NU;
      uint32_t ci = 0;
#pragma unroll
      for (uint32_t k = 0; k < NU; ++k) if (k == u) ci = c2_unit_comp<SP>(k);
      const ZhComp *cp = &M->comp[ci];
      u_comp[r] = ci; u_hto[r] = (uint32_t)cp->ht_off; u_mask[r] = cp->ht_mask; u_sb2[r] = (uint32_t)cp->arg[0] + 2u;
    }
    uint32_t mx_base[2] = {0, 0}, mx_size1[2] = {0, 0};
#pragma unroll
    for (uint32_t q = 0; q < SP::nmix; ++q) {
      const ZhComp &mc = M->comp[SP::mix_lane[q]];
      mx_base[q] = uni((uint32_t)mc.cm_off); mx_size1[q] = uni(mc.cm_mask);
    }
    uint32_t mix_blk = 0, mix_par = 0;                    // MIXLDS: block in S.mixblk[mix_par] (block 0 at the start: the decoder wave fills it)
    uint32_t hb = 0, hc = 0, hd = 0, hf = 0;              // committed HCOMP registers (A is the input at every run; M and H: S.mreg / S.hreg, zeroed by A)
    c2_put0(&S.mb_ack, cmd);
    uint32_t seq = 1;
    bool alive = true;
    uint32_t touch_a = 0, touch_b[6] = {0, 0, 0, 0, 0, 0};   // results of the line touches (dropped; kept so that nothing waits for them early)
    while (alive) {
      // ---- the first nibble of byte #seq
      uint32_t v;
      sp = 0;
      while ((((v = c2_ld(&S.mb_nib)) >> 8) ^ seq) & 0xFFFFFFu) {   // (this one is on the clock: the decoder wave has 4 bits to go; 24-bit sequence numbers)
        if (c2_ld(&S.mb_cmd) != seen_cmd || ++sp > kC2Spin) { alive = false; break; }
      }
      if (!alive) break;
      if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hb_t0)::"memory"); }
      const uint32_t x = (v & 15u) << 4 | cand;
      // ---- HCOMP for the candidates (all 64 lanes run it: four copies of each candidate)
      uint32_t sa = x, sb = hb, sc = hc, sd = hd, sf = hf;
      uint32_t wi = 0, wv = 0, wn = 0, hs[NH], wmask = 0;
#pragma unroll
      for (uint32_t d = 0; d < NH; ++d) hs[d] = 0;
      const SpecM sm{(lds_u8_p)lds_off(S.mreg), &wi, &wv, &wn};
      const SpecH<NH> sh{(lds_u32_p)lds_off(S.hreg), hs, &wmask};
      if constexpr (SP::id == 1) (void)zh_native_hcomp_min(sa, sb, sc, sd, sf, x, sm, (1u << SP::hm) - 1u, sh, NH - 1u, S.r, (Sink *)nullptr, L.budget);
      else if constexpr (SP::id == 3) (void)zh_native_hcomp_max(sa, sb, sc, sd, sf, x, sm, (1u << SP::hm) - 1u, sh, NH - 1u, S.r, (Sink *)nullptr, L.budget);
      else (void)zh_native_hcomp_mid(sa, sb, sc, sd, sf, x, sm, (1u << SP::hm) - 1u, sh, NH - 1u, S.r, (Sink *)nullptr, L.budget);
      if (grp == 0) {
#pragma unroll
        for (uint32_t d = 0; d < NH; ++d) S.hspec[d][cand] = (uint32_t)sh[d];
      }
      if constexpr (ROWS) {
      // ---- rows of the first nibble of the next byte (c8 = 1): Predictor.find's three candidates per component
      v4u rr[2][3];
      uint32_t cxts[2] = {0, 0};
#pragma unroll
      for (uint32_t r = 0; r < 2; ++r) {
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
      // ---- mixer rows for c8 = 1: weights grp, grp+4, grp+8, grp+12 of the row of candidate `cand`
      uint32_t mwv[2][4];
      v4u mrow8[4];
      if constexpr (LDS::kMixLds) {             // rows 0-7 of the candidate's block: 8 x m x 4 <= 256 bytes, 16 chunks of 16
        uint32_t hq = 0;
#pragma unroll
        for (uint32_t d = 0; d < NH; ++d) if ((SP::mix_lane[0] & (NH - 1u)) == d) hq = (uint32_t)sh[d];
        const uint32_t b0 = mx_base[0] + (hq & mx_size1[0] & ~255u) * (SP::mix_m[0] * 4u);
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) mrow8[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, b0 + (grp + 4u * t) * 16u, 0, 0);
      }
#pragma unroll
      for (uint32_t q = 0; q < (LDS::kMixLds ? 0u : SP::nmix); ++q) {
        uint32_t hq = 0;
#pragma unroll
        for (uint32_t d = 0; d < NH; ++d) if ((SP::mix_lane[q] & (NH - 1u)) == d) hq = (uint32_t)sh[d];
        const ZhComp &mc = M->comp[SP::mix_lane[q]];
        const uint32_t row = mx_base[q] + ((hq + (1u & (uint32_t)mc.arg[4])) & mx_size1[q]) * (SP::mix_m[q] * 4u);
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) {
          const uint32_t jj = grp + 4u * t;
          mwv[q][t] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, jj < SP::mix_m[q] ? row + jj * 4u : kOob, 0, 0);
        }
      }
#pragma unroll
      for (uint32_t r = 0; r < 2; ++r) {
        if (u_on[r]) {
#pragma unroll
          for (uint32_t k = 0; k < 3; ++k) *(lds_u4_p)lds_off(&S.rowst[grp + 4u * r][k][cand]) = rr[r][k];
#ifdef C2_FINDB
          if (C2_FINDB) {
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
            *(lds_u4_p)lds_off(&S.selrow[grp + 4u * r][cand]) = row;
            S.seloff[grp + 4u * r][cand] = sel;
          }
#endif
        }
      }
#pragma unroll
      for (uint32_t q = 0; q < (LDS::kMixLds ? 0u : SP::nmix); ++q)
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) S.mixst[q][cand][grp + 4u * t] = mwv[q][t];
      if constexpr (LDS::kMixLds) {
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) *(lds_u4_p)lds_off(&S.mixrow8[cand][(grp + 4u * t) * 4u]) = mrow8[t];
      }
      }
      asm volatile("" ::: "memory");
      c2_put0(&S.mb_ready, seq);
      if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hb_t1)::"memory"); hb_busy += hb_t1 - hb_t0; }
      if constexpr (TOUCH) {
        // The decoder wave will request the candidate's rows itself once the byte is known; one dword per (candidate,
        // component) now brings the 64-byte line that holds all three probes of Predictor.find (h0, h0^16, h0^32) out of
        // HBM, likewise the mixer row and the `sse 16` row pair.  Nothing is kept: the values are dropped.
        uint32_t t[RN + 2];
#pragma unroll
        for (uint32_t r = 0; r < RN; ++r) {
          const uint32_t hval = S.hspec[u_comp[r] & (NH - 1u)][cand];
          const uint32_t h0 = ((hval + 16u) * 16u) & (u_mask[r] - 15u);
          t[r] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, u_on[r] ? u_hto[r] + h0 : kOob, 0, 0);
        }
        {
          const uint32_t hq = S.hspec[SP::mix_lane[0] & (NH - 1u)][cand];
          const uint32_t row = mx_base[0] + ((hq + 1u) & mx_size1[0]) * (SP::mix_m[0] * 4u);
          t[RN] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, grp < 2 ? row + grp * (SP::mix_m[0] * 4u - 4u) : kOob, 0, 0);
          const ZhComp &sc20 = M->comp[20];
          const uint32_t srow = ((S.hspec[20][cand] * 32u) & sc20.cm_mask) * 4u + (uint32_t)sc20.cm_off;
          t[RN + 1] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, SP::has_tail ? srow + grp * 64u : kOob, 0, 0);
        }
#pragma unroll
        for (uint32_t r = 0; r < RN + 2; ++r) asm volatile("" ::"v"(t[r]));
      }
      // ---- the byte: commit its candidate
      sp = 0;
      while ((((v = c2_ld(&S.mb_byte)) >> 8) ^ seq) & 0xFFFFFFu) {
        if (c2_ld(&S.mb_cmd) != seen_cmd || ++sp > kC2Spin) { alive = false; break; }
        __builtin_amdgcn_s_sleep(1);                     // (nothing to do until the byte is known: poll gently)
      }
      if (!alive) break;
      if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hb_t0)::"memory"); hb_slack += hb_t0 - hb_t1; }
      const uint32_t lo = v & 15u;
      hb = rdlane(sb, lo); hc = rdlane(sc, lo); hd = rdlane(sd, lo); hf = rdlane(sf, lo);
      const uint32_t cwi = rdlane(wi, lo), cwv = rdlane(wv, lo), cwn = rdlane(wn, lo);
      if (cwn && lane == 0) S.mreg[cwi] = (uint8_t)cwv;
      if (lane < NH) { const uint32_t hv_ = S.hspec[lane][lo]; S.hreg[lane] = hv_; }
      if constexpr (LDS::kMixLds) {
        // the block of mixer rows the NEXT byte uses (its context is known now) -> the other half of mixblk; a byte
        // that keeps the context keeps the block, which the decoder wave has kept up to date
        const uint32_t blk = S.hspec[SP::mix_lane[0] & (NH - 1u)][lo] & mx_size1[0] & ~255u;
        if (blk != mix_blk) {
          mix_blk = blk; mix_par ^= 1u;
          const uint32_t b0 = mx_base[0] + blk * (SP::mix_m[0] * 4u);
          v4u chunk[7];
#pragma unroll
          for (uint32_t t = 0; t < 7; ++t) chunk[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, b0 + (lane + 64u * t) * 16u, 0, 0);
#pragma unroll
          for (uint32_t t = 0; t < 7; ++t) *(lds_u4_p)lds_off(&S.mixblk[mix_par][(lane + 64u * t) * 4u]) = chunk[t];
        }
        asm volatile("" ::: "memory");
        c2_put0(&S.mb_blk, seq << 1 | mix_par);
      }
      ++seq;
    }
  }
}

}  // namespace
