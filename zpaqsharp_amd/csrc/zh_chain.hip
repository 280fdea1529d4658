// zh_chain.hip — lane-per-component decode kernel: component chains of up to 64
// components (ICM / ISSE / MATCH / MIX / MIX2 / SSE / AVG / CM / CONST), i.e. the
// reference's built-in min / mid / max models (BASELINE configs 3-5).
//
// One wavefront owns one block; lane i owns component i of the model
// (Predictor.cs:245-475 runs them one after another, here they run side by side):
//
//  * All table traffic of a bit is issued by all components at once.  The pieces a
//    component can touch during a NIBBLE are fetched once per nibble into a 64-byte
//    per-lane LDS slot: the 16-byte hash row of an ICM/ISSE (Predictor.find,
//    Predictor.cs:550-567 — the three candidate rows are probed in parallel) or the
//    64-byte line of a CM.  Bit-history -> probability / weight tables of ICM and
//    ISSE (1-2 KiB each) live in LDS for the whole block.
//  * The inter-component dependencies (ISSE/AVG/MIX2/SSE/MIX inputs) are resolved
//    level by level; the host computes each component's level.  Cross-lane operands
//    travel by ds_bpermute; a MIX is a wave reduction over its input lanes (DPP), and
//    each input lane owns, loads, trains and stores ITS weight of the mixer row
//    (Predictor.cs:302-316, :427-439) — coalesced row traffic, no serial loop.
//  * The final probability goes through v_readlane to the scalar unit, which runs
//    the arithmetic decoder step shared with zh_cm.hip (Decoder.cs:136-158).
//  * MATCH verifies a candidate match with all 64 lanes comparing history bytes and
//    one ballot (Predictor.cs:403-405 is a serial loop of up to 255 steps).
//  * HCOMP / PCOMP run on the scalar core (zh_core.h) with H and M in LDS when small.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zh_core.h"
#include "zh_dev.h"
#include "zh_model.h"
#include "zh_zpaql_native.h"
#include "zh_zpaql_pcomp.h"
#define ZH_CHAIN_SPEC_DEVICE 1
#include "zh_chain_spec.h"

using namespace zhcore;
using namespace zhdev;

// LDS byte offsets are turned into address_space(3) pointers (32-bit on the device); the host pass of hipcc, which
// never runs this code, sees 64-bit pointers there and would warn.
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"

namespace {

constexpr int kSmallWords = 16384;        // LDS pool for ICM (256 words) / ISSE (512 words) tables
constexpr int kHWords = 256;              // HCOMP H kept in LDS when 2^hh <= 256
constexpr int kMBytes = 4096;             // HCOMP M kept in LDS when 2^hm <= 4096
constexpr int kMaxMix = 4;
constexpr int kCodeBytes = 2048;          // HCOMP program window kept in LDS when it fits
constexpr int kPHWords = 256;             // PCOMP H in LDS when 2^ph <= 256
constexpr int kPMBytes = 1024;            // PCOMP M in LDS when 2^pm <= 1024

struct alignas(16) ChainLds {
  ZhTables t;
  uint32_t small[kSmallWords];
  uint8_t slot[64][64];                   // per-lane nibble cache (hash row or CM line)
  uint32_t sserow[64];                    // specialised kernels: the 32-entry table row of up to two SSE components for this bit
  uint32_t dummy[64];                     // per-lane sink for the stores of lanes a branch-free step does not concern
  uint32_t hreg[kHWords];
  uint8_t mreg[kMBytes];
  uint32_t r[256], pr[256];
  uint8_t code[kCodeBytes];
  uint32_t phreg[kPHWords];
  uint8_t pmreg[kPMBytes];
  Vm hz, pz;
  Sink sink;
  alignas(16) uint32_t pimm[64];          // operands of a structurally matched PCOMP (zh_zpaql_pcomp.h)
};
static_assert(sizeof(ChainLds) <= 163840, "LDS budget");

// Products of a 20-bit weight or error term and a 12-bit stretched prediction use the full-rate 24-bit multiplier
// (__mul24 -> v_mul_i32_i24); both operands are bounded by clamp512k / clamp2k / squash, so the results are exact.
__device__ __forceinline__ int clampk(int x, int lo, int hi) { return x < lo ? lo : x > hi ? hi : x; }
// Typed LDS accesses by LDS byte offset: lets one ds_* instruction serve lanes of different component types
// (the address is selected per lane) instead of one divergent branch per type.
typedef __attribute__((address_space(3))) uint8_t *lds_u8_p;
typedef __attribute__((address_space(3))) uint16_t *lds_u16_p;
typedef __attribute__((address_space(3))) uint32_t *lds_u32_p;
__device__ __forceinline__ uint32_t lds_off(const void *p) { return (uint32_t)(uintptr_t)p; }

// Per-lane view of one component (Component.cs:18-57 + its header arguments).
struct ZhSpec_generic {                    // run-time everything (any header the host accepts for this family)
  static constexpr uint32_t id = 0u, types = 0x3ffu, nmix = 0u;
};

struct Lane {
  uint32_t type, a0, a1, a2, a3, a4, level;
  uint32_t cmo, hto;                      // arena offsets of the tables (the arena is < 4 GiB for this family);
                                          // pointers are formed as slot_mem + offset so they stay GLOBAL, not flat
  uint32_t cm_mask, ht_mask;
  uint32_t sbase;                         // word offset of the ICM/ISSE table in S.small
  uint32_t limit, cxt, a, b, c;           // Component state
  uint32_t h;                             // h[i]
  int p, pj, pk;                          // own prediction and the inputs it was computed from
  int w0, w1;                             // ISSE weights / MIX2 weight / SSE entries of this bit
  uint32_t mbyte, mcur;                   // MATCH: predicted byte, byte being assembled
  int mw[kMaxMix];                        // weights of this lane in each mixer row
  uint32_t memb;                          // bit q set: this lane feeds mixer q
  bool rowvalid;                          // slot holds a row/line that must be written back
};

template <bool PROF, class SP, bool PCALL>
__device__ __forceinline__ void decode_chain_body(const ZhLaunch &L, ChainLds &S) {
  constexpr bool kSpec = SP::id != 0;
  constexpr bool kDefer = SP::id == 1 || SP::id == 2;   // deferred mixer-weight store (measured: helps min / mid, not max)
#define ZH_HAS(t) ((SP::types >> (t)) & 1u)
  uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tprev = 0;
  const uint32_t lane = threadIdx.x;

  {  // model-independent tables -> LDS
    const uint4 *src = reinterpret_cast<const uint4 *>(L.tables);
    uint4 *dst = reinterpret_cast<uint4 *>(&S.t);
    for (uint32_t i = lane; i < sizeof(ZhTables) / 16; i += 64) dst[i] = src[i];
  }
  __syncthreads();

  uint8_t *slot_mem = L.arena + (uint64_t)blockIdx.x * L.arena_stride;
  uint8_t *myslot = &S.slot[lane][0];

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
    const uint32_t n = uni(M->n), depth = uni(M->depth);
    const uint32_t hh = uni(M->hh), hmb = uni(M->hm);

    // ---- Predictor.init (Predictor.cs:82-171): arena tables by all lanes, component by component
    for (uint32_t i = 0; i < n; ++i) {
      const ZhComp &cp = M->comp[i];
      const uint32_t type = uni(cp.type);
      uint8_t *cm = slot_mem + uni64(cp.cm_off), *ht = slot_mem + uni64(cp.ht_off);
      const uint64_t cmb = uni64(cp.cm_bytes), htb = uni64(cp.ht_bytes);
      uint4 pat = make_uint4(0, 0, 0, 0);
      bool fill_cm = false;
      if (type == ZH_CM) { pat = make_uint4(0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u); fill_cm = true; }
      else if (type == ZH_MATCH) fill_cm = true;
      else if (type == ZH_MIX2) { pat = make_uint4(0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u); fill_cm = true; }
      else if (type == ZH_MIX) { const uint32_t w = 65536u / uni(cp.arg[2]); pat = make_uint4(w, w, w, w); fill_cm = true; }
      if (fill_cm) { uint4 *q = reinterpret_cast<uint4 *>(cm); for (uint64_t k = lane; k < cmb / 16; k += 64) q[k] = pat; }
      if (type == ZH_SSE) {                              // squash((j&31)*64-992)<<17 | start, period 32 entries
        const uint32_t start = uni(cp.arg[2]);
        uint4 *q = reinterpret_cast<uint4 *>(cm);
        for (uint64_t k = lane; k < cmb / 16; k += 64) {
          const uint32_t j = (uint32_t)(k * 4) & 31;
          uint4 v;
          v.x = (uint32_t)S.t.squash[(j + 0) * 64 - 992 + 2048] << 17 | start;
          v.y = (uint32_t)S.t.squash[(j + 1) * 64 - 992 + 2048] << 17 | start;
          v.z = (uint32_t)S.t.squash[(j + 2) * 64 - 992 + 2048] << 17 | start;
          v.w = (uint32_t)S.t.squash[(j + 3) * 64 - 992 + 2048] << 17 | start;
          q[k] = v;
        }
      }
      if (type == ZH_ICM || type == ZH_ISSE || type == ZH_MATCH) {
        uint4 *q = reinterpret_cast<uint4 *>(ht);
        for (uint64_t k = lane; k < htb / 16; k += 64) q[k] = make_uint4(0, 0, 0, 0);
      }
    }
    {  // VM memories: arena tail (H, M, PCOMP H/M/program) zeroed; LDS copies zeroed
      const uint64_t h_off = uni64(M->h_off), tail = uni64(M->arena_bytes) - h_off;
      uint4 *z = reinterpret_cast<uint4 *>(slot_mem + h_off);
      for (uint64_t i = lane; i < tail / 16; i += 64) z[i] = make_uint4(0, 0, 0, 0);
      for (uint32_t i = lane; i < 256; i += 64) { S.r[i] = 0; S.pr[i] = 0; S.hreg[i] = 0; }
      for (uint32_t i = lane; i < kMBytes / 4; i += 64) reinterpret_cast<uint32_t *>(S.mreg)[i] = 0;
      for (uint32_t i = lane; i < kPHWords; i += 64) S.phreg[i] = 0;
      for (uint32_t i = lane; i < kPMBytes / 4; i += 64) reinterpret_cast<uint32_t *>(S.pmreg)[i] = 0;
    }

    __syncthreads();
    // ---- this lane's component
    Lane me;
    {
      const bool act = lane < n;
      const ZhComp *cp = &M->comp[act ? lane : 0];
      me.type = act ? cp->type : (uint32_t)ZH_NONE;
      me.a0 = cp->arg[0]; me.a1 = cp->arg[1]; me.a2 = cp->arg[2]; me.a3 = cp->arg[3]; me.a4 = cp->arg[4];
      me.level = cp->level;
      me.cmo = (uint32_t)cp->cm_off; me.hto = (uint32_t)cp->ht_off;
      me.cm_mask = cp->cm_mask; me.ht_mask = cp->ht_mask;
      me.sbase = (uint32_t)cp->small_unit * 256u;
      me.limit = me.cxt = me.a = me.b = me.c = 0; me.h = 0;
      me.p = me.pj = me.pk = 0; me.w0 = me.w1 = 0; me.mbyte = me.mcur = 0; me.memb = 0; me.rowvalid = false;
      for (int q = 0; q < kMaxMix; ++q) me.mw[q] = 0;
      switch (me.type) {                                 // scalar parts of Predictor.init
        case ZH_CONS: me.p = ((int)me.a0 - 128) * 4; break;
        case ZH_CM: me.limit = me.a1 * 4; break;
        case ZH_ICM:
          me.limit = 1023;
          for (uint32_t j = 0; j < 256; ++j) {
            const uint32_t n0 = S.t.ns[j * 4 + 2], n1 = S.t.ns[j * 4 + 3];
            S.small[me.sbase + j] = ((n1 * 2 + 1) << 22) / (n0 + n1 + 1);                 // StateTable.cminit
          }
          break;
        case ZH_ISSE:
          for (uint32_t j = 0; j < 256; ++j) {
            const uint32_t n0 = S.t.ns[j * 4 + 2], n1 = S.t.ns[j * 4 + 3];
            const uint32_t ci = ((n1 * 2 + 1) << 22) / (n0 + n1 + 1);
            S.small[me.sbase + 2 * j] = 1u << 15;
            S.small[me.sbase + 2 * j + 1] = (uint32_t)clamp512k(S.t.stretch[ci >> 8] * 1024);
          }
          break;
        case ZH_MATCH: (slot_mem + me.hto)[0] = 1; break;
        case ZH_MIX2: case ZH_MIX: me.c = me.cm_mask + 1; break;
        case ZH_SSE: me.limit = me.a3 * 4; break;
        default: break;
      }
    }
    // mixers (wave-uniform table in LDS) and which of them each lane feeds
    uint32_t nmix = 0;
    uint32_t mx_lane[kMaxMix] = {0, 0, 0, 0}, mx_j0[kMaxMix] = {0, 0, 0, 0}, mx_m[kMaxMix] = {0, 0, 0, 0}, mx_lv[kMaxMix] = {0, 0, 0, 0};
    uint32_t mx_off[kMaxMix] = {0, 0, 0, 0};            // arena offsets of the mixer tables (pointers formed at use stay GLOBAL)
    {
      const uint64_t mm = __ballot(me.type == ZH_MIX);
#pragma unroll
      for (int q = 0; q < kMaxMix; ++q) {
        uint64_t rest = mm;
        for (int k = 0; k < q; ++k) rest &= rest - 1;             // drop the q lowest set bits
        if (!rest) break;
        const uint32_t ml = (uint32_t)__builtin_ctzll(rest);
        mx_lane[q] = ml; mx_j0[q] = rdlane(me.a1, ml); mx_m[q] = rdlane(me.a2, ml); mx_lv[q] = rdlane(me.level, ml);
        mx_off[q] = rdlane(me.cmo, ml);
        if (lane >= mx_j0[q] && lane < mx_j0[q] + mx_m[q]) me.memb |= 1u << q;
        nmix = (uint32_t)q + 1;
      }
    }
    // Level descriptors, one per level, held in lane `level` of lvl_desc:
    //   bits 0-6  : the lane of the level's only non-MIX component, 64 = several, 65 = none
    //   bits 8-11 : its type      bits 12-14 : 1 + index of the mixer evaluated at this level (0 = none)
    //   bits 16-21: first input   bits 24-29 : second input
    uint32_t lvl_desc = 65;
    for (uint32_t lv = 1; lv <= depth && lv < 64; ++lv) {
      const uint64_t at = __ballot(me.level == lv && me.type != ZH_MIX && lane < n);
      uint32_t dsc = at ? 64u : 65u;
      if (__builtin_popcountll(at) == 1) {
        const uint32_t cl = (uint32_t)__builtin_ctzll(at);
        const uint32_t sj = rdlane(me.type == ZH_AVG ? me.a0 : me.a1, cl), sk = rdlane(me.type == ZH_AVG ? me.a1 : me.a2, cl);
        dsc = cl | rdlane(me.type, cl) << 8 | (sj & 63) << 16 | (sk & 63) << 24;
      }
#pragma unroll
      for (int q = 0; q < kMaxMix; ++q)
        if ((uint32_t)q < nmix && mx_lv[q] == lv && !(dsc >> 12 & 7)) dsc |= (uint32_t)(q + 1) << 12;
      // a second mixer on the same level falls back to the generic test below
      uint32_t cnt = 0;
#pragma unroll
      for (int q = 0; q < kMaxMix; ++q) cnt += (uint32_t)q < nmix && mx_lv[q] == lv;
      if (cnt > 1) dsc |= 7u << 12;
      if (lane == lv) lvl_desc = dsc;
    }
    __syncthreads();

    // per-lane constants of the branch-free ICM / ISSE steps
    const bool is_icm = me.type == ZH_ICM, is_isse = me.type == ZH_ISSE, is_ii = is_icm || is_isse, is_match = me.type == ZH_MATCH;
    const uint32_t ii_tab = lds_off(&S.small[0]) + me.sbase * 4;      // table of this lane (others: the pool's start, read only)
    const uint32_t ii_sh = is_isse ? 3u : 2u;                         // 8-byte weight pairs / 4-byte probabilities
    const uint32_t slot_off = lds_off(myslot), dummy_off = lds_off(&S.dummy[lane]), ns_off = lds_off(&S.t.ns[0]);
    int pm0 = 0, pm1 = 0;                                             // MATCH: stretch of +-dt2k[len] for this byte

    // HCOMP machine (ZPAQL.cs:1010-1026): H and M in LDS when they fit
    Vm &hz = S.hz;
    hz.a = hz.b = hz.c = hz.d = hz.f = 0;
    hz.len = uni(M->hcomp_len);
    {
      const uint8_t *gcode = L.code + uni(M->code_off);            // padded window: PAD | program | PAD
      const uint32_t win = hz.len + 2 * ZH_CODE_PAD;
      if (win <= (uint32_t)kCodeBytes) {
        for (uint32_t i = lane; i < win; i += 64) S.code[i] = gcode[i];
        hz.prog = S.code + ZH_CODE_PAD;
      } else hz.prog = gcode + ZH_CODE_PAD;
    }
    hz.hmask = (uint32_t)((1ull << hh) - 1); hz.mmask = (uint32_t)((1ull << hmb) - 1);
    hz.h = (1u << hh) <= (uint32_t)kHWords && hh < 31 ? S.hreg : reinterpret_cast<uint32_t *>(slot_mem + uni64(M->h_off));
    hz.m = hmb < 31 && (1u << hmb) <= (uint32_t)kMBytes ? S.mreg : slot_mem + uni64(M->m_off);
    hz.r = S.r;
    const uint32_t *Hptr = hz.h;
    const uint32_t hmask = hz.hmask;
    const bool h_lds = hz.h == S.hreg && hz.m == S.mreg;
    const uint32_t hnative = h_lds ? (uni(M->kind) >> 8) & 255 : 0;   // ahead-of-time translated HCOMP, if known
    uint32_t ha = 0, hb = 0, hc = 0, hd = 0, hf = 0;                  // HCOMP registers A B C D F (wave-uniform)
    const lds_u8_p lds_m = (lds_u8_p)lds_off(S.mreg);                 // address-space-qualified views for the native programs:
    const lds_u32_p lds_h = (lds_u32_p)lds_off(S.hreg);               // ds_* instead of flat_* accesses

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
    uint32_t pnative = 0;                                 // set when the loaded PCOMP is a known program
    uint32_t pskel = 0;                                   // ... or has the structure of one of the reference's generated ones (zh_zpaql_pcomp.h)
    uint32_t pa = 0, pb = 0, pc_ = 0, pd = 0, pf = 0;
    uint8_t *pzbuf = slot_mem + uni64(M->pz_off) + ZH_CODE_PAD;

    Dec d;
    d.low = 1; d.high = 0xFFFFFFFFu; d.curr = 0;
    OutBuf ob;
    ob.base = L.out + b_out_off; ob.cap = b_out_cap; ob.len = 0; ob.stored = 0; ob.word = 0; ob.park = 0;
    out_room(ob);
    Sink &sink = S.sink;
    sink.out = ob.base; sink.cap = ob.cap; sink.len = 0;
    __syncthreads();
    InBuf in;
    in.stream = L.in; in.total = L.in_total; in.cbase = 0; in.k = 0; in.avail = 0; in.cur = 0;

    uint32_t c8 = 1, hmap4 = 1;                        // Predictor.cs:20-21

    // Start of a nibble (c8 == 1 or 16 <= c8 < 32): write the old row/line back, fetch the new one.
    // Split in two so that other global traffic can be put in flight between issue and use.
    // The three candidate rows of an ICM / ISSE (Predictor.find, Predictor.cs:550-567) are requested by EVERY lane
    // (other lanes read the first bytes of the arena slot and ignore them): no divergent region, so the probes leave
    // back to back and are waited for once.  A CM's 64-byte line takes the same path through four registers.
    uint4 nr0 = make_uint4(0, 0, 0, 0), nr1 = nr0, nr2 = nr0, nr3 = nr0;   // rows in flight across the byte boundary
    uint32_t nh0 = 0;
    auto rows_issue = [&](uint4 &r0, uint4 &r1, uint4 &r2, uint4 &r3, uint32_t &h0) __attribute__((always_inline)) {
      if (is_ii && me.rowvalid) *reinterpret_cast<uint4 *>(slot_mem + me.hto + me.c) = *reinterpret_cast<const uint4 *>(myslot);
      const uint32_t cxt = me.h + 16u * c8;
      h0 = is_ii ? (cxt * 16u) & (me.ht_mask - 15u) : 0u;
      const uint8_t *tb = slot_mem + (is_ii ? me.hto : 0u);
      r0 = *reinterpret_cast<const uint4 *>(tb + h0);
      r1 = *reinterpret_cast<const uint4 *>(tb + (h0 ^ 16));
      r2 = *reinterpret_cast<const uint4 *>(tb + (h0 ^ 32));
      if (ZH_HAS(ZH_CM) && me.type == ZH_CM) {
        uint4 *g = reinterpret_cast<uint4 *>(slot_mem + me.cmo) + (size_t)me.c * 4;
        const uint4 *l = reinterpret_cast<const uint4 *>(myslot);
        if (me.rowvalid) { g[0] = l[0]; g[1] = l[1]; g[2] = l[2]; g[3] = l[3]; }
        me.c = ((me.h ^ hmap4) & me.cm_mask) >> 4;     // 16-entry line of this nibble
        g = reinterpret_cast<uint4 *>(slot_mem + me.cmo) + (size_t)me.c * 4;
        r0 = g[0]; r1 = g[1]; r2 = g[2]; r3 = g[3];
      }
    };
    auto rows_finish = [&](const uint4 &r0, const uint4 &r1, const uint4 &r2, const uint4 &r3, uint32_t h0) __attribute__((always_inline)) {
      const uint32_t chk = ((me.h + 16u * c8) >> (me.a0 + 2)) & 255;
      const bool m0 = (r0.x & 255) == chk, m1 = (r1.x & 255) == chk, m2 = (r2.x & 255) == chk;
      const uint32_t p0 = (r0.x >> 8) & 255, p1 = (r1.x >> 8) & 255, p2 = (r2.x >> 8) & 255;
      const uint32_t victim = (p0 <= p1 && p0 <= p2) ? h0 : p1 < p2 ? h0 ^ 16 : h0 ^ 32;
      const uint32_t sel = m0 ? h0 : m1 ? h0 ^ 16 : m2 ? h0 ^ 32 : victim;
      const uint4 fresh = make_uint4(chk, 0, 0, 0);
      const uint4 row = m0 ? r0 : m1 ? r1 : m2 ? r2 : fresh;
      if (is_ii) {
        *reinterpret_cast<uint4 *>(myslot) = row;
        me.c = sel;
        me.rowvalid = true;
      }
      if (ZH_HAS(ZH_CM) && me.type == ZH_CM) {
        uint4 *l = reinterpret_cast<uint4 *>(myslot);
        l[0] = r0; l[1] = r1; l[2] = r2; l[3] = r3;
        me.rowvalid = true;
      }
    };
    auto nibble_issue = [&]() __attribute__((always_inline)) { rows_issue(nr0, nr1, nr2, nr3, nh0); };
    auto nibble_finish = [&]() __attribute__((always_inline)) { rows_finish(nr0, nr1, nr2, nr3, nh0); };
    // issue and use back to back (block start, middle of a byte): the rows live in temporaries, not in the
    // loop-carried registers of the split form
    auto nibble_refresh = [&]() __attribute__((always_inline)) {
      uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0, a2 = a0, a3 = a0;
      uint32_t ah = 0;
      rows_issue(a0, a1, a2, a3, ah);
      rows_finish(a0, a1, a2, a3, ah);
    };

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
      if (s == 0) nibble_refresh();                    // first nibble of the block (h[] = 0)

      for (;;) {                                       // one decoded byte per iteration
        // ---- Decoder.decompress prologue (Decoder.cs:36-45)
        if (UNLIKELY(d.curr == 0)) {
          uint32_t cu = 0;
          for (int i = 0; i < 4; ++i) cu = cu << 8 | (uint32_t)in_get(in, lane);
          d.curr = uni(cu);
        }
        uint32_t bad = 0, rn, j = 0, err = 0;
        d.low = uni(d.low); d.high = uni(d.high); d.curr = uni(d.curr);
        ZH_DEC_STEP(d, 0u, j, bad, rn);                // EOS flag: p = 0
        if (UNLIKELY(bad)) { status = ZH_E_CORRUPT; break; }
        if (UNLIKELY(rn)) { if (dec_renorm(d, in, lane)) { status = ZH_E_EOF; break; } }
        int c;
        if (UNLIKELY(j)) {
          if (d.curr != 0) { status = ZH_E_EOS; break; }
          c = -1;
        } else {
          uint32_t prow[kMaxMix] = {~0u, ~0u, ~0u, ~0u};          // row whose weight (pmw) of the previous bit is not stored yet
          int pmw[kMaxMix] = {0, 0, 0, 0};
          for (int bit = 0; bit < 8; ++bit) {
            if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }
            c8 = uni(c8); hmap4 = uni(hmap4);
            const uint32_t hm15 = hmap4 & 15;
            // ================= predict, level 0 (Predictor.cs:259-343) =================
            uint32_t rows[kMaxMix] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t q = 0; q < (uint32_t)kMaxMix; ++q) {      // mixer rows: every input lane loads its own weight
              if (q >= (kSpec ? SP::nmix : nmix)) break;
              const uint32_t rowv = ((me.h + (c8 & me.a4)) & (me.c - 1)) * mx_m[q];    // valid in the mixer lane
              rows[q] = rdlane(rowv, mx_lane[q]);
              uint32_t *mrow = reinterpret_cast<uint32_t *>(slot_mem + mx_off[q]) + (lane - mx_j0[q]);
              if constexpr (kDefer) {
                // The weight trained at the previous bit is stored only now, AFTER this bit's load has been issued: the
                // store's round trip then overlaps a whole bit instead of being waited for at the top of the loop.
                // (Same row twice in a row — a mixer that does not select by c8 — keeps program order.)
                const bool same = prow[q] == rows[q];          // prow == ~0u: nothing pending
                if (same && (me.memb >> q & 1)) mrow[prow[q]] = (uint32_t)pmw[q];
                if (me.memb >> q & 1) me.mw[q] = (int)mrow[rows[q]];
                if (!same && prow[q] != ~0u && (me.memb >> q & 1)) mrow[prow[q]] = (uint32_t)pmw[q];
                prow[q] = ~0u;
              } else {
                if (me.memb >> q & 1) me.mw[q] = (int)mrow[rows[q]];
              }
            }
            // Everything update() will need from LDS is fetched here, before the bit is decoded: the entry
            // itself (pv), its adaptation rate (pdt) and BOTH successor states (pns); update() is then
            // register arithmetic and stores only.
            uint32_t pv = 0, pns = 0;
            int pdt = 0;
            if (ZH_HAS(ZH_CM) && me.type == ZH_CM) {
              me.cxt = (me.h ^ hmap4) & 15;
              pv = reinterpret_cast<const uint32_t *>(myslot)[me.cxt];
              me.p = S.t.stretch[pv >> 17];
              pdt = S.t.dt[pv & 0x3ff];
            }
            uint32_t ii_a = 0;
            if (ZH_HAS(ZH_ICM) || ZH_HAS(ZH_ISSE)) {
              // every lane walks the same three dependent LDS reads (row byte -> table entry -> stretch);
              // lanes of other types read harmless locations and keep nothing
              const uint32_t sw = *(lds_u32_p)(slot_off + (hm15 & 12));
              const uint32_t st = (sw >> ((hm15 & 3) * 8)) & 255;               // the bit history of this context
              ii_a = ii_tab + (st << ii_sh);
              const uint32_t w_x = *(lds_u32_p)ii_a, w_y = *(lds_u32_p)(ii_a + 4);
              const uint32_t nsv = *(lds_u16_p)(ns_off + st * 4);               // next(state, 0) | next(state, 1) << 8
              const int stv = S.t.stretch[is_icm ? w_x >> 8 : 0];
              if (is_ii) { me.cxt = st; pns = nsv; pv = w_x; me.w0 = (int)w_x; me.w1 = (int)w_y; }
              if (is_icm) me.p = stv;
            }
            if (ZH_HAS(ZH_MATCH)) {
              const uint32_t cbit = (me.mbyte >> (7 - (me.cxt & 7))) & 1;
              if (is_match) { me.c = me.a ? cbit : me.c; me.p = me.a ? (cbit ? pm1 : pm0) : 0; }
            }
            if (ZH_HAS(ZH_MIX2) && me.type == ZH_MIX2) {
              me.cxt = (me.h + (c8 & me.a4)) & (me.c - 1);
              me.w0 = reinterpret_cast<const uint16_t *>(slot_mem + me.cmo)[me.cxt];
            }
            ZH_STAMP(0);
            // ================= predict, dependent levels =================
            if constexpr (SP::id == 1) zh_spec_levels_min(me, lane, c8, S.t.stretch, slot_mem, S.sserow);
            else if constexpr (SP::id == 2) zh_spec_levels_mid(me, lane, c8, S.t.stretch, slot_mem, S.sserow);
            else if constexpr (SP::id == 3) zh_spec_levels_max(me, lane, c8, S.t.stretch, slot_mem, S.sserow);
            else {
            for (uint32_t lv = 1; lv <= depth; ++lv) {
              const uint32_t desc = rdlane(lvl_desc, lv & 63);
              const uint32_t one = desc & 127, typ = (desc >> 8) & 15;
              if (LIKELY(one < 64)) {
                // a single component at this level: wave-uniform control flow, operands by v_readlane,
                // every lane computes, only lane `one` keeps the result
                const int pj = (int)rdlane((uint32_t)me.p, (desc >> 16) & 63);
                const bool mine = lane == one;
                if (LIKELY(typ == ZH_ISSE)) {
                  const int v = clamp2k((__mul24(me.w0, pj) + me.w1 * 64) >> 16);
                  me.p = mine ? v : me.p; me.pj = mine ? pj : me.pj;
                } else if (typ == ZH_MIX2) {
                  const int pk = (int)rdlane((uint32_t)me.p, (desc >> 24) & 63);
                  const int v = (__mul24(me.w0, pj) + __mul24(65536 - me.w0, pk)) >> 16;
                  me.p = mine ? v : me.p; me.pj = mine ? pj : me.pj; me.pk = mine ? pk : me.pk;
                } else if (typ == ZH_AVG) {
                  const int pk = (int)rdlane((uint32_t)me.p, (desc >> 24) & 63);
                  const int v = (pj * (int)me.a2 + pk * (256 - (int)me.a2)) >> 8;
                  me.p = mine ? v : me.p;
                } else if (mine) {                       // SSE (Predictor.cs:327-340)
                  me.pj = pj;
                  me.cxt = (me.h + c8) * 32u;
                  int pq = clampk(pj + 992, 0, 1983);
                  const int wt = pq & 63;
                  pq >>= 6;
                  me.cxt += (uint32_t)pq;
                  const uint32_t *cm = reinterpret_cast<const uint32_t *>(slot_mem + me.cmo);
                  const uint32_t e0 = cm[me.cxt & me.cm_mask], e1 = cm[(me.cxt + 1) & me.cm_mask];
                  me.p = S.t.stretch[((e0 >> 10) * (uint32_t)(64 - wt) + (e1 >> 10) * (uint32_t)wt) >> 13];
                  me.cxt += (uint32_t)(wt >> 5);
                  me.w0 = (int)((wt >> 5) ? e1 : e0);    // the entry train() will update
                }
              } else if (one == 64) {
                // several components at this level: every lane gathers its own operands (ds_bpermute)
                const int pj = __shfl(me.p, (int)(me.type == ZH_AVG ? me.a0 : me.a1));
                const int pk = __shfl(me.p, (int)(me.type == ZH_AVG ? me.a1 : me.a2));
                if (me.level == lv) {
                  me.pj = pj; me.pk = pk;
                  switch (me.type) {
                    case ZH_ISSE: me.p = clamp2k((__mul24(me.w0, pj) + me.w1 * 64) >> 16); break;
                    case ZH_AVG: me.p = (pj * (int)me.a2 + pk * (256 - (int)me.a2)) >> 8; break;
                    case ZH_MIX2: me.p = (__mul24(me.w0, pj) + __mul24(65536 - me.w0, pk)) >> 16; break;
                    case ZH_SSE: {
                      me.cxt = (me.h + c8) * 32u;
                      int pq = clampk(pj + 992, 0, 1983);
                      const int wt = pq & 63;
                      pq >>= 6;
                      me.cxt += (uint32_t)pq;
                      const uint32_t *cm = reinterpret_cast<const uint32_t *>(slot_mem + me.cmo);
                      const uint32_t e0 = cm[me.cxt & me.cm_mask], e1 = cm[(me.cxt + 1) & me.cm_mask];
                      me.p = S.t.stretch[((e0 >> 10) * (uint32_t)(64 - wt) + (e1 >> 10) * (uint32_t)wt) >> 13];
                      me.cxt += (uint32_t)(wt >> 5);
                      me.w0 = (int)((wt >> 5) ? e1 : e0);
                      break;
                    }
                    default: break;
                  }
                }
              }
              const uint32_t mq = (desc >> 12) & 7;
              if (mq) {                                   // a MIX: wave reduction over its input lanes
#pragma unroll
                for (uint32_t q = 0; q < (uint32_t)kMaxMix; ++q) {
                  if (q >= (kSpec ? SP::nmix : nmix)) break;
                  if (mq != 7 ? mq != q + 1 : mx_lv[q] != lv) continue;
                  const int term = (me.memb >> q & 1) ? __mul24(me.mw[q] >> 8, me.p) : 0;
                  const int sum = wave_sum(term);
                  if (lane == mx_lane[q]) me.p = clamp2k(sum >> 8);
                }
              }
            }
            }
            ZH_STAMP(1);
            // ================= decode the bit =================
            const int sqp = (int)S.t.squash[me.p + 2048];          // squash(p[i]) of every lane, one LDS pass
            const uint32_t pr = rdlane((uint32_t)sqp, n - 1);
            const uint32_t ps = (pr * 2 + 1) << 16;
            ZH_DEC_STEP(d, ps, j, bad, rn);
            if (UNLIKELY(rn)) { if (dec_renorm(d, in, lane) && !err) err = bad ? (uint32_t)-ZH_E_CORRUPT : (uint32_t)-ZH_E_EOF; }
            const int y = (int)uni(j & 1);

            ZH_STAMP(2);
            // ================= update (Predictor.cs:363-461) =================
            const int emix = __mul24(y * 32767 - sqp, (int)me.a3) >> 4;   // MIX error term (meaningful in mixer lanes)
#pragma unroll
            for (uint32_t q = 0; q < (uint32_t)kMaxMix; ++q) {      // MIX: error from the mixer lane, weights in the input lanes
              if (q >= (kSpec ? SP::nmix : nmix)) break;
              const int eq = (int)rdlane((uint32_t)emix, mx_lane[q]);
              if (me.memb >> q & 1) me.mw[q] = clamp512k(me.mw[q] + ((__mul24(eq, me.p) + (1 << 12)) >> 13));
              if (kDefer && bit < 7) { pmw[q] = me.mw[q]; prow[q] = rows[q]; }    // stored after the next bit's load
              else if (me.memb >> q & 1) reinterpret_cast<uint32_t *>(slot_mem + mx_off[q])[rows[q] + (lane - mx_j0[q])] = (uint32_t)me.mw[q];
            }
            if (ZH_HAS(ZH_CM) && me.type == ZH_CM) {
              const uint32_t cnt = pv & 0x3ff;
              const int e = y * 32767 - (int)(pv >> 17);
              reinterpret_cast<uint32_t *>(myslot)[me.cxt] = pv + (((uint32_t)e * (uint32_t)pdt) & 0xFFFFFC00u) + (cnt < me.limit);
            }
            if (ZH_HAS(ZH_SSE) && me.type == ZH_SSE) {
              const uint32_t v = (uint32_t)me.w0, cnt = v & 0x3ff;
              const int e = y * 32767 - (int)(v >> 17);
              reinterpret_cast<uint32_t *>(slot_mem + me.cmo)[me.cxt & me.cm_mask] =
                  v + (((uint32_t)e * (uint32_t)S.t.dt[cnt]) & 0xFFFFFC00u) + (cnt < me.limit);
            }
            if (ZH_HAS(ZH_ICM) || ZH_HAS(ZH_ISSE)) {            // all lanes, stores of unconcerned lanes go to their dummy cell
              *(lds_u8_p)(is_ii ? slot_off + hm15 : dummy_off) = (uint8_t)(pns >> (y * 8));   // StateTable.next
              const int e = y * 32767 - sqp;
              const uint32_t n0 = is_icm ? pv + (uint32_t)((int)(y * 32767 - (int)(pv >> 8)) >> 2)
                                         : (uint32_t)clamp512k(me.w0 + ((__mul24(e, me.pj) + (1 << 12)) >> 13));
              const uint32_t n1 = (uint32_t)clamp512k(me.w1 + ((e + 16) >> 5));
              *(lds_u32_p)(is_ii ? ii_a : dummy_off) = n0;
              *(lds_u32_p)(is_isse ? ii_a + 4 : dummy_off) = n1;
            }
            if (ZH_HAS(ZH_MATCH) && me.type == ZH_MATCH) {
              if ((int)me.c != y) me.a = 0;
              me.mcur = (me.mcur * 2 + (uint32_t)y) & 255;
              ++me.cxt;                                  // finished at the byte boundary below
            }
            if (ZH_HAS(ZH_MIX2) && me.type == ZH_MIX2) {
              const int e = __mul24(y * 32767 - sqp, (int)me.a3) >> 5;
              int w = me.w0 + ((e * (me.pj - me.pk) + (1 << 12)) >> 13);
              w = clampk(w, 0, 65535);
              reinterpret_cast<uint16_t *>(slot_mem + me.cmo)[me.cxt] = (uint16_t)w;
            }
            ZH_STAMP(3);
            // ---- c8 / hmap4 bookkeeping (Predictor.cs:463-474)
            c8 = uni(c8 * 2 + (uint32_t)y);                 // pinned to the scalar unit: LLVM's uniformity analysis otherwise
                                                            // treats the byte state as divergent and runs this loop on exec masks
            if (c8 >= 256) break;                        // byte complete: handled below
            if (c8 >= 16 && c8 < 32) {
              hmap4 = uni((hmap4 & 0xf) << 5 | (uint32_t)y << 4 | 1);
              nibble_refresh();
              ZH_STAMP(4);
            } else hmap4 = uni((hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + (uint32_t)y) & 0xf));
          }
          if (UNLIKELY(err | bad)) { status = err ? -(int)err : ZH_E_CORRUPT; break; }
          c = (int)(c8 - 256);

          // ---- MATCH at the byte boundary (Predictor.cs:391-410)
          if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }
          {
            const bool ism = me.type == ZH_MATCH;
            uint32_t need = 0;
            if (ism) {
              me.cxt = 0;
              (slot_mem + me.hto)[me.limit & me.ht_mask] = (uint8_t)me.mcur;   // the assembled byte; ht(0)=1 is overwritten like the reference
              me.mcur = 0;
              me.limit = (me.limit + 1) & me.ht_mask;
            }
            uint32_t cmv = 0;
            if (ism) {                                     // still with the h[i] of the byte just coded (update0 runs before z.run)
              uint32_t *cm = reinterpret_cast<uint32_t *>(slot_mem + me.cmo);
              cmv = cm[me.h & me.cm_mask];                 // consumed after HCOMP: the load travels meanwhile
              cm[me.h & me.cm_mask] = me.limit;
            }
            ZH_STAMP(5);
            // h[] for the next byte: z.run(c), then H(i) (Predictor.cs:465-469)
            int rc;
            switch (hnative) {                             // native forms of the known programs (tools/gen_zpaql_native.py)
              case ZH_NATIVE_HCOMP_MIN: rc = zh_native_hcomp_min(ha, hb, hc, hd, hf, (uint32_t)c, lds_m, hz.mmask, lds_h, hmask, S.r, (Sink *)nullptr, L.budget); break;
              case ZH_NATIVE_HCOMP_MID: rc = zh_native_hcomp_mid(ha, hb, hc, hd, hf, (uint32_t)c, lds_m, hz.mmask, lds_h, hmask, S.r, (Sink *)nullptr, L.budget); break;
              case ZH_NATIVE_HCOMP_MAX: rc = zh_native_hcomp_max(ha, hb, hc, hd, hf, (uint32_t)c, lds_m, hz.mmask, lds_h, hmask, S.r, (Sink *)nullptr, L.budget); break;
              default: rc = vm_run(hz, (uint32_t)c, nullptr, L.budget); break;
            }
            rc = (int)uni((uint32_t)rc);
            if (rc) { status = rc; break; }
            me.h = h_lds ? lds_h[lane & hmask] : Hptr[lane & hmask];   // typed LDS read when H lives there (not a flat access)
            hmap4 = 1; c8 = 1;
            nibble_issue();                                // rows of the next byte's first nibble: in flight during MATCH
            ZH_STAMP(6);
            if (ism) {
              if (me.a == 0) {
                me.b = me.limit - cmv;
                need = (me.b & me.ht_mask) != 0;
              } else me.a += me.a < 255;
            }
            uint64_t nm = __ballot(need != 0);
            while (nm) {                                   // verify candidates with the whole wave
              const uint32_t ml = (uint32_t)__builtin_ctzll(nm);
              nm &= nm - 1;
              const uint32_t lim = rdlane(me.limit, ml), off = rdlane(me.b, ml), msk = rdlane(me.ht_mask, ml);
              const uint8_t *hp = slot_mem + rdlane(me.hto, ml);
              uint32_t len = 0;
              for (uint32_t base = 0; base < 256; base += 64) {
                const uint32_t t = base + lane;
                const bool eq = t < 255 && hp[(lim - t - 1) & msk] == hp[(lim - t - off - 1) & msk];
                const uint64_t mism = __ballot(!eq);
                if (mism) { len += (uint32_t)__builtin_ctzll(mism); break; }
                len += 64;
              }
              if (lane == ml) me.a = len > 255 ? 255 : len;
            }
            if (ism) me.mbyte = (slot_mem + me.hto)[(me.limit - me.b) & me.ht_mask];
            if (ZH_HAS(ZH_MATCH)) {                        // the two predictions a match of this length can make (Predictor.cs:273-287)
              const int dk = S.t.dt2k[is_match ? me.a : 0];
              pm0 = S.t.stretch[dk & 32767];
              pm1 = S.t.stretch[(-dk) & 32767];
            }
            nibble_finish();
          }

          ZH_STAMP(7);
        }

        // ---- PostProcessor.write(c) (PostProcessor.cs:37-86)
        c = (int)uni((uint32_t)c);
        if (LIKELY(pp_state == 1)) {
          if (LIKELY(c >= 0)) out_put(ob, (uint32_t)c, lane);
        } else if (pp_state == 5) {
          int rc;
          if (pnative == ZH_NATIVE_PCOMP_E8E9)
            rc = zh_native_pcomp_e8e9(pa, pb, pc_, pd, pf, (uint32_t)c, (lds_u8_p)lds_off(S.pmreg), pz.mmask, (lds_u32_p)lds_off(S.phreg), pz.hmask, S.pr, &sink, L.budget);
          else if (PCALL && pskel) {
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
            __syncthreads();
            pz.prog = pzbuf; pz.len = pp_len;
            pz.a = pz.b = pz.c = pz.d = pz.f = 0;
            pnative = p_lds ? uni(zh_native_pcomp_lookup(pzbuf, pp_len)) : 0;
            if constexpr (PCALL) {
              pskel = pnative ? 0u : uni(zh_pcomp_lookup(pzbuf, pp_len));
              if (lane == 0) zh_pcomp_operands(pskel, pzbuf, S.pimm);
              __syncthreads();
            }
            pp_state = 5;
          }
        }
        if (c < 0) break;
      }

      if (pp_state != 5) out_flush(ob, lane);
      const uint64_t produced = pp_state == 5 ? uni64(sink.len) : ob.len;
      if (!status && produced > b_out_cap) status = ZH_E_OUTPUT_FULL;
      if (status && status != ZH_E_OUTPUT_FULL) failed = 1;
      if (lane == 0) {
        ZhSegResult res;
        res.status = status; res.pp_state = (uint32_t)pp_state | (uint32_t)pp_hsize << 8;   // PCOMP length rides in bits 8-23
        res.out_off = b_out_off + produced0; res.out_len = produced - produced0;
        res.in_used = in_pos(in) - seg_off;
        L.results[si] = res;
      }
    }
    if (PROF && lane == 0 && L.debug)
      for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long *)&L.debug[i], (unsigned long long)prof[i]);
    __syncthreads();
  }
}

}  // namespace

#undef ZH_HAS

#define ZH_CHAIN_KERNEL(name, prof, spec, pcall)                                       \
  extern "C" __global__ __launch_bounds__(64) void name(ZhLaunch L) {                  \
    __shared__ ChainLds S;                                                             \
    decode_chain_body<prof, spec, pcall>(L, S);                                        \
  }
ZH_CHAIN_KERNEL(zh_decode_chain, false, ZhSpec_generic, false)
// ... the same with the translated post-processors of zh_zpaql_pcomp.h behind a call: launched for models that give their
// PCOMP memory (ph or pm > 0: lazy2, lzpre, bwtrle).  Kept apart because the mere presence of a call costs the bit loop
// ~10 % (scalar registers reserved for the stack), which blocks without such a post-processor should not pay.
ZH_CHAIN_KERNEL(zh_decode_chain_pc, false, ZhSpec_generic, true)
ZH_CHAIN_KERNEL(zh_decode_chain_min, false, ZhSpec_min, false)
ZH_CHAIN_KERNEL(zh_decode_chain_mid, false, ZhSpec_mid, false)
ZH_CHAIN_KERNEL(zh_decode_chain_max, false, ZhSpec_max, false)
ZH_CHAIN_KERNEL(zh_decode_chain_prof, true, ZhSpec_generic, false)
ZH_CHAIN_KERNEL(zh_decode_chain_mid_prof, true, ZhSpec_mid, false)
ZH_CHAIN_KERNEL(zh_decode_chain_max_prof, true, ZhSpec_max, false)

// spec: 0 generic, 1 min, 2 mid, 3 max (zh_chain_spec.h); prof: diagnostic build with stamps; pcall: some model of the
// launch has PCOMP memory
extern "C" hipError_t zh_launch_chain(const ZhLaunch *L, uint32_t grid, hipStream_t stream, uint32_t spec, int prof, int pcall) {
  void (*k)(ZhLaunch) = zh_decode_chain;
  if (prof) k = spec == 2 ? zh_decode_chain_mid_prof : spec == 3 ? zh_decode_chain_max_prof : zh_decode_chain_prof;
  else k = spec == 1 ? zh_decode_chain_min : spec == 2 ? zh_decode_chain_mid : spec == 3 ? zh_decode_chain_max
           : pcall ? zh_decode_chain_pc : zh_decode_chain;
  hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, stream, *L);
  return hipGetLastError();
}
