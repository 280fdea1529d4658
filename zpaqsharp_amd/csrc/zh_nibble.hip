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
// The helper wavefront (zh_c2_common.h: HCOMP for the 16 values the byte can still take, and the hash rows / mixer row
// the next byte starts with) is unchanged.  Results are bit-exact with zh_chain2.hip and the oracle (tests/).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zh_core.h"
#include "zh_dev.h"
#include "zh_model.h"
#include "zh_zpaql_native.h"

using namespace zhcore;
using namespace zhdev;

#pragma clang diagnostic ignored "-Wint-to-pointer-cast"

#define C2_TOUCH 0
#define C2_FINDB 1
#include "zh_c2_common.h"

namespace {

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
  uint32_t hspec[kSpecH][16];
  v4u_ rowst[kSpecUnits][3][16];
  uint32_t mixst[2][16][16];
  v4u_ selrow[kSpecUnits][16];
  uint32_t seloff[kSpecUnits][16];
  uint32_t mb_nib, mb_byte, mb_ready;
  uint32_t mb_cmd, mb_ack, mb_model;
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

template <class SP, bool PROF, class LDS>
__device__ __forceinline__ void decode_nibble_body(const ZhLaunch &L, LDS &S) {
  constexpr uint32_t NC = SP::id == 1 ? 2u : 8u;          // lanes of a group
  constexpr uint32_t NG = 8u;                               // groups: the 3-bit prefixes of a nibble's path
  constexpr uint64_t kII = SP::icm | SP::isse;
  static_assert(SP::n <= NC && NC * NG <= 64, "lane budget");
  uint64_t prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tprev = 0;
  const uint32_t lane = threadIdx.x & 63u;
  const bool wave_a = (threadIdx.x >> 6) == 0;
  const uint32_t ci = lane & (NC - 1u), g = (lane / NC) & (NG - 1u);
  const bool act = lane < NC * NG;                         // min: lanes 16-63 repeat lanes 0-15 and never write
  const uint32_t b1 = (g >> 2) & 1u, b2 = (g >> 1) & 1u, b3 = g & 1u;
  const bool l_isse = (SP::isse >> ci) & 1, l_ii = (kII >> ci) & 1;
  const bool l_match = SP::match_lane >= 0 && ci == (uint32_t)SP::match_lane;

  if (wave_a) {  // model-independent tables -> LDS
    const ZhTables *T = L.tables;
    for (uint32_t i = lane; i < 32768 / 8; i += 64) reinterpret_cast<uint4 *>(S.stretch)[i] = reinterpret_cast<const uint4 *>(T->stretch)[i];
    for (uint32_t i = lane; i < 4096 / 8; i += 64) reinterpret_cast<uint4 *>(S.squash)[i] = reinterpret_cast<const uint4 *>(T->squash)[i];
    for (uint32_t i = lane; i < 1024 / 16; i += 64) reinterpret_cast<uint4 *>(S.ns)[i] = reinterpret_cast<const uint4 *>(T->ns)[i];
    if (lane == 0) { S.zrow = v4u_{0, 0, 0, 0}; S.mb_cmd = 0; S.mb_ack = 0; S.mb_nib = 0; S.mb_byte = 0; S.mb_ready = 0; }
  }
  __syncthreads();                                       // the only workgroup barrier of the kernel
  if (!wave_a) { c2_helper<SP, LDS, PROF>(L, S, lane, blockIdx.x); return; }
  uint32_t cmd_seq = 0;
  for (uint32_t i = lane; i < 256; i += 64) {            // the two predictions of a match of length i
    const int dk = L.tables->dt2k[i];
    const uint32_t lo = (uint16_t)S.stretch[dk & 32767], hi = (uint16_t)S.stretch[(-dk) & 32767];
    S.pm01[i] = i ? lo | hi << 16 : 0u;
  }
  nb_wave_sync();

  uint8_t *slot_mem = L.arena + (uint64_t)blockIdx.x * L.arena_stride;
  const lds_i16_p lds_stretch = (lds_i16_p)lds_off(S.stretch);
  const lds_u16_p lds_squash = (lds_u16_p)lds_off(S.squash);
  const uint32_t ns_off = lds_off(S.ns);

  // ---- per-lane constants of the path this lane's group walks
  const uint32_t sh2 = 16u + 8u * b1;                   // node 2 + b1: byte 2 / 3 of row dword 0
  const uint32_t sh3 = 8u * (2u * b1 + b2);             // node 4 + 2 b1 + b2: a byte of dword 1
  const uint32_t sh4 = 8u * (2u * b2 + b3);             // node 8 + 4 b1 + 2 b2 + b3: a byte of dword 2 (b1 = 0) / 3
  const uint32_t node[5] = {0u, 1u, 2u + b1, 4u + 2u * b1 + b2, 8u + 4u * b1 + 2u * b2 + b3};
  const uint32_t pre[5] = {0u, 0u, b1, 2u * b1 + b2, 4u * b1 + 2u * b2 + b3};      // c8 of level d = (c8 of the nibble << (d-1)) + pre[d]
  const uint32_t ybit[4] = {0u, b1, b2, b3};             // the bit this group assumes at level d (d = 1..3)

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
    const uint32_t arena_bytes = uni((uint32_t)M->arena_bytes);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(slot_mem, 0, (int)arena_bytes, 0x00020000);

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
    const uint32_t unit = (uint32_t)__builtin_popcountll(kII & ((1ull << ci) - 1));
    {
      for (uint32_t j = lane; j < 256; j += 64) {
        const uint32_t n0 = S.ns[j * 4 + 2], n1 = S.ns[j * 4 + 3];
        const uint32_t cinit = ((n1 * 2 + 1) << 22) / (n0 + n1 + 1);                // StateTable.cminit
        const int stv = S.stretch[cinit >> 8];
        const v2u e_icm = {cinit, (uint32_t)stv};
        const v2u e_isse = {1u << 15, (uint32_t)clamp512k(stv * 1024)};
        uint32_t u = 0;
        for (uint32_t i = 0; i < SP::n; ++i) {
          if (!((kII >> i) & 1)) continue;
          S.ent[u][j] = ((SP::icm >> i) & 1) ? e_icm : e_isse;
          ++u;
        }
      }
      if (lane < 8) { S.slot[lane] = v4u{0, 0, 0, 0}; S.lent[lane] = v2u{0, 0}; S.lsink[lane] = 0; S.slotoff[lane] = 0; S.mixb[lane] = 0; }
    }
    nb_wave_sync();

    // ---- per-lane constants of the component
    const ZhComp *mycp = &M->comp[ci < SP::n ? ci : 0];
    const uint32_t hto = l_ii || l_match ? (uint32_t)mycp->ht_off : 0u, ht_mask = mycp->ht_mask;
    const uint32_t cmo = (uint32_t)mycp->cm_off, cm_mask = mycp->cm_mask;
    const uint32_t sizebits2 = (uint32_t)mycp->arg[0] + 2;
    const uint32_t tab = l_ii ? lds_off(&S.ent[unit][0]) : lds_off(&S.lent[ci]);      // entry table of this lane's component
    const uint32_t wrow = l_ii ? lds_off(&S.slot[ci]) : lds_off(&S.lsink[ci]);        // where its bit histories are written (+ node)
    const uint32_t wrow_mask = l_ii ? 15u : 0u;
    const int isse_m = l_isse ? -1 : 0;
    const uint32_t cshift = l_isse ? 6u : 16u;
    int pself = 0;                                       // prediction of a lane that is neither ICM nor ISSE (MATCH, CONST)
    if (ci < SP::n && mycp->type == ZH_CONS) pself = ((int)mycp->arg[0] - 128) * 4;
    if (l_match && lane == (uint32_t)SP::match_lane) (slot_mem + hto)[0] = 1;           // Predictor.cs:121 ht(0)=1 ... overwritten like the reference

    // the mixer kept in HBM: lane (g, k) owns weight k of the rows its group reads
    uint32_t vo_mix = kOob;
    uint32_t mx_base = 0, mx_size1 = 0;
    int mx_rate = 0;
    constexpr uint32_t mx_m4 = SP::mix_m[0] * 4u;
    if (SP::nmix) {
      const ZhComp &mc = M->comp[SP::mix_lane[0]];
      mx_base = uni((uint32_t)mc.cm_off);
      mx_size1 = uni(mc.cm_mask);
      mx_rate = (int)uni((uint32_t)mc.arg[3]);
      asm volatile("" : "+v"(mx_rate));
      if (ci >= SP::mix_j0[0] && ci < SP::mix_j0[0] + SP::mix_m[0]) vo_mix = (ci - SP::mix_j0[0]) * 4u;
    }
    const bool l_feed = vo_mix != kOob;

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
    uint32_t pnative = 0;
    uint32_t pa = 0, pb = 0, pc_ = 0, pd = 0, pf = 0;
    uint8_t *pzbuf = slot_mem + uni64(M->pz_off) + ZH_CODE_PAD;

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
    if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }
    InBuf in;
    in.stream = L.in; in.total = L.in_total; in.cbase = 0; in.k = 0; in.avail = 0; in.cur = 0;

    // ---- state carried from nibble to nibble (the same in every group)
    uint32_t hv = 0;                                    // h[component] (Predictor.cs:469)
    uint32_t rowoff = 0;                                // place of the hash row of the current nibble (in S.slot[ci] and below)
    uint32_t row_x = 0, row_q1 = 0, row_q2 = 0, row_q3 = 0;   // the row as it was when the nibble began (zero for components without a table)
    bool rowvalid = false;
    int mwl[5] = {0, 0, 0, 0, 0};                       // mixer: this lane's weight in the row of level d of the current nibble
    uint32_t mrowl[5] = {0, 0, 0, 0, 0};                // ... and its buffer offset
    uint32_t mx_rb = kOob;                              // this lane's buffer offset in row 0 of the byte's block of mixer rows
    // MATCH (Predictor.cs:273-287, 382-411): the Component fields, the same in every lane that stands for it
    uint32_t m_len = 0, m_ptr = 0, m_limit = 0, m_byte = 0;
    int pm0 = 0, pm1 = 0;                               // stretch of -+dt2k[len] for this byte; 0 once the match has failed
    uint32_t cm_pre = 0, va_pre = 0, vb_pre = 0, mbn_pre = 0, mbc_pre = 0;      // see zh_chain2.hip
    v4u oldb = {0, 0, 0, 0}; uint32_t oldb_off = 0; bool oldb_valid = false;    // the row written back at the last byte boundary
    auto match_prefetch = [&]() __attribute__((always_inline)) {
      const uint32_t ml = (uint32_t)(SP::match_lane >= 0 ? SP::match_lane : 0);
      const uint32_t msk = rdlane(ht_mask, ml), base = rdlane(hto, ml);
      const uint32_t lim = (rdlane(m_limit, ml) + 1u) & msk;                 // m_limit once this byte is stored
      const uint32_t off = lim - rdlane(cm_pre, ml);                        // the candidate's distance, should the byte end unmatched
      va_pre = __builtin_amdgcn_raw_buffer_load_b8(rsrc, base + ((lim - lane - 1u) & msk), 0, 0);
      vb_pre = __builtin_amdgcn_raw_buffer_load_b8(rsrc, base + ((lim - lane - off - 1u) & msk), 0, 0);
      mbn_pre = __builtin_amdgcn_raw_buffer_load_b8(rsrc, l_match ? base + ((lim - off) & msk) : kOob, 0, 0);
      mbc_pre = __builtin_amdgcn_raw_buffer_load_b8(rsrc, l_match ? base + ((lim - m_ptr) & msk) : kOob, 0, 0);
    };
    // Predictor.update's MATCH part at the byte boundary (Predictor.cs:391-410); c is already in the history and the
    // hash index, m_limit advanced
    auto match_boundary = [&](uint32_t cb) __attribute__((always_inline)) {
      const bool zero = m_len == 0;
      const uint32_t nptr = m_limit - cm_pre;
      const bool need = l_match && zero && (nptr & ht_mask) != 0;
      m_ptr = (l_match && zero) ? nptr : m_ptr;
      m_len = (l_match && !zero && m_len < 255) ? m_len + 1 : m_len;
      if (__ballot(need) != 0) {                         // verify the candidate with the whole wave (Predictor.cs:403-405)
        const uint32_t ml = (uint32_t)SP::match_lane;
        const uint32_t lim = rdlane(m_limit, ml), off = rdlane(m_ptr, ml), msk = rdlane(ht_mask, ml);
        const uint32_t a = lane == 0 ? cb : (va_pre & 255u);
        const uint32_t b = ((lane + off) & msk) == 0 ? cb : (vb_pre & 255u);
        uint64_t mism = __ballot(a != b);
        uint32_t len = 64;
        if (LIKELY(mism != 0)) len = (uint32_t)__builtin_ctzll(mism);
        else {
          const uint8_t *hp = slot_mem + rdlane(hto, ml);
          for (uint32_t base = 64; base < 256; base += 64) {
            const uint32_t t = base + lane;
            const bool eq = t < 255 && hp[(lim - t - 1) & msk] == hp[(lim - t - off - 1) & msk];
            mism = __ballot(!eq);
            if (mism) { len += (uint32_t)__builtin_ctzll(mism); break; }
            len += 64;
          }
        }
        const uint32_t nl = len > 255 ? 255 : len;
        m_len = l_match ? nl : m_len;
        m_byte = l_match ? (((off - 1u) & msk) == 0 ? cb : (mbn_pre & 255u)) : m_byte;
      } else {
        const uint32_t cont = ((m_ptr - 1u) & ht_mask) == 0 ? cb : (mbc_pre & 255u);
        m_byte = (l_match && m_len) ? cont : m_byte;
      }
      const uint32_t pw = *(lds_u32_p)(lds_off(S.pm01) + m_len * 4u);      // m_len stays 0 in the other lanes
      pm0 = (int)(int16_t)(pw & 0xffffu); pm1 = (int)pw >> 16;
    };

    // Hash rows of a nibble (c8 == 1 or 16 <= c8 < 32), Predictor.find (Predictor.cs:550-567): see zh_chain2.hip
    struct Probe { v4u r0, r1, r2; uint32_t h0, chk; };
    auto rows_issue = [&](uint32_t c8, Probe &pr, bool on) __attribute__((always_inline)) {
      const uint32_t cxt = hv + 16u * c8;
      pr.chk = (cxt >> sizebits2) & 255;
      pr.h0 = (cxt * 16u) & (ht_mask - 15u);
      const uint32_t vo = (l_ii && on) ? hto + pr.h0 : kOob;
      pr.r0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, 0, 0);
      pr.r1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo ^ 16u, 0, 0);
      pr.r2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo ^ 32u, 0, 0);
    };
    // find on the three probes; rows this wave evicted after the probes' loads may have been issued (olda, old) are taken
    // from the copies.  Result: the row and its place (nothing is written).
    auto rows_pick = [&](const Probe &pr, const v4u &olda, uint32_t olda_off, bool olda_valid, const v4u &old, uint32_t old_off,
                         bool old_valid, bool guard, v4u &row, uint32_t &sel) __attribute__((always_inline)) {
      const uint32_t h0 = pr.h0, h1 = h0 ^ 16u, h2 = h0 ^ 32u;
      v4u r0 = pr.r0, r1 = pr.r1, r2 = pr.r2;
      const bool near = (olda_valid && ((olda_off ^ h0) & ~48u) == 0) || (old_valid && ((old_off ^ h0) & ~48u) == 0);
      if (!guard || UNLIKELY(__ballot(near) != 0)) {
        if (olda_valid && olda_off == h0) r0 = olda;
        if (olda_valid && olda_off == h1) r1 = olda;
        if (olda_valid && olda_off == h2) r2 = olda;
        if (old_valid && old_off == h0) r0 = old;
        if (old_valid && old_off == h1) r1 = old;
        if (old_valid && old_off == h2) r2 = old;
      }
      const uint32_t chk = pr.chk;
      const bool m0 = (r0.x & 255) == chk, m1 = (r1.x & 255) == chk, m2 = (r2.x & 255) == chk;
      const uint32_t p0 = (r0.x >> 8) & 255, p1 = (r1.x >> 8) & 255, p2 = (r2.x >> 8) & 255;
      const uint32_t victim = (p0 <= p1 && p0 <= p2) ? h0 : p1 < p2 ? h1 : h2;
      sel = m0 ? h0 : m1 ? h1 : m2 ? h2 : victim;
      const v4u fresh = {chk, 0, 0, 0};
      row = m0 ? r0 : m1 ? r1 : m2 ? r2 : fresh;
    };
    auto row_take = [&](const v4u &row, uint32_t sel) __attribute__((always_inline)) {     // every lane holds the row of its component
      rowoff = sel; rowvalid = true;
      row_x = l_ii ? row.x : 0u; row_q1 = l_ii ? row.y : 0u; row_q2 = l_ii ? row.z : 0u; row_q3 = l_ii ? row.w : 0u;
    };
    // write the row of the finished nibble back (fire and forget) and hand its content to the caller
    auto row_evict = [&](v4u &old, uint32_t &old_off, bool &old_valid) __attribute__((always_inline)) {
      old = *(lds_u4_p)lds_off(&S.slot[ci]);
      old_off = rowoff; old_valid = rowvalid && l_ii;
      __builtin_amdgcn_raw_buffer_store_b128(old, rsrc, (old_valid && lane < NC) ? hto + rowoff : kOob, 0, 0);
    };
    auto mix_set = [&](uint32_t hq) __attribute__((always_inline)) {
      mx_rb = vo_mix + (mx_base + __umul24(uni(hq) & mx_size1 & ~255u, mx_m4));
    };
    // rows of levels 2..4 of a nibble whose first row is c8n (1, or 16 + first nibble); level 1 comes staged / fetched ahead
    auto mix_rows = [&](uint32_t c8n) __attribute__((always_inline)) {
#pragma unroll
      for (int dd = 1; dd <= 4; ++dd) mrowl[dd] = mx_rb + __umul24((c8n << (dd - 1)) + pre[dd], mx_m4);
#pragma unroll
      for (int dd = 2; dd <= 4; ++dd) mwl[dd] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, mrowl[dd], 0, 0);
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
      if (s == 0) {                                      // first nibble of the block (h[] = 0)
        Probe pr;
        rows_issue(1u, pr, true);
        v4u row; uint32_t sel;
        rows_pick(pr, v4u{0, 0, 0, 0}, 0u, false, v4u{0, 0, 0, 0}, 0u, false, false, row, sel);
        if (lane < NC) *(lds_u4_p)lds_off(&S.slot[ci]) = row;
        row_take(row, sel);
        if (SP::nmix) {
          mix_set(0u);
          mix_rows(1u);
          mwl[1] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, mrowl[1], 0, 0);
        }
      }

      v4u old1 = {0, 0, 0, 0}; uint32_t old1_off = 0; bool old1_valid = false;   // the first nibble's row as it was evicted
      int w1_new = 0;                                  // mixer: row c8 = 1 of the byte's block after the first nibble
      // ---- the eight bits of a byte (Decoder.cs:48-55 around Predictor.predict / update); j, bad, err as in zh_chain2.hip
      auto decode_byte = [&](uint32_t &j, uint32_t &bad, uint32_t &err) __attribute__((always_inline)) -> uint32_t {
          NB_STAMP(10);
          int rnd12 = 1 << 12;                           // the weight updates' rounding addend, in a VGPR (zh_chain2.hip, C2V 64)
          asm volatile("" : "+v"(rnd12));
          // ---- the second nibble's hash rows, for the two values of the first nibble this group's prefix leaves open
          // (16 candidates over the 8 groups), and the mixer weights of the rows they start with
          Probe cand[2];
          int cmw[2] = {0, 0};
#pragma unroll
          for (uint32_t k = 0; k < 2; ++k) {
            rows_issue(16u + 2u * g + k, cand[k], act);
            if (SP::nmix) cmw[k] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, act ? mx_rb + __umul24(16u + 2u * g + k, mx_m4) : kOob, 0, 0);
          }
          uint32_t cbyte = 0;
          int p_l1 = 0, sqm_l1 = 0, mw_l1 = 0;            // mixer: level 1 of the first nibble (the same in every group)
#pragma unroll
          for (int nib = 0; nib < 2; ++nib) {
            // ================= one nibble: four levels, every group on its own path =================
            uint32_t st[5], ea[5], nsp[5], nA[5], nB[5], nsb[5];
            v2u et[5];
            int nmw[5] = {0, 0, 0, 0, 0};
            st[1] = __builtin_amdgcn_ubfe(row_x, 8u, 8u);
            st[2] = __builtin_amdgcn_ubfe(row_x, sh2, 8u);
            st[3] = __builtin_amdgcn_ubfe(row_q1, sh3, 8u);
            st[4] = __builtin_amdgcn_ubfe(b1 ? row_q3 : row_q2, sh4, 8u);
#pragma unroll
            for (int dd = 1; dd <= 4; ++dd) {
              ea[dd] = tab + st[dd] * 8u;
              et[dd] = *(lds_u2_p)ea[dd];
              nsp[dd] = *(lds_u16_p)(ns_off + st[dd] * 4u);      // next(state, 0) | next(state, 1) << 8
            }
            // MATCH: the nibble the match predicts; a group whose path left it predicts 0 from there on
            uint32_t ex = 0, mism = 0;
            if (SP::match_lane >= 0) {
              ex = (m_byte >> (nib ? 0u : 4u)) & 15u;
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
              int xs = pself;
              if (SP::match_lane >= 0) {
                const uint32_t cbit = (ex >> (4 - dd)) & 1u;
                int pmv = cbit ? pm1 : pm0;
                const uint32_t left = dd == 1 ? 0u : dd == 2 ? (mism & 4u) : dd == 3 ? (mism & 6u) : mism;
                pmv = left ? 0 : pmv;
                xs = l_match ? pmv : pself;
              }
              const int x = l_ii ? (int)eB : xs;
              const int cw0 = (int)eA & isse_m;
              const int cw1m = (int)((uint32_t)x << cshift);
              p = x;
#pragma unroll
              for (uint32_t t = 0; t < SP::depth; ++t) p = med3i((__mul24(shr1(p), cw0) + cw1m) >> 16, -2048, 2047);
              uint32_t psv;
              if (SP::nmix) {
                const int term = sum8(__mul24(mwl[dd] >> 8, p));     // lanes that do not feed the mixer hold weight 0
                const int pmx = med3i(term >> 8, -2048, 2047);
                if (ci == SP::mix_lane[0]) p = pmx;
                sqm = (int)*(lds_u16_p)((uint32_t)(uintptr_t)lds_squash + (uint32_t)(pmx + 2048) * 2u);
                psv = ((uint32_t)sqm << 17) | 0x10000u;
              }
              sqp = (int)*(lds_u16_p)((uint32_t)(uintptr_t)lds_squash + (uint32_t)(p + 2048) * 2u);
              if (!SP::nmix) psv = ((uint32_t)sqp << 17) | 0x10000u;
              if (SP::nmix && nib == 0 && dd == 1) { p_l1 = p; sqm_l1 = sqm; mw_l1 = mwl[1]; }
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
                const uint32_t was = bad;
                uint32_t later = 0;                       // after the byte's last bit the next EOS step re-checks by itself
                if (dec_renorm_chk(d, in, lane, (nib == 1 && dd == 4) ? later : bad) && !err) err = was ? (uint32_t)-ZH_E_CORRUPT : (uint32_t)-ZH_E_EOF;
              }
              const uint32_t y = uni(j & 1u);
              nv = nv * 2u + y;
              // ---- update (Predictor.cs:363-461): levels 1-3 with the group's own bit, level 4 with the decoded one
              const uint32_t yl = dd < 4 ? ybit[dd] : y;
              const int ey = yl ? 32767 : 0;
              const int e = ey - sqp;
              nsb[dd] = __builtin_amdgcn_ubfe(nsp[dd], yl * 8u, 8u);
              const uint32_t ncm = eA + (uint32_t)((int)(ey - (int)(eA >> 8)) >> 2);
              const int npst = *(lds_i16_p)((uint32_t)(uintptr_t)lds_stretch + ((ncm >> 7) & 0x1fffeu));
              const int nw0 = med3i((int)eA + ((__mul24(e, pj) + rnd12) >> 13), -(1 << 19), (1 << 19) - 1);
              const int nw1 = med3i((int)eB + ((e + 16) >> 5), -(1 << 19), (1 << 19) - 1);
              nA[dd] = l_isse ? (uint32_t)nw0 : ncm;
              nB[dd] = l_isse ? (uint32_t)nw1 : (uint32_t)npst;
              if (SP::nmix) {                            // MIX (Predictor.cs:427-439)
                const int eq = __mul24(ey - sqm, mx_rate) >> 4;
                nmw[dd] = med3i(mwl[dd] + ((__mul24(eq, p) + rnd12) >> 13), -(1 << 19), (1 << 19) - 1);
              }
              if (dd == 2 && nib == 0 && SP::match_lane >= 0) { /* nothing: MATCH's boundary requests go out at the nibble switch */ }
            }
            NB_STAMP(nib);
            // ---- commit: the group the nibble's first three bits name
            const uint32_t gw = nv >> 1;
            if (act && g == gw) {
#pragma unroll
              for (int dd = 1; dd <= 4; ++dd) *(lds_u2_p)ea[dd] = v2u{nA[dd], nB[dd]};
#pragma unroll
              for (int dd = 1; dd <= 4; ++dd) *(lds_u8_p)(wrow + (node[dd] & wrow_mask)) = (uint8_t)nsb[dd];
            }
            if (SP::nmix) {
#pragma unroll
              for (int dd = 1; dd <= 4; ++dd) __builtin_amdgcn_raw_buffer_store_b32((uint32_t)nmw[dd], rsrc, (act && g == gw) ? mrowl[dd] : kOob, 0, 0);
            }
            if (SP::nmix && nib == 0) {
              // row c8 = 1 of this byte's block as it is now, in every group (level 1 is the same everywhere; y1 is known): should
              // the next byte have the same mixer context, the helper's copy of that row may predate the store above
              const int eq1 = __mul24(((nv & 8u) ? 32767 : 0) - sqm_l1, mx_rate) >> 4;
              w1_new = med3i(mw_l1 + ((__mul24(eq1, p_l1) + rnd12) >> 13), -(1 << 19), (1 << 19) - 1);
            }
            if (SP::match_lane >= 0) {                    // MATCH (Predictor.cs:383-384): a miss ends the match
              const bool miss = nv != ex;
              m_len = miss ? 0u : m_len; pm0 = miss ? 0 : pm0; pm1 = miss ? 0 : pm1;
            }
            NB_STAMP(2 + nib);
            if (nib == 0) {
              // ---- second nibble (Predictor.cs:267-270: c8 & 0xf0 == 16): its rows were requested when the byte began
              cbyte = nv;
              v4u old; uint32_t old_off; bool old_valid;
              row_evict(old, old_off, old_valid);
              old1 = old; old1_off = old_off; old1_valid = old_valid;
              if (SP::nmix > 0) c2_put0(&S.mb_nib, bseq << 8 | (nv & 15u));       // (see zh_chain2.hip for what the helper's loads must see)
              const bool k1 = (nv & 1u) != 0;
              Probe pr;
              pr.h0 = k1 ? cand[1].h0 : cand[0].h0; pr.chk = k1 ? cand[1].chk : cand[0].chk;
              pr.r0 = k1 ? cand[1].r0 : cand[0].r0; pr.r1 = k1 ? cand[1].r1 : cand[0].r1; pr.r2 = k1 ? cand[1].r2 : cand[0].r2;
              v4u row; uint32_t sel;
              rows_pick(pr, oldb, oldb_off, oldb_valid, old, old_off, old_valid, false, row, sel);   // (oldb: written back just before the candidates were requested)
              if (act && g == gw) {
                *(lds_u4_p)lds_off(&S.slot[ci]) = row;
                S.slotoff[ci] = sel;
                if (SP::nmix) S.mixb[ci] = (uint32_t)(k1 ? cmw[1] : cmw[0]);
              }
              asm volatile("" ::: "memory");
              {
                const v4u rowb = *(lds_u4_p)lds_off(&S.slot[ci]);
                const uint32_t selb = S.slotoff[ci];
                row_take(rowb, selb);
                if (SP::nmix) { mwl[1] = l_feed ? (int)S.mixb[ci] : 0; mix_rows(16u + nv); }
              }
              if (SP::nmix == 0) c2_put0(&S.mb_nib, bseq << 8 | (nv & 15u));      // min: after the rows requested at the byte's start were consumed
              if (SP::match_lane >= 0) match_prefetch();
              NB_STAMP(4);
            } else cbyte = cbyte * 16u + nv;
          }
          return cbyte;
      };
      // ---- byte boundary: MATCH (Predictor.cs:391-410), h[] and the rows of the next byte from the helper wave
      auto boundary = [&](int c) __attribute__((always_inline)) -> bool {
            v4u sg_row = {0, 0, 0, 0}; uint32_t sg_sel = 0; int sg_mw = 0;
            if (SP::match_lane >= 0) {                   // still with the h[i] of the byte just coded (update0 runs before z.run)
              __builtin_amdgcn_raw_buffer_store_b8((uint8_t)c, rsrc, (l_match && lane < NC) ? hto + (m_limit & ht_mask) : kOob, 0, 0);
              m_limit = l_match ? (m_limit + 1) & ht_mask : m_limit;
              __builtin_amdgcn_raw_buffer_store_b32(m_limit, rsrc, (l_match && lane < NC) ? cmo + (hv & cm_mask) * 4u : kOob, 0, 0);   // (its old value: cm_pre)
            }
            c2_put0(&S.mb_byte, bseq << 8 | (uint32_t)c);
            NB_STAMP(5);
            const uint32_t lo_ = (uint32_t)c & 15u, un_ = unit < (uint32_t)kSpecUnits ? unit : 0u;
            auto read_staged = [&]() __attribute__((always_inline)) {
              hv = S.hspec[ci & ((1u << SP::hh) - 1u)][lo_];
              sg_row = *(lds_u4_p)lds_off(&S.selrow[un_][lo_]); sg_sel = S.seloff[un_][lo_];
              if (SP::nmix) { const uint32_t jj = ci - SP::mix_j0[0]; sg_mw = (int)S.mixst[0][lo_][jj & 15u]; }
            };
            {
              const uint32_t rdy_v = __hip_atomic_load(&S.mb_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              asm volatile("" ::: "memory");             // (the flag first: what is read behind it is what the flag vouches for)
              read_staged();
              if (UNLIKELY(uni(rdy_v) != bseq)) {        // not yet: wait, then read again
                if (helper_ok) helper_ok = c2_wait(&S.mb_ready, bseq);
                asm volatile("" ::: "memory");
                read_staged();
              }
            }
            if (!helper_ok) return false;                  // the helper wavefront stopped answering (cannot happen by design)
            ++bseq;
            NB_STAMP(6);
            const uint32_t lo = (uint32_t)c & 15u;
            v4u old; uint32_t old_off; bool old_valid;
            row_evict(old, old_off, old_valid);
            Probe pr;
            {
              const uint32_t cxt = hv + 16u;
              pr.chk = (cxt >> sizebits2) & 255;
              pr.h0 = (cxt * 16u) & (ht_mask - 15u);
            }
            if (SP::nmix) {
              const uint32_t rb_was = mx_rb;
              mix_set(rdlane(hv, SP::mix_lane[0]));
              mwl[1] = l_feed ? (mx_rb == rb_was ? w1_new : sg_mw) : 0;
              mix_rows(1u);
            }
            if (SP::match_lane >= 0) {
              match_boundary((uint32_t)c);
              cm_pre = __builtin_amdgcn_raw_buffer_load_b32(rsrc, l_match ? cmo + (hv & cm_mask) * 4u : kOob, 0, 0);
            }
            const bool near = (old1_valid && ((old1_off ^ pr.h0) & ~48u) == 0) || (old_valid && ((old_off ^ pr.h0) & ~48u) == 0);
            v4u row = sg_row; uint32_t sel = sg_sel;
            if (UNLIKELY(__ballot(near) != 0)) {         // something this wave wrote late lies in a probed bucket: the probes, patched
              pr.r0 = *(lds_u4_p)lds_off(&S.rowst[un_][0][lo]);
              pr.r1 = *(lds_u4_p)lds_off(&S.rowst[un_][1][lo]);
              pr.r2 = *(lds_u4_p)lds_off(&S.rowst[un_][2][lo]);
              rows_pick(pr, old1, old1_off, old1_valid, old, old_off, old_valid, false, row, sel);
            }
            if (lane < NC) *(lds_u4_p)lds_off(&S.slot[ci]) = row;
            row_take(row, sel);
            oldb = old; oldb_off = old_off; oldb_valid = old_valid;
            asm volatile("" ::: "memory");
            NB_STAMP(7);
            return true;
      };
      for (;;) {                                       // one decoded byte per iteration
        uint32_t bad = 0, rn = 0, j = 0, err = 0;
        bool after_eos = false;
        // ---- the common case as a loop of its own: post-processor in PASS state, nothing unusual in the byte.  Whatever else
        // happens (priming, end of segment, a renormalisation or a range error at the EOS flag) leaves it for the general form below
        if (LIKELY(pp_state == 1)) {
          for (;;) {
            d.low = uni(d.low); d.high = uni(d.high); d.curr = uni(d.curr);
            if (UNLIKELY(d.curr == 0)) break;
            bad = 0; j = 0;
            ZH_DEC_STEP(d, 0u, j, bad, rn);            // EOS flag: p = 0
            if (UNLIKELY((bad | rn | j) != 0)) { after_eos = true; break; }
            err = 0;
            const uint32_t cb = decode_byte(j, bad, err);
            if (UNLIKELY((err | bad) != 0)) { status = err ? -(int)err : ZH_E_CORRUPT; break; }
            if (UNLIKELY(!boundary((int)cb))) { status = -24; break; }     // ZPAQHIP_E_HIP
            out_put(ob, cb, lane);
            NB_STAMP(9);
          }
          if (status) break;
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
          c = (int)decode_byte(j, bad, err);
          if (UNLIKELY(err | bad)) { status = err ? -(int)err : ZH_E_CORRUPT; break; }
          if (UNLIKELY(!boundary(c))) { status = -24; break; }             // ZPAQHIP_E_HIP
        }

        // ---- PostProcessor.write(c) (PostProcessor.cs:37-86)
        c = (int)uni((uint32_t)c);
        if (LIKELY(pp_state == 1)) {
          if (LIKELY(c >= 0)) out_put(ob, (uint32_t)c, lane);
        } else if (pp_state == 5) {
          int rc;
          if (pnative == ZH_NATIVE_PCOMP_E8E9)
            rc = zh_native_pcomp_e8e9(pa, pb, pc_, pd, pf, (uint32_t)c, (lds_u8_p)lds_off(S.pmreg), pz.mmask, (lds_u32_p)lds_off(S.phreg), pz.hmask, S.pr, &sink, L.budget);
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
            pnative = p_lds ? uni(zh_native_lookup(pzbuf, pp_len)) : 0;
            pp_state = 5;
          }
        }
        NB_STAMP(9);
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
      for (int i = 0; i < 16; ++i) atomicAdd((unsigned long long *)&L.debug[i], (unsigned long long)prof[i]);
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

// spec: 1 min, 2 mid (zh_chain_spec.h ids)
extern "C" hipError_t zh_launch_nibble(const ZhLaunch *L, uint32_t grid, hipStream_t stream, uint32_t spec, int prof) {
  void (*k)(ZhLaunch) = spec == 1 ? (prof ? zh_decode_nb_min_prof : zh_decode_nb_min) : spec == 2 ? (prof ? zh_decode_nb_mid_prof : zh_decode_nb_mid) : nullptr;
  if (!k) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k, dim3(grid), dim3(128), 0, stream, *L);     // decoder wave + helper wave
  return hipGetLastError();
}
extern "C" int zh_nibble_has(uint32_t spec) { return spec == 1 || spec == 2; }
