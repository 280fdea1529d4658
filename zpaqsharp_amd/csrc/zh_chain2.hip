// zh_chain2.hip — the bit loop of the reference's three built-in models (min / mid / max, Compressor.cs:48-74),
// written per model for one gfx950 wavefront.  It replaces the model-specialised instances of zh_chain.hip; that
// file stays as the kernel for any other component chain (and as the cross-check, opts.kernel = 5).
//
// The reference specialises the predictor per block header at run time by emitting x86 code
// (Predictor.assemble_p, Predictor.cs:579-1356).  The analogue here is ahead-of-time: the host selects this kernel
// only on an exact match of the header's COMP section (zh_chain_spec.h, zh_spec_lookup).
//
// What is different from zh_chain.hip (same lane-per-component mapping, same results):
//
//  * ONE-BIT-AHEAD, BOTH-WAYS FETCH.  Inside a nibble the node after j is 2j or 2j+1, and the bit histories of those two
//    nodes are ADJACENT bytes of the hash row (Predictor.cs:269-272: ht[c + (hmap4 & 15)]).  While bit k is being
//    predicted and decoded, every ICM / ISSE lane reads that byte pair and both table entries it selects; when y is
//    known the right one is picked with a v_cndmask.  The only way bit k can change what bit k+1 reads is by training
//    the very entry bit k+1 uses (same bit-history state); that case is detected by comparing entry addresses and
//    served from the registers update() has just computed.  The dependent LDS walk (row byte -> table entry ->
//    stretch) that cost ~670 cycles per bit in zh_chain.hip is off the critical path for 6 of the 8 bits of a byte.
//    Mixer weight rows (Predictor.cs:302-316) are fetched the same way: rows c8*2 and c8*2+1 while bit c8 is decoded.
//  * An ICM entry carries its stretched probability next to it ({cm, stretch(cm >> 8)}, 8 bytes like an ISSE's weight
//    pair), maintained by update(): predict needs no table walk for it.
//  * The ISSE chain (each ISSE takes the prediction of the component before it) is a SYSTOLIC step executed by all
//    lanes: p <- clamp2k((w0 * p[lane-1] + w1x) >> 16), one DPP row_shr:1 multiply, one add, one shift, one med3.
//    Lanes that are not an ISSE run it with w0 = 0 and w1x = p << 16, which reproduces p, so there is no per-level
//    lane select; after `depth` steps every lane holds its final value.
//  * All arena traffic goes through buffer instructions: 32-bit lane offset + scalar row offset, lanes that have
//    nothing to load or store carry an out-of-range offset and are dropped by the range check (no exec-mask code).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zh_core.h"
#include "zh_dev.h"
#include "zh_model.h"
#include "zh_zpaql_native.h"

using namespace zhcore;
using namespace zhdev;

#pragma clang diagnostic ignored "-Wint-to-pointer-cast"

// Build-time variants for same-box A/B runs (tools/ab_bench.sh, tools/build_variants.sh): -DC2V=<mask>.  The shipped build is
// 999: every bit below; 32 for the max model only (profiles/r04/ab_notes.txt).  (Round 4's variants 8 — the helper wave touching
// the lines of the mixer rows ahead: mid +0.6 %, max +1 % for 3.5 TB / 21 TB more HBM reads per GiB — and 16 — a one-way entry
// fetch: fewer instructions, slower — were measured, not shipped, and are gone from the source: profiles/r04/ab_notes.txt.)
//   1  the decoder step hands y to the vector side itself (select mask, ey, y made under the split's SCC: ZH_DEC_STEP_Y)
//   2  what a bit trains but the NEXT bit cannot read — mixer weights (their row changes with every bit), max's SSE entries
//      and `mix2 8` weight — is computed one bit later, in the shadow of that bit's squash look-up (an s_load or ds_read
//      round trip during which round 3's wave issued nothing)
//   4  the hash row of the nibble stays in four VGPRs: the bit histories of the next bit's two candidate nodes are bit
//      fields of a register, not an LDS read in front of the entry reads
//  32  MATCH's prediction as pm0 + bit * (pm1 - pm0) and a miss resetting it to the lane's constant: 7 instructions for 10
//  64  the 12 hash-row requests of the second nibble's candidates go out BEHIND the mixer weights of bit 3 instead of in front
//      of them: vector memory returns in issue order, and round 4's per-bit stamps show mid's bit 2 waiting ~360 cycles for
//      its two weight dwords queued behind those rows (profiles/r04/prof_vm_wait.txt); the rounding addend of the weight
//      updates in a VGPR (a v_mad_i32_i24 whose ADDEND is an SGPR takes 8.3 cycles against 4.9: tools/ubench/sgpr_bench)
// 128  (mid, with 1) the weights of bits 6 and 7 — rows 64-255 of the byte's block, a new line with every bit, the only ones
//      round 4's per-bit stamps still find the wave waiting for (~95 cycles each) — are requested TWO bits ahead, four
//      candidate rows each, and picked by the two bits decoded meanwhile
// 256  (min, mid) Predictor.find at the byte boundary done by the HELPER wave for its 16 candidates (check compare and victim
//      choice on the three probes it holds anyway): the decoder wave reads one row and its place instead of three rows,
//      a patch and the selection — unless a row it evicted after the helper's loads lies in the same bucket (then the
//      round-3 path)
// 512  the byte boundary reads the helper's "ready" word and everything it staged for the byte in ONE batch of LDS reads and
//      checks the word afterwards (the helper is ready ~550 cycles early, round-4 stamps): one LDS round trip where round 3
//      had three in a row (wait for ready, then h[], then the rows)
#ifndef C2V
#define C2V 999
#endif
#define C2_FINDB ((C2V & 256) != 0)
#define C2_SPECRD ((C2V & 512) != 0)

#include "zh_c2_common.h"

namespace {

// Diagnostic build (PROF): cycles per stage, summed per block into L.debug[0..7].  Stamps wait for LDS/scalar results
// only (global memory stays in flight, as in the real kernel).
#ifndef C2_PROF_MASK
#define C2_PROF_MASK 0x0fff                     /* stages the *_prof kernels stamp (fewer stamps: less distortion) */
#endif
#define C2_STAMP(i)                                                                                  \
  do {                                                                                               \
    if (PROF && ((C2_PROF_MASK >> (i)) & 1)) {                                                                                   \
      uint64_t now_;                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      prof[i] += now_ - tprev;                                                                       \
      tprev = now_;                                                                                  \
    }                                                                                                \
  } while (0)

__device__ __forceinline__ void c2_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }   // one wave: no barrier
// scalar x vector 24-bit multiply, by hand: where the compiler can prove that a 24-bit multiply equals the 32-bit one it
// selects v_mul_lo_u32 (+ narrowing moves) for an SGPR x VGPR product; the 24-bit form is one instruction (a dependent
// v_mul_lo_u32 chain issues at the rate of v_add_u32 on a lone gfx950 wave: tools/ubench/exec_bench — the gain is the count)
__device__ __forceinline__ int mul24_sv(int sc, int vec) {
  int r;
  asm("v_mul_i32_i24_e32 %0, %1, %2" : "=v"(r) : "s"(sc), "v"(vec));
  return r;
}


constexpr bool kYsel = (C2V & 1) != 0, kDefer = (C2V & 2) != 0, kRowReg = (C2V & 4) != 0;
constexpr bool kRowsLate = (C2V & 64) != 0, kFar2 = (C2V & 128) != 0 && kYsel;

template <class SP, bool PROF, int HELP, class LDS>
__device__ __forceinline__ void decode_chain2_body(const ZhLaunch &L, LDS &S) {
  constexpr bool kMatch2 = (C2V & 32) != 0 && SP::id == 3;   // (same-box A/B: max + E8E9 +0.6 %, mid -0.7 %)
  uint64_t prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tprev = 0;
  const uint32_t lane = threadIdx.x & 63u;
  const bool wave_a = (threadIdx.x >> 6) == 0;
  constexpr uint64_t kII = SP::icm | SP::isse;
  const bool l_isse = (SP::isse >> lane) & 1, l_ii = (kII >> lane) & 1;
  const bool l_match = SP::match_lane >= 0 && lane == (uint32_t)SP::match_lane;

  if (wave_a) {  // model-independent tables -> LDS (ZhTables: squash, stretch, dt, dt2k, ns)
    const ZhTables *T = L.tables;
    for (uint32_t i = lane; i < 32768 / 8; i += 64) reinterpret_cast<uint4 *>(S.stretch)[i] = reinterpret_cast<const uint4 *>(T->stretch)[i];
    for (uint32_t i = lane; i < 4096 / 8; i += 64) reinterpret_cast<uint4 *>(S.squash)[i] = reinterpret_cast<const uint4 *>(T->squash)[i];
    for (uint32_t i = lane; i < 1024 / 4; i += 64) reinterpret_cast<uint4 *>(S.dt)[i] = reinterpret_cast<const uint4 *>(T->dt)[i];
    for (uint32_t i = lane; i < 1024 / 16; i += 64) reinterpret_cast<uint4 *>(S.ns)[i] = reinterpret_cast<const uint4 *>(T->ns)[i];
    if (lane == 0) { S.zrow = v4u_{0, 0, 0, 0}; S.mb_cmd = 0; S.mb_ack = 0; S.mb_nib = 0; S.mb_byte = 0; S.mb_ready = 0; }
  }
  __syncthreads();                                       // the only workgroup barrier of the kernel
  if constexpr (HELP) { if (!wave_a) { c2_helper<SP, LDS, PROF>(L, S, lane, blockIdx.x); return; } }
  uint32_t cmd_seq = 0;                                  // commands issued to the helper wave
  const uint32_t *ps_tab = reinterpret_cast<const ZhTablesX *>(L.tables + 1)->ps;
  for (uint32_t i = lane; i < 256; i += 64) {            // the two predictions of a match of length i
    const int dk = L.tables->dt2k[i];
    const uint32_t lo = (uint16_t)S.stretch[dk & 32767], hi = (uint16_t)S.stretch[(-dk) & 32767];
    S.pm01[i] = i ? lo | hi << 16 : 0u;
  }
  c2_wave_sync();

  uint8_t *slot_mem = L.arena + (uint64_t)blockIdx.x * L.arena_stride;
  const lds_i16_p lds_stretch = (lds_i16_p)lds_off(S.stretch);
  const lds_u16_p lds_squash = (lds_u16_p)lds_off(S.squash);
  const uint32_t ns_off = lds_off(S.ns);

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
      else if (type == ZH_MIX2) { pat = make_uint4(0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u); fill_cm = true; }
      else if (type == ZH_MIX) { const uint32_t w = 65536u / uni(cp.arg[2]); pat = make_uint4(w, w, w, w); fill_cm = true; }
      if (fill_cm) { uint4 *q = reinterpret_cast<uint4 *>(cm); for (uint64_t k = lane; k < cmb / 16; k += 64) q[k] = pat; }
      if (type == ZH_SSE) {                              // squash((j&31)*64-992)<<17 | start, period 32 entries
        const uint32_t start = uni(cp.arg[2]);
        uint4 *q = reinterpret_cast<uint4 *>(cm);
        for (uint64_t k = lane; k < cmb / 16; k += 64) {
          const uint32_t j = (uint32_t)(k * 4) & 31;
          uint4 v;
          v.x = (uint32_t)S.squash[(j + 0) * 64 - 992 + 2048] << 17 | start;
          v.y = (uint32_t)S.squash[(j + 1) * 64 - 992 + 2048] << 17 | start;
          v.z = (uint32_t)S.squash[(j + 2) * 64 - 992 + 2048] << 17 | start;
          v.w = (uint32_t)S.squash[(j + 3) * 64 - 992 + 2048] << 17 | start;
          q[k] = v;
        }
      }
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
    // ---- ICM / ISSE entry tables in LDS.  Unit u of S.ent belongs to the u-th ICM/ISSE lane.
    const uint32_t unit = (uint32_t)__builtin_popcountll(kII & ((1ull << lane) - 1));
    {
      // the 256 initial entries are the same for every ICM and for every ISSE: compute them once with 64 lanes
      for (uint32_t j = lane; j < 256; j += 64) {
        const uint32_t n0 = S.ns[j * 4 + 2], n1 = S.ns[j * 4 + 3];
        const uint32_t ci = ((n1 * 2 + 1) << 22) / (n0 + n1 + 1);                  // StateTable.cminit
        const int stv = S.stretch[ci >> 8];
        const v2u e_icm = {ci, (uint32_t)stv};
        const v2u e_isse = {1u << 15, (uint32_t)clamp512k(stv * 1024)};
        uint32_t u = 0;
        for (uint32_t i = 0; i < SP::n; ++i) {
          if (!((kII >> i) & 1)) continue;
          S.ent[u][j] = ((SP::icm >> i) & 1) ? e_icm : e_isse;
          ++u;
        }
      }
      if (SP::has_tail) {
        for (uint32_t k = lane; k < 256 * 32; k += 64) S.sse18[k] = (uint32_t)S.squash[(k & 31) * 64 - 992 + 2048] << 17 | C2Max::sse_start;
        for (uint32_t k = lane; k < 256; k += 64) S.a19[k] = 32768;
      }
      S.slot[lane] = v4u{0, 0, 0, 0};
      S.lent[lane] = v2u{0, 0};
      S.lsink[lane] = 0;
    }
    c2_wave_sync();

    // ---- per-lane constants
    const ZhComp *mycp = &M->comp[lane < SP::n ? lane : 0];
    const uint32_t hto = l_ii || l_match ? (uint32_t)mycp->ht_off : 0u, ht_mask = mycp->ht_mask;
    const uint32_t cmo = (uint32_t)mycp->cm_off, cm_mask = mycp->cm_mask;
    const uint32_t sizebits2 = (uint32_t)mycp->arg[0] + 2;
    const uint32_t tab = l_ii ? lds_off(&S.ent[unit][0]) : lds_off(&S.lent[lane]);     // entry table of this lane
    const uint32_t rrow = l_ii ? lds_off(&S.slot[lane]) : lds_off(&S.zrow);            // row it reads bit histories from
    const uint32_t wrow = l_ii ? lds_off(&S.slot[lane]) : lds_off(&S.lsink[lane]);     // ... and writes them to (+ node index)
    const uint32_t wrow_mask = l_ii ? 15u : 0u;
    const int isse_m = l_isse ? -1 : 0;
    const uint32_t cshift = l_isse ? 6u : 16u;
    int pself = 0;                                       // prediction of a lane that is neither ICM nor ISSE (MATCH, CONST)
    if (lane < SP::n && mycp->type == ZH_CONS) pself = ((int)mycp->arg[0] - 128) * 4;
    if (l_match && lane == (uint32_t)SP::match_lane) (slot_mem + hto)[0] = 1;           // Predictor.cs:121 ht(0)=1 ... overwritten like the reference

    // mixers kept in HBM: lane j0+k owns weight k of the current row
    uint32_t vo_mix[2] = {kOob, kOob};                   // lane offset inside a row, out of range for lanes that do not feed it
    uint32_t mx_base[2] = {0, 0}, mx_m4[2] = {0, 0}, mx_size1[2] = {0, 0}, mx_cmask[2] = {0, 0};
    int mx_rate[2] = {0, 0};
#pragma unroll
    for (uint32_t q = 0; q < SP::nmix; ++q) {
      const ZhComp &mc = M->comp[SP::mix_lane[q]];
      mx_base[q] = uni((uint32_t)mc.cm_off);
      mx_m4[q] = SP::mix_m[q] * 4u;
      mx_size1[q] = uni(mc.cm_mask);                      // contexts - 1
      mx_cmask[q] = uni((uint32_t)mc.arg[4]);
      mx_rate[q] = (int)uni((uint32_t)mc.arg[3]);
      asm volatile("" : "+v"(mx_rate[q]));           // held in a VGPR: the error scaling below stays on the vector unit
      if (lane >= SP::mix_j0[q] && lane < SP::mix_j0[q] + SP::mix_m[q]) vo_mix[q] = (lane - SP::mix_j0[q]) * 4u;
    }

    // HCOMP machine (ZPAQL.cs:1010-1026): H and M in LDS
    Vm &hz = S.hz;
    hz.a = hz.b = hz.c = hz.d = hz.f = 0;
    hz.len = uni(M->hcomp_len);
    {
      const uint8_t *gcode = L.code + uni(M->code_off);
      const uint32_t win = hz.len + 2 * ZH_CODE_PAD;
      if (win <= (uint32_t)kCodeBytes) {
        for (uint32_t i = lane; i < win; i += 64) S.code[i] = gcode[i];
        hz.prog = S.code + ZH_CODE_PAD;
      } else hz.prog = gcode + ZH_CODE_PAD;
    }
    hz.hmask = (uint32_t)((1ull << hh) - 1); hz.mmask = (uint32_t)((1ull << hmb) - 1);
    hz.h = S.hreg; hz.m = S.mreg; hz.r = S.r;             // the three models have hh <= 5, hm <= 9
    const uint32_t hmask = hz.hmask;
    const uint32_t hnative = (uni(M->kind) >> 8) & 255;
    uint32_t ha = 0, hb = 0, hc = 0, hd = 0, hf = 0;
    constexpr int kMRegs = SP::id == 3 ? 2 : 1;
    MRegs<kMRegs> mregs;
    for (int i = 0; i < kMRegs; ++i) mregs.v[i] = 0;
    const MView<kMRegs> reg_m{&mregs};                   // M of the native programs (hm <= 8 + log2 kMRegs, checked by the host)
    const lds_u32_p lds_h = (lds_u32_p)lds_off(S.hreg);

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
    c2_wave_sync();
    uint32_t bseq = 1;                                  // bytes of this block decoded so far + 1 (the helper wave's clock)
    bool helper_ok = true;
    if (HELP) {                                         // wake the helper wave for this block (tables and VM memories are ready)
      if (lane == 0) { S.mb_model = model_i; S.mb_nib = 0; S.mb_byte = 0; S.mb_ready = 0; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      ++cmd_seq;
      c2_put0(&S.mb_cmd, cmd_seq << 2 | kC2New);
      helper_ok = c2_wait(&S.mb_ack, cmd_seq << 2 | kC2New);
    }
    if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }   // stage 0 also takes what lies between stamps
    InBuf in;
    in.stream = L.in; in.total = L.in_total; in.cbase = 0; in.k = 0; in.avail = 0; in.cur = 0;

    // ---- state carried from bit to bit
    uint32_t hv = 0;                                    // h[lane] (Predictor.cs:469)
    uint32_t rowoff = 0;                                // arena offset of the hash row held in S.slot[lane]
    uint32_t row_x = 0;                                 // first dword of that row as it was when the nibble began
    uint32_t row_q1 = 0, row_q2 = 0, row_q3 = 0;        // kRowReg: dwords 1-3 likewise (histories of nodes 4-15; zero in lanes without a table)
    bool rowvalid = false;
    uint32_t ea = tab;                                  // LDS address of the entry the current bit uses
    uint32_t st = 0;                                    // its bit-history state
    uint32_t eA = 0, eB = 0;                            // its value
    int mw[2] = {0, 0};                                 // weight of this lane in the current row of each HBM mixer
    uint32_t mrow[2] = {0, 0};                          // buffer offset of this lane's weight in that row
    // MATCH (Predictor.cs:273-287, 382-411): the lane's Component fields
    uint32_t m_len = 0, m_ptr = 0, m_limit = 0, m_byte = 0;
    int pm0 = kMatch2 ? pself : 0, pm1 = 0;             // stretch of -+dt2k[len] for this byte; 0 once the match has failed (kMatch2: see C2V)
    // What the byte boundary will want from HBM is requested half a byte early (match_prefetch, at bit 4): cm_pre is
    // the hash-index entry of the current h[i] (read when h[i] was set: nothing else writes the index before the next
    // boundary), va/vb the first 64 byte pairs of the candidate's verification, mbn/mbc the byte predicted by the
    // candidate / by the match that continues.  The one byte these cannot hold — the one being decoded — is put in at
    // the boundary.
    uint32_t cm_pre = 0, va_pre = 0, vb_pre = 0, mbn_pre = 0, mbc_pre = 0;
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
      if (kMatch2) { pm1 = l_match ? pm1 - pm0 : 0; pm0 = l_match ? pm0 : pself; }   // difference form, see C2V
    };

    // Hash rows of the nibble that starts now (c8 == 1 or 16 <= c8 < 32), Predictor.find (Predictor.cs:550-567).
    // issue: the three candidate rows are requested; `old` is the row this lane holds (just evicted, maybe still in
    // flight to HBM): a candidate at the same place is taken from it, so the requests need not wait for the write-back.
    struct Probe { v4u r0, r1, r2; uint32_t h0, chk; };
    auto rows_issue = [&](uint32_t c8, Probe &pr) __attribute__((always_inline)) {
      const uint32_t cxt = hv + 16u * c8;
      pr.chk = (cxt >> sizebits2) & 255;
      pr.h0 = (cxt * 16u) & (ht_mask - 15u);
      const uint32_t vo = l_ii ? hto + pr.h0 : kOob;
      pr.r0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, 0, 0);
      pr.r1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo ^ 16u, 0, 0);
      pr.r2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo ^ 32u, 0, 0);
    };
    // (olda, oldb: rows this wave evicted after the probe's loads may have been issued — by itself or by the helper wave)
    auto rows_finish2 = [&](const Probe &pr, const v4u &olda, uint32_t olda_off, bool olda_valid, const v4u &old, uint32_t old_off,
                            bool old_valid, bool guard) __attribute__((always_inline)) {
      const uint32_t h0 = pr.h0, h1 = h0 ^ 16u, h2 = h0 ^ 32u;
      v4u r0 = pr.r0, r1 = pr.r1, r2 = pr.r2;
      // an evicted row lands in the new bucket only when two contexts share it.  guard: one test for the wave and the
      // patch behind a branch (the byte boundary); without it the selects run always (the nibble switch, where the four
      // copies of this code would each bring a taken branch)
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
      const uint32_t sel = m0 ? h0 : m1 ? h1 : m2 ? h2 : victim;
      const v4u fresh = {chk, 0, 0, 0};
      const v4u row = m0 ? r0 : m1 ? r1 : m2 ? r2 : fresh;
      *(lds_u4_p)lds_off(&S.slot[lane]) = row;           // lanes without a hash table never read their slot
      rowoff = sel; rowvalid = true;
      row_x = l_ii ? row.x : 0u;                          // bytes 1-3: the histories of nodes 1, 2, 3 (no LDS round trip for them)
      if (kRowReg) { row_q1 = l_ii ? row.y : 0u; row_q2 = l_ii ? row.z : 0u; row_q3 = l_ii ? row.w : 0u; }
    };
    auto rows_finish = [&](const Probe &pr, const v4u &old, uint32_t old_off, bool old_valid) __attribute__((always_inline)) {
      rows_finish2(pr, old, 0u, false, old, old_off, old_valid, false);
    };
    // write the row of the finished nibble back (fire and forget) and hand its content to the caller
    auto row_evict = [&](v4u &old, uint32_t &old_off, bool &old_valid) __attribute__((always_inline)) {
      old = *(lds_u4_p)lds_off(&S.slot[lane]);
      old_off = rowoff; old_valid = rowvalid && l_ii;
      __builtin_amdgcn_raw_buffer_store_b128(old, rsrc, old_valid ? hto + rowoff : kOob, 0, 0);
    };
    // first bit of a nibble: node 1 of the row now in S.slot, read directly
    auto l0_direct = [&]() __attribute__((always_inline)) {
      st = (row_x >> 8) & 255u;
      ea = tab + st * 8u;
      const v2u e = *(lds_u2_p)ea;
      eA = e.x; eB = e.y;
    };
    // Mixer rows (Predictor.cs:302-316: row (h[i] + (c8 & mask)) & (size - 1)).  A block comes here only with the model's
    // own COMP list and HCOMP (zh_framing.cpp), whose mixer contexts are multiples of 256 with mask 255: the 255 rows of
    // a byte are consecutive, mx_rb + c8 * row bytes.
    uint32_t mx_h[2] = {0, 0}, mx_rb[2] = {0, 0};        // h[] of the mixer components; this lane's buffer offset in row 0 of the byte
    auto mix_set = [&](uint32_t q, uint32_t hq) __attribute__((always_inline)) {
      mx_h[q] = uni(hq);
      mx_rb[q] = vo_mix[q] + (mx_base[q] + __umul24(mx_h[q] & mx_size1[q] & ~255u, mx_m4[q]));   // < 2^16 x < 2^8: exact in 24 bits
    };
    // The row's place is a per-lane buffer offset (lane's weight inside the row + row): all of it vector arithmetic, so that
    // no value has to cross from the vector to the scalar unit on the way to the load (round 2 kept the row part in an SGPR:
    // under this kernel's scalar-register pressure the compiler held its operands in VGPRs anyway and paid a
    // v_readfirstlane, ~24 cycles, in front of every bit's loads).  Lanes that do not feed the mixer stay out of range.
    auto mix_row = [&](uint32_t q, uint32_t c8) __attribute__((always_inline)) -> uint32_t {
      return mx_rb[q] + __umul24(c8 & 255u, SP::mix_m[q] * 4u);
    };
    // ---- tail of the max model (components 17-21); the host routes a block here only with the built-in HCOMP, which
    // leaves h[17] = h[18] = h[19] = h[21] = 0 and h[20] = byte << 9 (even), so the two rows a bit can lead to are one
    // aligned row pair of each SSE table
    int w17 = 32768, w21 = 32768;                        // mix2 with a single weight (sizebits 0): kept in registers
    uint32_t w19 = 32768, a19i = 0;                      // mix2 19: current weight and its index in S.a19
    uint32_t row20 = 0;                                  // per lane: entry (lane & 31) of row 2r + (lane >> 5) of `sse 16 19` (HBM)
    uint32_t t_h20 = 0;
    const uint32_t sse20_base = SP::has_tail ? uni((uint32_t)M->comp[20].cm_off) : 0u, sse20_mask = SP::has_tail ? uni(M->comp[20].cm_mask) : 0u;
    auto row20_load = [&](uint32_t c8x) __attribute__((always_inline)) -> uint32_t {
      const uint32_t r = (((t_h20 + (c8x & ~1u)) * 32u) & sse20_mask) * 4u + sse20_base;   // (a per-lane offset: no scalar round trip, see mix_row)
      return __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4u + r, 0, 0);
    };
    auto stretch_u = [&](uint32_t ix) __attribute__((always_inline)) -> int {             // stretch() of a wave-uniform argument (stays a vector value)
      return (int)*(lds_i16_p)((uint32_t)(uintptr_t)lds_stretch + ix * 2u);
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
        rows_issue(1u, pr);
        rows_finish(pr, v4u{0, 0, 0, 0}, 0u, false);
#pragma unroll
        for (uint32_t q = 0; q < SP::nmix; ++q) {
          mix_set(q, 0u);
          mrow[q] = mix_row(q, 1u);
          mw[q] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, mrow[q], 0, 0);
        }
        if (SP::has_tail) { row20 = row20_load(1u); a19i = 1u; w19 = uni((uint32_t)S.a19[1]); }
      }

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
        // (the 8 bit steps below do not repeat decode()'s range test, Decoder.cs:138: a split keeps low <= curr <= high,
        // so only a renormalisation can break it, and every renormalisation but the byte's last — which the EOS step of
        // the next byte covers — raises `bad` itself: dec_renorm_chk)
        if (UNLIKELY(rn)) { if (dec_renorm_chk(d, in, lane, bad)) { status = ZH_E_EOF; break; } }
        int c;
        if (UNLIKELY(j)) {
          if (d.curr != 0) { status = ZH_E_EOS; break; }
          c = -1;
        } else {
          uint32_t c8 = 1;
          // ======== one bit.  NODE = position in the nibble (0..3); the node index is hm (1, 2-3, 4-7, 8-15).
          uint32_t hm = 1;
          uint32_t pairS = 0;                            // bit histories of nodes 2hm, 2hm+1
          C2_STAMP(10);
          Probe spec[4];                                 // candidate rows of the second nibble
          v4u old1 = {0, 0, 0, 0}; uint32_t old1_off = 0; bool old1_valid = false;   // the first nibble's row as it was evicted
          l0_direct();
          // kDefer: what bit k-1 left for bit k's squash shadow (Dq), see C2V
          int rnd12 = 1 << 12;                           // the weight updates' rounding addend
          if (kRowsLate) asm volatile("" : "+v"(rnd12));   // ... in a VGPR (C2V 64)
          struct Dq {
            int p, e, mw[2]; uint32_t mrow[2];
            uint32_t sel18, ti18, c8, a19i, sel20, ti20; int dtv18, dtv20, p17, p18, w19, ey;
          } dq = {};
          auto mix_train = [&](int pp, int ee, const int (&mww)[2], const uint32_t (&mrr)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (uint32_t q = 0; q < SP::nmix; ++q) {     // MIX (Predictor.cs:427-439): error from the mixer lane
              const int eq = mul24_sv((int)rdlane((uint32_t)ee, SP::mix_lane[q]), mx_rate[q]) >> 4;
              const int nmw = med3i(mww[q] + ((__mul24(eq, pp) + rnd12) >> 13), -(1 << 19), (1 << 19) - 1);
              __builtin_amdgcn_raw_buffer_store_b32((uint32_t)nmw, rsrc, mrr[q], 0, 0);
            }
          };
          auto mix2_train = [&](int w, int rate, uint32_t ln, int ee, int pj_, int pk_) __attribute__((always_inline)) -> int {   // Predictor.cs:414-426
            int vrate = rate;
            asm volatile("" : "+v"(vrate));
            const int er = mul24_sv((int)rdlane((uint32_t)ee, ln), vrate) >> 5;
            w += (__mul24(er, pj_ - pk_) + rnd12) >> 13;
            return w < 0 ? 0 : w > 65535 ? 65535 : w;
          };
          auto sse_train = [&](uint32_t pn, int dtv, int eyy) __attribute__((always_inline)) -> uint32_t {                      // Predictor.train, :1031-1036 form
            const uint32_t count = pn & 0x3ffu;
            const int error = eyy - (int)(pn >> 17);
            return pn + ((uint32_t)__mul24(error, dtv) & 0xFFFFFC00u) + (count < C2Max::sse_limit);   // |error| < 2^15, dt < 2^16: the low 32 bits are the reference's wrapping product
          };
          // the part of max's tail training that the next bit cannot read (rows / indices move with c8)
          auto tail_train_late = [&](const Dq &t) __attribute__((always_inline)) {
            if constexpr (SP::id == 3) {
              const uint32_t n18 = sse_train(t.sel18, t.dtv18, t.ey);
              *(lds_u32_p)(lds_off(S.sse18) + ((t.c8 & 255u) * 32u + t.ti18) * 4u) = n18;
              const uint32_t nw19 = (uint32_t)mix2_train(t.w19, C2Max::rate19, 19, t.e, t.p17, t.p18);
              *(lds_u16_p)(lds_off(S.a19) + t.a19i * 2u) = (uint16_t)nw19;
              const uint32_t n20 = sse_train(t.sel20, t.dtv20, t.ey);
              const uint32_t off = ((((t_h20 + t.c8) * 32u + t.ti20) & sse20_mask) * 4u) + sse20_base;
              __builtin_amdgcn_raw_buffer_store_b32(n20, rsrc, lane == 0 ? off : kOob, 0, 0);
            }
          };
          uint32_t rzw = 0;                              // kRowReg: dword 2 or 3 of the row, by the nibble's first bit
          uint32_t y_prev = 0;                           // the bit before (wave-uniform)
          uint64_t ym_prev = 0;                          // ... as a select mask
          int far_w[2][2][4] = {};                       // kFar2: the four candidate rows of bits 6 and 7
#pragma unroll
          for (int bit = 0; bit < 8; ++bit) {
            const bool pre_ii = (bit & 3) != 3;          // the next bit stays in this nibble: fetch both of its nodes
            const bool pre_mx = bit != 7;                // the next bit stays in this byte: fetch both of its mixer rows
            hm = uni(hm); c8 = uni(c8);
            if (PROF && ((C2_PROF_MASK >> 12) & 1)) {   // diagnostic: what is still in flight from the bit before (loads and stores), drained here
              uint64_t now_;
              __builtin_amdgcn_sched_barrier(0);
              asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");
              __builtin_amdgcn_sched_barrier(0);
              prof[12] += now_ - tprev; tprev = now_;
            }
            // ---- (a) requests for the NEXT bit, both ways
            if (SP::match_lane >= 0 && bit == 4) match_prefetch();
            uint32_t ea0 = 0, ea1 = 0, st0 = 0, st1 = 0;
            v2u e0 = {0, 0}, e1 = e0;
            if (pre_ii) {
              if (kRowReg) {
                // nodes 2hm, 2hm+1: bytes of the row held in registers (hm = 1: bytes 2, 3; hm = 2, 3: dword 1; hm = 4..7: dword 2 / 3)
                const uint32_t x = (bit & 3) == 0 ? row_x : (bit & 3) == 1 ? row_q1 : rzw;
                const uint32_t sh = (bit & 3) == 0 ? 16u : y_prev * 16u;
                st0 = __builtin_amdgcn_ubfe(x, sh, 8u);
                st1 = __builtin_amdgcn_ubfe(x, sh + 8u, 8u);
              } else {
                pairS = (bit & 3) == 0 ? row_x >> 16 : (uint32_t)*(lds_u16_p)(rrow + 2u * hm);
              }
            }
            int mwc0[2] = {0, 0}, mwc1[2] = {0, 0};
            uint32_t mrow0[2] = {0, 0}, mrow1[2] = {0, 0};
            const bool far2 = kFar2 && SP::id == 2;        // see C2V 128
            if (pre_mx) {
#pragma unroll
              for (uint32_t q = 0; q < SP::nmix; ++q) {
                // c8 < 128 here (no mask); v_mad_u32_u24 by hand: the compiler turns this into v_mad_u64_u32 and narrowing moves,
                // whatever the source says about the operands' width
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(mrow0[q]) : "s"(c8), "v"(SP::mix_m[q] * 8u), "v"(mx_rb[q]));
                mrow1[q] = mrow0[q] + SP::mix_m[q] * 4u;
                if (!(far2 && bit >= 5)) {               // (bits 6 and 7: requested two bits ago)
                  mwc0[q] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, mrow0[q], 0, 0);
                  mwc1[q] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, mrow1[q], 0, 0);
                }
                if (far2 && (bit == 4 || bit == 5)) {    // rows 4 c8 .. 4 c8 + 3: what the bit after next can use
                  const uint32_t r4 = mrow0[q] * 2u - mx_rb[q];            // mx_rb + 4 c8 * row bytes
#pragma unroll
                  for (uint32_t i = 0; i < 4; ++i) far_w[bit - 4][q][i] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, r4 + i * (SP::mix_m[q] * 4u), 0, 0);
                }
              }
            }
            if (bit == 2 && kRowsLate && SP::nmix > 0) {   // (see C2V 64: behind this bit's weight requests; c8 is what it was at the end of bit 1)
#pragma unroll
              for (uint32_t k = 0; k < 4; ++k) rows_issue(c8 * 4u + k, spec[k]);
            }
            uint32_t row20n = 0, w19n0 = 0, w19n1 = 0;
            if (SP::has_tail && pre_mx) {
              row20n = row20_load(c8 * 2u);
              const uint32_t wp = *(lds_u32_p)(lds_off(S.a19) + ((c8 * 2u) & 254u) * 2u);       // entries 2c8, 2c8+1
              w19n0 = wp & 0xffffu; w19n1 = wp >> 16;       // (stay on the vector unit: every reader is a vector instruction)
            }
            // ---- (b) predict: operands of the systolic ISSE step
            int xs = pself;
            if (SP::match_lane >= 0) {                   // MATCH predicts the next bit of the byte it points at
              const uint32_t cbit = (m_byte >> (7 - bit)) & 1;
              if (kMatch2) xs = __mul24((int)cbit, pm1) + pm0;       // pm1 holds the DIFFERENCE here; other lanes: pm0 = pself, pm1 = 0
              else xs = l_match ? (cbit ? pm1 : pm0) : pself;
            }
            const int x = l_ii ? (int)eB : xs;
            const int cw0 = (int)eA & isse_m;
            const int cw1m = (int)((uint32_t)x << cshift);
            int p = x;
            if (pre_ii) {                    // second half of (a): the two entries (their addresses came back)
              if (!kRowReg) { st0 = pairS & 255u; st1 = pairS >> 8; }
              ea0 = tab + st0 * 8u;
              ea1 = tab + st1 * 8u;
              e0 = *(lds_u2_p)ea0;
              e1 = *(lds_u2_p)ea1;
            }
#pragma unroll
            for (uint32_t t = 0; t < SP::depth; ++t) p = med3i((__mul24(shr1(p), cw0) + cw1m) >> 16, -2048, 2047);
            C2_STAMP(0);
            // ---- (c) mixers
            int p15 = 0, p16 = 0, p17 = 0, p18 = 0, p19 = 0, p20 = 0;       // max: the serial tail runs on the scalar unit
            uint32_t sel18 = 0, sel20 = 0, ti18 = 0, ti20 = 0;
            int dtv18 = 0, dtv20 = 0;
            if constexpr (SP::id == 3) {
              // MIX 15 over lanes 0-14, MIX 16 over lanes 0-15 (Predictor.cs:302-316): row sums land in lane 15
              // (the two row sums are independent up to MIX 16's last input, which is MIX 15's output: they interleave)
              const int w1hi = mw[1] >> 8;
              int t0 = __mul24(mw[0] >> 8, p);
              int t1 = lane == 15u ? 0 : __mul24(w1hi, p);
              t0 += dpp_shr(t0, 1); t1 += dpp_shr(t1, 1);
              t0 += dpp_shr(t0, 2); t1 += dpp_shr(t1, 2);
              t0 += dpp_shr(t0, 4); t1 += dpp_shr(t1, 4);
              t0 += dpp_shr(t0, 8); t1 += dpp_shr(t1, 8);
              // (shift before the v_readlane, clamp after it as a vector instruction taking the SGPR: a scalar instruction
              // between a v_readlane and the vector code that follows costs two hand-overs, ~20 cycles, tools/ubench/step_bench)
              p15 = med3i((int)rdlane((uint32_t)(t0 >> 8), 15), -2048, 2047);
              p = lane == 15u ? p15 : p;
              p16 = med3i(((int)rdlane((uint32_t)t1, 15) + mul24_sv((int)rdlane((uint32_t)w1hi, 15), p15)) >> 8, -2048, 2047);
              // (weights < 2^17, predictions < 2^12, SSE entries >> 10 < 2^22, errors < 2^16: every product of this tail is exact
              // in 24-bit multiplies)
              p17 = (__mul24(w17, p15) + __mul24(65536 - w17, p16)) >> 16;   // MIX2 17 (Predictor.cs:291-301)
              auto sse = [&](int pin, uint32_t rowv, int &pout, uint32_t &sel, uint32_t &ti, int &dtv) __attribute__((always_inline)) {   // dtv: per-lane copy, made scalar in update()
                int pq = pin + 992;                                          // SSE (Predictor.cs:327-340)
                pq = pq < 0 ? 0 : pq > 1983 ? 1983 : pq;
                const uint32_t wt = (uint32_t)pq & 63u, iq = (uint32_t)pq >> 6;
                const uint32_t lo = (c8 & 1u) * 32u + iq;                    // this bit's row is the (c8 & 1) half of the pair
                const uint32_t e0 = rdlane(rowv, lo), e1 = rdlane(rowv, lo + 1u);
                pout = stretch_u((__umul24(e0 >> 10, 64u - wt) + __umul24(e1 >> 10, wt)) >> 13);
                sel = (wt >> 5) ? e1 : e0;                                   // the entry train() will update
                ti = iq + (wt >> 5);
                dtv = S.dt[sel & 0x3ffu];                                    // wanted only after the bit is known
              };
              {  // SSE 18 lives in LDS: its two entries are read where they are needed (one ds_read2 by a vector address — no
                 // row held in a register, no lane select that has to go through the scalar unit)
                int pq = p17 + 992;
                pq = pq < 0 ? 0 : pq > 1983 ? 1983 : pq;
                const uint32_t wt = (uint32_t)pq & 63u, iq = (uint32_t)pq >> 6;                 // iq <= 30
                const uint32_t ea18 = lds_off(S.sse18) + ((c8 & 255u) * 32u + iq) * 4u;
                const uint32_t e0 = *(lds_u32_p)ea18, e1 = *(lds_u32_p)(ea18 + 4u);
                if (kDefer && bit > 0) {                 // (the SSE 18 entry read: ~64 cycles before its data is back)
                  __builtin_amdgcn_sched_barrier(0);
                  mix_train(dq.p, dq.e, dq.mw, dq.mrow);
                  __builtin_amdgcn_sched_barrier(0);
                }
                p18 = (int)*(lds_i16_p)((uint32_t)(uintptr_t)lds_stretch + ((__umul24(e0 >> 10, 64u - wt) + __umul24(e1 >> 10, wt)) >> 13) * 2u);
                sel18 = (wt >> 5) ? e1 : e0;
                ti18 = iq + (wt >> 5);
                dtv18 = S.dt[sel18 & 0x3ffu];
              }
              p19 = (__mul24((int)w19, p17) + __mul24(65536 - (int)w19, p18)) >> 16;  // MIX2 19
              sse(p19, row20, p20, sel20, ti20, dtv20);
              const int p21 = (__mul24(w21, p19) + __mul24(65536 - w21, p20)) >> 16;   // MIX2 21
              p = lane == 16u ? p16 : p;
              p = lane == 17u ? p17 : p;
              p = lane == 19u ? p19 : p;
              p = lane == 21u ? p21 : p;                   // lanes 18 and 20 (SSE) need no squash
            } else if (SP::nmix >= 1) {
              int term = __mul24(mw[0] >> 8, p);         // lanes that do not feed the mixer hold weight 0
              term += dpp_shr(term, 1); term += dpp_shr(term, 2); term += dpp_shr(term, 4);
              if (SP::mix_m[0] > 8) term += dpp_shr(term, 8);
              const int pmx = med3i(term >> 8, -2048, 2047);
              if (lane == SP::mix_lane[0]) p = pmx;
            }
            C2_STAMP(1);
            // ---- (d) decode
            // smem_ps: the final prediction's split factor comes through the scalar cache (ZhTablesX) and every lane's own
            // squash(p), wanted by the update only, follows from LDS under the decoder's shadow (measured: mid +1.4 %, min -2 %)
            uint32_t ps;
            if (SP::smem_ps) {
              uint32_t pv4 = ((uint32_t)p << 2) + 8192u;       // (p + 2048) * 4, scaled on the vector side: the v_readlane feeds the s_load directly
              asm("" : "+v"(pv4));                             // (the compiler would move the arithmetic behind the v_readlane, onto the scalar unit)
              const uint32_t pso = rdlane(pv4, SP::final_lane);
              if (kDefer) asm volatile("s_load_dword %0, %1, %2" : "=s"(ps) : "s"(ps_tab), "s"(pso));
              else asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(ps) : "s"(ps_tab), "s"(pso));
            }
            const int sqp = (int)*(lds_u16_p)((uint32_t)(uintptr_t)lds_squash + (uint32_t)(p + 2048) * 2u);
            const int pj = shr1(p);                        // ISSE update: the prediction of the component before (y-independent)
            if (kDefer) {
              // the look-up is on its way (s_load ~80 cycles, ds_read ~64 + a v_readlane): the wave spends them on the
              // training the bit before left over
              __builtin_amdgcn_sched_barrier(0);
              if (bit > 0) {
                if (SP::id != 3) mix_train(dq.p, dq.e, dq.mw, dq.mrow);
                tail_train_late(dq);
              }
              __builtin_amdgcn_sched_barrier(0);
              if (SP::smem_ps) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ps));
            }
            if (!SP::smem_ps && SP::id == 3) {           // (squash * 2 + 1) << 16 on the vector side (max +0.3 %; min -0.3 %: not there)
              uint32_t psv = ((uint32_t)sqp << 17) | 0x10000u;
              asm("" : "+v"(psv));
              ps = rdlane(psv, SP::final_lane);
            } else if (!SP::smem_ps) ps = (rdlane((uint32_t)sqp, SP::final_lane) * 2 + 1) << 16;
            uint32_t jb = j, xr;
            uint64_t ym = 0;
            uint32_t ey_s = 0, y = 0;
            if (kYsel) ZH_DEC_STEP_Y(d, ps, jb, xr, ym, ey_s, y);
            else ZH_DEC_STEP_LITE(d, ps, jb, xr);
            j = jb;
            if (UNLIKELY(xr < 0x1000000u)) {
              const uint32_t was = bad;
              uint32_t later = 0;                         // after the byte's last bit the next EOS step re-checks by itself
              if (dec_renorm_chk(d, in, lane, bit == 7 ? later : bad) && !err) err = was ? (uint32_t)-ZH_E_CORRUPT : (uint32_t)-ZH_E_EOF;
            }
            if (!kYsel) { y = uni(j & 1); ey_s = y ? 32767u : 0u; }
            const int ey = (int)ey_s;
            auto pick = [&](uint32_t a0, uint32_t a1) __attribute__((always_inline)) -> uint32_t { return kYsel ? sel_y(a0, a1, ym) : (y ? a1 : a0); };
            C2_STAMP(2);
            // ---- (e) update (Predictor.cs:363-461)
            const int e = ey - sqp;
            // bit history of this node: next(state, y) -> row byte
            const uint32_t nsb = *(lds_u8_p)(ns_off + st * 4u + y);
            // ICM (Predictor.cs:375-381) and ISSE (:440-449), every lane computes both
            const uint32_t ncm = eA + (uint32_t)((int)(ey - (int)(eA >> 8)) >> 2);
            const int npst = *(lds_i16_p)((uint32_t)(uintptr_t)lds_stretch + ((ncm >> 7) & 0x1fffeu));
            const int nw0 = med3i((int)eA + ((__mul24(e, pj) + rnd12) >> 13), -(1 << 19), (1 << 19) - 1);
            const int nw1 = med3i((int)eB + ((e + 16) >> 5), -(1 << 19), (1 << 19) - 1);
            if (kDefer) {
              dq.p = p; dq.e = e; dq.ey = ey;
#pragma unroll
              for (uint32_t q = 0; q < SP::nmix; ++q) { dq.mw[q] = mw[q]; dq.mrow[q] = mrow[q]; }
            } else mix_train(p, e, mw, mrow);
            if constexpr (SP::id == 3) {
              w17 = mix2_train(w17, C2Max::rate17, 17, e, p15, p16);
              if (kDefer) {
                dq.sel18 = sel18; dq.ti18 = ti18; dq.dtv18 = dtv18; dq.c8 = c8; dq.a19i = a19i; dq.w19 = (int)w19;
                dq.p17 = p17; dq.p18 = p18; dq.sel20 = sel20; dq.ti20 = ti20; dq.dtv20 = dtv20;
              } else {
                Dq now_{};
                now_.e = e; now_.ey = ey; now_.sel18 = sel18; now_.ti18 = ti18; now_.dtv18 = dtv18; now_.c8 = c8; now_.a19i = a19i; now_.w19 = (int)w19;
                now_.p17 = p17; now_.p18 = p18; now_.sel20 = sel20; now_.ti20 = ti20; now_.dtv20 = dtv20;
                tail_train_late(now_);
              }
              w21 = mix2_train(w21, C2Max::rate21, 21, e, p19, p20);
            }
            if (SP::match_lane >= 0) {                   // MATCH (Predictor.cs:383-384): a miss ends the match
              const uint32_t cbit = (m_byte >> (7 - bit)) & 1;
              const bool miss = cbit != y;
              if (kMatch2) { m_len = miss ? 0u : m_len; pm0 = miss ? pself : pm0; pm1 = miss ? 0 : pm1; }   // (pself is 0 in the MATCH lane; m_byte and pm1 are 0 elsewhere)
              else { m_len = miss ? 0u : m_len; pm0 = miss ? 0 : pm0; pm1 = miss ? 0 : pm1; }
            }
            const uint32_t nA = l_isse ? (uint32_t)nw0 : ncm, nB = l_isse ? (uint32_t)nw1 : (uint32_t)npst;
            *(lds_u2_p)ea = v2u{nA, nB};
            *(lds_u8_p)(wrow + (hm & wrow_mask)) = (uint8_t)nsb;
            C2_STAMP(3);
            // ---- (f) bookkeeping (Predictor.cs:463-474) and hand-over to the next bit
            if (PROF && ((C2_PROF_MASK >> 13) & 1) && pre_mx && SP::nmix > 0) {   // diagnostic: how long the next bit's weights (requested at this bit's start) are still away
              C2_STAMP(13);
              __builtin_amdgcn_sched_barrier(0);
              if (SP::id == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   // (younger than the loads: two weight stores and the SSE 20 store — this bit's, or kDefer: the bit before's)
              else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
              __builtin_amdgcn_sched_barrier(0);
              { const uint64_t t13 = tprev; C2_STAMP(14); prof[bit] += tprev - t13; }   // ... and by bit position (stages 0-6 of a build that stamps only 13 / 14)
            }
            c8 = c8 * 2u + y;
            if (pre_mx) {
#pragma unroll
              for (uint32_t q = 0; q < SP::nmix; ++q) {
                if (far2 && bit >= 5) {                  // the four rows requested at bit - 1, by that bit's y and this one's
                  const int (&f4)[4] = far_w[bit - 5][q];
                  const uint32_t lo = sel_y((uint32_t)f4[0], (uint32_t)f4[2], ym_prev), hi = sel_y((uint32_t)f4[1], (uint32_t)f4[3], ym_prev);
                  mw[q] = (int)pick(lo, hi);
                } else mw[q] = (int)pick((uint32_t)mwc0[q], (uint32_t)mwc1[q]);
                mrow[q] = pick(mrow0[q], mrow1[q]);
              }
            }
            if (SP::has_tail && pre_mx) {
              row20 = row20n;
              w19 = pick(w19n0, w19n1); a19i = c8 & 255u;
            }
            if (pre_ii) {
              hm = hm * 2u + y;
              const uint32_t nea = pick(ea0, ea1);
              const bool same = nea == ea;               // bit k trained the entry bit k+1 predicts from
              st = pick(st0, st1);
              uint32_t cA = pick(e0.x, e1.x), cB = pick(e0.y, e1.y);
              if (kYsel) asm volatile("" : "+v"(cA), "+v"(cB));      // (made here: the compiler would sink the two selects into a branch over `same`)
              eA = same ? nA : cA;
              eB = same ? nB : cB;
              ea = nea;
              if (kRowReg && (bit & 3) == 0) rzw = pick(row_q2, row_q3);
            } else if (bit == 3) {
              // ---- second nibble (Predictor.cs:267-270: c8 & 0xf0 == 16): new rows, requested two bits ago
              v4u old; uint32_t old_off; bool old_valid;
              row_evict(old, old_off, old_valid);
              old1 = old; old1_off = old_off; old1_valid = old_valid;
              // The helper wave is told the first nibble: it now prepares the next byte for the 16 values this one can
              // still take.  What its loads must see of this wave's stores — the last byte boundary's row write-back, the
              // c8 = 1 mixer row written at bit 0 (kDefer: in bit 1's shadow) — has reached memory by now: vector memory
              // operations complete in issue order, and the value published here depends on data requested after those
              // stores (mid, max: bit 3's decision went through mixer weights loaded at bit 2; min has no such load and
              // publishes below, after rows_finish has consumed the rows requested at bit 1).  The two hash rows this wave
              // writes later — the one just evicted and the second nibble's at the byte's end — are patched in from the
              // copies kept here (old1, old) when the staged rows are taken.
              if (HELP && SP::nmix > 0) c2_put0(&S.mb_nib, bseq << 8 | (c8 & 15u));
              switch (c8 & 3u) {                         // wave-uniform: four copies of the selection code, no data selects
                case 0: rows_finish(spec[0], old, old_off, old_valid); break;
                case 1: rows_finish(spec[1], old, old_off, old_valid); break;
                case 2: rows_finish(spec[2], old, old_off, old_valid); break;
                default: rows_finish(spec[3], old, old_off, old_valid); break;
              }
              if (HELP && SP::nmix == 0) c2_put0(&S.mb_nib, bseq << 8 | (c8 & 15u));     // min: see above
              hm = 1;
              l0_direct();
              C2_STAMP(4);
            }
            if (bit == 1 && !(kRowsLate && SP::nmix > 0)) {
              // Two bits of the first nibble are known: the second nibble's context is one of c8*4 .. c8*4+3.  The hash
              // rows of all four are requested now, so that the HBM round trip runs under bits 2 and 3 (a candidate
              // that coincides with the row still held in LDS is patched from it when the nibble ends: rows_finish).
#pragma unroll
              for (uint32_t k = 0; k < 4; ++k) rows_issue(c8 * 4u + k, spec[k]);
            }
            y_prev = y; ym_prev = ym;
            C2_STAMP(8);
          }
          if (kDefer) {                                  // the last bit's left-over training (its rows are not the next byte's)
            if (SP::nmix > 0) mix_train(dq.p, dq.e, dq.mw, dq.mrow);
            tail_train_late(dq);
          }
          if (UNLIKELY(err | bad)) { status = err ? -(int)err : ZH_E_CORRUPT; break; }
          c = (int)(c8 - 256);

          // ---- byte boundary: MATCH (Predictor.cs:391-410), HCOMP, h[], rows of the next byte
          {
            v4u sg_row = {0, 0, 0, 0}; uint32_t sg_sel = 0; int sg_mw[2] = {0, 0};   // what the helper wave staged for this byte's value
            uint32_t ms_a1 = kOob, ms_a2 = kOob;         // MATCH's two stores of this byte: their places now, the stores themselves at ms_store()
            if (SP::match_lane >= 0) {                   // still with the h[i] of the byte just coded (update0 runs before z.run)
              ms_a1 = l_match ? hto + (m_limit & ht_mask) : kOob;
              m_limit = l_match ? (m_limit + 1) & ht_mask : m_limit;
              ms_a2 = l_match ? cmo + (hv & cm_mask) * 4u : kOob;                                                     // (its old value: cm_pre)
            }
            auto ms_store = [&]() __attribute__((always_inline)) {
              if (SP::match_lane >= 0) {
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)c, rsrc, ms_a1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(m_limit, rsrc, ms_a2, 0, 0);
              }
            };
            if (HELP == 1) ms_store();
            if (HELP) {
              // ---- two-wave form: the helper wave ran HCOMP for this byte's value among 16 (and staged what follows)
              c2_put0(&S.mb_byte, bseq << 8 | (uint32_t)c);
              C2_STAMP(11);
              const uint32_t lo_ = (uint32_t)c & 15u, un_ = unit < (uint32_t)kSpecUnits ? unit : 0u;
              auto read_staged = [&]() __attribute__((always_inline)) {
                hv = S.hspec[lane & ((1u << SP::hh) - 1u)][lo_];
                if (HELP == 1 && C2_FINDB) { sg_row = *(lds_u4_p)lds_off(&S.selrow[un_][lo_]); sg_sel = S.seloff[un_][lo_]; }
                if (HELP == 1) {
#pragma unroll
                  for (uint32_t q = 0; q < SP::nmix; ++q) { const uint32_t jj = lane - SP::mix_j0[q]; sg_mw[q] = (int)S.mixst[q][lo_][jj & 15u]; }
                }
              };
              if (C2_SPECRD) {
                const uint32_t rdy_v = __hip_atomic_load(&S.mb_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");             // (the flag first: what is read behind it is what the flag vouches for)
                read_staged();
                if (UNLIKELY(uni(rdy_v) != bseq)) {        // not yet: wait, then read again
                  if (helper_ok) helper_ok = c2_wait(&S.mb_ready, bseq);
                  asm volatile("" ::: "memory");
                  read_staged();
                }
              } else {
                if (helper_ok) helper_ok = c2_wait(&S.mb_ready, bseq);
                asm volatile("" ::: "memory");
                read_staged();
              }
              if (!helper_ok) { status = -24; break; }       // ZPAQHIP_E_HIP: the helper wavefront stopped answering (cannot happen by design)
              ++bseq;
            }
            if (HELP == 1) {
              const uint32_t lo = (uint32_t)c & 15u;
              C2_STAMP(5);
              v4u old; uint32_t old_off; bool old_valid;
              row_evict(old, old_off, old_valid);
              Probe pr;
              {
                const uint32_t cxt = hv + 16u;
                pr.chk = (cxt >> sizebits2) & 255;
                pr.h0 = (cxt * 16u) & (ht_mask - 15u);
                const uint32_t un = unit < (uint32_t)kSpecUnits ? unit : 0u;
                if (!C2_FINDB) {
                  pr.r0 = *(lds_u4_p)lds_off(&S.rowst[un][0][lo]);
                  pr.r1 = *(lds_u4_p)lds_off(&S.rowst[un][1][lo]);
                  pr.r2 = *(lds_u4_p)lds_off(&S.rowst[un][2][lo]);
                }
              }
#pragma unroll
              for (uint32_t q = 0; q < SP::nmix; ++q) {
                mix_set(q, rdlane(hv, SP::mix_lane[q]));
                mrow[q] = mix_row(q, 1u);
                const uint32_t jj = lane - SP::mix_j0[q];
                mw[q] = jj < SP::mix_m[q] ? sg_mw[q] : 0;
              }
              if (SP::match_lane >= 0) {
                match_boundary((uint32_t)c);
                cm_pre = __builtin_amdgcn_raw_buffer_load_b32(rsrc, l_match ? cmo + (hv & cm_mask) * 4u : kOob, 0, 0);
              }
              C2_STAMP(6);
              bool taken = false;
              if (C2_FINDB) {
                const bool near = (old1_valid && ((old1_off ^ pr.h0) & ~48u) == 0) || (old_valid && ((old_off ^ pr.h0) & ~48u) == 0);
                if (LIKELY(__ballot(near) == 0)) {       // nothing this wave wrote late lies in a probed bucket: the helper's answer stands
                  const v4u row = sg_row;
                  const uint32_t sel = sg_sel;
                  *(lds_u4_p)lds_off(&S.slot[lane]) = row;
                  rowoff = sel; rowvalid = true;
                  row_x = l_ii ? row.x : 0u;
                  if (kRowReg) { row_q1 = l_ii ? row.y : 0u; row_q2 = l_ii ? row.z : 0u; row_q3 = l_ii ? row.w : 0u; }
                  taken = true;
                }
              }
              if (!taken) {
                if (C2_FINDB) {
                  const uint32_t un = unit < (uint32_t)kSpecUnits ? unit : 0u;
                  pr.r0 = *(lds_u4_p)lds_off(&S.rowst[un][0][lo]);
                  pr.r1 = *(lds_u4_p)lds_off(&S.rowst[un][1][lo]);
                  pr.r2 = *(lds_u4_p)lds_off(&S.rowst[un][2][lo]);
                }
                rows_finish2(pr, old1, old1_off, old1_valid, old, old_off, old_valid, SP::guard_rows);
              }
              asm volatile("" ::: "memory");
              C2_STAMP(7);
            } else {
            if (!HELP) {
            int rc;
            switch (hnative) {
              case ZH_NATIVE_HCOMP_MIN: rc = zh_native_hcomp_min(ha, hb, hc, hd, hf, (uint32_t)c, reg_m, hz.mmask, lds_h, hmask, S.r, (Sink *)nullptr, L.budget); break;
              case ZH_NATIVE_HCOMP_MID: rc = zh_native_hcomp_mid(ha, hb, hc, hd, hf, (uint32_t)c, reg_m, hz.mmask, lds_h, hmask, S.r, (Sink *)nullptr, L.budget); break;
              case ZH_NATIVE_HCOMP_MAX: rc = zh_native_hcomp_max(ha, hb, hc, hd, hf, (uint32_t)c, reg_m, hz.mmask, lds_h, hmask, S.r, (Sink *)nullptr, L.budget); break;
              default: rc = vm_run(hz, (uint32_t)c, nullptr, L.budget); break;
            }
            rc = (int)uni((uint32_t)rc);
            if (rc) { status = rc; break; }
            hv = lds_h[lane & hmask];
            }
            C2_STAMP(5);
            // (round 5) the row's write-back is issued BEHIND the next byte's loads: vector memory completes in issue order (one
            // vmcnt for loads and stores), so rows_finish below, waiting for probes requested behind a store, also waited for
            // that store's acknowledgement — at every byte boundary of the max model (profiles/r05/nb_stage_notes.txt)
            v4u old = *(lds_u4_p)lds_off(&S.slot[lane]);
            const uint32_t old_off = rowoff;
            const bool old_valid = rowvalid && l_ii;
            Probe pr;
            rows_issue(1u, pr);
#pragma unroll
            for (uint32_t q = 0; q < SP::nmix; ++q) {
              mix_set(q, rdlane(hv, SP::mix_lane[q]));
              mrow[q] = mix_row(q, 1u);
              mw[q] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, mrow[q], 0, 0);
            }
            if (SP::has_tail) {
              t_h20 = rdlane(hv, 20);
              row20 = row20_load(1u);
              a19i = 1u; w19 = uni((uint32_t)S.a19[1]);
            }
            if (HELP != 1) ms_store();                    // (behind the loads above, in front of the index load below)
            __builtin_amdgcn_raw_buffer_store_b128(old, rsrc, old_valid ? hto + old_off : kOob, 0, 0);
            if (SP::match_lane >= 0) {
              match_boundary((uint32_t)c);
              cm_pre = __builtin_amdgcn_raw_buffer_load_b32(rsrc, l_match ? cmo + (hv & cm_mask) * 4u : kOob, 0, 0);
            }
            C2_STAMP(6);
            rows_finish(pr, old, old_off, old_valid);
            asm volatile("" ::: "memory");
            C2_STAMP(7);
            }
          }
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
            c2_wave_sync();
            pz.prog = pzbuf; pz.len = pp_len;
            pz.a = pz.b = pz.c = pz.d = pz.f = 0;
            pnative = p_lds ? uni(zh_native_pcomp_lookup(pzbuf, pp_len)) : 0;
            pp_state = 5;
          }
        }
        C2_STAMP(9);
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
    if (HELP) {                                         // the helper wave leaves the block; its late commit of the last byte
      ++cmd_seq;                                        // (S.mreg / S.hreg) must be in LDS before this wave zeroes them again
      c2_put0(&S.mb_cmd, cmd_seq << 2 | kC2End);
      (void)c2_wait(&S.mb_ack, cmd_seq << 2 | kC2End);
    }
    c2_wave_sync();
  }
  if (HELP) { ++cmd_seq; c2_put0(&S.mb_cmd, cmd_seq << 2 | kC2Exit); }
}

}  // namespace

// The two-wave protocol rests on properties of gfx9-family hardware in its default (non-tgsplit) mode: both waves of the
// workgroup run on one CU and share its vector L1 and its LDS; a wave's vector memory operations complete in issue order
// (one vmcnt for loads and stores).  LLVM AMDGPU memory model, "Memory Model gfx90a/gfx942": in non-threadgroup-split
// mode the wavefronts of a work-group share the L1, so workgroup-scope ordering needs no cache maintenance.
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "zh_chain2.hip: the helper-wave protocol is written for gfx9-family CUs (shared vector L1, in-order vmcnt)"
#endif
#define ZH_CHAIN2_KERNEL(name, spec, prof)                                             \
  extern "C" __global__ __launch_bounds__(spec::helper ? 128 : 64) void name(ZhLaunch L) {  \
    typedef C2LdsT<spec::has_tail, spec::helper> Lds;                                  \
    __shared__ Lds S;                                                                  \
    decode_chain2_body<spec, prof, spec::helper, Lds>(L, S);                           \
  }
ZH_CHAIN2_KERNEL(zh_decode_c2_min, C2Min, false)
ZH_CHAIN2_KERNEL(zh_decode_c2_mid, C2Mid, false)
ZH_CHAIN2_KERNEL(zh_decode_c2_max, C2Max, false)
ZH_CHAIN2_KERNEL(zh_decode_c2_mid_prof, C2Mid, true)
ZH_CHAIN2_KERNEL(zh_decode_c2_max_prof, C2Max, true)

// spec: 1 min, 2 mid, 3 max (zh_chain_spec.h ids).  Returns hipErrorInvalidValue for a spec this file has no kernel for.
extern "C" hipError_t zh_launch_chain2(const ZhLaunch *L, uint32_t grid, hipStream_t stream, uint32_t spec, int prof) {
  void (*k)(ZhLaunch) = spec == 1 ? zh_decode_c2_min : spec == 2 ? zh_decode_c2_mid : spec == 3 ? zh_decode_c2_max : nullptr;
  if (prof && spec >= 2) k = spec == 2 ? zh_decode_c2_mid_prof : zh_decode_c2_max_prof;
  if (!k) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k, dim3(grid), dim3(128), 0, stream, *L);     // decoder wave + helper wave
  return hipGetLastError();
}
extern "C" int zh_chain2_has(uint32_t spec) { return spec >= 1 && spec <= 3; }
