// zh_cm.hip — lane-parallel decode kernel for models that are ONE direct context
// model (n == 1, component CM with >= 9 size bits): BASELINE configs 1-2 ("L1").
//
// One wavefront owns one block.  What makes this path MI355X-shaped:
//
//  * The CM table slice a byte can touch is one 2 KiB "window": the context
//    index is h[0] ^ hmap4 with hmap4 < 512 (Predictor.cs:263-266, :463-474), so
//    all 8 bit-contexts of a byte lie in the 512 entries around h[0].  Windows
//    are cached in LDS (44 x 2 KiB, fully associative: one tag per lane, lookup =
//    one v_cmp + s_ff1; FIFO replacement).  For text-like data every context
//    window stays in LDS for the whole block, so HBM sees only the stream and the
//    plaintext.  Victims are written back / windows loaded with coalesced 16-byte
//    accesses.
//  * Within a byte each bit position uses a DIFFERENT table entry, so all the
//    probabilities a byte can need (15 for the first nibble, 240 for the second)
//    are looked up at once by the 64 lanes (squash(stretch(cm>>17)) fused into
//    one 64 KiB LDS table); lane j of a 16-lane group holds the entry of nibble
//    context j, so the bit-serial decoder selects with v_readlane(j).
//  * The arithmetic decoder (Decoder.decode, Decoder.cs:136-158) is a hand-
//    scheduled 14-instruction scalar sequence per bit.  A lone wave on a CU pays
//    ~4 cycles per instruction and ~20 per taken branch (tools/ubench), so the hot
//    path is straight-line: errors and renormalisation are flags tested with
//    not-taken branches, and (range*p)>>16 is one s_mul_hi_u32 against p<<16.
//  * The 8 entries a byte visited are trained by 8 lanes at once afterwards
//    (Predictor.train, Predictor.cs:486-493 / :1031-1036).
//  * Compressed bytes come from a 256-byte register buffer (one aligned dword per
//    lane); plaintext is packed into dwords on the scalar unit, parked in a VGPR
//    with v_writelane and leaves as one coalesced 256-byte store per 256 bytes.
//
// Anything this kernel does not specialise (PCOMP programs, unusual HCOMP) runs
// through the same scalar core as the generic kernel, still on the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zh_core.h"
#include "zh_dev.h"
#include "zh_model.h"

using namespace zhcore;
using namespace zhdev;

namespace {

constexpr int kWin = 44;                  // LDS-resident CM windows
constexpr uint32_t kNoWin = 0xFFFFFFFFu;

struct alignas(16) CmLds {
  uint16_t fused[32768];                  // squash(stretch(x)) * 2 + 1, x = cm >> 17
  int32_t dt[1024];
  uint32_t win[kWin][512];
  uint32_t r[256];                        // HCOMP R (generic HCOMP fallback)
  uint32_t pr[256];                       // PCOMP R
  Vm hz, pz;                              // cold machine state lives here, not in registers
  Sink sink;                              // output of a PCOMP program (lane 0 only)
};
static_assert(sizeof(CmLds) <= 163840, "LDS budget");

// Window cache miss: pick the next FIFO victim, write it back, load window w.
__device__ __forceinline__ uint32_t win_miss(uint32_t w, uint32_t &tag, uint32_t &fifo, CmLds &S, uint32_t *table,
                                          uint32_t lane) {
  const uint32_t slot = fifo;
  fifo = fifo + 1 == (uint32_t)kWin ? 0 : fifo + 1;
  const uint32_t old = rdlane(tag, slot);
  uint4 *l = reinterpret_cast<uint4 *>(&S.win[slot][0]);
  if (old != kNoWin) {                    // write the victim back (coalesced, 2 x 1 KiB)
    uint4 *g = reinterpret_cast<uint4 *>(table + (uint64_t)old * 512);
    g[lane] = l[lane];
    g[lane + 64] = l[lane + 64];
  }
  const uint4 *gn = reinterpret_cast<const uint4 *>(table + (uint64_t)w * 512);
  uint4 a = gn[lane], b = gn[lane + 64];
  l[lane] = a;
  l[lane + 64] = b;
  if (lane == slot) tag = w;
  return slot;
}

template <bool PROF>
__device__ __forceinline__ void decode_cm_body(const ZhLaunch &L, const uint16_t *__restrict__ fused_g, CmLds &S) {
  uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tprev = 0;
  const uint32_t lane = threadIdx.x;

  {  // model-independent tables -> LDS
    const uint4 *src = reinterpret_cast<const uint4 *>(fused_g);
    uint4 *dst = reinterpret_cast<uint4 *>(S.fused);
    for (uint32_t i = lane; i < sizeof(S.fused) / 16; i += 64) dst[i] = src[i];
    const uint4 *s2 = reinterpret_cast<const uint4 *>(L.tables->dt);
    uint4 *d2 = reinterpret_cast<uint4 *>(S.dt);
    for (uint32_t i = lane; i < sizeof(S.dt) / 16; i += 64) d2[i] = s2[i];
  }
  __syncthreads();

  // per-lane constants of the lane <-> table-entry mapping
  const uint32_t l15 = lane & 15;                      // nibble context j held by this lane
  const uint32_t lgrp = lane >> 4;                     // 16-lane group
  const uint32_t ltt = 31 - __clz((int)(l15 | 1));     // depth of context j in the nibble tree (0..3)
  const uint32_t lsh_vis = 4 - ltt, lsh_y = 3 - ltt;

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
    const ZhComp *cp = &M->comp[0];
    const uint32_t cm_mask = uni(cp->cm_mask);
    const uint32_t limit = (uint32_t)uni(cp->arg[1]) * 4;
    const uint64_t cm_off = uni64(cp->cm_off), cm_bytes = uni64(cp->cm_bytes);
    const uint32_t kind = uni(M->kind);
    const uint32_t hk = (kind >> 8) & 255, hshift = (kind >> 16) & 255;
    uint32_t *table = reinterpret_cast<uint32_t *>(slot_mem + cm_off);

    // Predictor.init: CM table = 0x80000000 (Predictor.cs:103-104); VM memories zeroed.
    {
      const uint4 v = make_uint4(0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u);
      uint4 *q = reinterpret_cast<uint4 *>(table);
      for (uint64_t i = lane; i < cm_bytes / 16; i += 64) q[i] = v;
      const uint64_t h_off = uni64(M->h_off), tail = uni64(M->arena_bytes) - h_off;
      uint4 *z = reinterpret_cast<uint4 *>(slot_mem + h_off);
      for (uint64_t i = lane; i < tail / 16; i += 64) z[i] = make_uint4(0, 0, 0, 0);
      for (uint32_t i = lane; i < 256; i += 64) { S.r[i] = 0; S.pr[i] = 0; }
    }
    __syncthreads();

    uint32_t tag = kNoWin;                             // per-lane window directory (lanes >= kWin never match)
    uint32_t fifo = 0;

    Vm &hz = S.hz;                                     // HCOMP machine (generic form)
    hz.a = hz.b = hz.c = hz.d = hz.f = 0;
    hz.prog = L.code + uni(M->code_off) + ZH_CODE_PAD;
    hz.len = uni(M->hcomp_len);
    hz.m = slot_mem + uni64(M->m_off); hz.mmask = (uint32_t)((1ull << uni(M->hm)) - 1);
    hz.h = reinterpret_cast<uint32_t *>(slot_mem + uni64(M->h_off)); hz.hmask = (uint32_t)((1ull << uni(M->hh)) - 1);
    hz.r = S.r;
    uint32_t h0 = 0;                                   // h[0] = z.H(0)

    int pp_state = 0, pp_hsize = 0;                    // PostProcessor (PostProcessor.cs:12-16)
    uint32_t pp_len = 0;
    Vm &pz = S.pz;
    pz.a = pz.b = pz.c = pz.d = pz.f = 0;
    pz.prog = nullptr; pz.len = 0;
    pz.m = slot_mem + uni64(M->pm_off); pz.mmask = (uint32_t)((1ull << uni(M->pm)) - 1);
    pz.h = reinterpret_cast<uint32_t *>(slot_mem + uni64(M->ph_off)); pz.hmask = (uint32_t)((1ull << uni(M->ph)) - 1);
    pz.r = S.pr;
    uint8_t *pzbuf = slot_mem + uni64(M->pz_off) + ZH_CODE_PAD;

    Dec d;
    d.low = 1; d.high = 0xFFFFFFFFu; d.curr = 0;

    OutBuf ob;
    ob.base = L.out + b_out_off; ob.cap = b_out_cap; ob.len = 0; ob.stored = 0; ob.word = 0; ob.park = 0;
    out_room(ob);
    Sink &sink = S.sink;                               // used only when a PCOMP program emits output
    sink.out = ob.base; sink.cap = ob.cap; sink.len = 0;

    __syncthreads();
    InBuf in;
    in.stream = L.in; in.total = L.in_total; in.cbase = 0; in.k = 0; in.avail = 0; in.cur = 0;

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

      // One decoded byte: Decoder.decompress() (Decoder.cs:32-56) with predict/update folded in.
      // Returns 0..255, -1 at EOS, -2 on error (status set).
      auto decode_byte = [&]() __attribute__((always_inline)) -> int {
        if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }
        // ---- Decoder.decompress prologue (Decoder.cs:36-45)
        if (UNLIKELY(d.curr == 0)) {
          uint32_t cu = 0;
          for (int i = 0; i < 4; ++i) cu = cu << 8 | (uint32_t)in_get(in, lane);
          d.curr = uni(cu);
        }
        uint32_t bad = 0, rn, j = 0;
        d.low = uni(d.low); d.high = uni(d.high); d.curr = uni(d.curr);   // folds away where already scalar
        ZH_DEC_STEP(d, 0u, j, bad, rn);                // EOS flag: p = 0
        if (UNLIKELY(bad)) { status = ZH_E_CORRUPT; return -2; }
        if (UNLIKELY(rn)) { if (dec_renorm(d, in, lane)) { status = ZH_E_EOF; return -2; } }
        int c;
        if (UNLIKELY(j)) {
          if (d.curr != 0) { status = ZH_E_EOS; return -2; }
          c = -1;
        } else {
          ZH_STAMP(0);
          // ---- probabilities for every context this byte can reach
          const uint32_t hm = h0 & cm_mask;
          const uint32_t w = hm >> 9, lo9 = hm & 511, g0 = lo9 >> 4, x = lo9 & 15;
          const uint64_t hit = __ballot(tag == w);
          uint32_t slot;
          if (LIKELY(hit != 0)) slot = (uint32_t)__builtin_ctzll(hit);
          else slot = win_miss(w, tag, fifo, S, table, lane);
          ZH_STAMP(1);
          uint32_t *win = &S.win[slot][0];
          // lane (g, j) reads position j ^ x of a group: first nibble group g0, second nibble groups (g0^16)^n
          const uint32_t pos = l15 ^ x;
          const uint32_t ia = (g0 << 4) | pos;
          const uint32_t ib0 = ((((g0 ^ 16) & 16) | lgrp) << 4) | pos;   // + 64*k entries for k = 0..3
          const uint32_t cma = win[ia];
          uint32_t cmb[4], pb[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) cmb[k] = win[ib0 + 64 * k];
          const uint32_t pa = (uint32_t)S.fused[cma >> 17] << 16;
#pragma unroll
          for (int k = 0; k < 4; ++k) pb[k] = (uint32_t)S.fused[cmb[k] >> 17] << 16;
          ZH_STAMP(2);

          // ---- first nibble: context j lives in lane j
          // A failed renormalisation (end of stream) is recorded, not branched on: the coder
          // keeps running on garbage for at most a nibble; the FIRST error is what is reported.
          uint32_t err = 0;
          j = 1;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const uint32_t ps = rdlane(pa, j);
            ZH_DEC_STEP(d, ps, j, bad, rn);
            if (UNLIKELY(rn)) { if (dec_renorm(d, in, lane) && !err) err = bad ? (uint32_t)-ZH_E_CORRUPT : (uint32_t)-ZH_E_EOF; }
          }
          if (UNLIKELY(err | bad)) { status = err ? -(int)err : ZH_E_CORRUPT; return -2; }
          ZH_STAMP(3);
          const uint32_t n1 = j & 15;
          // ---- second nibble: group ((g0 ^ n1) & 15) is held by lane group (ga & 3), register ga >> 2
          const uint32_t ga = uni((g0 ^ n1) & 15), kb = ga >> 2, lb = (ga & 3) * 16;
          const uint32_t psel = kb == 0 ? pb[0] : kb == 1 ? pb[1] : kb == 2 ? pb[2] : pb[3];
          uint32_t j2 = 1;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const uint32_t ps = rdlane(psel, lb + j2);
            ZH_DEC_STEP(d, ps, j2, bad, rn);
            if (UNLIKELY(rn)) { if (dec_renorm(d, in, lane) && !err) err = bad ? (uint32_t)-ZH_E_CORRUPT : (uint32_t)-ZH_E_EOF; }
          }
          if (UNLIKELY(err | bad)) { status = err ? -(int)err : ZH_E_CORRUPT; return -2; }
          ZH_STAMP(4);
          const uint32_t n2 = j2 & 15;
          c = (int)(n1 << 4 | n2);

          // ---- train the 8 visited entries (Predictor.train), one pass over the lanes:
          // lane group (ga&3) updates the second-nibble entries, group (ga&3)^1 the first-nibble ones.
          {
            const bool isb = lgrp == (ga & 3);
            const bool isa = lgrp == ((ga & 3) ^ 1);
            const uint32_t cmsel = kb == 0 ? cmb[0] : kb == 1 ? cmb[1] : kb == 2 ? cmb[2] : cmb[3];
            const uint32_t cm = isb ? cmsel : cma;
            const uint32_t idx = isb ? ib0 + 64 * kb : ia;
            const uint32_t nib = isb ? n2 : n1;
            const bool vis = (isa || isb) && l15 != 0 && l15 == ((16 | nib) >> lsh_vis);
            const uint32_t yy = (nib >> lsh_y) & 1;
            const uint32_t cnt = cm & 0x3ff;
            const int err = (int)(yy * 32767) - (int)(cm >> 17);
            const uint32_t nv = cm + (((uint32_t)err * (uint32_t)S.dt[cnt]) & 0xFFFFFC00u) + (cnt < limit);
            if (vis) win[idx] = nv;
          }
          ZH_STAMP(5);

          // ---- HCOMP (Predictor.cs:464-470): h[0] = H(0) after z.run(c)
          if (LIKELY(hk == ZH_HK_SHIFT)) h0 = (uint32_t)c << hshift;     // "a<<= K  *d=a  halt"
          else {
            int rc = (int)uni((uint32_t)vm_run(hz, (uint32_t)c, nullptr, L.budget));
            if (rc) { status = rc; return -2; }
            h0 = uni(hz.h[0]);
          }
        }

        return (int)uni((uint32_t)c);
      };

      for (;;) {
        if (LIKELY(pp_state == 1)) {
          // ---- steady state: PASS post-processor (PostProcessor.cs:49-51).  Nothing of the cold
          // state machine below is live in this loop.
          // Entering the steady-state loop: pin every loop-carried scalar to the scalar unit, so
          // that LLVM's uniformity analysis sees a loop whose state is uniform on entry and on the
          // back edge (a value it believes divergent anywhere outside would otherwise drag the
          // whole loop onto the vector unit with exec-mask control flow).
          d.low = uni(d.low); d.high = uni(d.high); d.curr = uni(d.curr);
          in.cbase = uni64(in.cbase); in.k = uni(in.k); in.avail = uni(in.avail);
          ob.len = uni64(ob.len); ob.stored = uni64(ob.stored); ob.room = uni(ob.room); ob.word = uni(ob.word);
          h0 = uni(h0); fifo = uni(fifo);
          int c;
          for (;;) {
            c = (int)uni((uint32_t)decode_byte());
            if (UNLIKELY(c < 0)) break;
            out_put(ob, (uint32_t)c, lane);
            ZH_STAMP(6);
          }
          break;                                          // EOS (c == -1) or error (c == -2)
        }
        int c = decode_byte();
        if (c == -2) break;
        // ---- PostProcessor.write(c) (PostProcessor.cs:37-86)
        if (pp_state == 5) {
          // every lane runs the program (same inputs, same stores): keeps control flow wave-uniform
          int rc = (int)uni((uint32_t)vm_run(pz, (uint32_t)c, &sink, L.budget));
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
          pzbuf[pp_len] = (uint8_t)c;                  // all lanes store the same byte
          if ((int)++pp_len == pp_hsize) {
            __syncthreads();
            pz.prog = pzbuf; pz.len = pp_len;
            pz.a = pz.b = pz.c = pz.d = pz.f = 0;
            pp_state = 5;
          }
        }
        if (c < 0) break;
      }

      if (pp_state != 5) out_flush(ob, lane);
      uint64_t produced = pp_state == 5 ? uni64(sink.len) : ob.len;
      if (!status && produced > b_out_cap) status = ZH_E_OUTPUT_FULL;
      if (status && status != ZH_E_OUTPUT_FULL) failed = 1;
      if (lane == 0) {
        ZhSegResult res;
        res.status = status; res.pp_state = (uint32_t)pp_state;
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

extern "C" __global__ __launch_bounds__(64) void zh_decode_cm(ZhLaunch L, const uint16_t *__restrict__ fused_g) {
  __shared__ CmLds S;
  decode_cm_body<false>(L, fused_g, S);
}

extern "C" __global__ __launch_bounds__(64) void zh_decode_cm_prof(ZhLaunch L, const uint16_t *__restrict__ fused_g) {
  __shared__ CmLds S;
  decode_cm_body<true>(L, fused_g, S);
}

extern "C" hipError_t zh_launch_cm_prof(const ZhLaunch *L, const uint16_t *fused, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(zh_decode_cm_prof, dim3(grid), dim3(64), 0, stream, *L, fused);
  return hipGetLastError();
}

extern "C" hipError_t zh_launch_cm(const ZhLaunch *L, const uint16_t *fused, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(zh_decode_cm, dim3(grid), dim3(64), 0, stream, *L, fused);
  return hipGetLastError();
}
