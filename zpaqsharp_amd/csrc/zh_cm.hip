// zh_cm.hip — lane-parallel decode kernel for models that are ONE direct context
// model (n == 1, component CM with >= 9 size bits): BASELINE configs 1-2 ("L1").
//
// One wavefront owns one block.  What makes this path MI355X-shaped:
//
//  * The CM table slice a byte can touch is one 2 KiB "window": the context
//    index is h[0] ^ hmap4 with hmap4 < 512 (Predictor.cs:263-266, :463-474), so
//    all 8 bit-contexts of a byte lie in the 512 entries around h[0].  Windows
//    are cached in LDS (44 x 2 KiB, fully associative, tags held one per lane
//    and matched with a single ballot; LRU by per-lane stamps).  For text-like
//    data every context window lives in LDS for the whole block, so HBM sees only
//    the stream and the plaintext.  Evicted windows are written back / reloaded
//    with coalesced 16-byte accesses.
//  * Within a byte each bit position uses a DIFFERENT table entry, so all the
//    probabilities a byte can need (15 for the first nibble, 240 for the second)
//    are looked up at once by the 64 lanes (squash(stretch(cm>>17)) fused into
//    one 64 KiB LDS table) and the bit-serial arithmetic decoder then picks them
//    with v_readlane: its loop is pure scalar-unit code (Decoder.cs:136-158).
//  * The 8 entries a byte visited are trained in parallel afterwards
//    (Predictor.train, Predictor.cs:486-493 / :1031-1036).
//  * Compressed bytes arrive through a 256-byte register buffer (one dword per
//    lane, next chunk prefetched); plaintext leaves through an LDS stage that is
//    flushed with 16-byte coalesced stores.
//
// Anything this kernel does not specialise (PCOMP programs, unusual HCOMP) runs
// through the same scalar core as the generic kernel, still on the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zh_core.h"
#include "zh_model.h"

using namespace zhcore;

namespace {

constexpr int kWin = 44;                  // LDS-resident CM windows
constexpr uint32_t kNoWin = 0xFFFFFFFFu;
constexpr int kStage = 2048;              // plaintext staging bytes

struct alignas(16) CmLds {
  uint16_t fused[32768];                  // (squash(stretch(x)) * 2 + 1), x = cm >> 17
  int32_t dt[1024];
  uint32_t win[kWin][512];
  uint8_t stage[kStage];
  uint32_t r[256];                        // HCOMP R (generic HCOMP fallback)
  uint32_t pr[256];                       // PCOMP R
};
static_assert(sizeof(CmLds) <= 163840, "LDS budget");

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
  return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v);
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}

// Compressed-byte reader: 256-byte aligned chunks of the stream, one dword per lane.
struct InBuf {
  const uint8_t *stream;                  // whole stream (uniform)
  uint64_t total;                         // stream length
  uint64_t pos, end;                      // absolute byte cursor / end of this segment
  uint32_t cur, nxt;                      // per-lane dwords of chunk(pos) and the next chunk
  uint64_t chunk;                         // index of the chunk held in `cur`
};

__device__ __forceinline__ uint32_t load_chunk(const InBuf &in, uint64_t chunk, uint32_t lane) {
  uint64_t off = (chunk << 8) + 4ull * lane;
  // whole-dword reads; the stream buffer is readable up to its 4-byte rounded end
  return off < in.total ? *reinterpret_cast<const uint32_t *>(in.stream + off) : 0u;
}

__device__ __forceinline__ void in_open(InBuf &in, uint64_t off, uint32_t lane) {
  // Decoder.get() reads the caller's Reader, which does not stop at the segment end
  // (Decoder.cs:112-122): only the end of the stream is EOF.
  in.pos = off; in.end = in.total;
  in.chunk = off >> 8;
  in.cur = load_chunk(in, in.chunk, lane);
  in.nxt = load_chunk(in, in.chunk + 1, lane);
}

// Decoder.get() (Decoder.cs:112-122): next coded byte, or -1 past the segment.
__device__ __forceinline__ int in_get(InBuf &in, uint32_t lane) {
  if (in.pos >= in.end) return -1;
  uint64_t ch = in.pos >> 8;
  if (ch != in.chunk) {                   // crossed into the prefetched chunk
    in.cur = in.nxt;
    in.chunk = ch;
    in.nxt = load_chunk(in, ch + 1, lane);
  }
  uint32_t k = (uint32_t)in.pos & 255;
  ++in.pos;
  return (int)((rdlane(in.cur, k >> 2) >> ((k & 3) * 8)) & 255);
}

// Plaintext writer: bytes collect in LDS and leave in 16-byte coalesced stores.
struct OutBuf {
  uint8_t *base;                          // block's output base (uniform)
  uint64_t cap, len;                      // capacity / bytes produced
  uint64_t flushed;                       // bytes already in HBM
};

__device__ void out_flush(OutBuf &o, CmLds &S, uint32_t lane, bool final) {
  // stage[] holds bytes [flushed, min(len,cap)); LDS index = absolute address & (kStage-1),
  // so 16-byte groups of LDS line up with 16-byte groups of the destination.
  const uint64_t lim = o.len < o.cap ? o.len : o.cap;
  const uintptr_t base = (uintptr_t)o.base;
  const uintptr_t a0 = base + o.flushed, a1 = base + lim;
  if (a0 >= a1) return;
  uintptr_t v0 = (a0 + 15) & ~(uintptr_t)15;           // end of the unaligned head
  if (v0 > a1) v0 = a1;
  uintptr_t v1 = a1 & ~(uintptr_t)15;                  // end of the 16-byte groups
  if (v1 < v0) v1 = v0;
  for (uintptr_t a = a0 + lane; a < v0; a += 64) *(uint8_t *)a = S.stage[a & (kStage - 1)];
  for (uintptr_t a = v0 + 16ull * lane; a < v1; a += 16 * 64)
    *reinterpret_cast<uint4 *>(a) = *reinterpret_cast<const uint4 *>(&S.stage[a & (kStage - 1)]);
  if (final) {
    for (uintptr_t a = v1 + lane; a < a1; a += 64) *(uint8_t *)a = S.stage[a & (kStage - 1)];
    v1 = a1;
  }
  o.flushed = v1 - base;
}

__device__ __forceinline__ void out_put(OutBuf &o, CmLds &S, uint32_t c, uint32_t lane) {
  if (o.len < o.cap) {
    uintptr_t a = (uintptr_t)o.base + o.len;
    if (lane == 0) S.stage[a & (kStage - 1)] = (uint8_t)c;
    ++o.len;
    if (o.len - o.flushed >= kStage - 32) out_flush(o, S, lane, false);
  } else ++o.len;
}

struct Dec { uint32_t low, high, curr; };

// Decoder.decode (Decoder.cs:136-158) on the scalar unit.  p16 = predict()*2+1 or 0.
// Returns the bit, or a negative status.
__device__ __forceinline__ int dec_step(Dec &d, uint32_t p16, InBuf &in, uint32_t lane) {
  uint32_t range = d.high - d.low;
  uint32_t t = d.curr - d.low;
  if (t > range) return ZH_E_CORRUPT;     // curr < low || curr > high
  uint32_t off = (uint32_t)(((uint64_t)range * p16) >> 16);
  int y = t <= off;
  if (y) d.high = d.low + off;
  else d.low = d.low + off + 1;
  while ((d.high ^ d.low) < 0x1000000u) {
    d.high = d.high << 8 | 255;
    d.low = d.low << 8;
    d.low += (d.low == 0);
    int c = in_get(in, lane);
    if (c < 0) return ZH_E_EOF;
    d.curr = d.curr << 8 | (uint32_t)c;
  }
  return y;
}

// Window cache: returns the LDS slot holding window w, loading it on a miss.
__device__ uint32_t win_get(uint32_t w, uint32_t &tag, uint32_t &stamp, uint32_t clock, CmLds &S, uint32_t *table,
                            uint32_t lane) {
  uint64_t hit = __ballot(tag == w);
  uint32_t slot;
  if (hit) slot = (uint32_t)__ffsll((long long)hit) - 1;
  else {
    uint32_t key = lane < (uint32_t)kWin ? stamp : 0xFFFFFFFFu;
    uint32_t m = key;
    for (int o = 32; o; o >>= 1) { uint32_t x = (uint32_t)__shfl_xor((int)m, o); m = x < m ? x : m; }
    uint64_t vm = __ballot(key == m && lane < (uint32_t)kWin);
    slot = (uint32_t)__ffsll((long long)vm) - 1;
    uint32_t old = rdlane(tag, slot);
    uint4 *l = reinterpret_cast<uint4 *>(&S.win[slot][0]);
    if (old != kNoWin) {                   // write the victim back (coalesced, 2 x 1 KiB)
      uint4 *g = reinterpret_cast<uint4 *>(table + (uint64_t)old * 512);
      g[lane] = l[lane];
      g[lane + 64] = l[lane + 64];
    }
    const uint4 *gn = reinterpret_cast<const uint4 *>(table + (uint64_t)w * 512);
    uint4 a = gn[lane], b = gn[lane + 64];
    l[lane] = a;
    l[lane + 64] = b;
    if (lane == slot) tag = w;
  }
  if (lane == slot) stamp = clock;
  return slot;
}

}  // namespace

// In-kernel stamps (diagnostic build only: zh_decode_cm_prof): cycles spent per
// stage of a byte, summed per block and written to L.debug[blockIdx*8 + stage].
#define ZH_STAMP(i)                                                              \
  do {                                                                           \
    if (PROF) {                                                                  \
      uint64_t now_;                                                             \
      __builtin_amdgcn_sched_barrier(0);                                         \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
      __builtin_amdgcn_sched_barrier(0);                                         \
      prof[i] += now_ - tprev;                                                   \
      tprev = now_;                                                              \
    }                                                                            \
  } while (0)

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

    uint32_t tag = kNoWin, stamp = 0;                  // per-lane window cache directory
    uint32_t clock = 1;

    Vm hz;                                             // HCOMP machine (generic form)
    hz.a = hz.b = hz.c = hz.d = hz.f = 0;
    hz.prog = L.code + uni(M->code_off) + ZH_CODE_PAD;
    hz.len = uni(M->hcomp_len);
    hz.m = slot_mem + uni64(M->m_off); hz.mmask = (uint32_t)((1ull << uni(M->hm)) - 1);
    hz.h = reinterpret_cast<uint32_t *>(slot_mem + uni64(M->h_off)); hz.hmask = (uint32_t)((1ull << uni(M->hh)) - 1);
    hz.r = S.r;
    uint32_t h0 = 0;                                   // h[0] = z.H(0)

    int pp_state = 0, pp_hsize = 0;
    uint32_t pp_len = 0;
    Vm pz;
    pz.a = pz.b = pz.c = pz.d = pz.f = 0;
    pz.prog = nullptr; pz.len = 0;
    pz.m = slot_mem + uni64(M->pm_off); pz.mmask = (uint32_t)((1ull << uni(M->pm)) - 1);
    pz.h = reinterpret_cast<uint32_t *>(slot_mem + uni64(M->ph_off)); pz.hmask = (uint32_t)((1ull << uni(M->ph)) - 1);
    pz.r = S.pr;
    uint8_t *pzbuf = slot_mem + uni64(M->pz_off) + ZH_CODE_PAD;

    Dec d;
    d.low = 1; d.high = 0xFFFFFFFFu; d.curr = 0;

    OutBuf ob;
    ob.base = L.out + b_out_off; ob.cap = b_out_cap; ob.len = 0; ob.flushed = 0;
    Sink sink;                                         // used only when a PCOMP program emits output
    sink.out = ob.base; sink.cap = ob.cap; sink.len = 0;

    InBuf in;
    in.stream = L.in; in.total = L.in_total;

    int failed = 0;
    for (uint32_t s = 0; s < n_seg; ++s) {
      const uint32_t si = first_seg + s;
      const uint64_t produced0 = pp_state == 5 ? sink.len : ob.len;
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
      in_open(in, seg_off, lane);

      for (;;) {                                       // one decoded byte per iteration
        // ---- Decoder.decompress prologue (Decoder.cs:36-45)
        if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }
        if (d.curr == 0)
          for (int i = 0; i < 4; ++i) d.curr = d.curr << 8 | (uint32_t)in_get(in, lane);
        int y = dec_step(d, 0, in, lane);              // EOS flag
        if (y < 0) { status = y; break; }
        int c;
        if (y) {
          if (d.curr != 0) { status = ZH_E_EOS; break; }
          c = -1;
        } else {
          ZH_STAMP(0);
          // ---- probabilities for every context this byte can reach
          const uint32_t hm = h0 & cm_mask;
          const uint32_t w = hm >> 9, lo9 = hm & 511, g0 = lo9 >> 4, x = lo9 & 15;
          const uint32_t slot = win_get(w, tag, stamp, clock++, S, table, lane);
          ZH_STAMP(1);
          uint32_t *win = &S.win[slot][0];
          const uint32_t gb = g0 ^ 16;                 // groups of the second nibble: gb ^ n
          const uint32_t ia = (g0 << 4) | (lane & 15);
          uint32_t ib[4], cmb[4], pb[4];
          uint32_t cma = win[ia];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            ib[k] = (((gb & 16) | ((lane >> 4) + 4 * k)) << 4) | (lane & 15);
            cmb[k] = win[ib[k]];
          }
          uint32_t pa = S.fused[cma >> 17];
#pragma unroll
          for (int k = 0; k < 4; ++k) pb[k] = S.fused[cmb[k] >> 17];

          ZH_STAMP(2);
          // ---- first nibble
          uint32_t j = 1;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            y = dec_step(d, rdlane(pa, x ^ j), in, lane);
            if (y < 0) break;
            j = j * 2 + (uint32_t)y;
          }
          if (y < 0) { status = y; break; }
          ZH_STAMP(3);
          const uint32_t n1 = j & 15;
          // ---- second nibble: group (gb ^ n1), held by lanes (ga&3)*16.. in register ga>>2
          const uint32_t ga = (gb ^ n1) & 15, kb = ga >> 2, lb = (ga & 3) * 16;
          uint32_t psel = kb == 0 ? pb[0] : kb == 1 ? pb[1] : kb == 2 ? pb[2] : pb[3];
          uint32_t j2 = 1;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            y = dec_step(d, rdlane(psel, lb + (x ^ j2)), in, lane);
            if (y < 0) break;
            j2 = j2 * 2 + (uint32_t)y;
          }
          if (y < 0) { status = y; break; }
          ZH_STAMP(4);
          const uint32_t n2 = j2 & 15;
          c = (int)(n1 << 4 | n2);

          // ---- train the 8 visited entries (Predictor.train), lanes in parallel
          {
            // first nibble: lane<16 holds position (lane), i.e. slot j = lane ^ x
            uint32_t jj = (lane & 15) ^ x;
            uint32_t tt = 31 - __clz((int)(jj | 1));                       // bit position 0..3
            bool vis = lane < 16 && jj != 0 && jj == ((16 | n1) >> (4 - tt));
            uint32_t yy = (n1 >> (3 - tt)) & 1;
            uint32_t cnt = cma & 0x3ff;
            int err = (int)(yy * 32767) - (int)(cma >> 17);
            uint32_t nv = cma + (((uint32_t)err * (uint32_t)S.dt[cnt]) & 0xFFFFFC00u) + (cnt < limit);
            if (vis) win[ia] = nv;
            // second nibble
            uint32_t cmsel = kb == 0 ? cmb[0] : kb == 1 ? cmb[1] : kb == 2 ? cmb[2] : cmb[3];
            uint32_t isel = kb == 0 ? ib[0] : kb == 1 ? ib[1] : kb == 2 ? ib[2] : ib[3];
            bool vis2 = (lane >> 4) == (ga & 3) && jj != 0 && jj == ((16 | n2) >> (4 - tt));
            uint32_t yy2 = (n2 >> (3 - tt)) & 1;
            uint32_t cnt2 = cmsel & 0x3ff;
            int err2 = (int)(yy2 * 32767) - (int)(cmsel >> 17);
            uint32_t nv2 = cmsel + (((uint32_t)err2 * (uint32_t)S.dt[cnt2]) & 0xFFFFFC00u) + (cnt2 < limit);
            if (vis2) win[isel] = nv2;
          }

          ZH_STAMP(5);
          // ---- HCOMP (Predictor.cs:464-470): h[0] = H(0) after z.run(c)
          if (hk == 1) h0 = (uint32_t)c << hshift;     // "a<<= K  *d=a  halt"
          else {
            int rc = vm_run(hz, (uint32_t)c, nullptr, L.budget);
            if (rc) { status = rc; break; }
            h0 = uni(hz.h[0]);
          }
        }

        // ---- PostProcessor.write(c) (PostProcessor.cs:37-86)
        if (pp_state == 1) {
          if (c >= 0) out_put(ob, S, (uint32_t)c, lane);
        } else if (pp_state == 5) {
          if (ob.flushed < ob.len) out_flush(ob, S, lane, true);
          int rc = 0;
          if (lane == 0) rc = vm_run(pz, (uint32_t)c, &sink, L.budget);
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
          if (lane == 0) pzbuf[pp_len] = (uint8_t)c;
          if ((int)++pp_len == pp_hsize) {
            __syncthreads();
            pz.prog = pzbuf; pz.len = pp_len;
            pz.a = pz.b = pz.c = pz.d = pz.f = 0;
            pp_state = 5;
          }
        }
        ZH_STAMP(6);
        if (c < 0) break;
      }

      if (pp_state != 5) out_flush(ob, S, lane, true);
      // a PCOMP program keeps its sink length in lane 0
      uint64_t produced = pp_state == 5 ? uni64(sink.len) : ob.len;
      if (pp_state == 5) sink.len = produced;
      if (!status && produced > b_out_cap) status = ZH_E_OUTPUT_FULL;
      if (status && status != ZH_E_OUTPUT_FULL) failed = 1;
      if (lane == 0) {
        ZhSegResult res;
        res.status = status; res.pp_state = (uint32_t)pp_state;
        res.out_off = b_out_off + produced0; res.out_len = produced - produced0;
        res.in_used = in.pos - seg_off;
        L.results[si] = res;
      }
    }
    if (PROF && lane == 0 && L.debug)
      for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long *)&L.debug[i], (unsigned long long)prof[i]);
    __syncthreads();
  }
}

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
