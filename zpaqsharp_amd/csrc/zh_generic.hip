// zh_generic.hip — generic ZPAQ block decoder for gfx950: any model, any ZPAQL.
//
// One wavefront owns one block at a time and pulls blocks from a device-scope
// work queue.  All 64 lanes initialise the block's model tables in the arena
// slot (coalesced 16-byte stores) and stage the model-independent tables in
// LDS; lane 0 then runs the bit-serial chain:
//     Decoder.decompress (Decoder.cs:32-68)  ->  Predictor.predict0 /
//     update0 (Predictor.cs:245-475)  ->  ZPAQL.run0 (ZPAQL.cs:1028-1265)
//     ->  PostProcessor.write (PostProcessor.cs:37-86).
// This kernel is the always-correct path for arbitrary headers (n up to 255,
// any HCOMP/PCOMP); zh_cm.hip (single direct CM), zh_chain2.hip (the built-in
// min / mid / max models) and zh_chain.hip (any chain of <= 64 components) hold
// the lane-parallel kernels the host routes recognised models to.  Nothing here
// runs on the CPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zh_core.h"
#include "zh_model.h"
#include "zh_zpaql_pcomp.h"

using namespace zhcore;

namespace {

struct Coder { uint32_t low, high, curr; };

// Decoder.cs:136-158.  y in bit 0; negative = status.
__device__ __forceinline__ int decode_bit(Coder &d, Src &in, uint32_t p) {
  if (d.curr < d.low || d.curr > d.high) return ZH_E_CORRUPT;
  uint32_t mid = d.low + (uint32_t)(((uint64_t)(d.high - d.low) * p) >> 16);
  int y;
  if (d.curr <= mid) { y = 1; d.high = mid; }
  else { y = 0; d.low = mid + 1; }
  while ((d.high ^ d.low) < 0x1000000u) {
    d.high = d.high << 8 | 255;
    d.low = d.low << 8;
    d.low += (d.low == 0);
    int c = src_get(in);
    if (c < 0) return ZH_E_EOF;
    d.curr = d.curr << 8 | (uint32_t)c;
  }
  return y;
}

struct PostProc {      // PostProcessor.cs:12-16
  int state, hsize;
  uint32_t plen;       // PCOMP bytes loaded so far
  uint32_t native;     // zh_zpaql_pcomp.h: ahead-of-time translation of the loaded program, 0 = interpret
  Vm z;
};

// PostProcessor.cs:37-86.  c is 0..255 or -1.  Returns 0 or a status.
__device__ int pp_write(PostProc &pp, int c, Sink &out, const ZhModel *M, uint8_t *slot, uint64_t budget, uint32_t *pimm) {
  switch (pp.state) {
    case 0:
      if (c < 0) return ZH_E_PP_EOS;
      pp.state = c + 1;
      if (pp.state > 2) return ZH_E_PP_TYPE;
      break;
    case 1:
      if (c >= 0) sink_put(out, (uint32_t)c);
      break;
    case 2:
      if (c < 0) return ZH_E_PP_EOS;
      pp.hsize = c;
      pp.state = 3;
      break;
    case 3:
      if (c < 0) return ZH_E_PP_EOS;
      pp.hsize += c * 256;
      if (pp.hsize < 1) return ZH_E_PP_EMPTY;
      pp.plen = 0;
      pp.state = 4;
      break;
    case 4: {
      if (c < 0) return ZH_E_PP_EOS;
      uint8_t *buf = slot + M->pz_off + ZH_CODE_PAD;
      buf[pp.plen++] = (uint8_t)c;
      if ((int)pp.plen == pp.hsize) {
        // z.initp(): H/M were zeroed with the slot; registers start at 0 (ZPAQL.cs:1010-1026)
        pp.z.prog = buf;
        pp.z.len = pp.plen;
        pp.z.a = pp.z.b = pp.z.c = pp.z.d = pp.z.f = 0;
        pp.native = zh_pcomp_lookup(buf, pp.plen);
        zh_pcomp_operands(pp.native, buf, pimm);
        pp.state = 5;
      }
      break;
    }
    default:
      if (pp.native) {
        ZhPcRegs r{pp.z.a, pp.z.b, pp.z.c, pp.z.d, pp.z.f, 0};
        r = zh_pcomp_call(pp.native, r, (uint32_t)c, pp.z.m, pp.z.mmask, pp.z.h, pp.z.hmask, pp.z.r, &out, budget, pimm);
        pp.z.a = r.a; pp.z.b = r.b; pp.z.c = r.c; pp.z.d = r.d; pp.z.f = r.f;
        return r.rc;
      }
      return vm_run(pp.z, (uint32_t)c, &out, budget);
  }
  return 0;
}

__device__ void fill16(uint8_t *dst, uint64_t bytes, uint4 pat, uint32_t lane) {
  uint4 *q = reinterpret_cast<uint4 *>(dst);
  for (uint64_t i = lane; i < bytes / 16; i += 64) q[i] = pat;
}

// Predictor.init() per component (Predictor.cs:94-167) + ZPAQL.init (ZPAQL.cs:1010-1026),
// executed by all 64 lanes.
__device__ void init_slot(const ZhModel *M, uint8_t *slot, GenLds *S, uint32_t lane) {
  const uint4 z4 = make_uint4(0, 0, 0, 0);
  for (uint32_t i = 0; i < M->n; ++i) {
    const ZhComp &cp = M->comp[i];
    uint8_t *cm = slot + cp.cm_off, *ht = slot + cp.ht_off;
    switch (cp.type) {
      case ZH_CM:
        fill16(cm, cp.cm_bytes, make_uint4(0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u), lane);
        break;
      case ZH_ICM:
        fill16(ht, cp.ht_bytes, z4, lane);
        for (uint32_t j = lane; j < 256; j += 64) {
          uint32_t n0 = S->t.ns[j * 4 + 2], n1 = S->t.ns[j * 4 + 3];
          ((uint32_t *)cm)[j] = ((n1 * 2 + 1) << 22) / (n0 + n1 + 1);       // StateTable.cminit
        }
        break;
      case ZH_MATCH:
        fill16(cm, cp.cm_bytes, z4, lane);
        fill16(ht, cp.ht_bytes, z4, lane);
        for (uint64_t j = (cp.ht_bytes & ~15ull) + lane; j < cp.ht_bytes; j += 64) ht[j] = 0;
        break;
      case ZH_MIX2:
        fill16(cm, cp.cm_bytes, make_uint4(0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u), lane);
        for (uint64_t j = (cp.cm_bytes & ~15ull) / 2 + lane; j < cp.cm_bytes / 2; j += 64) ((uint16_t *)cm)[j] = 32768;
        break;
      case ZH_MIX: {
        uint32_t w = 65536u / cp.arg[2];
        fill16(cm, cp.cm_bytes, make_uint4(w, w, w, w), lane);
        for (uint64_t j = (cp.cm_bytes & ~15ull) / 4 + lane; j < cp.cm_bytes / 4; j += 64) ((uint32_t *)cm)[j] = w;
        break;
      }
      case ZH_ISSE:
        fill16(ht, cp.ht_bytes, z4, lane);
        for (uint32_t j = lane; j < 256; j += 64) {
          uint32_t n0 = S->t.ns[j * 4 + 2], n1 = S->t.ns[j * 4 + 3];
          uint32_t ci = ((n1 * 2 + 1) << 22) / (n0 + n1 + 1);
          ((int *)cm)[j * 2] = 1 << 15;
          ((int *)cm)[j * 2 + 1] = clamp512k(S->t.stretch[ci >> 8] * 1024);
        }
        break;
      case ZH_SSE: {
        // cm[j] = squash((j&31)*64-992)<<17 | start ; period 32 entries = 128 bytes
        uint32_t start = cp.arg[2];
        uint4 *q = reinterpret_cast<uint4 *>(cm);
        for (uint64_t k = lane; k < cp.cm_bytes / 16; k += 64) {
          uint32_t j = (uint32_t)(k * 4) & 31;
          uint4 v;
          v.x = (uint32_t)S->t.squash[(j + 0) * 64 - 992 + 2048] << 17 | start;
          v.y = (uint32_t)S->t.squash[(j + 1) * 64 - 992 + 2048] << 17 | start;
          v.z = (uint32_t)S->t.squash[(j + 2) * 64 - 992 + 2048] << 17 | start;
          v.w = (uint32_t)S->t.squash[(j + 3) * 64 - 992 + 2048] << 17 | start;
          q[k] = v;
        }
        break;
      }
      default: break;
    }
  }
  // VM memories: everything from h_off to the end of the slot is zero-filled.
  fill16(slot + M->h_off, M->arena_bytes - M->h_off, z4, lane);
}

// Decoder.decompress() for one byte (Decoder.cs:32-68).  Returns 0..255, -1 (EOS) or a status < -1.
__device__ int decode_byte(Pred &P, Coder &d, Src &in, uint64_t budget) {
  if (P.n) {
    if (d.curr == 0)
      for (int i = 0; i < 4; ++i) d.curr = d.curr << 8 | (uint32_t)src_get(in);
    int y = decode_bit(d, in, 0);
    if (y < 0) return y - 100;
    if (y) return d.curr != 0 ? ZH_E_EOS - 100 : -1;
    int c = 1;
    while (c < 256) {
      uint32_t p = (uint32_t)predict(P) * 2 + 1;
      y = decode_bit(d, in, p);
      if (y < 0) return y - 100;
      c += c + y;
      int rc = update(P, y, budget);
      if (rc) return rc - 100;
    }
    return c - 256;
  }
  if (d.curr == 0) {
    for (int i = 0; i < 4; ++i) d.curr = d.curr << 8 | (uint32_t)src_get(in);
    if (d.curr == 0) return -1;
  }
  --d.curr;
  return src_get(in);
}

}  // namespace

extern "C" __global__ __launch_bounds__(64) void zh_decode_generic(ZhLaunch L) {
  __shared__ GenLds S;
  const uint32_t lane = threadIdx.x;

  {  // stage the model-independent tables in LDS (coalesced 16-byte loads)
    const uint4 *src = reinterpret_cast<const uint4 *>(L.tables);
    uint4 *dst = reinterpret_cast<uint4 *>(&S.t);
    for (uint32_t i = lane; i < sizeof(ZhTables) / 16; i += 64) dst[i] = src[i];
  }
  __syncthreads();

  uint8_t *slot = L.arena + (uint64_t)blockIdx.x * L.arena_stride;

  for (;;) {
    uint32_t bi = 0;
    if (lane == 0) bi = atomicAdd(L.queue, 1u);
    bi = __shfl(bi, 0);
    if (bi >= L.n_blocks) break;                       // every wave reaches this exit

    const ZhBlockDesc bd = L.blocks[bi];
    const ZhModel *M = &L.models[bd.model];
    const uint32_t n = M->n;

    init_slot(M, slot, &S, lane);
    for (uint32_t i = lane; i < 256; i += 64) {
      S.p[i] = 0; S.h[i] = 0; S.r[i] = 0; S.pr[i] = 0;
      S.cs[i] = CompSt{0, 0, 0, 0, 0};
    }
    if (n <= ZH_MAX_LDS_COMP)
      for (uint32_t i = lane; i < n; i += 64) S.cd[i] = M->comp[i];
    __syncthreads();

    if (lane == 0) {
      Pred P;
      P.S = &S;
      P.cd = n <= ZH_MAX_LDS_COMP ? S.cd : M->comp;
      P.slot = slot;
      P.n = n;
      P.c8 = 1; P.hmap4 = 1;
      P.z.a = P.z.b = P.z.c = P.z.d = P.z.f = 0;
      P.z.prog = L.code + M->code_off + ZH_CODE_PAD;
      P.z.len = M->hcomp_len;
      P.z.m = slot + M->m_off; P.z.mmask = (uint32_t)((1ull << M->hm) - 1);
      P.z.h = (uint32_t *)(slot + M->h_off); P.z.hmask = (uint32_t)((1ull << M->hh) - 1);
      P.z.r = S.r;
      for (uint32_t i = 0; i < n; ++i) {               // scalar parts of Predictor.init
        const ZhComp &cp = P.cd[i];
        switch (cp.type) {
          case ZH_CONS: S.p[i] = ((int)cp.arg[0] - 128) * 4; break;
          case ZH_CM: S.cs[i].limit = (uint32_t)cp.arg[1] * 4; break;
          case ZH_ICM: S.cs[i].limit = 1023; break;
          case ZH_MATCH: (slot + cp.ht_off)[0] = 1; break;
          case ZH_MIX2: case ZH_MIX: S.cs[i].c = cp.cm_mask + 1; break;   // #contexts (see host)
          case ZH_SSE: S.cs[i].limit = (uint32_t)cp.arg[3] * 4; break;
          default: break;
        }
      }
      Coder d;
      if (n) { d.low = 1; d.high = 0xFFFFFFFFu; d.curr = 0; }
      else d.low = d.high = d.curr = 0;
      PostProc pp;
      pp.state = 0; pp.hsize = 0; pp.plen = 0; pp.native = 0;
      pp.z.a = pp.z.b = pp.z.c = pp.z.d = pp.z.f = 0;
      pp.z.prog = nullptr; pp.z.len = 0;
      pp.z.m = slot + M->pm_off; pp.z.mmask = (uint32_t)((1ull << M->pm) - 1);
      pp.z.h = (uint32_t *)(slot + M->ph_off); pp.z.hmask = (uint32_t)((1ull << M->ph) - 1);
      pp.z.r = S.pr;

      Sink &out = S.sink;
      out.out = L.out + bd.out_off; out.cap = bd.out_cap; out.len = 0;

      int failed = 0;
      for (uint32_t s = 0; s < bd.n_seg; ++s) {
        const uint32_t si = bd.first_seg + s;
        ZhSegResult res;
        res.out_off = bd.out_off + out.len;
        if (failed) {
          res.status = ZH_E_SKIPPED; res.pp_state = (uint32_t)pp.state; res.out_len = 0; res.in_used = 0;
          L.results[si] = res;
          continue;
        }
        const ZhSegDesc sd = L.segs[si];
        Src in;
        // Decoder.get() reads the caller's Reader, which does not stop at the segment end
        // (Decoder.cs:112-122): only the end of the stream is EOF.
        in.p = L.in + sd.in_off; in.end = L.in + L.in_total;
        const uint64_t start = out.len;
        int status = 0;
        for (;;) {                                     // Decompresser.decompress(-1), Decompresser.cs:121-153
          int c = decode_byte(P, d, in, L.budget);
          if (c < -1) { status = c + 100; break; }
          int rc = pp_write(pp, c, out, M, slot, L.budget, S.pimm);
          if (rc) { status = rc; break; }
          if (c == -1) break;
          if ((L.flags & ZH_LAUNCH_PP_ONLY) && (pp.state == 1 || pp.state == 5)) { status = ZH_E_STOPPED; break; }   // pcomp() read-back
        }
        if (!status && out.len > out.cap) status = ZH_E_OUTPUT_FULL;
        if (status && status != ZH_E_OUTPUT_FULL) failed = 1;
        res.status = status; res.pp_state = (uint32_t)pp.state | (uint32_t)pp.hsize << 8; res.out_len = out.len - start;
        res.in_used = (uint64_t)(in.p - (L.in + sd.in_off));
        L.results[si] = res;
      }
    }
    __syncthreads();
  }
}

extern "C" hipError_t zh_launch_generic(const ZhLaunch *L, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(zh_decode_generic, dim3(grid), dim3(64), 0, stream, *L);
  return hipGetLastError();
}
