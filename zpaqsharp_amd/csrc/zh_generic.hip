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
// any HCOMP/PCOMP); zh_lanes.hip holds the lane-parallel kernels used for the
// models the host recognises.  Nothing here runs on the CPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zh_model.h"

namespace {

struct CompSt { uint32_t limit, cxt, a, b, c; };   // Component.cs:20-22

struct __align__(16) GenLds {
  ZhTables t;
  int32_t p[256];
  uint32_t h[256];
  uint32_t r[256];     // HCOMP R
  uint32_t pr[256];    // PCOMP R
  CompSt cs[256];
  ZhComp cd[ZH_MAX_LDS_COMP];
};

struct Sink {          // Writer for one block (ZPAQL.outc/flush, ZPAQL.cs:194-207)
  uint8_t *out;
  uint64_t cap, len;
};
__device__ __forceinline__ void sink_put(Sink &s, uint32_t c) {
  if (s.len < s.cap) s.out[s.len] = (uint8_t)c;
  ++s.len;
}

struct Src {           // Reader over one segment's coded bytes (Decoder.get, Decoder.cs:112-122)
  const uint8_t *p, *end;
};
__device__ __forceinline__ int src_get(Src &s) { return s.p < s.end ? (int)*s.p++ : -1; }

struct Vm {            // ZPAQL machine state (ZPAQL.cs:209-223)
  uint32_t a, b, c, d, f;
  const uint8_t *prog; // first program byte; ZH_CODE_PAD zero bytes on both sides
  uint32_t len;        // hend - hbegin
  uint8_t *m;  uint32_t mmask;
  uint32_t *h; uint32_t hmask;
  uint32_t *r;
};

// ZPAQL.cs:1028-1251 execute() + :1253-1265 run0(), decoded by opcode field
// (ISA: ZPAQL.cs:238-321).  Returns 0, ZH_E_ZPAQL or ZH_E_BUDGET.
__device__ int vm_run(Vm &z, uint32_t input, Sink *out, uint64_t budget) {
  const uint8_t *hd = z.prog;
  int pc = 0;
  uint32_t a = input, b = z.b, c = z.c, d = z.d, f = z.f;
  int rc = 0;
  for (;;) {
    if (budget-- == 0) { rc = ZH_E_BUDGET; break; }
    uint32_t op = hd[pc++];
    if (op < 64) {
      uint32_t ddd = op >> 3, x = op & 7;
      if (x == 7) {
        uint32_t n = hd[pc++];
        int off = (int)((n + 128) & 255) - 128;
        switch (ddd) {
          case 0: a = z.r[n]; break;
          case 1: b = z.r[n]; break;
          case 2: c = z.r[n]; break;
          case 3: d = z.r[n]; break;
          case 4: if (f) pc += off; break;             // JT
          case 5: if (!f) pc += off; break;            // JF
          case 6: z.r[n] = a; break;                   // R=A
          default: pc += off; break;                   // JMP
        }
        continue;
      }
      if (ddd == 7) {
        if (x == 0) break;                                                   // HALT
        if (x == 1) { if (out) sink_put(*out, a & 255); continue; }         // OUT
        if (x == 3) { a = (a + z.m[b & z.mmask] + 512u) * 773u; continue; }  // HASH
        if (x == 4) { uint32_t *q = &z.h[d & z.hmask]; *q = (*q + a + 512u) * 773u; continue; }  // HASHD
        rc = ZH_E_ZPAQL; break;
      }
      if (x > 4 || op == 0) { rc = ZH_E_ZPAQL; break; }
      uint32_t v;
      switch (ddd) {
        case 0: v = a; break;
        case 1: v = b; break;
        case 2: v = c; break;
        case 3: v = d; break;
        case 4: v = z.m[b & z.mmask]; break;
        case 5: v = z.m[c & z.mmask]; break;
        default: v = z.h[d & z.hmask]; break;
      }
      uint32_t olda = a;
      switch (x) {
        case 0:                                        // <>a ; *b/*c swap the low byte only (ZPAQL.cs:1298-1303)
          if (ddd == 4 || ddd == 5) { a = (a & ~255u) | (v & 255u); v = olda & 255u; }
          else { a = v; v = olda; }
          break;
        case 1: ++v; break;
        case 2: --v; break;
        case 3: v = ~v; break;
        default: v = 0; break;
      }
      switch (ddd) {
        case 0: a = v; break;
        case 1: b = v; break;
        case 2: c = v; break;
        case 3: d = v; break;
        case 4: z.m[b & z.mmask] = (uint8_t)v; break;
        case 5: z.m[c & z.mmask] = (uint8_t)v; break;
        default: z.h[d & z.hmask] = v; break;
      }
      continue;
    }
    if (op == 255) {                                   // LJ
      uint32_t t = hd[pc] + 256u * hd[pc + 1];
      if (t >= z.len) { rc = ZH_E_ZPAQL; break; }
      pc = (int)t;
      continue;
    }
    uint32_t s;
    switch (op & 7) {
      case 0: s = a; break;
      case 1: s = b; break;
      case 2: s = c; break;
      case 3: s = d; break;
      case 4: s = z.m[b & z.mmask]; break;
      case 5: s = z.m[c & z.mmask]; break;
      case 6: s = z.h[d & z.hmask]; break;
      default: s = hd[pc++]; break;
    }
    if (op < 128) {
      switch ((op >> 3) & 7) {
        case 0: a = s; break;
        case 1: b = s; break;
        case 2: c = s; break;
        case 3: d = s; break;
        case 4: z.m[b & z.mmask] = (uint8_t)s; break;
        case 5: z.m[c & z.mmask] = (uint8_t)s; break;
        case 6: z.h[d & z.hmask] = s; break;
        default: rc = ZH_E_ZPAQL; break;
      }
      if (rc) break;
      continue;
    }
    switch ((op >> 3) & 15) {
      case 0: a += s; break;
      case 1: a -= s; break;
      case 2: a *= s; break;
      case 3: a = s ? a / s : 0; break;
      case 4: a = s ? a % s : 0; break;
      case 5: a &= s; break;
      case 6: a &= ~s; break;
      case 7: a |= s; break;
      case 8: a ^= s; break;
      case 9: a <<= (s & 31); break;
      case 10: a >>= (s & 31); break;
      case 11: f = a == s; break;
      case 12: f = a < s; break;
      case 13: f = a > s; break;
      default: rc = ZH_E_ZPAQL; break;
    }
    if (rc) break;
  }
  z.a = a; z.b = b; z.c = c; z.d = d; z.f = f;
  return rc;
}

// ---- model-independent arithmetic (Predictor.cs:496-543, intended bounds) ----
__device__ __forceinline__ int clamp2k(int x) { return x < -2048 ? -2048 : x > 2047 ? 2047 : x; }
__device__ __forceinline__ int clamp512k(int x) {
  return x < -(1 << 19) ? -(1 << 19) : x >= (1 << 19) ? (1 << 19) - 1 : x;
}

struct Pred {
  GenLds *S;
  const ZhComp *cd;    // component descriptors (LDS copy or global)
  uint8_t *slot;       // arena slot base
  uint32_t n;
  int c8, hmap4;
  Vm z;                // HCOMP machine
};

__device__ __forceinline__ int squash(const GenLds *S, int x) { return S->t.squash[x + 2048]; }
__device__ __forceinline__ int stretch(const GenLds *S, int x) { return S->t.stretch[x]; }

// Predictor.cs:550-567
__device__ uint32_t find_row(uint8_t *ht, uint32_t ht_mask, int sizebits, uint32_t cxt) {
  uint32_t chk = (cxt >> sizebits) & 255;
  uint32_t h0 = (cxt * 16u) & (ht_mask - 15u);
  if (ht[h0] == chk) return h0;
  uint32_t h1 = h0 ^ 16;
  if (ht[h1] == chk) return h1;
  uint32_t h2 = h0 ^ 32;
  if (ht[h2] == chk) return h2;
  uint32_t v;
  uint8_t p0 = ht[h0 + 1], p1 = ht[h1 + 1], p2 = ht[h2 + 1];
  if (p0 <= p1 && p0 <= p2) v = h0;
  else if (p1 < p2) v = h1;
  else v = h2;
  uint4 zero = make_uint4(chk, 0, 0, 0);
  *reinterpret_cast<uint4 *>(ht + v) = zero;          // rows are 16-byte aligned
  return v;
}

// Predictor.cs:245-350
__device__ int predict(Pred &P) {
  GenLds *S = P.S;
  int *p = S->p; const uint32_t *h = S->h;
  const int c8 = P.c8, hmap4 = P.hmap4;
  for (uint32_t i = 0; i < P.n; ++i) {
    const ZhComp &cp = P.cd[i];
    CompSt &cr = S->cs[i];
    switch (cp.type) {
      case ZH_CONS: break;
      case ZH_CM: {
        uint32_t *cm = (uint32_t *)(P.slot + cp.cm_off);
        cr.cxt = h[i] ^ (uint32_t)hmap4;
        p[i] = stretch(S, cm[cr.cxt & cp.cm_mask] >> 17);
        break;
      }
      case ZH_ICM: {
        uint8_t *ht = P.slot + cp.ht_off;
        uint32_t *cm = (uint32_t *)(P.slot + cp.cm_off);
        if (c8 == 1 || (c8 & 0xf0) == 16) cr.c = find_row(ht, cp.ht_mask, cp.arg[0] + 2, h[i] + 16u * (uint32_t)c8);
        cr.cxt = ht[cr.c + (uint32_t)(hmap4 & 15)];
        p[i] = stretch(S, cm[cr.cxt & cp.cm_mask] >> 8);
        break;
      }
      case ZH_MATCH: {
        if (cr.a == 0) p[i] = 0;
        else {
          uint8_t *ht = P.slot + cp.ht_off;
          cr.c = (ht[(cr.limit - cr.b) & cp.ht_mask] >> (7 - cr.cxt)) & 1;
          p[i] = stretch(S, (S->t.dt2k[cr.a] * (1 - 2 * (int)cr.c)) & 32767);
        }
        break;
      }
      case ZH_AVG:
        p[i] = (p[cp.arg[0]] * cp.arg[2] + p[cp.arg[1]] * (256 - cp.arg[2])) >> 8;
        break;
      case ZH_MIX2: {
        uint16_t *a16 = (uint16_t *)(P.slot + cp.cm_off);
        cr.cxt = (h[i] + (uint32_t)(c8 & cp.arg[4])) & (cr.c - 1);
        int w = a16[cr.cxt];
        p[i] = (w * p[cp.arg[1]] + (65536 - w) * p[cp.arg[2]]) >> 16;
        break;
      }
      case ZH_MIX: {
        int m = cp.arg[2];
        int *cm = (int *)(P.slot + cp.cm_off);
        cr.cxt = ((h[i] + (uint32_t)(c8 & cp.arg[4])) & (cr.c - 1)) * (uint32_t)m;
        const int *wt = &cm[cr.cxt];
        int s = 0;
        for (int j = 0; j < m; ++j) s += (wt[j] >> 8) * p[cp.arg[1] + j];
        p[i] = clamp2k(s >> 8);
        break;
      }
      case ZH_ISSE: {
        uint8_t *ht = P.slot + cp.ht_off;
        int *cm = (int *)(P.slot + cp.cm_off);
        if (c8 == 1 || (c8 & 0xf0) == 16) cr.c = find_row(ht, cp.ht_mask, cp.arg[0] + 2, h[i] + 16u * (uint32_t)c8);
        cr.cxt = ht[cr.c + (uint32_t)(hmap4 & 15)];
        const int *wt = &cm[cr.cxt * 2];
        p[i] = clamp2k((wt[0] * p[cp.arg[1]] + wt[1] * 64) >> 16);
        break;
      }
      case ZH_SSE: {
        uint32_t *cm = (uint32_t *)(P.slot + cp.cm_off);
        cr.cxt = (h[i] + (uint32_t)c8) * 32u;
        int pq = p[cp.arg[1]] + 992;
        pq = pq < 0 ? 0 : pq > 1983 ? 1983 : pq;
        int wt = pq & 63;
        pq >>= 6;
        cr.cxt += (uint32_t)pq;
        p[i] = stretch(S, ((cm[cr.cxt & cp.cm_mask] >> 10) * (uint32_t)(64 - wt) +
                           (cm[(cr.cxt + 1) & cp.cm_mask] >> 10) * (uint32_t)wt) >> 13);
        cr.cxt += (uint32_t)(wt >> 5);
        break;
      }
      default: break;
    }
  }
  return squash(S, p[P.n - 1]);
}

// Predictor.cs:486-493 in the intended form kept at Predictor.cs:1031-1036
__device__ __forceinline__ void train(const GenLds *S, uint32_t *pn, uint32_t limit, int y) {
  uint32_t v = *pn;
  uint32_t count = v & 0x3ff;
  int error = y * 32767 - (int)(v >> 17);
  *pn = v + (((uint32_t)error * (uint32_t)S->t.dt[count]) & 0xFFFFFC00u) + (count < limit);
}

// Predictor.cs:353-475.  Returns 0 or a ZPAQL status from the HCOMP run.
__device__ int update(Pred &P, int y, uint64_t budget) {
  GenLds *S = P.S;
  int *p = S->p; uint32_t *h = S->h;
  const int hmap4 = P.hmap4;
  for (uint32_t i = 0; i < P.n; ++i) {
    const ZhComp &cp = P.cd[i];
    CompSt &cr = S->cs[i];
    switch (cp.type) {
      case ZH_CM:
      case ZH_SSE: {
        uint32_t *cm = (uint32_t *)(P.slot + cp.cm_off);
        train(S, &cm[cr.cxt & cp.cm_mask], cr.limit, y);
        break;
      }
      case ZH_ICM: {
        uint8_t *bh = P.slot + cp.ht_off + cr.c + (uint32_t)(hmap4 & 15);
        uint32_t *cm = (uint32_t *)(P.slot + cp.cm_off);
        *bh = S->t.ns[*bh * 4 + y];
        uint32_t *pn = &cm[cr.cxt & cp.cm_mask];
        *pn += (uint32_t)((int)(y * 32767 - (int)(*pn >> 8)) >> 2);
        break;
      }
      case ZH_MATCH: {
        uint8_t *ht = P.slot + cp.ht_off;
        uint32_t *cm = (uint32_t *)(P.slot + cp.cm_off);
        if ((int)cr.c != y) cr.a = 0;
        uint8_t *bp = &ht[cr.limit & cp.ht_mask];
        *bp = (uint8_t)(*bp + *bp + y);
        if (++cr.cxt == 8) {
          cr.cxt = 0;
          cr.limit = (cr.limit + 1) & cp.ht_mask;
          if (cr.a == 0) {
            cr.b = cr.limit - cm[h[i] & cp.cm_mask];
            if (cr.b & cp.ht_mask)
              while (cr.a < 255 && ht[(cr.limit - cr.a - 1) & cp.ht_mask] == ht[(cr.limit - cr.a - cr.b - 1) & cp.ht_mask])
                ++cr.a;
          } else cr.a += cr.a < 255;
          cm[h[i] & cp.cm_mask] = cr.limit;
        }
        break;
      }
      case ZH_MIX2: {
        uint16_t *a16 = (uint16_t *)(P.slot + cp.cm_off);
        int err = (y * 32767 - squash(S, p[i])) * cp.arg[3] >> 5;
        int w = a16[cr.cxt];
        w += (err * (p[cp.arg[1]] - p[cp.arg[2]]) + (1 << 12)) >> 13;
        w = w < 0 ? 0 : w > 65535 ? 65535 : w;
        a16[cr.cxt] = (uint16_t)w;
        break;
      }
      case ZH_MIX: {
        int m = cp.arg[2];
        int *wt = (int *)(P.slot + cp.cm_off) + cr.cxt;
        int err = (y * 32767 - squash(S, p[i])) * cp.arg[3] >> 4;
        for (int j = 0; j < m; ++j)
          wt[j] = clamp512k(wt[j] + ((err * p[cp.arg[1] + j] + (1 << 12)) >> 13));
        break;
      }
      case ZH_ISSE: {
        int *wt = (int *)(P.slot + cp.cm_off) + cr.cxt * 2;
        int err = y * 32767 - squash(S, p[i]);
        wt[0] = clamp512k(wt[0] + ((err * p[cp.arg[1]] + (1 << 12)) >> 13));
        wt[1] = clamp512k(wt[1] + ((err + 16) >> 5));
        P.slot[cp.ht_off + cr.c + (uint32_t)(hmap4 & 15)] = S->t.ns[cr.cxt * 4 + y];
        break;
      }
      default: break;
    }
  }
  // Predictor.cs:463-474
  P.c8 += P.c8 + y;
  if (P.c8 >= 256) {
    int rc = vm_run(P.z, (uint32_t)(P.c8 - 256), nullptr, budget);
    if (rc) return rc;
    P.hmap4 = 1;
    P.c8 = 1;
    for (uint32_t i = 0; i < P.n; ++i) h[i] = P.z.h[i & P.z.hmask];
  } else if (P.c8 >= 16 && P.c8 < 32)
    P.hmap4 = (P.hmap4 & 0xf) << 5 | y << 4 | 1;
  else
    P.hmap4 = (P.hmap4 & 0x1f0) | (((P.hmap4 & 0xf) * 2 + y) & 0xf);
  return 0;
}

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
  Vm z;
};

// PostProcessor.cs:37-86.  c is 0..255 or -1.  Returns 0 or a status.
__device__ int pp_write(PostProc &pp, int c, Sink &out, const ZhModel *M, uint8_t *slot, uint64_t budget) {
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
        pp.state = 5;
      }
      break;
    }
    default:
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
      pp.state = 0; pp.hsize = 0; pp.plen = 0;
      pp.z.a = pp.z.b = pp.z.c = pp.z.d = pp.z.f = 0;
      pp.z.prog = nullptr; pp.z.len = 0;
      pp.z.m = slot + M->pm_off; pp.z.mmask = (uint32_t)((1ull << M->pm) - 1);
      pp.z.h = (uint32_t *)(slot + M->ph_off); pp.z.hmask = (uint32_t)((1ull << M->ph) - 1);
      pp.z.r = S.pr;

      Sink out;
      out.out = L.out + bd.out_off; out.cap = bd.out_cap; out.len = 0;

      int failed = 0;
      for (uint32_t s = 0; s < bd.n_seg; ++s) {
        const uint32_t si = bd.first_seg + s;
        ZhSegResult res;
        res.out_off = bd.out_off + out.len;
        if (failed) {
          res.status = ZH_E_SKIPPED; res.pp_state = (uint32_t)pp.state; res.out_len = 0;
          L.results[si] = res;
          continue;
        }
        const ZhSegDesc sd = L.segs[si];
        Src in;
        in.p = L.in + sd.in_off; in.end = in.p + sd.in_len;
        const uint64_t start = out.len;
        int status = 0;
        for (;;) {                                     // Decompresser.decompress(-1), Decompresser.cs:121-153
          int c = decode_byte(P, d, in, L.budget);
          if (c < -1) { status = c + 100; break; }
          int rc = pp_write(pp, c, out, M, slot, L.budget);
          if (rc) { status = rc; break; }
          if (c == -1) break;
        }
        if (!status && out.len > out.cap) status = ZH_E_OUTPUT_FULL;
        if (status && status != ZH_E_OUTPUT_FULL) failed = 1;
        res.status = status; res.pp_state = (uint32_t)pp.state; res.out_len = out.len - start;
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
