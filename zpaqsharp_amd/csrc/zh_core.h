// zh_core.h — the scalar (one-lane) form of the ZPAQ model: ZPAQL interpreter,
// Predictor.predict0/update0 and their helpers, written once and compiled
//   * for gfx950 as the body of the generic decode kernel (zh_generic.hip), and
//   * for the host as the predictor of the CPU stream *encoder* (../gen/), which
//     needs exactly the same model to write test/benchmark streams.
// The decompression path never runs this code on the host: libzpaqhip.so
// instantiates it in device code only.
#pragma once
#include <stdint.h>

#include "zh_model.h"

#if defined(__HIPCC__)
#define ZH_HD __host__ __device__
#else
#define ZH_HD
#endif

namespace zhcore {

struct alignas(16) ZhRow16 { uint32_t w[4]; };

struct CompSt { uint32_t limit, cxt, a, b, c; };   // Component.cs:20-22



struct Sink {          // Writer for one block (ZPAQL.outc/flush, ZPAQL.cs:194-207)
  uint8_t *out;
  uint64_t cap, len;
};
ZH_HD inline void sink_put(Sink &s, uint32_t c) {
  if (s.len < s.cap) s.out[s.len] = (uint8_t)c;
  ++s.len;
}

struct alignas(16) GenLds {
  ZhTables t;
  int32_t p[256];
  uint32_t h[256];
  uint32_t r[256];     // HCOMP R
  uint32_t pr[256];    // PCOMP R
  CompSt cs[256];
  ZhComp cd[ZH_MAX_LDS_COMP];
  alignas(16) uint32_t pimm[64];   // operands of a structurally matched PCOMP (zh_zpaql_pcomp.h)
  Sink sink;                       // the block's Writer (in LDS: the translated PCOMPs are real calls and take its address)
};

struct Src {           // Reader over one segment's coded bytes (Decoder.get, Decoder.cs:112-122)
  const uint8_t *p, *end;
};
ZH_HD inline int src_get(Src &s) { return s.p < s.end ? (int)*s.p++ : -1; }

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
ZH_HD inline int vm_run(Vm &z, uint32_t input, Sink *out, uint64_t budget) {
  // The machine description is copied to locals: `z` usually lives in LDS / memory and
  // must not be re-read around every M/H access.
  const uint8_t *hd = z.prog;
  uint8_t *const zm = z.m;
  uint32_t *const zh = z.h;
  uint32_t *const zr = z.r;
  const uint32_t mmask = z.mmask, hmask = z.hmask, zlen = z.len;
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
          case 0: a = zr[n]; break;
          case 1: b = zr[n]; break;
          case 2: c = zr[n]; break;
          case 3: d = zr[n]; break;
          case 4: if (f) pc += off; break;             // JT
          case 5: if (!f) pc += off; break;            // JF
          case 6: zr[n] = a; break;                   // R=A
          default: pc += off; break;                   // JMP
        }
        continue;
      }
      if (ddd == 7) {
        if (x == 0) break;                                                   // HALT
        if (x == 1) { if (out) sink_put(*out, a & 255); continue; }         // OUT
        if (x == 3) { a = (a + zm[b & mmask] + 512u) * 773u; continue; }  // HASH
        if (x == 4) { uint32_t *q = &zh[d & hmask]; *q = (*q + a + 512u) * 773u; continue; }  // HASHD
        rc = ZH_E_ZPAQL; break;
      }
      if (x > 4 || op == 0) { rc = ZH_E_ZPAQL; break; }
      uint32_t v;
      switch (ddd) {
        case 0: v = a; break;
        case 1: v = b; break;
        case 2: v = c; break;
        case 3: v = d; break;
        case 4: v = zm[b & mmask]; break;
        case 5: v = zm[c & mmask]; break;
        default: v = zh[d & hmask]; break;
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
        case 4: zm[b & mmask] = (uint8_t)v; break;
        case 5: zm[c & mmask] = (uint8_t)v; break;
        default: zh[d & hmask] = v; break;
      }
      continue;
    }
    if (op == 255) {                                   // LJ
      uint32_t t = hd[pc] + 256u * hd[pc + 1];
      if (t >= zlen) { rc = ZH_E_ZPAQL; break; }
      pc = (int)t;
      continue;
    }
    uint32_t s;
    switch (op & 7) {
      case 0: s = a; break;
      case 1: s = b; break;
      case 2: s = c; break;
      case 3: s = d; break;
      case 4: s = zm[b & mmask]; break;
      case 5: s = zm[c & mmask]; break;
      case 6: s = zh[d & hmask]; break;
      default: s = hd[pc++]; break;
    }
    if (op < 128) {
      switch ((op >> 3) & 7) {
        case 0: a = s; break;
        case 1: b = s; break;
        case 2: c = s; break;
        case 3: d = s; break;
        case 4: zm[b & mmask] = (uint8_t)s; break;
        case 5: zm[c & mmask] = (uint8_t)s; break;
        case 6: zh[d & hmask] = s; break;
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
ZH_HD inline int clamp2k(int x) { return x < -2048 ? -2048 : x > 2047 ? 2047 : x; }
ZH_HD inline int clamp512k(int x) {
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

ZH_HD inline int squash(const GenLds *S, int x) { return S->t.squash[x + 2048]; }
ZH_HD inline int stretch(const GenLds *S, int x) { return S->t.stretch[x]; }

// Predictor.cs:550-567
ZH_HD inline uint32_t find_row(uint8_t *ht, uint32_t ht_mask, int sizebits, uint32_t cxt) {
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
  ZhRow16 row = {{chk, 0, 0, 0}};
  *reinterpret_cast<ZhRow16 *>(ht + v) = row;         // rows are 16-byte aligned
  return v;
}

// Predictor.cs:245-350
ZH_HD inline int predict(Pred &P) {
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
ZH_HD inline void train(const GenLds *S, uint32_t *pn, uint32_t limit, int y) {
  uint32_t v = *pn;
  uint32_t count = v & 0x3ff;
  int error = y * 32767 - (int)(v >> 17);
  *pn = v + (((uint32_t)error * (uint32_t)S->t.dt[count]) & 0xFFFFFC00u) + (count < limit);
}

// Predictor.cs:353-475.  Returns 0 or a ZPAQL status from the HCOMP run.
ZH_HD inline int update(Pred &P, int y, uint64_t budget) {
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


}  // namespace zhcore
