/* oracle/zpaq_oracle.c — CPU oracle, see zpaq_oracle.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C restatement of the reference's interpreter semantics.  Every
 * function cites the reference file:line (relative to
 * /root/reference/ZPAQSharp/) it follows.  Defects of the C# transliteration
 *  are NOT reproduced; the intended libzpaq arithmetic, which
 * the reference keeps verbatim in comments, is used instead.
 */
#include "zpaq_oracle.h"

#include <math.h>
#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t U8;
typedef uint16_t U16;
typedef uint32_t U32;
typedef uint64_t U64;

/* ======================================================================
 * Model-independent tables                       Predictor.cs:48-79
 * ====================================================================== */
static U16 squasht[4096];
static int16_t stretcht[32768];
static int dt[1024];
static int dt2k[256];
static U8 ns[1024];
static int tables_ready = 0;
static U32 pin_stsum, pin_sqsum, pin_sns;

static U32 crc32_update(U32 crc, const U8 *p, size_t n) {
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) {
    crc ^= p[i];
    for (int k = 0; k < 8; ++k) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1)));
  }
  return ~crc;
}

/* Bit-history state table.  The reference embeds the finished table
 * (StateTable.cs:21-149); it is regenerated here from the ZPAQ bit-history
 * rules (bounded (n0,n1) counts, discount of the opposite count, one or two
 * states per count pair) and pinned by CRC-32 0x77a1e24c of the 1024 bytes. */
static int st_num_states(int n0, int n1) {
  static const int bound[6] = {20, 48, 15, 8, 6, 5};
  if (n0 < n1) return st_num_states(n1, n0);
  if (n0 < 0 || n1 < 0 || n1 >= 6 || n0 > bound[n1]) return 0;
  return 1 + (n1 > 0 && n0 + n1 <= 17);
}
static int st_discount(int n) {
  return (n >= 1) + (n >= 2) + (n >= 3) + (n >= 4) + (n >= 5) + (n >= 7) + (n >= 8);
}
static void st_next(int *n0, int *n1, int y) {
  if (*n0 < *n1) { st_next(n1, n0, 1 - y); return; }
  if (y) { ++*n1; *n0 = st_discount(*n0); }
  else   { ++*n0; *n1 = st_discount(*n1); }
  while (!st_num_states(*n0, *n1)) {
    if (*n1 < 2) --*n0;
    else { *n0 = (*n0 * (*n1 - 1) + (*n1 / 2)) / *n1; --*n1; }
  }
}
static void build_state_table(void) {
  enum { N = 50 };
  static U8 t[N][N][2];
  int state = 0;
  memset(t, 0, sizeof t);
  for (int i = 0; i < N; ++i)
    for (int n1 = 0; n1 <= i; ++n1) {
      int n0 = i - n1, n = st_num_states(n0, n1);
      if (n) { t[n0][n1][0] = (U8)state; t[n0][n1][1] = (U8)(state + n - 1); state += n; }
    }
  memset(ns, 0, sizeof ns);
  for (int n0 = 0; n0 < N; ++n0)
    for (int n1 = 0; n1 < N; ++n1)
      for (int y = 0; y < st_num_states(n0, n1); ++y) {
        int s = t[n0][n1][y], a = n0, b = n1;
        st_next(&a, &b, 0); ns[s * 4 + 0] = t[a][b][0];
        a = n0; b = n1;
        st_next(&a, &b, 1); ns[s * 4 + 1] = t[a][b][1];
        ns[s * 4 + 2] = (U8)n0; ns[s * 4 + 3] = (U8)n1;
      }
}

/* StateTable.cs:151-162 */
static inline int st_nex(int state, int y) { return ns[state * 4 + y]; }
static inline U32 st_cminit(int state) {
  return (U32)(((ns[state * 4 + 3] * 2 + 1) << 22) / (ns[state * 4 + 2] + ns[state * 4 + 3] + 1));
}

static void init_tables(void) {
  if (tables_ready) return;
  /* Predictor.cs:1358,1394 generator comments: dt2k[i]=2^11/i, dt[i]=2^17/(2i+3)*2 */
  dt2k[0] = 0;
  for (int i = 1; i < 256; ++i) dt2k[i] = 2048 / i;
  for (int i = 0; i < 1024; ++i) dt[i] = (1 << 17) / (i * 2 + 3) * 2;
  /* Predictor.cs:54-58: squash, middle 1344 entries by formula, tails 0/32767 */
  for (int i = 0; i < 4096; ++i) {
    if (i < 1376) squasht[i] = 0;
    else if (i >= 2720) squasht[i] = 32767;
    else squasht[i] = (U16)(int)(32768.0 / (1 + exp((i - 2048) * (-1.0 / 64))));
  }
  /* Predictor.cs:60-67: stretch, odd-symmetric */
  for (int i = 16384; i < 32768; ++i)
    stretcht[i] = (int16_t)((int)(log((i + 0.5) / (32767.5 - i)) * 64 + 0.5 + 100000) - 100000);
  for (int i = 0; i < 16384; ++i) stretcht[i] = (int16_t)-stretcht[32767 - i];
  build_state_table();
  /* Predictor.cs:71-77 self-check constants */
  U32 sq = 0, st = 0;
  for (int i = 32767; i >= 0; --i) st = st * 3 + (U32)(int)stretcht[i];
  for (int i = 4095; i >= 0; --i) sq = sq * 3 + squasht[i];
  pin_stsum = st; pin_sqsum = sq; pin_sns = crc32_update(0, ns, 1024);
  if (st != 3887533746u || sq != 2278286169u || pin_sns != 0x77a1e24cu) {
    fprintf(stderr, "zpaq_oracle: table pins failed (%u %u %08x)\n", st, sq, pin_sns);
    abort();
  }
  tables_ready = 1;
}

int zo_table_pins(U32 *stsum, U32 *sqsum, U32 *sns_crc32) {
  init_tables();
  *stsum = pin_stsum; *sqsum = pin_sqsum; *sns_crc32 = pin_sns;
  return 0;
}
void zo_tables(U16 *sq, int16_t *st, int32_t *d, int32_t *d2, U8 *n) {
  init_tables();
  memcpy(sq, squasht, sizeof squasht); memcpy(st, stretcht, sizeof stretcht);
  memcpy(d, dt, sizeof dt); memcpy(d2, dt2k, sizeof dt2k); memcpy(n, ns, sizeof ns);
}

/* Predictor.cs:496-543 with the intended bounds (SURVEY §8a defect table) */
static inline int squash(int x) { return squasht[x + 2048]; }
static inline int stretch(int x) { return stretcht[x]; }
static inline int clamp2k(int x) { return x < -2048 ? -2048 : x > 2047 ? 2047 : x; }
static inline int clamp512k(int x) {
  return x < -(1 << 19) ? -(1 << 19) : x >= (1 << 19) ? (1 << 19) - 1 : x;
}

/* ======================================================================
 * SHA-1 (FIPS 180-4).  The reference defers to the .NET BCL (ZPAQL.cs:187).
 * ====================================================================== */
typedef struct { U32 h[5]; U64 len; U8 buf[64]; int fill; } sha1_t;
static void sha1_init(sha1_t *s) {
  s->h[0] = 0x67452301; s->h[1] = 0xEFCDAB89; s->h[2] = 0x98BADCFE;
  s->h[3] = 0x10325476; s->h[4] = 0xC3D2E1F0; s->len = 0; s->fill = 0;
}
static inline U32 rol(U32 x, int n) { return x << n | x >> (32 - n); }
static void sha1_block(sha1_t *s, const U8 *p) {
  U32 w[80], a = s->h[0], b = s->h[1], c = s->h[2], d = s->h[3], e = s->h[4];
  for (int i = 0; i < 16; ++i)
    w[i] = (U32)p[i * 4] << 24 | (U32)p[i * 4 + 1] << 16 | (U32)p[i * 4 + 2] << 8 | p[i * 4 + 3];
  for (int i = 16; i < 80; ++i) w[i] = rol(w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
  for (int i = 0; i < 80; ++i) {
    U32 f, k;
    if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999; }
    else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1; }
    else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDC; }
    else { f = b ^ c ^ d; k = 0xCA62C1D6; }
    U32 t = rol(a, 5) + f + e + k + w[i];
    e = d; d = c; c = rol(b, 30); b = a; a = t;
  }
  s->h[0] += a; s->h[1] += b; s->h[2] += c; s->h[3] += d; s->h[4] += e;
}
static void sha1_write(sha1_t *s, const U8 *p, size_t n) {
  s->len += n;
  while (n) {
    size_t k = 64 - (size_t)s->fill; if (k > n) k = n;
    memcpy(s->buf + s->fill, p, k); s->fill += (int)k; p += k; n -= k;
    if (s->fill == 64) { sha1_block(s, s->buf); s->fill = 0; }
  }
}
static void sha1_result(sha1_t *s, U8 out[20]) {
  U64 bits = s->len * 8; U8 pad = 0x80;
  sha1_write(s, &pad, 1); pad = 0;
  while (s->fill != 56) sha1_write(s, &pad, 1);
  U8 l[8]; for (int i = 0; i < 8; ++i) l[i] = (U8)(bits >> (56 - 8 * i));
  sha1_write(s, l, 8);
  for (int i = 0; i < 5; ++i) { out[i*4] = (U8)(s->h[i] >> 24); out[i*4+1] = (U8)(s->h[i] >> 16);
    out[i*4+2] = (U8)(s->h[i] >> 8); out[i*4+3] = (U8)s->h[i]; }
}
void zo_sha1(const U8 *p, size_t n, U8 out[20]) { sha1_t s; sha1_init(&s); sha1_write(&s, p, n); sha1_result(&s, out); }

/* ======================================================================
 * error() — must not return (LibZPAQ.cs:22-24, LICENSE:41-46)
 * ====================================================================== */
typedef struct { jmp_buf jb; char msg[96]; int armed; } err_t;
static void zerror(err_t *e, const char *msg) {
  snprintf(e->msg, sizeof e->msg, "%s", msg);
  if (!e->armed) { fprintf(stderr, "zpaq_oracle: unarmed error: %s\n", msg); abort(); }
  longjmp(e->jb, 1);
}

/* ======================================================================
 * Byte source / sink                       Reader.cs:9-25, Writer.cs:14-24
 * ====================================================================== */
typedef struct { const U8 *p; size_t n, pos; } rd_t;
static inline int rd_get(rd_t *r) { return r->pos < r->n ? r->p[r->pos++] : -1; }
typedef struct { U8 *p; size_t cap, len; int overflow; } wr_t;
static inline void wr_put(wr_t *w, int c) {
  if (w->len < w->cap) w->p[w->len] = (U8)c; else w->overflow = 1;
  ++w->len;
}

/* ======================================================================
 * ZPAQL virtual machine                                 ZPAQL.cs
 * ====================================================================== */
enum { NONE, CONS, CM, ICM, MATCH, AVG, MIX2, MIX, ISSE, SSE };      /* LibZPAQ.cs:51-63 */
static const int compsize[10] = {0, 2, 3, 2, 3, 4, 6, 6, 3, 5};      /* Component.cs:27-43 */

typedef struct {
  U8 *header; size_t header_len;           /* hsize[2] hh hm ph pm n COMP 0 gap HCOMP 0 */
  int cend, hbegin, hend;
  U8 *m; U32 *h; U32 r[256];
  U64 msize, hsize;                        /* element counts, powers of 2 */
  U32 a, b, c, d; int f; int pc;
  wr_t *output; sha1_t *sha1;
  err_t *err;
} vm_t;

static void vm_clear(vm_t *z) {                                      /* ZPAQL.cs:33-42 */
  free(z->header); free(z->m); free(z->h);
  z->header = 0; z->m = 0; z->h = 0; z->header_len = 0; z->msize = z->hsize = 0;
  z->cend = z->hbegin = z->hend = 0;
  z->a = z->b = z->c = z->d = 0; z->f = 0; z->pc = 0;
}

static void vm_init(vm_t *z, int hbits, int mbits) {                 /* ZPAQL.cs:1010-1026 */
  if (hbits > 32) zerror(z->err, "H too big");
  if (mbits > 32) zerror(z->err, "M too big");
  if (hbits > 28 || mbits > 30) zerror(z->err, "oracle: H/M beyond host memory");
  free(z->h); free(z->m);
  z->hsize = (U64)1 << hbits; z->msize = (U64)1 << mbits;
  z->h = calloc(z->hsize, 4); z->m = calloc(z->msize, 1);
  if (!z->h || !z->m) zerror(z->err, "Out of memory");
  memset(z->r, 0, sizeof z->r);
  z->a = z->b = z->c = z->d = 0; z->pc = 0; z->f = 0;
}
static void vm_inith(vm_t *z) { vm_init(z, z->header[2], z->header[3]); }   /* ZPAQL.cs:44-50 */
static void vm_initp(vm_t *z) { vm_init(z, z->header[4], z->header[5]); }   /* ZPAQL.cs:52-56 */

static double pow2(int x) { double r = 1; for (; x > 0; --x) r += r; return r; }
static double vm_memory(const vm_t *z) {                             /* ZPAQL.cs:58-81 */
  const U8 *hd = z->header;
  double mem = pow2(hd[2] + 2) + pow2(hd[3]) + pow2(hd[4] + 2) + pow2(hd[5]) + (double)z->header_len;
  int cp = 7;
  for (int i = 0; i < hd[6]; ++i) {
    double size = pow2(hd[cp + 1]);
    switch (hd[cp]) {
      case CM: mem += 4 * size; break;
      case ICM: mem += 64 * size + 1024; break;
      case MATCH: mem += 4 * size + pow2(hd[cp + 2]); break;
      case MIX2: mem += 2 * size; break;
      case MIX: mem += 4 * size * hd[cp + 3]; break;
      case ISSE: mem += 64 * size + 2048; break;
      case SSE: mem += 128 * size; break;
    }
    cp += compsize[hd[cp]];
  }
  return mem;
}

/* ZPAQL.cs:112-156 — parse block header from a byte source */
static int vm_read(vm_t *z, rd_t *in) {
  int hsize = rd_get(in);
  int hi = rd_get(in);
  if (hsize < 0 || hi < 0) zerror(z->err, "unexpected end of file");
  hsize += hi * 256;
  free(z->header);
  z->header_len = (size_t)hsize + 300;
  z->header = calloc(z->header_len, 1);
  z->cend = z->hbegin = z->hend = 0;
  z->header[z->cend++] = (U8)(hsize & 255);
  z->header[z->cend++] = (U8)(hsize >> 8);
  while (z->cend < 7) {
    int c = rd_get(in);
    if (c < 0) zerror(z->err, "unexpected end of file");
    z->header[z->cend++] = (U8)c;
  }
  int n = z->header[z->cend - 1];
  for (int i = 0; i < n; ++i) {
    int type = rd_get(in);
    if (type < 0 || type > 255) zerror(z->err, "unexpected end of file");
    z->header[z->cend++] = (U8)type;
    int size = type < 10 ? compsize[type] : 0;
    if (size < 1) zerror(z->err, "Invalid component type");
    if (z->cend + size > hsize) zerror(z->err, "COMP overflows header");
    for (int j = 1; j < size; ++j) {
      int c = rd_get(in);
      if (c < 0) zerror(z->err, "unexpected end of file");
      z->header[z->cend++] = (U8)c;
    }
  }
  int c = rd_get(in);
  if (c < 0) zerror(z->err, "unexpected end of file");
  if ((z->header[z->cend++] = (U8)c) != 0) zerror(z->err, "missing COMP END");
  z->hbegin = z->hend = z->cend + 128;
  if (z->hend > hsize + 129) zerror(z->err, "missing HCOMP");
  while (z->hend < hsize + 129) {
    int op = rd_get(in);
    if (op == -1) zerror(z->err, "unexpected end of file");
    z->header[z->hend++] = (U8)op;
  }
  c = rd_get(in);
  if (c < 0) zerror(z->err, "unexpected end of file");
  if ((z->header[z->hend++] = (U8)c) != 0) zerror(z->err, "missing HCOMP END");
  return z->cend + z->hend - z->hbegin;
}

/* ZPAQL.cs:158-179 */
static long vm_write(const vm_t *z, U8 *buf, size_t cap, int pp) {
  if (z->header_len <= 6) return 0;
  size_t o = 0;
  if (!pp) { for (int i = 0; i < z->cend; ++i) { if (o < cap) buf[o] = z->header[i]; ++o; } }
  else {
    if (o < cap) buf[o] = (U8)((z->hend - z->hbegin) & 255);
    ++o;
    if (o < cap) buf[o] = (U8)((z->hend - z->hbegin) >> 8);
    ++o;
  }
  for (int i = z->hbegin; i < z->hend; ++i) { if (o < cap) buf[o] = z->header[i]; ++o; }
  return (long)o;
}

/* ZPAQL.cs:194-207 (outbuf staging collapsed: bytes go straight to the sink
 * and the SHA-1, which is what flush() amounts to). */
static inline void vm_outc(vm_t *z, int ch) {
  if (ch < 0) return;
  U8 b = (U8)ch;
  if (z->output) wr_put(z->output, b);
  if (z->sha1) sha1_write(z->sha1, &b, 1);
}

#define M(i) z->m[(i) & (z->msize - 1)]
#define H(i) z->h[(i) & (z->hsize - 1)]

/* ZPAQL.cs:1028-1251 execute(), 1253-1265 run0(), 1267-1303 div/mod/swap.
 * Written from the ISA description (ZPAQL.cs:238-321) as a decode of the
 * opcode fields rather than a 256-way case list. */
/* Test guard, not reference behaviour: instructions one run() may execute (0 = unlimited, as in the reference).  A
 * post-processor fed bytes no encoder writes need not terminate (bwtrle's list traversal on a garbage BWT index); the
 * device bounds every run with zpaqhip_opts.zpaql_budget, and the arbitrary-input tests bound the oracle likewise. */
static U64 g_vm_budget = 0;
void zo_set_zpaql_budget(U64 per_run) { g_vm_budget = per_run; }

static void vm_run(vm_t *z, U32 input) {
  const U8 *hd = z->header;
  int pc = z->hbegin;
  U32 a = input, b = z->b, c = z->c, d = z->d; int f = z->f;
  U64 left = g_vm_budget ? g_vm_budget : ~(U64)0;
  for (;;) {
    if (left-- == 0) {
      z->a = a; z->b = b; z->c = c; z->d = d; z->f = f; z->pc = pc;
      zerror(z->err, "ZPAQL instruction budget exhausted");
    }
    int op = hd[pc++];
    if (op < 64) {
      int ddd = op >> 3, x = op & 7;
      if (x == 7) {                       /* 2-byte forms of group 00 */
        int n = hd[pc++];
        switch (ddd) {
          case 0: a = z->r[n]; break; case 1: b = z->r[n]; break;
          case 2: c = z->r[n]; break; case 3: d = z->r[n]; break;
          case 4: if (f) pc += ((n + 128) & 255) - 128; break;          /* JT */
          case 5: if (!f) pc += ((n + 128) & 255) - 128; break;         /* JF */
          case 6: z->r[n] = a; break;                                   /* R=A */
          case 7: pc += ((n + 128) & 255) - 128; break;                 /* JMP */
        }
        continue;
      }
      if (ddd == 7) {                     /* specials 56..62 */
        if (x == 0) break;                                              /* HALT */
        else if (x == 1) vm_outc(z, (int)(a & 255));                    /* OUT */
        else if (x == 3) a = (a + M(b) + 512) * 773;                    /* HASH */
        else if (x == 4) H(d) = (H(d) + a + 512) * 773;                 /* HASHD */
        else goto bad;
        continue;
      }
      if (x > 4 || op == 0) goto bad;
      /* unary op x on destination ddd: 0 <>a, 1 ++, 2 --, 3 !, 4 =0 */
      U32 v;
      switch (ddd) { case 0: v = a; break; case 1: v = b; break; case 2: v = c; break;
        case 3: v = d; break; case 4: v = M(b); break; case 5: v = M(c); break; default: v = H(d); }
      U32 olda = a;
      switch (x) {
        case 0: if (ddd == 4 || ddd == 5) { a = (a & ~255u) | (v & 255); v = olda & 255; }
                else { a = v; v = olda; } break;         /* *b<>a swaps low byte only :1298 */
        case 1: ++v; break; case 2: --v; break; case 3: v = ~v; break; default: v = 0;
      }
      switch (ddd) { case 0: if (x) a = v; break; case 1: b = v; break; case 2: c = v; break;
        case 3: d = v; break; case 4: M(b) = (U8)v; break; case 5: M(c) = (U8)v; break;
        default: H(d) = v; }
      continue;
    }
    if (op == 255) {                                                    /* LJ */
      pc = z->hbegin + hd[pc] + 256 * hd[pc + 1];
      if (pc >= z->hend) goto bad;
      continue;
    }
    {
      int sss = op & 7; U32 s;
      switch (sss) { case 0: s = a; break; case 1: s = b; break; case 2: s = c; break;
        case 3: s = d; break; case 4: s = M(b); break; case 5: s = M(c); break;
        case 6: s = H(d); break; default: s = hd[pc++]; }
      if (op < 128) {                     /* assignment 01dddsss */
        int ddd = (op >> 3) & 7;
        switch (ddd) { case 0: a = s; break; case 1: b = s; break; case 2: c = s; break;
          case 3: d = s; break; case 4: M(b) = (U8)s; break; case 5: M(c) = (U8)s; break;
          case 6: H(d) = s; break; default: goto bad; }
        continue;
      }
      switch ((op >> 3) & 15) {
        case 0: a += s; break; case 1: a -= s; break; case 2: a *= s; break;
        case 3: a = s ? a / s : 0; break; case 4: a = s ? a % s : 0; break;
        case 5: a &= s; break; case 6: a &= ~s; break; case 7: a |= s; break;
        case 8: a ^= s; break; case 9: a <<= (s & 31); break; case 10: a >>= (s & 31); break;
        case 11: f = a == s; break; case 12: f = a < s; break; case 13: f = a > s; break;
        default: goto bad;
      }
    }
  }
  z->a = a; z->b = b; z->c = c; z->d = d; z->f = f; z->pc = pc;
  return;
bad:
  z->a = a; z->b = b; z->c = c; z->d = d; z->f = f; z->pc = pc;
  zerror(z->err, "ZPAQL execution error");                              /* ZPAQL.cs:1314-1317 */
}
#undef M
#undef H

/* ======================================================================
 * Predictor                                           Predictor.cs
 * ====================================================================== */
typedef struct {                                   /* Component.cs:18-57 */
  U64 limit, cxt, a, b, c;
  U32 *cm; U64 cm_n;
  U8 *ht; U64 ht_n;
  U16 *a16; U64 a16_n;
} comp_t;

typedef struct {
  int c8, hmap4;
  int p[256]; U32 h[256];
  vm_t *z;
  comp_t comp[256];
  err_t *err;
  /* trace */
  U32 *tr; size_t tr_cap, *tr_n; U32 tr_crc; U32 tr_bits;
} pred_t;

static void comp_free(comp_t *c) { free(c->cm); free(c->ht); free(c->a16); memset(c, 0, sizeof *c); }

static void *xcalloc(err_t *e, U64 n, size_t sz) {
  if (n * sz > ((U64)1 << 33)) zerror(e, "oracle: table beyond host memory");
  void *p = calloc(n ? n : 1, sz);
  if (!p) zerror(e, "Out of memory");
  return p;
}

static int pred_is_modeled(const pred_t *pr) { return pr->z->header[6] != 0; }   /* Predictor.cs:229-233 */

/* Predictor.cs:39-172 */
static void pred_init(pred_t *pr) {
  vm_t *z = pr->z; err_t *e = pr->err;
  init_tables();
  vm_inith(z);
  for (int i = 0; i < 256; ++i) { pr->h[i] = 0; pr->p[i] = 0; comp_free(&pr->comp[i]); }
  pr->c8 = 1; pr->hmap4 = 1;
  int n = z->header[6];
  const U8 *cp = &z->header[7];
  for (int i = 0; i < n; ++i) {
    comp_t *cr = &pr->comp[i];
    switch (cp[0]) {
      case CONS: pr->p[i] = (cp[1] - 128) * 4; break;
      case CM:
        if (cp[1] > 32) zerror(e, "max size for CM is 32");
        cr->cm_n = (U64)1 << cp[1]; cr->cm = xcalloc(e, cr->cm_n, 4);
        cr->limit = (U64)cp[2] * 4;
        for (U64 j = 0; j < cr->cm_n; ++j) cr->cm[j] = 0x80000000u;
        break;
      case ICM:
        if (cp[1] > 26) zerror(e, "max size for ICM is 26");
        cr->limit = 1023;
        cr->cm_n = 256; cr->cm = xcalloc(e, 256, 4);
        cr->ht_n = (U64)64 << cp[1]; cr->ht = xcalloc(e, cr->ht_n, 1);
        for (int j = 0; j < 256; ++j) cr->cm[j] = st_cminit(j);
        break;
      case MATCH:
        if (cp[1] > 32 || cp[2] > 32) zerror(e, "max size for MATCH is 32 32");
        cr->cm_n = (U64)1 << cp[1]; cr->cm = xcalloc(e, cr->cm_n, 4);
        cr->ht_n = (U64)1 << cp[2]; cr->ht = xcalloc(e, cr->ht_n, 1);
        cr->ht[0] = 1;
        break;
      case AVG:
        if (cp[1] >= i) zerror(e, "AVG j >= i");
        if (cp[2] >= i) zerror(e, "AVG k >= i");
        break;
      case MIX2:
        if (cp[1] > 32) zerror(e, "max size for MIX2 is 32");
        if (cp[3] >= i) zerror(e, "MIX2 k >= i");
        if (cp[2] >= i) zerror(e, "MIX2 j >= i");
        cr->c = (U64)1 << cp[1];
        cr->a16_n = cr->c; cr->a16 = xcalloc(e, cr->a16_n, 2);
        for (U64 j = 0; j < cr->a16_n; ++j) cr->a16[j] = 32768;
        break;
      case MIX: {
        if (cp[1] > 32) zerror(e, "max size for MIX is 32");
        if (cp[2] >= i) zerror(e, "MIX j >= i");
        if (cp[3] < 1 || cp[3] > i - cp[2]) zerror(e, "MIX m not in 1..i-j");
        int m = cp[3];
        cr->c = (U64)1 << cp[1];
        cr->cm_n = (U64)m << cp[1]; cr->cm = xcalloc(e, cr->cm_n, 4);
        for (U64 j = 0; j < cr->cm_n; ++j) cr->cm[j] = (U32)(65536 / m);
        break;
      }
      case ISSE:
        if (cp[1] > 32) zerror(e, "max size for ISSE is 32");
        if (cp[2] >= i) zerror(e, "ISSE j >= i");
        cr->ht_n = (U64)64 << cp[1]; cr->ht = xcalloc(e, cr->ht_n, 1);
        cr->cm_n = 512; cr->cm = xcalloc(e, 512, 4);
        for (int j = 0; j < 256; ++j) {
          cr->cm[j * 2] = 1 << 15;
          cr->cm[j * 2 + 1] = (U32)clamp512k(stretch((int)(st_cminit(j) >> 8)) * 1024);
        }
        break;
      case SSE:
        if (cp[1] > 32) zerror(e, "max size for SSE is 32");
        if (cp[2] >= i) zerror(e, "SSE j >= i");
        if (cp[3] > cp[4] * 4) zerror(e, "SSE start > limit*4");
        cr->cm_n = (U64)32 << cp[1]; cr->cm = xcalloc(e, cr->cm_n, 4);
        cr->limit = (U64)cp[4] * 4;
        for (U64 j = 0; j < cr->cm_n; ++j)
          cr->cm[j] = (U32)squash((int)(j & 31) * 64 - 992) << 17 | cp[3];
        break;
      default: zerror(e, "unknown component type");
    }
    cp += compsize[cp[0]];
  }
}

#define CMX(cr, i) (cr)->cm[(i) & ((cr)->cm_n - 1)]     /* Array "()" masked index */
#define HTX(cr, i) (cr)->ht[(i) & ((cr)->ht_n - 1)]

/* Predictor.cs:550-567 */
static U64 pred_find(comp_t *cr, int sizebits, U32 cxt) {
  U8 *ht = cr->ht;
  int chk = (cxt >> sizebits) & 255;
  U64 h0 = ((U64)(U32)(cxt * 16u)) & (cr->ht_n - 16);
  if (ht[h0] == chk) return h0;
  U64 h1 = h0 ^ 16;
  if (ht[h1] == chk) return h1;
  U64 h2 = h0 ^ 32;
  if (ht[h2] == chk) return h2;
  U64 v;
  if (ht[h0 + 1] <= ht[h1 + 1] && ht[h0 + 1] <= ht[h2 + 1]) v = h0;
  else if (ht[h1 + 1] < ht[h2 + 1]) v = h1;
  else v = h2;
  memset(&ht[v], 0, 16); ht[v] = (U8)chk;
  return v;
}

/* Predictor.cs:245-350 */
static int pred_predict(pred_t *pr) {
  const vm_t *z = pr->z;
  int n = z->header[6];
  const U8 *cp = &z->header[7];
  int *p = pr->p; const U32 *h = pr->h; const int c8 = pr->c8, hmap4 = pr->hmap4;
  for (int i = 0; i < n; ++i) {
    comp_t *cr = &pr->comp[i];
    switch (cp[0]) {
      case CONS: break;
      case CM:
        cr->cxt = h[i] ^ (U32)hmap4;
        p[i] = stretch((int)(CMX(cr, cr->cxt) >> 17));
        break;
      case ICM:
        if (c8 == 1 || (c8 & 0xf0) == 16) cr->c = pred_find(cr, cp[1] + 2, h[i] + 16u * (U32)c8);
        cr->cxt = cr->ht[cr->c + (U64)(hmap4 & 15)];
        p[i] = stretch((int)(CMX(cr, cr->cxt) >> 8));
        break;
      case MATCH:
        if (cr->a == 0) p[i] = 0;
        else {
          cr->c = (HTX(cr, cr->limit - cr->b) >> (7 - cr->cxt)) & 1;
          p[i] = stretch((dt2k[cr->a] * (1 - 2 * (int)cr->c)) & 32767);
        }
        break;
      case AVG:
        p[i] = (p[cp[1]] * cp[3] + p[cp[2]] * (256 - cp[3])) >> 8;
        break;
      case MIX2: {
        cr->cxt = (h[i] + (U32)(c8 & cp[5])) & (cr->c - 1);
        int w = cr->a16[cr->cxt];
        p[i] = (w * p[cp[2]] + (65536 - w) * p[cp[3]]) >> 16;
        break;
      }
      case MIX: {
        int m = cp[3];
        cr->cxt = h[i] + (U32)(c8 & cp[5]);
        cr->cxt = (cr->cxt & (cr->c - 1)) * (U64)m;
        const int *wt = (const int *)&cr->cm[cr->cxt];
        int s = 0;
        for (int j = 0; j < m; ++j) s += (wt[j] >> 8) * p[cp[2] + j];
        p[i] = clamp2k(s >> 8);
        break;
      }
      case ISSE: {
        if (c8 == 1 || (c8 & 0xf0) == 16) cr->c = pred_find(cr, cp[1] + 2, h[i] + 16u * (U32)c8);
        cr->cxt = cr->ht[cr->c + (U64)(hmap4 & 15)];
        const int *wt = (const int *)&cr->cm[cr->cxt * 2];
        p[i] = clamp2k((wt[0] * p[cp[2]] + wt[1] * 64) >> 16);
        break;
      }
      case SSE: {
        cr->cxt = (U64)(U32)((h[i] + (U32)c8) * 32u);
        int pq = p[cp[2]] + 992;
        if (pq < 0) pq = 0;
        if (pq > 1983) pq = 1983;
        int wt = pq & 63;
        pq >>= 6;
        cr->cxt += (U64)pq;
        p[i] = stretch((int)(((CMX(cr, cr->cxt) >> 10) * (U32)(64 - wt) +
                              (CMX(cr, cr->cxt + 1) >> 10) * (U32)wt) >> 13));
        cr->cxt += (U64)(wt >> 5);
        break;
      }
      default: zerror(pr->err, "component predict not implemented");
    }
    cp += compsize[cp[0]];
  }
  return squash(p[n - 1]);
}

/* Predictor.cs:486-493 in the intended form kept at :1031-1036 */
static inline void pred_train(comp_t *cr, int y) {
  U32 *pn = &CMX(cr, cr->cxt);
  U32 count = *pn & 0x3ff;
  int error = y * 32767 - (int)(*pn >> 17);
  *pn += ((U32)error * (U32)dt[count] & 0xFFFFFC00u) + (count < cr->limit);
}

/* Predictor.cs:353-475 */
static void pred_update(pred_t *pr, int y) {
  vm_t *z = pr->z;
  int n = z->header[6];
  const U8 *cp = &z->header[7];
  int *p = pr->p; U32 *h = pr->h; const int hmap4 = pr->hmap4;
  for (int i = 0; i < n; ++i) {
    comp_t *cr = &pr->comp[i];
    switch (cp[0]) {
      case CONS: break;
      case CM: pred_train(cr, y); break;
      case ICM: {
        U8 *bh = &cr->ht[cr->c + (U64)(hmap4 & 15)];
        *bh = (U8)st_nex(*bh, y);
        U32 *pn = &CMX(cr, cr->cxt);
        *pn += (U32)((int)(y * 32767 - (int)(*pn >> 8)) >> 2);
        break;
      }
      case MATCH: {
        if ((int)cr->c != y) cr->a = 0;
        U8 *bp = &HTX(cr, cr->limit);
        *bp = (U8)(*bp + *bp + y);
        if (++cr->cxt == 8) {
          cr->cxt = 0;
          ++cr->limit;
          cr->limit &= ((U64)1 << cp[2]) - 1;
          if (cr->a == 0) {
            cr->b = cr->limit - CMX(cr, h[i]);
            if (cr->b & (cr->ht_n - 1))
              while (cr->a < 255 && HTX(cr, cr->limit - cr->a - 1) == HTX(cr, cr->limit - cr->a - cr->b - 1))
                ++cr->a;
          } else cr->a += cr->a < 255;
          CMX(cr, h[i]) = (U32)cr->limit;
        }
        break;
      }
      case AVG: break;
      case MIX2: {
        int err = (y * 32767 - squash(p[i])) * cp[4] >> 5;
        int w = cr->a16[cr->cxt];
        w += (err * (p[cp[2]] - p[cp[3]]) + (1 << 12)) >> 13;
        if (w < 0) w = 0;
        if (w > 65535) w = 65535;
        cr->a16[cr->cxt] = (U16)w;
        break;
      }
      case MIX: {
        int m = cp[3];
        int err = (y * 32767 - squash(p[i])) * cp[4] >> 4;
        int *wt = (int *)&cr->cm[cr->cxt];
        for (int j = 0; j < m; ++j)
          wt[j] = clamp512k(wt[j] + ((err * p[cp[2] + j] + (1 << 12)) >> 13));
        break;
      }
      case ISSE: {
        int err = y * 32767 - squash(p[i]);
        int *wt = (int *)&cr->cm[cr->cxt * 2];
        wt[0] = clamp512k(wt[0] + ((err * p[cp[2]] + (1 << 12)) >> 13));
        wt[1] = clamp512k(wt[1] + ((err + 16) >> 5));
        cr->ht[cr->c + (U64)(hmap4 & 15)] = (U8)st_nex((int)cr->cxt, y);
        break;
      }
      case SSE: pred_train(cr, y); break;
    }
    cp += compsize[cp[0]];
  }
  /* Predictor.cs:463-474 */
  pr->c8 += pr->c8 + y;
  if (pr->c8 >= 256) {
    vm_run(z, (U32)(pr->c8 - 256));
    pr->hmap4 = 1;
    pr->c8 = 1;
    for (int i = 0; i < n; ++i) h[i] = z->h[(U64)i & (z->hsize - 1)];    /* H(i), masked */
  } else if (pr->c8 >= 16 && pr->c8 < 32)
    pr->hmap4 = (pr->hmap4 & 0xf) << 5 | y << 4 | 1;
  else
    pr->hmap4 = (pr->hmap4 & 0x1f0) | (((pr->hmap4 & 0xf) * 2 + y) & 0xf);
}

static inline void pred_trace(pred_t *pr, int p, int y) {
  if (!pr->tr) return;
  U8 b[2] = {(U8)((p << 1 | y) & 255), (U8)((p << 1 | y) >> 8)};
  pr->tr_crc = crc32_update(pr->tr_crc, b, 2);
  if (++pr->tr_bits == 4096) {
    if (*pr->tr_n < pr->tr_cap) pr->tr[*pr->tr_n] = pr->tr_crc;
    ++*pr->tr_n; pr->tr_bits = 0;
  }
}

/* ======================================================================
 * Decoder                                               Decoder.cs
 * ====================================================================== */
typedef struct {
  rd_t in;
  U32 low, high, curr;
  pred_t pr;
  err_t *err;
} decoder_t;

/* Decoder.cs:136-158 */
static inline int dec_decode(decoder_t *d, int p) {
  if (d->curr < d->low || d->curr > d->high) zerror(d->err, "archive corrupted");
  U32 mid = d->low + (U32)(((U64)(d->high - d->low) * (U32)p) >> 16);
  int y;
  if (d->curr <= mid) { y = 1; d->high = mid; }
  else { y = 0; d->low = mid + 1; }
  while ((d->high ^ d->low) < 0x1000000) {
    d->high = d->high << 8 | 255;
    d->low = d->low << 8;
    d->low += (d->low == 0);
    int c = rd_get(&d->in);
    if (c < 0) zerror(d->err, "unexpected end of file");
    d->curr = d->curr << 8 | (U32)c;
  }
  return y;
}

/* Decoder.cs:32-68 */
static int dec_decompress(decoder_t *d) {
  if (pred_is_modeled(&d->pr)) {
    if (d->curr == 0)   /* get() == -1 at EOF sets all 32 bits, as the int -> U32 conversion of Decoder.cs:38-39 does */
      for (int i = 0; i < 4; ++i) d->curr = d->curr << 8 | (U32)rd_get(&d->in);
    if (dec_decode(d, 0)) {
      if (d->curr != 0) zerror(d->err, "decoding end of stream");
      return -1;
    }
    int c = 1;
    while (c < 256) {
      int p = pred_predict(&d->pr) * 2 + 1;
      int y = dec_decode(d, p);
      c += c + y;
      pred_trace(&d->pr, p, y);
      pred_update(&d->pr, y);
    }
    return c - 256;
  } else {
    if (d->curr == 0) {
      for (int i = 0; i < 4; ++i) d->curr = d->curr << 8 | (U32)rd_get(&d->in);   /* Decoder.cs:62: EOF -> all ones */
      if (d->curr == 0) return -1;
    }
    --d->curr;
    return rd_get(&d->in);
  }
}

/* Decoder.cs:70-98 */
static int dec_skip(decoder_t *d) {
  int c = -1;
  if (pred_is_modeled(&d->pr)) {
    while (d->curr == 0) d->curr = (U32)rd_get(&d->in);
    while (d->curr && (c = rd_get(&d->in)) >= 0) d->curr = d->curr << 8 | (U32)c;
    while ((c = rd_get(&d->in)) == 0) ;
    return c;
  } else {
    if (d->curr == 0)
      for (int i = 0; i < 4 && (c = rd_get(&d->in)) >= 0; ++i) d->curr = d->curr << 8 | (U32)c;
    while (d->curr > 0) {
      while (d->curr > 0) {
        --d->curr;
        if (rd_get(&d->in) < 0) zerror(d->err, "skipped to EOF");
      }
      for (int i = 0; i < 4 && (c = rd_get(&d->in)) >= 0; ++i) d->curr = d->curr << 8 | (U32)c;
    }
    if (c >= 0) c = rd_get(&d->in);
    return c;
  }
}

/* Decoder.cs:100-105 */
static void dec_init(decoder_t *d) {
  pred_init(&d->pr);
  if (pred_is_modeled(&d->pr)) { d->low = 1; d->high = 0xFFFFFFFFu; d->curr = 0; }
  else d->low = d->high = d->curr = 0;
}

/* ======================================================================
 * PostProcessor                                   PostProcessor.cs
 * ====================================================================== */
typedef struct { int state, hsize, ph, pm; vm_t z; err_t *err; } pp_t;

static void pp_init(pp_t *pp, int h, int m) {                        /* :27-33 */
  pp->state = pp->hsize = 0; pp->ph = h; pp->pm = m;
  wr_t *o = pp->z.output; sha1_t *s = pp->z.sha1;
  vm_clear(&pp->z); pp->z.output = o; pp->z.sha1 = s;
}

static int pp_write(pp_t *pp, int c) {                               /* :37-86 */
  vm_t *z = &pp->z; err_t *e = pp->err;
  switch (pp->state) {
    case 0:
      if (c < 0) zerror(e, "Unexpected EOS");
      pp->state = c + 1;
      if (pp->state > 2) zerror(e, "unknown post processing type");
      if (pp->state == 1) { wr_t *o = z->output; sha1_t *s = z->sha1; vm_clear(z); z->output = o; z->sha1 = s; }
      break;
    case 1: vm_outc(z, c); break;
    case 2:
      if (c < 0) zerror(e, "Unexpected EOS");
      pp->hsize = c; pp->state = 3; break;
    case 3:
      if (c < 0) zerror(e, "Unexpected EOS");
      pp->hsize += c * 256;
      if (pp->hsize < 1) zerror(e, "Empty PCOMP");
      free(z->header);
      z->header_len = (size_t)pp->hsize + 300;
      z->header = calloc(z->header_len, 1);
      z->cend = 8; z->hbegin = z->hend = z->cend + 128;
      z->header[4] = (U8)pp->ph; z->header[5] = (U8)pp->pm;
      pp->state = 4; break;
    case 4:
      if (c < 0) zerror(e, "Unexpected EOS");
      z->header[z->hend++] = (U8)c;
      if (z->hend - z->hbegin == pp->hsize) {
        pp->hsize = z->cend - 2 + z->hend - z->hbegin;
        z->header[0] = (U8)(pp->hsize & 255);
        z->header[1] = (U8)(pp->hsize >> 8);
        vm_initp(z);
        pp->state = 5;
      }
      break;
    case 5:
      vm_run(z, (U32)c);
      break;
  }
  return pp->state;
}

/* ======================================================================
 * Decompresser                                     Decompresser.cs
 * ====================================================================== */
enum { S_BLOCK, S_FILENAME, S_COMMENT, S_DATA, S_SEGEND };
enum { D_FIRSTSEG, D_SEG, D_SKIP };

struct zo_dec {
  vm_t z; decoder_t dec; pp_t pp;
  int state, decode_state;
  err_t err; wr_t out; sha1_t sha1;
};

zo_dec *zo_dec_new(void) {
  init_tables();
  zo_dec *d = calloc(1, sizeof *d);
  d->z.err = &d->err; d->pp.z.err = &d->err; d->pp.err = &d->err;
  d->dec.err = &d->err; d->dec.pr.err = &d->err; d->dec.pr.z = &d->z;
  d->dec.low = 1; d->dec.high = 0xFFFFFFFFu; d->dec.pr.c8 = 1; d->dec.pr.hmap4 = 1;
  d->state = S_BLOCK; d->decode_state = D_FIRSTSEG;
  return d;
}
void zo_dec_free(zo_dec *d) {
  if (!d) return;
  for (int i = 0; i < 256; ++i) comp_free(&d->dec.pr.comp[i]);
  vm_clear(&d->z); vm_clear(&d->pp.z); free(d);
}
const char *zo_dec_error(const zo_dec *d) { return d->err.msg; }
void zo_dec_set_input(zo_dec *d, const U8 *p, size_t n) { d->dec.in.p = p; d->dec.in.n = n; d->dec.in.pos = 0; }
size_t zo_dec_tell(const zo_dec *d) { return d->dec.in.pos; }
void zo_dec_set_trace(zo_dec *d, U32 *dg, size_t cap, size_t *n) {
  d->dec.pr.tr = dg; d->dec.pr.tr_cap = cap; d->dec.pr.tr_n = n; d->dec.pr.tr_crc = 0; d->dec.pr.tr_bits = 0;
  if (n) *n = 0;
}
void zo_dec_state(const zo_dec *d, U32 st[8]) {
  st[0] = d->dec.low; st[1] = d->dec.high; st[2] = d->dec.curr;
  st[3] = (U32)d->dec.pr.c8; st[4] = (U32)d->dec.pr.hmap4;
  st[5] = d->dec.pr.h[0]; st[6] = d->dec.pr.h[1]; st[7] = d->dec.pr.h[2];
}

#define GUARD(d) do { (d)->err.msg[0] = 0; (d)->err.armed = 1; \
  if (setjmp((d)->err.jb)) { (d)->err.armed = 0; return -1; } } while (0)
#define DONE(d, v) do { (d)->err.armed = 0; return (v); } while (0)

/* Decompresser.cs:29-58 */
int zo_dec_find_block(zo_dec *d, double *mem) {
  GUARD(d);
  U32 h1 = 0x3D49B113, h2 = 0x29EB7F93, h3 = 0x2614BE13, h4 = 0x3828EB13;
  int c;
  while ((c = rd_get(&d->dec.in)) != -1) {
    h1 = h1 * 12 + (U32)c; h2 = h2 * 20 + (U32)c; h3 = h3 * 28 + (U32)c; h4 = h4 * 44 + (U32)c;
    if (h1 == 0xB16B88F1 && h2 == 0xFF5376F1 && h3 == 0x72AC5BF1 && h4 == 0x2F909AF1) break;
  }
  if (c == -1) DONE(d, 0);
  if ((c = rd_get(&d->dec.in)) != 1 && c != 2) zerror(&d->err, "unsupported ZPAQ level");
  if (rd_get(&d->dec.in) != 1) zerror(&d->err, "unsupported ZPAQL type");
  vm_read(&d->z, &d->dec.in);
  if (c == 1 && d->z.header_len > 6 && d->z.header[6] == 0)
    zerror(&d->err, "ZPAQ level 1 requires at least 1 component");
  if (mem) *mem = vm_memory(&d->z);
  d->state = S_FILENAME; d->decode_state = D_FIRSTSEG;
  DONE(d, 1);
}
void zo_tag_hash(const U8 *p, size_t n, U32 h[4]) {
  U32 h1 = 0x3D49B113, h2 = 0x29EB7F93, h3 = 0x2614BE13, h4 = 0x3828EB13;
  for (size_t i = 0; i < n; ++i) { h1 = h1 * 12 + p[i]; h2 = h2 * 20 + p[i]; h3 = h3 * 28 + p[i]; h4 = h4 * 44 + p[i]; }
  h[0] = h1; h[1] = h2; h[2] = h3; h[3] = h4;
}

/* Decompresser.cs:67-93 */
int zo_dec_find_filename(zo_dec *d, char *buf, size_t cap) {
  GUARD(d);
  if (d->state != S_FILENAME) zerror(&d->err, "oracle: findFilename out of order");
  size_t o = 0;
  int c = rd_get(&d->dec.in);
  if (c == 1) {
    for (;;) {
      c = rd_get(&d->dec.in);
      if (c == -1) zerror(&d->err, "unexpected EOF");
      if (c == 0) { if (buf && o < cap) buf[o] = 0; d->state = S_COMMENT; DONE(d, 1); }
      if (buf && o + 1 < cap) buf[o++] = (char)c;
    }
  } else if (c == 255) { d->state = S_BLOCK; DONE(d, 0); }
  else zerror(&d->err, "missing segment or end of block");
  DONE(d, 0);
}

/* Decompresser.cs:96-108 */
int zo_dec_read_comment(zo_dec *d, char *buf, size_t cap) {
  GUARD(d);
  if (d->state != S_COMMENT) zerror(&d->err, "oracle: readComment out of order");
  d->state = S_DATA;
  size_t o = 0;
  for (;;) {
    int c = rd_get(&d->dec.in);
    if (c == -1) zerror(&d->err, "unexpected EOF");
    if (c == 0) break;
    if (buf && o + 1 < cap) buf[o++] = (char)c;
  }
  if (buf && o < cap) buf[o] = 0;
  if (rd_get(&d->dec.in) != 0) zerror(&d->err, "missing reserved byte");
  DONE(d, 0);
}

/* Decompresser.cs:121-153 */
int zo_dec_decompress(zo_dec *d, long n_, U8 *out, size_t cap, size_t *len) {
  volatile long n = n_;
  GUARD(d);
  if (d->state != S_DATA) zerror(&d->err, "oracle: decompress out of order");
  if (d->decode_state == D_SKIP) zerror(&d->err, "decompression after skipped segment");
  d->out.p = out; d->out.cap = cap; d->out.len = *len; d->out.overflow = 0;
  d->pp.z.output = &d->out;
  if (d->decode_state == D_FIRSTSEG) {
    dec_init(&d->dec);
    pp_init(&d->pp, d->z.header[4], d->z.header[5]);
    d->decode_state = D_SEG;
  }
  while ((d->pp.state & 3) != 1) pp_write(&d->pp, dec_decompress(&d->dec));
  int more = 1;
  while (n) {
    int c = dec_decompress(&d->dec);
    pp_write(&d->pp, c);
    if (c == -1) { d->state = S_SEGEND; more = 0; break; }
    if (n > 0) --n;
  }
  *len = d->out.len;
  if (d->out.overflow) zerror(&d->err, "oracle: output buffer too small");
  DONE(d, more);
}

long zo_dec_hcomp(zo_dec *d, U8 *buf, size_t cap) { return vm_write(&d->z, buf, cap, 0); }
long zo_dec_pcomp(zo_dec *d, U8 *buf, size_t cap) { return vm_write(&d->pp.z, buf, cap, 1); }

/* Decompresser.cs:163-194 */
int zo_dec_read_segment_end(zo_dec *d, U8 sha1[21]) {
  GUARD(d);
  int c = 0;
  if (d->state == S_DATA) { c = dec_skip(&d->dec); d->decode_state = D_SKIP; }
  else if (d->state == S_SEGEND) c = rd_get(&d->dec.in);
  else zerror(&d->err, "oracle: readSegmentEnd out of order");
  d->state = S_FILENAME;
  if (c == 254) { if (sha1) sha1[0] = 0; }
  else if (c == 253) {
    if (sha1) sha1[0] = 1;
    for (int i = 1; i <= 20; ++i) { c = rd_get(&d->dec.in); if (sha1) sha1[i] = (U8)c; }
  } else zerror(&d->err, "missing end of segment marker");
  DONE(d, 0);
}

/* LibZPAQ.cs:65-79 */
long zo_decompress(const U8 *in, size_t n, U8 *out, size_t cap, char *err, size_t errcap) {
  zo_dec *d = zo_dec_new();
  zo_dec_set_input(d, in, n);
  size_t len = 0; int r; long ret = -1;
  if (err && errcap) err[0] = 0;
  while ((r = zo_dec_find_block(d, 0)) == 1) {
    while ((r = zo_dec_find_filename(d, 0, 0)) == 1) {
      if (zo_dec_read_comment(d, 0, 0) < 0) goto fail;
      if (zo_dec_decompress(d, -1, out, cap, &len) < 0) goto fail;
      if (zo_dec_read_segment_end(d, 0) < 0) goto fail;
    }
    if (r < 0) goto fail;
  }
  if (r < 0) goto fail;
  ret = (long)len;
fail:
  if (ret < 0 && err && errcap) snprintf(err, errcap, "%s", zo_dec_error(d));
  zo_dec_free(d);
  return ret;
}

/* ======================================================================
 * Encoder / Compressor mirror              Encoder.cs, Compressor.cs
 * ====================================================================== */
struct zo_enc {
  vm_t z; pred_t pr; wr_t out; err_t err;
  U32 low, high;
  int state;     /* 0 INIT, 1 BLOCK1, 2 SEG1, 3 BLOCK2, 4 SEG2   Compressor.cs:314-321 */
};

zo_enc *zo_enc_new(U8 *out, size_t cap) {
  init_tables();
  zo_enc *e = calloc(1, sizeof *e);
  e->out.p = out; e->out.cap = cap;
  e->z.err = &e->err; e->pr.err = &e->err; e->pr.z = &e->z;
  e->low = 1; e->high = 0xFFFFFFFFu; e->pr.c8 = 1; e->pr.hmap4 = 1;
  return e;
}
void zo_enc_free(zo_enc *e) {
  if (!e) return;
  for (int i = 0; i < 256; ++i) comp_free(&e->pr.comp[i]);
  vm_clear(&e->z); free(e);
}
const char *zo_enc_error(const zo_enc *e) { return e->err.msg; }
size_t zo_enc_tell(const zo_enc *e) { return e->out.len; }

/* Compressor.cs:27-43 */
int zo_enc_write_tag(zo_enc *e) {
  static const U8 tag[13] = {0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3};
  for (int i = 0; i < 13; ++i) wr_put(&e->out, tag[i]);
  return 0;
}

/* Compressor.cs:85-99 */
int zo_enc_start_block(zo_enc *e, const U8 *hdr, size_t hdrlen) {
  GUARD(e);
  rd_t r = {hdr, hdrlen, 0};
  vm_read(&e->z, &r);
  wr_put(&e->out, 'z'); wr_put(&e->out, 'P'); wr_put(&e->out, 'Q');
  wr_put(&e->out, 1 + (e->z.header[6] == 0));
  wr_put(&e->out, 1);
  for (int i = 0; i < e->z.cend; ++i) wr_put(&e->out, e->z.header[i]);          /* ZPAQL.write(out,false) */
  for (int i = e->z.hbegin; i < e->z.hend; ++i) wr_put(&e->out, e->z.header[i]);
  e->state = 1;
  DONE(e, 0);
}

/* Compressor.cs:133-146 */
int zo_enc_start_segment(zo_enc *e, const char *filename, const char *comment) {
  wr_put(&e->out, 1);
  while (filename && *filename) wr_put(&e->out, (U8)*filename++);
  wr_put(&e->out, 0);
  while (comment && *comment) wr_put(&e->out, (U8)*comment++);
  wr_put(&e->out, 0);
  wr_put(&e->out, 0);
  if (e->state == 1) e->state = 2;
  if (e->state == 3) e->state = 4;
  return 0;
}

/* Encoder.cs:87-103 */
static inline void enc_encode(zo_enc *e, int y, int p) {
  U32 mid = e->low + (U32)(((U64)(e->high - e->low) * (U32)p) >> 16);
  if (y) e->high = mid; else e->low = mid + 1;
  while ((e->high ^ e->low) < 0x1000000) {
    wr_put(&e->out, (int)(e->high >> 24));
    e->high = e->high << 8 | 255;
    e->low = e->low << 8;
    e->low += (e->low == 0);
  }
}

/* Encoder.cs:39-73 (modelled path; the n=0 store path is below) */
static void enc_compress(zo_enc *e, int c) {
  if (c == -1) enc_encode(e, 1, 0);
  else {
    enc_encode(e, 0, 0);
    for (int i = 7; i >= 0; --i) {
      int p = pred_predict(&e->pr) * 2 + 1;
      int y = c >> i & 1;
      enc_encode(e, y, p);
      pred_update(&e->pr, y);
    }
  }
}

/* Compressor.cs:156-190 */
int zo_enc_post_process(zo_enc *e, const U8 *pcomp, size_t len) {
  GUARD(e);
  if (e->state == 4) DONE(e, 0);
  if (e->state != 2) zerror(&e->err, "oracle: postProcess out of order");
  e->low = 1; e->high = 0xFFFFFFFFu;                                   /* Encoder.cs:26-37 */
  pred_init(&e->pr);
  if (!pred_is_modeled(&e->pr)) zerror(&e->err, "oracle: unmodelled (n=0) encode not supported");
  if (pcomp && len > 0) {
    enc_compress(e, 1);
    enc_compress(e, (int)(len & 255));
    enc_compress(e, (int)((len >> 8) & 255));
    for (size_t i = 0; i < len; ++i) enc_compress(e, pcomp[i]);
  } else enc_compress(e, 0);
  e->state = 4;
  DONE(e, 0);
}

/* Test hook (no reference counterpart): starts the coder of a segment like postProcess does but emits no
 * post-processor selector, so that the caller's bytes ARE the decoded stream PostProcessor.write sees
 * (malformed selectors, truncated PCOMP headers: PostProcessor.cs:43-68). */
int zo_enc_begin_raw(zo_enc *e) {
  GUARD(e);
  if (e->state == 4) DONE(e, 0);
  if (e->state != 2) zerror(&e->err, "oracle: begin_raw out of order");
  e->low = 1; e->high = 0xFFFFFFFFu;
  pred_init(&e->pr);
  if (!pred_is_modeled(&e->pr)) zerror(&e->err, "oracle: unmodelled (n=0) encode not supported");
  e->state = 4;
  DONE(e, 0);
}

/* Compressor.cs:193-221 */
int zo_enc_compress(zo_enc *e, const U8 *data, size_t n) {
  if (e->state == 2 && zo_enc_post_process(e, 0, 0) < 0) return -1;
  GUARD(e);
  if (e->state != 4) zerror(&e->err, "oracle: compress out of order");
  for (size_t i = 0; i < n; ++i) enc_compress(e, data[i]);
  DONE(e, 0);
}

/* Compressor.cs:224-248 */
int zo_enc_end_segment(zo_enc *e, const U8 sha1[20]) {
  if (e->state == 2 && zo_enc_post_process(e, 0, 0) < 0) return -1;
  GUARD(e);
  if (e->state != 4) zerror(&e->err, "oracle: endSegment out of order");
  enc_compress(e, -1);
  for (int i = 0; i < 4; ++i) wr_put(&e->out, 0);
  if (sha1) { wr_put(&e->out, 253); for (int i = 0; i < 20; ++i) wr_put(&e->out, sha1[i]); }
  else wr_put(&e->out, 254);
  e->state = 3;
  DONE(e, 0);
}

/* Compressor.cs:294-299 */
int zo_enc_end_block(zo_enc *e) { wr_put(&e->out, 255); e->state = 0; return e->out.overflow ? -1 : 0; }

/* ======================================================================
 * helpers
 * ====================================================================== */
/* LibZPAQ.cs:372-384 */
void zo_e8e9(U8 *buf, size_t n) {
  for (long i = (long)n - 5; i >= 0; --i) {
    if (((buf[i] & 254) == 0xe8) && ((buf[i + 4] + 1) & 254) == 0) {
      unsigned a = (buf[i + 1] | buf[i + 2] << 8 | buf[i + 3] << 16) + (unsigned)i;
      buf[i + 1] = (U8)a; buf[i + 2] = (U8)(a >> 8); buf[i + 3] = (U8)(a >> 16);
    }
  }
}

long zo_run_pcomp(const U8 *pcomp, size_t plen, int ph, int pm, const U8 *in, size_t n, U8 *out, size_t cap) {
  err_t err; memset(&err, 0, sizeof err);
  pp_t pp; memset(&pp, 0, sizeof pp);
  wr_t w = {out, cap, 0, 0};
  pp.err = &err; pp.z.err = &err; pp.z.output = &w;
  err.armed = 1;
  if (setjmp(err.jb)) { vm_clear(&pp.z); return -1; }
  pp_init(&pp, ph, pm);
  pp_write(&pp, 1); pp_write(&pp, (int)(plen & 255)); pp_write(&pp, (int)(plen >> 8));
  for (size_t i = 0; i < plen; ++i) pp_write(&pp, pcomp[i]);
  for (size_t i = 0; i < n; ++i) pp_write(&pp, in[i]);
  pp_write(&pp, -1);
  vm_clear(&pp.z);
  return (long)w.len;
}
