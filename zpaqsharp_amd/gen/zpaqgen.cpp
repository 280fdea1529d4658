// zpaqgen.cpp — libzpaqgen.so: CPU-side ZPAQ stream *writer* and synthetic
// workload generator.  This is the compress side the benchmarks and the
// full-size tests need to obtain ZPAQ streams (the reference ships none):
//   Encoder     Encoder.cs:26-103      (arithmetic coder, mirror of Decoder)
//   Compressor  Compressor.cs:27-299   (block / segment framing, PCOMP injection)
//   e8e9        LibZPAQ.cs:372-384     (forward E8E9 transform)
//   compress    LibZPAQ.cs:84-108,296-323 (one block, one segment, comment = size, SHA-1)
// The model (Predictor) is the scalar core of ../csrc/zh_core.h compiled for the
// host.  This library exports no decompression entry point at all: decoding is
// done only by libzpaqhip.so on the GPU.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../csrc/zh_core.h"
#include "../csrc/zh_host.h"

using namespace zhcore;

namespace {

// ---------------------------------------------------------------------------
// Host-side Predictor.init (Predictor.cs:82-171) for one arena slot.
// ---------------------------------------------------------------------------
uint32_t cminit(const ZhTables &t, int j) {          // StateTable.cs:158-162
  return (uint32_t)(((t.ns[j * 4 + 3] * 2 + 1) << 22) / (t.ns[j * 4 + 2] + t.ns[j * 4 + 3] + 1));
}

void host_init_slot(const ZhModel &M, uint8_t *slot, GenLds &S) {
  const ZhTables &t = S.t;
  for (uint32_t i = 0; i < M.n; ++i) {
    const ZhComp &cp = M.comp[i];
    uint32_t *cm = (uint32_t *)(slot + cp.cm_off);
    uint8_t *ht = slot + cp.ht_off;
    switch (cp.type) {
      case ZH_CM:
        for (uint64_t j = 0; j < cp.cm_bytes / 4; ++j) cm[j] = 0x80000000u;
        break;
      case ZH_ICM:
        memset(ht, 0, cp.ht_bytes);
        for (int j = 0; j < 256; ++j) cm[j] = cminit(t, j);
        break;
      case ZH_MATCH:
        memset(cm, 0, cp.cm_bytes);
        memset(ht, 0, cp.ht_bytes);
        ht[0] = 1;
        break;
      case ZH_MIX2:
        for (uint64_t j = 0; j < cp.cm_bytes / 2; ++j) ((uint16_t *)cm)[j] = 32768;
        break;
      case ZH_MIX:
        for (uint64_t j = 0; j < cp.cm_bytes / 4; ++j) cm[j] = 65536u / cp.arg[2];
        break;
      case ZH_ISSE:
        memset(ht, 0, cp.ht_bytes);
        for (int j = 0; j < 256; ++j) {
          ((int *)cm)[j * 2] = 1 << 15;
          ((int *)cm)[j * 2 + 1] = clamp512k(t.stretch[cminit(t, j) >> 8] * 1024);
        }
        break;
      case ZH_SSE:
        for (uint64_t j = 0; j < cp.cm_bytes / 4; ++j)
          cm[j] = (uint32_t)t.squash[(int)(j & 31) * 64 - 992 + 2048] << 17 | cp.arg[2];
        break;
      default: break;
    }
  }
  memset(slot + M.h_off, 0, M.arena_bytes - M.h_off);
  for (int i = 0; i < 256; ++i) {
    S.p[i] = 0; S.h[i] = 0; S.r[i] = 0; S.pr[i] = 0;
    S.cs[i] = CompSt{0, 0, 0, 0, 0};
  }
  for (uint32_t i = 0; i < M.n; ++i) {
    const ZhComp &cp = M.comp[i];
    switch (cp.type) {
      case ZH_CONS: S.p[i] = ((int)cp.arg[0] - 128) * 4; break;
      case ZH_CM: S.cs[i].limit = (uint32_t)cp.arg[1] * 4; break;
      case ZH_ICM: S.cs[i].limit = 1023; break;
      case ZH_MIX2: case ZH_MIX: S.cs[i].c = cp.cm_mask + 1; break;
      case ZH_SSE: S.cs[i].limit = (uint32_t)cp.arg[3] * 4; break;
      default: break;
    }
  }
}

struct Out {
  std::vector<uint8_t> v;
  void put(int c) { v.push_back((uint8_t)c); }
  void put(const void *p, size_t n) { v.insert(v.end(), (const uint8_t *)p, (const uint8_t *)p + n); }
};

// One block writer: Compressor + Encoder for a fixed model.
struct BlockWriter {
  ZhModel M;
  std::vector<uint8_t> code, hdr;
  std::unique_ptr<GenLds> S;
  std::vector<uint8_t> slot;
  Pred P;
  uint32_t low = 1, high = 0xFFFFFFFFu;

  int setup(const uint8_t *h, size_t n) {
    zpaqhip_err err;
    hdr.assign(h, h + n);
    int rc = zh::build_model(h, n, M, code, &err);
    if (rc) return rc;
    if (M.n == 0) return ZPAQHIP_E_ARG;               // unmodelled store blocks are not generated here
    S.reset(new GenLds);
    S->t = zh::host_tables();
    slot.resize(M.arena_bytes);
    return 0;
  }

  void start_block() {                                 // Encoder.init, Encoder.cs:26-37 + Predictor.init
    host_init_slot(M, slot.data(), *S);
    P.S = S.get();
    P.cd = M.comp;
    P.slot = slot.data();
    P.n = M.n;
    P.c8 = 1; P.hmap4 = 1;
    P.z.a = P.z.b = P.z.c = P.z.d = P.z.f = 0;
    P.z.prog = code.data() + M.code_off + ZH_CODE_PAD;
    P.z.len = M.hcomp_len;
    P.z.m = slot.data() + M.m_off; P.z.mmask = (uint32_t)((1ull << M.hm) - 1);
    P.z.h = (uint32_t *)(slot.data() + M.h_off); P.z.hmask = (uint32_t)((1ull << M.hh) - 1);
    P.z.r = S->r;
    low = 1; high = 0xFFFFFFFFu;
  }

  inline void encode(Out &o, int y, uint32_t p) {      // Encoder.cs:87-103
    uint32_t mid = low + (uint32_t)(((uint64_t)(high - low) * p) >> 16);
    if (y) high = mid; else low = mid + 1;
    while ((high ^ low) < 0x1000000u) {
      o.put((int)(high >> 24));
      high = high << 8 | 255;
      low = low << 8;
      low += (low == 0);
    }
  }

  int compress_byte(Out &o, int c) {                   // Encoder.cs:39-60
    if (c < 0) { encode(o, 1, 0); return 0; }
    encode(o, 0, 0);
    for (int i = 7; i >= 0; --i) {
      uint32_t p = (uint32_t)predict(P) * 2 + 1;
      int y = c >> i & 1;
      encode(o, y, p);
      int rc = update(P, y, 1ull << 32);
      if (rc) return rc;
    }
    return 0;
  }

  // tag + block header + one segment + end of block (LibZPAQ.cs:296-323 framing).
  int write_block(Out &o, const uint8_t *pcomp, size_t plen, const uint8_t *data, size_t n, const char *filename,
                  const char *comment, const uint8_t *sha /* 20 bytes or null */, bool tag) {
    static const uint8_t kTag[13] = {0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3};
    if (tag) o.put(kTag, 13);                          // Compressor.writeTag, Compressor.cs:27-43
    o.put('z'); o.put('P'); o.put('Q'); o.put(1); o.put(1);   // startBlock, Compressor.cs:92-96 (n>0 -> level 1)
    o.put(hdr.data(), hdr.size());
    o.put(1);                                          // startSegment, Compressor.cs:133-146
    if (filename) o.put(filename, strlen(filename));
    o.put(0);
    if (comment) o.put(comment, strlen(comment));
    o.put(0); o.put(0);
    start_block();
    int rc = 0;
    if (pcomp && plen) {                               // postProcess, Compressor.cs:156-190
      rc |= compress_byte(o, 1);
      rc |= compress_byte(o, (int)(plen & 255));
      rc |= compress_byte(o, (int)(plen >> 8 & 255));
      for (size_t i = 0; i < plen && !rc; ++i) rc |= compress_byte(o, pcomp[i]);
    } else rc |= compress_byte(o, 0);
    for (size_t i = 0; i < n && !rc; ++i) rc = compress_byte(o, data[i]);
    if (rc) return rc;
    compress_byte(o, -1);                              // endSegment, Compressor.cs:224-248
    o.put(0); o.put(0); o.put(0); o.put(0);
    if (sha) { o.put(253); o.put(sha, 20); } else o.put(254);
    o.put(255);                                        // endBlock, Compressor.cs:294-299
    return 0;
  }
};

// ---------------------------------------------------------------------------
// Synthetic plaintext: T text-like, X x86-like, R random.
// ---------------------------------------------------------------------------
struct SplitMix {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint32_t below(uint32_t n) { return (uint32_t)((next() >> 32) * (uint64_t)n >> 32); }
};

struct Vocab {
  std::vector<std::string> words;
  std::vector<uint32_t> cdf;                           // 32-bit cumulative Zipf(1.1) weights
  Vocab() {
    SplitMix r{1};
    for (int i = 0; i < 4096; ++i) {
      int len = 2 + (int)r.below(8);
      std::string w;
      for (int k = 0; k < len; ++k) w.push_back((char)('a' + r.below(26)));
      words.push_back(w);
    }
    std::vector<double> w(4096);
    double tot = 0;
    for (int i = 0; i < 4096; ++i) tot += (w[i] = 1.0 / pow(i + 1.0, 1.1));
    double acc = 0;
    for (int i = 0; i < 4096; ++i) {
      acc += w[i];
      cdf.push_back(i == 4095 ? 0xFFFFFFFFu : (uint32_t)(acc / tot * 4294967295.0));
    }
  }
  const std::string &pick(SplitMix &r) const {
    uint32_t u = (uint32_t)(r.next() >> 32);
    size_t lo = 0, hi = 4095;
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (cdf[mid] < u) lo = mid + 1; else hi = mid; }
    return words[lo];
  }
};

const Vocab &vocab() { static const Vocab v; return v; }

void gen_plain(int kind, uint64_t block, uint8_t *out, size_t n) {
  SplitMix r{0x5A50415153484152ull ^ block};
  if (kind == 2) {                                     // R
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t v = r.next(); memcpy(out + i, &v, 8); }
    for (uint64_t v = r.next(); i < n; ++i, v >>= 8) out[i] = (uint8_t)v;
    return;
  }
  if (kind == 0) {                                     // T
    const Vocab &V = vocab();
    size_t i = 0;
    int until = 8 + (int)r.below(9);
    while (i < n) {
      const std::string &w = V.pick(r);
      for (char ch : w) { if (i < n) out[i++] = (uint8_t)ch; }
      if (--until == 0) {
        for (char ch : {'.', ' ', '\n'}) if (i < n) out[i++] = (uint8_t)ch;
        until = 8 + (int)r.below(9);
      } else if (i < n) out[i++] = ' ';
    }
    return;
  }
  // X: opcode-ish filler with a CALL/JMP rel32 every 16-64 bytes whose target is
  // one of a few absolute addresses (so the forward E8E9 transform helps).
  static const uint8_t ops[16] = {0x8B, 0x89, 0x48, 0x83, 0xC3, 0x55, 0x5D, 0x0F, 0x85, 0x74, 0xFF, 0x24, 0x8D, 0x4C, 0x00, 0x01};
  uint32_t targets[8];
  for (auto &t : targets) t = r.below(1u << 22);
  size_t i = 0;
  while (i < n) {
    size_t run = 16 + r.below(49);
    for (size_t k = 0; k < run && i < n; ++k) {
      uint32_t v = r.below(64);
      out[i++] = v < 48 ? ops[v & 15] : (uint8_t)r.below(256);
    }
    if (i + 5 <= n) {
      uint32_t rel = targets[r.below(8)] - (uint32_t)(i + 5);      // rel32 = target - next ip
      out[i] = r.below(2) ? 0xE8 : 0xE9;
      out[i + 1] = (uint8_t)rel; out[i + 2] = (uint8_t)(rel >> 8); out[i + 3] = (uint8_t)(rel >> 16);
      out[i + 4] = (rel >> 24) & 0x80 ? 0xFF : 0x00;
      i += 5;
    }
  }
}

void e8e9_forward(uint8_t *buf, size_t n) {            // LibZPAQ.cs:372-384
  for (long i = (long)n - 5; i >= 0; --i)
    if ((buf[i] & 254) == 0xe8 && ((buf[i + 4] + 1) & 254) == 0) {
      unsigned a = (buf[i + 1] | buf[i + 2] << 8 | buf[i + 3] << 16) + (unsigned)i;
      buf[i + 1] = (uint8_t)a; buf[i + 2] = (uint8_t)(a >> 8); buf[i + 3] = (uint8_t)(a >> 16);
    }
}

// ---------------------------------------------------------------------------
// Pre-processors in the reference's code formats (LZBuffer.cs:96-115), for benchmark streams of its methods "1".."3":
// only the FORMAT has to agree with the reference (the post-processors invert it); the parse is a plain greedy one
// through a hash of the last position of every 4 (level 1) / m (level 2) bytes.
// ---------------------------------------------------------------------------
static int lg32(uint32_t x) { int r = 0; while (x) { ++r; x >>= 1; } return r; }   // floor(log2 x) + 1

struct BitW {
  std::vector<uint8_t> &v; uint64_t acc = 0; int n = 0;
  explicit BitW(std::vector<uint8_t> &o) : v(o) {}
  void put(uint32_t x, int bits) {                     // LSB first
    if (!bits) return;
    acc |= (uint64_t)(x & (bits == 32 ? 0xFFFFFFFFu : ((1u << bits) - 1))) << n; n += bits;
    while (n >= 8) { v.push_back((uint8_t)acc); acc >>= 8; n -= 8; }
  }
  void flush() { if (n) { v.push_back((uint8_t)acc); acc = 0; n = 0; } }
};

template <class Lit, class Match>
static void greedy_parse(const uint8_t *d, size_t n, uint32_t min_match, uint32_t max_match, uint32_t max_off, Lit lit, Match match) {
  const uint32_t hb = 20;
  std::vector<uint32_t> last(1u << hb, 0xFFFFFFFFu);
  auto hash = [&](size_t i) { uint32_t v; memcpy(&v, d + i, 4); return (v * 2654435761u) >> (32 - hb); };
  size_t i = 0, lit0 = 0;
  const uint32_t key = min_match < 4 ? 4 : min_match;
  while (i + key <= n) {
    const uint32_t h = hash(i);
    const uint32_t j = last[h];
    last[h] = (uint32_t)i;
    uint32_t m = 0;
    if (j != 0xFFFFFFFFu && i - j <= max_off) {
      while (m < max_match && i + m < n && d[j + m] == d[i + m]) ++m;
    }
    if (m >= min_match) {
      if (i > lit0) lit(lit0, i);
      match(m, (uint32_t)(i - j));
      for (size_t k = i + 1; k < i + m && k + 4 <= n; k += 2) last[hash(k)] = (uint32_t)k;
      i += m; lit0 = i;
    } else ++i;
  }
  if (n > lit0) lit(lit0, n);
}

// level 1 (lazy2's input): 00,n,L[n] literals; mm,mmm,n,ll,r,q matches (LZBuffer.cs:96-107)
static void pre_lz1(const uint8_t *d, size_t n, const int *args, std::vector<uint8_t> &out) {
  const int rb = args[0] > 4 ? args[0] - 4 : 0;
  const uint32_t min_match = args[2] < 4 ? 4 : (uint32_t)args[2];
  BitW w(out);
  greedy_parse(d, n, min_match, 1u << 16, (1u << 23) - 1,
    [&](size_t a, size_t b) {
      const uint32_t litn = (uint32_t)(b - a);
      int ll = lg32(litn) - 1;
      w.put(0, 2);
      while (ll > 0) { --ll; w.put(1, 1); w.put((litn >> ll) & 1, 1); }
      w.put(0, 1);
      for (size_t k = a; k < b; ++k) w.put(d[k], 8);
    },
    [&](uint32_t ln, uint32_t off) {
      int ll = lg32(ln) - 1;
      off += (1u << rb) - 1;
      const int lo = lg32(off) - 1 - rb;
      w.put((uint32_t)((lo + 8) >> 3), 2);
      w.put((uint32_t)(lo & 7), 3);
      while (ll > 2) { --ll; w.put(1, 1); w.put((ln >> ll) & 1, 1); }
      w.put(0, 1);
      w.put(ln & 3, 2);
      w.put(off, rb);
      w.put(off >> rb, lo);
    });
  w.flush();
}

// level 2 (lzpre's input): 00xxxxxx x+1 literals; yyxxxxxx y+1 offset bytes, length x+m (LZBuffer.cs:109-112)
static void pre_lz2(const uint8_t *d, size_t n, const int *args, std::vector<uint8_t> &out) {
  const uint32_t m = (uint32_t)args[2];
  greedy_parse(d, n, m < 3 ? 3 : m, m + 63 + 4 * 64, (1u << 24) - 1,
    [&](size_t a, size_t b) {
      while (a < b) {
        const size_t k = b - a < 64 ? b - a : 64;
        out.push_back((uint8_t)(k - 1));
        out.insert(out.end(), d + a, d + a + k);
        a += k;
      }
    },
    [&](uint32_t ln, uint32_t off) {
      off -= 1;
      while (ln > 0) {
        const uint32_t len1 = ln > m * 2 + 63 ? m + 63 : ln > m + 63 ? ln - m : ln;
        if (off < (1u << 16)) { out.push_back((uint8_t)(64 + len1 - m)); out.push_back((uint8_t)(off >> 8)); out.push_back((uint8_t)off); }
        else { out.push_back((uint8_t)(128 + len1 - m)); out.push_back((uint8_t)(off >> 16)); out.push_back((uint8_t)(off >> 8)); out.push_back((uint8_t)off); }
        ln -= len1;
      }
    });
}

// level 3 (bwtrle's input): BWT, end of string coded as 255, its position in the last 4 bytes (LZBuffer.cs:113-115, :228-240)
static void pre_bwt(const uint8_t *d, size_t n, std::vector<uint8_t> &out) {
  if (n == 0) { out.assign({255, 0, 0, 0, 0}); return; }
  std::vector<uint32_t> sa(n);
  // bucket by the first two bytes, then sort every bucket by plain suffix comparison (the end of the block is smaller than
  // any byte): synthetic text repeats itself over tens of bytes at most
  std::vector<uint32_t> cnt(65537, 0);
  auto key2 = [&](size_t i) -> uint32_t { return (uint32_t)d[i] << 8 | (i + 1 < n ? d[i + 1] : 0u); };
  for (size_t i = 0; i < n; ++i) ++cnt[key2(i) + 1];
  for (size_t k = 1; k <= 65536; ++k) cnt[k] += cnt[k - 1];
  { std::vector<uint32_t> pos(cnt.begin(), cnt.end() - 1); for (size_t i = 0; i < n; ++i) sa[pos[key2(i)]++] = (uint32_t)i; }
  auto less = [&](uint32_t a, uint32_t b) {
    if (a == b) return false;
    const size_t la = n - a, lb = n - b, l = la < lb ? la : lb;
    const int c = memcmp(d + a, d + b, l);
    return c ? c < 0 : la < lb;
  };
  for (size_t k = 0; k < 65536; ++k)
    if (cnt[k + 1] - cnt[k] > 1) std::sort(sa.begin() + cnt[k], sa.begin() + cnt[k + 1], less);
  out.reserve(n + 5);
  out.push_back(d[n - 1]);
  uint32_t idx = 0;
  for (size_t i = 1; i <= n; ++i) {
    const uint32_t s0 = sa[i - 1];
    if (s0 == 0) { idx = (uint32_t)i; out.push_back(255); } else out.push_back(d[s0 - 1]);
  }
  for (int k = 0; k < 4; ++k) out.push_back((uint8_t)(idx >> (8 * k)));
}

static void preprocess(const uint8_t *d, size_t n, const int *args, std::vector<uint8_t> &out) {
  out.clear();
  const int level = args[1] & 3;
  if (level == 1) pre_lz1(d, n, args, out);
  else if (level == 2) pre_lz2(d, n, args, out);
  else if (level == 3) pre_bwt(d, n, out);
  else out.assign(d, d + n);
}

struct Stream {
  std::vector<uint8_t> bytes;
  std::vector<uint64_t> offsets;                       // nblocks + 1
  std::string error;
};

}  // namespace

extern "C" {

int zpaqgen_version(void) { return 1; }

void zpaqgen_plain(int kind, uint64_t block_index, uint8_t *out, size_t n) { gen_plain(kind, block_index, out, n); }

void zpaqgen_e8e9(uint8_t *buf, size_t n) { e8e9_forward(buf, n); }

void zpaqgen_sha1(const uint8_t *p, size_t n, uint8_t out[20]) { zh::sha1(p, n, out); }

// Compress `data` as one block.  `sha_src` (may equal data) is what the stored
// SHA-1 and the size comment describe — the ORIGINAL bytes when `data` has been
// pre-transformed for a PCOMP.  flags: bit0 store SHA-1, bit1 write the 13-byte tag.
// Returns bytes written, or a negative status; -20 with *need set if cap is small.
long zpaqgen_compress_block(const uint8_t *hdr, size_t hdrlen, const uint8_t *pcomp, size_t plen, const uint8_t *data,
                            size_t n, const uint8_t *sha_src, size_t sha_n, const char *filename, const char *comment,
                            int flags, uint8_t *out, size_t cap, size_t *need) {
  if (!zh::host_tables_ok()) return ZPAQHIP_E_NO_DEVICE;
  BlockWriter w;
  int rc = w.setup(hdr, hdrlen);
  if (rc) return rc;
  Out o;
  o.v.reserve(n / 2 + 4096);
  uint8_t sha[20];
  if (flags & 1) zh::sha1(sha_src ? sha_src : data, sha_src ? sha_n : n, sha);
  char cbuf[32];
  if (!comment) { snprintf(cbuf, sizeof cbuf, "%zu", sha_src ? sha_n : n); comment = cbuf; }
  rc = w.write_block(o, pcomp, plen, data, n, filename, comment, (flags & 1) ? sha : nullptr, (flags & 2) != 0);
  if (rc) return rc;
  if (need) *need = o.v.size();
  if (o.v.size() > cap) return ZPAQHIP_E_OUTPUT_FULL;
  memcpy(out, o.v.data(), o.v.size());
  return (long)o.v.size();
}

// Whole synthetic stream: blocks first_block .. first_block+nblocks-1, each
// `block_size` bytes of plaintext of the given kind, compressed independently on
// `threads` host threads.  e8e9 != 0 applies the forward transform before coding.
void *zpaqgen_stream_new(const uint8_t *hdr, size_t hdrlen, const uint8_t *pcomp, size_t plen, int kind, int e8e9,
                         uint64_t first_block, uint32_t nblocks, size_t block_size, int threads) {
  Stream *s = new Stream();
  if (!zh::host_tables_ok()) { s->error = "table pins failed"; return s; }
  std::vector<std::vector<uint8_t>> parts(nblocks);
  std::atomic<uint32_t> next{0};
  std::atomic<int> failed{0};
  if (threads < 1) threads = 1;
  std::vector<std::thread> th;
  for (int t = 0; t < threads; ++t)
    th.emplace_back([&] {
      BlockWriter w;
      if (w.setup(hdr, hdrlen)) { failed = 1; return; }
      std::vector<uint8_t> plain(block_size), enc;
      for (;;) {
        uint32_t b = next.fetch_add(1);
        if (b >= nblocks || failed) break;
        gen_plain(kind, first_block + b, plain.data(), block_size);
        uint8_t sha[20];
        zh::sha1(plain.data(), block_size, sha);
        const uint8_t *src = plain.data();
        if (e8e9) { enc = plain; e8e9_forward(enc.data(), enc.size()); src = enc.data(); }
        Out o;
        o.v.reserve(block_size / 2 + 4096);
        char comment[32];
        snprintf(comment, sizeof comment, "%zu", block_size);
        if (w.write_block(o, pcomp, plen, src, block_size, "", comment, sha, true)) { failed = 1; break; }
        parts[b].swap(o.v);
      }
    });
  for (auto &t : th) t.join();
  if (failed) { s->error = "stream generation failed"; return s; }
  size_t total = 0;
  for (auto &p : parts) total += p.size();
  s->bytes.reserve(total);
  for (auto &p : parts) {
    s->offsets.push_back(s->bytes.size());
    s->bytes.insert(s->bytes.end(), p.begin(), p.end());
    std::vector<uint8_t>().swap(p);
  }
  s->offsets.push_back(s->bytes.size());
  return s;
}

// What the pre-processor of a method makes of `data` (args[0..8] = the method's numbers, args[1] & 3 = level; the E8E9
// variants expect the caller to have applied the forward transform).  Returns the size, or -20 with *need set.
long zpaqgen_preprocess(const int *args, const uint8_t *data, size_t n, uint8_t *out, size_t cap, size_t *need) {
  std::vector<uint8_t> v;
  preprocess(data, n, args, v);
  if (need) *need = v.size();
  if (v.size() > cap) return ZPAQHIP_E_OUTPUT_FULL;
  if (!v.empty()) memcpy(out, v.data(), v.size());
  return (long)v.size();
}

// Synthetic stream of a METHOD (LibZPAQ.compressBlock's framing): every block's plaintext goes through the method's
// pre-processor and is then coded with the model of `hdr` — or, for n = 0, stored in length-prefixed chunks
// (Encoder.cs:39-73) behind the selector and the PCOMP program.
void *zpaqgen_method_stream_new(const uint8_t *hdr, size_t hdrlen, const uint8_t *pcomp, size_t plen, const int *args, int kind,
                                uint64_t first_block, uint32_t nblocks, size_t block_size, int threads) {
  Stream *s = new Stream();
  if (!zh::host_tables_ok()) { s->error = "table pins failed"; return s; }
  const bool stored = hdrlen > 6 && hdr[6] == 0;
  const bool doe8 = args[1] >= 4 && args[1] <= 7;
  std::vector<std::vector<uint8_t>> parts(nblocks);
  std::atomic<uint32_t> next{0};
  std::atomic<int> failed{0};
  if (threads < 1) threads = 1;
  std::vector<std::thread> th;
  for (int t = 0; t < threads; ++t)
    th.emplace_back([&] {
      BlockWriter w;
      if (!stored && w.setup(hdr, hdrlen)) { failed = 1; return; }
      std::vector<uint8_t> plain(block_size), enc, pre;
      for (;;) {
        uint32_t b = next.fetch_add(1);
        if (b >= nblocks || failed) break;
        gen_plain(kind, first_block + b, plain.data(), block_size);
        uint8_t sha[20];
        zh::sha1(plain.data(), block_size, sha);
        const uint8_t *src = plain.data();
        if (doe8) { enc = plain; e8e9_forward(enc.data(), enc.size()); src = enc.data(); }
        preprocess(src, block_size, args, pre);
        char comment[32];
        snprintf(comment, sizeof comment, "%zu", block_size);
        Out o;
        if (!stored) {
          o.v.reserve(pre.size() / 2 + 4096);
          if (w.write_block(o, pcomp, plen, pre.data(), pre.size(), "", comment, sha, true)) { failed = 1; break; }
        } else {
          static const uint8_t kTag[13] = {0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3};
          o.v.reserve(pre.size() + plen + 4096);
          o.put(kTag, 13);
          o.put('z'); o.put('P'); o.put('Q'); o.put(2); o.put(1);      // n = 0 -> level 2 (Compressor.cs:92-96)
          o.put(hdr, hdrlen);
          o.put(1); o.put(0);
          o.put(comment, strlen(comment));
          o.put(0); o.put(0);
          std::vector<uint8_t> dec;                                      // the decoded stream: selector [+ program] + data
          if (pcomp && plen) { dec.push_back(1); dec.push_back((uint8_t)(plen & 255)); dec.push_back((uint8_t)(plen >> 8)); dec.insert(dec.end(), pcomp, pcomp + plen); }
          else dec.push_back(0);
          dec.insert(dec.end(), pre.begin(), pre.end());
          for (size_t i = 0; i < dec.size(); i += 65536) {
            const size_t k = dec.size() - i < 65536 ? dec.size() - i : 65536;
            o.put((int)(k >> 24 & 255)); o.put((int)(k >> 16 & 255)); o.put((int)(k >> 8 & 255)); o.put((int)(k & 255));
            o.put(dec.data() + i, k);
          }
          o.put(0); o.put(0); o.put(0); o.put(0);
          o.put(253); o.put(sha, 20);
          o.put(255);
        }
        parts[b].swap(o.v);
      }
    });
  for (auto &t : th) t.join();
  if (failed) { s->error = "stream generation failed"; return s; }
  size_t total = 0;
  for (auto &p : parts) total += p.size();
  s->bytes.reserve(total);
  for (auto &p : parts) {
    s->offsets.push_back(s->bytes.size());
    s->bytes.insert(s->bytes.end(), p.begin(), p.end());
    std::vector<uint8_t>().swap(p);
  }
  s->offsets.push_back(s->bytes.size());
  return s;
}

const char *zpaqgen_stream_error(void *h) { return ((Stream *)h)->error.c_str(); }
size_t zpaqgen_stream_size(void *h) { return ((Stream *)h)->bytes.size(); }
void zpaqgen_stream_copy(void *h, uint8_t *out, uint64_t *offsets) {
  Stream *s = (Stream *)h;
  if (out) memcpy(out, s->bytes.data(), s->bytes.size());
  if (offsets) memcpy(offsets, s->offsets.data(), s->offsets.size() * 8);
}
void zpaqgen_stream_free(void *h) { delete (Stream *)h; }

}  // extern "C"
