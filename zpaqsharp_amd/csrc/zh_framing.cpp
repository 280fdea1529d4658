// zh_framing.cpp — host-side stream framing and header parsing.
//
// scan_stream() walks the archive grammar exactly as the reference's
// Decompresser state machine does, but without decoding:
//   findBlock        Decompresser.cs:29-58   (16-byte rolling-hash locator)
//   ZPAQL.read       ZPAQL.cs:112-156        (block header)
//   findFilename     Decompresser.cs:67-93
//   readComment      Decompresser.cs:96-108
//   Decoder.skip     Decoder.cs:70-98        (end of coded data = run of >=4 zero bytes)
//   readSegmentEnd   Decompresser.cs:163-194
// Its output is the block/segment table the GPU decode works from.
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "zh_host.h"
#include "zh_zpaql_native.h"
#include "zh_chain_spec.h"

namespace zh {

const char *status_message(int code) {
  switch (code) {
    case ZPAQHIP_OK: return "ok";
    case ZPAQHIP_E_CORRUPT: return "archive corrupted";
    case ZPAQHIP_E_EOF: return "unexpected end of file";
    case ZPAQHIP_E_EOS: return "decoding end of stream";
    case ZPAQHIP_E_ZPAQL: return "ZPAQL execution error";
    case ZPAQHIP_E_HEADER: return "invalid block header";
    case ZPAQHIP_E_HM_TOO_BIG: return "H too big";
    case ZPAQHIP_E_COMPONENT: return "invalid component";
    case ZPAQHIP_E_PP_EOS: return "Unexpected EOS";
    case ZPAQHIP_E_PP_TYPE: return "unknown post processing type";
    case ZPAQHIP_E_PP_EMPTY: return "Empty PCOMP";
    case ZPAQHIP_E_LEVEL: return "unsupported ZPAQ level";
    case ZPAQHIP_E_SEGMENT: return "missing segment or end of block";
    case ZPAQHIP_E_FRAMING_EOF: return "unexpected EOF";
    case ZPAQHIP_E_RESERVED: return "missing reserved byte";
    case ZPAQHIP_E_SEGEND: return "missing end of segment marker";
    case ZPAQHIP_E_OUTPUT_FULL: return "output buffer too small";
    case ZPAQHIP_E_SHA1: return "SHA-1 checksum mismatch";
    case ZPAQHIP_E_NO_DEVICE: return "no usable HIP device";
    case ZPAQHIP_E_DEVICE_MEM: return "Out of memory";
    case ZPAQHIP_E_HIP: return "HIP runtime error";
    case ZPAQHIP_E_ARG: return "bad argument";
    case ZPAQHIP_E_BUDGET: return "ZPAQL instruction budget exhausted";
    case ZPAQHIP_E_CALLBACK: return "read/write callback failed";
    case ZH_E_SKIPPED: return "segment skipped after an earlier error in its block";
    default: return "unknown status";
  }
}

static const int kCompSize[10] = {0, 2, 3, 2, 3, 4, 6, 6, 3, 5};   // Component.cs:27-43


void set_err(zpaqhip_err *err, int code, int block, int seg, const char *msg) {
  if (!err) return;
  err->code = code; err->block = block; err->segment = seg;
  snprintf(err->msg, sizeof err->msg, "%s", msg ? msg : status_message(code));
}

namespace {

struct Cur {
  const uint8_t *p; size_t n, pos;
  bool eof_hit = false;                   // a get() ran into the end of the buffer
  int get() { if (pos < n) return p[pos++]; eof_hit = true; return -1; }
};

// ZPAQL.memory(), ZPAQL.cs:58-81 (header_len = hsize + 300 as allocated by ZPAQL.read).
double model_memory(const uint8_t *hd, size_t hsize) {
  auto pow2 = [](int x) { double r = 1; for (; x > 0; --x) r += r; return r; };
  double mem = pow2(hd[2] + 2) + pow2(hd[3]) + pow2(hd[4] + 2) + pow2(hd[5]) + (double)(hsize + 300);
  size_t cp = 7;
  for (int i = 0; i < hd[6]; ++i) {
    double size = pow2(hd[cp + 1]);
    switch (hd[cp]) {
      case ZH_CM: mem += 4 * size; break;
      case ZH_ICM: mem += 64 * size + 1024; break;
      case ZH_MATCH: mem += 4 * size + pow2(hd[cp + 2]); break;
      case ZH_MIX2: mem += 2 * size; break;
      case ZH_MIX: mem += 4 * size * hd[cp + 3]; break;
      case ZH_ISSE: mem += 64 * size + 2048; break;
      case ZH_SSE: mem += 128 * size; break;
    }
    cp += kCompSize[hd[cp]];
  }
  return mem;
}

// Structural checks of ZPAQL.read on header bytes at in[off..].  Returns the
// header length (hsize+2) or a negative status.
long check_header(Cur &c, const char **msg) {
  size_t start = c.pos;
  int lo = c.get(), hi = c.get();
  if (lo < 0 || hi < 0) { *msg = "unexpected end of file"; return ZPAQHIP_E_EOF; }
  size_t hsize = (size_t)lo + 256 * (size_t)hi;
  size_t cend = 2;
  uint8_t fixed[5];
  for (int i = 0; i < 5; ++i) {
    int v = c.get();
    if (v < 0) { *msg = "unexpected end of file"; return ZPAQHIP_E_EOF; }
    fixed[i] = (uint8_t)v; ++cend;
  }
  int n = fixed[4];
  for (int i = 0; i < n; ++i) {
    int type = c.get();
    if (type < 0) { *msg = "unexpected end of file"; return ZPAQHIP_E_EOF; }
    ++cend;
    int size = type < 10 ? kCompSize[type] : 0;
    if (size < 1) { *msg = "Invalid component type"; return ZPAQHIP_E_HEADER; }
    if (cend + size > hsize) { *msg = "COMP overflows header"; return ZPAQHIP_E_HEADER; }
    for (int j = 1; j < size; ++j) {
      if (c.get() < 0) { *msg = "unexpected end of file"; return ZPAQHIP_E_EOF; }
      ++cend;
    }
  }
  int e = c.get();
  if (e < 0) { *msg = "unexpected end of file"; return ZPAQHIP_E_EOF; }
  ++cend;
  if (e != 0) { *msg = "missing COMP END"; return ZPAQHIP_E_HEADER; }
  // hbegin = cend+128; must not exceed hsize+129  (ZPAQL.cs:137-138)
  if (cend + 128 > hsize + 129) { *msg = "missing HCOMP"; return ZPAQHIP_E_HEADER; }
  size_t hlen = hsize + 129 - (cend + 128);          // HCOMP bytes before the END byte
  for (size_t i = 0; i < hlen; ++i)
    if (c.get() < 0) { *msg = "unexpected end of file"; return ZPAQHIP_E_EOF; }
  e = c.get();
  if (e < 0) { *msg = "unexpected end of file"; return ZPAQHIP_E_EOF; }
  if (e != 0) { *msg = "missing HCOMP END"; return ZPAQHIP_E_HEADER; }
  return (long)(c.pos - start);
}

uint64_t parse_size(const uint8_t *p, size_t n) {
  if (n == 0 || p[0] < '0' || p[0] > '9') return UINT64_MAX;
  uint64_t v = 0;
  for (size_t i = 0; i < n && p[i] >= '0' && p[i] <= '9'; ++i) {
    if (v > (UINT64_MAX - 9) / 10) return UINT64_MAX;
    v = v * 10 + (p[i] - '0');
  }
  return v;
}

}  // namespace

// On an error the blocks (and only their segments) parsed before the damaged one are kept in `out`, so that a
// caller can deliver them first, as the reference's Decompresser would (it only fails on reaching the damage).
// out.resume_off = where an incremental caller continues once more bytes have arrived; out.hit_eof = the failing
// block ran into the end of the buffer (more input may complete it).
static int scan_blocks(Cur &c, const uint8_t *in, ScanOut &out, zpaqhip_err *err, const ScanLimit &lim);

int scan_stream(const uint8_t *in, size_t n, ScanOut &out, zpaqhip_err *err, const ScanLimit &lim) {
  Cur c{in, n, 0};
  out.blocks.clear();
  out.segs.clear();
  out.resume_off = 0;
  out.hit_eof = false;
  out.stopped = false;
  const int rc = scan_blocks(c, in, out, err, lim);
  if (out.stopped) { out.resume_off = (size_t)out.blocks.back().end_off; return ZPAQHIP_OK; }
  const size_t good_end = out.blocks.empty() ? 0 : (size_t)out.blocks.back().end_off;
  if (rc) {
    if (!out.blocks.empty()) out.segs.resize(out.blocks.back().first_seg + out.blocks.back().n_seg);
    else out.segs.clear();
    out.hit_eof = c.eof_hit;
    out.resume_off = good_end;           // the damaged / unfinished block is scanned again from its tag
  } else {
    // no further tag: only the last 15 bytes can still be the beginning of one
    out.resume_off = std::max(good_end, n > 15 ? n - 15 : (size_t)0);
  }
  return rc;
}

static int scan_blocks(Cur &c, const uint8_t *in, ScanOut &out, zpaqhip_err *err, const ScanLimit &lim) {
  bool single_only = true;                               // every block so far has one component (the single-CM kernel's two-per-CU form)
  for (;;) {
    if (!out.blocks.empty() && out.blocks.back().n_comp != 1) single_only = false;
    const size_t min_blocks = (lim.min_blocks_other && !single_only) ? lim.min_blocks_other : lim.min_blocks;
    if (!out.blocks.empty() &&
        (out.blocks.size() >= lim.max_blocks || (out.blocks.size() >= min_blocks && out.blocks.back().end_off >= lim.min_bytes))) {
      out.stopped = true;                                // a batch is complete: the caller continues from resume_off
      return ZPAQHIP_OK;
    }
    // ---- findBlock: Decompresser.cs:34-45
    uint32_t h1 = 0x3D49B113, h2 = 0x29EB7F93, h3 = 0x2614BE13, h4 = 0x3828EB13;
    int ch;
    while ((ch = c.get()) != -1) {
      h1 = h1 * 12 + (uint32_t)ch; h2 = h2 * 20 + (uint32_t)ch;
      h3 = h3 * 28 + (uint32_t)ch; h4 = h4 * 44 + (uint32_t)ch;
      if (h1 == 0xB16B88F1 && h2 == 0xFF5376F1 && h3 == 0x72AC5BF1 && h4 == 0x2F909AF1) break;
    }
    if (ch == -1) return ZPAQHIP_OK;
    const int bi = (int)out.blocks.size();
    zpaqhip_block b;
    memset(&b, 0, sizeof b);
    b.tag_off = c.pos >= 16 ? c.pos - 16 : 0;
    int level = c.get();
    if (level != 1 && level != 2) { set_err(err, ZPAQHIP_E_LEVEL, bi, -1, "unsupported ZPAQ level"); return ZPAQHIP_E_LEVEL; }
    if (c.get() != 1) { set_err(err, ZPAQHIP_E_LEVEL, bi, -1, "unsupported ZPAQL type"); return ZPAQHIP_E_LEVEL; }
    b.level = (uint8_t)level;
    b.hdr_off = c.pos;
    const char *msg = nullptr;
    long hl = check_header(c, &msg);
    if (hl < 0) { set_err(err, (int)hl, bi, -1, msg); return (int)hl; }
    b.hdr_len = (uint32_t)hl;
    const uint8_t *hd = in + b.hdr_off;
    b.hh = hd[2]; b.hm = hd[3]; b.ph = hd[4]; b.pm = hd[5]; b.n_comp = hd[6];
    if (level == 1 && b.n_comp == 0) {
      set_err(err, ZPAQHIP_E_LEVEL, bi, -1, "ZPAQ level 1 requires at least 1 component");
      return ZPAQHIP_E_LEVEL;
    }
    b.model_mem = model_memory(hd, (size_t)hl - 2);
    b.first_seg = (uint32_t)out.segs.size();
    b.usize_hint = 0;
    // ---- segments
    for (;;) {
      int t = c.get();                                   // findFilename, Decompresser.cs:67-93
      if (t == 255) break;
      if (t != 1) { set_err(err, ZPAQHIP_E_SEGMENT, bi, (int)out.segs.size(), "missing segment or end of block"); return ZPAQHIP_E_SEGMENT; }
      const int si = (int)out.segs.size();
      zpaqhip_segment s;
      memset(&s, 0, sizeof s);
      s.block = (uint32_t)bi;
      s.name_off = c.pos;
      for (;;) {
        int v = c.get();
        if (v == -1) { set_err(err, ZPAQHIP_E_FRAMING_EOF, bi, si, "unexpected EOF"); return ZPAQHIP_E_FRAMING_EOF; }
        if (v == 0) break;
      }
      s.name_len = (uint32_t)(c.pos - 1 - s.name_off);
      s.comment_off = c.pos;                             // readComment, Decompresser.cs:96-108
      for (;;) {
        int v = c.get();
        if (v == -1) { set_err(err, ZPAQHIP_E_FRAMING_EOF, bi, si, "unexpected EOF"); return ZPAQHIP_E_FRAMING_EOF; }
        if (v == 0) break;
      }
      s.comment_len = (uint32_t)(c.pos - 1 - s.comment_off);
      if (c.get() != 0) { set_err(err, ZPAQHIP_E_RESERVED, bi, si, "missing reserved byte"); return ZPAQHIP_E_RESERVED; }
      s.usize_hint = parse_size(in + s.comment_off, s.comment_len);
      s.data_off = c.pos;
      int nx;                                            // byte after the coded data
      if (b.n_comp) {
        // Decoder.skip, modelled branch (Decoder.cs:73-81): the coded data ends
        // with a run of at least four zero bytes; the whole run belongs to it.
        uint32_t curr = 0;
        int v = -1;
        while (curr == 0) {
          v = c.get();
          if (v < 0) break;
          curr = (uint32_t)v;
        }
        if (v >= 0) {
          // `while (curr && (v = get()) >= 0) curr = curr << 8 | v`: stops behind the first four consecutive zero
          // bytes after the non-zero byte just read.  Searched with memchr (coded data is ~1/256 zero bytes).
          const uint8_t *q = c.p + c.pos, *end = c.p + c.n;
          const uint8_t *hit = nullptr;
          while (q < end) {
            const uint8_t *z = (const uint8_t *)memchr(q, 0, (size_t)(end - q));
            if (!z) break;
            if (end - z >= 4 && z[1] == 0 && z[2] == 0 && z[3] == 0) { hit = z; break; }
            q = z + 1;
            while (q < end && *q == 0) ++q;               // a run shorter than four: skip it whole
            if (q - z >= 4) { hit = z; break; }          // (cannot happen: kept for clarity of the invariant)
          }
          if (hit) c.pos = (size_t)(hit - c.p) + 4;
          else { c.pos = c.n; c.eof_hit = true; }
        }
        while ((nx = c.get()) == 0) {}
      } else {
        // unmodelled store path (Decoder.cs:84-96): [len32 big-endian, bytes]* , len 0 ends
        uint32_t curr = 0;
        int v = 0;
        for (int i = 0; i < 4 && (v = c.get()) >= 0; ++i) curr = curr << 8 | (uint32_t)v;
        while (curr > 0 && v >= 0) {
          if (c.n - c.pos < curr) {                    // the chunk runs past what has been read: more input may repair it
            c.eof_hit = true;
            set_err(err, ZPAQHIP_E_EOF, bi, si, "skipped to EOF");
            return ZPAQHIP_E_EOF;
          }
          c.pos += curr;
          curr = 0;
          for (int i = 0; i < 4 && (v = c.get()) >= 0; ++i) curr = curr << 8 | (uint32_t)v;
        }
        nx = v >= 0 ? c.get() : -1;
      }
      s.data_len = (nx >= 0 ? c.pos - 1 : c.pos) - s.data_off;
      // readSegmentEnd, Decompresser.cs:177-193
      if (nx == 254) s.flags = 0;
      else if (nx == 253) {
        s.flags = 1;
        for (int i = 0; i < 20; ++i) {
          int v = c.get();
          s.sha1[i] = (uint8_t)v;            // the reference stores get() even at EOF
        }
      } else { set_err(err, ZPAQHIP_E_SEGEND, bi, si, "missing end of segment marker"); return ZPAQHIP_E_SEGEND; }
      if (b.usize_hint != UINT64_MAX) {
        if (s.usize_hint == UINT64_MAX) b.usize_hint = UINT64_MAX;
        else b.usize_hint += s.usize_hint;
      }
      out.segs.push_back(s);
    }
    b.n_seg = (uint32_t)out.segs.size() - b.first_seg;
    b.end_off = c.pos;
    out.blocks.push_back(b);
  }
}

static uint64_t up256(uint64_t x) { return (x + 255) & ~255ull; }

int build_model(const uint8_t *hdr, size_t len, ZhModel &m, std::vector<uint8_t> &code, zpaqhip_err *err) {
  memset(&m, 0, sizeof m);
  if (len < 9 || (size_t)hdr[0] + 256 * (size_t)hdr[1] + 2 != len) {
    set_err(err, ZPAQHIP_E_HEADER, -1, -1, "COMP overflows header");
    return ZPAQHIP_E_HEADER;
  }
  m.hh = hdr[2]; m.hm = hdr[3]; m.ph = hdr[4]; m.pm = hdr[5]; m.n = hdr[6];
  if (m.hh > 32 || m.ph > 32) { set_err(err, ZPAQHIP_E_HM_TOO_BIG, -1, -1, "H too big"); return ZPAQHIP_E_HM_TOO_BIG; }
  if (m.hm > 32 || m.pm > 32) { set_err(err, ZPAQHIP_E_HM_TOO_BIG, -1, -1, "M too big"); return ZPAQHIP_E_HM_TOO_BIG; }
  const uint64_t kMaxTable = 1ull << 36;
  uint64_t off = 0;
  size_t cp = 7;
  auto fail = [&](const char *msg) { set_err(err, ZPAQHIP_E_COMPONENT, -1, -1, msg); return ZPAQHIP_E_COMPONENT; };
  for (uint32_t i = 0; i < m.n; ++i) {
    if (cp >= len || hdr[cp] >= 10 || kCompSize[hdr[cp]] < 1 || cp + kCompSize[hdr[cp]] > len) {
      set_err(err, ZPAQHIP_E_HEADER, -1, -1, "Invalid component type");
      return ZPAQHIP_E_HEADER;
    }
    ZhComp &c = m.comp[i];
    c.type = hdr[cp];
    for (int k = 1; k < kCompSize[c.type]; ++k) c.arg[k - 1] = hdr[cp + k];
    const uint8_t *a = c.arg;
    uint64_t cm_elems = 0, cm_esize = 4, ht_bytes = 0, mask_elems = 0;
    switch (c.type) {                                   // Predictor.cs:94-167
      case ZH_CONS: break;
      case ZH_CM:
        if (a[0] > 32) return fail("max size for CM is 32");
        cm_elems = mask_elems = 1ull << a[0];
        break;
      case ZH_ICM:
        if (a[0] > 26) return fail("max size for ICM is 26");
        cm_elems = mask_elems = 256; ht_bytes = 64ull << a[0];
        break;
      case ZH_MATCH:
        if (a[0] > 32 || a[1] > 32) return fail("max size for MATCH is 32 32");
        cm_elems = mask_elems = 1ull << a[0]; ht_bytes = 1ull << a[1];
        break;
      case ZH_AVG:
        if (a[0] >= i) return fail("AVG j >= i");
        if (a[1] >= i) return fail("AVG k >= i");
        break;
      case ZH_MIX2:
        if (a[0] > 32) return fail("max size for MIX2 is 32");
        if (a[2] >= i) return fail("MIX2 k >= i");
        if (a[1] >= i) return fail("MIX2 j >= i");
        cm_elems = mask_elems = 1ull << a[0]; cm_esize = 2;
        break;
      case ZH_MIX:
        if (a[0] > 32) return fail("max size for MIX is 32");
        if (a[1] >= i) return fail("MIX j >= i");
        if (a[2] < 1 || a[2] > i - a[1]) return fail("MIX m not in 1..i-j");
        mask_elems = 1ull << a[0];                       // contexts; a row is m weights
        cm_elems = mask_elems * a[2];
        break;
      case ZH_ISSE:
        if (a[0] > 32) return fail("max size for ISSE is 32");
        if (a[1] >= i) return fail("ISSE j >= i");
        cm_elems = mask_elems = 512; ht_bytes = 64ull << a[0];
        break;
      case ZH_SSE:
        if (a[0] > 32) return fail("max size for SSE is 32");
        if (a[1] >= i) return fail("SSE j >= i");
        if (a[2] > a[3] * 4) return fail("SSE start > limit*4");
        cm_elems = mask_elems = 32ull << a[0];
        break;
      default:
        return fail("unknown component type");
    }
    if (cm_elems * cm_esize > kMaxTable || ht_bytes > kMaxTable || mask_elems > (1ull << 32) || ht_bytes > (1ull << 32)) {
      set_err(err, ZPAQHIP_E_DEVICE_MEM, -1, -1, "Out of memory");
      return ZPAQHIP_E_DEVICE_MEM;
    }
    c.cm_mask = mask_elems ? (uint32_t)(mask_elems - 1) : 0;
    c.ht_mask = ht_bytes ? (uint32_t)(ht_bytes - 1) : 0;
    c.cm_off = off; c.cm_bytes = (cm_elems * cm_esize + 15) & ~15ull; off = up256(off + c.cm_bytes);
    c.ht_off = off; c.ht_bytes = (ht_bytes + 15) & ~15ull; off = up256(off + c.ht_bytes);
    cp += kCompSize[c.type];
  }
  if (cp >= len || hdr[cp] != 0) { set_err(err, ZPAQHIP_E_HEADER, -1, -1, "missing COMP END"); return ZPAQHIP_E_HEADER; }
  ++cp;
  if (hdr[len - 1] != 0 || cp >= len) { set_err(err, ZPAQHIP_E_HEADER, -1, -1, "missing HCOMP END"); return ZPAQHIP_E_HEADER; }
  // everything from h_off on is zero-filled at block start
  m.h_off = off; off = up256(off + (4ull << m.hh));
  m.m_off = off; off = up256(off + (1ull << m.hm));
  m.ph_off = off; off = up256(off + (4ull << m.ph));
  m.pm_off = off; off = up256(off + (1ull << m.pm));
  m.pz_off = off; off = up256(off + ZH_PCOMP_BUF);
  m.arena_bytes = off;
  m.hcomp_len = (uint32_t)(len - cp);
  m.code_off = (uint32_t)code.size();
  code.insert(code.end(), ZH_CODE_PAD, 0);
  code.insert(code.end(), hdr + cp, hdr + len);
  code.insert(code.end(), ZH_CODE_PAD, 0);
  while (code.size() & 15) code.push_back(0);
  // ---- dependency levels and LDS placement for the lane-per-component kernel
  {
    uint32_t units = 0, nmix = 0, depth = 0;
    bool ok = m.n >= 1 && m.n <= 64;
    for (uint32_t i = 0; i < m.n; ++i) {
      ZhComp &c = m.comp[i];
      const uint8_t *a = c.arg;
      uint32_t lv = 0;
      auto in = [&](uint32_t j) { lv = std::max<uint32_t>(lv, m.comp[j].level + 1u); };
      switch (c.type) {
        case ZH_AVG: in(a[0]); in(a[1]); break;
        case ZH_MIX2: in(a[1]); in(a[2]); break;
        case ZH_MIX: for (uint32_t j = a[1]; j < (uint32_t)a[1] + a[2]; ++j) in(j); ++nmix; break;
        case ZH_ISSE: in(a[1]); break;
        case ZH_SSE: in(a[1]); break;
        default: break;
      }
      c.level = (uint8_t)lv;
      depth = std::max(depth, lv);
      c.small_unit = 0;
      if (c.type == ZH_ICM || c.type == ZH_ISSE) {
        if (units > 255) ok = false; else c.small_unit = (uint8_t)units;
        units += c.type == ZH_ICM ? 1 : 2;
      }
      if (c.type == ZH_CM && a[0] < 4) ok = false;       // a nibble's 16 entries must be distinct
    }
    m.depth = depth;
    m.kind = (ok && units <= 64 && nmix <= 4 && m.arena_bytes < (1ull << 32)) ? ZH_FAM_CHAIN : ZH_FAM_GENERIC;
    if (m.n == 0) m.kind = ZH_FAM_STORE;                  // stored bytes: the wave-wide store / LZ77 kernel (zh_store.hip)
  }
  // ---- specialisation the kernels may use (never changes results)
  if ((m.kind & 255u) == ZH_FAM_CHAIN) {
    uint32_t spec = zh_spec_lookup(hdr, len);                            // {0 none, 1 min, 2 mid, 3 max}
    const uint32_t native = zh_native_lookup(hdr + cp, m.hcomp_len);     // native HCOMP id or 0
    // The models LibZPAQ.makeConfig writes for the method strings `ci1,1,1,1,2am` (level 4) and `ci1` (BWT, level 3) have
    // the component lists of mid and min with other sizes (they follow the block-size argument) and their own HCOMP.
    // zh_nibble.hip takes table sizes from the model at run time and lets its helper wave run the program the model names
    // (bits 8-15 of kind), so these models join mid's / min's family.  Its 32-bit buffer offsets want the arena below 2 GiB.
    bool method_model = false;
    if (spec == 0 && native == ZH_NATIVE_HCOMP_M4 && m.hh == 9 && m.hm == 16 && m.n == 8 && m.arena_bytes < (1ull << 31)) {
      bool ok4 = m.comp[0].type == ZH_ICM && m.comp[6].type == ZH_MATCH && m.comp[7].type == ZH_MIX;
      for (uint32_t i = 1; ok4 && i <= 5; ++i) ok4 = m.comp[i].type == ZH_ISSE && m.comp[i].arg[1] == i - 1;
      const uint8_t *mx = m.comp[7].arg;                                 // mix N 0 7 rate 255 over components 0..6
      if (ok4 && mx[1] == 0 && mx[2] == 7 && mx[4] == 255 && mx[0] >= 8) { spec = 2; method_model = true; }
    }
    // ... and its text variant `ci1,1,1,1,2awm`: a word-model ICM as the eighth mixer input (H[8], the mixer's context, is
    // never written by that program: the kernel takes it as 0)
    bool mid8 = false;
    if (spec == 0 && native == ZH_NATIVE_HCOMP_M4W && m.hh == 9 && m.hm == 16 && m.n == 9 && m.arena_bytes < (1ull << 31)) {
      bool ok4 = m.comp[0].type == ZH_ICM && m.comp[6].type == ZH_MATCH && m.comp[7].type == ZH_ICM && m.comp[8].type == ZH_MIX;
      for (uint32_t i = 1; ok4 && i <= 5; ++i) ok4 = m.comp[i].type == ZH_ISSE && m.comp[i].arg[1] == i - 1;
      const uint8_t *mx = m.comp[8].arg;                                 // mix N 0 8 rate 255 over components 0..7
      if (ok4 && mx[1] == 0 && mx[2] == 8 && mx[4] == 255 && mx[0] >= 8) mid8 = true;
    }
    if (spec == 0 && (native == ZH_NATIVE_HCOMP_M3 || native == ZH_NATIVE_HCOMP_M2 || native == ZH_NATIVE_HCOMP_M2E) && m.hh == 9 && m.hm == 16 && m.n == 2 && m.arena_bytes < (1ull << 31) &&
        m.comp[0].type == ZH_ICM && m.comp[1].type == ZH_ISSE && m.comp[1].arg[1] == 0) {
      spec = 1; method_model = true;
    }
    // zh_chain2.hip's max code relies on what the built-in HCOMP leaves in h[17..21] (zeros, and an even h[20])
    if (spec == 3 && (native != ZH_NATIVE_HCOMP_MAX || m.hh != 5 || m.hm != 9)) spec = 0;
    // ... its min / mid code hands HCOMP to a helper wavefront that runs the translated program of that model for 16
    // candidate bytes at once, with H and M of exactly the built-in sizes
    if (spec == 1 && !method_model && (native != ZH_NATIVE_HCOMP_MIN || m.hh != 1 || m.hm != 2)) spec = 0;
    if (spec == 2 && !method_model && (native != ZH_NATIVE_HCOMP_MID || m.hh != 3 || m.hm != 3)) spec = 0;
    // ... and it keeps H in 256 LDS words and M in one or two vector registers (256 / 512 bytes)
    if (spec && !method_model && (m.hh > 8 || m.hm > (spec == 3 ? 9u : 8u))) spec = 0;
    // ... and level 4's model for barely compressible data, `...,5,0,7,..,1c0,0,511`: one ICM (the LZ77 + CM program without the
    // ISSE's context)
    const bool min1 = spec == 0 && !mid8 && (native == ZH_NATIVE_HCOMP_M2S || native == ZH_NATIVE_HCOMP_M2SE) && m.hh == 9 && m.hm == 16 &&
                      m.n == 1 && m.comp[0].type == ZH_ICM && m.arena_bytes < (1ull << 31);
    m.kind += spec;
    if (mid8) m.kind = ZH_FAM_CHAIN_MID8;
    if (min1) m.kind = ZH_FAM_CHAIN_MIN1;
    m.kind |= native << 8;
  }
  // Single direct CM whose HCOMP is "a<<= K  *d=a  halt" (D is 0 at every entry) with K >= 9: the low 9 bits of the
  // context hash are zero, which is what zh_cm.hip's compact window cache relies on.  Other single-CM models keep
  // the family chosen above.
  if (m.n == 1 && m.comp[0].type == ZH_CM && m.comp[0].arg[0] >= 9) {
    const uint8_t *hc = hdr + cp;
    // ... and its swap wave addresses the table with 32-bit buffer offsets (window * 2048, a drop sentinel at 2 GiB): a table
    // of more than 2 GiB (cm 30 and up) stays with the family chosen above
    if (m.hcomp_len == 5 && hc[0] == 207 && hc[2] == 112 && hc[3] == 56 && hc[4] == 0 && (hc[1] & 31) >= 9 &&
        m.comp[0].cm_bytes <= (1ull << 31))
      m.kind = ZH_FAM_CM1 | ZH_HK_SHIFT << 8 | (uint32_t)(hc[1] & 31) << 16;
  }
  return ZPAQHIP_OK;
}

}  // namespace zh
