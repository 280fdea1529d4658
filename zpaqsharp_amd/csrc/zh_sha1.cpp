// zh_sha1.cpp — SHA-1 (FIPS 180-4) for the optional per-segment checksum check.
// The reference defers to System.Security.Cryptography.SHA1 (ZPAQL.cs:187,
// Decompresser.cs:115-118); the stored digest is of the segment's plaintext.
#include <string.h>

#include "zh_host.h"

namespace zh {
namespace {
inline uint32_t rotl(uint32_t x, int n) { return x << n | x >> (32 - n); }

void compress(uint32_t st[5], const uint8_t *blk) {
  uint32_t w[80];
  for (int t = 0; t < 16; ++t)
    w[t] = (uint32_t)blk[4 * t] << 24 | (uint32_t)blk[4 * t + 1] << 16 | (uint32_t)blk[4 * t + 2] << 8 | blk[4 * t + 3];
  for (int t = 16; t < 80; ++t) w[t] = rotl(w[t - 3] ^ w[t - 8] ^ w[t - 14] ^ w[t - 16], 1);
  uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4];
  for (int t = 0; t < 80; ++t) {
    uint32_t f, k;
    switch (t / 20) {
      case 0: f = d ^ (b & (c ^ d)); k = 0x5A827999u; break;
      case 1: f = b ^ c ^ d; k = 0x6ED9EBA1u; break;
      case 2: f = (b & c) | (d & (b | c)); k = 0x8F1BBCDCu; break;
      default: f = b ^ c ^ d; k = 0xCA62C1D6u; break;
    }
    uint32_t tmp = rotl(a, 5) + f + e + k + w[t];
    e = d; d = c; c = rotl(b, 30); b = a; a = tmp;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e;
}
}  // namespace

void sha1(const uint8_t *p, size_t n, uint8_t out[20]) {
  uint32_t st[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
  size_t full = n / 64;
  for (size_t i = 0; i < full; ++i) compress(st, p + 64 * i);
  uint8_t tail[128];
  size_t rem = n - 64 * full;
  memset(tail, 0, sizeof tail);
  if (rem) memcpy(tail, p + 64 * full, rem);
  tail[rem] = 0x80;
  size_t tl = rem + 9 <= 64 ? 64 : 128;
  uint64_t bits = (uint64_t)n * 8;
  for (int i = 0; i < 8; ++i) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
  compress(st, tail);
  if (tl == 128) compress(st, tail + 64);
  for (int i = 0; i < 5; ++i) {
    out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16);
    out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i];
  }
}
}  // namespace zh
