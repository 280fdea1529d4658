// zh_sha1.hip — SHA-1 of decoded segments on the device (FIPS 180-4), for the verification the
// documented caller loop does per segment (Decompresser.cs:115-118, :183-191; LICENSE:305-309).
// Segments are independent: one lane per segment, each walking its plaintext in HBM 64 bytes at a
// time.  The digest comparison stays on the host (20 bytes per segment cross PCIe instead of the
// plaintext being hashed after it arrived).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ uint32_t rol(uint32_t x, int n) { return x << n | x >> (32 - n); }
__device__ __forceinline__ uint32_t be32(const uint8_t *p) {
  return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3];
}

__device__ void sha1_block(uint32_t h[5], const uint32_t win[16]) {
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = win[i];
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
#pragma unroll
  for (int t = 0; t < 80; ++t) {
    if (t >= 16) w[t & 15] = rol(w[(t + 13) & 15] ^ w[(t + 8) & 15] ^ w[(t + 2) & 15] ^ w[t & 15], 1);
    uint32_t f, k;
    if (t < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
    else if (t < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
    else if (t < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
    else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
    const uint32_t tmp = rol(a, 5) + f + e + k + w[t & 15];
    e = d; d = c; c = rol(b, 30); b = a; a = tmp;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
}

}  // namespace

// seg[2*i] = byte offset of segment i in `data`, seg[2*i+1] = its length; digest[5*i..] = H0..H4 (big-endian words).
extern "C" __global__ void zh_sha1_segments(const uint8_t *data, const uint64_t *seg, uint32_t n_seg, uint32_t *digest) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_seg) return;
  const uint8_t *p = data + seg[2 * i];
  const uint64_t len = seg[2 * i + 1];
  uint32_t h[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
  uint32_t w[16];
  uint64_t pos = 0;
  const bool aligned = ((uintptr_t)p & 3) == 0;
  for (; pos + 64 <= len; pos += 64) {
    if (aligned) {
      const uint4 *q = reinterpret_cast<const uint4 *>(p + pos);   // 16-byte loads when the segment start allows (p + pos is 4-aligned; 16 only if p is)
      if (((uintptr_t)p & 15) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint4 v = q[k];
          w[4 * k] = __builtin_bswap32(v.x); w[4 * k + 1] = __builtin_bswap32(v.y);
          w[4 * k + 2] = __builtin_bswap32(v.z); w[4 * k + 3] = __builtin_bswap32(v.w);
        }
      } else {
        const uint32_t *d = reinterpret_cast<const uint32_t *>(p + pos);
#pragma unroll
        for (int k = 0; k < 16; ++k) w[k] = __builtin_bswap32(d[k]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) w[k] = be32(p + pos + 4 * k);
    }
    sha1_block(h, w);
  }
  // tail: remaining bytes, 0x80, zero padding, 64-bit bit length
  uint8_t tail[128];
  const uint32_t rem = (uint32_t)(len - pos);
  for (uint32_t k = 0; k < 128; ++k) tail[k] = 0;
  for (uint32_t k = 0; k < rem; ++k) tail[k] = p[pos + k];
  tail[rem] = 0x80;
  const uint32_t tl = rem < 56 ? 64u : 128u;
  const uint64_t bits = len * 8;
  for (int k = 0; k < 8; ++k) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
  for (uint32_t o = 0; o < tl; o += 64) {
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = be32(tail + o + 4 * k);
    sha1_block(h, w);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) digest[5 * i + k] = h[k];
}

extern "C" hipError_t zh_launch_sha1(const uint8_t *data, const uint64_t *seg, uint32_t n_seg, uint32_t *digest, hipStream_t stream) {
  if (n_seg == 0) return hipSuccess;
  hipLaunchKernelGGL(zh_sha1_segments, dim3((n_seg + 63) / 64), dim3(64), 0, stream, data, seg, n_seg, digest);
  return hipGetLastError();
}
