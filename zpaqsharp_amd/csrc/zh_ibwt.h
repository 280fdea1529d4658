// zh_ibwt.h — the inverse BWT of the reference's `bwtrle` post-processor (LibZPAQ.cs:642-795) as a wave-wide device function,
// shared by zh_store.hip (unmodelled BWT blocks) and zh_nibble.hip (the modelled BWT method `ci1`).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zh_dev.h"

namespace {
using zhdev::uni;

// operands of the reference's two bwtrle programs (zh_zpaql_pcomp.h lists the disassembly; zh_store.hip says why every operand
// is compared): blocks up to 16 MiB / any size
// ... and of its lzpre program (LibZPAQ.cs:575-639)
__device__ const uint8_t kLzpre108[12] = {255, 0, 6, 1, 63, 0 /* minimum match */, 1, 0, 2, 8, 8, 0};
__device__ const uint8_t kBwt123[11] = {255, 8, 8, 8, 0, 255, 1, 255, 8, 0, 8};
__device__ const uint8_t kBwt106[9] = {255, 8, 8, 8, 0, 255, 1, 255, 0};

// ---------------------------------------------------------------------------------------------------------------
// Inverse BWT of `bwtrle` (LibZPAQ.cs:642-795) for a well-formed block, wave-wide.
//
// The program collects the segment in M, and at its end: idx = the last 4 bytes, n = the rest; counts the bytes; builds
// the list T[1 + #(bytes < c) + #(earlier c)] = b for every position b != idx with byte c (a stable counting sort by
// byte value, slot 0 left to the end-of-string symbol, which sits at idx coded as 255); then walks it:
// d = idx; while (d) { d = T[d]; out(M[d]); } — one dependent load per byte, 4 million in a row for a 4 MiB block.
// Here:  (1) per-lane histograms of 64 contiguous chunks of M in LDS, (2) their prefix sums, (3) the stable scatter
// T[pos] = b by chunk (T is the program's H array in the arena), (4) the walk cut at SPLITTERS — every position p with
// (p - 1) % step == 0, a few thousand of them, and idx: every lane walks from one splitter to the next and notes where
// it ended and after how many steps, lanes take splitters from a counter; one lane strings the pieces together from
// idx (offsets = prefix sums of the lengths); a second walk writes every piece's bytes at its offset.  64 independent
// chains per wavefront instead of one.
// Taken only when the program would do exactly this: one segment in the block, 1 <= idx < n, M[idx] == 255 (so the
// counts are those of a BWT), n + 256 <= |H| and n + 4 <= |M| (no wrap), and the walk from idx reaches 0 without a
// cycle; anything else is left to the generic kernel (returns false).  *out_len = bytes the program would have written.
// ---------------------------------------------------------------------------------------------------------------
template <uint32_t kSplitMax>
struct BwtLdsT {                                        // zh_store.hip: overlays StoreLds::ring (128 KiB), 4096 splitters; zh_nibble.hip: its tables, 1024
  uint32_t hist[256][64];                               // [byte][chunk]: counts, then first free list position (lanes side by side: no bank conflicts)
  uint32_t s_len[kSplitMax + 1], s_next[kSplitMax + 1], s_off[kSplitMax + 1];   // per splitter (+ idx as the last one)
  uint32_t total[256];
  uint32_t counter, bad, out_total;
};

// *touched: the list T (the program's H array) has been written — a caller that falls back to the program itself zeroes it first
template <uint32_t kSplitMax>
__device__ __attribute__((noinline)) bool ibwt_block(BwtLdsT<kSplitMax> &B, const uint8_t *Mp, uint32_t *T, uint32_t n_in, uint64_t msize, uint64_t hsize,
                                                     uint8_t *outp, uint64_t out_cap, uint32_t *out_len, uint32_t lane, bool *touched = nullptr) {
  if (touched) *touched = false;
  if (n_in < 6u || (uint64_t)n_in > msize) return false;
  const uint32_t n = n_in - 4u;
  const uint32_t idx = uni((uint32_t)Mp[n] | (uint32_t)Mp[n + 1] << 8 | (uint32_t)Mp[n + 2] << 16 | (uint32_t)Mp[n + 3] << 24);
  if (idx == 0u || idx >= n || (uint64_t)n + 256u > hsize || n >= 0x7FFFFFFFu) return false;
  if (uni((uint32_t)Mp[idx]) != 255u) return false;
  // ---- (1) histograms: lane l counts chunk l
  const uint32_t ch = (n + 63u) / 64u, b0 = lane * ch, b1 = b0 + ch < n ? b0 + ch : n;
  for (uint32_t c = 0; c < 256; ++c) B.hist[c][lane] = 0;
  if (lane == 0) { B.counter = 0; B.bad = 0; B.out_total = 0; }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  for (uint32_t b = b0; b < b1; ++b) {
    if (b == idx) continue;
    const uint32_t c = Mp[b];
    B.hist[c][lane] += 1u;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  // ---- (2) first list position of every (chunk, byte): 1 + #(bytes < c) + #(c in earlier chunks); lane l does bytes 4l..4l+3
  for (uint32_t q = 0; q < 4; ++q) {
    const uint32_t c = lane * 4u + q;
    uint32_t sum = 0;
    for (uint32_t l = 0; l < 64; ++l) sum += B.hist[c][l];
    B.total[c] = sum;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  for (uint32_t q = 0; q < 4; ++q) {
    const uint32_t c = lane * 4u + q;
    uint32_t base = 1u;
    for (uint32_t c2 = 0; c2 < c; ++c2) base += B.total[c2];
    for (uint32_t l = 0; l < 64; ++l) { const uint32_t h = B.hist[c][l]; B.hist[c][l] = base; base += h; }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  // ---- (3) the list: stable within a byte value (chunks in order, positions in order inside a chunk)
  if (touched) *touched = true;
  for (uint32_t b = b0; b < b1; ++b) {
    if (b == idx) continue;
    const uint32_t c = Mp[b];
    const uint32_t pos = B.hist[c][lane];
    B.hist[c][lane] = pos + 1u;
    T[pos] = b;
  }
  if (lane == 0) T[0] = 0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  __builtin_amdgcn_s_waitcnt(0);
  // ---- (4) the walk, cut at the splitters
  uint32_t step = 1u;
  while ((uint64_t)step * kSplitMax < (uint64_t)n) step <<= 1;
  const uint32_t nsplit = (n - 1u + step - 1u) / step;  // grid positions 1, 1 + step, ... below n
  const uint32_t smask = step - 1u;
  for (int pass = 0; pass < 2; ++pass) {
    if (lane == 0) B.counter = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // every lane walks a piece; a lane that ends one takes the next splitter at once (the wave iterates while any lane
    // has work: pieces differ a lot in length)
    bool busy = false, finished = false;
    uint32_t i = 0, p = 0, cnt = 0, off = 0;
    while (__ballot(!finished) != 0) {
      if (!busy && !finished) {
        i = atomicAdd(&B.counter, 1u);                    // (the last index: idx itself)
        if (i > nsplit) finished = true;
        else {
          p = i == nsplit ? idx : 1u + i * step;
          cnt = 0;
          off = pass ? B.s_off[i] : 0u;
          if (i != nsplit && p == idx) { if (pass == 0) { B.s_len[i] = 0; B.s_next[i] = 0xFFFFFFFFu; } }   // (idx is the last splitter)
          else if (pass == 1 && off == 0xFFFFFFFFu) {}                                              // (not on the way from idx)
          else busy = true;
        }
      }
      if (busy) {
        p = __builtin_nontemporal_load(&T[p]);
        if (pass) { const uint8_t v = Mp[p]; if ((uint64_t)off + cnt < out_cap) outp[off + cnt] = v; }
        ++cnt;
        uint32_t nxt = 0xFFFFFFFDu;
        if (p == 0u) nxt = 0xFFFFFFFEu;                                // the end of the text
        else if (p == idx) nxt = nsplit;
        else if (((p - 1u) & smask) == 0u) nxt = (p - 1u) / step;
        else if (cnt > n) { B.bad = 1u; nxt = 0xFFFFFFFFu; }          // (a cycle: not a BWT)
        if (nxt != 0xFFFFFFFDu) {
          if (pass == 0) { B.s_len[i] = cnt; B.s_next[i] = nxt; }
          busy = false;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (pass == 0) {
      // one lane strings the pieces together from idx
      for (uint32_t i = lane; i <= nsplit; i += 64) B.s_off[i] = 0xFFFFFFFFu;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      if (lane == 0) {
        uint32_t i = nsplit, off = 0, hops = 0;
        for (;;) {
          if (B.s_off[i] != 0xFFFFFFFFu || ++hops > nsplit + 2u) { B.bad = 1u; break; }      // a piece twice: a cycle
          B.s_off[i] = off;
          off += B.s_len[i];
          const uint32_t nx = B.s_next[i];
          if (nx == 0xFFFFFFFEu) break;
          if (nx == 0xFFFFFFFFu) { B.bad = 1u; break; }
          i = nx;
        }
        B.out_total = off;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      if (uni(B.bad)) return false;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  __builtin_amdgcn_s_waitcnt(0);
  *out_len = uni(B.out_total);
  return true;
}
}  // namespace
