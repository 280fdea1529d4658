// zh_store.hip — unmodelled (n = 0) blocks: Decoder.decompress's store path (Decoder.cs:58-67: [length32 big-endian,
// bytes]*, 0) and behind it the reference's LZ77 post-processors as wave-wide kernels.
//
// The reference's methods "1" and "2" (LibZPAQ.cs:174-186) write blocks with NO model: all of the work is the
// post-processor the block carries — `lazy2` (bit-packed LZ77, LibZPAQ.cs:427-572) or `lzpre` (byte-aligned LZ77,
// :575-639), formats LZBuffer.cs:96-115.  zh_generic.hip runs such a block on one lane through the program's
// translation, a byte at a time.  Here a workgroup of two wavefronts owns a block:
//
//   * PARSER WAVE.  Reads the coded bytes through a 256-byte register window (one aligned dword per lane, mirrored in
//     LDS), parses the LZ77 codes on the scalar unit following the PROGRAM's state machine byte for byte — same
//     registers (bits, n, state, len, m, ptr, r), same order of the tests a byte goes through, same 32-bit arithmetic —,
//     so that the result is the program's on any input, not only on what an encoder writes.  Matches and literal runs
//     are copied by all 64 lanes into a RING of the last 128 KiB of output in LDS: a literal run of `lazy2` is a funnel
//     shift of the coded bytes, a match whose distance is shorter than its length is a periodic fill, a match inside the
//     ring never leaves the CU.  This wave issues no global store while it decodes, so its loads (window refills, matches
//     from further back than the ring) never queue behind a store's acknowledgement (one in-order vmcnt on gfx9).
//   * FLUSHER WAVE.  Follows the parser's write position and moves the ring's bytes to the Writer and to the program's M
//     array (the LZ77 window, 2^pm bytes in the arena slot), which it keeps exactly as the program would: zeros before
//     the first write, contents surviving the end of a segment.  A match from beyond the ring is read from M once the
//     flusher has passed it (its stores are complete before it says so; the parser's loads go to L2).
//
// Only programs that ARE the reference's (structure AND every operand, zh_zpaql_pcomp.h) are taken; anything else —
// other programs, the E8E9 variants, the BWT, a stream that ends inside a chunk, a copy longer than 2^24 — makes the
// block report ZH_E_RETRY: the host then runs it on zh_generic.hip, which is the complete implementation (it starts the
// block from scratch).  PASS blocks (stored bytes, no program) are copied wave-wide by the parser wave.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zh_core.h"
#include "zh_dev.h"
#include "zh_model.h"
#include "zh_zpaql_pcomp.h"
#include "zh_ibwt.h"

using namespace zhcore;
using namespace zhdev;

namespace {

constexpr uint32_t kRing = 128u << 10, kRM = kRing - 1u;   // the last 128 KiB of output
constexpr uint32_t kPiece = 32768u;                        // a copy goes into the ring in pieces of this size at most
constexpr uint32_t kRingKeep = kRing - kPiece - 4096u;     // unflushed bytes the parser may leave behind it
constexpr uint32_t kStSpin = 1u << 26;

struct Lz {
  uint32_t k, avail, left, kdisc;                       // window cursor, valid bytes, bytes left in the chunk, start of consecutive payload
  uint32_t nb, bits, st, len, mbits, r5, off;           // the program's registers (lazy2: d = n, c = bits, R1, R2, R3, R5; lzpre: d = st, R1 = len, R2 = off)
  uint32_t wp, wpub, seg_base, fpos_seen;               // ring: bytes written, last position published, the segment's first, the flusher's last known
  uint32_t rb, minlen, mmask, prog;                     // the method's parameters
  uint32_t flags;                                       // 1: the flusher is lost, 2: not for this kernel (ZH_E_RETRY)
};

struct alignas(16) StoreLds {
  uint8_t ring[kRing];
  uint32_t win[64];                                     // the register window's 256 bytes
  uint8_t csink[64];                                    // lazy2 fast loop: where the lanes beyond a copy's length read and write (no exec switch)
  uint32_t words[64];                                   // operands of the block's program (before: the block's description)
  uint32_t wpos, fpos;                                  // parser: bytes written into the ring; flusher: bytes moved out (per block)
  uint32_t cmd, ack;                                    // parser -> flusher: sequence << 2 | kCmd*; flusher: last command seen
  uint32_t seg_base;                                    // ring position of the segment's first byte (the program's ptr = position - seg_base)
  uint32_t flush_to;                                    // the flusher also moves a last partial round up to here
  uint32_t skip_to;                                     // PASS: the parser wrote the Writer itself up to here
  Lz z;                                                 // the parser's state between two calls of lz_window
};

typedef __attribute__((address_space(3))) uint8_t *lds_u8_p;
__device__ __forceinline__ uint32_t lds_off(const void *p) { return (uint32_t)(uintptr_t)p; }

__device__ __forceinline__ uint32_t st_ld(const uint32_t *p) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// one dword from lane 0; LDS serves a CU's requests in arrival order (see zh_cm.hip): what this wave wrote to LDS before
// is seen by whoever sees this word
__device__ __forceinline__ void st_put0(const uint32_t *where, uint32_t val) {
  const uint32_t addr = (uint32_t)(uintptr_t)where;
  asm volatile("s_mov_b64 exec, 1\n\tds_write_b32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(addr), "v"(val) : "memory");
}
enum : uint32_t { kCmdBlock = 1, kCmdEnd = 2, kCmdExit = 3 };

// ---- the stored byte stream: chunks behind a register window
struct Rd {
  InBuf in;
  uint32_t left;                                        // payload bytes left in the current chunk (Decoder.curr)
  uint32_t kdisc;                                       // window index from which the bytes before the cursor are consecutive
                                                        // payload (the window began, or a chunk header ended, there)
};
enum : int { kEos = -1, kDamaged = -2 };

__device__ __forceinline__ void rd_seek(Rd &r, StoreLds &S, uint64_t pos, uint32_t lane) {
  in_seek(r.in, pos, lane);
  S.win[lane] = r.in.cur;                               // (literals are copied out of the window through LDS)
  r.kdisc = r.in.k;
}
__device__ __forceinline__ int rd_byte(Rd &r, StoreLds &S, uint32_t lane) {
  if (UNLIKELY(r.in.k >= r.in.avail)) {
    rd_seek(r, S, in_pos(r.in), lane);
    if (r.in.k >= r.in.avail) return -1;
  }
  const uint32_t k = r.in.k++;
  return (int)((rdlane(r.in.cur, k >> 2) >> ((k & 3) * 8)) & 255);
}
// chunk header (4 bytes, big-endian) when the chunk is used up; kEos on a zero length.  kDamaged: the stream ends inside
// the header or the chunk runs past the stream (Decoder.get would return -1 in the middle: left to the generic kernel)
__device__ __forceinline__ int rd_chunk(Rd &r, StoreLds &S, uint32_t lane) {
  if (r.left) return 0;
  uint32_t v = 0;
  for (int i = 0; i < 4; ++i) {
    const int c = rd_byte(r, S, lane);
    if (c < 0) return kDamaged;
    v = v << 8 | (uint32_t)c;
  }
  v = uni(v);
  if (v == 0) return kEos;
  if ((uint64_t)v > r.in.total - in_pos(r.in)) return kDamaged;
  r.left = v;
  r.kdisc = r.in.k;
  return 0;
}

// operands of the reference's programs (zh_zpaql_pcomp.h lists the disassembly): a block whose program differs in ANY
// operand that is not a parameter of the method is not taken here
// (kLzpre108: zh_ibwt.h, shared with zh_nibble.hip)
__device__ const uint8_t kLazy302[43] = {255, 8, 0, 1, 3, 0, 3, 2, 7, 3, 5, 1, 2, 3, 1, 2, 1, 1, 1, 1, 1, 1, 2, 3, 2, 2,
                                          2, 1, 0, 3, 1, 1, 1, 1, 1, 1, 1, 4, 4, 7, 8, 8, 0};
// lazy2 with rb > 0 (blocks over 16 MiB): [26..35] = 5, rb-1, (1<<rb)-1, rb, rb, 2, 2, 1, rb, (1<<rb)-1
__device__ const uint8_t kLazy337[51] = {255, 8, 0, 1, 3, 0, 3, 2, 7, 3, 5, 1, 2, 3, 1, 2, 1, 1, 1, 1, 1, 1, 2, 3, 2, 5,
                                          5, 0, 0, 0, 0, 2, 2, 1, 0, 0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 4, 4, 7, 8, 8, 0};

// bwtrle (LibZPAQ.cs:642-795), the two list traversals without E8E9: 16 MiB blocks at most / any size

enum : uint32_t { kProgNone = 0, kProgLzpre = 1, kProgLazy2 = 2, kProgBwt = 3 };

// ---------------------------------------------------------------------------------------------------------------
// FLUSHER WAVE: ring -> Writer and M
// ---------------------------------------------------------------------------------------------------------------
__device__ void store_flusher(const ZhLaunch &L, StoreLds &S, uint32_t lane) {
  uint8_t *slot = L.arena + (uint64_t)blockIdx.x * L.arena_stride;
  uint32_t seen = 0;
  for (;;) {
    uint32_t cmd, sp = 0;
    while ((cmd = st_ld(&S.cmd)) == seen) { __builtin_amdgcn_s_sleep(8); if (++sp > kStSpin) return; }
    seen = cmd;
    if ((cmd & 3u) == kCmdExit) return;
    if ((cmd & 3u) != kCmdBlock) { st_put0(&S.ack, cmd); continue; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // the block: words[] carries its description until a program is loaded (see the parser)
    const uint64_t out_off = uni64((uint64_t)S.words[0] | (uint64_t)S.words[1] << 32), cap = uni64((uint64_t)S.words[2] | (uint64_t)S.words[3] << 32);
    const uint64_t m_off = uni64((uint64_t)S.words[4] | (uint64_t)S.words[5] << 32);
    const uint32_t mmask = uni(S.words[6]);
    uint8_t *out = L.out + out_off;
    uint8_t *Mp = slot + m_off;
    asm volatile("" ::: "memory");
    st_put0(&S.ack, cmd);
    uint32_t fl = 0;                                    // position flushed so far (== Writer position inside this block)
    for (;;) {
      const uint32_t c2 = st_ld(&S.cmd);
      const uint32_t sk = st_ld(&S.skip_to);
      if ((int32_t)(sk - fl) > 0) { fl = sk; st_put0(&S.fpos, fl); }        // PASS: nothing of this comes from the ring
      const uint32_t w = st_ld(&S.wpos), ft = st_ld(&S.flush_to), sb = st_ld(&S.seg_base);
      // whole rounds of 64 up to the parser's position; everything up to flush_to
      uint32_t lim = fl + ((w - fl) & ~63u);
      if ((int32_t)(ft - lim) > 0 && (int32_t)(w - ft) >= 0) lim = ft;
      if ((int32_t)(lim - fl) <= 0) {
        if (c2 != seen) break;                            // the block is over (everything was flushed before the command)
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      while (fl != lim) {
        const uint32_t n = lim - fl < 64u ? lim - fl : 64u;
        if (lane < n) {
          const uint32_t pos = fl + lane;
          const uint8_t v = *(lds_u8_p)(lds_off(S.ring) + (pos & kRM));
          Mp[(pos - sb) & mmask] = v;
          if ((uint64_t)pos < cap) out[pos] = v;
        }
        fl += n;
      }
      __builtin_amdgcn_s_waitcnt(0);                      // the stores are done (L2) before the parser is told
      asm volatile("" ::: "memory");
      st_put0(&S.fpos, fl);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// PARSER WAVE, the hot part: the bytes of the window, as long as the chunk and the window last.  A function of its own
// with its state handed over in LDS: what the block's driver keeps alive (pointers, sizes, results) would otherwise
// crowd the scalar registers of this loop (the first form spilled 87 of them).
// ---------------------------------------------------------------------------------------------------------------

__device__ __attribute__((noinline)) void lz_window(StoreLds &S, Lz &Z, uint32_t cur, const uint8_t *Mp, uint32_t lane, uint64_t *dbg) {
#ifdef ST_PROF
  uint64_t pt0, pt1, pf0 = 0, pf1 = 0, p_far = 0; uint32_t n_a0 = 0, n_a = 0, n_b = 0, n_far = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt0)::"memory");
#endif
  const uint32_t ring = lds_off(S.ring);
  uint32_t k = uni(Z.k);
  const uint32_t k0 = k, avail = uni(Z.avail), left = uni(Z.left), kdisc = uni(Z.kdisc);
  uint32_t nb = uni(Z.nb), bits = uni(Z.bits), st = uni(Z.st), len = uni(Z.len), mbits = uni(Z.mbits), r5 = uni(Z.r5), off = uni(Z.off);
  uint32_t wp = uni(Z.wp), wpub = uni(Z.wpub), fpos_seen = uni(Z.fpos_seen);
  const uint32_t seg_base = uni(Z.seg_base), rb = uni(Z.rb), minlen = uni(Z.minlen), mmask = uni(Z.mmask), prog = uni(Z.prog);
  bool lost = false, retry = false;
  auto wait_flushed = [&](uint32_t need) __attribute__((always_inline)) {
    uint32_t sp = 0;
    while ((int32_t)(need - fpos_seen) > 0) {
      fpos_seen = st_ld(&S.fpos);
      if (++sp > kStSpin) { lost = true; break; }
    }
  };
  auto publish = [&]() __attribute__((always_inline)) { st_put0(&S.wpos, wp); wpub = wp; };
  auto flush_all = [&]() __attribute__((always_inline)) {
    st_put0(&S.flush_to, wp);
    publish();
    wait_flushed(wp);
  };
  // room for n more bytes in the ring (n <= kPiece): the flusher must not be more than kRingKeep behind
  auto ring_room = [&](uint32_t n) __attribute__((always_inline)) {
    if (UNLIKELY(wp + n - fpos_seen > kRingKeep)) {
      publish();
      wait_flushed(wp + n - kRingKeep);
    }
  };
  // n bytes from `dist` back in the ring to wp (the program: `a=*c *b=a c++ b++ out`, a byte at a time: a distance
  // shorter than the length repeats what the copy itself wrote).  64 lanes per round; LDS operations of a wave are
  // performed in order, so a round reads what earlier rounds wrote.
  auto copy_ring = [&](uint32_t dist, uint32_t n) __attribute__((always_inline)) {
    const uint32_t src = wp - dist;
    if (dist >= 64u) {
      for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t j = base + lane;
        if (j < n) {
          const uint8_t v = *(lds_u8_p)(ring + ((src + j) & kRM));
          *(lds_u8_p)(ring + ((wp + j) & kRM)) = v;
        }
      }
    } else {
      const uint32_t rcp = (65536u + dist - 1u) / dist;  // floor(r / dist) == (r * rcp) >> 16 for r < 128
      uint32_t phase = 0;
      for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t j = base + lane;
        const uint32_t r = phase + lane;
        const uint32_t k = r - ((r * rcp) >> 16) * dist;
        if (j < n) {
          const uint8_t v = *(lds_u8_p)(ring + ((src + k) & kRM));
          *(lds_u8_p)(ring + ((wp + j) & kRM)) = v;
        }
        const uint32_t r2 = phase + 64u;
        phase = uni(r2 - ((r2 * rcp) >> 16) * dist);
      }
    }
    wp += n;
  };
  // a match of the program: n bytes M[ptr - a + i] -> M[ptr + i] and the Writer, i = 0..n-1
  auto copy_match = [&](uint32_t a, uint32_t n) __attribute__((always_inline)) {
    const uint32_t dist = uni(a & mmask);              // (M is addressed modulo its size)
    uint32_t done = 0;
    while (done < n && !lost) {
      const uint32_t ptr = wp - seg_base;
      const uint32_t piece = n - done < kPiece ? n - done : kPiece;
      ring_room(piece);
      if (lost) break;
      if (dist != 0u && dist <= ptr && dist <= kRing - 64u) {
        copy_ring(dist, piece);                          // the source is output of this segment still in the ring
      } else if (dist >= piece && (wp - fpos_seen <= dist - piece || (fpos_seen = st_ld(&S.fpos), wp - fpos_seen <= dist - piece))) {
        // from M, further back than the ring (or from before the segment): the flusher has passed the source and the
        // copy does not reach its own output
        const uint32_t p = ptr - dist;
#ifdef ST_PROF
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pf0)::"memory"); ++n_far;
#endif
        for (uint32_t base = 0; base < piece; base += 256) {
          uint32_t v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) { const uint32_t j = base + 64u * k + lane; v[k] = j < piece ? __builtin_nontemporal_load(&Mp[(p + j) & mmask]) : 0u; }
#pragma unroll
          for (int k = 0; k < 4; ++k) { const uint32_t j = base + 64u * k + lane; if (j < piece) *(lds_u8_p)(ring + ((wp + j) & kRM)) = (uint8_t)v[k]; }
        }
#ifdef ST_PROF
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pf1)::"memory"); p_far += pf1 - pf0;
#endif
        wp += piece;
      } else {
        // the rest (a source the flusher has not reached, distance 0 = M rewritten in place, a short period over
        // old data): a round of 64 at a time from M with everything flushed before it
        for (uint32_t base = 0; base < piece && !lost; base += 64) {
          flush_all();
          const uint32_t pc = wp - seg_base;
          const uint32_t cnt = piece - base < 64u ? piece - base : 64u;
          const uint32_t q = dist == 0u ? pc + lane : pc - dist + (dist < 64u ? lane % dist : lane);
          const uint8_t v = lane < cnt ? __builtin_nontemporal_load(&Mp[q & mmask]) : 0;
          if (lane < cnt) *(lds_u8_p)(ring + ((wp + lane) & kRM)) = v;
          wp += cnt;
        }
      }
      done += piece;
    }
  };


  // ---- a program: the bytes of the window, as long as the chunk and the window last (tight loop: the rounds of the
  // outer loop cost a hundred instructions of house-keeping each)
  {
    const uint32_t wbase = lds_off(S.win);
    auto next_byte = [&]() __attribute__((always_inline)) -> uint32_t {
      const uint32_t kk = k++;
      return (rdlane(cur, kk >> 2) >> ((kk & 3u) * 8u)) & 255u;
    };
    if (prog == kProgLzpre) {
      // lzpre (LibZPAQ.cs:575-639): d = state, b = ptr, R1 = len, R2 = offset so far
      const uint32_t kend = avail - k < left ? avail : k + left;        // the window or the chunk ends here
      while (k < kend && !lost) {
        if (st == 1u) {                                                // literals: straight out of the window
          const uint32_t n = len < kend - k ? len : kend - k;
          ring_room(n);
          for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t j = base + lane;
            if (j < n) *(lds_u8_p)(ring + ((wp + j) & kRM)) = *(lds_u8_p)(wbase + k + j);
          }
          wp += n; k += n; len -= n;
          if (len == 0u) st = 0u;
          continue;
        }
        const uint32_t cc = next_byte();
        if (st == 0u) {
          st = (cc >> 6) + 1u;
          if (st == 1u) { len = 1u + cc; off = 0; }
          else { ++st; len = (cc & 63u) + minlen; off = 0; }
        } else if (st > 2u) { off = off << 8 | cc; --st; }
        else {
          off = off << 8 | cc;
          if (len == 0u || len > (1u << 24)) { retry = true; break; }    // (do ... while: 2^32 rounds in the program)
          copy_match(off + 1u, len);
          st = 0;                                                        // (d counted down to 0)
          if (wp - wpub >= 1024u) publish();
        }
      }
    } else {
      // lazy2 (LibZPAQ.cs:427-572): c = bits, d = n, R1 = state, R2 = len, R3 = m, R4 = ptr, R5 = r
      const uint32_t kend = avail - k < left ? avail : k + left;
      // (A) WHOLE CODES.  The program is driven by byte arrivals: a byte is added to the bit buffer, then the stages
      // run once in the order 0 1 5 2 3 4, each only in its state and only with enough bits (n > 2, n > rb - 1,
      // n >= m, n > 1, n > 7); a code that ends (state 0) is followed by the next one only at the next arrival.
      // So a code is a pure function of the bits from where the last one ended, and the number of bytes the
      // program has taken when it acts is: one for stage 0, one more each time a field lacks bits.  That is what
      // (A) does on a 64-bit look at the window — fields by shifts, the byte count by arithmetic — while 16 bytes
      // of window and chunk lie ahead; otherwise, and for anything odd (a length code over 32 bits), the
      // byte-driven pass (B), which IS the program's, goes on from the same state.  (A) does not keep `bits`:
      // they are the n bits before byte k of the window, taken out when (B) wants them.
      bool stale = false;
      auto bits_from_window = [&]() __attribute__((always_inline)) {
        const uint32_t bp2 = 8u * k - nb, ix2 = bp2 >> 5, sh2 = bp2 & 31u;
        const uint32_t x0 = rdlane(cur, ix2), x1 = rdlane(cur, (ix2 + 1u) & 63u);
        const uint64_t qq = ((uint64_t)x1 << 32 | x0) >> sh2;
        bits = (uint32_t)qq & (nb >= 32u ? 0xFFFFFFFFu : (1u << nb) - 1u);
        stale = false;
      };
      const uint32_t v_sink = lds_off(S.csink) + lane;
      while (k < kend && !lost) {
        // (A0) THE COMMON CODE, in assembly (round 4).  On text 96 % of the codes are matches of 4-15 bytes (two length pairs
        // at most) from inside the ring that do not overlap their own output.  The general form (A) below is ~300
        // instructions and ~20 taken branches for one (1 850 cycles per code at 2.2 GB/s: the parser wave is bound by its own
        // instruction stream, nothing else); the compiler's rendering of a straight-line C++ form still took ~870 cycles
        // (mask juggling for every condition, the LDS read waited for at once).  Here: ~85 instructions per code, the
        // conditions as compare + not-taken branch, the copy WRITTEN ONE CODE LATE (the byte a lane has read for code i
        // goes into the ring in front of the read of code i + 1 — LDS serves a wave's requests in order, so that read sees
        // it — and the parse of code i + 1 runs under the LDS round trip), lanes beyond the length on a sink byte instead of
        // an exec switch.  Same arithmetic and byte-arrival accounting as (A); whatever is not this kind of code — a
        // literal run, a longer length, a source beyond the ring or overlapping its output, no room in the ring, the chunk's
        // end, 1 KiB written since the flusher was last told — changes no state and leaves the loop.
        if (LIKELY(st == 0u && rb == 0u)) {
          uint32_t took = 0;
          asm volatile(
              "v_mov_b32_e32 v251, %[snk]\n\t"          /* pending write: address (sink), value */
              "v_mov_b32_e32 v250, 0\n"
              ".Lz2_loop_%=:\n\t"
              "s_sub_u32 s80, %[kend], %[k]\n\t"
              "s_cmp_lt_u32 s80, 9\n\t"
              "s_cbranch_scc1 .Lz2_out_%=\n\t"
              "s_sub_u32 s80, %[k], %[kdisc]\n\t"
              "s_lshl_b32 s80, s80, 3\n\t"
              "s_cmp_lt_u32 s80, %[nb]\n\t"
              "s_cbranch_scc1 .Lz2_out_%=\n\t"
              "s_lshl_b32 s80, %[k], 3\n\t"
              "s_sub_u32 s80, s80, %[nb]\n\t"           /* window bit position of the first unused bit */
              "s_lshr_b32 s81, s80, 5\n\t"
              "s_add_u32 s83, s81, 1\n\t"
              "s_and_b32 s82, s80, 31\n\t"
              "s_add_u32 s88, %[k], 1\n\t"              /* kv0: stage 0 has taken a byte */
              "s_add_u32 s89, %[nb], 3\n\t"             /* nv0 = n + 8 - 5 (`mm mmm`) */
              "s_cmp_le_u32 s89, 2\n\t"
              "s_cselect_b32 s90, 1, 0\n\t"
              "v_readlane_b32 s84, %[cur], s81\n\t"
              "v_readlane_b32 s85, %[cur], s83\n\t"
              "s_add_u32 s88, s88, s90\n\t"
              "s_lshl_b32 s90, s90, 3\n\t"
              "s_add_u32 s89, s89, s90\n\t"
              "s_lshr_b64 s[84:85], s[84:85], s82\n\t"  /* q: at least 33 valid bits */
              "s_and_b32 s86, s84, 3\n\t"
              "s_cmp_eq_u32 s86, 0\n\t"
              "s_cbranch_scc1 .Lz2_out_%=\n\t"          /* 00: a literal run */
              "s_sub_u32 s86, s86, 1\n\t"
              "s_lshl_b32 s86, s86, 3\n\t"
              "s_bfe_u32 s87, s84, 0x30002\n\t"
              "s_add_u32 s86, s86, s87\n\t"             /* m: offset bits */
              "s_bfe_u32 s91, s84, 0x30005\n\t"         /* first length triple */
              "s_bitcmp1_b32 s91, 0\n\t"
              "s_cbranch_scc1 .Lz2_two_%=\n\t"
              "s_lshr_b32 s92, s91, 1\n\t"
              "s_add_u32 s92, s92, 4\n\t"               /* length 4..7 */
              "s_sub_u32 s89, s89, 3\n\t"
              "s_mov_b32 s93, 8\n"
              ".Lz2_got_%=:\n\t"
              "s_sub_u32 s94, s86, s89\n\t"             /* stage 2: bytes taken until n >= m */
              "s_add_u32 s94, s94, 7\n\t"
              "s_lshr_b32 s94, s94, 3\n\t"
              "s_cmp_lt_u32 s89, s86\n\t"
              "s_cselect_b32 s94, s94, 0\n\t"
              "s_add_u32 s88, s88, s94\n\t"
              "s_lshl_b32 s94, s94, 3\n\t"
              "s_add_u32 s89, s89, s94\n\t"
              "s_sub_u32 s89, s89, s86\n\t"
              "s_cmp_gt_u32 s88, %[kend]\n\t"
              "s_cbranch_scc1 .Lz2_out_%=\n\t"
              "s_lshr_b64 s[94:95], s[84:85], s93\n\t"
              "s_bfm_b32 s95, s86, 0\n\t"
              "s_and_b32 s94, s94, s95\n\t"
              "s_lshl_b32 s95, 1, s86\n\t"
              "s_add_u32 s94, s94, s95\n\t"
              "s_and_b32 s94, s94, %[mmask]\n\t"        /* distance (M is addressed modulo its size) */
              "s_cmp_lt_u32 s94, s92\n\t"
              "s_cbranch_scc1 .Lz2_out_%=\n\t"          /* would read its own output */
              "s_sub_u32 s95, %[wp], %[segb]\n\t"
              "s_min_u32 s95, s95, 0x1ffc0\n\t"
              "s_cmp_lt_u32 s95, s94\n\t"
              "s_cbranch_scc1 .Lz2_out_%=\n\t"          /* from before the segment, or from beyond the ring */
              "s_add_u32 s96, %[wp], s92\n\t"
              "s_sub_u32 s95, s96, %[fpos]\n\t"
              "s_cmp_gt_u32 s95, 0x17000\n\t"
              "s_cbranch_scc1 .Lz2_out_%=\n\t"          /* the flusher is too far behind */
              "s_sub_u32 s95, %[wp], s94\n\t"
              "v_add_u32_e32 v252, s95, %[lane]\n\t"
              "v_add_u32_e32 v253, %[wp], %[lane]\n\t"
              "v_and_b32_e32 v252, 0x1ffff, v252\n\t"
              "v_and_b32_e32 v253, 0x1ffff, v253\n\t"
              "v_cmp_gt_u32_e32 vcc, s92, %[lane]\n\t"
              "v_add_u32_e32 v252, %[ring], v252\n\t"
              "v_add_u32_e32 v253, %[ring], v253\n\t"
              "v_cndmask_b32_e32 v252, %[snk], v252, vcc\n\t"
              "s_waitcnt lgkmcnt(0)\n\t"                /* v250 is the ds_read_u8 of the code before: LDS returns are not interlocked with VGPR reads */
              "ds_write_b8 v251, v250\n\t"              /* the code before */
              "v_cndmask_b32_e32 v251, %[snk], v253, vcc\n\t"
              "ds_read_u8 v250, v252\n\t"
              "s_mov_b32 %[wp], s96\n\t"
              "s_mov_b32 %[k], s88\n\t"
              "s_mov_b32 %[nb], s89\n\t"
              "s_mov_b32 %[mb], s86\n\t"
              "s_mov_b32 %[ln], s92\n\t"
              "s_add_u32 %[took], %[took], 1\n\t"
              "s_sub_u32 s95, s96, %[wpub]\n\t"
              "s_cmp_lt_u32 s95, 0x400\n\t"
              "s_cbranch_scc1 .Lz2_loop_%=\n\t"
              "s_branch .Lz2_out_%=\n"
              ".Lz2_two_%=:\n\t"                         /* 1b: a second triple decides */
              "s_sub_u32 s89, s89, 2\n\t"
              "s_cmp_le_u32 s89, 2\n\t"
              "s_cselect_b32 s90, 1, 0\n\t"
              "s_add_u32 s88, s88, s90\n\t"
              "s_lshl_b32 s90, s90, 3\n\t"
              "s_add_u32 s89, s89, s90\n\t"
              "s_bfe_u32 s93, s84, 0x30007\n\t"
              "s_bitcmp1_b32 s93, 0\n\t"
              "s_cbranch_scc1 .Lz2_out_%=\n\t"          /* three pairs or more: (A) */
              "s_bfe_u32 s92, s91, 0x10001\n\t"
              "s_add_u32 s92, s92, 2\n\t"
              "s_lshl_b32 s92, s92, 2\n\t"
              "s_lshr_b32 s93, s93, 1\n\t"
              "s_add_u32 s92, s92, s93\n\t"             /* length 8..15 */
              "s_sub_u32 s89, s89, 3\n\t"
              "s_mov_b32 s93, 10\n\t"
              "s_branch .Lz2_got_%=\n"
              ".Lz2_out_%=:\n\t"
              "s_waitcnt lgkmcnt(0)\n\t"                /* (the read whose byte is written below) */
              "ds_write_b8 v251, v250\n\t"              /* nothing stays pending outside */
              "s_waitcnt lgkmcnt(0)\n\t"
              : [k] "+s"(k), [nb] "+s"(nb), [wp] "+s"(wp), [mb] "+s"(mbits), [ln] "+s"(len), [took] "+s"(took)
              : [kend] "s"(kend), [kdisc] "s"(kdisc), [mmask] "s"(mmask), [segb] "s"(seg_base), [fpos] "s"(fpos_seen), [wpub] "s"(wpub),
                [cur] "v"(cur), [lane] "v"(lane), [ring] "v"(ring), [snk] "v"(v_sink)
              : "memory", "scc", "vcc", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92",
                "s93", "s94", "s95", "s96", "v250", "v251", "v252", "v253");
          if (took) {
            stale = true;
            if (wp - wpub >= 1024u) publish();
            continue;                                     // (the loop's own test first: the window may be used up)
          }
        }
        if (st == 0u && kend - k >= 16u && 8u * (k - kdisc) >= nb) {
          const uint32_t bp = 8u * k - nb;                             // window bit position of the first unused bit
          const uint32_t ix = bp >> 5, sh = bp & 31u;
          const uint32_t w0 = rdlane(cur, ix), w1 = rdlane(cur, ix + 1u), w2 = rdlane(cur, ix + 2u);
          uint64_t q = (uint64_t)w1 << 32 | w0;
          if (sh) q = q >> sh | (uint64_t)w2 << (64u - sh);
          uint32_t kv = k + 1u, nv = nb + 8u, u, ln = 1u;              // stage 0: a byte has arrived
          const uint32_t t = (uint32_t)q & 3u;
          bool odd = false;
          if (t) {                                                     // match: mm mmm, length, [r,] m offset bits
            const uint32_t mb = ((t - 1u) << 3) + (((uint32_t)q >> 2) & 7u);
            u = 5u; nv -= 5u;
            for (;;) {                                                 // stage 1: length, interleaved Elias gamma
              if (nv <= 2u) { ++kv; nv += 8u; }
              const uint32_t f3 = (uint32_t)(q >> u) & 7u;
              if (f3 & 1u) { ln += ln + ((f3 >> 1) & 1u); u += 2u; nv -= 2u; if (u > 32u) { odd = true; break; } }
              else { ln = (ln << 2) + (f3 >> 1); u += 3u; nv -= 3u; break; }
            }
            uint32_t rr = 0;
            if (rb) {                                                  // stage 5: the low bits of the offset
              if (nv <= rb - 1u) { ++kv; nv += 8u; }
              rr = (uint32_t)(q >> u) & ((1u << rb) - 1u); u += rb; nv -= rb;
            }
            while (nv < mb) { ++kv; nv += 8u; }                        // stage 2: m offset bits
            const uint32_t one = 1u << (mb & 31u);
            uint32_t a = ((one - 1u) & (uint32_t)(q >> (u & 63u))) + one;
            if (rb) a = (a << rb) + rr - ((1u << rb) - 1u);
            nv -= mb;
            if (!odd && kv <= kend && ln <= (1u << 24)) {
              r5 = rb ? rr : r5; mbits = mb;
              if (ln) copy_match(a, ln);
#ifdef ST_PROF
              ++n_a;
#endif
              len = ln; k = kv; nb = nv; stale = true;
              if (wp - wpub >= 1024u) publish();
              continue;
            }
          } else {                                                     // literals: 00, length, then the bytes
            u = 2u; nv -= 2u;
            for (;;) {                                                 // stage 3
              if (nv <= 1u) { ++kv; nv += 8u; }
              const uint32_t f2 = (uint32_t)(q >> u) & 3u;
              if (f2 & 1u) { ln += ln + (f2 >> 1); u += 2u; nv -= 2u; if (u > 32u) { odd = true; break; } }
              else { u += 1u; nv -= 1u; break; }
            }
            if (!odd && kv <= kend) {
              st = 4u;
              if (nv > 7u) {                                           // stage 4 of the same arrival: one literal
                ring_room(1u);
                if (lane == 0) *(lds_u8_p)(ring + (wp & kRM)) = (uint8_t)(q >> u);
                ++wp; nv -= 8u;
                if (--ln == 0u) st = 0u;
              }
              k = kv; nb = nv; len = ln; stale = true;
              continue;                                                // (the run itself: the bulk path below)
            }
          }
          // (not taken: the pass below does this byte)
        }
        if (stale) bits_from_window();
        if (st == 4u && nb < 8u) {
          // state 4 with n < 8: every further byte adds 8 bits and gives one literal (LibZPAQ.cs:551-563):
          // literal j = (carry | byte_j << n) & 255, carry = what byte_{j-1} leaves above bit 8 (bits < 2^n)
          const uint32_t n = len < kend - k ? len : kend - k;
          ring_room(n);
          const uint32_t w0 = wbase + k;
          for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t j = base + lane;
            if (j < n) {
              const uint32_t cur = *(lds_u8_p)(w0 + j);
              const uint32_t prev = j ? (uint32_t)*(lds_u8_p)(w0 + j - 1u) : 0u;
              const uint32_t carry = j ? ((prev << nb) >> 8) : bits;
              *(lds_u8_p)(ring + ((wp + j) & kRM)) = (uint8_t)(carry | cur << nb);
            }
          }
          const uint32_t i1 = k + n - 1u;
          const uint32_t last = (rdlane(cur, i1 >> 2) >> ((i1 & 3u) * 8u)) & 255u;
          uint32_t lastc = bits;
          if (n > 1u) { const uint32_t i2 = i1 - 1u; lastc = ((((rdlane(cur, i2 >> 2) >> ((i2 & 3u) * 8u)) & 255u) << nb) >> 8); }
          bits = ((lastc & 255u) | last << nb) >> 8;
          wp += n; k += n; len -= n;
          if (len == 0u) st = 0u;
          if (wp - wpub >= 1024u) publish();
          continue;
        }
#ifdef ST_PROF
        ++n_b;
#endif
        bits += next_byte() << (nb & 31u);
        nb += 8u;
        if (st == 0u) {                                                  // expect a new code
          len = 1u;
          const uint32_t t = bits & 3u;
          if (t > 0u) {
            mbits = (t - 1u) << 3; bits >>= 2;
            mbits += bits & 7u; bits >>= 3;
            nb -= 5u; st = 1u;
          } else { bits >>= 2; nb -= 2u; st = 3u; }
        }
        while (st == 1u && nb > 2u) {                                    // match length, interleaved Elias gamma
          if (bits & 1u) { bits >>= 1; len += len + (bits & 1u); bits >>= 1; nb -= 2u; }
          else { bits >>= 1; len = (len << 2) + (bits & 3u); bits >>= 2; nb -= 3u; st = rb ? 5u : 2u; }
        }
        if (rb && st == 5u && nb > rb - 1u) { r5 = bits & ((1u << rb) - 1u); bits >>= rb; nb -= rb; st = 2u; }
        if (st == 2u && !(mbits > nb)) {                                 // m offset bits
          const uint32_t one = 1u << (mbits & 31u);
          uint32_t a = ((one - 1u) & bits) + one;
          if (rb) a = (a << rb) + r5 - ((1u << rb) - 1u);
          if (len > (1u << 24)) { retry = true; break; }
          if (len) copy_match(a, len);
          bits >>= (mbits & 31u); nb -= mbits;
          st = 0u;
          if (wp - wpub >= 1024u) publish();
        }
        while (st == 3u && nb > 1u) {                                    // literal length
          if (bits & 1u) { bits >>= 1; len += len + (bits & 1u); bits >>= 1; nb -= 2u; }
          else { bits >>= 1; nb -= 1u; st = 4u; }
        }
        if (st == 4u && nb > 7u) {
          ring_room(1u);
          if (lane == 0) *(lds_u8_p)(ring + (wp & kRM)) = (uint8_t)bits;
          ++wp;
          bits >>= 8; nb -= 8u;
          if (--len == 0u) st = 0u;
        }
      }
      if (stale) bits_from_window();
    }
  }

#ifdef ST_PROF
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt1)::"memory");
  if (lane == 0 && dbg) {
    atomicAdd((unsigned long long *)&dbg[0], (unsigned long long)(pt1 - pt0)); atomicAdd((unsigned long long *)&dbg[1], 1ull);
    atomicAdd((unsigned long long *)&dbg[2], (unsigned long long)n_a0); atomicAdd((unsigned long long *)&dbg[3], (unsigned long long)n_a);
    atomicAdd((unsigned long long *)&dbg[4], (unsigned long long)n_b); atomicAdd((unsigned long long *)&dbg[5], (unsigned long long)n_far);
    atomicAdd((unsigned long long *)&dbg[6], (unsigned long long)p_far);
  }
#endif
  if (lane == 0) {
    Z.k = k; Z.left = left - (k - k0);
    Z.nb = nb; Z.bits = bits; Z.st = st; Z.len = len; Z.mbits = mbits; Z.r5 = r5; Z.off = off;
    Z.wp = wp; Z.wpub = wpub; Z.fpos_seen = fpos_seen;
    Z.flags = (lost ? 1u : 0u) | (retry ? 2u : 0u);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

typedef BwtLdsT<4096u> BwtLds;
static_assert(sizeof(BwtLds) <= kRing, "the inverse BWT's tables live in the ring");
}  // namespace

extern "C" __global__ __launch_bounds__(128) void zh_decode_store(ZhLaunch L) {
  __shared__ StoreLds S;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uni(threadIdx.x >> 6);
  uint8_t *slot = L.arena + (uint64_t)blockIdx.x * L.arena_stride;
  if (threadIdx.x == 0) { S.cmd = 0; S.ack = 0; S.wpos = 0; S.fpos = 0; S.flush_to = 0; S.seg_base = 0; S.skip_to = 0; }
  __syncthreads();
  if (wave == 1) { store_flusher(L, S, lane); return; }

  uint32_t cmd_seq = 0;
  const uint32_t ring = lds_off(S.ring);
  for (;;) {
    uint32_t bi = 0;
    if (lane == 0) bi = atomicAdd(L.queue, 1u);
    bi = uni((uint32_t)__shfl((int)bi, 0));
    if (bi >= L.n_blocks) break;                        // every wave reaches an exit (the flusher on kCmdExit)
    const ZhBlockDesc *bdp = &L.blocks[bi];
    const ZhModel *Mo = &L.models[uni(bdp->model)];
    const uint32_t first_seg = uni(bdp->first_seg), n_seg = uni(bdp->n_seg);
    const uint64_t b_out_off = uni64(bdp->out_off), b_out_cap = uni64(bdp->out_cap);
    const uint32_t pmb = uni(Mo->pm);
    const uint8_t *Mp = slot + uni64(Mo->pm_off);
    const uint32_t mmask = pmb < 32 ? (uint32_t)((1ull << pmb) - 1) : 0xFFFFFFFFu;
    uint8_t *pzbuf = slot + uni64(Mo->pz_off) + ZH_CODE_PAD;
    uint8_t *outp = L.out + b_out_off;

    uint32_t wp = 0, wpub = 0;                          // bytes produced by this block = ring position = Writer position; last published
    uint32_t seg_base = 0;                              // ... when the segment began: the program's ptr = wp - seg_base
    int pp_state = 0, pp_hsize = 0;                      // PostProcessor (PostProcessor.cs:12-16)
    uint32_t pp_len = 0, prog = kProgNone;
    uint32_t minlen = 0, rb = 0, bw_n = 0;               // (bwtrle: bytes of the segment collected in M)
    bool retry = false, failed = false, lost = false;
    uint32_t fpos_seen = 0;                             // last value read from the flusher

    // the parser waits until the flusher has passed `need`
    auto wait_flushed = [&](uint32_t need) __attribute__((always_inline)) {
      uint32_t sp = 0;
      while ((int32_t)(need - fpos_seen) > 0) {
        fpos_seen = st_ld(&S.fpos);
        if (++sp > kStSpin) { lost = true; break; }
      }
    };
    auto publish = [&]() __attribute__((always_inline)) { st_put0(&S.wpos, wp); wpub = wp; };
    auto flush_all = [&]() __attribute__((always_inline)) {
      st_put0(&S.flush_to, wp);
      publish();
      wait_flushed(wp);
    };
    // ---- the flusher gets the block
    if (lane == 0) {
      S.words[0] = (uint32_t)b_out_off; S.words[1] = (uint32_t)(b_out_off >> 32);
      S.words[2] = (uint32_t)b_out_cap; S.words[3] = (uint32_t)(b_out_cap >> 32);
      const uint64_t mo = Mo->pm_off;
      S.words[4] = (uint32_t)mo; S.words[5] = (uint32_t)(mo >> 32);
      S.words[6] = mmask;
      S.wpos = 0; S.fpos = 0; S.flush_to = 0; S.seg_base = 0; S.skip_to = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    ++cmd_seq;
    st_put0(&S.cmd, cmd_seq << 2 | kCmdBlock);
    {
      uint32_t sp = 0;
      while (st_ld(&S.ack) != (cmd_seq << 2 | kCmdBlock)) { if (++sp > kStSpin) { lost = true; break; } }
    }
    const bool flusher_on = !lost;

    for (uint32_t s = 0; s < n_seg && !retry && !lost; ++s) {
      const uint32_t si = first_seg + s;
      const uint32_t produced0 = wp;
      int status = 0;
      if (failed) {
        if (lane == 0) {
          ZhSegResult res;
          res.status = ZH_E_SKIPPED; res.pp_state = (uint32_t)pp_state; res.out_off = b_out_off + produced0; res.out_len = 0; res.in_used = 0;
          L.results[si] = res;
        }
        continue;
      }
      const uint64_t seg_off = uni64(L.segs[si].in_off);
      Rd r;
      r.in.stream = L.in; r.in.total = L.in_total; r.in.cbase = 0; r.in.k = 0; r.in.avail = 0; r.in.cur = 0;
      r.left = 0; r.kdisc = 0;
      rd_seek(r, S, seg_off, lane);
      seg_base = wp;                                     // (the flusher is idle: everything before was flushed at the last segment's end)
      st_put0(&S.seg_base, seg_base);

      for (;;) {
        // (one value per wave, said once per round: the state machine below then runs on the scalar unit)
        pp_state = (int)uni((uint32_t)pp_state); prog = uni(prog);
        wp = uni(wp); wpub = uni(wpub); r.left = uni(r.left);
        r.in.k = uni(r.in.k); r.in.avail = uni(r.in.avail); r.in.cbase = uni64(r.in.cbase); r.kdisc = uni(r.kdisc);
        rb = uni(rb); minlen = uni(minlen); fpos_seen = uni(fpos_seen); seg_base = uni(seg_base); bw_n = uni(bw_n);
        if (UNLIKELY(lost)) break;
        if (UNLIKELY(wp > 0xF0000000u)) { retry = true; break; }      // (positions are 32 bits)
        if (wp - wpub >= 1024u && pp_state == 5) publish();            // the flusher follows in steps of 1 KiB
        // the chunk the next byte comes from (Decoder.decompress, store path: header when the last chunk is used up)
        const int hrc = rd_chunk(r, S, lane);
        if (hrc == kDamaged) { retry = true; break; }
        const bool eos = hrc == kEos;
        // ---- PASS: the chunks' payload is the plaintext (copied past the ring: the flusher has nothing to do)
        if (pp_state == 1) {
          if (eos) break;
          const uint64_t from = in_pos(r.in);
          const uint32_t n = r.left;
          for (uint32_t base = 0; base < n; base += 256) {
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const uint32_t j = base + 64u * k + lane; v[k] = j < n ? L.in[from + j] : 0u; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const uint32_t j = base + 64u * k + lane;
              if (j < n && (uint64_t)wp + j < b_out_cap) outp[wp + j] = (uint8_t)v[k];
            }
          }
          wp += n;
          r.left = 0;
          rd_seek(r, S, from + n, lane);
          continue;
        }
        // ---- bwtrle: the segment is collected in M (`*b=a b++`), the inverse BWT runs at its end
        if (pp_state == 5 && prog == kProgBwt) {
          if (!eos) {
            const uint64_t from = in_pos(r.in);
            const uint32_t n = r.left;
            if ((uint64_t)bw_n + n > (uint64_t)mmask + 1u) { retry = true; break; }      // (M would wrap)
            uint8_t *Mw = slot + uni64(Mo->pm_off);
            for (uint32_t base = 0; base < n; base += 256) {
              uint32_t v[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) { const uint32_t j = base + 64u * k + lane; v[k] = j < n ? L.in[from + j] : 0u; }
#pragma unroll
              for (int k = 0; k < 4; ++k) { const uint32_t j = base + 64u * k + lane; if (j < n) Mw[bw_n + j] = (uint8_t)v[k]; }
            }
            bw_n += n;
            r.left = 0;
            rd_seek(r, S, from + n, lane);
            continue;
          }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
          __builtin_amdgcn_s_waitcnt(0);
          uint32_t produced_b = 0;
          const bool okb = n_seg == 1u && ibwt_block(*reinterpret_cast<BwtLds *>(S.ring), Mp, reinterpret_cast<uint32_t *>(slot + uni64(Mo->ph_off)), bw_n, (uint64_t)mmask + 1u,
                                                     1ull << uni(Mo->ph), outp, b_out_cap, &produced_b, lane);
          if (!okb) { retry = true; break; }
          wp += produced_b;
          break;
        }
        // ---- a program: the bytes of the window, as long as the chunk and the window last (lz_window)
        if (pp_state == 5 && !eos) {
          if (r.in.k >= r.in.avail) { rd_seek(r, S, in_pos(r.in), lane); if (r.in.k >= r.in.avail) { retry = true; break; } continue; }
          if (lane == 0) {
            S.z.k = r.in.k; S.z.avail = r.in.avail; S.z.left = r.left; S.z.kdisc = r.kdisc;
            S.z.wp = wp; S.z.wpub = wpub; S.z.fpos_seen = fpos_seen; S.z.seg_base = seg_base;
          }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          lz_window(S, S.z, r.in.cur, Mp, lane, L.debug);
          r.in.k = uni(S.z.k); r.left = uni(S.z.left);
          wp = uni(S.z.wp); wpub = uni(S.z.wpub); fpos_seen = uni(S.z.fpos_seen);
          const uint32_t fl = uni(S.z.flags);
          if (fl & 1u) lost = true;
          if (fl & 2u) { retry = true; break; }
          continue;
        }
        // ---- one decoded byte (the end of the segment for a program; the post-processor's header)
        int c = -1;
        if (!eos) { --r.left; c = rd_byte(r, S, lane); }
        if (pp_state == 5) {
          // a> 255: the programs reset their state (ptr = 0: the next segment's base) and halt
          if (lane == 0) { S.z.bits = 0; S.z.nb = 0; S.z.st = 0; S.z.len = 0; S.z.mbits = 0; S.z.off = 0; }
          break;
        }
        // ---- PostProcessor.write, states 0 and 2-4 (PostProcessor.cs:37-79)
        if (pp_state == 0) {
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
        } else {                                        // state 4: the program's bytes
          if (c < 0) { status = ZH_E_PP_EOS; break; }
          if (lane == 0) pzbuf[pp_len] = (uint8_t)c;
          if ((int)++pp_len == pp_hsize) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            __builtin_amdgcn_s_waitcnt(0);
            const uint32_t id = uni(zh_pcomp_lookup(pzbuf, pp_len));
            zh_pcomp_operands(id, pzbuf, S.words);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            bool ok = false;
            if (id == ZH_PCOMP_LZPRE_108) {
              ok = true;
              for (int k = 0; k < 12; ++k) ok = ok && (k == 5 || S.words[k] == kLzpre108[k]);
              minlen = uni(S.words[5]); prog = kProgLzpre;
            } else if (id == ZH_PCOMP_LAZY2_302) {
              ok = true;
              for (int k = 0; k < 43; ++k) ok = ok && S.words[k] == kLazy302[k];
              rb = 0; prog = kProgLazy2;
            } else if (id == ZH_PCOMP_LAZY2_337) {
              rb = uni(S.words[27]) + 1u;
              ok = rb >= 1u && rb <= 8u;
              const uint32_t mk = (1u << rb) - 1u;
              for (int k = 0; k < 51 && ok; ++k) {
                uint32_t want = kLazy337[k];
                if (k == 27) want = rb - 1u;
                else if (k == 28 || k == 35) want = mk;
                else if (k == 29 || k == 30 || k == 34) want = rb;
                ok = S.words[k] == want;
              }
              prog = kProgLazy2;
            }
            else if (id == ZH_PCOMP_BWTRLE_123 || id == ZH_PCOMP_BWTRLE_106) {
              ok = true;
              const int nk = id == ZH_PCOMP_BWTRLE_123 ? 11 : 9;
              for (int k = 0; k < nk; ++k) ok = ok && S.words[k] == (id == ZH_PCOMP_BWTRLE_123 ? kBwt123[k] : kBwt106[k]);
              ok = ok && uni(Mo->ph) <= 31u;
              prog = kProgBwt;
            }
            ok = uni(ok ? 1u : 0u) != 0u;
            // every other program (and an M that 32 bits of mask cannot address) is the generic kernel's
            if (!ok || pmb > 31u) { retry = true; break; }
            {                                           // ZPAQL.initp: M starts as zeros (ZPAQL.cs:1010-1026)
              uint4 *q = reinterpret_cast<uint4 *>(slot + uni64(Mo->pm_off));
              const uint64_t n16 = ((uint64_t)mmask + 1u) / 16u;
              for (uint64_t i = lane; i < n16; i += 64) q[i] = make_uint4(0, 0, 0, 0);
              if (mmask < 15u) { if (lane <= mmask) (slot + uni64(Mo->pm_off))[lane] = 0; }
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
              __builtin_amdgcn_s_waitcnt(0);
            }
            if (lane == 0) {
              S.z.bits = S.z.nb = S.z.st = S.z.len = S.z.mbits = S.z.r5 = S.z.off = 0;
              S.z.rb = rb; S.z.minlen = minlen; S.z.mmask = mmask; S.z.prog = prog;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            pp_state = 5;
          }
        }
      }

      // ---- end of the segment: everything out of the ring before the results are written and M is read again
      if (flusher_on && !lost) {
        if (pp_state == 1 || prog == kProgBwt) { st_put0(&S.skip_to, wp); wait_flushed(wp); }     // (PASS and the inverse BWT wrote the Writer themselves)
        else flush_all();
      }
      if (retry || lost) break;
      const uint64_t produced = wp;
      if (!status && produced > b_out_cap) status = ZH_E_OUTPUT_FULL;
      if (status && status != ZH_E_OUTPUT_FULL) failed = true;
      if (lane == 0) {
        ZhSegResult res;
        res.status = status; res.pp_state = (uint32_t)pp_state | (uint32_t)pp_hsize << 8;
        res.out_off = b_out_off + produced0; res.out_len = produced - produced0;
        res.in_used = in_pos(r.in) - seg_off;
        L.results[si] = res;
      }
    }
    if (flusher_on) {                                   // the flusher leaves the block
      ++cmd_seq;
      st_put0(&S.cmd, cmd_seq << 2 | kCmdEnd);
      uint32_t sp = 0;
      while (st_ld(&S.ack) != (cmd_seq << 2 | kCmdEnd)) { if (++sp > kStSpin) { lost = true; break; } }
    }
    if ((retry || lost) && lane == 0) {                  // the whole block goes to the generic kernel
      for (uint32_t s = 0; s < n_seg; ++s) {
        ZhSegResult res;
        res.status = ZH_E_RETRY; res.pp_state = 0; res.out_off = b_out_off; res.out_len = 0; res.in_used = 0;
        L.results[first_seg + s] = res;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  }
  ++cmd_seq;
  st_put0(&S.cmd, cmd_seq << 2 | kCmdExit);
}

extern "C" hipError_t zh_launch_store(const ZhLaunch *L, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(zh_decode_store, dim3(grid), dim3(128), 0, stream, *L);    // parser wave + flusher wave
  return hipGetLastError();
}
