// zh_dev.h — device-side helpers shared by the lane-parallel kernels (zh_cm.hip,
// zh_chain.hip): wave-uniform value pinning, the register-buffered stream reader, the
// register-parked plaintext writer and the hand-scheduled scalar arithmetic-decoder step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zh_model.h"

namespace zhdev {

#define LIKELY(x) __builtin_expect(!!(x), 1)
#define UNLIKELY(x) __builtin_expect(!!(x), 0)

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
  return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v);
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}
// park[lane] = val with both operands scalar: the lane select has to go through M0 (one
// constant-bus operand per VALU instruction on gfx9); M0 is saved and restored.
__device__ __forceinline__ uint32_t wrlane(uint32_t val, uint32_t lane, uint32_t park) {
  uint32_t keep;
  asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
               : "+v"(park), "=&s"(keep) : "s"(uni(val)), "s"(uni(lane)));
  return park;
}

// NOTE on 64-bit values in this file: the scalar unit has no ordered 64-bit compare, so any
// `a < b` on uint64_t is selected onto the vector unit and turns a uniform branch into an
// exec-mask dance.  The per-byte paths therefore only ever compare 32-bit counters.

// ---- compressed-byte reader: one 256-byte chunk of the stream, a dword per lane ----
struct InBuf {
  const uint8_t *stream;                  // whole stream (uniform)
  uint64_t total;                         // stream length = EOF (Decoder.get reads the caller's Reader,
                                          // which does not stop at a segment end: Decoder.cs:112-122)
  uint64_t cbase;                         // stream offset of the chunk in `cur` (multiple of 4)
  uint32_t k;                             // cursor inside the chunk: position = cbase + k
  uint32_t avail;                         // valid bytes in the chunk (<= 256)
  uint32_t cur;                           // per-lane dword of the chunk
};

__device__ __forceinline__ uint64_t in_pos(const InBuf &in) { return in.cbase + in.k; }

__device__ __forceinline__ void in_seek(InBuf &in, uint64_t pos, uint32_t lane) {
  in.cbase = pos & ~3ull;
  in.k = (uint32_t)(pos & 3);
  const uint64_t left = in.total > in.cbase ? in.total - in.cbase : 0;
  in.avail = left < 256 ? (uint32_t)left : 256u;
  const uint32_t o = 4u * lane;           // whole-dword reads; the stream buffer is readable up to its 4-byte rounded end
  in.cur = o < in.avail ? *reinterpret_cast<const uint32_t *>(in.stream + in.cbase + o) : 0u;
}

// Decoder.get(): next coded byte, or -1 at the end of the stream.
__device__ __forceinline__ int in_get(InBuf &in, uint32_t lane) {
  if (UNLIKELY(in.k >= in.avail)) {
    in_seek(in, in_pos(in), lane);
    if (in.k >= in.avail) return -1;
  }
  const uint32_t k = in.k++;
  return (int)((rdlane(in.cur, k >> 2) >> ((k & 3) * 8)) & 255);
}

// ---- plaintext writer: dwords assembled on the SALU, parked in a VGPR, 256-byte stores ----
struct OutBuf {
  uint8_t *base;                          // block's output base (uniform)
  uint64_t cap, len;                      // capacity / bytes produced (len counts past cap)
  uint64_t stored;                        // bytes already in HBM
  uint32_t room;                          // bytes that may still be stored (clamped to 32 bits, refreshed per chunk)
  uint32_t word;                          // bytes of the current dword (scalar)
  uint32_t park;                          // per-lane: dword (vpos>>2)&63 of the current 256-byte chunk
};

__device__ __forceinline__ void out_room(OutBuf &o) {
  const uint64_t r = o.cap > o.len ? o.cap - o.len : 0;
  o.room = r > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)r;
}

// Writes bytes [stored, lim) out of the parked chunk.  vpos = (base & 255) + position, so the
// chunk in `park` is 256-byte aligned in memory and lane l holds its bytes 4l..4l+3.
__device__ __forceinline__ void out_flush(OutBuf &o, uint32_t lane) {
  const uint64_t lim = o.len < o.cap ? o.len : o.cap;
  if (lim > o.stored) {
    const uint32_t phase = (uint32_t)((uintptr_t)o.base & 255);
    const uint64_t vend = phase + lim;
    uint32_t park = o.park;
    if (vend & 3) park = wrlane(o.word, (uint32_t)((vend >> 2) & 63), park);
    const uint64_t v0 = phase + o.stored, v1 = vend;                 // virtual byte range to write
    const uint64_t chunk_v = (v1 - 1) & ~255ull;                     // the parked chunk (holds the last byte)
    const uint64_t lv = chunk_v + 4ull * lane;                       // this lane's dword, virtual
    uint8_t *dst = o.base + (lv - phase);
    if (lv >= v0 && lv + 4 <= v1) *reinterpret_cast<uint32_t *>(dst) = park;
    else
      for (int i = 0; i < 4; ++i)
        if (lv + i >= v0 && lv + i < v1) dst[i] = (uint8_t)(park >> (8 * i));
    o.stored = lim;
  }
  out_room(o);
}

__device__ __forceinline__ void out_put(OutBuf &o, uint32_t c, uint32_t lane) {
  const uint32_t v = (uint32_t)(uintptr_t)o.base + (uint32_t)o.len;  // low bits of the virtual position
  ++o.len;
  if (UNLIKELY(o.room == 0)) return;                                 // count-only past the capacity
  --o.room;
  const uint32_t sh = (v & 3) * 8;
  o.word = sh ? (o.word | c << sh) : c;
  if ((v & 3) == 3) {
    o.park = wrlane(o.word, (v >> 2) & 63, o.park);
    if (UNLIKELY((v & 255) == 255)) out_flush(o, lane);
  }
}

struct Dec { uint32_t low, high, curr; };

// One Decoder.decode step (Decoder.cs:136-158) on the scalar unit.
//   ps  = p16 << 16 where p16 = predict()*2+1 (or 0 for the EOS flag): (range*p16)>>16 == mulhi(range, ps)
//   j   = (j << 1) | y
//   bad |= (curr < low || curr > high)            ("archive corrupted")
//   rn  = (high ^ low) < 2^24  after the split    (renormalisation needed)
#define ZH_DEC_STEP(d, ps, j, bad, rn)                                                              \
  do {                                                                                              \
    uint32_t r_, t_, off_, mid_, m1_, x_;                                                           \
    asm volatile(                                                                                   \
        "s_sub_u32 %[r], %[high], %[low]\n\t"                                                       \
        "s_sub_u32 %[t], %[curr], %[low]\n\t"                                                       \
        "s_mul_hi_u32 %[off], %[r], %[p]\n\t"                                                       \
        "s_cmp_gt_u32 %[t], %[r]\n\t"                                                               \
        "s_cselect_b32 %[bd], 1, %[bd]\n\t"                                                         \
        "s_add_u32 %[mid], %[low], %[off]\n\t"                                                      \
        "s_add_u32 %[m1], %[mid], 1\n\t"                                                            \
        "s_cmp_le_u32 %[t], %[off]\n\t"                                                             \
        "s_cselect_b32 %[high], %[mid], %[high]\n\t"                                                \
        "s_cselect_b32 %[low], %[low], %[m1]\n\t"                                                   \
        "s_addc_u32 %[jj], %[jj], %[jj]\n\t"                                                        \
        "s_xor_b32 %[x], %[high], %[low]\n\t"                                                       \
        "s_cmp_lt_u32 %[x], 0x1000000\n\t"                                                          \
        "s_cselect_b32 %[rn_], 1, 0"                                                                \
        : [low] "+s"(d.low), [high] "+s"(d.high), [curr] "+s"(d.curr), [jj] "+s"(j), [bd] "+s"(bad), \
          [rn_] "=s"(rn),                                                                           \
          [r] "=&s"(r_), [t] "=&s"(t_), [off] "=&s"(off_), [mid] "=&s"(mid_), [m1] "=&s"(m1_),      \
          [x] "=&s"(x_)                                                                             \
        : [p] "s"(ps)                                                                               \
        : "scc");                                                                                   \
  } while (0)

// The same step without the range check and with the renormalisation test left to the caller
// (x = high ^ low after the split; renormalise when x < 2^24).  The check may be dropped because
// the split keeps low <= curr <= high: only priming and the `low += (low == 0)` of a
// renormalisation can break the invariant, and dec_renorm_chk reports exactly that.
#define ZH_DEC_STEP_LITE(d, ps, j, x)                                                               \
  do {                                                                                              \
    uint32_t r_, off_, mid_, m1_;                                                                   \
    asm volatile(                                                                                   \
        "s_sub_u32 %[r], %[high], %[low]\n\t"                                                       \
        "s_mul_hi_u32 %[off], %[r], %[p]\n\t"                                                       \
        "s_add_u32 %[mid], %[low], %[off]\n\t"                                                      \
        "s_add_u32 %[m1], %[mid], 1\n\t"                                                            \
        "s_cmp_le_u32 %[curr], %[mid]\n\t"                                                          \
        "s_cselect_b32 %[high], %[mid], %[high]\n\t"                                                \
        "s_cselect_b32 %[low], %[low], %[m1]\n\t"                                                   \
        "s_addc_u32 %[jj], %[jj], %[jj]\n\t"                                                        \
        "s_xor_b32 %[xx], %[high], %[low]"                                                          \
        : [low] "+s"(d.low), [high] "+s"(d.high), [curr] "+s"(d.curr), [jj] "+s"(j), [xx] "=s"(x),  \
          [r] "=&s"(r_), [off] "=&s"(off_), [mid] "=&s"(mid_), [m1] "=&s"(m1_)                      \
        : [p] "s"(ps)                                                                               \
        : "scc");                                                                                   \
  } while (0)

// The same step handing the decoded bit to the VECTOR side in the forms its update wants, made while SCC still holds
// y = (curr <= mid): ym = y ? ~0 : 0 as an SGPR pair (the select mask of every v_cndmask that picks between the two
// pre-fetched candidates of the next bit: sel_y below), ey = y ? 32767 : 0 (Predictor.cs:366 `y * 32767`), yy = y.
// Round 3 let the compiler derive them from j afterwards: s_and, s_bfe_i32, s_and, s_cmp, s_cselect_b64 per bit.
#define ZH_DEC_STEP_Y(d, ps, j, x, ym, ey, yy)                                                      \
  do {                                                                                              \
    uint32_t r_, off_, mid_, m1_;                                                                   \
    asm volatile(                                                                                   \
        "s_sub_u32 %[r], %[high], %[low]\n\t"                                                       \
        "s_mul_hi_u32 %[off], %[r], %[p]\n\t"                                                       \
        "s_add_u32 %[mid], %[low], %[off]\n\t"                                                      \
        "s_add_u32 %[m1], %[mid], 1\n\t"                                                            \
        "s_cmp_le_u32 %[curr], %[mid]\n\t"                                                          \
        "s_cselect_b32 %[high], %[mid], %[high]\n\t"                                                \
        "s_cselect_b32 %[low], %[low], %[m1]\n\t"                                                   \
        "s_cselect_b64 %[ym_], -1, 0\n\t"                                                           \
        "s_cselect_b32 %[ey_], 0x7fff, 0\n\t"                                                       \
        "s_cselect_b32 %[yy_], 1, 0\n\t"                                                            \
        "s_addc_u32 %[jj], %[jj], %[jj]\n\t"                                                        \
        "s_xor_b32 %[xx], %[high], %[low]"                                                          \
        : [low] "+s"(d.low), [high] "+s"(d.high), [curr] "+s"(d.curr), [jj] "+s"(j), [xx] "=s"(x),  \
          [ym_] "=&s"(ym), [ey_] "=&s"(ey), [yy_] "=&s"(yy),                                        \
          [r] "=&s"(r_), [off] "=&s"(off_), [mid] "=&s"(mid_), [m1] "=&s"(m1_)                      \
        : [p] "s"(ps)                                                                               \
        : "scc");                                                                                   \
  } while (0)
// a1 where the wave-uniform mask says 1, else a0 (one v_cndmask_b32 on the SGPR pair made by ZH_DEC_STEP_Y)
__device__ __forceinline__ uint32_t sel_y(uint32_t a0, uint32_t a1, uint64_t ym) {
  uint32_t r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a0), "v"(a1), "s"(ym));   // (not volatile: the scheduler places it; a select that waits for a load must not hold up the ones behind it)
  return r;
}

// Renormalisation loop of Decoder.decode (Decoder.cs:148-156).  Returns 0 or ZH_E_EOF.
__device__ __forceinline__ int dec_renorm(Dec &d, InBuf &in, uint32_t lane) {
  int rc = 0;
  uint32_t low = d.low, high = d.high, curr = d.curr;
  do {
    high = high << 8 | 255;
    low = low << 8;
    low = low ? low : 1u;                 // low += (low == 0)
    int c = in_get(in, lane);
    if (c < 0) { rc = ZH_E_EOF; break; }
    curr = curr << 8 | (uint32_t)c;
  } while ((high ^ low) < 0x1000000u);
  // this rare path may be selected onto the vector unit; hand the state back as scalars
  d.low = uni(low); d.high = uni(high); d.curr = uni(curr);
  return (int)uni((uint32_t)rc);
}

// dec_renorm plus the range test the NEXT Decoder.decode call would make (Decoder.cs:138):
// `bad` is raised when the renormalised state has curr outside [low, high].  Not touched on EOF.
__device__ __forceinline__ int dec_renorm_chk(Dec &d, InBuf &in, uint32_t lane, uint32_t &bad) {
  const int rc = dec_renorm(d, in, lane);
  if (!rc) bad = uni(bad | (uint32_t)(d.curr < d.low) | (uint32_t)(d.curr > d.high));
  return rc;
}

// Wave-wide integer sum (DPP row shifts + row broadcasts); result is wave-uniform.
__device__ __forceinline__ int wave_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
  return (int)rdlane((uint32_t)v, 63);
}

// In-kernel stamps (diagnostic build only: *_prof kernels): cycles spent per
// stage of a byte, summed per block and added to L.debug[stage].
// ZH_STAMP_WAIT: what a stamp drains first.  Default: everything (a stage pays for its own memory latency);
// make CXXFLAGS+=-DZH_STAMP_NOVM leaves global memory operations in flight (stages are charged the way the
// non-diagnostic build runs them).
#if defined(ZH_STAMP_NOVM)
#define ZH_STAMP_WAIT "s_waitcnt lgkmcnt(0)"
#else
#define ZH_STAMP_WAIT "s_waitcnt vmcnt(0) lgkmcnt(0)"
#endif
#define ZH_STAMP(i)                                                              \
  do {                                                                           \
    if (PROF) {                                                                  \
      uint64_t now_;                                                             \
      __builtin_amdgcn_sched_barrier(0);                                         \
      asm volatile(ZH_STAMP_WAIT "\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
      __builtin_amdgcn_sched_barrier(0);                                         \
      prof[i] += now_ - tprev;                                                   \
      tprev = now_;                                                              \
    }                                                                            \
  } while (0)


}  // namespace zhdev
