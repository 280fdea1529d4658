// zh_cm.hip — two-wave decode kernel for models that are ONE direct context model whose HCOMP is
// "a<<= K  *d=a  halt" with K >= 9 (n == 1, component CM): BASELINE configs 1-2 ("L1").
//
// One workgroup of two wavefronts owns one block.  What makes this path MI355X-shaped:
//
//  * The CM table slice a byte can touch is one 512-entry "window": the context index is
//    h[0] ^ hmap4 with hmap4 < 512 (Predictor.cs:263-266, :463-474).  With K >= 9 the low 9 bits of
//    h[0] are zero, so a byte uses group 0 of the window for its first nibble and one of the groups
//    16..31 for its second: 272 of the 512 entries.  Those 272 entries of up to 64 windows are
//    cached in LDS (fully associative: one tag per lane, lookup = one v_cmp + s_ff1; LRU), each
//    together with its ready-made 16-bit decoder probability squash(stretch(cm >> 17)) * 2 + 1.
//    For text-like data every context window stays in LDS for the whole block, so HBM sees only the
//    stream and the plaintext.  Victims are written back / windows loaded with 16-byte accesses.
//  * WAVE A runs nothing but the arithmetic decoder (Decoder.decode, Decoder.cs:136-158) against
//    that probability cache; its byte loop is hand-written assembly (zh_cm_fast.h).  Within a byte
//    each bit uses a DIFFERENT entry, so all probabilities the byte can need are fetched at once
//    (lane j <- first-nibble node j; lane (q, j) <- node j of second-nibble groups q, q+4, q+8, q+12)
//    and the bit-serial decoder selects with v_readlane.
//  * WAVE B, on its own SIMD of the same CU, receives each decoded byte through an LDS ring, trains
//    the 8 entries it visited (Predictor.train, Predictor.cs:486-493 / :1031-1036) with 8 lanes,
//    refreshes their cached probabilities and writes the plaintext (dwords packed on the scalar
//    unit, parked in a VGPR, one coalesced 256-byte store per 256 bytes).  It also swaps windows.
//  * The only true dependency — a byte whose window was touched by a byte B has not finished — is
//    tracked per window slot; the same per-slot stamp is the LRU clock.
//  * Compressed bytes come from a 256-byte register buffer (one aligned dword per lane).
//
// The first bytes of a segment (post-processor header) and PCOMP programs run on wave A alone
// through the scalar core shared with the generic kernel, on the same window cache.  Single-CM
// models of any other shape are decoded by zh_chain.hip / zh_generic.hip (zh_framing.cpp decides).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "zh_core.h"
#include "zh_dev.h"
#include "zh_cm_fast.h"
#include "zh_model.h"

using namespace zhcore;
using namespace zhdev;

// LDS byte offsets are turned into address_space(3) pointers (32-bit on the device); the host pass of hipcc, which
// never runs this code, sees 64-bit pointers there and would warn.
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"

namespace {

constexpr uint32_t kNoWin = 0xFFFFFFFFu;
constexpr uint32_t kRing = 16;            // A -> B message ring (entries)
enum : uint32_t { kMsgByte = 0, kMsgLeave = 2 };
enum : uint32_t { kCmdEnter = 1, kCmdExit = 3 };
constexpr uint32_t kSpinSection = 1u << 27;   // bounded waits: nothing may hang the GPU
constexpr uint64_t kSpinIdle = 1ull << 33;
#define ZH_E_HELPER (-24)                  // = ZPAQHIP_E_HIP: the helper wavefront stopped answering (cannot happen by design)

// The model-independent tables (read-only after the kernel's one barrier) are shared by every block of the workgroup;
// everything else belongs to ONE block: its window cache, its mailboxes, its cold machine state.  A workgroup holds one
// block (NW = 64 windows, two wavefronts: zh_decode_cm) or — when a launch has more blocks than the GPU has CUs — two
// (NW = 32 windows each, four wavefronts: zh_decode_cm_x2), so that an archive of many blocks fills more than one SIMD
// pair of every CU.
struct alignas(64) CmTabs {
  int16_t sh[16384];                      // stretch(x) for x in [16384, 32768); stretch(x) = -stretch(32767 - x) below
  uint16_t sq[4096];                      // squash
  int32_t dt[1024];
};
template <uint32_t NW>
struct alignas(64) CmBlkT {
  static constexpr uint32_t kNW = NW;
  uint32_t winB[NW][256];                 // CM entries of groups 16..31 of the resident windows (second nibble)
  uint32_t winA[NW][16];                  // CM entries of group 0 (first nibble)
  uint16_t p16B[NW][256];                 // predict()*2+1 of every winB entry, order p16b_pos; kept current by wave B
  uint16_t p16A[NW][16];                  // same for winA
  alignas(64) uint32_t ring[kRing];       // A -> B messages: tag(7) | type(2) | byte(8) | 0(9) | slot(6); 64-byte aligned (zh_cm_fast.h steps the address with v_bfi)
  uint32_t dummy[64];                     // where the lanes of wave A other than lane 0 put their copy of a message (no exec switch)
  uint32_t aux[kRing][2];                 // MISS: new window, victim window
  uint32_t tags[64];                      // ENTER: slot s was trained by wave A alone since B last saw it (its p16 is stale)
  uint4 wsink[64];                        // where lanes that hold no part of a window put their LDS write during a swap (no exec switch)
  // wave A -> swap wave (wave C): mold = the victim's window (kNoWin: the slot was empty), then mreq = n(8) << 23 | window(16) << 6
  // | slot(6) (bit 22: the window id is in aux[0][0]); wave C -> wave A: mdone = n of the last swap installed.  mcfg counts the
  // sections (C re-reads the table's place when it changes); ~0 = leave the kernel
  alignas(8) uint32_t mreq;               // (8-byte aligned: wave C reads {mreq, mold} as one word pair; zh_cm_fast.h addresses mold / mdone as mreq + 4 / + 8)
  uint32_t mold, mdone, mcfg;
  uint32_t t0, b_seq;                     // message count at section start / messages completed by B
  uint32_t cmd_seq, cmd_code, cmd_ack;    // A -> B commands outside a section
  uint32_t limit, ob_word, ob_room;
  uint64_t table, ob_base, ob_cap, ob_len, ob_stored;
  uint32_t table_bytes;                   // size of the CM table (buffer descriptor of wave B's window swaps)
  uint32_t pr[256];                       // PCOMP R
  Vm pz;                                  // cold machine state lives here, not in registers
  Sink sink;                              // output of a PCOMP program
};
template <uint32_t NW, uint32_t NP>
struct alignas(64) CmLdsT {
  CmTabs T;
  CmBlkT<NW> B[NP];
};
static_assert(sizeof(CmLdsT<64, 1>) <= 163840 && sizeof(CmLdsT<32, 2>) <= 163840, "LDS budget");
static_assert(offsetof(CmBlkT<64>, mold) == offsetof(CmBlkT<64>, mreq) + 4 && offsetof(CmBlkT<64>, mdone) == offsetof(CmBlkT<64>, mreq) + 8 && offsetof(CmBlkT<64>, mreq) % 8 == 0, "swap mailbox layout");

__device__ __forceinline__ uint32_t lds_ld(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(uint32_t *p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// One dword into LDS from lane 0 only, without the compiler's exec-mask dance.  The calling wave
// runs with all 64 lanes enabled (uniform code), so exec is restored to all ones.
__device__ __forceinline__ void lds_put0(const uint32_t *where, uint32_t val) {
  const uint32_t addr = (uint32_t)(uintptr_t)where;        // low half of a generic LDS pointer = LDS offset
  asm volatile("s_mov_b64 exec, 1\n\tds_write_b32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(addr), "v"(val) : "memory");
}
// The LDS unit serves one CU's requests in arrival order and a wave issues its LDS instructions
// in program order, so "write data, then write flag" / "read flag, then read data" need no
// s_waitcnt between them: only the compiler must not reorder.
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }
// Typed LDS accesses by LDS byte offset (keeps ds_* with folded address arithmetic).
typedef __attribute__((address_space(3))) uint16_t *lds_u16_p;
typedef __attribute__((address_space(3))) uint32_t *lds_u32_p;
typedef __attribute__((address_space(3))) const uint64_t *lds_u64_p;
__device__ __forceinline__ uint32_t lds_u16(uint32_t off) { return *(lds_u16_p)off; }
__device__ __forceinline__ uint64_t lds_u64(uint32_t off) { return *(lds_u64_p)off; }
__device__ __forceinline__ uint32_t lds_off(const void *p) { return (uint32_t)(uintptr_t)p; }
// Where the probability of second-nibble entry (group 16 + q, position P) is kept inside p16B[slot][]:
// quad q & 3, position, element q >> 2 — so that the four candidates of a lane of wave A (groups
// q, q+4, q+8, q+12, same position) are adjacent: one ds_read_b64.
__device__ __forceinline__ uint32_t p16b_pos(uint32_t q, uint32_t P) { return ((q & 3) << 6) | (P << 2) | (q >> 2); }
__device__ __forceinline__ uint32_t ring_tag(uint32_t u) { return u & 127u; }   // differs between the messages u, u + 16, ... u + 112 that share a ring entry
static_assert(kRing == 16, "ring_tag");
// single-wave replacement of __syncthreads(): wave A must never wait on a workgroup barrier (wave B idles in a mailbox loop)
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// predict()*2+1 for a CM entry: squash(stretch(cm >> 17)) (Predictor.cs:263-266, :349) through the half stretch table
__device__ __forceinline__ uint32_t p16_of(const CmTabs &T, uint32_t cm) {
  const uint32_t xv = cm >> 17;
  const int st = xv >= 16384 ? (int)T.sh[xv - 16384] : -(int)T.sh[16383 - xv];
  return (uint32_t)T.sq[st + 2048] * 2 + 1;
}

// Window <-> table.  A window is 512 entries = 2 KiB of the reference table; this kernel only ever touches group 0
// (uint4 0..3) and groups 16..31 (uint4 64..127) of it, so the 960 bytes in between are free: the HBM copy of a window
// carries its probability cache there (p16A at uint4 4..5, p16B at uint4 8..39).  A window that comes back from HBM is
// then usable at once — the miss path has no table walk (stretch, squash) for its 272 entries.
template <class BLK>
__device__ __forceinline__ void win_store(const BLK &S, uint32_t slot, uint32_t *table, uint32_t w, uint32_t lane) {
  uint4 *g = reinterpret_cast<uint4 *>(table + (uint64_t)w * 512);
  g[64 + lane] = reinterpret_cast<const uint4 *>(&S.winB[slot][0])[lane];
  if (lane < 4) g[lane] = reinterpret_cast<const uint4 *>(&S.winA[slot][0])[lane];
  if (lane < 32) g[8 + lane] = reinterpret_cast<const uint4 *>(&S.p16B[slot][0])[lane];
  if (lane >= 32 && lane < 34) g[4 + lane - 32] = reinterpret_cast<const uint4 *>(&S.p16A[slot][0])[lane - 32];
}
template <class BLK>
__device__ __forceinline__ void win_load(BLK &S, uint32_t slot, const uint32_t *table, uint32_t w, uint32_t lane) {
  const uint4 *g = reinterpret_cast<const uint4 *>(table + (uint64_t)w * 512);
  const uint4 b = g[64 + lane];
  uint4 a = make_uint4(0, 0, 0, 0), p = a;
  if (lane < 4) a = g[lane];
  if (lane < 32) p = g[8 + lane];
  if (lane >= 32 && lane < 34) p = g[4 + lane - 32];
  reinterpret_cast<uint4 *>(&S.winB[slot][0])[lane] = b;
  if (lane < 4) reinterpret_cast<uint4 *>(&S.winA[slot][0])[lane] = a;
  if (lane < 32) reinterpret_cast<uint4 *>(&S.p16B[slot][0])[lane] = p;
  if (lane >= 32 && lane < 34) reinterpret_cast<uint4 *>(&S.p16A[slot][0])[lane - 32] = p;
}
// probability cache of one slot from its entries (whole wave)
template <class BLK>
__device__ __forceinline__ void p16_rebuild(const CmTabs &T, BLK &S, uint32_t slot, uint32_t lane) {
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) {
    const uint32_t e = lane + 64 * k;
    S.p16B[slot][p16b_pos(e >> 4, e & 15)] = (uint16_t)p16_of(T, S.winB[slot][e]);
  }
  if (lane < 16) S.p16A[slot][lane] = (uint16_t)p16_of(T, S.winA[slot][lane]);
}
// Replacement (round 4): the first slot whose last use is older than message `thr` — empty slots carry 0, lanes that stand
// for no slot ~0 and are never taken; when no slot is that old the threshold moves up to 31 (24 with 32 slots) messages ago.  A handful of instructions where true LRU needed a wave-wide maximum; on the
// x86-like generator it misses as rarely as LRU (20.7 % against 20.5 %; first-in-first-out: 27 %).  zh_cm_fast.h does the
// same test inline and leaves to the C++ body only when the threshold has to move.
template <uint32_t NW> constexpr uint32_t kVictimBack = NW >= 40 ? 31u : NW - 8u;
template <uint32_t NW>
__device__ __forceinline__ uint32_t pick_victim(uint32_t lastuse, uint32_t now, uint32_t &thr) {
  // every byte stamps one slot with its own message number, so at most kBack slots (and one being swapped in) carry a stamp
  // of the last kBack messages: with kBack < NW - 1 a victim exists; kBack > 13 keeps it clear of what wave B may still owe
  constexpr uint32_t kBack = kVictimBack<NW>;
  static_assert(kBack > 13 && kBack + 1 < NW, "replacement threshold");
  uint64_t old = __ballot(lastuse < thr);
  if (UNLIKELY(old == 0)) { thr = now > kBack ? now - kBack : 1u; old = __ballot(lastuse < thr); }
  return (uint32_t)__builtin_ctzll(old | 1ull << 63);      // (never empty; the guard keeps a lost invariant from indexing past the slots)
}

// ---------------------------------------------------------------------------------------
// Wave B: model trainer / probability cache / plaintext writer of the steady state.
// ---------------------------------------------------------------------------------------
template <bool PROF, class BLK>
__device__ void helper_wave(const ZhLaunch &L, const CmTabs &T, BLK &S, uint32_t lane) {
  constexpr uint32_t kWin = BLK::kNW;
  uint64_t busy = 0, tb0 = 0, tb1 = 0, idle = 0;
  const uint32_t l15 = lane & 15;
  const uint32_t ltt = 31 - __clz((int)(l15 | 1));
  const uint32_t lsh_vis = 4 - ltt, lsh_y = 3 - ltt;
  const uint32_t offA = lds_off(&S.winA[0][0]) + l15 * 4, offB = lds_off(&S.winB[0][0]) + l15 * 4;
  const uint32_t poffA = lds_off(&S.p16A[0][0]) + l15 * 2, poffB = lds_off(&S.p16B[0][0]);
  uint32_t seen = 0;
  for (;;) {
    uint64_t spin = 0;
    uint32_t cs;
    while ((cs = lds_ld(&S.cmd_seq)) == seen) {            // idle between sections
      __builtin_amdgcn_s_sleep(8);
      if (++spin > kSpinIdle) return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    seen = cs;
    if (uni(lds_ld(&S.cmd_code)) == kCmdExit) return;

    // ---- ENTER: take over the window cache and the output
    uint32_t *table = reinterpret_cast<uint32_t *>(L.arena + uni64(S.table));   // offsets, so that accesses stay global_*
    const uint32_t limit = uni(S.limit);
    OutBuf ob;
    ob.base = L.out + uni64(S.ob_base); ob.cap = uni64(S.ob_cap); ob.len = uni64(S.ob_len);
    ob.stored = uni64(S.ob_stored); ob.word = uni(S.ob_word); ob.room = uni(S.ob_room); ob.park = 0;
    for (uint32_t sl = 0; sl < kWin; ++sl) {               // usually one slot (the byte that names the post-processor)
      if (uni(S.tags[sl]) == 0) continue;
      p16_rebuild(T, S, sl, lane);
    }
    uint32_t u = uni(S.t0), sp = 0;                        // next message to take
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    lds_st(&S.cmd_ack, cs);

    // A byte message is taken to its end at once: entry -> dt -> train -> write, then stretch -> squash of the trained entry ->
    // cached probability -> report.  (Rounds 1-3 ran the second half together with the NEXT message's first half, to overlap
    // their LDS round trips; completion of a byte then waited for the message behind it, wave B was two messages behind
    // wave A as a rule, and round 4's stamps found A waiting for B on every byte whose window one of the last two bytes had
    // used — 21 % of the bytes of text, ~560 cycles each, profiles/r04/stages_l1_miss.txt.  B has the time: it was busy
    // ~580 of A's ~1 070 cycles per byte.)  Masked writes go through a per-lane address (unvisited lanes: a sink word):
    // no exec-mask branches on the path.
    bool leave = false;
    auto stretch_idx = [&](uint32_t nv) { const uint32_t xv = nv >> 17; return xv >= 16384 ? xv - 16384 : 16383 - xv; };
    const bool second = (lane & 16) != 0;                  // lanes 0-15 (and their copies 32-47): first nibble (group 0); 16-31 (48-63): second nibble, group 16 + n1
    const uint32_t l_sinkw = lds_off(&S.wsink[lane]);
    while (!leave) {
      const uint32_t m0 = uni(lds_ld(&S.ring[u & (kRing - 1)]));
      const bool have = (m0 >> 25) == ring_tag(u);
      const uint32_t type = (m0 >> 23) & 3, slot = m0 & 63;
      lds_order();
      if (LIKELY(have && type == kMsgByte)) {
        if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb0)::"memory"); }
        sp = 0;
        const uint32_t c = (m0 >> 15) & 255, n1 = c >> 4, n2 = c & 15;
        const uint32_t nib = second ? n2 : n1;
        const bool vis = lane < 32 && l15 != 0 && l15 == ((16 | nib) >> lsh_vis);
        const uint32_t eoff = second ? offB + slot * 1024 + n1 * 64 : offA + slot * 64;
        const uint32_t p_off = second ? poffB + slot * 512 + p16b_pos(n1, l15) * 2 : poffA + slot * 32;
        // every lane reads (harmless for the unvisited ones); only the writes are masked
        const uint32_t cm = *(lds_u32_p)eoff;
        const uint32_t cnt = cm & 0x3ff;
        const int dtv = T.dt[cnt];
        const uint32_t yy = (nib >> lsh_y) & 1;
        const int err = (int)(yy * 32767) - (int)(cm >> 17);            // Predictor.train (Predictor.cs:1031-1036)
        const uint32_t nv = cm + (((uint32_t)err * (uint32_t)dtv) & 0xFFFFFC00u) + (cnt < limit);
        *(lds_u32_p)(vis ? eoff : l_sinkw) = nv;
        const int shv = T.sh[stretch_idx(nv)];
        const uint32_t sqv = T.sq[((nv >> 17) >= 16384 ? shv : -shv) + 2048];
        *(lds_u16_p)(vis ? p_off : l_sinkw) = (uint16_t)(sqv * 2 + 1);
        ++u;
        lds_order();
        lds_put0(&S.b_seq, u);                                          // every message before u is complete
        out_put(ob, c, lane);                              // PostProcessor PASS: the byte is the plaintext
        if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb1)::"memory"); busy += tb1 - tb0; }
      } else if (!have) {
        if (PROF) ++idle;
        if (++sp > kSpinSection) return;
      } else {                                             // LEAVE: hand the output state back
        out_flush(ob, lane);
        if (PROF && lane == 0 && L.debug) {
          atomicAdd((unsigned long long *)&L.debug[5], (unsigned long long)busy);
          atomicAdd((unsigned long long *)&L.debug[7], (unsigned long long)idle);
        }
        busy = 0; idle = 0;
        if (lane == 0) { S.ob_len = ob.len; S.ob_stored = ob.stored; S.ob_word = ob.word; S.ob_room = ob.room; }
        ++u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        lds_st(&S.b_seq, u);
        leave = true;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Wave C: window swaps (round 4).  A miss used to travel through wave B's message ring, behind the byte before it, and
// wave A then waited for B to train that byte AND swap the window: ~3 000 cycles, of which the memory round trip was ~130
// (profiles/r04/stages_l1_miss_before.txt).  Nothing in a swap depends on B: the victim is a slot nobody has used for more
// than 31 messages (B is never more than 13 behind), the new window arrives with its probability cache (win_store).  So a
// third wavefront does nothing but swaps: wave A hands it slot, window and victim through three LDS words and goes on as
// soon as the window is installed, while B is still training the byte before.
// Buffer accesses with one offset per lane and the window's place as the scalar offset — lanes that hold no part of a
// window carry an offset beyond the table and are dropped, their LDS writes go to a sink: no exec-mask code, no 64-bit
// address arithmetic.  Stateless between requests (the directory stays with wave A).
// ---------------------------------------------------------------------------------------
template <bool PROF, class BLK>
__device__ void swap_wave(const ZhLaunch &L, BLK &S, uint32_t lane) {
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) v4u *lds_v4_p;
  constexpr uint32_t kDrop = 0x80000000u;
  const uint32_t g_offB = (64u + lane) * 16u, g_offA = lane < 4 ? lane * 16u : kDrop,
                 g_offP = lane < 32 ? (8u + lane) * 16u : lane < 34 ? (4u + lane - 32u) * 16u : kDrop;
  const uint32_t l_sink = lds_off(&S.wsink[lane]);
  const uint32_t l_cB = lds_off(&S.winB[0][0]) + lane * 16u;
  const uint32_t l_cA = lane < 4 ? lds_off(&S.winA[0][0]) + lane * 16u : l_sink, l_mA = lane < 4 ? 64u : 0u;
  const uint32_t l_cP = lane < 32 ? lds_off(&S.p16B[0][0]) + lane * 16u : lane < 34 ? lds_off(&S.p16A[0][0]) + (lane - 32u) * 16u : l_sink,
                 l_mP = lane < 32 ? 512u : lane < 34 ? 32u : 0u;
  uint64_t t_req = 0, t_out = 0, t_back = 0, t_done = 0, c_a = 0, c_b = 0, c_c = 0, n_sw = 0;
  __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void *)L.arena, 0, 0, 0x00020000);
  uint32_t cfg = 0, done = 0;
  uint64_t idle = 0;
  for (;;) {
    // {mreq, mold} in one 8-byte read (wave A writes mold first and LDS serves a wave's requests in order, so a new request
    // word comes with its victim), the section counter beside it: one LDS round trip per poll
    const uint64_t rq = __hip_atomic_load(reinterpret_cast<const uint64_t *>(&S.mreq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (an atomic load: a plain one would be hoisted out of the poll)
    const uint32_t cfg_now = lds_ld(&S.mcfg);
    const uint32_t r = uni((uint32_t)rq);
    const uint32_t n = (r >> 23) & 255u;
    if (n == done) {                                       // nothing to do: is the kernel over?
      const uint32_t c = uni(cfg_now);
      if (c == ~0u) break;
      if (++idle > 64) __builtin_amdgcn_s_sleep(2);        // (a miss-heavy stream keeps this wave awake)
      if (idle > kSpinIdle) break;
      continue;
    }
    idle = 0;
    if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_req)::"memory"); }
    const uint32_t c = uni(cfg_now);
    if (UNLIKELY(c != cfg)) {                              // a new section (or block): where its table is
      cfg = c;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      trs = __builtin_amdgcn_make_buffer_rsrc((void *)(L.arena + uni64(S.table)), 0, (int)uni(S.table_bytes), 0x00020000);
    }
    const uint32_t sl = r & 63u;
    const uint32_t neww = (r >> 22) & 1u ? uni(S.aux[0][0]) : (r >> 6) & 0xFFFFu;
    const uint32_t oldw = uni((uint32_t)(rq >> 32));
    const uint32_t aB = l_cB + sl * 1024u, aA = l_cA + sl * l_mA, aP = l_cP + sl * l_mP;
    const uint32_t so_new = neww * 2048u;
    const v4u nb = __builtin_amdgcn_raw_buffer_load_b128(trs, g_offB, so_new, 0);   // its probability cache travels with it (win_store)
    const v4u na = __builtin_amdgcn_raw_buffer_load_b128(trs, g_offA, so_new, 0);
    const v4u np = __builtin_amdgcn_raw_buffer_load_b128(trs, g_offP, so_new, 0);
    if (oldw != kNoWin) {                                  // the victim goes back while the new window travels
      const v4u vb = *(lds_v4_p)aB, va = *(lds_v4_p)aA, vp = *(lds_v4_p)aP;
      const uint32_t so_old = oldw * 2048u;
      __builtin_amdgcn_raw_buffer_store_b128(vb, trs, g_offB, so_old, 0);
      __builtin_amdgcn_raw_buffer_store_b128(va, trs, g_offA, so_old, 0);
      __builtin_amdgcn_raw_buffer_store_b128(vp, trs, g_offP, so_old, 0);
    }
    if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_out)::"memory"); asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_back)::"memory"); }
    *(lds_v4_p)aB = nb;
    *(lds_v4_p)aA = na;
    *(lds_v4_p)aP = np;
    // LDS data, then LDS flag, in program order: no fence — a release fence would also wait for the victim's write-back, a
    // store round trip wave A has no reason to sit through (only this wave ever reads that window back, and its own memory
    // operations stay in order)
    lds_order();
    lds_put0(&S.mdone, n);
    done = n;
    if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_done)::"memory");
                c_a += t_out - t_req; c_b += t_back - t_out; c_c += t_done - t_back; ++n_sw; }
  }
  if (PROF && lane == 0 && L.debug) {
    atomicAdd((unsigned long long *)&L.debug[13], (unsigned long long)c_a);
    atomicAdd((unsigned long long *)&L.debug[14], (unsigned long long)c_b);
    atomicAdd((unsigned long long *)&L.debug[15], (unsigned long long)c_c);
    atomicAdd((unsigned long long *)&L.debug[9], (unsigned long long)n_sw);
  }
}

template <bool PROF, uint32_t NW, uint32_t NP>
__device__ __forceinline__ void decode_cm_body(const ZhLaunch &L, CmLdsT<NW, NP> &SS) {
  uint64_t prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tprev = 0, t_exit = 0, t_pub = 0;           // PROF: stamps of the miss path (asm loop left, miss published)
  bool was_miss = false;
  // a block = three wavefronts: A (decoder), B (model, probability cache, output), C (window swaps)
  // Two blocks per workgroup: wavefronts in the order A0 A1 B0 B1 C0 C1.  A CU hands its wavefronts to its four SIMDs in turn, so
  // each decoder wave shares its SIMD only with its own swap wave (asleep unless that decoder waits for it), the two model waves
  // have a SIMD each.  (Round 4, first form: A0 B0 C0 A1 B1 C1 — decoder 0 shared its SIMD with model wave 1, decoder 1 had one
  // to itself, and the launch waited for decoder 0.)
  const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6, pair = NP > 1 ? wid % NP : 0u, wave = NP > 1 ? wid / NP : wid;
  CmTabs &T = SS.T;
  CmBlkT<NW> &S = SS.B[pair];

  {  // model-independent tables -> LDS (both waves)
    const uint4 *s0 = reinterpret_cast<const uint4 *>(L.tables->stretch + 16384);
    uint4 *d0 = reinterpret_cast<uint4 *>(T.sh);
    for (uint32_t i = threadIdx.x; i < sizeof(T.sh) / 16; i += 192 * NP) d0[i] = s0[i];
    const uint4 *s1 = reinterpret_cast<const uint4 *>(L.tables->squash);
    uint4 *d1 = reinterpret_cast<uint4 *>(T.sq);
    for (uint32_t i = threadIdx.x; i < sizeof(T.sq) / 16; i += 192 * NP) d1[i] = s1[i];
    const uint4 *s2 = reinterpret_cast<const uint4 *>(L.tables->dt);
    uint4 *d2 = reinterpret_cast<uint4 *>(T.dt);
    for (uint32_t i = threadIdx.x; i < sizeof(T.dt) / 16; i += 192 * NP) d2[i] = s2[i];
    if (wave == 0 && lane == 0) { S.cmd_seq = 0; S.cmd_code = 0; S.cmd_ack = 0; S.t0 = 0; S.b_seq = 0; S.mreq = 0; S.mold = 0; S.mdone = 0; S.mcfg = 0; }
  }
  __syncthreads();                                       // the only workgroup barrier of the kernel
  if (wave == 1) { helper_wave<PROF>(L, T, S, lane); return; }
  if (wave == 2) { swap_wave<PROF>(L, S, lane); return; }
  uint32_t n_miss = 0, sec_cfg = 0;                      // swap requests made to wave C so far / sections entered (S.mcfg)
  uint32_t cmd_seq = 0;                                  // commands issued to wave B so far

  // per-lane constants of the lane <-> table-entry mapping
  const uint32_t l15 = lane & 15;                      // nibble context j held by this lane
  const uint32_t lgrp = lane >> 4;                     // 16-lane group
  const uint32_t ltt = 31 - __clz((int)(l15 | 1));     // depth of context j in the nibble tree (0..3)
  const uint32_t lsh_vis = 4 - ltt, lsh_y = 3 - ltt;
  // LDS offsets of what this lane reads per byte: probability of first-nibble node l15 (+ slot * 32) and the
  // four second-nibble candidates of quad lgrp (+ slot * 512)
  const uint32_t p_la = lds_off(&S.p16A[0][0]) + l15 * 2, p_lb = lds_off(&S.p16B[0][0]) + lgrp * 128 + l15 * 8;
  const uint32_t ring_addr = lds_off(&S.ring[0]), bseq_addr = lds_off(&S.b_seq), mreq_addr = lds_off(&S.mreq);   // (mold, mdone follow mreq)
  const uint32_t dummy_addr = lds_off(&S.dummy[lane]), ring_step = lane == 0 ? 63u : 0u;   // see ZH_FAST_EPILOGUE

  uint8_t *slot_mem = L.arena + (uint64_t)(blockIdx.x * NP + pair) * L.arena_stride;   // the host sizes the arena for grid x NP slots

  for (;;) {
    uint32_t bi = 0;
    if (lane == 0) bi = atomicAdd(L.queue, 1u);
    bi = uni((uint32_t)__shfl((int)bi, 0));
    if (bi >= L.n_blocks) break;                       // every wave reaches this exit

    const ZhBlockDesc *bdp = &L.blocks[bi];
    const uint32_t model_i = uni(bdp->model);
    const uint32_t first_seg = uni(bdp->first_seg), n_seg = uni(bdp->n_seg);
    const uint64_t b_out_off = uni64(bdp->out_off), b_out_cap = uni64(bdp->out_cap);
    const ZhModel *M = &L.models[model_i];
    const ZhComp *cp = &M->comp[0];
    const uint32_t cm_mask = uni(cp->cm_mask);
    const uint32_t limit = (uint32_t)uni(cp->arg[1]) * 4;
    const uint64_t cm_off = uni64(cp->cm_off), cm_bytes = uni64(cp->cm_bytes);
    const uint32_t win_bfe = 9u | ((uint32_t)__builtin_popcount(cm_mask) - 9u) << 16;   // s_bfe operand: window id = bits 9.. of h[0] & cm_mask
    const uint32_t hshift = (uni(M->kind) >> 16) & 255;    // HCOMP "a<<= K  *d=a  halt": 9 <= K <= 31 (zh_framing.cpp)
    uint32_t *table = reinterpret_cast<uint32_t *>(slot_mem + cm_off);

    // Predictor.init: CM table = 0x80000000 (Predictor.cs:103-104); VM memories zeroed.
    {
      const uint4 v = make_uint4(0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u);
      uint4 *q = reinterpret_cast<uint4 *>(table);
      for (uint64_t i = lane; i < cm_bytes / 16; i += 64) q[i] = v;
      // ... and the probability cache every window carries in HBM (win_store): predict()*2+1 of a fresh entry
      const uint32_t pv = p16_of(T, 0x80000000u), pp = pv | pv << 16;
      const uint4 ppat = make_uint4(pp, pp, pp, pp);
      const uint64_t nwin = cm_bytes / 2048;
      for (uint64_t i = lane; i < nwin * 34; i += 64) {
        const uint64_t w = i / 34, k = i % 34;
        q[w * 128 + (k < 32 ? 8 + k : 4 + (k - 32))] = ppat;
      }
      const uint64_t h_off = uni64(M->h_off), tail = uni64(M->arena_bytes) - h_off;
      uint4 *z = reinterpret_cast<uint4 *>(slot_mem + h_off);
      for (uint64_t i = lane; i < tail / 16; i += 64) z[i] = make_uint4(0, 0, 0, 0);
      for (uint32_t i = lane; i < 256; i += 64) S.pr[i] = 0;
    }
    wave_sync();

    uint32_t tag = kNoWin;                             // per-lane window directory: lane s = slot s
    uint32_t stale = 0;                                // per-lane: slot s was trained outside a two-wave section (p16 not current)
    uint32_t lastuse = lane < NW ? 0u : ~0u;           // per-lane: value of t after the last byte / message that used slot s (lanes beyond the slots: never a victim)
    uint32_t thr = 1;                                  // replacement threshold (pick_victim)
    uint32_t t = 0;                                    // bytes decoded + window swaps so far = messages published to wave B
    uint32_t h0 = 0;                                   // h[0] = z.H(0)

    int pp_state = 0, pp_hsize = 0;                    // PostProcessor (PostProcessor.cs:12-16)
    uint32_t pp_len = 0;
    Vm &pz = S.pz;
    pz.a = pz.b = pz.c = pz.d = pz.f = 0;
    pz.prog = nullptr; pz.len = 0;
    pz.m = slot_mem + uni64(M->pm_off); pz.mmask = (uint32_t)((1ull << uni(M->pm)) - 1);
    pz.h = reinterpret_cast<uint32_t *>(slot_mem + uni64(M->ph_off)); pz.hmask = (uint32_t)((1ull << uni(M->ph)) - 1);
    pz.r = S.pr;
    uint8_t *pzbuf = slot_mem + uni64(M->pz_off) + ZH_CODE_PAD;

    Dec d;
    d.low = 1; d.high = 0xFFFFFFFFu; d.curr = 0;

    OutBuf ob;
    ob.base = L.out + b_out_off; ob.cap = b_out_cap; ob.len = 0; ob.stored = 0; ob.word = 0; ob.park = 0;
    out_room(ob);
    Sink &sink = S.sink;                               // used only when a PCOMP program emits output
    sink.out = ob.base; sink.cap = ob.cap; sink.len = 0;

    wave_sync();
    InBuf in;
    in.stream = L.in; in.total = L.in_total; in.cbase = 0; in.k = 0; in.avail = 0; in.cur = 0;

    int failed = 0;
    for (uint32_t s = 0; s < n_seg; ++s) {
      const uint32_t si = first_seg + s;
      const uint64_t produced0 = pp_state == 5 ? uni64(sink.len) : ob.len;
      int status = 0;
      if (failed) {
        if (lane == 0) {
          ZhSegResult res;
          res.status = ZH_E_SKIPPED; res.pp_state = (uint32_t)pp_state; res.out_off = b_out_off + produced0; res.out_len = 0;
          res.in_used = 0;
          L.results[si] = res;
        }
        continue;
      }
      const uint64_t seg_off = uni64(L.segs[si].in_off);
      in_seek(in, seg_off, lane);

      // One decoded byte on wave A alone: Decoder.decompress() (Decoder.cs:32-56) with predict/update folded in.
      // Returns 0..255, -1 at EOS, -2 on error (status set).
      auto decode_byte = [&]() __attribute__((always_inline)) -> int {
        if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }
        // ---- Decoder.decompress prologue (Decoder.cs:36-45)
        if (UNLIKELY(d.curr == 0)) {
          uint32_t cu = 0;
          for (int i = 0; i < 4; ++i) cu = cu << 8 | (uint32_t)in_get(in, lane);
          d.curr = uni(cu);
        }
        uint32_t bad = 0, rn, j = 0;
        d.low = uni(d.low); d.high = uni(d.high); d.curr = uni(d.curr);   // folds away where already scalar
        ZH_DEC_STEP(d, 0u, j, bad, rn);                // EOS flag: p = 0
        if (UNLIKELY(bad)) { status = ZH_E_CORRUPT; return -2; }
        if (UNLIKELY(rn)) { if (dec_renorm(d, in, lane)) { status = ZH_E_EOF; return -2; } }
        int c;
        if (UNLIKELY(j)) {
          if (d.curr != 0) { status = ZH_E_EOS; return -2; }
          c = -1;
        } else {
          ZH_STAMP(0);
          // ---- probabilities for every context this byte can reach
          const uint32_t w = (h0 & cm_mask) >> 9;
          const uint64_t hit = __ballot(tag == w);
          uint32_t slot;
          if (LIKELY(hit != 0)) slot = (uint32_t)__builtin_ctzll(hit);
          else {                                        // swap the window in (wave A alone: nothing is in flight)
            slot = uni(pick_victim<NW>(lastuse, t, thr));
            const uint32_t old = rdlane(tag, slot);
            if (old != kNoWin) {
              if (rdlane(stale, slot)) { p16_rebuild(T, S, slot, lane); wave_sync(); }   // trained here, cache not refreshed yet
              win_store(S, slot, table, old, lane);
            }
            win_load(S, slot, table, w, lane);
            tag = lane == slot ? w : tag;
            wave_sync();
          }
          ++t;
          lastuse = lane == slot ? t : lastuse;
          stale = lane == slot ? 1u : stale;
          ZH_STAMP(1);
          // lane j of any quad: first-nibble node j; lane (q, j): node j of second-nibble groups q, q+4, q+8, q+12
          uint32_t *wa = &S.winA[slot][0], *wb = &S.winB[slot][0];
          const uint32_t ib0 = (lgrp << 4) | l15;       // + 64*k entries for k = 0..3
          const uint32_t cma = wa[l15];
          uint32_t cmb[4], pb[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) cmb[k] = wb[ib0 + 64 * k];
          const uint32_t pa = p16_of(T, cma) << 16;
#pragma unroll
          for (int k = 0; k < 4; ++k) pb[k] = p16_of(T, cmb[k]) << 16;
          ZH_STAMP(2);

          // ---- first nibble: context j lives in lane j
          // A failed renormalisation (end of stream) is recorded, not branched on: the coder
          // keeps running on garbage for at most a nibble; the FIRST error is what is reported.
          uint32_t err = 0;
          j = 1;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t ps = rdlane(pa, j);
            ZH_DEC_STEP(d, ps, j, bad, rn);
            if (UNLIKELY(rn)) { if (dec_renorm(d, in, lane) && !err) err = bad ? (uint32_t)-ZH_E_CORRUPT : (uint32_t)-ZH_E_EOF; }
          }
          if (UNLIKELY(err | bad)) { status = err ? -(int)err : ZH_E_CORRUPT; return -2; }
          ZH_STAMP(3);
          const uint32_t n1 = j & 15;
          // ---- second nibble: group 16 + n1 is held by quad (n1 & 3), register n1 >> 2
          const uint32_t ga = uni(n1), kb = ga >> 2, lb = (ga & 3) * 16;
          const uint32_t psel = kb == 0 ? pb[0] : kb == 1 ? pb[1] : kb == 2 ? pb[2] : pb[3];
          uint32_t j2 = 1;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t ps = rdlane(psel, lb + j2);
            ZH_DEC_STEP(d, ps, j2, bad, rn);
            if (UNLIKELY(rn)) { if (dec_renorm(d, in, lane) && !err) err = bad ? (uint32_t)-ZH_E_CORRUPT : (uint32_t)-ZH_E_EOF; }
          }
          if (UNLIKELY(err | bad)) { status = err ? -(int)err : ZH_E_CORRUPT; return -2; }
          ZH_STAMP(4);
          const uint32_t n2 = j2 & 15;
          c = (int)(n1 << 4 | n2);

          // ---- train the 8 visited entries (Predictor.train), one pass over the lanes:
          // quad (n1 & 3) updates the second-nibble entries, quad (n1 & 3) ^ 1 the first-nibble ones.
          {
            const bool isb = lgrp == (ga & 3);
            const bool isa = lgrp == ((ga & 3) ^ 1);
            const uint32_t cmsel = kb == 0 ? cmb[0] : kb == 1 ? cmb[1] : kb == 2 ? cmb[2] : cmb[3];
            const uint32_t cm = isb ? cmsel : cma;
            const uint32_t nib = isb ? n2 : n1;
            const bool vis = (isa || isb) && l15 != 0 && l15 == ((16 | nib) >> lsh_vis);
            const uint32_t yy = (nib >> lsh_y) & 1;
            const uint32_t cnt = cm & 0x3ff;
            const int e = (int)(yy * 32767) - (int)(cm >> 17);
            const uint32_t nv = cm + (((uint32_t)e * (uint32_t)T.dt[cnt]) & 0xFFFFFC00u) + (cnt < limit);
            if (vis) { if (isb) wb[ib0 + 64 * kb] = nv; else wa[l15] = nv; }
          }
          ZH_STAMP(5);
          h0 = (uint32_t)c << hshift;                    // HCOMP "a<<= K  *d=a  halt" (Predictor.cs:464-470)
        }
        return (int)uni((uint32_t)c);
      };

      for (;;) {
        if (LIKELY(pp_state == 1)) {
          // ===== steady state: PASS post-processor (PostProcessor.cs:49-51), two wavefronts =====
          // hand the window cache and the output to wave B
          S.tags[lane] = stale;
          stale = 0;
          if (lane < kRing) S.ring[lane] = ((ring_tag(t) + 64) & 127) << 25;   // never the tag of messages t .. t+15
          out_flush(ob, lane);
          if (lane == 0) {
            S.table = (uint64_t)(reinterpret_cast<uint8_t *>(table) - L.arena); S.limit = limit;
            S.table_bytes = (uint32_t)(cm_bytes > 0xFFFFF000ull ? 0xFFFFF000ull : cm_bytes);
            S.mcfg = ++sec_cfg;                           // wave C: a new section (published with the fence below, before any request)
            S.ob_base = (uint64_t)(ob.base - L.out); S.ob_cap = ob.cap; S.ob_len = ob.len; S.ob_stored = ob.stored;
            S.ob_word = ob.word; S.ob_room = ob.room;
            S.t0 = t; S.b_seq = t; S.cmd_code = kCmdEnter;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          ++cmd_seq;
          lds_st(&S.cmd_seq, cmd_seq);
          bool helper_ok = true;
          {
            uint32_t sp = 0;
            while (lds_ld(&S.cmd_ack) != cmd_seq) { if (++sp > kSpinSection) { helper_ok = false; break; } }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          }
          if (!helper_ok) { status = ZH_E_HELPER; break; }
          uint32_t b_done = t;                            // messages known completed by wave B
          auto publish = [&](uint32_t m0) __attribute__((always_inline)) {
            lds_order();
            lds_put0(&S.ring[t & (kRing - 1)], ring_tag(t) << 25 | m0);
            ++t;
          };
          auto wait_done = [&](uint32_t upto) __attribute__((always_inline)) -> bool {   // until b_seq >= upto
            uint32_t sp = 0;
            while ((int32_t)(b_done - upto) < 0) {
              b_done = uni(lds_ld(&S.b_seq));
              if (++sp > kSpinSection) return false;
            }
            lds_order();
            return true;
          };
          // The loop has ONE exit (the EOS test): an error lets the byte run to its end on whatever
          // state it has, commits nothing, and forces that test; the common path carries no exit bookkeeping.
          enum : uint32_t { kEvEos = 1, kEvCorrupt, kEvEof, kEvHelper };
          uint32_t ev = 0;
          uint32_t tq;
          for (;;) {
            // hand-written loop for the common case (zh_cm_fast.h); the C++ body below is the same
            // algorithm and takes every byte the fast loop declines
            {
              if (in.avail - in.k < 49 && in.avail == 256) in_seek(in, in_pos(in), lane);   // re-centre the chunk (the fast loop wants 48 bytes ahead)
              uint32_t code;
              d.low = uni(d.low); d.high = uni(d.high); d.curr = uni(d.curr); in.k = uni(in.k); in.avail = uni(in.avail);
              t = uni(t); h0 = uni(h0); b_done = uni(b_done); n_miss = uni(n_miss); thr = uni(thr);
              uint64_t f0 = 0, f1 = 0;
              const uint32_t t_in = t;
              if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(f0)::"memory"); }
              // the loop's invariant (an out-of-range state is the C++ body's to report); its lag compare is signed and uses 2^30 as "never"
              if (LIKELY(d.curr - d.low <= d.high - d.low && t < 0x3e000000u)) {
                // a byte consumes at most 9 x 4 coded bytes; the loop tests k < klim in the shadow of a byte's SEVENTH step, ahead of
                // the renormalisations of its last two steps (at most 4 each): 48 ahead then is 40 at the next byte's start
                const uint32_t klim = uni(in.avail >= 48 ? in.avail - 48 : 0u);
                uint32_t vr = lane == 0 ? ring_addr + (t & (kRing - 1)) * 4u : dummy_addr;
                if (PROF) {
                  uint32_t sp_lo = 0, sp_hi = 0, nsp = 0;
                  ZH_CM_FAST_LOOP_PROF(d.low, d.high, d.curr, in.k, t, h0, b_done, lastuse, code, klim, uni(win_bfe), uni(hshift),
                                       vr, ring_step, bseq_addr, in.cur, tag, p_la, p_lb, thr, n_miss, mreq_addr, uni(kVictimBack<NW>), sp_lo, sp_hi, nsp);
                  prof[11] += (uint64_t)sp_hi << 32 | sp_lo; prof[12] += nsp;     // cycles / times wave A waited in the spin
                } else {
                ZH_CM_FAST_LOOP(d.low, d.high, d.curr, in.k, t, h0, b_done, lastuse, code, klim, uni(win_bfe), uni(hshift),
                                vr, ring_step, bseq_addr, in.cur, tag, p_la, p_lb, thr, n_miss, mreq_addr, uni(kVictimBack<NW>));
                }
              } else code = 0;
              if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(f1)::"memory"); prof[2] += f1 - f0; prof[3] += t - t_in; t_exit = f1; }
              if (UNLIKELY(code)) { ev = kEvCorrupt; d.low = d.high = d.curr = 1; }   // the EOS test below ends the section
            }
            if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }
            // ---- Decoder.decompress prologue (Decoder.cs:36-45)
            if (UNLIKELY(d.curr == 0)) {
              uint32_t cu = 0;
              for (int i = 0; i < 4; ++i) cu = cu << 8 | (uint32_t)in_get(in, lane);
              d.curr = uni(cu);
            }
            // ---- EOS flag, p = 0 (Decoder.cs:136-158 with mid = low): y = (curr <= low)
            tq = d.curr - d.low;
            if (UNLIKELY(tq - 1 >= d.high - d.low)) break;   // tq == 0: y = 1;  tq > high - low: "archive corrupted"
            // ---- the byte's window (before any coder state changes, so that a miss can simply start over)
            const uint32_t w = (h0 & cm_mask) >> 9;
            const uint64_t hit = __ballot(tag == w);
            if (UNLIKELY(hit == 0)) {                     // window miss: wave C swaps the window in (the fast loop asks for that itself
              thr = uni(thr);                             // unless the replacement threshold has to move or the id is too wide for the request word)
              const uint32_t vs = uni(pick_victim<NW>(lastuse, t, thr));
              const uint32_t old = rdlane(tag, vs);
              tag = lane == vs ? w : tag;
              n_miss = uni(n_miss + 1);
              const uint32_t nn = n_miss & 255u;
              if (w >= 0x10000u && lane == 0) S.aux[0][0] = w;
              lds_st(&S.mold, old);
              lds_order();
              lds_put0(&S.mreq, nn << 23 | (w >= 0x10000u ? 1u << 22 : w << 6) | vs);
              // a window fresh from the table has nothing outstanding with wave B: stamped as if used 13 messages ago (the ring's depth)
              lastuse = lane == vs ? (t > 13u ? t - 13u : 1u) : lastuse;
              if (PROF) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_pub)::"memory"); prof[8] += t_pub - t_exit; prof[10] += 1; }   // loop left -> request published
              { uint32_t spn = 0; while (uni(lds_ld(&S.mdone)) != nn) { if (++spn > kSpinSection) { ev = kEvHelper; d.low = d.high = d.curr = 1; break; } } }
              lds_order();
              continue;                                   // the fast loop takes the byte
            }
            const uint32_t slot = (uint32_t)__builtin_ctzll(hit);
            uint32_t bad = 0, err = 0, helper_lost = 0;
            {  // a swap the fast loop asked for and gave up waiting on (it never does in practice) must be in before the window is read
              uint32_t spn = 0;
              while (UNLIKELY(uni(lds_ld(&S.mdone)) != (n_miss & 255u))) { if (++spn > kSpinSection) { helper_lost = 1; break; } }
              lds_order();
            }
            d.low += 1;
            if (UNLIKELY((d.high ^ d.low) < 0x1000000u)) {
              if (dec_renorm_chk(d, in, lane, bad)) err = kEvEof;
            }
            ZH_STAMP(0);
            // The cached probabilities of this window must include every earlier byte that used it, and
            // the ring must not overrun: B may be at most `back` messages behind.
            {
              const uint32_t since = t - rdlane(lastuse, slot);
              const uint32_t back = since < kRing - 3 ? since : kRing - 3;
              if (UNLIKELY(t - b_done > back)) { if (!wait_done(t - back)) helper_lost = 1; }
            }
            ZH_STAMP(1);
            const uint32_t pa = (uint32_t)lds_u16(p_la + slot * 32) << 16;
            const uint64_t pb = lds_u64(p_lb + slot * 512);
            uint32_t j = 1;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
              const uint32_t ps = rdlane(pa, j);
              uint32_t xr;
              ZH_DEC_STEP_LITE(d, ps, j, xr);
              if (UNLIKELY(xr < 0x1000000u)) {
                const uint32_t was = bad;
                if (dec_renorm_chk(d, in, lane, bad) && !err) err = was ? kEvCorrupt : kEvEof;
              }
            }
            // second nibble: group 16 + n1 = quad (n1 & 3), element n1 >> 2
            const uint32_t ga = uni(j & 15), lb = (ga & 3) * 16;
            const uint32_t psel = (uint32_t)(pb >> ((ga >> 2) * 16)) << 16;
            uint32_t j2 = 1;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
              const uint32_t ps = rdlane(psel, lb + j2);
              uint32_t xr;
              ZH_DEC_STEP_LITE(d, ps, j2, xr);
              if (UNLIKELY(xr < 0x1000000u)) {
                // after the byte's last bit the next call re-primes / re-checks by itself (Decoder.cs:36-45)
                uint32_t later = 0;
                const uint32_t was = bad;
                if (dec_renorm_chk(d, in, lane, tt == 3 ? later : bad) && !err) err = was ? kEvCorrupt : kEvEof;
              }
            }
            ZH_STAMP(4);
            if (LIKELY((err | bad | helper_lost) == 0)) {
              const uint32_t cc = (j << 4) + j2 - 272;     // j = 16 | n1, j2 = 16 | n2
              // the byte goes to wave B: training, probability refresh and output happen there
              publish(kMsgByte << 23 | cc << 15 | slot);
              lastuse = lane == slot ? t : lastuse;
              h0 = cc << hshift;                           // HCOMP "a<<= K  *d=a  halt" (Predictor.cs:464-470)
            } else {
              // an error ends the section through the loop's only exit: make the next EOS test fire
              ev = helper_lost ? (uint32_t)kEvHelper : err ? err : (uint32_t)kEvCorrupt;
              d.low = d.high = d.curr = 1;
            }
            ZH_STAMP(6);
          }
          if (!ev) ev = tq ? kEvCorrupt : kEvEos;
          if (ev == kEvEos) {                             // y = 1: high = mid = low, then the usual renormalisation
            d.high = d.low;
            if (dec_renorm(d, in, lane)) status = ZH_E_EOF;
            else if (d.curr != 0) status = ZH_E_EOS;
          } else {
            status = ev == kEvCorrupt ? ZH_E_CORRUPT : ev == kEvEof ? ZH_E_EOF : ZH_E_HELPER;
          }
          // leave the section: B flushes the output and hands its state back
          publish(kMsgLeave << 23);
          if (!wait_done(t) && !status) status = ZH_E_HELPER;
          ob.len = uni64(S.ob_len); ob.stored = uni64(S.ob_stored); ob.word = uni(S.ob_word); ob.room = uni(S.ob_room);
          ob.park = 0;
          break;                                          // end of segment or error
        }
        int c = decode_byte();
        if (c == -2) break;
        // ---- PostProcessor.write(c) (PostProcessor.cs:37-86)
        if (pp_state == 5) {
          // every lane runs the program (same inputs, same stores): keeps control flow wave-uniform
          int rc = (int)uni((uint32_t)vm_run(pz, (uint32_t)c, &sink, L.budget));
          if (rc) { status = rc; break; }
        } else if (pp_state == 0) {
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
        } else {                                        // state 4: PCOMP bytes
          if (c < 0) { status = ZH_E_PP_EOS; break; }
          pzbuf[pp_len] = (uint8_t)c;                  // all lanes store the same byte
          if ((int)++pp_len == pp_hsize) {
            wave_sync();
            pz.prog = pzbuf; pz.len = pp_len;
            pz.a = pz.b = pz.c = pz.d = pz.f = 0;
            pp_state = 5;
          }
        }
        if (c < 0) break;
      }

      if (pp_state != 5) out_flush(ob, lane);
      uint64_t produced = pp_state == 5 ? uni64(sink.len) : ob.len;
      if (!status && produced > b_out_cap) status = ZH_E_OUTPUT_FULL;
      if (status && status != ZH_E_OUTPUT_FULL) failed = 1;
      if (lane == 0) {
        ZhSegResult res;
        res.status = status; res.pp_state = (uint32_t)pp_state | (uint32_t)pp_hsize << 8;   // PCOMP length rides in bits 8-23
        res.out_off = b_out_off + produced0; res.out_len = produced - produced0;
        res.in_used = in_pos(in) - seg_off;
        L.results[si] = res;
      }
    }
    if (PROF && lane == 0 && L.debug)
      for (int i = 0; i < 13; ++i) if (i != 5 && i != 7) atomicAdd((unsigned long long *)&L.debug[i], (unsigned long long)prof[i]);
    wave_sync();
  }
  if (lane == 0) { S.cmd_code = kCmdExit; S.mcfg = ~0u; }   // release waves B and C
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  lds_st(&S.cmd_seq, cmd_seq + 1);
}

}  // namespace

extern "C" __global__ __launch_bounds__(192) void zh_decode_cm(ZhLaunch L) {
  __shared__ CmLdsT<64, 1> S;
  decode_cm_body<false, 64, 1>(L, S);
}

// Two blocks per workgroup (four wavefronts, 32 windows per block): taken by the host when a launch has more blocks than
// the GPU has CUs (zh_api.cpp), so that an archive of many blocks uses all four SIMDs of every CU.
extern "C" __global__ __launch_bounds__(384) void zh_decode_cm_x2(ZhLaunch L) {
  __shared__ CmLdsT<32, 2> S;
  decode_cm_body<false, 32, 2>(L, S);
}

extern "C" __global__ __launch_bounds__(192) void zh_decode_cm_prof(ZhLaunch L) {
  __shared__ CmLdsT<64, 1> S;
  decode_cm_body<true, 64, 1>(L, S);
}

extern "C" hipError_t zh_launch_cm_prof(const ZhLaunch *L, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(zh_decode_cm_prof, dim3(grid), dim3(192), 0, stream, *L);
  return hipGetLastError();
}

extern "C" hipError_t zh_launch_cm(const ZhLaunch *L, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(zh_decode_cm, dim3(grid), dim3(192), 0, stream, *L);
  return hipGetLastError();
}

// grid workgroups of two blocks each: the arena must hold 2 x grid slots
extern "C" hipError_t zh_launch_cm_x2(const ZhLaunch *L, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(zh_decode_cm_x2, dim3(grid), dim3(384), 0, stream, *L);
  return hipGetLastError();
}
