// zh_chain3.hip — the reference's built-in mid / max models (Compressor.cs:53-72) on THREE cooperating wavefronts per
// block: a decoder wave, a model wave and the helper wave of zh_chain2.hip.
//
// Why: one wavefront issues at most one instruction per 4 cycles, so a block decoded by one wave costs
// (instructions per bit) x 4 cycles at best, whatever the 63 other lanes could do (zh_chain2.hip: ~150 per bit for mid).
// Per bit only  predictions -> mixer(s) -> squash -> Decoder.decode (Decoder.cs:136-158)  has to wait for the bit
// before; the rest of Predictor.update (Predictor.cs:353-475) has to be ready one bit later.  So the work is cut in two
// along that line and the two halves run on two SIMDs of the CU:
//
//   * DECODER WAVE (wave 0, the block's master): the final mixer(s) — mid: MIX 7; max: MIX 15/16, MIX2 17/19/21,
//     SSE 18/20 — their training, the arithmetic decoder, the post-processor and the output.  Its loop per bit is
//     read p[] -> dot product -> squash -> decode -> publish y.
//   * MODEL WAVE (wave 1): every ICM / ISSE / MATCH / CONST component (lane = component), their hash rows and bit
//     histories.  It SPECULATES over the bit being decoded: lanes 0-15 update the model as if that bit were 0,
//     lanes 16-31 as if it were 1 (one instruction stream, the hypothesis is a per-lane constant; a DPP row is 16 lanes,
//     so the systolic ISSE step works inside each half unchanged), and both halves publish the predictions p[] that
//     follow.  When the decoder wave has the bit it reads the half that was right — without waiting for this wave; the
//     model wave learns the bit one hand-over later, commits the winning half's writes (table entry, bit history) and
//     copies its state to the other half through LDS.  Bits whose successor needs new hash rows (the last bit of a
//     nibble) are not speculated: the wave waits for the bit and then works as zh_chain2.hip does.
//   * HELPER WAVE (wave 2): zh_c2_common.h, unchanged — HCOMP for the 16 values the byte can still take, the next
//     byte's hash rows and mixer row staged in LDS.
//
// Hand-overs go through LDS words written by one wave each (yv, pv, the helper's mailboxes); a hand-over costs 130-190
// cycles (tools/ubench/hop_bench.hip), which is why only ONE of them (bit -> model wave) is left per bit and why it is
// off the decoder wave's path.  Every wait is bounded and watches the block command word.
#include "zh_c2_common.h"

namespace {

constexpr uint32_t kC3Spin = 1u << 24;
__device__ __forceinline__ void c2_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }   // this wave's LDS / memory operations are done

// ---------------------------------------------------------------------------------------------------------------
// MODEL WAVE
// ---------------------------------------------------------------------------------------------------------------
template <class SP, bool SPEC, bool PROF, class LDS>
__device__ void c3_model(const ZhLaunch &L, LDS &S, uint32_t lane) {
  // diagnostic build: cycles this wave waits for a bit — in speculated steps [12], at the nibble switch [13], at the end of
  // the byte [14] — and for the helper wave [15]
  uint64_t mprof[4] = {0, 0, 0, 0};
  auto now = [&]() __attribute__((always_inline)) -> uint64_t { uint64_t t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; };
  constexpr uint64_t kII = SP::icm | SP::isse;
  constexpr int HELP = SP::helper;
  const uint32_t g = (lane >> 4) & 1u, li = lane & 15u;       // hypothesis of this half, component of this lane
  const bool l_isse = (SP::isse >> li) & 1, l_ii = (kII >> li) & 1;
  const bool l_match = SP::match_lane >= 0 && li == (uint32_t)SP::match_lane;
  uint8_t *slot_mem = L.arena + (uint64_t)blockIdx.x * L.arena_stride;
  const lds_i16_p lds_stretch = (lds_i16_p)lds_off(S.stretch);
  const lds_u16_p lds_squash = (lds_u16_p)lds_off(S.squash);
  const uint32_t ns_off = lds_off(S.ns);
  uint32_t seen_cmd = 0;

  for (;;) {
    uint32_t cmd, sp = 0;
    while ((cmd = c2_ld(&S.mb_cmd)) == seen_cmd) { __builtin_amdgcn_s_sleep(4); if (++sp > kC3Spin) return; }
    seen_cmd = cmd;
    if ((cmd & 3u) == kC2Exit) return;
    if ((cmd & 3u) != kC2New) { c2_put0(&S.mb_ack2, cmd); continue; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const ZhModel *M = &L.models[uni(c2_ld(&S.mb_model))];
    const uint32_t arena_bytes = uni((uint32_t)M->arena_bytes);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(slot_mem, 0, (int)arena_bytes, 0x00020000);

    // ---- Predictor.init (Predictor.cs:82-171): the arena tables this kernel keeps in HBM (all 64 lanes fill)
    for (uint32_t i = 0; i < SP::n; ++i) {
      const ZhComp &cp = M->comp[i];
      const uint32_t type = uni(cp.type);
      uint8_t *cm = slot_mem + uni64(cp.cm_off), *ht = slot_mem + uni64(cp.ht_off);
      const uint64_t cmb = uni64(cp.cm_bytes), htb = uni64(cp.ht_bytes);
      uint4 pat = make_uint4(0, 0, 0, 0);
      bool fill_cm = false;
      if (type == ZH_MATCH) fill_cm = true;
      else if (type == ZH_MIX2) { pat = make_uint4(0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u); fill_cm = true; }
      else if (type == ZH_MIX) { const uint32_t w = 65536u / uni(cp.arg[2]); pat = make_uint4(w, w, w, w); fill_cm = true; }
      if (fill_cm) { uint4 *q = reinterpret_cast<uint4 *>(cm); for (uint64_t k = lane; k < cmb / 16; k += 64) q[k] = pat; }
      if (type == ZH_SSE) {                              // squash((j&31)*64-992)<<17 | start, period 32 entries
        const uint32_t start = uni(cp.arg[2]);
        uint4 *q = reinterpret_cast<uint4 *>(cm);
        for (uint64_t k = lane; k < cmb / 16; k += 64) {
          const uint32_t j = (uint32_t)(k * 4) & 31;
          uint4 v;
          v.x = (uint32_t)S.squash[(j + 0) * 64 - 992 + 2048] << 17 | start;
          v.y = (uint32_t)S.squash[(j + 1) * 64 - 992 + 2048] << 17 | start;
          v.z = (uint32_t)S.squash[(j + 2) * 64 - 992 + 2048] << 17 | start;
          v.w = (uint32_t)S.squash[(j + 3) * 64 - 992 + 2048] << 17 | start;
          q[k] = v;
        }
      }
      if (type == ZH_ICM || type == ZH_ISSE || type == ZH_MATCH) {
        uint4 *q = reinterpret_cast<uint4 *>(ht);
        for (uint64_t k = lane; k < htb / 16; k += 64) q[k] = make_uint4(0, 0, 0, 0);
      }
    }
    // ---- ICM / ISSE entry tables in LDS.  Unit u of S.ent belongs to the u-th ICM/ISSE component.
    const uint32_t unit = (uint32_t)__builtin_popcountll(kII & ((1ull << li) - 1));
    for (uint32_t j = lane; j < 256; j += 64) {
      const uint32_t n0 = S.ns[j * 4 + 2], n1 = S.ns[j * 4 + 3];
      const uint32_t ci = ((n1 * 2 + 1) << 22) / (n0 + n1 + 1);                  // StateTable.cminit
      const int stv = S.stretch[ci >> 8];
      const v2u e_icm = {ci, (uint32_t)stv};
      const v2u e_isse = {1u << 15, (uint32_t)clamp512k(stv * 1024)};
      uint32_t u = 0;
      for (uint32_t i = 0; i < SP::n; ++i) {
        if (!((kII >> i) & 1)) continue;
        S.ent[u][j] = ((SP::icm >> i) & 1) ? e_icm : e_isse;
        ++u;
      }
    }
    S.slot[lane] = v4u{0, 0, 0, 0};
    S.lent[lane] = v2u{0, 0};
    S.lsink[lane] = 0;
    c2_wave_sync();

    // ---- per-lane constants (lane = hypothesis half g, component li)
    const ZhComp *mycp = &M->comp[li < SP::n ? li : 0];
    const uint32_t hto = l_ii || l_match ? (uint32_t)mycp->ht_off : 0u, ht_mask = mycp->ht_mask;
    const uint32_t cmo = (uint32_t)mycp->cm_off, cm_mask = mycp->cm_mask;
    const uint32_t sizebits2 = (uint32_t)mycp->arg[0] + 2;
    const uint32_t tab = l_ii ? lds_off(&S.ent[unit][0]) : lds_off(&S.lent[lane]);     // entry table of this lane
    const uint32_t rrow = l_ii ? lds_off(&S.slot[li]) : lds_off(&S.zrow);              // row it reads bit histories from (both halves: one row)
    const uint32_t wrow = l_ii ? lds_off(&S.slot[li]) : lds_off(&S.lsink[lane]);       // ... and writes them to (+ node index)
    const uint32_t wrow_mask = l_ii ? 15u : 0u;
    const int isse_m = l_isse ? -1 : 0;
    const uint32_t cshift = l_isse ? 6u : 16u;
    int pself = 0;                                       // prediction of a lane that is neither ICM nor ISSE (MATCH, CONST)
    if (li < SP::n && mycp->type == ZH_CONS) pself = ((int)mycp->arg[0] - 128) * 4;
    if (l_match) (slot_mem + hto)[0] = 1;                // Predictor.cs:121 ht(0)=1 ... overwritten like the reference
    const uint32_t pv_off = lds_off(&S.pv[0][g][li]);    // + 128 for odd sequence numbers
    const uint32_t xf_own = lds_off(&S.xfer[lane]), xf_base = lds_off(&S.xfer[li]);     // + 256 when the bit was 1

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the fills are in memory before the decoder wave (and the helper) read them
    __builtin_amdgcn_s_waitcnt(0);
    c2_put0(&S.mb_ack2, cmd);

    // ---- state carried from bit to bit
    uint32_t hv = 0;                                    // h[li] (Predictor.cs:469)
    uint32_t rowoff = 0;                                // arena offset of the hash row held in S.slot[li]
    bool rowvalid = false;
    uint32_t ea = tab, st = 0, eA = 0, eB = 0;          // entry the current bit predicts from: LDS address, bit-history state, value
    int p = 0, sqp = 2048;                              // this lane's prediction for the current bit and its squash
    uint32_t m_len = 0, m_ptr = 0, m_limit = 0, m_byte = 0;
    int pm0 = 0, pm1 = 0;
    uint32_t cm_pre = 0, va_pre = 0, vb_pre = 0, mbn_pre = 0, mbc_pre = 0;
    uint32_t bs = 1, bseq = 1;                          // sequence number of the bit / byte being decoded
    bool alive = true;

    auto match_prefetch = [&]() __attribute__((always_inline)) {
      const uint32_t ml = (uint32_t)(SP::match_lane >= 0 ? SP::match_lane : 0);
      const uint32_t msk = rdlane(ht_mask, ml), base = rdlane(hto, ml);
      const uint32_t lim = (rdlane(m_limit, ml) + 1u) & msk;
      const uint32_t off = lim - rdlane(cm_pre, ml);
      va_pre = __builtin_amdgcn_raw_buffer_load_b8(rsrc, base + ((lim - lane - 1u) & msk), 0, 0);
      vb_pre = __builtin_amdgcn_raw_buffer_load_b8(rsrc, base + ((lim - lane - off - 1u) & msk), 0, 0);
      mbn_pre = __builtin_amdgcn_raw_buffer_load_b8(rsrc, l_match ? base + ((lim - off) & msk) : kOob, 0, 0);
      mbc_pre = __builtin_amdgcn_raw_buffer_load_b8(rsrc, l_match ? base + ((lim - m_ptr) & msk) : kOob, 0, 0);
    };
    // Predictor.update's MATCH part at the byte boundary (Predictor.cs:391-410): as zh_chain2.hip; the verification runs
    // on all 64 lanes of this wave (va_pre / vb_pre are per full lane)
    auto match_boundary = [&](uint32_t cb) __attribute__((always_inline)) {
      const bool zero = m_len == 0;
      const uint32_t nptr = m_limit - cm_pre;
      const bool need = l_match && zero && (nptr & ht_mask) != 0;
      m_ptr = (l_match && zero) ? nptr : m_ptr;
      m_len = (l_match && !zero && m_len < 255) ? m_len + 1 : m_len;
      if (__ballot(need) != 0) {
        const uint32_t ml = (uint32_t)SP::match_lane;
        const uint32_t lim = rdlane(m_limit, ml), off = rdlane(m_ptr, ml), msk = rdlane(ht_mask, ml);
        const uint32_t a = lane == 0 ? cb : (va_pre & 255u);
        const uint32_t b = ((lane + off) & msk) == 0 ? cb : (vb_pre & 255u);
        uint64_t mism = __ballot(a != b);
        uint32_t len = 64;
        if (LIKELY(mism != 0)) len = (uint32_t)__builtin_ctzll(mism);
        else {
          const uint8_t *hp = slot_mem + rdlane(hto, ml);
          for (uint32_t base = 64; base < 256; base += 64) {
            const uint32_t t = base + lane;
            const bool eq = t < 255 && hp[(lim - t - 1) & msk] == hp[(lim - t - off - 1) & msk];
            mism = __ballot(!eq);
            if (mism) { len += (uint32_t)__builtin_ctzll(mism); break; }
            len += 64;
          }
        }
        const uint32_t nl = len > 255 ? 255 : len;
        m_len = l_match ? nl : m_len;
        m_byte = l_match ? (((off - 1u) & msk) == 0 ? cb : (mbn_pre & 255u)) : m_byte;
      } else {
        const uint32_t cont = ((m_ptr - 1u) & ht_mask) == 0 ? cb : (mbc_pre & 255u);
        m_byte = (l_match && m_len) ? cont : m_byte;
      }
      const uint32_t pw = *(lds_u32_p)(lds_off(S.pm01) + m_len * 4u);
      pm0 = (int)(int16_t)(pw & 0xffffu); pm1 = (int)pw >> 16;
    };

    // Hash rows of the nibble that starts now, Predictor.find (Predictor.cs:550-567): as zh_chain2.hip
    struct Probe { v4u r0, r1, r2; uint32_t h0, chk; };
    auto rows_issue = [&](uint32_t c8, Probe &pr) __attribute__((always_inline)) {
      const uint32_t cxt = hv + 16u * c8;
      pr.chk = (cxt >> sizebits2) & 255;
      pr.h0 = (cxt * 16u) & (ht_mask - 15u);
      const uint32_t vo = l_ii ? hto + pr.h0 : kOob;
      pr.r0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, 0, 0);
      pr.r1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo ^ 16u, 0, 0);
      pr.r2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo ^ 32u, 0, 0);
    };
    uint32_t row_x = 0;
    auto rows_finish2 = [&](const Probe &pr, const v4u &olda, uint32_t olda_off, bool olda_valid, const v4u &old, uint32_t old_off,
                            bool old_valid) __attribute__((always_inline)) {
      const uint32_t h0 = pr.h0, h1 = h0 ^ 16u, h2 = h0 ^ 32u;
      v4u r0 = pr.r0, r1 = pr.r1, r2 = pr.r2;
      if (olda_valid && olda_off == h0) r0 = olda;
      if (olda_valid && olda_off == h1) r1 = olda;
      if (olda_valid && olda_off == h2) r2 = olda;
      if (old_valid && old_off == h0) r0 = old;
      if (old_valid && old_off == h1) r1 = old;
      if (old_valid && old_off == h2) r2 = old;
      const uint32_t chk = pr.chk;
      const bool m0 = (r0.x & 255) == chk, m1 = (r1.x & 255) == chk, m2 = (r2.x & 255) == chk;
      const uint32_t p0 = (r0.x >> 8) & 255, p1 = (r1.x >> 8) & 255, p2 = (r2.x >> 8) & 255;
      const uint32_t victim = (p0 <= p1 && p0 <= p2) ? h0 : p1 < p2 ? h1 : h2;
      const uint32_t sel = m0 ? h0 : m1 ? h1 : m2 ? h2 : victim;
      const v4u fresh = {chk, 0, 0, 0};
      const v4u row = m0 ? r0 : m1 ? r1 : m2 ? r2 : fresh;
      if (g == 0 && lane < 16) *(lds_u4_p)lds_off(&S.slot[li]) = row;     // one copy of the row for both halves
      rowoff = sel; rowvalid = true;
      row_x = l_ii ? row.x : 0u;
    };
    auto row_evict = [&](v4u &old, uint32_t &old_off, bool &old_valid) __attribute__((always_inline)) {
      old = *(lds_u4_p)lds_off(&S.slot[li]);
      old_off = rowoff; old_valid = rowvalid && l_ii;
      __builtin_amdgcn_raw_buffer_store_b128(old, rsrc, old_valid && lane < 16 ? hto + rowoff : kOob, 0, 0);
    };
    // first bit of a nibble: node 1 of the row just selected (its histories are in row_x: no LDS round trip)
    auto l0_direct = [&]() __attribute__((always_inline)) {
      st = (row_x >> 8) & 255u;
      ea = tab + st * 8u;
      const v2u e = *(lds_u2_p)ea;
      eA = e.x; eB = e.y;
    };
    // the systolic ISSE step over the entries just selected -> p, published for the decoder wave under sequence number sq
    auto predict_publish = [&](int bitpos, uint32_t sq) __attribute__((always_inline)) {
      int xs = pself;
      if (SP::match_lane >= 0) {
        const uint32_t cbit = (m_byte >> (7 - bitpos)) & 1;
        xs = l_match ? (cbit ? pm1 : pm0) : pself;
      }
      const int x = l_ii ? (int)eB : xs;
      const int cw0 = (int)eA & isse_m;
      const int cw1m = (int)((uint32_t)x << cshift);
      int q = x;
#pragma unroll
      for (uint32_t t = 0; t < SP::depth; ++t) q = med3i((__mul24(shr1(q), cw0) + cw1m) >> 16, -2048, 2047);
      p = q;
      *(lds_u32_p)(pv_off + (sq & 1u) * 128u) = (sq << 12) | ((uint32_t)q & 0xfffu);
      sqp = (int)*(lds_u16_p)((uint32_t)(uintptr_t)lds_squash + (uint32_t)(q + 2048) * 2u);
    };
    // the bit with sequence number sq, as soon as the decoder wave has published it
    auto wait_bit = [&](uint32_t sq, int slot) __attribute__((always_inline)) -> uint32_t {
      const uint32_t addr = lds_off(&S.yv);
      uint32_t v = 0, d = 0, tmp, rounds = 0;
      uint64_t t0 = 0;
      if (PROF) t0 = now();
      for (;;) {
        uint32_t left = 1u << 14;
        asm volatile(
            ".Lwb_%=:\n\t"
            "ds_read_b32 %[t], %[a]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_readfirstlane_b32 %[v], %[t]\n\t"
            "s_lshr_b32 %[d], %[v], 8\n\t"
            "s_sub_u32 %[d], %[d], %[q]\n\t"
            "s_and_b32 %[d], %[d], 0xffffff\n\t"
            "s_cmp_lt_u32 %[d], 2\n\t"                   // (the decoder wave is at most one bit ahead)
            "s_cbranch_scc1 .Lwd_%=\n\t"
            "s_sub_u32 %[l], %[l], 1\n\t"
            "s_cmp_lg_u32 %[l], 0\n\t"
            "s_cbranch_scc1 .Lwb_%=\n"
            ".Lwd_%=:"
            : [t] "=&v"(tmp), [v] "=&s"(v), [d] "=&s"(d), [l] "+s"(left)
            : [a] "v"(addr), [q] "s"(sq)
            : "memory", "scc");
        if (LIKELY(left != 0)) break;
        if (c2_ld(&S.mb_cmd) != seen_cmd || ++rounds > (1u << 10)) { alive = false; return 0u; }   // the block was given up
      }
      if (PROF) mprof[slot] += now() - t0;
      return (v >> d) & 1u;
    };

    // ---- first nibble of the block (h[] = 0)
    {
      Probe pr;
      rows_issue(1u, pr);
      rows_finish2(pr, v4u{0, 0, 0, 0}, 0u, false, v4u{0, 0, 0, 0}, 0u, false);
      asm volatile("" ::: "memory");
      l0_direct();
      predict_publish(0, bs);
    }

    while (alive) {                                      // one byte per iteration
      uint32_t c8 = 1, hm = 1;
      Probe spec[4];
      v4u old1 = {0, 0, 0, 0}; uint32_t old1_off = 0; bool old1_valid = false;
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
        const bool in_nib = (bit & 3) != 3;              // the next bit stays in this nibble
        const bool spec_step = SPEC && in_nib;
        hm = uni(hm); c8 = uni(c8);
        if (SP::match_lane >= 0 && bit == 4) match_prefetch();
        uint32_t y = 0;
        if (!spec_step) { y = wait_bit(bs, bit == 3 ? 1 : 2); if (UNLIKELY(!alive)) break; }
        const uint32_t yh = spec_step ? g : y;            // the bit this lane works with
        // ---- Predictor.update for bit `bs` under yh (Predictor.cs:363-461)
        const int ey = yh ? 32767 : 0;
        const int e = ey - sqp;
        const uint32_t nsb = *(lds_u8_p)(ns_off + st * 4u + yh);
        const uint32_t ncm = eA + (uint32_t)((int)(ey - (int)(eA >> 8)) >> 2);
        const int npst = *(lds_i16_p)((uint32_t)(uintptr_t)lds_stretch + ((ncm >> 7) & 0x1fffeu));
        const int pj = shr1(p);
        const int nw0 = med3i((int)eA + ((__mul24(e, pj) + (1 << 12)) >> 13), -(1 << 19), (1 << 19) - 1);
        const int nw1 = med3i((int)eB + ((e + 16) >> 5), -(1 << 19), (1 << 19) - 1);
        const uint32_t nA = l_isse ? (uint32_t)nw0 : ncm, nB = l_isse ? (uint32_t)nw1 : (uint32_t)npst;
        const uint32_t cbit = SP::match_lane >= 0 ? (m_byte >> (7 - bit)) & 1 : 0u;
        uint32_t nst = 0, nea = tab, neA = 0, neB = 0;
        if (in_nib) {                                     // the node after this one: 2 hm + yh of the same row
          const uint32_t hmn = hm * 2u + yh;
          nst = *(lds_u8_p)(rrow + hmn);
          nea = tab + nst * 8u;
          const v2u en = *(lds_u2_p)nea;
          const bool same = nea == ea;                    // this bit trains the entry the next one predicts from
          neA = same ? nA : en.x; neB = same ? nB : en.y;
        }
        if (spec_step) {
          // ---- speculative step: predictions for the next bit under both values of this one; then the bit itself
          const uint32_t oea = ea, ohm = hm;
          const int opm0 = pm0, opm1 = pm1;
          if (SP::match_lane >= 0 && cbit != yh) { pm0 = 0; pm1 = 0; }     // (this half only: restored below)
          st = nst; ea = nea; eA = neA; eB = neB;
          predict_publish(bit + 1, bs + 1u);
          *(lds_u4_p)xf_own = v4u{eA, eB, (uint32_t)sqp, st};
          pm0 = opm0; pm1 = opm1;
          y = wait_bit(bs, 0);
          if (UNLIKELY(!alive)) break;
          if (g == y) {                                   // the half that was right commits its writes
            *(lds_u2_p)oea = v2u{nA, nB};
            *(lds_u8_p)(wrow + (ohm & wrow_mask)) = (uint8_t)nsb;
          }
          const v4u w = *(lds_u4_p)(xf_base + y * 256u);  // ... and its state becomes everybody's
          const uint32_t pvw = *(lds_u32_p)(lds_off(&S.pv[0][0][li]) + ((bs + 1u) & 1u) * 128u + y * 64u);
          eA = w.x; eB = w.y; sqp = (int)w.z; st = w.w; ea = tab + st * 8u;
          p = (int)(pvw << 20) >> 20;
        } else {
          *(lds_u2_p)ea = v2u{nA, nB};
          *(lds_u8_p)(wrow + (hm & wrow_mask)) = (uint8_t)nsb;
        }
        if (SP::match_lane >= 0) {                       // MATCH (Predictor.cs:383-384): a miss ends the match
          const bool miss = cbit != y;
          m_len = miss ? 0u : m_len; pm0 = miss ? 0 : pm0; pm1 = miss ? 0 : pm1;
        }
        c8 = c8 * 2u + y;
        if (in_nib) {
          hm = hm * 2u + y;
          if (!spec_step) { st = nst; ea = nea; eA = neA; eB = neB; predict_publish(bit + 1, bs + 1u); }
        } else if (bit == 3) {
          // ---- second nibble (Predictor.cs:267-270): new rows, requested two bits ago
          v4u old; uint32_t old_off; bool old_valid;
          row_evict(old, old_off, old_valid);
          old1 = old; old1_off = old_off; old1_valid = old_valid;
          switch (c8 & 3u) {
            case 0: rows_finish2(spec[0], old, 0u, false, old, old_off, old_valid); break;
            case 1: rows_finish2(spec[1], old, 0u, false, old, old_off, old_valid); break;
            case 2: rows_finish2(spec[2], old, 0u, false, old, old_off, old_valid); break;
            default: rows_finish2(spec[3], old, 0u, false, old, old_off, old_valid); break;
          }
          hm = 1;
          l0_direct();
          predict_publish(4, bs + 1u);
        }
        if (bit == 1) {
#pragma unroll
          for (uint32_t k = 0; k < 4; ++k) rows_issue(c8 * 4u + k, spec[k]);
        }
        ++bs;
      }
      if (!alive) break;
      const uint32_t c = c8 - 256u;

      // ---- byte boundary: MATCH (Predictor.cs:391-410), h[] from the helper wave, rows of the next byte
      if (SP::match_lane >= 0) {
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)c, rsrc, l_match && lane < 16 ? hto + (m_limit & ht_mask) : kOob, 0, 0);
        m_limit = l_match ? (m_limit + 1) & ht_mask : m_limit;
        __builtin_amdgcn_raw_buffer_store_b32(m_limit, rsrc, l_match && lane < 16 ? cmo + (hv & cm_mask) * 4u : kOob, 0, 0);
      }
      {
        uint32_t spin = 0;
        uint64_t t0 = 0;
        if (PROF) t0 = now();
        while (c2_ld(&S.mb_ready) != bseq) {
          if ((++spin & 63u) == 0 && (c2_ld(&S.mb_cmd) != seen_cmd || spin > kC3Spin)) { alive = false; break; }
        }
        if (!alive) break;
        if (PROF) mprof[3] += now() - t0;
      }
      asm volatile("" ::: "memory");
      hv = S.hspec[li & ((1u << SP::hh) - 1u)][c & 15u];
      ++bseq;
      v4u old; uint32_t old_off; bool old_valid;
      row_evict(old, old_off, old_valid);
      Probe pr;
      if (HELP == 1) {
        const uint32_t lo = c & 15u;
        const uint32_t cxt = hv + 16u;
        pr.chk = (cxt >> sizebits2) & 255;
        pr.h0 = (cxt * 16u) & (ht_mask - 15u);
        const uint32_t un = unit < (uint32_t)kSpecUnits ? unit : 0u;
        pr.r0 = *(lds_u4_p)lds_off(&S.rowst[un][0][lo]);
        pr.r1 = *(lds_u4_p)lds_off(&S.rowst[un][1][lo]);
        pr.r2 = *(lds_u4_p)lds_off(&S.rowst[un][2][lo]);
      } else {
        rows_issue(1u, pr);
      }
      if (SP::match_lane >= 0) {
        match_boundary(c);
        cm_pre = __builtin_amdgcn_raw_buffer_load_b32(rsrc, l_match ? cmo + (hv & cm_mask) * 4u : kOob, 0, 0);
      }
      rows_finish2(pr, old1, old1_off, old1_valid, old, old_off, old_valid);
      asm volatile("" ::: "memory");
      l0_direct();
      predict_publish(0, bs);
    }
    if (PROF && lane == 0 && L.debug)
      for (int i = 0; i < 4; ++i) { atomicAdd((unsigned long long *)&L.debug[12 + i], (unsigned long long)mprof[i]); mprof[i] = 0; }
    // the block ended (or the decoder wave gave it up): back to the command loop, which acknowledges End
  }
}

// ---------------------------------------------------------------------------------------------------------------
// DECODER WAVE (master of the block)
// ---------------------------------------------------------------------------------------------------------------
template <class SP, bool SPEC, bool PROF, class LDS>
__device__ void c3_decoder(const ZhLaunch &L, LDS &S, uint32_t lane) {
  constexpr int HELP = SP::helper;
  // diagnostic build: cycles this wave waits for the model wave's predictions (bits 0-2, 5, 6 [0]; 3 [1]; 4 [2]; 7 [3]), from
  // having them to the publication of the bit [4], from there to the end of the bit [5], between bits (requests of later
  // rows, byte prologue, boundary) [6]; per byte in total [8], waiting for the helper wave [9], post-processor [10], bytes [11]
  uint64_t prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto now = [&]() __attribute__((always_inline)) -> uint64_t { uint64_t t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; };
  uint32_t cmd_seq = 0;
  const uint32_t *ps_tab = reinterpret_cast<const ZhTablesX *>(L.tables + 1)->ps;
  uint8_t *slot_mem = L.arena + (uint64_t)blockIdx.x * L.arena_stride;
  const lds_i16_p lds_stretch = (lds_i16_p)lds_off(S.stretch);
  const lds_u16_p lds_squash = (lds_u16_p)lds_off(S.squash);
  const uint32_t pv_lane = lds_off(&S.pv[0][0][lane & 15u]);

  for (;;) {
    uint32_t bi = 0;
    if (lane == 0) bi = atomicAdd(L.queue, 1u);
    bi = uni((uint32_t)__shfl((int)bi, 0));
    if (bi >= L.n_blocks) break;                       // every wave reaches the exit (the others on kC2Exit)

    const ZhBlockDesc *bdp = &L.blocks[bi];
    const uint32_t model_i = uni(bdp->model);
    const uint32_t first_seg = uni(bdp->first_seg), n_seg = uni(bdp->n_seg);
    const uint64_t b_out_off = uni64(bdp->out_off), b_out_cap = uni64(bdp->out_cap);
    const ZhModel *M = &L.models[model_i];
    const uint32_t arena_bytes = uni((uint32_t)M->arena_bytes);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(slot_mem, 0, (int)arena_bytes, 0x00020000);

    {  // VM memories: arena tail zeroed; LDS copies zeroed; the tables of the tail components that live in LDS
      const uint64_t h_off = uni64(M->h_off), tail = uni64(M->arena_bytes) - h_off;
      uint4 *z = reinterpret_cast<uint4 *>(slot_mem + h_off);
      for (uint64_t i = lane; i < tail / 16; i += 64) z[i] = make_uint4(0, 0, 0, 0);
      for (uint32_t i = lane; i < 256; i += 64) { S.r[i] = 0; S.pr[i] = 0; S.hreg[i] = 0; }
      for (uint32_t i = lane; i < kMBytes / 4; i += 64) reinterpret_cast<uint32_t *>(S.mreg)[i] = 0;
      for (uint32_t i = lane; i < kPHWords; i += 64) S.phreg[i] = 0;
      for (uint32_t i = lane; i < kPMBytes / 4; i += 64) reinterpret_cast<uint32_t *>(S.pmreg)[i] = 0;
      if (SP::has_tail) {
        for (uint32_t k = lane; k < 256 * 32; k += 64) S.sse18[k] = (uint32_t)S.squash[(k & 31) * 64 - 992 + 2048] << 17 | C2Max::sse_start;
        for (uint32_t k = lane; k < 256; k += 64) S.a19[k] = 32768;
      }
      if (lane < 64) { reinterpret_cast<uint32_t *>(S.pv)[lane] = 0; }
      if (lane == 0) { S.yv = 0; S.mb_model = model_i; S.mb_block = bi; S.mb_nib = 0; S.mb_byte = 0; S.mb_ready = 0; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    ++cmd_seq;
    c2_put0(&S.mb_cmd, cmd_seq << 2 | kC2New);           // the model wave builds the model, the helper wave gets ready
    bool helper_ok = c2_wait(&S.mb_ack, cmd_seq << 2 | kC2New);
    const bool model_ok = c2_wait(&S.mb_ack2, cmd_seq << 2 | kC2New);
    const int lost0 = !helper_ok ? 0x3000 : !model_ok ? 0x3001 : 0;
    helper_ok = helper_ok && model_ok;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

    // Mixers (Predictor.cs:302-316, 427-439): lane k owns weight k of the current row.
    //  * max: rows in HBM, the two rows the next bit can lead to requested while this bit is decoded (zh_chain2.hip); this
    //    wave's step is long enough for that.
    //  * mid (MIXLDS): the step is short, a row requested when its bit begins would arrive hundreds of cycles late, and
    //    waiting on vmcnt behind the training stores is worse still.  The helper wave brings the byte's whole block of rows
    //    into LDS (zh_c2_common.h); this wave reads weights from there, trains them there and writes the trained row to
    //    HBM without ever waiting for memory.
    constexpr bool MIXLDS = LDS::kMixLds;
    static_assert(!MIXLDS || (SP::nmix == 1 && SP::mix_j0[0] == 0 && SP::mix_m[0] == 7), "MIXLDS is laid out for the mid model's mixer");
    uint32_t vo_mix[2] = {kOob, kOob};
    uint32_t mx_base[2] = {0, 0}, mx_m4[2] = {0, 0}, mx_size1[2] = {0, 0};
    int mx_rate[2] = {0, 0};
#pragma unroll
    for (uint32_t q = 0; q < SP::nmix; ++q) {
      const ZhComp &mc = M->comp[SP::mix_lane[q]];
      mx_base[q] = uni((uint32_t)mc.cm_off);
      mx_m4[q] = SP::mix_m[q] * 4u;
      mx_size1[q] = uni(mc.cm_mask);
      mx_rate[q] = (int)uni((uint32_t)mc.arg[3]);
      if (lane >= SP::mix_j0[q] && lane < SP::mix_j0[q] + SP::mix_m[q]) vo_mix[q] = (lane - SP::mix_j0[q]) * 4u;
    }
    const bool l_w = lane < SP::mix_m[0];                // this lane holds a weight of mixer 0

    int pp_state = 0, pp_hsize = lost0;                // PostProcessor (PostProcessor.cs:12-16)
    uint32_t pp_len = 0;
    Vm &pz = S.pz;
    pz.a = pz.b = pz.c = pz.d = pz.f = 0;
    pz.prog = nullptr; pz.len = 0;
    const uint32_t phb = uni(M->ph), pmb = uni(M->pm);
    pz.mmask = (uint32_t)((1ull << pmb) - 1); pz.hmask = (uint32_t)((1ull << phb) - 1);
    pz.m = pmb < 31 && (1u << pmb) <= (uint32_t)kPMBytes ? S.pmreg : slot_mem + uni64(M->pm_off);
    pz.h = phb < 31 && (1u << phb) <= (uint32_t)kPHWords ? S.phreg : reinterpret_cast<uint32_t *>(slot_mem + uni64(M->ph_off));
    pz.r = S.pr;
    const bool p_lds = pz.m == S.pmreg && pz.h == S.phreg;
    uint32_t pnative = 0;
    uint32_t pa = 0, pb = 0, pc_ = 0, pd = 0, pf = 0;
    uint8_t *pzbuf = slot_mem + uni64(M->pz_off) + ZH_CODE_PAD;

    Dec d;
    d.low = 1; d.high = 0xFFFFFFFFu; d.curr = 0;
    OutBuf ob;
    ob.base = L.out + b_out_off; ob.cap = b_out_cap; ob.len = 0; ob.stored = 0; ob.word = 0; ob.park = 0;
    out_room(ob);
    Sink &sink = S.sink;
    sink.out = ob.base; sink.cap = ob.cap; sink.len = 0;
    c2_wave_sync();
    uint32_t bseq = 1, bs = 1, yprev = 0;
    InBuf in;
    in.stream = L.in; in.total = L.in_total; in.cbase = 0; in.k = 0; in.avail = 0; in.cur = 0;

    int mw[2] = {0, 0};                                 // this lane's weight in the current row of each mixer
    uint32_t mrow[2] = {0, 0};                          // max: arena offset of that row
    // MIXLDS: block (context) whose rows are in S.mixblk[mpar]; whether it was there when the byte began; what was
    // trained before it arrived
    uint32_t mblk = 0, mpar = 0;
    bool mres = true;
    uint32_t wsrc = 0;                                  // LDS address of this lane's weight in row 0 of the source in use
    int sv_w[2] = {0, 0};
    uint32_t sv_c8[2] = {0, 0};
    uint32_t mx_h[2] = {0, 0}, mx_rb[2] = {0, 0};
    auto mix_set = [&](uint32_t q, uint32_t hq) __attribute__((always_inline)) {
      mx_h[q] = uni(hq);
      mx_rb[q] = uni(mx_base[q] + (mx_h[q] & mx_size1[q] & ~255u) * mx_m4[q]);
    };
    auto mix_row = [&](uint32_t q, uint32_t c8) __attribute__((always_inline)) -> uint32_t {
      return uni(mx_rb[q] + (c8 & 255u) * (SP::mix_m[q] * 4u));
    };
    // ---- tail of the max model (components 17-21), as zh_chain2.hip
    int w17 = 32768, w21 = 32768;
    uint32_t w19 = 32768, a19i = 0;
    uint32_t row18 = 0;
    uint32_t t_h20 = 0;
    const uint32_t sse20_base = SP::has_tail ? uni((uint32_t)M->comp[20].cm_off) : 0u, sse20_mask = SP::has_tail ? uni(M->comp[20].cm_mask) : 0u;
    auto row18_load = [&](uint32_t c8x) __attribute__((always_inline)) -> uint32_t {
      return *(lds_u32_p)(lds_off(S.sse18) + ((c8x & 254u) * 128u) + lane * 4u);
    };
    auto row20_load = [&](uint32_t c8x) __attribute__((always_inline)) -> uint32_t {
      const uint32_t r = uni((((t_h20 + (c8x & ~1u)) * 32u) & sse20_mask) * 4u + sse20_base);
      return __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4u, r, 0);
    };
    auto stretch_u = [&](uint32_t ix) __attribute__((always_inline)) -> int {
      return (int)uni((uint32_t)(int)*(lds_i16_p)((uint32_t)(uintptr_t)lds_stretch + ix * 2u));
    };
    // predictions of the model wave for bit `sq`, the half that assumed the bit before to be `yp`
    auto wait_p = [&](uint32_t sq, uint32_t yp, bool &ok) __attribute__((always_inline)) -> int {
      const uint32_t a = pv_lane + (sq & 1u) * 128u + yp * 64u;
      const uint32_t tag = sq & 0xFFFFFu;
      uint32_t v, spin = 0;
      for (;;) {
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");   // (a poll: re-read every time)
        if (LIKELY(__ballot((v >> 12) != tag) == 0)) break;
        if (++spin > kC3Spin) { ok = false; break; }
      }
      return (int)(v << 20) >> 20;
    };

    uint32_t row20 = 0;
    // what bit 0 of a byte uses (c8 = 1), once the byte's contexts are known; lo = low nibble of the byte before
    auto byte_start_rows = [&](bool staged, uint32_t lo) __attribute__((always_inline)) {
      if constexpr (MIXLDS) {
        const uint32_t blk = uni(mx_h[0] & mx_size1[0] & ~255u);
        mres = uni(blk == mblk ? 1u : 0u) != 0u;
        if (!mres) { mblk = blk; mpar ^= 1u; }
        mpar = uni(mpar); mblk = uni(mblk);
        wsrc = (mres ? lds_off(&S.mixblk[mpar][0]) : lds_off(&S.mixrow8[lo][0])) + lane * 4u;
        const int w = *(lds_u32_p)(wsrc + 7u * 4u);
        mw[0] = l_w ? w : 0;
      } else {
#pragma unroll
        for (uint32_t q = 0; q < SP::nmix; ++q) {
          mrow[q] = mix_row(q, 1u);
          if (staged) { const uint32_t jj = lane - SP::mix_j0[q]; mw[q] = jj < SP::mix_m[q] ? (int)S.mixst[HELP == 1 ? q : 0][lo][jj & 15u] : 0; }
          else mw[q] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, vo_mix[q], mrow[q], 0);
        }
      }
      if (SP::has_tail) { row18 = row18_load(1u); row20 = row20_load(1u); a19i = 1u; w19 = uni((uint32_t)S.a19[1]); }
    };
#pragma unroll
    for (uint32_t q = 0; q < SP::nmix; ++q) mix_set(q, 0u);      // first byte of the block (h[] = 0)
    if constexpr (MIXLDS) {                              // ... whose block is the freshly initialised one (Predictor.cs:141-143)
      const uint32_t w0 = 65536u / SP::mix_m[0];
      for (uint32_t i = lane; i < 1792u; i += 64) S.mixblk[0][i] = w0;
      if (lane == 0) S.mb_blk = 0;
      c2_wave_sync();
    }
    byte_start_rows(false, 0u);

    int failed = 0;
    uint64_t tbyte = 0, tlast = 0;
    if (PROF) { tbyte = now(); tlast = tbyte; }
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

      for (;;) {                                       // one decoded byte per iteration
        // Everything the per-byte control flow tests is one value for the wave; said so here, once per byte, so that
        // the branches below are scalar branches and not exec-mask regions with their register copies.
        pp_state = (int)uni((uint32_t)pp_state); pp_hsize = (int)uni((uint32_t)pp_hsize); pp_len = uni(pp_len); pnative = uni(pnative);
        ob.len = uni64(ob.len); ob.stored = uni64(ob.stored); ob.room = uni(ob.room); ob.word = uni(ob.word);
        in.k = uni(in.k); in.avail = uni(in.avail); in.cbase = uni64(in.cbase);
        bs = uni(bs); bseq = uni(bseq); yprev = uni(yprev);
        // ---- Decoder.decompress prologue (Decoder.cs:36-45)
        if (UNLIKELY(d.curr == 0)) {
          uint32_t cu = 0;
          for (int i = 0; i < 4; ++i) cu = cu << 8 | (uint32_t)in_get(in, lane);
          d.curr = uni(cu);
        }
        uint32_t bad = 0, rn, j = 0, err = 0;
        d.low = uni(d.low); d.high = uni(d.high); d.curr = uni(d.curr);
        ZH_DEC_STEP(d, 0u, j, bad, rn);                // EOS flag: p = 0
        if (UNLIKELY(bad)) { status = ZH_E_CORRUPT; break; }
        if (UNLIKELY(rn)) { if (dec_renorm(d, in, lane)) { status = ZH_E_EOF; break; } }
        int c;
        if (UNLIKELY(j)) {
          if (d.curr != 0) { status = ZH_E_EOS; break; }
          c = -1;
        } else {
          uint32_t c8 = 1;
#pragma unroll
          for (int bit = 0; bit < 8; ++bit) {
            const bool pre_mx = bit != 7;                // the next bit stays in this byte: fetch both of its mixer rows
            c8 = uni(c8);
            int mwc0[2] = {0, 0}, mwc1[2] = {0, 0};
            uint32_t mrow0[2] = {0, 0}, mrow1[2] = {0, 0};
            if constexpr (MIXLDS) {
              if (bit == 2 && uni(mres ? 0u : 1u)) {
                // from bit 3 on the rows come from the block the helper wave has been bringing in since the byte began
                uint32_t v, spin = 0;
                while ((((v = c2_ld(&S.mb_blk)) >> 1) ^ (bseq - 1u)) & 0x7FFFFFFFu) { if (++spin > kC3Spin) { helper_ok = false; break; } }
                helper_ok = uni(helper_ok ? 1u : 0u) != 0u;
                if (UNLIKELY(!helper_ok)) { pp_hsize = 0x4000; break; }
                wsrc = lds_off(&S.mixblk[mpar][0]) + lane * 4u;
                if (l_w) {                                 // what bits 0 and 1 trained meanwhile
                  *(lds_u32_p)(wsrc + sv_c8[0] * 28u) = (uint32_t)sv_w[0];
                  *(lds_u32_p)(wsrc + sv_c8[1] * 28u) = (uint32_t)sv_w[1];
                }
                asm volatile("" ::: "memory");
              }
              if (pre_mx) {
                const uint32_t a0 = wsrc + c8 * 56u;       // rows 2 c8, 2 c8 + 1: 28 bytes apart
                mwc0[0] = *(lds_u32_p)a0;
                mwc1[0] = *(lds_u32_p)(a0 + 28u);
              }
            } else if (pre_mx) {
#pragma unroll
              for (uint32_t q = 0; q < SP::nmix; ++q) {
                mrow0[q] = mix_row(q, c8 * 2u);
                mrow1[q] = mrow0[q] + SP::mix_m[q] * 4u;
                mwc0[q] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, vo_mix[q], mrow0[q], 0);
                mwc1[q] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, vo_mix[q], mrow1[q], 0);
              }
            }
            uint32_t row18n = 0, row20n = 0, w19n0 = 0, w19n1 = 0;
            if (SP::has_tail && pre_mx) {
              row18n = row18_load(c8 * 2u);
              row20n = row20_load(c8 * 2u);
              const uint32_t wp = *(lds_u32_p)(lds_off(S.a19) + ((c8 * 2u) & 254u) * 2u);
              w19n0 = uni(wp) & 0xffffu; w19n1 = uni(wp) >> 16;
            }
            // ---- the model wave's predictions for this bit
            bool okp = true;
            uint64_t tw0 = 0;
            if (PROF) tw0 = now();
            int p = wait_p(bs, yprev, okp);
            uint64_t tw1 = 0;
            if (PROF) { tw1 = now(); prof[bit == 3 ? 1 : bit == 4 ? 2 : bit == 7 ? 3 : 0] += tw1 - tw0; }
            if (UNLIKELY(!okp)) { helper_ok = false; pp_hsize = 0x1000 + bit; break; }   // (debug aid: where the partner was lost)
            // ---- mixers (Predictor.cs:302-316) and, for max, the serial tail on wave-uniform values
            int p15 = 0, p16 = 0, p17 = 0, p18 = 0, p19 = 0, p20 = 0;
            uint32_t sel18 = 0, sel20 = 0, ti18 = 0, ti20 = 0;
            int dtv18 = 0, dtv20 = 0;
            int pt = 0;                                  // max: the tail's predictions, one per lane (15, 16, 17, 19, 21 -> lanes 0..4)
            if constexpr (SP::id == 3) {
              const uint32_t ln = 15u;
              const uint32_t lw = lane;
              p = lane < 16u ? p : 0;                     // (the predictions are replicated over the DPP rows)
              const int w1hi = mw[1] >> 8;
              int t0 = __mul24(mw[0] >> 8, p);            // (component 15 itself publishes 0: lane 15 adds nothing)
              int t1 = lw == 15u ? 0 : __mul24(w1hi, p);
              t0 += dpp_shr(t0, 1); t1 += dpp_shr(t1, 1);
              t0 += dpp_shr(t0, 2); t1 += dpp_shr(t1, 2);
              t0 += dpp_shr(t0, 4); t1 += dpp_shr(t1, 4);
              t0 += dpp_shr(t0, 8); t1 += dpp_shr(t1, 8);
              p15 = med3i((int)rdlane((uint32_t)t0, ln) >> 8, -2048, 2047);
              p = lw == 15u ? p15 : p;                     // MIX 16's last input is MIX 15's output
              p16 = med3i(((int)rdlane((uint32_t)t1, ln) + (int)rdlane((uint32_t)w1hi, ln) * p15) >> 8, -2048, 2047);
              p17 = (w17 * p15 + (65536 - w17) * p16) >> 16;                 // MIX2 17 (Predictor.cs:291-301)
              auto sse = [&](int pin, uint32_t rowv, int &pout, uint32_t &sel, uint32_t &ti, int &dtv) __attribute__((always_inline)) {
                int pq = pin + 992;                                          // SSE (Predictor.cs:327-340)
                pq = pq < 0 ? 0 : pq > 1983 ? 1983 : pq;
                const uint32_t wt = (uint32_t)pq & 63u, iq = (uint32_t)pq >> 6;
                const uint32_t lo = (c8 & 1u) * 32u + iq;
                const uint32_t e0 = rdlane(rowv, lo), e1 = rdlane(rowv, lo + 1u);
                pout = stretch_u(((e0 >> 10) * (64u - wt) + (e1 >> 10) * wt) >> 13);
                sel = (wt >> 5) ? e1 : e0;
                ti = iq + (wt >> 5);
                dtv = S.dt[sel & 0x3ffu];
              };
              sse(p17, row18, p18, sel18, ti18, dtv18);
              p19 = (int)((int)w19 * p17 + (65536 - (int)w19) * p18) >> 16;  // MIX2 19
              sse(p19, row20, p20, sel20, ti20, dtv20);
              const int p21 = (w21 * p19 + (65536 - w21) * p20) >> 16;       // MIX2 21
              pt = lane == 0u ? p15 : lane == 1u ? p16 : lane == 2u ? p17 : lane == 3u ? p19 : lane == 4u ? p21 : 0;
            } else {
              int term = __mul24(mw[0] >> 8, p);         // lanes that do not feed the mixer hold weight 0
              term += dpp_shr(term, 1); term += dpp_shr(term, 2); term += dpp_shr(term, 4);
              if (SP::mix_m[0] > 8) term += dpp_shr(term, 8);
              pt = med3i((int)rdlane((uint32_t)term, SP::mix_m[0] > 8 ? 15u : 7u) >> 8, -2048, 2047);   // wave-uniform
            }
            // ---- decode
            uint32_t ps;
            int sqt = 0;                                 // squash of the tail's predictions (max: per lane; mid: the mixer's)
            if (SP::smem_ps) {
              const uint32_t pso = uni(((uint32_t)pt + 2048u) << 2);
              asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(ps) : "s"(ps_tab), "s"(pso));
              sqt = (int)(ps >> 17);
            } else {
              sqt = (int)*(lds_u16_p)((uint32_t)(uintptr_t)lds_squash + (uint32_t)(pt + 2048) * 2u);
              ps = (rdlane((uint32_t)sqt, SP::id == 3 ? 4u : 0u) * 2 + 1) << 16;
            }
            uint32_t jb = j;
            ZH_DEC_STEP(d, ps, jb, bad, rn);
            j = jb;
            const uint32_t y = uni(j & 1);
            c2_put0(&S.yv, bs << 8 | (j & 255u));         // the model wave takes it from here
            if (HELP && bit == 3) c2_put0(&S.mb_nib, bseq << 8 | (j & 15u));      // first nibble -> helper wave
            uint64_t tw2 = 0;
            if (PROF) { tw2 = now(); prof[4] += tw2 - tw1; }
            if (UNLIKELY(rn)) { if (dec_renorm(d, in, lane) && !err) err = bad ? (uint32_t)-ZH_E_CORRUPT : (uint32_t)-ZH_E_EOF; }
            const int ey = y ? 32767 : 0;
            // ---- training of what this wave owns (Predictor.cs:414-439, train :1031-1036)
            const int e = ey - sqt;                      // mid: one value; max: lane k = error of tail component k
#pragma unroll
            for (uint32_t q = 0; q < SP::nmix; ++q) {
              const int eq = __mul24(SP::id == 3 ? (int)rdlane((uint32_t)e, q) : e, mx_rate[q]) >> 4;
              const int nmw = med3i(mw[q] + ((__mul24(eq, p) + (1 << 12)) >> 13), -(1 << 19), (1 << 19) - 1);
              __builtin_amdgcn_raw_buffer_store_b32((uint32_t)nmw, rsrc, vo_mix[q], MIXLDS ? mix_row(q, c8) : uni(mrow[q]), 0);
              if constexpr (MIXLDS) {                     // ... and the copy in LDS (a byte that keeps the context re-reads it from there)
                if (mres || bit >= 2) { if (l_w) *(lds_u32_p)(lds_off(&S.mixblk[mpar][0]) + lane * 4u + c8 * 28u) = (uint32_t)nmw; }
                else { sv_w[bit & 1] = nmw; sv_c8[bit & 1] = c8; }
              }
            }
            if constexpr (SP::id == 3) {
              auto mix2_train = [&](int w, int rate, uint32_t ln, int pj_, int pk_) __attribute__((always_inline)) -> int {
                const int er = ((int)rdlane((uint32_t)e, ln) * rate) >> 5;
                w += (er * (pj_ - pk_) + (1 << 12)) >> 13;
                return w < 0 ? 0 : w > 65535 ? 65535 : w;
              };
              auto sse_train = [&](uint32_t pn, int dtv) __attribute__((always_inline)) -> uint32_t {
                const uint32_t count = pn & 0x3ffu;
                const int error = ey - (int)(pn >> 17);
                return pn + (((uint32_t)error * uni((uint32_t)dtv)) & 0xFFFFFC00u) + (count < C2Max::sse_limit);
              };
              w17 = mix2_train(w17, C2Max::rate17, 2, p15, p16);
              const uint32_t n18 = sse_train(sel18, dtv18);
              *(lds_u32_p)(lds_off(S.sse18) + ((c8 & 255u) * 32u + ti18) * 4u) = n18;
              w19 = (uint32_t)mix2_train((int)w19, C2Max::rate19, 3, p17, p18);
              *(lds_u16_p)(lds_off(S.a19) + a19i * 2u) = (uint16_t)w19;
              const uint32_t n20 = sse_train(sel20, dtv20);
              {
                const uint32_t off = uni(((((t_h20 + c8) * 32u + ti20) & sse20_mask) * 4u) + sse20_base);
                __builtin_amdgcn_raw_buffer_store_b32(n20, rsrc, lane == 0 ? 0u : kOob, off, 0);
              }
              w21 = mix2_train(w21, C2Max::rate21, 4, p19, p20);
            }
            c8 = c8 * 2u + y;
            if (pre_mx) {
#pragma unroll
              for (uint32_t q = 0; q < SP::nmix; ++q) {
                mw[q] = y ? mwc1[q] : mwc0[q];
                if (MIXLDS) mw[q] = l_w ? mw[q] : 0;
                else mrow[q] = uni(y ? mrow1[q] : mrow0[q]);
              }
            }
            if (SP::has_tail && pre_mx) {
              row18 = row18n; row20 = row20n;
              w19 = y ? w19n1 : w19n0; a19i = c8 & 255u;
            }
            yprev = y;
            ++bs;
            if (PROF) { const uint64_t tw3 = now(); prof[5] += tw3 - tw2; prof[6] += tw0 - tlast; tlast = tw3; }
          }
          if (UNLIKELY(!helper_ok)) { status = -24; break; }      // ZPAQHIP_E_HIP: a partner wavefront stopped answering (cannot happen by design)
          if (UNLIKELY(err | bad)) { status = err ? -(int)err : ZH_E_CORRUPT; break; }
          c = (int)(c8 - 256);
          c2_put0(&S.mb_byte, bseq << 8 | (uint32_t)c);   // the helper wave commits this byte's candidate
        }

        // ---- PostProcessor.write(c) (PostProcessor.cs:37-86): while the model wave works on the byte boundary
        c = (int)uni((uint32_t)c);
        uint64_t tp0 = 0;
        if (PROF) tp0 = now();
        if (LIKELY(pp_state == 1)) {
          if (LIKELY(c >= 0)) out_put(ob, (uint32_t)c, lane);
        } else if (pp_state == 5) {
          int rc;
          if (pnative == ZH_NATIVE_PCOMP_E8E9)
            rc = zh_native_pcomp_e8e9(pa, pb, pc_, pd, pf, (uint32_t)c, (lds_u8_p)lds_off(S.pmreg), pz.mmask, (lds_u32_p)lds_off(S.phreg), pz.hmask, S.pr, &sink, L.budget);
          else rc = vm_run(pz, (uint32_t)c, &sink, L.budget);
          rc = (int)uni((uint32_t)rc);
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
          pzbuf[pp_len] = (uint8_t)c;
          if ((int)++pp_len == pp_hsize) {
            c2_wave_sync();
            pz.prog = pzbuf; pz.len = pp_len;
            pz.a = pz.b = pz.c = pz.d = pz.f = 0;
            pnative = p_lds ? uni(zh_native_lookup(pzbuf, pp_len)) : 0;
            pp_state = 5;
          }
        }
        if (c < 0) break;
        uint64_t tp1 = 0;
        if (PROF) { tp1 = now(); prof[10] += tp1 - tp0; }

        // ---- byte boundary, this wave's part: h[] of the mixers from the helper wave, the rows for c8 = 1
        if (helper_ok) { helper_ok = c2_wait(&S.mb_ready, bseq); if (!helper_ok) pp_hsize = 0x2000; }
        if (PROF) { const uint64_t t2 = now(); prof[9] += t2 - tp1; prof[8] += t2 - tbyte; tbyte = t2; prof[11] += 1; }
        if (!helper_ok) { status = -24; break; }
        asm volatile("" ::: "memory");
        {
          const uint32_t lo = (uint32_t)c & 15u;
#pragma unroll
          for (uint32_t q = 0; q < SP::nmix; ++q) mix_set(q, S.hspec[SP::mix_lane[q] & ((1u << SP::hh) - 1u)][lo]);
          if (SP::has_tail) t_h20 = uni(S.hspec[SP::has_tail ? 20 : 0][lo]);
          byte_start_rows(HELP == 1, lo);
        }
        ++bseq;
      }

      if (pp_state != 5) out_flush(ob, lane);
      const uint64_t produced = pp_state == 5 ? uni64(sink.len) : ob.len;
      if (!status && produced > b_out_cap) status = ZH_E_OUTPUT_FULL;
      if (status && status != ZH_E_OUTPUT_FULL) failed = 1;
      if (lane == 0) {
        ZhSegResult res;
        res.status = status; res.pp_state = (uint32_t)pp_state | (uint32_t)pp_hsize << 8;
        res.out_off = b_out_off + produced0; res.out_len = produced - produced0;
        res.in_used = in_pos(in) - seg_off;
        L.results[si] = res;
      }
    }
    if (PROF && lane == 0 && L.debug)
      for (int i = 0; i < 12; ++i) atomicAdd((unsigned long long *)&L.debug[i], (unsigned long long)prof[i]);
    // the partner waves leave the block; their last LDS writes are in before this wave re-initialises for the next one
    ++cmd_seq;
    c2_put0(&S.mb_cmd, cmd_seq << 2 | kC2End);
    (void)c2_wait(&S.mb_ack, cmd_seq << 2 | kC2End);
    (void)c2_wait(&S.mb_ack2, cmd_seq << 2 | kC2End);
    c2_wave_sync();
  }
  ++cmd_seq;
  c2_put0(&S.mb_cmd, cmd_seq << 2 | kC2Exit);
}

template <class SP, bool SPEC, bool PROF, class LDS>
__device__ __forceinline__ void decode_chain3_body(const ZhLaunch &L, LDS &S) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = uni(threadIdx.x >> 6);
  if (wave == 0) {  // model-independent tables -> LDS (ZhTables: squash, stretch, dt, ns), the MATCH prediction table, the mailboxes
    const ZhTables *T = L.tables;
    for (uint32_t i = lane; i < 32768 / 8; i += 64) reinterpret_cast<uint4 *>(S.stretch)[i] = reinterpret_cast<const uint4 *>(T->stretch)[i];
    for (uint32_t i = lane; i < 4096 / 8; i += 64) reinterpret_cast<uint4 *>(S.squash)[i] = reinterpret_cast<const uint4 *>(T->squash)[i];
    for (uint32_t i = lane; i < 1024 / 4; i += 64) reinterpret_cast<uint4 *>(S.dt)[i] = reinterpret_cast<const uint4 *>(T->dt)[i];
    for (uint32_t i = lane; i < 1024 / 16; i += 64) reinterpret_cast<uint4 *>(S.ns)[i] = reinterpret_cast<const uint4 *>(T->ns)[i];
    if (lane == 0) {
      S.zrow = v4u_{0, 0, 0, 0};
      S.mb_cmd = 0; S.mb_ack = 0; S.mb_ack2 = 0; S.mb_nib = 0; S.mb_byte = 0; S.mb_ready = 0; S.yv = 0;
    }
    __builtin_amdgcn_s_waitcnt(0);
    for (uint32_t i = lane; i < 256; i += 64) {            // the two predictions of a match of length i (Predictor.cs:273-287)
      const int dk = T->dt2k[i];
      const int lo = T->stretch[dk & 32767], hi = T->stretch[(-dk) & 32767];
      S.pm01[i] = i ? ((uint32_t)(uint16_t)lo | (uint32_t)(uint16_t)hi << 16) : 0u;
    }
  }
  __syncthreads();                                       // the only workgroup barrier of the kernel
  if (wave == 0) c3_decoder<SP, SPEC, PROF>(L, S, lane);
  else if (wave == 1) c3_model<SP, SPEC, PROF>(L, S, lane);
  else c2_helper<SP>(L, S, lane, blockIdx.x);
}

}  // namespace

#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "zh_chain3.hip: the multi-wave protocol is written for gfx9-family CUs (one LDS, shared vector L1, in-order vmcnt)"
#endif
#define ZH_CHAIN3_KERNEL(name, spec, speculate, prof)                                  \
  extern "C" __global__ __launch_bounds__(192) void name(ZhLaunch L) {                 \
    typedef C2LdsT<spec::has_tail, spec::helper, spec::id == 2> Lds;                   \
    __shared__ Lds S;                                                                  \
    decode_chain3_body<spec, speculate, prof, Lds>(L, S);                              \
  }
ZH_CHAIN3_KERNEL(zh_decode_c3_mid, C2Mid, true, false)
ZH_CHAIN3_KERNEL(zh_decode_c3_max, C2Max, true, false)
ZH_CHAIN3_KERNEL(zh_decode_c3_mid_sync, C2Mid, false, false)     // no speculation: the model wave waits for every bit (cross-check, A/B)
ZH_CHAIN3_KERNEL(zh_decode_c3_max_sync, C2Max, false, false)
ZH_CHAIN3_KERNEL(zh_decode_c3_mid_prof, C2Mid, true, true)
ZH_CHAIN3_KERNEL(zh_decode_c3_max_prof, C2Max, true, true)

// spec: 2 mid, 3 max (zh_chain_spec.h ids).  variant 1 = without speculation, 2 = diagnostic build (ZPAQHIP_PROF).
extern "C" hipError_t zh_launch_chain3(const ZhLaunch *L, uint32_t grid, hipStream_t stream, uint32_t spec, int variant) {
  void (*k)(ZhLaunch) = spec == 2 ? (variant == 2 ? zh_decode_c3_mid_prof : variant ? zh_decode_c3_mid_sync : zh_decode_c3_mid)
                       : spec == 3 ? (variant == 2 ? zh_decode_c3_max_prof : variant ? zh_decode_c3_max_sync : zh_decode_c3_max) : nullptr;
  if (!k) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k, dim3(grid), dim3(192), 0, stream, *L);     // decoder wave + model wave + helper wave
  return hipGetLastError();
}
extern "C" int zh_chain3_has(uint32_t spec) { return spec == 2 || spec == 3; }
