// zh_api.cpp — C-ABI entry points of libzpaqhip.so (include/zpaqhip.h) and the
// host driver that turns a block table into kernel launches.
//
// There is deliberately no CPU decode path in this file or library: every
// decoded byte comes out of a HIP kernel, and every entry point that needs a
// GPU fails with ZPAQHIP_E_NO_DEVICE when none is usable.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "zh_host.h"

extern "C" hipError_t zh_launch_generic(const ZhLaunch *L, uint32_t grid, hipStream_t stream);
extern "C" hipError_t zh_launch_cm(const ZhLaunch *L, uint32_t grid, hipStream_t stream);
extern "C" hipError_t zh_launch_sha1(const uint8_t *data, const uint64_t *seg, uint32_t n_seg, uint32_t *digest, hipStream_t stream);
extern "C" hipError_t zh_launch_chain(const ZhLaunch *L, uint32_t grid, hipStream_t stream, uint32_t spec, int prof);
extern "C" hipError_t zh_launch_cm_prof(const ZhLaunch *L, uint32_t grid, hipStream_t stream);
extern "C" hipError_t zh_launch_chain2(const ZhLaunch *L, uint32_t grid, hipStream_t stream, uint32_t spec, int prof);
extern "C" int zh_chain2_has(uint32_t spec);


using namespace zh;

namespace {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) cap = n;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct zpaqhip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  DevBuf tables, arena, models, code, bdesc, sdesc, results, queue, in, out;
  zpaqhip_stats stats{};
  std::vector<uint32_t> raw_pp;           // last decode: per segment pp_state | PCOMP length << 8, as the kernels report it
};

namespace {

#define HIPCHK(expr)                                                          \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) {                                                   \
      char m_[112];                                                           \
      snprintf(m_, sizeof m_, "HIP: %s (%s)", hipGetErrorString(e_), #expr);  \
      set_err(err, ZPAQHIP_E_HIP, -1, -1, m_);                                \
      return ZPAQHIP_E_HIP;                                                   \
    }                                                                         \
  } while (0)

constexpr size_t kQueueBytes = 256, kDebugBytes = 64;    // work-queue heads (32 B per family) and the *_prof kernels' sums
constexpr uint64_t kPpOnlyMagic = 0x5A50505F4F4E4C59ull;   // internal: zpaqhip_block_pcomp -> decode_blocks_device ("ZPP_ONLY")

zpaqhip_opts resolve_opts(const zpaqhip_opts *o) {
  zpaqhip_opts r;
  memset(&r, 0, sizeof r);
  if (o) memcpy(&r, o, std::min<size_t>(sizeof r, o->struct_size ? o->struct_size : sizeof r));
  if (!r.zpaql_budget) r.zpaql_budget = 1ull << 32;
  return r;
}

}  // namespace

extern "C" {

int zpaqhip_version(void) { return ZPAQHIP_ABI_VERSION; }

const char *zpaqhip_strerror(int status) { return status_message(status); }

int zpaqhip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int zpaqhip_ctx_create(int device, zpaqhip_ctx **out, zpaqhip_err *err) {
  if (!out) { set_err(err, ZPAQHIP_E_ARG, -1, -1); return ZPAQHIP_E_ARG; }
  *out = nullptr;
  if (!host_tables_ok()) {
    set_err(err, ZPAQHIP_E_NO_DEVICE, -1, -1, "model tables failed their reference checksums");
    return ZPAQHIP_E_NO_DEVICE;
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
    set_err(err, ZPAQHIP_E_NO_DEVICE, -1, -1);
    return ZPAQHIP_E_NO_DEVICE;
  }
  HIPCHK(hipSetDevice(device));
  zpaqhip_ctx *c = new zpaqhip_ctx();
  c->device = device;
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess ||
      (e = c->tables.reserve(sizeof(ZhTables))) != hipSuccess ||
      (e = hipMemcpy(c->tables.p, &host_tables(), sizeof(ZhTables), hipMemcpyHostToDevice)) != hipSuccess) {
    char m[112];
    snprintf(m, sizeof m, "HIP: %s (context setup)", hipGetErrorString(e));
    set_err(err, ZPAQHIP_E_HIP, -1, -1, m);
    zpaqhip_ctx_destroy(c);
    return ZPAQHIP_E_HIP;
  }
  *out = c;
  return ZPAQHIP_OK;
}

void zpaqhip_ctx_destroy(zpaqhip_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (DevBuf *b : {&c->tables, &c->arena, &c->models, &c->code, &c->bdesc, &c->sdesc, &c->results, &c->queue, &c->in, &c->out})
    b->release();
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int zpaqhip_last_stats(const zpaqhip_ctx *c, zpaqhip_stats *out) {
  if (!c || !out) return ZPAQHIP_E_ARG;
  *out = c->stats;
  return ZPAQHIP_OK;
}

int zpaqhip_scan(const uint8_t *in, size_t in_len, zpaqhip_block *blocks, size_t block_cap, size_t *n_blocks,
                 zpaqhip_segment *segs, size_t seg_cap, size_t *n_segs, zpaqhip_err *err) {
  if ((!in && in_len) || !n_blocks || !n_segs) { set_err(err, ZPAQHIP_E_ARG, -1, -1); return ZPAQHIP_E_ARG; }
  ScanOut so;
  int rc = scan_stream(in, in_len, so, err);
  *n_blocks = so.blocks.size();
  *n_segs = so.segs.size();
  if (rc) return rc;
  if (so.blocks.size() > block_cap || so.segs.size() > seg_cap) {
    if (block_cap || seg_cap) set_err(err, ZPAQHIP_E_ARG, -1, -1, "block/segment table too small");
    return (block_cap || seg_cap) ? ZPAQHIP_E_ARG : ZPAQHIP_OK;
  }
  if (blocks && !so.blocks.empty()) memcpy(blocks, so.blocks.data(), so.blocks.size() * sizeof(zpaqhip_block));
  if (segs && !so.segs.empty()) memcpy(segs, so.segs.data(), so.segs.size() * sizeof(zpaqhip_segment));
  return ZPAQHIP_OK;
}

int zpaqhip_read_device_tables(zpaqhip_ctx *c, uint16_t *squash, int16_t *stretch, int32_t *dt, int32_t *dt2k,
                               uint8_t *ns, zpaqhip_err *err) {
  if (!c) { set_err(err, ZPAQHIP_E_ARG, -1, -1); return ZPAQHIP_E_ARG; }
  HIPCHK(hipSetDevice(c->device));
  std::vector<uint8_t> buf(sizeof(ZhTables));
  HIPCHK(hipMemcpy(buf.data(), c->tables.p, sizeof(ZhTables), hipMemcpyDeviceToHost));
  const ZhTables *t = reinterpret_cast<const ZhTables *>(buf.data());
  if (squash) memcpy(squash, t->squash, sizeof t->squash);
  if (stretch) memcpy(stretch, t->stretch, sizeof t->stretch);
  if (dt) memcpy(dt, t->dt, sizeof t->dt);
  if (dt2k) memcpy(dt2k, t->dt2k, sizeof t->dt2k);
  if (ns) memcpy(ns, t->ns, sizeof t->ns);
  return ZPAQHIP_OK;
}

}  // extern "C"

// `h_hdrs`: the block headers are needed on the host to build the model
// descriptors; they are fetched from the device stream (a few hundred bytes per
// distinct model) so the ABI stays a pure device-buffer interface.
// The call has two halves so that the whole-stream forms can keep the host busy (callbacks, scanning, copies of the
// neighbouring batches) while a batch's kernels run: decode_launch() enqueues everything on `stream` and returns,
// decode_finish() waits for the stream and turns the device results into zpaqhip_seg_result records.  One launch may be
// outstanding per context (the descriptor / result / arena buffers belong to the context).
struct zh_pending {
  std::vector<uint32_t> sel;
  const zpaqhip_block *blocks = nullptr;
  const zpaqhip_segment *segs = nullptr;
  size_t n_segs = 0;
  hipStream_t stream = nullptr;
  uint64_t total_in = 0, total_model = 0;
  uint32_t launches = 0, slots = 0, kind_used = 0;
  bool prof = false, active = false;
};

static int decode_launch(zpaqhip_ctx *c, const void *d_in, const uint8_t *h_in, size_t in_len,
                                 const zpaqhip_block *blocks,
                                 size_t n_blocks, const zpaqhip_segment *segs, size_t n_segs, const uint32_t *ids,
                                 size_t n_ids, void *d_out, const uint64_t *out_off, const uint64_t *out_cap,
                                 const zpaqhip_opts &opts, hipStream_t stream, zh_pending &P,
                                 zpaqhip_err *err) {
  P = zh_pending();
  P.blocks = blocks; P.segs = segs; P.n_segs = n_segs; P.stream = stream;
  HIPCHK(hipSetDevice(c->device));
  memset(&c->stats, 0, sizeof c->stats);

  std::vector<uint32_t> &sel = P.sel;
  if (ids) sel.assign(ids, ids + n_ids);
  else { sel.resize(n_blocks); std::iota(sel.begin(), sel.end(), 0u); }
  if (sel.empty()) return ZPAQHIP_OK;
  for (uint32_t b : sel)
    if (b >= n_blocks) { set_err(err, ZPAQHIP_E_ARG, (int)b, -1, "block id out of range"); return ZPAQHIP_E_ARG; }

  // ---- models: one per distinct header
  std::map<std::string, uint32_t> model_of;
  std::vector<ZhModel> models;
  std::vector<uint8_t> code;
  std::vector<ZhBlockDesc> bd(sel.size());
  std::vector<ZhSegDesc> sd(n_segs);
  std::vector<uint8_t> hdr;
  uint64_t total_in = 0, total_model = 0;
  for (size_t k = 0; k < sel.size(); ++k) {
    const zpaqhip_block &b = blocks[sel[k]];
    if (b.hdr_off + b.hdr_len > in_len || (uint64_t)b.first_seg + b.n_seg > n_segs) {
      set_err(err, ZPAQHIP_E_ARG, (int)sel[k], -1, "block table does not match the stream");
      return ZPAQHIP_E_ARG;
    }
    hdr.resize(b.hdr_len);
    if (h_in) memcpy(hdr.data(), h_in + b.hdr_off, b.hdr_len);
    else HIPCHK(hipMemcpy(hdr.data(), (const uint8_t *)d_in + b.hdr_off, b.hdr_len, hipMemcpyDeviceToHost));
    std::string key((const char *)hdr.data(), hdr.size());
    auto it = model_of.find(key);
    if (it == model_of.end()) {
      ZhModel m;
      int rc = build_model(hdr.data(), hdr.size(), m, code, err);
      if (rc) { if (err) err->block = (int)sel[k]; return rc; }
      it = model_of.emplace(key, (uint32_t)models.size()).first;
      models.push_back(m);
    }
    const ZhModel &m = models[it->second];
    total_model += m.arena_bytes;
    bd[k].model = it->second;
    bd[k].first_seg = b.first_seg;
    bd[k].n_seg = b.n_seg;
    bd[k].out_off = out_off ? out_off[k] : 0;
    bd[k].out_cap = out_cap ? out_cap[k] : 0;
    for (uint32_t s = 0; s < b.n_seg; ++s) {
      const zpaqhip_segment &sg = segs[b.first_seg + s];
      if (sg.data_off + sg.data_len > in_len) {
        set_err(err, ZPAQHIP_E_ARG, (int)sel[k], (int)(b.first_seg + s), "segment table does not match the stream");
        return ZPAQHIP_E_ARG;
      }
      sd[b.first_seg + s].in_off = sg.data_off;
      sd[b.first_seg + s].in_len = sg.data_len;
      total_in += sg.data_len;
    }
  }
  std::vector<uint64_t> weight(sel.size(), 0);
  for (size_t k = 0; k < sel.size(); ++k)
    for (uint32_t i = 0; i < bd[k].n_seg; ++i) weight[k] += sd[bd[k].first_seg + i].in_len;

  // ---- kernel family per block (opts.kernel == 1 forces the generic kernel)
  auto family = [&](size_t k) -> uint32_t {
    uint32_t f = models[bd[k].model].kind & 255u;
    if (opts.kernel == 1) f = ZH_FAM_GENERIC;              // force the generic kernel
    if (opts.kernel == 3 && f == ZH_FAM_CM1) f = ZH_FAM_CHAIN;   // force the lane-per-component kernel
    if (opts.kernel == 4 && f > ZH_FAM_CHAIN) f = ZH_FAM_CHAIN;  // lane-per-component kernel without model specialisation
    return f;
  };
  std::vector<std::vector<uint32_t>> groups(ZH_NFAM);
  for (size_t k = 0; k < sel.size(); ++k) groups[family(k)].push_back((uint32_t)k);

  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  free_b += c->arena.cap;                               // our own cached arena is reusable
  const uint64_t mem_budget = free_b > (1ull << 30) ? free_b - (1ull << 30) : free_b / 2;

  HIPCHK(c->models.reserve(models.size() * sizeof(ZhModel)));
  HIPCHK(c->code.reserve(code.size() + 16));
  HIPCHK(c->bdesc.reserve(sel.size() * sizeof(ZhBlockDesc)));
  HIPCHK(c->sdesc.reserve(sd.size() * sizeof(ZhSegDesc)));
  HIPCHK(c->results.reserve(n_segs * sizeof(ZhSegResult)));
  static_assert(32 * ZH_NFAM <= kQueueBytes, "one 32-byte work-queue head per kernel family");
  HIPCHK(c->queue.reserve(kQueueBytes + kDebugBytes));   // [heads | diagnostic cycle sums]: the two never overlap
  HIPCHK(hipMemcpyAsync(c->models.p, models.data(), models.size() * sizeof(ZhModel), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(c->code.p, code.data(), code.size(), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(c->sdesc.p, sd.data(), sd.size() * sizeof(ZhSegDesc), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemsetAsync(c->results.p, 0xff, n_segs * sizeof(ZhSegResult), stream));
  HIPCHK(hipMemsetAsync(c->queue.p, 0, kQueueBytes + kDebugBytes, stream));

  // arena: sized for the most demanding group
  uint32_t slots_of[ZH_NFAM] = {};
  uint64_t stride_of[ZH_NFAM], arena_need = 0;
  for (auto &x : stride_of) x = 256;
  for (uint32_t g = 0; g < ZH_NFAM; ++g) {
    if (groups[g].empty()) continue;
    for (uint32_t k : groups[g]) stride_of[g] = std::max<uint64_t>(stride_of[g], models[bd[k].model].arena_bytes);
    uint64_t max_slots = mem_budget / stride_of[g];
    if (max_slots == 0) { set_err(err, ZPAQHIP_E_DEVICE_MEM, -1, -1, "Out of memory"); return ZPAQHIP_E_DEVICE_MEM; }
    uint32_t want = opts.max_concurrent ? opts.max_concurrent : 256u;   // one wave per CU by default
    slots_of[g] = (uint32_t)std::min<uint64_t>({(uint64_t)want, max_slots, (uint64_t)groups[g].size()});
    arena_need = std::max<uint64_t>(arena_need, slots_of[g] * stride_of[g]);
  }
  HIPCHK(c->arena.reserve((size_t)arena_need));

  std::vector<ZhBlockDesc> bd_sorted;
  bd_sorted.reserve(sel.size());
  size_t base_of[ZH_NFAM] = {};
  for (uint32_t g = 0; g < ZH_NFAM; ++g) {
    // Longest block first: the work queue then balances the tail (LPT order).
    std::stable_sort(groups[g].begin(), groups[g].end(), [&](uint32_t a, uint32_t b) { return weight[a] > weight[b]; });
    base_of[g] = bd_sorted.size();
    for (uint32_t k : groups[g]) bd_sorted.push_back(bd[k]);
  }
  HIPCHK(hipMemcpyAsync(c->bdesc.p, bd_sorted.data(), bd_sorted.size() * sizeof(ZhBlockDesc), hipMemcpyHostToDevice, stream));

  uint32_t launches = 0, slots = 0, kind_used = 0;
  HIPCHK(hipEventRecord(c->ev0, stream));
  for (uint32_t g = 0; g < ZH_NFAM; ++g) {
    if (groups[g].empty()) continue;
    ZhLaunch L;
    memset(&L, 0, sizeof L);
    L.in = (const uint8_t *)d_in;
    L.in_total = in_len;
    L.models = (const ZhModel *)c->models.p;
    L.code = (const uint8_t *)c->code.p;
    L.blocks = (const ZhBlockDesc *)c->bdesc.p + base_of[g];
    L.segs = (const ZhSegDesc *)c->sdesc.p;
    L.results = (ZhSegResult *)c->results.p;
    L.out = (uint8_t *)d_out;
    L.arena = (uint8_t *)c->arena.p;
    L.arena_stride = stride_of[g];
    L.tables = (const ZhTables *)c->tables.p;
    L.queue = (uint32_t *)c->queue.p + 8 * g;           // one work-queue head per launch
    L.n_blocks = (uint32_t)groups[g].size();
    L.budget = opts.zpaql_budget;
    L.flags = opts.reserved[0] == kPpOnlyMagic ? ZH_LAUNCH_PP_ONLY : 0u;
    const bool prof = getenv("ZPAQHIP_PROF") != nullptr;   // diagnostic build with in-kernel stamps
    if (prof) L.debug = (uint64_t *)((uint8_t *)c->queue.p + kQueueBytes);
    if (g == ZH_FAM_CM1 && prof) HIPCHK(zh_launch_cm_prof(&L, slots_of[g], stream));
    else if (g == ZH_FAM_CM1) HIPCHK(zh_launch_cm(&L, slots_of[g], stream));
    else if (g > ZH_FAM_CHAIN && zh_chain2_has(g - ZH_FAM_CHAIN) && opts.kernel != 5)   // per-model bit loop (zh_chain2.hip)
      HIPCHK(zh_launch_chain2(&L, slots_of[g], stream, g - ZH_FAM_CHAIN, prof));
    else if (g >= ZH_FAM_CHAIN) HIPCHK(zh_launch_chain(&L, slots_of[g], stream, g - ZH_FAM_CHAIN, prof));
    else HIPCHK(zh_launch_generic(&L, slots_of[g], stream));
    ++launches;
    slots = std::max(slots, slots_of[g]);
    kind_used = std::max(kind_used, std::min(g, (uint32_t)ZH_FAM_CHAIN) + 1);
  }
  HIPCHK(hipEventRecord(c->ev1, stream));
  P.total_in = total_in; P.total_model = total_model;
  P.launches = launches; P.slots = slots; P.kind_used = kind_used;
  P.prof = getenv("ZPAQHIP_PROF") != nullptr;
  P.active = true;
  return ZPAQHIP_OK;
}

static int decode_finish(zpaqhip_ctx *c, zh_pending &P, zpaqhip_seg_result *results, zpaqhip_err *err) {
  if (!P.active) return ZPAQHIP_OK;                     // nothing was selected
  P.active = false;
  const std::vector<uint32_t> &sel = P.sel;
  const zpaqhip_block *blocks = P.blocks;
  const zpaqhip_segment *segs = P.segs;
  const size_t n_segs = P.n_segs;
  hipStream_t stream = P.stream;
  const uint64_t total_in = P.total_in, total_model = P.total_model;
  const uint32_t launches = P.launches, slots = P.slots, kind_used = P.kind_used;
  if (P.prof) {
    uint64_t dbg[8];
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipMemcpy(dbg, (uint8_t *)c->queue.p + kQueueBytes, 64, hipMemcpyDeviceToHost));
    fprintf(stderr, "ZPAQHIP_PROF cycles:");
    for (int i = 0; i < 8; ++i) fprintf(stderr, " %llu", (unsigned long long)dbg[i]);
    fprintf(stderr, "\n");
  }
  std::vector<ZhSegResult> res(n_segs);
  c->raw_pp.assign(n_segs, 0);
  HIPCHK(hipMemcpyAsync(res.data(), c->results.p, n_segs * sizeof(ZhSegResult), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));

  int first_bad = ZPAQHIP_OK;
  uint64_t total_out = 0;
  for (size_t k = 0; k < sel.size(); ++k) {
    const zpaqhip_block &b = blocks[sel[k]];
    for (uint32_t s = 0; s < b.n_seg; ++s) {
      const uint32_t si = b.first_seg + s;
      if (res[si].status == ZH_E_STOPPED) res[si].status = ZPAQHIP_OK, res[si].in_used = segs[si].data_len;   // ended on request
      results[si].status = res[si].status;
      results[si].pp_state = res[si].pp_state & 255u;
      c->raw_pp[si] = res[si].pp_state;
      results[si].out_off = res[si].out_off;
      results[si].out_len = res[si].out_len;
      results[si].in_used = res[si].in_used;
      // A decoder that stopped anywhere but at the scanned end of the coded data means a
      // damaged stream: the reference would now read its end-of-segment marker from the
      // wrong place (Decompresser.cs:163-194).
      if (res[si].status == ZPAQHIP_OK && res[si].in_used != segs[si].data_len) results[si].status = res[si].status = ZPAQHIP_E_SEGEND;
      total_out += res[si].out_len;
      if (res[si].status != ZPAQHIP_OK && res[si].status != ZPAQHIP_E_OUTPUT_FULL && first_bad == ZPAQHIP_OK &&
          res[si].status != ZH_E_SKIPPED) {
        first_bad = res[si].status;
        set_err(err, first_bad, (int)sel[k], (int)si);
      }
    }
  }
  c->stats.kernel_ms = ms;
  c->stats.blocks = sel.size();
  c->stats.in_bytes = total_in;
  c->stats.out_bytes = total_out;
  c->stats.model_bytes = total_model;
  c->stats.launches = launches;
  c->stats.concurrent = slots;
  c->stats.kernel_kind = kind_used;
  return first_bad;
}

extern "C" int zpaqhip_decode_blocks_device(zpaqhip_ctx *c, const void *d_in, const uint8_t *h_in, size_t in_len,
                                 const zpaqhip_block *blocks,
                                 size_t n_blocks, const zpaqhip_segment *segs, size_t n_segs, const uint32_t *ids,
                                 size_t n_ids, void *d_out, const uint64_t *out_off, const uint64_t *out_cap,
                                 zpaqhip_seg_result *results, const zpaqhip_opts *opts_in, void *hip_stream,
                                 zpaqhip_err *err) {
  if (!c || !blocks || !segs || !results || (!d_in && in_len) || (ids == nullptr && n_ids != 0 && n_ids != n_blocks)) {
    set_err(err, ZPAQHIP_E_ARG, -1, -1);
    return ZPAQHIP_E_ARG;
  }
  const zpaqhip_opts opts = resolve_opts(opts_in);
  zh_pending P;
  int rc = decode_launch(c, d_in, h_in, in_len, blocks, n_blocks, segs, n_segs, ids, n_ids, d_out, out_off, out_cap, opts,
                         hip_stream ? (hipStream_t)hip_stream : c->stream, P, err);
  if (rc) return rc;
  return decode_finish(c, P, results, err);
}


// ---------------------------------------------------------------------------
// Whole-stream forms
// ---------------------------------------------------------------------------
namespace {

// Decodes a host-resident stream; the plaintext ends up in ctx->out (device),
// laid out in stream order.  Returns total plaintext length in *total.
bool data_error(int rc) {                 // per-segment outcomes, as opposed to failures of the call itself
  return rc < 0 && rc != ZPAQHIP_E_HIP && rc != ZPAQHIP_E_ARG && rc != ZPAQHIP_E_DEVICE_MEM && rc != ZPAQHIP_E_NO_DEVICE &&
         rc != ZPAQHIP_E_HEADER && rc != ZPAQHIP_E_COMPONENT && rc != ZPAQHIP_E_HM_TOO_BIG;
}

int decode_stream_to_device(zpaqhip_ctx *c, const uint8_t *in, size_t in_len, const zpaqhip_opts &opts, ScanOut &so,
                            std::vector<zpaqhip_seg_result> &res, uint64_t *total, zpaqhip_err *err,
                            bool tolerate = false) {
  int rc = scan_stream(in, in_len, so, err);
  if (rc) return rc;
  *total = 0;
  res.assign(so.segs.size(), zpaqhip_seg_result{});
  if (so.blocks.empty()) return ZPAQHIP_OK;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(c->in.reserve(in_len + 16));
  hipEvent_t e0 = c->ev0, e1 = c->ev1;
  HIPCHK(hipEventRecord(e0, c->stream));
  HIPCHK(hipMemcpyAsync(c->in.p, in, in_len, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipEventRecord(e1, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  float h2d = 0;
  HIPCHK(hipEventElapsedTime(&h2d, e0, e1));

  const size_t nb = so.blocks.size();
  std::vector<uint64_t> off(nb), cap(nb);
  // pass 1: place by the decimal sizes in the segment comments (LibZPAQ.compress
  // writes them, LICENSE:57-58); blocks without a size are decoded in count-only mode.
  uint64_t pos = 0;
  bool exact_possible = true;
  for (size_t b = 0; b < nb; ++b) {
    uint64_t hint = so.blocks[b].usize_hint;
    if (hint == UINT64_MAX || hint > (1ull << 40)) { hint = 0; exact_possible = false; }
    off[b] = pos; cap[b] = hint; pos += hint;
  }
  HIPCHK(c->out.reserve((size_t)pos + 16));
  rc = zpaqhip_decode_blocks_device(c, c->in.p, in, in_len, so.blocks.data(), nb, so.segs.data(), so.segs.size(), nullptr, 0,
                                    c->out.p, off.data(), cap.data(), res.data(), &opts, c->stream, err);
  zpaqhip_stats st1 = c->stats;
  if (rc && !(tolerate && data_error(rc))) return rc;
  // did every block land exactly where the final layout wants it?
  std::vector<uint64_t> real(nb, 0);
  bool ok = exact_possible;
  uint64_t pos2 = 0;
  for (size_t b = 0; b < nb; ++b) {
    for (uint32_t s = 0; s < so.blocks[b].n_seg; ++s) real[b] += res[so.blocks[b].first_seg + s].out_len;
    if (real[b] != cap[b] || pos2 != off[b]) ok = false;
    pos2 += real[b];
  }
  *total = pos2;
  if (!ok) {
    // pass 2: sizes are now known exactly; decode again into the final layout.
    pos = 0;
    for (size_t b = 0; b < nb; ++b) { off[b] = pos; cap[b] = real[b]; pos += real[b]; }
    HIPCHK(c->out.reserve((size_t)pos + 16));
    rc = zpaqhip_decode_blocks_device(c, c->in.p, in, in_len, so.blocks.data(), nb, so.segs.data(), so.segs.size(), nullptr,
                                      0, c->out.p, off.data(), cap.data(), res.data(), &opts, c->stream, err);
    c->stats.kernel_ms += st1.kernel_ms;
    c->stats.launches += st1.launches;
    if (rc && !(tolerate && data_error(rc))) return rc;
  }
  c->stats.h2d_ms = h2d;
  return ZPAQHIP_OK;
}

// SHA-1 of every decoded segment that stores one, computed ON THE DEVICE over ctx->out (zh_sha1_dev.hip) while the
// plaintext is still there; bad[s] = 1 where the digest differs from the stored one (Decompresser.cs:183-191).
int sha1_mismatches(zpaqhip_ctx *c, const ScanOut &so, const std::vector<zpaqhip_seg_result> &res, std::vector<int> &bad,
                    zpaqhip_err *err) {
  const size_t ns = so.segs.size();
  bad.assign(ns, 0);
  std::vector<uint64_t> tab;
  std::vector<uint32_t> which;
  for (size_t s = 0; s < ns; ++s) {
    if (!(so.segs[s].flags & 1) || res[s].status != ZPAQHIP_OK) continue;
    tab.push_back(res[s].out_off);
    tab.push_back(res[s].out_len);
    which.push_back((uint32_t)s);
  }
  if (which.empty()) return ZPAQHIP_OK;
  const size_t tab_bytes = tab.size() * 8, dig_bytes = which.size() * 20;
  HIPCHK(c->sdesc.reserve(tab_bytes + dig_bytes + 64));          // the descriptors of the finished decode are not needed any more
  uint8_t *d_tab = (uint8_t *)c->sdesc.p, *d_dig = d_tab + ((tab_bytes + 15) & ~(size_t)15);
  HIPCHK(hipMemcpyAsync(d_tab, tab.data(), tab_bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(zh_launch_sha1((const uint8_t *)c->out.p, (const uint64_t *)d_tab, (uint32_t)which.size(), (uint32_t *)d_dig, c->stream));
  std::vector<uint32_t> dig(which.size() * 5);
  HIPCHK(hipMemcpyAsync(dig.data(), d_dig, dig_bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (size_t k = 0; k < which.size(); ++k) {
    uint8_t d[20];
    for (int i = 0; i < 5; ++i) {
      const uint32_t v = dig[5 * k + i];
      d[4 * i] = (uint8_t)(v >> 24); d[4 * i + 1] = (uint8_t)(v >> 16); d[4 * i + 2] = (uint8_t)(v >> 8); d[4 * i + 3] = (uint8_t)v;
    }
    bad[which[k]] = memcmp(d, so.segs[which[k]].sha1, 20) != 0;
  }
  return ZPAQHIP_OK;
}

int verify_sha1(zpaqhip_ctx *c, const ScanOut &so, const std::vector<zpaqhip_seg_result> &res, zpaqhip_err *err) {
  std::vector<int> bad;
  int rc = sha1_mismatches(c, so, res, bad, err);
  if (rc) return rc;
  for (size_t s = 0; s < bad.size(); ++s)
    if (bad[s]) { set_err(err, ZPAQHIP_E_SHA1, (int)so.segs[s].block, (int)s); return ZPAQHIP_E_SHA1; }
  return ZPAQHIP_OK;
}

}  // namespace

extern "C" {

int zpaqhip_decompress(zpaqhip_ctx *c, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len,
                       const zpaqhip_opts *opts_in, zpaqhip_err *err) {
  if (!c || (!in && in_len) || !out_len || (!out && out_cap)) { set_err(err, ZPAQHIP_E_ARG, -1, -1); return ZPAQHIP_E_ARG; }
  const zpaqhip_opts opts = resolve_opts(opts_in);
  ScanOut so;
  std::vector<zpaqhip_seg_result> res;
  uint64_t total = 0;
  *out_len = 0;
  int rc = decode_stream_to_device(c, in, in_len, opts, so, res, &total, err);
  if (rc) return rc;
  *out_len = (size_t)total;
  if (total > out_cap) { set_err(err, ZPAQHIP_E_OUTPUT_FULL, -1, -1); return ZPAQHIP_E_OUTPUT_FULL; }
  if (opts.verify_sha1) { rc = verify_sha1(c, so, res, err); if (rc) return rc; }      // on the device, before the copy back
  if (total) {
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    HIPCHK(hipMemcpyAsync(out, c->out.p, total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->stats.d2h_ms = ms;
  }
  return ZPAQHIP_OK;
}

int zpaqhip_decompress_segments(zpaqhip_ctx *c, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                size_t *out_len, zpaqhip_seg_result *results, size_t result_cap, size_t *n_results,
                                const zpaqhip_opts *opts_in, zpaqhip_err *err) {
  if (!c || (!in && in_len) || !out_len || !n_results || (!out && out_cap) || (!results && result_cap)) {
    set_err(err, ZPAQHIP_E_ARG, -1, -1);
    return ZPAQHIP_E_ARG;
  }
  const zpaqhip_opts opts = resolve_opts(opts_in);
  ScanOut so;
  std::vector<zpaqhip_seg_result> res;
  uint64_t total = 0;
  *out_len = 0; *n_results = 0;
  int rc = decode_stream_to_device(c, in, in_len, opts, so, res, &total, err, true);
  if (rc) return rc;
  *out_len = (size_t)total;
  *n_results = res.size();
  if (res.size() > result_cap) { set_err(err, ZPAQHIP_E_ARG, -1, -1, "result table too small"); return ZPAQHIP_E_ARG; }
  if (!res.empty()) memcpy(results, res.data(), res.size() * sizeof(zpaqhip_seg_result));
  if (total > out_cap) { set_err(err, ZPAQHIP_E_OUTPUT_FULL, -1, -1); return ZPAQHIP_E_OUTPUT_FULL; }
  if (total) HIPCHK(hipMemcpy(out, c->out.p, total, hipMemcpyDeviceToHost));
  if (opts.verify_sha1) {                   // mismatches become per-segment statuses
    std::vector<int> bad;
    rc = sha1_mismatches(c, so, res, bad, err);
    if (rc) return rc;
    for (size_t s = 0; s < res.size(); ++s)
      if (bad[s]) results[s].status = ZPAQHIP_E_SHA1;
  }
  return ZPAQHIP_OK;
}

// Decompresser.pcomp (Decompresser.cs:155-158 -> ZPAQL.write(out, true), ZPAQL.cs:158-179): the PCOMP program a block's
// first segment carries, as "length lo, length hi, program bytes"; *out_len = 0 when the block has none.  The program
// travels inside the coded data, so the start of the block is decoded (the generic kernel stops once the post-processor
// header is complete) and the program is read back from the arena slot.
int zpaqhip_block_pcomp(zpaqhip_ctx *c, const uint8_t *in, size_t in_len, uint32_t block, uint8_t *out, size_t out_cap,
                        size_t *out_len, zpaqhip_err *err) {
  if (!c || (!in && in_len) || !out_len || (!out && out_cap)) { set_err(err, ZPAQHIP_E_ARG, -1, -1); return ZPAQHIP_E_ARG; }
  *out_len = 0;
  ScanOut so;
  int rc = scan_stream(in, in_len, so, err);
  if (rc) return rc;
  if (block >= so.blocks.size()) { set_err(err, ZPAQHIP_E_ARG, (int)block, -1, "block id out of range"); return ZPAQHIP_E_ARG; }
  const zpaqhip_block &b = so.blocks[block];
  if (b.n_seg == 0) return ZPAQHIP_OK;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(c->in.reserve(in_len + 16));
  HIPCHK(hipMemcpy(c->in.p, in, in_len, hipMemcpyHostToDevice));
  HIPCHK(c->out.reserve(16));
  zpaqhip_opts o;
  memset(&o, 0, sizeof o);
  o.struct_size = sizeof o;
  o.max_concurrent = 1;                                  // the block runs in arena slot 0
  o.kernel = 1;                                          // the generic kernel knows how to stop after the header
  o.reserved[0] = kPpOnlyMagic;
  std::vector<zpaqhip_seg_result> res(so.segs.size());
  const uint32_t id = block;
  const uint64_t off0 = 0, cap0 = 0;                     // count-only: nothing is written
  rc = zpaqhip_decode_blocks_device(c, c->in.p, in, in_len, so.blocks.data(), so.blocks.size(), so.segs.data(), so.segs.size(),
                                    &id, 1, c->out.p, &off0, &cap0, res.data(), &o, c->stream, err);
  const uint32_t raw = c->raw_pp.size() > b.first_seg ? c->raw_pp[b.first_seg] : 0;
  const uint32_t state = raw & 255u, hsize = raw >> 8;
  if (state < 5) {                                       // PASS, or the stream broke before the program was complete
    const int st0 = res[b.first_seg].status;
    if (rc && st0 != ZPAQHIP_OK && st0 != ZPAQHIP_E_OUTPUT_FULL) return rc;
    return ZPAQHIP_OK;
  }
  *out_len = (size_t)hsize + 2;
  if (*out_len > out_cap) { set_err(err, ZPAQHIP_E_OUTPUT_FULL, (int)block, (int)b.first_seg); return ZPAQHIP_E_OUTPUT_FULL; }
  std::vector<uint8_t> code;
  ZhModel m;
  rc = build_model(in + b.hdr_off, b.hdr_len, m, code, err);
  if (rc) return rc;
  out[0] = (uint8_t)(hsize & 255);
  out[1] = (uint8_t)(hsize >> 8);
  HIPCHK(hipMemcpy(out + 2, (const uint8_t *)c->arena.p + m.pz_off + ZH_CODE_PAD, hsize, hipMemcpyDeviceToHost));
  return ZPAQHIP_OK;
}

int zpaqhip_decompress_cb(zpaqhip_ctx *c, zpaqhip_read_fn read_fn, zpaqhip_write_fn write_fn, void *user,
                          const zpaqhip_opts *opts_in, zpaqhip_err *err) {
  if (!c || !read_fn || !write_fn) { set_err(err, ZPAQHIP_E_ARG, -1, -1); return ZPAQHIP_E_ARG; }
  const zpaqhip_opts opts = resolve_opts(opts_in);
  std::vector<uint8_t> in;
  for (;;) {                                            // Reader.read until EOF (Reader.cs:14-25)
    const int chunk = 1 << 20;
    size_t old = in.size();
    in.resize(old + chunk);
    int n = read_fn(user, in.data() + old, chunk);
    if (n < 0) { set_err(err, ZPAQHIP_E_CALLBACK, -1, -1); return ZPAQHIP_E_CALLBACK; }
    in.resize(old + (size_t)n);
    if (n == 0) break;
  }
  ScanOut so;
  std::vector<zpaqhip_seg_result> res;
  uint64_t total = 0;
  int rc = decode_stream_to_device(c, in.data(), in.size(), opts, so, res, &total, err);
  if (rc) return rc;
  std::vector<uint8_t> out((size_t)total);
  if (total) HIPCHK(hipMemcpy(out.data(), c->out.p, total, hipMemcpyDeviceToHost));
  if (opts.verify_sha1) { rc = verify_sha1(c, so, res, err); if (rc) return rc; }
  for (size_t p = 0; p < out.size();) {                 // Writer.write (Writer.cs:19-24)
    int n = (int)std::min<size_t>(out.size() - p, 1 << 20);
    if (write_fn(user, out.data() + p, n) < 0) { set_err(err, ZPAQHIP_E_CALLBACK, -1, -1); return ZPAQHIP_E_CALLBACK; }
    p += (size_t)n;
  }
  return ZPAQHIP_OK;
}

}  // extern "C"
