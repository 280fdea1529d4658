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
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "zh_host.h"
#include "zh_zpaql_native.h"

extern "C" hipError_t zh_launch_generic(const ZhLaunch *L, uint32_t grid, hipStream_t stream);
extern "C" hipError_t zh_launch_cm(const ZhLaunch *L, uint32_t grid, hipStream_t stream);
extern "C" hipError_t zh_launch_cm_x2(const ZhLaunch *L, uint32_t grid, hipStream_t stream);   // two blocks per workgroup
extern "C" hipError_t zh_launch_sha1(const uint8_t *data, const uint64_t *seg, uint32_t n_seg, uint32_t *digest, hipStream_t stream);
extern "C" hipError_t zh_launch_chain(const ZhLaunch *L, uint32_t grid, hipStream_t stream, uint32_t spec, int prof, int pcall);
extern "C" hipError_t zh_launch_cm_prof(const ZhLaunch *L, uint32_t grid, hipStream_t stream);
extern "C" hipError_t zh_launch_chain2(const ZhLaunch *L, uint32_t grid, hipStream_t stream, uint32_t spec, int prof);
extern "C" int zh_chain2_has(uint32_t spec);
extern "C" hipError_t zh_launch_nibble(const ZhLaunch *L, uint32_t grid, hipStream_t stream, uint32_t spec, int prof);
extern "C" int zh_nibble_has(uint32_t spec);
static size_t pool_trim_device(int device, size_t keep);   // idle contexts of zpaqhip_decompress_multi's pool (below)
#ifdef ZH_WITH_CHAIN3   // make EXPERIMENTS=1: tools/experiments/zh_chain3.hip (three-wave form, measured slower; not in the product build)
extern "C" hipError_t zh_launch_chain3(const ZhLaunch *L, uint32_t grid, hipStream_t stream, uint32_t spec, int variant);
extern "C" int zh_chain3_has(uint32_t spec);
#endif
extern "C" hipError_t zh_launch_store(const ZhLaunch *L, uint32_t grid, hipStream_t stream);


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
  // whole-stream pipeline (created on first use): copy streams, double-buffered device input / staging / second-pass
  // buffers, pinned chunks for the Writer callback
  hipStream_t s_in = nullptr, s_out = nullptr;
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_pin[2] = {nullptr, nullptr}, ev_h0 = nullptr, ev_h1 = nullptr;
  hipEvent_t ev_d[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // D2H of a batch slot: first copy enqueued / last copy done
  bool d_pending[2] = {false, false};
  float d2h_acc = 0;
  DevBuf in2[2], out2[2], out_fix[2];
  uint8_t *pin[2] = {nullptr, nullptr};
  zpaqhip_stats stats{};
  uint32_t mem_share = 1;                 // contexts of this process that share the device (zpaqhip_decompress_multi): divides the memory budgets
  // a launch whose blocks need several kernel families (an archive that mixes models): one stream per family, forked from
  // and joined to the launch stream, each family with its own region of the arena
  hipStream_t fam_stream[ZH_NFAM_HOST] = {};
  hipEvent_t fam_ev[ZH_NFAM_HOST] = {};
  hipEvent_t fork_ev = nullptr;
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

constexpr size_t kQueueBytes = 512, kDebugBytes = 128;    // work-queue heads (32 B per family) and the *_prof kernels' sums
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
      (e = c->tables.reserve(sizeof(ZhTables) + sizeof(ZhTablesX))) != hipSuccess ||
      (e = hipMemcpy(c->tables.p, &host_tables(), sizeof(ZhTables), hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy((uint8_t *)c->tables.p + sizeof(ZhTables), &host_tables_x(), sizeof(ZhTablesX), hipMemcpyHostToDevice)) != hipSuccess) {
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
  for (DevBuf *b : {&c->tables, &c->arena, &c->models, &c->code, &c->bdesc, &c->sdesc, &c->results, &c->queue, &c->in, &c->out,
                    &c->in2[0], &c->in2[1], &c->out2[0], &c->out2[1], &c->out_fix[0], &c->out_fix[1]})
    b->release();
  for (int i = 0; i < 2; ++i) {
    if (c->pin[i]) (void)hipHostFree(c->pin[i]);
    if (c->ev_in[i]) (void)hipEventDestroy(c->ev_in[i]);
    if (c->ev_pin[i]) (void)hipEventDestroy(c->ev_pin[i]);
    for (int k = 0; k < 2; ++k) if (c->ev_d[i][k]) (void)hipEventDestroy(c->ev_d[i][k]);
  }
  if (c->ev_h0) (void)hipEventDestroy(c->ev_h0);
  if (c->ev_h1) (void)hipEventDestroy(c->ev_h1);
  if (c->s_in) (void)hipStreamDestroy(c->s_in);
  if (c->s_out) (void)hipStreamDestroy(c->s_out);
  for (uint32_t g = 0; g < ZH_NFAM_HOST; ++g) {
    if (c->fam_ev[g]) (void)hipEventDestroy(c->fam_ev[g]);
    if (c->fam_stream[g]) (void)hipStreamDestroy(c->fam_stream[g]);
  }
  if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
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
  zpaqhip_err e2{};
  const int rc = scan_stream(in, in_len, so, &e2);
  // On a framing error the blocks before the damage are still handed back (counts and tables), with the error code:
  // the reference's Decompresser delivers them too and only fails on reaching the damaged block.
  *n_blocks = so.blocks.size();
  *n_segs = so.segs.size();
  if (so.blocks.size() > block_cap || so.segs.size() > seg_cap) {
    if (block_cap || seg_cap) { set_err(err, ZPAQHIP_E_ARG, -1, -1, "block/segment table too small"); return ZPAQHIP_E_ARG; }
    if (rc && err) *err = e2;
    return rc;
  }
  if (blocks && !so.blocks.empty()) memcpy(blocks, so.blocks.data(), so.blocks.size() * sizeof(zpaqhip_block));
  if (segs && !so.segs.empty()) memcpy(segs, so.segs.data(), so.segs.size() * sizeof(zpaqhip_segment));
  if (rc && err) *err = e2;
  return rc;
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
  // sources of the launch's asynchronous H2D copies: they must outlive decode_launch (the API does not promise that a
  // pageable hipMemcpyAsync has read its source when it returns)
  std::vector<ZhModel> models;
  std::vector<uint8_t> code;
  std::vector<ZhSegDesc> sd;
  std::vector<ZhBlockDesc> bd_sorted;
  // zh_store.hip hands blocks it does not take (ZH_E_RETRY) back: they run on the generic kernel in decode_finish
  bool has_store = false;
  ZhLaunch store_launch{};
  size_t store_base = 0, store_count = 0;
  uint32_t store_slots = 0;
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
  std::vector<ZhModel> &models = P.models;
  std::vector<uint8_t> &code = P.code;
  std::vector<ZhBlockDesc> bd(sel.size());
  std::vector<ZhSegDesc> &sd = P.sd;
  sd.assign(n_segs, ZhSegDesc{});
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
    if (f == ZH_FAM_STORE && (opts.reserved[0] == kPpOnlyMagic)) f = ZH_FAM_GENERIC;
    if (opts.kernel == 3 && f == ZH_FAM_CM1) f = ZH_FAM_CHAIN;   // force the lane-per-component kernel
    if (opts.kernel == 4 && f > ZH_FAM_CHAIN) f = ZH_FAM_CHAIN;  // lane-per-component kernel without model specialisation
    // the method models of min's / mid's shape (zh_framing.cpp) are known to zh_nibble.hip only: the older kernels of these
    // families carry the built-in HCOMP programs
    if ((opts.kernel == 9 || opts.kernel == 7 || opts.kernel == 8 || opts.kernel == 5) && f > ZH_FAM_CHAIN && f < ZH_FAM_STORE) {
      const uint32_t hk = (models[bd[k].model].kind >> 8) & 255u;
      if (hk == ZH_NATIVE_HCOMP_M4 || hk == ZH_NATIVE_HCOMP_M3 || hk == ZH_NATIVE_HCOMP_M2 || hk == ZH_NATIVE_HCOMP_M2E) f = ZH_FAM_CHAIN;
    }
    if ((opts.kernel == 9 || opts.kernel == 7 || opts.kernel == 8 || opts.kernel == 5 || opts.kernel == 4) && (f == ZH_FAM_CHAIN_MID8 || f == ZH_FAM_CHAIN_MIN1)) f = ZH_FAM_CHAIN;
    return f;
  };
  std::vector<std::vector<uint32_t>> groups(ZH_NFAM_HOST);
  for (size_t k = 0; k < sel.size(); ++k) groups[family(k)].push_back((uint32_t)k);

  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  free_b += c->arena.cap;                               // our own cached arena is reusable
  uint64_t mem_budget = (free_b > (1ull << 30) ? free_b - (1ull << 30) : free_b / 2) / std::max(1u, c->mem_share);
  {
    // Idle contexts zpaqhip_decompress_multi has pooled on this device keep their arenas (tens of GB for the larger models).
    // If this launch could not give every block (up to 256) its slot, they are released first and the budget taken again
    // (multi([0,0,0,0]) followed by a max-model decode on a context of one's own saw a quarter of the memory).
    uint64_t wish = 0;
    for (uint32_t g = 0; g < ZH_NFAM_HOST; ++g) {
      uint64_t stride = 0;
      for (uint32_t k : groups[g]) stride = std::max<uint64_t>(stride, models[bd[k].model].arena_bytes);
      wish += stride * std::min<uint64_t>(groups[g].size(), opts.max_concurrent ? opts.max_concurrent : 256u);
    }
    if (wish > mem_budget && pool_trim_device(c->device, 0)) {
      HIPCHK(hipMemGetInfo(&free_b, &total_b));
      free_b += c->arena.cap;
      mem_budget = (free_b > (1ull << 30) ? free_b - (1ull << 30) : free_b / 2) / std::max(1u, c->mem_share);
    }
  }

  HIPCHK(c->models.reserve(models.size() * sizeof(ZhModel)));
  HIPCHK(c->code.reserve(code.size() + 16));
  HIPCHK(c->bdesc.reserve(sel.size() * sizeof(ZhBlockDesc)));
  HIPCHK(c->sdesc.reserve(sd.size() * sizeof(ZhSegDesc)));
  HIPCHK(c->results.reserve(n_segs * sizeof(ZhSegResult)));
  static_assert(32 * ZH_NFAM_HOST <= kQueueBytes, "one 32-byte work-queue head per kernel family");
  HIPCHK(c->queue.reserve(kQueueBytes + kDebugBytes));   // [heads | diagnostic cycle sums]: the two never overlap
  HIPCHK(hipMemcpyAsync(c->models.p, models.data(), models.size() * sizeof(ZhModel), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(c->code.p, code.data(), code.size(), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(c->sdesc.p, sd.data(), sd.size() * sizeof(ZhSegDesc), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemsetAsync(c->results.p, 0xff, n_segs * sizeof(ZhSegResult), stream));
  HIPCHK(hipMemsetAsync(c->queue.p, 0, kQueueBytes + kDebugBytes, stream));

  // arena: one region per kernel family when all of them fit at once (the families then run side by side: an archive
  // that mixes models fills the GPU with whatever blocks it has), else one region sized for the most demanding family,
  // used by one family after the other
  uint32_t slots_of[ZH_NFAM_HOST] = {};
  bool cm_x2[ZH_NFAM_HOST] = {};
  uint64_t stride_of[ZH_NFAM_HOST], arena_need = 0, arena_sum = 0, arena_off[ZH_NFAM_HOST] = {};
  uint32_t n_fam = 0;
  for (auto &x : stride_of) x = 256;
  for (uint32_t g = 0; g < ZH_NFAM_HOST; ++g) {
    if (groups[g].empty()) continue;
    for (uint32_t k : groups[g]) stride_of[g] = std::max<uint64_t>(stride_of[g], models[bd[k].model].arena_bytes);
    uint64_t max_slots = mem_budget / stride_of[g];
    if (max_slots == 0) { set_err(err, ZPAQHIP_E_DEVICE_MEM, -1, -1, "Out of memory"); return ZPAQHIP_E_DEVICE_MEM; }
    uint32_t want = opts.max_concurrent ? opts.max_concurrent : 256u;   // one block per CU by default
    // single-CM blocks beyond one per CU: two per workgroup (zh_decode_cm_x2: 32 LDS windows each instead of 64), so a stream
    // of many blocks uses all four SIMDs of a CU; opts.kernel == 6 forces that form, 2 the one-block form (A/B runs, tests)
    cm_x2[g] = g == ZH_FAM_CM1 && !getenv("ZPAQHIP_PROF") && opts.kernel != 2 &&
               (opts.kernel == 6 || (!opts.max_concurrent && groups[g].size() > 256));
    if (cm_x2[g]) want = opts.max_concurrent ? opts.max_concurrent : 512u;
    slots_of[g] = (uint32_t)std::min<uint64_t>({(uint64_t)want, max_slots, (uint64_t)groups[g].size()});
    if (cm_x2[g]) { slots_of[g] = (slots_of[g] + 1u) & ~1u; if (slots_of[g] > max_slots) { slots_of[g] = (uint32_t)(max_slots & ~1ull); cm_x2[g] = slots_of[g] >= 2; if (!cm_x2[g]) slots_of[g] = 1; } }
    arena_need = std::max<uint64_t>(arena_need, slots_of[g] * stride_of[g]);
    arena_off[g] = arena_sum;
    arena_sum += (slots_of[g] * stride_of[g] + 255) & ~255ull;
    ++n_fam;
  }
  const bool side_by_side = n_fam > 1 && arena_sum <= mem_budget && !getenv("ZPAQHIP_SERIAL_FAMILIES");
  if (!side_by_side) for (auto &x : arena_off) x = 0;
  HIPCHK(c->arena.reserve((size_t)(side_by_side ? arena_sum : arena_need)));
  if (side_by_side && !c->fork_ev) HIPCHK(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));

  std::vector<ZhBlockDesc> &bd_sorted = P.bd_sorted;
  bd_sorted.reserve(sel.size());
  size_t base_of[ZH_NFAM_HOST] = {};
  for (uint32_t g = 0; g < ZH_NFAM_HOST; ++g) {
    // Longest block first: the work queue then balances the tail (LPT order).
    std::stable_sort(groups[g].begin(), groups[g].end(), [&](uint32_t a, uint32_t b) { return weight[a] > weight[b]; });
    base_of[g] = bd_sorted.size();
    for (uint32_t k : groups[g]) bd_sorted.push_back(bd[k]);
  }
  HIPCHK(hipMemcpyAsync(c->bdesc.p, bd_sorted.data(), bd_sorted.size() * sizeof(ZhBlockDesc), hipMemcpyHostToDevice, stream));

  uint32_t launches = 0, slots = 0, kind_used = 0;
  HIPCHK(hipEventRecord(c->ev0, stream));
  if (side_by_side) HIPCHK(hipEventRecord(c->fork_ev, stream));      // the tables and descriptors above are on `stream`
  hipStream_t const launch_stream = stream;
  for (uint32_t g = 0; g < ZH_NFAM_HOST; ++g) {
    if (groups[g].empty()) continue;
    hipStream_t stream = launch_stream;                  // (shadows: the family's own stream when families run side by side)
    if (side_by_side) {
      if (!c->fam_stream[g]) {
        HIPCHK(hipStreamCreateWithFlags(&c->fam_stream[g], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->fam_ev[g], hipEventDisableTiming));
      }
      stream = c->fam_stream[g];
      HIPCHK(hipStreamWaitEvent(stream, c->fork_ev, 0));
    }
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
    L.arena = (uint8_t *)c->arena.p + arena_off[g];
    L.arena_stride = stride_of[g];
    L.tables = (const ZhTables *)c->tables.p;
    L.queue = (uint32_t *)c->queue.p + 8 * g;           // one work-queue head per launch
    L.n_blocks = (uint32_t)groups[g].size();
    L.budget = opts.zpaql_budget;
    L.flags = opts.reserved[0] == kPpOnlyMagic ? ZH_LAUNCH_PP_ONLY : 0u;
    const bool prof = getenv("ZPAQHIP_PROF") != nullptr;   // diagnostic build with in-kernel stamps
    if (prof) L.debug = (uint64_t *)((uint8_t *)c->queue.p + kQueueBytes);
    if (g == ZH_FAM_STORE) {
      HIPCHK(zh_launch_store(&L, slots_of[g], stream));
      P.has_store = true; P.store_launch = L; P.store_base = base_of[g]; P.store_count = groups[g].size(); P.store_slots = slots_of[g];
    }
    else if (g == ZH_FAM_CM1 && prof) HIPCHK(zh_launch_cm_prof(&L, slots_of[g], stream));
    else if (g == ZH_FAM_CM1 && cm_x2[g]) HIPCHK(zh_launch_cm_x2(&L, slots_of[g] / 2, stream));
    else if (g == ZH_FAM_CM1) HIPCHK(zh_launch_cm(&L, slots_of[g], stream));
    else if (g == ZH_FAM_CHAIN_MID8) HIPCHK(zh_launch_nibble(&L, slots_of[g], stream, 5, prof));   // mid's shape, eight mixer inputs
    else if (g == ZH_FAM_CHAIN_MIN1) HIPCHK(zh_launch_nibble(&L, slots_of[g], stream, 6, prof));   // one ICM on min's loop
#ifdef ZH_WITH_CHAIN3
    else if (g > ZH_FAM_CHAIN && zh_chain3_has(g - ZH_FAM_CHAIN) && (opts.kernel == 7 || opts.kernel == 8))   // decoder ‖ model ‖ helper wave: experiment build only
      HIPCHK(zh_launch_chain3(&L, slots_of[g], stream, g - ZH_FAM_CHAIN, prof ? 2 : opts.kernel == 7));
#endif
    else if (g > ZH_FAM_CHAIN && zh_nibble_has(g - ZH_FAM_CHAIN) && opts.kernel != 9 && opts.kernel != 5)   // min / mid a nibble at a time (zh_nibble.hip); 9: the bit-at-a-time form below
      HIPCHK(zh_launch_nibble(&L, slots_of[g], stream, g - ZH_FAM_CHAIN, prof));
    else if (g > ZH_FAM_CHAIN && zh_chain2_has(g - ZH_FAM_CHAIN) && opts.kernel != 5)   // per-model bit loop (zh_chain2.hip)
      HIPCHK(zh_launch_chain2(&L, slots_of[g], stream, g - ZH_FAM_CHAIN, prof));
    else if (g >= ZH_FAM_CHAIN) {
      int pcall = 0;                                     // any model with PCOMP memory: the variant with translated post-processors
      for (uint32_t k : groups[g]) pcall |= (models[bd[k].model].ph | models[bd[k].model].pm) != 0;
      HIPCHK(zh_launch_chain(&L, slots_of[g], stream, g - ZH_FAM_CHAIN, prof, pcall));
    }
    else HIPCHK(zh_launch_generic(&L, slots_of[g], stream));
    if (side_by_side) {                                  // join: the launch stream goes on when every family is done
      HIPCHK(hipEventRecord(c->fam_ev[g], stream));
      HIPCHK(hipStreamWaitEvent(launch_stream, c->fam_ev[g], 0));
    }
    ++launches;
    slots = std::max(slots, slots_of[g]);
    kind_used = std::max(kind_used, g == ZH_FAM_STORE ? 1u : std::min(g, (uint32_t)ZH_FAM_CHAIN) + 1);
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
    uint64_t dbg[16];
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipMemcpy(dbg, (uint8_t *)c->queue.p + kQueueBytes, kDebugBytes, hipMemcpyDeviceToHost));
    fprintf(stderr, "ZPAQHIP_PROF cycles:");
    for (int i = 0; i < 16; ++i) fprintf(stderr, " %llu", (unsigned long long)dbg[i]);
    fprintf(stderr, "\n");
  }
  std::vector<ZhSegResult> res(n_segs);
  c->raw_pp.assign(n_segs, 0);
  HIPCHK(hipMemcpyAsync(res.data(), c->results.p, n_segs * sizeof(ZhSegResult), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  uint32_t extra_launches = 0;
  if (P.has_store) {
    // blocks the store kernel handed back (programs other than the reference's LZ77 ones, damaged chunks): the generic
    // kernel is the complete implementation.  Same arena slots, same output places.
    std::vector<ZhBlockDesc> redo;
    for (size_t k = 0; k < P.store_count; ++k) {
      const ZhBlockDesc &b = P.bd_sorted[P.store_base + k];
      if (b.n_seg && res[b.first_seg].status == ZH_E_RETRY) redo.push_back(b);
    }
    if (!redo.empty()) {
      ZhLaunch L = P.store_launch;
      HIPCHK(hipMemcpyAsync((void *)L.blocks, redo.data(), redo.size() * sizeof(ZhBlockDesc), hipMemcpyHostToDevice, stream));
      HIPCHK(hipMemsetAsync(L.queue, 0, 32, stream));
      L.n_blocks = (uint32_t)redo.size();
      HIPCHK(hipEventRecord(c->ev0, stream));
      HIPCHK(zh_launch_generic(&L, std::min<uint32_t>(P.store_slots, L.n_blocks), stream));
      HIPCHK(hipEventRecord(c->ev1, stream));
      HIPCHK(hipMemcpyAsync(res.data(), c->results.p, n_segs * sizeof(ZhSegResult), hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
      float ms2 = 0;
      HIPCHK(hipEventElapsedTime(&ms2, c->ev0, c->ev1));
      ms += ms2;
      ++extra_launches;
    }
  }

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
  c->stats.launches = launches + extra_launches;
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
// Whole-stream forms: a three-stage pipeline over batches of whole blocks
//
//   source (caller's buffer, or Reader-shaped callback read incrementally)
//     -> scan a batch (>= 512 blocks when the stream has them)            host
//     -> H2D of the batch's bytes                                         stream s_in
//     -> decode kernels (zh_* families)                                   stream c->stream
//     -> D2H of the plaintext, block by block, into stream order          stream s_out
//   sink (caller's buffer, or Writer-shaped callback fed from pinned chunks)
//
// While the kernels of batch k run, the host drains the output of batch k-1 and reads / scans / uploads batch k+1
// (double-buffered device input and staging buffers).  Blocks whose plaintext size is unknown (no decimal size in the
// segment comment, LICENSE:57-58) get a provisional staging slot and are decoded ONCE; only a block that overflows its
// slot is decoded a second time, alone, with the exact size it reported.
// ---------------------------------------------------------------------------
namespace {

bool data_error(int rc) {                 // per-segment outcomes, as opposed to failures of the call itself
  return rc < 0 && rc != ZPAQHIP_E_HIP && rc != ZPAQHIP_E_ARG && rc != ZPAQHIP_E_DEVICE_MEM && rc != ZPAQHIP_E_NO_DEVICE &&
         rc != ZPAQHIP_E_HEADER && rc != ZPAQHIP_E_COMPONENT && rc != ZPAQHIP_E_HM_TOO_BIG && rc != ZPAQHIP_E_CALLBACK;
}

struct Source {
  const uint8_t *mem = nullptr;           // memory form
  size_t mem_len = 0, mem_pos = 0;
  zpaqhip_read_fn rd = nullptr;           // callback form (Reader.read, Reader.cs:14-25)
  void *user = nullptr;
  std::vector<uint8_t> buf;               // bytes read and not yet handed to a batch
  bool eof = false;
  uint64_t consumed = 0;                  // stream offset of the next unconsumed byte
};

struct Batch {
  std::vector<uint8_t> own;               // callback form: the batch's bytes (moved out of Source::buf)
  const uint8_t *h = nullptr;             // host bytes [0, len): whole blocks; tables in `so` are relative to h
  size_t len = 0;
  uint64_t stream_off = 0;
  ScanOut so;
  size_t blk0 = 0, seg0 = 0;              // global index of the first block / segment
  std::vector<uint64_t> off, cap, real;   // staging offset / capacity / plaintext length per block
  std::vector<zpaqhip_seg_result> res;    // per segment; out_off relative to the staging buffer
  int slot = 0;
  int rc_after = 0;                       // framing error that follows the last block of this batch ...
  zpaqhip_err err_after{};                // ... reported once everything before it has been delivered
  bool last = false;
};

struct Batch;
int sha1_mismatches(zpaqhip_ctx *c, const Batch &bt, std::vector<int> &bad, zpaqhip_err *err);

struct Sinkk {
  uint8_t *mem = nullptr; size_t cap = 0; // memory form
  zpaqhip_write_fn wr = nullptr;          // callback form (Writer.write, Writer.cs:19-24)
  void *user = nullptr;
  uint64_t total = 0;                     // plaintext bytes so far (counted past `cap` too)
  // multi-device form: block k of this source goes to mem + place[k] (sizes promised by the caller); sizes[k] = what
  // it really produced
  const uint64_t *place = nullptr;
  std::vector<uint64_t> *sizes = nullptr;
  // ... and at most place_cap[k] bytes of it; a block that is larger (a wrong or missing size in its comment) is kept
  // whole in spill[k] and put in place by the caller once every size is known
  const uint64_t *place_cap = nullptr;
  std::vector<std::vector<uint8_t>> *spill = nullptr;
};

constexpr size_t kBatchMinBlocks = 512, kBatchMinBlocksOther = 256, kBatchMinBytes = 32u << 20, kBatchMaxBlocks = 4096;   // 512: single-CM blocks run two per CU (zh_decode_cm_x2); every other kernel holds 256 blocks per launch, and a second batch overlaps its copies with this one's kernels
constexpr size_t kReadChunk = 4u << 20, kPinChunk = 32u << 20;

// A header the framing scan accepts can still be refused when the model is built (component limits, table sizes:
// Predictor.cs:94-167).  The reference raises that at the block's Predictor.init, after everything before the block has
// been written; so the batch is cut in front of the first such block and the error is reported behind the cut.
void cut_at_bad_model(Batch &bt) {
  std::map<std::string, int> seen;
  for (size_t b = 0; b < bt.so.blocks.size(); ++b) {
    const zpaqhip_block &B = bt.so.blocks[b];
    std::string key((const char *)bt.h + B.hdr_off, B.hdr_len);
    auto it = seen.find(key);
    zpaqhip_err e2{};
    if (it == seen.end()) {
      ZhModel m;
      std::vector<uint8_t> code;
      it = seen.emplace(key, build_model(bt.h + B.hdr_off, B.hdr_len, m, code, &e2)).first;
    } else if (it->second) {
      ZhModel m;
      std::vector<uint8_t> code;
      (void)build_model(bt.h + B.hdr_off, B.hdr_len, m, code, &e2);     // (again, for the message)
    }
    if (!it->second) continue;
    e2.block = (int32_t)b; e2.segment = -1;
    bt.rc_after = it->second; bt.err_after = e2; bt.last = true;
    bt.so.blocks.resize(b);
    bt.so.segs.resize(b ? bt.so.blocks.back().first_seg + bt.so.blocks.back().n_seg : 0);
    bt.so.stopped = false;
    bt.len = b ? (size_t)bt.so.blocks.back().end_off : 0;
    return;
  }
}

// Next batch of whole blocks, or batch.so.blocks.empty() at the end of the stream.  Returns a call-level error only.
int next_batch(Source &src, Batch &bt, size_t blk0, size_t seg0, size_t batch_blocks, zpaqhip_err *err) {
  ScanLimit lim;
  lim.min_blocks = kBatchMinBlocks; lim.min_blocks_other = kBatchMinBlocksOther; lim.min_bytes = kBatchMinBytes; lim.max_blocks = kBatchMaxBlocks;
  if (batch_blocks) { lim.min_blocks = lim.max_blocks = batch_blocks; lim.min_blocks_other = 0; lim.min_bytes = 0; }   // zpaqhip_opts.batch_blocks
  bt = Batch();
  bt.blk0 = blk0; bt.seg0 = seg0;
  zpaqhip_err e2{};
  if (src.mem || !src.rd) {
    const uint8_t *w = src.mem + src.mem_pos;
    const size_t wl = src.mem_len - src.mem_pos;
    int rc = scan_stream(w, wl, bt.so, &e2, lim);
    bt.h = w; bt.stream_off = src.mem_pos;
    bt.len = bt.so.blocks.empty() ? 0 : (size_t)bt.so.blocks.back().end_off;
    if (rc) { bt.rc_after = rc; bt.err_after = e2; bt.last = true; src.mem_pos = src.mem_len; }
    else if (bt.so.stopped) src.mem_pos += bt.so.resume_off;
    else { bt.last = true; src.mem_pos = src.mem_len; }
    cut_at_bad_model(bt);
    if (bt.last) src.mem_pos = src.mem_len;
    return ZPAQHIP_OK;
  }
  for (;;) {                                            // Reader form: read until a batch is complete or the input ends
    int rc = scan_stream(src.buf.data(), src.buf.size(), bt.so, &e2, lim);
    const bool complete = bt.so.stopped || src.eof;
    if (rc && !(bt.so.hit_eof && !src.eof)) {           // damage that more input cannot repair
      bt.rc_after = rc; bt.err_after = e2; bt.last = true;
    } else if (!complete) {
      const size_t old = src.buf.size();
      src.buf.resize(old + kReadChunk);
      int n = src.rd(src.user, src.buf.data() + old, (int)kReadChunk);
      if (n < 0) { set_err(err, ZPAQHIP_E_CALLBACK, -1, -1); return ZPAQHIP_E_CALLBACK; }
      src.buf.resize(old + (size_t)n);
      if (n == 0) src.eof = true;
      continue;
    } else if (rc) {                                    // input ended inside a block
      bt.rc_after = rc; bt.err_after = e2; bt.last = true;
    }
    if (!bt.so.stopped) bt.last = true;
    bt.len = bt.so.blocks.empty() ? 0 : (size_t)bt.so.blocks.back().end_off;
    bt.stream_off = src.consumed;
    // hand the batch its bytes; keep the unconsumed tail for the next one
    const size_t keep_from = bt.last ? src.buf.size() : bt.so.resume_off;
    bt.own.assign(src.buf.begin(), src.buf.begin() + (ptrdiff_t)bt.len);
    src.buf.erase(src.buf.begin(), src.buf.begin() + (ptrdiff_t)keep_from);
    src.consumed += keep_from;
    bt.h = bt.own.data();
    cut_at_bad_model(bt);
    return ZPAQHIP_OK;
  }
}

void rebase_err(zpaqhip_err *err, const Batch &bt) {
  if (!err) return;
  if (err->block >= 0) err->block += (int)bt.blk0;
  if (err->segment >= 0) err->segment += (int)bt.seg0;
}

// Staging layout of a batch: the size hint of a block when the stream carries one, else a provisional slot.
int plan_staging(zpaqhip_ctx *c, Batch &bt, zpaqhip_err *err) {
  const size_t nb = bt.so.blocks.size();
  bt.off.assign(nb, 0); bt.cap.assign(nb, 0); bt.real.assign(nb, 0);
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  free_b += c->out2[0].cap + c->out2[1].cap;
  const uint64_t budget = free_b / 4 / std::max(1u, c->mem_share);   // per staging buffer; the arena needs the rest
  uint64_t sum = 0;
  for (size_t b = 0; b < nb; ++b) {
    const zpaqhip_block &B = bt.so.blocks[b];
    uint64_t coded = 0;
    for (uint32_t i = 0; i < B.n_seg; ++i) coded += bt.so.segs[B.first_seg + i].data_len;
    uint64_t cap = B.usize_hint;
    // The hint is untrusted text (a comment that merely starts with digits, or a hostile archive): it only places the
    // block when it is plausible; anything else gets the provisional slot and, if that overflows, an exact second pass.
    if (cap == UINT64_MAX || cap > (1ull << 40) || cap > budget)
      cap = std::min<uint64_t>(std::max<uint64_t>(16 * coded, 64u << 10), 256ull << 20);
    bt.cap[b] = cap;
    sum += cap;
  }
  if (sum > budget) {                                   // does not fit: count-only for the largest slots until it does
    std::vector<size_t> order(nb);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b2) { return bt.cap[a] > bt.cap[b2]; });
    for (size_t k = 0; k < nb && sum > budget; ++k) { sum -= bt.cap[order[k]]; bt.cap[order[k]] = 0; }
  }
  uint64_t pos = 0;
  for (size_t b = 0; b < nb; ++b) { bt.off[b] = pos; pos += bt.cap[b]; }
  if (c->out2[bt.slot].reserve((size_t)pos + 16) != hipSuccess) {   // not even that: everything count-only, exact second pass
    (void)hipGetLastError();
    std::fill(bt.cap.begin(), bt.cap.end(), 0ull);
    std::fill(bt.off.begin(), bt.off.end(), 0ull);
    HIPCHK(c->out2[bt.slot].reserve(16));
  }
  return ZPAQHIP_OK;
}

// SHA-1 of every decoded segment of a batch that stores one, computed ON THE DEVICE over the staging buffers
// (zh_sha1_dev.hip) while the plaintext is still there; bad[s] = 1 where the digest differs from the stored one
// (Decompresser.cs:183-191).
int sha1_mismatches(zpaqhip_ctx *c, const Batch &bt, std::vector<int> &bad, zpaqhip_err *err) {
  const size_t ns = bt.so.segs.size();
  bad.assign(ns, 0);
  // two passes: segments in the staging buffer, segments in the second-pass buffer
  for (int where = 0; where < 2; ++where) {
    std::vector<uint64_t> tab;
    std::vector<uint32_t> which;
    for (size_t b = 0; b < bt.so.blocks.size(); ++b) {
      if ((int)(bt.off[b] >> 63) != where) continue;
      const zpaqhip_block &B = bt.so.blocks[b];
      for (uint32_t i = 0; i < B.n_seg; ++i) {
        const size_t s = B.first_seg + i;
        if (!(bt.so.segs[s].flags & 1) || bt.res[s].status != ZPAQHIP_OK) continue;
        tab.push_back(bt.res[s].out_off);
        tab.push_back(bt.res[s].out_len);
        which.push_back((uint32_t)s);
      }
    }
    if (which.empty()) continue;
    const uint8_t *base = where ? (const uint8_t *)c->out_fix[bt.slot].p : (const uint8_t *)c->out2[bt.slot].p;
    const size_t tab_bytes = tab.size() * 8, dig_bytes = which.size() * 20;
    HIPCHK(c->sdesc.reserve(tab_bytes + dig_bytes + 64));        // the descriptors of the finished decode are not needed any more
    uint8_t *d_tab = (uint8_t *)c->sdesc.p, *d_dig = d_tab + ((tab_bytes + 15) & ~(size_t)15);
    HIPCHK(hipMemcpyAsync(d_tab, tab.data(), tab_bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(zh_launch_sha1(base, (const uint64_t *)d_tab, (uint32_t)which.size(), (uint32_t *)d_dig, c->stream));
    std::vector<uint32_t> dig(which.size() * 5);
    HIPCHK(hipMemcpyAsync(dig.data(), d_dig, dig_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < which.size(); ++k) {
      uint8_t d[20];
      for (int i = 0; i < 5; ++i) {
        const uint32_t v = dig[5 * k + i];
        d[4 * i] = (uint8_t)(v >> 24); d[4 * i + 1] = (uint8_t)(v >> 16); d[4 * i + 2] = (uint8_t)(v >> 8); d[4 * i + 3] = (uint8_t)v;
      }
      bad[which[k]] = memcmp(d, bt.so.segs[which[k]].sha1, 20) != 0;
    }
  }
  return ZPAQHIP_OK;
}

// After decode_finish: plaintext sizes, and a second decode of the blocks that did not fit their staging slot.
int settle_batch(zpaqhip_ctx *c, Batch &bt, const zpaqhip_opts &opts, bool tolerate, zpaqhip_stats &acc, zpaqhip_err *err) {
  const size_t nb = bt.so.blocks.size();
  std::vector<uint32_t> redo;
  for (size_t b = 0; b < nb; ++b) {
    const zpaqhip_block &B = bt.so.blocks[b];
    uint64_t real = 0;
    bool full = false;
    for (uint32_t i = 0; i < B.n_seg; ++i) {
      real += bt.res[B.first_seg + i].out_len;
      full |= bt.res[B.first_seg + i].status == ZPAQHIP_E_OUTPUT_FULL;
    }
    bt.real[b] = real;
    if (full || real > bt.cap[b]) redo.push_back((uint32_t)b);
  }
  if (redo.empty()) return ZPAQHIP_OK;
  // exact sizes are known now: these blocks go to a buffer of their own
  std::vector<uint64_t> off2(redo.size()), cap2(redo.size());
  uint64_t pos = 0;
  for (size_t k = 0; k < redo.size(); ++k) { off2[k] = pos; cap2[k] = bt.real[redo[k]]; pos += cap2[k]; }
  HIPCHK(c->out_fix[bt.slot].reserve((size_t)pos + 16));
  zh_pending P;
  int rc = decode_launch(c, c->in2[bt.slot].p, bt.h, bt.len, bt.so.blocks.data(), nb, bt.so.segs.data(), bt.so.segs.size(),
                         redo.data(), redo.size(), c->out_fix[bt.slot].p, off2.data(), cap2.data(), opts, c->stream, P, err);
  if (!rc) rc = decode_finish(c, P, bt.res.data(), err);
  acc.kernel_ms += c->stats.kernel_ms; acc.launches += c->stats.launches;
  if (rc && !(tolerate && data_error(rc)) && !data_error(rc)) return rc;
  for (size_t k = 0; k < redo.size(); ++k) {            // mark: lives in out_fix (offset | top bit)
    bt.off[redo[k]] = off2[k] | (1ull << 63);
    bt.cap[redo[k]] = cap2[k];
  }
  return ZPAQHIP_OK;
}

const uint8_t *block_dev_ptr(zpaqhip_ctx *c, const Batch &bt, size_t b) {
  const uint64_t o = bt.off[b];
  return (o >> 63) ? (const uint8_t *)c->out_fix[bt.slot].p + (o & ~(1ull << 63)) : (const uint8_t *)c->out2[bt.slot].p + o;
}

// Deliver the first `upto_blocks` blocks (plus `partial` bytes of the next one) of a decoded batch to the sink, in order.
// D2H time of a batch slot's last drain (memory and placed forms), added to c->d2h_acc once the copies are done
int harvest_d2h(zpaqhip_ctx *c, int slot, zpaqhip_err *err) {
  if (!c->d_pending[slot]) return ZPAQHIP_OK;
  c->d_pending[slot] = false;
  HIPCHK(hipEventSynchronize(c->ev_d[slot][1]));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, c->ev_d[slot][0], c->ev_d[slot][1]));
  c->d2h_acc += ms;
  return ZPAQHIP_OK;
}

int drain_batch(zpaqhip_ctx *c, const Batch &bt, size_t upto_blocks, uint64_t partial, Sinkk &sink, zpaqhip_err *err) {
  struct Piece { const uint8_t *p; uint64_t n; };
  const bool timed = sink.place || sink.mem || !sink.wr;
  if (timed) {
    int hrc = harvest_d2h(c, bt.slot, err);
    if (hrc) return hrc;
    HIPCHK(hipEventRecord(c->ev_d[bt.slot][0], c->s_out));
  }
  std::vector<Piece> pieces;
  for (size_t b = 0; b <= upto_blocks && b < bt.so.blocks.size(); ++b) {
    const uint64_t n = b < upto_blocks ? bt.real[b] : std::min<uint64_t>(partial, bt.real[b]);
    if (!n) continue;
    const uint8_t *p = block_dev_ptr(c, bt, b);
    if (!pieces.empty() && pieces.back().p + pieces.back().n == p) pieces.back().n += n;     // contiguous on the device too
    else pieces.push_back({p, n});
  }
  if (sink.place) {                                     // placed form: every block at its own offset
    for (size_t b = 0; b < upto_blocks && b < bt.so.blocks.size(); ++b) {
      const uint64_t n = bt.real[b], at = sink.place[bt.blk0 + b];
      if (sink.sizes) (*sink.sizes)[bt.blk0 + b] = n;
      if (sink.place_cap && n > sink.place_cap[bt.blk0 + b]) {
        if (sink.spill) {
          std::vector<uint8_t> &sp = (*sink.spill)[bt.blk0 + b];
          sp.resize((size_t)n);
          HIPCHK(hipMemcpyAsync(sp.data(), block_dev_ptr(c, bt, b), (size_t)n, hipMemcpyDeviceToHost, c->s_out));
        }
      } else if (n && at < sink.cap)
        HIPCHK(hipMemcpyAsync(sink.mem + at, block_dev_ptr(c, bt, b), (size_t)std::min<uint64_t>(n, sink.cap - at), hipMemcpyDeviceToHost, c->s_out));
      sink.total += n;
    }
    HIPCHK(hipEventRecord(c->ev_d[bt.slot][1], c->s_out));
    c->d_pending[bt.slot] = true;
    return ZPAQHIP_OK;
  }
  if (sink.mem || !sink.wr) {
    for (const Piece &pc : pieces) {
      if (sink.total < sink.cap) {
        const uint64_t n = std::min<uint64_t>(pc.n, sink.cap - sink.total);
        HIPCHK(hipMemcpyAsync(sink.mem + sink.total, pc.p, (size_t)n, hipMemcpyDeviceToHost, c->s_out));
      }
      sink.total += pc.n;
    }
    HIPCHK(hipEventRecord(c->ev_d[bt.slot][1], c->s_out));
    c->d_pending[bt.slot] = true;
    return ZPAQHIP_OK;
  }
  // Writer form: pinned double buffer; the copy of chunk i+1 runs while write_fn consumes chunk i
  for (int i = 0; i < 2; ++i)
    if (!c->pin[i]) HIPCHK(hipHostMalloc((void **)&c->pin[i], kPinChunk, hipHostMallocDefault));
  struct Chunk { uint64_t n; };
  int cur = 0;
  uint64_t fill[2] = {0, 0};
  bool inflight[2] = {false, false};
  auto flush = [&](int i) -> int {
    if (!inflight[i]) return ZPAQHIP_OK;
    HIPCHK(hipEventSynchronize(c->ev_pin[i]));
    inflight[i] = false;
    for (uint64_t p = 0; p < fill[i];) {
      const int n = (int)std::min<uint64_t>(fill[i] - p, 1u << 30);
      if (sink.wr(sink.user, c->pin[i] + p, n) < 0) { set_err(err, ZPAQHIP_E_CALLBACK, -1, -1); return ZPAQHIP_E_CALLBACK; }
      p += (uint64_t)n;
    }
    sink.total += fill[i];
    fill[i] = 0;
    return ZPAQHIP_OK;
  };
  for (const Piece &pc : pieces) {
    for (uint64_t done = 0; done < pc.n;) {
      const uint64_t n = std::min<uint64_t>(pc.n - done, kPinChunk - fill[cur]);
      HIPCHK(hipMemcpyAsync(c->pin[cur] + fill[cur], pc.p + done, (size_t)n, hipMemcpyDeviceToHost, c->s_out));
      fill[cur] += n; done += n;
      if (fill[cur] == kPinChunk) {
        HIPCHK(hipEventRecord(c->ev_pin[cur], c->s_out));
        inflight[cur] = true;
        cur ^= 1;
        int rc = flush(cur);                            // the other buffer must be free before it is refilled
        if (rc) return rc;
      }
    }
  }
  if (fill[cur]) { HIPCHK(hipEventRecord(c->ev_pin[cur], c->s_out)); inflight[cur] = true; }
  int rc = flush(cur ^ 1);
  if (!rc) rc = flush(cur);
  return rc;
}

struct SegSink {                          // zpaqhip_decompress_segments: per-segment records in stream order
  std::vector<zpaqhip_seg_result> res;
  std::vector<zpaqhip_segment> segs;
};

// The pipeline.  `tolerate`: per-segment data errors do not end the call (zpaqhip_decompress_segments).
// *stream_error: the returned code describes the stream (damage behind the delivered blocks), not a failure of the call.
int run_pipeline_impl(zpaqhip_ctx *c, Source &src, Sinkk &sink, const zpaqhip_opts &opts, bool tolerate, SegSink *segsink,
                      zpaqhip_err *err, bool *stream_error);
// Every exit with an error leaves nothing in flight: copies on s_out / s_in may still be writing caller-owned (pageable) host
// memory or reading staging the caller is about to free (the multi-device worker destroyed its spill vectors and
// the context right after an early return).
int run_pipeline(zpaqhip_ctx *c, Source &src, Sinkk &sink, const zpaqhip_opts &opts, bool tolerate, SegSink *segsink,
                 zpaqhip_err *err, bool *stream_error = nullptr) {
  const int rc = run_pipeline_impl(c, src, sink, opts, tolerate, segsink, err, stream_error);
  if (rc) {
    if (c->s_out) (void)hipStreamSynchronize(c->s_out);
    if (c->s_in) (void)hipStreamSynchronize(c->s_in);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
  }
  return rc;
}
int run_pipeline_impl(zpaqhip_ctx *c, Source &src, Sinkk &sink, const zpaqhip_opts &opts, bool tolerate, SegSink *segsink,
                      zpaqhip_err *err, bool *stream_error) {
  if (stream_error) *stream_error = false;
  HIPCHK(hipSetDevice(c->device));
  if (!c->s_in) {
    HIPCHK(hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      HIPCHK(hipEventCreate(&c->ev_in[i])); HIPCHK(hipEventCreate(&c->ev_pin[i]));
      HIPCHK(hipEventCreate(&c->ev_d[i][0])); HIPCHK(hipEventCreate(&c->ev_d[i][1]));
    }
    HIPCHK(hipEventCreate(&c->ev_h0)); HIPCHK(hipEventCreate(&c->ev_h1));
  }
  zpaqhip_stats acc{};
  c->d2h_acc = 0; c->d_pending[0] = c->d_pending[1] = false;
  Batch bt[2];
  zh_pending P;
  size_t blk0 = 0, seg0 = 0;
  int final_rc = ZPAQHIP_OK;
  zpaqhip_err final_err{};
  bool have_prev = false, stop = false;
  int cur = 0;
  float h2d_ms = 0;

  auto upload = [&](Batch &b) -> int {
    if (b.so.blocks.empty()) return ZPAQHIP_OK;
    HIPCHK(c->in2[b.slot].reserve(b.len + 16));
    HIPCHK(hipEventRecord(c->ev_h0, c->s_in));
    HIPCHK(hipMemcpyAsync(c->in2[b.slot].p, b.h, b.len, hipMemcpyHostToDevice, c->s_in));
    HIPCHK(hipEventRecord(c->ev_h1, c->s_in));
    HIPCHK(hipEventRecord(c->ev_in[b.slot], c->s_in));
    return ZPAQHIP_OK;
  };

  int rc = next_batch(src, bt[cur], blk0, seg0, (size_t)opts.batch_blocks, err);
  if (rc) return rc;
  bt[cur].slot = cur;
  rc = upload(bt[cur]);
  if (rc) return rc;
  for (;;) {
    Batch &B = bt[cur];
    const bool have = !B.so.blocks.empty();
    if (have) {
      rc = plan_staging(c, B, err);
      if (rc) return rc;
      B.res.assign(B.so.segs.size(), zpaqhip_seg_result{});
      HIPCHK(hipStreamWaitEvent(c->stream, c->ev_in[B.slot], 0));
      rc = decode_launch(c, c->in2[B.slot].p, B.h, B.len, B.so.blocks.data(), B.so.blocks.size(), B.so.segs.data(),
                         B.so.segs.size(), nullptr, 0, c->out2[B.slot].p, B.off.data(), B.cap.data(), opts, c->stream, P, err);
      if (rc) {                                          // (device memory, HIP): what is already decoded is delivered first
        rebase_err(err, B);
        if (have_prev) { zpaqhip_err e3{}; (void)drain_batch(c, bt[cur ^ 1], bt[cur ^ 1].so.blocks.size(), 0, sink, &e3); (void)hipStreamSynchronize(c->s_out); }
        return rc;
      }
    }
    // ---- while the kernels run: deliver the previous batch, fetch and upload the next one
    Batch &prev = bt[cur ^ 1];
    if (have_prev) {
      rc = drain_batch(c, prev, prev.so.blocks.size(), 0, sink, err);
      if (rc) return rc;
      have_prev = false;
    }
    bool more = have && !B.last;
    if (more) {
      { float ms = 0; if (hipEventElapsedTime(&ms, c->ev_h0, c->ev_h1) == hipSuccess) h2d_ms += ms; else (void)hipGetLastError(); }
      HIPCHK(hipStreamSynchronize(c->s_out));          // prev's device buffers are about to be reused
      rc = next_batch(src, prev, blk0 + B.so.blocks.size(), seg0 + B.so.segs.size(), (size_t)opts.batch_blocks, err);
      if (rc) return rc;
      prev.slot = cur ^ 1;
      rc = upload(prev);
      if (rc) return rc;
    }
    if (!have) {
      if (B.rc_after) { final_rc = B.rc_after; final_err = B.err_after; rebase_err(&final_err, B); }
      break;
    }
    // ---- finish this batch
    rc = decode_finish(c, P, B.res.data(), err);
    acc.kernel_ms += c->stats.kernel_ms; acc.launches += c->stats.launches;
    acc.blocks += c->stats.blocks; acc.in_bytes += c->stats.in_bytes; acc.model_bytes += c->stats.model_bytes;
    acc.concurrent = std::max(acc.concurrent, c->stats.concurrent); acc.kernel_kind = std::max(acc.kernel_kind, c->stats.kernel_kind);
    if (rc && !data_error(rc)) { rebase_err(err, B); return rc; }
    rc = settle_batch(c, B, opts, tolerate, acc, err);
    if (rc) { rebase_err(err, B); return rc; }
    if (opts.verify_sha1) {
      std::vector<int> bad;
      rc = sha1_mismatches(c, B, bad, err);
      if (rc) return rc;
      for (size_t s2 = 0; s2 < bad.size(); ++s2)
        if (bad[s2] && B.res[s2].status == ZPAQHIP_OK) B.res[s2].status = ZPAQHIP_E_SHA1;
    }
    // first segment that failed: everything before it is delivered, then the call ends with its error (the reference
    // raises on reaching the damaged segment, Decompresser.cs:121-153)
    size_t ok_blocks = B.so.blocks.size();
    uint64_t partial = 0;
    if (!tolerate) {
      for (size_t b = 0; b < B.so.blocks.size() && !stop; ++b) {
        const zpaqhip_block &Bk = B.so.blocks[b];
        uint64_t before = 0;
        for (uint32_t i = 0; i < Bk.n_seg; ++i) {
          const zpaqhip_seg_result &r = B.res[Bk.first_seg + i];
          if (r.status != ZPAQHIP_OK) {
            stop = true; ok_blocks = b; partial = before;
            final_rc = r.status == ZH_E_SKIPPED ? ZPAQHIP_E_CORRUPT : r.status;
            set_err(&final_err, final_rc, (int)(B.blk0 + b), (int)(B.seg0 + Bk.first_seg + i));
            break;
          }
          before += r.out_len;
        }
      }
    }
    if (segsink) {
      uint64_t base = sink.total;                       // stream-order offset of this batch's first byte ...
      if (have_prev) base += 0;
      uint64_t run = 0;
      for (size_t b = 0; b < B.so.blocks.size(); ++b) {
        const zpaqhip_block &Bk = B.so.blocks[b];
        const uint64_t st_off = (B.off[b] >> 63) ? (B.off[b] & ~(1ull << 63)) : B.off[b];
        for (uint32_t i = 0; i < Bk.n_seg; ++i) {
          zpaqhip_seg_result r = B.res[Bk.first_seg + i];
          r.out_off = base + run + (r.out_off - st_off);
          segsink->res.push_back(r);
          zpaqhip_segment sg = B.so.segs[Bk.first_seg + i];
          sg.block += (uint32_t)B.blk0;
          segsink->segs.push_back(sg);
        }
        run += B.real[b];
      }
    }
    if (stop) {
      rc = drain_batch(c, B, ok_blocks, partial, sink, err);
      if (rc) return rc;
      break;
    }
    if (!more) {
      rc = drain_batch(c, B, B.so.blocks.size(), 0, sink, err);
      if (rc) return rc;
      if (B.rc_after) { final_rc = B.rc_after; final_err = B.err_after; rebase_err(&final_err, B); }
      break;
    }
    have_prev = true;
    blk0 += B.so.blocks.size(); seg0 += B.so.segs.size();
    cur ^= 1;
  }
  { float ms = 0; if (hipEventElapsedTime(&ms, c->ev_h0, c->ev_h1) == hipSuccess) h2d_ms += ms; else (void)hipGetLastError(); }
  HIPCHK(hipStreamSynchronize(c->s_out));
  for (int i = 0; i < 2; ++i) { int hrc = harvest_d2h(c, i, err); if (hrc) return hrc; }
  c->stats = acc;
  c->stats.h2d_ms = h2d_ms;
  c->stats.d2h_ms = c->d2h_acc;                          // copy-stream time of the plaintext (memory forms; it overlaps the kernels)
  c->stats.out_bytes = sink.total;
  if (final_rc) { if (err) *err = final_err; if (stream_error) *stream_error = true; return final_rc; }
  return ZPAQHIP_OK;
}

}  // namespace

extern "C" {

int zpaqhip_decompress(zpaqhip_ctx *c, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len,
                       const zpaqhip_opts *opts_in, zpaqhip_err *err) {
  if (!c || (!in && in_len) || !out_len || (!out && out_cap)) { set_err(err, ZPAQHIP_E_ARG, -1, -1); return ZPAQHIP_E_ARG; }
  const zpaqhip_opts opts = resolve_opts(opts_in);
  Source src; src.mem = in ? in : (const uint8_t *)""; src.mem_len = in_len;
  Sinkk sink; sink.mem = out; sink.cap = out_cap;
  *out_len = 0;
  int rc = run_pipeline(c, src, sink, opts, false, nullptr, err);
  *out_len = (size_t)sink.total;
  if (rc) return rc;
  if (sink.total > out_cap) { set_err(err, ZPAQHIP_E_OUTPUT_FULL, -1, -1); return ZPAQHIP_E_OUTPUT_FULL; }
  return ZPAQHIP_OK;
}

// Estimated decode cost of every block: plaintext bytes (the decimal size in the comments when it is plausible, else
// 4 x the coded bytes) x the cycles per plaintext byte of the kernel the block's header selects (measured, profiles/r03;
// a block's decode time is its bit count x the depth of its model, not its coded size).  Host-side only.
static constexpr uint64_t kCm1Marker = ~0ull;
int zpaqhip_block_costs(const uint8_t *in, size_t in_len, const zpaqhip_block *blocks, size_t n_blocks,
                        const zpaqhip_segment *segs, size_t n_segs, uint64_t *cost, zpaqhip_err *err) {
  if ((!in && in_len) || (!blocks && n_blocks) || (!segs && n_segs) || (!cost && n_blocks)) { set_err(err, ZPAQHIP_E_ARG, -1, -1); return ZPAQHIP_E_ARG; }
  std::map<std::string, uint64_t> per_byte;
  for (size_t b = 0; b < n_blocks; ++b) {
    const zpaqhip_block &B = blocks[b];
    if (B.hdr_off + B.hdr_len > in_len || (uint64_t)B.first_seg + B.n_seg > n_segs) {
      set_err(err, ZPAQHIP_E_ARG, (int)b, -1, "block table does not match the stream");
      return ZPAQHIP_E_ARG;
    }
    std::string key((const char *)in + B.hdr_off, B.hdr_len);
    auto it = per_byte.find(key);
    if (it == per_byte.end()) {
      ZhModel m;
      std::vector<uint8_t> code;
      zpaqhip_err e2{};
      uint64_t w = 6000;                                  // a header the model builder refuses costs nothing real; keep it finite
      if (build_model(in + B.hdr_off, B.hdr_len, m, code, &e2) == ZPAQHIP_OK) {
        const uint32_t fam = m.kind & 255u;
        const bool pcomp = (m.ph | m.pm) != 0;
        // measured cycles of the owning workgroup per plaintext byte (profiles/r03: 256 x 4 MiB, one block per CU), rounded;
        // estimates for the fallback kernels
        w = fam == ZH_FAM_STORE ? (pcomp ? 120u : 30u)                // zh_store.hip: wave-wide copy / LZ77 / inverse BWT
            : fam == ZH_FAM_CM1 ? kCm1Marker                         // zh_cm.hip: by coded / plain ratio, below
            : fam == ZH_FAM_CHAIN + 1 ? 3800u : fam == ZH_FAM_CHAIN + 2 ? 6800u : fam == ZH_FAM_CHAIN + 3 ? 16600u   // zh_nibble.hip min / mid, zh_chain2.hip max (profiles/r05)
            : fam == ZH_FAM_CHAIN_MID8 ? 7000u                        // zh_nibble.hip, eight mixer inputs
            : fam == ZH_FAM_CHAIN_MIN1 ? 3800u                        // zh_nibble.hip, one ICM on min's loop
            : fam == ZH_FAM_CHAIN ? 4000u + 2200u * m.n               // zh_chain.hip: level walk at run time
            : 10000u + 16000u * m.n;                                  // zh_generic.hip: one lane, tables in HBM
        if (pcomp && fam != ZH_FAM_STORE) w += w == kCm1Marker ? (uint64_t)-1 : 1500u;       // (marker - 1: single CM with a post-processor)
      }
      it = per_byte.emplace(key, w).first;
    }
    uint64_t coded = 0;
    for (uint32_t i = 0; i < B.n_seg; ++i) coded += segs[B.first_seg + i].data_len;
    const bool hinted = B.usize_hint != UINT64_MAX && B.usize_hint <= (1ull << 40);
    const uint64_t plain = hinted ? B.usize_hint : coded * 4;
    uint64_t w = it->second;
    if (w >= kCm1Marker - 1) {
      const uint64_t pp = w == kCm1Marker ? 0u : 1500u;
      // zh_cm.hip: the byte loop costs the same on any data, a window miss ~1 400 cycles more, and the miss rate of an
      // order-1 context follows the data's entropy — which the block shows as coded / plain.  Measured at 256 blocks per GPU
      // (profiles/r04): text 0.45 -> 920 cycles per byte, x86-like 0.75 -> 1 250, random 1.03 -> 2 020; piecewise linear
      // between them (a block without a size hint counts as text)
      const double rho = hinted && plain ? (double)coded / (double)plain : 0.45;
      w = pp + (rho <= 0.45 ? 920u : rho <= 0.75 ? 920u + (uint64_t)((rho - 0.45) * 1100.0) : 1250u + (uint64_t)((std::min(rho, 1.1) - 0.75) * 2750.0));
    }
    cost[b] = std::max<uint64_t>(1, plain) * w;
  }
  return ZPAQHIP_OK;
}

// Contexts zpaqhip_decompress_multi has used stay alive between calls (the arena of a context is tens of GB for the larger
// models: allocating and faulting it in on every call cost more than the decode of a small archive): a device thread takes
// one of its device's from this pool, or makes one, and puts it back.  zpaqhip_multi_trim() destroys the idle ones.
namespace {
std::mutex g_pool_mu;
std::map<int, std::vector<zpaqhip_ctx *>> g_pool;
zpaqhip_ctx *pool_take(int device, int *rc, zpaqhip_err *err) {
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto &v = g_pool[device];
    if (!v.empty()) { zpaqhip_ctx *c = v.back(); v.pop_back(); *rc = ZPAQHIP_OK; return c; }
  }
  zpaqhip_ctx *c = nullptr;
  *rc = zpaqhip_ctx_create(device, &c, err);
  return *rc ? nullptr : c;
}
void pool_give(zpaqhip_ctx *c) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_pool[c->device].push_back(c);
}
}  // namespace
// Destroys the idle pooled contexts of `device` beyond the first `keep`; returns how many went.
static size_t pool_trim_device(int device, size_t keep) {
  std::vector<zpaqhip_ctx *> gone;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool.find(device);
    if (it == g_pool.end()) return 0;
    while (it->second.size() > keep) { gone.push_back(it->second.back()); it->second.pop_back(); }
  }
  for (zpaqhip_ctx *c : gone) zpaqhip_ctx_destroy(c);
  return gone.size();
}

extern "C" void zpaqhip_multi_trim(void) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (auto &kv : g_pool) for (zpaqhip_ctx *c : kv.second) zpaqhip_ctx_destroy(c);
  g_pool.clear();
}

// Several GPUs of one node: one context and one host thread per entry of `devices`, all pulling from ONE work queue.
//   queue   the blocks sorted by estimated cost (zpaqhip_block_costs) are dealt into K = ceil(n / queue_blocks) chunks,
//           chunk k = every K-th block of that order starting at k (each chunk is a cross-section of the cost
//           distribution; queue_blocks defaults to min(256 — one block per CU —, a quarter of a device's share)); a device thread takes the
//           next chunk (one atomic counter) whenever it has finished one, so a GPU that is faster, or whose chunks
//           turned out cheaper than estimated, simply takes more of them
//   placing block b goes to out + (sum of the plausible comment sizes before it); a block whose comment carries no
//           plausible size, or the wrong one, is kept whole in a host buffer by the thread that decoded it and only
//           THAT block (and the blocks behind it, by memmove) is put in place when all sizes are known — nothing is
//           decoded twice
//   errors  a damaged block ends its chunk there; the other chunks go on, so that every block before the first damaged
//           one of the stream is delivered, *out_len says how many bytes that is, and the call returns that block's error
int zpaqhip_decompress_multi_stats(const int *devices, size_t n_dev, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                   size_t *out_len, const zpaqhip_opts *opts_in, zpaqhip_stats *per_device, zpaqhip_err *err) {
  if (!devices || !n_dev || (!in && in_len) || !out_len || (!out && out_cap)) { set_err(err, ZPAQHIP_E_ARG, -1, -1); return ZPAQHIP_E_ARG; }
  zpaqhip_opts opts = resolve_opts(opts_in);
  *out_len = 0;
  if (per_device) memset(per_device, 0, n_dev * sizeof(zpaqhip_stats));
  ScanOut so;
  zpaqhip_err scan_err{};
  const int scan_rc = scan_stream(in ? in : (const uint8_t *)"", in_len, so, &scan_err);   // blocks before damage are decoded
  const size_t nb = so.blocks.size();
  if (!nb) { if (scan_rc && err) *err = scan_err; return scan_rc; }
  // ---- the queue
  std::vector<uint64_t> cost(nb, 0);
  { int rc = zpaqhip_block_costs(in, in_len, so.blocks.data(), nb, so.segs.data(), so.segs.size(), cost.data(), err); if (rc) return rc; }
  std::vector<size_t> order(nb);
  std::iota(order.begin(), order.end(), (size_t)0);
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
  // blocks per pull: what the caller says, or a chunk that fills a GPU (256: one block per CU; 512 where every block is a
  // single-CM one: two per CU) — but at least four pulls per device, or the queue has nothing to rebalance with
  // (multigpu.default_queue_blocks is the same rule)
  size_t chunk_blocks = (size_t)opts.queue_blocks;
  if (!chunk_blocks) {
    bool all_cm1 = true;
    for (size_t b = 0; b < nb; ++b) all_cm1 = all_cm1 && so.blocks[b].n_comp == 1;
    const size_t full = all_cm1 ? 512 : 256, per4 = (nb + 4 * n_dev - 1) / (4 * n_dev);
    chunk_blocks = n_dev <= 1 ? full : std::max<size_t>(1, std::min(full, per4));     // (one taker: nothing to balance)
  }
  const size_t K = (nb + chunk_blocks - 1) / chunk_blocks;
  std::atomic<size_t> next_chunk{0};
  std::atomic<bool> abort_all{false};
  // ---- placing
  std::vector<uint64_t> slot_off(nb + 1, 0), slot_cap(nb, 0), real(nb, 0);
  for (size_t b = 0; b < nb; ++b) {
    const uint64_t h = so.blocks[b].usize_hint;
    const bool plausible = h != UINT64_MAX && h <= (1ull << 40) && slot_off[b] + h <= out_cap;
    slot_cap[b] = plausible ? h : 0;
    slot_off[b + 1] = slot_off[b] + slot_cap[b];
  }
  std::vector<std::vector<uint8_t>> spill(nb);
  std::vector<uint8_t> done(nb, 0);
  {  // idle pooled contexts beyond what this call uses on a device give their memory back before the workers size their arenas
    std::map<int, size_t> use;
    for (size_t r = 0; r < n_dev; ++r) ++use[devices[r]];
    for (auto &kv : use) pool_trim_device(kv.first, kv.second);
  }
  struct Job { int rc = 0; zpaqhip_err err{}; long bad = -1; zpaqhip_stats st{}; };
  std::vector<Job> jobs(n_dev);
  auto worker = [&](size_t r) {
    Job &J = jobs[r];
    zpaqhip_ctx *c = pool_take(devices[r], &J.rc, &J.err);
    if (!c) { abort_all = true; return; }
    c->mem_share = 1;
    for (size_t k = 0; k < n_dev; ++k) c->mem_share += k != r && devices[k] == devices[r];
    std::vector<uint8_t> sub;
    std::vector<size_t> ids;
    std::vector<uint64_t> place, cap, sizes;
    std::vector<std::vector<uint8_t>> sp;
    while (!abort_all) {
      const size_t k = next_chunk.fetch_add(1);
      if (k >= K) break;
      ids.clear();
      for (size_t i = k; i < nb; i += K) ids.push_back(order[i]);
      std::sort(ids.begin(), ids.end());                 // stream order inside the chunk: the pipeline delivers in order
      sub.clear(); place.clear(); cap.clear();
      // a chunk of consecutive blocks (one chunk for the whole archive, typically) is decoded where it lies; a strided one
      // is gathered first
      const bool run = ids.back() - ids.front() + 1 == ids.size();
      for (size_t b : ids) {
        if (!run) sub.insert(sub.end(), in + so.blocks[b].tag_off, in + so.blocks[b].end_off);
        place.push_back(slot_off[b]); cap.push_back(slot_cap[b]);
      }
      sizes.assign(ids.size(), 0);
      sp.assign(ids.size(), std::vector<uint8_t>());
      Source src;
      if (run) { src.mem = in + so.blocks[ids.front()].tag_off; src.mem_len = (size_t)(so.blocks[ids.back()].end_off - so.blocks[ids.front()].tag_off); }
      else { src.mem = sub.data(); src.mem_len = sub.size(); }
      Sinkk sink;
      sink.mem = out; sink.cap = out_cap; sink.place = place.data(); sink.place_cap = cap.data(); sink.sizes = &sizes; sink.spill = &sp;
      zpaqhip_err e2{};
      bool stream_error = false;
      const int rc = run_pipeline(c, src, sink, opts, false, nullptr, &e2, &stream_error);
      J.st.kernel_ms += c->stats.kernel_ms; J.st.h2d_ms += c->stats.h2d_ms; J.st.d2h_ms += c->stats.d2h_ms;
      J.st.blocks += c->stats.blocks; J.st.in_bytes += c->stats.in_bytes; J.st.out_bytes += c->stats.out_bytes;
      J.st.model_bytes += c->stats.model_bytes; J.st.launches += 1;
      J.st.concurrent = std::max(J.st.concurrent, c->stats.concurrent); J.st.kernel_kind = std::max(J.st.kernel_kind, c->stats.kernel_kind);
      size_t good = ids.size();
      if (rc) {
        const bool at_block = (stream_error || data_error(rc)) && e2.block >= 0 && (size_t)e2.block < ids.size();
        if (!at_block) { if (!J.rc) { J.rc = rc; J.err = e2; J.bad = -1; } abort_all = true; break; }   // the call itself failed
        good = (size_t)e2.block;
        const long g = (long)ids[good];
        if (J.bad < 0 || g < J.bad) { J.rc = rc; J.err = e2; J.bad = g; }
      }
      for (size_t j = 0; j < good; ++j) { real[ids[j]] = sizes[j]; spill[ids[j]].swap(sp[j]); done[ids[j]] = 1; }
    }
    if (J.rc && J.bad < 0) zpaqhip_ctx_destroy(c);       // a failure of the call itself (HIP, memory): do not keep that context
    else pool_give(c);
  };
  {
    std::vector<std::thread> th;
    for (size_t r = 0; r < n_dev; ++r) th.emplace_back(worker, r);
    for (auto &t : th) t.join();
  }
  if (per_device) for (size_t r = 0; r < n_dev; ++r) per_device[r] = jobs[r].st;
  // ---- outcome: a failure of the call itself first; else the first damaged block of the stream
  for (size_t r = 0; r < n_dev; ++r)
    if (jobs[r].rc && jobs[r].bad < 0) { if (err) *err = jobs[r].err; return jobs[r].rc; }
  size_t limit = nb;
  int data_rc = 0;
  zpaqhip_err data_err{};
  for (size_t r = 0; r < n_dev; ++r)
    if (jobs[r].rc && (size_t)jobs[r].bad < limit) { limit = (size_t)jobs[r].bad; data_rc = jobs[r].rc; data_err = jobs[r].err; data_err.block = (int32_t)limit; data_err.segment = -1; }
  for (size_t b = 0; b < limit; ++b)
    if (!done[b]) { set_err(err, ZPAQHIP_E_HIP, (int)b, -1, "multi-device queue: a block was left undecoded"); return ZPAQHIP_E_HIP; }
  // ---- put the blocks whose size was not the promised one in place
  std::vector<uint64_t> new_off(limit + 1, 0);
  bool moved = false;
  for (size_t b = 0; b < limit; ++b) { new_off[b + 1] = new_off[b] + real[b]; moved |= real[b] != slot_cap[b]; }
  *out_len = (size_t)new_off[limit];
  if (new_off[limit] > out_cap) { set_err(err, ZPAQHIP_E_OUTPUT_FULL, -1, -1); return ZPAQHIP_E_OUTPUT_FULL; }
  if (moved) {
    // blocks that lie in their slot (real <= promised) and move up: last first; those that move down: first first;
    // a block kept in a host buffer is written last, when every other block is where it belongs
    for (size_t b = limit; b-- > 0;)
      if (spill[b].empty() && real[b] && new_off[b] > slot_off[b]) memmove(out + new_off[b], out + slot_off[b], (size_t)real[b]);
    for (size_t b = 0; b < limit; ++b)
      if (spill[b].empty() && real[b] && new_off[b] < slot_off[b]) memmove(out + new_off[b], out + slot_off[b], (size_t)real[b]);
    for (size_t b = 0; b < limit; ++b)
      if (!spill[b].empty()) memcpy(out + new_off[b], spill[b].data(), (size_t)real[b]);
  }
  if (data_rc) { if (err) *err = data_err; return data_rc; }
  if (scan_rc) { if (err) *err = scan_err; return scan_rc; }
  return ZPAQHIP_OK;
}

int zpaqhip_decompress_multi(const int *devices, size_t n_dev, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                             size_t *out_len, const zpaqhip_opts *opts_in, zpaqhip_err *err) {
  return zpaqhip_decompress_multi_stats(devices, n_dev, in, in_len, out, out_cap, out_len, opts_in, nullptr, err);
}

int zpaqhip_decompress_segments(zpaqhip_ctx *c, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                size_t *out_len, zpaqhip_seg_result *results, size_t result_cap, size_t *n_results,
                                const zpaqhip_opts *opts_in, zpaqhip_err *err) {
  if (!c || (!in && in_len) || !out_len || !n_results || (!out && out_cap) || (!results && result_cap)) {
    set_err(err, ZPAQHIP_E_ARG, -1, -1);
    return ZPAQHIP_E_ARG;
  }
  const zpaqhip_opts opts = resolve_opts(opts_in);
  Source src; src.mem = in ? in : (const uint8_t *)""; src.mem_len = in_len;
  Sinkk sink; sink.mem = out; sink.cap = out_cap;
  SegSink ss;
  *out_len = 0; *n_results = 0;
  bool stream_error = false;
  int rc = run_pipeline(c, src, sink, opts, true, &ss, err, &stream_error);
  *out_len = (size_t)sink.total;
  *n_results = ss.res.size();
  if (rc && !stream_error) return rc;                    // the call itself failed (device, arguments, callback)
  const int frame_rc = rc;                               // a framing error after the last good block: reported below
  if (ss.res.size() > result_cap) { set_err(err, ZPAQHIP_E_ARG, -1, -1, "result table too small"); return ZPAQHIP_E_ARG; }
  if (!ss.res.empty()) memcpy(results, ss.res.data(), ss.res.size() * sizeof(zpaqhip_seg_result));
  if (sink.total > out_cap) { set_err(err, ZPAQHIP_E_OUTPUT_FULL, -1, -1); return ZPAQHIP_E_OUTPUT_FULL; }
  return frame_rc;
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
  Source src; src.rd = read_fn; src.user = user;
  Sinkk sink; sink.wr = write_fn; sink.user = user;
  return run_pipeline(c, src, sink, opts, false, nullptr, err);
}

}  // extern "C"
