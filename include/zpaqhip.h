/* include/zpaqhip.h — C ABI of libzpaqhip.so: MI355X (gfx950) ZPAQ block decompression.
 *
 * This is the drop-in boundary for the reference's decompression path.  The
 * reference (mnadareski/ZPAQSharp) has no FFI layer of its own — the path is a
 * plain class API (Decompresser.cs:11-221 driven by LibZPAQ.decompress,
 * LibZPAQ.cs:65-79).  A maintainer binds these entry points with
 * [DllImport("zpaqhip", CallingConvention = CallingConvention.Cdecl)] and keeps
 * the Reader/Writer-facing classes unchanged; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain C types only; no exceptions, exit() or longjmp cross the boundary;
 *   - every function returns 0 (ZPAQHIP_OK) or a negative zpaqhip_status; the
 *     codes map 1:1 onto the reference's error() messages (zpaqhip_strerror);
 *   - the library never keeps a caller pointer after a call returns;
 *   - a zpaqhip_ctx is bound to one GPU and is not thread-safe; distinct
 *     contexts are independent (libzpaq contract, reference LICENSE:44-46);
 *   - there is NO CPU decode path in this library: without a usable HIP device
 *     zpaqhip_ctx_create fails with ZPAQHIP_E_NO_DEVICE.
 */
#ifndef ZPAQHIP_H
#define ZPAQHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZPAQHIP_ABI_VERSION 1

typedef enum zpaqhip_status {
  ZPAQHIP_OK = 0,
  /* Decoder.cs */
  ZPAQHIP_E_CORRUPT = -1,        /* "archive corrupted"               Decoder.cs:141 */
  ZPAQHIP_E_EOF = -2,            /* "unexpected end of file"          Decoder.cs:154 */
  ZPAQHIP_E_EOS = -3,            /* "decoding end of stream"          Decoder.cs:43  */
  /* ZPAQL.cs */
  ZPAQHIP_E_ZPAQL = -4,          /* "ZPAQL execution error"           ZPAQL.cs:1314-1317 */
  ZPAQHIP_E_HEADER = -5,         /* header errors                     ZPAQL.cs:128-148 */
  ZPAQHIP_E_HM_TOO_BIG = -6,     /* "H too big" / "M too big"         ZPAQL.cs:1021-1022 */
  /* Predictor.cs:100-166 component limit checks */
  ZPAQHIP_E_COMPONENT = -7,
  /* PostProcessor.cs */
  ZPAQHIP_E_PP_EOS = -8,         /* "Unexpected EOS"                  PostProcessor.cs:43,52,57,68 */
  ZPAQHIP_E_PP_TYPE = -9,        /* "unknown post processing type"    PostProcessor.cs:45 */
  ZPAQHIP_E_PP_EMPTY = -10,      /* "Empty PCOMP"                     PostProcessor.cs:59 */
  /* Decompresser.cs framing */
  ZPAQHIP_E_LEVEL = -11,         /* "unsupported ZPAQ level"/"ZPAQL type" Decompresser.cs:49-50 */
  ZPAQHIP_E_SEGMENT = -12,       /* "missing segment or end of block" Decompresser.cs:90 */
  ZPAQHIP_E_FRAMING_EOF = -13,   /* "unexpected EOF"                  Decompresser.cs:77,103 */
  ZPAQHIP_E_RESERVED = -14,      /* "missing reserved byte"           Decompresser.cs:107 */
  ZPAQHIP_E_SEGEND = -15,        /* "missing end of segment marker"   Decompresser.cs:193 */
  /* build-specific */
  ZPAQHIP_E_OUTPUT_FULL = -20,   /* caller's output buffer too small (out_len still reports the need) */
  ZPAQHIP_E_SHA1 = -21,          /* stored SHA-1 does not match decoded segment */
  ZPAQHIP_E_NO_DEVICE = -22,     /* no usable HIP device / kernels not loadable */
  ZPAQHIP_E_DEVICE_MEM = -23,    /* model does not fit device memory ("Out of memory") */
  ZPAQHIP_E_HIP = -24,           /* HIP runtime error (message in zpaqhip_err) */
  ZPAQHIP_E_ARG = -25,           /* bad argument / table too small */
  ZPAQHIP_E_BUDGET = -26,        /* ZPAQL instruction budget exhausted (runaway program guard) */
  ZPAQHIP_E_CALLBACK = -27       /* read/write callback failed */
} zpaqhip_status;

typedef struct zpaqhip_err {
  int32_t code;        /* zpaqhip_status */
  int32_t block;       /* block index the error belongs to, or -1 */
  int32_t segment;     /* segment index (global), or -1 */
  char msg[116];       /* NUL-terminated English message (reference wording) */
} zpaqhip_err;

/* One block of the stream, as located by zpaqhip_scan (Decompresser.findBlock,
 * Decompresser.cs:29-58 + ZPAQL.read, ZPAQL.cs:112-156). */
typedef struct zpaqhip_block {
  uint64_t tag_off;      /* offset of the 13-byte tag + "zPQ" locator (start of the 16-byte string) */
  uint64_t hdr_off;      /* offset of the header (hsize low byte) */
  uint32_t hdr_len;      /* hsize + 2 */
  uint8_t level;         /* 1 or 2 */
  uint8_t n_comp;        /* header[6] */
  uint8_t hh, hm, ph, pm;
  uint16_t reserved;
  uint32_t first_seg;    /* index into the segment table */
  uint32_t n_seg;
  uint64_t end_off;      /* offset just past the block's 255 terminator */
  double model_mem;      /* ZPAQL.memory(), ZPAQL.cs:58-81 */
  uint64_t usize_hint;   /* sum of decimal sizes in the segment comments, or UINT64_MAX */
} zpaqhip_block;

/* One segment (Decompresser.findFilename/readComment/readSegmentEnd,
 * Decompresser.cs:67-108,163-194). */
typedef struct zpaqhip_segment {
  uint32_t block;        /* owning block */
  uint32_t flags;        /* bit0: SHA-1 present */
  uint64_t name_off;     /* filename bytes [name_off, name_off+name_len) */
  uint32_t name_len;
  uint32_t comment_len;
  uint64_t comment_off;
  uint64_t data_off;     /* first coded byte */
  uint64_t data_len;     /* coded bytes incl. the 4 zero bytes that end the segment */
  uint64_t usize_hint;   /* decimal size from the comment, or UINT64_MAX */
  uint8_t sha1[20];      /* stored checksum if flags&1 */
  uint32_t reserved;
} zpaqhip_segment;

/* Per-segment result of a decode. */
typedef struct zpaqhip_seg_result {
  int32_t status;        /* zpaqhip_status */
  uint32_t pp_state;     /* PostProcessor state at the end (1 PASS, 5 PROG) */
  uint64_t out_off;      /* where the segment's plaintext starts in the output */
  uint64_t out_len;      /* plaintext bytes produced (counted even past capacity) */
  uint64_t in_used;      /* coded bytes the decoder consumed; != data_len means the stream is damaged */
} zpaqhip_seg_result;

typedef struct zpaqhip_opts {
  uint32_t struct_size;       /* = sizeof(zpaqhip_opts) */
  uint32_t verify_sha1;       /* 1: check stored SHA-1 of every segment (Decompresser.cs:183-191 contract); hashed on the GPU */
  uint32_t max_concurrent;    /* blocks in flight per launch; 0 = auto (memory-bound) */
  uint32_t kernel;            /* 0 auto; 1 force the generic (one-lane) kernel; 2 / 6: single-CM blocks one / two per workgroup
                                 (auto: two when a launch has more than 256 of them); 3 prefer the lane-per-component kernel;
                                 4 lane-per-component without model specialisation; 5 the run-time-level form of the
                                 lane-per-component kernel also for the built-in min/mid/max models (cross-check);
                                 7 / 8: ignored (= auto) by the product build; a library built with `make EXPERIMENTS=1` runs the
                                 measured-and-not-kept three-wave form of mid/max there (tools/experiments/zh_chain3.hip);
                                 9: the built-in min / mid models on the bit-at-a-time kernels of rounds 2-4 (zh_chain2.hip) instead of
                                 the nibble-at-a-time ones (zh_nibble.hip): cross-check and A/B runs.  The models LibZPAQ.makeConfig
                                 writes for levels 3 / 4 (`ci1`, `...,1c0,0,511i2`, `ci1,1,1,1,2am`, `...2awm`) are known to
                                 zh_nibble.hip only: under 4, 5 and 9 they run on the lane-per-component kernel */
  uint64_t zpaql_budget;      /* runaway-program guard, per run() call: max ZPAQL instructions on the interpreter, max backward
                                 jumps in an ahead-of-time translated program (a translation checks where it can loop);
                                 0 = default (1<<32).  Exceeding it ends the block with ZPAQHIP_E_BUDGET */
  uint64_t batch_blocks;      /* whole-stream forms: blocks per pipeline batch; 0 = default (at least 512 blocks and 32 MiB
                                 of coded bytes per batch, so that every CU has a block) */
  uint64_t queue_blocks;      /* zpaqhip_decompress_multi: blocks per pull from the shared work queue; 0 = default: 256 (one block per
                                 CU of the device that takes the chunk; 512 where every block is a single-CM one), but no more than a
                                 quarter of a device's share of the blocks, so that every device pulls at least four times */
  uint64_t reserved[2];
} zpaqhip_opts;

/* Timing / accounting of the last decode call on a context (the reference's
 * stat() hooks are stubs: Predictor.cs:224-227, Decompresser.cs:196-199). */
typedef struct zpaqhip_stats {
  double kernel_ms;           /* HIP-event time of the decode kernel(s), on the launch stream */
  double init_ms;             /* HIP-event time spent initialising model tables (0 if fused) */
  double h2d_ms, d2h_ms;      /* host<->device copies done by the call (0 for the device form) */
  uint64_t blocks;            /* blocks decoded */
  uint64_t in_bytes;          /* coded bytes consumed */
  uint64_t out_bytes;         /* plaintext bytes produced */
  uint64_t model_bytes;       /* per-block model state (sum over decoded blocks) */
  uint32_t launches;          /* decode kernel launches */
  uint32_t concurrent;        /* blocks in flight per launch */
  uint32_t kernel_kind;       /* most specialised kernel used: 1 generic, 2 single-CM lanes, 3 lane-per-component */
  uint32_t reserved;
} zpaqhip_stats;

typedef struct zpaqhip_ctx zpaqhip_ctx;

/* Reader.read / Writer.write shaped callbacks (Reader.cs:14-25, Writer.cs:19-24). */
typedef int (*zpaqhip_read_fn)(void *user, uint8_t *buf, int n);        /* bytes read, 0 = EOF, <0 = error */
typedef int (*zpaqhip_write_fn)(void *user, const uint8_t *buf, int n); /* 0 = ok, <0 = error */

/* ---- library / context ------------------------------------------------- */
int zpaqhip_version(void);                       /* ZPAQHIP_ABI_VERSION */
const char *zpaqhip_strerror(int status);        /* reference message for a status */
int zpaqhip_device_count(void);                  /* usable HIP devices (0 without a GPU) */
int zpaqhip_ctx_create(int device, zpaqhip_ctx **ctx, zpaqhip_err *err);
void zpaqhip_ctx_destroy(zpaqhip_ctx *ctx);
int zpaqhip_last_stats(const zpaqhip_ctx *ctx, zpaqhip_stats *out);

/* ---- framing: replaces findBlock/findFilename/readComment/readSegmentEnd --
 * (Decompresser.cs:29-108,163-194; Decoder.skip, Decoder.cs:70-98).  Host-side,
 * no GPU needed.  Pass NULL tables with zero capacity to count only; returns
 * ZPAQHIP_E_ARG (with *n_blocks / *n_segs = required) if a table is too small. */
int zpaqhip_scan(const uint8_t *in, size_t in_len,
                 zpaqhip_block *blocks, size_t block_cap, size_t *n_blocks,
                 zpaqhip_segment *segs, size_t seg_cap, size_t *n_segs,
                 zpaqhip_err *err);

/* ---- whole stream, host buffers: replaces LibZPAQ.decompress(Reader, Writer)
 * (LibZPAQ.cs:65-79).  Blocks are decoded concurrently on the context's GPU;
 * plaintext is concatenated in stream order.  *out_len is always the total
 * plaintext size; ZPAQHIP_E_OUTPUT_FULL if it exceeds out_cap. */
int zpaqhip_decompress(zpaqhip_ctx *ctx, const uint8_t *in, size_t in_len,
                       uint8_t *out, size_t out_cap, size_t *out_len,
                       const zpaqhip_opts *opts, zpaqhip_err *err);

/* ---- same, but per-segment outcomes are handed back instead of aborting at the first bad
 * block: this is what a step-wise Decompresser mirror (findBlock / findFilename /
 * decompress(n) / readSegmentEnd, Decompresser.cs:29-194) needs to raise an error only when
 * the caller reaches the failing segment, as the reference does.  results[i] describes
 * segment i of zpaqhip_scan's table (out_off relative to `out`).  Returns
 * ZPAQHIP_E_ARG with *n_results = required count if result_cap is too small. */
int zpaqhip_decompress_segments(zpaqhip_ctx *ctx, const uint8_t *in, size_t in_len,
                                uint8_t *out, size_t out_cap, size_t *out_len,
                                zpaqhip_seg_result *results, size_t result_cap, size_t *n_results,
                                const zpaqhip_opts *opts, zpaqhip_err *err);

/* ---- same, streaming through Reader/Writer-shaped callbacks -------------- */
int zpaqhip_decompress_cb(zpaqhip_ctx *ctx, zpaqhip_read_fn read_fn, zpaqhip_write_fn write_fn,
                          void *user, const zpaqhip_opts *opts, zpaqhip_err *err);

/* ---- whole stream on several GPUs of one node (BASELINE.json configs[3]) ----
 * LibZPAQ.decompress(Reader, Writer) (LibZPAQ.cs:65-79) for a caller that owns more than one GPU and no
 * torch.distributed ranks (the C# host): one context and one host thread per entry of `devices` (a device may be
 * listed more than once; the contexts then share its memory budget).  The threads pull chunks of
 * opts->queue_blocks blocks from ONE work queue ordered by estimated cost (zpaqhip_block_costs), so a device that is
 * faster takes more chunks; plaintext arrives in `out` in stream order.  With a decimal size in every segment comment
 * (LibZPAQ.compressBlock writes it, LibZPAQ.cs:298-300) every device copies its blocks straight to their final place;
 * a block without a plausible size, or with a wrong one, is kept in a host buffer and put in place (it and what
 * follows it) when all sizes are known — no block is decoded twice.  A damaged block ends the call with its error
 * after every block before it has been delivered (*out_len = their bytes).  (Difference to zpaqhip_decompress /
 * zpaqhip_decompress_cb, which write as they decode like the reference: those also deliver the bytes the damaged block
 * itself produced before its error; this entry point places whole blocks only, so its plaintext is a prefix of theirs.
 * Pinned by test_multi_device_entry_point_with_contexts_sharing_this_gpu.)  No context is needed; the ones the call
 * makes are kept for the next call (zpaqhip_multi_trim); idle ones beyond what a call uses on a device, and all idle
 * ones of a device whose memory a later launch needs, are released by the library itself.
 * The _stats form also fills per_device[0..n_devices) (kernel_ms, blocks, ... summed over the chunks a device took;
 * launches = chunks). */
int zpaqhip_decompress_multi(const int *devices, size_t n_devices, const uint8_t *in, size_t in_len,
                             uint8_t *out, size_t out_cap, size_t *out_len,
                             const zpaqhip_opts *opts, zpaqhip_err *err);
int zpaqhip_decompress_multi_stats(const int *devices, size_t n_devices, const uint8_t *in, size_t in_len,
                                   uint8_t *out, size_t out_cap, size_t *out_len,
                                   const zpaqhip_opts *opts, zpaqhip_stats *per_device, zpaqhip_err *err);

/* zpaqhip_decompress_multi keeps the contexts it has used (one per device thread: tables, arena, streams) for its next
 * call; this destroys the idle ones and gives their device memory back. */
void zpaqhip_multi_trim(void);

/* Estimated decode cost of each block of a scanned stream, the weight every multi-GPU plan here uses: plaintext bytes
 * (the comment's decimal size when plausible, else 4 x coded bytes) x the cycles per plaintext byte of the kernel
 * the block's header selects (measured).  Decode time of a block is its bit count times the depth of its model
 * (Predictor.cs:245-475 runs once per bit), not its coded size.  Host-side, no GPU needed. */
int zpaqhip_block_costs(const uint8_t *in, size_t in_len, const zpaqhip_block *blocks, size_t n_blocks,
                        const zpaqhip_segment *segs, size_t n_segs, uint64_t *cost, zpaqhip_err *err);

/* ---- explicit block-table form, device-resident buffers ------------------
 * Replaces the per-block inner loop Decompresser.decompress(-1)
 * (Decompresser.cs:121-153) for a set of blocks.  `d_in` is the whole stream in
 * device memory; it must be 4-byte aligned and readable up to in_len rounded up to
 * a multiple of 4 (the kernels fetch the coded bytes as aligned dwords); `ids[0..n_ids)` selects the blocks this GPU decodes (NULL =
 * all, in table order) — the multi-GPU scheduler gives each rank its shard.
 * Block ids[i] writes its plaintext (all segments, concatenated) at
 * d_out + out_off[i], at most out_cap[i] bytes; bytes past the capacity are
 * counted, not written (status ZPAQHIP_E_OUTPUT_FULL).  results[] has one entry
 * per segment of the table (entries of blocks not in ids are left untouched);
 * out_off in a result is relative to d_out.  `hip_stream` is a hipStream_t (NULL
 * = the context's own stream); the call returns after the stream work has
 * completed.  `h_in` is an optional host copy of the same stream; when NULL the
 * few header bytes the host needs are fetched from d_in. */
int zpaqhip_decode_blocks_device(zpaqhip_ctx *ctx, const void *d_in, const uint8_t *h_in, size_t in_len,
                                 const zpaqhip_block *blocks, size_t n_blocks,
                                 const zpaqhip_segment *segs, size_t n_segs,
                                 const uint32_t *ids, size_t n_ids,
                                 void *d_out, const uint64_t *out_off, const uint64_t *out_cap,
                                 zpaqhip_seg_result *results,
                                 const zpaqhip_opts *opts, void *hip_stream, zpaqhip_err *err);

/* Device-side copies of the model-independent tables, for parity tests
 * (Predictor.cs:48-79 squash/stretch/dt/dt2k, StateTable.cs:21-149).
 * squash 4096 u16, stretch 32768 i16, dt 1024 i32, dt2k 256 i32, ns 1024 u8;
 * any pointer may be NULL.  The values are read back FROM THE DEVICE. */
int zpaqhip_read_device_tables(zpaqhip_ctx *ctx, uint16_t *squash, int16_t *stretch,
                               int32_t *dt, int32_t *dt2k, uint8_t *ns, zpaqhip_err *err);

/* Decompresser.pcomp(Writer) (Decompresser.cs:155-158, ZPAQL.write(out, true) ZPAQL.cs:158-179): the PCOMP
 * program of block `block` (index into zpaqhip_scan's table) as the reference writes it: length low byte, length
 * high byte, program bytes.  *out_len = 0 and ZPAQHIP_OK when the block has no PCOMP (pcomp() returns false).
 * ZPAQHIP_E_OUTPUT_FULL with *out_len = bytes needed when out_cap is too small.  The program is part of the coded
 * data: this call decodes the first bytes of the block (up to the end of the post-processor header) on the GPU. */
int zpaqhip_block_pcomp(zpaqhip_ctx *ctx, const uint8_t *in, size_t in_len, uint32_t block,
                        uint8_t *out, size_t out_cap, size_t *out_len, zpaqhip_err *err);

#ifdef __cplusplus
}
#endif
#endif
