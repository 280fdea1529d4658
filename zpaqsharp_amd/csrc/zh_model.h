// zh_model.h — structures shared by the host driver and the HIP kernels.
//
// Data layout in HBM:
//   stream   : the caller's compressed bytes, untouched.
//   models   : one ZhModel per distinct block header (component table, arena map).
//   code     : per model, the HCOMP program in a zero-padded window
//              [ZH_CODE_PAD zeros | program | ZH_CODE_PAD zeros] so that short
//              jumps that leave the program land on opcode 0 = error, as they do
//              in the reference's header layout (ZPAQL.cs:112-156, 128-byte gap).
//   arena    : one slot per block in flight; a slot holds every table of the
//              model (cm / ht / a16 per component), the HCOMP H and M arrays,
//              the PCOMP H and M arrays and the PCOMP program buffer.
//   out      : plaintext, block b at out_off[b].
#pragma once
#include <stdint.h>

enum ZhCompType : uint8_t {  // LibZPAQ.cs:51-63
  ZH_NONE = 0, ZH_CONS = 1, ZH_CM = 2, ZH_ICM = 3, ZH_MATCH = 4, ZH_AVG = 5,
  ZH_MIX2 = 6, ZH_MIX = 7, ZH_ISSE = 8, ZH_SSE = 9
};

#define ZH_CODE_PAD 160u          // >= 128 (max short jump) + operand bytes
#define ZH_PCOMP_BUF (65536u + 2u * ZH_CODE_PAD)
#define ZH_MAX_LDS_COMP 32        // component descriptors cached in LDS up to this n

#define ZH_FAM_GENERIC 0u        // zh_generic.hip
#define ZH_FAM_CM1 1u            // zh_cm.hip: n == 1, one CM with >= 9 size bits
#define ZH_FAM_CHAIN 2u          // zh_chain.hip: lane-per-component, n <= 64; +1/+2/+3 = specialised for min/mid/max
#define ZH_FAM_STORE 6u         // zh_store.hip: n == 0 (stored bytes, the post-processor is the whole work)
#define ZH_NFAM 7u
#define ZH_HK_GENERIC 0u         // interpret HCOMP
#define ZH_HK_SHIFT 1u           // HCOMP == "a<<= K  *d=a  halt" with D == 0: H[0] = c << K

// Per-segment status codes written by the kernels == zpaqhip_status values.
#define ZH_OK 0
#define ZH_E_CORRUPT (-1)
#define ZH_E_EOF (-2)
#define ZH_E_EOS (-3)
#define ZH_E_ZPAQL (-4)
#define ZH_E_PP_EOS (-8)
#define ZH_E_PP_TYPE (-9)
#define ZH_E_PP_EMPTY (-10)
#define ZH_E_OUTPUT_FULL (-20)
#define ZH_E_BUDGET (-26)
#define ZH_E_SKIPPED (-100)       // an earlier segment of the block failed
#define ZH_E_STOPPED (-101)       // decode ended on request (ZH_LAUNCH_PP_ONLY); never leaves the library
#define ZH_E_RETRY (-102)         // zh_store.hip hands the block to zh_generic.hip; never leaves the library
#define ZH_LAUNCH_PP_ONLY 1u

struct ZhComp {            // one component of a model (Component.cs:18-57 + header args)
  uint8_t type;            // ZhCompType
  uint8_t arg[5];          // header bytes cp[1..5]
  uint8_t level;           // dependency depth: 0 = no prediction inputs, else 1 + max(level of inputs)
  uint8_t small_unit;      // zh_chain: offset of the ICM/ISSE state table in the LDS pool, in 256-word units
  uint32_t cm_mask;        // (#elements of cm / a16) - 1 where the reference indexes masked
  uint32_t ht_mask;        // (#bytes of ht) - 1
  uint64_t cm_off;         // byte offset of cm (u32[]) or a16 (u16[]) in the arena slot
  uint64_t ht_off;         // byte offset of ht (u8[])
  uint64_t cm_bytes;
  uint64_t ht_bytes;
};

struct ZhModel {
  uint32_t n;              // components
  uint8_t hh, hm, ph, pm;
  uint32_t code_off;       // offset of the padded HCOMP window in the code blob
  uint32_t hcomp_len;      // program bytes incl. trailing 0
  uint32_t kind;           // host-chosen specialisation: bits 0-7 kernel family (ZH_FAM_*),
                           // bits 8-15 HCOMP form (ZH_HK_*), bits 16-23 its parameter
  uint32_t depth;          // max component level
  uint64_t h_off, m_off;   // HCOMP H (u32[1<<hh]) and M (u8[1<<hm])
  uint64_t ph_off, pm_off; // PCOMP H and M
  uint64_t pz_off;         // PCOMP program buffer, ZH_PCOMP_BUF bytes
  uint64_t arena_bytes;    // slot size (multiple of 256)
  ZhComp comp[255];
};

struct ZhBlockDesc {
  uint32_t model;          // index into models
  uint32_t first_seg, n_seg;
  uint32_t pad;
  uint64_t out_off, out_cap;
};

struct ZhSegDesc {
  uint64_t in_off, in_len; // coded bytes of the segment within the stream
};

struct ZhSegResult {       // == zpaqhip_seg_result
  int32_t status;
  uint32_t pp_state;
  uint64_t out_off, out_len;
  uint64_t in_used;        // coded bytes consumed by the decoder
};

struct ZhTables {          // Predictor.cs:48-79 + StateTable.cs
  uint16_t squash[4096];
  int16_t stretch[32768];
  int32_t dt[1024];
  int32_t dt2k[256];
  uint8_t ns[1024];
};

// Follows ZhTables in the same device buffer.  ps[p + 2048] = (squash(p) * 2 + 1) << 16: the arithmetic decoder's split
// factor for prediction p (Decoder.cs:136-140), ready for s_mul_hi_u32; read through the scalar cache by kernels whose
// critical path is one wave (a dependent s_load costs ~80 cycles there, ds_read + v_readlane ~120: tools/ubench/lat_bench).
struct ZhTablesX {
  uint32_t ps[4096];
};

struct ZhLaunch {          // kernel arguments (one struct, passed by value)
  const uint8_t *in;
  const ZhModel *models;
  const uint8_t *code;
  const ZhBlockDesc *blocks;
  const ZhSegDesc *segs;
  ZhSegResult *results;
  uint8_t *out;
  uint8_t *arena;
  uint64_t arena_stride;
  const ZhTables *tables;
  uint32_t *queue;         // work-queue head (device-scope atomic)
  uint32_t n_blocks;
  uint32_t flags;          // ZH_LAUNCH_PP_ONLY: stop a block once its post-processor header is complete (generic kernel)
  uint64_t budget;         // ZPAQL instructions per run()
  uint64_t in_total;       // length of the whole stream at `in`
  uint64_t *debug;         // diagnostic builds only (cycle sums); NULL otherwise
};
