// zh_host.h — internal host-side declarations of libzpaqhip.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/zpaqhip.h"
#include "zh_model.h"

// Kernel families known to the host only (the kernels never look at a family number): zh_model.h's, and
#define ZH_FAM_CHAIN_MID8 7u     // zh_nibble.hip: mid's shape with EIGHT mixer inputs (icm, five isse, match, icm; mix) — the level-4 text model
#define ZH_FAM_CHAIN_MIN1 8u     // zh_nibble.hip: ONE ICM on min's loop — level 4's model for barely compressible data
#define ZH_NFAM_HOST 9u

namespace zh {

// zh_tables.cpp — model-independent tables, generated and pinned against the
// reference's self-check constants (Predictor.cs:71-77) and the state-table CRC.
const ZhTables &host_tables();           // aborts the call chain via ok=false if pins fail
const ZhTablesX &host_tables_x();        // derived (zh_model.h); uploaded right behind ZhTables
bool host_tables_ok();

// zh_framing.cpp
struct ScanOut {
  std::vector<zpaqhip_block> blocks;
  std::vector<zpaqhip_segment> segs;
  size_t resume_off = 0;                  // see scan_stream
  bool hit_eof = false;
  bool stopped = false;                   // the scan ended on its ScanLimit, not at the end of the buffer
};
// Stop after a block once `max_blocks` are found, or `min_blocks` are found and they span `min_bytes` (batching of the
// streaming forms).  The default never stops early.
struct ScanLimit {
  size_t min_blocks = SIZE_MAX, min_bytes = 0, max_blocks = SIZE_MAX;
  size_t min_blocks_other = 0;            // != 0: the minimum for a batch that holds any block with other than one component
};
int scan_stream(const uint8_t *in, size_t n, ScanOut &out, zpaqhip_err *err, const ScanLimit &lim = ScanLimit());
// Parses a stream-form header (hsize[2] hh hm ph pm n COMP 0 HCOMP 0) into a
// ZhModel + padded code window.  Mirrors ZPAQL.read (ZPAQL.cs:112-156) and the
// limit checks of Predictor.init (Predictor.cs:94-167).
int build_model(const uint8_t *hdr, size_t len, ZhModel &m, std::vector<uint8_t> &code, zpaqhip_err *err);

const char *status_message(int code);
void set_err(zpaqhip_err *err, int code, int block, int seg, const char *msg = nullptr);

// zh_sha1.cpp
void sha1(const uint8_t *p, size_t n, uint8_t out[20]);

}  // namespace zh
