// zh_tables.cpp — builds the model-independent tables once on the host and
// checks them against the reference's own pins before they are uploaded:
//   stsum == 3887533746, sqsum == 2278286169   (Predictor.cs:71-77)
//   CRC-32 of the 1024-byte state table == 0x77a1e24c (StateTable.cs:21-149)
// A mismatch makes zpaqhip_ctx_create fail; nothing is ever decoded with
// unpinned tables.
#include <math.h>
#include <string.h>

#include "zh_host.h"

namespace zh {
namespace {

// Bit-history state machine generator (ZPAQ specification).  A state is a pair
// of bounded counts (n0, n1), plus for small totals which bit came last.
struct StateGen {
  static int states_for(int n0, int n1) {
    static const int cap[6] = {20, 48, 15, 8, 6, 5};
    if (n0 < n1) return states_for(n1, n0);
    if (n0 < 0 || n1 < 0 || n1 >= 6 || n0 > cap[n1]) return 0;
    return (n1 > 0 && n0 + n1 <= 17) ? 2 : 1;
  }
  static int fade(int n) {  // count kept of the bit that did NOT occur
    int r = 0;
    for (int t : {1, 2, 3, 4, 5, 7, 8}) r += n >= t;
    return r;
  }
  static void step(int &n0, int &n1, int y) {
    if (n0 < n1) { step(n1, n0, 1 - y); return; }
    if (y) { ++n1; n0 = fade(n0); } else { ++n0; n1 = fade(n1); }
    while (!states_for(n0, n1)) {
      if (n1 < 2) --n0;
      else { n0 = (n0 * (n1 - 1) + n1 / 2) / n1; --n1; }
    }
  }
  static void build(uint8_t ns[1024]) {
    const int N = 50;
    static uint8_t id[N][N][2];
    memset(id, 0, sizeof id);
    int next = 0;
    for (int tot = 0; tot < N; ++tot)
      for (int n1 = 0; n1 <= tot; ++n1) {
        int n0 = tot - n1, k = states_for(n0, n1);
        if (!k) continue;
        id[n0][n1][0] = (uint8_t)next;
        id[n0][n1][1] = (uint8_t)(next + k - 1);
        next += k;
      }
    memset(ns, 0, 1024);
    for (int n0 = 0; n0 < N; ++n0)
      for (int n1 = 0; n1 < N; ++n1)
        for (int y = 0; y < states_for(n0, n1); ++y) {
          uint8_t *row = &ns[id[n0][n1][y] * 4];
          int a = n0, b = n1;
          step(a, b, 0); row[0] = id[a][b][0];
          a = n0; b = n1;
          step(a, b, 1); row[1] = id[a][b][1];
          row[2] = (uint8_t)n0; row[3] = (uint8_t)n1;
        }
  }
};

uint32_t crc32(const uint8_t *p, size_t n) {
  uint32_t c = ~0u;
  for (size_t i = 0; i < n; ++i) {
    c ^= p[i];
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1)));
  }
  return ~c;
}

struct Built {
  ZhTables t;
  ZhTablesX x;
  bool ok;
  Built() {
    t.dt2k[0] = 0;
    for (int i = 1; i < 256; ++i) t.dt2k[i] = 2048 / i;
    for (int i = 0; i < 1024; ++i) t.dt[i] = (1 << 17) / (i * 2 + 3) * 2;
    for (int i = 0; i < 4096; ++i)
      t.squash[i] = i < 1376 ? 0 : i >= 2720 ? 32767 : (uint16_t)(int)(32768.0 / (1 + exp((i - 2048) * (-1.0 / 64))));
    for (int i = 16384; i < 32768; ++i)
      t.stretch[i] = (int16_t)((int)(log((i + 0.5) / (32767.5 - i)) * 64 + 0.5 + 100000) - 100000);
    for (int i = 0; i < 16384; ++i) t.stretch[i] = (int16_t)-t.stretch[32767 - i];
    for (int i = 0; i < 4096; ++i) x.ps[i] = ((uint32_t)t.squash[i] * 2 + 1) << 16;
    StateGen::build(t.ns);
    uint32_t st = 0, sq = 0;
    for (int i = 32767; i >= 0; --i) st = st * 3 + (uint32_t)(int)t.stretch[i];
    for (int i = 4095; i >= 0; --i) sq = sq * 3 + t.squash[i];
    ok = st == 3887533746u && sq == 2278286169u && crc32(t.ns, 1024) == 0x77a1e24cu;
  }
};

const Built &built() {
  static const Built b;
  return b;
}

}  // namespace

const ZhTables &host_tables() { return built().t; }
const ZhTablesX &host_tables_x() { return built().x; }
bool host_tables_ok() { return built().ok; }

}  // namespace zh
