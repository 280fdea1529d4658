// What the first SALU instruction behind a VALU instruction that WRITES an SGPR costs a lone wavefront, by what the SALU
// instruction reads and by how many independent VALU instructions sit between the two (round 4: the L1 step with the next
// bit's probability fetched both ways took the v_readlane's SGPR off every nearby instruction's operands and the ~30 cycles
// behind the v_readlane stayed — profiles/r04/ab_notes.txt, call 14).  Each case: 32 repetitions, cycles by s_memtime.
// Run on the GPU box: gpurun -- tools/ubench/sgprw_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP32(x) REP16(x) REP16(x)
#define STAMP(v) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#define CASE(body, ...)                                                  \
  do {                                                                   \
    STAMP(t0);                                                           \
    asm volatile(body : __VA_ARGS__);                                    \
    STAMP(t1);                                                           \
    if (threadIdx.x == 0) out[n] = t1 - t0;                              \
    ++n;                                                                 \
  } while (0)
#define VA "v_add_u32_e32 %1, %2, %1\n\t"
#define RL "v_readlane_b32 s20, %0, 3\n\t"
#define CM "v_cmp_eq_u32_e64 s[22:23], %0, %2\n\t"
#define CV "v_cmp_eq_u32_e32 vcc, %0, %2\n\t"
#define MV "v_mov_b32_e32 %3, %0\n\t"
#define SI "s_add_u32 %4, %4, 1\n\t"              /* reads nothing the VALU wrote */
#define SD "s_add_u32 %4, %4, s20\n\t"            /* reads the v_readlane's SGPR */
#define OPS "+v"(a), "+v"(b), "+v"(kk), "+v"(c), "+s"(s1) : : "s20", "s22", "s23", "vcc", "scc"

__global__ void k(uint64_t *out, uint32_t seed) {
  uint64_t t0, t1;
  uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 5u, kk = seed | 3u;
  uint32_t s1 = seed | 5u;
  int n = 0;
  CASE(REP32(MV SI), OPS);                        // 0 plain VALU, then SALU (64 instr)
  CASE(REP32(RL SI), OPS);                        // 1 v_readlane, independent SALU
  CASE(REP32(RL SD), OPS);                        // 2 v_readlane, SALU reads its SGPR
  CASE(REP32(RL VA SI), OPS);                     // 3 ... one VALU between (96)
  CASE(REP32(RL VA VA SI), OPS);                  // 4 two (128)
  CASE(REP32(RL VA VA VA VA SI), OPS);            // 5 four (192)
  CASE(REP32(RL VA VA VA VA VA VA VA VA SI), OPS);   // 6 eight (320)
  CASE(REP32(MV VA VA VA VA SI), OPS);            // 7 the same without an SGPR write (192)
  CASE(REP32(MV VA VA VA VA VA VA VA VA SI), OPS);   // 8 (320)
  CASE(REP32(CM SI), OPS);                        // 9 v_cmp into an SGPR pair, independent SALU
  CASE(REP32(CM VA VA SI), OPS);                  // 10
  CASE(REP32(CM VA VA VA VA SI), OPS);            // 11
  CASE(REP32(CV SI), OPS);                        // 12 v_cmp into vcc, independent SALU
  CASE(REP32(CV VA VA VA VA SI), OPS);            // 13
  CASE(REP32(RL RL SI), OPS);                     // 14 two v_readlane, one SALU (96)
  CASE(REP32(RL SI SI SI SI), OPS);               // 15 v_readlane, four SALU (160)
  CASE(REP32(MV SI SI SI SI), OPS);               // 16 plain VALU, four SALU (160)
  CASE(REP32(RL VA VA VA VA SD), OPS);            // 17 four between, SALU reads the SGPR (192)
  if (threadIdx.x == 0) out[31] = a + b + c + s1 + kk;
}

int main() {
  uint64_t *o;
  hipMalloc(&o, 8 * 32);
  const char *names[] = {"v_mov, s_add (64)", "v_readlane, s_add indep (64)", "v_readlane, s_add reads it (64)", "v_readlane, 1 valu, s_add indep (96)",
                         "v_readlane, 2 valu, s_add indep (128)", "v_readlane, 4 valu, s_add indep (192)", "v_readlane, 8 valu, s_add indep (320)",
                         "v_mov, 4 valu, s_add (192)", "v_mov, 8 valu, s_add (320)", "v_cmp_e64 sgpr pair, s_add indep (64)", "v_cmp_e64, 2 valu, s_add (128)",
                         "v_cmp_e64, 4 valu, s_add (192)", "v_cmp vcc, s_add indep (64)", "v_cmp vcc, 4 valu, s_add (192)", "2 v_readlane, s_add (96)",
                         "v_readlane, 4 s_add (160)", "v_mov, 4 s_add (160)", "v_readlane, 4 valu, s_add reads it (192)"};
  const int instr[] = {64, 64, 64, 96, 128, 192, 320, 192, 320, 64, 128, 192, 64, 192, 96, 160, 160, 192};
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, 7u + rep);
    hipDeviceSynchronize();
  }
  uint64_t h[32];
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 18; ++i)
    printf("  %-44s %6llu ticks  %5.1f per instr  %6.1f per repetition\n", names[i], (unsigned long long)h[i], (double)h[i] / instr[i], (double)h[i] / 32);
  return 0;
}
