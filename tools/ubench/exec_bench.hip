// Does a vector instruction of a wave64 get cheaper when most of its lanes are switched off?
// The chain kernels use 8 (mid) to 22 (max) lanes of the decoder wave: if the SIMD skipped the 16-lane passes of an
// instruction whose EXEC bits are all zero there, running them under a narrow EXEC would shorten every VALU instruction.
//   * dependent and independent v_add / v_mad_i32_i24 / v_mul_lo_u32 / DPP chains under EXEC = 64, 32, 16, 1 lanes
// Run on the GPU box: gpurun -- tools/ubench/exec_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define STAMP(v) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")

#define CASE(body, ...)                                                  \
  do {                                                                   \
    STAMP(t0);                                                           \
    asm volatile(body : __VA_ARGS__);                                    \
    STAMP(t1);                                                           \
    if (threadIdx.x == 0) out[n] = t1 - t0;                              \
    ++n;                                                                 \
  } while (0)

__global__ void k(uint64_t *out, uint64_t mask, uint32_t seed) {
  uint64_t t0, t1;
  uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 5u, d = a + 9u;
  const uint32_t kk = seed | 3u;
  int n = 0;
  uint64_t saved;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1" : "=s"(saved) : "s"(mask));
  CASE(REP64("v_add_u32 %0, %0, %1\n\t"), "+v"(a) : "v"(kk));
  CASE(REP16("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t"), "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(kk));
  CASE(REP64("v_mad_i32_i24 %0, %0, %1, %1\n\t"), "+v"(a) : "v"(kk));
  CASE(REP64("v_mul_lo_u32 %0, %0, %1\n\t"), "+v"(a) : "v"(kk));
  CASE(REP16("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4\n\t"), "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(kk));
  CASE(REP16("s_nop 1\n\tv_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mad_i32_i24 %0, %1, %2, %2\n\tv_ashrrev_i32 %0, 3, %0\n\tv_med3_i32 %0, %0, %2, %3\n\t"),
       "+v"(a), "+v"(b) : "v"(kk), "v"(d));
  CASE(REP16("v_readlane_b32 s20, %0, 3\n\ts_add_u32 s20, s20, 1\n\tv_add_u32 %0, %0, s20\n\t"), "+v"(a) : : "s20", "scc");
  // the same step without a separate DPP move, and without DPP at all (the input of lane t comes through an SGPR)
  CASE(REP64("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"), "+v"(a) :);
  CASE(REP16("s_nop 1\n\tv_mul_i32_i24_dpp %1, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_u32 %0, %1, %2\n\tv_ashrrev_i32 %0, 3, %0\n\tv_med3_i32 %0, %0, %2, %3\n\t"),
       "+v"(a), "+v"(b) : "v"(kk), "v"(d));
  CASE(REP16("s_nop 0\n\tv_readlane_b32 s20, %0, 3\n\tv_mad_i32_i24 %1, s20, %2, %2\n\tv_ashrrev_i32 %1, 3, %1\n\tv_med3_i32 %0, %1, %2, %3\n\tv_cndmask_b32 %4, %4, %0, vcc\n\t"),
       "+v"(a), "+v"(b) : "v"(kk), "v"(d), "v"(c) : "s20");
  CASE(REP16("v_mad_i32_i24 %0, %0, %2, %2\n\tv_ashrrev_i32 %0, 3, %0\n\tv_med3_i32 %0, %0, %2, %3\n\tv_xor_b32 %1, %1, %2\n\t"),
       "+v"(a), "+v"(b) : "v"(kk), "v"(d));
  asm volatile("s_mov_b64 exec, %0" ::"s"(saved));
  if (threadIdx.x == 0) out[15] = a + b + c + d;
}

int main() {
  uint64_t *o;
  hipMalloc(&o, 8 * 16);
  const char *names[] = {"dep v_add x64", "4 indep v_add x16 (64)", "dep v_mad_i32_i24 x64", "dep v_mul_lo_u32 x64", "4 indep v_mul_lo_u32 x16 (64)",
                         "ISSE step (nop,dpp,mad,ashr,med3) x16 (80)", "readlane->s_add->v_add x16 (48)",
                         "dep (nop, v_mov_dpp) x64", "ISSE step (nop,mul_dpp,add,ashr,med3) x16 (80)", "ISSE step (nop0,readlane,mad,ashr,med3,cndmask) x16 (96)",
                         "4 plain VALU x16 (64), no DPP"};
  const struct { uint64_t m; const char *what; } masks[] = {{~0ull, "EXEC = 64 lanes"}, {0xffffffffull, "EXEC = lanes 0-31"}, {0xffffull, "EXEC = lanes 0-15"},
                                                             {~0ull, "EXEC = 64 lanes again"}};
  for (auto &mk : masks) {
    uint64_t r[16];
    for (int rep = 0; rep < 2; ++rep) {                      // second launch: instruction cache warm
      hipMemset(o, 0, 8 * 16);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, mk.m, 12345u);
      hipDeviceSynchronize();
    }
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    printf("%s\n", mk.what);
    for (int i = 0; i < 11; ++i) printf("  %-44s %6llu ticks\n", names[i], (unsigned long long)r[i]);
  }
  return 0;
}
