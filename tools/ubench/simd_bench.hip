// Where do the wavefronts of one workgroup land, and what does a busy-polling neighbour cost?
//   * HW_ID of every wave of a 192-thread workgroup (SIMD, CU): do the three waves of zh_chain3's block sit on three SIMDs?
//   * a dependent VALU chain on wave 0 alone, with waves 1-2 polling an LDS word, and with all three computing
//   * the same with a dependent SALU chain
//   * core clock under that load: s_memtime ticks vs wall time (hipEvent)
// Run on the GPU box: gpurun -- tools/ubench/simd_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define DEV __device__ __forceinline__
DEV uint64_t now() { uint64_t t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }

template <int N> DEV int vchain(int v, int k) {
#pragma unroll
  for (int i = 0; i < N; ++i) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(k)); }
  return v;
}
template <int N> DEV uint32_t schain(uint32_t v, uint32_t k) {
#pragma unroll
  for (int i = 0; i < N; ++i) { asm volatile("s_add_u32 %0, %0, %1" : "+s"(v) : "s"(k) : "scc"); }
  return v;
}

// mode: 0 wave 0 computes, others exit; 1 others poll LDS until wave 0 is done; 2 all compute; 3 others poll with s_sleep
template <int MODE, int SCALAR>
__global__ __launch_bounds__(192) void k_busy(uint64_t *out, int iters) {
  __shared__ uint32_t flag;
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (threadIdx.x == 0) flag = 0;
  __syncthreads();
  uint32_t hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (lane == 0) out[8 + blockIdx.x * 4 + w] = hwid;
  if (w == 0 || MODE == 2) {
    int acc = (int)lane; uint32_t sacc = w;
    uint64_t t0 = now();
    for (int i = 0; i < iters; ++i) { if (SCALAR) sacc = schain<64>(sacc, 3u); else acc = vchain<64>(acc, 3); }
    uint64_t t1 = now();
    if (lane == 0 && blockIdx.x == 0) { out[w] = t1 - t0; out[4 + w] = (uint32_t)acc + sacc; }
    if (w == 0) { __hip_atomic_store(&flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
  } else if (MODE == 1 || MODE == 3) {
    uint32_t spins = 0;
    while (__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0 && spins < (1u << 28)) { ++spins; if (MODE == 3) __builtin_amdgcn_s_sleep(2); }
  }
}

int main() {
  uint64_t *o; hipMalloc(&o, 8 * 4200);
  std::vector<uint64_t> r(4200);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(mode, sc, grid, name)                                                                 \
  hipMemset(o, 0, 8 * 4200);                                                                      \
  hipLaunchKernelGGL((k_busy<mode, sc>), dim3(grid), dim3(192), 0, 0, o, 100);                    \
  hipEventRecord(e0); hipLaunchKernelGGL((k_busy<mode, sc>), dim3(grid), dim3(192), 0, 0, o, iters); hipEventRecord(e1); \
  hipEventSynchronize(e1); { float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(r.data(), o, 8 * 4200, hipMemcpyDeviceToHost); \
  printf("%-44s grid %3d: wave0 %.2f ticks/instr (w1 %.2f w2 %.2f), kernel %.3f ms -> %.0f ticks/us\n", name, grid,            \
         (double)r[0] / (64.0 * iters), (double)r[1] / (64.0 * iters), (double)r[2] / (64.0 * iters), ms, (double)r[0] / (ms * 1000.0)); }
  RUN(0, 0, 1, "VALU chain, wave 0 alone")
  printf("HW_ID of the 3 waves of block 0: ");
  for (int w = 0; w < 3; ++w) printf("[wave %d: simd %u cu %u se %u wave_slot %u] ", w, (unsigned)((r[8 + w] >> 4) & 3), (unsigned)((r[8 + w] >> 8) & 15), (unsigned)((r[8 + w] >> 13) & 7), (unsigned)(r[8 + w] & 15));
  printf("\n");
  RUN(1, 0, 1, "VALU chain, waves 1-2 polling LDS")
  RUN(3, 0, 1, "VALU chain, waves 1-2 polling with s_sleep")
  RUN(2, 0, 1, "VALU chain, all three computing")
  RUN(0, 1, 1, "SALU chain, wave 0 alone")
  RUN(1, 1, 1, "SALU chain, waves 1-2 polling LDS")
  RUN(2, 1, 1, "SALU chain, all three computing")
  RUN(2, 0, 256, "VALU chain, all three computing")
  { int same = 0; for (int b = 0; b < 256; ++b) { unsigned s0 = (r[8 + b * 4] >> 4) & 3, s1 = (r[8 + b * 4 + 1] >> 4) & 3, s2 = (r[8 + b * 4 + 2] >> 4) & 3; if (s0 == s1 || s0 == s2 || s1 == s2) ++same; }
    printf("256 workgroups: %d have two waves on one SIMD\n", same); }
  RUN(1, 0, 256, "VALU chain, waves 1-2 polling LDS")
  RUN(2, 1, 256, "SALU chain, all three computing")
  return 0;
}
