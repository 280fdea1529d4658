// Dependent-chain latency of the three ways a lone wave can look a table value up: LDS (ds_read + readlane), scalar memory
// (s_load_dword through the scalar cache) and a VGPR-resident table (v_readlane with an SGPR lane index).
// Run on the GPU box: gpurun -- tools/ubench/lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_lds(const uint32_t *tab, uint64_t *out, int iters) {
  __shared__ uint32_t t[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) t[i] = tab[i];
  __syncthreads();
  uint32_t idx = 1;
  uint64_t t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
    uint32_t v = t[(idx + threadIdx.x) & 4095];
    idx = __builtin_amdgcn_readlane(v, 7) & 4095;
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = idx; }
}
__global__ void k_smem(const uint32_t *tab, uint64_t *out, int iters) {
  uint32_t idx = 1;
  uint64_t t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
    uint32_t v;
    uint32_t off = idx * 4;
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(tab), "s"(off) : "memory");
    idx = v & 4095;
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = idx; }
}
__global__ void k_vgpr(const uint32_t *tab, uint64_t *out, int iters) {
  uint32_t mine = tab[threadIdx.x];
  uint32_t idx = 1;
  uint64_t t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
    uint32_t v = __builtin_amdgcn_readlane(mine, idx & 63);
    idx = (v * 5 + 1) & 63;
    asm volatile("" : "+s"(idx));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = idx; }
}

int main() {
  std::vector<uint32_t> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (uint32_t)((i * 1103515245u + 12345u) >> 8);
  uint32_t *d; uint64_t *o;
  hipMalloc(&d, 16384); hipMalloc(&o, 16);
  hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
  const int iters = 100000;
  uint64_t r[2];
  for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, d, o, iters); hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
    printf("lds  ds_read+readlane chain: %.1f ticks/iter\n", (double)r[0] / iters);
    hipLaunchKernelGGL(k_smem, dim3(1), dim3(64), 0, 0, d, o, iters); hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
    printf("smem s_load_dword chain:     %.1f ticks/iter\n", (double)r[0] / iters);
    hipLaunchKernelGGL(k_vgpr, dim3(1), dim3(64), 0, 0, d, o, iters); hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
    printf("vgpr v_readlane(sgpr) chain: %.1f ticks/iter\n", (double)r[0] / iters);
  }
  // s_memtime tick rate: time a fixed busy kernel with events
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k_smem, dim3(1), dim3(64), 0, 0, d, o, iters * 10); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
  printf("s_memtime: %.1f ticks per microsecond\n", (double)r[0] / (ms * 1000.0));
  return 0;
}
