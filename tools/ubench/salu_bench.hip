// Micro-benchmark: what does ONE wavefront alone on a CU pay per instruction?
// Build: hipcc --offload-arch=gfx950 -O2 salu_bench.hip -o salu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define STAMP(v) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")

__global__ void k(uint64_t *out, uint32_t seed, int16_t *lds_src) {
  __shared__ uint32_t lds[4096];
  uint32_t lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = (i * 7 + 3) & 4095;
  __syncthreads();
  uint64_t t0, t1;
  uint32_t a = seed, b = seed * 3 + 1;
  int n = 0;
  // 1: dependent s_add chain (64)
  STAMP(t0);
  asm volatile(REP64("s_add_u32 %0, %0, %1\n\t") : "+s"(a) : "s"(b));
  STAMP(t1); out[n++] = t1 - t0;
  // 2: independent s_add (alternate two regs)
  STAMP(t0);
  asm volatile(REP16("s_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, %2\n\ts_xor_b32 s20, %2, %2\n\ts_xor_b32 s21, %2, %2\n\t") : "+s"(a), "+s"(b) : "s"(seed) : "s20", "s21");
  STAMP(t1); out[n++] = t1 - t0;
  // 3: dependent s_mul_hi chain
  STAMP(t0);
  asm volatile(REP64("s_mul_hi_u32 %0, %0, %1\n\t") : "+s"(a) : "s"(b));
  STAMP(t1); out[n++] = t1 - t0;
  // 4: dependent v_add chain
  uint32_t v = lane + seed;
  STAMP(t0);
  asm volatile(REP64("v_add_u32 %0, %0, %1\n\t") : "+v"(v) : "v"(lane));
  STAMP(t1); out[n++] = t1 - t0;
  // 5: v_readlane -> s_add -> v_add using sgpr (VALU<->SALU ping-pong), 16 rounds
  STAMP(t0);
  asm volatile(REP16("v_readlane_b32 %0, %1, 3\n\ts_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, %0\n\t") : "+s"(a), "+v"(v));
  STAMP(t1); out[n++] = t1 - t0;
  // 6: v_readlane with SGPR lane select (dependent): 16 rounds
  uint32_t sel = seed & 7;
  STAMP(t0);
  asm volatile(REP16("s_and_b32 %0, %0, 63\n\ts_nop 3\n\tv_readlane_b32 %0, %1, %0\n\t") : "+s"(sel) : "v"(v));
  STAMP(t1); out[n++] = t1 - t0;
  // 7: dependent LDS read chain (pointer chase) 16
  uint32_t idx = lane & 63;
  STAMP(t0);
  for (int i = 0; i < 16; ++i) idx = lds[idx];
  STAMP(t1); out[n++] = t1 - t0;
  // 8: taken branch chain: 64 unconditional short branches
  STAMP(t0);
  asm volatile(REP64("s_branch 1f\n\ts_nop 0\n\t1:\n\t"));
  STAMP(t1); out[n++] = t1 - t0;
  // 9: not-taken conditional branches 64
  STAMP(t0);
  asm volatile("s_cmp_eq_u32 %0, %0\n\t" REP64("s_cbranch_scc0 1f\n\t1:\n\t") ::"s"(a));
  STAMP(t1); out[n++] = t1 - t0;
  // 10: taken conditional branches 64
  STAMP(t0);
  asm volatile("s_cmp_eq_u32 %0, %0\n\t" REP64("s_cbranch_scc1 1f\n\ts_nop 0\n\t1:\n\t") ::"s"(a));
  STAMP(t1); out[n++] = t1 - t0;
  // 11: back-to-back empty stamps
  STAMP(t0);
  STAMP(t1); out[n++] = t1 - t0;
  // 12: s_cselect/s_cmp dependent mix 16x(cmp, cselect, addc)
  STAMP(t0);
  asm volatile(REP16("s_cmp_le_u32 %0, %1\n\ts_cselect_b32 %0, %1, %0\n\ts_addc_u32 %1, %1, %1\n\ts_xor_b32 %0, %0, %1\n\t") : "+s"(a), "+s"(b));
  STAMP(t1); out[n++] = t1 - t0;
  // 13: global load dependent chain (L2/HBM resident small) 8
  const int16_t *p = lds_src;
  uint32_t gi = lane & 31;
  STAMP(t0);
  for (int i = 0; i < 8; ++i) gi = (uint32_t)p[gi] & 1023;
  STAMP(t1); out[n++] = t1 - t0;
  // 14: s_mul_i32+s_mul_hi+s_lshr_b64 realistic triple, 16 rounds dependent
  STAMP(t0);
  asm volatile(REP16("s_mul_hi_u32 s21, %0, %1\n\ts_mul_i32 s20, %0, %1\n\ts_lshr_b64 s[20:21], s[20:21], 16\n\ts_add_u32 %0, %0, s20\n\t") : "+s"(a) : "s"(b) : "s20", "s21");
  STAMP(t1); out[n++] = t1 - t0;
  out[n++] = a + b + v + idx + gi + sel;
}

int main() {
  uint64_t *d; int16_t *src;
  hipMalloc(&d, 256); hipMalloc(&src, 4096);
  int16_t h[2048]; for (int i = 0; i < 2048; ++i) h[i] = (i * 5 + 1) & 1023;
  hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 12345u, src);
    hipDeviceSynchronize();
  }
  uint64_t o[32]; hipMemcpy(o, d, 256, hipMemcpyDeviceToHost);
  const char *names[] = {"dep s_add x64", "4 indep salu x16 (64)", "dep s_mul_hi x64", "dep v_add x64", "readlane->s_add->v_add x16 (48)",
    "s_and,nop3,readlane(sgpr sel) x16", "dep LDS read x16", "taken s_branch x64", "not-taken cbranch x64", "taken cbranch x64", "empty stamp pair",
    "cmp/cselect/addc/xor x16 (64)", "dep global load x8", "mulhi/mul/lshr64/add x16 (64)"};
  for (int i = 0; i < 14; ++i) printf("%-40s %6llu cycles\n", names[i], (unsigned long long)o[i]);
  return 0;
}
