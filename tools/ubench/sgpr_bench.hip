// What an SGPR operand costs a VALU instruction of a lone wavefront (round 4; follow-up of mix_bench: a dependent
// v_mad_i32_i24 with an SGPR addend ran at 9.6 cycles per instruction against 4.7, and a v_add that reads an SGPR the SALU
// has just written at 8.6).  By operand position and encoding, constant and freshly written; VMEM issue cost by the
// number of live lanes; the two ways the chain kernels turn the final prediction into the decoder's split factor.
// Run on the GPU box: gpurun -- tools/ubench/sgpr_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP32(x) REP16(x) REP16(x)
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

__global__ void k(uint64_t *out, uint32_t seed, const uint32_t *gmem) {
  __shared__ uint32_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (i * 8u) & 4095u;
  __syncthreads();
  uint64_t t0, t1;
  uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 5u, d = a + 9u;
  const uint32_t kk = seed | 3u, hi = 0x7fffffu;
  uint32_t s1 = seed | 5u, s2 = 3u;
  const uint32_t lbase = (uint32_t)(uintptr_t)lds;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)gmem, 0, 16384, 0x00020000);
  const uint32_t off_all = (threadIdx.x * 4u) & 0xffcu, off7 = threadIdx.x < 7 ? threadIdx.x * 4u : 0x80000000u;
  uint32_t w0 = 0;
  int n = 0;
  CASE(REP64("v_med3_i32 %0, %0, %1, %2\n\t"), "+v"(a) : "s"(s2), "v"(hi));                                    // 0
  CASE(REP64("v_med3_i32 %0, %0, %1, %2\n\t"), "+v"(a) : "v"(kk), "v"(hi));                                    // 1
  CASE(REP64("v_add_u32_e32 %0, %1, %0\n\t"), "+v"(a) : "s"(s1));                                              // 2
  CASE(REP64("v_sub_u32_e32 %0, %1, %0\n\t"), "+v"(a) : "s"(s1));                                              // 3
  CASE(REP64("v_mad_i32_i24 %0, %0, %1, %2\n\t"), "+v"(a) : "v"(kk), "s"(s2));                                 // 4 SGPR src2
  CASE(REP64("v_mad_i32_i24 %0, %1, %0, %2\n\t"), "+v"(a) : "s"(s2), "v"(kk));                                 // 5 SGPR src0
  CASE(REP64("v_mad_i32_i24 %0, %0, %1, %2\n\t"), "+v"(a) : "v"(kk), "v"(hi));                                 // 6 VGPRs
  CASE(REP64("v_mul_i32_i24_e32 %0, %1, %0\n\t"), "+v"(a) : "s"(s2));                                          // 7
  CASE(REP64("v_lshl_add_u32 %0, %0, 2, %1\n\t"), "+v"(a) : "s"(s2));                                          // 8
  CASE(REP64("v_lshl_add_u32 %0, %0, 2, %1\n\t"), "+v"(a) : "v"(kk));                                          // 9
  CASE(REP64("v_and_b32_e32 %0, %1, %0\n\t"), "+v"(a) : "s"(s1));                                              // 10
  CASE(REP32("s_add_u32 %1, %1, 1\n\tv_add_u32_e32 %0, %1, %0\n\t"), "+v"(a), "+s"(s1) : : "scc");             // 11 fresh, no distance
  CASE(REP32("s_add_u32 %1, %1, 1\n\tv_xor_b32 %2, %2, %3\n\tv_add_u32_e32 %0, %1, %0\n\t"), "+v"(a), "+s"(s1), "+v"(b) : "v"(kk) : "scc");                      // 12 one VALU between (96)
  CASE(REP32("s_add_u32 %1, %1, 1\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4\n\tv_add_u32_e32 %0, %1, %0\n\t"), "+v"(a), "+s"(s1), "+v"(b), "+v"(c) : "v"(kk) : "scc");   // 13 two between (128)  [c is read-write in fact]
  CASE(REP32("s_add_u32 %1, %1, 1\n\ts_nop 0\n\tv_add_u32_e32 %0, %1, %0\n\t"), "+v"(a), "+s"(s1) : : "scc");  // 14 s_nop 0 between (96)
  CASE(REP32("s_add_u32 %1, %1, 1\n\tv_mov_b32_e32 %2, %1\n\tv_add_u32_e32 %0, %2, %0\n\tv_xor_b32 %0, %2, %0\n\t"), "+v"(a), "+s"(s1), "+v"(b) : : "scc");      // 15 v_mov once, two VGPR uses (128)
  CASE(REP64("v_cmp_eq_u32_e64 s[20:21], %1, %0\n\t"), : "v"(a), "s"(s2) : "s20", "s21");                      // 16 compare with an SGPR operand (independent)
  CASE(REP16("buffer_load_dword %0, %1, %2, 0 offen\n\t") "s_waitcnt vmcnt(0)\n\t", "=&v"(w0) : "v"(off_all), "s"(rs));   // 17 64 live lanes, one 256-byte run
  CASE(REP16("buffer_load_dword %0, %1, %2, 0 offen\n\t") "s_waitcnt vmcnt(0)\n\t", "=&v"(w0) : "v"(off7), "s"(rs));      // 18 7 live lanes (the rest out of range)
  CASE(REP16("buffer_store_dword %0, %1, %2, 0 offen\n\t") "s_waitcnt vmcnt(0)\n\t", : "v"(a), "v"(off7), "s"(rs) : "memory");   // 19 16 stores, 7 live lanes
  CASE(REP4("buffer_load_dword %0, %1, %2, 0 offen\n\t") "s_waitcnt vmcnt(0)\n\t", "=&v"(w0) : "v"(off7), "s"(rs));       // 20 4 loads + latency
  CASE("buffer_load_dword %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)\n\t", "=&v"(w0) : "v"(off7), "s"(rs));               // 21 1 load + latency
  // the final prediction -> split factor: (a) v_readlane -> s_load -> scalar consumer; (b) ds_read_u16 -> v_readlane -> scalar consumer
  CASE(REP16("v_lshl_add_u32 %1, %0, 2, %2\n\tv_and_b32 %1, 0xffc, %1\n\ts_nop 0\n\tv_readlane_b32 s20, %1, 7\n\ts_load_dword s21, %3, s20\n\ts_waitcnt lgkmcnt(0)\n\ts_mul_hi_u32 s22, s21, s21\n\ts_and_b32 s22, s22, 0xff\n\tv_add_u32 %0, s22, %0\n\t"),
       "+v"(a), "+v"(b) : "v"(kk), "s"(gmem) : "s20", "s21", "s22", "scc");                                     // 22 (9 instr per iteration)
  CASE(REP16("v_lshl_add_u32 %1, %0, 1, %2\n\tv_and_b32 %1, 0xffe, %1\n\tv_add_u32 %1, %1, %3\n\tds_read_u16 %1, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_readlane_b32 s21, %1, 7\n\ts_mul_hi_u32 s22, s21, s21\n\ts_and_b32 s22, s22, 0xff\n\tv_add_u32 %0, s22, %0\n\t"),
       "+v"(a), "+v"(b) : "v"(kk), "v"(lbase) : "s21", "s22", "scc");                                           // 23 (9 instr per iteration)
  CASE(REP64("v_cndmask_b32_e32 %0, %0, %1, vcc\n\t"), "+v"(a) : "v"(kk) : );                                   // 24 select on vcc, e32
  CASE(REP16("s_nop 0\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_u32 %1, %1, %2\n\t"), "+v"(a), "+v"(b) : "v"(kk));   // 25 s_nop 0 + dpp + 1 filler (48)
  CASE(REP16("v_add_u32 %1, %1, %2\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"), "+v"(a), "+v"(b) : "v"(kk));              // 26 dpp + 1 filler, no nop (hazard: for timing only) (32)
  if (threadIdx.x == 0) out[31] = a + b + c + d + w0 + s1;
}

int main() {
  uint64_t *o;
  uint32_t *g;
  hipMalloc(&o, 8 * 32);
  hipMalloc(&g, 16384);
  hipMemset(g, 0, 16384);
  const char *names[] = {"v_med3 v,v,S,v x64", "v_med3 v,v,v,v x64", "v_add e32 S,v x64", "v_sub e32 S,v x64", "v_mad_i32_i24 v,v,S x64", "v_mad_i32_i24 S,v,v x64",
                         "v_mad_i32_i24 v,v,v x64", "v_mul_i32_i24 e32 S,v x64", "v_lshl_add v,2,S x64", "v_lshl_add v,2,v x64", "v_and e32 S,v x64",
                         "s_add -> v_add reads it x32 (64)", "s_add, 1 valu, v_add reads x32 (96)", "s_add, 2 valu, v_add reads x32 (128)", "s_add, s_nop 0, v_add reads x32 (96)",
                         "s_add, v_mov, 2 vgpr uses x32 (128)", "v_cmp_e64 S,v indep x64", "16 buffer_load 64 lanes + wait", "16 buffer_load 7 lanes + wait",
                         "16 buffer_store 7 lanes + wait", "4 buffer_load 7 lanes + wait", "1 buffer_load 7 lanes + wait", "p -> readlane -> s_load -> salu x16 (144)",
                         "p -> ds_read -> readlane -> salu x16 (144)", "dep v_cndmask e32 vcc x64", "s_nop 0, dpp, 1 filler x16 (48)", "1 filler, dpp (no nop) x16 (32)"};
  uint64_t r[32];
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(o, 0, 8 * 32);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, 12345u, g);
    hipDeviceSynchronize();
  }
  hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  for (int i = 0; i < 27; ++i) printf("  %-46s %6llu ticks\n", names[i], (unsigned long long)r[i]);
  return 0;
}
