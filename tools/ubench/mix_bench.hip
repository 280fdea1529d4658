// What the instruction KINDS of the chain kernels' bit loop cost a lone wavefront (round 4): the round-3 table
// (salu_bench, exec_bench) prices plain e32 VALU and SALU chains at ~4.7 cycles per instruction, and the mid kernel runs
// at 5.95 per instruction.  This prices what that loop is actually made of: VOP3 forms, selects on SGPR masks, SALU <-> VALU
// alternation, LDS / VMEM issue, waits with nothing outstanding, DPP with fillers.
// Run on the GPU box: gpurun -- tools/ubench/mix_bench
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
  const uint32_t kk = seed | 3u, lo = 5u, hi = 0x7fffffu;
  const uint32_t lbase = (uint32_t)(uintptr_t)lds;
  uint32_t la = lbase + (threadIdx.x & 7) * 8u;
  uint64_t m64 = 0x5555555555555555ull;
  uint32_t s1 = seed, s2 = 1;
  uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
  int n = 0;
  CASE(REP64("v_add_u32 %0, %0, %1\n\t"), "+v"(a) : "v"(kk));                                                        // 0
  CASE(REP64("v_med3_i32 %0, %0, %1, %2\n\t"), "+v"(a) : "v"(lo), "v"(hi));                                            // 1 VOP3 dep
  CASE(REP64("v_cndmask_b32_e64 %0, %0, %1, %2\n\t"), "+v"(a) : "v"(kk), "s"(m64));                                    // 2 select on an SGPR pair
  CASE(REP32("s_add_u32 %1, %1, 1\n\tv_add_u32 %0, %0, %2\n\t"), "+v"(a), "+s"(s1) : "v"(kk) : "scc");                 // 3 alternate, independent
  CASE(REP32("s_add_u32 %1, %1, 1\n\tv_add_u32 %0, %0, %1\n\t"), "+v"(a), "+s"(s1) : : "scc");                         // 4 VALU reads the SGPR the SALU just wrote
  CASE(REP32("v_cmp_lt_u32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]\n\t"), "+v"(a) : "v"(hi), "v"(kk) : "s20", "s21");   // 5 VALU -> SGPR pair -> VALU
  CASE(REP64("v_and_b32 %0, 0x1fffe, %0\n\t"), "+v"(a) :);                                                             // 6 literal
  CASE(REP64("v_bfe_u32 %0, %0, %1, 8\n\t"), "+v"(a) : "s"(s2));                                                       // 7 VOP3 with an SGPR operand
  CASE(REP64("v_lshl_add_u32 %0, %0, 3, %1\n\t"), "+v"(a) : "v"(kk));                                                  // 8
  CASE(REP16("ds_read_b64 v[240:241], %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0xff8, v240\n\tv_add_u32 %0, %0, %1\n\t"), "+v"(la) : "v"(lbase) : "v240", "v241");    // 9 dependent ds_read_b64 (64)
  CASE(REP16("ds_read_b64 v[240:241], %0\n\tds_read_b64 v[242:243], %0 offset:8\n\tds_read_b64 v[244:245], %0 offset:16\n\tds_read_b64 v[246:247], %0 offset:24\n\t") "s_waitcnt lgkmcnt(0)\n\t",
       : "v"(la) : "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247");                                    // 10 64 LDS reads issued back to back
  CASE(REP64("s_nop 0\n\t"), :);                                                                                       // 11
  CASE(REP64("s_nop 1\n\t"), :);                                                                                       // 12
  CASE(REP64("s_waitcnt lgkmcnt(0)\n\t"), :);                                                                          // 13 nothing outstanding
  CASE(REP64("s_waitcnt vmcnt(0)\n\t"), :);                                                                            // 14
  CASE(REP16("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_u32 %1, %1, %3\n\tv_add_u32 %2, %2, %3\n\t"),
       "+v"(a), "+v"(b), "+v"(c) : "v"(kk));                                                                           // 15 DPP + two fillers (48)
  CASE(REP16("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"), "+v"(a) :); // 16 DPP + s_nop 1 (32)
  CASE(REP16("v_readlane_b32 s20, %0, 7\n\tv_mul_i32_i24 %0, s20, %1\n\t"), "+v"(a) : "v"(kk) : "s20");                // 17 readlane -> VALU (32)
  CASE(REP16("v_readlane_b32 s20, %0, 7\n\ts_lshl_b32 s20, s20, 2\n\ts_and_b32 s20, s20, 0xffc\n\ts_load_dword s21, %1, s20\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, s21\n\t"),
       "+v"(a) : "s"(gmem) : "s20", "s21", "scc");                                                                     // 18 readlane -> s_load -> VALU (per iteration)
  CASE(REP16("s_cmp_le_u32 %1, %2\n\ts_cselect_b64 s[20:21], -1, 0\n\tv_cndmask_b32_e64 %0, %0, %3, s[20:21]\n\t"), "+v"(a) : "s"(s1), "s"(s2), "v"(kk) : "s20", "s21", "scc");  // 19
  CASE(REP16("v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mad_i32_i24 %0, %1, %2, %2\n\tv_ashrrev_i32 %0, 16, %0\n\tv_med3_i32 %0, %0, %2, %3\n\tv_add_u32 %4, %4, %2\n\tv_add_u32 %5, %5, %2\n\t"),
       "+v"(a), "+v"(b) : "v"(kk), "v"(hi), "v"(c), "v"(d));                                                           // 20 ISSE step + two fillers (96)
  CASE(REP16("ds_write_b64 %0, v[240:241]\n\t") "s_waitcnt lgkmcnt(0)\n\t", : "v"(la) : "v240", "v241", "memory");       // 21 16 LDS writes issued back to back
  CASE(REP16("buffer_load_dword %0, %1, %2, 0 offen\n\t") "s_waitcnt vmcnt(0)\n\t", "=&v"(w0) : "v"(b & 0xffcu), "s"(__builtin_amdgcn_make_buffer_rsrc((void *)gmem, 0, 16384, 0x00020000)));   // 22 16 loads issued back to back (same line set)
  CASE(REP64("v_mad_i32_i24 %0, %0, %1, %2\n\t"), "+v"(a) : "v"(kk), "s"(s2));                                         // 23 VOP3 + SGPR operand
  CASE(REP64("v_sub_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"), "+v"(a) : "v"(kk));   // 24 SDWA
  if (threadIdx.x == 0) out[31] = a + b + c + d + w0 + w1 + w2 + w3 + (uint32_t)m64 + s1;
}

int main() {
  uint64_t *o;
  uint32_t *g;
  hipMalloc(&o, 8 * 32);
  hipMalloc(&g, 16384);
  hipMemset(g, 0, 16384);
  const char *names[] = {"dep v_add_u32 e32 x64", "dep v_med3_i32 (VOP3) x64", "dep v_cndmask e64, SGPR-pair mask x64", "s_add ; v_add independent x32 (64)",
                         "s_add -> v_add reads it x32 (64)", "v_cmp_e64 -> v_cndmask_e64 x32 (64)", "dep v_and literal x64", "dep v_bfe_u32 SGPR shift x64",
                         "dep v_lshl_add_u32 x64", "ds_read_b64 -> wait -> v_and x16 (48)", "4 ds_read_b64 x16 then wait (64)", "s_nop 0 x64", "s_nop 1 x64",
                         "s_waitcnt lgkmcnt(0), idle x64", "s_waitcnt vmcnt(0), idle x64", "dpp add + 2 fillers x16 (48)", "s_nop 1 + dpp add x16 (32)",
                         "v_readlane -> v_mul x16 (32)", "readlane,2 salu,s_load,wait,v_add x16 (96)", "s_cmp,s_cselect_b64,v_cndmask x16 (48)",
                         "ISSE step + 2 fillers x16 (96)", "16 ds_write_b64 then wait", "16 buffer_load then wait", "dep v_mad_i32_i24 SGPR operand x64", "dep v_sub_sdwa x64"};
  uint64_t r[32];
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(o, 0, 8 * 32);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, 12345u, g);
    hipDeviceSynchronize();
  }
  hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  for (int i = 0; i < 25; ++i) printf("  %-46s %6llu ticks\n", names[i], (unsigned long long)r[i]);
  return 0;
}
