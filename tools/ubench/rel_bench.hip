// Checks the two relative-addressing forms zh_cm_fast.h relies on, on the GPU box: s_movrels_b32 (SGPR[base + M0]) and the
// VGPR index mode (s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off).  Prints what each index returned.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t *out) {
  const uint32_t lane = threadIdx.x & 63;
  uint32_t v = lane * 1000u + 7u;
  for (uint32_t j = 1; j < 16; ++j) {
    uint32_t r;
    asm volatile(
        "v_readlane_b32 s65, %1, 1\n\tv_readlane_b32 s66, %1, 2\n\tv_readlane_b32 s67, %1, 3\n\tv_readlane_b32 s68, %1, 4\n\t"
        "v_readlane_b32 s69, %1, 5\n\tv_readlane_b32 s70, %1, 6\n\tv_readlane_b32 s71, %1, 7\n\tv_readlane_b32 s72, %1, 8\n\t"
        "v_readlane_b32 s73, %1, 9\n\tv_readlane_b32 s74, %1, 10\n\tv_readlane_b32 s75, %1, 11\n\tv_readlane_b32 s76, %1, 12\n\t"
        "v_readlane_b32 s77, %1, 13\n\tv_readlane_b32 s78, %1, 14\n\tv_readlane_b32 s79, %1, 15\n\t"
        "s_mov_b32 m0, %2\n\ts_nop 1\n\ts_movrels_b32 %0, s64\n\t"
        : "=s"(r) : "v"(v), "s"(j)
        : "m0", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79");
    if (lane == 0) out[j] = r;
  }
  for (uint32_t i = 0; i < 8; ++i) {
    uint32_t r;
    asm volatile(
        "v_mov_b32 v240, 100\n\tv_mov_b32 v241, 101\n\tv_mov_b32 v242, 102\n\tv_mov_b32 v243, 103\n\t"
        "v_mov_b32 v244, 104\n\tv_mov_b32 v245, 105\n\tv_mov_b32 v246, 106\n\tv_mov_b32 v247, 107\n\t"
        "s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_mov_b32_e32 %0, v240\n\ts_set_gpr_idx_off\n\t"
        : "=v"(r) : "s"(i)
        : "m0", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247");
    if (lane == 0) out[16 + i] = r;
  }
}
int main() {
  uint32_t *d, h[32] = {0};
  hipMalloc(&d, sizeof h);
  hipMemset(d, 0, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("s_movrels:");
  for (int j = 1; j < 16; ++j) printf(" %u", h[j]);
  printf("\n(expected: j * 1000 + 7)\ngpr idx:");
  for (int i = 0; i < 8; ++i) printf(" %u", h[16 + i]);
  printf("\n(expected: 100 .. 107)\n");
  return 0;
}
