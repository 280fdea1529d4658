// Checks zh_cm_fast.h's second-nibble selection on the GPU box: 32 bytes per lane from LDS (ds_read_b128 x2), register
// picked in VGPR-index mode, half picked by a shift: lane P must end with (P * 16 + n1) << 16 for every n1.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t *out) {
  __shared__ __attribute__((aligned(64))) uint16_t tab[256];
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t i = lane; i < 256; i += 64) tab[i] = (uint16_t)(i + 1000);
  __syncthreads();
  const uint32_t lb = (uint32_t)(uintptr_t)&tab[0] + (lane & 15) * 32;
  for (uint32_t n1 = 0; n1 < 16; ++n1) {
    uint32_t r;
    const uint32_t s90 = 16 | n1;
    asm volatile(
        "ds_read_b128 v[240:243], %1\n\t"
        "ds_read_b128 v[244:247], %1 offset:16\n\t"
        "s_bfe_u32 s84, %2, 0x30001\n\t"
        "s_bitcmp1_b32 %2, 0\n\t"
        "s_cselect_b32 s80, 0, 16\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_set_gpr_idx_on s84, gpr_idx(SRC0)\n\t"
        "v_mov_b32_e32 v250, v240\n\t"
        "s_set_gpr_idx_off\n\t"
        "v_lshlrev_b32_e32 v250, s80, v250\n\t"
        "v_and_b32_e32 %0, 0xffff0000, v250\n\t"
        : "=v"(r) : "v"(lb), "s"(s90)
        : "memory", "scc", "m0", "s80", "s84", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v250");
    if (lane < 16) out[n1 * 16 + lane] = r >> 16;
  }
}
int main() {
  uint32_t *d, h[256];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int n1 = 0; n1 < 16; ++n1)
    for (int p = 0; p < 16; ++p)
      if (h[n1 * 16 + p] != (uint32_t)(p * 16 + n1 + 1000)) { if (bad++ < 8) printf("n1 %d pos %d: got %u want %u\n", n1, p, h[n1 * 16 + p], p * 16 + n1 + 1000); }
  printf("%d mismatches of 256\n", bad);
  return 0;
}
