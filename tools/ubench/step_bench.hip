// Micro-benchmark: cost of the arithmetic-decoder bit step of zh_cm_fast.h on one wavefront,
// alone on its CU and with a second wavefront of the same workgroup polling LDS (as wave B does).
// Build: hipcc --offload-arch=gfx950 -O2 step_bench.hip -o step_bench ; run: ./step_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define STAMP(v) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")

#define STEP                                   \
  "v_readlane_b32 s94, %[pv], %[j]\n\t"        \
  "s_sub_u32 s84, %[high], %[low]\n\t"         \
  "s_sub_u32 s85, %[curr], %[low]\n\t"         \
  "s_mul_hi_u32 s86, s84, s94\n\t"             \
  "s_add_u32 s87, %[low], s86\n\t"             \
  "s_add_u32 s88, s87, 1\n\t"                  \
  "s_cmp_le_u32 s85, s86\n\t"                  \
  "s_cselect_b32 %[high], s87, %[high]\n\t"    \
  "s_cselect_b32 %[low], %[low], s88\n\t"      \
  "s_addc_u32 %[j], %[j], %[j]\n\t"            \
  "s_and_b32 %[j], %[j], 63\n\t"               \
  "s_xor_b32 s84, %[high], %[low]\n\t"         \
  "s_cmp_lt_u32 s84, 0x100\n\t"                \
  "s_cbranch_scc1 9f\n\t"

// same arithmetic, range kept as (low, range) so that the multiply does not wait for a subtract
#define STEP2                                  \
  "v_readlane_b32 s94, %[pv], %[j]\n\t"        \
  "s_mul_hi_u32 s86, %[high], s94\n\t"         \
  "s_sub_u32 s85, %[curr], %[low]\n\t"         \
  "s_not_b32 s87, s86\n\t"                     \
  "s_add_u32 s88, %[high], s87\n\t"            \
  "s_cmp_le_u32 s85, s86\n\t"                  \
  "s_cselect_b32 %[high], s86, s88\n\t"        \
  "s_cselect_b32 s87, 0, s87\n\t"              \
  "s_addc_u32 %[j], %[j], %[j]\n\t"            \
  "s_and_b32 %[j], %[j], 63\n\t"               \
  "s_sub_u32 %[low], %[low], s87\n\t"          \
  "s_cmp_lt_u32 %[high], 0x100\n\t"            \
  "s_cbranch_scc1 9f\n\t"


// ---- one nibble (4 steps): probability of node j fetched by a dependent v_readlane per step ...
#define CORE(P)                                \
  "s_sub_u32 s84, %[high], %[low]\n\t"         \
  "s_sub_u32 s85, %[curr], %[low]\n\t"         \
  "s_mul_hi_u32 s86, s84, " P "\n\t"           \
  "s_add_u32 s87, %[low], s86\n\t"             \
  "s_add_u32 s88, s87, 1\n\t"
#define TAIL                                   \
  "s_addc_u32 %[j], %[j], %[j]\n\t"            \
  "s_xor_b32 s84, %[high], %[low]\n\t"         \
  "s_cmp_lt_u32 s84, 0x100\n\t"                \
  "s_cbranch_scc1 9f\n\t"
#define SPLIT "s_cmp_le_u32 s85, s86\n\ts_cselect_b32 %[high], s87, %[high]\n\ts_cselect_b32 %[low], %[low], s88\n\t"
#define NIB_A                                  \
  "s_mov_b32 %[j], 1\n\t"                      \
  "v_readlane_b32 s94, %[pv], %[j]\n\t" CORE("s94") SPLIT TAIL \
  "v_readlane_b32 s94, %[pv], %[j]\n\t" CORE("s94") SPLIT TAIL \
  "v_readlane_b32 s94, %[pv], %[j]\n\t" CORE("s94") SPLIT TAIL \
  "v_readlane_b32 s94, %[pv], %[j]\n\t" CORE("s94") SPLIT TAIL
// ... versus lane j holding both children of node j: the fetch for level k+1 is issued one level early
#define SEL "s_lshl_b32 s96, s95, 16\n\ts_and_b32 s95, s95, 0xffff0000\n\t" SPLIT "s_cselect_b32 s94, s95, s96\n\t"
#define NIB_B                                  \
  "s_mov_b32 %[j], 1\n\t"                      \
  "v_readlane_b32 s94, %[pv], 0\n\t"           \
  "v_readlane_b32 s95, %[pv], 1\n\t"           \
  "s_and_b32 s94, s94, 0xffff0000\n\t"         \
  CORE("s94") SEL TAIL                         \
  "v_readlane_b32 s95, %[pv], %[j]\n\t" CORE("s94") SEL TAIL \
  "v_readlane_b32 s95, %[pv], %[j]\n\t" CORE("s94") SEL TAIL \
  CORE("s94") SPLIT TAIL

// ... versus the same per-step fetch issued as soon as the node index is known (j is updated right after the
// compare; the bit is then recovered from j for the two selects), alternating two probability registers
#define STEP_C(PCUR, PNEXT)                    \
  "s_sub_u32 s84, %[high], %[low]\n\t"         \
  "s_sub_u32 s85, %[curr], %[low]\n\t"         \
  "s_mul_hi_u32 s86, s84, " PCUR "\n\t"        \
  "s_add_u32 s87, %[low], s86\n\t"             \
  "s_add_u32 s88, s87, 1\n\t"                  \
  "s_cmp_le_u32 s85, s86\n\t"                  \
  "s_addc_u32 %[j], %[j], %[j]\n\t"            \
  "v_readlane_b32 " PNEXT ", %[pv], %[j]\n\t"  \
  "s_bitcmp1_b32 %[j], 0\n\t"                  \
  "s_cselect_b32 %[high], s87, %[high]\n\t"    \
  "s_cselect_b32 %[low], %[low], s88\n\t"      \
  "s_xor_b32 s84, %[high], %[low]\n\t"         \
  "s_cmp_lt_u32 s84, 0x100\n\t"                \
  "s_cbranch_scc1 9f\n\t"
#define NIB_C                                  \
  "s_mov_b32 %[j], 1\n\t"                      \
  "v_readlane_b32 s94, %[pv], %[j]\n\t"        \
  STEP_C("s94", "s95") STEP_C("s95", "s94") STEP_C("s94", "s95") STEP_C("s95", "s94")


// ... versus selecting the lane through exec: s_lshl_b64 exec, 1, j ; v_readfirstlane (no SGPR lane select)
#define NIB_D                                  \
  "s_mov_b32 %[j], 1\n\t"                      \
  "s_lshl_b64 exec, 1, %[j]\n\tv_readfirstlane_b32 s94, %[pv]\n\t" CORE("s94") SPLIT TAIL \
  "s_lshl_b64 exec, 1, %[j]\n\tv_readfirstlane_b32 s94, %[pv]\n\t" CORE("s94") SPLIT TAIL \
  "s_lshl_b64 exec, 1, %[j]\n\tv_readfirstlane_b32 s94, %[pv]\n\t" CORE("s94") SPLIT TAIL \
  "s_lshl_b64 exec, 1, %[j]\n\tv_readfirstlane_b32 s94, %[pv]\n\t" CORE("s94") SPLIT TAIL \
  "s_mov_b64 exec, -1\n\t"
// ... versus all 15 nodes of the nibble read into SGPRs with constant lane selects, narrowed by s_cselect as bits arrive
// s[60:61]=n2,n3  s[62:65]=n4..n7  s[66:73]=n8..n15 ; after bit1: s[62:63] pair, s[66:69] quad; after bit2: s62, s[66:67]; after bit3: s66
#define RL(S, L) "v_readlane_b32 " S ", %[pv], " L "\n\t"
#define NIB_E                                  \
  "s_mov_b32 %[j], 1\n\t"                      \
  RL("s94","1") RL("s60","2") RL("s61","3") RL("s62","4") RL("s63","5") RL("s64","6") RL("s65","7") \
  RL("s66","8") RL("s67","9") RL("s68","10") RL("s69","11") RL("s70","12") RL("s71","13") RL("s72","14") RL("s73","15") \
  CORE("s94") SPLIT                            \
  "s_cselect_b32 s94, s61, s60\n\ts_cselect_b64 s[62:63], s[64:65], s[62:63]\n\ts_cselect_b64 s[66:67], s[70:71], s[66:67]\n\ts_cselect_b64 s[68:69], s[72:73], s[68:69]\n\t" TAIL \
  CORE("s94") SPLIT                            \
  "s_cselect_b32 s94, s63, s62\n\ts_cselect_b64 s[66:67], s[68:69], s[66:67]\n\t" TAIL \
  CORE("s94") SPLIT                            \
  "s_cselect_b32 s94, s67, s66\n\t" TAIL       \
  CORE("s94") SPLIT TAIL

__global__ void k(uint64_t *out, uint32_t seed, int spin) {
  __shared__ uint32_t flag[4];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) flag[0] = 0;
  __syncthreads();
  if (wave == 1) {
    if (!spin) return;
    for (uint32_t i = 0; i < (1u << 22); ++i) {
      if (__hip_atomic_load(&flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
      if (spin == 2) __builtin_amdgcn_s_sleep(2);
    }
    return;
  }
  uint64_t t0, t1;
  uint32_t low = 1, high = 0xFFFFFFF0u, curr = seed * 2654435761u, j = 1;
  uint32_t pv = (lane * 2654435761u) | 0x40000000u;
  int n = 0;
  for (int rep = 0; rep < 2; ++rep) {
    STAMP(t0);
    asm volatile(REP16(STEP) "9:\n\t" : [low] "+s"(low), [high] "+s"(high), [curr] "+s"(curr), [j] "+s"(j) : [pv] "v"(pv)
                 : "scc", "s84", "s85", "s86", "s87", "s88", "s94");
    STAMP(t1); out[n++] = t1 - t0;
  }
  low = 1; high = 0xFFFFFFF0u;
  for (int rep = 0; rep < 2; ++rep) {
    STAMP(t0);
    asm volatile(REP16(STEP2) "9:\n\t" : [low] "+s"(low), [high] "+s"(high), [curr] "+s"(curr), [j] "+s"(j) : [pv] "v"(pv)
                 : "scc", "s84", "s85", "s86", "s87", "s88", "s94");
    STAMP(t1); out[n++] = t1 - t0;
  }
  low = 1; high = 0xFFFFFFF0u;
  for (int rep = 0; rep < 2; ++rep) {
    STAMP(t0);
    asm volatile(REP4(NIB_A) "9:\n\t" : [low] "+s"(low), [high] "+s"(high), [curr] "+s"(curr), [j] "+s"(j) : [pv] "v"(pv)
                 : "scc", "s84", "s85", "s86", "s87", "s88", "s94", "s95", "s96");
    STAMP(t1); out[n++] = t1 - t0;
  }
  low = 1; high = 0xFFFFFFF0u;
  for (int rep = 0; rep < 2; ++rep) {
    STAMP(t0);
    asm volatile(REP4(NIB_B) "9:\n\t" : [low] "+s"(low), [high] "+s"(high), [curr] "+s"(curr), [j] "+s"(j) : [pv] "v"(pv)
                 : "scc", "s84", "s85", "s86", "s87", "s88", "s94", "s95", "s96");
    STAMP(t1); out[n++] = t1 - t0;
  }
  low = 1; high = 0xFFFFFFF0u;
  for (int rep = 0; rep < 2; ++rep) {
    STAMP(t0);
    asm volatile(REP4(NIB_C) "9:\n\t" : [low] "+s"(low), [high] "+s"(high), [curr] "+s"(curr), [j] "+s"(j) : [pv] "v"(pv)
                 : "scc", "s84", "s85", "s86", "s87", "s88", "s94", "s95", "s96");
    STAMP(t1); out[n++] = t1 - t0;
  }
  low = 1; high = 0xFFFFFFF0u;
  for (int rep = 0; rep < 2; ++rep) {
    STAMP(t0);
    asm volatile(REP4(NIB_D) "9:\n\ts_mov_b64 exec, -1\n\t" : [low] "+s"(low), [high] "+s"(high), [curr] "+s"(curr), [j] "+s"(j) : [pv] "v"(pv)
                 : "scc", "s84", "s85", "s86", "s87", "s88", "s94", "s95", "s96");
    STAMP(t1); out[n++] = t1 - t0;
  }
  low = 1; high = 0xFFFFFFF0u;
  for (int rep = 0; rep < 2; ++rep) {
    STAMP(t0);
    asm volatile(REP4(NIB_E) "9:\n\t" : [low] "+s"(low), [high] "+s"(high), [curr] "+s"(curr), [j] "+s"(j) : [pv] "v"(pv)
                 : "scc", "s84", "s85", "s86", "s87", "s88", "s94", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73");
    STAMP(t1); out[n++] = t1 - t0;
  }
  // pieces
  STAMP(t0);
  asm volatile(REP16("v_readlane_b32 s94, %[pv], %[j]\n\ts_add_u32 %[j], s94, 1\n\ts_and_b32 %[j], %[j], 63\n\t") : [j] "+s"(j) : [pv] "v"(pv) : "scc", "s94");
  STAMP(t1); out[n++] = t1 - t0;       // readlane(sgpr sel) -> salu -> salu, x16
  STAMP(t0);
  asm volatile(REP16("s_mul_hi_u32 %[j], %[j], %[h]\n\ts_add_u32 %[j], %[j], 77\n\t") : [j] "+s"(j) : [h] "s"(high) : "scc");
  STAMP(t1); out[n++] = t1 - t0;       // mul_hi -> add x16
  STAMP(t0);
  asm volatile(REP16("s_cmp_le_u32 %[j], %[h]\n\ts_cselect_b32 %[j], %[h], %[j]\n\t") : [j] "+s"(j) : [h] "s"(high) : "scc");
  STAMP(t1); out[n++] = t1 - t0;       // cmp -> cselect x16
  {  // VALU -> SGPR -> VALU: v_readlane feeding a vector instruction through its scalar operand (no SALU in between)
    uint32_t vv = pv;
    STAMP(t0);
    asm volatile(REP16("v_readlane_b32 s94, %[v], 7\n\tv_add_u32_e32 %[v], s94, %[v]\n\tv_xor_b32_e32 %[v], 0x55, %[v]\n\t") : [v] "+v"(vv) : : "s94");
    STAMP(t1); out[n++] = t1 - t0;     // readlane -> valu(sgpr operand) -> valu, x16
    STAMP(t0);
    asm volatile(REP16("v_readfirstlane_b32 s94, %[v]\n\ts_add_u32 s94, s94, 3\n\tv_add_u32_e32 %[v], s94, %[v]\n\t") : [v] "+v"(vv) : : "s94", "scc");
    STAMP(t1); out[n++] = t1 - t0;     // readfirstlane -> salu -> valu(sgpr operand), x16
    STAMP(t0);
    asm volatile(REP16("v_add_u32_e32 %[v], 3, %[v]\n\tv_xor_b32_e32 %[v], 0x55, %[v]\n\tv_mul_u32_u24_e32 %[v], 3, %[v]\n\t") : [v] "+v"(vv));
    STAMP(t1); out[n++] = t1 - t0;     // three dependent valu, x16
    STAMP(t0);
    asm volatile(REP16("ds_bpermute_b32 %[v], %[v], %[v]\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32_e32 %[v], 0xfc, %[v]\n\t") : [v] "+v"(vv) : : "memory");
    STAMP(t1); out[n++] = t1 - t0;     // ds_bpermute -> wait -> valu, x16
    curr += vv;
  }
  STAMP(t0);
  STAMP(t1); out[n++] = t1 - t0;
  out[n++] = low + high + curr + j;
  if (lane == 0) __hip_atomic_store(&flag[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

int main() {
  uint64_t *d;
  hipMalloc(&d, 256);
  const char *names[] = {"STEP x16 (cold)", "STEP x16", "STEP2 x16 (cold)", "STEP2 x16", "4 nibbles, readlane per step (cold)", "4 nibbles, readlane per step", "4 nibbles, children pairs (cold)", "4 nibbles, children pairs", "4 nibbles, early fetch (cold)", "4 nibbles, early fetch", "4 nibbles, exec + readfirstlane (cold)", "4 nibbles, exec + readfirstlane", "4 nibbles, 15 nodes in SGPRs (cold)", "4 nibbles, 15 nodes in SGPRs", "readlane->2 salu x16 (48)", "mul_hi->add x16 (32)",
                         "cmp->cselect x16 (32)", "readlane->valu(sgpr)->valu x16 (48)", "readfirstlane->salu->valu x16 (48)", "3 dependent valu x16 (48)", "ds_bpermute->wait->valu x16", "empty stamp pair"};
  for (int spin = 0; spin < 2; ++spin) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, 12345u, spin);
      hipDeviceSynchronize();
    }
    uint64_t o[32];
    hipMemcpy(o, d, 256, hipMemcpyDeviceToHost);
    printf("second wave: %s\n", spin == 0 ? "exits at once" : spin == 1 ? "polls LDS flat out" : "polls LDS with s_sleep 2");
    for (int i = 0; i < 22; ++i) printf("  %-32s %6llu cycles\n", names[i], (unsigned long long)o[i]);
  }
  return 0;
}
