// What a hand-over between two wavefronts of one workgroup costs when it goes through LDS, and what a per-bit
// two-stage pipeline (model wave -> decoder wave -> model wave) built on such hand-overs can reach.
// Run on the GPU box: gpurun -- tools/ubench/hop_bench
//   pingpong_simple : lane-0 flag write, scalar poll loop (ds_read + readfirstlane + compare) on the other side
//   pingpong_vec    : 64-lane payload with a tag in every lane (one ds_write_b32), reader tests all lanes (v_cmp + vcc)
//   pingpong_pipe   : as simple, but the poller keeps two reads in flight
//   pingpong_bar    : s_barrier as the signal, the value read from LDS behind it
//   stage_pipe      : wave C: wait y, NC dependent VALU ops, write p[] (tagged);  wave D: wait p[], ND dependent VALU ops,
//                     one dependent s_load, 14 dependent SALU ops, write y.  Both also run `side` independent VALU ops
//                     after their write (the work that overlaps the other wave's stage).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define DEV __device__ __forceinline__

DEV void put0(uint32_t addr, uint32_t val) {
  asm volatile("s_mov_b64 exec, 1\n\tds_write_b32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(addr), "v"(val) : "memory");
}
DEV void wait_eq(uint32_t addr, uint32_t want, uint32_t &g_left) {
  uint32_t got, tmp;
  asm volatile(
      ".Lw_%=:\n\t"
      "ds_read_b32 %[t], %[a]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_readfirstlane_b32 %[g], %[t]\n\t"
      "s_cmp_eq_u32 %[g], %[w]\n\t"
      "s_cbranch_scc1 .Lwd_%=\n\t"
      "s_sub_u32 %[l], %[l], 1\n\t"
      "s_cmp_lg_u32 %[l], 0\n\t"
      "s_cbranch_scc1 .Lw_%=\n"
      ".Lwd_%=:"
      : [t] "=&v"(tmp), [g] "=&s"(got), [l] "+s"(g_left)
      : [a] "v"(addr), [w] "s"(want)
      : "memory", "scc");
}
// two reads in flight: the answer is at most half a round trip old when it is looked at
DEV void wait_eq_pipe(uint32_t addr, uint32_t want, uint32_t &g_left) {
  uint32_t g0, g1, t0, t1;
  asm volatile(
      "ds_read_b32 %[t0], %[a]\n\t"
      ".Lp_%=:\n\t"
      "ds_read_b32 %[t1], %[a]\n\t"
      "s_waitcnt lgkmcnt(1)\n\t"
      "v_readfirstlane_b32 %[g0], %[t0]\n\t"
      "s_cmp_eq_u32 %[g0], %[w]\n\t"
      "s_cbranch_scc1 .Ld_%=\n\t"
      "ds_read_b32 %[t0], %[a]\n\t"
      "s_waitcnt lgkmcnt(1)\n\t"
      "v_readfirstlane_b32 %[g1], %[t1]\n\t"
      "s_cmp_eq_u32 %[g1], %[w]\n\t"
      "s_cbranch_scc1 .Ld_%=\n\t"
      "s_sub_u32 %[l], %[l], 1\n\t"
      "s_cmp_lg_u32 %[l], 0\n\t"
      "s_cbranch_scc1 .Lp_%=\n"
      ".Ld_%=:\n\t"
      "s_waitcnt lgkmcnt(0)"
      : [t0] "=&v"(t0), [t1] "=&v"(t1), [g0] "=&s"(g0), [g1] "=&s"(g1), [l] "+s"(g_left)
      : [a] "v"(addr), [w] "s"(want)
      : "memory", "scc");
}
// all 64 lanes: value >> 12 == want
DEV uint32_t wait_vec(uint32_t addr_lane, uint32_t want, uint32_t &g_left) {
  uint32_t v;
  asm volatile(
      ".Lv_%=:\n\t"
      "ds_read_b32 %[v], %[a]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_lshrrev_b32 %[v], 12, %[v]\n\t"
      "v_cmp_ne_u32 vcc, %[w], %[v]\n\t"
      "s_cbranch_vccz .Lvd_%=\n\t"
      "s_sub_u32 %[l], %[l], 1\n\t"
      "s_cmp_lg_u32 %[l], 0\n\t"
      "s_cbranch_scc1 .Lv_%=\n"
      ".Lvd_%=:"
      : [v] "=&v"(v), [l] "+s"(g_left)
      : [a] "v"(addr_lane), [w] "s"(want)
      : "memory", "vcc", "scc");
  return v;
}
DEV uint64_t now() { uint64_t t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }

template <int MODE>
__global__ __launch_bounds__(128) void k_pingpong(uint64_t *out, int iters) {
  __shared__ uint32_t X, Y, PV[64];
  const uint32_t lane = threadIdx.x & 63, w = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x == 0) { X = 0; Y = 0; }
  if (threadIdx.x < 64) PV[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t ax = (uint32_t)(uintptr_t)&X, ay = (uint32_t)(uintptr_t)&Y, apv = (uint32_t)(uintptr_t)&PV[lane];
  uint32_t left = 1u << 27;                    // every poll draws on one budget: a protocol error ends the kernel
  uint64_t t0 = now();
#define LOOP(body) for (int i = 1; i <= iters; ++i) { const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane(i); body }
  if (MODE == 0) {
    if (w == 0) { LOOP(put0(ax, s); wait_eq(ay, s, left);) } else { LOOP(wait_eq(ax, s, left); put0(ay, s);) }
  } else if (MODE == 1) {
    if (w == 0) { LOOP(asm volatile("ds_write_b32 %0, %1" ::"v"(apv), "v"(s << 12 | lane) : "memory"); wait_eq(ay, s, left);) }
    else { LOOP((void)wait_vec(apv, s, left); put0(ay, s);) }
  } else if (MODE == 2) {
    if (w == 0) { LOOP(put0(ax, s); wait_eq_pipe(ay, s, left);) } else { LOOP(wait_eq_pipe(ax, s, left); put0(ay, s);) }
  } else {
    // barrier as the signal: the writer stores, both meet, the reader loads behind the barrier
    for (int i = 1; i <= iters; ++i) {
      const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane(i);
      if (w == 0) { put0(ax, s); }
      __builtin_amdgcn_s_barrier();
      uint32_t v = 0;
      if (w == 1) { v = *(volatile uint32_t *)&X; put0(ay, v); }
      __builtin_amdgcn_s_barrier();
      if (w == 0) { v = *(volatile uint32_t *)&Y; asm volatile("" ::"v"(v)); }
    }
  }
  uint64_t t1 = now();
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

// dependent VALU chain of n ops
template <int N>
DEV int vchain(int v, int k) {
#pragma unroll
  for (int i = 0; i < N; ++i) { v = v * 3 + k; asm volatile("" : "+v"(v)); }
  return v;
}
template <int N>
DEV uint32_t schain(uint32_t v, uint32_t k) {
#pragma unroll
  for (int i = 0; i < N; ++i) { v = v + k; asm volatile("" : "+s"(v)); }
  return v;
}

template <int NC, int ND, int SIDE, int PIPE>
__global__ __launch_bounds__(128) void k_stage(const uint32_t *tab, uint64_t *out, int iters) {
  __shared__ uint32_t Yw, PV[64];
  const uint32_t lane = threadIdx.x & 63, w = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x == 0) { Yw = 0; }
  if (threadIdx.x < 64) PV[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t ay = (uint32_t)(uintptr_t)&Yw, apv = (uint32_t)(uintptr_t)&PV[lane];
  int acc = (int)lane;
  int side = (int)lane * 7;
  uint32_t left = 1u << 27;
  uint64_t t0 = now();
  if (w == 0) {
    // wave C: publishes p[] for bit 1 at once, then one per y
    for (int i = 1; i <= iters; ++i) {
      const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane(i);
      acc = vchain<NC>(acc, (int)s);
      asm volatile("ds_write_b32 %0, %1" ::"v"(apv), "v"(s << 12 | ((uint32_t)acc & 0xfffu)) : "memory");
      side = vchain<SIDE>(side, 5);
      if (PIPE) wait_eq_pipe(ay, s, left); else wait_eq(ay, s, left);
    }
  } else {
    for (int i = 1; i <= iters; ++i) {
      const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane(i);
      uint32_t v = wait_vec(apv, s, left);
      acc = vchain<ND>((int)v, acc);
      uint32_t idx = (uint32_t)__builtin_amdgcn_readlane(acc, 7) & 1023u, ps;
      idx *= 4;
      asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(ps) : "s"(tab), "s"(idx) : "memory");
      ps = schain<14>(ps, s);
      asm volatile("" ::"s"(ps));
      put0(ay, s);
      side = vchain<SIDE>(side, 5);
    }
  }
  uint64_t t1 = now();
  if (lane == 0) { out[w] = t1 - t0; out[2 + w] = (uint64_t)(uint32_t)(acc + side); }
}

__global__ void k_rate(const uint32_t *tab, uint64_t *out, int iters) {
  uint32_t idx = 1;
  uint64_t t0 = now();
  for (int i = 0; i < iters; ++i) {
    uint32_t v, off = idx * 4;
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(tab), "s"(off) : "memory");
    idx = v & 1023;
  }
  uint64_t t1 = now();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = idx; }
}

int main() {
  std::vector<uint32_t> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (uint32_t)((i * 1103515245u + 12345u) >> 8);
  uint32_t *d; uint64_t *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 64);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  const int iters = 200000;
  uint64_t r[4];
  // tick rate of s_memtime
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate, dim3(1), dim3(64), 0, 0, d, o, 1000);
  hipEventRecord(e0); hipLaunchKernelGGL(k_rate, dim3(1), dim3(64), 0, 0, d, o, iters * 10); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
  printf("s_memtime: %.1f ticks per microsecond (kernel %.3f ms)\n", (double)r[0] / (ms * 1000.0), ms);
  for (int pass = 0; pass < 2; ++pass) {
#define PP(mode, name)                                                                              \
    hipLaunchKernelGGL(k_pingpong<mode>, dim3(1), dim3(128), 0, 0, o, iters);                       \
    hipMemcpy(r, o, 8, hipMemcpyDeviceToHost);                                                      \
    printf("%-18s %.1f ticks per round trip (two hand-overs)\n", name, (double)r[0] / iters);
    PP(0, "pingpong_simple") PP(1, "pingpong_vec") PP(2, "pingpong_pipe") PP(3, "pingpong_bar")
#define ST(nc, nd, side, pipe)                                                                      \
    hipLaunchKernelGGL((k_stage<nc, nd, side, pipe>), dim3(1), dim3(128), 0, 0, d, o, iters);       \
    hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);                                                     \
    printf("stage_pipe C=%2d D=%2d side=%2d pipe=%d: %.1f ticks per bit\n", nc, nd, side, pipe, (double)r[0] / iters);
    ST(0, 0, 0, 0) ST(0, 0, 0, 1) ST(30, 8, 0, 0) ST(30, 8, 0, 1) ST(30, 8, 20, 1) ST(30, 8, 40, 1) ST(30, 8, 60, 1) ST(45, 25, 40, 1)
  }
  // many workgroups at once (one per CU, as the decoder runs): does the hand-over change under load?
  hipLaunchKernelGGL((k_stage<30, 8, 40, 1>), dim3(256), dim3(128), 0, 0, d, o, iters);
  hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
  printf("stage_pipe C=30 D= 8 side=40 pipe=1, 256 workgroups: %.1f ticks per bit\n", (double)r[0] / iters);
  return 0;
}
