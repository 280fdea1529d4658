// zh_e8e9.h — the reference's E8E9 post-processor (LibZPAQ.cs:802-826, zh_native_pcomp_e8e9 is its instruction-for-instruction
// translation) as the few scalar operations it amounts to per byte, for the kernels that have matched a block's PCOMP against
// that exact program (zh_native_lookup) in a model whose M is one byte (pm = 0: `*b=b` and `a=*b` then name the same cell).
//
// The program keeps the last four input bytes in B (oldest in the low byte) and the number of bytes seen in C:
//   *b=b                                   M[0] = the byte about to leave B
//   a<<= 24 d=a a=b a>>= 8 a+=d b=a c++    B = B >> 8 | input << 24; C++
//   a=c a> 4 if  a=*b out                  from the fifth byte on: the byte that left B is written
//     a&= 254 a== 232 if                   ... and if it is E8 / E9
//       a=b a>>= 24 a++ a&= 254 a== 0 if   ... and the newest byte (the address's top byte) is 00 or FF:
//         a=b a>>= 24 a<<= 24 d=a  a=b a-=c a+= 5  a<<= 8 a>>= 8 a|=d b=a      the 24 low bits of B become B - C + 5
// M[0] is written and read inside one run, so (B, C) is all the state a run leaves behind (A, D, F are set before they are
// read in every run, the end-of-segment run included).
#pragma once
#include <stdint.h>

namespace {
// One input byte.  true: `outb` is the byte the program writes in this run.
__device__ __forceinline__ bool zh_e8e9_step(uint32_t &B, uint32_t &C, uint32_t x, uint32_t &outb) {
  const uint32_t m0 = B & 255u;
  B = (B >> 8) | (x << 24);
  ++C;
  if (C <= 4u) return false;
  outb = m0;
  if ((m0 & 254u) == 232u) {
    const uint32_t t = B >> 24;
    if (((t + 1u) & 254u) == 0u) B = ((B - C + 5u) & 0xFFFFFFu) | (t << 24);
  }
  return true;
}
}  // namespace
