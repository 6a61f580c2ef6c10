// Shared helpers for the gfx950 (MI355X / CDNA4) kernels of the VQA hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LCT_OK 0
#define LCT_EINVAL (-1)
#define LCT_EUNSUPPORTED (-2)
#define LCT_EWORKSPACE (-3)

void lct_set_error(const char* fmt, ...);
int lct_check_launch(const char* what);

#define LCT_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      lct_set_error(__VA_ARGS__);         \
      return LCT_EINVAL;                  \
    }                                     \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// async global -> LDS copy, 16 B per lane.  LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Same copy, but opaque to the compiler's waitcnt insertion.  With the builtin the compiler must assume that any later
// LDS read may alias the in-flight DMA and drains vmcnt to 0 in front of it; kernels that keep unrelated global loads
// in flight across LDS reads use this form and order the DMA themselves (s_waitcnt vmcnt(N) + barrier).
__device__ __forceinline__ void glds16_opaque(const float* g, float* lds_wave_base) {
  const unsigned l = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory", "m0");
}

// The same copy with BUFFER addressing: a wave-uniform 128-bit descriptor (lct_make_rsrc on a wave-uniform base pointer), the lane's byte
// offset (32 bits, usually loop-invariant) and a wave-uniform byte offset in an SGPR (the loop's advance): no 64-bit vector address
// arithmetic per copy.  Opaque to the compiler's waitcnt bookkeeping like glds16_opaque.  Every lane must address bytes inside the
// buffer (a range-checked lane does not write its LDS slot): callers clamp, they do not mask.
typedef int lct_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lct_i32x4 lct_make_rsrc(const void* base) {
  const unsigned long a = (unsigned long)(uintptr_t)base;
  lct_i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);     // stride 0: raw buffer
  r[2] = 0x7ffffff0;                                                             // bytes addressable from base
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void bufdma16(unsigned voff, lct_i32x4 rsrc, unsigned soff, float* lds_wave_base) {
  const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(l) : "memory", "m0");
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
