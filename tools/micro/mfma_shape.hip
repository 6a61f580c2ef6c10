// Micro-benchmark: the two bf16 MFMA shapes of gfx950 in the register tiling of the bf16 convolution (wave tile 128 x 64, every operand
// fragment re-read from LDS by ds_read_b128, random data, two workgroups of 4 waves per CU): does v_mfma_f32_16x16x32_bf16 sustain more
// FLOP/s than v_mfma_f32_32x32x16_bf16 under load (MI355X_MICROARCH.md, DVFS item 7)?   hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];                       // 32 KB of random bf16 pairs
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4* L = (const f32x4*)lds;
  if (SHAPE == 32) {
    f32x16 acc[4][2];
    for (int m = 0; m < 4; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    for (int it = 0; it < iters; it++) {
      const int base = ((it * 7 + wave * 64) & 255) * 4;                           // walk the buffer
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        bf16x8 b[2], a[4];
#pragma unroll
        for (int n = 0; n < 2; n++) b[n] = __builtin_bit_cast(bf16x8, L[(base + (kk * 6 + n) * 64 + lane) & 2047]);
#pragma unroll
        for (int m = 0; m < 4; m++) a[m] = __builtin_bit_cast(bf16x8, L[(base + (kk * 6 + 2 + m) * 64 + lane) & 2047]);
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
          for (int n = 0; n < 2; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int m = 0; m < 4; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) s += acc[m][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x4 acc[8][4];
    for (int m = 0; m < 8; m++) for (int n = 0; n < 4; n++) for (int r = 0; r < 4; r++) acc[m][n][r] = 0.f;
    for (int it = 0; it < iters; it++) {
      const int base = ((it * 7 + wave * 64) & 255) * 4;
      bf16x8 b[4], a[8];                                                           // K = 32 per MFMA: the 12 fragments of one K-step
#pragma unroll
      for (int n = 0; n < 4; n++) b[n] = __builtin_bit_cast(bf16x8, L[(base + n * 64 + lane) & 2047]);
#pragma unroll
      for (int m = 0; m < 8; m++) a[m] = __builtin_bit_cast(bf16x8, L[(base + (4 + m) * 64 + lane) & 2047]);
#pragma unroll
      for (int m = 0; m < 8; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    float s = 0.f;
    for (int m = 0; m < 8; m++) for (int n = 0; n < 4; n++) for (int r = 0; r < 4; r++) s += acc[m][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

int main() {
  const int iters = 20000, grid = 256 * 2 * 8;
  std::vector<unsigned short> h(16384);
  srand(1);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 0.25f; unsigned int u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  float *src, *out;
  hipMalloc(&src, 32768); hipMalloc(&out, (size_t)grid * 256 * 4);
  hipMemcpy(src, h.data(), 32768, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = (double)grid * 4 * iters * 2.0 * 128 * 64 * 32;              // per wave and iteration: 128 x 64 x 32 MACs
  for (int rep = 0; rep < 3; rep++)
    for (int shape : {32, 16}) {
      hipEventRecord(e0);
      if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(grid), dim3(256), 0, 0, src, out, iters);
      else hipLaunchKernelGGL(k<16>, dim3(grid), dim3(256), 0, 0, src, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("shape %2d: %8.3f ms  %7.1f TFLOP/s\n", shape, ms, flop / ms / 1e9);
    }
  return 0;
}
