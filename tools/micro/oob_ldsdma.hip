// What does a range-checked lane of `buffer_load_dwordx4 ... offen lds` do to its LDS slot: write zeros, or leave it alone?
// hipcc --offload-arch=gfx950 -O2 -I lct_vqa_amd/csrc tools/micro/oob_ldsdma.hip -o tools/micro/oob_ldsdma && tools/micro/oob_ldsdma
#include "common.h"
#include <cstdio>
#include <vector>
void lct_set_error(const char*, ...) {}
int lct_check_launch(const char*) { return 0; }

__global__ void probe(const float* src, int nbytes, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 7.f;
  __syncthreads();
  lct_i32x4 r = lct_make_rsrc(src);
  r[2] = nbytes;                                            // 512 bytes: lanes 32..63 are out of range
  bufdma16((unsigned)threadIdx.x * 16u, r, 0u, lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}

int main() {
  float *src, *out;
  hipMalloc(&src, 4096); hipMalloc(&out, 1024);
  std::vector<float> h(1024, 3.f);
  hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, 512, out);
  std::vector<float> o(256);
  hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
  printf("in-range slot (lane 0): %g   out-of-range slot (lane 40): %g %g %g %g   (7 = untouched, 0 = zero-filled, 3 = loaded)\n", o[0], o[160], o[161], o[162], o[163]);
  return 0;
}
