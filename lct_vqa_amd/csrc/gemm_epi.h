// The GEMM epilogue descriptor shared by the fp32 GEMM (gemm.hip) and the bf16 GEMM (gemm_bf16.hip), and the host-side launcher
// of the split-K slab reduction that applies it (defined in gemm.hip).
#pragma once
#include "common.h"

struct GemmEpi {
  // fused A-operand prologue of the fusion head (reference basic_vqa/models.py:96-98 feeding fc1), fp32 general kernel only:
  //   A(m,k) := amask[m][k] ? ascale * tanh(A(m,k) * a2[m*a_ms + k*a_ks]) : 0      (a2 == null: plain A)
  const float* a2;
  const uint8_t* amask;  // dense [M][K] or null
  float ascale;
  const float* bias;     // [N] or null
  const float* addend;   // [M][ld_add] or null
  long ld_add;
  const uint8_t* mask;   // [M][ld_mask] or null (0/1)
  long ld_mask;
  float mask_scale;
  int act;               // 0 none, 1 relu, 2 tanh
};

__device__ __forceinline__ float apply_epi(float v, int m, int n, const GemmEpi& e) {
  if (e.bias) v += e.bias[n];
  if (e.addend) v += e.addend[(long)m * e.ld_add + n];
  if (e.act == 1) v = fmaxf(v, 0.f);
  else if (e.act == 2) v = tanhf(v);
  if (e.mask) v = e.mask[(long)m * e.ld_mask + n] ? v * e.mask_scale : 0.f;
  return v;
}

// C[m][n] = epi( sum_s ws[s][m][n] ) for the [splits][M][N] fp32 slabs of a split-K product (fixed order: deterministic)
int lct_splitk_reduce(const float* ws, int splits, float* C, long ldc, int M, int N, const GemmEpi& epi, hipStream_t st);
