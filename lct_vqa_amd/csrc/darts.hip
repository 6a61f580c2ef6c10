// PC-DARTS search-cell primitives (reference basic_vqa/pcdarts/operations.py:4-104, model_search.py:11-91)
// on NHWC fp32 activations.  Channel counts are tiny (C/4 = 4..16 per mixed op, <= 256 per cell output), so
// every kernel here is HBM-bound VALU work: one pass, channel-fastest thread mapping (coalesced NHWC rows),
// no reshaping into MFMA tiles (the 1x1 convolutions go through the strided GEMM with fused input ReLU).
// Reductions over pixels (BatchNorm statistics, depthwise-weight / architecture-weight gradients) are
// two-stage and deterministic: per-workgroup partials, then a fixed-order finish.
#include "common.h"

static inline int gs_blocks(long n, int cap = 8192) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}
#define GRID_STRIDE(o, total) for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < (total); o += (long)gridDim.x * blockDim.x)

// ---------------------------------------------------------------------------------------------------------
// channel slice copy  y[p][0:Cn] = x[p][c0:c0+Cn]   and its backward (dx[p][c] = c in slice ? dy : 0)
// ---------------------------------------------------------------------------------------------------------
__global__ void slice_channels_kernel(const float* __restrict__ x, long ldx, int c0, float* __restrict__ y, int Cn, long P) {
  GRID_STRIDE(o, P * Cn) { long p = o / Cn; int c = o % Cn; y[o] = x[p * ldx + c0 + c]; }
}
__global__ void slice_channels_bwd_kernel(const float* __restrict__ dy, int Cn, int c0, float* __restrict__ dx, int C, long P) {
  GRID_STRIDE(o, P * C) { long p = o / C; int c = o % C; dx[o] = (c >= c0 && c < c0 + Cn) ? dy[p * Cn + c - c0] : 0.f; }
}

// ---------------------------------------------------------------------------------------------------------
// depthwise KxK convolution (groups = C), stride s, padding p, dilation d, optional ReLU on the input
// (operations.py:41-42,56-61: ReLU -> depthwise conv).  x [B,H,W,C] dense, w [C][K][K], y [B,Ho,Wo,C]
// ---------------------------------------------------------------------------------------------------------
__global__ void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int B, int H,
                                  int W, int C, int K, int s, int p, int d, int Ho, int Wo, int relu_in) {
  GRID_STRIDE(o, (long)B * Ho * Wo * C) {
    int c = o % C; long q = o / C; int ox = q % Wo; int oy = (q / Wo) % Ho; int n = q / ((long)Wo * Ho);
    float a = 0.f;
    for (int ky = 0; ky < K; ky++) {
      int iy = oy * s - p + ky * d;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < K; kx++) {
        int ix = ox * s - p + kx * d;
        if (ix < 0 || ix >= W) continue;
        float v = x[(((long)n * H + iy) * W + ix) * C + c];
        if (relu_in) v = fmaxf(v, 0.f);
        a = fmaf(v, w[(c * K + ky) * K + kx], a);
      }
    }
    y[o] = a;
  }
}
__global__ void dwconv_bwd_input_kernel(const float* __restrict__ dy, const float* __restrict__ w, const float* __restrict__ x,
                                        float* __restrict__ dx, int B, int H, int W, int C, int K, int s, int p, int d, int Ho,
                                        int Wo, int relu_in) {
  GRID_STRIDE(o, (long)B * H * W * C) {
    int c = o % C; long q = o / C; int ix = q % W; int iy = (q / W) % H; int n = q / ((long)W * H);
    float a = 0.f;
    if (!relu_in || x[o] > 0.f) {
      for (int ky = 0; ky < K; ky++) {
        int ty = iy + p - ky * d;
        if (ty < 0 || ty % s) continue;
        int oy = ty / s;
        if (oy >= Ho) continue;
        for (int kx = 0; kx < K; kx++) {
          int tx = ix + p - kx * d;
          if (tx < 0 || tx % s) continue;
          int ox = tx / s;
          if (ox >= Wo) continue;
          a = fmaf(dy[(((long)n * Ho + oy) * Wo + ox) * C + c], w[(c * K + ky) * K + kx], a);
        }
      }
    }
    dx[o] = a;
  }
}
// partial[blk][tap*C + c] = sum over the block's output pixels of dy[p][c] * relu?(x[p shifted by tap][c]).
// thread = (weight index wi = t % NWc, pixel lane t / NWc): all 256 threads stream pixels, lanes meet in LDS.
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                float* __restrict__ part, int B, int H, int W, int C, int K,
                                                                int s, int p, int d, int Ho, int Wo, int relu_in, long per_blk) {
  __shared__ float red[256];
  const int KK = K * K, NW = KK * C;
  const long P = (long)B * Ho * Wo;
  const long q0 = blockIdx.x * per_blk, q1 = min(P, q0 + per_blk);
  for (int w0 = 0; w0 < NW; w0 += 256) {
    const int NWc = min(256, NW - w0);            // weights handled in this pass
    const int PL = 256 / NWc;                     // pixel lanes
    const int wi = threadIdx.x % NWc, pl = threadIdx.x / NWc;
    float a = 0.f;
    if (pl < PL) {
      const int t = w0 + wi, c = t % C, tap = t / C, ky = tap / K, kx = tap % K;
      for (long q = q0 + pl; q < q1; q += PL) {
        int ox = q % Wo; int oy = (q / Wo) % Ho; int n = q / ((long)Wo * Ho);
        int iy = oy * s - p + ky * d, ix = ox * s - p + kx * d;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        float v = x[(((long)n * H + iy) * W + ix) * C + c];
        if (relu_in) v = fmaxf(v, 0.f);
        a = fmaf(dy[q * C + c], v, a);
      }
    }
    __syncthreads();
    red[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < NWc) {
      float sum = 0.f;
      for (int l = 0; l < PL; l++) sum += red[l * NWc + threadIdx.x];
      part[(long)blockIdx.x * NW + w0 + threadIdx.x] = sum;
    }
  }
}
// dw[c][ky][kx] = sum_blk part[blk][tap*C + c]   (one wave per weight, strided over the partials)
__global__ __launch_bounds__(64) void dwconv_bwd_weight_finish_kernel(const float* __restrict__ part, int nblk, int C, int KK,
                                                                      float* __restrict__ dw) {
  const int t = blockIdx.x;
  const int c = t % C, tap = t / C;
  float a = 0.f;
  for (int b = threadIdx.x; b < nblk; b += 64) a += part[(long)b * C * KK + t];
  a = wave_sum(a);
  if (threadIdx.x == 0) dw[c * KK + tap] = a;
}

// float4 forms of the two kernels above for C % 4 == 0: thread = (pixel, channel quad); the K*K weights of the four
// channels are gathered per tap (w is [C][K][K]: stride K*K between channels, L1-resident).
__global__ void dwconv_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int B, int H,
                                      int W, int C, int K, int s, int p, int d, int Ho, int Wo, int relu_in) {
  const int CQ = C >> 2, KK = K * K;
  const f32x4* x4 = (const f32x4*)x;
  GRID_STRIDE(o, (long)B * Ho * Wo * CQ) {
    const int cq = o % CQ; const long q = o / CQ; const int ox = q % Wo, oy = (q / Wo) % Ho, n = q / ((long)Wo * Ho);
    const float* wc = w + (cq * 4) * KK;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < K; ky++) {
      const int iy = oy * s - p + ky * d;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < K; kx++) {
        const int ix = ox * s - p + kx * d;
        if (ix < 0 || ix >= W) continue;
        f32x4 v = x4[(((long)n * H + iy) * W + ix) * CQ + cq];
        if (relu_in) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        const int t = ky * K + kx;
        const f32x4 wv = {wc[t], wc[KK + t], wc[2 * KK + t], wc[3 * KK + t]};
        a += v * wv;
      }
    }
    ((f32x4*)y)[o] = a;
  }
}
__global__ void dwconv_bwd_input_vec_kernel(const float* __restrict__ dy, const float* __restrict__ w, const float* __restrict__ x,
                                            float* __restrict__ dx, int B, int H, int W, int C, int K, int s, int p, int d, int Ho,
                                            int Wo, int relu_in) {
  const int CQ = C >> 2, KK = K * K;
  const f32x4* dy4 = (const f32x4*)dy;
  GRID_STRIDE(o, (long)B * H * W * CQ) {
    const int cq = o % CQ; const long q = o / CQ; const int ix = q % W, iy = (q / W) % H, n = q / ((long)W * H);
    const float* wc = w + (cq * 4) * KK;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < K; ky++) {
      const int ty = iy + p - ky * d;
      if (ty < 0 || ty % s) continue;
      const int oy = ty / s;
      if (oy >= Ho) continue;
      for (int kx = 0; kx < K; kx++) {
        const int tx = ix + p - kx * d;
        if (tx < 0 || tx % s) continue;
        const int ox = tx / s;
        if (ox >= Wo) continue;
        const int t = ky * K + kx;
        const f32x4 wv = {wc[t], wc[KK + t], wc[2 * KK + t], wc[3 * KK + t]};
        a += dy4[(((long)n * Ho + oy) * Wo + ox) * CQ + cq] * wv;
      }
    }
    if (relu_in) {
      const f32x4 xv = ((const f32x4*)x)[o];
      a[0] = xv[0] > 0.f ? a[0] : 0.f; a[1] = xv[1] > 0.f ? a[1] : 0.f; a[2] = xv[2] > 0.f ? a[2] : 0.f; a[3] = xv[3] > 0.f ? a[3] : 0.f;
    }
    ((f32x4*)dx)[o] = a;
  }
}

// Compile-time-K forms of the two float4 kernels above (K = 3, 5, 7; item count < 2^31): the per-channel-quad weights are
// staged once per workgroup in LDS as float4 [tap][quad] (the loop reads them as broadcasts), every tap is loaded from a
// clamped, always valid address and zeroed afterwards when it lies outside the image, so the K*K loads of an item are
// independent and all in flight together (with a branch per tap each load waited for the previous one).
template <int K>
__global__ __launch_bounds__(256) void dwconv_fwd_vecK_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              float* __restrict__ y, int B, int H, int W, int C, int s, int p, int d,
                                                              int Ho, int Wo, int relu_in) {
  constexpr int KK = K * K;
  extern __shared__ __attribute__((aligned(16))) f32x4 swq[];     // [KK][CQ]
  const int CQ = C >> 2;
  for (int t = threadIdx.x; t < KK * CQ; t += blockDim.x) {
    const int tap = t / CQ, c = (t % CQ) * 4;
    swq[t] = f32x4{w[c * KK + tap], w[(c + 1) * KK + tap], w[(c + 2) * KK + tap], w[(c + 3) * KK + tap]};
  }
  __syncthreads();
  const f32x4* x4 = (const f32x4*)x;
  const int items = B * Ho * Wo * CQ;
  const int cq = threadIdx.x % CQ;                              // grid stride is a multiple of CQ
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < items; i += gridDim.x * blockDim.x) {
    const int q = i / CQ;
    const int ox = q % Wo, oy = (q / Wo) % Ho, n = q / (Wo * Ho);
    f32x4 v[KK];
    bool ok[KK];
#pragma unroll
    for (int ky = 0; ky < K; ky++) {
      const int iy = oy * s - p + ky * d;
      const bool oky = iy >= 0 && iy < H;
      const int rowb = (n * H + min(max(iy, 0), H - 1)) * W;
#pragma unroll
      for (int kx = 0; kx < K; kx++) {
        const int ix = ox * s - p + kx * d;
        ok[ky * K + kx] = oky && ix >= 0 && ix < W;
        v[ky * K + kx] = x4[(long)(rowb + min(max(ix, 0), W - 1)) * CQ + cq];
      }
    }
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KK; t++) {
      f32x4 u = v[t];
      if (relu_in) { u[0] = fmaxf(u[0], 0.f); u[1] = fmaxf(u[1], 0.f); u[2] = fmaxf(u[2], 0.f); u[3] = fmaxf(u[3], 0.f); }
      if (!ok[t]) u = f32x4{0.f, 0.f, 0.f, 0.f};
      a += u * swq[t * CQ + cq];
    }
    ((f32x4*)y)[i] = a;
  }
}

template <int K>
__global__ __launch_bounds__(256) void dwconv_bwd_input_vecK_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                    const float* __restrict__ x, float* __restrict__ dx, int B, int H,
                                                                    int W, int C, int s, int p, int d, int Ho, int Wo, int relu_in) {
  constexpr int KK = K * K;
  extern __shared__ __attribute__((aligned(16))) f32x4 swq[];     // [KK][CQ]
  const int CQ = C >> 2;
  for (int t = threadIdx.x; t < KK * CQ; t += blockDim.x) {
    const int tap = t / CQ, c = (t % CQ) * 4;
    swq[t] = f32x4{w[c * KK + tap], w[(c + 1) * KK + tap], w[(c + 2) * KK + tap], w[(c + 3) * KK + tap]};
  }
  __syncthreads();
  const f32x4* dy4 = (const f32x4*)dy;
  const int items = B * H * W * CQ;
  const int cq = threadIdx.x % CQ;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < items; i += gridDim.x * blockDim.x) {
    const int q = i / CQ;
    const int ix = q % W, iy = (q / W) % H, n = q / (W * H);
    f32x4 v[KK];
    bool ok[KK];
#pragma unroll
    for (int ky = 0; ky < K; ky++) {
      const int ty = iy + p - ky * d;                           // = oy * s for the output row that used this tap
      const int oy = max(ty, 0) / s;
      const bool oky = ty >= 0 && (ty % s) == 0 && oy < Ho;
      const int rowb = (n * Ho + min(oy, Ho - 1)) * Wo;
#pragma unroll
      for (int kx = 0; kx < K; kx++) {
        const int tx = ix + p - kx * d;
        const int ox = max(tx, 0) / s;
        ok[ky * K + kx] = oky && tx >= 0 && (tx % s) == 0 && ox < Wo;
        v[ky * K + kx] = dy4[(long)(rowb + min(ox, Wo - 1)) * CQ + cq];
      }
    }
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KK; t++) {
      f32x4 u = v[t];
      if (!ok[t]) u = f32x4{0.f, 0.f, 0.f, 0.f};
      a += u * swq[t * CQ + cq];
    }
    if (relu_in) {
      const f32x4 xv = ((const f32x4*)x)[i];
      a[0] = xv[0] > 0.f ? a[0] : 0.f; a[1] = xv[1] > 0.f ? a[1] : 0.f; a[2] = xv[2] > 0.f ? a[2] : 0.f; a[3] = xv[3] > 0.f ? a[3] : 0.f;
    }
    ((f32x4*)dx)[i] = a;
  }
}

// Same reduction for C % 4 == 0 (every channel count of the search and derived networks): thread = (output pixel,
// channel quad), one float4 of dy and K*K float4 of x per pixel (neighbours hit L1), K*K float4 accumulators in
// registers; lanes that own the same channel quad meet by shuffles, waves by LDS.  16-B coalesced loads instead of the
// scalar 4-B gathers of the generic kernel.   part[blk][tap*C + c] as above.
// v + (v rotated right by N lanes inside its row of 16 lanes): one DPP add, no LDS
template <int N>
__device__ __forceinline__ float row_ror_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xF, 0xF, false));
}

template <int K>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_vec_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                    float* __restrict__ part, int B, int H, int W, int C, int s, int p,
                                                                    int d, int Ho, int Wo, int relu_in) {
  constexpr int KK = K * K;
  __shared__ f32x4 red[16][KK][4];                              // [wave * 4 + row of 16 lanes][tap][channel quad (<= 4 per pass)]
  const int CQ = C >> 2;                                        // power of two, 1..16 (host-checked)
  const int items = B * Ho * Wo * CQ;                           // < 2^31 (host-checked)
  const f32x4* dy4 = (const f32x4*)dy;
  const f32x4* x4 = (const f32x4*)x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the grid stride is a multiple of CQ (blockDim = 256), so a thread keeps its channel quad
  const int cq = threadIdx.x % CQ;
  f32x4 acc[KK];
#pragma unroll
  for (int t = 0; t < KK; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < items; i += gridDim.x * blockDim.x) {
    const int q = i / CQ;
    const int ox = q % Wo, oy = (q / Wo) % Ho, n = q / (Wo * Ho);
    const f32x4 g = dy4[i];
    // every tap is loaded from a clamped (always valid) address and zeroed afterwards when it lies outside the image:
    // no branches, so the K*K loads of an item are all in flight together
    f32x4 v[KK];
    bool ok[KK];
#pragma unroll
    for (int ky = 0; ky < K; ky++) {
      const int iy = oy * s - p + ky * d;
      const bool oky = iy >= 0 && iy < H;
      const int rowb = (n * H + min(max(iy, 0), H - 1)) * W;
#pragma unroll
      for (int kx = 0; kx < K; kx++) {
        const int ix = ox * s - p + kx * d;
        ok[ky * K + kx] = oky && ix >= 0 && ix < W;
        v[ky * K + kx] = x4[(long)(rowb + min(max(ix, 0), W - 1)) * CQ + cq];
      }
    }
#pragma unroll
    for (int t = 0; t < KK; t++) {
      f32x4 u = v[t];
      if (relu_in) { u[0] = fmaxf(u[0], 0.f); u[1] = fmaxf(u[1], 0.f); u[2] = fmaxf(u[2], 0.f); u[3] = fmaxf(u[3], 0.f); }
      if (!ok[t]) u = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[t] += g * u;
    }
  }
  // lanes with equal (lane % CQ) hold partial sums of the same channel quad: fold them inside each row of 16 lanes by DPP
  // rotations (lane l += lane l+8, +4, ... down to CQ), the 16 rows of the workgroup meet in LDS
#pragma unroll
  for (int t = 0; t < KK; t++) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float a = acc[t][e];
      if (CQ <= 8) a = row_ror_add<8>(a);
      if (CQ <= 4) a = row_ror_add<4>(a);
      if (CQ <= 2) a = row_ror_add<2>(a);
      if (CQ <= 1) a = row_ror_add<1>(a);
      acc[t][e] = a;
    }
  }
  const int rl = lane & 15, row = wave * 4 + (lane >> 4);
  for (int c0 = 0; c0 < CQ; c0 += 4) {          // CQ <= 4 in one pass through the LDS buffer, 8 / 16 in two / four
    __syncthreads();
    if (rl >= c0 && rl < c0 + 4 && rl < CQ) {
#pragma unroll
      for (int t = 0; t < KK; t++) red[row][t][rl - c0] = acc[t];
    }
    __syncthreads();
    const int nq = min(4, CQ - c0);
    for (int o = threadIdx.x; o < KK * nq; o += blockDim.x) {
      const int t = o / nq, qd = o % nq;
      f32x4 sum = red[0][t][qd];
#pragma unroll
      for (int r = 1; r < 16; r++) sum += red[r][t][qd];
      float* dst = part + (long)blockIdx.x * KK * C + t * C + (c0 + qd) * 4;
      dst[0] = sum[0]; dst[1] = sum[1]; dst[2] = sum[2]; dst[3] = sum[3];
    }
  }
}

// 1x1 convolution for small channel counts (Cin, Cout multiples of 4, Cout <= 64: the DARTS cells), forward and input
// gradient in one kernel:   out[p][o] = sum_i relu?(in[p][i]) * W(o,i)   [* (mask[p][o] > 0)]
//   forward : in = x,  W(o,i) = w[o][i],  relu? = the ReLU in front of the conv
//   dgrad   : in = dy, W(o,i) = w[i][o] (TRANSPOSE),  mask = x (gradient of the same ReLU)
// A P x 16 x 16 product is pure memory traffic; the generic 128x128 MFMA GEMM spends its time on padding.  thread = pixel:
// reads its input row once (float4s), keeps all outputs in registers, takes the weights from LDS (every lane of a wave
// reads the same address: broadcast), writes float4s.
template <int NOUT, bool TRANSPOSE>
__global__ __launch_bounds__(256) void pwconv_small_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                           const float* __restrict__ mask, float* __restrict__ out, long P, int NIN,
                                                           int relu_in) {
  extern __shared__ __attribute__((aligned(16))) float sw[];    // [NOUT][NIN]
  for (int t = threadIdx.x; t < NOUT * NIN; t += blockDim.x) {
    const int o = t / NIN, i = t % NIN;
    sw[t] = TRANSPOSE ? w[(long)i * NOUT + o] : w[t];
  }
  __syncthreads();
  const int NQ = NIN >> 2;
  const f32x4* in4 = (const f32x4*)in;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    float acc[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; o++) acc[o] = 0.f;
    for (int q = 0; q < NQ; q++) {
      f32x4 v = in4[p * NQ + q];
      if (relu_in) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
#pragma unroll
      for (int o = 0; o < NOUT; o++) {
        const f32x4 wv = *(const f32x4*)(sw + o * NIN + 4 * q);
        acc[o] = fmaf(v[0], wv[0], fmaf(v[1], wv[1], fmaf(v[2], wv[2], fmaf(v[3], wv[3], acc[o]))));
      }
    }
    f32x4* o4 = (f32x4*)(out + p * NOUT);
#pragma unroll
    for (int oq = 0; oq < NOUT / 4; oq++) {
      f32x4 r = {acc[4 * oq], acc[4 * oq + 1], acc[4 * oq + 2], acc[4 * oq + 3]};
      if (mask) {
        const f32x4 m = ((const f32x4*)(mask + p * NOUT))[oq];
        r[0] = m[0] > 0.f ? r[0] : 0.f; r[1] = m[1] > 0.f ? r[1] : 0.f; r[2] = m[2] > 0.f ? r[2] : 0.f; r[3] = m[3] > 0.f ? r[3] : 0.f;
      }
      o4[oq] = r;
    }
  }
}

// 1x1-convolution weight gradient for small channel counts (Cout * Cin <= 4096; the DARTS cells: 4..64 channels):
//   dw[co][ci] = sum_p dy[p][co] * relu?(x[p][ci]),  P = B*H*W pixels (10^5..10^6): a 16x16 "GEMM" with K = P, for which a
// 128x128 MFMA tile is >98 % padding.  A block stages 256 pixels of dy and x in LDS (coalesced), thread = (output o,
// pixel lane): the outputs are spread over all 256 threads, lanes meet in LDS; part[blk][o], finished by the same
// column-sum kernel as the depthwise gradient.
__global__ __launch_bounds__(256) void pwconv_wgrad_small_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 float* __restrict__ part, long P, int Cout, int Cin, int relu_in,
                                                                 long per_blk) {
  constexpr int TP = 256;
  extern __shared__ float sm[];
  float* sdy = sm;                       // [TP][Cout]
  float* sx = sm + TP * Cout;            // [TP][Cin]
  __shared__ float red[256];
  const int NO = Cout * Cin;
  const long q0 = blockIdx.x * per_blk, q1 = min(P, q0 + per_blk);
  const int NOc = min(256, NO);                                 // outputs per pass
  const int PL = 256 / NOc;                                     // pixel lanes
  for (int o0 = 0; o0 < NO; o0 += 256) {
    const int oi = threadIdx.x % NOc, pl = threadIdx.x / NOc;
    const int o = o0 + oi;
    const bool live = pl < PL && o < NO;
    const int co = live ? o / Cin : 0, ci = live ? o % Cin : 0;
    float a = 0.f;
    for (long qb = q0; qb < q1; qb += TP) {
      const int np = (int)min((long)TP, q1 - qb);
      __syncthreads();
      for (int t = threadIdx.x; t < np * Cout; t += 256) sdy[t] = dy[qb * Cout + t];
      for (int t = threadIdx.x; t < np * Cin; t += 256) {
        float v = x[qb * Cin + t];
        sx[t] = relu_in ? fmaxf(v, 0.f) : v;
      }
      __syncthreads();
      if (live)
        for (int pp = pl; pp < np; pp += PL) a = fmaf(sdy[pp * Cout + co], sx[pp * Cin + ci], a);
    }
    __syncthreads();
    red[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < NOc && o0 + threadIdx.x < NO) {
      float sum = 0.f;
      for (int l = 0; l < PL; l++) sum += red[l * NOc + threadIdx.x];
      part[(long)blockIdx.x * NO + o0 + threadIdx.x] = sum;
    }
  }
}
// The same gradient for the cells' own 1x1 convs (Cin in {4, 8, 16}, Cout in {4, 8, 16}: every pointwise conv inside a MixedOp):
// thread = (pixel, quad of output channels); one float4 of dy and CIN/4 float4s of x per item, the 4 x CIN outer product
// accumulates in registers, so the kernel streams dy and x once with 16-B coalesced loads and no barrier in the loop.  Lanes
// of one output quad fold inside each row of 16 lanes by DPP rotations, the 16 rows of the workgroup meet in LDS
// (the structure of dwconv_bwd_weight_vec_kernel).  part[blk][co * CIN + ci] as above.
template <int CIN>
__global__ __launch_bounds__(256) void pwconv_wgrad_quad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                float* __restrict__ part, int P, int Cout, int relu_in) {
  constexpr int NQ = CIN / 4;
  __shared__ f32x4 red[16][4][4][NQ];                           // [row of 16 lanes][output quad][e][input quad]
  const int CQ = Cout >> 2;                                     // 1, 2 or 4
  const int items = P * CQ;
  const f32x4* dy4 = (const f32x4*)dy;
  const f32x4* x4 = (const f32x4*)x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[4][NQ];
#pragma unroll
  for (int e = 0; e < 4; e++)
#pragma unroll
    for (int n = 0; n < NQ; n++) acc[e][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < items; i += gridDim.x * blockDim.x) {
    const int pix = i / CQ;
    const f32x4 g = dy4[i];
    f32x4 xv[NQ];
#pragma unroll
    for (int n = 0; n < NQ; n++) xv[n] = x4[(long)pix * NQ + n];
#pragma unroll
    for (int n = 0; n < NQ; n++) {
      f32x4 u = xv[n];
      if (relu_in) { u[0] = fmaxf(u[0], 0.f); u[1] = fmaxf(u[1], 0.f); u[2] = fmaxf(u[2], 0.f); u[3] = fmaxf(u[3], 0.f); }
#pragma unroll
      for (int e = 0; e < 4; e++) acc[e][n] += g[e] * u;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; e++)
#pragma unroll
    for (int n = 0; n < NQ; n++)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float a = acc[e][n][k];
        a = row_ror_add<8>(a);
        a = row_ror_add<4>(a);
        if (CQ <= 2) a = row_ror_add<2>(a);
        if (CQ <= 1) a = row_ror_add<1>(a);
        acc[e][n][k] = a;
      }
  const int rl = lane & 15, row = wave * 4 + (lane >> 4);
  if (rl < CQ) {
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int n = 0; n < NQ; n++) red[row][rl][e][n] = acc[e][n];
  }
  __syncthreads();
  const int NO = Cout * CIN;
  for (int o = threadIdx.x; o < NO; o += blockDim.x) {
    const int co = o / CIN, ci = o % CIN;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) sum += red[r][co >> 2][co & 3][ci >> 2][ci & 3];
    part[(long)blockIdx.x * NO + o] = sum;
  }
}

// dW[co][ci] = sum_p dY[p][co] * (relu) X[p][ci] for the preprocess 1x1 convolutions (16..64 outputs, 48..256 inputs, P = 0.2 .. 3.2 M
// pixels): a reduction over the pixels, not a GEMM shape -- as a split-K TN GEMM it ran at 1.2 TB/s of its operands.  Workgroup = a slice of
// the pixels, staged TP rows at a time in LDS (whole rows, 16-byte loads); thread = a 4 x RBI block of dW (its RBI inputs interleaved in
// groups of four NBI*4 apart: neighbouring lanes read neighbouring 16-byte words) x a pixel lane; one 16-byte read of dY and RBI/4 of X
// per 4*RBI multiply-adds.  Per-workgroup partials, summed in a fixed order by colsum_partials_kernel.
template <int RBI>
__global__ __launch_bounds__(256) void pwconv_wgrad_blk_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part,
                                                               long P, int Cout, int Cin, int relu_in, long per_blk, int TP) {
  extern __shared__ __align__(16) float smw[];
  f32x4* sdy = (f32x4*)smw;                       // [TP][Cout/4]
  f32x4* sx = sdy + TP * (Cout / 4);              // [TP][Cin/4]
  const int NO4 = Cout / 4, NI4 = Cin / 4, NBI = Cin / RBI, NBT = NO4 * NBI, PL = 256 / NBT;
  const int bt = threadIdx.x % NBT, pl = threadIdx.x / NBT;
  const int bo = bt / NBI, bi = bt % NBI;
  const bool live = pl < PL;
  f32x4 acc[RBI];                                  // acc[4k + j][a]: dW[4 bo + a][4 (bi + NBI k) + j]
#pragma unroll
  for (int i = 0; i < RBI; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long q0 = blockIdx.x * per_blk, q1 = min(P, q0 + per_blk);
  for (long qb = q0; qb < q1; qb += TP) {
    const int np = (int)min((long)TP, q1 - qb);
    __syncthreads();
    const f32x4* gd = (const f32x4*)(dy + qb * Cout);
    const f32x4* gx = (const f32x4*)(x + qb * Cin);
    for (int t = threadIdx.x; t < np * NO4; t += 256) sdy[t] = gd[t];
    for (int t = threadIdx.x; t < np * NI4; t += 256) {
      f32x4 v = gx[t];
      if (relu_in) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
      sx[t] = v;
    }
    __syncthreads();
    if (live)
      for (int pp = pl; pp < np; pp += PL) {
        const f32x4 g4 = sdy[pp * NO4 + bo];
#pragma unroll
        for (int k = 0; k < RBI / 4; k++) {
          const f32x4 x4 = sx[pp * NI4 + bi + NBI * k];
#pragma unroll
          for (int j = 0; j < 4; j++)
#pragma unroll
            for (int a = 0; a < 4; a++) acc[4 * k + j][a] = fmaf(g4[a], x4[j], acc[4 * k + j][a]);
        }
      }
  }
  // fold the pixel lanes (fixed order) and write this workgroup's partial row
  float* prow = part + (long)blockIdx.x * Cout * Cin;
  if (PL == 1) {
    if (live) {
#pragma unroll
      for (int k = 0; k < RBI / 4; k++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int a = 0; a < 4; a++) prow[(4 * bo + a) * Cin + 4 * (bi + NBI * k) + j] = acc[4 * k + j][a];
    }
    return;
  }
  __syncthreads();
  float* red = smw;                               // [PL][NBT][4 * RBI]   (fits: checked on the host)
  if (live) {
#pragma unroll
    for (int i = 0; i < RBI; i++) *(f32x4*)(red + ((pl * NBT + bt) * RBI + i) * 4) = acc[i];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NBT * RBI * 4; e += 256) {
    float sum = 0.f;
    for (int l = 0; l < PL; l++) sum += red[l * NBT * RBI * 4 + e];
    const int b = e / (RBI * 4), i = (e / 4) % RBI, a = e % 4;
    const int eo = b / NBI, ei = b % NBI;
    prow[(4 * eo + a) * Cin + 4 * (ei + NBI * (i / 4)) + i % 4] = sum;
  }
}

// out[t] = sum_blk part[blk][t]   (one wave per output)
__global__ __launch_bounds__(64) void colsum_partials_kernel(const float* __restrict__ part, int nblk, int NO, float* __restrict__ out) {
  const int t = blockIdx.x;
  float a = 0.f;
  for (int b = threadIdx.x; b < nblk; b += 64) a += part[(long)b * NO + t];
  a = wave_sum(a);
  if (threadIdx.x == 0) out[t] = a;
}

// ---------------------------------------------------------------------------------------------------------
// 3x3 pooling, padding 1, stride s (operations.py:6-7): kind 0 = max, 1 = avg (count_include_pad=False)
// ---------------------------------------------------------------------------------------------------------
__global__ void pool3x3_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int s, int Ho,
                                   int Wo, int kind) {
  GRID_STRIDE(o, (long)B * Ho * Wo * C) {
    int c = o % C; long q = o / C; int ox = q % Wo; int oy = (q / Wo) % Ho; int n = q / ((long)Wo * Ho);
    float a = kind ? 0.f : -INFINITY;
    int cnt = 0;
    for (int ky = 0; ky < 3; ky++) {
      int iy = oy * s - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; kx++) {
        int ix = ox * s - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        float v = x[(((long)n * H + iy) * W + ix) * C + c];
        if (kind) a += v; else a = fmaxf(a, v);
        cnt++;
      }
    }
    y[o] = kind ? a / (float)cnt : a;
  }
}
// the same, four channels per thread (16-byte loads and stores; C % 4 == 0)
__global__ void pool3x3_fwd_vec_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C4, int s, int Ho,
                                       int Wo, int kind) {
  GRID_STRIDE(o, (long)B * Ho * Wo * C4) {
    int c = o % C4; long q = o / C4; int ox = q % Wo; int oy = (q / Wo) % Ho; int n = q / ((long)Wo * Ho);
    f32x4 a = kind ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int cnt = 0;
    for (int ky = 0; ky < 3; ky++) {
      int iy = oy * s - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; kx++) {
        int ix = ox * s - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const f32x4 v = *(const f32x4*)(x + ((((long)n * H + iy) * W + ix) * C4 + c) * 4);
        if (kind) a += v;
        else { a[0] = fmaxf(a[0], v[0]); a[1] = fmaxf(a[1], v[1]); a[2] = fmaxf(a[2], v[2]); a[3] = fmaxf(a[3], v[3]); }
        cnt++;
      }
    }
    if (kind) a = a / (float)cnt;
    *(f32x4*)(y + o * 4) = a;
  }
}
// thread per INPUT element: gathers the gradient of every window that selected it (max: first maximal element in
// scan order, like PyTorch) or covers it (avg)
__global__ void pool3x3_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int B,
                                   int H, int W, int C, int s, int Ho, int Wo, int kind) {
  GRID_STRIDE(o, (long)B * H * W * C) {
    int c = o % C; long q = o / C; int ix = q % W; int iy = (q / W) % H; int n = q / ((long)W * H);
    float a = 0.f;
    for (int oy = (iy + 1 - 2 + s - 1) / s; oy * s - 1 <= iy && oy < Ho; oy++) {
      if (oy < 0) continue;
      for (int ox = (ix + 1 - 2 + s - 1) / s; ox * s - 1 <= ix && ox < Wo; ox++) {
        if (ox < 0) continue;
        const float g = dy[(((long)n * Ho + oy) * Wo + ox) * C + c];
        if (kind) {
          int y0 = max(oy * s - 1, 0), y1 = min(oy * s + 1, H - 1), x0 = max(ox * s - 1, 0), x1 = min(ox * s + 1, W - 1);
          a += g / (float)((y1 - y0 + 1) * (x1 - x0 + 1));
        } else {
          // is (iy,ix) the first maximum of window (oy,ox)?
          float best = -INFINITY; int by = -1, bx = -1;
          for (int ky = 0; ky < 3; ky++) {
            int yy = oy * s - 1 + ky;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < 3; kx++) {
              int xx = ox * s - 1 + kx;
              if (xx < 0 || xx >= W) continue;
              float v = x[(((long)n * H + yy) * W + xx) * C + c];
              if (v > best) { best = v; by = yy; bx = xx; }
            }
          }
          if (by == iy && bx == ix) a += g;
        }
      }
    }
    dx[o] = a;
  }
}

// float4 form (C % 4 == 0): thread = (input pixel, channel quad); the first-maximum search of every covering window runs
// on four channels at once.
__global__ void pool3x3_bwd_vec_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int B,
                                       int H, int W, int C, int s, int Ho, int Wo, int kind) {
  const int CQ = C >> 2;
  const f32x4* dy4 = (const f32x4*)dy;
  const f32x4* x4 = (const f32x4*)x;
  GRID_STRIDE(o, (long)B * H * W * CQ) {
    const int cq = o % CQ; const long q = o / CQ; const int ix = q % W, iy = (q / W) % H, n = q / ((long)W * H);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int oy = (iy + 1 - 2 + s - 1) / s; oy * s - 1 <= iy && oy < Ho; oy++) {
      if (oy < 0) continue;
      for (int ox = (ix + 1 - 2 + s - 1) / s; ox * s - 1 <= ix && ox < Wo; ox++) {
        if (ox < 0) continue;
        const f32x4 g = dy4[(((long)n * Ho + oy) * Wo + ox) * CQ + cq];
        if (kind) {
          const int y0 = max(oy * s - 1, 0), y1 = min(oy * s + 1, H - 1), x0 = max(ox * s - 1, 0), x1 = min(ox * s + 1, W - 1);
          a += g / (float)((y1 - y0 + 1) * (x1 - x0 + 1));
        } else {
          // per channel: is (iy,ix) the FIRST maximum (scan order) of window (oy,ox)?
          f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
          int pos[4] = {-1, -1, -1, -1};
          for (int ky = 0; ky < 3; ky++) {
            const int yy = oy * s - 1 + ky;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < 3; kx++) {
              const int xx = ox * s - 1 + kx;
              if (xx < 0 || xx >= W) continue;
              const f32x4 v = x4[(((long)n * H + yy) * W + xx) * CQ + cq];
#pragma unroll
              for (int e = 0; e < 4; e++)
                if (v[e] > best[e]) { best[e] = v[e]; pos[e] = yy * W + xx; }
            }
          }
          const int me = iy * W + ix;
#pragma unroll
          for (int e = 0; e < 4; e++)
            if (pos[e] == me) a[e] += g[e];
        }
      }
    }
    ((f32x4*)dx)[o] = a;
  }
}

// ---------------------------------------------------------------------------------------------------------
// spatial subsample with optional ReLU (FactorizedReduce, operations.py:100-104):
//   y[n,oy,ox,:] = relu?(x[n, 2*oy+off, 2*ox+off, :])   and its backward (scatter, zero elsewhere)
// ---------------------------------------------------------------------------------------------------------
__global__ void subsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int off,
                                      int relu) {
  const int Ho = H / 2, Wo = W / 2;
  GRID_STRIDE(o, (long)B * Ho * Wo * C) {
    int c = o % C; long q = o / C; int ox = q % Wo; int oy = (q / Wo) % Ho; int n = q / ((long)Wo * Ho);
    float v = x[(((long)n * H + 2 * oy + off) * W + 2 * ox + off) * C + c];
    y[o] = relu ? fmaxf(v, 0.f) : v;
  }
}
// dx = sum over the two taps (off 0 from dy0, off 1 from dy1), masked by relu'(x)
__global__ void subsample2_bwd_kernel(const float* __restrict__ dy0, const float* __restrict__ dy1, const float* __restrict__ x,
                                      float* __restrict__ dx, int B, int H, int W, int C, int relu) {
  const int Ho = H / 2, Wo = W / 2;
  GRID_STRIDE(o, (long)B * H * W * C) {
    int c = o % C; long q = o / C; int ix = q % W; int iy = (q / W) % H; int n = q / ((long)W * H);
    float a = 0.f;
    if ((iy & 1) == (ix & 1)) {
      const int off = iy & 1;
      const float* src = off ? dy1 : dy0;
      if (src) a = src[(((long)n * Ho + (iy >> 1)) * Wo + (ix >> 1)) * C + c];
    }
    if (relu && !(x[o] > 0.f)) a = 0.f;
    dx[o] = a;
  }
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm2d (operations.py:29,43,62,98; model_search.py:37,113): training = batch statistics over B*H*W per
// channel (biased variance for normalisation, unbiased for the running estimate), eval = running statistics.
// x may be a channel slice: row stride ldx.  Statistics in fp64 partials (12.8 M pixels per channel at 224^2).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, long ldx, long P, int C, long per_blk,
                                                               double* __restrict__ part) {
  __shared__ double s1[256], s2[256];
  const int nrl = 256 / C;                  // row lanes (C <= 256)
  const int c = threadIdx.x % C, rl = threadIdx.x / C;
  const long q0 = blockIdx.x * per_blk, q1 = min(P, q0 + per_blk);
  double a = 0.0, b = 0.0;
  if (rl < nrl)
    for (long q = q0 + rl; q < q1; q += nrl) { double v = x[q * ldx + c]; a += v; b += v * v; }
  s1[threadIdx.x] = a; s2[threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.x < C) {
    double ta = 0.0, tb = 0.0;
    for (int r = 0; r < nrl; r++) { ta += s1[r * C + threadIdx.x]; tb += s2[r * C + threadIdx.x]; }
    part[((long)blockIdx.x * 2 + 0) * C + threadIdx.x] = ta;
    part[((long)blockIdx.x * 2 + 1) * C + threadIdx.x] = tb;
  }
}
// The same sums with 16-byte loads, four rows in flight per thread (thread = (row lane, channel quad); C % 4 == 0, 16-byte aligned rows):
// the scalar form above issued one 4-byte load per dependent iteration and ran at 1.5 TB/s.  Same fp64 accumulation; the row lanes are
// folded in the same fixed order.  MODE 0: sum x, sum x^2;  MODE 1: sum dy, sum dy * xhat.
template <int MODE>
__global__ __launch_bounds__(256) void bn_sums_vec_kernel(const float* __restrict__ a_, long lda, const float* __restrict__ x, long ldx,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd, long P, int C,
                                                          long per_blk, double* __restrict__ part) {
  __shared__ double s1[256 * 4 / 4 * 4], s2[256 * 4 / 4 * 4];    // [row lane][C]
  const int CQ = C / 4, nrl = 256 / CQ;
  const int q4 = threadIdx.x % CQ, rl = threadIdx.x / CQ;
  const long q0 = blockIdx.x * per_blk, q1 = min(P, q0 + per_blk);
  double sa[4] = {0.0, 0.0, 0.0, 0.0}, sb[4] = {0.0, 0.0, 0.0, 0.0};
  if (rl < nrl) {
    f32x4 m = {0.f, 0.f, 0.f, 0.f}, is = m;
    if (MODE == 1) { m = *(const f32x4*)(mean + 4 * q4); is = *(const f32x4*)(invstd + 4 * q4); }
    long q = q0 + rl;
    for (; q + 3L * nrl < q1; q += 4L * nrl) {
      f32x4 va[4], vx[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        va[u] = *(const f32x4*)(a_ + (q + (long)u * nrl) * lda + 4 * q4);
        if (MODE == 1) vx[u] = *(const f32x4*)(x + (q + (long)u * nrl) * ldx + 4 * q4);
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const double g = va[u][e];
          sa[e] += g;
          sb[e] += MODE == 1 ? g * (double)((vx[u][e] - m[e]) * is[e]) : g * g;
        }
    }
    for (; q < q1; q += nrl) {
      const f32x4 va = *(const f32x4*)(a_ + q * lda + 4 * q4);
      f32x4 vx = va;
      if (MODE == 1) vx = *(const f32x4*)(x + q * ldx + 4 * q4);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const double g = va[e];
        sa[e] += g;
        sb[e] += MODE == 1 ? g * (double)((vx[e] - m[e]) * is[e]) : g * g;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; e++) { s1[rl * C + 4 * q4 + e] = sa[e]; s2[rl * C + 4 * q4 + e] = sb[e]; }
  __syncthreads();
  if (threadIdx.x < C) {
    double ta = 0.0, tb = 0.0;
    for (int r = 0; r < nrl; r++) { ta += s1[r * C + threadIdx.x]; tb += s2[r * C + threadIdx.x]; }
    part[((long)blockIdx.x * 2 + 0) * C + threadIdx.x] = ta;
    part[((long)blockIdx.x * 2 + 1) * C + threadIdx.x] = tb;
  }
}
static bool bn_vec_ok(const float* a, long lda, const float* x, long ldx, int C) {
  return C % 4 == 0 && C <= 256 && 256 % (C / 4) == 0 && lda % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)a | (uintptr_t)x) & 15) == 0;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// mean/invstd (+ running stats update: momentum m, unbiased variance); one wave per channel
__global__ __launch_bounds__(64) void bn_stats_finish_kernel(const double* __restrict__ part, int nblk, int C, long P, float eps,
                                                             float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                                             float* __restrict__ run_mean, float* __restrict__ run_var,
                                                             long* __restrict__ batches_tracked) {
  const int c = blockIdx.x;
  if (batches_tracked && c == 0 && threadIdx.x == 0) batches_tracked[0] += 1;   // nn.BatchNorm2d.num_batches_tracked
  double a = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 64) { a += part[((long)k * 2) * C + c]; b += part[((long)k * 2 + 1) * C + c]; }
  a = wave_sum_d(a); b = wave_sum_d(b);
  if (threadIdx.x != 0) return;
  const double m = a / (double)P;
  double var = b / (double)P - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
    run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * m);
    run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
  }
}
__global__ void bn_eval_stats_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var, float eps, int C,
                                     float* __restrict__ mean, float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { mean[c] = run_mean[c]; invstd[c] = 1.f / sqrtf(run_var[c] + eps); }
}
__global__ void bn_apply_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ mean,
                                const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                float* __restrict__ y, long ldy, long P, int C) {
  GRID_STRIDE(o, P * C) {
    long p = o / C; int c = o % C;
    float v = (x[p * ldx + c] - mean[c]) * invstd[c];
    if (gamma) v = v * gamma[c] + beta[c];
    y[p * ldy + c] = v;
  }
}
// partial sums of dy and dy*xhat per channel (xhat recomputed from x)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                             long ldx, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, long P, int C, long per_blk,
                                                             double* __restrict__ part) {
  __shared__ double s1[256], s2[256];
  const int nrl = 256 / C;
  const int c = threadIdx.x % C, rl = threadIdx.x / C;
  const long q0 = blockIdx.x * per_blk, q1 = min(P, q0 + per_blk);
  double a = 0.0, b = 0.0;
  if (rl < nrl) {
    const float m = mean[c], is = invstd[c];
    for (long q = q0 + rl; q < q1; q += nrl) {
      double g = dy[q * lddy + c];
      a += g; b += g * (double)((x[q * ldx + c] - m) * is);
    }
  }
  s1[threadIdx.x] = a; s2[threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.x < C) {
    double ta = 0.0, tb = 0.0;
    for (int r = 0; r < nrl; r++) { ta += s1[r * C + threadIdx.x]; tb += s2[r * C + threadIdx.x]; }
    part[((long)blockIdx.x * 2 + 0) * C + threadIdx.x] = ta;
    part[((long)blockIdx.x * 2 + 1) * C + threadIdx.x] = tb;
  }
}
__global__ __launch_bounds__(64) void bn_bwd_finish_kernel(const double* __restrict__ part, int nblk, int C, float* __restrict__ sum_dy,
                                                           float* __restrict__ sum_dy_xhat) {
  const int c = blockIdx.x;
  double a = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 64) { a += part[((long)k * 2) * C + c]; b += part[((long)k * 2 + 1) * C + c]; }
  a = wave_sum_d(a); b = wave_sum_d(b);
  if (threadIdx.x == 0) { sum_dy[c] = (float)a; sum_dy_xhat[c] = (float)b; }
}
// training: dx = g*invstd*(dy - sum_dy/P - xhat*sum_dy_xhat/P);  eval: dx = g*invstd*dy
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ sum_dy,
                                    const float* __restrict__ sum_dy_xhat, float* __restrict__ dx, long P, int C, int training) {
  const float invP = 1.f / (float)P;
  GRID_STRIDE(o, P * C) {
    long p = o / C; int c = o % C;
    const float g = gamma ? gamma[c] : 1.f;
    float v = dy[p * lddy + c];
    if (training) {
      const float xh = (x[p * ldx + c] - mean[c]) * invstd[c];
      v = v - sum_dy[c] * invP - xh * sum_dy_xhat[c] * invP;
    }
    dx[o] = v * g * invstd[c];
  }
}

// ---------------------------------------------------------------------------------------------------------
// MixedOp tail (model_search.py:41-55) + edge accumulation (model_search.py:87):
//   out[p][4i+0] (+)= beta * sum_k w_k * branch_k[p][i]
//   out[p][4i+g] (+)= beta * bypass[p'][g*Cq+i]         g = 1..3   (bypass = x, or MaxPool2d(2,2)(x) in a reduction edge)
// = beta * channel_shuffle(cat(temp1, xtemp2), 4) accumulated into the node state (row stride ldo: the state may
// be a channel slice of the cell's concatenated output).
// ---------------------------------------------------------------------------------------------------------
struct BranchPtrs { const float* p[8]; };
__global__ void mixed_combine_fwd_kernel(BranchPtrs br, int nbr, const float* __restrict__ wts /* [nbr] device */,
                                         const float* __restrict__ beta, const float* __restrict__ x, int B, int Ho, int Wo, int C,
                                         int reduce, float* __restrict__ out, long ldo, int accumulate) {
  const int Cq = C / 4;
  const float bt = beta[0];
  GRID_STRIDE(o, (long)B * Ho * Wo * C) {
    int oc = o % C; long q = o / C;
    int g = oc & 3, i = oc >> 2;
    float v;
    if (g == 0) {
      v = 0.f;
      for (int k = 0; k < nbr; k++) v = fmaf(wts[k], br.p[k][q * Cq + i], v);
    } else {
      const int c = g * Cq + i;
      if (!reduce) v = x[q * C + c];
      else {
        int ox = q % Wo; int oy = (q / Wo) % Ho; int n = q / ((long)Wo * Ho);
        const int H = Ho * 2, W = Wo * 2;
        const float* b0 = x + (((long)n * H + 2 * oy) * W + 2 * ox) * C + c;
        v = fmaxf(fmaxf(b0[0], b0[C]), fmaxf(b0[(long)W * C], b0[(long)W * C + C]));
      }
    }
    v *= bt;
    float* dst = out + q * ldo + oc;
    *dst = accumulate ? *dst + v : v;
  }
}
// gradients w.r.t. the branches:  dbr_k[p][i] = beta * w_k * dout[p][4i]
__global__ void mixed_combine_bwd_branch_kernel(const float* __restrict__ dout, long ldd, const float* __restrict__ wts, int k,
                                                const float* __restrict__ beta, float* __restrict__ dbr, long P, int Cq) {
  const float s = beta[0] * wts[k];
  GRID_STRIDE(o, P * Cq) { long p = o / Cq; int i = o % Cq; dbr[o] = s * dout[p * ldd + 4 * i]; }
}
// gradient w.r.t. x through the bypass channels (first Cq channels get 0 here: they flow through the branches)
__global__ void mixed_combine_bwd_x_kernel(const float* __restrict__ dout, long ldd, const float* __restrict__ beta,
                                           const float* __restrict__ x, float* __restrict__ dx, int B, int H, int W, int C,
                                           int reduce) {
  const int Cq = C / 4;
  const float bt = beta[0];
  GRID_STRIDE(o, (long)B * H * W * C) {
    int c = o % C; long q = o / C;
    float v = 0.f;
    if (c >= Cq) {
      const int g = c / Cq, i = c % Cq;
      if (!reduce) v = bt * dout[q * ldd + 4 * i + g];
      else {
        int ix = q % W; int iy = (q / W) % H; int n = q / ((long)W * H);
        const int Ho = H / 2, Wo = W / 2, oy = iy >> 1, ox = ix >> 1;
        const float* b0 = x + (((long)n * H + 2 * oy) * W + 2 * ox) * C + c;
        const float v00 = b0[0], v01 = b0[C], v10 = b0[(long)W * C], v11 = b0[(long)W * C + C];
        int best = 0; float bv = v00;
        if (v01 > bv) { bv = v01; best = 1; }
        if (v10 > bv) { bv = v10; best = 2; }
        if (v11 > bv) { bv = v11; best = 3; }
        if (best == (iy & 1) * 2 + (ix & 1)) v = bt * dout[(((long)n * Ho + oy) * Wo + ox) * ldd + 4 * i + g];
      }
    }
    dx[o] = v;
  }
}
// scalar reductions: part[blk][0..nbr-1] = sum dout[p][4i]*br_k[p][i];  part[blk][8] = sum over bypass channels of
// dout[p][4i+g]*bypass[p][gCq+i]  (deterministic two-stage; finish on the host side via colsum of [nblk][9])
__global__ __launch_bounds__(256) void mixed_combine_bwd_scalars_kernel(const float* __restrict__ dout, long ldd, BranchPtrs br,
                                                                        int nbr, const float* __restrict__ x, int B, int Ho, int Wo,
                                                                        int C, int reduce, long per_blk, float* __restrict__ part) {
  // thread = (pixel q, channel i of the first quarter): the four shuffled output channels 4i..4i+3 are one float4 of dout
  // (g = 0: the weighted branch sum of channel i; g = 1..3: bypass channels g*Cq + i), so every lane does the same work
  // and the branch / bypass reads are contiguous over i.
  __shared__ float red[9][4];
  const int Cq = C / 4;
  const long P = (long)B * Ho * Wo;
  const long q0 = blockIdx.x * per_blk, q1 = min(P, q0 + per_blk);
  const bool vec = ((ldd & 3) == 0) && (((uintptr_t)dout & 15) == 0);
  float acc[9];
#pragma unroll
  for (int k = 0; k < 9; k++) acc[k] = 0.f;
  for (long o = q0 * Cq + threadIdx.x; o < q1 * Cq; o += blockDim.x) {
    const int i = o % Cq; const long q = o / Cq;
    float d0, d1, d2, d3;
    if (vec) {
      const f32x4 dv = *(const f32x4*)(dout + q * ldd + 4 * i);
      d0 = dv[0]; d1 = dv[1]; d2 = dv[2]; d3 = dv[3];
    } else {
      const float* dp = dout + q * ldd + 4 * i;
      d0 = dp[0]; d1 = dp[1]; d2 = dp[2]; d3 = dp[3];
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (k < nbr) acc[k] = fmaf(d0, br.p[k][o], acc[k]);
    float v1, v2, v3;
    if (!reduce) {
      const float* xp = x + q * C + i;
      v1 = xp[Cq]; v2 = xp[2 * Cq]; v3 = xp[3 * Cq];
    } else {
      const int ox = q % Wo, oy = (q / Wo) % Ho, n = q / ((long)Wo * Ho);
      const int H = Ho * 2, W = Wo * 2;
      const float* b0 = x + (((long)n * H + 2 * oy) * W + 2 * ox) * C + i;
      const long rs = (long)W * C;
      v1 = fmaxf(fmaxf(b0[Cq], b0[C + Cq]), fmaxf(b0[rs + Cq], b0[rs + C + Cq]));
      v2 = fmaxf(fmaxf(b0[2 * Cq], b0[C + 2 * Cq]), fmaxf(b0[rs + 2 * Cq], b0[rs + C + 2 * Cq]));
      v3 = fmaxf(fmaxf(b0[3 * Cq], b0[C + 3 * Cq]), fmaxf(b0[rs + 3 * Cq], b0[rs + C + 3 * Cq]));
    }
    acc[8] = fmaf(d1, v1, fmaf(d2, v2, fmaf(d3, v3, acc[8])));
  }
#pragma unroll
  for (int k = 0; k < 9; k++) {
    float s = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x < 9) part[(long)blockIdx.x * 9 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}
// dw[k] = beta * S_k ;  dbeta = sum_k w_k S_k + S_8   (one wave, strided over the partials)
__global__ __launch_bounds__(64) void mixed_combine_bwd_scalars_finish_kernel(const float* __restrict__ part, int nblk, int nbr,
                                                        const float* __restrict__ wts, const float* __restrict__ beta,
                                                        float* __restrict__ dw, float* __restrict__ dbeta) {
  double S[9];
#pragma unroll
  for (int k = 0; k < 9; k++) S[k] = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64)
#pragma unroll
    for (int k = 0; k < 9; k++) S[k] += part[(long)b * 9 + k];
#pragma unroll
  for (int k = 0; k < 9; k++) S[k] = wave_sum_d(S[k]);
  if (threadIdx.x != 0) return;
  double db = S[8];
  for (int k = 0; k < nbr; k++) { dw[k] = (float)(beta[0] * S[k]); db += wts[k] * S[k]; }
  dbeta[0] = (float)db;
}

// y[p][c0:c0+Cn] of a strided destination = x dense   (channel concat of the two FactorizedReduce halves)
__global__ void copy_into_channels_kernel(const float* __restrict__ x, int Cn, float* __restrict__ y, long ldy, int c0, long P) {
  GRID_STRIDE(o, P * Cn) { long p = o / Cn; int c = o % Cn; y[p * ldy + c0 + c] = x[o]; }
}

// blocks of the float4 depthwise weight-gradient kernels: one per CU -- their epilogue (K*K*4 wave reductions per wave)
// costs as much as ~10 items of the main loop, so fewer, longer-running waves win over occupancy
static int dw_wgrad_blocks(long items) {
  long nb = (items + 255) / 256;
  if (nb > 512) nb = 512;                               // two resident workgroups per CU (227 VGPRs at K = 5)
  return (int)(nb < 1 ? 1 : nb);
}
static int red_blocks(long P, long* per_blk) {
  long nb = (P + 511) / 512;
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  *per_blk = (P + nb - 1) / nb;
  return (int)((P + *per_blk - 1) / *per_blk);
}

template <bool TR>
static int launch_pwconv_small(const float* in, const float* w, const float* mask, float* out, long P, int NOUT, int NIN, int relu_in,
                               hipStream_t st) {
  long nb = (P + 255) / 256;
  if (nb > 4096) nb = 4096;
  size_t lds = (size_t)NOUT * NIN * sizeof(float);
#define PW_CASE(N) case N: hipLaunchKernelGGL((pwconv_small_kernel<N, TR>), dim3((int)nb), dim3(256), lds, st, in, w, mask, out, P, NIN, relu_in); break;
  switch (NOUT) {
    PW_CASE(4) PW_CASE(8) PW_CASE(16) PW_CASE(32) PW_CASE(64)
    default: lct_set_error("pwconv_small: unsupported output width %d", NOUT); return LCT_EUNSUPPORTED;
  }
#undef PW_CASE
  return lct_check_launch("pwconv_small");
}

extern "C" {

int lctvqa_slice_channels(const float* x, long ldx, int c0, float* y, int Cn, long P, void* stream) {
  hipLaunchKernelGGL(slice_channels_kernel, dim3(gs_blocks(P * Cn)), dim3(256), 0, (hipStream_t)stream, x, ldx, c0, y, Cn, P);
  return lct_check_launch("slice_channels");
}
int lctvqa_slice_channels_bwd(const float* dy, int Cn, int c0, float* dx, int C, long P, void* stream) {
  hipLaunchKernelGGL(slice_channels_bwd_kernel, dim3(gs_blocks(P * C)), dim3(256), 0, (hipStream_t)stream, dy, Cn, c0, dx, C, P);
  return lct_check_launch("slice_channels_bwd");
}
int lctvqa_copy_into_channels(const float* x, int Cn, float* y, long ldy, int c0, long P, void* stream) {
  hipLaunchKernelGGL(copy_into_channels_kernel, dim3(gs_blocks(P * Cn)), dim3(256), 0, (hipStream_t)stream, x, Cn, y, ldy, c0, P);
  return lct_check_launch("copy_into_channels");
}

int lctvqa_dwconv_fwd(const float* x, const float* w, float* y, int B, int H, int W, int C, int K, int stride, int pad, int dil,
                      int relu_in, void* stream) {
  LCT_REQUIRE(K >= 1 && stride >= 1 && dil >= 1, "dwconv: bad args");
  int Ho = (H + 2 * pad - dil * (K - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const bool pow2q = C % 4 == 0 && ((C / 4) & (C / 4 - 1)) == 0 && C <= 256;
  if (pow2q && (K == 3 || K == 5 || K == 7) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && (long)B * H * W * C < (1L << 31)) {
    const dim3 grid(gs_blocks((long)B * Ho * Wo * (C / 4), 2048));
    const size_t lds = (size_t)K * K * (C / 4) * sizeof(f32x4);
    hipStream_t st = (hipStream_t)stream;
    if (K == 3) hipLaunchKernelGGL(dwconv_fwd_vecK_kernel<3>, grid, dim3(256), lds, st, x, w, y, B, H, W, C, stride, pad, dil, Ho, Wo, relu_in);
    else if (K == 5) hipLaunchKernelGGL(dwconv_fwd_vecK_kernel<5>, grid, dim3(256), lds, st, x, w, y, B, H, W, C, stride, pad, dil, Ho, Wo, relu_in);
    else hipLaunchKernelGGL(dwconv_fwd_vecK_kernel<7>, grid, dim3(256), lds, st, x, w, y, B, H, W, C, stride, pad, dil, Ho, Wo, relu_in);
  } else if (C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0)
    hipLaunchKernelGGL(dwconv_fwd_vec_kernel, dim3(gs_blocks((long)B * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, w, y,
                       B, H, W, C, K, stride, pad, dil, Ho, Wo, relu_in);
  else
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(gs_blocks((long)B * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, w, y, B, H, W,
                       C, K, stride, pad, dil, Ho, Wo, relu_in);
  return lct_check_launch("dwconv_fwd");
}
int lctvqa_dwconv_bwd_input(const float* dy, const float* w, const float* x, float* dx, int B, int H, int W, int C, int K,
                            int stride, int pad, int dil, int relu_in, void* stream) {
  int Ho = (H + 2 * pad - dil * (K - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const bool pow2q = C % 4 == 0 && ((C / 4) & (C / 4 - 1)) == 0 && C <= 256;
  if (pow2q && (K == 3 || K == 5 || K == 7) && (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx) & 15) == 0 &&
      (long)B * H * W * C < (1L << 31)) {
    const dim3 grid(gs_blocks((long)B * H * W * (C / 4), 2048));
    const size_t lds = (size_t)K * K * (C / 4) * sizeof(f32x4);
    hipStream_t st = (hipStream_t)stream;
    if (K == 3) hipLaunchKernelGGL(dwconv_bwd_input_vecK_kernel<3>, grid, dim3(256), lds, st, dy, w, x, dx, B, H, W, C, stride, pad, dil, Ho, Wo, relu_in);
    else if (K == 5) hipLaunchKernelGGL(dwconv_bwd_input_vecK_kernel<5>, grid, dim3(256), lds, st, dy, w, x, dx, B, H, W, C, stride, pad, dil, Ho, Wo, relu_in);
    else hipLaunchKernelGGL(dwconv_bwd_input_vecK_kernel<7>, grid, dim3(256), lds, st, dy, w, x, dx, B, H, W, C, stride, pad, dil, Ho, Wo, relu_in);
  } else if (C % 4 == 0 && (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx) & 15) == 0)
    hipLaunchKernelGGL(dwconv_bwd_input_vec_kernel, dim3(gs_blocks((long)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, w,
                       x, dx, B, H, W, C, K, stride, pad, dil, Ho, Wo, relu_in);
  else
    hipLaunchKernelGGL(dwconv_bwd_input_kernel, dim3(gs_blocks((long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy, w, x, dx,
                       B, H, W, C, K, stride, pad, dil, Ho, Wo, relu_in);
  return lct_check_launch("dwconv_bwd_input");
}
long lctvqa_dwconv_bwd_weight_scratch_elems(int B, int H, int W, int C, int K, int stride, int pad, int dil) {
  int Ho = (H + 2 * pad - dil * (K - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  long per;
  int nb = red_blocks((long)B * Ho * Wo, &per);
  return (long)nb * C * K * K;
}
int lctvqa_dwconv_bwd_weight(const float* dy, const float* x, float* dw, float* scratch, int B, int H, int W, int C, int K,
                             int stride, int pad, int dil, int relu_in, void* stream) {
  int Ho = (H + 2 * pad - dil * (K - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  long per;
  int nb = red_blocks((long)B * Ho * Wo, &per);
  const bool vec = (C % 4 == 0) && C <= 64 && ((C / 4) & (C / 4 - 1)) == 0 && (K == 3 || K == 5 || K == 7) &&
                   (((uintptr_t)dy | (uintptr_t)x) & 15) == 0 && (long)B * H * W * C < (1L << 31);
  if (vec) {            // at most nb blocks and the same part[blk][tap*C + c] layout, grid-stride over (pixel, channel quad) items
    nb = min(nb, dw_wgrad_blocks((long)B * Ho * Wo * (C / 4)));
    if (K == 3)
      hipLaunchKernelGGL(dwconv_bwd_weight_vec_kernel<3>, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, x, scratch, B, H, W, C, stride,
                         pad, dil, Ho, Wo, relu_in);
    else if (K == 5)
      hipLaunchKernelGGL(dwconv_bwd_weight_vec_kernel<5>, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, x, scratch, B, H, W, C, stride,
                         pad, dil, Ho, Wo, relu_in);
    else
      hipLaunchKernelGGL(dwconv_bwd_weight_vec_kernel<7>, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, x, scratch, B, H, W, C, stride,
                         pad, dil, Ho, Wo, relu_in);
  } else {
    hipLaunchKernelGGL(dwconv_bwd_weight_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, x, scratch, B, H, W, C, K, stride,
                       pad, dil, Ho, Wo, relu_in, per);
  }
  int rc = lct_check_launch("dwconv_bwd_weight");
  if (rc) return rc;
  hipLaunchKernelGGL(dwconv_bwd_weight_finish_kernel, dim3(C * K * K), dim3(64), 0, (hipStream_t)stream, scratch, nb, C, K * K, dw);
  return lct_check_launch("dwconv_bwd_weight_finish");
}

// forward (transpose = 0: x [P][Cin], w [Cout][Cin] -> y [P][Cout], relu_in on x) and input gradient (transpose = 1:
// dy [P][Cout], the same w -> dx [P][Cin], masked by relu_mask_x > 0 when given) of a 1x1 conv with few channels
int lctvqa_pwconv_small_supported(long P, int Cout, int Cin) {
  const int ok_out[] = {4, 8, 16, 32, 64};
  bool o = false, i = false;
  for (int k = 0; k < 5; k++) { o |= (Cout == ok_out[k]); i |= (Cin == ok_out[k]); }
  return P >= 1 && o && i;
}
int lctvqa_pwconv_small(const float* in, const float* w, const float* relu_mask_x, float* out, long P, int Cout, int Cin, int relu_in,
                        int transpose, void* stream) {
  LCT_REQUIRE(lctvqa_pwconv_small_supported(P, Cout, Cin), "pwconv_small: unsupported P=%ld Cout=%d Cin=%d", P, Cout, Cin);
  LCT_REQUIRE((((uintptr_t)in | (uintptr_t)out | (uintptr_t)relu_mask_x) & 15) == 0, "pwconv_small: buffers must be 16-B aligned");
  if (transpose) return launch_pwconv_small<true>(in, w, relu_mask_x, out, P, Cin, Cout, 0, (hipStream_t)stream);
  return launch_pwconv_small<false>(in, w, nullptr, out, P, Cout, Cin, relu_in, (hipStream_t)stream);
}

long lctvqa_pwconv_wgrad_small_scratch_elems(long P, int Cout, int Cin) {
  long per;
  int nb = red_blocks(P, &per);
  return (long)nb * Cout * Cin;
}
// the register-blocked form: which input block size, 0 = not this shape
static int pw_wgrad_rbi(int Cout, int Cin) {
  if (Cout % 4 || Cin % 16 || Cout > 64 || Cin > 256) return 0;
  for (int rbi = 4; rbi <= 16; rbi *= 2)
    if (Cin % rbi == 0 && (Cout / 4) * (Cin / rbi) <= 256) return rbi;
  return 0;
}
static int pw_wgrad_tp(int Cout, int Cin) {        // rows staged at a time: about 32 KB of LDS
  int tp = 8192 / (Cout + Cin);
  return tp > 256 ? 256 : (tp < 8 ? 8 : tp);
}
int lctvqa_pwconv_wgrad_small_supported(long P, int Cout, int Cin) {
  if (P >= 1 && pw_wgrad_rbi(Cout, Cin)) return 1;
  return P >= 1 && Cout >= 1 && Cin >= 1 && (long)Cout * Cin <= 4096 && (Cout + Cin) * 256 * 4 <= 60 * 1024;
}
int lctvqa_pwconv_wgrad_small(const float* dy, const float* x, float* dw, float* scratch, long P, int Cout, int Cin, int relu_in,
                              void* stream) {
  LCT_REQUIRE(lctvqa_pwconv_wgrad_small_supported(P, Cout, Cin), "pwconv_wgrad_small: unsupported P=%ld Cout=%d Cin=%d", P, Cout, Cin);
  long per;
  int nb = red_blocks(P, &per);
  const bool quad = (Cin == 4 || Cin == 8 || Cin == 16) && (Cout == 4 || Cout == 8 || Cout == 16) && P * (Cout / 4) < (1L << 31) &&
                    (((uintptr_t)dy | (uintptr_t)x) & 15) == 0;
  if (quad) {           // at most nb blocks (the scratch is sized for nb), same part layout
    nb = (int)min((long)nb, (P * (Cout / 4) + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (Cin == 4) hipLaunchKernelGGL(pwconv_wgrad_quad_kernel<4>, dim3(nb), dim3(256), 0, st, dy, x, scratch, (int)P, Cout, relu_in);
    else if (Cin == 8) hipLaunchKernelGGL(pwconv_wgrad_quad_kernel<8>, dim3(nb), dim3(256), 0, st, dy, x, scratch, (int)P, Cout, relu_in);
    else hipLaunchKernelGGL(pwconv_wgrad_quad_kernel<16>, dim3(nb), dim3(256), 0, st, dy, x, scratch, (int)P, Cout, relu_in);
  } else if (const int rbi = pw_wgrad_rbi(Cout, Cin); rbi && (((uintptr_t)dy | (uintptr_t)x) & 15) == 0) {
    const int tp = pw_wgrad_tp(Cout, Cin);
    const int nbt = (Cout / 4) * (Cin / rbi), pl = 256 / nbt;
    size_t lds = (size_t)tp * (Cout + Cin) * sizeof(float);
    const size_t red = pl > 1 ? (size_t)pl * nbt * rbi * 4 * sizeof(float) : 0;
    if (red > lds) lds = red;
    hipStream_t st = (hipStream_t)stream;
    if (rbi == 4) hipLaunchKernelGGL(pwconv_wgrad_blk_kernel<4>, dim3(nb), dim3(256), lds, st, dy, x, scratch, P, Cout, Cin, relu_in, per, tp);
    else if (rbi == 8) hipLaunchKernelGGL(pwconv_wgrad_blk_kernel<8>, dim3(nb), dim3(256), lds, st, dy, x, scratch, P, Cout, Cin, relu_in, per, tp);
    else hipLaunchKernelGGL(pwconv_wgrad_blk_kernel<16>, dim3(nb), dim3(256), lds, st, dy, x, scratch, P, Cout, Cin, relu_in, per, tp);
  } else {
    LCT_REQUIRE((long)Cout * Cin <= 4096 && (Cout + Cin) * 256 * 4 <= 60 * 1024, "pwconv_wgrad_small: unaligned operands for Cout=%d Cin=%d", Cout, Cin);
    size_t lds = (size_t)(Cout + Cin) * 256 * sizeof(float);
    hipLaunchKernelGGL(pwconv_wgrad_small_kernel, dim3(nb), dim3(256), lds, (hipStream_t)stream, dy, x, scratch, P, Cout, Cin, relu_in, per);
  }
  int rc = lct_check_launch("pwconv_wgrad_small");
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_partials_kernel, dim3(Cout * Cin), dim3(64), 0, (hipStream_t)stream, scratch, nb, Cout * Cin, dw);
  return lct_check_launch("pwconv_wgrad_small_finish");
}

int lctvqa_pool3x3_fwd(const float* x, float* y, int B, int H, int W, int C, int stride, int kind, void* stream) {
  int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  if (C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0)
    hipLaunchKernelGGL(pool3x3_fwd_vec_kernel, dim3(gs_blocks((long)B * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W,
                       C / 4, stride, Ho, Wo, kind);
  else
    hipLaunchKernelGGL(pool3x3_fwd_kernel, dim3(gs_blocks((long)B * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C,
                       stride, Ho, Wo, kind);
  return lct_check_launch("pool3x3_fwd");
}
int lctvqa_pool3x3_bwd(const float* dy, const float* x, float* dx, int B, int H, int W, int C, int stride, int kind, void* stream) {
  int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  if (C % 4 == 0 && (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx) & 15) == 0)
    hipLaunchKernelGGL(pool3x3_bwd_vec_kernel, dim3(gs_blocks((long)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, x, dx,
                       B, H, W, C, stride, Ho, Wo, kind);
  else
    hipLaunchKernelGGL(pool3x3_bwd_kernel, dim3(gs_blocks((long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, B, H, W,
                       C, stride, Ho, Wo, kind);
  return lct_check_launch("pool3x3_bwd");
}

int lctvqa_subsample2_fwd(const float* x, float* y, int B, int H, int W, int C, int off, int relu, void* stream) {
  LCT_REQUIRE(!(H & 1) && !(W & 1) && (off == 0 || off == 1), "subsample2: need even H,W and off in {0,1}");
  hipLaunchKernelGGL(subsample2_fwd_kernel, dim3(gs_blocks((long)B * (H / 2) * (W / 2) * C)), dim3(256), 0, (hipStream_t)stream, x, y,
                     B, H, W, C, off, relu);
  return lct_check_launch("subsample2_fwd");
}
int lctvqa_subsample2_bwd(const float* dy0, const float* dy1, const float* x, float* dx, int B, int H, int W, int C, int relu,
                          void* stream) {
  hipLaunchKernelGGL(subsample2_bwd_kernel, dim3(gs_blocks((long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy0, dy1, x, dx,
                     B, H, W, C, relu);
  return lct_check_launch("subsample2_bwd");
}

// scratch: doubles; elems = lctvqa_bn_scratch_doubles(P, C)
long lctvqa_bn_scratch_doubles(long P, int C) {
  long per;
  int nb = red_blocks(P, &per);
  return (long)nb * 2 * C;
}
// training != 0: batch statistics (+ running update when run_mean != NULL, + 1 on the int64 batches_tracked counter when it
// is given); else running statistics.
// writes mean[C], invstd[C] (saved for backward) and y.
int lctvqa_batchnorm_fwd(const float* x, long ldx, float* y, long ldy, long P, int C, const float* gamma, const float* beta,
                         float* run_mean, float* run_var, long* batches_tracked, float* mean, float* invstd, double* scratch,
                         int training, float momentum, float eps, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  LCT_REQUIRE(C >= 1 && C <= 256 && P >= 1, "batchnorm: need 1 <= C <= 256");
  if (training) {
    long per;
    int nb = red_blocks(P, &per);
    if (bn_vec_ok(x, ldx, x, ldx, C))
      hipLaunchKernelGGL(bn_sums_vec_kernel<0>, dim3(nb), dim3(256), 0, st, x, ldx, x, ldx, (const float*)nullptr, (const float*)nullptr, P, C, per, scratch);
    else
      hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nb), dim3(256), 0, st, x, ldx, P, C, per, scratch);
    int rc = lct_check_launch("bn_stats_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3(C), dim3(64), 0, st, scratch, nb, C, P, eps, momentum, mean, invstd, run_mean,
                       run_var, batches_tracked);
  } else {
    LCT_REQUIRE(run_mean && run_var, "batchnorm eval: running statistics required");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, run_mean, run_var, eps, C, mean, invstd);
  }
  int rc = lct_check_launch("bn_stats");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(gs_blocks(P * C)), dim3(256), 0, st, x, ldx, mean, invstd, gamma, beta, y, ldy, P, C);
  return lct_check_launch("bn_apply");
}
// dx dense [P][C]; dgamma/dbeta [C] (may be NULL when gamma is NULL); sums: 2*C floats of scratch
int lctvqa_batchnorm_bwd(const float* dy, long lddy, const float* x, long ldx, const float* mean, const float* invstd,
                         const float* gamma, float* dx, float* dgamma, float* dbeta, float* sums, double* scratch, long P, int C,
                         int training, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  LCT_REQUIRE(C >= 1 && C <= 256, "batchnorm_bwd: need 1 <= C <= 256");
  long per;
  int nb = red_blocks(P, &per);
  if (bn_vec_ok(dy, lddy, x, ldx, C) && (((uintptr_t)mean | (uintptr_t)invstd) & 15) == 0)
    hipLaunchKernelGGL(bn_sums_vec_kernel<1>, dim3(nb), dim3(256), 0, st, dy, lddy, x, ldx, mean, invstd, P, C, per, scratch);
  else
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nb), dim3(256), 0, st, dy, lddy, x, ldx, mean, invstd, P, C, per, scratch);
  int rc = lct_check_launch("bn_bwd_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(C), dim3(64), 0, st, scratch, nb, C, sums, sums + C);
  rc = lct_check_launch("bn_bwd_finish");
  if (rc) return rc;
  if (dgamma) {
    hipMemcpyAsync(dbeta, sums, C * sizeof(float), hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(dgamma, sums + C, C * sizeof(float), hipMemcpyDeviceToDevice, st);
  }
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(gs_blocks(P * C)), dim3(256), 0, st, dy, lddy, x, ldx, mean, invstd, gamma, sums,
                     sums + C, dx, P, C, training);
  return lct_check_launch("bn_bwd_apply");
}

// branches: array of nbr (<= 8) device pointers (host array); wts: device [nbr]; beta: device scalar
int lctvqa_mixed_combine_fwd(const float* const* branches, int nbr, const float* wts, const float* beta, const float* x, int B,
                             int Ho, int Wo, int C, int reduce, float* out, long ldo, int accumulate, void* stream) {
  LCT_REQUIRE(nbr >= 0 && nbr <= 8 && C % 4 == 0, "mixed_combine: need nbr <= 8 and C %% 4 == 0");
  BranchPtrs bp;
  for (int k = 0; k < 8; k++) bp.p[k] = k < nbr ? branches[k] : nullptr;
  hipLaunchKernelGGL(mixed_combine_fwd_kernel, dim3(gs_blocks((long)B * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, bp, nbr, wts,
                     beta, x, B, Ho, Wo, C, reduce, out, ldo, accumulate);
  return lct_check_launch("mixed_combine_fwd");
}
int lctvqa_mixed_combine_bwd_branch(const float* dout, long ldd, const float* wts, int k, const float* beta, float* dbr, long P, int Cq,
                                    void* stream) {
  hipLaunchKernelGGL(mixed_combine_bwd_branch_kernel, dim3(gs_blocks(P * Cq)), dim3(256), 0, (hipStream_t)stream, dout, ldd, wts, k,
                     beta, dbr, P, Cq);
  return lct_check_launch("mixed_combine_bwd_branch");
}
int lctvqa_mixed_combine_bwd_x(const float* dout, long ldd, const float* beta, const float* x, float* dx, int B, int H, int W, int C,
                               int reduce, void* stream) {
  hipLaunchKernelGGL(mixed_combine_bwd_x_kernel, dim3(gs_blocks((long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dout, ldd,
                     beta, x, dx, B, H, W, C, reduce);
  return lct_check_launch("mixed_combine_bwd_x");
}
long lctvqa_mixed_combine_scratch_elems(int B, int Ho, int Wo) {
  long per;
  int nb = red_blocks((long)B * Ho * Wo, &per);
  return (long)nb * 9;
}
int lctvqa_mixed_combine_bwd_scalars(const float* dout, long ldd, const float* const* branches, int nbr, const float* wts,
                                     const float* beta, const float* x, int B, int Ho, int Wo, int C, int reduce, float* dw,
                                     float* dbeta, float* scratch, void* stream) {
  BranchPtrs bp;
  for (int k = 0; k < 8; k++) bp.p[k] = k < nbr ? branches[k] : nullptr;
  long per;
  int nb = red_blocks((long)B * Ho * Wo, &per);
  hipLaunchKernelGGL(mixed_combine_bwd_scalars_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dout, ldd, bp, nbr, x, B, Ho, Wo, C,
                     reduce, per, scratch);
  int rc = lct_check_launch("mixed_combine_bwd_scalars");
  if (rc) return rc;
  hipLaunchKernelGGL(mixed_combine_bwd_scalars_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, nb, nbr, wts, beta, dw,
                     dbeta);
  return lct_check_launch("mixed_combine_bwd_scalars_finish");
}

}  // extern "C"
