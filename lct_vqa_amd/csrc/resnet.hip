// Pieces of a FROZEN ResNet-152 image trunk that the VGG/DARTS kernels do not already provide (BASELINE.json configs[2]:
// "ResNet-152+GRU"; the reference itself has no ResNet -- SURVEY.md section 0.1 -- so the architecture is torchvision's published
// resnet152 (Bottleneck, stride on the 3x3 convolution), restated in oracle/resnet_oracle.py).
//
// A frozen trunk runs its BatchNorms on running statistics, i.e. as a per-channel affine map that FOLDS into the preceding
// convolution: w'[co] = w[co] * gamma[co] / sqrt(var[co] + eps), b'[co] = beta[co] - mean[co] * gamma[co] / sqrt(var[co] + eps).
// With that every layer of the trunk is an existing kernel with its fused bias + ReLU (+ residual) epilogue:
//   3x3 convolutions   -> conv3x3 (Winograd F(4x4,3x3) / direct MFMA), stride 2 = stride 1 followed by a 2x subsample
//   1x1 convolutions   -> the strided fp32 MFMA GEMM on NHWC rows, epilogue relu(acc + b' + identity)
//   MaxPool2d(3, 2, 1) -> pool3x3
// and this file adds the 7x7 / stride-2 stem, the weight folding and the global average pool.
#include "common.h"

namespace {

// w' = w * scale[co], shift[co] = beta - mean * scale, scale = gamma / sqrt(var + eps);  w [Cout][K] row-major
__global__ void fold_bn_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps, int Cout, long K,
                               float* __restrict__ w_out, float* __restrict__ shift) {
  const long total = (long)Cout * K;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int co = (int)(o / K);
    const float sc = gamma[co] / sqrtf(var[co] + eps);
    w_out[o] = w[o] * sc;
    if (o % K == 0) shift[co] = beta[co] - mean[co] * sc;
  }
}

// Stem: Conv2d(3, 64, 7, stride 2, padding 3) + bias + ReLU, fp32 NCHW image -> NHWC.  One workgroup per output row; the
// 7 x 3 zero-padded input rows and the [147][64] weights sit in LDS; lane = output channel, a wave walks 8 output pixels at
// a time (inputs as wave-uniform 16-byte broadcasts).  147 MACs per output: VALU work, HBM-write bound beyond that.
__global__ __launch_bounds__(256) void conv7x7_s2_stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y, int B, int H,
                                                              int W, int relu) {
  constexpr int PX = 8, CO = 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int WP = (W + 6 + 2 * PX + 7 + 3) & ~3;                  // padded row: index i <-> input x = i - 3; tail over-reads zeros
  float* rows = smem;                                            // [21 (c, ky)][WP]
  float* wsm = smem + 21 * WP;                                   // [147 (c, ky, kx)][64]
  const int R = blockIdx.x, n = R / Ho, oy = R % Ho;
  for (int i = threadIdx.x; i < 21 * WP; i += blockDim.x) {
    const int cr = i / WP, xx = i % WP - 3;
    const int c = cr / 7, iy = 2 * oy - 3 + cr % 7;
    const bool ok = iy >= 0 && iy < H && xx >= 0 && xx < W;
    rows[i] = ok ? x[(((long)n * 3 + c) * H + iy) * W + xx] : 0.f;
  }
  for (int i = threadIdx.x; i < 147 * CO; i += blockDim.x) wsm[i] = w[(i & 63) * 147 + (i >> 6)];   // [co][c][ky][kx] -> [k][co]
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const float bv = bias ? bias[lane] : 0.f;
  for (int x0 = wave * PX; x0 < Wo; x0 += nw * PX) {
    float a[PX];
#pragma unroll
    for (int i = 0; i < PX; i++) a[i] = bv;
#pragma unroll 1
    for (int cr = 0; cr < 21; cr++) {
      // outputs x0 .. x0+7 read inputs 2*x0 - 3 .. 2*x0 + 17 -> row indices 2*x0 .. 2*x0 + 20 (24 floats, 16-byte aligned)
      const float* row = rows + cr * WP + 2 * x0;
      float in[24];
#pragma unroll
      for (int v = 0; v < 6; v++) {
        const f32x4 t = *(const f32x4*)(row + 4 * v);
        in[4 * v] = t[0]; in[4 * v + 1] = t[1]; in[4 * v + 2] = t[2]; in[4 * v + 3] = t[3];
      }
#pragma unroll
      for (int k = 0; k < 7; k++) {
        const float wv = wsm[(cr * 7 + k) * CO + lane];
#pragma unroll
        for (int i = 0; i < PX; i++) a[i] = fmaf(in[2 * i + k], wv, a[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < PX; i++)
      if (x0 + i < Wo) y[(((long)n * Ho + oy) * Wo + x0 + i) * CO + lane] = relu ? fmaxf(a[i], 0.f) : a[i];
  }
}

// dst[n][oy + y][ox + x][:] = src[n][sy + y][sx + x][:] for an h x w window (NHWC, C % 4 == 0): zero-padding a small map into a
// bordered buffer and cropping the interior back out (the 7x7 maps of layer4, ops._conv3x3_small_map)
__global__ void copy_window_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int Hs, int Ws, int Hd, int Wd, int C4,
                                   int sy, int sx, int oy, int ox, int h, int w, int relu) {
  const long total = (long)B * h * w * C4;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int c = (int)(o % C4);
    long r = o / C4;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const long n = r / h;
    f32x4 v = ((const f32x4*)src)[((n * Hs + sy + y) * Ws + sx + x) * C4 + c];
    if (relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    ((f32x4*)dst)[((n * Hd + oy + y) * Wd + ox + x) * C4 + c] = v;
  }
}

// y[n][c] = mean over the H*W pixels of x[n][:, :][c]   (AdaptiveAvgPool2d(1) + flatten)
__global__ void global_avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int HW, int C) {
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < (long)B * C; o += (long)gridDim.x * blockDim.x) {
    const int c = (int)(o % C);
    const long n = o / C;
    float a = 0.f;
    for (int p = 0; p < HW; p++) a += x[(n * HW + p) * C + c];
    y[o] = a / (float)HW;
  }
}

}  // namespace

extern "C" {

int lctvqa_fold_bn(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, float eps, int Cout,
                   long K, float* w_out, float* shift, void* stream) {
  LCT_REQUIRE(Cout > 0 && K > 0, "fold_bn: bad shape");
  long nb = ((long)Cout * K + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(fold_bn_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, w, gamma, beta, mean, var, eps, Cout, K, w_out, shift);
  return lct_check_launch("fold_bn");
}

// fp32 NCHW [B,3,H,W] -> NHWC [B,Ho,Wo,64], Ho = (H - 1) / 2 + 1
int lctvqa_conv7x7_s2_stem(const float* x_nchw, const float* w_oihw, const float* bias, float* y_nhwc, int B, int H, int W, int relu,
                           void* stream) {
  LCT_REQUIRE(B > 0 && H >= 7 && W >= 7, "conv7x7_s2_stem: bad shape");
  const int Ho = (H - 1) / 2 + 1;
  const int WP = (W + 6 + 16 + 7 + 3) & ~3;
  const size_t lds = (size_t)(21 * WP + 147 * 64) * sizeof(float);
  LCT_REQUIRE(lds <= 160 * 1024, "conv7x7_s2_stem: W too large");
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)conv7x7_s2_stem_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(conv7x7_s2_stem_kernel, dim3(B * Ho), dim3(256), lds, (hipStream_t)stream, x_nchw, w_oihw, bias, y_nhwc, B, H, W, relu);
  return lct_check_launch("conv7x7_s2_stem");
}

int lctvqa_copy_window_nhwc(const float* src, float* dst, int B, int Hs, int Ws, int Hd, int Wd, int C, int sy, int sx, int oy, int ox,
                            int h, int w, int relu, void* stream) {
  LCT_REQUIRE(C % 4 == 0 && sy >= 0 && sx >= 0 && oy >= 0 && ox >= 0 && sy + h <= Hs && sx + w <= Ws && oy + h <= Hd && ox + w <= Wd,
              "copy_window_nhwc: window out of range");
  LCT_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "copy_window_nhwc: 16-B alignment");
  long nb = ((long)B * h * w * (C / 4) + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(copy_window_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, src, dst, B, Hs, Ws, Hd, Wd, C / 4, sy, sx, oy, ox,
                     h, w, relu);
  return lct_check_launch("copy_window");
}

int lctvqa_global_avgpool_nhwc(const float* x, float* y, int B, int HW, int C, void* stream) {
  long nb = ((long)B * C + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(global_avgpool_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, x, y, B, HW, C);
  return lct_check_launch("global_avgpool");
}

}  // extern "C"
