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
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void conv7x7_s2_stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, void* __restrict__ yv, int B, int H,
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
      if (x0 + i < Wo) {
        const float v = relu ? fmaxf(a[i], 0.f) : a[i];
        const long o = (((long)n * Ho + oy) * Wo + x0 + i) * CO + lane;
        if (OUT_BF16) { const __bf16 b = (__bf16)v; ((unsigned short*)yv)[o] = __builtin_bit_cast(unsigned short, b); }
        else ((float*)yv)[o] = v;
      }
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


// ------------------------------------------------------------------------------------------------------------------
// BatchNorm2d over dense NHWC rows [P][C] for ANY channel count (the DARTS kernels stop at 256), fused with what follows it in a
// bottleneck: y = relu?( (x - mean) * invstd * gamma + beta (+ residual) ).  Training: batch statistics (fp64 partial sums per
// workgroup, fixed-order finish, running statistics + batch counter updated); eval: running statistics.
// Thread = (row lane, channel quad): a wave reads 64 consecutive quads (1 KB) of a row.
// ------------------------------------------------------------------------------------------------------------------
// part[blk][C][2]: per-workgroup (sum, sum of squares) -- or (sum g, sum g * xhat) for the backward -- of its rows
template <bool BWD>
__global__ __launch_bounds__(256) void bn_wide_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd, long P, int C,
                                                              long per_blk, double* __restrict__ part) {
  __shared__ double red[2][4][64 * 4];
  const int NQ = C / 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long r0 = blockIdx.x * per_blk, r1 = min(P, r0 + per_blk);
  for (int q0 = 0; q0 < NQ; q0 += 64) {                           // 256 channels at a time: lane = quad, wave = row lane
    const int q = q0 + lane;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (q < NQ) {
      f32x4 m = {0.f, 0.f, 0.f, 0.f}, is = m;
      if (BWD) { m = *(const f32x4*)(mean + 4 * q); is = *(const f32x4*)(invstd + 4 * q); }
      // four rows of this wave's share per iteration: their loads are independent and issued together (one load in flight per
      // thread kept the whole chip at 4 MB in flight, about 2 TB/s)
      auto add_row = [&](const f32x4 v, f32x4 g, const f32x4 yo) __attribute__((always_inline)) {
        if (!BWD) {
#pragma unroll
          for (int e = 0; e < 4; e++) { a[e] += (double)v[e]; b[e] += (double)v[e] * (double)v[e]; }
        } else {
          if (y) {
#pragma unroll
            for (int e = 0; e < 4; e++) g[e] = yo[e] > 0.f ? g[e] : 0.f;
          }
#pragma unroll
          for (int e = 0; e < 4; e++) { a[e] += (double)g[e]; b[e] += (double)(g[e] * ((v[e] - m[e]) * is[e])); }
        }
      };
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      long r = r0 + wv;
      for (; r + 12 < r1; r += 16) {
        f32x4 v[4], g[4], yo[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const long o = (r + 4 * u) * C + 4 * q;
          v[u] = *(const f32x4*)(x + o);
          g[u] = BWD ? *(const f32x4*)(dy + o) : z4;
          yo[u] = (BWD && y) ? *(const f32x4*)(y + o) : z4;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) add_row(v[u], g[u], yo[u]);
      }
      for (; r < r1; r += 4) {
        const long o = r * C + 4 * q;
        add_row(*(const f32x4*)(x + o), BWD ? *(const f32x4*)(dy + o) : z4, (BWD && y) ? *(const f32x4*)(y + o) : z4);
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; e++) { red[0][wv][lane * 4 + e] = a[e]; red[1][wv][lane * 4 + e] = b[e]; }
    __syncthreads();
    const int c = threadIdx.x;                                    // 256 channels of this group
    if (q0 * 4 + c < C) {
      const double sa = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
      const double sb = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
      part[((long)blockIdx.x * C + q0 * 4 + c) * 2 + 0] = sa;
      part[((long)blockIdx.x * C + q0 * 4 + c) * 2 + 1] = sb;
    }
  }
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// one wave per channel: forward -> mean, invstd (+ running update); backward -> sums[c] = sum g, sums[C + c] = sum g * xhat
__global__ __launch_bounds__(64) void bn_wide_finish_kernel(const double* __restrict__ part, int nblk, int C, long P, float eps, float momentum,
                                                            int bwd, float* __restrict__ out0, float* __restrict__ out1,
                                                            float* __restrict__ run_mean, float* __restrict__ run_var, long* __restrict__ tracked,
                                                            double* __restrict__ sums64) {
  const int c = blockIdx.x;
  double a = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 64) { a += part[((long)k * C + c) * 2]; b += part[((long)k * C + c) * 2 + 1]; }
  a = wave_sum_f64(a); b = wave_sum_f64(b);
  if (threadIdx.x != 0) return;
  if (sums64) { sums64[2 * c] = a; sums64[2 * c + 1] = b; return; }          // SyncBN: the per-channel sums, for the all-reduce
  if (bwd) { out0[c] = (float)a; out1[c] = (float)b; return; }
  const double m = a / (double)P;
  double var = b / (double)P - m * m;
  if (var < 0.0) var = 0.0;
  out0[c] = (float)m;
  out1[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
    run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * m);
    run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
  }
  if (c == 0 && tracked) tracked[0] += 1;
}
__global__ void bn_wide_eval_stats_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var, float eps, int C,
                                          float* __restrict__ mean, float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { mean[c] = run_mean[c]; invstd[c] = 1.f / sqrtf(run_var[c] + eps); }
}
__global__ void bn_wide_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ res, int relu,
                                     float* __restrict__ y, long P, int C) {
  const int NQ = C / 4;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < P * NQ; o += (long)gridDim.x * blockDim.x) {
    const int q = (int)(o % NQ);
    f32x4 v = ((const f32x4*)x)[o];
    const f32x4 m = *(const f32x4*)(mean + 4 * q), is = *(const f32x4*)(invstd + 4 * q);
    const f32x4 g = *(const f32x4*)(gamma + 4 * q), b = *(const f32x4*)(beta + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = (v[e] - m[e]) * is[e] * g[e] + b[e];
    if (res) v += ((const f32x4*)res)[o];
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
    }
    ((f32x4*)y)[o] = v;
  }
}
// g = dy * (y > 0) (when y is given); dres = g; training: dx = gamma invstd (g - sum g / P - xhat sum(g xhat) / P), eval: gamma invstd g
__global__ void bn_wide_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
                                         const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                         const float* __restrict__ sums, int training, float* __restrict__ dx, float* __restrict__ dres, long P,
                                         int C, double count) {
  const int NQ = C / 4;
  const float invP = (float)(1.0 / count);
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < P * NQ; o += (long)gridDim.x * blockDim.x) {
    const int q = (int)(o % NQ);
    f32x4 g = ((const f32x4*)dy)[o];
    if (y) {
      const f32x4 yo = ((const f32x4*)y)[o];
#pragma unroll
      for (int e = 0; e < 4; e++) g[e] = yo[e] > 0.f ? g[e] : 0.f;
    }
    if (dres) ((f32x4*)dres)[o] = g;
    const f32x4 v = ((const f32x4*)x)[o];
    const f32x4 m = *(const f32x4*)(mean + 4 * q), is = *(const f32x4*)(invstd + 4 * q), ga = *(const f32x4*)(gamma + 4 * q);
    f32x4 r;
    if (training) {
      const f32x4 s0 = *(const f32x4*)(sums + 4 * q), s1 = *(const f32x4*)(sums + C + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; e++) r[e] = ga[e] * is[e] * (g[e] - s0[e] * invP - (v[e] - m[e]) * is[e] * s1[e] * invP);
    } else {
#pragma unroll
      for (int e = 0; e < 4; e++) r[e] = ga[e] * is[e] * g[e];
    }
    ((f32x4*)dx)[o] = r;
  }
}

// dx[n][p][c] = dy[n][c] / HW   (backward of the global average pool)
__global__ void global_avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int HW, int C) {
  const float s = 1.f / (float)HW;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < (long)B * HW * C; o += (long)gridDim.x * blockDim.x) {
    const int c = (int)(o % C);
    const long n = o / ((long)HW * C);
    dx[o] = dy[n * C + c] * s;
  }
}

// Weight gradient of the 7x7 / stride-2 stem: dW[co][c][ky][kx] = sum over (n, oy, ox) of dz[n][oy][ox][co] * x[n][c][2oy-3+ky][2ox-3+kx].
// Persistent workgroups over output rows; lane = output channel, wave w owns the taps k = w, w+4, ... (37 of the 147), one
// register accumulator each; the input rows of an output row sit zero-padded in LDS (wave-uniform broadcast reads).
// part[blk][147][64] per-workgroup partials, summed by the caller (colsum).
template <bool DZ_BF16>
__global__ __launch_bounds__(256) void conv7x7_s2_stem_wgrad_kernel(const float* __restrict__ x, const void* __restrict__ dzv, float* __restrict__ part,
                                                                    int B, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int WP = W + 6;
  float* rows = smem;                                             // [21][WP]: index i <-> input x = i - 3
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[37];
#pragma unroll
  for (int i = 0; i < 37; i++) acc[i] = 0.f;
  for (long R = blockIdx.x; R < (long)B * Ho; R += gridDim.x) {
    const int n = (int)(R / Ho), oy = (int)(R % Ho);
    __syncthreads();
    for (int i = threadIdx.x; i < 21 * WP; i += blockDim.x) {
      const int cr = i / WP, xx = i % WP - 3;
      const int c = cr / 7, iy = 2 * oy - 3 + cr % 7;
      const bool ok = iy >= 0 && iy < H && xx >= 0 && xx < W;
      rows[i] = ok ? x[(((long)n * 3 + c) * H + iy) * W + xx] : 0.f;
    }
    __syncthreads();
    for (int ox = 0; ox < Wo; ox++) {
      const long go = (((long)n * Ho + oy) * Wo + ox) * 64 + lane;
      const float g = DZ_BF16 ? __builtin_bit_cast(float, (unsigned int)((const unsigned short*)dzv)[go] << 16) : ((const float*)dzv)[go];
#pragma unroll
      for (int i = 0; i < 37; i++) {
        const int k = wave + 4 * i;                                // tap index (c, ky, kx) = k / 7 (row cr) and k % 7
        if (k < 147) acc[i] = fmaf(g, rows[(k / 7) * WP + 2 * ox + k % 7], acc[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 37; i++) {
    const int k = wave + 4 * i;
    if (k < 147) part[((long)blockIdx.x * 147 + k) * 64 + lane] = acc[i];
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

static int launch_stem(const float* x_nchw, const float* w_oihw, const float* bias, void* y_nhwc, int B, int H, int W, int relu, int out_bf16,
                       void* stream) {
  LCT_REQUIRE(B > 0 && H >= 7 && W >= 7, "conv7x7_s2_stem: bad shape");
  const int Ho = (H - 1) / 2 + 1;
  const int WP = (W + 6 + 16 + 7 + 3) & ~3;
  const size_t lds = (size_t)(21 * WP + 147 * 64) * sizeof(float);
  LCT_REQUIRE(lds <= 160 * 1024, "conv7x7_s2_stem: W too large");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv7x7_s2_stem_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv7x7_s2_stem_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (out_bf16)
    hipLaunchKernelGGL(conv7x7_s2_stem_kernel<true>, dim3(B * Ho), dim3(256), lds, (hipStream_t)stream, x_nchw, w_oihw, bias, y_nhwc, B, H, W, relu);
  else
    hipLaunchKernelGGL(conv7x7_s2_stem_kernel<false>, dim3(B * Ho), dim3(256), lds, (hipStream_t)stream, x_nchw, w_oihw, bias, y_nhwc, B, H, W, relu);
  return lct_check_launch("conv7x7_s2_stem");
}
// fp32 NCHW [B,3,H,W] -> NHWC [B,Ho,Wo,64] fp32 (or bf16: the _bf16 form), Ho = (H - 1) / 2 + 1
int lctvqa_conv7x7_s2_stem(const float* x_nchw, const float* w_oihw, const float* bias, float* y_nhwc, int B, int H, int W, int relu,
                           void* stream) {
  return launch_stem(x_nchw, w_oihw, bias, y_nhwc, B, H, W, relu, 0, stream);
}
int lctvqa_conv7x7_s2_stem_bf16(const float* x_nchw, const float* w_oihw, const float* bias, void* y_nhwc_bf16, int B, int H, int W, int relu,
                                void* stream) {
  return launch_stem(x_nchw, w_oihw, bias, y_nhwc_bf16, B, H, W, relu, 1, stream);
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

// rows of the fp64 partial buffer of the wide BatchNorm for P rows
int lctvqa_bn_wide_blocks(long P) {
  long nb = (P + 255) / 256;
  if (nb > 2048) nb = 2048;
  return (int)(nb < 1 ? 1 : nb);
}
// y = relu?(bn(x) (+ residual)) on dense [P][C] rows, C % 4 == 0.  part: lctvqa_bn_wide_blocks(P) * C * 2 doubles.
int lctvqa_bn_wide_fwd(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var, long* tracked,
                       const float* residual, int relu, float* y, float* mean, float* invstd, double* part, long P, int C, int training,
                       float momentum, float eps, void* stream) {
  LCT_REQUIRE(C % 4 == 0 && P >= 1, "bn_wide_fwd: need C%%4==0");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (training) {
    const int nb = lctvqa_bn_wide_blocks(P);
    const long per = (P + nb - 1) / nb;
    hipLaunchKernelGGL(bn_wide_partial_kernel<false>, dim3(nb), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, P, C, per, part);
    if ((rc = lct_check_launch("bn_wide_partial"))) return rc;
    hipLaunchKernelGGL(bn_wide_finish_kernel, dim3(C), dim3(64), 0, st, part, nb, C, P, eps, momentum, 0, mean, invstd, run_mean, run_var, tracked,
                       (double*)nullptr);
  } else {
    LCT_REQUIRE(run_mean && run_var, "bn_wide_fwd eval: running statistics required");
    hipLaunchKernelGGL(bn_wide_eval_stats_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, run_mean, run_var, eps, C, mean, invstd);
  }
  if ((rc = lct_check_launch("bn_wide_stats"))) return rc;
  long nbk = (P * (C / 4) + 255) / 256;
  if (nbk > 8192) nbk = 8192;
  hipLaunchKernelGGL(bn_wide_apply_kernel, dim3((int)nbk), dim3(256), 0, st, x, mean, invstd, gamma, beta, residual, relu, y, P, C);
  return lct_check_launch("bn_wide_apply");
}
// y_out: the forward's output when a ReLU followed (its sign is the mask), else NULL; dres (or NULL) receives the gradient of the
// residual input; dgamma / dbeta [C]; sums: 2*C floats of scratch
int lctvqa_bn_wide_bwd(const float* dy, const float* y_out, const float* x, const float* mean, const float* invstd, const float* gamma,
                       float* dx, float* dres, float* dgamma, float* dbeta, float* sums, double* part, long P, int C, int training,
                       void* stream) {
  LCT_REQUIRE(C % 4 == 0 && P >= 1, "bn_wide_bwd: need C%%4==0");
  hipStream_t st = (hipStream_t)stream;
  const int nb = lctvqa_bn_wide_blocks(P);
  const long per = (P + nb - 1) / nb;
  int rc;
  hipLaunchKernelGGL(bn_wide_partial_kernel<true>, dim3(nb), dim3(256), 0, st, x, dy, y_out, mean, invstd, P, C, per, part);
  if ((rc = lct_check_launch("bn_wide_bwd_partial"))) return rc;
  hipLaunchKernelGGL(bn_wide_finish_kernel, dim3(C), dim3(64), 0, st, part, nb, C, P, 0.f, 0.f, 1, sums, sums + C, (float*)nullptr,
                     (float*)nullptr, (long*)nullptr, (double*)nullptr);
  if ((rc = lct_check_launch("bn_wide_bwd_finish"))) return rc;
  (void)hipMemcpyAsync(dbeta, sums, C * sizeof(float), hipMemcpyDeviceToDevice, st);
  (void)hipMemcpyAsync(dgamma, sums + C, C * sizeof(float), hipMemcpyDeviceToDevice, st);
  long nbk = (P * (C / 4) + 255) / 256;
  if (nbk > 8192) nbk = 8192;
  hipLaunchKernelGGL(bn_wide_bwd_apply_kernel, dim3((int)nbk), dim3(256), 0, st, dy, y_out, x, mean, invstd, gamma, sums, training, dx, dres, P, C,
                     (double)P);
  return lct_check_launch("bn_wide_bwd_apply");
}
// ---- SyncBN forms (data-parallel training, SURVEY.md section 8e BN caveat): the same kernels cut at the point where the per-channel
// sums exist, so that the caller can all-reduce them.  sums64 [C][2] fp64: (sum x, sum x^2) or (sum g, sum g xhat) of THIS rank.
int lctvqa_bn_wide_sums(const float* x, const float* dy, const float* y_out, const float* mean, const float* invstd, long P, int C, int bwd,
                        double* part, double* sums64, void* stream) {
  LCT_REQUIRE(C % 4 == 0 && P >= 1 && sums64, "bn_wide_sums: need C%%4==0");
  hipStream_t st = (hipStream_t)stream;
  const int nb = lctvqa_bn_wide_blocks(P);
  const long per = (P + nb - 1) / nb;
  if (bwd) hipLaunchKernelGGL(bn_wide_partial_kernel<true>, dim3(nb), dim3(256), 0, st, x, dy, y_out, mean, invstd, P, C, per, part);
  else hipLaunchKernelGGL(bn_wide_partial_kernel<false>, dim3(nb), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr,
                          (const float*)nullptr, (const float*)nullptr, P, C, per, part);
  int rc = lct_check_launch("bn_wide_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_wide_finish_kernel, dim3(C), dim3(64), 0, st, part, nb, C, P, 0.f, 0.f, bwd, (float*)nullptr, (float*)nullptr,
                     (float*)nullptr, (float*)nullptr, (long*)nullptr, sums64);
  return lct_check_launch("bn_wide_sums");
}
// forward from all-reduced sums over `count` elements per channel: statistics (+ running estimates), then y = relu?(bn(x) (+ residual))
int lctvqa_bn_wide_fwd_from_sums(const float* x, const double* sums64, double count, const float* gamma, const float* beta, float* run_mean,
                                 float* run_var, long* tracked, const float* residual, int relu, float* y, float* mean, float* invstd, long P,
                                 int C, float momentum, float eps, void* stream) {
  LCT_REQUIRE(C % 4 == 0 && P >= 1 && count >= 1.0, "bn_wide_fwd_from_sums: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_wide_finish_kernel, dim3(C), dim3(64), 0, st, sums64, 1, C, (long)count, eps, momentum, 0, mean, invstd, run_mean, run_var,
                     tracked, (double*)nullptr);
  int rc = lct_check_launch("bn_wide_finish");
  if (rc) return rc;
  long nbk = (P * (C / 4) + 255) / 256;
  if (nbk > 8192) nbk = 8192;
  hipLaunchKernelGGL(bn_wide_apply_kernel, dim3((int)nbk), dim3(256), 0, st, x, mean, invstd, gamma, beta, residual, relu, y, P, C);
  return lct_check_launch("bn_wide_apply");
}
// backward from sums: dgamma / dbeta from THIS rank's sums (the data-parallel reducer averages them), dx from the all-reduced ones
int lctvqa_bn_wide_bwd_from_sums(const float* dy, const float* y_out, const float* x, const float* mean, const float* invstd, const float* gamma,
                                 const double* sums_local, const double* sums_global, double count, float* dx, float* dres, float* dgamma,
                                 float* dbeta, float* sums, long P, int C, void* stream) {
  LCT_REQUIRE(C % 4 == 0 && P >= 1 && count >= 1.0, "bn_wide_bwd_from_sums: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_wide_finish_kernel, dim3(C), dim3(64), 0, st, sums_local, 1, C, P, 0.f, 0.f, 1, dbeta, dgamma, (float*)nullptr,
                     (float*)nullptr, (long*)nullptr, (double*)nullptr);
  hipLaunchKernelGGL(bn_wide_finish_kernel, dim3(C), dim3(64), 0, st, sums_global, 1, C, P, 0.f, 0.f, 1, sums, sums + C, (float*)nullptr,
                     (float*)nullptr, (long*)nullptr, (double*)nullptr);
  int rc = lct_check_launch("bn_wide_bwd_finish");
  if (rc) return rc;
  long nbk = (P * (C / 4) + 255) / 256;
  if (nbk > 8192) nbk = 8192;
  hipLaunchKernelGGL(bn_wide_bwd_apply_kernel, dim3((int)nbk), dim3(256), 0, st, dy, y_out, x, mean, invstd, gamma, sums, 1, dx, dres, P, C, count);
  return lct_check_launch("bn_wide_bwd_apply");
}
int lctvqa_global_avgpool_nhwc_bwd(const float* dy, float* dx, int B, int HW, int C, void* stream) {
  long nb = ((long)B * HW * C + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(global_avgpool_bwd_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, dy, dx, B, HW, C);
  return lct_check_launch("global_avgpool_bwd");
}
// workgroups (= partial rows) of the stem weight gradient
int lctvqa_conv7x7_s2_stem_wgrad_blocks(int B, int H) {
  const long rows = (long)B * ((H - 1) / 2 + 1);
  return (int)(rows < 1024 ? rows : 1024);
}
// part [blocks][147][64] floats; dW[co][k] = sum over blocks of part[.][k][co] (the caller reduces and transposes)
int lctvqa_conv7x7_s2_stem_wgrad(const float* x_nchw, const float* dz_nhwc, float* part, int B, int H, int W, void* stream) {
  LCT_REQUIRE(B > 0 && H >= 7 && W >= 7, "conv7x7_s2_stem_wgrad: bad shape");
  const size_t lds = (size_t)21 * (W + 6) * sizeof(float);
  LCT_REQUIRE(lds <= 64 * 1024, "conv7x7_s2_stem_wgrad: W too large");
  hipLaunchKernelGGL(conv7x7_s2_stem_wgrad_kernel<false>, dim3(lctvqa_conv7x7_s2_stem_wgrad_blocks(B, H)), dim3(256), lds, (hipStream_t)stream,
                     x_nchw, (const void*)dz_nhwc, part, B, H, W);
  return lct_check_launch("conv7x7_s2_stem_wgrad");
}
// the same with a bf16 output gradient (the bf16-native trainable trunk, train_bf16.hip)
int lctvqa_conv7x7_s2_stem_wgrad_bf16(const float* x_nchw, const void* dz_nhwc_bf16, float* part, int B, int H, int W, void* stream) {
  LCT_REQUIRE(B > 0 && H >= 7 && W >= 7, "conv7x7_s2_stem_wgrad_bf16: bad shape");
  const size_t lds = (size_t)21 * (W + 6) * sizeof(float);
  LCT_REQUIRE(lds <= 64 * 1024, "conv7x7_s2_stem_wgrad_bf16: W too large");
  hipLaunchKernelGGL(conv7x7_s2_stem_wgrad_kernel<true>, dim3(lctvqa_conv7x7_s2_stem_wgrad_blocks(B, H)), dim3(256), lds, (hipStream_t)stream,
                     x_nchw, dz_nhwc_bf16, part, B, H, W);
  return lct_check_launch("conv7x7_s2_stem_wgrad_bf16");
}

}  // extern "C"
