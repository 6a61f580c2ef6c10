// Backward of the VGG19 conv blocks (trainable trunk: reference models_lct.py:62-63, ImgEncoderFixed with
// pretrained=False; darts-free EF model of config 4).
//
//  * dgrad   = the forward MFMA kernel run on dZ with flipped/transposed weights: only a pack kernel here.
//  * wgrad   = dW[co][ci][tap] = sum_pixels dZ[p][co] * X[p+tap][ci] on v_mfma_f32_32x32x2_f32 with the
//              PIXELS as the K dimension: a workgroup owns a (64 co x 64 ci) block of dW for ALL 9 taps
//              (9 accumulator tiles per wave), walks its share of TH x TW pixel patches, stages the dZ patch
//              and the zero-padded X halo patch HBM->LDS with global_load_lds (rows are NHWC 256-B channel
//              segments: no transposition needed) and reuses the halo for the 9 shifted taps.  Partial sums
//              go to per-split slabs that a second kernel reduces in a fixed order (deterministic, no atomics).
//  * ReLU / MaxPool2d(2,2) backward fused in one HBM-bound pass; bias gradient via the column-sum kernel.
#include "common.h"

__device__ __attribute__((aligned(64))) float g_zero_page_b[64];
#define g_zero_page g_zero_page_b

// packed weights of the dgrad convolution: conv3x3(dZ[.,Cout], W') with W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]
// layout = forward packing for a conv with Cin' = Cout, Cout' = Cin: [Cin/32][(Cout/16)*9][4][32][4]
__global__ void conv3x3_pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin) {
  const long total = (long)Cout * Cin * 9;
  const int S = (Cout / 16) * 9;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    int e = o & 3;
    int j = (o >> 2) & 31;
    int c = (o >> 7) & 3;
    long rest = o >> 9;
    int s = rest % S;
    int nt = rest / S;
    int chunk = s / 9, tap = s % 9;
    int ci = nt * 32 + j;                 // output channel of the dgrad conv
    int co = chunk * 16 + c * 4 + e;      // its input channel
    wp[o] = w[((long)co * Cin + ci) * 9 + (8 - tap)];
  }
}

// dZ = relu'(A) * unpool(dOut):  A [B,H,W,C] post-ReLU activation; dOut [B,H/2,W/2,C] (pool) or [B,H,W,C].
// MaxPool2d routes the gradient to the FIRST maximal element of each 2x2 window (scan order), like PyTorch.
__global__ void relu_pool_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ act,
                                     float* __restrict__ dz, int B, int H, int W, int C, int pool) {
  const int C4 = C / 4;
  if (!pool) {
    const long total = (long)B * H * W * C4;
    for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
      f32x4 a = ((const f32x4*)act)[o], g = ((const f32x4*)dout)[o], r;
#pragma unroll
      for (int k = 0; k < 4; k++) r[k] = a[k] > 0.f ? g[k] : 0.f;
      ((f32x4*)dz)[o] = r;
    }
    return;
  }
  const int Ho = H / 2, Wo = W / 2;
  const long total = (long)B * Ho * Wo * C4;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    int c4 = o % C4;
    long pix = o / C4;
    int ox = pix % Wo;
    int oy = (pix / Wo) % Ho;
    int n = pix / ((long)Wo * Ho);
    const long base = (((long)n * H + 2 * oy) * W + 2 * ox) * C4 + c4;
    const long offs[4] = {0, C4, (long)W * C4, (long)W * C4 + C4};
    f32x4 a[4];
#pragma unroll
    for (int q = 0; q < 4; q++) a[q] = ((const f32x4*)act)[base + offs[q]];
    f32x4 g = ((const f32x4*)dout)[o];
    f32x4 r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int best = 0;
      float bv = a[0][k];
#pragma unroll
      for (int q = 1; q < 4; q++)
        if (a[q][k] > bv) { bv = a[q][k]; best = q; }
#pragma unroll
      for (int q = 0; q < 4; q++) r[q][k] = (q == best && bv > 0.f) ? g[k] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) ((f32x4*)dz)[base + offs[q]] = r[q];
  }
}

// backward of AdaptiveAvgPool2d((7,7)) + flatten(NCHW order):  dy [B, C*49] -> dx NHWC [B,H,W,C]
__global__ void adaptive_avgpool7_flatten_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H,
                                                     int W, int C) {
  long total = (long)B * H * W * C;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    int c = o % C;
    long pix = o / C;
    int ix = pix % W, iy = (pix / W) % H;
    int n = pix / ((long)W * H);
    float a = 0.f;
    // bins (oy,ox) whose window [floor(o*H/7), ceil((o+1)*H/7)) contains (iy,ix)
    for (int oy = 0; oy < 7; oy++) {
      int ys = (oy * H) / 7, ye = ((oy + 1) * H + 6) / 7;
      if (iy < ys || iy >= ye) continue;
      for (int ox = 0; ox < 7; ox++) {
        int xs = (ox * W) / 7, xe = ((ox + 1) * W + 6) / 7;
        if (ix < xs || ix >= xe) continue;
        a += dy[((long)n * C + c) * 49 + oy * 7 + ox] / (float)((ye - ys) * (xe - xs));
      }
    }
    dx[o] = a;
  }
}

// ----------------------------------------------------------------------------------------------------------
// wgrad MFMA kernel
// ----------------------------------------------------------------------------------------------------------
template <int TH, int TW, int RS>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                               float* __restrict__ slab, int Btot, int H, int W, int Cin,
                                                               int Cout, int nsplit, int G) {
  constexpr int NP = TH * TW;                   // pixels per patch = K extent of one patch
  static_assert(NP % 2 == 0, "patch must hold an even number of pixels");
  constexpr int HR = TH + 4;
  constexpr int PX = HR * RS;                   // halo pixels (allocated)
  constexpr int ZCH = NP * 16;                  // 16-B chunks of the dZ tile  [NP][64 co]
  constexpr int XCH = PX * 16;                  // 16-B chunks of the X halo   [PX][64 ci]
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* zt = smem;                             // [NP][64]
  float* xt = smem + NP * 64;                   // [PX][64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, j = lane & 31, h = lane >> 5;
  const int NCT = Cin / 64, MCT = Cout / 64;
  int bid = blockIdx.x;
  const int sp = bid % nsplit; bid /= nsplit;
  const int nt = bid % NCT, mt = bid / NCT;
  const int co0 = mt * 64, ci0 = nt * 64;
  const int CT = W / TW;
  const long Rtot = (long)Btot * H;
  const int npatch = (int)((Rtot + TH - 1) / TH) * CT;
  const int PLIVE = (TH + 2 + G) * RS;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

  for (int pi = sp; pi < npatch; pi += nsplit) {
    const int ct = pi % CT, rt = pi / CT;
    const int R0 = rt * TH, x0 = ct * TW;
    const int n0img = R0 / H, y0 = R0 % H;
    __syncthreads();                            // previous patch fully consumed
    // ---- stage dZ patch: chunk q -> pixel i = q>>4 (patch-linear), 16-B channel chunk q&15 ----
    for (int q0 = 0; q0 < ZCH; q0 += 256) {
      const int q = q0 + tid;
      const int i = q >> 4, c16 = q & 15;
      const int ty = i / TW, tx = i % TW;
      const long R = (long)R0 + ty;
      const bool ok = (q < ZCH) && (R < Rtot);
      const float* src = ok ? dz + (R * W + x0 + tx) * (long)Cout + co0 + c16 * 4 : g_zero_page;
      if (q < ZCH) glds16(src, zt + (q0 + wave * 64) * 4);
    }
    // ---- stage X halo: chunk q -> halo pixel p = q>>4 ----
    for (int q0 = 0; q0 < XCH; q0 += 256) {
      const int q = q0 + tid;
      const int p = q >> 4, c16 = q & 15;
      const int hr = p / RS, hx = p % RS;
      const int xx = x0 - 1 + hx;
      int yv = y0 - 1 + hr, n = n0img;
      while (yv > H) { yv -= (H + 1); n++; }
      const bool ok = (p < PLIVE) && (hx < TW + 2) && (yv >= 0) && (yv < H) && (n < Btot) && (xx >= 0) && (xx < W);
      const float* src = ok ? x + (((long)n * H + yv) * W + xx) * (long)Cin + ci0 + c16 * 4 : g_zero_page;
      if (q0 < PLIVE * 16 && q < XCH) glds16(src, xt + (q0 + wave * 64) * 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- K loop over the patch's pixels, 2 per MFMA ----
#pragma unroll 4
    for (int ks = 0; ks < NP / 2; ks++) {
      const int i = 2 * ks + h;
      const int ty = i / TW, tx = i % TW;
      const int yy = y0 + ty;
      const int g = (yy >= H) + (yy >= 2 * H);
      const float a = zt[i * 64 + wm * 32 + j];
      const float* xb = xt + ((ty + g) * RS + tx) * 64 + wn * 32 + j;
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const float b = xb[((t / 3) * RS + (t % 3)) * 64];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
      }
    }
  }
  // ---- write this split's partial dW block to its slab, OIHW order ----
  float* out = slab + (long)sp * Cout * Cin * 9;
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int ci = ci0 + wn * 32 + j;
      out[((long)co * Cin + ci) * 9 + t] = acc[t][r];
    }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slab, int nsplit, long n, float* __restrict__ out) {
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) {
    float a = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;                 // four slabs per iteration: independent loads in flight
    int s = 0;
    for (; s + 3 < nsplit; s += 4) {
      const float* p = slab + (long)s * n + o;
      a += p[0]; a1 += p[n]; a2 += p[2 * n]; a3 += p[3 * n];
    }
    for (; s < nsplit; s++) a += slab[(long)s * n + o];
    out[o] = (a + a1) + (a2 + a3);
  }
}

// naive wgrad fallback (tiny maps / channel counts the MFMA kernel does not tile)
__global__ void conv3x3_wgrad_direct_kernel(const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ dw,
                                            int B, int H, int W, int Cin, int Cout) {
  const long total = (long)Cout * Cin * 9;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int tap = o % 9, ci = (o / 9) % Cin, co = o / (9L * Cin);
    const int ky = tap / 3, kx = tap % 3;
    float a = 0.f;
    for (int n = 0; n < B; n++)
      for (int y = 0; y < H; y++) {
        const int iy = y + ky - 1;
        if (iy < 0 || iy >= H) continue;
        for (int xx = 0; xx < W; xx++) {
          const int ix = xx + kx - 1;
          if (ix < 0 || ix >= W) continue;
          a = fmaf(dz[(((long)n * H + y) * W + xx) * Cout + co], x[(((long)n * H + iy) * W + ix) * Cin + ci], a);
        }
      }
    dw[o] = a;
  }
}

// first layer (Cin = 3, NCHW image): dW[co][c][ky][kx] and db[co];  lane = co, workgroup = one stacked row,
// per-row partials [rows][28][Cout] reduced by the column-sum kernel (28th column = bias gradient).
__global__ __launch_bounds__(256) void conv3x3_first_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                   float* __restrict__ part, int B, int H, int W, int Cout) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [9][WP] input rows + [4][28][64] wave partials
  const int WP = W + 2;
  const int R = blockIdx.x, n = R / H, yy = R % H;
  for (int i0 = 0; i0 < WP; i0 += blockDim.x) {
    const int i = i0 + threadIdx.x, xx = i - 1;
    if (i < WP) {
#pragma unroll
      for (int cr = 0; cr < 9; cr++) {
        const int c = cr / 3, iy = yy + (cr % 3) - 1;
        const bool ok = (iy >= 0) && (iy < H) && (xx >= 0) && (xx < W);
        smem[cr * WP + i] = ok ? x[(((long)n * 3 + c) * H + iy) * W + xx] : 0.f;
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float* wpart = smem + 9 * WP;
  for (int cg = 0; cg < Cout; cg += 64) {
    const int co = cg + lane;
    float a[28];
#pragma unroll
    for (int k = 0; k < 28; k++) a[k] = 0.f;
    for (int xx = wave; xx < W; xx += nw) {
      const float g = (co < Cout) ? dz[((long)R * W + xx) * Cout + co] : 0.f;
#pragma unroll
      for (int cr = 0; cr < 9; cr++)
#pragma unroll
        for (int k = 0; k < 3; k++) a[cr * 3 + k] = fmaf(g, smem[cr * WP + xx + k], a[cr * 3 + k]);
      a[27] += g;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 28; k++) wpart[(wave * 28 + k) * 64 + lane] = a[k];
    __syncthreads();
    for (int i = threadIdx.x; i < 28 * 64; i += blockDim.x) {
      const int k = i >> 6, l = i & 63;
      float s = 0.f;
      for (int w2 = 0; w2 < nw; w2++) s += wpart[(w2 * 28 + k) * 64 + l];
      if (cg + l < Cout) part[((long)R * 28 + k) * Cout + cg + l] = s;
    }
  }
}
// part [rows][28][Cout] summed over rows -> dw [Cout][27] (OIHW) and db [Cout]
__global__ void conv3x3_first_wgrad_finish_kernel(const float* __restrict__ colsum28, float* __restrict__ dw,
                                                  float* __restrict__ db, int Cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 28 * Cout) return;
  const int k = i / Cout, co = i % Cout;
  if (k < 27) dw[co * 27 + k] = colsum28[i];
  else if (db) db[co] = colsum28[i];
}

static int max_boundaries_b(int H, int TH, long Rtot) {
  int best = 0;
  long lim = Rtot < (long)H * TH * 2 ? Rtot : (long)H * TH * 2;
  for (long R0 = 0; R0 < lim; R0 += TH) {
    int g = (int)((R0 + TH - 1) / H - R0 / H);
    if (g > best) best = g;
  }
  return best;
}

static int wgrad_cfg(int B, int H, int W, int Cin, int Cout, int* th, int* tw) {
  if (Cin % 64 || Cout % 64) return 0;
  if (W % 8 == 0) { *th = 8; *tw = 8; }
  else if (W % 14 == 0) { *th = 4; *tw = 14; }
  else return 0;
  if (max_boundaries_b(H, *th, (long)B * H) > 2) return 0;
  return 1;
}

extern "C" {

int lctvqa_conv3x3_pack_weights_dgrad(const float* w_oihw, float* w_packed, int Cout, int Cin, void* stream) {
  LCT_REQUIRE(Cout % 16 == 0 && Cin % 32 == 0, "pack_weights_dgrad: need Cout%%16==0, Cin%%32==0 (got %d,%d)", Cout, Cin);
  long total = (long)Cout * Cin * 9;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv3x3_pack_dgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, w_packed, Cout, Cin);
  return lct_check_launch("conv3x3_pack_dgrad");
}

int lctvqa_relu_pool_bwd_nhwc(const float* dout, const float* act, float* dz, int B, int H, int W, int C, int pool, void* stream) {
  LCT_REQUIRE(C % 4 == 0, "relu_pool_bwd: need C%%4==0");   // odd H/W with pool: the caller pre-zeroes dz (uncovered last row/col)
  long t = pool ? (long)B * (H / 2) * (W / 2) * (C / 4) : (long)B * H * W * (C / 4);
  int b = (int)((t + 255) / 256);
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(relu_pool_bwd_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, dout, act, dz, B, H, W, C, pool);
  return lct_check_launch("relu_pool_bwd");
}

int lctvqa_adaptive_avgpool7_flatten_bwd(const float* dy, float* dx_nhwc, int B, int H, int W, int C, void* stream) {
  long t = (long)B * H * W * C;
  int b = (int)((t + 255) / 256);
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(adaptive_avgpool7_flatten_bwd_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, dy, dx_nhwc, B, H, W, C);
  return lct_check_launch("adaptive_avgpool7_flatten_bwd");
}

// floats of scratch needed by lctvqa_conv3x3_wgrad_nhwc for this shape
long lctvqa_conv3x3_wgrad_scratch_elems(int B, int H, int W, int Cin, int Cout) {
  int th, tw;
  if (!wgrad_cfg(B, H, W, Cin, Cout, &th, &tw)) return 0;
  int tiles = (Cin / 64) * (Cout / 64);
  int nsplit = 512 / tiles;
  if (nsplit < 1) nsplit = 1;
  return (long)nsplit * Cout * Cin * 9;
}

// dW (OIHW [Cout][Cin][3][3]) = sum over pixels of dz[p][co] * x[p+tap][ci];  x, dz NHWC
int lctvqa_conv3x3_wgrad_nhwc(const float* x, const float* dz, float* dw_oihw, float* scratch, int B, int H, int W, int Cin,
                              int Cout, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  LCT_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_wgrad: bad shape");
  int th, tw;
  if (!wgrad_cfg(B, H, W, Cin, Cout, &th, &tw)) {
    long total = (long)Cout * Cin * 9;
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(conv3x3_wgrad_direct_kernel, dim3(blocks), dim3(256), 0, st, x, dz, dw_oihw, B, H, W, Cin, Cout);
    return lct_check_launch("conv3x3_wgrad_direct");
  }
  LCT_REQUIRE(scratch != nullptr, "conv3x3_wgrad: scratch required");
  int tiles = (Cin / 64) * (Cout / 64);
  int nsplit = 512 / tiles;
  if (nsplit < 1) nsplit = 1;
  int G = max_boundaries_b(H, th, (long)B * H);
  long n = (long)Cout * Cin * 9;
  if (th == 8) {
    constexpr int TH = 8, TW = 8, RS = 10;
    size_t lds = (size_t)(TH * TW * 64 + (TH + 4) * RS * 64) * sizeof(float);
    auto k = conv3x3_wgrad_kernel<TH, TW, RS>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(tiles * nsplit), dim3(256), lds, st, x, dz, scratch, B, H, W, Cin, Cout, nsplit, G);
  } else {
    constexpr int TH = 4, TW = 14, RS = 16;
    size_t lds = (size_t)(TH * TW * 64 + (TH + 4) * RS * 64) * sizeof(float);
    auto k = conv3x3_wgrad_kernel<TH, TW, RS>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(tiles * nsplit), dim3(256), lds, st, x, dz, scratch, B, H, W, Cin, Cout, nsplit, G);
  }
  int rc = lct_check_launch("conv3x3_wgrad");
  if (rc) return rc;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, scratch, nsplit, n, dw_oihw);
  return lct_check_launch("wgrad_slab_reduce");
}

// first layer: x NCHW [B,3,H,W], dz NHWC [B,H,W,Cout] -> dw [Cout][3][3][3], db [Cout] (may be NULL).
// scratch: B*H*28*Cout + 28*Cout + colsum scratch (see lctvqa_conv3x3_first_wgrad_scratch_elems)
long lctvqa_conv3x3_first_wgrad_scratch_elems(int B, int H, int W, int Cout);
int lctvqa_colsum_f32(const float* X, long ld, int M, int N, float* out, float* scratch, void* stream);
long lctvqa_colsum_scratch_elems(int M, int N);

long lctvqa_conv3x3_first_wgrad_scratch_elems(int B, int H, int W, int Cout) {
  long rows = (long)B * H;
  return rows * 28 * Cout + 28L * Cout + lctvqa_colsum_scratch_elems((int)rows, 28 * Cout);
}

int lctvqa_conv3x3_first_wgrad(const float* x_nchw, const float* dz_nhwc, float* dw, float* db, float* scratch, int B, int H,
                               int W, int Cout, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  LCT_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && scratch != nullptr, "conv3x3_first_wgrad: bad args");
  long rows = (long)B * H;
  float* part = scratch;
  float* sums = part + rows * 28 * Cout;
  float* cs = sums + 28L * Cout;
  size_t lds = (size_t)(9 * (W + 2) + 4 * 28 * 64) * sizeof(float);
  LCT_REQUIRE(lds <= 64 * 1024, "conv3x3_first_wgrad: W too large");
  hipLaunchKernelGGL(conv3x3_first_wgrad_kernel, dim3((int)rows), dim3(256), lds, st, x_nchw, dz_nhwc, part, B, H, W, Cout);
  int rc = lct_check_launch("conv3x3_first_wgrad");
  if (rc) return rc;
  rc = lctvqa_colsum_f32(part, 28L * Cout, (int)rows, 28 * Cout, sums, cs, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(conv3x3_first_wgrad_finish_kernel, dim3(cdiv(28 * Cout, 256)), dim3(256), 0, st, sums, dw, db, Cout);
  return lct_check_launch("conv3x3_first_wgrad_finish");
}

}  // extern "C"
