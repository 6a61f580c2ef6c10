// Weight gradient of a 3x3 convolution in the Winograd domain, fp32, NHWC.
//
// Operator: dW[co][ci][ky][kx] = sum_{n,y,x} dZ[n][y][x][co] * X[n][y+ky-1][x+kx-1][ci]   (zero padding) - the weight gradient of
// nn.Conv2d(Cin,Cout,3,padding=1) inside the reference's trainable VGG19 trunk (basic_vqa/models_lct.py:62-63, pretrained=False)
// and of the ResNet bottleneck's 3x3 convolution.  Per 4x4 tile of dZ and the 6x6 tile of X around it the 3x3 result is the
// minimal-filtering algorithm F(3x3, 4x4) (the transpose of the forward pass's F(4x4, 3x3), same interpolation points):
//     dW += A'^T [ (G' dZ_t G'^T)(6x6) (.) (B^T X_t B)(6x6) ] A'
// The sum over tiles and the contraction over nothing else commute with A'^T . A', so the 36 (xi,nu) planes are 36 independent
// products  M[xi,nu][co][ci] = sum_t U[xi,nu][t][co] * V[xi,nu][t][ci]  = 36 fp32 GEMMs with K = tiles, a quarter of the direct
// algorithm's multiply-adds, run by the LDS-DMA GEMM (gemm.hip) on both operands row-contiguous, the tile axis cut into S slices
// so that 36 * S * (tiles of the Cout x Cin result) fills the chip.  Three HBM-bound kernels around it:
//   wino4_wgrad_xform_x  : X  (NHWC)  -> V[36][Tp][Cin]   (B^T d B, the forward pass's input transform)
//   wino4_wgrad_xform_dz : dZ (NHWC)  -> U[36][Tp][Cout]  (G' g G'^T)
//   wino4_wgrad_finish   : M[36][S][Cout][Cin] -> sum over S, A'^T M A' -> dW (OIHW)
// Tp = S * kper (tile rows T..Tp-1 are written as zeros).  Worth it where Cout and Cin fill the GEMM's 256 x 128 tile:
// lctvqa_conv3x3_wino4_wgrad_supported says so; the direct MFMA weight gradient (conv3x3_bwd.hip) serves the rest.
#include "common.h"

extern "C" int lctvqa_gemm_f32_batched(const float* A, long a_ms, long a_ks, long sA, const float* B, long b_ns, long b_ks, long sB,
                                       float* C, long ldc, long sC, int batch, int M, int N, int K, const float* bias, int act,
                                       void* stream);

namespace {

struct WgPlan {
  int Hp, Wp, tpi, T, S, kper, Tp;
};

static WgPlan wg_plan(int B, int H, int W, int Cin, int Cout) {
  WgPlan p;
  p.Hp = (H + 3) / 4 * 4;
  p.Wp = (W + 3) / 4 * 4;
  p.tpi = (p.Hp / 4) * (p.Wp / 4);
  p.T = B * p.tpi;
  const int tiles = cdiv(Cout, 256) * cdiv(Cin, Cin > 64 ? 128 : 64) * 36;
  int S = cdiv(768, tiles);                      // about three workgroups per CU
  const int maxs = p.T / 256;                    // at least 256 tiles (16 K steps) per slice
  if (S > maxs) S = maxs;
  if (S < 1) S = 1;
  p.kper = ((cdiv(p.T, S) + 15) / 16) * 16;
  p.S = cdiv(p.T, p.kper);
  p.Tp = p.S * p.kper;
  return p;
}

// thread = (tile t, channel quad q): V[xn][t][4q..4q+3] = (B^T d B)[xi][nu] of the 6x6 patch around tile t; t >= T: zeros
__global__ __launch_bounds__(256) void wino4_wgrad_xform_x(const float* __restrict__ x, float* __restrict__ V, int H, int W, int C,
                                                           int Hp4, int Wp4, int T, int Tp) {
  const int CQ = C >> 2;
  const long total = (long)Tp * CQ;
  const long plane = (long)Tp * C;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int q = (int)(o % CQ);
    const int t = (int)(o / CQ);
    float* dst = V + (long)t * C + 4 * q;
    if (t >= T) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int xn = 0; xn < 36; xn++) *(f32x4*)(dst + xn * plane) = z;
      continue;
    }
    const int tpi = Hp4 * Wp4;
    const int n = t / tpi, r = t % tpi;
    const int y0 = (r / Wp4) * 4 - 1, x0 = (r % Wp4) * 4 - 1;
    f32x4 tb[6][6];                                            // (B^T d)[xi][col]
#pragma unroll
    for (int c = 0; c < 6; c++) {
      f32x4 d[6];
      const int xx = x0 + c;
#pragma unroll
      for (int rr = 0; rr < 6; rr++) {
        const int yy = y0 + rr;
        d[rr] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? *(const f32x4*)(x + (((long)n * H + yy) * W + xx) * C + 4 * q)
                                                        : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      tb[0][c] = 4.f * d[0] - 5.f * d[2] + d[4];
      tb[1][c] = (d[4] - 4.f * d[2]) + (d[3] - 4.f * d[1]);
      tb[2][c] = (d[4] - 4.f * d[2]) - (d[3] - 4.f * d[1]);
      tb[3][c] = (d[4] - d[2]) + 2.f * (d[3] - d[1]);
      tb[4][c] = (d[4] - d[2]) - 2.f * (d[3] - d[1]);
      tb[5][c] = 4.f * d[1] - 5.f * d[3] + d[5];
    }
#pragma unroll
    for (int xi = 0; xi < 6; xi++) {
      const f32x4* e = tb[xi];
      float* v = dst + (long)(xi * 6) * plane;
      *(f32x4*)(v) = 4.f * e[0] - 5.f * e[2] + e[4];
      *(f32x4*)(v + plane) = (e[4] - 4.f * e[2]) + (e[3] - 4.f * e[1]);
      *(f32x4*)(v + 2 * plane) = (e[4] - 4.f * e[2]) - (e[3] - 4.f * e[1]);
      *(f32x4*)(v + 3 * plane) = (e[4] - e[2]) + 2.f * (e[3] - e[1]);
      *(f32x4*)(v + 4 * plane) = (e[4] - e[2]) - 2.f * (e[3] - e[1]);
      *(f32x4*)(v + 5 * plane) = 4.f * e[1] - 5.f * e[3] + e[5];
    }
  }
}

// G' g for one column of four values: the rows of G' = diag(1/4, -1/6, -1/6, 1/24, 1/24, 1) [p^k], p = 0, 1, -1, 2, -2, inf
__device__ __forceinline__ void g4(const f32x4 g0, const f32x4 g1, const f32x4 g2, const f32x4 g3, f32x4 (&o)[6]) {
  const f32x4 s02 = g0 + g2, s13 = g1 + g3;
  const f32x4 a = g0 + 4.f * g2, b = 2.f * g1 + 8.f * g3;
  o[0] = 0.25f * g0;
  o[1] = (-1.f / 6) * (s02 + s13);
  o[2] = (-1.f / 6) * (s02 - s13);
  o[3] = (1.f / 24) * (a + b);
  o[4] = (1.f / 24) * (a - b);
  o[5] = g3;
}

// thread = (tile t, channel quad q): U[xn][t][4q..] = (G' g G'^T)[xi][nu] of the 4x4 tile t of dZ; t >= T: zeros
__global__ __launch_bounds__(256) void wino4_wgrad_xform_dz(const float* __restrict__ dz, float* __restrict__ U, int H, int W, int C,
                                                            int Hp4, int Wp4, int T, int Tp) {
  const int CQ = C >> 2;
  const long total = (long)Tp * CQ;
  const long plane = (long)Tp * C;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int q = (int)(o % CQ);
    const int t = (int)(o / CQ);
    float* dst = U + (long)t * C + 4 * q;
    if (t >= T) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int xn = 0; xn < 36; xn++) *(f32x4*)(dst + xn * plane) = z;
      continue;
    }
    const int tpi = Hp4 * Wp4;
    const int n = t / tpi, r = t % tpi;
    const int y0 = (r / Wp4) * 4, x0 = (r % Wp4) * 4;
    f32x4 tg[6][4];                                            // (G' g)[xi][col]
#pragma unroll
    for (int c = 0; c < 4; c++) {
      f32x4 g[4];
      const int xx = x0 + c;
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int yy = y0 + rr;
        g[rr] = (yy < H && xx < W) ? *(const f32x4*)(dz + (((long)n * H + yy) * W + xx) * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      f32x4 o6[6];
      g4(g[0], g[1], g[2], g[3], o6);
#pragma unroll
      for (int xi = 0; xi < 6; xi++) tg[xi][c] = o6[xi];
    }
#pragma unroll
    for (int xi = 0; xi < 6; xi++) {
      f32x4 o6[6];
      g4(tg[xi][0], tg[xi][1], tg[xi][2], tg[xi][3], o6);
      float* v = dst + (long)(xi * 6) * plane;
#pragma unroll
      for (int nu = 0; nu < 6; nu++) *(f32x4*)(v + nu * plane) = o6[nu];
    }
  }
}

// thread = (co, ci): m[xn] = sum_s M[xn][s][co][ci];  dW[co][ci] = A'^T m A',  A'^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 1]
__global__ __launch_bounds__(256) void wino4_wgrad_finish(const float* __restrict__ M, float* __restrict__ dw, int Cout, int Cin, int S) {
  const long total = (long)Cout * Cin;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    float r[6][3];                                             // (m A')[xi][j]
#pragma unroll
    for (int xi = 0; xi < 6; xi++) {
      float m[6];
#pragma unroll
      for (int nu = 0; nu < 6; nu++) {
        const float* p = M + ((long)(xi * 6 + nu) * S) * total + o;
        float a = 0.f, a1 = 0.f;                                  // two slices per iteration (36 planes already give 36 loads in flight)
        int s = 0;
        for (; s + 1 < S; s += 2) { a += p[(long)s * total]; a1 += p[(long)(s + 1) * total]; }
        if (s < S) a += p[(long)s * total];
        m[nu] = a + a1;
      }
      r[xi][0] = m[0] + m[1] + m[2] + m[3] + m[4];
      r[xi][1] = (m[1] - m[2]) + 2.f * (m[3] - m[4]);
      r[xi][2] = (m[1] + m[2]) + 4.f * (m[3] + m[4]) + m[5];
    }
    float* d = dw + o * 9;
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      d[0 * 3 + jj] = r[0][jj] + r[1][jj] + r[2][jj] + r[3][jj] + r[4][jj];
      d[1 * 3 + jj] = (r[1][jj] - r[2][jj]) + 2.f * (r[3][jj] - r[4][jj]);
      d[2 * 3 + jj] = (r[1][jj] + r[2][jj]) + 4.f * (r[3][jj] + r[4][jj]) + r[5][jj];
    }
  }
}

// ---- split forward path (built as an A/B): F(4x4,3x3) as three kernels -- the input transform above
// (V = B^T d B), 36 batched fp32 GEMMs M[xn][t][co] = sum_ci V[xn][t][ci] * U[xn][co][ci], and the output transform below
// (Y = A^T M A + bias, ReLU, optional 2x2 max pool) -- instead of the fused conv3x3_wino4_kernel.  U[36][Cout][Cin] = G g G^T.
__global__ void wino4_split_pack_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin) {
  const long total = (long)Cout * Cin;
  const long plane = total;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const float* g = w + o * 9;                                  // OIHW: [co][ci][3][3], o = co * Cin + ci
    float t[6][3];                                               // G g
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float g0 = g[c], g1 = g[3 + c], g2 = g[6 + c];
      t[0][c] = 0.25f * g0;
      t[1][c] = (-1.f / 6) * (g0 + g1 + g2);
      t[2][c] = (-1.f / 6) * (g0 - g1 + g2);
      t[3][c] = (1.f / 24) * g0 + (1.f / 12) * g1 + (1.f / 6) * g2;
      t[4][c] = (1.f / 24) * g0 - (1.f / 12) * g1 + (1.f / 6) * g2;
      t[5][c] = g2;
    }
#pragma unroll
    for (int xi = 0; xi < 6; xi++) {
      const float a = t[xi][0], b = t[xi][1], c = t[xi][2];
      float* u = U + (long)(xi * 6) * plane + o;
      u[0] = 0.25f * a;
      u[plane] = (-1.f / 6) * (a + b + c);
      u[2 * plane] = (-1.f / 6) * (a - b + c);
      u[3 * plane] = (1.f / 24) * a + (1.f / 12) * b + (1.f / 6) * c;
      u[4 * plane] = (1.f / 24) * a - (1.f / 12) * b + (1.f / 6) * c;
      u[5 * plane] = c;
    }
  }
}
// thread = (tile t, channel quad q): the 4x4 output tile of M[36][Tp][C] -> y NHWC (+ bias, ReLU, 2x2 max pool)
__global__ __launch_bounds__(256) void wino4_split_out_kernel(const float* __restrict__ M, const float* __restrict__ bias, float* __restrict__ y,
                                                              int H, int W, int C, int Hp4, int Wp4, int T, int Tp, int relu, int pool) {
  const int CQ = C >> 2;
  const long total = (long)T * CQ;
  const long plane = (long)Tp * C;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int q = (int)(o % CQ);
    const int t = (int)(o / CQ);
    const float* src = M + (long)t * C + 4 * q;
    f32x4 ta[4][6];                                              // (A^T m)[i][nu]
#pragma unroll
    for (int nu = 0; nu < 6; nu++) {
      f32x4 m[6];
#pragma unroll
      for (int xi = 0; xi < 6; xi++) m[xi] = *(const f32x4*)(src + (long)(xi * 6 + nu) * plane);
      const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
      ta[0][nu] = m[0] + s12 + s34;
      ta[1][nu] = d12 + 2.f * d34;
      ta[2][nu] = s12 + 4.f * s34;
      ta[3][nu] = d12 + 8.f * d34 + m[5];
    }
    const f32x4 bv = bias ? *(const f32x4*)(bias + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 out[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const f32x4* e = ta[i];
      const f32x4 s12 = e[1] + e[2], d12 = e[1] - e[2], s34 = e[3] + e[4], d34 = e[3] - e[4];
      out[i][0] = e[0] + s12 + s34 + bv;
      out[i][1] = d12 + 2.f * d34 + bv;
      out[i][2] = s12 + 4.f * s34 + bv;
      out[i][3] = d12 + 8.f * d34 + e[5] + bv;
    }
    if (relu) {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int jx = 0; jx < 4; jx++)
#pragma unroll
          for (int e = 0; e < 4; e++) out[i][jx][e] = fmaxf(out[i][jx][e], 0.f);
    }
    const int tpi = Hp4 * Wp4;
    const int n = t / tpi, r = t % tpi;
    const int y0 = (r / Wp4) * 4, x0 = (r % Wp4) * 4;
    if (!pool) {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int jx = 0; jx < 4; jx++)
          if (y0 + i < H && x0 + jx < W) *(f32x4*)(y + (((long)n * H + y0 + i) * W + x0 + jx) * C + 4 * q) = out[i][jx];
    } else {
      const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int jx = 0; jx < 2; jx++) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; e++)
            v[e] = fmaxf(fmaxf(out[2 * i][2 * jx][e], out[2 * i][2 * jx + 1][e]), fmaxf(out[2 * i + 1][2 * jx][e], out[2 * i + 1][2 * jx + 1][e]));
          const int oy = (y0 >> 1) + i, ox = (x0 >> 1) + jx;
          if (oy < Ho && ox < Wo) *(f32x4*)(y + (((long)n * Ho + oy) * Wo + ox) * C + 4 * q) = v;
        }
    }
  }
}

}  // namespace

extern "C" {

long lctvqa_conv3x3_wino4_split_workspace_elems(int B, int H, int W, int Cin, int Cout) {
  const long T = (long)B * ((H + 3) / 4) * ((W + 3) / 4);
  const long Tp = (T + 3) / 4 * 4;
  return 36L * Tp * (Cin + Cout);
}
int lctvqa_conv3x3_wino4_split_pack_weights(const float* w_oihw, float* U, int Cout, int Cin, void* stream) {
  long nb = ((long)Cout * Cin + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(wino4_split_pack_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, w_oihw, U, Cout, Cin);
  return lct_check_launch("wino4_split_pack");
}
// y = relu?(conv3x3(x) + bias) (+ 2x2 max pool), NHWC fp32, F(4x4,3x3) as transform -> 36 batched GEMMs -> transform.
// U: lctvqa_conv3x3_wino4_split_pack_weights; workspace: lctvqa_conv3x3_wino4_split_workspace_elems floats.
int lctvqa_conv3x3_wino4_split_nhwc(const float* x, const float* U, const float* bias, float* y, float* workspace, long workspace_elems,
                                    int B, int H, int W, int Cin, int Cout, int relu, int pool, void* stream) {
  LCT_REQUIRE(B >= 1 && H >= 2 && W >= 2 && Cin % 4 == 0 && Cout % 4 == 0 && (!pool || (H % 2 == 0 && W % 2 == 0)), "conv3x3_wino4_split: bad shape");
  LCT_REQUIRE(workspace && workspace_elems >= lctvqa_conv3x3_wino4_split_workspace_elems(B, H, W, Cin, Cout), "conv3x3_wino4_split: workspace too small");
  LCT_REQUIRE((((uintptr_t)x | (uintptr_t)U | (uintptr_t)y | (uintptr_t)workspace) & 15) == 0, "conv3x3_wino4_split: tensors must be 16-B aligned");
  hipStream_t st = (hipStream_t)stream;
  const int Hp4 = (H + 3) / 4, Wp4 = (W + 3) / 4;
  const long T = (long)B * Hp4 * Wp4, Tp = (T + 3) / 4 * 4;
  float* V = workspace;
  float* M = V + 36L * Tp * Cin;
  auto blocks = [](long threads) { long b = (threads + 255) / 256; return (int)(b > 16384 ? 16384 : b); };
  hipLaunchKernelGGL(wino4_wgrad_xform_x, dim3(blocks(Tp * (Cin / 4))), dim3(256), 0, st, x, V, H, W, Cin, Hp4, Wp4, (int)T, (int)Tp);
  int rc = lct_check_launch("wino4_split_xform_x");
  if (rc) return rc;
  rc = lctvqa_gemm_f32_batched(V, Cin, 1, Tp * Cin, U, Cin, 1, (long)Cout * Cin, M, Cout, Tp * Cout, 36, (int)Tp, Cout, Cin, nullptr, 0, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(wino4_split_out_kernel, dim3(blocks(T * (Cout / 4))), dim3(256), 0, st, M, bias, y, H, W, Cout, Hp4, Wp4, (int)T, (int)Tp,
                     relu, pool);
  return lct_check_launch("wino4_split_out");
}

// 1 where the Winograd-domain weight gradient is the better choice (measured, B = 256: 2.0-2.8x at 256/512 channels, 1.65x at
// 128->256, 1.35x at 14x14 256->256, a loss at 128->128): both channel counts at least 128 and one at least 256, the map wastes at
// most a quarter of the 4x4 tile grid, and there are enough tiles for 16 K steps
int lctvqa_conv3x3_wino4_wgrad_supported(int B, int H, int W, int Cin, int Cout) {
  if (B < 1 || H < 2 || W < 2 || Cin % 4 || Cout % 4 || Cin < 128 || Cout < 128 || (long)Cin * Cout < 128 * 256) return 0;
  const WgPlan p = wg_plan(B, H, W, Cin, Cout);
  if ((long)H * W * 4 < (long)p.Hp * p.Wp * 3) return 0;
  if (p.T < 256) return 0;
  return 1;
}

// floats of workspace: U[36][Tp][Cout] + V[36][Tp][Cin] + M[36][S][Cout][Cin]
long lctvqa_conv3x3_wino4_wgrad_workspace_elems(int B, int H, int W, int Cin, int Cout) {
  const WgPlan p = wg_plan(B, H, W, Cin, Cout);
  return 36L * p.Tp * (Cin + Cout) + 36L * p.S * Cout * Cin;
}

int lctvqa_conv3x3_wino4_wgrad(const float* x, const float* dz, float* dw_oihw, float* workspace, long workspace_elems, int B, int H,
                               int W, int Cin, int Cout, void* stream) {
  LCT_REQUIRE(lctvqa_conv3x3_wino4_wgrad_supported(B, H, W, Cin, Cout), "conv3x3_wino4_wgrad: unsupported shape B=%d H=%d W=%d Cin=%d Cout=%d",
              B, H, W, Cin, Cout);
  LCT_REQUIRE(workspace && workspace_elems >= lctvqa_conv3x3_wino4_wgrad_workspace_elems(B, H, W, Cin, Cout),
              "conv3x3_wino4_wgrad: workspace too small");
  LCT_REQUIRE((((uintptr_t)x | (uintptr_t)dz | (uintptr_t)workspace) & 15) == 0, "conv3x3_wino4_wgrad: tensors must be 16-B aligned");
  hipStream_t st = (hipStream_t)stream;
  const WgPlan p = wg_plan(B, H, W, Cin, Cout);
  float* U = workspace;
  float* V = U + 36L * p.Tp * Cout;
  float* M = V + 36L * p.Tp * Cin;
  auto blocks = [](long threads) { long b = (threads + 255) / 256; return (int)(b > 16384 ? 16384 : b); };
  hipLaunchKernelGGL(wino4_wgrad_xform_dz, dim3(blocks((long)p.Tp * (Cout / 4))), dim3(256), 0, st, dz, U, H, W, Cout, p.Hp / 4, p.Wp / 4,
                     p.T, p.Tp);
  int rc = lct_check_launch("wino4_wgrad_xform_dz");
  if (rc) return rc;
  hipLaunchKernelGGL(wino4_wgrad_xform_x, dim3(blocks((long)p.Tp * (Cin / 4))), dim3(256), 0, st, x, V, H, W, Cin, p.Hp / 4, p.Wp / 4, p.T,
                     p.Tp);
  rc = lct_check_launch("wino4_wgrad_xform_x");
  if (rc) return rc;
  // batch member (xn, s): A(m = co, k = tile) = U[xn][s*kper + k][co], B(n = ci, k) = V[xn][s*kper + k][ci]; Tp = S*kper makes the
  // member stride uniform
  rc = lctvqa_gemm_f32_batched(U, 1, Cout, (long)p.kper * Cout, V, 1, Cin, (long)p.kper * Cin, M, Cin, (long)Cout * Cin, 36 * p.S, Cout,
                               Cin, p.kper, nullptr, 0, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(wino4_wgrad_finish, dim3(blocks((long)Cout * Cin)), dim3(256), 0, st, M, dw_oihw, Cout, Cin, p.S);
  return lct_check_launch("wino4_wgrad_finish");
}

}  // extern "C"
