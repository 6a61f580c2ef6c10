// Winograd F(2x2,3x3) convolution for the VGG19 trunk on gfx950, fp32 in / fp32 accumulate, NHWC.
//
// Same operator as conv3x3.hip (nn.Conv2d(Cin,Cout,3,padding=1) + bias + ReLU [+ MaxPool2d(2,2)], reference
// basic_vqa/models.py:36 via torchvision vgg19.features) computed with the minimal-filtering identity
//     Y = A^T [ sum_cin (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray 2016, F(2x2,3x3))
// i.e. 16 multiplies per 2x2 output tile and (cin,cout) pair instead of 36: 2.25x fewer MFMA FLOPs.  All
// arithmetic is fp32 (v_mfma_f32_32x32x2_f32 for the 16 transformed-domain GEMMs); the transforms only add /
// subtract (B, A) or scale by 1/2 (G), so the result differs from the direct kernel by ordinary fp32 rounding
// (measured <= 2e-6 relative on the test shapes; parity gate 1e-3 on logits, argmax identical).
//
// Structure (one workgroup = 4 waves, ONE wave per SIMD with the whole 512-register budget):
//  * a workgroup owns 32*WM Winograd tiles (a TH x TW patch of output pixels) x 32*WN output channels; wave
//    (wm, wn) keeps the 16 transformed-domain accumulator tiles (xi,nu) of its 32 tiles x 32 couts in 256 VGPRs;
//  * per 16-channel chunk: the zero-padded raw halo patch is DMA'd HBM->LDS (same planar layout, same
//    stacked-row / shared-zero-row trick as the direct kernel), every thread transforms one (tile, 4-channel)
//    item B^T d B in registers (32 float4 add/sub) and writes the 16 (xi,nu) planes V[xi,nu][chunk][tile][4ch];
//  * the MFMA phase reads V fragments with ds_read_b128 (consecutive tiles -> conflict free) and takes the
//    pre-transformed weights U = G g G^T STRAIGHT FROM L2 INTO REGISTERS (2 KB per (cout group, chunk, xi,nu),
//    lane-contiguous float4s, software-prefetched one group of 4 (xi,nu) ahead): no weight staging in LDS, so the
//    only barriers are two per chunk (128 MFMAs per wave in between);
//  * epilogue: the 16 (xi,nu) values of a (tile, cout) sit in ONE lane (same register index of the 16
//    accumulator tiles), so A^T M A, bias, ReLU and the 2x2 max-pool (= max over the tile's 4 outputs) are
//    in-register; NHWC stores are 128-B contiguous per half wave.
#include "common.h"

__device__ __attribute__((aligned(64))) float g_zero_page_w[64];

// U[cout/32][Cin/16][16 xi*4+nu][4 chunks][32 cout][4 ch] = G g G^T.   flip_t: weights of the dgrad convolution
// (g'[ci][co][ky][kx] = g[co][ci][2-ky][2-kx]; Cout/Cin then name the dgrad conv's own output/input channels).
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, int flip_t) {
  const long total = (long)Cout * Cin * 16;
  const int NCH = Cin / 16;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int e = o & 3, j = (o >> 2) & 31, c = (o >> 7) & 3, xn = (o >> 9) & 15;
    const long rest = o >> 13;
    const int chunk = rest % NCH, grp = rest / NCH;
    const int co = grp * 32 + j, ci = chunk * 16 + c * 4 + e;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++)
        g[a][b] = flip_t ? w[((long)ci * Cout + co) * 9 + (2 - a) * 3 + (2 - b)] : w[((long)co * Cin + ci) * 9 + a * 3 + b];
    // t = G g (4x3), U = t G^T (4x4);  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; b++) {
      t[0][b] = g[0][b];
      t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      t[3][b] = g[2][b];
    }
    const int xi = xn >> 2, nu = xn & 3;
    float v;
    if (nu == 0) v = t[xi][0];
    else if (nu == 1) v = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]);
    else if (nu == 2) v = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]);
    else v = t[xi][2];
    u[o] = v;
  }
}

template <int TH, int TW, int WM, int WN, bool POOL>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_kernel(const float* __restrict__ x, const float* __restrict__ up,
                                                              const float* __restrict__ bias, float* __restrict__ y, int Btot,
                                                              int H, int W, int Cin, int Cout, int relu, int G) {
  static_assert(WM * WN == 4, "4 waves");
  static_assert(TH % 2 == 0 && TW % 2 == 0, "patch must hold whole 2x2 tiles");
  constexpr int TX = TW / 2, TY = TH / 2, NTILE = TX * TY;     // live Winograd tiles of the patch
  constexpr int MTILES = 32 * WM;                               // tile slots (MFMA rows) of the workgroup
  static_assert(NTILE <= MTILES, "patch holds too many tiles");
  constexpr int RS = TW + 2;
  constexpr int HRMAX = TH + 4;
  constexpr int PMAX = HRMAX * RS;
  constexpr int AROUNDS = (PMAX + 63) / 64;
  constexpr int PPAD = AROUNDS * 64;
  constexpr int PLANE = PPAD * 4;                               // floats per raw chunk plane
  constexpr int RAWSZ = 4 * PLANE;
  constexpr int VPL = MTILES * 4;                               // floats per (xi,nu,chunk) plane of V
  constexpr int VSZ = 16 * 4 * VPL;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* raw = smem;                 // [2][RAWSZ]
  float* V = smem + 2 * RAWSZ;       // [16][4][MTILES][4]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int j = lane & 31, h = lane >> 5;
  const int NT = Cout / (32 * WN);
  const int n_tile = blockIdx.x % NT, m_tile = blockIdx.x / NT;
  const int CT = W / TW;
  const int ct = m_tile % CT, rt = m_tile / CT;
  const int R0 = rt * TH, x0 = ct * TW;
  const int Rtot = Btot * H;
  const int n0img = R0 / H, y0 = R0 % H;
  const int NCH = Cin / 16;
  const int PLIVE = (TH + 2 + G) * RS;

  // ---- halo staging offsets (identical scheme to conv3x3_mfma_kernel) ----
  int aoff[AROUNDS];
#pragma unroll
  for (int r = 0; r < AROUNDS; r++) {
    int p = r * 64 + lane;
    int hr = p / RS, hx = p % RS;
    int xx = x0 - 1 + hx;
    int yv = y0 - 1 + hr;
    int n = n0img;
    while (yv > H) { yv -= (H + 1); n++; }
    bool ok = (p < PLIVE) && (yv >= 0) && (yv < H) && (n < Btot) && (xx >= 0) && (xx < W);
    aoff[r] = ok ? (((n * H + yv) * W + xx) * Cin + wave * 4) : -1;
  }
  auto stage_raw = [&](int chunk, float* dst) {
#pragma unroll
    for (int r = 0; r < AROUNDS; r++) {
      if (r * 64 < PLIVE) {
        const float* src = (aoff[r] >= 0) ? (x + (long)aoff[r] + chunk * 16) : g_zero_page_w;
        glds16(src, dst + wave * PLANE + r * 256);
      }
    }
  };

  // ---- transform item of this thread: (tile slot tt, channel chunk tc); items = MTILES*4 <= 256 ----
  const int tt = tid % MTILES, tc = tid / MTILES;
  const bool titem = (tid < MTILES * 4);
  const bool tlive = titem && (tt < NTILE);
  int tbase = 0;                                   // float offset of input pixel (2ty, 2tx) [halo coords] in plane tc
  if (tlive) {
    const int ty = tt / TX, tx = tt % TX;
    const int yy = y0 + 2 * ty;
    const int g = (yy >= H) + (yy >= 2 * H);       // tiles never straddle an image boundary (H even)
    tbase = tc * PLANE + ((2 * ty + g) * RS + 2 * tx) * 4;
  }
  auto transform = [&](const float* rw) {
    if (!titem) return;
    // t = B^T d  (column by column: 4 loads live at a time), v = t B;  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
    f32x4 t[4][4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
      f32x4 d0, d1, d2, d3;
      if (tlive) {
        d0 = *(const f32x4*)(rw + tbase + (0 * RS + b) * 4);
        d1 = *(const f32x4*)(rw + tbase + (1 * RS + b) * 4);
        d2 = *(const f32x4*)(rw + tbase + (2 * RS + b) * 4);
        d3 = *(const f32x4*)(rw + tbase + (3 * RS + b) * 4);
      } else {
        d0 = d1 = d2 = d3 = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      t[0][b] = d0 - d2;
      t[1][b] = d1 + d2;
      t[2][b] = d2 - d1;
      t[3][b] = d1 - d3;
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
      float* vp = V + ((a * 4) * 4 + tc) * VPL + tt * 4;
      *(f32x4*)(vp) = t[a][0] - t[a][2];
      *(f32x4*)(vp + 4 * VPL) = t[a][1] + t[a][2];
      *(f32x4*)(vp + 8 * VPL) = t[a][2] - t[a][1];
      *(f32x4*)(vp + 12 * VPL) = t[a][1] - t[a][3];
    }
  };

  f32x16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; q++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[q][r] = 0.f;

  // transformed weights of this wave's cout group: [chunk][16][4][32][4]; lane reads float4 at chunk plane c = 2kk+h, cout j
  const float* ub = up + (long)(n_tile * WN + wn) * NCH * 8192 + (h * 32 + j) * 4;
  const int abase = (wm * 32 + j) * 4 + h * VPL;            // float offset of this lane's A fragment inside a (xi,nu) slab

  f32x4 B0[4][2], B1[4][2];
  auto ldB = [&](const float* base, int grp, f32x4 (&Bf)[4][2]) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int kk = 0; kk < 2; kk++) Bf[q][kk] = *(const f32x4*)(base + ((grp * 4 + q) * 4 + 2 * kk) * 128);
  };

  stage_raw(0, raw);
  if (NCH > 1) stage_raw(1, raw + RAWSZ);
  ldB(ub, 0, B0);

  for (int chunk = 0; chunk < NCH; chunk++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // raw[chunk&1] landed (issued a whole MFMA phase ago)
    __syncthreads();                                         // ... for every wave; V is free (previous MFMA phase done)
    transform(raw + (chunk & 1) * RAWSZ);
    __syncthreads();                                         // V complete; raw[chunk&1] may be overwritten
    if (chunk + 2 < NCH) stage_raw(chunk + 2, raw + (chunk & 1) * RAWSZ);
    const float* uc = ub + (long)chunk * 8192;
    // ---- MFMA phase: 16 (xi,nu) x K=16, weights prefetched one group of 4 ahead ----
#define WINO_GROUP(GI, BCUR)                                                                          \
    {                                                                                                 \
      _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                  \
        _Pragma("unroll") for (int kk = 0; kk < 2; kk++) {                                             \
          const f32x4 af = *(const f32x4*)(V + (((GI) * 4 + q) * 4 + 2 * kk) * VPL + abase);          \
          _Pragma("unroll") for (int e = 0; e < 4; e++)                                                \
            acc[(GI) * 4 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], BCUR[q][kk][e], acc[(GI) * 4 + q], 0, 0, 0); \
        }                                                                                              \
      }                                                                                               \
    }
    ldB(uc, 1, B1);
    WINO_GROUP(0, B0)
    ldB(uc, 2, B0);
    WINO_GROUP(1, B1)
    ldB(uc, 3, B1);
    WINO_GROUP(2, B0)
    if (chunk + 1 < NCH) ldB(uc + 8192, 0, B0);
    WINO_GROUP(3, B1)
#undef WINO_GROUP
  }

  // ---- epilogue: Y = A^T M A per (tile, cout) in registers; bias + ReLU (+ 2x2 max pool = max over the tile) ----
  const int co = (n_tile * WN + wn) * 32 + j;
  const float bv = bias ? bias[co] : 0.f;
  const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int tile = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    // T = A^T M : T[i][nu],  A^T = [[1,1,1,0],[0,1,-1,-1]]
    float T0[4], T1[4];
#pragma unroll
    for (int nu = 0; nu < 4; nu++) {
      T0[nu] = acc[0 * 4 + nu][r] + acc[1 * 4 + nu][r] + acc[2 * 4 + nu][r];
      T1[nu] = acc[1 * 4 + nu][r] - acc[2 * 4 + nu][r] - acc[3 * 4 + nu][r];
    }
    float y00 = T0[0] + T0[1] + T0[2] + bv, y01 = T0[1] - T0[2] - T0[3] + bv;
    float y10 = T1[0] + T1[1] + T1[2] + bv, y11 = T1[1] - T1[2] - T1[3] + bv;
    if (relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
    if (tile < NTILE) {
      const int ty = tile / TX, tx = tile % TX;
      const int R = R0 + 2 * ty;
      if (R < Rtot) {
        if (POOL) {
          const int n = R / H, yy = R % H;
          const long opix = ((long)n * Ho + (yy >> 1)) * Wo + ((x0 >> 1) + tx);
          y[opix * Cout + co] = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11));
        } else {
          float* o0 = y + ((long)R * W + x0 + 2 * tx) * Cout + co;
          o0[0] = y00; o0[Cout] = y01;
          o0[(long)W * Cout] = y10; o0[(long)W * Cout + Cout] = y11;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// v2 (1x4 wave layout, Cout % 128 == 0): software-pipelined.  V is double buffered; the input transform of chunk
// c+1 is split into four column stages that sit between the four (xi,nu) groups of chunk c's MFMA phase, A fragments
// and weights are prefetched one group ahead, and there is ONE barrier per chunk (128 MFMAs per wave).
// Transform work is spread over all 256 threads: thread = (tile, channel chunk, half), each half produces two of the
// four B^T-rows of its item.
// ---------------------------------------------------------------------------------------------------------------
template <int TH, int TW, bool POOL>
__global__ __launch_bounds__(256, 1) void conv3x3_wino2_kernel(const float* __restrict__ x, const float* __restrict__ up,
                                                               const float* __restrict__ bias, float* __restrict__ y, int Btot,
                                                               int H, int W, int Cin, int Cout, int relu, int G) {
  constexpr int TX = TW / 2, TY = TH / 2, NTILE = TX * TY;
  static_assert(NTILE <= 32, "patch holds too many tiles");
  constexpr int RS = TW + 2;
  constexpr int HRMAX = TH + 4;
  constexpr int PMAX = HRMAX * RS;
  constexpr int AROUNDS = (PMAX + 63) / 64;
  constexpr int PLANE = AROUNDS * 64 * 4;
  constexpr int RAWSZ = 4 * PLANE;
  constexpr int VPL = 32 * 4;                                   // floats per (xi,nu,chunk) plane
  constexpr int VBUF = 16 * 4 * VPL;                            // floats per V buffer (32 KB)

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* raw = smem;                 // [2][RAWSZ]
  float* V = smem + 2 * RAWSZ;       // [2][VBUF]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave;
  const int j = lane & 31, h = lane >> 5;
  const int NT = Cout / 128;
  const int n_tile = blockIdx.x % NT, m_tile = blockIdx.x / NT;
  const int CT = W / TW;
  const int ct = m_tile % CT, rt = m_tile / CT;
  const int R0 = rt * TH, x0 = ct * TW;
  const int Rtot = Btot * H;
  const int n0img = R0 / H, y0 = R0 % H;
  const int NCH = Cin / 16;
  const int PLIVE = (TH + 2 + G) * RS;

  int aoff[AROUNDS];
#pragma unroll
  for (int r = 0; r < AROUNDS; r++) {
    int p = r * 64 + lane;
    int hr = p / RS, hx = p % RS;
    int xx = x0 - 1 + hx;
    int yv = y0 - 1 + hr;
    int n = n0img;
    while (yv > H) { yv -= (H + 1); n++; }
    bool ok = (p < PLIVE) && (yv >= 0) && (yv < H) && (n < Btot) && (xx >= 0) && (xx < W);
    aoff[r] = ok ? (((n * H + yv) * W + xx) * Cin + wave * 4) : -1;
  }
  auto stage_raw = [&](int chunk, float* dst) {
#pragma unroll
    for (int r = 0; r < AROUNDS; r++) {
      if (r * 64 < PLIVE) {
        const float* src = (aoff[r] >= 0) ? (x + (long)aoff[r] + chunk * 16) : g_zero_page_w;
        glds16(src, dst + wave * PLANE + r * 256);
      }
    }
  };

  // transform item: tile tt, channel chunk tc, half hlf (wave-uniform: waves 0,1 -> rows 0,1 of B^T d; waves 2,3 -> rows 2,3)
  const int item = tid & 127, tt = item & 31, tc = item >> 5;
  const int hlf = __builtin_amdgcn_readfirstlane(tid >> 7);
  const bool tlive = tt < NTILE;
  int tbase = 0;
  if (tlive) {
    const int ty = tt / TX, tx = tt % TX;
    const int yy = y0 + 2 * ty;
    const int g = (yy >= H) + (yy >= 2 * H);
    tbase = tc * PLANE + ((2 * ty + g) * RS + 2 * tx) * 4;
  }
  const int vwbase = tc * VPL + tt * 4;                         // + (xn*4)*VPL
  f32x4 tA[4], tB[4];                                           // the two B^T-rows of this thread, over columns b
  auto tpart = [&](const float* rw, int b) {
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* p0 = rw + tbase + b * 4;
    if (hlf == 0) {
      const f32x4 d0 = tlive ? *(const f32x4*)(p0) : z, d1 = tlive ? *(const f32x4*)(p0 + RS * 4) : z,
                  d2 = tlive ? *(const f32x4*)(p0 + 2 * RS * 4) : z;
      tA[b] = d0 - d2;          // row 0
      tB[b] = d1 + d2;          // row 1
    } else {
      const f32x4 d1 = tlive ? *(const f32x4*)(p0 + RS * 4) : z, d2 = tlive ? *(const f32x4*)(p0 + 2 * RS * 4) : z,
                  d3 = tlive ? *(const f32x4*)(p0 + 3 * RS * 4) : z;
      tA[b] = d2 - d1;          // row 2
      tB[b] = d1 - d3;          // row 3
    }
  };
  auto twrite = [&](float* Vd) {
    float* va = Vd + ((2 * hlf) * 4) * 4 * VPL + vwbase;        // xn = a*4 + nu, a = 2*hlf (+1)
    float* vb = va + 4 * 4 * VPL;
    *(f32x4*)(va) = tA[0] - tA[2];
    *(f32x4*)(va + 4 * VPL) = tA[1] + tA[2];
    *(f32x4*)(va + 8 * VPL) = tA[2] - tA[1];
    *(f32x4*)(va + 12 * VPL) = tA[1] - tA[3];
    *(f32x4*)(vb) = tB[0] - tB[2];
    *(f32x4*)(vb + 4 * VPL) = tB[1] + tB[2];
    *(f32x4*)(vb + 8 * VPL) = tB[2] - tB[1];
    *(f32x4*)(vb + 12 * VPL) = tB[1] - tB[3];
  };

  f32x16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; q++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[q][r] = 0.f;

  const float* ub = up + (long)(n_tile * 4 + wn) * NCH * 8192 + (h * 32 + j) * 4;
  const int abase = j * 4 + h * VPL;

  f32x4 B0[4][2], B1[4][2], A0[4][2], A1[4][2];
  auto ldB = [&](const float* base, int grp, f32x4 (&Bf)[4][2]) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int kk = 0; kk < 2; kk++) Bf[q][kk] = *(const f32x4*)(base + ((grp * 4 + q) * 4 + 2 * kk) * 128);
  };
  auto ldA = [&](const float* Vc, int grp, f32x4 (&Af)[4][2]) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int kk = 0; kk < 2; kk++) Af[q][kk] = *(const f32x4*)(Vc + ((grp * 4 + q) * 4 + 2 * kk) * VPL + abase);
  };
#define WINO2_GROUP(GI, ACUR, BCUR)                                                                   \
  {                                                                                                   \
    _Pragma("unroll") for (int q = 0; q < 4; q++)                                                      \
      _Pragma("unroll") for (int kk = 0; kk < 2; kk++)                                                 \
        _Pragma("unroll") for (int e = 0; e < 4; e++)                                                  \
          acc[(GI) * 4 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ACUR[q][kk][e], BCUR[q][kk][e], acc[(GI) * 4 + q], 0, 0, 0); \
  }

  stage_raw(0, raw);
  if (NCH > 1) stage_raw(1, raw + RAWSZ);
  ldB(ub, 0, B0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int b = 0; b < 4; b++) tpart(raw, b);
  twrite(V);
  __syncthreads();
  if (NCH > 2) stage_raw(2, raw);

  for (int chunk = 0; chunk < NCH; chunk++) {
    const float* Vc = V + (chunk & 1) * VBUF;
    float* Vn = V + ((chunk + 1) & 1) * VBUF;
    const float* rwn = raw + ((chunk + 1) & 1) * RAWSZ;
    const bool nxt = chunk + 1 < NCH;
    const float* uc = ub + (long)chunk * 8192;
    ldA(Vc, 0, A0);
    ldB(uc, 1, B1);
    ldA(Vc, 1, A1);
    if (nxt) tpart(rwn, 0);
    WINO2_GROUP(0, A0, B0)
    ldB(uc, 2, B0);
    ldA(Vc, 2, A0);
    if (nxt) tpart(rwn, 1);
    WINO2_GROUP(1, A1, B1)
    ldB(uc, 3, B1);
    ldA(Vc, 3, A1);
    if (nxt) tpart(rwn, 2);
    WINO2_GROUP(2, A0, B0)
    if (nxt) {
      ldB(uc + 8192, 0, B0);
      tpart(rwn, 3);
      twrite(Vn);
    }
    WINO2_GROUP(3, A1, B1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // raw of chunk+2 (staged a phase ago) and the B0 prefetch landed
    __syncthreads();                                         // V[next] complete, V[cur] and raw[(chunk+1)&1] free
    if (chunk + 3 < NCH) stage_raw(chunk + 3, raw + ((chunk + 1) & 1) * RAWSZ);
  }
#undef WINO2_GROUP

  const int co = (n_tile * 4 + wn) * 32 + j;
  const float bv = bias ? bias[co] : 0.f;
  const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int tile = (r & 3) + 8 * (r >> 2) + 4 * h;
    float T0[4], T1[4];
#pragma unroll
    for (int nu = 0; nu < 4; nu++) {
      T0[nu] = acc[0 * 4 + nu][r] + acc[1 * 4 + nu][r] + acc[2 * 4 + nu][r];
      T1[nu] = acc[1 * 4 + nu][r] - acc[2 * 4 + nu][r] - acc[3 * 4 + nu][r];
    }
    float y00 = T0[0] + T0[1] + T0[2] + bv, y01 = T0[1] - T0[2] - T0[3] + bv;
    float y10 = T1[0] + T1[1] + T1[2] + bv, y11 = T1[1] - T1[2] - T1[3] + bv;
    if (relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
    if (tile < NTILE) {
      const int ty = tile / TX, tx = tile % TX;
      const int R = R0 + 2 * ty;
      if (R < Rtot) {
        if (POOL) {
          const int n = R / H, yy = R % H;
          const long opix = ((long)n * Ho + (yy >> 1)) * Wo + ((x0 >> 1) + tx);
          y[opix * Cout + co] = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11));
        } else {
          float* o0 = y + ((long)R * W + x0 + 2 * tx) * Cout + co;
          o0[0] = y00; o0[Cout] = y01;
          o0[(long)W * Cout] = y10; o0[(long)W * Cout + Cout] = y11;
        }
      }
    }
  }
}

static int max_boundaries_w(int H, int TH, long Rtot) {
  int best = 0;
  long lim = Rtot < (long)H * TH * 2 ? Rtot : (long)H * TH * 2;
  for (long R0 = 0; R0 < lim; R0 += TH) {
    int g = (int)((R0 + TH - 1) / H - R0 / H);
    if (g > best) best = g;
  }
  return best;
}

// patch selection: 0 unsupported; Cout % 128 == 0 -> 32-tile patches (1x4 waves); Cout == 64 -> 64-tile patches (2x2 waves)
static int wino_select(int W, int Cout) {
  if (Cout % 128 == 0) {
    if (W % 16 == 0) return 1;   // (8,16)
    if (W % 8 == 0) return 2;    // (16,8)
    if (W % 4 == 0) return 3;    // (32,4)
    if (W % 14 == 0) return 4;   // (8,14): 28 of 32 tile slots
    return 0;
  }
  if (Cout % 64 == 0) {
    if (W % 32 == 0) return 5;   // (8,32)
    if (W % 16 == 0) return 6;   // (16,16)
    if (W % 8 == 0) return 7;    // (32,8)
  }
  return 0;
}
static int wino_th(int sel) {
  switch (sel) { case 1: return 8; case 2: return 16; case 3: return 32; case 4: return 8; case 5: return 8; case 6: return 16; default: return 32; }
}

template <int TH, int TW, int WM, int WN>
static int launch_wino(const float* x, const float* up, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                       int relu, int pool, int G, hipStream_t st) {
  constexpr int RS = TW + 2;
  constexpr int AROUNDS = ((TH + 4) * RS + 63) / 64;
  constexpr int RAWSZ = AROUNDS * 64 * 16;
  constexpr int VSZ = 16 * 4 * 32 * WM * 4;
  size_t lds = (size_t)(2 * RAWSZ + VSZ) * sizeof(float);
  long Rtot = (long)B * H;
  int mtiles = cdiv(Rtot, TH) * (W / TW);
  int ntiles = Cout / (32 * WN);
  dim3 grid(mtiles * ntiles);
  if (pool) {
    auto k = conv3x3_wino_kernel<TH, TW, WM, WN, true>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, x, up, bias, y, B, H, W, Cin, Cout, relu, G);
  } else {
    auto k = conv3x3_wino_kernel<TH, TW, WM, WN, false>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, x, up, bias, y, B, H, W, Cin, Cout, relu, G);
  }
  return lct_check_launch("conv3x3_wino");
}

#ifndef WINO_V2
#define WINO_V2 1
#endif
template <int TH, int TW>
static int launch_wino2(const float* x, const float* up, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                        int relu, int pool, int G, hipStream_t st) {
  constexpr int RS = TW + 2;
  constexpr int AROUNDS = ((TH + 4) * RS + 63) / 64;
  constexpr int RAWSZ = AROUNDS * 64 * 16;
  constexpr int VBUF = 16 * 4 * 32 * 4;
  size_t lds = (size_t)(2 * RAWSZ + 2 * VBUF) * sizeof(float);
  long Rtot = (long)B * H;
  dim3 grid(cdiv(Rtot, TH) * (W / TW) * (Cout / 128));
  if (pool) {
    auto k = conv3x3_wino2_kernel<TH, TW, true>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, x, up, bias, y, B, H, W, Cin, Cout, relu, G);
  } else {
    auto k = conv3x3_wino2_kernel<TH, TW, false>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, x, up, bias, y, B, H, W, Cin, Cout, relu, G);
  }
  return lct_check_launch("conv3x3_wino2");
}

extern "C" {

long lctvqa_conv3x3_wino_packed_elems(int Cout, int Cin) { return (long)Cout * Cin * 16; }

int lctvqa_conv3x3_wino_supported(int B, int H, int W, int Cin, int Cout, int pool) {
  if (Cin % 16 || Cout % 64 || (H & 1) || (W & 1)) return 0;
  const int sel = wino_select(W, Cout);
  if (!sel) return 0;
  if (max_boundaries_w(H, wino_th(sel), (long)B * H) > 2) return 0;
  if ((long)B * H * W * Cin >= (1L << 31)) return 0;
  return 1;
}

// flip_transpose = 0: forward weights (w OIHW [Cout][Cin][3][3]);  1: weights of the dgrad convolution, in which case
// Cout/Cin are the dgrad conv's OWN output/input channel counts (= the forward layer's Cin/Cout) and w is still the
// forward OIHW parameter [Cin][Cout][3][3].
int lctvqa_conv3x3_wino_pack_weights(const float* w_oihw, float* u_packed, int Cout, int Cin, int flip_transpose, void* stream) {
  LCT_REQUIRE(Cin % 16 == 0 && Cout % 32 == 0, "wino_pack_weights: need Cin%%16==0, Cout%%32==0 (got %d,%d)", Cin, Cout);
  long total = (long)Cout * Cin * 16;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, u_packed, Cout, Cin, flip_transpose);
  return lct_check_launch("wino_pack");
}

int lctvqa_conv3x3_wino_nhwc(const float* x, const float* u_packed, const float* bias, float* y, int B, int H, int W, int Cin,
                             int Cout, int relu, int pool, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  LCT_REQUIRE(lctvqa_conv3x3_wino_supported(B, H, W, Cin, Cout, pool), "conv3x3_wino: unsupported shape B=%d H=%d W=%d Cin=%d Cout=%d", B, H,
              W, Cin, Cout);
  const int sel = wino_select(W, Cout);
  const int G = max_boundaries_w(H, wino_th(sel), (long)B * H);
  switch (sel) {
#if WINO_V2
    case 1: return launch_wino2<8, 16>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 2: return launch_wino2<16, 8>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 3: return launch_wino2<32, 4>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 4: return launch_wino2<8, 14>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
#else
    case 1: return launch_wino<8, 16, 1, 4>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 2: return launch_wino<16, 8, 1, 4>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 3: return launch_wino<32, 4, 1, 4>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 4: return launch_wino<8, 14, 1, 4>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
#endif
    case 5: return launch_wino<8, 32, 2, 2>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 6: return launch_wino<16, 16, 2, 2>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 7: return launch_wino<32, 8, 2, 2>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    default: break;
  }
  lct_set_error("conv3x3_wino: no tiling");
  return LCT_EUNSUPPORTED;
}

}  // extern "C"
