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
// Common structure of the two kernels below (v2: 128 output channels per workgroup, one wave per SIMD; v3: 64 output
// channels, two workgroups per CU):
//  * a workgroup owns 32 Winograd tiles (a TH x TW patch of output pixels); per 16-channel chunk the zero-padded raw
//    halo patch is DMA'd HBM->LDS (same planar layout, stacked-row / shared-zero-row trick as the direct kernel),
//    every thread transforms (tile, 4-channel) items B^T d B in registers and writes the 16 (xi,nu) planes
//    V[xi,nu][k-quad][tile][4ch];
//  * the MFMA phase reads V fragments with ds_read_b128 (consecutive tiles -> conflict free) and takes the
//    pre-transformed weights U = G g G^T STRAIGHT FROM L2 INTO REGISTERS (2 KB per (cout group, chunk, xi,nu),
//    lane-contiguous float4s): no weight staging in LDS;
//  * epilogue: the (xi,nu) values of a (tile, cout) sit in ONE lane (same register index of the accumulator
//    tiles), so A^T M A, bias, ReLU and the 2x2 max-pool (= max over the tile's 4 outputs) are in-register;
//    NHWC stores are 128-B contiguous per half wave.
// Measured (profiles/r01_wino_*): the MFMA pipe is busy 73-75 % of the time; the rest is instruction-issue cost of
// the memory operations a 32x32x2 fp32 MFMA needs per 64 cycles (weight loads 6 %, raw-patch LDS-DMA 5 %, input
// transform 8 %, prologue/epilogue 6 %, tail 2 %) -- see DESIGN.md.
#include "common.h"

__device__ __attribute__((aligned(64))) float g_zero_page_w[64];

#ifndef WINO_XCD_MAP
#define WINO_XCD_MAP 1
#endif
static inline int wino_grid(int mtiles, int ntiles) {
#if WINO_XCD_MAP
  return ((mtiles + 7) / 8) * 8 * ntiles;
#else
  return mtiles * ntiles;
#endif
}

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

#ifdef WINO_DIAG
  const long dg0 = __builtin_amdgcn_s_memtime();
  const long dr0 = __builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* raw = smem;                 // [2][RAWSZ]
  float* V = smem + 2 * RAWSZ;       // [2][VBUF]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave;
  const int j = lane & 31, h = lane >> 5;
  const int NT = Cout / 128;
#if WINO_XCD_MAP
  // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), each XCD with its own L2: the NT workgroups that
  // read the same input patch are given consecutive slots of ONE XCD so that all but the first hit its L2.
  const int n_tile = (blockIdx.x >> 3) % NT, m_tile = ((blockIdx.x >> 3) / NT) * 8 + (blockIdx.x & 7);
  if (m_tile >= ((Btot * H + TH - 1) / TH) * (W / TW)) return;
#else
  const int n_tile = blockIdx.x % NT, m_tile = blockIdx.x / NT;
#endif
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
        glds16_opaque(src, dst + wave * PLANE + r * 256);
      }
    }
  };

  // transform item: tile tt, channel chunk tc, half hlf (wave-uniform: waves 0,1 -> rows 0,1 of B^T d; waves 2,3 -> rows 2,3).
  // Both halves read three consecutive patch rows e0,e1,e2 starting at row hlf:
  //   hlf 0: rows (0,1) = (e0-e2, e1+e2);   hlf 1: rows (2,3) = (e1-e0, e0-e2).
  // The main loop is instantiated once per half (HF compile-time) so that the transform is straight-line code that
  // can be pinned, a few VALU ops at a time, between the MFMAs.  Dead tile slots (NTILE < 32) transform whatever
  // sits at the start of the plane: their accumulator rows are never stored.
  const int item = tid & 127, tt = item & 31, tc = item >> 5;
  const int hlf = __builtin_amdgcn_readfirstlane(tid >> 7);
  int tbase = tc * PLANE;
  if (tt < NTILE) {
    const int ty = tt / TX, tx = tt % TX;
    const int yy = y0 + 2 * ty;
    const int g = (yy >= H) + (yy >= 2 * H);
    tbase = tc * PLANE + ((2 * ty + g + hlf) * RS + 2 * tx) * 4;
  }
  const int vwbase = tc * VPL + tt * 4;                         // + (xn*4)*VPL
  f32x4 tA[4], tB[4];                                           // the two B^T-rows of this thread, over columns b
  f32x4 te0, te1, te2;
  auto tload = [&](const float* rw, int b) {
    const float* p0 = rw + tbase + b * 4;
    te0 = *(const f32x4*)(p0);
    te1 = *(const f32x4*)(p0 + RS * 4);
    te2 = *(const f32x4*)(p0 + 2 * RS * 4);
  };
  // quarter q of the row pass of column b: q = 0,1 -> tA components (0,1),(2,3); q = 2,3 -> tB.  The empty asm pins the
  // two results where they are computed (IR-level passes otherwise merge the quarters back into one block).
  auto tcomb_q = [&](int b, int q, int hf) {
    const int e = (q & 1) * 2;
    float r0, r1;
    if (q < 2) {
      r0 = hf ? te1[e] - te0[e] : te0[e] - te2[e];
      r1 = hf ? te1[e + 1] - te0[e + 1] : te0[e + 1] - te2[e + 1];
    } else {
      r0 = hf ? te0[e] - te2[e] : te1[e] + te2[e];
      r1 = hf ? te0[e + 1] - te2[e + 1] : te1[e + 1] + te2[e + 1];
    }
    asm volatile("" : "+v"(r0), "+v"(r1));
    if (q < 2) { tA[b][e] = r0; tA[b][e + 1] = r1; } else { tB[b][e] = r0; tB[b][e + 1] = r1; }
  };
  auto tpart = [&](const float* rw, int b) {
    tload(rw, b);
#pragma unroll
    for (int q = 0; q < 4; q++) tcomb_q(b, q, hlf);
  };
  // column pass + store of one of the 8 (xi,nu) planes this thread owns: i<4 from tA (xi = 2*hlf), else tB (xi = 2*hlf+1)
  auto twrite_part = [&](float* Vd, int i) {
    float* v = Vd + ((2 * hlf + (i >> 2)) * 4 + (i & 3)) * 4 * VPL + vwbase;
    const f32x4* t = (i < 4) ? tA : tB;
    const int nu = i & 3;
    *(f32x4*)v = nu == 0 ? t[0] - t[2] : nu == 1 ? t[1] + t[2] : nu == 2 ? t[2] - t[1] : t[1] - t[3];
  };
  auto twrite = [&](float* Vd) {
#pragma unroll
    for (int i = 0; i < 8; i++) twrite_part(Vd, i);
  };
  // transform work of the main loop, cut into slices placed between the MFMAs of step xn (slot i = 0..3):
  // column b is loaded in step 3b, row-combined in step 3b+1, and the eight planes are written in steps 12..15
  auto tslice = [&](const float* rwn, float* Vn, int xn, int i, int hf) {
    if (xn < 12 && xn % 3 == 0 && i == 0) tload(rwn, xn / 3);
    if (xn < 12 && xn % 3 == 1) tcomb_q(xn / 3, i, hf);
    if (xn >= 12 && (i & 1) == 0) twrite_part(Vn, (xn - 12) * 2 + (i >> 1));
  };

  f32x16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; q++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[q][r] = 0.f;

  const float* ub = up + (long)(n_tile * 4 + wn) * NCH * 8192 + (h * 32 + j) * 4;
  const int abase = j * 4 + h * VPL;

  // Weights: one register slot per (xi,nu), refilled with the NEXT chunk's weights right after this chunk's MFMAs
  // consumed it, so every weight load has a full chunk phase (~8000 cycles) to come back from L2.  A fragments
  // come from LDS just in time (one (xi,nu) ahead).
  f32x4 Bw[16][2];
  f32x4 Aa[2], Ab[2];
  auto ldBx = [&](const float* base, int xn) {
#pragma unroll
    for (int kk = 0; kk < 2; kk++) Bw[xn][kk] = *(const f32x4*)(base + (xn * 4 + 2 * kk) * 128);
  };
  auto ldAx = [&](const float* Vc, int xn, f32x4 (&Af)[2]) {
#pragma unroll
    for (int kk = 0; kk < 2; kk++) Af[kk] = *(const f32x4*)(Vc + (xn * 4 + 2 * kk) * VPL + abase);
  };
  // One (xi,nu): 8 MFMAs.  Everything else is pinned between them in source order (sched_barrier): left to itself the
  // scheduler bunches the transform and the loads at the end of the phase, where nothing hides them.
#define WINO2_SB() __builtin_amdgcn_sched_barrier(0)
#define WINO2_MF(XN, ACUR, KK, E) \
  acc[XN] = __builtin_amdgcn_mfma_f32_32x32x2f32(ACUR[KK][E], Bw[XN][KK][E], acc[XN], 0, 0, 0)
#define WINO2_STEP(XN, ACUR, ANEXT, NXT, HF)                                                             \
  {                                                                                                   \
    WINO2_MF(XN, ACUR, 0, 0);                                                                         \
    if ((XN) + 1 < 16) ANEXT[0] = *(const f32x4*)(Vc + (((XN) + 1) * 4) * VPL + abase); \
    WINO2_SB();                                                                                       \
    WINO2_MF(XN, ACUR, 0, 1);                                                                         \
    if ((XN) + 1 < 16) ANEXT[1] = *(const f32x4*)(Vc + (((XN) + 1) * 4 + 2) * VPL + abase); \
    WINO2_SB();                                                                                       \
    WINO2_MF(XN, ACUR, 0, 2);                                                                         \
    if (NXT) tslice(rwn, Vn, XN, 0, HF);                                                                \
    WINO2_SB();                                                                                       \
    WINO2_MF(XN, ACUR, 0, 3);                                                                         \
    if (NXT) tslice(rwn, Vn, XN, 1, HF);                                                                \
    WINO2_SB();                                                                                       \
    WINO2_MF(XN, ACUR, 1, 0);                                                                         \
    if (NXT) Bw[XN][0] = *(const f32x4*)(uc + 8192 + ((XN) * 4) * 128);               \
    WINO2_SB();                                                                                       \
    WINO2_MF(XN, ACUR, 1, 1);                                                                         \
    if (NXT) tslice(rwn, Vn, XN, 2, HF);                                                                \
    WINO2_SB();                                                                                       \
    WINO2_MF(XN, ACUR, 1, 2);                                                                         \
    if (NXT) tslice(rwn, Vn, XN, 3, HF);                                                                \
    WINO2_SB();                                                                                       \
    WINO2_MF(XN, ACUR, 1, 3);                                                                         \
    if (NXT) Bw[XN][1] = *(const f32x4*)(uc + 8192 + ((XN) * 4 + 2) * 128);           \
    WINO2_SB();                                                                                       \
  }
#define WINO2_PHASE(NXT, HF)                                                                            \
  {                                                                                                   \
    const float* Vc = V + (chunk & 1) * VBUF;                                                         \
    float* Vn = V + ((chunk + 1) & 1) * VBUF;                                                         \
    const float* rwn = raw + ((chunk + 1) & 1) * RAWSZ;                                               \
    const float* uc = ub + (long)chunk * 8192;                                                        \
    ldAx(Vc, 0, Aa);                                                                                  \
    WINO2_STEP(0, Aa, Ab, NXT, HF)  WINO2_STEP(1, Ab, Aa, NXT, HF)  WINO2_STEP(2, Aa, Ab, NXT, HF)  WINO2_STEP(3, Ab, Aa, NXT, HF)     \
    WINO2_STEP(4, Aa, Ab, NXT, HF)  WINO2_STEP(5, Ab, Aa, NXT, HF)  WINO2_STEP(6, Aa, Ab, NXT, HF)  WINO2_STEP(7, Ab, Aa, NXT, HF)     \
    WINO2_STEP(8, Aa, Ab, NXT, HF)  WINO2_STEP(9, Ab, Aa, NXT, HF)  WINO2_STEP(10, Aa, Ab, NXT, HF) WINO2_STEP(11, Ab, Aa, NXT, HF)    \
    WINO2_STEP(12, Aa, Ab, NXT, HF) WINO2_STEP(13, Ab, Aa, NXT, HF) WINO2_STEP(14, Aa, Ab, NXT, HF) WINO2_STEP(15, Ab, Aa, NXT, HF)    \
  }

  stage_raw(0, raw);
  if (NCH > 1) stage_raw(1, raw + RAWSZ);
#pragma unroll
  for (int xn = 0; xn < 16; xn++) ldBx(ub, xn);
  // builtin (not asm) so that the compiler's waitcnt bookkeeping sees the prologue loads retired: the counted waits
  // in the loop are then derived from the loop's own issue order only.  0x0F70 = vmcnt(0), expcnt/lgkmcnt untouched.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
#pragma unroll
  for (int b = 0; b < 4; b++) tpart(raw, b);
  twrite(V);
  __syncthreads();
  if (NCH > 2) stage_raw(2, raw);

#define WINO2_MAIN(HF)                                                                                \
  {                                                                                                   \
    int chunk = 0;                                                                                    \
    for (; chunk < NCH - 1; chunk++) {                                                                \
      WINO2_PHASE(true, HF)                                                                           \
      asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)\n\ts_barrier" ::: "memory");   \
      if (chunk + 3 < NCH) stage_raw(chunk + 3, raw + ((chunk + 1) & 1) * RAWSZ);   \
    }                                                                                                 \
    WINO2_PHASE(false, HF)                                                                            \
  }
  // End of a phase: the raw patch of chunk+2 (LDS-DMA issued at the start of the phase) must have landed; the 32 weight
  // loads of the next chunk, issued after it, stay in flight (vmcnt retires in issue order).  Bare barrier instead of
  // __syncthreads(), whose fence would drain that prefetch; lgkmcnt(0) retires the transform's LDS stores.  After the
  // barrier V[next] is complete and V[cur], raw[(chunk+1)&1] are free.
#ifdef WINO_DIAG
  const long dg1 = __builtin_amdgcn_s_memtime();
#endif
  if (hlf == 0) WINO2_MAIN(0) else WINO2_MAIN(1)
#ifdef WINO_DIAG
  const long dg2 = __builtin_amdgcn_s_memtime();
#endif
#undef WINO2_MAIN
#undef WINO2_PHASE
#undef WINO2_STEP
#undef WINO2_MF
#undef WINO2_SB

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
#ifdef WINO_DIAG
    if (tile < NTILE && relu == 12345) {   // diagnostic build: no stores (the stamps go to y), arithmetic stays live
#else
    if (tile < NTILE) {
#endif
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
#ifdef WINO_DIAG
  __syncthreads();
  if (tid == 0) {
    const long dg3 = __builtin_amdgcn_s_memtime();
    float* dg = y + (long)blockIdx.x * 4;
    const long dr3 = __builtin_amdgcn_s_memrealtime();
    dg[0] = (float)(dg1 - dg0); dg[1] = (float)(dg2 - dg1); dg[2] = (float)(dg3 - dg2); dg[3] = (float)(dr3 - dr0);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// v3: two workgroups per CU.  Block = 4 waves = 32 tiles x 64 output channels; wave (xh, nh) accumulates the 8
// transformed-domain GEMMs xi in {2xh, 2xh+1} (all nu) for channels nh*32..+31: 8 x 16 = 128 accumulator registers,
// so two waves fit on a SIMD and the second workgroup's MFMAs fill every bubble of the first (barriers, the input
// transform, weight-load issue, prologue and epilogue), which the one-wave-per-SIMD v2 kernel has to schedule
// around by hand.  That allows the simple structure: per 16-channel chunk
//     MFMA phase on V  ->  barrier  ->  B^T d B of the next chunk (raw -> V)  ->  barrier  ->  LDS-DMA of the chunk after,
// with V and the raw patch single-buffered (48 KB of LDS per workgroup).  Weights: one register slot per (xi,nu),
// refilled with the next chunk's weights right after use (a full phase of lead).  The two xi-halves are combined in
// the epilogue: Y = A^T M A is linear in M, each wave output-transforms its own half and the xh=1 waves hand their
// partial 2x2 tiles to the xh=0 waves through LDS.
// ---------------------------------------------------------------------------------------------------------------
template <int TH, int TW, bool POOL>
__global__ __launch_bounds__(256, 2) void conv3x3_wino3_kernel(const float* __restrict__ x, const float* __restrict__ up,
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
  constexpr int VPL = 32 * 4;                                   // floats per (xi,nu,k-quad) plane
  constexpr int VBUF = 16 * 4 * VPL;                            // 8192 floats = 32 KB

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* V = smem;                   // [16 xi,nu][4 k-quads][32 tiles][4]
  float* raw = smem + VBUF;          // [4 channel quads][PLANE]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nh = wave & 1, xh = wave >> 1;
  const int j = lane & 31, h = lane >> 5;
  const int NT = Cout / 64;
#if WINO_XCD_MAP
  // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), each XCD with its own L2: the NT workgroups that
  // read the same input patch are given consecutive slots of ONE XCD so that all but the first hit its L2.
  const int n_tile = (blockIdx.x >> 3) % NT, m_tile = ((blockIdx.x >> 3) / NT) * 8 + (blockIdx.x & 7);
  if (m_tile >= ((Btot * H + TH - 1) / TH) * (W / TW)) return;
#else
  const int n_tile = blockIdx.x % NT, m_tile = blockIdx.x / NT;
#endif
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
  auto stage_raw = [&](int chunk) {
#pragma unroll
    for (int r = 0; r < AROUNDS; r++) {
      if (r * 64 < PLIVE) {
        const float* src = (aoff[r] >= 0) ? (x + (long)aoff[r] + chunk * 16) : g_zero_page_w;
        glds16_opaque(src, raw + wave * PLANE + r * 256);
      }
    }
  };

  // input transform item: tile tt, channel quad tc, half hlf (wave-uniform): rows (0,1) or (2,3) of B^T d, then the
  // column pass, written to the 8 (xi,nu) planes xi in {2 hlf, 2 hlf + 1}
  const int item = tid & 127, tt = item & 31, tc = item >> 5;
  const int hlf = __builtin_amdgcn_readfirstlane(tid >> 7);
  int tbase = tc * PLANE;
  if (tt < NTILE) {
    const int ty = tt / TX, tx = tt % TX;
    const int yy = y0 + 2 * ty;
    const int g = (yy >= H) + (yy >= 2 * H);
    tbase = tc * PLANE + ((2 * ty + g + hlf) * RS + 2 * tx) * 4;
  }
  float* vw = V + (2 * hlf * 4) * 4 * VPL + tc * VPL + tt * 4;
  // one B^T-row at a time (16 live registers instead of 32: the accumulators and the weight slots leave little room)
  auto transform = [&]() {
#pragma unroll
    for (int rw = 0; rw < 2; rw++) {
      f32x4 t[4];
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const float* p0 = raw + tbase + b * 4;                   // patch rows hlf, hlf+1, hlf+2 at +0, +RS*4, +2*RS*4
        if (hlf == 0) {
          if (rw == 0) t[b] = *(const f32x4*)(p0) - *(const f32x4*)(p0 + 2 * RS * 4);                 // row 0: d0 - d2
          else         t[b] = *(const f32x4*)(p0 + RS * 4) + *(const f32x4*)(p0 + 2 * RS * 4);        // row 1: d1 + d2
        } else {
          if (rw == 0) t[b] = *(const f32x4*)(p0 + RS * 4) - *(const f32x4*)(p0);                     // row 2: d2 - d1
          else         t[b] = *(const f32x4*)(p0) - *(const f32x4*)(p0 + 2 * RS * 4);                 // row 3: d1 - d3
        }
      }
      float* v = vw + rw * 16 * VPL;
      *(f32x4*)(v) = t[0] - t[2];
      *(f32x4*)(v + 4 * VPL) = t[1] + t[2];
      *(f32x4*)(v + 8 * VPL) = t[2] - t[1];
      *(f32x4*)(v + 12 * VPL) = t[1] - t[3];
    }
  };

  f32x16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; q++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[q][r] = 0.f;

  // packed weights: [32-channel group][chunk][xi,nu][k-quad][32 ch][4]; this wave: group n_tile*2+nh, planes xh*8..+7
  const float* ub = up + (long)(n_tile * 2 + nh) * NCH * 8192 + (xh * 8) * 512 + (h * 32 + j) * 4;
  const float* va = V + (xh * 8) * 4 * VPL + j * 4 + h * VPL;

  // weights: ring of four (xi,nu) slots; the slot of step q is refilled right behind its MFMAs with the weights of
  // step q+4 (this chunk's for q < 4, the next chunk's q-4 otherwise): three to four steps of lead, 32 registers
  f32x4 Bw[4][2];
  f32x4 Aa[2], Ab[2];
#define WINO3_STEP(Q, ACUR, ANEXT, NXT)                                                                \
  {                                                                                                   \
    if ((Q) + 1 < 8) {                                                                                \
      ANEXT[0] = *(const f32x4*)(va + (((Q) + 1) * 4) * VPL);                                          \
      ANEXT[1] = *(const f32x4*)(va + (((Q) + 1) * 4 + 2) * VPL);                                      \
    }                                                                                                 \
    _Pragma("unroll") for (int kk = 0; kk < 2; kk++)                                                   \
      _Pragma("unroll") for (int e = 0; e < 4; e++)                                                    \
        acc[Q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ACUR[kk][e], Bw[(Q) & 3][kk][e], acc[Q], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);   /* keep the refill right behind its MFMAs */                  \
    if ((Q) < 4) {                                                                                    \
      Bw[(Q) & 3][0] = *(const f32x4*)(uc + (((Q) + 4) * 4) * 128);                                    \
      Bw[(Q) & 3][1] = *(const f32x4*)(uc + (((Q) + 4) * 4 + 2) * 128);                                \
    } else if (NXT) {                                                                                 \
      Bw[(Q) & 3][0] = *(const f32x4*)(uc + 8192 + (((Q) - 4) * 4) * 128);                             \
      Bw[(Q) & 3][1] = *(const f32x4*)(uc + 8192 + (((Q) - 4) * 4 + 2) * 128);                         \
    }                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  }
#define WINO3_PHASE(NXT)                                                                              \
  {                                                                                                   \
    const float* uc = ub + (long)chunk * 8192;                                                        \
    Aa[0] = *(const f32x4*)(va);                                                                      \
    Aa[1] = *(const f32x4*)(va + 2 * VPL);                                                            \
    WINO3_STEP(0, Aa, Ab, NXT) WINO3_STEP(1, Ab, Aa, NXT) WINO3_STEP(2, Aa, Ab, NXT) WINO3_STEP(3, Ab, Aa, NXT)  \
    WINO3_STEP(4, Aa, Ab, NXT) WINO3_STEP(5, Ab, Aa, NXT) WINO3_STEP(6, Aa, Ab, NXT) WINO3_STEP(7, Ab, Aa, NXT)  \
  }

  stage_raw(0);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    Bw[q][0] = *(const f32x4*)(ub + (q * 4) * 128);
    Bw[q][1] = *(const f32x4*)(ub + (q * 4 + 2) * 128);
  }
  // builtin (not asm): the compiler's waitcnt bookkeeping then sees the prologue loads retired and derives the counted
  // waits of the loop from the loop's own issue order.  0x0F70 = vmcnt(0), expcnt/lgkmcnt untouched.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  transform();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (NCH > 1) stage_raw(1);

  int chunk = 0;
  for (; chunk < NCH - 1; chunk++) {
    WINO3_PHASE(true)
    // raw patch of chunk+1 (LDS-DMA issued a phase ago) landed; the 16 younger weight loads stay in flight.  Bare
    // barriers: __syncthreads() would drain them.  #1: every wave is done reading V.  #2: V(chunk+1) complete, raw free.
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
    transform();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (chunk + 2 < NCH) stage_raw(chunk + 2);
  }
  WINO3_PHASE(false)
#undef WINO3_PHASE
#undef WINO3_STEP

  // ---- epilogue: partial output transform of this wave's xi-half, exchange, bias/ReLU/pool, store
  float* xch = V;                                               // [nh][r][i*2+jj][lane]: 32 KB, V is dead after the barrier
  __syncthreads();
  const int co = (n_tile * 2 + nh) * 32 + j;
  const float bv = bias ? bias[co] : 0.f;
  const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    float T0[2], T1[2];                                         // T[xi_local][jj]
    T0[0] = acc[0][r] + acc[1][r] + acc[2][r];
    T0[1] = acc[1][r] - acc[2][r] - acc[3][r];
    T1[0] = acc[4][r] + acc[5][r] + acc[6][r];
    T1[1] = acc[5][r] - acc[6][r] - acc[7][r];
    if (xh == 1) {                                              // xi = 2,3:  P[0] = T(xi=2),  P[1] = -T(2) - T(3)
      float* d = xch + ((nh * 16 + r) * 4) * 64 + lane;
      d[0] = T0[0]; d[64] = T0[1];
      d[128] = -T0[0] - T1[0]; d[192] = -T0[1] - T1[1];
    } else {                                                    // xi = 0,1:  P[0] = T(0) + T(1),  P[1] = T(1)
      acc[0][r] = T0[0] + T1[0]; acc[1][r] = T0[1] + T1[1];
      acc[2][r] = T1[0];         acc[3][r] = T1[1];
    }
  }
  __syncthreads();
  if (xh == 0) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float* d = xch + ((nh * 16 + r) * 4) * 64 + lane;
      float y00 = acc[0][r] + d[0] + bv, y01 = acc[1][r] + d[64] + bv;
      float y10 = acc[2][r] + d[128] + bv, y11 = acc[3][r] + d[192] + bv;
      if (relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
      const int tile = (r & 3) + 8 * (r >> 2) + 4 * h;
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

// kernel / patch selection: 0 unsupported; 1..4 = v2 (Cout % 128 == 0 and Cin > 64: its single 128-channel
// workgroup re-uses the transformed input twice as often), 11..14 = v3 (Cout % 64 == 0: the only option for 64
// output channels, and the faster one for short K loops, where its second workgroup hides prologue and epilogue).
// Patches (TH,TW): (8,16), (16,8), (32,4), (8,14 -> 28 of 32 tile slots).
static int wino_select(int W, int Cin, int Cout) {
  int patch = 0;
  if (W % 16 == 0) patch = 1;
  else if (W % 8 == 0) patch = 2;
  else if (W % 4 == 0) patch = 3;
  else if (W % 14 == 0) patch = 4;
  if (!patch || Cout % 64) return 0;
  return (Cout % 128 == 0 && Cin > 64) ? patch : 10 + patch;
}
static int wino_th(int sel) {
  switch (sel % 10) { case 1: return 8; case 2: return 16; case 3: return 32; default: return 8; }
}

template <int TH, int TW>
static int launch_wino2(const float* x, const float* up, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                        int relu, int pool, int G, hipStream_t st) {
  constexpr int RS = TW + 2;
  constexpr int AROUNDS = ((TH + 4) * RS + 63) / 64;
  constexpr int RAWSZ = AROUNDS * 64 * 16;
  constexpr int VBUF = 16 * 4 * 32 * 4;
  size_t lds = (size_t)(2 * RAWSZ + 2 * VBUF) * sizeof(float);
  long Rtot = (long)B * H;
  dim3 grid(wino_grid(cdiv(Rtot, TH) * (W / TW), Cout / 128));
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

template <int TH, int TW>
static int launch_wino3(const float* x, const float* up, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                        int relu, int pool, int G, hipStream_t st) {
  constexpr int RS = TW + 2;
  constexpr int AROUNDS = ((TH + 4) * RS + 63) / 64;
  constexpr int RAWSZ = AROUNDS * 64 * 16;
  constexpr int VBUF = 16 * 4 * 32 * 4;
  size_t lds = (size_t)(RAWSZ + VBUF) * sizeof(float);
  long Rtot = (long)B * H;
  dim3 grid(wino_grid(cdiv(Rtot, TH) * (W / TW), Cout / 64));
  if (pool) {
    auto k = conv3x3_wino3_kernel<TH, TW, true>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, x, up, bias, y, B, H, W, Cin, Cout, relu, G);
  } else {
    auto k = conv3x3_wino3_kernel<TH, TW, false>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, x, up, bias, y, B, H, W, Cin, Cout, relu, G);
  }
  return lct_check_launch("conv3x3_wino3");
}

extern "C" {

long lctvqa_conv3x3_wino_packed_elems(int Cout, int Cin) { return (long)Cout * Cin * 16; }

int lctvqa_conv3x3_wino_supported(int B, int H, int W, int Cin, int Cout, int pool) {
  if (Cin % 16 || Cout % 64 || (H & 1) || (W & 1)) return 0;
  const int sel = wino_select(W, Cin, Cout);
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
  const int sel = wino_select(W, Cin, Cout);
  const int G = max_boundaries_w(H, wino_th(sel), (long)B * H);
  switch (sel) {
    case 1: return launch_wino2<8, 16>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 2: return launch_wino2<16, 8>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 3: return launch_wino2<32, 4>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 4: return launch_wino2<8, 14>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 11: return launch_wino3<8, 16>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 12: return launch_wino3<16, 8>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 13: return launch_wino3<32, 4>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 14: return launch_wino3<8, 14>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    default: break;
  }
  lct_set_error("conv3x3_wino: no tiling");
  return LCT_EUNSUPPORTED;
}

}  // extern "C"
