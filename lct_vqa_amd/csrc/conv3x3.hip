// 3x3 / stride 1 / pad 1 convolution for the VGG19 trunk on gfx950, fp32 in / fp32 accumulate.
//
// Replaces the cuDNN/ATen dispatch behind torchvision vgg19.features as called at
// reference basic_vqa/models.py:36 and basic_vqa/models_lct.py:61-63.
//
// Design (MI355X-first, not a GEMM library call, no im2col buffer):
//  * activations are NHWC fp32 in HBM ("stacked rows": R = n*H + y, pixel = R*W + x);
//  * implicit GEMM  M = pixels, N = Cout, K = 9*Cin on v_mfma_f32_32x32x2_f32
//    (exact fp32 FMA chains, 64 FLOP/clk/SIMD -> 157.3 TFLOP/s chip peak);
//  * a workgroup (4 waves) owns a TH x TW patch of output pixels and BN = 32*WN output
//    channels.  For every 16-channel chunk of Cin the input HALO patch
//    ((TH+2+gaps) x (TW+2) pixels x 16 ch, zero padding materialised at staging time)
//    is brought HBM->LDS once with global_load_lds (no VGPR round trip) and reused by all
//    9 filter taps; each (chunk, tap) step streams one pre-packed 2 KB x WN weight tile;
//  * image boundaries inside a patch of stacked rows are handled by giving every image
//    one shared all-zero LDS row after its last row ("expanded" row coordinate
//    E = n*(H+1)+y), so the inner loop has no masks and no per-tap address math beyond
//    one add + a 2-bit XOR swizzle;
//  * LDS tiles are PLANAR per 16-B channel chunk: halo [4 chunks][pixel][4 ch], weights
//    [4 chunks][32 cout][4 ch].  A lane's ds_read_b128 address is then
//    (per-lane base) + (compile-time immediate for tap / k-group / buffer): consecutive
//    lanes read consecutive 16-B slots (bank-conflict free) and the inner loop carries no
//    address arithmetic at all.  The four waves of a workgroup each gather one chunk plane
//    of the same 64 pixels, so the 64-B pixel segments are fetched from L2 once;
//  * double-buffered halo + weight tiles, one barrier per (chunk, tap) step (56-64 MFMAs
//    = ~4000 cycles per wave), two workgroups per CU so one computes while the other waits.
//    (Measured A/B on one device, tools/ab_conv.py: per-wave private weight rings without the
//    per-step barrier, a 3-deep ring with counted vmcnt, and e-outer MFMA ordering were all
//    2-3 % SLOWER than this simple structure; they are not used.);
//  * n-tile = blockIdx % NT, so (blocks are dealt round-robin over the 8 XCDs) every XCD
//    streams the weights of a single n-tile and keeps them in its private 4 MiB L2.
//  * epilogue fuses bias + ReLU (+ 2x2 max-pool when requested) and writes NHWC.
#include "common.h"

__device__ __attribute__((aligned(64))) float g_zero_page[64];

// ----------------------------------------------------------------------------------------
// weight packing:  OIHW [Cout][Cin][3][3]  ->  [Cout/32][S = (Cin/16)*9][4 chunks][32 cout][4 ch]
// Runs every call (parameters are read by pointer each time: the architects of the
// reference overwrite parameters in place between forwards, architect_lct.py:142-155,194-207).
// ----------------------------------------------------------------------------------------
__global__ void conv3x3_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin) {
  const long total = (long)Cout * Cin * 9;
  const int S = (Cin / 16) * 9;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    int e = o & 3;
    int j = (o >> 2) & 31;
    int c = (o >> 7) & 3;
    long rest = o >> 9;
    int s = rest % S;
    int nt = rest / S;
    int chunk = s / 9, tap = s % 9;
    int co = nt * 32 + j;
    int ci = chunk * 16 + c * 4 + e;
    wp[o] = w[((long)co * Cin + ci) * 9 + tap];
  }
}

// Pixel mapping inside a 32-row MFMA sub-tile.  A sub-tile covers SH x SW pixels (SH*SW = 32).
// Row rr = 8a + 4hh + b of the MFMA C/D layout (reg = 4a + b, hh = lane>>5) is mapped so that the
// four pixels of every 2x2 pooling window live in ONE lane (dx = b&1, dy = a&1): the fused
// max-pool is then four in-register fmaxf, no LDS, no cross-lane traffic.
template <int SH, int SW, int TW>
__device__ __forceinline__ void subtile_pixel(int msub, int rr, int& ty, int& tx) {
  static_assert(SH * SW == 32 && SH % 2 == 0 && SW % 2 == 0, "sub-tile shape");
  const int a = rr >> 3, hh = (rr >> 2) & 1, b = rr & 3;
  const int u = (a >> 1) * 4 + hh * 2 + (b >> 1);        // pooled cell 0..7
  const int cx = u % (SW / 2), cy = u / (SW / 2);
  const int sx = 2 * cx + (b & 1), sy = 2 * cy + (a & 1);
  constexpr int SPR = TW / SW;                            // sub-tiles per patch row
  ty = (msub / SPR) * SH + sy;
  tx = (msub % SPR) * SW + sx;
}

template <int TH, int TW, int RS, int SH, int SW, int WM, int WN, int MT, bool POOL>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ y, int Btot, int H, int W, int Cin, int Cout, int relu, int G) {
  static_assert(TH * TW == WM * MT * 32, "tile mismatch");
  static_assert(TW % SW == 0 && TH % SH == 0, "sub-tile must tile the patch");
  static_assert(WM * WN == 4, "4 waves");
  static_assert(RS >= TW + 2, "LDS row stride must cover the halo row");
  constexpr int TWP = RS;                             // LDS row stride in pixels (>= TW+2; padded where it dodges bank conflicts)
  constexpr int HRMAX = TH + 4;                       // TH + 2 halo + up to 2 gap rows
  constexpr int PMAX = HRMAX * TWP;                   // halo pixels
  constexpr int AROUNDS = (PMAX + 63) / 64;           // each wave gathers one chunk plane of 64 pixels per round
  constexpr int PPAD = AROUNDS * 64;                  // pixels per plane (padded)
  constexpr int PLANE = PPAD * 4;                     // floats per chunk plane
  constexpr int ASZ = 4 * PLANE;                      // floats per halo buffer
  constexpr int WSZ = WN * 512;                       // floats per weight step tile
  constexpr int WCHUNKS = WN * 128;                   // 16-B chunks per weight step tile

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* abuf = smem;                // [2][ASZ]
  float* wbuf = smem + 2 * ASZ;      // [2][WSZ]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int j = lane & 31, h = lane >> 5;

  const int NT = Cout / (32 * WN);
  const int n_tile = blockIdx.x % NT;
  const int m_tile = blockIdx.x / NT;
  const int CT = W / TW;
  const int ct = m_tile % CT, rt = m_tile / CT;
  const int R0 = rt * TH, x0 = ct * TW;
  const int Rtot = Btot * H;
  const int n0img = R0 / H, y0 = R0 % H;
  const int NCH = Cin / 16;
  const int S = NCH * 9;
  const int PLIVE = (TH + 2 + G) * TWP;               // live halo pixels

  // ---- per-lane source offsets of the halo staging (same for every channel chunk) ----
  // round r: this lane gathers 16 B (channel chunk = wave id) of halo pixel p = r*64 + lane
  int aoff[AROUNDS];
#pragma unroll
  for (int r = 0; r < AROUNDS; r++) {
    int p = r * 64 + lane;
    int hr = p / TWP, hx = p % TWP;
    int xx = x0 - 1 + hx;
    int yv = y0 - 1 + hr;
    int n = n0img;
    while (yv > H) { yv -= (H + 1); n++; }
    bool ok = (p < PLIVE) && (hx < TW + 2) && (yv >= 0) && (yv < H) && (n < Btot) && (xx >= 0) && (xx < W);
    aoff[r] = ok ? (((n * H + yv) * W + xx) * Cin + wave * 4) : -1;
  }
  // ---- per-lane LDS pixel index (tap (0,0)) of each M-subtile row this lane feeds ----
  int base[MT];
#pragma unroll
  for (int m = 0; m < MT; m++) {
    int ty, tx;
    subtile_pixel<SH, SW, TW>(wm * MT + m, j, ty, tx);
    int yy = y0 + ty;
    int g = (yy >= H) + (yy >= 2 * H);
    base[m] = ((ty + g) * TWP + tx) * 4 + h * PLANE;     // float offset inside a halo buffer
  }
  const int bbase = wn * 512 + h * 128 + j * 4;           // float offset inside a weight step tile
  const float* wsrc = wp + (long)(n_tile * WN) * S * 512;   // [WN pieces][S][512] -> piece stride S*512

  auto stage_a = [&](int chunk, float* dst) {
#pragma unroll
    for (int r = 0; r < AROUNDS; r++) {
      if (r * 64 < PLIVE) {
        const float* src = (aoff[r] >= 0) ? (x + (long)aoff[r] + chunk * 16) : g_zero_page;
        glds16(src, dst + wave * PLANE + r * 256);
      }
    }
  };
  auto stage_w = [&](int s, float* dst) {
#pragma unroll
    for (int r = 0; r < (WCHUNKS + 255) / 256; r++) {
      int q = r * 256 + tid;
      if (WCHUNKS >= 256 || tid < WCHUNKS) {
        int piece = q >> 7, within = q & 127;
        glds16(wsrc + ((long)piece * S + s) * 512 + within * 4, dst + (r * 256 + wave * 64) * 4);
      }
    }
  };

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[m][r] = 0.f;

  stage_a(0, abuf);
  stage_w(0, wbuf);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int s = 0;
  for (int chunk = 0; chunk < NCH; chunk++) {
    const float* ab = abuf + (chunk & 1) * ASZ;
#pragma unroll
    for (int tap = 0; tap < 9; tap++, s++) {
      if (s + 1 < S) stage_w(s + 1, wbuf + ((s + 1) & 1) * WSZ);
      if (tap == 0 && chunk + 1 < NCH) stage_a(chunk + 1, abuf + ((chunk + 1) & 1) * ASZ);
      const float* wb = wbuf + (s & 1) * WSZ + bbase;
      const int tapoff = ((tap / 3) * TWP + (tap % 3)) * 4;
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        const f32x4 bf = *(const f32x4*)(wb + kk * 256);
#pragma unroll
        for (int m = 0; m < MT; m++) {
          const f32x4 af = *(const f32x4*)(ab + base[m] + tapoff + kk * 2 * PLANE);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc[m], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue: bias + ReLU (+ in-register 2x2 max pool), NHWC store ----
  const int co = (n_tile * WN + wn) * 32 + j;
  const float bv = bias ? bias[co] : 0.f;
  if (!POOL) {
#pragma unroll
    for (int m = 0; m < MT; m++) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
        int ty, tx;
        subtile_pixel<SH, SW, TW>(wm * MT + m, rr, ty, tx);
        const int R = R0 + ty;
        if (R < Rtot) {
          float v = acc[m][r] + bv;
          if (relu) v = fmaxf(v, 0.f);
          y[((long)R * W + x0 + tx) * Cout + co] = v;
        }
      }
    }
  } else {
    const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
    for (int m = 0; m < MT; m++) {
#pragma unroll
      for (int ap = 0; ap < 2; ap++) {
#pragma unroll
        for (int bp = 0; bp < 2; bp++) {
          const int r00 = 8 * ap + 2 * bp;            // reg of (dy=0,dx=0): a = 2ap, b = 2bp
          float v = fmaxf(fmaxf(acc[m][r00], acc[m][r00 + 1]), fmaxf(acc[m][r00 + 4], acc[m][r00 + 5])) + bv;
          if (relu) v = fmaxf(v, 0.f);
          const int rr = (2 * bp) + 8 * (2 * ap) + 4 * h;   // row of the window's (0,0) pixel
          int ty, tx;
          subtile_pixel<SH, SW, TW>(wm * MT + m, rr, ty, tx);
          const int R = R0 + ty;
          if (R < Rtot) {
            const int n = R / H, yy = R % H;
            const long opix = ((long)n * Ho + (yy >> 1)) * Wo + ((x0 + tx) >> 1);
            y[opix * Cout + co] = v;
          }
        }
      }
    }
  }
}

// naive fallback for shapes the MFMA kernel does not tile (tiny test images, Cin % 16 != 0)
__global__ void conv3x3_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                      const float* __restrict__ bias, float* __restrict__ y, int B, int H,
                                      int W, int Cin, int Cout, int relu) {
  long total = (long)B * H * W * Cout;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    int co = o % Cout;
    long pix = o / Cout;
    int xx = pix % W;
    int yy = (pix / W) % H;
    int n = pix / ((long)W * H);
    float a = bias ? bias[co] : 0.f;
    for (int ky = 0; ky < 3; ky++) {
      int iy = yy + ky - 1;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; kx++) {
        int ix = xx + kx - 1;
        if (ix < 0 || ix >= W) continue;
        const float* xp = x + (((long)n * H + iy) * W + ix) * Cin;
        const float* wq = w + (long)co * Cin * 9 + ky * 3 + kx;
        for (int ci = 0; ci < Cin; ci++) a = fmaf(xp[ci], wq[(long)ci * 9], a);
      }
    }
    if (relu) a = fmaxf(a, 0.f);
    y[o] = a;
  }
}

// ----------------------------------------------------------------------------------------
// first layer (Cin = 3, K = 27): bandwidth-bound, VALU.  NCHW image in -> NHWC out.
// lane = output channel (Cout = 64 = one wave), a wave walks a strip of pixels of one row;
// the 3 x 3 x (W+2) input rows of the strip are staged zero-padded in LDS and read as
// wave-uniform broadcasts.
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3x3_first_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ y, int B, int H, int W,
                                                            int Cout, int relu) {
  constexpr int PX = 8;                  // output pixels per wave iteration (register blocking)
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [3 ch][3 rows][WP], index i <-> x = i-1
  const int WP = (W + 2 + PX + 3) & ~3;  // padded (multiple of 4 floats: rows stay 16-B aligned; tail over-reads zeros)
  const int R = blockIdx.x;              // stacked row n*H + y
  const int n = R / H, yy = R % H;
  for (int i0 = 0; i0 < WP; i0 += blockDim.x) {       // all 9 row loads in flight before the first LDS store
    const int i = i0 + threadIdx.x, xx = i - 1;
    float v[9];
#pragma unroll
    for (int cr = 0; cr < 9; cr++) {
      const int c = cr / 3, iy = yy + (cr % 3) - 1;
      const bool ok = (iy >= 0) && (iy < H) && (xx >= 0) && (xx < W);
      const long off = (((long)n * 3 + c) * H + (ok ? iy : 0)) * W + (ok ? xx : 0);
      const float t = x[off];
      v[cr] = ok ? t : 0.f;
    }
    if (i < WP) {
#pragma unroll
      for (int cr = 0; cr < 9; cr++) smem[cr * WP + i] = v[cr];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float* wsm = smem + 9 * WP;            // [27][64] weights of the current 64-channel group, cout fastest
  for (int cg = 0; cg < Cout; cg += 64) {
    const int co = cg + lane;
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * 64; i += blockDim.x) {
      const int k = i >> 6, c = min(cg + (i & 63), Cout - 1);
      wsm[i] = w[c * 27 + k];                                      // [co][ci][ky][kx]
    }
    __syncthreads();
    const float bv = (bias && co < Cout) ? bias[co] : 0.f;
    for (int x0 = wave * PX; x0 < W; x0 += nw * PX) {
      float a[PX];
#pragma unroll
      for (int i = 0; i < PX; i++) a[i] = bv;
#pragma unroll 1
      for (int cr = 0; cr < 9; cr++) {                   // rolled: keeps ~30 live VGPRs -> 8 waves/SIMD
        const float* row = smem + cr * WP + x0;          // wave-uniform, 16-B aligned: 3 broadcast LDS reads
        const f32x4 i0 = *(const f32x4*)row, i1 = *(const f32x4*)(row + 4), i2 = *(const f32x4*)(row + 8);
        float in[12];
#pragma unroll
        for (int i = 0; i < 4; i++) { in[i] = i0[i]; in[4 + i] = i1[i]; in[8 + i] = i2[i]; }
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const float wv = wsm[(cr * 3 + k) * 64 + lane];
#pragma unroll
          for (int i = 0; i < PX; i++) a[i] = fmaf(in[i + k], wv, a[i]);
        }
      }
      if (co < Cout) {
#pragma unroll
        for (int i = 0; i < PX; i++)
          if (x0 + i < W) {
            float v = relu ? fmaxf(a[i], 0.f) : a[i];
            y[((long)R * W + x0 + i) * Cout + co] = v;
          }
      }
    }
  }
}

// The first layer on the matrix cores in exact fp32 (v_mfma_f32_32x32x2_f32: fp32 FMA chains): implicit GEMM with K = 27 (c, ky, kx)
// padded to 28 = 14 MFMA steps per 32-pixel x 32-channel tile.  The VALU kernel above spends 27 FMAs per output at the scalar-FMA rate
// (22 G multiply-adds per B=256 batch = 0.57 ms at peak, 0.85 ms measured); here the multiply-adds cost 0.3 ms of MFMA time under the
// 3.29 GB of output the layer has to write, which becomes its bound.  Workgroup = 4 waves = 8 output rows x 128 columns x 64 channels;
// wave w owns columns 32w .. 32w+31.  The zero-padded image rows sit in LDS as [3][6][RL] (index j <-> ix = ox0 + j - 1); a lane's A
// value of step s is the pixel's tap k = 2s + (lane >> 5), one ds_read_b32 (consecutive lanes, consecutive addresses); its B values
// (the weights of its output channel for the two taps of each step, both 32-channel halves) live in 28 registers for the whole kernel.
constexpr int FM_ROWS = 8, FM_NR = FM_ROWS + 2, FM_RL = 132, FM_TS = 68;
__global__ __launch_bounds__(256, 2) void conv3x3_first_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ bias, float* __restrict__ y, int B, int H,
                                                                    int W, int relu) {
  __shared__ float xin[3 * FM_NR * FM_RL];
  __shared__ __attribute__((aligned(16))) float tile[4 * 32 * FM_TS];
  const int gx = (W + 127) / 128, gy = (H + FM_ROWS - 1) / FM_ROWS;
  const int xh = blockIdx.x % gx, yb = (blockIdx.x / gx) % gy, n = blockIdx.x / (gx * gy);
  const int ox0 = 128 * xh, oy0 = FM_ROWS * yb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  // image rows: every load issued before the first LDS store
  constexpr int NXI = (3 * FM_NR * FM_RL + 255) / 256;
  float xv[NXI];
#pragma unroll
  for (int k = 0; k < NXI; k++) {
    const int i = tid + 256 * k;
    const int jj = i % FM_RL, rr = (i / FM_RL) % FM_NR, c = i / (FM_RL * FM_NR);
    const int iy = oy0 - 1 + rr, ix = ox0 + jj - 1;
    const bool ok = i < 3 * FM_NR * FM_RL && iy >= 0 && iy < H && ix >= 0 && ix < W;
    xv[k] = ok ? x[(((long)n * 3 + c) * H + iy) * W + ix] : 0.f;
  }
  // weights through LDS: a coalesced copy of the 64 x 27 floats, then this lane's 28 values wr[t][s] = w[32 t + j][k = 2 s + h]
  // (k = 27: zero) -- fetched straight from global each lane's loads touch 64 cache lines per instruction
  float wv[7];
#pragma unroll
  for (int k = 0; k < 7; k++) wv[k] = (tid + 256 * k < 64 * 27) ? w[tid + 256 * k] : 0.f;
#pragma unroll
  for (int k = 0; k < NXI; k++)
    if (tid + 256 * k < 3 * FM_NR * FM_RL) xin[tid + 256 * k] = xv[k];
#pragma unroll
  for (int k = 0; k < 7; k++)
    if (tid + 256 * k < 64 * 27) tile[tid + 256 * k] = wv[k];      // the output tile's space: free until the first row is done
  __syncthreads();
  float wr[2][14];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int s = 0; s < 14; s++) {
      const int k = 2 * s + h;
      wr[t][s] = k < 27 ? tile[(32 * t + j) * 27 + k] : 0.f;      // stride 27 floats between lanes: conflict-free (27 is odd)
    }
  __syncthreads();                                               // every wave has its weights before any wave reuses the tile
  const int lx = 32 * wave + j;                                  // this lane's pixel column inside the 128-column chunk
  if (ox0 + 32 * wave >= W) return;                              // a wave entirely right of the image (no barrier follows)
  // LDS offset of tap k = 2 s + h relative to (row oyl, column lx): (c * FM_NR + ky) * FM_RL + kx
  int koff[14];
#pragma unroll
  for (int s = 0; s < 14; s++) {
    const int k = min(2 * s + h, 26);
    koff[s] = ((k / 9) * FM_NR + (k % 9) / 3) * FM_RL + k % 3 + lx;
  }
  const float b0 = bias ? bias[j] : 0.f, b1 = bias ? bias[32 + j] : 0.f;
  for (int oyl = 0; oyl < FM_ROWS; oyl++) {
    const int oy = oy0 + oyl;
    if (oy >= H) break;                                          // uniform
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = b0; acc1[r] = b1; }
    const float* xr = xin + oyl * FM_RL;
#pragma unroll
    for (int s = 0; s < 14; s++) {
      const float a = xr[koff[s]];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wr[0][s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wr[1][s], acc1, 0, 0, 0);
    }
    // D[row][col]: this lane holds channel col = j of pixels row = (r & 3) + 8 (r >> 2) + 4 h.  Stored from here a store instruction
    // would write 4 bytes per lane (two 128-byte pieces); through the wave's private LDS tile a lane stores 16 bytes of one pixel and an
    // instruction four whole 256-byte pixel rows.  LDS operations of one wave execute in order: no barrier.
    float* T = tile + wave * 32 * FM_TS;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      float v0 = acc0[r], v1 = acc1[r];
      if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      T[row * FM_TS + j] = v0;
      T[row * FM_TS + 32 + j] = v1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* yb = y + (((long)n * H + oy) * W + ox0 + 32 * wave) * 64 + 4 * (lane & 15);
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const int pr = (lane >> 4) + 4 * c;
      const f32x4 v = *(const f32x4*)(T + pr * FM_TS + 4 * (lane & 15));
      if (ox0 + 32 * wave + pr < W) *(f32x4*)(yb + (long)pr * 64) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the tile is rewritten by the next row
  }
}

__global__ void maxpool2x2_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W,
                                       int C) {
  const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
  long total = (long)B * Ho * Wo * C4;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    int c4 = o % C4;
    long pix = o / C4;
    int ox = pix % Wo;
    int oy = (pix / Wo) % Ho;
    int n = pix / ((long)Wo * Ho);
    const f32x4* p = (const f32x4*)(x + (((long)n * H + 2 * oy) * W + 2 * ox) * C) + c4;
    f32x4 a = p[0], b = p[C4], c = p[(long)W * C4], d = p[(long)W * C4 + C4];
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = fmaxf(fmaxf(a[k], b[k]), fmaxf(c[k], d[k]));
    ((f32x4*)y)[o] = r;
  }
}

// AdaptiveAvgPool2d((7,7)) + flatten in NCHW order (torchvision vgg19.avgpool + torch.flatten):
// NHWC [B,H,W,C] -> [B, C*49] with feature index c*49 + oy*7 + ox.  Identity pooling at H=W=7.
__global__ void adaptive_avgpool7_flatten_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H,
                                                 int W, int C) {
  long total = (long)B * C * 49;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    // iterate with c fastest for coalesced NHWC reads; scatter the (small) output
    int c = o % C;
    int cell = (o / C) % 49;
    int n = o / ((long)C * 49);
    int oy = cell / 7, ox = cell % 7;
    int ys = (oy * H) / 7, ye = ((oy + 1) * H + 6) / 7;
    int xs = (ox * W) / 7, xe = ((ox + 1) * W + 6) / 7;
    float a = 0.f;
    for (int iy = ys; iy < ye; iy++)
      for (int ix = xs; ix < xe; ix++) a += x[(((long)n * H + iy) * W + ix) * C + c];
    y[((long)n * C + c) * 49 + cell] = a / (float)((ye - ys) * (xe - xs));
  }
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
static int max_boundaries(int H, int TH, long Rtot) {
  // largest number of image boundaries strictly inside a tile of TH stacked rows starting at k*TH
  int best = 0;
  long lim = Rtot < (long)H * TH * 2 ? Rtot : (long)H * TH * 2;   // pattern repeats with lcm(H,TH) <= H*TH
  for (long R0 = 0; R0 < lim; R0 += TH) {
    int g = (int)((R0 + TH - 1) / H - R0 / H);
    if (g > best) best = g;
  }
  return best;
}

template <int TH, int TW, int RS, int SH, int SW, int WM, int WN, int MT>
static int launch_conv(const float* x, const float* wp, const float* bias, float* y, int B, int H, int W, int Cin,
                       int Cout, int relu, int pool, int G, hipStream_t st) {
  constexpr int TWP = RS;
  constexpr int AROUNDS = ((TH + 4) * TWP + 63) / 64;
  constexpr int ASZ = AROUNDS * 64 * 16;
  constexpr int WSZ = WN * 512;
  size_t lds = (size_t)(2 * ASZ + 2 * WSZ) * sizeof(float);
  long Rtot = (long)B * H;
  int mtiles = cdiv(Rtot, TH) * (W / TW);
  int ntiles = Cout / (32 * WN);
  dim3 grid(mtiles * ntiles);
  if (pool) {
    auto k = conv3x3_mfma_kernel<TH, TW, RS, SH, SW, WM, WN, MT, true>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, x, wp, bias, y, B, H, W, Cin, Cout, relu, G);
  } else {
    auto k = conv3x3_mfma_kernel<TH, TW, RS, SH, SW, WM, WN, MT, false>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, x, wp, bias, y, B, H, W, Cin, Cout, relu, G);
  }
  return lct_check_launch("conv3x3_mfma");
}

extern "C" {

long lctvqa_conv3x3_packed_elems(int Cout, int Cin) { return (long)Cout * Cin * 9; }

// patch selection: 0 none, 1 (8,32), 2 (8,28), 3 (16,16), 4 (16,14), 5 (32,8).  The 7-sub-tile patches
// (224 pixels) only exist in the 1x4 wave layout, i.e. for Cout % 128 == 0.
static int select_patch(int W, int Cout) {
  const bool n4 = (Cout % 128 == 0);
  if (W % 32 == 0) return 1;
  if (n4 && W % 28 == 0) return 2;
  if (W % 16 == 0) return 3;
  if (n4 && W % 14 == 0) return 4;
  if (W % 8 == 0) return 5;
  return 0;
}
static int patch_th(int sel) { return sel == 1 || sel == 2 ? 8 : (sel == 5 ? 32 : 16); }

// returns 1 when the MFMA kernel tiles this shape, 0 when the direct fallback is used
int lctvqa_conv3x3_uses_mfma(int B, int H, int W, int Cin, int Cout, int pool) {
  if (Cin % 16 || Cout % 64) return 0;
  if (pool && ((H & 1) || (W & 1))) return 0;
  const int sel = select_patch(W, Cout);
  if (!sel) return 0;
  if (max_boundaries(H, patch_th(sel), (long)B * H) > 2) return 0;
  if ((long)B * H * W * Cin >= (1L << 31)) return 0;      // 32-bit element offsets in the halo staging
  return 1;
}

int lctvqa_conv3x3_pack_weights(const float* w_oihw, float* w_packed, int Cout, int Cin, void* stream) {
  LCT_REQUIRE(Cin % 16 == 0 && Cout % 32 == 0, "conv3x3_pack_weights: need Cin%%16==0, Cout%%32==0 (got %d,%d)", Cin, Cout);
  long total = (long)Cout * Cin * 9;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv3x3_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, w_packed, Cout, Cin);
  return lct_check_launch("conv3x3_pack");
}

// x NHWC [B,H,W,Cin] -> y NHWC [B,H,W,Cout] (or [B,H/2,W/2,Cout] when pool).  w_oihw is the
// PyTorch parameter, w_packed the output of lctvqa_conv3x3_pack_weights (may be NULL when
// the shape falls back to the direct kernel).
int lctvqa_conv3x3_nhwc(const float* x, const float* w_oihw, const float* w_packed, const float* bias, float* y,
                        float* pool_tmp, int B, int H, int W, int Cin, int Cout, int relu, int pool, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  LCT_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_nhwc: bad shape");
  LCT_REQUIRE(!pool || (!(H & 1) && !(W & 1)), "conv3x3_nhwc: pooling needs even H,W");
  if (!lctvqa_conv3x3_uses_mfma(B, H, W, Cin, Cout, pool)) {
    float* dst = pool ? pool_tmp : y;
    LCT_REQUIRE(dst != nullptr, "conv3x3_nhwc: pool_tmp required for the fallback path with pooling");
    long total = (long)B * H * W * Cout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(conv3x3_direct_kernel, dim3(blocks), dim3(256), 0, st, x, w_oihw, bias, dst, B, H, W, Cin, Cout, relu);
    int rc = lct_check_launch("conv3x3_direct");
    if (rc) return rc;
    if (pool) {
      long t2 = (long)B * (H / 2) * (W / 2) * (Cout / 4);
      int b2 = (int)((t2 + 255) / 256);
      if (b2 > 8192) b2 = 8192;
      if (b2 < 1) b2 = 1;
      hipLaunchKernelGGL(maxpool2x2_nhwc_kernel, dim3(b2), dim3(256), 0, st, dst, y, B, H, W, Cout);
      return lct_check_launch("maxpool2x2");
    }
    return LCT_OK;
  }
  LCT_REQUIRE(w_packed != nullptr, "conv3x3_nhwc: packed weights required");
  long Rtot = (long)B * H;
  const bool n4 = (Cout % 128 == 0);
#define LAUNCH(TH, TW, RS, SH, SW)                                                                                          \
  do {                                                                                                          \
    int G = max_boundaries(H, TH, Rtot);                                                                        \
    constexpr int BMs = (TH * TW) / 32;                                                                         \
    if (n4) return launch_conv<TH, TW, RS, SH, SW, 1, 4, BMs>(x, w_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st); \
    if constexpr (BMs % 2 == 0)                                                                                 \
      return launch_conv<TH, TW, RS, SH, SW, 2, 2, BMs / 2>(x, w_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);   \
  } while (0)
  switch (select_patch(W, Cout)) {
    case 1: LAUNCH(8, 32, 34, 2, 16); break;
    case 2: LAUNCH(8, 28, 30, 8, 4); break;
    case 3: LAUNCH(16, 16, 18, 2, 16); break;
    case 4: LAUNCH(16, 14, 18, 16, 2); break;
    case 5: LAUNCH(32, 8, 10, 4, 8); break;
    default: break;
  }
#undef LAUNCH
  lct_set_error("conv3x3_nhwc: no tiling for H=%d W=%d Cin=%d Cout=%d", H, W, Cin, Cout);
  return LCT_EUNSUPPORTED;
}

int lctvqa_conv3x3_first_nchw(const float* x_nchw, const float* w_oihw, const float* bias, float* y_nhwc, int B,
                              int H, int W, int Cout, int relu, void* stream) {
  LCT_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0, "conv3x3_first: bad shape");
  static int use_mfma = -1;                                      // LCTVQA_FIRST_MFMA=0: the VALU kernel (A/B)
  if (use_mfma < 0) { const char* e = getenv("LCTVQA_FIRST_MFMA"); use_mfma = (e && e[0] == '0') ? 0 : 1; }
  if (use_mfma && Cout == 64 && W >= 32) {
    const long grid = (long)B * ((H + FM_ROWS - 1) / FM_ROWS) * ((W + 127) / 128);
    LCT_REQUIRE(grid < (1L << 31), "conv3x3_first: grid too large");
    hipLaunchKernelGGL(conv3x3_first_mfma_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x_nchw, w_oihw, bias, y_nhwc,
                       B, H, W, relu);
    return lct_check_launch("conv3x3_first_mfma");
  }
  size_t lds = (size_t)(9 * ((W + 2 + 8 + 3) & ~3) + 27 * 64) * sizeof(float);
  LCT_REQUIRE(lds <= 64 * 1024, "conv3x3_first: W too large");
  hipLaunchKernelGGL(conv3x3_first_kernel, dim3(B * H), dim3(256), lds, (hipStream_t)stream, x_nchw, w_oihw, bias,
                     y_nhwc, B, H, W, Cout, relu);
  return lct_check_launch("conv3x3_first");
}

int lctvqa_maxpool2x2_nhwc(const float* x, float* y, int B, int H, int W, int C, void* stream) {
  LCT_REQUIRE(C % 4 == 0 && H >= 2 && W >= 2, "maxpool2x2: need C%%4==0 and H,W >= 2 (odd sizes floor like nn.MaxPool2d)");
  long t = (long)B * (H / 2) * (W / 2) * (C / 4);
  int b = (int)((t + 255) / 256);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(maxpool2x2_nhwc_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C);
  return lct_check_launch("maxpool2x2");
}

int lctvqa_adaptive_avgpool7_flatten(const float* x_nhwc, float* y, int B, int H, int W, int C, void* stream) {
  long t = (long)B * C * 49;
  int b = (int)((t + 255) / 256);
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(adaptive_avgpool7_flatten_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, x_nhwc, y, B, H, W, C);
  return lct_check_launch("adaptive_avgpool7_flatten");
}

}  // extern "C"
