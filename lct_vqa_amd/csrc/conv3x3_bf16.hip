// 3x3 / stride 1 / pad 1 convolution of a FROZEN VGG19 trunk in bf16 (fp32 accumulate) on gfx950: the mixed-precision
// form of the path in conv3x3.hip, for BASELINE.json's bf16 configurations.  The reference is fp32 throughout
// (basic_vqa/models.py:23-42); this is an opt-in compute dtype (ops.set_compute_dtype('bf16')) whose results are
// compared with the fp32 oracle under a stated tolerance, never the headline path.
//
// Same design as the fp32 direct kernel (conv3x3.hip): NHWC activations as "stacked rows", implicit GEMM M = pixels,
// N = Cout, K = 9*Cin with no im2col buffer, zero-padded halo patch brought HBM->LDS by LDS-DMA once per channel chunk
// and reused by the 9 taps, image boundaries inside a patch as one shared zero row per image, planar 16-byte LDS slots
// (every fragment read is ds_read_b128 at base + immediate), fused bias + ReLU + 2x2 max-pool epilogue.  What changes:
//   * a 16-byte slot holds 8 bf16 channels, a channel chunk is 32 channels (4 planes), and the contraction runs on
//     v_mfma_f32_32x32x16_bf16: lane half h of k16-step kk reads plane 2*kk + h -- 16x the MFMA rate of the fp32 form,
//     so a (chunk, tap) step is only ~500 MFMA cycles and the kernel lives on two workgroups per CU covering each
//     other's barriers, and on a wave tile with TWO 32-channel N-subtiles (4 A + 2 B fragment reads per 8 MFMAs)
//     to stay off the LDS bandwidth;
//   * weights are packed to bf16 [Cout/32][(Cin/32)*9][4 planes][32 cout][8 ch]; activations are stored bf16.
// Roofline: MFMA (dense bf16 peak ~2.5 PFLOP/s); algorithmic bytes per launch = 2 B * (B*H*W*Cin + B*Ho*Wo*Cout).
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __attribute__((aligned(64))) float g_zero_page_bf16[64];

// 1: epilogue through a per-wave LDS tile, 16-byte stores of consecutive channels; 0: 2-byte stores straight from the accumulators
#ifndef BF16_EPI_LDS
#define BF16_EPI_LDS 1
#endif
// 1: halo patch staged pixel-major with swizzled 16-byte slots (16 cache lines per DMA instruction); 0: planar, one octet plane per wave
#ifndef BF16_HALO_PX
#define BF16_HALO_PX 1
#endif

__device__ __forceinline__ unsigned short f2bf(float v) {
  const __bf16 b = (__bf16)v;                                   // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(unsigned short, b);
}

// OIHW fp32 [Cout][Cin][3][3] -> bf16 [Cout/32][S = (Cin/32)*9][4 planes][32 cout][8 ch]
// DGRAD: the packed weights of the convolution that maps dz to dx: W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx] (Cout, Cin are W''s)
template <bool DGRAD>
__global__ void conv3x3_bf16_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cout, int Cin) {
  const long total = (long)Cout * Cin * 9;
  const int S = (Cin / 32) * 9;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int e = o & 7;
    const int j = (o >> 3) & 31;
    const int c = (o >> 8) & 3;
    const long rest = o >> 10;
    const int s = rest % S;
    const int nt = rest / S;
    const int chunk = s / 9, tap = s % 9;
    const int co = nt * 32 + j;
    const int ci = chunk * 32 + c * 8 + e;
    wp[o] = DGRAD ? f2bf(w[((long)ci * Cout + co) * 9 + (8 - tap)]) : f2bf(w[((long)co * Cin + ci) * 9 + tap]);
  }
}

// pixel of row rr of M-subtile msub (see conv3x3.hip: the four pixels of a 2x2 pooling window share a lane)
template <int SH, int SW, int TW>
__device__ __forceinline__ void subtile_pixel(int msub, int rr, int& ty, int& tx) {
  static_assert(SH * SW == 32 && SH % 2 == 0 && SW % 2 == 0, "sub-tile shape");
  const int a = rr >> 3, hh = (rr >> 2) & 1, b = rr & 3;
  const int u = (a >> 1) * 4 + hh * 2 + (b >> 1);
  const int cx = u % (SW / 2), cy = u / (SW / 2);
  const int sx = 2 * cx + (b & 1), sy = 2 * cy + (a & 1);
  constexpr int SPR = TW / SW;
  ty = (msub / SPR) * SH + sy;
  tx = (msub % SPR) * SW + sx;
}

// WM x WNW waves (= 4); a wave owns MT M-subtiles (32 pixels) x NT2 N-subtiles (32 channels); BN = 32 * WNW * NT2
template <int TH, int TW, int RS, int SH, int SW, int WM, int WNW, int MT, int NT2, bool POOL>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ wp,
                                                              const float* __restrict__ bias, unsigned short* __restrict__ y, int Btot,
                                                              int H, int W, int Cin, int Cout, int relu, int G) {
  static_assert(TH * TW == WM * MT * 32, "tile mismatch");
  static_assert(TW % SW == 0 && TH % SH == 0, "sub-tile must tile the patch");
  static_assert(WM * WNW == 4, "4 waves");
  static_assert(RS >= TW + 2, "LDS row stride must cover the halo row");
  constexpr int TWP = RS;
  constexpr int HRMAX = TH + 4;
  constexpr int PMAX = HRMAX * TWP;
  constexpr int AROUNDS = (PMAX + 63) / 64;
  constexpr int PPAD = AROUNDS * 64;
  constexpr int PLANE = PPAD * 4;                     // floats (16 B per pixel) per 8-channel plane
  constexpr int ASZ = 4 * PLANE;
  constexpr int NP = WNW * NT2;                       // 32-channel weight pieces per step
  constexpr int WSZ = NP * 512;                       // floats per weight step tile (2 KB per piece)
  constexpr int WCHUNKS = NP * 128;                   // 16-B slots per weight step tile

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* abuf = smem;                // [2][ASZ]
  float* wbuf = smem + 2 * ASZ;      // [2][WSZ]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;
  const int j = lane & 31, h = lane >> 5;

  const int NT = Cout / (32 * NP);
  const int n_tile = blockIdx.x % NT;                 // blocks are dealt round-robin over the XCDs: one n-tile's weights per L2
  const int m_tile = blockIdx.x / NT;
  const int CT = W / TW;
  const int ct = m_tile % CT, rt = m_tile / CT;
  const int R0 = rt * TH, x0 = ct * TW;
  const int Rtot = Btot * H;
  const int n0img = R0 / H, y0 = R0 % H;
  const int NCH = Cin / 32;
  const int S = NCH * 9;
  const int PLIVE = (TH + 2 + G) * TWP;

  // Halo patch in LDS, two forms.  HPX (the 256-pixel patches, MT = 4): [pixel][four 16-byte slots of 8 channels], the slot XOR-swizzled
  // by (pixel>>2)&3 (conflict-free ds_read_b128 of consecutive pixels).  The DMA writes LDS linearly, so the swizzle is applied to the
  // source: a piece = 16 pixels x 4 slots, lane = (pixel lane>>2, slot lane&3) fetches channel octet (lane&3) ^ ((pixel>>2)&3): the four
  // lanes of a pixel read its 64 contiguous bytes and an instruction touches 16 cache lines (-3 ... -12 % per layer).  Planar (the
  // 224-pixel patches, MT = 7: lane = pixel, one octet plane per wave, 64 lines per instruction): their 63 per-tap fragment offsets
  // would not fit the register file beside 112 accumulators (measured: 50 spilled VGPRs, +25 %).
  constexpr bool HPX = BF16_HALO_PX && MT <= 4;
  int aoff[AROUNDS];                                  // element offset of this lane's 16-byte slot in round r (or -1)
#pragma unroll
  for (int r = 0; r < AROUNDS; r++) {
    const int p = HPX ? (wave + 4 * r) * 16 + (lane >> 2) : r * 64 + lane;
    const int sq = HPX ? ((lane & 3) ^ ((p >> 2) & 3)) : wave;
    const int hr = p / TWP, hx = p % TWP;
    const int xx = x0 - 1 + hx;
    int yv = y0 - 1 + hr;
    int n = n0img;
    while (yv > H) { yv -= (H + 1); n++; }
    const bool ok = (p < PLIVE) && (hx < TW + 2) && (yv >= 0) && (yv < H) && (n < Btot) && (xx >= 0) && (xx < W);
    aoff[r] = ok ? (((n * H + yv) * W + xx) * Cin + sq * 8) : -1;
  }
  // LDS float offset of this lane's fragment of sub-tile m: HPX: afr[m][tap] for half-step kk = 0 (kk = 1: the same offset ^ 8);
  // planar: base[m] + tap and plane offsets that are immediates
  int afr[HPX ? MT : 1][9];
  int base[MT];
#pragma unroll
  for (int m = 0; m < MT; m++) {
    int ty, tx;
    subtile_pixel<SH, SW, TW>(wm * MT + m, j, ty, tx);
    const int yy = y0 + ty;
    const int g = (yy >= H) + (yy >= 2 * H);
    base[m] = ((ty + g) * TWP + tx) * 4 + h * PLANE;
    if (HPX) {
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const int P = (ty + g + t / 3) * TWP + tx + t % 3;
        afr[HPX ? m : 0][t] = P * 16 + ((h ^ ((P >> 2) & 3)) << 2);
      }
    }
  }
  const int bbase = (wn * NT2) * 512 + h * 128 + j * 4;
  const unsigned short* wsrc = wp + (long)(n_tile * NP) * S * 1024;      // [NP pieces][S][1024 bf16]

  auto stage_a = [&](int chunk, float* dst) {
#pragma unroll
    for (int r = 0; r < AROUNDS; r++) {
      if (HPX ? ((wave + 4 * r) * 16 < PLIVE) : (r * 64 < PLIVE)) {                   // wave-uniform
        const float* src = (aoff[r] >= 0) ? (const float*)(x + (long)aoff[r] + chunk * 32) : g_zero_page_bf16;
        glds16(src, HPX ? dst + (wave + 4 * r) * 256 : dst + wave * PLANE + r * 256);
      }
    }
  };
  // The step's weight tile.  BUFFER-addressed LDS-DMA (common.h: bufdma16): the lane's byte offset inside its piece is loop-invariant and
  // the step a scalar byte offset -- no per-step 64-bit address arithmetic on the vector ALU.  Same process, B = 256 (tools/ab_conv_bf16.py,
  // -DBF16_W_GLOBAL for the A/B): -1.4 ... -3.8 % on the 112x112 and 56x56 layers, +-0 at 28x28 / 14x14, +4.6 % on conv1_2's 8x32 patch,
  // which keeps the global_load_lds form.
#ifdef BF16_W_GLOBAL
  constexpr bool WBUF = false;
#else
  constexpr bool WBUF = TW != 32;
#endif
  const lct_i32x4 wrs = lct_make_rsrc(wsrc);
  unsigned wvo[(WCHUNKS + 255) / 256];
#pragma unroll
  for (int r = 0; r < (WCHUNKS + 255) / 256; r++) {
    const int q = min(r * 256 + tid, WCHUNKS - 1);
    wvo[r] = (unsigned)((((long)(q >> 7) * S) * 1024 + (q & 127) * 8) * 2);
  }
  auto stage_w = [&](int s, float* dst) {
    if constexpr (WBUF) {
      const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)s * 2048u);
#pragma unroll
      for (int r = 0; r < (WCHUNKS + 255) / 256; r++) {
        // (WCHUNKS is a multiple of 128: a wave is inside a round or outside it as a whole)
        if (WCHUNKS >= 256 * (r + 1) || r * 256 + wave * 64 < WCHUNKS) bufdma16(wvo[r], wrs, so, dst + (r * 256 + wave * 64) * 4);
      }
    } else {
#pragma unroll
      for (int r = 0; r < (WCHUNKS + 255) / 256; r++) {
        const int q = r * 256 + tid;
        if (WCHUNKS >= 256 * (r + 1) || q < WCHUNKS) {
          const int piece = q >> 7, within = q & 127;
          glds16((const float*)(wsrc + ((long)piece * S + s) * 1024 + within * 8), dst + (r * 256 + wave * 64) * 4);
        }
      }
    }
  };

  f32x16 acc[MT][NT2];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int nn = 0; nn < NT2; nn++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][nn][r] = 0.f;

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
        bf16x8 bf[NT2];
#pragma unroll
        for (int nn = 0; nn < NT2; nn++) bf[nn] = __builtin_bit_cast(bf16x8, *(const f32x4*)(wb + nn * 512 + kk * 256));
#pragma unroll
        for (int m = 0; m < MT; m++) {
          const bf16x8 af = __builtin_bit_cast(bf16x8, *(const f32x4*)(HPX ? ab + (afr[HPX ? m : 0][tap] ^ (kk ? 8 : 0))
                                                                           : ab + base[m] + tapoff + kk * 2 * PLANE));
#pragma unroll
          for (int nn = 0; nn < NT2; nn++) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[nn], acc[m][nn], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue: bias + ReLU (+ in-register 2x2 max pool), bf16 NHWC store ----
#if BF16_EPI_LDS
  // Each 32-pixel x (32*NT2)-channel block of a wave goes through a private fp32 LDS tile (the operand buffers are free: the loop
  // ended on a barrier) so that a lane stores 16 bytes of consecutive channels of ONE pixel: 64..128-byte row segments per store
  // instruction instead of 2-byte scatters (32 store instructions per block became 2-4; a Cin = 64 layer spent a third of its time here).
  {
    typedef unsigned short us8 __attribute__((ext_vector_type(8)));
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    constexpr int TC = 32 * NT2, TS = TC + 1;                      // tile columns, padded row stride (floats)
    float* T = smem + wave * 32 * TS;
    const int co0 = (n_tile * NP + wn * NT2) * 32;
    if (!POOL) {
      constexpr int LPP = TC / 8, PPI = 64 / LPP;                  // lanes per pixel (16 B each), pixels per store instruction
      const int tq = lane % LPP, tp = lane / LPP;                  // a store instruction writes PPI whole 64..128-byte pixel rows
      float bvv[8];
#pragma unroll
      for (int e = 0; e < 8; e++) bvv[e] = bias ? bias[co0 + 8 * tq + e] : 0.f;
#pragma unroll
      for (int m = 0; m < MT; m++) {
#pragma unroll
        for (int nn = 0; nn < NT2; nn++)
#pragma unroll
          for (int r = 0; r < 16; r++) T[((r & 3) + 8 * (r >> 2) + 4 * h) * TS + nn * 32 + j] = acc[m][nn][r];
#pragma unroll
        for (int c = 0; c < 32 / PPI; c++) {
          const int pr = tp + PPI * c;
          int ty, tx;
          subtile_pixel<SH, SW, TW>(wm * MT + m, pr, ty, tx);
          const int R = R0 + ty;
          us8 ov;
#pragma unroll
          for (int e = 0; e < 8; e++) {
            float v = T[pr * TS + 8 * tq + e] + bvv[e];
            if (relu) v = fmaxf(v, 0.f);
            ov[e] = f2bf(v);
          }
          if (R < Rtot) *(us8*)(y + ((long)R * W + x0 + tx) * Cout + co0 + 8 * tq) = ov;
        }
      }
    } else {
      const int Ho = H >> 1, Wo = W >> 1;
      constexpr int CW = TC / 8;                                   // channels per lane of the read-back: 8 pooled pixels x 8 lanes
      const int px = lane >> 3, tcol = (lane & 7) * CW;
      float bvv[CW];
#pragma unroll
      for (int e = 0; e < CW; e++) bvv[e] = bias ? bias[co0 + tcol + e] : 0.f;
      const int rr = 2 * (px & 1) + 16 * ((px >> 1) & 1) + 4 * (px >> 2);    // row of the (0,0) pixel of pooled pixel px = (h, ap, bp)
#pragma unroll
      for (int m = 0; m < MT; m++) {
#pragma unroll
        for (int nn = 0; nn < NT2; nn++)
#pragma unroll
          for (int ap = 0; ap < 2; ap++)
#pragma unroll
            for (int bp = 0; bp < 2; bp++) {
              const int r00 = 8 * ap + 2 * bp;                     // reg of (dy=0,dx=0): a = 2ap, b = 2bp
              T[(4 * h + 2 * ap + bp) * TS + nn * 32 + j] =
                  fmaxf(fmaxf(acc[m][nn][r00], acc[m][nn][r00 + 1]), fmaxf(acc[m][nn][r00 + 4], acc[m][nn][r00 + 5]));
            }
        int ty, tx;
        subtile_pixel<SH, SW, TW>(wm * MT + m, rr, ty, tx);
        const int R = R0 + ty;
        const int n = R / H, yy = R % H;
        unsigned short* dst = y + (((long)n * Ho + (yy >> 1)) * Wo + ((x0 + tx) >> 1)) * Cout + co0 + tcol;
        if (CW == 8) {
          us8 ov;
#pragma unroll
          for (int e = 0; e < 8; e++) {
            float v = T[px * TS + tcol + e] + bvv[e];
            if (relu) v = fmaxf(v, 0.f);
            ov[e] = f2bf(v);
          }
          if (R < Rtot) *(us8*)dst = ov;
        } else {
          us4 ov;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float v = T[px * TS + tcol + e] + bvv[e];
            if (relu) v = fmaxf(v, 0.f);
            ov[e] = f2bf(v);
          }
          if (R < Rtot) *(us4*)dst = ov;
        }
      }
    }
  }
#else
#pragma unroll
  for (int nn = 0; nn < NT2; nn++) {
    const int co = (n_tile * NP + wn * NT2 + nn) * 32 + j;
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
            float v = acc[m][nn][r] + bv;
            if (relu) v = fmaxf(v, 0.f);
            y[((long)R * W + x0 + tx) * Cout + co] = f2bf(v);
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
            const int r00 = 8 * ap + 2 * bp;
            float v = fmaxf(fmaxf(acc[m][nn][r00], acc[m][nn][r00 + 1]), fmaxf(acc[m][nn][r00 + 4], acc[m][nn][r00 + 5])) + bv;
            if (relu) v = fmaxf(v, 0.f);
            const int rr = (2 * bp) + 8 * (2 * ap) + 4 * h;
            int ty, tx;
            subtile_pixel<SH, SW, TW>(wm * MT + m, rr, ty, tx);
            const int R = R0 + ty;
            if (R < Rtot) {
              const int n = R / H, yy = R % H;
              const long opix = ((long)n * Ho + (yy >> 1)) * Wo + ((x0 + tx) >> 1);
              y[opix * Cout + co] = f2bf(v);
            }
          }
        }
      }
    }
  }
#endif
}

// first layer (Cin = 3): fp32 NCHW image -> bf16 NHWC, bias + ReLU.  HBM-write bound; lane = output channel, 8 pixels of
// a row per wave iteration held in registers, the three zero-padded input rows of each channel in LDS read as wave-uniform
// 16-byte broadcasts (the scheme of conv3x3_first_kernel in conv3x3.hip).
__global__ __launch_bounds__(256) void conv3x3_first_bf16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, unsigned short* __restrict__ y, int B, int H,
                                                                 int W, int Cout, int relu) {
  constexpr int PX = 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [9 (ch, row)][WP] with index i <-> x = i - 1, then [27][64] weights
  const int WP = (W + 2 + PX + 3) & ~3;
  const int R = blockIdx.x, n = R / H, yy = R % H;
  for (int i = threadIdx.x; i < 9 * WP; i += blockDim.x) {
    const int cr = i / WP, xx = i % WP - 1;
    const int c = cr / 3, iy = yy + (cr % 3) - 1;
    const bool ok = iy >= 0 && iy < H && xx >= 0 && xx < W;
    smem[i] = ok ? x[(((long)n * 3 + c) * H + iy) * W + xx] : 0.f;
  }
  float* wsm = smem + 9 * WP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int cg = 0; cg < Cout; cg += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * 64; i += blockDim.x) wsm[i] = w[min(cg + (i & 63), Cout - 1) * 27 + (i >> 6)];
    __syncthreads();
    const int co = cg + lane;
    const float bv = (bias && co < Cout) ? bias[co] : 0.f;
    for (int x0 = wave * PX; x0 < W; x0 += nw * PX) {
      float a[PX];
#pragma unroll
      for (int i = 0; i < PX; i++) a[i] = bv;
#pragma unroll 1
      for (int cr = 0; cr < 9; cr++) {
        const float* row = smem + cr * WP + x0;                  // wave-uniform, 16-byte aligned
        const f32x4 i0 = *(const f32x4*)row, i1 = *(const f32x4*)(row + 4), i2 = *(const f32x4*)(row + 8);
        const float in[12] = {i0[0], i0[1], i0[2], i0[3], i1[0], i1[1], i1[2], i1[3], i2[0], i2[1], i2[2], i2[3]};
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
          if (x0 + i < W) y[((long)R * W + x0 + i) * Cout + co] = f2bf(relu ? fmaxf(a[i], 0.f) : a[i]);
      }
    }
  }
}

// The first layer on the matrix cores: implicit GEMM with K = 9 rows (c,ky) x 8 (kx 0..2, five zeros) padded to 5 k16-steps -- the
// MFMA work is a rounding error (62 % of it multiplies zeros), what counts is that it leaves the VALU form's 27 FMAs per output behind
// (0.88 ms, compute-bound) for the 1.65 GB of bf16 output.  A lane's eight k values are eight consecutive pixels of a bf16 row image in
// LDS; two copies of every row, one shifted by a pixel, keep the window of an odd output column 4-byte aligned.  Workgroup = 4 waves =
// 4 output rows x 128 columns x 64 channels.  Image and weights are rounded to bf16 (the VALU form rounds only its output).
constexpr int F1_ROWS = 4, F1_NR = F1_ROWS + 2, F1_RL = 136;
__global__ __launch_bounds__(256, 2) void conv3x3_first_bf16_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                         const float* __restrict__ bias, unsigned short* __restrict__ y,
                                                                         int B, int H, int W, int relu) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned short us8 __attribute__((ext_vector_type(8)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* xin = (unsigned short*)smem_raw;                          // [2 copies][3][F1_NR][F1_RL]: copy k, index j <-> ix = ox0 + j - 1 + k
  unsigned short* wB = xin + 2 * 3 * F1_NR * F1_RL;                          // [10 rows][64 co][8]
  float* Tall = (float*)(wB + 10 * 64 * 8);                                 // [4 waves][32][65]
  const int gx = (W + 127) / 128, gy = (H + F1_ROWS - 1) / F1_ROWS;
  const int xh = blockIdx.x % gx, yb = (blockIdx.x / gx) % gy, n = blockIdx.x / (gx * gy);
  const int ox0 = 128 * xh, oy0 = F1_ROWS * yb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // every load of the staging is issued before the first LDS store (39 + 20 per thread: one dependent round trip instead of ~8)
  constexpr int NXI = (2 * 3 * F1_NR * F1_RL + 255) / 256, NWI = (10 * 64 * 8) / 256;
  float xv[NXI], wv[NWI];
#pragma unroll
  for (int k = 0; k < NXI; k++) {
    const int i = tid + 256 * k;
    const int j = i % F1_RL, rr = (i / F1_RL) % F1_NR, c = (i / (F1_RL * F1_NR)) % 3, cp = i / (F1_RL * F1_NR * 3);
    const int iy = oy0 - 1 + rr, ix = ox0 + j - 1 + cp;
    const bool ok = i < 2 * 3 * F1_NR * F1_RL && iy >= 0 && iy < H && ix >= 0 && ix < W;
    xv[k] = ok ? x[(((long)n * 3 + c) * H + iy) * W + ix] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < NWI; k++) {
    const int i = tid + 256 * k;
    const int kx = i & 7, co = (i >> 3) & 63, row = i >> 9;
    wv[k] = (row < 9 && kx < 3) ? w[(co * 9 + row) * 3 + kx] : 0.f;          // w[co][c][ky][kx], row = c*3 + ky
  }
#pragma unroll
  for (int k = 0; k < NXI; k++)
    if (tid + 256 * k < 2 * 3 * F1_NR * F1_RL) xin[tid + 256 * k] = f2bf(xv[k]);
#pragma unroll
  for (int k = 0; k < NWI; k++) wB[tid + 256 * k] = f2bf(wv[k]);
  __syncthreads();
  const int j = lane & 31, h = lane >> 5;
  const int lx = 32 * wave + j;
  const int cpy = lx & 1, jst = lx - cpy;                                    // odd columns read the shifted copy one slot earlier
  float* T = Tall + wave * 32 * 65;
  const int tq = lane & 7, tp = lane >> 3;                                   // read-back: 8 lanes x 16 B = the 128 bytes of one pixel
  float bvv[8];
#pragma unroll
  for (int e = 0; e < 8; e++) bvv[e] = bias ? bias[8 * tq + e] : 0.f;
  for (int oyl = 0; oyl < F1_ROWS; oyl++) {
    const int oy = oy0 + oyl;
    if (oy >= H) break;                                                      // uniform
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 5; s++) {
      const int row = 2 * s + h, rowc = row < 9 ? row : 8;                   // row 9 does not exist: its weights are zero
      const int c = rowc / 3, ky = rowc % 3;
      const unsigned int* ap = (const unsigned int*)(xin + ((cpy * 3 + c) * F1_NR + oyl + ky) * F1_RL + jst);
      const u32x4 av = {ap[0], ap[1], ap[2], ap[3]};
      const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const bf16x8 b = __builtin_bit_cast(bf16x8, *(const u32x4*)(wB + ((row * 64) + 32 * t + j) * 8));
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) T[((r & 3) + 8 * (r >> 2) + 4 * h) * 65 + t * 32 + j] = acc[t][r];
#pragma unroll
    for (int c8 = 0; c8 < 4; c8++) {                                        // a store instruction = 8 whole pixels (8 x 128 B)
      const int pr = tp + 8 * c8, oxs = ox0 + 32 * wave + pr;
      us8 ov;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float v = T[pr * 65 + 8 * tq + e] + bvv[e];
        if (relu) v = fmaxf(v, 0.f);
        ov[e] = f2bf(v);
      }
      if (oxs < W) *(us8*)(y + ((((long)n * H + oy) * W + oxs) * 64 + 8 * tq)) = ov;
    }
  }
}

// AdaptiveAvgPool2d((7,7)) + flatten in NCHW order: bf16 NHWC [B,H,W,C] -> fp32 [B, C*49]
__global__ void adaptive_avgpool7_flatten_bf16_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C) {
  const long total = (long)B * C * 49;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int c = o % C;
    const int cell = (o / C) % 49;
    const int n = o / ((long)C * 49);
    const int oy = cell / 7, ox = cell % 7;
    const int ys = (oy * H) / 7, ye = ((oy + 1) * H + 6) / 7;
    const int xs = (ox * W) / 7, xe = ((ox + 1) * W + 6) / 7;
    float a = 0.f;
    for (int iy = ys; iy < ye; iy++)
      for (int ix = xs; ix < xe; ix++) {
        const unsigned int bits = (unsigned int)x[(((long)n * H + iy) * W + ix) * C + c] << 16;
        a += __builtin_bit_cast(float, bits);
      }
    y[((long)n * C + c) * 49 + cell] = a / (float)((ye - ys) * (xe - xs));
  }
}

int max_boundaries(int H, int TH, long Rtot) {
  int best = 0;
  const long lim = Rtot < (long)H * TH * 2 ? Rtot : (long)H * TH * 2;
  for (long R0 = 0; R0 < lim; R0 += TH) {
    const int g = (int)((R0 + TH - 1) / H - R0 / H);
    if (g > best) best = g;
  }
  return best;
}

template <int TH, int TW, int RS, int SH, int SW, int WM, int WNW, int MT, int NT2>
int launch_conv(const unsigned short* x, const unsigned short* wp, const float* bias, unsigned short* y, int B, int H, int W, int Cin,
                int Cout, int relu, int pool, int G, hipStream_t st) {
  constexpr int AROUNDS = ((TH + 4) * RS + 63) / 64;
  constexpr int ASZ = AROUNDS * 64 * 16;
  constexpr int WSZ = WNW * NT2 * 512;
  const size_t lds = (size_t)(2 * ASZ + 2 * WSZ) * sizeof(float);
  const long Rtot = (long)B * H;
  const int mtiles = cdiv(Rtot, TH) * (W / TW);
  const int ntiles = Cout / (32 * WNW * NT2);
  const dim3 grid(mtiles * ntiles);
  if (pool) {
    auto k = conv3x3_bf16_kernel<TH, TW, RS, SH, SW, WM, WNW, MT, NT2, true>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, x, wp, bias, y, B, H, W, Cin, Cout, relu, G);
  } else {
    auto k = conv3x3_bf16_kernel<TH, TW, RS, SH, SW, WM, WNW, MT, NT2, false>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, x, wp, bias, y, B, H, W, Cin, Cout, relu, G);
  }
  return lct_check_launch("conv3x3_bf16");
}

// patch: 1 (8,32), 2 (8,28), 4 (16,14), 5 (32,8); 0 none
int select_patch(int W) {
  if (W % 32 == 0) return 1;
  if (W % 8 == 0) return 5;
  if (W % 28 == 0) return 2;
  if (W % 14 == 0) return 4;
  return 0;
}
int patch_th(int sel) { return sel == 1 || sel == 2 ? 8 : (sel == 5 ? 32 : 16); }

}  // namespace

extern "C" {

// 1 when the bf16 MFMA kernel tiles this layer
int lctvqa_conv3x3_bf16_supported(int B, int H, int W, int Cin, int Cout, int pool) {
  if (Cin % 32 || Cout % 64) return 0;
  if (pool && ((H & 1) || (W & 1))) return 0;
  const int sel = select_patch(W);
  if (!sel) return 0;
  if ((sel == 2 || sel == 4) && Cout % 128) return 0;         // the 224-pixel patches exist in the 1 x 4 wave layout only
  if (max_boundaries(H, patch_th(sel), (long)B * H) > 2) return 0;
  if ((long)B * H * W * Cin >= (1L << 31)) return 0;
  return 1;
}

// fp32 OIHW parameter -> bf16 kernel layout (Cout*Cin*9 bf16)
int lctvqa_conv3x3_bf16_pack_weights(const float* w_oihw, void* w_packed, int Cout, int Cin, void* stream) {
  LCT_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, "conv3x3_bf16_pack_weights: need Cin%%32==0, Cout%%32==0 (got %d,%d)", Cin, Cout);
  const long total = (long)Cout * Cin * 9;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv3x3_bf16_pack_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, (unsigned short*)w_packed, Cout, Cin);
  return lct_check_launch("conv3x3_bf16_pack");
}
// packed weights of the data-gradient convolution of a layer with fp32 OIHW weights [Cout][Cin][3][3]: a [Cin]-output,
// [Cout]-input 3x3 convolution with transposed channels and flipped taps (the backward of models_lct.py:62-63's trainable trunk)
int lctvqa_conv3x3_bf16_pack_weights_dgrad(const float* w_oihw, void* w_packed, int Cout, int Cin, void* stream) {
  LCT_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, "conv3x3_bf16_pack_weights_dgrad: need Cin%%32==0, Cout%%32==0 (got %d,%d)", Cin, Cout);
  const long total = (long)Cout * Cin * 9;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv3x3_bf16_pack_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, (unsigned short*)w_packed, Cin, Cout);
  return lct_check_launch("conv3x3_bf16_pack_dgrad");
}

// x bf16 NHWC [B,H,W,Cin] -> y bf16 NHWC [B,H,W,Cout] (or [B,H/2,W/2,Cout] when pool); bias fp32; fp32 accumulation
int lctvqa_conv3x3_bf16_nhwc(const void* x, const void* w_packed, const float* bias, void* y, int B, int H, int W, int Cin, int Cout,
                             int relu, int pool, void* stream) {
  LCT_REQUIRE(lctvqa_conv3x3_bf16_supported(B, H, W, Cin, Cout, pool), "conv3x3_bf16_nhwc: unsupported shape H=%d W=%d Cin=%d Cout=%d", H, W, Cin, Cout);
  LCT_REQUIRE((((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y) & 15) == 0, "conv3x3_bf16_nhwc: tensors must be 16-B aligned");
  hipStream_t st = (hipStream_t)stream;
  const unsigned short* xs = (const unsigned short*)x;
  const unsigned short* ws = (const unsigned short*)w_packed;
  unsigned short* ys = (unsigned short*)y;
  const long Rtot = (long)B * H;
  const bool wide = (Cout % 128 == 0);
  // 256-pixel patches: 2 x 2 waves, wave = 4 M-subtiles x 2 N-subtiles (128 channels) or x 1 (64 channels);
  // 224-pixel patches: 1 x 4 waves, wave = 7 M-subtiles x 1 N-subtile (128 channels)
#define LAUNCH256(TH, TW, RS, SH, SW)                                                                                       \
  do {                                                                                                                     \
    const int G = max_boundaries(H, TH, Rtot);                                                                             \
    if (wide) return launch_conv<TH, TW, RS, SH, SW, 2, 2, 4, 2>(xs, ws, bias, ys, B, H, W, Cin, Cout, relu, pool, G, st);  \
    return launch_conv<TH, TW, RS, SH, SW, 2, 2, 4, 1>(xs, ws, bias, ys, B, H, W, Cin, Cout, relu, pool, G, st);            \
  } while (0)
#define LAUNCH224(TH, TW, RS, SH, SW)                                                                                       \
  do {                                                                                                                     \
    const int G = max_boundaries(H, TH, Rtot);                                                                             \
    return launch_conv<TH, TW, RS, SH, SW, 1, 4, 7, 1>(xs, ws, bias, ys, B, H, W, Cin, Cout, relu, pool, G, st);            \
  } while (0)
  switch (select_patch(W)) {
    case 1: LAUNCH256(8, 32, 34, 2, 16); break;
    case 5: LAUNCH256(32, 8, 10, 4, 8); break;
    case 2: LAUNCH224(8, 28, 30, 8, 4); break;
    case 4: LAUNCH224(16, 14, 18, 16, 2); break;
    default: break;
  }
#undef LAUNCH256
#undef LAUNCH224
  lct_set_error("conv3x3_bf16_nhwc: no tiling for H=%d W=%d Cin=%d Cout=%d", H, W, Cin, Cout);
  return LCT_EUNSUPPORTED;
}

// first VGG layer: fp32 NCHW image [B,3,H,W] -> bf16 NHWC [B,H,W,Cout]
int lctvqa_conv3x3_first_nchw_bf16(const float* x_nchw, const float* w_oihw, const float* bias, void* y_nhwc, int B, int H, int W, int Cout,
                                   int relu, void* stream) {
  LCT_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0, "conv3x3_first_bf16: bad shape");
  const size_t lds = (size_t)(9 * ((W + 2 + 8 + 3) & ~3) + 27 * 64) * sizeof(float);
  LCT_REQUIRE(lds <= 64 * 1024, "conv3x3_first_bf16: W too large");
  hipLaunchKernelGGL(conv3x3_first_bf16_kernel, dim3(B * H), dim3(256), lds, (hipStream_t)stream, x_nchw, w_oihw, bias,
                     (unsigned short*)y_nhwc, B, H, W, Cout, relu);
  return lct_check_launch("conv3x3_first_bf16");
}

// the first layer on the matrix cores (Cout = 64 only): image and weights rounded to bf16, fp32 accumulation
int lctvqa_conv3x3_first_nchw_bf16_mfma(const float* x_nchw, const float* w_oihw /* [64][3][3][3] */, const float* bias, void* y_nhwc,
                                        int B, int H, int W, int relu, void* stream) {
  LCT_REQUIRE(B > 0 && H > 0 && W > 0, "conv3x3_first_bf16_mfma: bad shape");
  LCT_REQUIRE(((uintptr_t)y_nhwc & 15) == 0, "conv3x3_first_bf16_mfma: output must be 16-byte aligned");
  const size_t lds = (size_t)(2 * 3 * F1_NR * F1_RL + 10 * 64 * 8) * 2 + (size_t)4 * 32 * 65 * 4;
  auto k = conv3x3_first_bf16_mfma_kernel;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const long grid = (long)B * ((H + F1_ROWS - 1) / F1_ROWS) * ((W + 127) / 128);
  LCT_REQUIRE(grid < (1L << 31), "conv3x3_first_bf16_mfma: grid too large");
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, x_nchw, w_oihw, bias, (unsigned short*)y_nhwc, B, H, W, relu);
  return lct_check_launch("conv3x3_first_bf16_mfma");
}

int lctvqa_adaptive_avgpool7_flatten_bf16(const void* x_nhwc, float* y, int B, int H, int W, int C, void* stream) {
  const long t = (long)B * C * 49;
  int b = (int)((t + 255) / 256);
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(adaptive_avgpool7_flatten_bf16_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x_nhwc, y, B, H, W, C);
  return lct_check_launch("adaptive_avgpool7_flatten_bf16");
}

}  // extern "C"
