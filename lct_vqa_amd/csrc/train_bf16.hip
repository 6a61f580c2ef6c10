// bf16-native TRAINABLE convolution trunk (BASELINE.json configs[2]: "ResNet-152+GRU bf16 ... encoder trainable"; the reference's
// trainable-trunk semantics are basic_vqa/models_lct.py:56-63 -- a trunk without no_grad whose every layer gets a gradient).
// Activations and their gradients live in HBM as bf16 NHWC rows [P = B*H*W][C]; weights are fp32 masters converted per step;
// every product accumulates in fp32 on v_mfma_f32_32x32x16_bf16; BatchNorm statistics and parameter gradients are fp32 / fp64.
//
//   conv_gemm_bf16   z[P][N] = sum_tap sum_ci x[p + tap][ci] * w[n][tap][ci]          (1x1: taps = 1; 3x3 on any map: taps = 9)
//                    NT product, operand tiles by LDS-DMA (rows outside the image read a zero page), epilogue: per-workgroup
//                    column (sum, sum of squares) of the fp32 accumulators = the BatchNorm statistics of the layer's output,
//                    bf16 addend (residual gradient) and bf16 ReLU mask (out = mask > 0 ? out : 0) for the data-gradient use.
//   wgrad_bf16_tn    dW[tap][co][ci] = sum_p dz[p][co] * x[p + tap][ci]                 (both operands K(=pixel)-strided)
//                    tiles [32 pixels][64 channels] by LDS-DMA in 16-row x 64-byte sub-tiles, MFMA operands by the hardware
//                    transpose read ds_read_b64_tr_b16, split over the pixels into fp32 slabs, fixed-order reduction.
//   bn_*             BatchNorm2d (batch statistics) on bf16 rows: statistics (when the producer did not take them), coefficient
//                    finish in fp64, apply fused with residual add + ReLU; backward sums (ReLU mask applied on the fly, masked
//                    gradient written once for the residual branch), coefficient finish, apply.  HBM-bound: 2 B per element and pass.
#include "common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __attribute__((aligned(64))) float g_zero_page_tb[64];

__device__ __forceinline__ unsigned short f2bf(float v) {
  const __bf16 b = (__bf16)v;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned int)u << 16); }

// ------------------------------------------------------------------------------------------------------------------
// NT product with taps: the structure of gemm_bf16_kernel (gemm_bf16.hip: 256 x 128|64 tile, 2 x 2 waves of 4 x NT2 32x32
// accumulators, K steps of 32 in 64-byte swizzled LDS rows, double buffered LDS-DMA, one barrier per step).
// ------------------------------------------------------------------------------------------------------------------
struct CgArgs {
  const unsigned short* A; long lda;          // [M][lda] bf16 rows (NHWC pixels), Cin channels used
  const unsigned short* B; long ldb;          // [N][ldb] bf16, k = tap * Cin + ci
  unsigned short* C; long ldc;                // [M][ldc] bf16
  int M, N, Cin, taps, H, W;
  const unsigned short* addend; long ld_add;  // bf16 [M][ld_add] or null (may alias C)
  const unsigned short* mask; long ld_mask;   // bf16 [M][ld_mask] or null: out = mask > 0 ? out : 0
  float* stats;                               // [2 * ceil(M/256)][2][N] or null
};

template <int NT2, int TAPS, int MT>
__global__ __launch_bounds__(256, (TAPS == 1 && MT == 4) ? 3 : 2) void conv_gemm_bf16_kernel(const CgArgs a) {
  constexpr int BM = 64 * MT, BN = 64 * NT2;
  constexpr int APL = BM * 4, BPL = BN * 4;
  constexpr int ASZ = 4 * APL, BSZ = 4 * BPL;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* abuf = smem;
  float* bbuf = smem + 2 * ASZ;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int j = lane & 31, h = lane >> 5;
  // the n tiles of one m tile read the same A rows: they are dealt to ONE XCD (blocks b and b + 8 share an XCD; speed only), so
  // that A leaves HBM once instead of once per XCD (PMC, round 4: 2-3x the algorithmic bytes fetched with n tiles on consecutive blocks)
  const int NTL = a.N / BN;
  const int MTL = (a.M + BM - 1) / BM;
  const int grp8 = blockIdx.x / (8 * NTL), rem8 = blockIdx.x % (8 * NTL);
  const int m_tile = grp8 * 8 + (rem8 & 7), n_tile = rem8 >> 3;
  if (m_tile >= MTL) return;                                      // uniform: the grid is rounded up to whole groups of 8 m tiles
  const long m0 = (long)m_tile * BM;
  const int n0 = n_tile * BN;
  const int spt = a.Cin >> 5;                                    // K steps per tap
  const int S = TAPS * spt;
  const int rl = lane >> 2, sq = (lane & 3) ^ ((rl >> 2) & 3);
  int arow[MT], ay[TAPS == 9 ? MT : 1], ax[TAPS == 9 ? MT : 1];
#pragma unroll
  for (int r = 0; r < MT; r++) {
    const long row = m0 + 16 * (wave + 4 * r) + rl;
    arow[r] = row < a.M ? (int)row : -1;
    if (TAPS == 9) { ax[r] = (int)(row % a.W); ay[r] = (int)((row / a.W) % a.H); }
  }
  long boff[NT2];
#pragma unroll
  for (int r = 0; r < NT2; r++) boff[r] = (long)(n0 + 16 * (wave + 4 * r) + rl) * a.ldb + sq * 8;
  auto stage = [&](int tap, int ks, float* ad, float* bd) {
    int dy = 0, dx = 0;
    if (TAPS == 9) { dy = tap / 3 - 1; dx = tap - 3 * (tap / 3) - 1; }
    const int dro = dy * a.W + dx;
#pragma unroll
    for (int r = 0; r < MT; r++) {
      bool ok = arow[r] >= 0;
      if (TAPS == 9) ok = ok && (unsigned)(ay[r] + dy) < (unsigned)a.H && (unsigned)(ax[r] + dx) < (unsigned)a.W;
      const float* src = ok ? (const float*)(a.A + (long)(arow[r] + dro) * a.lda + sq * 8 + ks * 32) : g_zero_page_tb;
      glds16(src, ad + (wave + 4 * r) * 256);
    }
#pragma unroll
    for (int r = 0; r < NT2; r++) glds16((const float*)(a.B + boff[r] + (long)(tap * spt + ks) * 32), bd + (wave + 4 * r) * 256);
  };
  f32x16 acc[MT][NT2];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int nn = 0; nn < NT2; nn++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][nn][r] = 0.f;
  stage(0, 0, abuf, bbuf);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int swz = (j >> 2) & 3;
  const int fo[2] = {((0 + h) ^ swz) << 2, ((2 + h) ^ swz) << 2};
  const int abase = (wm * MT * 32 + j) * 16;
  const int bbase = (wn * NT2 * 32 + j) * 16;
  int ntap = 0, nks = 0;
  for (int s = 0; s < S; s++) {
    if (++nks == spt) { nks = 0; ntap++; }
    if (s + 1 < S) stage(ntap, nks, abuf + ((s + 1) & 1) * ASZ, bbuf + ((s + 1) & 1) * BSZ);
    const float* ab = abuf + (s & 1) * ASZ + abase;
    const float* bb = bbuf + (s & 1) * BSZ + bbase;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      bf16x8 bf[NT2];
#pragma unroll
      for (int nn = 0; nn < NT2; nn++) bf[nn] = __builtin_bit_cast(bf16x8, *(const f32x4*)(bb + nn * 512 + fo[kk]));
#pragma unroll
      for (int m = 0; m < MT; m++) {
        const bf16x8 af = __builtin_bit_cast(bf16x8, *(const f32x4*)(ab + m * 512 + fo[kk]));
#pragma unroll
        for (int nn = 0; nn < NT2; nn++) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[nn], acc[m][nn], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // BatchNorm statistics of the layer output: column sums of the fp32 accumulators of this wave's 128 rows (rows beyond M are
  // products of the zero page: they add nothing).  Slab (2 m_tile + wm) of [slabs][2][N].
  if (a.stats) {
#pragma unroll
    for (int nn = 0; nn < NT2; nn++) {
      float cs = 0.f, cq = 0.f;
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) { const float v = acc[m][nn][r]; cs += v; cq = fmaf(v, v, cq); }
      cs += __shfl_xor(cs, 32, 64);
      cq += __shfl_xor(cq, 32, 64);
      if (h == 0) {
        float* sp = a.stats + ((long)(m_tile * 2 + wm) * 2) * a.N + n0 + (wn * NT2 + nn) * 32 + j;     // slab = 32 * MT rows
        sp[0] = cs;
        sp[a.N] = cq;
      }
    }
  }
  // bf16 output through the wave's private LDS tile: 16-byte stores of consecutive channels (gemm_bf16_kernel's epilogue)
  constexpr int TC = 32 * NT2, TS = TC + 1;
  constexpr int LPP = TC / 8, PPI = 64 / LPP;
  float* T = smem + wave * 32 * TS;
  const int tq = lane % LPP, tp = lane / LPP;
  const int col0 = n0 + wn * TC + 8 * tq;
#pragma unroll
  for (int m = 0; m < MT; m++) {
    __syncthreads();
#pragma unroll
    for (int nn = 0; nn < NT2; nn++)
#pragma unroll
      for (int r = 0; r < 16; r++) T[((r & 3) + 8 * (r >> 2) + 4 * h) * TS + nn * 32 + j] = acc[m][nn][r];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 32 / PPI; c++) {
      const int pr = tp + PPI * c;
      const long row = m0 + wm * (32 * MT) + m * 32 + pr;
      if (row < a.M) {
        us8 av = {0, 0, 0, 0, 0, 0, 0, 0}, mv = av;
        if (a.addend) av = *(const us8*)(a.addend + row * a.ld_add + col0);
        if (a.mask) mv = *(const us8*)(a.mask + row * a.ld_mask + col0);
        us8 ov;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          float v = T[pr * TS + 8 * tq + e];
          if (a.mask) v = bf2f(mv[e]) > 0.f ? v : 0.f;
          if (a.addend) v += bf2f(av[e]);
          ov[e] = f2bf(v);
        }
        *(us8*)(a.C + row * a.ldc + col0) = ov;
      }
    }
  }
}

// ---- wide form of the 1x1 product for N % 256 == 0, K >= 256: 256 x 256 tile on EIGHT waves (2 x 4, wave tile 128 x 64 as above),
// one workgroup per CU, D stages of LDS-DMA in flight with counted vmcnt (the form of wgrad_bf16_ring_kernel).  With 128 x 128 tiles
// every operand byte of these products is staged 3-4 times through L2 (staged bytes = flops * 2 B / tile edge) and the K loop runs at one
// memory latency per 32-deep step; here it is staged half as often and three stages are always in flight.  The n tiles of one m tile
// are dealt to ONE XCD (blocks b and b + 8 share an XCD: speed only) so that they read the same A rows from that XCD's L2.
template <int D>
__global__ __launch_bounds__(512, 1) void conv_gemm_bf16_wide_kernel(const CgArgs a) {
  constexpr int NI = 4;                                           // DMA instructions per wave and stage: 2 of A, 2 of B
  constexpr int ASZ = 256 * 16, STG = 2 * ASZ;                    // floats: A tile [256 rows][64 B], then the B tile
  static_assert(NI * (D - 1) <= 63, "vmcnt budget");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int j = lane & 31, h = lane >> 5;
  const int NTL = a.N >> 8;
  const int MTL = (a.M + 255) >> 8;
  const int grp = blockIdx.x / (8 * NTL), rem = blockIdx.x % (8 * NTL);
  const int m_tile = grp * 8 + (rem & 7), n_tile = rem >> 3;
  if (m_tile >= MTL) return;                                      // uniform: the grid is rounded up to whole groups of 8 m tiles
  const long m0 = (long)m_tile * 256;
  const int n0 = n_tile * 256;
  const int S = a.Cin >> 5;
  const int rl = lane >> 2, sq = (lane & 3) ^ ((rl >> 2) & 3);
  long aoff[2], boff[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const long row = m0 + 16 * (wave + 8 * r) + rl;
    aoff[r] = row < a.M ? row * a.lda + sq * 8 : -1;
    boff[r] = (long)(n0 + 16 * (wave + 8 * r) + rl) * a.ldb + sq * 8;
  }
  auto stage = [&](int s, float* buf) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const float* src = (aoff[r] >= 0 && s < S) ? (const float*)(a.A + aoff[r] + (long)s * 32) : g_zero_page_tb;
      glds16_opaque(src, buf + (wave + 8 * r) * 256);
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const float* src = s < S ? (const float*)(a.B + boff[r] + (long)s * 32) : g_zero_page_tb;
      glds16_opaque(src, buf + ASZ + (wave + 8 * r) * 256);
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int nn = 0; nn < 2; nn++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][nn][r] = 0.f;
  const int swz = (j >> 2) & 3;
  const int fo[2] = {((0 + h) ^ swz) << 2, ((2 + h) ^ swz) << 2};
  const int abase = (wm * 128 + j) * 16;
  const int bbase = ASZ + (wn * 64 + j) * 16;
#pragma unroll
  for (int d = 0; d < D - 1; d++) stage(d, smem + d * STG);
  for (int s = 0; s < S; s++) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI * (D - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    stage(s + D - 1, smem + ((s + D - 1) % D) * STG);
    const float* ab = smem + (s % D) * STG + abase;
    const float* bb = smem + (s % D) * STG + bbase;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      bf16x8 bf[2];
#pragma unroll
      for (int nn = 0; nn < 2; nn++) bf[nn] = __builtin_bit_cast(bf16x8, *(const f32x4*)(bb + nn * 512 + fo[kk]));
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const bf16x8 af = __builtin_bit_cast(bf16x8, *(const f32x4*)(ab + m * 512 + fo[kk]));
#pragma unroll
        for (int nn = 0; nn < 2; nn++) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[nn], acc[m][nn], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the tail's zero-page DMAs
  if (a.stats) {
#pragma unroll
    for (int nn = 0; nn < 2; nn++) {
      float cs = 0.f, cq = 0.f;
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) { const float v = acc[m][nn][r]; cs += v; cq = fmaf(v, v, cq); }
      cs += __shfl_xor(cs, 32, 64);
      cq += __shfl_xor(cq, 32, 64);
      if (h == 0) {
        float* sp = a.stats + ((long)(m_tile * 2 + wm) * 2) * a.N + n0 + wn * 64 + nn * 32 + j;     // slab = 128 rows
        sp[0] = cs;
        sp[a.N] = cq;
      }
    }
  }
  constexpr int TS = 65;
  float* T = smem + wave * 32 * TS;
  const int tq = lane & 7, tp = lane >> 3;                        // 8 lanes x 16 B per row of 64 columns, 8 rows per store instruction
  const int col0 = n0 + wn * 64 + 8 * tq;
#pragma unroll
  for (int m = 0; m < 4; m++) {
    __syncthreads();
#pragma unroll
    for (int nn = 0; nn < 2; nn++)
#pragma unroll
      for (int r = 0; r < 16; r++) T[((r & 3) + 8 * (r >> 2) + 4 * h) * TS + nn * 32 + j] = acc[m][nn][r];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int pr = tp + 8 * c;
      const long row = m0 + wm * 128 + m * 32 + pr;
      if (row < a.M) {
        us8 av = {0, 0, 0, 0, 0, 0, 0, 0}, mv = av;
        if (a.addend) av = *(const us8*)(a.addend + row * a.ld_add + col0);
        if (a.mask) mv = *(const us8*)(a.mask + row * a.ld_mask + col0);
        us8 ov;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          float v = T[pr * TS + 8 * tq + e];
          if (a.mask) v = bf2f(mv[e]) > 0.f ? v : 0.f;
          if (a.addend) v += bf2f(av[e]);
          ov[e] = f2bf(v);
        }
        *(us8*)(a.C + row * a.ldc + col0) = ov;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// TN product (weight gradient).  Workgroup = 4 waves as WM x WN, wave tile 64 (co) x 64 (one column block: a tap and 64 input
// channels); K step = 32 pixels.  LDS image of a step: 1-KB sub-tiles [16 pixels][32 channels] (one LDS-DMA instruction each: lane
// = (pixel lane>>2, 16-byte slot lane&3)), which the transposed read consumes conflict-free (a 32-lane half reads 4 pixel rows x
// 64 B = 256 contiguous bytes).  A (dz) sub-tile index ((mb * 2 + rh) * 2 + c32), B (x) likewise behind them.
// ------------------------------------------------------------------------------------------------------------------
struct WgArgs {
  const unsigned short* dZ; long ldz;         // [P][ldz] bf16
  const unsigned short* X; long ldx;          // [P][ldx] bf16
  float* slabs;                               // [splits][taps][Cout][Cin] fp32
  int P, Cout, Cin, taps, H, W;
  int steps_per_split;                        // 32-pixel steps per split
  int ncb;                                    // column blocks = taps * Cin / 64
  int ntile_m, ntile_n;
  int xcd_map;                                // two-stage form: deal the tiles of one pixel range to one XCD (1x1 products)
};

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* sub, int lane_off) {
  // rows 8h + 0..3 and 8h + 4..7 of the sub-tile, transposed: element e = pixel 8h + e of this lane's channel
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sub + lane_off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sub + lane_off + 256));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_tn_kernel(const WgArgs a) {
  static_assert(WM * WN == 4, "four waves");
  constexpr int NA = 4 * WM, NB = 4 * WN;                         // sub-tiles (= DMA instructions) per stage
  constexpr int STAGE = (NA + NB) * 1024;                         // bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ntiles = a.ntile_m * a.ntile_n;
  // the tiles of one pixel range read the same dz / x rows: XCD = blockIdx % 8 owns the splits = XCD (mod 8) (as in the ring form)
  // (for the 1x1 products: PMC showed 2.6x the algorithmic bytes fetched otherwise; the nine taps of a 3x3 product stage every x row
  // nine times, which one XCD's L2 cannot feed: there the tiles of a pixel range stay spread over all eight)
  const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
  const int tile = a.xcd_map ? qq % ntiles : blockIdx.x % ntiles;
  const int split = a.xcd_map ? (qq / ntiles) * 8 + xcd : blockIdx.x / ntiles;
  const int tile_m = tile % a.ntile_m, tile_n = tile / a.ntile_m;
  const int cpb = a.Cin >> 6;                                     // column blocks per tap
  const int steps_total = (a.P + 31) >> 5;
  const int s_begin = split * a.steps_per_split;
  const int s_end = min(steps_total, s_begin + a.steps_per_split);
  const int S = s_end - s_begin;
  if (S <= 0) return;                                             // uniform: the grid rounds the splits up to a multiple of 8
  // this lane's two pixel rows of a step (rh = 0, 1) and their image coordinates
  const int prl = lane >> 2, slot = lane & 3;
  long p_cur[2];
  int py[2], px[2];
#pragma unroll
  for (int rh = 0; rh < 2; rh++) {
    p_cur[rh] = (long)s_begin * 32 + 16 * rh + prl;
    px[rh] = py[rh] = 0;
    if (a.taps == 9) { px[rh] = (int)(p_cur[rh] % a.W); py[rh] = (int)((p_cur[rh] / a.W) % a.H); }
  }
  const int xinc = 32 % a.W, yinc = 32 / a.W;
  const bool tiny = yinc + 1 > a.H;                               // maps of a few rows: one conditional subtract does not wrap y
  auto stage = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < (NA + NB + 3) / 4; i++) {
      const int idx = wave + 4 * i;                               // wave-uniform
      if (idx >= NA + NB) break;
      if (idx < NA) {
        const int mb = idx >> 2, rh = (idx >> 1) & 1, c32 = idx & 1;
        const float* src = p_cur[rh] < a.P
            ? (const float*)(a.dZ + p_cur[rh] * a.ldz + ((tile_m * WM + mb) << 6) + 32 * c32 + 8 * slot) : g_zero_page_tb;
        glds16(src, (float*)(buf + idx * 1024));
      } else {
        const int ib = idx - NA;
        const int nb = ib >> 2, rh = (ib >> 1) & 1, c32 = ib & 1;
        const int cb = tile_n * WN + nb;
        if (cb < a.ncb) {
          const int tap = cb / cpb, ci0 = (cb - tap * cpb) << 6;
          int dy = 0, dx = 0;
          if (a.taps == 9) { dy = tap / 3 - 1; dx = tap - 3 * (tap / 3) - 1; }
          bool ok = p_cur[rh] < a.P;
          if (a.taps == 9) ok = ok && (unsigned)(py[rh] + dy) < (unsigned)a.H && (unsigned)(px[rh] + dx) < (unsigned)a.W;
          const float* src = ok ? (const float*)(a.X + (p_cur[rh] + dy * a.W + dx) * a.ldx + ci0 + 32 * c32 + 8 * slot) : g_zero_page_tb;
          glds16(src, (float*)(buf + idx * 1024));
        }
      }
    }
  };
  auto advance = [&]() {
#pragma unroll
    for (int rh = 0; rh < 2; rh++) {
      p_cur[rh] += 32;
      if (a.taps == 9) {
        if (tiny) {
          px[rh] = (int)(p_cur[rh] % a.W);
          py[rh] = (int)((p_cur[rh] / a.W) % a.H);
        } else {
          px[rh] += xinc; py[rh] += yinc;
          if (px[rh] >= a.W) { px[rh] -= a.W; py[rh]++; }
          if (py[rh] >= a.H) py[rh] -= a.H;
        }
      }
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; mi++)
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;
  const int cbw = tile_n * WN + wn;
  const bool active = cbw < a.ncb;
  // transposed-read lane offset inside a sub-tile: group g = lane>>4 -> (k half g>>1, 16-channel half g&1); lane 4q+pp of the
  // group addresses pixel row q, channels 4pp..4pp+3
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int lane_off = (8 * (g >> 1) + q) * 64 + (16 * (g & 1) + 4 * pp) * 2;
  if (S > 0) stage(smem_raw);
  advance();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = 0; s < S; s++) {
    if (s + 1 < S) stage(smem_raw + ((s + 1) & 1) * STAGE);
    advance();
    if (active) {
      const unsigned char* ab = smem_raw + (s & 1) * STAGE;
      const unsigned char* bb = ab + NA * 1024;
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int mi = 0; mi < 2; mi++) af[mi] = tr_frag(ab + (((wm * 2 + kk) * 2 + mi) << 10), lane_off);
#pragma unroll
        for (int ni = 0; ni < 2; ni++) bf[ni] = tr_frag(bb + (((wn * 2 + kk) * 2 + ni) << 10), lane_off);
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
          for (int ni = 0; ni < 2; ni++) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (active) {
    const int tap = cbw / cpb, ci0 = (cbw - tap * cpb) << 6;
    const int co0 = (tile_m * WM + wm) << 6;
    const int j = lane & 31, h = lane >> 5;
    float* out = a.slabs + ((long)split * a.taps + tap) * a.Cout * a.Cin;
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
      for (int ni = 0; ni < 2; ni++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int co = co0 + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * h;
          out[(long)co * a.Cin + ci0 + 32 * ni + j] = acc[mi][ni][r];
        }
  }
}

// ---- ring form (the default): ONE workgroup per CU keeps D stages of LDS-DMA in flight (counted vmcnt, bare barrier: the DMA of
// stage s + D - 1 is issued while stage s is multiplied), so a CU always has ~100 KB of loads outstanding -- the two-stage form above
// waits a full memory latency per 32-pixel step.  Work items (tile, pixel split) are dealt so that the tiles of one pixel range run on
// ONE XCD (blocks b and b + 8 share an XCD; speed only): they re-read the same dz / x rows from that XCD's L2, HBM sees them once.
// Every wave issues exactly NI DMA instructions per stage (a zero-page source where there is nothing to load) so that
// s_waitcnt vmcnt(NI * (D - 2)) means "my share of stage s has landed" at every iteration, the tail included.
template <int WM, int WN, int D>
__global__ __launch_bounds__(256, 1) void wgrad_bf16_ring_kernel(const WgArgs a) {
  static_assert(WM * WN == 4, "four waves");
  constexpr int NA = 4 * WM, NB = 4 * WN;
  constexpr int NI = (NA + NB) / 4;
  constexpr int STAGE = (NA + NB) * 1024;
  static_assert((NA + NB) % 4 == 0 && NI * (D - 1) <= 63, "vmcnt budget");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ntiles = a.ntile_m * a.ntile_n;
  // item -> (tile, split): XCD = blockIdx % 8 owns the splits = XCD (mod 8)
  const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
  const int tile = qq % ntiles, split = (qq / ntiles) * 8 + xcd;
  const int tile_m = tile % a.ntile_m, tile_n = tile / a.ntile_m;
  const int cpb = a.Cin >> 6;
  const int steps_total = (a.P + 31) >> 5;
  const int s_begin = split * a.steps_per_split;
  const int s_end = min(steps_total, s_begin + a.steps_per_split);
  const int S = s_end - s_begin;
  if (S <= 0) return;                                             // uniform: a split beyond the pixel range (splits rounded up to 8)
  const int prl = lane >> 2, slot = lane & 3;
  long p_cur[2];
  int py[2], px[2];
#pragma unroll
  for (int rh = 0; rh < 2; rh++) {
    p_cur[rh] = (long)s_begin * 32 + 16 * rh + prl;
    px[rh] = py[rh] = 0;
    if (a.taps == 9) { px[rh] = (int)(p_cur[rh] % a.W); py[rh] = (int)((p_cur[rh] / a.W) % a.H); }
  }
  const long p_end = min((long)a.P, (long)s_end * 32);
  const int xinc = 32 % a.W, yinc = 32 / a.W;
  const bool tiny = yinc + 1 > a.H;
  auto stage = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int idx = wave + 4 * i;
      const float* src = g_zero_page_tb;
      if (idx < NA) {
        const int mb = idx >> 2, rh = (idx >> 1) & 1, c32 = idx & 1;
        if (p_cur[rh] < p_end) src = (const float*)(a.dZ + p_cur[rh] * a.ldz + ((tile_m * WM + mb) << 6) + 32 * c32 + 8 * slot);
      } else {
        const int ib = idx - NA;
        const int nb = ib >> 2, rh = (ib >> 1) & 1, c32 = ib & 1;
        const int cb = tile_n * WN + nb;
        if (cb < a.ncb) {
          const int tap = cb / cpb, ci0 = (cb - tap * cpb) << 6;
          int dy = 0, dx = 0;
          if (a.taps == 9) { dy = tap / 3 - 1; dx = tap - 3 * (tap / 3) - 1; }
          bool ok = p_cur[rh] < p_end;
          if (a.taps == 9) ok = ok && (unsigned)(py[rh] + dy) < (unsigned)a.H && (unsigned)(px[rh] + dx) < (unsigned)a.W;
          if (ok) src = (const float*)(a.X + (p_cur[rh] + dy * a.W + dx) * a.ldx + ci0 + 32 * c32 + 8 * slot);
        }
      }
      glds16_opaque(src, (float*)(buf + idx * 1024));
    }
#pragma unroll
    for (int rh = 0; rh < 2; rh++) {
      p_cur[rh] += 32;
      if (a.taps == 9) {
        if (tiny) {
          px[rh] = (int)(p_cur[rh] % a.W);
          py[rh] = (int)((p_cur[rh] / a.W) % a.H);
        } else {
          px[rh] += xinc; py[rh] += yinc;
          if (px[rh] >= a.W) { px[rh] -= a.W; py[rh]++; }
          if (py[rh] >= a.H) py[rh] -= a.H;
        }
      }
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; mi++)
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;
  const int cbw = tile_n * WN + wn;
  const bool active = cbw < a.ncb;
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int lane_off = (8 * (g >> 1) + q) * 64 + (16 * (g & 1) + 4 * pp) * 2;
  // prologue: stages 0 .. D-2 (rows beyond this split's range read the zero page: they add nothing)
#pragma unroll
  for (int d = 0; d < D - 1; d++) stage(smem_raw + d * STAGE);
  for (int s = 0; s < S; s++) {
    // my DMA of stage s has landed when at most the NI * (D - 2) younger ones are outstanding; the barrier extends that to every
    // wave's share and tells that every wave is done reading stage s - 1, whose slot the next issue overwrites
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI * (D - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    stage(smem_raw + ((s + D - 1) % D) * STAGE);
    if (active) {
      const unsigned char* ab = smem_raw + (s % D) * STAGE;
      const unsigned char* bb = ab + NA * 1024;
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int mi = 0; mi < 2; mi++) af[mi] = tr_frag(ab + (((wm * 2 + kk) * 2 + mi) << 10), lane_off);
#pragma unroll
        for (int ni = 0; ni < 2; ni++) bf[ni] = tr_frag(bb + (((wn * 2 + kk) * 2 + ni) << 10), lane_off);
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
          for (int ni = 0; ni < 2; ni++) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the tail's zero-page DMAs must not outlive the workgroup's LDS
  if (active) {
    const int tap = cbw / cpb, ci0 = (cbw - tap * cpb) << 6;
    const int co0 = (tile_m * WM + wm) << 6;
    const int j = lane & 31, h = lane >> 5;
    float* out = a.slabs + ((long)split * a.taps + tap) * a.Cout * a.Cin;
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
      for (int ni = 0; ni < 2; ni++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int co = co0 + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * h;
          out[(long)co * a.Cin + ci0 + 32 * ni + j] = acc[mi][ni][r];
        }
  }
}

// ---- 3x3 weight gradient on 2-D pixel patches (round 4).  The tap-as-column-block form above stages every x row nine times (once per
// tap) and a dz row once per (tap, ci block): 16 KB of LDS-DMA per 32 MFMAs, and LDS-DMA delivers only about the HBM rate per CU.  Here
// a K step is a PATCH of PH x PWV output pixels (k = py * PWS + px; slots px >= PWV are zero rows of dz) and the x operand is the
// patch's (PH+2) x (PWV+2) halo, staged ONCE (image borders read the zero page at staging time): the nine taps are nine constant row
// offsets dy * (PWV+2) + dx into the same LDS image, taken by the per-lane row addresses of ds_read_b64_tr_b16.  Workgroup = 64 co x 64
// ci x 9 taps (2 x 2 waves, nine 32x32 accumulators each); 12 KB staged per 72 MFMAs.
struct Wg3Args {
  const unsigned short* dZ; long ldz;
  const unsigned short* X; long ldx;
  float* slabs;                               // [splits][9][Cout][Cin]
  int B, H, W, Cout, Cin;
  int patches_per_split, npatch, pby, pbx;    // patches per image row band / per image: pby = H / PH, pbx = W / PWV
};

template <int PH, int PWS, int PWV>
__global__ __launch_bounds__(256, 2) void wgrad3x3_bf16_kernel(const Wg3Args a) {
  static_assert(PH * PWS == 32 && PWV <= PWS && (PH + 2) * (PWV + 2) <= 64 && PWS % 4 == 0, "patch shape");
  constexpr int HW_ = PWV + 2;                                    // halo row length
  constexpr int STAGE = 12 * 1024, XB = 4 * 1024;                 // bytes: dz sub-tiles (rh, c32), then x [2 ch halves][64 halo pixels][64 B]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntm = a.Cout >> 6, ntiles = ntm * (a.Cin >> 6);
  const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;           // the tiles of one patch range on one XCD (speed only)
  const int tile = qq % ntiles, split = (qq / ntiles) * 8 + xcd;
  const int co0 = (tile % ntm) << 6, ci0 = (tile / ntm) << 6;
  const int t_begin = split * a.patches_per_split;
  const int t_end = min(a.npatch, t_begin + a.patches_per_split);
  const int S = t_end - t_begin;
  if (S <= 0) return;
  const int prl = lane >> 2, slot = lane & 3;
  const int ppi = a.pby * a.pbx;
  auto stage = [&](int t, unsigned char* buf) {
    const int n = t / ppi, rem = t - n * ppi;
    const int y0 = (rem / a.pbx) * PH, x0 = (rem % a.pbx) * PWV;
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const int i = wave + 4 * jj;                                // wave-uniform
      const float* src = g_zero_page_tb;
      if (i < 4) {
        const int rh = i >> 1, c32 = i & 1;
        const int k = 16 * rh + prl, py = k / PWS, px = k % PWS;
        if (px < PWV) src = (const float*)(a.dZ + ((long)(n * a.H + y0 + py) * a.W + x0 + px) * a.ldz + co0 + 32 * c32 + 8 * slot);
      } else {
        const int ix = i - 4, c32 = ix >> 2, hr = 16 * (ix & 3) + prl;
        const int hy = hr / HW_, hx = hr % HW_;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        if (hr < (PH + 2) * HW_ && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W)
          src = (const float*)(a.X + ((long)(n * a.H + yy) * a.W + xx) * a.ldx + ci0 + 32 * c32 + 8 * slot);
      }
      glds16(src, (float*)(buf + i * 1024));
    }
  };
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int coloff = (16 * (g & 1) + 4 * pp) * 2;
  const int a_off = (8 * (g >> 1) + q) * 64 + coloff;             // dz: row 8h + q of a 16-pixel sub-tile (+256: rows +4)
  int b_off[2][2];                                                // x: halo pixel of k = 16 kk + 8h + q (+4), tap (0, 0)
#pragma unroll
  for (int kk = 0; kk < 2; kk++)
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int k = 16 * kk + 8 * (g >> 1) + q + 4 * u;
      const int py = k / PWS, px = min(k % PWS, PWV - 1);         // padded slots: any in-range halo pixel (their dz rows are zero)
      b_off[kk][u] = XB + wn * 4096 + ((py + 1) * HW_ + px + 1) * 64 + coloff;
    }
  stage(t_begin, smem_raw);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = 0; s < S; s++) {
    if (s + 1 < S) stage(t_begin + s + 1, smem_raw + ((s + 1) & 1) * STAGE);
    const unsigned char* sb = smem_raw + (s & 1) * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      const bf16x8 af = tr_frag(sb + ((kk * 2 + wm) << 10), a_off);
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const int d = ((t / 3 - 1) * HW_ + (t % 3 - 1)) * 64;     // compile-time row shift of the tap
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb + b_off[kk][0] + d));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb + b_off[kk][1] + d));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, v), acc[t], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 9; t++) {
    float* out = a.slabs + ((long)split * 9 + t) * a.Cout * a.Cin;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int co = co0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h;
      out[(long)co * a.Cin + ci0 + 32 * wn + j] = acc[t][r];
    }
  }
}

// out[co][ci][tap] (OIHW when taps == 9, [co][ci] when taps == 1) (+)= sum over splits of slabs[split][tap][co][ci]
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, int splits, int taps, int Cout, int Cin, float* __restrict__ out, int accumulate) {
  // thread = 4 consecutive (tap, co, ci) entries (Cin % 64 == 0: never across a row); the slab loop keeps 8 independent 16-byte
  // loads in flight (one dependent load per split made the kernel latency-bound: 64 splits x ~1 us) and adds them in a fixed order
  const long per = (long)taps * Cout * Cin, per4 = per >> 2;
  const f32x4* s4 = (const f32x4*)slabs;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < per4; o += (long)gridDim.x * blockDim.x) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= splits; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = s4[(k + u) * per4 + o];
#pragma unroll
      for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; k < splits; k++) s += s4[k * per4 + o];
    const long e = o << 2;
    const int ci = (int)(e % Cin);
    const long t2 = e / Cin;
    const int co = (int)(t2 % Cout), tap = (int)(t2 / Cout);
    if (taps == 1) {
      f32x4* d = (f32x4*)(out + (long)co * Cin + ci);
      *d = accumulate ? *d + s : s;
    } else {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const long d = ((long)co * Cin + ci + u) * taps + tap;
        out[d] = accumulate ? out[d] + s[u] : s[u];
      }
    }
  }
}

// fp32 OIHW [Cout][Cin][3][3] -> bf16 [Cout][tap][Cin] (forward) or, dgrad: [Cin][tap][Cout] with the taps flipped
// (the weights of the convolution that maps dz to dx)
__global__ void pack3x3_taps_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int Cout, int Cin, int dgrad) {
  const long total = (long)Cout * Cin * 9;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    if (!dgrad) {
      const int ci = (int)(o % Cin);
      const long t = o / Cin;
      const int tap = (int)(t % 9), co = (int)(t / 9);
      dst[o] = f2bf(w[((long)co * Cin + ci) * 9 + tap]);
    } else {
      const int co = (int)(o % Cout);
      const long t = o / Cout;
      const int tap = (int)(t % 9), ci = (int)(t / 9);
      dst[o] = f2bf(w[((long)co * Cin + ci) * 9 + (8 - tap)]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// BatchNorm on bf16 rows [P][C], C in {64 .. 2048} a power of two.  thread = (row lane, 8-channel group).
// ------------------------------------------------------------------------------------------------------------------
// part[blk][2][C]: (sum z, sum z^2) of the block's rows;  BWD: (sum g, sum g z) with g = dy (* (ymask > 0)), g optionally stored
template <bool BWD>
__global__ __launch_bounds__(256) void bn_sums_bf16_kernel(const unsigned short* __restrict__ z, const unsigned short* __restrict__ dy,
                                                           const unsigned short* __restrict__ ymask, unsigned short* __restrict__ gout,
                                                           long P, int C, long per_blk, float* __restrict__ part) {
  __shared__ float red[2][256 * 8];
  const int CG = C >> 3, RPP = 256 / CG;
  const int cg = threadIdx.x % CG, rlane = threadIdx.x / CG;
  const long r0 = blockIdx.x * per_blk, r1 = min(P, r0 + per_blk);
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; e++) s[e] = q[e] = 0.f;
  auto add = [&](const us8 zv, const us8 dv, const us8 yv, long o) __attribute__((always_inline)) {
    if (!BWD) {
#pragma unroll
      for (int e = 0; e < 8; e++) { const float v = bf2f(zv[e]); s[e] += v; q[e] = fmaf(v, v, q[e]); }
    } else {
      us8 gv = dv;
      if (ymask) {
#pragma unroll
        for (int e = 0; e < 8; e++) gv[e] = bf2f(yv[e]) > 0.f ? dv[e] : (unsigned short)0;
      }
      if (gout) *(us8*)(gout + o) = gv;
#pragma unroll
      for (int e = 0; e < 8; e++) { const float gg = bf2f(gv[e]); s[e] += gg; q[e] = fmaf(gg, bf2f(zv[e]), q[e]); }
    }
  };
  const us8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
  long r = r0 + rlane;
  for (; r + 3L * RPP < r1; r += 4L * RPP) {                      // four rows in flight per thread
    us8 zv[4], dv[4], yv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const long o = (r + (long)u * RPP) * C + 8 * cg;
      zv[u] = *(const us8*)(z + o);
      dv[u] = BWD ? *(const us8*)(dy + o) : z8;
      yv[u] = (BWD && ymask) ? *(const us8*)(ymask + o) : z8;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) add(zv[u], dv[u], yv[u], (r + (long)u * RPP) * C + 8 * cg);
  }
  for (; r < r1; r += RPP) {
    const long o = r * C + 8 * cg;
    add(*(const us8*)(z + o), BWD ? *(const us8*)(dy + o) : z8, (BWD && ymask) ? *(const us8*)(ymask + o) : z8, o);
  }
#pragma unroll
  for (int e = 0; e < 8; e++) { red[0][threadIdx.x * 8 + e] = s[e]; red[1][threadIdx.x * 8 + e] = q[e]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float sa = 0.f, sb = 0.f;
    for (int k = 0; k < RPP; k++) {
      const int i = (k * CG + (c >> 3)) * 8 + (c & 7);
      sa += red[0][i];
      sb += red[1][i];
    }
    part[((long)blockIdx.x * 2) * C + c] = sa;
    part[((long)blockIdx.x * 2 + 1) * C + c] = sb;
  }
}

// out[chunk][w] (fp64) = sum of in[r][w] over the chunk's rows; grid (ceil(W2 / 64), nch), block = 64 columns x 4 row lanes
template <typename Tin>
__global__ __launch_bounds__(256) void colreduce_kernel(const Tin* __restrict__ in, int n, int W2, int per, double* __restrict__ out) {
  __shared__ double red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * per, r1 = min(n, r0 + per);
  double acc = 0.0;
  if (col < W2) {
    int r = r0 + rl;
    for (; r + 28 < r1; r += 32) {                                // eight independent loads in flight, added in a fixed order
      Tin v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = in[(long)(r + 4 * u) * W2 + col];
#pragma unroll
      for (int u = 0; u < 8; u++) acc += (double)v[u];
    }
    for (; r < r1; r += 4) acc += (double)in[(long)r * W2 + col];
  }
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && col < W2) out[(long)blockIdx.y * W2 + col] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// coef[0..3][C] = mean, invstd, scale = gamma * invstd, shift = beta - mean * scale.  training: from the partial sums (n rows of
// [2][C]) and the element count; else from the running statistics.  Running statistics / batch counter updated when training.
template <typename Tin>
__global__ void bn_fwd_final_kernel(const Tin* __restrict__ part, int n, int C, double count, float eps, float momentum,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ run_mean,
                                    float* __restrict__ run_var, long* __restrict__ tracked, int training, float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double m, var;
  if (training) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < n; k++) { a += (double)part[((long)k * 2) * C + c]; b += (double)part[((long)k * 2 + 1) * C + c]; }
    m = a / count;
    var = b / count - m * m;
    if (var < 0.0) var = 0.0;
    if (run_mean) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * m);
      run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
    }
    if (c == 0 && tracked) tracked[0] += 1;
  } else {
    m = run_mean[c];
    var = run_var[c];
  }
  const double is = 1.0 / sqrt(var + (double)eps);
  const double sc = (double)gamma[c] * is;
  coef[c] = (float)m;
  coef[C + c] = (float)is;
  coef[2 * C + c] = (float)sc;
  coef[3 * C + c] = (float)((double)beta[c] - m * sc);
}

// From the backward sums S0 = sum g, Sz = sum g z:  S1 = sum g xhat = invstd (Sz - mean S0);  dbeta (+)= S0_local, dgamma (+)= S1_local;
// dz = A g + Bc z + Cc with  A = gamma invstd,  Bc = -A invstd S1 / count,  Cc = A (mean invstd S1 - S0) / count  (training; the
// sums of `glob` and `count` are the whole batch's: all ranks' under SyncBN);  eval: A only.
template <typename Tin>
__global__ void bn_bwd_final_kernel(const Tin* __restrict__ loc, int nloc, const double* __restrict__ glob, int C, double count,
                                    const float* __restrict__ coef, const float* __restrict__ gamma, int training,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, float* __restrict__ bcoef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s0 = 0.0, sz = 0.0;
  for (int k = 0; k < nloc; k++) { s0 += (double)loc[((long)k * 2) * C + c]; sz += (double)loc[((long)k * 2 + 1) * C + c]; }
  const double m = coef[c], is = coef[C + c];
  const double s1 = is * (sz - m * s0);
  dbeta[c] = (float)((accumulate ? (double)dbeta[c] : 0.0) + s0);
  dgamma[c] = (float)((accumulate ? (double)dgamma[c] : 0.0) + s1);
  double g0 = s0, g1 = s1;
  if (glob) { g0 = glob[c]; g1 = is * (glob[C + c] - m * g0); }
  const double A = (double)gamma[c] * is;
  bcoef[c] = (float)A;
  bcoef[C + c] = training ? (float)(-A * is * g1 / count) : 0.f;
  bcoef[2 * C + c] = training ? (float)(A * (m * is * g1 - g0) / count) : 0.f;
}

// y = relu?(z * scale + shift (+ res));  thread = 8 channels, channel group fixed per thread (the grid stride is a multiple of C/8)
__global__ __launch_bounds__(256) void bn_apply_bf16_kernel(const unsigned short* __restrict__ z, const unsigned short* __restrict__ res,
                                                            unsigned short* __restrict__ y, const float* __restrict__ coef, long P, int C, int relu) {
  const int CG = C >> 3;
  const long total = P * CG;
  const long t0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const int cg = (int)(t0 % CG);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { sc[e] = coef[2 * C + 8 * cg + e]; sh[e] = coef[3 * C + 8 * cg + e]; }
  for (long o = t0; o < total; o += (long)gridDim.x * blockDim.x) {
    const us8 zv = *(const us8*)(z + o * 8);
    us8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (res) rv = *(const us8*)(res + o * 8);
    us8 ov;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float v = fmaf(bf2f(zv[e]), sc[e], sh[e]);
      if (res) v += bf2f(rv[e]);
      if (relu) v = fmaxf(v, 0.f);
      ov[e] = f2bf(v);
    }
    *(us8*)(y + o * 8) = ov;
  }
}

// dz = A g + Bc z + Cc, g = dy (* (ymask > 0))
__global__ __launch_bounds__(256) void bn_bwd_apply_bf16_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ ymask,
                                                                const unsigned short* __restrict__ z, unsigned short* __restrict__ dz,
                                                                const float* __restrict__ bcoef, long P, int C) {
  const int CG = C >> 3;
  const long total = P * CG;
  const long t0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const int cg = (int)(t0 % CG);
  float ca[8], cb[8], cc[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { ca[e] = bcoef[8 * cg + e]; cb[e] = bcoef[C + 8 * cg + e]; cc[e] = bcoef[2 * C + 8 * cg + e]; }
  for (long o = t0; o < total; o += (long)gridDim.x * blockDim.x) {
    const us8 gv = *(const us8*)(dy + o * 8);
    const us8 zv = *(const us8*)(z + o * 8);
    us8 yv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ymask) yv = *(const us8*)(ymask + o * 8);
    us8 ov;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float gg = bf2f(gv[e]);
      if (ymask) gg = bf2f(yv[e]) > 0.f ? gg : 0.f;
      ov[e] = f2bf(fmaf(ca[e], gg, fmaf(cb[e], bf2f(zv[e]), cc[e])));
    }
    *(us8*)(dz + o * 8) = ov;
  }
}

// ---- small bf16 helpers of the trainable trunk ----
// MaxPool2d(3, 2, 1) that also records which of the 9 window positions won (first maximum in scan order, like PyTorch)
__global__ void maxpool3x3_s2_bf16_idx_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, uint8_t* __restrict__ idx,
                                              int B, int H, int W, int C8, int Ho, int Wo) {
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < (long)B * Ho * Wo * C8; o += (long)gridDim.x * blockDim.x) {
    const int c = (int)(o % C8);
    long qq = o / C8;
    const int ox = (int)(qq % Wo); qq /= Wo;
    const int oy = (int)(qq % Ho);
    const long n = qq / Ho;
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { best[e] = -INFINITY; bi[e] = 0; }
    for (int ky = 0; ky < 3; ky++) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; kx++) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const us8 v = *(const us8*)(x + (((n * H + iy) * W + ix) * C8 + c) * 8);
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float f = bf2f(v[e]);
          if (f > best[e]) { best[e] = f; bi[e] = ky * 3 + kx; }
        }
      }
    }
    us8 r;
    unsigned long long packed = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) { r[e] = f2bf(best[e]); packed |= (unsigned long long)bi[e] << (8 * e); }
    *(us8*)(y + o * 8) = r;
    *(unsigned long long*)(idx + o * 8) = packed;
  }
}
// thread = (input pixel, 8 channels): gathers dy of the (up to 4) windows whose recorded winner is this pixel
__global__ void maxpool3x3_s2_bf16_bwd_kernel(const unsigned short* __restrict__ dy, const uint8_t* __restrict__ idx, unsigned short* __restrict__ dx,
                                              int B, int H, int W, int C8, int Ho, int Wo) {
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < (long)B * H * W * C8; o += (long)gridDim.x * blockDim.x) {
    const int c = (int)(o % C8);
    long qq = o / C8;
    const int ix = (int)(qq % W); qq /= W;
    const int iy = (int)(qq % H);
    const long n = qq / H;
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; e++) a[e] = 0.f;
    for (int oy = iy >> 1; oy <= (iy + 1) >> 1; oy++) {           // windows with 2 oy - 1 <= iy <= 2 oy + 1
      if (oy >= Ho) continue;
      const int ky = iy - (2 * oy - 1);
      for (int ox = ix >> 1; ox <= (ix + 1) >> 1; ox++) {
        if (ox >= Wo) continue;
        const int want = ky * 3 + (ix - (2 * ox - 1));
        const long w = (((n * Ho + oy) * Wo + ox) * C8 + c) * 8;
        const unsigned long long packed = *(const unsigned long long*)(idx + w);
        const us8 g = *(const us8*)(dy + w);
#pragma unroll
        for (int e = 0; e < 8; e++)
          if ((int)((packed >> (8 * e)) & 0xff) == want) a[e] += bf2f(g[e]);
      }
    }
    us8 r;
#pragma unroll
    for (int e = 0; e < 8; e++) r[e] = f2bf(a[e]);
    *(us8*)(dx + o * 8) = r;
  }
}
// dst[n][y][x][:] = (y, x both even) ? src[n][y/2][x/2][:] : 0   (the backward of a 2x subsample)
__global__ void upsample2_zero_bf16_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, int B, int H, int W, int C8) {
  const int Hs = H / 2, Ws = W / 2;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < (long)B * H * W * C8; o += (long)gridDim.x * blockDim.x) {
    const int c = (int)(o % C8);
    long qq = o / C8;
    const int x = (int)(qq % W); qq /= W;
    const int y = (int)(qq % H);
    const long n = qq / H;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!((x | y) & 1)) v = ((const f32x4*)src)[((n * Hs + (y >> 1)) * Ws + (x >> 1)) * C8 + c];
    ((f32x4*)dst)[o] = v;
  }
}
// dx[n][p][c] (bf16) = dy[n][c] (fp32) / HW
__global__ void global_avgpool_bf16_bwd_kernel(const float* __restrict__ dy, unsigned short* __restrict__ dx, int B, int HW, int C8) {
  const float s = 1.f / (float)HW;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < (long)B * HW * C8; o += (long)gridDim.x * blockDim.x) {
    const int c = (int)(o % C8);
    const long n = o / ((long)HW * C8);
    const f32x4 a = *(const f32x4*)(dy + (n * C8 + c) * 8), b = *(const f32x4*)(dy + (n * C8 + c) * 8 + 4);
    const us8 r = {f2bf(a[0] * s), f2bf(a[1] * s), f2bf(a[2] * s), f2bf(a[3] * s), f2bf(b[0] * s), f2bf(b[1] * s), f2bf(b[2] * s), f2bf(b[3] * s)};
    *(us8*)(dx + o * 8) = r;
  }
}

// ---- trainable VGG trunk (reference basic_vqa/models_lct.py:56-63, pretrained=False) on bf16 activations ----
// MaxPool2d(2, 2) on bf16 NHWC (floor: an odd last row / column is dropped)
__global__ void maxpool2x2_bf16_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int B, int H, int W, int C8) {
  const int Ho = H / 2, Wo = W / 2;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < (long)B * Ho * Wo * C8; o += (long)gridDim.x * blockDim.x) {
    const int c = (int)(o % C8);
    long qq = o / C8;
    const int ox = (int)(qq % Wo); qq /= Wo;
    const int oy = (int)(qq % Ho);
    const long n = qq / Ho;
    const long base = ((n * H + 2 * oy) * W + 2 * ox) * C8 + c;
    const us8 a = ((const us8*)x)[base], b = ((const us8*)x)[base + C8], cc = ((const us8*)x)[base + (long)W * C8], d = ((const us8*)x)[base + (long)W * C8 + C8];
    us8 r;
#pragma unroll
    for (int e = 0; e < 8; e++) r[e] = f2bf(fmaxf(fmaxf(bf2f(a[e]), bf2f(b[e])), fmaxf(bf2f(cc[e]), bf2f(d[e]))));
    ((us8*)y)[o] = r;
  }
}
// Backward of (ReLU -> [MaxPool2d(2,2)]) in one pass: act = the conv layer's post-ReLU output [B,H,W,C].
//   pool = 0: dz = d * (act > 0), d [B,H,W,C];   pool = 1: d [B,H/2,W/2,C] goes to the FIRST maximum of its 2x2 window (PyTorch's
//   tie rule) where that maximum is positive; an odd last row / column gets zeros.  thread = (2x2 window | pixel, 8 channels).
__global__ void relu_pool_bwd_bf16_kernel(const unsigned short* __restrict__ d, const unsigned short* __restrict__ act, unsigned short* __restrict__ dz,
                                          int B, int H, int W, int C8, int pool) {
  if (!pool) {
    for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < (long)B * H * W * C8; o += (long)gridDim.x * blockDim.x) {
      const us8 g = ((const us8*)d)[o], a = ((const us8*)act)[o];
      us8 r;
#pragma unroll
      for (int e = 0; e < 8; e++) r[e] = bf2f(a[e]) > 0.f ? g[e] : (unsigned short)0;
      ((us8*)dz)[o] = r;
    }
    return;
  }
  const int Hc = (H + 1) / 2, Wc = (W + 1) / 2, Ho = H / 2, Wo = W / 2;     // windows incl. the dropped odd edge
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < (long)B * Hc * Wc * C8; o += (long)gridDim.x * blockDim.x) {
    const int c = (int)(o % C8);
    long qq = o / C8;
    const int ox = (int)(qq % Wc); qq /= Wc;
    const int oy = (int)(qq % Hc);
    const long n = qq / Hc;
    const bool live = oy < Ho && ox < Wo;
    us8 g = {0, 0, 0, 0, 0, 0, 0, 0};
    if (live) g = ((const us8*)d)[((n * Ho + oy) * Wo + ox) * C8 + c];
    us8 a[4];
    long pos[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int iy = 2 * oy + (k >> 1), ix = 2 * ox + (k & 1);
      ok[k] = iy < H && ix < W;
      pos[k] = ((n * H + iy) * W + ix) * C8 + c;
      a[k] = ok[k] ? ((const us8*)act)[pos[k]] : us8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    us8 r[4];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      int best = 0;
      float bv = bf2f(a[0][e]);
#pragma unroll
      for (int k = 1; k < 4; k++) {
        const float v = bf2f(a[k][e]);
        if (v > bv) { bv = v; best = k; }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) r[k][e] = (live && k == best && bv > 0.f) ? g[e] : (unsigned short)0;
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (ok[k]) ((us8*)dz)[pos[k]] = r[k];
  }
}
// backward of AdaptiveAvgPool2d((7,7)) + flatten (torchvision's (c, y, x) feature order): fp32 dy [B][C*49] -> bf16 dx [B,H,W,C]
__global__ void adaptive_avgpool7_flatten_bwd_bf16_kernel(const float* __restrict__ dy, unsigned short* __restrict__ dx, int B, int H, int W, int C) {
  const long total = (long)B * H * W * C;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int c = (int)(o % C);
    const long pix = o / C;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H);
    const long n = pix / ((long)W * H);
    float a = 0.f;
    for (int oy = 0; oy < 7; oy++) {
      const int ys = (oy * H) / 7, ye = ((oy + 1) * H + 6) / 7;
      if (iy < ys || iy >= ye) continue;
      for (int ox = 0; ox < 7; ox++) {
        const int xs = (ox * W) / 7, xe = ((ox + 1) * W + 6) / 7;
        if (ix < xs || ix >= xe) continue;
        a += dy[(n * C + c) * 49 + oy * 7 + ox] / (float)((ye - ys) * (xe - xs));
      }
    }
    dx[o] = f2bf(a);
  }
}

// Stem weight gradient as a product: cols[p][k] (bf16, k = (c, ky, kx) < 147, zero up to 192) = the 7x7 / stride-2 / padding-3 window
// of output pixel p of the fp32 NCHW image; dW = dz^T . cols then runs on wgrad_bf16 (the VALU form, resnet.hip
// conv7x7_s2_stem_wgrad_kernel, spends 2.6 ms per B = 256 step on 60 GFLOP).  thread = (output pixel, 8 consecutive k).
__global__ void stem_im2col_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ cols, int B, int H, int W, int Ho, int Wo) {
  const long total = (long)B * Ho * Wo * 24;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int kg = (int)(o % 24);
    long p = o / 24;
    const int ox = (int)(p % Wo);
    const int oy = (int)((p / Wo) % Ho);
    const long n = p / ((long)Wo * Ho);
    us8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int k = kg * 8 + e;
      float f = 0.f;
      if (k < 147) {
        const int c = k / 49, ky = (k % 49) / 7, kx = k % 7;
        const int iy = 2 * oy - 3 + ky, ix = 2 * ox - 3 + kx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) f = x[((n * 3 + c) * H + iy) * W + ix];
      }
      v[e] = f2bf(f);
    }
    *(us8*)(cols + p * 192 + kg * 8) = v;
  }
}

int gs_blocks_tb(long n, int cap = 16384) {
  long nb = (n + 255) / 256;
  if (nb > cap) nb = cap;
  return (int)(nb < 1 ? 1 : nb);
}
// grid of the channel-fixed elementwise kernels: a multiple of 256 threads keeps (thread % (C/8)) constant along the grid stride
int apply_blocks(long groups) {
  long nb = (groups + 255) / 256;
  if (nb > 8192) nb = 8192;
  return (int)(nb < 1 ? 1 : nb);
}
bool bn_c_ok(int C) { return C >= 64 && C <= 2048 && (C & (C - 1)) == 0; }

}  // namespace

extern "C" {

// ---- conv_gemm_bf16 ----
int lctvqa_conv_gemm_bf16_supported(long M, int N, int Cin, int taps) {
  return (M >= 1 && M < (1L << 31) && N >= 64 && N % 64 == 0 && Cin >= 32 && Cin % 32 == 0 && (taps == 1 || taps == 9)) ? 1 : 0;
}
// rows per workgroup tile: 128 for the 1x1 products (95 VGPRs, 33 KB of LDS: four workgroups per CU cover each other's load
// latency and store tails; these products are HBM-bound), 256 for the tap loop and the 64-column tiles
// 1 when the wide (256 x 256, eight waves, ring) form takes the product
static int cg_wide(long M, int N, int Cin, int taps) {
  static const int forced = getenv("LCTVQA_CG_WIDE") ? atoi(getenv("LCTVQA_CG_WIDE")) : -1;
  const bool able = taps == 1 && N % 256 == 0 && Cin >= 128 && M >= 1024;
  if (forced >= 0) return forced && able;
  return able && Cin >= 1024 && N == 256 && M >= 32768;           // the one ResNet shape family it wins on (14x14 1024->256: 42 vs 52 us);
                                                                  // profiles/r04_conv_gemm_bf16_forms.txt has the whole A/B
}
static int cg_tile_rows(long M, int N, int taps) {
  static const int forced = getenv("LCTVQA_CG_ROWS") ? atoi(getenv("LCTVQA_CG_ROWS")) : 0;
  if (forced == 128 || forced == 256) return forced;
  (void)M;
  return (taps == 1 && N > 64) ? 128 : 256;     // measured per shape at B = 256 (tools/bench_conv_gemm_bf16.py): 128 rows = 4 workgroups per CU
}
// slabs (rows) of the statistics buffer [slabs][2][N] for this product
int lctvqa_conv_gemm_bf16_stats_slabs(long M, int N, int Cin, int taps) {
  const int bm = cg_wide(M, N, Cin, taps) ? 256 : cg_tile_rows(M, N, taps);
  return (int)(2 * ((M + bm - 1) / bm));
}
// C[M][N] bf16 = sum_tap sum_ci A[row + tap][ci] * B[n][tap * Cin + ci]  (* mask, + addend); taps = 9: rows are pixels of
// [.,H,W] images, taps outside the image contribute zero.  stats (or NULL): per-slab column (sum, sum sq) of the fp32 result.
int lctvqa_conv_gemm_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, long M, int N, int Cin, int taps, int H, int W,
                          const void* addend, long ld_add, const void* mask, long ld_mask, float* stats, void* stream) {
  LCT_REQUIRE(lctvqa_conv_gemm_bf16_supported(M, N, Cin, taps), "conv_gemm_bf16: need N %% 64 == 0, Cin %% 32 == 0, taps 1|9 (M=%ld N=%d Cin=%d taps=%d)", M, N, Cin, taps);
  LCT_REQUIRE((lda & 7) == 0 && (ldb & 7) == 0 && (ldc & 7) == 0 && (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0 && lda >= Cin &&
              ldb >= (long)taps * Cin && ldc >= N, "conv_gemm_bf16: operands must be 16-byte aligned bf16 rows");
  LCT_REQUIRE((!addend || ((ld_add & 7) == 0 && ((uintptr_t)addend & 15) == 0)) && (!mask || ((ld_mask & 7) == 0 && ((uintptr_t)mask & 15) == 0)),
              "conv_gemm_bf16: addend / mask rows must be 16-byte aligned");
  LCT_REQUIRE(taps == 1 || (H >= 1 && W >= 1 && M % ((long)H * W) == 0), "conv_gemm_bf16: taps = 9 needs M = B*H*W");
  CgArgs a{(const unsigned short*)A, lda, (const unsigned short*)B, ldb, (unsigned short*)C, ldc, (int)M, N, Cin, taps, H, W,
           (const unsigned short*)addend, ld_add, (const unsigned short*)mask, ld_mask, stats};
  hipStream_t st = (hipStream_t)stream;
  if (cg_wide(M, N, Cin, taps)) {
    static const int depth = getenv("LCTVQA_CG_WIDE_D") ? atoi(getenv("LCTVQA_CG_WIDE_D")) : 3;
    const long mtl = (M + 255) / 256, groups = (mtl + 7) / 8;
    const dim3 grid((unsigned)(groups * 8 * (N / 256)));
    if (depth == 4) {
      const size_t lds = (size_t)4 * 2 * 256 * 16 * sizeof(float);            // 128 KB
      auto k = conv_gemm_bf16_wide_kernel<4>;
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k, grid, dim3(512), lds, st, a);
    } else {
      const size_t lds = (size_t)3 * 2 * 256 * 16 * sizeof(float);            // 96 KB (the epilogue's 8 x 32 x 65 floats fit inside)
      auto k = conv_gemm_bf16_wide_kernel<3>;
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k, grid, dim3(512), lds, st, a);
    }
    return lct_check_launch("conv_gemm_bf16_wide");
  }
  const int bm = cg_tile_rows(M, N, taps);
  const long mt = (M + bm - 1) / bm;
#define CG_LAUNCH(NT2v, TAPSv, MTv)                                                                                \
  {                                                                                                               \
    size_t lds = (size_t)(2 * 64 * MTv * 16 + 2 * 64 * NT2v * 16) * sizeof(float);                                 \
    const size_t epi = (size_t)4 * 32 * (32 * NT2v + 1) * sizeof(float);                                           \
    if (lds < epi) lds = epi;                                                                                      \
    auto k = conv_gemm_bf16_kernel<NT2v, TAPSv, MTv>;                                                               \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
    hipLaunchKernelGGL(k, dim3((unsigned)(((mt + 7) / 8) * 8 * (N / (64 * NT2v)))), dim3(256), lds, st, a);        \
  }
  if (bm == 128) { if (N % 128 == 0) CG_LAUNCH(2, 1, 2) else CG_LAUNCH(1, 1, 2) }
  else if (N % 128 == 0) { if (taps == 1) CG_LAUNCH(2, 1, 4) else CG_LAUNCH(2, 9, 4) }
  else { if (taps == 1) CG_LAUNCH(1, 1, 4) else CG_LAUNCH(1, 9, 4) }
#undef CG_LAUNCH
  return lct_check_launch("conv_gemm_bf16");
}

// ---- wgrad_bf16 ----
// forms: 1 = ring, 8 stages, one workgroup per CU; 2 = two-stage, ~1024 workgroups; 3 = ring, 4 stages, two workgroups per CU;
// 4 = two-stage, ~512 workgroups.  LCTVQA_WGRAD_MODE overrides the per-shape choice (A/B: tools/bench_wgrad_bf16.py).
static int wgrad_mode(long P, int taps) {
  static const int forced = getenv("LCTVQA_WGRAD_MODE") ? atoi(getenv("LCTVQA_WGRAD_MODE")) : 0;
  if (forced) return forced;
  return (taps == 1 && P >= 150000) ? 3 : 2;                     // measured per shape at B = 256: profiles/r04_wgrad_bf16_forms.txt
}
// patch form of the 3x3 weight gradient: (PH, PWS, PWV) by map size, 0 when no patch shape tiles the map
static int wg3_shape(int H, int W, int Cout, int Cin, int taps) {
  static const int off = getenv("LCTVQA_WGRAD3_OFF") != nullptr;
  if (off || taps != 9 || Cout % 64 || Cin % 64 || H < 1 || W < 1) return 0;
  if (H % 4 == 0 && W % 8 == 0) return 1;                         // 4 x 8 patches
  if (H % 4 == 0 && W % 7 == 0) return 2;                         // 4 x 7 of 8 slots (28 x 28)
  if (H % 2 == 0 && W % 14 == 0) return 3;                        // 2 x 14 of 16 slots (14 x 14)
  return 0;
}
static void wg3_plan(long P, int H, int W, int Cout, int Cin, int shape, int* npatch, int* pps, int* splits) {
  const int ph = shape == 3 ? 2 : 4, pwv = shape == 1 ? 8 : (shape == 2 ? 7 : 14);
  const long B = P / ((long)H * W);
  *npatch = (int)(B * (H / ph) * (W / pwv));
  const long tiles = (long)(Cout / 64) * (Cin / 64);
  long s = 512 / tiles;                                           // about two workgroups per CU
  if (s > *npatch / 8) s = *npatch / 8;
  s = (s / 8) * 8;
  if (s < 8) s = 8;
  *pps = (int)((*npatch + s - 1) / s);
  *splits = (int)s;
}
static void wgrad_plan(long P, int Cout, int Cin, int taps, int* wm, int* ntm, int* ntn, int* splits, int* sps) {
  *wm = Cout >= 128 ? 2 : 1;
  const int wn = 4 / *wm;
  const int ncb = taps * (Cin / 64);
  *ntm = Cout / (64 * *wm);
  *ntn = (ncb + wn - 1) / wn;
  const long tiles = (long)*ntm * *ntn;
  const int steps_total = (int)((P + 31) / 32);
  const int mode = wgrad_mode(P, taps);
  if (mode == 2 || mode == 4) {
    long s = (mode == 2 ? 1024 : 512) / tiles;
    if (s > steps_total / 8) s = steps_total / 8;
    if (s >= 8 && taps == 1) s = (s / 8) * 8;                     // whole groups of 8: every XCD owns the same number of splits
    if (s < 1) s = 1;
    *sps = (int)((steps_total + s - 1) / s);
    *splits = (steps_total + *sps - 1) / *sps;
    return;
  }
  // ring forms: one (two) workgroups per CU, splits a multiple of 8 (one residue class per XCD), at least 16 steps per split
  long s = ((mode == 1 ? 256 : 512) + tiles - 1) / tiles;
  s = ((s + 7) / 8) * 8;
  while (s > 8 && steps_total / s < 16) s -= 8;
  *sps = (int)((steps_total + s - 1) / s);
  *splits = (int)s;
}
long lctvqa_wgrad_bf16_workspace_bytes(long P, int Cout, int Cin, int taps, int H, int W) {
  int wm, ntm, ntn, splits, sps;
  const int shape = wg3_shape(H, W, Cout, Cin, taps);
  if (shape) {
    int npatch, pps;
    wg3_plan(P, H, W, Cout, Cin, shape, &npatch, &pps, &splits);
  } else {
    wgrad_plan(P, Cout, Cin, taps, &wm, &ntm, &ntn, &splits, &sps);
  }
  return (long)splits * taps * Cout * Cin * (long)sizeof(float);
}
// dW (+)= sum_p dz[p][co] * x[p + tap][ci]: dW fp32 [Cout][Cin] (taps = 1) or OIHW [Cout][Cin][3][3] (taps = 9; rows are pixels
// of [.,H,W] images).  workspace: lctvqa_wgrad_bf16_workspace_bytes.
int lctvqa_wgrad_bf16(const void* dz, long ldz, const void* x, long ldx, long P, int Cout, int Cin, int taps, int H, int W, float* dW,
                      int accumulate, float* workspace, long workspace_bytes, void* stream) {
  LCT_REQUIRE(P >= 1 && P < (1L << 31) && Cout >= 64 && Cout % 64 == 0 && Cin >= 64 && Cin % 64 == 0 && (taps == 1 || taps == 9) &&
              (Cout == 64 || Cout % 128 == 0), "wgrad_bf16: need Cout, Cin multiples of 64 (Cout 64 or a multiple of 128), taps 1|9");
  LCT_REQUIRE((ldz & 7) == 0 && (ldx & 7) == 0 && (((uintptr_t)dz | (uintptr_t)x) & 15) == 0 && ldz >= Cout && ldx >= Cin,
              "wgrad_bf16: operands must be 16-byte aligned bf16 rows");
  LCT_REQUIRE(taps == 1 || (H >= 1 && W >= 1 && P % ((long)H * W) == 0), "wgrad_bf16: taps = 9 needs P = B*H*W");
  int wm, ntm, ntn, splits, sps;
  hipStream_t st = (hipStream_t)stream;
  const int shape = wg3_shape(H, W, Cout, Cin, taps);
  if (shape) {
    int npatch, pps;
    wg3_plan(P, H, W, Cout, Cin, shape, &npatch, &pps, &splits);
    LCT_REQUIRE(workspace && workspace_bytes >= (long)splits * 9 * Cout * Cin * (long)sizeof(float), "wgrad_bf16: workspace too small");
    const int ph = shape == 3 ? 2 : 4, pwv = shape == 1 ? 8 : (shape == 2 ? 7 : 14);
    Wg3Args g{(const unsigned short*)dz, ldz, (const unsigned short*)x, ldx, workspace, (int)(P / ((long)H * W)), H, W, Cout, Cin, pps, npatch,
              H / ph, W / pwv};
    const unsigned grid3 = (unsigned)((long)(Cout / 64) * (Cin / 64) * splits);
    const size_t lds3 = 2 * 12 * 1024;
#define WG3(PHv, PWSv, PWVv)                                                                                   \
    {                                                                                                          \
      auto k = wgrad3x3_bf16_kernel<PHv, PWSv, PWVv>;                                                            \
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);         \
      hipLaunchKernelGGL(k, dim3(grid3), dim3(256), lds3, st, g);                                               \
    }
    if (shape == 1) WG3(4, 8, 8) else if (shape == 2) WG3(4, 8, 7) else WG3(2, 16, 14)
#undef WG3
    int rc3 = lct_check_launch("wgrad3x3_bf16");
    if (rc3) return rc3;
    const int live3 = (npatch + pps - 1) / pps;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gs_blocks_tb((long)9 * Cout * Cin, 4096)), dim3(256), 0, st, workspace, live3, 9, Cout, Cin, dW,
                       accumulate);
    return lct_check_launch("wgrad_reduce");
  }
  wgrad_plan(P, Cout, Cin, taps, &wm, &ntm, &ntn, &splits, &sps);
  LCT_REQUIRE(workspace && workspace_bytes >= (long)splits * taps * Cout * Cin * (long)sizeof(float), "wgrad_bf16: workspace too small");
  WgArgs a{(const unsigned short*)dz, ldz, (const unsigned short*)x, ldx, workspace, (int)P, Cout, Cin, taps, H > 0 ? H : 1, W > 0 ? W : 1, sps,
           taps * (Cin / 64), ntm, ntn, taps == 1 ? 1 : 0};
  const unsigned grid = (unsigned)((long)ntm * ntn * (((splits + 7) / 8) * 8));       // both forms deal splits to XCDs in groups of 8
  int live = splits;                                              // slabs that are written (splits beyond the pixel range are not)
  const int mode = wgrad_mode(P, taps);
#define WG_RING(WMv, WNv, Dv)                                                                                    \
  {                                                                                                              \
    const size_t lds = (size_t)Dv * (4 * WMv + 4 * WNv) * 1024;                                                   \
    auto k = wgrad_bf16_ring_kernel<WMv, WNv, Dv>;                                                                \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);                                                     \
  }
  if (mode == 2 || mode == 4) {
    if (wm == 2) {
      const size_t lds = 2 * (size_t)(8 + 8) * 1024;
      auto k = wgrad_bf16_tn_kernel<2, 2>;
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
    } else {
      const size_t lds = 2 * (size_t)(4 + 16) * 1024;
      auto k = wgrad_bf16_tn_kernel<1, 4>;
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
    }
  } else {
    const int steps_total = (int)((P + 31) / 32);
    live = (steps_total + sps - 1) / sps;
    if (mode == 1) { if (wm == 2) WG_RING(2, 2, 8) else WG_RING(1, 4, 6) }
    else { if (wm == 2) WG_RING(2, 2, 4) else WG_RING(1, 4, 3) }
  }
#undef WG_RING
  int rc = lct_check_launch("wgrad_bf16");
  if (rc) return rc;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gs_blocks_tb((long)taps * Cout * Cin, 4096)), dim3(256), 0, st, workspace, live, taps, Cout, Cin, dW,
                     accumulate);
  return lct_check_launch("wgrad_reduce");
}

int lctvqa_pack3x3_taps_bf16(const float* w_oihw, void* dst, int Cout, int Cin, int dgrad, void* stream) {
  LCT_REQUIRE(Cout > 0 && Cin > 0, "pack3x3_taps_bf16: bad shape");
  hipLaunchKernelGGL(pack3x3_taps_kernel, dim3(gs_blocks_tb((long)Cout * Cin * 9, 4096)), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     (unsigned short*)dst, Cout, Cin, dgrad);
  return lct_check_launch("pack3x3_taps_bf16");
}

// ---- BatchNorm on bf16 rows ----
int lctvqa_bn_bf16_blocks(long P, int C) {
  if (!bn_c_ok(C)) return 0;
  const int rpp = 256 / (C / 8);
  long nb = (P + 16L * rpp - 1) / (16L * rpp);                   // >= 16 rows per thread
  if (nb > 2048) nb = 2048;
  return (int)(nb < 1 ? 1 : nb);
}
// part [lctvqa_bn_bf16_blocks][2][C] = per-workgroup (sum z, sum z^2)
int lctvqa_bn_bf16_stats(const void* z, long P, int C, float* part, void* stream) {
  LCT_REQUIRE(bn_c_ok(C) && P >= 1 && ((uintptr_t)z & 15) == 0, "bn_bf16_stats: C must be a power of two in 64..2048");
  const int nb = lctvqa_bn_bf16_blocks(P, C);
  hipLaunchKernelGGL(bn_sums_bf16_kernel<false>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)z, (const unsigned short*)nullptr,
                     (const unsigned short*)nullptr, (unsigned short*)nullptr, P, C, (P + nb - 1) / nb, part);
  return lct_check_launch("bn_bf16_stats");
}
// part [blocks][2][C] = (sum g, sum g z), g = dy * (ymask > 0) (ymask NULL: g = dy); g_out (or NULL) receives g
int lctvqa_bn_bf16_bwd_sums(const void* dy, const void* ymask, const void* z, void* g_out, long P, int C, float* part, void* stream) {
  LCT_REQUIRE(bn_c_ok(C) && P >= 1 && (((uintptr_t)z | (uintptr_t)dy | (uintptr_t)ymask | (uintptr_t)g_out) & 15) == 0,
              "bn_bf16_bwd_sums: C must be a power of two in 64..2048, tensors 16-byte aligned");
  const int nb = lctvqa_bn_bf16_blocks(P, C);
  hipLaunchKernelGGL(bn_sums_bf16_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)z, (const unsigned short*)dy,
                     (const unsigned short*)ymask, (unsigned short*)g_out, P, C, (P + nb - 1) / nb, part);
  return lct_check_launch("bn_bf16_bwd_sums");
}
// out [chunks][W2] fp64 = column sums of in [n][W2] (fp32) over `chunks` row ranges (fixed order)
int lctvqa_colreduce_f32(const float* in, int n, int W2, int chunks, double* out, void* stream) {
  LCT_REQUIRE(n >= 1 && W2 >= 1 && chunks >= 1, "colreduce: bad shape");
  const int per = (n + chunks - 1) / chunks;
  hipLaunchKernelGGL(colreduce_kernel<float>, dim3((W2 + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, in, n, W2, per, out);
  return lct_check_launch("colreduce_f32");
}
// coef [4][C] = mean, invstd, scale, shift.  part: n rows of [2][C] (sum, sum sq), fp64 when part_f64 else fp32; count = elements
// per channel behind those sums (all ranks' under SyncBN).  training = 0: running statistics, part unused.
int lctvqa_bn_bf16_fwd_final(const void* part, int part_f64, int n, int C, double count, float eps, float momentum, const float* gamma,
                             const float* beta, float* run_mean, float* run_var, long* tracked, int training, float* coef, void* stream) {
  LCT_REQUIRE(C >= 1 && (training ? (part && n >= 1 && count >= 1.0) : (run_mean && run_var)), "bn_bf16_fwd_final: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (part_f64)
    hipLaunchKernelGGL(bn_fwd_final_kernel<double>, dim3((C + 63) / 64), dim3(64), 0, st, (const double*)part, n, C, count, eps, momentum, gamma, beta,
                       run_mean, run_var, tracked, training, coef);
  else
    hipLaunchKernelGGL(bn_fwd_final_kernel<float>, dim3((C + 63) / 64), dim3(64), 0, st, (const float*)part, n, C, count, eps, momentum, gamma, beta,
                       run_mean, run_var, tracked, training, coef);
  return lct_check_launch("bn_bf16_fwd_final");
}
// bcoef [3][C] (A, Bc, Cc of dz = A g + Bc z + Cc); dgamma / dbeta (+)= from the LOCAL sums `loc` (n rows of [2][C]); glob
// (or NULL = loc): fp64 [2][C] whole-batch sums with `count` elements per channel.
int lctvqa_bn_bf16_bwd_final(const void* loc, int loc_f64, int n, const double* glob, int C, double count, const float* coef, const float* gamma,
                             int training, float* dgamma, float* dbeta, int accumulate, float* bcoef, void* stream) {
  LCT_REQUIRE(C >= 1 && loc && n >= 1 && count >= 1.0, "bn_bf16_bwd_final: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (loc_f64)
    hipLaunchKernelGGL(bn_bwd_final_kernel<double>, dim3((C + 63) / 64), dim3(64), 0, st, (const double*)loc, n, glob, C, count, coef, gamma, training,
                       dgamma, dbeta, accumulate, bcoef);
  else
    hipLaunchKernelGGL(bn_bwd_final_kernel<float>, dim3((C + 63) / 64), dim3(64), 0, st, (const float*)loc, n, glob, C, count, coef, gamma, training,
                       dgamma, dbeta, accumulate, bcoef);
  return lct_check_launch("bn_bf16_bwd_final");
}
// y = relu?(z * coef.scale + coef.shift (+ res)) on bf16 rows [P][C]
int lctvqa_bn_bf16_apply(const void* z, const void* res, void* y, const float* coef, long P, int C, int relu, void* stream) {
  LCT_REQUIRE(bn_c_ok(C) && P >= 1 && (((uintptr_t)z | (uintptr_t)res | (uintptr_t)y) & 15) == 0, "bn_bf16_apply: bad shape / alignment");
  hipLaunchKernelGGL(bn_apply_bf16_kernel, dim3(apply_blocks(P * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)z,
                     (const unsigned short*)res, (unsigned short*)y, coef, P, C, relu);
  return lct_check_launch("bn_bf16_apply");
}
// dz = A g + Bc z + Cc, g = dy * (ymask > 0) (ymask NULL: g = dy)
int lctvqa_bn_bf16_bwd_apply(const void* dy, const void* ymask, const void* z, void* dz, const float* bcoef, long P, int C, void* stream) {
  LCT_REQUIRE(bn_c_ok(C) && P >= 1 && (((uintptr_t)z | (uintptr_t)dy | (uintptr_t)ymask | (uintptr_t)dz) & 15) == 0, "bn_bf16_bwd_apply: bad shape / alignment");
  hipLaunchKernelGGL(bn_bwd_apply_bf16_kernel, dim3(apply_blocks(P * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)dy,
                     (const unsigned short*)ymask, (const unsigned short*)z, (unsigned short*)dz, bcoef, P, C);
  return lct_check_launch("bn_bf16_bwd_apply");
}

// ---- helpers ----
int lctvqa_maxpool3x3_s2_bf16_idx(const void* x, void* y, void* idx_u8, int B, int H, int W, int C, void* stream) {
  LCT_REQUIRE(C % 8 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)idx_u8) & 15) == 0, "maxpool3x3_s2_bf16_idx: need C %% 8 == 0, aligned tensors");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3_s2_bf16_idx_kernel, dim3(gs_blocks_tb((long)B * Ho * Wo * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x, (unsigned short*)y, (uint8_t*)idx_u8, B, H, W, C / 8, Ho, Wo);
  return lct_check_launch("maxpool3x3_s2_bf16_idx");
}
int lctvqa_maxpool3x3_s2_bf16_bwd(const void* dy, const void* idx_u8, void* dx, int B, int H, int W, int C, void* stream) {
  LCT_REQUIRE(C % 8 == 0 && (((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)idx_u8) & 15) == 0, "maxpool3x3_s2_bf16_bwd: need C %% 8 == 0, aligned tensors");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3_s2_bf16_bwd_kernel, dim3(gs_blocks_tb((long)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)dy, (const uint8_t*)idx_u8, (unsigned short*)dx, B, H, W, C / 8, Ho, Wo);
  return lct_check_launch("maxpool3x3_s2_bf16_bwd");
}
// dst [B][H][W][C] = src [B][H/2][W/2][C] at the even pixels, zero elsewhere
int lctvqa_upsample2_zero_bf16(const void* src, void* dst, int B, int H, int W, int C, void* stream) {
  LCT_REQUIRE(C % 8 == 0 && H % 2 == 0 && W % 2 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "upsample2_zero_bf16: need C %% 8 == 0, even H, W");
  hipLaunchKernelGGL(upsample2_zero_bf16_kernel, dim3(gs_blocks_tb((long)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)src, (unsigned short*)dst, B, H, W, C / 8);
  return lct_check_launch("upsample2_zero_bf16");
}
int lctvqa_global_avgpool_bf16_bwd(const float* dy, void* dx, int B, int HW, int C, void* stream) {
  LCT_REQUIRE(C % 8 == 0 && (((uintptr_t)dy | (uintptr_t)dx) & 15) == 0, "global_avgpool_bf16_bwd: need C %% 8 == 0, aligned tensors");
  hipLaunchKernelGGL(global_avgpool_bf16_bwd_kernel, dim3(gs_blocks_tb((long)B * HW * (C / 8))), dim3(256), 0, (hipStream_t)stream, dy,
                     (unsigned short*)dx, B, HW, C / 8);
  return lct_check_launch("global_avgpool_bf16_bwd");
}

// cols [B*Ho*Wo][192] bf16 = the stem convolution's input windows (k = (c, ky, kx), zero beyond 147), Ho = (H-1)/2+1
int lctvqa_stem_im2col_bf16(const float* x_nchw, void* cols, int B, int H, int W, void* stream) {
  LCT_REQUIRE(B > 0 && H >= 7 && W >= 7 && ((uintptr_t)cols & 15) == 0, "stem_im2col_bf16: bad shape / alignment");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(stem_im2col_bf16_kernel, dim3(gs_blocks_tb((long)B * Ho * Wo * 24)), dim3(256), 0, (hipStream_t)stream, x_nchw,
                     (unsigned short*)cols, B, H, W, Ho, Wo);
  return lct_check_launch("stem_im2col_bf16");
}
int lctvqa_maxpool2x2_bf16(const void* x, void* y, int B, int H, int W, int C, void* stream) {
  LCT_REQUIRE(C % 8 == 0 && H >= 2 && W >= 2 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, "maxpool2x2_bf16: need C %% 8 == 0, aligned tensors");
  hipLaunchKernelGGL(maxpool2x2_bf16_kernel, dim3(gs_blocks_tb((long)B * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x, (unsigned short*)y, B, H, W, C / 8);
  return lct_check_launch("maxpool2x2_bf16");
}
// dz [B,H,W,C] = backward of ReLU (-> MaxPool2d(2,2) when pool) given the layer's post-ReLU output `act` [B,H,W,C] and the gradient
// d of what followed ([B,H,W,C], or [B,H/2,W/2,C] when pool)
int lctvqa_relu_pool_bwd_bf16(const void* d, const void* act, void* dz, int B, int H, int W, int C, int pool, void* stream) {
  LCT_REQUIRE(C % 8 == 0 && (((uintptr_t)d | (uintptr_t)act | (uintptr_t)dz) & 15) == 0, "relu_pool_bwd_bf16: need C %% 8 == 0, aligned tensors");
  const long n = pool ? (long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8) : (long)B * H * W * (C / 8);
  hipLaunchKernelGGL(relu_pool_bwd_bf16_kernel, dim3(gs_blocks_tb(n)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)d,
                     (const unsigned short*)act, (unsigned short*)dz, B, H, W, C / 8, pool);
  return lct_check_launch("relu_pool_bwd_bf16");
}
int lctvqa_adaptive_avgpool7_flatten_bwd_bf16(const float* dy, void* dx_nhwc_bf16, int B, int H, int W, int C, void* stream) {
  hipLaunchKernelGGL(adaptive_avgpool7_flatten_bwd_bf16_kernel, dim3(gs_blocks_tb((long)B * H * W * C, 8192)), dim3(256), 0, (hipStream_t)stream, dy,
                     (unsigned short*)dx_nhwc_bf16, B, H, W, C);
  return lct_check_launch("adaptive_avgpool7_flatten_bwd_bf16");
}

}  // extern "C"
