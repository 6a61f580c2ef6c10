// One PC-DARTS MixedOp edge (reference basic_vqa/pcdarts/model_search.py:22-55, operations.py:4-104) as a handful of
// tile kernels instead of one launch and one HBM tensor per primitive and per BatchNorm pass.
//
//   x [B,H,W,C] NHWC, Cq = C/4, x1 = x[..., :Cq]; stride s (2 on the input edges of a reduction cell); P = B*Ho*Wo
//   slot  branch                                 pre-BN tensor (all [P,Cq], batch-statistics BatchNorm, affine=False)
//    0    max_pool_3x3(s)                         u0
//    1    avg_pool_3x3(s), count_include_pad=0    u1
//    2    skip_connect: s=1 identity (no BN); s=2 FactorizedReduce = relu -> two 1x1 stride-2 convs (even / odd) -> cat   u2
//    3    sep_conv_3x3 stage a: relu -> dw3(s) -> 1x1                        t3     7  stage b: bn -> relu -> dw3(1) -> 1x1   u3
//    4    sep_conv_5x5 stage a: relu -> dw5(s) -> 1x1                        t4     8  stage b: bn -> relu -> dw5(1) -> 1x1   u4
//    5    dil_conv_3x3: relu -> dw3(s, dilation 2) -> 1x1                    u5
//    6    dil_conv_5x5: relu -> dw5(s, dilation 2) -> 1x1                    u6
//   mix = sum_k w_k bn_k(u_k);  out (+)= beta * channel_shuffle(cat(mix, bypass(x[..., Cq:])), 4)
//
// Forward = three tile kernels separated by the two batch-statistics reductions BatchNorm needs:
//   edge_fwd_a   x1 halo tile in LDS (read once for all seven branches) -> slots 0..6 + per-workgroup partial (sum, sum^2)
//   edge_fwd_b   t3 / t4 halo tiles, normalised + ReLU'd on load -> slots 7, 8 + partials
//   edge_fwd_c   normalise the seven final slots, alpha-weighted sum, bypass (2x2 max-pooled on reduction edges),
//                channel shuffle, beta-accumulate into the node's running sum
// with edge_stats_finish (fixed-order fp64 finish, running statistics, num_batches_tracked) after a and b.
// Thread = one output pixel with ALL Cq channels in registers (the 1x1 convolutions never leave the thread); LDS tiles are
// planar per channel quad ([quad][position] of float4) so neighbouring pixels read neighbouring 16-byte words.
// Roofline: HBM (Cq = 4..16 cannot feed an MFMA tile); algorithmic bytes per edge forward = (1 + 9 + 2 + 7 + 3 + 4 + 4) T
// with T = 4*P*Cq bytes.
#include "common.h"

#include <mutex>
#include <type_traits>
#include <unordered_map>

namespace {

constexpr int NSLOT = 9;
struct EdgeW {
  const float* dw[6];   // depthwise weights [Cq][K][K]:  0 sep3a 1 sep5a 2 dil3 3 dil5 4 sep3b 5 sep5b
  const float* pw[6];   // 1x1 weights [Cq][Cq] (out, in), same order
  const float* fr[2];   // FactorizedReduce conv_1 / conv_2 [Cq/2][Cq] (stride-2 edges)
};
struct EdgeBN {         // per slot: running statistics + batch counter (null when untracked / identity)
  float* run_mean[NSLOT];
  float* run_var[NSLOT];
  long* tracked[NSLOT];
};

// Storage type of the nine pre-BN tensors.  The file is compiled twice: as it is (fp32, the entry points lctvqa_mixed_edge_*) and from
// mixed_edge_bf16.hip with EDGE_PRE_BF16 (bf16 storage, entry points lctvqa_mixed_edge_*_bf16pre; darts_ops selects them under the bf16
// compute dtype).  In the bf16 build a value is ROUNDED (nearest even) before both its store and its BatchNorm statistics, so that the
// statistics are those of the stored tensor and every later reader (normalisation, backward x-hat, ReLU masks of the second stages) sees
// the same numbers; all arithmetic stays fp32.
#ifdef EDGE_PRE_BF16
typedef unsigned short pre_t;
#define EDGE_SYM(name) name##_bf16pre
__device__ __forceinline__ float rndp(float v) {
  unsigned int u = __builtin_bit_cast(unsigned int, v);
  u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
  return __builtin_bit_cast(float, u);
}
__device__ __forceinline__ f32x4 ldp4(const pre_t* p) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 r = *(const u32x2*)p;
  return f32x4{__builtin_bit_cast(float, r[0] << 16), __builtin_bit_cast(float, r[0] & 0xffff0000u), __builtin_bit_cast(float, r[1] << 16),
               __builtin_bit_cast(float, r[1] & 0xffff0000u)};
}
__device__ __forceinline__ void stp4(pre_t* p, f32x4 v) {      // v holds bf16-representable values (every caller has applied rndp): the low halves are zero
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  // (element copies first: __builtin_bit_cast applied directly to an ext-vector ELEMENT expression reads element 0 with this compiler)
  const float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
  const unsigned int a = __builtin_bit_cast(unsigned int, v0) >> 16, b = __builtin_bit_cast(unsigned int, v1) & 0xffff0000u;
  const unsigned int c = __builtin_bit_cast(unsigned int, v2) >> 16, d = __builtin_bit_cast(unsigned int, v3) & 0xffff0000u;
  *(u32x2*)p = u32x2{a | b, c | d};
}
#else
typedef float pre_t;
#define EDGE_SYM(name) name
__device__ __forceinline__ float rndp(float v) { return v; }
#endif
__device__ __forceinline__ f32x4 ldp4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ void stp4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ f32x4 rndp4(f32x4 v) { return f32x4{rndp(v[0]), rndp(v[1]), rndp(v[2]), rndp(v[3])}; }

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
}

// Zero-padded halo tile of NQ channel quads: tile[q * TH_*TW_ + ly*TW_ + lx] = src[n, y0+ly, x0+lx, 4q..4q+3] (row stride ld
// floats); ACT 0: as stored, 1: ReLU'd, 2: normalised + ReLU'd ((v - mean) * invstd, then max 0: the input of a sep_conv's
// second stage).  The trip counts are compile-time, so the loads of up to 8 rounds are issued back to back before the first
// LDS store (one memory round trip per batch instead of one per round).
template <int NQ, int ACT, int TH_, int TW_, int TPB, int BATCH_ = 0, typename ST = float>
__device__ __forceinline__ void load_tile(f32x4* __restrict__ tile, const ST* __restrict__ src, long ld, int n, int H, int W,
                                          int y0, int x0, const float* __restrict__ mean, const float* __restrict__ invstd, int tid_ = -1) {
  constexpr int NPOS = TH_ * TW_, TOT = NPOS * NQ, ROUNDS = (TOT + TPB - 1) / TPB, BATCH = BATCH_ ? BATCH_ : (NQ >= 4 ? 4 : 8);
  const int tid = tid_ < 0 ? (int)threadIdx.x : tid_;            // a caller may pass a laundered copy (keeps the index math inside its tile loop)
#pragma unroll
  for (int r0 = 0; r0 < ROUNDS; r0 += BATCH) {
    f32x4 v[BATCH];
    bool ok[BATCH];
#pragma unroll
    for (int r = 0; r < BATCH; r++) {
      const int idx = tid + (r0 + r) * TPB;
      const int q = idx % NQ, pos = idx / NQ;
      const int ly = pos / TW_, lx = pos - ly * TW_;
      const int y = y0 + ly, x = x0 + lx;
      ok[r] = (r0 + r < ROUNDS) && idx < TOT && y >= 0 && y < H && x >= 0 && x < W;
      v[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok[r]) v[r] = ldp4(src + (((long)n * H + y) * W + x) * ld + 4 * q);
    }
#pragma unroll
    for (int r = 0; r < BATCH; r++) {
      const int idx = tid + (r0 + r) * TPB;
      if (r0 + r < ROUNDS && idx < TOT) {
        const int q = idx % NQ, pos = idx / NQ;
        f32x4 t = v[r];
        if (ACT == 2) {
          if (ok[r]) {
#pragma unroll
            for (int e = 0; e < 4; e++) t[e] = fmaxf((t[e] - mean[4 * q + e]) * invstd[4 * q + e], 0.f);
          }
        } else if (ACT == 1) {
          t = relu4(t);
        }
        tile[q * NPOS + pos] = t;
      }
    }
  }
}

// depthwise K x K at one output pixel: taps at (cy + DIL*j, cx + DIL*i) of the tile (cy, cx = top-left tap), ReLU'd on read
// TAPS: weights staged tap-major [K*K][CQ] (stage_weights_taps: one 16-byte broadcast read per tap and channel quad) or
// channel-major [CQ][K*K] as the parameter stores them (four 4-byte reads).  Same accumulation order either way (bit-identical);
// measured per kernel: the forward kernels gain from the tap-major form at Cq = 4 and lose at Cq = 8 / 16, the backward kernel gains.
typedef const __attribute__((address_space(4))) float* const_f32_ptr;   // constant address space: uniform loads become s_load (SGPR operands)
typedef const __attribute__((address_space(4))) f32x4* const_f32x4_ptr;
__device__ __forceinline__ const_f32_ptr scalar_ptr(const float* p) {     // per use site: the loads are re-issued there instead of held in SGPRs
  unsigned long a = (unsigned long)(uintptr_t)p;
  asm volatile("" : "+s"(a));
  return (const_f32_ptr)a;
}

template <int CQ, int K, int DIL, bool RELU, bool TAPS, typename WP = const float*>
__device__ __forceinline__ void dw_at(const f32x4* __restrict__ tile, int npos, int tw, int cy, int cx,
                                      WP wl, float* __restrict__ out) {
#pragma unroll
  for (int q = 0; q < CQ / 4; q++) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < K; j++)
#pragma unroll
      for (int i = 0; i < K; i++) {
        f32x4 v = tile[q * npos + (cy + DIL * j) * tw + cx + DIL * i];
        if (RELU) v = relu4(v);
        if constexpr (TAPS) {
          const f32x4 w4 = *(const f32x4*)(wl + (j * K + i) * CQ + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; e++) a[e] = fmaf(w4[e], v[e], a[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) a[e] = fmaf(wl[(4 * q + e) * K * K + j * K + i], v[e], a[e]);
        }
      }
#pragma unroll
    for (int e = 0; e < 4; e++) out[4 * q + e] = a[e];
  }
}

// out[oc] = sum_c w[oc][c] * in[c]  (w in LDS, uniform address across the wave: broadcast reads)
template <int NO, int NI, typename WP = const float*>
__device__ __forceinline__ void pw_at(WP wl, const float* __restrict__ in, float* __restrict__ out) {
#pragma unroll
  for (int oc = 0; oc < NO; oc++) {            // one accumulation chain per output in channel order (the compiler merges the weight
    float a = 0.f;                             // reads into 8-byte pairs): splitting it into four partial sums changes the rounding
#pragma unroll                                 // of the forward values, and with it which elements sit on a ReLU boundary
    for (int c = 0; c < NI; c++) a = fmaf(wl[oc * NI + c], in[c], a);
    out[oc] = a;
  }
}

__device__ __forceinline__ void stage_weights(float* __restrict__ dst, const float* __restrict__ src, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
// depthwise weights [CQ][KK] (the parameter's layout) -> tap-major [KK][CQ] in LDS
__device__ __forceinline__ void stage_weights_taps(float* __restrict__ dst, const float* __restrict__ src, int cq, int kk) {
  for (int i = threadIdx.x; i < cq * kk; i += blockDim.x) dst[(i % kk) * cq + i / kk] = src[i];
}


// Per-workgroup (sum, sum of squares) of the NS x CQ values every thread holds for ITS pixel (u[k][c], zeros when the thread
// is not live): per channel quad the 4*NS columns go through LDS (red[col][thread]), 8 row segments per column are summed
// by one thread each, then one thread per column adds the 8 segment sums -- in a fixed order -- into the workgroup's fp64
// accumulators acc[(slot_base + k) * CQ + c] (sums) and acc[NSTAT * CQ + ...] (sums of squares).
template <int CQ, int NS, int NSTAT, int TPB>
__device__ __forceinline__ void phase_stats(const float (*u)[CQ], bool live, float* __restrict__ red, float* __restrict__ seg,
                                            double* __restrict__ acc, int slot_base) {
  constexpr int NC = 4 * NS;
  const int t = threadIdx.x;
#ifdef EDGE_ABL_NOSTATS
  return;                                                      // ablation build (tools/ab_edge.py): what the statistics reduction costs
#endif
#pragma unroll
  for (int q = 0; q < CQ / 4; q++) {
    __syncthreads();                                           // red[] of the previous round consumed
#pragma unroll
    for (int k = 0; k < NS; k++)
#pragma unroll
      for (int e = 0; e < 4; e++) red[(k * 4 + e) * (TPB + 1) + t] = live ? u[k][4 * q + e] : 0.f;
    __syncthreads();
    if (t < NC * 8) {
      const int col = t % NC, prt = t / NC;
      const float* r = red + col * (TPB + 1) + prt * (TPB / 8);
      float a = 0.f, b = 0.f;
#pragma unroll 8
      for (int i = 0; i < TPB / 8; i++) { const float v = r[i]; a += v; b = fmaf(v, v, b); }
      seg[col * 8 + prt] = a;
      seg[(NC + col) * 8 + prt] = b;
    }
    __syncthreads();
    if (t < NC) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int p = 0; p < 8; p++) { a += (double)seg[t * 8 + p]; b += (double)seg[(NC + t) * 8 + p]; }
      const int k = t / 4, e = t % 4;
      acc[(slot_base + k) * CQ + 4 * q + e] += a;
      acc[(NSTAT + slot_base + k) * CQ + 4 * q + e] += b;
    }
  }
}

template <int CQ, typename DT>
__device__ __forceinline__ void store_row(DT* __restrict__ dst, const float* __restrict__ v) {
#pragma unroll
  for (int q = 0; q < CQ / 4; q++) stp4(dst + 4 * q, f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
}
template <int NS, int CQ>
__device__ __forceinline__ void round_rows(float (*u)[CQ]) {       // bf16 build: the values that are stored are the values that are counted
#ifdef EDGE_PRE_BF16
#pragma unroll
  for (int k = 0; k < NS; k++)
#pragma unroll
    for (int c = 0; c < CQ; c++) u[k][c] = rndp(u[k][c]);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// edge_fwd_a: slots 0..6 from one x1 halo tile.  Tile = TH x TW output pixels (TH*TW threads), persistent over tiles.
// Input tile origin (S*ty0 - 4, S*tx0 - 4): halo 4 = the reach of dil_conv_5x5 (padding 4, dilation 2).
// ------------------------------------------------------------------------------------------------------------------
template <int CQ, int S, int TH, int TW>
__global__ __launch_bounds__(TH * TW) void edge_fwd_a_kernel(const float* __restrict__ x /* x1 rows */, long ldx, int B, int H, int W, EdgeW wts,
                                                             pre_t* __restrict__ pre /* [9][P][CQ] */,
                                                             double* __restrict__ part /* [grid][7*CQ][2] */) {
  constexpr int TPB = TH * TW, NQ = CQ / 4, C = 4 * CQ;
  constexpr int IH = S * (TH - 1) + 9, IW = S * (TW - 1) + 9, NPOS = IH * IW;
  extern __shared__ __align__(16) unsigned char smem[];
  f32x4* tile = (f32x4*)smem;                                   // [NQ][NPOS]
  float* wl = (float*)(tile + NQ * NPOS);                      // weights
  float* w_dw3a = wl, *w_dw5a = w_dw3a + CQ * 9, *w_dd3 = w_dw5a + CQ * 25, *w_dd5 = w_dd3 + CQ * 9;
  float* w_pw = w_dd5 + CQ * 25;                               // [4][CQ*CQ]: sep3a sep5a dil3 dil5
  float* w_fr = w_pw + 4 * CQ * CQ;                            // [2][CQ/2*CQ]
  float* red = w_fr + CQ * CQ;                                 // [12][TPB+1]
  float* seg = red + 12 * (TPB + 1);                           // [2][12][8]
  double* acc = (double*)(seg + 2 * 12 * 8);                   // [2][7*CQ] (an even number of floats precedes it)
  constexpr bool TAPS = CQ == 4;
  if (TAPS) {
    stage_weights_taps(w_dw3a, wts.dw[0], CQ, 9);
    stage_weights_taps(w_dw5a, wts.dw[1], CQ, 25);
    stage_weights_taps(w_dd3, wts.dw[2], CQ, 9);
    stage_weights_taps(w_dd5, wts.dw[3], CQ, 25);
  } else {
    stage_weights(w_dw3a, wts.dw[0], CQ * 9);
    stage_weights(w_dw5a, wts.dw[1], CQ * 25);
    stage_weights(w_dd3, wts.dw[2], CQ * 9);
    stage_weights(w_dd5, wts.dw[3], CQ * 25);
  }
  for (int k = 0; k < 4; k++) stage_weights(w_pw + k * CQ * CQ, wts.pw[k], CQ * CQ);
  if (S == 2) { stage_weights(w_fr, wts.fr[0], CQ / 2 * CQ); stage_weights(w_fr + CQ / 2 * CQ, wts.fr[1], CQ / 2 * CQ); }
  for (int i = threadIdx.x; i < 2 * 7 * CQ; i += TPB) acc[i] = 0.0;
  const int Ho = H / S, Wo = W / S;
  const int tiles_y = (Ho + TH - 1) / TH, tiles_x = (Wo + TW - 1) / TW;
  const long ntiles = (long)B * tiles_y * tiles_x;
  const long P = (long)B * Ho * Wo;
  const int ly = threadIdx.x / TW, lx = threadIdx.x % TW;
  for (long tidx = blockIdx.x; tidx < ntiles; tidx += gridDim.x) {
    const int n = (int)(tidx / (tiles_y * tiles_x));
    const int tr = (int)(tidx % (tiles_y * tiles_x));
    const int ty0 = (tr / tiles_x) * TH, tx0 = (tr % tiles_x) * TW;
    __syncthreads();                                           // previous tile's readers are done (also orders the weight staging)
    load_tile<NQ, 0, IH, IW, TPB>(tile, x, ldx, n, H, W, S * ty0 - 4, S * tx0 - 4, nullptr, nullptr);
    __syncthreads();
    const int oy = ty0 + ly, ox = tx0 + lx;
    const bool live = oy < Ho && ox < Wo;
    const int cy = S * ly + 4, cx = S * lx + 4;                  // tile position of input pixel (S*oy, S*ox)
    const long qo = ((long)n * Ho + oy) * Wo + ox;
    // The three groups of branches are computed, stored and reduced one after the other so that only one group's values are
    // live in registers at a time.
    {   // ---- pools (raw x1; validity by image coordinates) and, on a stride-2 edge, FactorizedReduce ----
      float u[3][CQ];
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sm = {0.f, 0.f, 0.f, 0.f};
        int cnt = 0;
#pragma unroll
        for (int j = -1; j <= 1; j++)
#pragma unroll
          for (int i = -1; i <= 1; i++) {
            const int iy = S * oy + j, ix = S * ox + i;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
              const f32x4 v = tile[q * NPOS + (cy + j) * IW + cx + i];
#pragma unroll
              for (int e = 0; e < 4; e++) { mx[e] = fmaxf(mx[e], v[e]); sm[e] += v[e]; }
              cnt++;
            }
          }
        const float inv = 1.f / (float)max(cnt, 1);
#pragma unroll
        for (int e = 0; e < 4; e++) { u[0][4 * q + e] = live ? mx[e] : 0.f; u[1][4 * q + e] = sm[e] * inv; }
      }
      if (S == 2) {
        float r0[CQ], r1[CQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const f32x4 a = relu4(tile[q * NPOS + cy * IW + cx]), b = relu4(tile[q * NPOS + (cy + 1) * IW + cx + 1]);
#pragma unroll
          for (int e = 0; e < 4; e++) { r0[4 * q + e] = a[e]; r1[4 * q + e] = b[e]; }
        }
#ifndef EDGE_FWD_SCALARW   // measured: scalar-loaded weights lose 7-15 % in the forward kernels at Cq = 4, 8 (tools/ab_edge.py), -6 % only on the stride-2 Cq = 16 edge
        pw_at<CQ / 2, CQ>(w_fr, r0, &u[2][0]);
        pw_at<CQ / 2, CQ>(w_fr + CQ / 2 * CQ, r1, &u[2][CQ / 2]);
#else
        pw_at<CQ / 2, CQ>(scalar_ptr(wts.fr[0]), r0, &u[2][0]);
        pw_at<CQ / 2, CQ>(scalar_ptr(wts.fr[1]), r1, &u[2][CQ / 2]);
#endif
      } else {
#pragma unroll
        for (int c = 0; c < CQ; c++) u[2][c] = 0.f;
      }
      round_rows<3, CQ>(u);
      if (live) {
        store_row<CQ>(pre + (0 * P + qo) * CQ, u[0]);
        store_row<CQ>(pre + (1 * P + qo) * CQ, u[1]);
        if (S == 2) store_row<CQ>(pre + (2 * P + qo) * CQ, u[2]);
      }
      phase_stats<CQ, 3, 7, TPB>(u, live, red, seg, acc, 0);
    }
    {   // ---- first stages of sep_conv_3x3 / sep_conv_5x5: relu -> depthwise (stride S) -> 1x1 ----
      float u[2][CQ], d[CQ];
#ifndef EDGE_FWD_SCALARW   // measured: scalar-loaded weights lose 7-15 % in the forward kernels at Cq = 4, 8 (tools/ab_edge.py), -6 % only on the stride-2 Cq = 16 edge
      dw_at<CQ, 3, 1, true, TAPS>(tile, NPOS, IW, cy - 1, cx - 1, w_dw3a, d);
      pw_at<CQ, CQ>(w_pw, d, u[0]);
      dw_at<CQ, 5, 1, true, TAPS>(tile, NPOS, IW, cy - 2, cx - 2, w_dw5a, d);
      pw_at<CQ, CQ>(w_pw + CQ * CQ, d, u[1]);
#else
      dw_at<CQ, 3, 1, true, false>(tile, NPOS, IW, cy - 1, cx - 1, scalar_ptr(wts.dw[0]), d);
      pw_at<CQ, CQ>(scalar_ptr(wts.pw[0]), d, u[0]);
      dw_at<CQ, 5, 1, true, false>(tile, NPOS, IW, cy - 2, cx - 2, scalar_ptr(wts.dw[1]), d);
      pw_at<CQ, CQ>(scalar_ptr(wts.pw[1]), d, u[1]);
#endif
      round_rows<2, CQ>(u);
      if (live) { store_row<CQ>(pre + (3 * P + qo) * CQ, u[0]); store_row<CQ>(pre + (4 * P + qo) * CQ, u[1]); }
      phase_stats<CQ, 2, 7, TPB>(u, live, red, seg, acc, 3);
    }
    {   // ---- dil_conv_3x3 / dil_conv_5x5: relu -> depthwise (stride S, dilation 2) -> 1x1 ----
      float u[2][CQ], d[CQ];
#ifndef EDGE_FWD_SCALARW   // measured: scalar-loaded weights lose 7-15 % in the forward kernels at Cq = 4, 8 (tools/ab_edge.py), -6 % only on the stride-2 Cq = 16 edge
      dw_at<CQ, 3, 2, true, TAPS>(tile, NPOS, IW, cy - 2, cx - 2, w_dd3, d);
      pw_at<CQ, CQ>(w_pw + 2 * CQ * CQ, d, u[0]);
      dw_at<CQ, 5, 2, true, TAPS>(tile, NPOS, IW, cy - 4, cx - 4, w_dd5, d);
      pw_at<CQ, CQ>(w_pw + 3 * CQ * CQ, d, u[1]);
#else
      dw_at<CQ, 3, 2, true, false>(tile, NPOS, IW, cy - 2, cx - 2, scalar_ptr(wts.dw[2]), d);
      pw_at<CQ, CQ>(scalar_ptr(wts.pw[2]), d, u[0]);
      dw_at<CQ, 5, 2, true, false>(tile, NPOS, IW, cy - 4, cx - 4, scalar_ptr(wts.dw[3]), d);
      pw_at<CQ, CQ>(scalar_ptr(wts.pw[3]), d, u[1]);
#endif
      round_rows<2, CQ>(u);
      if (live) { store_row<CQ>(pre + (5 * P + qo) * CQ, u[0]); store_row<CQ>(pre + (6 * P + qo) * CQ, u[1]); }
      phase_stats<CQ, 2, 7, TPB>(u, live, red, seg, acc, 5);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 7 * CQ; i += TPB) {
    part[((long)blockIdx.x * 7 * CQ + i) * 2 + 0] = acc[i];
    part[((long)blockIdx.x * 7 * CQ + i) * 2 + 1] = acc[7 * CQ + i];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// edge_fwd_a2 / edge_fwd_b2 (Cq = 8, 16): the same results as edge_fwd_a / edge_fwd_b, BIT FOR BIT, in the lane layouts of
// edge_bwd_conv2 (thread = pixel with all channels ran at 2.5 - 5x the bytes' time at these widths: every weight an LDS broadcast):
//   * the tiles are stored in "strided quads": plane q holds channels q, q + NQ, q + 2 NQ, q + 3 NQ of a position;
//   * depthwise phase: a wave owns one strided quad, a lane one pixel; the quad's weights are scalar loads from a tap-major copy
//     (edge_fwd_prep_kernel), the per-channel tap order is the first version's;
//   * 1x1 phase: lane = (pixel, strided quad) -- the B operand of v_mfma_f32_16x16x4_f32; instruction m consumes element m of every
//     lane, i.e. input channels NQ*m .. NQ*m + NQ - 1 of its pixel block: with the k order of the instruction that is the first
//     version's sequential chain over the input channels (zero A entries of other position blocks add exact zeros), and the result
//     rows are the natural output quads;
//   * BatchNorm partial sums: the values go through LDS as [pixel][channel] and are summed in exactly phase_stats' order (8 pixel
//     segments per channel sequentially in fp32, the segments in fp64).
// ------------------------------------------------------------------------------------------------------------------
// wq[off(conv) + (tap * NQ + q) * 4 + e] = dw[conv][(q + NQ * e) * K*K + tap]   (conv = 0..5, off = cumulative taps * CQ)
__global__ __launch_bounds__(256) void edge_fwd_prep_kernel(EdgeW wts, int CQ, float* __restrict__ wq) {
  const int taps[6] = {9, 25, 9, 25, 9, 25};
  const int NQ = CQ / 4;
  int off = 0;
  for (int i = 0; i < 6; i++) {
    const int kk = taps[i];
    for (int idx = threadIdx.x; idx < kk * CQ; idx += 256) {
      const int e = idx % 4, q = (idx / 4) % NQ, tap = idx / CQ;
      wq[off + idx] = wts.dw[i][(q + NQ * e) * kk + tap];
    }
    off += kk * CQ;
  }
}

// one activated-on-read depthwise K x K at a pixel for one strided quad (weights: scalar loads); tap order = dw_at's
template <int K, int DIL, bool RELU, int NQ>
__device__ __forceinline__ f32x4 dw_quad(const f32x4* __restrict__ tp /* plane + top-left tap */, int tw, const float* __restrict__ wq_conv, int q) {
  unsigned long wa = (unsigned long)(uintptr_t)(wq_conv + 4 * q);
  asm volatile("" : "+s"(wa));
  const const_f32x4_ptr wp = (const_f32x4_ptr)wa;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < K; j++)
#pragma unroll
    for (int i = 0; i < K; i++) {
      f32x4 v = tp[(DIL * j) * tw + DIL * i];
      if (RELU) v = relu4(v);
      const f32x4 w4 = wp[(j * K + i) * NQ];
#pragma unroll
      for (int e = 0; e < 4; e++) a[e] = fmaf(w4[e], v[e], a[e]);
    }
  return a;
}

// phase_stats' arithmetic on values staged as Us[k][pixel][channel] (zeros for pixels that are not live): NS slots starting at slot_base
template <int CQ, int NPX, int TPB>
__device__ __forceinline__ void stats_from_lds(const float* __restrict__ Us, int NS, float* __restrict__ seg, double* __restrict__ acc, int slot_base,
                                               int NSTAT) {
  const int NC = NS * CQ;
  for (int idx = threadIdx.x; idx < NC * 8; idx += TPB) {
    const int col = idx % NC, prt = idx / NC;
    const float* r = Us + (long)(col / CQ) * NPX * CQ + (prt * (NPX / 8)) * CQ + col % CQ;
    float a = 0.f, b = 0.f;
#pragma unroll 8
    for (int i = 0; i < NPX / 8; i++) { const float v = r[i * CQ]; a += v; b = fmaf(v, v, b); }
    seg[col * 8 + prt] = a;
    seg[(NC + col) * 8 + prt] = b;
  }
  __syncthreads();
  for (int col = threadIdx.x; col < NC; col += TPB) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int p = 0; p < 8; p++) { a += (double)seg[col * 8 + p]; b += (double)seg[(NC + col) * 8 + p]; }
    acc[(slot_base + col / CQ) * CQ + col % CQ] += a;
    acc[(NSTAT + slot_base + col / CQ) * CQ + col % CQ] += b;
  }
}

template <int CQ, int S, int TH, int TW, int NW>
__global__ __launch_bounds__(64 * NW, 2) void edge_fwd_a2_kernel(const float* __restrict__ x /* x1 rows */, long ldx, int B, int H, int W, EdgeW wts,
                                                                 const float* __restrict__ wq, pre_t* __restrict__ pre /* [9][P][CQ] */,
                                                                 double* __restrict__ part /* [grid][7*CQ][2] */) {
  constexpr int TPB = 64 * NW, NPX = TH * TW, NQ = CQ / 4;
  constexpr int IH = S * (TH - 1) + 9, IW = S * (TW - 1) + 9, NPOS = IH * IW;
  constexpr int PPW = 64 / NQ, NITP = NPX / PPW, ITP = NITP / NW;          // 1x1 phase: items of PPW pixels
  constexpr int NITC = NPX / 64, ITC = NITC * NQ / NW;                      // depthwise phase: (64-pixel item, quad) pairs per wave
  static_assert(NW % NQ == 0 && (NITC * NQ) % NW == 0 && NITP % NW == 0 && NPX % 64 == 0 && (TW & (TW - 1)) == 0 && NPX % 8 == 0, "tile / wave split");
  extern __shared__ __align__(16) unsigned char smem[];
  f32x4* tile = (f32x4*)smem;                                   // [NQ][NPOS] strided quads
  float* Ds = (float*)(tile + NQ * NPOS);                      // [2][NPX][NQ][4]: depthwise outputs, strided quads
  float* Us = Ds + 2 * NPX * CQ;                               // [3][NPX][CQ]: pre-BN outputs, natural channel order (statistics, pooled stores)
  float* seg = Us + 3 * NPX * CQ;                              // [2][3*CQ][8]
  double* acc = (double*)(seg + 2 * 3 * CQ * 8);               // [2][7*CQ]
  for (int i = threadIdx.x; i < 2 * 7 * CQ; i += TPB) acc[i] = 0.0;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n16 = lane & 15, jg = lane >> 4;
  const int qB = jg % NQ, blkB = jg / NQ, slotB = n16 + 16 * blkB;
  // A operands (block diagonal): A_m[i = (block, oc)][k-slot jg] = W[oc][in = qB + NQ m]
  float aop[4][4], afr0[4], afr1[4];
  {
    const bool on = n16 / CQ == blkB;
    const int oc = n16 % CQ;
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int in = qB + NQ * m;
#pragma unroll
      for (int k = 0; k < 4; k++) aop[k][m] = on ? wts.pw[k][oc * CQ + in] : 0.f;
      afr0[m] = (S == 2 && on && oc < CQ / 2) ? wts.fr[0][oc * CQ + in] : 0.f;
      afr1[m] = (S == 2 && on && oc >= CQ / 2) ? wts.fr[1][(oc - CQ / 2) * CQ + in] : 0.f;
    }
  }
  const int qC = wave % NQ, itc0 = wave / NQ;
  const float* wq0 = wq, *wq1 = wq0 + 9 * CQ, *wq2 = wq1 + 25 * CQ, *wq3 = wq2 + 9 * CQ;
  const int Ho = H / S, Wo = W / S;
  const int tiles_y = (Ho + TH - 1) / TH, tiles_x = (Wo + TW - 1) / TW;
  const long ntiles = (long)B * tiles_y * tiles_x;
  const long P = (long)B * Ho * Wo;
  for (long tidx = blockIdx.x; tidx < ntiles; tidx += gridDim.x) {
    const int n = (int)(tidx / (tiles_y * tiles_x));
    const int tr = (int)(tidx % (tiles_y * tiles_x));
    const int ty0 = (tr / tiles_x) * TH, tx0 = (tr % tiles_x) * TW;
    __syncthreads();
    int tid = threadIdx.x, slot = slotB, ln = lane;
    asm volatile("" : "+v"(tid), "+v"(slot), "+v"(ln));
    {   // ---- x1 halo tile (origin (S*ty0 - 4, S*tx0 - 4)) into strided-quad planes ----
      constexpr int TOT = NPOS * NQ, ROUNDS = (TOT + TPB - 1) / TPB, BATCH = 4;
      float* tf = (float*)tile;
#pragma unroll
      for (int r0 = 0; r0 < ROUNDS; r0 += BATCH) {
        f32x4 v[BATCH];
#pragma unroll
        for (int r = 0; r < BATCH; r++) {
          const int idx = tid + (r0 + r) * TPB;
          const int q = idx % NQ, pos = idx / NQ;
          const int ly = pos / IW, lx = pos - ly * IW;
          const int y = S * ty0 - 4 + ly, xx = S * tx0 - 4 + lx;
          v[r] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (r0 + r < ROUNDS && idx < TOT && y >= 0 && y < H && xx >= 0 && xx < W) v[r] = *(const f32x4*)(x + (((long)n * H + y) * W + xx) * ldx + 4 * q);
        }
#pragma unroll
        for (int r = 0; r < BATCH; r++) {
          const int idx = tid + (r0 + r) * TPB;
          if (r0 + r < ROUNDS && idx < TOT) {
            const int q = idx % NQ, pos = idx / NQ;
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const int c = 4 * q + e;                           // channel c -> plane c % NQ, element c / NQ
              tf[((c % NQ) * NPOS + pos) * 4 + c / NQ] = v[r][e];
            }
          }
        }
      }
    }
    __syncthreads();
    // ---- depthwise phase 1: pools (raw x1) and the first stages of sep_conv_3x3 / sep_conv_5x5 ----
#pragma unroll 1
    for (int t = 0; t < ITC; t++) {
      const int p = (itc0 + t * (NW / NQ)) * 64 + ln;
      const int ly = p / TW, lx = p % TW;
      const int oy = ty0 + ly, ox = tx0 + lx;
      const bool live = oy < Ho && ox < Wo;
      const f32x4* tp = tile + qC * NPOS + (S * ly + 4) * IW + S * lx + 4;   // input pixel (S*oy, S*ox)
      f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sm = {0.f, 0.f, 0.f, 0.f};
      int cnt = 0;
#pragma unroll
      for (int j = -1; j <= 1; j++)
#pragma unroll
        for (int i = -1; i <= 1; i++) {
          const int iy = S * oy + j, ix = S * ox + i;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const f32x4 v = tp[j * IW + i];
#pragma unroll
            for (int e = 0; e < 4; e++) { mx[e] = fmaxf(mx[e], v[e]); sm[e] += v[e]; }
            cnt++;
          }
        }
      const float inv = 1.f / (float)max(cnt, 1);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        Us[p * CQ + qC + NQ * e] = live ? rndp(mx[e]) : 0.f;
        Us[NPX * CQ + p * CQ + qC + NQ * e] = live ? rndp(sm[e] * inv) : 0.f;     // (the first version leaves sm * inv here too: zero padding sums to 0)
      }
      *(f32x4*)(Ds + (p * NQ + qC) * 4) = dw_quad<3, 1, true, NQ>(tp - IW - 1, IW, wq0, qC);
      *(f32x4*)(Ds + NPX * CQ + (p * NQ + qC) * 4) = dw_quad<5, 1, true, NQ>(tp - 2 * IW - 2, IW, wq1, qC);
    }
    __syncthreads();
    // ---- 1x1 phase 1: FactorizedReduce (stride 2), sep3.a, sep5.a ----
    f32x4 ukeep[ITP][2];
#pragma unroll
    for (int t = 0; t < ITP; t++) {
      const int pc = (wave + t * NW) * PPW + slot;
      const int ly = pc / TW, lx = pc % TW;
      const int oy = ty0 + ly, ox = tx0 + lx;
      const bool live = oy < Ho && ox < Wo;
      const long qo = ((long)n * Ho + oy) * Wo + ox;
      f32x4 u2 = {0.f, 0.f, 0.f, 0.f};
      if (S == 2) {
        const f32x4* tp = tile + qB * NPOS + (S * ly + 4) * IW + S * lx + 4;
        const f32x4 r0 = relu4(tp[0]), r1 = relu4(tp[IW + 1]);
#pragma unroll
        for (int m = 0; m < 4; m++) u2 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr0[m], r0[m], u2, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; m++) u2 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr1[m], r1[m], u2, 0, 0, 0);
      }
      const f32x4 d3 = *(const f32x4*)(Ds + (pc * NQ + qB) * 4), d5 = *(const f32x4*)(Ds + NPX * CQ + (pc * NQ + qB) * 4);
      f32x4 u3 = {0.f, 0.f, 0.f, 0.f}, u4 = u3;
#pragma unroll
      for (int m = 0; m < 4; m++) u3 = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[0][m], d3[m], u3, 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 4; m++) u4 = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[1][m], d5[m], u4, 0, 0, 0);
      u2 = rndp4(u2); u3 = rndp4(u3); u4 = rndp4(u4);
      if (live) {
        if (S == 2) stp4(pre + (2 * P + qo) * CQ + 4 * qB, u2);
        stp4(pre + (3 * P + qo) * CQ + 4 * qB, u3);
        stp4(pre + (4 * P + qo) * CQ + 4 * qB, u4);
      }
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *(f32x4*)(Us + 2 * NPX * CQ + pc * CQ + 4 * qB) = live ? u2 : z;
      ukeep[t][0] = live ? u3 : z;
      ukeep[t][1] = live ? u4 : z;
    }
    __syncthreads();
    // pooled outputs to memory (whole rows), statistics of slots 0..2
    for (int idx = tid; idx < NPX * NQ; idx += TPB) {
      const int pc = idx / NQ, q = idx % NQ;
      const int oy = ty0 + pc / TW, ox = tx0 + pc % TW;
      if (oy < Ho && ox < Wo) {
        const long qo = ((long)n * Ho + oy) * Wo + ox;
        stp4(pre + (0 * P + qo) * CQ + 4 * q, *(const f32x4*)(Us + pc * CQ + 4 * q));
        stp4(pre + (1 * P + qo) * CQ + 4 * q, *(const f32x4*)(Us + NPX * CQ + pc * CQ + 4 * q));
      }
    }
    stats_from_lds<CQ, NPX, TPB>(Us, 3, seg, acc, 0, 7);
    __syncthreads();
    // statistics of slots 3, 4; meanwhile the dilated depthwise outputs (Ds is free: its readers passed the barrier above)
#pragma unroll
    for (int t = 0; t < ITP; t++) {
      const int pc = (wave + t * NW) * PPW + slot;
      *(f32x4*)(Us + pc * CQ + 4 * qB) = ukeep[t][0];
      *(f32x4*)(Us + NPX * CQ + pc * CQ + 4 * qB) = ukeep[t][1];
    }
#pragma unroll 1
    for (int t = 0; t < ITC; t++) {
      const int p = (itc0 + t * (NW / NQ)) * 64 + ln;
      const int ly = p / TW, lx = p % TW;
      const f32x4* tp = tile + qC * NPOS + (S * ly + 4) * IW + S * lx + 4;
      *(f32x4*)(Ds + (p * NQ + qC) * 4) = dw_quad<3, 2, true, NQ>(tp - 2 * IW - 2, IW, wq2, qC);
      *(f32x4*)(Ds + NPX * CQ + (p * NQ + qC) * 4) = dw_quad<5, 2, true, NQ>(tp - 4 * IW - 4, IW, wq3, qC);
    }
    __syncthreads();
    stats_from_lds<CQ, NPX, TPB>(Us, 2, seg, acc, 3, 7);
    // ---- 1x1 phase 2: dil3, dil5 (results stay in registers until the statistics above have read Us) ----
#pragma unroll
    for (int t = 0; t < ITP; t++) {
      const int pc = (wave + t * NW) * PPW + slot;
      const int oy = ty0 + pc / TW, ox = tx0 + pc % TW;
      const bool live = oy < Ho && ox < Wo;
      const long qo = ((long)n * Ho + oy) * Wo + ox;
      const f32x4 d3 = *(const f32x4*)(Ds + (pc * NQ + qB) * 4), d5 = *(const f32x4*)(Ds + NPX * CQ + (pc * NQ + qB) * 4);
      f32x4 u5 = {0.f, 0.f, 0.f, 0.f}, u6 = u5;
#pragma unroll
      for (int m = 0; m < 4; m++) u5 = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[2][m], d3[m], u5, 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 4; m++) u6 = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[3][m], d5[m], u6, 0, 0, 0);
      u5 = rndp4(u5); u6 = rndp4(u6);
      if (live) {
        stp4(pre + (5 * P + qo) * CQ + 4 * qB, u5);
        stp4(pre + (6 * P + qo) * CQ + 4 * qB, u6);
      }
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      ukeep[t][0] = live ? u5 : z;
      ukeep[t][1] = live ? u6 : z;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < ITP; t++) {
      const int pc = (wave + t * NW) * PPW + slot;
      *(f32x4*)(Us + pc * CQ + 4 * qB) = ukeep[t][0];
      *(f32x4*)(Us + NPX * CQ + pc * CQ + 4 * qB) = ukeep[t][1];
    }
    __syncthreads();
    stats_from_lds<CQ, NPX, TPB>(Us, 2, seg, acc, 5, 7);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 7 * CQ; i += TPB) {
    part[((long)blockIdx.x * 7 * CQ + i) * 2 + 0] = acc[i];
    part[((long)blockIdx.x * 7 * CQ + i) * 2 + 1] = acc[7 * CQ + i];
  }
}

template <int CQ, int TH, int TW, int NW>
__global__ __launch_bounds__(64 * NW, 2) void edge_fwd_b2_kernel(int B, int Ho, int Wo, EdgeW wts, const float* __restrict__ wq,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 pre_t* __restrict__ pre, double* __restrict__ part /* [grid][2*CQ][2] */) {
  constexpr int TPB = 64 * NW, NPX = TH * TW, NQ = CQ / 4;
  constexpr int IH = TH + 4, IW = TW + 4, NPOS = IH * IW;
  constexpr int PPW = 64 / NQ, NITP = NPX / PPW, ITP = NITP / NW, NITC = NPX / 64, ITC = NITC * NQ / NW;
  static_assert(NW % NQ == 0 && (NITC * NQ) % NW == 0 && NITP % NW == 0 && NPX % 64 == 0 && (TW & (TW - 1)) == 0 && NPX % 8 == 0, "tile / wave split");
  extern __shared__ __align__(16) unsigned char smem[];
  f32x4* tile3 = (f32x4*)smem;                                  // [NQ][NPOS] strided quads of relu(bn(t3))
  f32x4* tile5 = tile3 + NQ * NPOS;
  float* Ds = (float*)(tile5 + NQ * NPOS);                     // [2][NPX][NQ][4]
  float* Us = Ds + 2 * NPX * CQ;                               // [2][NPX][CQ]
  float* ms = Us + 2 * NPX * CQ;                               // mean/invstd of slots 3, 4: [2][2][CQ]
  float* seg = ms + 4 * CQ;                                    // [2][2*CQ][8]
  double* acc = (double*)(seg + 2 * 2 * CQ * 8);               // [2][2*CQ]
  for (int i = threadIdx.x; i < 2 * CQ; i += TPB) {
    ms[i] = mean[3 * CQ + i];
    ms[2 * CQ + i] = invstd[3 * CQ + i];
  }
  for (int i = threadIdx.x; i < 4 * CQ; i += TPB) acc[i] = 0.0;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n16 = lane & 15, jg = lane >> 4;
  const int qB = jg % NQ, blkB = jg / NQ, slotB = n16 + 16 * blkB;
  float aop[2][4];
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int k = 0; k < 2; k++) aop[k][m] = (n16 / CQ == blkB) ? wts.pw[4 + k][(n16 % CQ) * CQ + qB + NQ * m] : 0.f;
  const int qC = wave % NQ, itc0 = wave / NQ;
  const float* wq4 = wq + 68 * CQ, *wq5 = wq4 + 9 * CQ;
  const int tiles_y = (Ho + TH - 1) / TH, tiles_x = (Wo + TW - 1) / TW;
  const long ntiles = (long)B * tiles_y * tiles_x;
  const long P = (long)B * Ho * Wo;
  for (long tidx = blockIdx.x; tidx < ntiles; tidx += gridDim.x) {
    const int n = (int)(tidx / (tiles_y * tiles_x));
    const int tr = (int)(tidx % (tiles_y * tiles_x));
    const int ty0 = (tr / tiles_x) * TH, tx0 = (tr % tiles_x) * TW;
    __syncthreads();
    int tid = threadIdx.x, slot = slotB, ln = lane;
    asm volatile("" : "+v"(tid), "+v"(slot), "+v"(ln));
    {   // ---- halo tiles of t3 / t4 (origin (ty0 - 2, tx0 - 2)), normalised + ReLU'd on load, into strided-quad planes ----
      constexpr int TOT = NPOS * NQ, ROUNDS = (TOT + TPB - 1) / TPB;
#pragma unroll 2
      for (int r = 0; r < ROUNDS; r++) {
        const int idx = tid + r * TPB;
        if (idx < TOT) {
          const int q = idx % NQ, pos = idx / NQ;
          const int ly = pos / IW, lx = pos - ly * IW;
          const int y = ty0 - 2 + ly, xx = tx0 - 2 + lx;
          const bool ok = y >= 0 && y < Ho && xx >= 0 && xx < Wo;
          f32x4 v3 = {0.f, 0.f, 0.f, 0.f}, v5 = v3;
          if (ok) {
            const long qi = ((long)n * Ho + y) * Wo + xx;
            v3 = ldp4(pre + (3 * P + qi) * CQ + 4 * q);
            v5 = ldp4(pre + (4 * P + qi) * CQ + 4 * q);
          }
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int c = 4 * q + e;
            float a = v3[e], b = v5[e];
            if (ok) {
              a = fmaxf((a - ms[c]) * ms[2 * CQ + c], 0.f);
              b = fmaxf((b - ms[CQ + c]) * ms[3 * CQ + c], 0.f);
            }
            ((float*)tile3)[((c % NQ) * NPOS + pos) * 4 + c / NQ] = a;
            ((float*)tile5)[((c % NQ) * NPOS + pos) * 4 + c / NQ] = b;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < ITC; t++) {
      const int p = (itc0 + t * (NW / NQ)) * 64 + ln;
      const int ly = p / TW, lx = p % TW;
      *(f32x4*)(Ds + (p * NQ + qC) * 4) = dw_quad<3, 1, false, NQ>(tile3 + qC * NPOS + (ly + 1) * IW + lx + 1, IW, wq4, qC);
      *(f32x4*)(Ds + NPX * CQ + (p * NQ + qC) * 4) = dw_quad<5, 1, false, NQ>(tile5 + qC * NPOS + ly * IW + lx, IW, wq5, qC);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < ITP; t++) {
      const int pc = (wave + t * NW) * PPW + slot;
      const int oy = ty0 + pc / TW, ox = tx0 + pc % TW;
      const bool live = oy < Ho && ox < Wo;
      const long qo = ((long)n * Ho + oy) * Wo + ox;
      const f32x4 d3 = *(const f32x4*)(Ds + (pc * NQ + qB) * 4), d5 = *(const f32x4*)(Ds + NPX * CQ + (pc * NQ + qB) * 4);
      f32x4 u7 = {0.f, 0.f, 0.f, 0.f}, u8 = u7;
#pragma unroll
      for (int m = 0; m < 4; m++) u7 = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[0][m], d3[m], u7, 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 4; m++) u8 = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[1][m], d5[m], u8, 0, 0, 0);
      u7 = rndp4(u7); u8 = rndp4(u8);
      if (live) {
        stp4(pre + (7 * P + qo) * CQ + 4 * qB, u7);
        stp4(pre + (8 * P + qo) * CQ + 4 * qB, u8);
      }
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *(f32x4*)(Us + pc * CQ + 4 * qB) = live ? u7 : z;
      *(f32x4*)(Us + NPX * CQ + pc * CQ + 4 * qB) = live ? u8 : z;
    }
    __syncthreads();
    stats_from_lds<CQ, NPX, TPB>(Us, 2, seg, acc, 0, 2);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * CQ; i += TPB) {
    part[((long)blockIdx.x * 2 * CQ + i) * 2 + 0] = acc[i];
    part[((long)blockIdx.x * 2 * CQ + i) * 2 + 1] = acc[2 * CQ + i];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// edge_fwd_b: second stage of sep_conv_3x3 / sep_conv_5x5: bn_a -> relu (on load) -> depthwise (stride 1) -> 1x1
// ------------------------------------------------------------------------------------------------------------------
template <int CQ, int TH, int TW>
__global__ __launch_bounds__(TH * TW) void edge_fwd_b_kernel(int B, int Ho, int Wo, EdgeW wts, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, pre_t* __restrict__ pre,
                                                             double* __restrict__ part /* [grid][2*CQ][2] */) {
  constexpr int TPB = TH * TW, NQ = CQ / 4;
  constexpr int IH = TH + 4, IW = TW + 4, NPOS = IH * IW;
  extern __shared__ __align__(16) unsigned char smem[];
  f32x4* tile3 = (f32x4*)smem;
  f32x4* tile5 = tile3 + NQ * NPOS;
  float* w_dw3 = (float*)(tile5 + NQ * NPOS), *w_dw5 = w_dw3 + CQ * 9, *w_pw = w_dw5 + CQ * 25;   // pw [2][CQ*CQ]
  float* ms = w_pw + 2 * CQ * CQ;                               // mean/invstd of slots 3, 4: [2][2][CQ]
  float* red = ms + 4 * CQ;                                    // [8][TPB+1]
  float* seg = red + 8 * (TPB + 1);                            // [2][8][8]
  double* acc = (double*)(seg + 128);                          // [2][2*CQ]
  constexpr bool TAPS = CQ == 4;
  if (TAPS) {
    stage_weights_taps(w_dw3, wts.dw[4], CQ, 9);
    stage_weights_taps(w_dw5, wts.dw[5], CQ, 25);
  } else {
    stage_weights(w_dw3, wts.dw[4], CQ * 9);
    stage_weights(w_dw5, wts.dw[5], CQ * 25);
  }
  stage_weights(w_pw, wts.pw[4], CQ * CQ);
  stage_weights(w_pw + CQ * CQ, wts.pw[5], CQ * CQ);
  for (int i = threadIdx.x; i < 2 * CQ; i += TPB) {
    ms[i] = mean[3 * CQ + i];                                  // slots 3 and 4 are consecutive
    ms[2 * CQ + i] = invstd[3 * CQ + i];
  }
  for (int i = threadIdx.x; i < 4 * CQ; i += TPB) acc[i] = 0.0;
  const int tiles_y = (Ho + TH - 1) / TH, tiles_x = (Wo + TW - 1) / TW;
  const long ntiles = (long)B * tiles_y * tiles_x;
  const long P = (long)B * Ho * Wo;
  const int ly = threadIdx.x / TW, lx = threadIdx.x % TW;
  for (long tidx = blockIdx.x; tidx < ntiles; tidx += gridDim.x) {
    const int n = (int)(tidx / (tiles_y * tiles_x));
    const int tr = (int)(tidx % (tiles_y * tiles_x));
    const int ty0 = (tr / tiles_x) * TH, tx0 = (tr % tiles_x) * TW;
    __syncthreads();
    load_tile<NQ, 2, IH, IW, TPB>(tile3, pre + 3 * P * CQ, CQ, n, Ho, Wo, ty0 - 2, tx0 - 2, ms, ms + 2 * CQ);
    load_tile<NQ, 2, IH, IW, TPB>(tile5, pre + 4 * P * CQ, CQ, n, Ho, Wo, ty0 - 2, tx0 - 2, ms + CQ, ms + 3 * CQ);
    __syncthreads();
    const int oy = ty0 + ly, ox = tx0 + lx;
    const bool live = oy < Ho && ox < Wo;
    float u[2][CQ], d[CQ];
#ifndef EDGE_FWD_SCALARW   // measured: scalar-loaded weights lose 7-15 % in the forward kernels at Cq = 4, 8 (tools/ab_edge.py), -6 % only on the stride-2 Cq = 16 edge
    dw_at<CQ, 3, 1, false, TAPS>(tile3, NPOS, IW, ly + 1, lx + 1, w_dw3, d);
    pw_at<CQ, CQ>(w_pw, d, u[0]);
    dw_at<CQ, 5, 1, false, TAPS>(tile5, NPOS, IW, ly, lx, w_dw5, d);
    pw_at<CQ, CQ>(w_pw + CQ * CQ, d, u[1]);
#else
    // every lane needs the same weights: scalar loads from the parameters themselves (same arithmetic, same order; no LDS reads for them)
    dw_at<CQ, 3, 1, false, false>(tile3, NPOS, IW, ly + 1, lx + 1, scalar_ptr(wts.dw[4]), d);
    pw_at<CQ, CQ>(scalar_ptr(wts.pw[4]), d, u[0]);
    dw_at<CQ, 5, 1, false, false>(tile5, NPOS, IW, ly, lx, scalar_ptr(wts.dw[5]), d);
    pw_at<CQ, CQ>(scalar_ptr(wts.pw[5]), d, u[1]);
#endif
    const long qo = ((long)n * Ho + oy) * Wo + ox;
    round_rows<2, CQ>(u);
    if (live) {
#pragma unroll
      for (int k = 0; k < 2; k++)
#pragma unroll
        for (int q = 0; q < NQ; q++)
          stp4(pre + ((long)(7 + k) * P + qo) * CQ + 4 * q, f32x4{u[k][4 * q], u[k][4 * q + 1], u[k][4 * q + 2], u[k][4 * q + 3]});
    }
    phase_stats<CQ, 2, 2, TPB>(u, live, red, seg, acc, 0);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * CQ; i += TPB) {
    part[((long)blockIdx.x * 2 * CQ + i) * 2 + 0] = acc[i];
    part[((long)blockIdx.x * 2 * CQ + i) * 2 + 1] = acc[2 * CQ + i];
  }
}

// mean / invstd of `nslot` consecutive slots starting at slot0 from the per-workgroup fp64 partials (fixed order), plus the
// running-statistics update of nn.BatchNorm2d (momentum m, unbiased variance) and its batch counter.  In eval mode
// (training = 0) the statistics are the running ones and nothing is updated.  One wave per (slot, channel).
__global__ __launch_bounds__(64) void edge_stats_finish_kernel(const double* __restrict__ part, int nblk, int nslot, int slot0, int CQ,
                                                               long P, float eps, float momentum, int training, EdgeBN bn,
                                                               int skip_slot, float* __restrict__ mean, float* __restrict__ invstd) {
  const int col = blockIdx.x;                                  // local slot * CQ + c
  const int slot = slot0 + col / CQ, c = col % CQ;
  if (slot == skip_slot) {                                     // identity skip: no BatchNorm
    if (threadIdx.x == 0) { mean[slot * CQ + c] = 0.f; invstd[slot * CQ + c] = 1.f; }
    return;
  }
  if (!training) {
    if (threadIdx.x == 0) {
      mean[slot * CQ + c] = bn.run_mean[slot][c];
      invstd[slot * CQ + c] = 1.f / sqrtf(bn.run_var[slot][c] + eps);
    }
    return;
  }
  double a = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 64) { a += part[((long)k * nslot * CQ + col) * 2]; b += part[((long)k * nslot * CQ + col) * 2 + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
  if (threadIdx.x != 0) return;
  const double m = a / (double)P;
  double var = b / (double)P - m * m;
  if (var < 0.0) var = 0.0;
  mean[slot * CQ + c] = (float)m;
  invstd[slot * CQ + c] = (float)(1.0 / sqrt(var + (double)eps));
  if (bn.run_mean[slot]) {
    const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
    bn.run_mean[slot][c] = (float)((1.0 - momentum) * bn.run_mean[slot][c] + momentum * m);
    bn.run_var[slot][c] = (float)((1.0 - momentum) * bn.run_var[slot][c] + momentum * unb);
  }
  if (c == 0 && bn.tracked[slot]) bn.tracked[slot][0] += 1;
}

// ------------------------------------------------------------------------------------------------------------------
// edge_fwd_c: out[q][4i] (+)= beta * sum_k w_k bn_k(u_k[q][i]);  out[q][4i+g] (+)= beta * bypass[q][g*Cq+i]
// thread = (output pixel, channel quad i..i+3) -> the 16 contiguous output channels 4i .. 4i+15
// ------------------------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void edge_fwd_c_kernel(const float* __restrict__ x, const pre_t* __restrict__ pre, const float* __restrict__ mean,
                                  const float* __restrict__ invstd, const float* __restrict__ wsel /* [7] */,
                                  const float* __restrict__ beta, int B, int Ho, int Wo, int CQ, float* __restrict__ out, long ldo,
                                  int accumulate) {
  const int NQ = CQ / 4, C = 4 * CQ;
  const long P = (long)B * Ho * Wo;
  const float bt = beta[0];
  float wk[7];
#pragma unroll
  for (int k = 0; k < 7; k++) wk[k] = wsel[k];
  const int slot_of[7] = {0, 1, 2, 7, 8, 5, 6};                // live branch k -> slot of its final pre-BN tensor
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < P * NQ; o += (long)gridDim.x * blockDim.x) {
    const int q4 = (int)(o % NQ);
    const long q = o / NQ;
    f32x4 mix = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 7; k++) {
      if (S == 1 && k == 2) continue;
      const int sl = slot_of[k];
      const f32x4 v = ldp4(pre + ((long)sl * P + q) * CQ + 4 * q4);
      const f32x4 m = *(const f32x4*)(mean + sl * CQ + 4 * q4), is = *(const f32x4*)(invstd + sl * CQ + 4 * q4);
#pragma unroll
      for (int e = 0; e < 4; e++) mix[e] = fmaf(wk[k], (v[e] - m[e]) * is[e], mix[e]);
    }
    f32x4 byp[3];
    if (S == 1) {
      const float* xp = x + q * C + 4 * q4;
      const f32x4 x1v = *(const f32x4*)xp;
#pragma unroll
      for (int e = 0; e < 4; e++) mix[e] = fmaf(wk[2], x1v[e], mix[e]);
#pragma unroll
      for (int g = 1; g < 4; g++) byp[g - 1] = *(const f32x4*)(xp + g * CQ);
    } else {
      const int ox = (int)(q % Wo), oy = (int)((q / Wo) % Ho), n = (int)(q / ((long)Wo * Ho));
      const int W = 2 * Wo, H = 2 * Ho;
      const float* b0 = x + (((long)n * H + 2 * oy) * W + 2 * ox) * C + 4 * q4;
      const long rs = (long)W * C;
#pragma unroll
      for (int g = 1; g < 4; g++) {
        const f32x4 a = *(const f32x4*)(b0 + g * CQ), b = *(const f32x4*)(b0 + C + g * CQ);
        const f32x4 c = *(const f32x4*)(b0 + rs + g * CQ), d = *(const f32x4*)(b0 + rs + C + g * CQ);
#pragma unroll
        for (int e = 0; e < 4; e++) byp[g - 1][e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
      }
    }
    float* dst = out + q * ldo + 16 * q4;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      f32x4 v = {mix[e] * bt, byp[0][e] * bt, byp[1][e] * bt, byp[2][e] * bt};
      if (accumulate) {
        const f32x4 old = *(const f32x4*)(dst + 4 * e);
        v += old;
      }
      *(f32x4*)(dst + 4 * e) = v;
    }
  }
}


// ==================================================================================================================
// Backward.  dOut = gradient of the node's running sum (row stride ldd).  With a_k = beta * w_k the gradient entering
// BatchNorm k is a_k * dY (dY[q][i] = dOut[q][4i]) for EVERY branch, so one pass over the pixels yields all the sums its
// backward needs -- S0[i] = sum_q dY, S_k[i] = sum_q dY * xhat_k -- and with them the architecture gradients
// dw_k = beta * sum_i S_k[i], dbeta = sum_k w_k sum_i S_k[i] + <dOut_bypass, bypass>.  Then
//   g_k = d/d(u_k) = A_k (dY - c1_k - xhat_k c2_k),  A_k = a_k invstd_k, c1_k = S0 / P, c2_k = S_k / P      (training)
// flows through the 1x1 / depthwise / pooling stencils of each branch back to x1.
//   edge_bwd_sums    dY compact, the sums above (per-workgroup partials)
//   edge_bwd_coef    finish: coefficients A, c1, c2 per slot; dw, dbeta
//   edge_bwd_pools   LAST: max / avg pool, skip (identity or FactorizedReduce) + the conv stages' dx1c + the bypass -> dx, whole rows
//   edge_bwd_conv    one (relu ->) depthwise -> 1x1 stage: dx1c (dense [B,H,W,Cq]) = / += (or the gradient entering a sep_conv's first BatchNorm
//                    + its sums), per-workgroup partial weight gradients; launched for dil3, dil5, sep3b, sep5b, then
//                    (after edge_bwd_coef_a) sep3a, sep5a
//   edge_wgrad_finish  fixed-order fp64 column sums of the partial weight gradients
// ==================================================================================================================
constexpr int COEF = 3 * 16;                                   // per slot: A[16], c1[16], c2[16]

template <int S>
__global__ __launch_bounds__(256) void edge_bwd_sums_kernel(const float* __restrict__ dout, long ldd, const float* __restrict__ x,
                                                            const pre_t* __restrict__ pre, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ beta,
                                                            int B, int Ho, int Wo, int CQ, float* __restrict__ dY,
                                                            float* __restrict__ part /* [grid][8*CQ+1] */,
                                                            float* __restrict__ x1d /* dense copy of x[..., :CQ] for the conv stages: [B,H,W,CQ] */) {
  __shared__ float red[4][33];
  const int NQ = CQ / 4, C = 4 * CQ;
  const long P = (long)B * Ho * Wo;
  const float bt = beta[0];
  const int slot_of[7] = {0, 1, 2, 7, 8, 5, 6};
  f32x4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float sb = 0.f;
  for (long o = blockIdx.x * 256L + threadIdx.x; o < P * NQ; o += gridDim.x * 256L) {
    const int q4 = (int)(o % NQ);
    const long q = o / NQ;
    f32x4 D[4];
#pragma unroll
    for (int e = 0; e < 4; e++) D[e] = *(const f32x4*)(dout + q * ldd + 16 * q4 + 4 * e);
    const f32x4 dy = {D[0][0], D[1][0], D[2][0], D[3][0]};
    *(f32x4*)(dY + q * CQ + 4 * q4) = dy;
    acc[7] += dy;
#pragma unroll
    for (int k = 0; k < 7; k++) {
      if (S == 1 && k == 2) continue;
      const int sl = slot_of[k];
      const f32x4 v = ldp4(pre + ((long)sl * P + q) * CQ + 4 * q4);
      const f32x4 m = *(const f32x4*)(mean + sl * CQ + 4 * q4), is = *(const f32x4*)(invstd + sl * CQ + 4 * q4);
#pragma unroll
      for (int e = 0; e < 4; e++) acc[k][e] = fmaf(dy[e], (v[e] - m[e]) * is[e], acc[k][e]);
    }
    if (S == 1) {
      const float* xp = x + q * C + 4 * q4;
      const f32x4 x1v = *(const f32x4*)xp;
      if (x1d) *(f32x4*)(x1d + q * CQ + 4 * q4) = x1v;
#pragma unroll
      for (int e = 0; e < 4; e++) acc[2][e] = fmaf(dy[e], x1v[e], acc[2][e]);
#pragma unroll
      for (int g = 1; g < 4; g++) {
        const f32x4 bv = *(const f32x4*)(xp + g * CQ);
#pragma unroll
        for (int e = 0; e < 4; e++) sb = fmaf(D[e][g], bv[e], sb);
      }
    } else {
      const int ox = (int)(q % Wo), oy = (int)((q / Wo) % Ho), n = (int)(q / ((long)Wo * Ho));
      const int W = 2 * Wo, H = 2 * Ho;
      const long base = (((long)n * H + 2 * oy) * W + 2 * ox) * C + 4 * q4;
      const long rs = (long)W * C;
      if (x1d) {   // the four x1 quads of the window (same rows as the bypass channels read below)
        const long pix = ((long)n * H + 2 * oy) * W + 2 * ox;
        *(f32x4*)(x1d + pix * CQ + 4 * q4) = *(const f32x4*)(x + base);
        *(f32x4*)(x1d + (pix + 1) * CQ + 4 * q4) = *(const f32x4*)(x + base + C);
        *(f32x4*)(x1d + (pix + W) * CQ + 4 * q4) = *(const f32x4*)(x + base + rs);
        *(f32x4*)(x1d + (pix + W + 1) * CQ + 4 * q4) = *(const f32x4*)(x + base + rs + C);
      }
#pragma unroll
      for (int g = 1; g < 4; g++) {
        const f32x4 a = *(const f32x4*)(x + base + g * CQ), b = *(const f32x4*)(x + base + C + g * CQ);
        const f32x4 c = *(const f32x4*)(x + base + rs + g * CQ), d = *(const f32x4*)(x + base + rs + C + g * CQ);
#pragma unroll
        for (int e = 0; e < 4; e++) sb = fmaf(D[e][g], fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e])), sb);
      }
    }
  }
  // lanes with equal (lane % NQ) hold the same channel quad: fold those lanes, then the four waves
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float vals[33];
#pragma unroll
  for (int k = 0; k < 8; k++)
#pragma unroll
    for (int e = 0; e < 4; e++) vals[k * 4 + e] = acc[k][e];
  vals[32] = sb;
#pragma unroll
  for (int i = 0; i < 33; i++) {
    float v = vals[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
      if (o >= NQ || i == 32) v += __shfl_xor(v, o, 64);
    vals[i] = v;
  }
  float* out = part + (long)blockIdx.x * (8 * CQ + 1);
  for (int qd = 0; qd < NQ; qd++) {
    __syncthreads();
    if (lane == qd) {
#pragma unroll
      for (int i = 0; i < 32; i++) red[wv][i] = vals[i];
      if (qd == 0) red[wv][32] = vals[32];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int k = threadIdx.x / 4, e = threadIdx.x % 4;
      out[k * CQ + 4 * qd + e] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
    if (qd == 0 && threadIdx.x == 32) out[8 * CQ] = red[0][32] + red[1][32] + red[2][32] + red[3][32];
  }
}

// one workgroup of 1024 threads: sums of the partials (fixed order, fp64; thread = (row group, column), so a row is read by
// neighbouring lanes) -> coefficients of the seven final slots, dw[7], dbeta
__global__ __launch_bounds__(1024) void edge_bwd_coef_kernel(const float* __restrict__ part, int nblk, int CQ, long P, int S, int training,
                                                             const float* __restrict__ wsel, const float* __restrict__ beta,
                                                             const float* __restrict__ invstd, float* __restrict__ coef,
                                                             float* __restrict__ dw, float* __restrict__ dbeta, EdgeW wts,
                                                             float* __restrict__ wtaps /* tap-major [K*K][CQ] copies of the six depthwise weights */) {
  {
    const int taps[6] = {9, 25, 9, 25, 9, 25};
    int off = 0;
    for (int i = 0; i < 6; i++) {
      const int kk = taps[i];
      for (int idx = threadIdx.x; idx < kk * CQ; idx += 1024) wtaps[off + (idx % kk) * CQ + idx / kk] = wts.dw[i][idx];
      off += kk * CQ;
    }
  }
  __shared__ double grp[7][8 * 16 + 1];
  __shared__ double tot[8 * 16 + 1];
  const int ncol = 8 * CQ + 1;
  const int ngrp = 1024 / ncol > 7 ? 7 : 1024 / ncol;            // 7 row groups at Cq = 4..16
  const int g = threadIdx.x / ncol, col = threadIdx.x % ncol;
  if (g < ngrp) {
    // sixteen independent loads in flight per thread (this single workgroup is latency-bound: up to 1536 partial rows / 7 groups)
    double a = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int b = g;
    for (; b + 15 * ngrp < nblk; b += 16 * ngrp) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = part[(long)(b + u * ngrp) * ncol + col];
#pragma unroll
      for (int u = 0; u < 16; u += 4) { a += (double)v[u]; a1 += (double)v[u + 1]; a2 += (double)v[u + 2]; a3 += (double)v[u + 3]; }
    }
    for (; b < nblk; b += ngrp) a += (double)part[(long)b * ncol + col];
    a = (a + a1) + (a2 + a3);
    grp[g][col] = a;
  }
  __syncthreads();
  if (threadIdx.x < ncol) {
    double a = 0.0;
    for (int k = 0; k < ngrp; k++) a += grp[k][threadIdx.x];
    tot[threadIdx.x] = a;
  }
  __syncthreads();
  const int slot_of[7] = {0, 1, 2, 7, 8, 5, 6};
  const float bt = beta[0];
  if (threadIdx.x < 7 * CQ) {
    const int k = threadIdx.x / CQ, c = threadIdx.x % CQ, sl = slot_of[k];
    const bool ident = (S == 1 && k == 2);
    float* cf = coef + sl * COEF;
    cf[c] = bt * wsel[k] * (ident ? 1.f : invstd[sl * CQ + c]);
    cf[16 + c] = (training && !ident) ? (float)(tot[7 * CQ + c] / (double)P) : 0.f;
    cf[32 + c] = (training && !ident) ? (float)(tot[k * CQ + c] / (double)P) : 0.f;
  }
  if (threadIdx.x == 0) {
    double db = tot[8 * CQ];
    for (int k = 0; k < 7; k++) {
      double sk = 0.0;
      for (int c = 0; c < CQ; c++) sk += tot[k * CQ + c];
      dw[k] = (float)(bt * sk);
      db += (double)wsel[k] * sk;
    }
    dbeta[0] = (float)db;
  }
}

// coefficients of a sep_conv's FIRST BatchNorm (slots 3, 4) from the partial sums its second stage's backward left:
// A = invstd, c1 = sum dBN / P, c2 = sum dBN * xhat / P
__global__ __launch_bounds__(64) void edge_bwd_coef_a_kernel(const double* __restrict__ part /* [2][G][2*CQ][2] */, int G, int nblk3, int nblk5,
                                                             int CQ, long P, int training, const float* __restrict__ invstd,
                                                             float* __restrict__ coef) {
  const int which = blockIdx.x / CQ, c = blockIdx.x % CQ;       // which: 0 sep3 (slot 3), 1 sep5 (slot 4)
  const double* pp = part + (long)which * G * 2 * CQ * 2;
  const int nblk = which ? nblk5 : nblk3;
  double a = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 64) { a += pp[((long)k * 2 * CQ + c) * 2]; b += pp[((long)k * 2 * CQ + CQ + c) * 2]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
  if (threadIdx.x != 0) return;
  float* cf = coef + (3 + which) * COEF;
  cf[c] = invstd[(3 + which) * CQ + c];
  cf[16 + c] = training ? (float)(a / (double)P) : 0.f;
  cf[32 + c] = training ? (float)(b / (double)P) : 0.f;
}

// g[oc] = A (dY - c1 - xhat c2) for one pixel, all CQ channels (rows of dY / u read from memory)
template <int CQ>
__device__ __forceinline__ void grad_row(const float* __restrict__ dyr, const pre_t* __restrict__ ur, const float* __restrict__ ms /* LDS: mean[CQ], invstd[CQ] */,
                                         const float* __restrict__ cf /* LDS: A[16], c1[16], c2[16] */, float* __restrict__ g) {
#pragma unroll
  for (int q = 0; q < CQ / 4; q++) {
    const f32x4 d = *(const f32x4*)(dyr + 4 * q), u = ldp4(ur + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int c = 4 * q + e;
      const float xh = (u[e] - ms[c]) * ms[CQ + c];
      g[c] = cf[c] * (d[e] - cf[16 + c] - xh * cf[32 + c]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// edge_bwd_pools: the LAST backward kernel of an edge; writes every channel of dx exactly once, in whole rows:
//   dx[..., :Cq]  = backward of max_pool (slot 0), avg_pool (slot 1), skip_connect (slot 2) + dx1c (what the conv stages
//                   accumulated, dense [B,H,W,Cq])
//   dx[..., Cq:]  = the bypass: beta * dOut[q][4i+g], routed to the first maximum of its 2x2 window on a reduction edge
// Tile = TH x TW INPUT pixels (one thread each, all Cq channels).  LDS: x1 tile with halo 2, and per covering OUTPUT pixel
// its max-pool gradient, its avg-pool gradient / window size, and the position (0..8) of its window's FIRST maximum per
// channel (computed once per output, not once per (input, output) pair).  FactorizedReduce weight gradients: partials.
// ------------------------------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int CQ, int S, int TH, int TW>
__global__ __launch_bounds__(TH * TW) void edge_bwd_pools_kernel(const float* __restrict__ x, const float* __restrict__ dY,
                                                                 const float* __restrict__ dout, long ldd, const float* __restrict__ beta,
                                                                 const pre_t* __restrict__ pre, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ coef,
                                                                 const float* __restrict__ dx1c, int B, int H, int W, EdgeW wts,
                                                                 float* __restrict__ dx, float* __restrict__ wpart, int wtot, int fr_off,
                                                                 const float* __restrict__ x1src, long ld1 /* where the x1 tile comes from */,
                                                                 int acc_dx /* dx += (a state with several consumers: one buffer, no add pass) */) {
  constexpr int TPB = TH * TW, NQ = CQ / 4, C = 4 * CQ;
  constexpr int XH = TH + 4, XW = TW + 4, XN = XH * XW;         // x1 tile, halo 2
  constexpr int GH = TH / S + 2, GW = TW / S + 2, GN = GH * GW;   // gradient tiles: outputs (ty0/S - 1) .. (ty0/S + TH/S)
  extern __shared__ __align__(16) unsigned char smem[];
  f32x4* xt = (f32x4*)smem;                                     // [NQ][XN]
  f32x4* g0t = xt + NQ * XN;                                    // [NQ][GN] max-pool gradient of the output
  f32x4* g1t = g0t + NQ * GN;                                   // [NQ][GN] avg-pool gradient / window size
  i32x4* apt = (i32x4*)(g1t + NQ * GN);                         // [NQ][GN] window position of the first maximum
  float* ms = (float*)(apt + NQ * GN);                         // mean/invstd of slots 0,1,2: [3][2][CQ]
  float* cf = ms + 6 * CQ;                                     // coef of slots 0,1,2: [3][48]
  float* w_fr = cf + 3 * COEF;                                 // [2][CQ/2][CQ]
  float* Gs = w_fr + CQ * CQ;                                  // S == 2: g2 rows of the tile's outputs [TPB/4][CQ]
  float* R0 = Gs + (TPB / 4) * CQ;                             //         relu(x1) at the even / odd taps [TPB/4][CQ] each
  float* R1 = R0 + (TPB / 4) * CQ;
  float* Xs = R1 + (TPB / 4) * CQ;                             // the tile's x1 gradients [TPB][CQ] on their way to the row writer
  const int Ho = H / S, Wo = W / S;
  const long P = (long)B * Ho * Wo;
  const float bt = beta[0];
  for (int i = threadIdx.x; i < 3 * CQ; i += TPB) {
    const int k = i / CQ, c = i % CQ;
    ms[k * 2 * CQ + c] = mean[k * CQ + c];
    ms[k * 2 * CQ + CQ + c] = invstd[k * CQ + c];
  }
  for (int i = threadIdx.x; i < 3 * COEF; i += TPB) cf[i] = coef[i];
  if (S == 2) { stage_weights(w_fr, wts.fr[0], CQ / 2 * CQ); stage_weights(w_fr + CQ / 2 * CQ, wts.fr[1], CQ / 2 * CQ); }
  float acc_fr = 0.f;                                          // thread (oc, c) of the two FactorizedReduce weight gradients
  const int tiles_y = (H + TH - 1) / TH, tiles_x = (W + TW - 1) / TW;
  const long ntiles = (long)B * tiles_y * tiles_x;
  const int ly = threadIdx.x / TW, lx = threadIdx.x % TW;
  for (long tidx = blockIdx.x; tidx < ntiles; tidx += gridDim.x) {
    const int n = (int)(tidx / (tiles_y * tiles_x));
    const int tr = (int)(tidx % (tiles_y * tiles_x));
    const int ty0 = (tr / tiles_x) * TH, tx0 = (tr % tiles_x) * TW;
    const int gy0 = ty0 / S - 1, gx0 = tx0 / S - 1;             // first output row / column of the gradient tiles
    __syncthreads();
    load_tile<NQ, 0, XH, XW, TPB>(xt, x1src, ld1, n, H, W, ty0 - 2, tx0 - 2, nullptr, nullptr);
    __syncthreads();
    for (int idx = threadIdx.x; idx < GN; idx += TPB) {
      const int gy = gy0 + idx / GW, gx = gx0 + idx % GW;
      const bool in = gy >= 0 && gy < Ho && gx >= 0 && gx < Wo;
      float a[CQ], b[CQ];
      float inv = 0.f;
      int wy0 = 0, wy1 = -1, wx0 = 0, wx1 = -1;
      if (in) {
        const long q = ((long)n * Ho + gy) * Wo + gx;
        grad_row<CQ>(dY + q * CQ, pre + (0 * P + q) * CQ, ms, cf, a);
        grad_row<CQ>(dY + q * CQ, pre + (1 * P + q) * CQ, ms + 2 * CQ, cf + COEF, b);
        wy0 = max(S * gy - 1, 0); wy1 = min(S * gy + 1, H - 1); wx0 = max(S * gx - 1, 0); wx1 = min(S * gx + 1, W - 1);
        inv = 1.f / (float)((wy1 - wy0 + 1) * (wx1 - wx0 + 1));
      } else {
#pragma unroll
        for (int c = 0; c < CQ; c++) { a[c] = 0.f; b[c] = 0.f; }
      }
#pragma unroll
      for (int q4 = 0; q4 < NQ; q4++) {
        // first maximum of the window in scan order (strict >), as a window position (ky * 3 + kx relative to S*g - 1)
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        i32x4 pos = {-1, -1, -1, -1};
        for (int yy = wy0; yy <= wy1; yy++)
          for (int xx = wx0; xx <= wx1; xx++) {
            const f32x4 v = xt[q4 * XN + (yy - ty0 + 2) * XW + (xx - tx0 + 2)];
            const int code = (yy - (S * gy - 1)) * 3 + (xx - (S * gx - 1));
#pragma unroll
            for (int e = 0; e < 4; e++)
              if (v[e] > best[e]) { best[e] = v[e]; pos[e] = code; }
          }
        g0t[q4 * GN + idx] = f32x4{a[4 * q4], a[4 * q4 + 1], a[4 * q4 + 2], a[4 * q4 + 3]};
        g1t[q4 * GN + idx] = f32x4{b[4 * q4] * inv, b[4 * q4 + 1] * inv, b[4 * q4 + 2] * inv, b[4 * q4 + 3] * inv};
        apt[q4 * GN + idx] = pos;
      }
    }
    __syncthreads();
    const int iy = ty0 + ly, ix = tx0 + lx;
    const bool live = iy < H && ix < W;
    float dxv[CQ];
    // ---- pools: every output window (oy, ox) that contains (iy, ix) ----
    const int oy_lo = max((iy - 1 + S - 1) / S, 0), oy_hi = min((iy + 1) / S, Ho - 1);    // ceil((iy-1)/S) .. floor((iy+1)/S)
    const int ox_lo = max((ix - 1 + S - 1) / S, 0), ox_hi = min((ix + 1) / S, Wo - 1);
#pragma unroll
    for (int q4 = 0; q4 < NQ; q4++) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      if (live) {
        for (int oy = oy_lo; oy <= oy_hi; oy++)
          for (int ox = ox_lo; ox <= ox_hi; ox++) {
            const int gi = (oy - gy0) * GW + (ox - gx0);
            const f32x4 gmax = g0t[q4 * GN + gi], gavg = g1t[q4 * GN + gi];
            const i32x4 pos = apt[q4 * GN + gi];
            const int code = (iy - (S * oy - 1)) * 3 + (ix - (S * ox - 1));
#pragma unroll
            for (int e = 0; e < 4; e++) a[e] += (pos[e] == code ? gmax[e] : 0.f) + gavg[e];
          }
        const f32x4 c1 = *(const f32x4*)(dx1c + (((long)n * H + iy) * W + ix) * CQ + 4 * q4);   // the conv stages' share
        a += c1;
      }
#pragma unroll
      for (int e = 0; e < 4; e++) dxv[4 * q4 + e] = a[e];
    }
    // ---- skip_connect ----
    if (S == 1) {
      if (live) {
        const long q = ((long)n * H + iy) * W + ix;
#pragma unroll
        for (int q4 = 0; q4 < NQ; q4++) {
          const f32x4 d = *(const f32x4*)(dY + q * CQ + 4 * q4);
#pragma unroll
          for (int e = 0; e < 4; e++) dxv[4 * q4 + e] = fmaf(cf[2 * COEF + 4 * q4 + e], d[e], dxv[4 * q4 + e]);
        }
      }
    } else {
      // FactorizedReduce: output (iy/2, ix/2) reads relu(x1) at (even, even) through conv_1 and at (odd, odd) through conv_2
      const bool even = !(iy & 1) && !(ix & 1), odd = (iy & 1) && (ix & 1);
      float g2[CQ];
      const int oy = iy >> 1, ox = ix >> 1;
      if (live && (even || odd)) {
        const long q = ((long)n * Ho + oy) * Wo + ox;
        grad_row<CQ>(dY + q * CQ, pre + (2 * P + q) * CQ, ms + 4 * CQ, cf + 2 * COEF, g2);
        const float* wf = w_fr + (odd ? CQ / 2 * CQ : 0);
        const int o0 = odd ? CQ / 2 : 0;
#pragma unroll
        for (int q4 = 0; q4 < NQ; q4++) {
          const f32x4 xp = xt[q4 * XN + (ly + 2) * XW + lx + 2];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int c = 4 * q4 + e;
            float a = 0.f;
#pragma unroll
            for (int oc = 0; oc < CQ / 2; oc++) a = fmaf(wf[oc * CQ + c], g2[o0 + oc], a);
            if (xp[e] > 0.f) dxv[c] += a;
          }
        }
      }
      // weight gradients: stage g2 and the two relu'd taps of every output of this tile, then thread (oc, c) sums over them
      __syncthreads();
      {
        const int lo = (ly >> 1) * (TW / 2) + (lx >> 1);          // local output index of this thread's 2x2 cell
        if (even) {
#pragma unroll
          for (int c = 0; c < CQ; c++) Gs[lo * CQ + c] = live ? g2[c] : 0.f;
        }
        if (even || odd) {
          float* R = even ? R0 : R1;
#pragma unroll
          for (int q4 = 0; q4 < NQ; q4++) {
            const f32x4 v = relu4(xt[q4 * XN + (ly + 2) * XW + lx + 2]);
#pragma unroll
            for (int e = 0; e < 4; e++) R[lo * CQ + 4 * q4 + e] = live ? v[e] : 0.f;
          }
        }
      }
      __syncthreads();
      if (threadIdx.x < CQ * CQ) {
        const int oc = threadIdx.x / CQ, c = threadIdx.x % CQ;
        const float* R = oc < CQ / 2 ? R0 : R1;
        float a = 0.f;
        for (int o = 0; o < TPB / 4; o++) a = fmaf(Gs[o * CQ + oc], R[o * CQ + c], a);
        acc_fr += a;
      }
    }
    // ---- write the tile's rows of dx: lane = (pixel, 16-byte chunk of its row), so that neighbouring lanes touch neighbouring chunks
    //      (one thread per whole row cost 64 cache lines per store instruction and made the accumulating form twice as slow) ----
    store_row<CQ>(Xs + threadIdx.x * CQ, dxv);                   // the x1 chunks go through LDS
    __syncthreads();
    constexpr int NCH = 4 * NQ;                                  // chunks of a row: NQ of x1, then the bypass groups g = 1..3
    if constexpr (S == 1) {
      // U items per thread at a time, every load of the batch requested before the first use (one item per iteration exposed a memory
      // round trip per iteration: 16 per tile at Cq = 16, most of the kernel's time)
      constexpr int UMAX = 8, U = NCH < UMAX ? NCH : UMAX;
      for (int it0 = threadIdx.x; it0 < TPB * NCH; it0 += U * TPB) {
        float dsc[U][4];
        f32x4 xw[S == 2 ? U : 1][4], old[U];
        bool ok[U];
  #pragma unroll
        for (int u = 0; u < U; u++) {
          const int it = it0 + u * TPB;
          const int pix = it / NCH, ch = it % NCH;
          const int g = ch / NQ, q4 = ch % NQ;
          const int jy = ty0 + pix / TW, jx = tx0 + pix % TW;
          ok[u] = it < TPB * NCH && jy < H && jx < W;
          old[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  #pragma unroll
          for (int e = 0; e < 4; e++) dsc[u][e] = 0.f;
          if (ok[u]) {
            if (acc_dx) old[u] = *(const f32x4*)(dx + (((long)n * H + jy) * W + jx) * C + g * CQ + 4 * q4);
            if (g > 0) {
              const long q = ((long)n * Ho + jy / S) * Wo + jx / S;
  #pragma unroll
              for (int e = 0; e < 4; e++) dsc[u][e] = dout[q * ldd + 16 * q4 + 4 * e + g];     // bypass channel g*Cq + i <- beta * dOut[q][4i + g]
              if (S == 2) {
                const float* xr = x + (((long)n * H + (jy & ~1)) * W + (jx & ~1)) * C + g * CQ + 4 * q4;   // top-left of the 2x2 window
                xw[S == 2 ? u : 0][0] = *(const f32x4*)xr; xw[S == 2 ? u : 0][1] = *(const f32x4*)(xr + C);
                xw[S == 2 ? u : 0][2] = *(const f32x4*)(xr + (long)W * C); xw[S == 2 ? u : 0][3] = *(const f32x4*)(xr + (long)W * C + C);
              }
            }
          }
        }
  #pragma unroll
        for (int u = 0; u < U; u++) {
          if (!ok[u]) continue;
          const int it = it0 + u * TPB;
          const int pix = it / NCH, ch = it % NCH;
          const int g = ch / NQ, q4 = ch % NQ;
          const int jy = ty0 + pix / TW, jx = tx0 + pix % TW;
          f32x4 v;
          if (g == 0) {
            v = *(const f32x4*)(Xs + pix * CQ + 4 * q4);
          } else {
  #pragma unroll
            for (int e = 0; e < 4; e++) v[e] = dsc[u][e] * bt;
            if (S == 2) {
              const int me = (jy & 1) * 2 + (jx & 1);
              const f32x4 va = xw[S == 2 ? u : 0][0], vb = xw[S == 2 ? u : 0][1], vc = xw[S == 2 ? u : 0][2], vd = xw[S == 2 ? u : 0][3];
  #pragma unroll
              for (int e = 0; e < 4; e++) {
                int best = 0; float bv = va[e];                   // first maximum in scan order, like MaxPool2d(2, 2)
                if (vb[e] > bv) { bv = vb[e]; best = 1; }
                if (vc[e] > bv) { bv = vc[e]; best = 2; }
                if (vd[e] > bv) { bv = vd[e]; best = 3; }
                if (best != me) v[e] = 0.f;
              }
            }
          }
          if (acc_dx) v += old[u];
          *(f32x4*)(dx + (((long)n * H + jy) * W + jx) * C + g * CQ + 4 * q4) = v;
        }
      }
    } else {   // stride 2: one item per iteration (the batched form lost 10 % there: registers)
      for (int it = threadIdx.x; it < TPB * NCH; it += TPB) {
        const int pix = it / NCH, ch = it % NCH;
        const int g = ch / NQ, q4 = ch % NQ;
        const int py = pix / TW, px = pix % TW;
        const int jy = ty0 + py, jx = tx0 + px;
        if (jy >= H || jx >= W) continue;
        float* dp = dx + (((long)n * H + jy) * W + jx) * C + g * CQ + 4 * q4;
        f32x4 v;
        if (g == 0) {
          v = *(const f32x4*)(Xs + pix * CQ + 4 * q4);
        } else {
          // bypass channel g*Cq + i <- beta * dOut[q][4i + g]
          const long q = ((long)n * Ho + jy / S) * Wo + jx / S;
          f32x4 D[4];
  #pragma unroll
          for (int e = 0; e < 4; e++) D[e] = *(const f32x4*)(dout + q * ldd + 16 * q4 + 4 * e);
  #pragma unroll
          for (int e = 0; e < 4; e++) v[e] = (g == 1 ? D[e][1] : g == 2 ? D[e][2] : D[e][3]) * bt;
          if (S == 2) {
            const float* xr = x + (((long)n * H + (jy & ~1)) * W + (jx & ~1)) * C + g * CQ + 4 * q4;   // top-left of the 2x2 window
            const int me = (jy & 1) * 2 + (jx & 1);
            const f32x4 va = *(const f32x4*)xr, vb = *(const f32x4*)(xr + C);
            const f32x4 vc = *(const f32x4*)(xr + (long)W * C), vd = *(const f32x4*)(xr + (long)W * C + C);
  #pragma unroll
            for (int e = 0; e < 4; e++) {
              int best = 0; float bv = va[e];                     // first maximum in scan order, like MaxPool2d(2, 2)
              if (vb[e] > bv) { bv = vb[e]; best = 1; }
              if (vc[e] > bv) { bv = vc[e]; best = 2; }
              if (vd[e] > bv) { bv = vd[e]; best = 3; }
              if (best != me) v[e] = 0.f;
            }
          }
        }
        if (acc_dx) v += *(const f32x4*)dp;
        *(f32x4*)dp = v;
      }
    }
  }
  if (S == 2 && threadIdx.x < CQ * CQ) wpart[(long)blockIdx.x * wtot + fr_off + threadIdx.x] = acc_fr;
}

// ------------------------------------------------------------------------------------------------------------------
// edge_bwd_conv: backward of one (relu ->) depthwise K x K (stride S, dilation DIL) -> 1x1 stage.
//   MODE 0  input relu(x1) (row stride C); output gradient from (dY, u, coef);  dx1 +=
//   MODE 1  input relu(bn_a(t)) (second stage of a sep_conv, S = 1);  writes dBN = d/d(bn_a(t)) and its BatchNorm-backward sums
//   MODE 2  input relu(x1); output gradient from (dBN, t, coef_a) (first stage of a sep_conv);  dx1 +=
// Tile = TH x TW OUTPUT pixels.  LDS: the activated input tile (forward halo), the tile of dd = Wpw^T g over the outputs that
// reach the owned input pixels (backward halo), and the per-pixel rows g, d (depthwise output) for the 1x1 weight gradient.
// ------------------------------------------------------------------------------------------------------------------
template <int CQ, int K, int DIL, int S, int MODE, int TH, int TW>
__global__ __launch_bounds__(TH * TW) void edge_bwd_conv_kernel(const typename std::conditional<MODE == 1, pre_t, float>::type* __restrict__ in, long ldin,
                                                                const float* __restrict__ gsrc /* dY or dBN */,
                                                                const pre_t* __restrict__ usrc /* u or t: pre-BN output of this stage */,
                                                                const float* __restrict__ mean_o, const float* __restrict__ invstd_o,
                                                                const float* __restrict__ coef_o, const float* __restrict__ mean_i,
                                                                const float* __restrict__ invstd_i, const float* __restrict__ w_dw,
                                                                const float* __restrict__ w_pw, int B, int H, int W,
                                                                float* __restrict__ dst /* dx1c [B,H,W,CQ] (MODE 0, 2) or dBN [P][CQ] (MODE 1) */,
                                                                int accumulate, double* __restrict__ spart /* MODE 1: [grid][2*CQ][2] */,
                                                                float* __restrict__ wpart, int wtot, int dw_off, int pw_off) {
  constexpr int TPB = TH * TW, NQ = CQ / 4, KK = K * K, PAD = DIL * (K - 1) / 2, HB = (PAD + S - 1) / S;
  constexpr int IH = S * (TH - 1) + 2 * PAD + 1, IW = S * (TW - 1) + 2 * PAD + 1, IN = IH * IW;
  constexpr int DH = TH + 2 * HB, DW = TW + 2 * HB, DN = DH * DW;
  constexpr int NE = NQ * KK;                                   // depthwise weight-gradient entries (channel quad, tap): <= 100
  constexpr int DSL = TPB / NE;                                 // pixel slices per entry
  constexpr int SL = TPB >= CQ * CQ ? TPB / (CQ * CQ) : 1;      // pixel slices of the 1x1 weight gradient
  constexpr int NPW = TPB >= CQ * CQ ? 1 : CQ * CQ / TPB;       // its entries per thread (narrow tiles at Cq = 16: 2)
  extern __shared__ __align__(16) unsigned char smem[];
  f32x4* rt = (f32x4*)smem;                                     // [NQ][IN] activated input
  f32x4* ddt = rt + NQ * IN;                                    // [NQ][DN]
  float* Gs = (float*)(ddt + NQ * DN);                         // [TPB][CQ]
  float* Ds = Gs + TPB * CQ;                                   // [TPB][CQ]
  float* wdl = Ds + TPB * CQ;                                  // [CQ][KK]
  float* wpl = wdl + CQ * KK;                                  // [CQ][CQ]
  float* mso = wpl + CQ * CQ;                                  // mean/invstd of the stage's output BN [2][CQ]
  float* cfo = mso + 2 * CQ;                                   // its coefficients [48]
  float* msi = cfo + COEF;                                     // MODE 1: mean/invstd of the input BN [2][CQ]
  float* red = msi + 2 * CQ;                                   // MODE 1: [8][TPB+1]
  float* seg = red + (MODE == 1 ? 8 * (TPB + 1) : 0);          // MODE 1: [2][8][8]
  double* acc = (double*)(seg + (MODE == 1 ? 128 : 0));        // MODE 1: [2][2*CQ]
  stage_weights_taps(wdl, w_dw, CQ, KK);
  stage_weights(wpl, w_pw, CQ * CQ);
  for (int i = threadIdx.x; i < CQ; i += TPB) {
    mso[i] = mean_o[i]; mso[CQ + i] = invstd_o[i];
    if (MODE == 1) { msi[i] = mean_i[i]; msi[CQ + i] = invstd_i[i]; }
  }
  for (int i = threadIdx.x; i < COEF; i += TPB) cfo[i] = coef_o[i];
  if (MODE == 1)
    for (int i = threadIdx.x; i < 4 * CQ; i += TPB) acc[i] = 0.0;
  const int Ho = H / S, Wo = W / S;
  const int tiles_y = (Ho + TH - 1) / TH, tiles_x = (Wo + TW - 1) / TW;
  const long ntiles = (long)B * tiles_y * tiles_x;
  const int ly = threadIdx.x / TW, lx = threadIdx.x % TW;
  f32x4 acc_dw = {0.f, 0.f, 0.f, 0.f};
  float acc_pw[NPW];
#pragma unroll
  for (int i = 0; i < NPW; i++) acc_pw[i] = 0.f;
  // 1x1 weight gradient for Cq >= 8: thread = (4 x 4 block of (output, input) channels, slice of the tile's pixels); a pixel costs two
  // 16-byte LDS reads for 16 multiply-adds (entry per thread: two 4-byte reads per multiply-add, 512 per tile at Cq = 16)
  constexpr bool PWB = CQ >= 8;
  constexpr int NB = (CQ / 4) * (CQ / 4), SLB = TPB / NB;
  float acc_pb[PWB ? 16 : 1];
#pragma unroll
  for (int i = 0; i < (PWB ? 16 : 1); i++) acc_pb[i] = 0.f;
  for (long tidx = blockIdx.x; tidx < ntiles; tidx += gridDim.x) {
    const int n = (int)(tidx / (tiles_y * tiles_x));
    const int tr = (int)(tidx % (tiles_y * tiles_x));
    const int ty0 = (tr / tiles_x) * TH, tx0 = (tr % tiles_x) * TW;
    __syncthreads();
    // activated input tile
    load_tile<NQ, MODE == 1 ? 2 : 1, IH, IW, TPB>(rt, in, ldin, n, H, W, S * ty0 - PAD, S * tx0 - PAD, msi, msi + CQ);
    // dd = Wpw^T g over the outputs (ty0 - HB .. ty0 + TH - 1 + HB); the rows of dY / u of all of a thread's positions are
    // requested before the first is used, and an owned position also leaves its g row for the 1x1 weight gradient
    {
      constexpr int ROUNDS = (DN + TPB - 1) / TPB, RB = NQ <= 2 ? ROUNDS : 1;    // rounds in flight together (registers)
#pragma unroll
      for (int r0 = 0; r0 < ROUNDS; r0 += RB) {
      f32x4 dv[RB][NQ], uv[RB][NQ];
      bool in[RB];
#pragma unroll
      for (int r = 0; r < RB; r++) {
        const int idx = threadIdx.x + (r0 + r) * TPB;
        const int oy = ty0 - HB + idx / DW, ox = tx0 - HB + idx % DW;
        in[r] = idx < DN && oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
        const long q = ((long)n * Ho + oy) * Wo + ox;
#pragma unroll
        for (int q4 = 0; q4 < NQ; q4++) {
          dv[r][q4] = f32x4{0.f, 0.f, 0.f, 0.f};
          uv[r][q4] = dv[r][q4];
          if (in[r]) { dv[r][q4] = *(const f32x4*)(gsrc + q * CQ + 4 * q4); uv[r][q4] = ldp4(usrc + q * CQ + 4 * q4); }
        }
      }
#pragma unroll
      for (int r = 0; r < RB; r++) {
        const int idx = threadIdx.x + (r0 + r) * TPB;
        if (idx < DN) {
          float g[CQ], dd[CQ];
#pragma unroll
          for (int q4 = 0; q4 < NQ; q4++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const int c = 4 * q4 + e;
              const float xh = (uv[r][q4][e] - mso[c]) * mso[CQ + c];
              g[c] = in[r] ? cfo[c] * (dv[r][q4][e] - cfo[16 + c] - xh * cfo[32 + c]) : 0.f;
            }
#pragma unroll
          for (int c4 = 0; c4 < NQ; c4++) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int oc = 0; oc < CQ; oc++) {
              const f32x4 w4 = *(const f32x4*)(wpl + oc * CQ + 4 * c4);
#pragma unroll
              for (int e = 0; e < 4; e++) a[e] = fmaf(w4[e], g[oc], a[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; e++) dd[4 * c4 + e] = a[e];
          }
#pragma unroll
          for (int q4 = 0; q4 < NQ; q4++) ddt[q4 * DN + idx] = f32x4{dd[4 * q4], dd[4 * q4 + 1], dd[4 * q4 + 2], dd[4 * q4 + 3]};
          const int py = idx / DW - HB, px = idx % DW - HB;       // position inside the owned tile?
          if (py >= 0 && py < TH && px >= 0 && px < TW) {
#pragma unroll
            for (int c = 0; c < CQ; c++) Gs[(py * TW + px) * CQ + c] = g[c];
          }
        }
      }
      }
    }
    __syncthreads();
    const int oy = ty0 + ly, ox = tx0 + lx;
    const bool live = oy < Ho && ox < Wo;
    // ---- own output pixel: the recomputed depthwise output d, staged beside g for the 1x1 weight gradient ----
    {
      float d[CQ];
      if (live) {
        dw_at<CQ, K, DIL, false, true>(rt, IN, IW, S * ly, S * lx, wdl, d);
      } else {
#pragma unroll
        for (int c = 0; c < CQ; c++) d[c] = 0.f;
      }
#pragma unroll
      for (int c = 0; c < CQ; c++) Ds[threadIdx.x * CQ + c] = d[c];
    }
    __syncthreads();
    if constexpr (PWB) {
      const int blk = threadIdx.x % NB, sl = threadIdx.x / NB;
      const int ob = blk / (CQ / 4), cb = blk % (CQ / 4);
      for (int p = sl; p < TPB; p += SLB) {
        const f32x4 g4 = *(const f32x4*)(Gs + p * CQ + 4 * ob), d4 = *(const f32x4*)(Ds + p * CQ + 4 * cb);
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < 4; b++) acc_pb[4 * a + b] = fmaf(g4[a], d4[b], acc_pb[4 * a + b]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPW; i++) {
        const int t = threadIdx.x + i * TPB;
        if (t < SL * CQ * CQ) {
          const int e = t % (CQ * CQ), sl = t / (CQ * CQ);
          const int oc = e / CQ, c = e % CQ;
          float a = 0.f;
          for (int p = sl; p < TPB; p += SL) a = fmaf(Gs[p * CQ + oc], Ds[p * CQ + c], a);
          acc_pw[i] += a;
        }
      }
    }
    // ---- depthwise weight gradient: entry (channel quad, tap) sums dd[o][c] * r[S*o - PAD + DIL*tap][c] over the tile's
    //      outputs; thread = entry x slice of the tile's pixels, 16-byte LDS reads ----
    if (threadIdx.x < DSL * NE) {
      const int ent = threadIdx.x % NE, sl = threadIdx.x / NE;
      const int q4 = ent / KK, tap = ent % KK, tj = tap / K, ti = tap % K;
      const f32x4* rp = rt + q4 * IN + DIL * tj * IW + DIL * ti;
      const f32x4* dp = ddt + q4 * DN + HB * DW + HB;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      for (int pp = sl; pp < TPB; pp += DSL) {
        const int py = pp / TW, px = pp % TW;
        const f32x4 dv = dp[py * DW + px], rv = rp[S * py * IW + S * px];
#pragma unroll
        for (int e = 0; e < 4; e++) a[e] = fmaf(dv[e], rv[e], a[e]);
      }
      acc_dw += a;
    }
    // ---- input gradient at the owned input pixels (S*oy + a, S*ox + b): gather over the taps ----
    float stat[2][CQ];
#pragma unroll
    for (int a = 0; a < S; a++)
#pragma unroll
      for (int b = 0; b < S; b++) {
        const int iy = S * oy + a, ix = S * ox + b;
        float din[CQ];
#pragma unroll
        for (int q4 = 0; q4 < NQ; q4++) {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < K; j++) {
            const int ty = a + PAD - DIL * j;
            if (ty % S != 0) continue;
#pragma unroll
            for (int i = 0; i < K; i++) {
              const int tx = b + PAD - DIL * i;
              if (tx % S != 0) continue;
              const f32x4 v = ddt[q4 * DN + (ly + HB + ty / S) * DW + lx + HB + tx / S];
              const f32x4 w4 = *(const f32x4*)(wdl + (j * K + i) * CQ + 4 * q4);
#pragma unroll
              for (int e = 0; e < 4; e++) sacc[e] = fmaf(w4[e], v[e], sacc[e]);
            }
          }
          const f32x4 r = rt[q4 * IN + (S * ly + a + PAD) * IW + S * lx + b + PAD];
#pragma unroll
          for (int e = 0; e < 4; e++) din[4 * q4 + e] = r[e] > 0.f ? sacc[e] : 0.f;
          if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 4; e++) { stat[0][4 * q4 + e] = din[4 * q4 + e]; stat[1][4 * q4 + e] = din[4 * q4 + e] * r[e]; }
          }
        }
        if (live && iy < H && ix < W) {
          float* dp = dst + (((long)n * H + iy) * W + ix) * CQ;
          if (MODE != 1 && accumulate) {
#pragma unroll
            for (int q4 = 0; q4 < NQ; q4++) {
              const f32x4 o = *(const f32x4*)(dp + 4 * q4);
#pragma unroll
              for (int e = 0; e < 4; e++) din[4 * q4 + e] += o[e];
            }
          }
          store_row<CQ>(dp, din);
        }
      }
    if (MODE == 1) phase_stats<CQ, 2, 2, TPB>(stat, live, red, seg, acc, 0);
  }
  // ---- per-workgroup partial weight gradients ----
  __syncthreads();
  float* wrow = wpart + (long)blockIdx.x * wtot;
  ((f32x4*)Ds)[threadIdx.x] = acc_dw;                            // [slice][entry] of float4
  __syncthreads();
  if (threadIdx.x < NE) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < DSL; sl++) a += ((const f32x4*)Ds)[sl * NE + threadIdx.x];
    const int q4 = threadIdx.x / KK, tap = threadIdx.x % KK;
#pragma unroll
    for (int e = 0; e < 4; e++) wrow[dw_off + (4 * q4 + e) * KK + tap] = a[e];
  }
  __syncthreads();
  if constexpr (PWB) {
    // [slice][block][16] through LDS (TPB * 16 floats = the Gs + Ds region at Cq = 8, half of it at Cq = 16), then one thread per entry
    // adds the slices in a fixed order
#pragma unroll
    for (int i = 0; i < 16; i++) Gs[threadIdx.x * 16 + i] = acc_pb[i];
    __syncthreads();
    for (int e = threadIdx.x; e < CQ * CQ; e += TPB) {
      const int oc = e / CQ, c = e % CQ;
      const int blk = (oc / 4) * (CQ / 4) + c / 4, within = (oc % 4) * 4 + c % 4;
      float a = 0.f;
      for (int sl = 0; sl < SLB; sl++) a += Gs[(sl * NB + blk) * 16 + within];
      wrow[pw_off + e] = a;
    }
  } else if (SL > 1) {
    Gs[threadIdx.x] = acc_pw[0];
    __syncthreads();
    if (threadIdx.x < CQ * CQ) {
      float a = 0.f;
      for (int sl = 0; sl < SL; sl++) a += Gs[sl * CQ * CQ + threadIdx.x];
      wrow[pw_off + threadIdx.x] = a;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NPW; i++)
      if (threadIdx.x + i * TPB < CQ * CQ) wrow[pw_off + threadIdx.x + i * TPB] = acc_pw[i];
  }
  if (MODE == 1) {
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * CQ; i += TPB) {
      spart[((long)blockIdx.x * 2 * CQ + i) * 2 + 0] = acc[i];
      spart[((long)blockIdx.x * 2 * CQ + i) * 2 + 1] = acc[2 * CQ + i];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// edge_bwd_conv2: the same stage backward (same arguments, same outputs), laid out for the machine instead of for the pixel:
//   phase B  dd = Wpw^T g on the halo of output positions: lane = (position, channel quad), the 1x1 product is four
//            v_mfma_f32_16x16x4_f32 (exact fp32) whose B operand IS that layout (k-slot (kk, lane group) = channel 4*quad + kk) and
//            whose result lands in it again; the 1x1 weights sit in four registers per lane (block-diagonal for Cq < 16: 16 / Cq
//            position blocks share one instruction).  No LDS traffic for the 1x1.
//   phase C  depthwise part: a WAVE owns one channel quad for the whole kernel, a lane one pixel of a 64-pixel item.  The quad's
//            weights are wave-uniform: scalar loads from a tap-major copy (SGPR operands of v_pk_fma_f32, no LDS, no VGPRs);
//            the recomputed depthwise output, the input gradient (gather over the taps of the dd tile) and the depthwise weight
//            gradient share the tile reads; the weight gradient is K*K float4 accumulators PER LANE carried over all tiles
//            (the first version re-read both tiles once per (tap, pixel) from a second thread mapping).
//   phase D  1x1 weight gradient = sum over pixels g (x) d on the matrix cores, pixels on the k index (A = Gs, B = Ds read as
//            4-byte words, lane = channel), accumulated in 4 registers per lane over all tiles.
// LDS bytes read per tile: about a third of the first version's; three barriers per tile instead of four.
// ------------------------------------------------------------------------------------------------------------------

template <int CQ, int K, int DIL, int S, int MODE, int TH, int TW, int NW, int WPE>
__global__ __launch_bounds__(64 * NW, WPE) void edge_bwd_conv2_kernel(const typename std::conditional<MODE == 1, pre_t, float>::type* __restrict__ in, long ldin,
                                                                 const float* __restrict__ gsrc /* dY or dBN */,
                                                                 const pre_t* __restrict__ usrc /* u or t: pre-BN output of this stage */,
                                                                 const float* __restrict__ mean_o, const float* __restrict__ invstd_o,
                                                                 const float* __restrict__ coef_o, const float* __restrict__ mean_i,
                                                                 const float* __restrict__ invstd_i, const float* __restrict__ w_taps /* [K*K][CQ] */,
                                                                 const float* __restrict__ w_pw, int B, int H, int W,
                                                                 float* __restrict__ dst /* dx1c [B,H,W,CQ] (MODE 0, 2) or dBN [P][CQ] (MODE 1) */,
                                                                 int accumulate, double* __restrict__ spart /* MODE 1: [grid][2*CQ][2] */,
                                                                 float* __restrict__ wpart, int wtot, int dw_off, int pw_off,
                                                                 int need_w /* 0: input gradient only (no weight-gradient work, wpart untouched) */) {
  constexpr int TPB = 64 * NW, NPX = TH * TW, NQ = CQ / 4, KK = K * K, PAD = DIL * (K - 1) / 2, HB = (PAD + S - 1) / S;
  constexpr int IH = S * (TH - 1) + 2 * PAD + 1, IW = S * (TW - 1) + 2 * PAD + 1, IN = IH * IW;
  constexpr int DH = TH + 2 * HB, DW = TW + 2 * HB, DN = DH * DW;
  constexpr int PPW = 64 / NQ;                                   // positions of one phase-B item (16 per position block)
  constexpr int NITB = (DN + PPW - 1) / PPW, ITB = (NITB + NW - 1) / NW;
  constexpr int NITC = NPX / 64, ITC = NITC * NQ / NW;           // 64-pixel items; (item, quad) pairs per wave
  constexpr int NBK = 16 / CQ;                                   // position blocks of one 16x16x4 instruction
  constexpr int NM = NPX / (4 * NBK), MPW = NM / NW;             // phase-D instructions per tile / per wave
  constexpr int RK = NW / NQ, NT = (KK + RK - 1) / RK;           // waves that share a quad split the K*K taps of the depthwise weight gradient
  static_assert(NW % NQ == 0 && (NITC * NQ) % NW == 0 && NPX % 64 == 0 && NM % NW == 0 && (TW & (TW - 1)) == 0, "tile / wave split");
  extern __shared__ __align__(16) unsigned char smem[];
  f32x4* rt = (f32x4*)smem;                                     // [NQ][IN] activated input
  f32x4* ddt = rt + NQ * IN;                                    // [NQ][DN]
  float* Gs = (float*)(ddt + NQ * DN);                         // [NPX][CQ]
  float* Ds = Gs + NPX * CQ;                                   // [NPX][CQ]
  float* cst = Ds + NPX * CQ;                                  // [5][CQ]: mean, invstd of the stage's output BN, its three coefficients
  float* msi = cst + 5 * CQ;                                   // MODE 1: mean/invstd of the input BN [2][CQ]
  for (int i = threadIdx.x; i < CQ; i += TPB) {
    cst[i] = mean_o[i]; cst[CQ + i] = invstd_o[i];
    cst[2 * CQ + i] = coef_o[i]; cst[3 * CQ + i] = coef_o[16 + i]; cst[4 * CQ + i] = coef_o[32 + i];
    if (MODE == 1) { msi[i] = mean_i[i]; msi[CQ + i] = invstd_i[i]; }
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n16 = lane & 15, jg = lane >> 4;
  // phase B: lane = (position slot n16 + 16 * block, channel quad qB)
  const int qB = jg % NQ, blkB = jg / NQ, slotB = n16 + 16 * blkB;
  float aop[4];                                                 // A[i = (block, c)][k-slot (kk, jg)] = Wpw[oc = 4 qB + kk][c] on the block diagonal
#pragma unroll
  for (int kk = 0; kk < 4; kk++) aop[kk] = (n16 / CQ == blkB) ? w_pw[(4 * qB + kk) * CQ + n16 % CQ] : 0.f;
  // phase C: this wave's channel quad and its weights (scalar loads)
  const int qC = wave % NQ, itc0 = wave / NQ;                   // itc0 = this wave's rank among the RK waves of its quad
  f32x4 accdw[NT];                                              // taps s * RK + rank (a clamped duplicate in the last slot of some ranks: never written out)
  int toff[NT];                                                 // their tile offsets (scalar)
#pragma unroll
  for (int t = 0; t < NT; t++) {
    accdw[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tap = (t * RK + itc0) < KK ? t * RK + itc0 : KK - 1;
    toff[t] = (DIL * (tap / K)) * IW + DIL * (tap % K);
  }
  f32x4 accp = {0.f, 0.f, 0.f, 0.f};
  f32x4 st0 = {0.f, 0.f, 0.f, 0.f}, st1 = st0;
  const int Ho = H / S, Wo = W / S;
  const int tiles_y = (Ho + TH - 1) / TH, tiles_x = (Wo + TW - 1) / TW;
  const long ntiles = (long)B * tiles_y * tiles_x;
  for (long tidx = blockIdx.x; tidx < ntiles; tidx += gridDim.x) {
    const int n = (int)(tidx / (tiles_y * tiles_x));
    const int tr = (int)(tidx % (tiles_y * tiles_x));
    const int ty0 = (tr / tiles_x) * TH, tx0 = (tr % tiles_x) * TW;
    __syncthreads();
    int tid = threadIdx.x, slot = slotB, ln = lane;
    asm volatile("" : "+v"(tid), "+v"(slot), "+v"(ln));           // per-tile copies: the index math of every round stays inside the loop
    {   // ---- phase B (the loads of its first rows are requested BEFORE the input tile: one memory round trip for both) ----
      constexpr int UB = ITB < 4 ? ITB : 4;
      f32x4 dv[UB], uv[UB];
      bool inb[UB];
      auto request = [&](int b0) {
#pragma unroll
        for (int u = 0; u < UB; u++) {
          const int idx = (wave + (b0 + u) * NW) * PPW + slot;
          const int oy = ty0 - HB + idx / DW, ox = tx0 - HB + idx % DW;
          inb[u] = (b0 + u < ITB) && idx < DN && oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
          dv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          uv[u] = dv[u];
          if (inb[u]) {
            const long q = ((long)n * Ho + oy) * Wo + ox;
#ifdef EDGE_ABL_NOGLOAD
            dv[u] = f32x4{1.f, 2.f, 3.f, (float)q}; uv[u] = dv[u];
#else
            dv[u] = *(const f32x4*)(gsrc + q * CQ + 4 * qB);
            uv[u] = ldp4(usrc + q * CQ + 4 * qB);
#endif
          }
        }
      };
      request(0);
#ifdef EDGE_ABL_NOTILE
      for (int i = tid; i < NQ * IN; i += TPB) rt[i] = f32x4{1.f, 0.f, 2.f, (float)i};
#else
      load_tile<NQ, MODE == 1 ? 2 : 1, IH, IW, TPB>(rt, in, ldin, n, H, W, S * ty0 - PAD, S * tx0 - PAD, msi, msi + CQ, tid);
#endif
      const f32x4 mn = *(const f32x4*)(cst + 4 * qB), is = *(const f32x4*)(cst + CQ + 4 * qB);
      const f32x4 c0 = *(const f32x4*)(cst + 2 * CQ + 4 * qB), c1 = *(const f32x4*)(cst + 3 * CQ + 4 * qB), c2 = *(const f32x4*)(cst + 4 * CQ + 4 * qB);
#pragma unroll
      for (int b0 = 0; b0 < ITB; b0 += UB) {
        if (b0) request(b0);
#pragma unroll
        for (int u = 0; u < UB; u++) {
          if (b0 + u < ITB && wave + (b0 + u) * NW < NITB) {      // wave-uniform
            const int idx = (wave + (b0 + u) * NW) * PPW + slot;
            f32x4 g;
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const float xh = (uv[u][e] - mn[e]) * is[e];
              g[e] = inb[u] ? c0[e] * (dv[u][e] - c1[e] - xh * c2[e]) : 0.f;
            }
            f32x4 dd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 4; kk++) dd = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[kk], g[kk], dd, 0, 0, 0);
            if (idx < DN) {
              ddt[qB * DN + idx] = dd;
              const int py = idx / DW - HB, px = idx % DW - HB;     // position inside the owned tile: leave g for the 1x1 weight gradient
              if (py >= 0 && py < TH && px >= 0 && px < TW) *(f32x4*)(Gs + (py * TW + px) * CQ + 4 * qB) = g;
            }
          }
        }
      }
    }
    __syncthreads();
    // ---- phase C ----  (one item at a time: unrolled items would keep several items' tile reads and all K*K weight quads live)
#pragma unroll 1
    for (int t = 0; t < ITC; t++) {
      unsigned long wqa = (unsigned long)(uintptr_t)(w_taps + 4 * qC);
      asm volatile("" : "+s"(wqa));                              // re-issue the scalar weight loads per item instead of holding 4*K*K SGPRs
      const const_f32x4_ptr wq = (const_f32x4_ptr)wqa;
      const int p = (itc0 + t * (NW / NQ)) * 64 + ln;
      const int ly = p / TW, lx = p % TW;
      const int oy = ty0 + ly, ox = tx0 + lx;
      const bool live = oy < Ho && ox < Wo;
      const f32x4* rp = rt + qC * IN + (S * ly) * IW + S * lx;
      const f32x4* dq = ddt + qC * DN + (ly + HB) * DW + lx + HB;
      f32x4 prev[S * S];                                           // what the earlier stages left in dst: requested now, used after the tap loops
#pragma unroll
      for (int ab = 0; ab < S * S; ab++) {
        prev[ab] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int iy = S * oy + ab / S, ix = S * ox + ab % S;
#ifndef EDGE_ABL_NOACC
        if (MODE != 1 && accumulate && live && iy < H && ix < W) prev[ab] = *(const f32x4*)(dst + (((long)n * H + iy) * W + ix) * CQ + 4 * qC);
#endif
      }
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < K; j++)
#pragma unroll
        for (int i = 0; i < K; i++) {
          const f32x4 v = rp[(DIL * j) * IW + DIL * i];
          const f32x4 w4 = wq[(j * K + i) * NQ];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            d[e] = fmaf(w4[e], v[e], d[e]);
          }
        }
      *(f32x4*)(Ds + p * CQ + 4 * qC) = d;
#pragma unroll
      for (int a = 0; a < S; a++)
#pragma unroll
        for (int b = 0; b < S; b++) {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < K; j++) {
            const int ty = a + PAD - DIL * j;
            if (ty % S != 0) continue;
#pragma unroll
            for (int i = 0; i < K; i++) {
              const int tx = b + PAD - DIL * i;
              if (tx % S != 0) continue;
              const f32x4 v = dq[(ty / S) * DW + tx / S];
              const f32x4 w4 = wq[(j * K + i) * NQ];
#pragma unroll
              for (int e = 0; e < 4; e++) sacc[e] = fmaf(w4[e], v[e], sacc[e]);
            }
          }
          const f32x4 r = rp[(a + PAD) * IW + b + PAD];
          f32x4 din;
#pragma unroll
          for (int e = 0; e < 4; e++) din[e] = r[e] > 0.f ? sacc[e] : 0.f;
          if (MODE == 1 && live) {
#pragma unroll
            for (int e = 0; e < 4; e++) { st0[e] += din[e]; st1[e] = fmaf(din[e], r[e], st1[e]); }
          }
          const int iy = S * oy + a, ix = S * ox + b;
          if (live && iy < H && ix < W) {
            float* dp = dst + (((long)n * H + iy) * W + ix) * CQ + 4 * qC;
            *(f32x4*)dp = din + prev[a * S + b];
          }
        }
    }
#ifndef EDGE_ABL_NODW
    // depthwise weight gradient: every wave walks ALL items of the tile for its quad with its own NT taps (K*K accumulators per lane
    // cost K = 5 its occupancy: 256 registers and spills; split over the RK waves of the quad they are 28 / 52 registers)
#pragma unroll 1
    for (int it = 0; it < (need_w ? NITC : 0); it++) {
      const int p = it * 64 + ln;
      const int ly = p / TW, lx = p % TW;
      const f32x4* rp = rt + qC * IN + (S * ly) * IW + S * lx;
      const f32x4 ddo = ddt[qC * DN + (ly + HB) * DW + lx + HB];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const f32x4 v = rp[toff[t]];
#pragma unroll
        for (int e = 0; e < 4; e++) accdw[t][e] = fmaf(ddo[e], v[e], accdw[t][e]);
      }
    }
#endif
    __syncthreads();
    // ---- phase D ----
    if (need_w) {
#pragma unroll
      for (int m = 0; m < MPW; m++) {
        const int pix = (wave * MPW + m) * (4 * NBK) + 4 * (n16 / CQ) + jg;
        accp = __builtin_amdgcn_mfma_f32_16x16x4f32(Gs[pix * CQ + n16 % CQ], Ds[pix * CQ + n16 % CQ], accp, 0, 0, 0);
      }
    }
  }
  // ---- per-workgroup partial weight gradients (fixed order) ----
  __syncthreads();
  if (!need_w && MODE != 1) return;
  float* wrow = wpart + (long)blockIdx.x * wtot;
  float* fin = (float*)smem;                                    // [NW][16][NT*4] then [NW][16][16] then [NW][8]
  float* fin2 = fin + NW * 16 * NT * 4;
  float* fin3 = fin2 + NW * 256;
#pragma unroll
  for (int t = 0; t < NT; t++) {
    f32x4 v = accdw[t];
#pragma unroll
    for (int e = 0; e < 4; e++) { v[e] += __shfl_xor(v[e], 32, 64); v[e] += __shfl_xor(v[e], 16, 64); }
    if (lane < 16) *(f32x4*)(fin + ((wave * 16 + lane) * NT + t) * 4) = v;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) fin2[wave * 256 + (4 * jg + r) * 16 + n16] = accp[r];
  if (MODE == 1) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float a = wave_sum(st0[e]), b = wave_sum(st1[e]);
      if (lane == 0) { fin3[wave * 8 + e] = a; fin3[wave * 8 + 4 + e] = b; }
    }
  }
  __syncthreads();
  for (int ent = threadIdx.x; ent < (need_w ? CQ * KK : 0); ent += TPB) {
    const int c = ent / KK, tap = ent % KK;
    const int w = c / 4 + NQ * (tap % RK), t = tap / RK;           // the one wave that owns (quad, tap)
    float a = 0.f;
    for (int l = 0; l < 16; l++) a += fin[((w * 16 + l) * NT + t) * 4 + c % 4];
    wrow[dw_off + ent] = a;
  }
  for (int ent = threadIdx.x; ent < (need_w ? CQ * CQ : 0); ent += TPB) {
    const int oc = ent / CQ, c = ent % CQ;
    float a = 0.f;
    for (int w = 0; w < NW; w++)
      for (int b = 0; b < NBK; b++) a += fin2[w * 256 + (b * CQ + oc) * 16 + b * CQ + c];
    wrow[pw_off + ent] = a;
  }
  if (MODE == 1) {
    for (int i = threadIdx.x; i < 2 * CQ; i += TPB) {
      const int k = i / CQ, c = i % CQ;
      double a = 0.0;
      for (int w = c / 4; w < NW; w += NQ) a += (double)fin3[w * 8 + 4 * k + c % 4];
      spart[((long)blockIdx.x * 2 * CQ + i) * 2 + 0] = a;
      spart[((long)blockIdx.x * 2 * CQ + i) * 2 + 1] = 0.0;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// edge_bwd_conv2_pair: the 3x3 and the 5x5 stage that read the SAME input (dil_conv_3x3 + dil_conv_5x5: MODE 0; the first stages of
// sep_conv_3x3 + sep_conv_5x5: MODE 2) in one pass over the tiles: one input tile (the 5x5 halo), one read-modify-write of dx1c
// instead of two, half the launches.  At Cq = 4 / 8 these stages move bytes, not flops (7 T each alone, 8.6 - 10.3 T as a pair).
// Same phases and lane layouts as edge_bwd_conv2_kernel; stage 0 = 3x3, stage 1 = 5x5.
// ------------------------------------------------------------------------------------------------------------------
struct StageArgs {
  const float* gsrc;      // dY or dBN
  const pre_t* usrc;      // pre-BN output of the stage
  const float* mean_o; const float* invstd_o; const float* coef_o;
  const float* w_taps;    // tap-major depthwise weights [K*K][CQ]
  const float* w_pw;
  int dw_off, pw_off;
};

template <int CQ, int DIL, int S, int MODE, int TH, int TW, int NW, int WPE>
__global__ __launch_bounds__(64 * NW, WPE) void edge_bwd_conv2_pair_kernel(const float* __restrict__ in, long ldin, StageArgs sa, StageArgs sb, int B, int H, int W,
                                                                      float* __restrict__ dst, int accumulate, float* __restrict__ wpart, int wtot,
                                                                      int need_w) {
  static_assert(MODE == 0 || MODE == 2, "stages that read x1");
  constexpr int TPB = 64 * NW, NPX = TH * TW, NQ = CQ / 4;
  constexpr int PAD = 2 * DIL;                                   // the 5x5 stage's reach: the shared input tile
  constexpr int IH = S * (TH - 1) + 2 * PAD + 1, IW = S * (TW - 1) + 2 * PAD + 1, IN = IH * IW;
  constexpr int PPW = 64 / NQ, NITC = NPX / 64, ITC = NITC * NQ / NW, NBK = 16 / CQ, NM = NPX / (4 * NBK), MPW = NM / NW, RK = NW / NQ;
  static_assert(NW % NQ == 0 && (NITC * NQ) % NW == 0 && NPX % 64 == 0 && NM % NW == 0 && (TW & (TW - 1)) == 0, "tile / wave split");
  // per stage: K, its padding inside the shared tile, halo of output positions, dd tile
  constexpr int K0 = 3, K1 = 5, P0 = DIL, P1 = 2 * DIL, HB0 = (P0 + S - 1) / S, HB1 = (P1 + S - 1) / S;
  constexpr int DW0 = TW + 2 * HB0, DN0 = (TH + 2 * HB0) * DW0, DW1 = TW + 2 * HB1, DN1 = (TH + 2 * HB1) * DW1;
  constexpr int NT0 = (K0 * K0 + RK - 1) / RK, NT1 = (K1 * K1 + RK - 1) / RK;
  extern __shared__ __align__(16) unsigned char smem[];
  f32x4* rt = (f32x4*)smem;                                     // [NQ][IN]
  f32x4* ddt0 = rt + NQ * IN;                                   // [NQ][DN0]
  f32x4* ddt1 = ddt0 + NQ * DN0;                                // [NQ][DN1]
  float* Gs0 = (float*)(ddt1 + NQ * DN1);                      // [NPX][CQ] each
  float* Gs1 = Gs0 + NPX * CQ;
  float* Ds0 = Gs1 + NPX * CQ;
  float* Ds1 = Ds0 + NPX * CQ;
  float* cst0 = Ds1 + NPX * CQ;                                // [5][CQ] each
  float* cst1 = cst0 + 5 * CQ;
  for (int i = threadIdx.x; i < CQ; i += TPB) {
    cst0[i] = sa.mean_o[i]; cst0[CQ + i] = sa.invstd_o[i];
    cst0[2 * CQ + i] = sa.coef_o[i]; cst0[3 * CQ + i] = sa.coef_o[16 + i]; cst0[4 * CQ + i] = sa.coef_o[32 + i];
    cst1[i] = sb.mean_o[i]; cst1[CQ + i] = sb.invstd_o[i];
    cst1[2 * CQ + i] = sb.coef_o[i]; cst1[3 * CQ + i] = sb.coef_o[16 + i]; cst1[4 * CQ + i] = sb.coef_o[32 + i];
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n16 = lane & 15, jg = lane >> 4;
  const int qB = jg % NQ, blkB = jg / NQ, slotB = n16 + 16 * blkB;
  float aop0[4], aop1[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) {
    aop0[kk] = (n16 / CQ == blkB) ? sa.w_pw[(4 * qB + kk) * CQ + n16 % CQ] : 0.f;
    aop1[kk] = (n16 / CQ == blkB) ? sb.w_pw[(4 * qB + kk) * CQ + n16 % CQ] : 0.f;
  }
  const int qC = wave % NQ, itc0 = wave / NQ;
  f32x4 acc0[NT0], acc1[NT1];
  int toff0[NT0], toff1[NT1];
#pragma unroll
  for (int t = 0; t < NT0; t++) {
    acc0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tap = (t * RK + itc0) < K0 * K0 ? t * RK + itc0 : K0 * K0 - 1;
    toff0[t] = (DIL * (tap / K0) + P1 - P0) * IW + DIL * (tap % K0) + P1 - P0;
  }
#pragma unroll
  for (int t = 0; t < NT1; t++) {
    acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tap = (t * RK + itc0) < K1 * K1 ? t * RK + itc0 : K1 * K1 - 1;
    toff1[t] = (DIL * (tap / K1)) * IW + DIL * (tap % K1);
  }
  f32x4 accp0 = {0.f, 0.f, 0.f, 0.f}, accp1 = accp0;
  const int Ho = H / S, Wo = W / S;
  const int tiles_y = (Ho + TH - 1) / TH, tiles_x = (Wo + TW - 1) / TW;
  const long ntiles = (long)B * tiles_y * tiles_x;
  for (long tidx = blockIdx.x; tidx < ntiles; tidx += gridDim.x) {
    const int n = (int)(tidx / (tiles_y * tiles_x));
    const int tr = (int)(tidx % (tiles_y * tiles_x));
    const int ty0 = (tr / tiles_x) * TH, tx0 = (tr % tiles_x) * TW;
    __syncthreads();
    int tid = threadIdx.x, slot = slotB, ln = lane;
    asm volatile("" : "+v"(tid), "+v"(slot), "+v"(ln));
    // ---- phase B of both stages (their loads are requested before the input tile) ----
    auto phase_b = [&](auto hbc, auto dwc, auto dnc, const StageArgs& st, const float* cst, const float* aop, f32x4* ddt, float* Gs, bool with_tile) {
      constexpr int HB = decltype(hbc)::value, DW = decltype(dwc)::value, DN = decltype(dnc)::value;
      constexpr int NITB = (DN + PPW - 1) / PPW, ITB = (NITB + NW - 1) / NW, UB = ITB < 4 ? ITB : 4;
      f32x4 dv[UB], uv[UB];
      bool inb[UB];
      auto request = [&](int b0) {
#pragma unroll
        for (int u = 0; u < UB; u++) {
          const int idx = (wave + (b0 + u) * NW) * PPW + slot;
          const int oy = ty0 - HB + idx / DW, ox = tx0 - HB + idx % DW;
          inb[u] = (b0 + u < ITB) && idx < DN && oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
          dv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          uv[u] = dv[u];
          if (inb[u]) {
            const long q = ((long)n * Ho + oy) * Wo + ox;
            dv[u] = *(const f32x4*)(st.gsrc + q * CQ + 4 * qB);
            uv[u] = ldp4(st.usrc + q * CQ + 4 * qB);
          }
        }
      };
      request(0);
      if (with_tile) load_tile<NQ, 1, IH, IW, TPB>(rt, in, ldin, n, H, W, S * ty0 - PAD, S * tx0 - PAD, nullptr, nullptr, tid);
      const f32x4 mn = *(const f32x4*)(cst + 4 * qB), is = *(const f32x4*)(cst + CQ + 4 * qB);
      const f32x4 c0 = *(const f32x4*)(cst + 2 * CQ + 4 * qB), c1 = *(const f32x4*)(cst + 3 * CQ + 4 * qB), c2 = *(const f32x4*)(cst + 4 * CQ + 4 * qB);
#pragma unroll
      for (int b0 = 0; b0 < ITB; b0 += UB) {
        if (b0) request(b0);
#pragma unroll
        for (int u = 0; u < UB; u++) {
          if (b0 + u < ITB && wave + (b0 + u) * NW < NITB) {
            const int idx = (wave + (b0 + u) * NW) * PPW + slot;
            f32x4 g;
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const float xh = (uv[u][e] - mn[e]) * is[e];
              g[e] = inb[u] ? c0[e] * (dv[u][e] - c1[e] - xh * c2[e]) : 0.f;
            }
            f32x4 dd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 4; kk++) dd = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[kk], g[kk], dd, 0, 0, 0);
            if (idx < DN) {
              ddt[qB * DN + idx] = dd;
              const int py = idx / DW - HB, px = idx % DW - HB;
              if (py >= 0 && py < TH && px >= 0 && px < TW) *(f32x4*)(Gs + (py * TW + px) * CQ + 4 * qB) = g;
            }
          }
        }
      }
    };
    phase_b(std::integral_constant<int, HB1>{}, std::integral_constant<int, DW1>{}, std::integral_constant<int, DN1>{}, sb, cst1, aop1, ddt1, Gs1, true);
    phase_b(std::integral_constant<int, HB0>{}, std::integral_constant<int, DW0>{}, std::integral_constant<int, DN0>{}, sa, cst0, aop0, ddt0, Gs0, false);
    __syncthreads();
    // ---- phase C: both stages of an (item, quad) pair, one write of dx1c ----
    auto stage_c = [&](auto kc, auto pc, auto hbc, auto dwc, auto dnc, const float* w_taps, const f32x4* ddt, float* Ds, int p, int ly, int lx, f32x4* sacc /* [S*S] */) {
      constexpr int K = decltype(kc)::value, PS = decltype(pc)::value, HB = decltype(hbc)::value, DW = decltype(dwc)::value, DN = decltype(dnc)::value;
      unsigned long wqa = (unsigned long)(uintptr_t)(w_taps + 4 * qC);
      asm volatile("" : "+s"(wqa));
      const const_f32x4_ptr wq = (const_f32x4_ptr)wqa;
      const f32x4* rp = rt + qC * IN + (S * ly + P1 - PS) * IW + S * lx + P1 - PS;
      const f32x4* dq = ddt + qC * DN + (ly + HB) * DW + lx + HB;
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < K; j++)
#pragma unroll
        for (int i = 0; i < K; i++) {
          const f32x4 v = rp[(DIL * j) * IW + DIL * i];
          const f32x4 w4 = wq[(j * K + i) * NQ];
#pragma unroll
          for (int e = 0; e < 4; e++) d[e] = fmaf(w4[e], v[e], d[e]);
        }
      *(f32x4*)(Ds + p * CQ + 4 * qC) = d;
#pragma unroll
      for (int a = 0; a < S; a++)
#pragma unroll
        for (int b = 0; b < S; b++) {
          f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < K; j++) {
            const int ty = a + PS - DIL * j;
            if (ty % S != 0) continue;
#pragma unroll
            for (int i = 0; i < K; i++) {
              const int tx = b + PS - DIL * i;
              if (tx % S != 0) continue;
              const f32x4 v = dq[(ty / S) * DW + tx / S];
              const f32x4 w4 = wq[(j * K + i) * NQ];
#pragma unroll
              for (int e = 0; e < 4; e++) sc[e] = fmaf(w4[e], v[e], sc[e]);
            }
          }
          sacc[a * S + b] += sc;
        }
    };
#pragma unroll 1
    for (int t = 0; t < ITC; t++) {
      const int p = (itc0 + t * (NW / NQ)) * 64 + ln;
      const int ly = p / TW, lx = p % TW;
      const int oy = ty0 + ly, ox = tx0 + lx;
      const bool live = oy < Ho && ox < Wo;
      f32x4 prev[S * S], sacc[S * S];
#pragma unroll
      for (int ab = 0; ab < S * S; ab++) {
        prev[ab] = f32x4{0.f, 0.f, 0.f, 0.f};
        sacc[ab] = prev[ab];
        const int iy = S * oy + ab / S, ix = S * ox + ab % S;
        if (accumulate && live && iy < H && ix < W) prev[ab] = *(const f32x4*)(dst + (((long)n * H + iy) * W + ix) * CQ + 4 * qC);
      }
      stage_c(std::integral_constant<int, K0>{}, std::integral_constant<int, P0>{}, std::integral_constant<int, HB0>{}, std::integral_constant<int, DW0>{},
              std::integral_constant<int, DN0>{}, sa.w_taps, ddt0, Ds0, p, ly, lx, sacc);
      stage_c(std::integral_constant<int, K1>{}, std::integral_constant<int, P1>{}, std::integral_constant<int, HB1>{}, std::integral_constant<int, DW1>{},
              std::integral_constant<int, DN1>{}, sb.w_taps, ddt1, Ds1, p, ly, lx, sacc);
      const f32x4* rc = rt + qC * IN + (S * ly + P1) * IW + S * lx + P1;
#pragma unroll
      for (int a = 0; a < S; a++)
#pragma unroll
        for (int b = 0; b < S; b++) {
          const f32x4 r = rc[a * IW + b];
          f32x4 din;
#pragma unroll
          for (int e = 0; e < 4; e++) din[e] = r[e] > 0.f ? sacc[a * S + b][e] : 0.f;
          const int iy = S * oy + a, ix = S * ox + b;
          if (live && iy < H && ix < W) *(f32x4*)(dst + (((long)n * H + iy) * W + ix) * CQ + 4 * qC) = din + prev[a * S + b];
        }
    }
    // depthwise weight gradients of both stages: every wave walks all items for its quad with its own taps
#pragma unroll 1
    for (int it = 0; it < (need_w ? NITC : 0); it++) {
      const int p = it * 64 + ln;
      const int ly = p / TW, lx = p % TW;
      const f32x4* rp = rt + qC * IN + (S * ly) * IW + S * lx;
      const f32x4 dd0 = ddt0[qC * DN0 + (ly + HB0) * DW0 + lx + HB0], dd1 = ddt1[qC * DN1 + (ly + HB1) * DW1 + lx + HB1];
#pragma unroll
      for (int t = 0; t < NT0; t++) {
        const f32x4 v = rp[toff0[t]];
#pragma unroll
        for (int e = 0; e < 4; e++) acc0[t][e] = fmaf(dd0[e], v[e], acc0[t][e]);
      }
#pragma unroll
      for (int t = 0; t < NT1; t++) {
        const f32x4 v = rp[toff1[t]];
#pragma unroll
        for (int e = 0; e < 4; e++) acc1[t][e] = fmaf(dd1[e], v[e], acc1[t][e]);
      }
    }
    __syncthreads();
    // ---- phase D ----
    if (need_w) {
#pragma unroll
      for (int m = 0; m < MPW; m++) {
        const int pix = (wave * MPW + m) * (4 * NBK) + 4 * (n16 / CQ) + jg;
        accp0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Gs0[pix * CQ + n16 % CQ], Ds0[pix * CQ + n16 % CQ], accp0, 0, 0, 0);
        accp1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Gs1[pix * CQ + n16 % CQ], Ds1[pix * CQ + n16 % CQ], accp1, 0, 0, 0);
      }
    }
  }
  // ---- per-workgroup partial weight gradients (fixed order) ----
  __syncthreads();
  if (!need_w) return;
  float* wrow = wpart + (long)blockIdx.x * wtot;
  float* fin0 = (float*)smem;                                   // [NW][16][NT0*4], [NW][16][NT1*4], 2 x [NW][16][16]
  float* fin1 = fin0 + NW * 16 * NT0 * 4;
  float* finp = fin1 + NW * 16 * NT1 * 4;
#pragma unroll
  for (int t = 0; t < NT0; t++) {
    f32x4 v = acc0[t];
#pragma unroll
    for (int e = 0; e < 4; e++) { v[e] += __shfl_xor(v[e], 32, 64); v[e] += __shfl_xor(v[e], 16, 64); }
    if (lane < 16) *(f32x4*)(fin0 + ((wave * 16 + lane) * NT0 + t) * 4) = v;
  }
#pragma unroll
  for (int t = 0; t < NT1; t++) {
    f32x4 v = acc1[t];
#pragma unroll
    for (int e = 0; e < 4; e++) { v[e] += __shfl_xor(v[e], 32, 64); v[e] += __shfl_xor(v[e], 16, 64); }
    if (lane < 16) *(f32x4*)(fin1 + ((wave * 16 + lane) * NT1 + t) * 4) = v;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    finp[wave * 256 + (4 * jg + r) * 16 + n16] = accp0[r];
    finp[NW * 256 + wave * 256 + (4 * jg + r) * 16 + n16] = accp1[r];
  }
  __syncthreads();
  for (int ent = threadIdx.x; ent < CQ * (K0 * K0 + K1 * K1); ent += TPB) {
    const bool second = ent >= CQ * K0 * K0;
    const int KK = second ? K1 * K1 : K0 * K0, NT = second ? NT1 : NT0, en = second ? ent - CQ * K0 * K0 : ent;
    const float* fin = second ? fin1 : fin0;
    const int c = en / KK, tap = en % KK;
    const int w = c / 4 + NQ * (tap % RK), t = tap / RK;
    float a = 0.f;
    for (int l = 0; l < 16; l++) a += fin[((w * 16 + l) * NT + t) * 4 + c % 4];
    wrow[(second ? sb.dw_off : sa.dw_off) + en] = a;
  }
  for (int ent = threadIdx.x; ent < 2 * CQ * CQ; ent += TPB) {
    const bool second = ent >= CQ * CQ;
    const int en = second ? ent - CQ * CQ : ent;
    const int oc = en / CQ, c = en % CQ;
    const float* f2 = finp + (second ? NW * 256 : 0);
    float a = 0.f;
    for (int w = 0; w < NW; w++)
      for (int b = 0; b < NBK; b++) a += f2[w * 256 + (b * CQ + oc) * 16 + b * CQ + c];
    wrow[(second ? sb.pw_off : sa.pw_off) + en] = a;
  }
}

// out[col] = sum over the partial rows of the kernel that produced the column (fixed order, fp64).  Column sections: the
// depthwise gradients of conv 0..5 (9, 25, 9, 25, 9, 25 taps x Cq), their 1x1 gradients (Cq x Cq each), FactorizedReduce;
// nb.n[i] = workgroups of conv kernel i (i = 0..5), nb.n[6] = of the pooling kernel.
struct SectionRows { int n[7]; };
__global__ __launch_bounds__(64) void edge_wgrad_finish_kernel(const float* __restrict__ part, SectionRows nb, int wtot, int CQ,
                                                               float* __restrict__ out) {
  const int col = blockIdx.x;
  const int taps[6] = {9, 25, 9, 25, 9, 25};
  int sec = 6, o = 0;
  if (col < 102 * CQ) {
    for (int i = 0; i < 6; i++) { o += taps[i] * CQ; if (col < o) { sec = i; break; } }
  } else if (col < 102 * CQ + 6 * CQ * CQ) {
    sec = (col - 102 * CQ) / (CQ * CQ);
  }
  const int nblk = nb.n[sec];
  double a = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 64) a += (double)part[(long)k * wtot + col];
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) a += __shfl_xor(a, o2, 64);
  if (threadIdx.x == 0) out[col] = (float)a;
}

template <int CQ, int S, int TH, int TW>
size_t bwd_pools_lds() {
  constexpr int TPB = TH * TW, NQ = CQ / 4, XN = (TH + 4) * (TW + 4), GN = (TH / S + 2) * (TW / S + 2);
  size_t floats = (size_t)NQ * XN * 4 + 3 * NQ * GN * 4 + 6 * CQ + 3 * COEF + CQ * CQ + 3 * (TPB / 4) * CQ + TPB * CQ;
  return floats * 4 + 16;
}
template <int CQ, int K, int DIL, int S, int MODE, int TH, int TW>
size_t bwd_conv_lds() {
  constexpr int TPB = TH * TW, NQ = CQ / 4, KK = K * K, PAD = DIL * (K - 1) / 2, HB = (PAD + S - 1) / S;
  constexpr int IN = (S * (TH - 1) + 2 * PAD + 1) * (S * (TW - 1) + 2 * PAD + 1), DN = (TH + 2 * HB) * (TW + 2 * HB);
  size_t floats = (size_t)NQ * IN * 4 + NQ * DN * 4 + 2 * TPB * CQ + CQ * KK + CQ * CQ + 2 * CQ + COEF + 2 * CQ;
  if (MODE == 1) floats += 8 * (TPB + 1) + 128 + 1;
  return floats * 4 + (MODE == 1 ? 4 * CQ * 8 : 0) + 16;
}

template <int CQ, int K, int DIL, int S, int MODE, int TH, int TW, int NW>
size_t bwd_conv2_lds() {
  constexpr int NPX = TH * TW, NQ = CQ / 4, KK0 = K * K, RK = NW / NQ, KK = (KK0 + RK - 1) / RK, PAD = DIL * (K - 1) / 2, HB = (PAD + S - 1) / S;
  constexpr int IN = (S * (TH - 1) + 2 * PAD + 1) * (S * (TW - 1) + 2 * PAD + 1), DN = (TH + 2 * HB) * (TW + 2 * HB);
  const size_t tile = (size_t)NQ * IN * 4 + NQ * DN * 4 + 2 * NPX * CQ + 7 * CQ;
  const size_t fin = (size_t)NW * 16 * KK * 4 + NW * 256 + NW * 8;
  return (tile > fin ? tile : fin) * 4 + 16;
}

template <int CQ, int DIL, int S, int TH, int TW, int NW>
size_t bwd_conv2_pair_lds() {
  constexpr int NPX = TH * TW, NQ = CQ / 4, RK = NW / NQ, P0 = DIL, P1 = 2 * DIL, HB0 = (P0 + S - 1) / S, HB1 = (P1 + S - 1) / S;
  constexpr int IN = (S * (TH - 1) + 2 * P1 + 1) * (S * (TW - 1) + 2 * P1 + 1);
  constexpr int DN0 = (TH + 2 * HB0) * (TW + 2 * HB0), DN1 = (TH + 2 * HB1) * (TW + 2 * HB1), NT0 = (9 + RK - 1) / RK, NT1 = (25 + RK - 1) / RK;
  const size_t tile = (size_t)NQ * IN * 4 + NQ * (DN0 + DN1) * 4 + 4 * NPX * CQ + 10 * CQ;
  const size_t fin = (size_t)NW * 16 * (NT0 + NT1) * 4 + 2 * NW * 256;
  return (tile > fin ? tile : fin) * 4 + 16;
}

template <int CQ, int S, int TH, int TW>
size_t fwd_a2_lds() {
  constexpr int NPX = TH * TW, NQ = CQ / 4, IH = S * (TH - 1) + 9, IW = S * (TW - 1) + 9;
  size_t floats = (size_t)NQ * IH * IW * 4 + 5 * NPX * CQ + 2 * 3 * CQ * 8;
  return floats * 4 + 2 * 7 * CQ * 8 + 16;
}
template <int CQ, int TH, int TW>
size_t fwd_b2_lds() {
  constexpr int NPX = TH * TW, NQ = CQ / 4, IH = TH + 4, IW = TW + 4;
  size_t floats = (size_t)2 * NQ * IH * IW * 4 + 4 * NPX * CQ + 4 * CQ + 2 * 2 * CQ * 8;
  return floats * 4 + 4 * CQ * 8 + 16;
}

template <int CQ, int S, int TH, int TW>
size_t fwd_a_lds() {
  constexpr int TPB = TH * TW, NQ = CQ / 4, IH = S * (TH - 1) + 9, IW = S * (TW - 1) + 9;
  size_t floats = (size_t)NQ * IH * IW * 4 + CQ * 68 + 4 * CQ * CQ + CQ * CQ + 12 * (TPB + 1) + 2 * 12 * 8;
  return floats * 4 + 2 * 7 * CQ * 8 + 16;
}
template <int CQ, int TH, int TW>
size_t fwd_b_lds() {
  constexpr int TPB = TH * TW, NQ = CQ / 4, IH = TH + 4, IW = TW + 4;
  size_t floats = (size_t)2 * NQ * IH * IW * 4 + CQ * 34 + 2 * CQ * CQ + 4 * CQ + 8 * (TPB + 1) + 128;
  return floats * 4 + 4 * CQ * 8 + 16;
}

// Workgroups of `kern` one CU holds at once (registers, LDS); also opts the kernel in to its dynamic LDS size.  Cached per kernel.
int blocks_per_cu(const void* kern, int tpb, size_t lds) {
  static std::mutex mu;
  static std::unordered_map<const void*, int> cache;
  std::lock_guard<std::mutex> guard(mu);
  auto it = cache.find(kern);
  if (it != cache.end()) return it->second;
  (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, tpb, lds) != hipSuccess || n < 1) n = 1;
  if (n > 8) n = 8;
  cache[kern] = n;
  return n;
}
int cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n < 1) n = 256;
  }
  return n;
}
// Persistent grid of a tile kernel: one co-resident wave of workgroups (every CU filled to the kernel's own occupancy, so
// there is no partial second round), never more than there are tiles or partial-buffer rows.
int tile_grid(const void* kern, int tpb, size_t lds, long ntiles, int max_rows) {
  long g = (long)cu_count() * blocks_per_cu(kern, tpb, lds);
  if (g > ntiles) g = ntiles;
  if (g > max_rows) g = max_rows;
  return (int)(g < 1 ? 1 : g);
}
constexpr int MAX_ROWS = 2048;

}  // namespace

extern "C" {

#ifdef EDGE_PRE_BF16   // the shape queries are defined once, by the fp32 build of this file
int lctvqa_mixed_edge_supported(int B, int H, int W, int C, int stride);
int lctvqa_mixed_edge_grid(int B, int H, int W, int C, int stride);
int lctvqa_mixed_edge_wgrad_elems(int C);
long lctvqa_mixed_edge_bwd_scratch_floats(int B, int H, int W, int C, int stride);
#else
// 1 when the fused edge kernels take this shape (Cq in {4, 8, 16}; stride-2 edges need even H, W), else 0
int lctvqa_mixed_edge_supported(int B, int H, int W, int C, int stride) {
  const int CQ = C / 4;
  if (C % 16 != 0 || (CQ != 4 && CQ != 8 && CQ != 16)) return 0;
  if (stride == 2) return (H % 2 == 0 && W % 2 == 0 && H >= 2 && W >= 2) ? 1 : 0;
  return stride == 1 ? 1 : 0;
}
// rows of the per-workgroup partial buffers of an edge (an upper bound of every tile kernel's grid)
int lctvqa_mixed_edge_grid(int B, int H, int W, int C, int stride) {
  const int Ho = H / stride, Wo = W / stride, CQ = C / 4;
  const int TH = 8, TW = (stride == 2 && CQ == 16) ? 16 : 32;
  const long t_out = (long)B * ((Ho + TH - 1) / TH) * ((Wo + TW - 1) / TW);
  const long t_in = (long)B * ((H + 7) / 8) * ((W + 31) / 32);
  const long t = t_out > t_in ? t_out : t_in;
  return (int)(t > MAX_ROWS ? MAX_ROWS : t);
}
// floats of the flat weight-gradient vector: dw[6] (9, 25, 9, 25, 9, 25 taps x Cq), pw[6] (Cq x Cq), fr[2] (Cq/2 x Cq)
int lctvqa_mixed_edge_wgrad_elems(int C) {
  const int CQ = C / 4;
  return 102 * CQ + 7 * CQ * CQ;
}
long lctvqa_mixed_edge_bwd_scratch_floats(int B, int H, int W, int C, int stride) {
  const int CQ = C / 4;
  const long P = (long)B * (H / stride) * (W / stride);
  const long G = lctvqa_mixed_edge_grid(B, H, W, C, stride);
  return 3 * P * CQ + 2L * B * H * W * CQ + 1536L * (8 * CQ + 1) + 9 * COEF + 102 * CQ + G * lctvqa_mixed_edge_wgrad_elems(C) + 2 * (2 * G * 2 * CQ * 2) + 8;
}
#endif

// Forward of one edge.  w: 14 pointers = dw[6], pw[6], fr[2] (fr null on stride-1 edges); bn: 27 pointers = run_mean[9],
// run_var[9], tracked[9] (null for the identity skip); pre [9][P][Cq]; stats mean/invstd [9][Cq]; part: fp64 scratch of
// lctvqa_mixed_edge_grid * 7*Cq * 2 doubles; wsel [7] live architecture weights (PRIMITIVES order without 'none'); beta [1];
// out row stride ldo.
// (EDGE_PRE_BF16 build: lctvqa_mixed_edge_fwd_bf16pre / _bwd_bf16pre, `pre` = [9][P][Cq] bf16, everything else unchanged)
int EDGE_SYM(lctvqa_mixed_edge_fwd)(const float* x, int B, int H, int W, int C, int stride, const void* const* w, void* const* bn,
                          pre_t* pre, float* mean, float* invstd, double* part, const float* wsel, const float* beta,
                          float* out, long ldo, int accumulate, int training, float momentum, float eps, const float* x1d, void* stream) {
  LCT_REQUIRE(lctvqa_mixed_edge_supported(B, H, W, C, stride), "mixed_edge_fwd: unsupported shape B=%d H=%d W=%d C=%d s=%d", B, H, W, C, stride);
  LCT_REQUIRE((((uintptr_t)x | (uintptr_t)pre | (uintptr_t)out | (uintptr_t)mean | (uintptr_t)invstd) & 15) == 0 && (ldo & 3) == 0,
              "mixed_edge_fwd: tensors must be 16-B aligned");
  hipStream_t st = (hipStream_t)stream;
  EdgeW ew;
  EdgeBN eb;
  for (int i = 0; i < 6; i++) { ew.dw[i] = (const float*)w[i]; ew.pw[i] = (const float*)w[6 + i]; }
  ew.fr[0] = (const float*)w[12]; ew.fr[1] = (const float*)w[13];
  for (int i = 0; i < NSLOT; i++) { eb.run_mean[i] = (float*)bn[i]; eb.run_var[i] = (float*)bn[9 + i]; eb.tracked[i] = (long*)bn[18 + i]; }
  const int CQ = C / 4, Ho = H / stride, Wo = W / stride;
  const long P = (long)B * Ho * Wo;
  LCT_REQUIRE((P * CQ * (long)sizeof(pre_t)) % 16 == 0, "mixed_edge: a slot of the pre-BN buffer must be a multiple of 16 bytes (P=%ld Cq=%d)", P, CQ);
  const int rows = lctvqa_mixed_edge_grid(B, H, W, C, stride);
  const int skip_slot = stride == 1 ? 2 : -1;
  int rc, grid = 1;
#define TILES(THv, TWv) ((long)B * ((Ho + (THv) - 1) / (THv)) * ((Wo + (TWv) - 1) / (TWv)))
#define EDGE_A(CQv, Sv, THv, TWv)                                                                                          \
  {                                                                                                                        \
    const size_t lds = fwd_a_lds<CQv, Sv, THv, TWv>();                                                                     \
    grid = tile_grid((const void*)edge_fwd_a_kernel<CQv, Sv, THv, TWv>, THv * TWv, lds, TILES(THv, TWv), rows);           \
    hipLaunchKernelGGL((edge_fwd_a_kernel<CQv, Sv, THv, TWv>), dim3(grid), dim3(THv * TWv), lds, st, x1d ? x1d : x, (long)(x1d ? CQv : C), B, H, W, ew, pre, part); \
  }
#define EDGE_B(CQv, THv, TWv)                                                                                              \
  {                                                                                                                        \
    const size_t lds = fwd_b_lds<CQv, THv, TWv>();                                                                         \
    grid = tile_grid((const void*)edge_fwd_b_kernel<CQv, THv, TWv>, THv * TWv, lds, TILES(THv, TWv), rows);               \
    hipLaunchKernelGGL((edge_fwd_b_kernel<CQv, THv, TWv>), dim3(grid), dim3(THv * TWv), lds, st, B, Ho, Wo, ew, mean, invstd, pre, part); \
  }
  // Cq = 8, 16: the matrix-core / scalar-weight form (bit-identical results); Cq = 4 moves bytes at the HBM roof in the first form
  float* wq = (float*)(part + (long)rows * 7 * CQ * 2);         // strided-quad tap-major depthwise weights (edge_fwd_prep_kernel)
#define EDGE_A2(CQv, Sv, THv, TWv, NWv)                                                                                    \
  {                                                                                                                        \
    const size_t lds = fwd_a2_lds<CQv, Sv, THv, TWv>();                                                                    \
    grid = tile_grid((const void*)edge_fwd_a2_kernel<CQv, Sv, THv, TWv, NWv>, 64 * NWv, lds, TILES(THv, TWv), rows);      \
    hipLaunchKernelGGL((edge_fwd_a2_kernel<CQv, Sv, THv, TWv, NWv>), dim3(grid), dim3(64 * NWv), lds, st, x1d ? x1d : x, (long)(x1d ? CQv : C), B, H, W, ew, wq, pre, part); \
  }
#define EDGE_B2(CQv, THv, TWv, NWv)                                                                                        \
  {                                                                                                                        \
    const size_t lds = fwd_b2_lds<CQv, THv, TWv>();                                                                        \
    grid = tile_grid((const void*)edge_fwd_b2_kernel<CQv, THv, TWv, NWv>, 64 * NWv, lds, TILES(THv, TWv), rows);          \
    hipLaunchKernelGGL((edge_fwd_b2_kernel<CQv, THv, TWv, NWv>), dim3(grid), dim3(64 * NWv), lds, st, B, Ho, Wo, ew, wq, mean, invstd, pre, part); \
  }
  // measured (tools/ab_edge.py, B=64): the stride-2 edges gain 14 % / 28 % of their forward (Cq = 8 / 16); at stride 1 the first form's
  // two or three workgroups per CU hide its barriers better than the new form's one (-DEDGE_FWD2_S1 runs it there on half tiles)
#ifdef EDGE_FWD_V1
  const bool v2 = false;
#elif defined(EDGE_FWD2_S1)
  const bool v2 = CQ >= 8;
#else
  const bool v2 = CQ >= 8 && stride == 2;
#endif
  if (v2) {
    hipLaunchKernelGGL(edge_fwd_prep_kernel, dim3(1), dim3(256), 0, st, ew, CQ, wq);
    if ((rc = lct_check_launch("edge_fwd_prep"))) return rc;
  }
  if (CQ == 4 && stride == 1) EDGE_A(4, 1, 8, 32)
  else if (CQ == 4) EDGE_A(4, 2, 8, 32)
  else if (v2 && CQ == 8 && stride == 1) EDGE_A2(8, 1, 8, 32, 4)
  else if (v2 && CQ == 8) EDGE_A2(8, 2, 8, 32, 8)
  else if (v2 && stride == 1) EDGE_A2(16, 1, 8, 16, 4)
  else if (v2) EDGE_A2(16, 2, 8, 16, 8)
  else if (CQ == 8 && stride == 1) EDGE_A(8, 1, 8, 32)
  else if (CQ == 8) EDGE_A(8, 2, 8, 32)
  else if (stride == 1) EDGE_A(16, 1, 8, 32)
  else EDGE_A(16, 2, 8, 16)
  if ((rc = lct_check_launch("edge_fwd_a"))) return rc;
  hipLaunchKernelGGL(edge_stats_finish_kernel, dim3(7 * CQ), dim3(64), 0, st, part, grid, 7, 0, CQ, P, eps, momentum, training, eb,
                     skip_slot, mean, invstd);
  if ((rc = lct_check_launch("edge_stats_finish(a)"))) return rc;
  const bool narrow = (stride == 2 && CQ == 16);
  if (CQ == 4) EDGE_B(4, 8, 32)
  else if (v2 && CQ == 8 && stride == 1) EDGE_B2(8, 8, 32, 4)
  else if (v2 && CQ == 8) EDGE_B2(8, 8, 32, 8)
  else if (v2 && !narrow) EDGE_B2(16, 8, 16, 4)
  else if (v2) EDGE_B2(16, 8, 16, 8)
  else if (CQ == 8) EDGE_B(8, 8, 32)
  else if (!narrow) EDGE_B(16, 8, 32)
  else EDGE_B(16, 8, 16)
  if ((rc = lct_check_launch("edge_fwd_b"))) return rc;
  hipLaunchKernelGGL(edge_stats_finish_kernel, dim3(2 * CQ), dim3(64), 0, st, part, grid, 2, 7, CQ, P, eps, momentum, training, eb, -1,
                     mean, invstd);
  if ((rc = lct_check_launch("edge_stats_finish(b)"))) return rc;
  long nb = (P * (CQ / 4) + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (stride == 1)
    hipLaunchKernelGGL(edge_fwd_c_kernel<1>, dim3((int)nb), dim3(256), 0, st, x, pre, mean, invstd, wsel, beta, B, Ho, Wo, CQ, out, ldo, accumulate);
  else
    hipLaunchKernelGGL(edge_fwd_c_kernel<2>, dim3((int)nb), dim3(256), 0, st, x, pre, mean, invstd, wsel, beta, B, Ho, Wo, CQ, out, ldo, accumulate);
  return lct_check_launch("edge_fwd_c");
#undef EDGE_A
#undef EDGE_B
#undef EDGE_A2
#undef EDGE_B2
#undef TILES
}

// Backward of one edge (see the block comment above edge_bwd_sums_kernel).  dx [B,H,W,C] is written completely (accumulate_dx = 0) or
// added to (1: the gradient buffer of a state that several edges read); dwsel [7],
// dbeta [1]; wgrad = flat weight gradients (lctvqa_mixed_edge_wgrad_elems, order dw[6], pw[6], fr[2]); scratch >=
// lctvqa_mixed_edge_bwd_scratch_floats floats, 16-byte aligned.
int EDGE_SYM(lctvqa_mixed_edge_bwd)(const float* dout, long ldd, const float* x, int B, int H, int W, int C, int stride, const void* const* w,
                          const pre_t* pre, const float* mean, const float* invstd, const float* wsel, const float* beta, float* dx,
                          float* dwsel, float* dbeta, float* wgrad, float* scratch, int training, int accumulate_dx, const float* x1d_in,
                          int need_wgrad, void* stream) {
  LCT_REQUIRE(lctvqa_mixed_edge_supported(B, H, W, C, stride), "mixed_edge_bwd: unsupported shape B=%d H=%d W=%d C=%d s=%d", B, H, W, C, stride);
  LCT_REQUIRE((((uintptr_t)dout | (uintptr_t)x | (uintptr_t)pre | (uintptr_t)dx | (uintptr_t)scratch | (uintptr_t)wgrad) & 15) == 0 && (ldd & 3) == 0,
              "mixed_edge_bwd: tensors must be 16-B aligned");
  hipStream_t st = (hipStream_t)stream;
  EdgeW ew;
  for (int i = 0; i < 6; i++) { ew.dw[i] = (const float*)w[i]; ew.pw[i] = (const float*)w[6 + i]; }
  ew.fr[0] = (const float*)w[12]; ew.fr[1] = (const float*)w[13];
  const int CQ = C / 4, Ho = H / stride, Wo = W / stride;
  const long P = (long)B * Ho * Wo;
  LCT_REQUIRE((P * CQ * (long)sizeof(pre_t)) % 16 == 0, "mixed_edge: a slot of the pre-BN buffer must be a multiple of 16 bytes (P=%ld Cq=%d)", P, CQ);
  const int G = lctvqa_mixed_edge_grid(B, H, W, C, stride);
  const int wtot = lctvqa_mixed_edge_wgrad_elems(C);
  int dw_off[6], pw_off[6];
  { const int taps[6] = {9, 25, 9, 25, 9, 25}; int o = 0; for (int i = 0; i < 6; i++) { dw_off[i] = o; o += taps[i] * CQ; } for (int i = 0; i < 6; i++) pw_off[i] = o + i * CQ * CQ; }
  const int fr_off = 102 * CQ + 6 * CQ * CQ;
  float* dY = scratch;
  float* dBN = dY + P * CQ;                                    // [2][P][CQ]
  float* dx1c = dBN + 2 * P * CQ;                              // [B][H][W][CQ]: what the conv stages send to x1
  float* x1s = dx1c + (long)B * H * W * CQ;                    // [B][H][W][CQ]: dense x[..., :CQ] (a 16..64-byte slice of every NHWC row of x otherwise),
  const float* x1d = x1d_in ? x1d_in : x1s;                    // written by the sums pass unless the caller kept the forward's copy
  float* part1 = x1s + (long)B * H * W * CQ;                   // [<=1536][8*CQ+1]
  float* coef = part1 + 1536L * (8 * CQ + 1);
  coef += (4 - ((coef - scratch) & 3)) & 3;
  float* wtaps = coef + 9 * COEF;                              // tap-major copies of the depthwise weights (16-byte aligned), offsets dw_off
  float* wpart = wtaps + 102 * CQ;                             // [G][wtot]
  float* sp = wpart + (long)G * wtot;
  sp += ((uintptr_t)sp & 7) ? 1 : 0;
  double* spart = (double*)sp;                                 // [2][G][2*CQ][2]
  SectionRows rows;
  for (int i = 0; i < 7; i++) rows.n[i] = 0;
  int rc, grid = 1, g3b = 1, g5b = 1;
  {
    long nb = (P * (CQ / 4) + 255) / 256;
    if (nb > 1536) nb = 1536;                                    // six workgroups per CU hide the load latency; one partial row each
    if (stride == 1)
      hipLaunchKernelGGL(edge_bwd_sums_kernel<1>, dim3((int)nb), dim3(256), 0, st, dout, ldd, x, pre, mean, invstd, beta, B, Ho, Wo, CQ, dY, part1, x1d_in ? nullptr : x1s);
    else
      hipLaunchKernelGGL(edge_bwd_sums_kernel<2>, dim3((int)nb), dim3(256), 0, st, dout, ldd, x, pre, mean, invstd, beta, B, Ho, Wo, CQ, dY, part1, x1d_in ? nullptr : x1s);
    if ((rc = lct_check_launch("edge_bwd_sums"))) return rc;
    hipLaunchKernelGGL(edge_bwd_coef_kernel, dim3(1), dim3(1024), 0, st, part1, (int)nb, CQ, P, stride, training, wsel, beta, invstd, coef, dwsel, dbeta, ew, wtaps);
    if ((rc = lct_check_launch("edge_bwd_coef"))) return rc;
  }
#define TILES(Hv, Wv, THv, TWv) ((long)B * (((Hv) + (THv) - 1) / (THv)) * (((Wv) + (TWv) - 1) / (TWv)))
  // CONV(CQ, K, DIL, S, MODE, TH, TW, conv index, slot of its output BN, input tensor, ldin, H, W, gradient source, dst, accumulate, spart)
#ifndef EDGE_CONV_V1
#ifndef EDGE_NW4
#define EDGE_NW4 4
#endif
#ifndef EDGE_NW8
#define EDGE_NW8 8
#endif
#ifndef EDGE_NW16
#define EDGE_NW16 8
#endif
#ifndef EDGE_WPE
#define EDGE_WPE(CQv, Kv) ((CQv) == 16 ? 2 : ((Kv) == 5 ? 3 : 4))
#endif
#define EDGE_NW(CQv, THv, TWv) ((CQv) == 16 ? EDGE_NW16 : ((CQv) == 8 ? EDGE_NW8 : EDGE_NW4))
#define CONV(CQv, Kv, DILv, Sv, MODEv, THv, TWv, ci, slot, inp, ldv, Hv, Wv, gs, dstp, accv, spp)                           \
  {                                                                                                                        \
    constexpr int NWv = EDGE_NW(CQv, THv, TWv);                                                                            \
    const size_t lds = bwd_conv2_lds<CQv, Kv, DILv, Sv, MODEv, THv, TWv, NWv>();                                           \
    grid = tile_grid((const void*)edge_bwd_conv2_kernel<CQv, Kv, DILv, Sv, MODEv, THv, TWv, NWv, EDGE_WPE(CQv, Kv)>, 64 * NWv, lds, \
                     TILES((Hv) / Sv, (Wv) / Sv, THv, TWv), G);                                                            \
    rows.n[ci] = grid;                                                                                                     \
    hipLaunchKernelGGL((edge_bwd_conv2_kernel<CQv, Kv, DILv, Sv, MODEv, THv, TWv, NWv, EDGE_WPE(CQv, Kv)>), dim3(grid), dim3(64 * NWv), lds, st, inp, (long)(ldv), gs, \
                       pre + (long)(slot) * P * CQ, mean + (slot) * CQ, invstd + (slot) * CQ, coef + (slot) * COEF,       \
                       mean + ((slot) - 4) * CQ, invstd + ((slot) - 4) * CQ, wtaps + dw_off[ci], ew.pw[ci], B, Hv, Wv, dstp, accv, spp, wpart, wtot, \
                       dw_off[ci], pw_off[ci], need_wgrad);                                                                \
  }
#else
#define CONV(CQv, Kv, DILv, Sv, MODEv, THv, TWv, ci, slot, inp, ldv, Hv, Wv, gs, dstp, accv, spp)                           \
  {                                                                                                                        \
    const size_t lds = bwd_conv_lds<CQv, Kv, DILv, Sv, MODEv, THv, TWv>();                                                 \
    grid = tile_grid((const void*)edge_bwd_conv_kernel<CQv, Kv, DILv, Sv, MODEv, THv, TWv>, THv * TWv, lds,               \
                     TILES((Hv) / Sv, (Wv) / Sv, THv, TWv), G);                                                            \
    rows.n[ci] = grid;                                                                                                     \
    hipLaunchKernelGGL((edge_bwd_conv_kernel<CQv, Kv, DILv, Sv, MODEv, THv, TWv>), dim3(grid), dim3(THv * TWv), lds, st, inp, (long)(ldv), gs, \
                       pre + (long)(slot) * P * CQ, mean + (slot) * CQ, invstd + (slot) * CQ, coef + (slot) * COEF,       \
                       mean + ((slot) - 4) * CQ, invstd + ((slot) - 4) * CQ, ew.dw[ci], ew.pw[ci], B, Hv, Wv, dstp, accv, spp, wpart, wtot, \
                       dw_off[ci], pw_off[ci]);                                                                            \
  }
#endif
  // PAIR(CQ, DIL, S, MODE, TH, TW, conv index of the 3x3 / of the 5x5 stage, their output-BN slots, their gradient sources, accumulate)
#define STAGE_ARGS(ci, slot, gs) StageArgs{gs, pre + (long)(slot) * P * CQ, mean + (slot) * CQ, invstd + (slot) * CQ, coef + (slot) * COEF, \
                                           wtaps + dw_off[ci], ew.pw[ci], dw_off[ci], pw_off[ci]}
#define PAIR(CQv, DILv, Sv, MODEv, THv, TWv, c0, c1, s0, s1, gs0, gs1, accv)                                                \
  {                                                                                                                        \
    constexpr int NWv = EDGE_NW(CQv, THv, TWv);                                                                            \
    const size_t lds = bwd_conv2_pair_lds<CQv, DILv, Sv, THv, TWv, NWv>();                                                 \
    grid = tile_grid((const void*)edge_bwd_conv2_pair_kernel<CQv, DILv, Sv, MODEv, THv, TWv, NWv, 2>, 64 * NWv, lds,      \
                     TILES(H / Sv, W / Sv, THv, TWv), G);                                                                  \
    rows.n[c0] = grid; rows.n[c1] = grid;                                                                                  \
    hipLaunchKernelGGL((edge_bwd_conv2_pair_kernel<CQv, DILv, Sv, MODEv, THv, TWv, NWv, 2>), dim3(grid), dim3(64 * NWv), lds, st, x1d, (long)CQv, \
                       STAGE_ARGS(c0, s0, gs0), STAGE_ARGS(c1, s1, gs1), B, H, W, dx1c, accv, wpart, wtot, need_wgrad);    \
  }
#define DIL_SINGLES(CQv, Sv, THv, TWv)                                                                                     \
    CONV(CQv, 3, 2, Sv, 0, THv, TWv, 2, 5, x1d, CQv, H, W, dY, dx1c, 0, (double*)nullptr)                                  \
    if ((rc = lct_check_launch("edge_bwd_conv(dil3)"))) return rc;                                                         \
    CONV(CQv, 5, 2, Sv, 0, THv, TWv, 3, 6, x1d, CQv, H, W, dY, dx1c, 1, (double*)nullptr)                                  \
    if ((rc = lct_check_launch("edge_bwd_conv(dil5)"))) return rc;
#define SEPA_SINGLES(CQv, Sv, THv, TWv)                                                                                    \
    CONV(CQv, 3, 1, Sv, 2, THv, TWv, 0, 3, x1d, CQv, H, W, dBN, dx1c, 1, (double*)nullptr)                                 \
    if ((rc = lct_check_launch("edge_bwd_conv(sep3a)"))) return rc;                                                        \
    CONV(CQv, 5, 1, Sv, 2, THv, TWv, 1, 4, x1d, CQv, H, W, dBN + P * CQv, dx1c, 1, (double*)nullptr)                       \
    if ((rc = lct_check_launch("edge_bwd_conv(sep5a)"))) return rc;
#define DIL_PAIR(CQv, Sv, THv, TWv)                                                                                        \
    PAIR(CQv, 2, Sv, 0, THv, TWv, 2, 3, 5, 6, dY, dY, 0)                                                                   \
    if ((rc = lct_check_launch("edge_bwd_conv_pair(dil)"))) return rc;
#define SEPA_PAIR(CQv, Sv, THv, TWv)                                                                                       \
    PAIR(CQv, 1, Sv, 2, THv, TWv, 0, 1, 3, 4, dBN, dBN + P * CQv, 1)                                                       \
    if ((rc = lct_check_launch("edge_bwd_conv_pair(sep.a)"))) return rc;
  // the pair form at Cq = 8 (measured: -6 % of the stride-2 edge's backward, -2 % stride 1); at Cq = 4 its 196 registers take the third wave per
  // SIMD that the single stages have (+3 %), at Cq = 16 it spills and the dil pair's tiles do not fit the LDS
#ifdef EDGE_NO_PAIR
#define DIL_STAGES_4 DIL_SINGLES
#define DIL_STAGES_8 DIL_SINGLES
#define SEPA_STAGES_4 SEPA_SINGLES
#define SEPA_STAGES_8 SEPA_SINGLES
#else
#define DIL_STAGES_4 DIL_SINGLES
#define DIL_STAGES_8 DIL_PAIR
#define SEPA_STAGES_4 SEPA_SINGLES
#define SEPA_STAGES_8 SEPA_PAIR
#endif
#ifdef EDGE_PAIR16
#define DIL_STAGES_16 DIL_PAIR
#define SEPA_STAGES_16 SEPA_PAIR
#else
#define DIL_STAGES_16 DIL_SINGLES
#define SEPA_STAGES_16 SEPA_SINGLES
#endif
#define DIL_STAGES(CQv, Sv, THv, TWv) DIL_STAGES_##CQv(CQv, Sv, THv, TWv)
#define SEPA_STAGES(CQv, Sv, THv, TWv) SEPA_STAGES_##CQv(CQv, Sv, THv, TWv)
#define STAGES(CQv, Sv, THv, TWv)                                                                                          \
  {                                                                                                                        \
    DIL_STAGES(CQv, Sv, THv, TWv)                                                                                          \
    CONV(CQv, 3, 1, 1, 1, 8, 32, 4, 7, pre + 3L * P * CQv, CQv, Ho, Wo, dY, dBN, 0, spart)                                 \
    g3b = grid;                                                                                                            \
    if ((rc = lct_check_launch("edge_bwd_conv(sep3b)"))) return rc;                                                        \
    CONV(CQv, 5, 1, 1, 1, 8, 32, 5, 8, pre + 4L * P * CQv, CQv, Ho, Wo, dY, dBN + P * CQv, 0, spart + (long)G * 2 * CQv * 2) \
    g5b = grid;                                                                                                            \
    if ((rc = lct_check_launch("edge_bwd_conv(sep5b)"))) return rc;                                                        \
    hipLaunchKernelGGL(edge_bwd_coef_a_kernel, dim3(2 * CQv), dim3(64), 0, st, spart, G, g3b, g5b, CQv, P, training, invstd, coef); \
    if ((rc = lct_check_launch("edge_bwd_coef_a"))) return rc;                                                             \
    SEPA_STAGES(CQv, Sv, THv, TWv)                                                                                         \
    {                                                                                                                      \
      const size_t lds = bwd_pools_lds<CQv, Sv, 8, 32>();                                                                  \
      grid = tile_grid((const void*)edge_bwd_pools_kernel<CQv, Sv, 8, 32>, 256, lds, TILES(H, W, 8, 32), G);               \
      rows.n[6] = grid;                                                                                                    \
      hipLaunchKernelGGL((edge_bwd_pools_kernel<CQv, Sv, 8, 32>), dim3(grid), dim3(256), lds, st, x, dY, dout, ldd, beta, pre, mean, invstd, \
                         coef, dx1c, B, H, W, ew, dx, wpart, wtot, fr_off, Sv == 1 ? x1d : x, (long)(Sv == 1 ? CQv : C), accumulate_dx); \
      if ((rc = lct_check_launch("edge_bwd_pools"))) return rc;                                                            \
    }                                                                                                                      \
  }
  if (CQ == 4 && stride == 1) STAGES(4, 1, 8, 32)
  else if (CQ == 4) STAGES(4, 2, 8, 32)
  else if (CQ == 8 && stride == 1) STAGES(8, 1, 8, 32)
  else if (CQ == 8) STAGES(8, 2, 8, 32)
  else if (stride == 1) STAGES(16, 1, 8, 32)
  else STAGES(16, 2, 8, 16)
  if (!need_wgrad) return LCT_OK;                              // (the architects' finite-difference passes: gradients of alpha / beta only)
  const int ncols = stride == 2 ? wtot : fr_off;               // stride-1 edges have no FactorizedReduce
  hipLaunchKernelGGL(edge_wgrad_finish_kernel, dim3(ncols), dim3(64), 0, st, wpart, rows, wtot, CQ, wgrad);
  return lct_check_launch("edge_wgrad_finish");
#undef STAGES
#undef DIL_STAGES
#undef SEPA_STAGES
#undef PAIR
#undef STAGE_ARGS
#undef CONV
#undef TILES
}

}  // extern "C"
