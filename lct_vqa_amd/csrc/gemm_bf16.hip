// bf16 GEMM with fp32 accumulation on v_mfma_f32_32x32x16_bf16, "NT" form: C[M][N] = epi( sum_k A[m][k] * B[n][k] ),
// A [M][K] and B [N][K] bf16 row-major (K contiguous: activations as NHWC rows, weights as stored), and the small bf16 helpers of
// the bf16 ResNet trunk (BASELINE.json configs[2] in bf16; the reference has neither ResNet nor bf16).
//   epi(v) = act( v + bias[n] + addend[m][n] ), output bf16 or fp32 (fp32 + in-place addend = accumulation over several calls).
// Structure = the bf16 convolution kernel (conv3x3_bf16.hip) without taps and halo: workgroup tile 256 x 128 (or 64), 2 x 2 waves
// of 4 x 2 (or 4 x 1) 32x32 accumulator tiles, K in steps of 32 (four 8-element planes of 16-byte LDS slots, lane half h of
// k16-step kk reads plane 2kk + h), operand tiles by LDS-DMA, double buffered, one barrier per step, two workgroups per CU.
// Rows beyond M read a zero page and are not stored.  Roofline: MFMA for K >= ~512, HBM below (1x1 convolutions with 64..256
// input channels move (K + N) * 2 bytes per row for 2*K*N flops).
#include "common.h"
#include "gemm_epi.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __attribute__((aligned(64))) float g_zero_page_gb[64];

__device__ __forceinline__ unsigned short f2bf(float v) {
  const __bf16 b = (__bf16)v;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned int)u << 16); }

template <int NT2, bool OUT_BF16>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const unsigned short* __restrict__ A, long lda, const unsigned short* __restrict__ B,
                                                           long ldb, void* __restrict__ Cv, long ldc, int M, int N, int K,
                                                           const float* __restrict__ bias, const void* __restrict__ addv, long ld_add, int act) {
  constexpr int BM = 256, BN = 64 * NT2, MT = 4;
  constexpr int APL = BM * 4, BPL = BN * 4;                      // floats per 8-element plane of the A / B step tile
  constexpr int ASZ = 4 * APL, BSZ = 4 * BPL;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* abuf = smem;                                           // [2][ASZ]
  float* bbuf = smem + 2 * ASZ;                                 // [2][BSZ]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int j = lane & 31, h = lane >> 5;
  const int NTL = N / BN;
  const int n_tile = blockIdx.x % NTL, m_tile = blockIdx.x / NTL;
  const long m0 = (long)m_tile * BM;
  const int n0 = n_tile * BN;
  const int S = K / 32;
  // Operand tiles in LDS as rows of 64 B (32 bf16 = four 16-byte slots), the slot XOR-swizzled by (row>>2)&3 (conflict-free
  // ds_read_b128 of 32 consecutive rows).  The DMA writes LDS linearly, so the swizzle is applied to the source: lane = (row lane>>2,
  // slot lane&3) fetches slot (lane&3) ^ ((row>>2)&3); the four lanes of a row cover its 64 contiguous bytes and an instruction
  // touches 16 cache lines (the planar form -- lane = row, one 16-byte slot per instruction -- touched 64).
  const int rl = lane >> 2, sq = (lane & 3) ^ ((rl >> 2) & 3);
  long aoff[4];                                                 // element offset of this lane's slot in rows 16 (wave + 4 r) + rl (or -1)
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const long row = m0 + 16 * (wave + 4 * r) + rl;
    aoff[r] = row < M ? row * lda + sq * 8 : -1;
  }
  long boff[NT2];
#pragma unroll
  for (int r = 0; r < NT2; r++) boff[r] = (long)(n0 + 16 * (wave + 4 * r) + rl) * ldb + sq * 8;
  auto stage = [&](int s, float* ad, float* bd) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float* src = aoff[r] >= 0 ? (const float*)(A + aoff[r] + (long)s * 32) : g_zero_page_gb;
      glds16(src, ad + (wave + 4 * r) * 256);
    }
#pragma unroll
    for (int r = 0; r < NT2; r++) glds16((const float*)(B + boff[r] + (long)s * 32), bd + (wave + 4 * r) * 256);
  };
  f32x16 acc[MT][NT2];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int nn = 0; nn < NT2; nn++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][nn][r] = 0.f;
  stage(0, abuf, bbuf);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int swz = (j >> 2) & 3;
  const int fo[2] = {((0 + h) ^ swz) << 2, ((2 + h) ^ swz) << 2};  // swizzled slot of half-step kk = 0, 1 (floats)
  const int abase = (wm * MT * 32 + j) * 16;
  const int bbase = (wn * NT2 * 32 + j) * 16;
  for (int s = 0; s < S; s++) {
    if (s + 1 < S) stage(s + 1, abuf + ((s + 1) & 1) * ASZ, bbuf + ((s + 1) & 1) * BSZ);
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
  if (OUT_BF16) {
    // bf16 output: every 32-row x (32*NT2)-column block of a wave goes through a private LDS tile so that a lane stores 32
    // consecutive bytes per 16 columns (whole 64..128-byte row segments per store instead of 2-byte scatters); bias, addend
    // and activation are applied on the row segments, the addend read as 16-byte vectors too.
    constexpr int TC = 32 * NT2, TS = TC + 1;                      // tile columns, padded row stride (floats)
    constexpr int LPP = TC / 8, PPI = 64 / LPP;                    // lanes per row (16 B each), rows per store instruction
    float* T = smem + wave * 32 * TS;
    const int tq = lane % LPP, tp = lane / LPP;                    // a store instruction writes PPI whole 64..128-byte row segments
    const int col0 = n0 + wn * TC + 8 * tq;
    float bvv[8];
#pragma unroll
    for (int e = 0; e < 8; e++) bvv[e] = bias ? bias[col0 + e] : 0.f;
#pragma unroll
    for (int m = 0; m < MT; m++) {
      __syncthreads();                                            // operand buffers / the previous block's read-back are done
#pragma unroll
      for (int nn = 0; nn < NT2; nn++)
#pragma unroll
        for (int r = 0; r < 16; r++) T[((r & 3) + 8 * (r >> 2) + 4 * h) * TS + nn * 32 + j] = acc[m][nn][r];
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 32 / PPI; c++) {
        typedef unsigned short us8 __attribute__((ext_vector_type(8)));
        const int pr = tp + PPI * c;
        const long row = m0 + wm * 128 + m * 32 + pr;
        if (row < M) {
          us8 av = {0, 0, 0, 0, 0, 0, 0, 0};
          if (addv) av = *(const us8*)((const unsigned short*)addv + row * ld_add + col0);
          us8 ov;
#pragma unroll
          for (int e = 0; e < 8; e++) {
            float v = T[pr * TS + 8 * tq + e] + bvv[e];
            if (addv) v += bf2f(av[e]);
            if (act == 1) v = fmaxf(v, 0.f);
            ov[e] = f2bf(v);
          }
          *(us8*)((unsigned short*)Cv + row * ldc + col0) = ov;
        }
      }
    }
  } else {
#pragma unroll
    for (int nn = 0; nn < NT2; nn++) {
      const int col = n0 + (wn * NT2 + nn) * 32 + j;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const long row = m0 + wm * 128 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < M) {
            float v = acc[m][nn][r] + bv;
            if (addv) v += ((const float*)addv)[row * ld_add + col];
            if (act == 1) v = fmaxf(v, 0.f);
            ((float*)Cv)[row * ldc + col] = v;
          }
        }
    }
  }
}


// ---- the general form behind every nn.Linear of the heads under the bf16 compute dtype (fp32 master weights converted per call,
// fp32 accumulation, fp32 output): ragged N (a store guard; the bf16 B copy is padded to the tile), split-K into fp32 slabs
// (blockIdx.z), the full fp32 epilogue descriptor (bias / addend / ReLU / tanh / dropout mask), and two epilogues that fuse the
// LCT token head (reference basic_vqa/models_lct.py:118-120, Linear(H, QV) over all tokens) with its cross entropy
// (models_lct.py:253-255) so that the [T*B, QV] logits never exist in HBM:
//   MODE 1  per-row (max, sum exp) partials of every 64-column chunk + the label's logit           (forward)
//   MODE 2  recompute the tile, dl = (softmax - onehot) * g / rows, stored as bf16 row-major AND transposed (the operands of
//           the two backward products), and per-128-row partial column sums (the bias gradient)   (backward)
struct Bf16Job {
  const unsigned short* A; long lda;       // [M][>= K]  bf16, K-contiguous, zero beyond the logical K
  const unsigned short* B; long ldb;       // [Np][>= K] bf16 (rows >= Nvalid: anything finite or not -- never stored)
  int M, Nvalid, Np, K, kper;              // K % 32 == 0, kper % 32 == 0: split z covers [z*kper, min(K, (z+1)*kper))
  float* C; long ldc; int direct;          // direct: C[m][n] = epi(.), else slab z of [splits][M][Nvalid]
  GemmEpi epi;
  // token head + cross entropy
  const long* labels;                      // [M] int64
  const float* lse;                        // [M]                        (MODE 2)
  float* pmax; float* psum;                // [Np/64][M]                 (MODE 1)
  float* picked;                           // [M] logit of the label     (MODE 1)
  const float* gscale; float inv_rows;     // dl scale = *gscale * inv_rows (MODE 2)
  unsigned short* dl; long ld_dl;          // [M][Np] bf16               (MODE 2)
  unsigned short* dlT; long ld_dlT;        // [Np][>= Mpad] bf16         (MODE 2)
  float* dbpart;                           // [ceil(M/128)][Np]          (MODE 2)
};

template <int NT2, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_bf16x_kernel(const Bf16Job job) {
  constexpr int BM = 256, BN = 64 * NT2, MT = 4;
  constexpr int APL = BM * 4, BPL = BN * 4;
  constexpr int ASZ = 4 * APL, BSZ = 4 * BPL;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* abuf = smem;
  float* bbuf = smem + 2 * ASZ;
  const unsigned short* __restrict__ A = job.A;
  const unsigned short* __restrict__ B = job.B;
  const long lda = job.lda, ldb = job.ldb;
  const int M = job.M, Nvalid = job.Nvalid;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int j = lane & 31, h = lane >> 5;
  const int NTL = job.Np / BN;
  const int n_tile = blockIdx.x % NTL, m_tile = blockIdx.x / NTL;
  const long m0 = (long)m_tile * BM;
  const int n0 = n_tile * BN;
  const int kbeg = blockIdx.z * job.kper;
  const int kend = min(job.K, kbeg + job.kper);
  const int S = (kend - kbeg) / 32;
  const int rl = lane >> 2, sq = (lane & 3) ^ ((rl >> 2) & 3);
  long aoff[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const long row = m0 + 16 * (wave + 4 * r) + rl;
    aoff[r] = row < M ? row * lda + sq * 8 + kbeg : -1;
  }
  long boff[NT2];
#pragma unroll
  for (int r = 0; r < NT2; r++) boff[r] = (long)(n0 + 16 * (wave + 4 * r) + rl) * ldb + sq * 8 + kbeg;
  auto stage = [&](int s, float* ad, float* bd) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float* src = aoff[r] >= 0 ? (const float*)(A + aoff[r] + (long)s * 32) : g_zero_page_gb;
      glds16(src, ad + (wave + 4 * r) * 256);
    }
#pragma unroll
    for (int r = 0; r < NT2; r++) glds16((const float*)(B + boff[r] + (long)s * 32), bd + (wave + 4 * r) * 256);
  };
  f32x16 acc[MT][NT2];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int nn = 0; nn < NT2; nn++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][nn][r] = 0.f;
  if (S > 0) stage(0, abuf, bbuf);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int swz = (j >> 2) & 3;
  const int fo[2] = {((0 + h) ^ swz) << 2, ((2 + h) ^ swz) << 2};
  const int abase = (wm * MT * 32 + j) * 16;
  const int bbase = (wn * NT2 * 32 + j) * 16;
  for (int s = 0; s < S; s++) {
    if (s + 1 < S) stage(s + 1, abuf + ((s + 1) & 1) * ASZ, bbuf + ((s + 1) & 1) * BSZ);
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
  if (MODE == 0) {
    const int direct = job.direct;
    float* __restrict__ out = direct ? job.C : job.C + (long)blockIdx.z * M * Nvalid;
    const long ldo = direct ? job.ldc : (long)Nvalid;
    const bool plain = !direct || (job.epi.addend == nullptr && job.epi.mask == nullptr && job.epi.act != 2);
#pragma unroll
    for (int nn = 0; nn < NT2; nn++) {
      const int col = n0 + (wn * NT2 + nn) * 32 + j;
      if (col >= Nvalid) continue;
      const float bv = (direct && job.epi.bias) ? job.epi.bias[col] : 0.f;
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const long row = m0 + wm * 128 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < M) {
            float v = acc[m][nn][r];
            if (plain) {
              v += bv;
              if (direct && job.epi.act == 1) v = fmaxf(v, 0.f);
            } else {
              v = apply_epi(v, (int)row, col, job.epi);
            }
            out[row * ldo + col] = v;
          }
        }
    }
  } else {
    // token head + cross entropy: each 32-row x 64-column block of a wave goes through the wave's private LDS tile
    static_assert(MODE == 0 || NT2 == 2, "the cross-entropy epilogues use 128-column tiles");
    constexpr int TS = 65;
    float* T = smem + wave * 32 * TS;
    const int col0 = n0 + wn * 64;
    float bvs[NT2];
#pragma unroll
    for (int nn = 0; nn < NT2; nn++) {
      const int col = col0 + nn * 32 + j;
      bvs[nn] = col < Nvalid ? (job.epi.bias ? job.epi.bias[col] : 0.f) : -INFINITY;     // padded columns: exp() = 0
    }
    const int row = lane >> 1, half = lane & 1;                     // row pass: two lanes per row, 32 columns each
    float dbacc = 0.f;                                              // MODE 2: column lane's sum over this wave's 128 rows
    const float gs = (MODE == 2) ? job.gscale[0] * job.inv_rows : 0.f;
#pragma unroll
    for (int m = 0; m < MT; m++) {
      __syncthreads();
#pragma unroll
      for (int nn = 0; nn < NT2; nn++)
#pragma unroll
        for (int r = 0; r < 16; r++) T[((r & 3) + 8 * (r >> 2) + 4 * h) * TS + nn * 32 + j] = acc[m][nn][r] + bvs[nn];
      __syncthreads();
      const long grow = m0 + wm * 128 + m * 32 + row;
      const float* tr = T + row * TS + half * 32;
      if (MODE == 1) {
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 32; c++) mx = fmaxf(mx, tr[c]);
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        float sm = 0.f;
        if (mx > -INFINITY) {
#pragma unroll
          for (int c = 0; c < 32; c++) sm += __expf(tr[c] - mx);
        }
        sm += __shfl_xor(sm, 1, 64);
        if (grow < M) {
          if (half == 0) {
            const long o = (long)(col0 >> 6) * M + grow;
            job.pmax[o] = mx;
            job.psum[o] = sm;
          }
          const long lab = job.labels[grow] - (col0 + half * 32);
          if (lab >= 0 && lab < 32) job.picked[grow] = tr[lab];
        }
      } else {
        float* tw = T + row * TS + half * 32;
        const bool live = grow < M;
        const float ls = live ? job.lse[grow] : 0.f;
        const long lab = live ? job.labels[grow] - (col0 + half * 32) : -1;
        typedef unsigned short us8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int q = 0; q < 4; q++) {
          us8 ov;
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const int c = 8 * q + e;
            float d = live ? (__expf(tr[c] - ls) - (c == lab ? 1.f : 0.f)) * gs : 0.f;
            tw[c] = d;
            ov[e] = f2bf(d);
          }
          if (live) *(us8*)(job.dl + grow * job.ld_dl + col0 + half * 32 + 8 * q) = ov;
        }
        __syncthreads();
        // column pass: lane = column, the 32 rows of the block are 64 contiguous bytes of the transposed copy
        const long rbase = m0 + wm * 128 + m * 32;
        float cs = 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          us8 ov;
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const float d = T[(8 * q + e) * TS + lane];
            cs += d;
            ov[e] = f2bf(d);
          }
          *(us8*)(job.dlT + (long)(col0 + lane) * job.ld_dlT + rbase + 8 * q) = ov;      // rows beyond M hold zeros (dl = 0 there)
        }
        dbacc += cs;
      }
    }
    if (MODE == 2) job.dbpart[(long)(m_tile * 2 + wm) * job.Np + col0 + lane] = dbacc;
  }
}

// lse[m] = log sum_n exp(logit) from the per-chunk (max, sum exp) partials; rowloss[m] = lse[m] - picked[m].
// Workgroup = 32 rows x 8 chunk groups: thread (row r = tid & 31, group g = tid >> 5) folds chunks g, g + 8, ... online, the 8
// groups of a row meet in LDS in a fixed order.  Reads are coalesced along m (partials are [chunk][M]).
__global__ __launch_bounds__(256) void xent_chunks_rows_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                               const float* picked, int chunks, int M,
                                                               float* __restrict__ lse, float* rowloss /* may alias picked */) {
  __shared__ float smx[8][32], ssm[8][32];
  const int r = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int m = blockIdx.x * 32 + r;
  float mx = -INFINITY, sm = 0.f;
  if (m < M) {
    for (int c = g; c < chunks; c += 8) {
      const float pm = pmax[(long)c * M + m], ps = psum[(long)c * M + m];
      if (pm > -INFINITY) {
        if (pm > mx) { sm = sm * __expf(mx - pm) + ps; mx = pm; }      // exp(-inf - pm) = 0 on the first chunk
        else sm += ps * __expf(pm - mx);
      }
    }
  }
  smx[g][r] = mx;
  ssm[g][r] = sm;
  __syncthreads();
  if (g == 0 && m < M) {
    float M2 = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; k++) M2 = fmaxf(M2, smx[k][r]);
    float S2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (smx[k][r] > -INFINITY) S2 += ssm[k][r] * __expf(smx[k][r] - M2);
    const float l = M2 + logf(S2);
    lse[m] = l;
    rowloss[m] = l - picked[m];
  }
}
// loss = inv_rows * sum_m rowloss[m] (one workgroup, fixed order)
__global__ __launch_bounds__(1024) void xent_rows_sum_kernel(const float* __restrict__ rowloss, int M, float inv_rows, float* __restrict__ loss) {
  __shared__ float red[1024];
  float local = 0.f;
  for (int m = threadIdx.x; m < M; m += 1024) local += rowloss[m];
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] * inv_rows;
}

// dst (bf16) [rows_pad][ld_dst >= cols_pad] = src (fp32) [rows][cols] (row stride ld_src), zero beyond rows / cols
__global__ void cvt_pad_bf16_kernel(const float* __restrict__ src, long ld_src, int rows, int cols, unsigned short* __restrict__ dst,
                                    long ld_dst, int rows_pad, int cols_pad) {
  typedef unsigned short us8 __attribute__((ext_vector_type(8)));
  const int g8 = cols_pad >> 3;
  const bool vec = ((ld_src & 3) == 0) && (((uintptr_t)src & 15) == 0);
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < (long)rows_pad * g8; o += (long)gridDim.x * blockDim.x) {
    const int g = (int)(o % g8);
    const long r = o / g8;
    const int c = g * 8;
    us8 ov = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < rows && c < cols) {
      const float* p = src + r * ld_src + c;
      if (vec && c + 8 <= cols) {
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
        ov = us8{f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
      } else {
#pragma unroll
        for (int e = 0; e < 8; e++) ov[e] = (c + e < cols) ? f2bf(p[e]) : (unsigned short)0;
      }
    }
    *(us8*)(dst + r * ld_dst + c) = ov;
  }
}
// transposed form: dst (bf16) [cols_pad][ld_dst >= rows_pad] = src^T; 64 x 64 tiles through LDS (coalesced on both sides)
__global__ __launch_bounds__(256) void cvt_pad_bf16_t_kernel(const float* __restrict__ src, long ld_src, int rows, int cols,
                                                             unsigned short* __restrict__ dst, long ld_dst, int rows_pad, int cols_pad,
                                                             unsigned short* __restrict__ plain, long ld_plain) {
  typedef unsigned short us8 __attribute__((ext_vector_type(8)));
  __shared__ float T[64][65];
  const int tr = rows_pad / 64;                                   // tiles along the source rows (= dst columns)
  const int r0 = (blockIdx.x % tr) * 64, c0 = (blockIdx.x / tr) * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int rr = (tid >> 6) + 4 * i, cc = tid & 63;
    const long r = r0 + rr, c = c0 + cc;
    T[rr][cc] = (r < rows && c < cols) ? src[r * ld_src + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int item = tid + 256 * i;                               // 64 dst rows x 8 groups of 8
    const int cc = item >> 3, g = item & 7;
    us8 ov;
#pragma unroll
    for (int e = 0; e < 8; e++) ov[e] = f2bf(T[8 * g + e][cc]);
    *(us8*)(dst + (long)(c0 + cc) * ld_dst + r0 + 8 * g) = ov;
  }
  if (plain) {                                                    // the un-transposed copy [rows_pad][cols_pad] from the same tile
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int item = tid + 256 * i;
      const int rr = item >> 3, g = item & 7;
      us8 ov;
#pragma unroll
      for (int e = 0; e < 8; e++) ov[e] = f2bf(T[rr][8 * g + e]);
      *(us8*)(plain + (long)(r0 + rr) * ld_plain + c0 + 8 * g) = ov;
    }
  }
}

// ---- elementwise / window helpers on bf16 NHWC tensors ----
__global__ void cvt_f32_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n4) {
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n4; o += (long)gridDim.x * blockDim.x) {
    const f32x4 v = ((const f32x4*)x)[o];
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    ((us4*)y)[o] = us4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
  }
}
__global__ void cvt_bf16_f32_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, long n4) {
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n4; o += (long)gridDim.x * blockDim.x) {
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    const us4 v = ((const us4*)x)[o];
    ((f32x4*)y)[o] = f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
  }
}
// MaxPool2d(3, stride 2, padding 1) on bf16 NHWC
// thread = (output pixel, 8 channels): nine 16-byte loads, one 16-byte store (C % 8 == 0)
__global__ void maxpool3x3_s2_bf16_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int B, int H, int W, int C8,
                                          int Ho, int Wo) {
  typedef unsigned short us8 __attribute__((ext_vector_type(8)));
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < (long)B * Ho * Wo * C8; o += (long)gridDim.x * blockDim.x) {
    const int c = (int)(o % C8);
    long q = o / C8;
    const int ox = (int)(q % Wo); q /= Wo;
    const int oy = (int)(q % Ho);
    const long n = q / Ho;
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; e++) a[e] = -INFINITY;
    for (int ky = 0; ky < 3; ky++) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; kx++) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const us8 v = *(const us8*)(x + (((n * H + iy) * W + ix) * C8 + c) * 8);
#pragma unroll
        for (int e = 0; e < 8; e++) a[e] = fmaxf(a[e], bf2f(v[e]));
      }
    }
    us8 r;
#pragma unroll
    for (int e = 0; e < 8; e++) r[e] = f2bf(a[e]);
    *(us8*)(y + o * 8) = r;
  }
}
// y[n][oy][ox][:] = x[n][2oy][2ox][:]  (bf16 NHWC, 16-byte vectors: C % 8 == 0)
__global__ void subsample2_bf16_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int B, int H, int W, int C8) {
  const int Ho = H / 2, Wo = W / 2;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < (long)B * Ho * Wo * C8; o += (long)gridDim.x * blockDim.x) {
    const int c = (int)(o % C8);
    long q = o / C8;
    const int ox = (int)(q % Wo); q /= Wo;
    const int oy = (int)(q % Ho);
    const long n = q / Ho;
    ((f32x4*)y)[o] = ((const f32x4*)x)[((n * H + 2 * oy) * W + 2 * ox) * C8 + c];
  }
}
// y[n][c] (fp32) = mean over the HW pixels of bf16 x[n][:][c]
__global__ void global_avgpool_bf16_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, int B, int HW, int C) {
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < (long)B * C; o += (long)gridDim.x * blockDim.x) {
    const int c = (int)(o % C);
    const long n = o / C;
    float a = 0.f;
    for (int p = 0; p < HW; p++) a += bf2f(x[(n * HW + p) * C + c]);
    y[o] = a / (float)HW;
  }
}


// ---- ResNet stem on the matrix cores: Conv2d(3,64,7,stride 2,padding 3) + shift + ReLU, fp32 NCHW image -> bf16 NHWC.
// Implicit GEMM with K = 3 x 7 rows x 8 (kx 0..6 and one zero) = 168, padded to 11 k16-steps: k16-step s feeds rows (c,ky) = 2s
// (lanes 0..31) and 2s+1 (lanes 32..63) of v_mfma_f32_32x32x16_bf16, a lane's 8 k values are the 8 consecutive input pixels
// 2ox-3 .. 2ox+4 of that row (four ds_read_b32 of the bf16 row image in LDS).  Workgroup = 4 waves = 4 output rows x up to 128
// output pixels x 64 channels; the 13 input rows and the weights (as B fragments [22 rows][64 co][8]) are staged once per workgroup.
// (The VALU form, resnet.hip conv7x7_s2_stem_kernel, spends 1.9 ms on the 60 GFLOP of a B=256 batch; this one is bound by its
// 0.41 GB of output.)  Image and weights are rounded to bf16 (the VALU form multiplies in fp32 and rounds only its output).
constexpr int STEM_ROWS = 4, STEM_NR = 2 * STEM_ROWS + 5, STEM_RL = 272;
__global__ __launch_bounds__(256, 2) void conv7x7_s2_stem_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                      const float* __restrict__ shift, unsigned short* __restrict__ y,
                                                                      int B, int H, int W, int relu) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned short us8 __attribute__((ext_vector_type(8)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* xin = (unsigned short*)smem_raw;                          // [3][STEM_NR][STEM_RL] bf16, index j <-> ix = j - 3
  unsigned short* wB = xin + 3 * STEM_NR * STEM_RL;                          // [22][64][8] bf16
  float* Tall = (float*)(wB + 22 * 64 * 8);                                 // [4 waves][32][65] fp32
  const int Ho = H >> 1, Wo = W >> 1;
  const int gpr = (Ho + STEM_ROWS - 1) / STEM_ROWS;
  const int n = blockIdx.x / gpr, oy0 = (blockIdx.x % gpr) * STEM_ROWS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // every load of the staging is issued before the first LDS store (42 + 44 per thread: one dependent round trip instead of ~11)
  constexpr int NXI = (3 * STEM_NR * STEM_RL + 255) / 256, NWI = (22 * 64 * 8) / 256;
  float xv[NXI], wv[NWI];
#pragma unroll
  for (int k = 0; k < NXI; k++) {
    const int i = tid + 256 * k;
    const int j = i % STEM_RL, rr = (i / STEM_RL) % STEM_NR, c = i / (STEM_RL * STEM_NR);
    const int iy = 2 * oy0 - 3 + rr, ix = j - 3;
    const bool ok = i < 3 * STEM_NR * STEM_RL && iy >= 0 && iy < H && ix >= 0 && ix < W;
    xv[k] = ok ? x[(((long)n * 3 + c) * H + iy) * W + ix] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < NWI; k++) {
    const int i = tid + 256 * k;
    const int kx = i & 7, co = (i >> 3) & 63, row = i >> 9;
    const int c = row / 7, ky = row % 7;
    wv[k] = (row < 21 && kx < 7) ? w[((co * 3 + c) * 7 + ky) * 7 + kx] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < NXI; k++)
    if (tid + 256 * k < 3 * STEM_NR * STEM_RL) xin[tid + 256 * k] = f2bf(xv[k]);
#pragma unroll
  for (int k = 0; k < NWI; k++) wB[tid + 256 * k] = f2bf(wv[k]);
  __syncthreads();
  const int j = lane & 31, h = lane >> 5;
  const int ox = 32 * wave + j;
  float* T = Tall + wave * 32 * 65;
  const int tq = lane & 7, tp = lane >> 3;                                   // read-back: 8 lanes x 16 B = the 128 bytes of one pixel
  float sh[8];
#pragma unroll
  for (int e = 0; e < 8; e++) sh[e] = shift ? shift[8 * tq + e] : 0.f;
  for (int oyl = 0; oyl < STEM_ROWS; oyl++) {
    const int oy = oy0 + oyl;
    if (oy >= Ho) break;                                                    // uniform
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 11; s++) {
      const int row = 2 * s + h, rowc = row < 21 ? row : 20;               // row 21 does not exist: its weights are zero
      const int c = rowc / 7, ky = rowc % 7;
      const unsigned int* ap = (const unsigned int*)(xin + (c * STEM_NR + 2 * oyl + ky) * STEM_RL + 2 * ox);
      const u32x4 av = {ap[0], ap[1], ap[2], ap[3]};
      const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const bf16x8 b = __builtin_bit_cast(bf16x8, *(const u32x4*)(wB + ((row * 64) + 32 * t + j) * 8));
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
      }
    }
    // 32 pixels x 64 channels through the wave's LDS tile: 16-byte stores of consecutive channels
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) T[((r & 3) + 8 * (r >> 2) + 4 * h) * 65 + t * 32 + j] = acc[t][r];
#pragma unroll
    for (int c8 = 0; c8 < 4; c8++) {                                        // a store instruction = 8 whole pixels (8 x 128 B)
      const int pr = tp + 8 * c8, oxs = 32 * wave + pr;
      us8 ov;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float v = T[pr * 65 + 8 * tq + e] + sh[e];
        if (relu) v = fmaxf(v, 0.f);
        ov[e] = f2bf(v);
      }
      if (oxs < Wo) *(us8*)(y + ((((long)n * Ho + oy) * Wo + oxs) * 64 + 8 * tq)) = ov;
    }
  }
}
}  // namespace

extern "C" {

// 1 when lctvqa_gemm_bf16 takes this problem
int lctvqa_gemm_bf16_supported(long M, int N, int K) { return (M >= 1 && N >= 64 && N % 64 == 0 && K >= 32 && K % 32 == 0) ? 1 : 0; }

// C[M][N] = act(A[M][K] . B[N][K]^T + bias[N] + addend[M][N]); A, B bf16 with row strides lda / ldb (elements, multiples of 8);
// out_bf16: C and addend are bf16, else fp32 (addend may alias C: accumulation).  act: 0 none, 1 ReLU.
int lctvqa_gemm_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, long M, int N, int K, const float* bias,
                     const void* addend, long ld_add, int act, int out_bf16, void* stream) {
  LCT_REQUIRE(lctvqa_gemm_bf16_supported(M, N, K), "gemm_bf16: need N %% 64 == 0, K %% 32 == 0 (got M=%ld N=%d K=%d)", M, N, K);
  LCT_REQUIRE((lda & 7) == 0 && (ldb & 7) == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0, "gemm_bf16: operands must be 16-B aligned rows");
  LCT_REQUIRE(!out_bf16 || ((ldc & 7) == 0 && ((uintptr_t)C & 15) == 0 && (!addend || ((ld_add & 7) == 0 && ((uintptr_t)addend & 15) == 0))),
              "gemm_bf16: bf16 output / addend rows must be 16-B aligned");
  LCT_REQUIRE(M < (1L << 31) && act >= 0 && act <= 1, "gemm_bf16: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const unsigned short* a = (const unsigned short*)A;
  const unsigned short* b = (const unsigned short*)B;
  const long mt = (M + 255) / 256;
#define GB(NT2v, OBF)                                                                                                       \
  {                                                                                                                        \
    const size_t lds = (size_t)(2 * 4 * 256 * 4 + 2 * 4 * 64 * NT2v * 4) * sizeof(float);                                  \
    auto k = gemm_bf16_kernel<NT2v, OBF>;                                                                                  \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
    hipLaunchKernelGGL(k, dim3((unsigned)(mt * (N / (64 * NT2v)))), dim3(256), lds, st, a, lda, b, ldb, C, ldc, (int)M, N, K, bias, addend, \
                       ld_add, act);                                                                                       \
  }
  if (N % 128 == 0) { if (out_bf16) GB(2, true) else GB(2, false) }
  else { if (out_bf16) GB(1, true) else GB(1, false) }
#undef GB
  return lct_check_launch("gemm_bf16");
}

static int bf16x_splits(long M, int Np, int K, int BN) {
  const long tiles = ((M + 255) / 256) * (Np / BN);
  int splits = 1;
  if (tiles < 192) {
    splits = (int)(512 / tiles);
    const int maxs = K / 128;
    if (splits > maxs) splits = maxs;
    if (splits > 32) splits = 32;
    if (splits < 1) splits = 1;
  }
  return splits;
}
// bytes of split-K workspace lctvqa_gemm_bf16x wants for this problem (0: no split)
long lctvqa_gemm_bf16x_workspace_bytes(long M, int N, int Np, int K) {
  const int s = bf16x_splits(M, Np, K, Np % 128 == 0 ? 128 : 64);
  return s > 1 ? (long)s * M * N * (long)sizeof(float) : 0;
}

static int bf16x_check(const void* A, long lda, const void* B, long ldb, long M, int N, int Np, int K) {
  LCT_REQUIRE(M >= 1 && M < (1L << 31) && N >= 1 && Np >= N && Np % 64 == 0 && K >= 32 && K % 32 == 0,
              "gemm_bf16x: need Np %% 64 == 0 >= N, K %% 32 == 0 (got M=%ld N=%d Np=%d K=%d)", M, N, Np, K);
  LCT_REQUIRE((lda & 7) == 0 && (ldb & 7) == 0 && lda >= K && ldb >= K && (((uintptr_t)A | (uintptr_t)B) & 15) == 0,
              "gemm_bf16x: operands must be 16-byte aligned bf16 rows of at least K elements");
  return LCT_OK;
}

// C[M][N] (fp32) = epi(A[M][K] . B[Np][K]^T), bf16 operands (zero-padded to K % 32 == 0; B padded to Np rows, Np % 64 == 0),
// epi(v) = mask * mask_scale * act(v + bias[n] + addend[m][n]) (act 0 none / 1 ReLU / 2 tanh).  Few-tile problems are split along K
// into fp32 slabs in `workspace` (lctvqa_gemm_bf16x_workspace_bytes) and reduced in a fixed order.
int lctvqa_gemm_bf16x(const void* A, long lda, const void* B, long ldb, float* C, long ldc, long M, int N, int Np, int K, const float* bias,
                      const float* addend, long ld_add, int act, const uint8_t* mask, long ld_mask, float mask_scale, float* workspace,
                      long workspace_bytes, void* stream) {
  int rc = bf16x_check(A, lda, B, ldb, M, N, Np, K);
  if (rc) return rc;
  LCT_REQUIRE(act >= 0 && act <= 2 && ldc >= N, "gemm_bf16x: bad act / ldc");
  hipStream_t st = (hipStream_t)stream;
  const int BN = Np % 128 == 0 ? 128 : 64;
  int splits = bf16x_splits(M, Np, K, BN);
  if (splits > 1) {
    const long cap = workspace ? workspace_bytes / (M * N * (long)sizeof(float)) : 0;
    if (cap < splits) splits = cap < 1 ? 1 : (int)cap;
  }
  int kper = ((cdiv(K, splits) + 31) / 32) * 32;
  splits = cdiv(K, kper);
  Bf16Job job{};
  job.A = (const unsigned short*)A; job.lda = lda; job.B = (const unsigned short*)B; job.ldb = ldb;
  job.M = (int)M; job.Nvalid = N; job.Np = Np; job.K = K; job.kper = kper;
  job.epi = GemmEpi{nullptr, nullptr, 1.f, bias, addend, ld_add, mask, ld_mask, mask_scale, act};
  job.direct = splits == 1;
  job.C = splits == 1 ? C : workspace; job.ldc = ldc;
  const long mt = (M + 255) / 256;
  const size_t lds = (size_t)(2 * 4 * 256 * 4 + 2 * 4 * BN * 4) * sizeof(float);
  dim3 grid((unsigned)(mt * (Np / BN)), 1, splits);
  if (BN == 128) {
    auto k = gemm_bf16x_kernel<2, 0>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, job);
  } else {
    auto k = gemm_bf16x_kernel<1, 0>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, job);
  }
  rc = lct_check_launch("gemm_bf16x");
  if (rc || splits == 1) return rc;
  return lct_splitk_reduce(workspace, splits, C, ldc, (int)M, N, job.epi, st);
}

// Token head fused with its cross entropy, forward (reference basic_vqa/models_lct.py:118-120 + 253-255):
//   loss = mean_m( logsumexp_n(X[m] . W[n] + b[n]) - (X[m] . W[labels[m]] + b[labels[m]]) ), lse[m] kept for the backward pass.
// X [M][K], W [Np][K] bf16 (Np % 128 == 0 >= N); part: 2 * (Np / 64) * M floats of scratch; picked: M floats.
long lctvqa_linear_xent_bf16_part_elems(long M, int Np) { return 2L * (Np / 64) * M; }
int lctvqa_linear_xent_bf16_fwd(const void* X, long ldx, const void* W, long ldw, const float* bias, const long* labels, long M, int N,
                                int Np, int K, float* part, float* picked, float* lse, float* loss, void* stream) {
  int rc = bf16x_check(X, ldx, W, ldw, M, N, Np, K);
  if (rc) return rc;
  LCT_REQUIRE(Np % 128 == 0, "linear_xent_bf16: Np %% 128");
  hipStream_t st = (hipStream_t)stream;
  Bf16Job job{};
  job.A = (const unsigned short*)X; job.lda = ldx; job.B = (const unsigned short*)W; job.ldb = ldw;
  job.M = (int)M; job.Nvalid = N; job.Np = Np; job.K = K; job.kper = K;
  job.epi = GemmEpi{nullptr, nullptr, 1.f, bias, nullptr, 0, nullptr, 0, 1.f, 0};
  job.labels = labels; job.pmax = part; job.psum = part + (long)(Np / 64) * M; job.picked = picked;
  const size_t lds = (size_t)(2 * 4 * 256 * 4 + 2 * 4 * 128 * 4) * sizeof(float);
  auto k = gemm_bf16x_kernel<2, 1>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3((unsigned)(((M + 255) / 256) * (Np / 128))), dim3(256), lds, st, job);
  rc = lct_check_launch("linear_xent_bf16_fwd");
  if (rc) return rc;
  // picked[] is consumed by the row kernel and then re-used as the per-row loss buffer of the final sum
  hipLaunchKernelGGL(xent_chunks_rows_kernel, dim3((unsigned)((M + 31) / 32)), dim3(256), 0, st, job.pmax, job.psum, picked, Np / 64, (int)M, lse,
                     picked);
  rc = lct_check_launch("xent_chunks_rows");
  if (rc) return rc;
  hipLaunchKernelGGL(xent_rows_sum_kernel, dim3(1), dim3(1024), 0, st, picked, (int)M, 1.f / (float)M, loss);
  return lct_check_launch("xent_rows_sum");
}
// backward: dl[m][n] = (exp(logit - lse[m]) - [n == labels[m]]) * g[0] / M as bf16, row-major [M][Np] and transposed [Np][ld_dlT >=
// 256-padded M] (padded rows / columns are zeros), and dbpart [ceil(M/256)*2][Np] partial column sums (sum them for the bias gradient).
int lctvqa_linear_xent_bf16_bwd(const void* X, long ldx, const void* W, long ldw, const float* bias, const long* labels, const float* lse,
                                const float* g, long M, int N, int Np, int K, void* dl, long ld_dl, void* dlT, long ld_dlT, float* dbpart,
                                void* stream) {
  int rc = bf16x_check(X, ldx, W, ldw, M, N, Np, K);
  if (rc) return rc;
  const long Mp = ((M + 255) / 256) * 256;
  LCT_REQUIRE(Np % 128 == 0 && ld_dl >= Np && (ld_dl & 7) == 0 && ld_dlT >= Mp && (ld_dlT & 7) == 0 &&
              (((uintptr_t)dl | (uintptr_t)dlT) & 15) == 0, "linear_xent_bf16_bwd: bad dl / dlT layout");
  hipStream_t st = (hipStream_t)stream;
  Bf16Job job{};
  job.A = (const unsigned short*)X; job.lda = ldx; job.B = (const unsigned short*)W; job.ldb = ldw;
  job.M = (int)M; job.Nvalid = N; job.Np = Np; job.K = K; job.kper = K;
  job.epi = GemmEpi{nullptr, nullptr, 1.f, bias, nullptr, 0, nullptr, 0, 1.f, 0};
  job.labels = labels; job.lse = lse; job.gscale = g; job.inv_rows = 1.f / (float)M;
  job.dl = (unsigned short*)dl; job.ld_dl = ld_dl; job.dlT = (unsigned short*)dlT; job.ld_dlT = ld_dlT; job.dbpart = dbpart;
  const size_t lds = (size_t)(2 * 4 * 256 * 4 + 2 * 4 * 128 * 4) * sizeof(float);
  auto k = gemm_bf16x_kernel<2, 2>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3((unsigned)(((M + 255) / 256) * (Np / 128))), dim3(256), lds, st, job);
  return lct_check_launch("linear_xent_bf16_bwd");
}

// dst (bf16) [rows_pad][ld_dst] = src (fp32) [rows][cols] zero-padded; transpose: dst [cols_pad][ld_dst] = src^T (rows_pad, cols_pad
// multiples of 64 then).  cols_pad (rows_pad when transposed) % 8 == 0, dst 16-byte aligned, ld_dst % 8 == 0.
int lctvqa_cvt_pad_bf16(const float* src, long ld_src, int rows, int cols, void* dst, long ld_dst, int rows_pad, int cols_pad, int transpose,
                        void* stream) {
  LCT_REQUIRE(rows >= 1 && cols >= 1 && rows_pad >= rows && cols_pad >= cols && (ld_dst & 7) == 0 && ((uintptr_t)dst & 15) == 0,
              "cvt_pad_bf16: bad shape / alignment");
  hipStream_t st = (hipStream_t)stream;
  if (!transpose) {
    LCT_REQUIRE(cols_pad % 8 == 0 && ld_dst >= cols_pad, "cvt_pad_bf16: cols_pad %% 8");
    long nb = ((long)rows_pad * (cols_pad / 8) + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(cvt_pad_bf16_kernel, dim3((int)nb), dim3(256), 0, st, src, ld_src, rows, cols, (unsigned short*)dst, ld_dst, rows_pad,
                       cols_pad);
    return lct_check_launch("cvt_pad_bf16");
  }
  LCT_REQUIRE(rows_pad % 64 == 0 && cols_pad % 64 == 0 && ld_dst >= rows_pad, "cvt_pad_bf16(transpose): pads must be multiples of 64");
  hipLaunchKernelGGL(cvt_pad_bf16_t_kernel, dim3((unsigned)((rows_pad / 64) * (cols_pad / 64))), dim3(256), 0, st, src, ld_src, rows, cols,
                     (unsigned short*)dst, ld_dst, rows_pad, cols_pad, (unsigned short*)nullptr, 0L);
  return lct_check_launch("cvt_pad_bf16_t");
}
// both copies of one fp32 matrix in one pass: plain [rows_pad][ld_plain >= cols_pad] and transposed [cols_pad][ld_t >= rows_pad]
// (pads multiples of 64): the gradient dY of a Linear is the A operand of dX = dY . W as it is and of dW = dY^T . X transposed
int lctvqa_cvt_pad_bf16_both(const float* src, long ld_src, int rows, int cols, void* plain, long ld_plain, void* dst_t, long ld_t,
                             int rows_pad, int cols_pad, void* stream) {
  LCT_REQUIRE(rows >= 1 && cols >= 1 && rows_pad >= rows && cols_pad >= cols && rows_pad % 64 == 0 && cols_pad % 64 == 0 &&
              ld_plain >= cols_pad && ld_t >= rows_pad && (ld_plain & 7) == 0 && (ld_t & 7) == 0 &&
              (((uintptr_t)plain | (uintptr_t)dst_t) & 15) == 0, "cvt_pad_bf16_both: bad shape / alignment");
  hipLaunchKernelGGL(cvt_pad_bf16_t_kernel, dim3((unsigned)((rows_pad / 64) * (cols_pad / 64))), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                     rows, cols, (unsigned short*)dst_t, ld_t, rows_pad, cols_pad, (unsigned short*)plain, ld_plain);
  return lct_check_launch("cvt_pad_bf16_both");
}

int lctvqa_cvt_f32_to_bf16(const float* x, void* y, long n, void* stream) {
  LCT_REQUIRE(n % 4 == 0, "cvt_f32_to_bf16: n %% 4");
  long nb = (n / 4 + 255) / 256; if (nb > 8192) nb = 8192; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(cvt_f32_bf16_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)y, n / 4);
  return lct_check_launch("cvt_f32_to_bf16");
}
int lctvqa_cvt_bf16_to_f32(const void* x, float* y, long n, void* stream) {
  LCT_REQUIRE(n % 4 == 0, "cvt_bf16_to_f32: n %% 4");
  long nb = (n / 4 + 255) / 256; if (nb > 8192) nb = 8192; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(cvt_bf16_f32_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, y, n / 4);
  return lct_check_launch("cvt_bf16_to_f32");
}
int lctvqa_maxpool3x3_s2_bf16(const void* x, void* y, int B, int H, int W, int C, void* stream) {
  LCT_REQUIRE(C % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, "maxpool3x3_s2_bf16: need C %% 8 == 0 and 16-byte aligned tensors");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  long nb = ((long)B * Ho * Wo * (C / 8) + 255) / 256; if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(maxpool3x3_s2_bf16_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, (unsigned short*)y, B, H, W, C / 8, Ho, Wo);
  return lct_check_launch("maxpool3x3_s2_bf16");
}
int lctvqa_subsample2_bf16(const void* x, void* y, int B, int H, int W, int C, void* stream) {
  LCT_REQUIRE(C % 8 == 0 && H % 2 == 0 && W % 2 == 0, "subsample2_bf16: need C %% 8 == 0 and even H, W");
  long nb = ((long)B * (H / 2) * (W / 2) * (C / 8) + 255) / 256; if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(subsample2_bf16_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, (unsigned short*)y, B, H, W, C / 8);
  return lct_check_launch("subsample2_bf16");
}
int lctvqa_global_avgpool_bf16(const void* x, float* y, int B, int HW, int C, void* stream) {
  long nb = ((long)B * C + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(global_avgpool_bf16_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, y, B, HW, C);
  return lct_check_launch("global_avgpool_bf16");
}

// 1 when the matrix-core stem takes this image size (even H, W; at most 128 output pixels per row)
int lctvqa_conv7x7_s2_stem_bf16_mfma_supported(int B, int H, int W) {
  return (B > 0 && H >= 8 && W >= 8 && H % 2 == 0 && W % 2 == 0 && W / 2 <= 128) ? 1 : 0;
}
int lctvqa_conv7x7_s2_stem_bf16_mfma(const float* x_nchw, const float* w_oihw /* [64][3][7][7] */, const float* shift, void* y_nhwc_bf16,
                                     int B, int H, int W, int relu, void* stream) {
  LCT_REQUIRE(lctvqa_conv7x7_s2_stem_bf16_mfma_supported(B, H, W), "conv7x7_s2_stem_bf16_mfma: unsupported image size %d x %d", H, W);
  LCT_REQUIRE(((uintptr_t)y_nhwc_bf16 & 15) == 0, "conv7x7_s2_stem_bf16_mfma: output must be 16-byte aligned");
  const size_t lds = (size_t)(3 * STEM_NR * STEM_RL + 22 * 64 * 8) * 2 + (size_t)4 * 32 * 65 * 4;
  auto k = conv7x7_s2_stem_mfma_kernel;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int Ho = H / 2;
  hipLaunchKernelGGL(k, dim3(B * ((Ho + STEM_ROWS - 1) / STEM_ROWS)), dim3(256), lds, (hipStream_t)stream, x_nchw, w_oihw, shift,
                     (unsigned short*)y_nhwc_bf16, B, H, W, relu);
  return lct_check_launch("conv7x7_s2_stem_bf16_mfma");
}

}  // extern "C"
