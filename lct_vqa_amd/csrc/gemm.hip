// fp32 MFMA GEMM with general operand strides, split-K and a fused epilogue.
//
//   C[m][n] = epi( sum_k A(m,k) * B(n,k) ),   A(m,k) = A[m*a_ms + k*a_ks],  B(n,k) = B[n*b_ns + k*b_ks]
//   epi(v)  = mask[m][n] * mask_scale * act( v + bias[n] + addend[m][n] )
//
// Replaces the cuBLAS dispatch behind every nn.Linear on the path (reference basic_vqa/models.py:29,37,
// 53,66,80,81,99,102; models_lct.py:87-88,116,120; the VGG classifier Linear layers behind models.py:36)
// and, through the stride arguments, their two backward products (dX = dY.W : b_ks = ldw, b_ns = 1;
// dW = dY^T.X : a_ms = 1, a_ks = ldy, b_ns = 1, b_ks = ldx).
//
// 128x128x16 tile, 4 waves (2x2), each wave a 64x64 output block = 2x2 v_mfma_f32_32x32x2_f32 tiles.
// Operands are staged HBM -> VGPR -> LDS (the VGPR hop is what lets one kernel serve K-contiguous,
// M-contiguous and 4-byte-aligned-only operands such as the reference's Linear(3129,3129));
// LDS rows are 64 B with the 16-B chunk XOR-swizzled by ((row>>2)&3): conflict-free ds_read_b128.
#include "common.h"
#include "gemm_epi.h"

enum { LD_KVEC = 0, LD_KSCALAR = 1, LD_MCONTIG = 2 };

template <int MODE>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long rs, long ks, int row0, int nrows,
                                          int k0, int kend, int tid, float (&r)[8]) {
  if (MODE == LD_KVEC) {
    const int row = row0 + (tid >> 1), k = k0 + (tid & 1) * 8;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (row < nrows) {
      const float* p = P + (long)row * rs + k;
      if (k < kend) a = *(const f32x4*)p;
      if (k + 4 < kend) b = *(const f32x4*)(p + 4);
    }
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  } else if (MODE == LD_KSCALAR) {
    const int row = row0 + (tid >> 1), k = k0 + (tid & 1) * 8;
    if (ks == 1 && (rs & 1) == 0 && (kend & 1) == 0 && ((uintptr_t)P & 7) == 0) {
      // K-contiguous rows that are only 8-byte aligned (row length even, not a multiple of 4: QV = 17 858): four 8-byte loads
      typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        f32x2 v = {0.f, 0.f};
        if (row < nrows && k + i < kend) v = *(const f32x2*)(P + (long)row * rs + k + i);
        r[i] = v[0]; r[i + 1] = v[1];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) r[i] = (row < nrows && k + i < kend) ? P[(long)row * rs + (long)(k + i) * ks] : 0.f;
    }
  } else {
    const int k = k0 + (tid >> 4);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int row = row0 + (tid & 15) + 16 * i;
      r[i] = (row < nrows && k < kend) ? P[(long)row * rs + (long)k * ks] : 0.f;
    }
  }
}

template <int MODE>
__device__ __forceinline__ void store_tile(float* __restrict__ L, int tid, const float (&r)[8]) {
  if (MODE == LD_MCONTIG) {
    const int k = tid >> 4;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int row = (tid & 15) + 16 * i;
      L[row * 16 + (((k >> 2) ^ ((row >> 2) & 3)) << 2) + (k & 3)] = r[i];
    }
  } else {
    const int row = tid >> 1, kc = (tid & 1) * 2, sw = (row >> 2) & 3;
    *(f32x4*)(L + row * 16 + ((kc ^ sw) << 2)) = f32x4{r[0], r[1], r[2], r[3]};
    *(f32x4*)(L + row * 16 + (((kc + 1) ^ sw) << 2)) = f32x4{r[4], r[5], r[6], r[7]};
  }
}

// A(m,k) := mask * scale * tanh(A * A2) applied to the 8 staged elements of this thread (K-contiguous modes only)
template <int MODE>
__device__ __forceinline__ void fuse_a(float (&r)[8], const GemmEpi& e, long rs, long ks, int row0, int nrows, int k0, int kend,
                                       int K, int tid) {
  float r2[8];
  load_tile<MODE>(e.a2, rs, ks, row0, nrows, k0, kend, tid, r2);
  const int row = row0 + (tid >> 1), k = k0 + (tid & 1) * 8;      // element mapping of LD_KVEC / LD_KSCALAR
#pragma unroll
  for (int i = 0; i < 8; i++) {
    float v = tanhf(r[i] * r2[i]);
    if (e.amask) v = (row < nrows && k + i < kend && e.amask[(long)row * K + k + i]) ? v * e.ascale : 0.f;
    r[i] = v;
  }
}

// Epilogue of one wave: MT x NT accumulator tiles whose first output is (mw, nw).  Interior tiles with the common epilogues
// (bias, residual addend, ReLU) take a path without per-element bounds, branches or index multiplies.
template <int MT, int NT>
__device__ __forceinline__ void gemm_store_tiles(const f32x16 (&acc)[MT][NT], float* __restrict__ out, long ldc, int mw, int nw,
                                                 int mlim, int nlim, int M, int N, const GemmEpi& epi, int direct, int j, int h) {
  const bool fast = (mlim <= M) && (nlim <= N) && (!direct || (epi.mask == nullptr && epi.act != 2));
  if (fast) {
    const bool relu = direct && epi.act == 1;
#pragma unroll
    for (int b = 0; b < NT; b++) {
      const int n = nw + b * 32 + j;
      const float bv = (direct && epi.bias) ? epi.bias[n] : 0.f;
#pragma unroll
      for (int a = 0; a < MT; a++) {
        const long m = mw + a * 32 + 4 * h;
        float* po = out + m * ldc + n;
        if (direct && epi.addend) {
          const float* pa = epi.addend + m * epi.ld_add + n;
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int ro = (r & 3) + 8 * (r >> 2);
            float v = acc[a][b][r] + bv + pa[ro * epi.ld_add];
            po[ro * ldc] = relu ? fmaxf(v, 0.f) : v;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int ro = (r & 3) + 8 * (r >> 2);
            float v = acc[a][b][r] + bv;
            po[ro * ldc] = relu ? fmaxf(v, 0.f) : v;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int a = 0; a < MT; a++)
#pragma unroll
    for (int b = 0; b < NT; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = mw + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int n = nw + b * 32 + j;
        if (m < M && n < N) {
          float v = acc[a][b][r];
          if (direct) v = apply_epi(v, m, n, epi);
          out[(long)m * ldc + n] = v;
        }
      }
}

template <int AMODE, int BMODE>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const float* __restrict__ A, long a_ms, long a_ks,
                                                          const float* __restrict__ B, long b_ns, long b_ks,
                                                          float* __restrict__ C, long ldc, int M, int N, int K,
                                                          int kper, GemmEpi epi, int direct, int relu_ab) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][128 * 16];   // [buf][A|B][row][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, j = lane & 31, h = lane >> 5;
  const int ntn = (N + 127) / 128;
  const int tn = blockIdx.x % ntn, tm = blockIdx.x / ntn;
  const int m0 = tm * 128, n0 = tn * 128;
  const int kbeg = blockIdx.z * kper;
  const int kend = min(K, kbeg + kper);
  const int nk = (kend - kbeg + 15) / 16;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  float ra[8], rb[8];
  if (nk > 0) {
    load_tile<AMODE>(A, a_ms, a_ks, m0, M, kbeg, kend, tid, ra);
    load_tile<BMODE>(B, b_ns, b_ks, n0, N, kbeg, kend, tid, rb);
    if (epi.a2) fuse_a<AMODE>(ra, epi, a_ms, a_ks, m0, M, kbeg, kend, K, tid);
    if (relu_ab & 1) {
#pragma unroll
      for (int i = 0; i < 8; i++) ra[i] = fmaxf(ra[i], 0.f);
    }
    if (relu_ab & 2) {
#pragma unroll
      for (int i = 0; i < 8; i++) rb[i] = fmaxf(rb[i], 0.f);
    }
    store_tile<AMODE>(lds[0][0], tid, ra);
    store_tile<BMODE>(lds[0][1], tid, rb);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; kt++) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      load_tile<AMODE>(A, a_ms, a_ks, m0, M, kbeg + (kt + 1) * 16, kend, tid, ra);
      load_tile<BMODE>(B, b_ns, b_ks, n0, N, kbeg + (kt + 1) * 16, kend, tid, rb);
      if (epi.a2) fuse_a<AMODE>(ra, epi, a_ms, a_ks, m0, M, kbeg + (kt + 1) * 16, kend, K, tid);
      if (relu_ab & 1) {
#pragma unroll
        for (int i = 0; i < 8; i++) ra[i] = fmaxf(ra[i], 0.f);
      }
      if (relu_ab & 2) {
#pragma unroll
        for (int i = 0; i < 8; i++) rb[i] = fmaxf(rb[i], 0.f);
      }
    }
    const float* la = lds[cur][0];
    const float* lb = lds[cur][1];
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      const int c = 2 * kk + h;
      f32x4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int ar = wm * 64 + i * 32 + j, br = wn * 64 + i * 32 + j;
        af[i] = *(const f32x4*)(la + ar * 16 + ((c ^ ((ar >> 2) & 3)) << 2));
        bf[i] = *(const f32x4*)(lb + br * 16 + ((c ^ ((br >> 2) & 3)) << 2));
      }
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int b = 0; b < 2; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][e], bf[b][e], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      store_tile<AMODE>(lds[cur ^ 1][0], tid, ra);
      store_tile<BMODE>(lds[cur ^ 1][1], tid, rb);
    }
    __syncthreads();
  }

  float* out = direct ? C : C + (long)blockIdx.z * M * ldc;   // split-K slab [z][M][ldc]
  gemm_store_tiles<2, 2>(acc, out, ldc, m0 + wm * 64, n0 + wn * 64, m0 + 128, n0 + 128, M, N, epi, direct, j, h);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same product for operands the LDS-DMA can stage directly (16-byte aligned, K- or row-contiguous, no fused A prologue; an
// operand ReLU is applied to the fragments after their LDS read):
// 256 x (64*NTW) x 16 tile, 4 waves (2 x 2), wave = 4 x NTW v_mfma_f32_32x32x2_f32 tiles (128 x 32*NTW outputs), both operand
// tiles HBM -> LDS by global_load_lds (16 B per lane, no VGPR hop, no address VALU in the loop), double buffered with one
// barrier per K step of 16 while the next step's 6 DMA instructions per wave are in flight.
//   K-contiguous operand  X(r,k) = X[r*ld + k]: LDS rows of 64 B, the 16-B chunk XOR-swizzled by (row>>2)&3 exactly as in the
//     general kernel (conflict-free ds_read_b128, one per four MFMAs; lane half h of half-step kk reads quad 2kk+h).  The DMA
//     writes LDS linearly, so the swizzle is applied to the SOURCE: lane = (row, slot c) fetches quad c ^ ((row>>2)&3); the four
//     lanes of a row still cover its 64 contiguous bytes;
//   row-contiguous operand X(r,k) = X[k*ld + r] (the transposed operand of dX = dY.W and dW = dY^T.X): LDS [k][row], a fragment
//     element is one conflict-free ds_read_b32 per MFMA; k = 8kk + 4h + e keeps the pairing of the two operand kinds equal.
// blockIdx.y = batch (operand strides sA, sB, sC; the transformed-domain products of a Winograd convolution), blockIdx.z = K split.
__device__ __attribute__((aligned(64))) float g_zero_page_g[64];

template <bool AKM, bool BKM, int NTW>
__global__ __launch_bounds__(256, 2) void gemm_f32_dma_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                                              long ldb, float* __restrict__ C, long ldc, int M, int N, int K,
                                                              int kper, GemmEpi epi, int direct, long sA, long sB, long sC, int relu_ab,
                                                              int flat_nb) {
  constexpr int BM = 256, BN = 64 * NTW;
  constexpr int ASZ = BM * 16, BSZ = BN * 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* abuf = smem;                 // [2][ASZ]
  float* bbuf = smem + 2 * ASZ;       // [2][BSZ]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, j = lane & 31, h = lane >> 5;
  const int ntn = (N + BN - 1) / BN, ntm = (M + BM - 1) / BM;
  // workgroups are dealt to the 8 XCDs round-robin: with at least 8 row blocks the ntn workgroups that read one A row block are
  // given ids that share an XCD (its L2); fewer row blocks keep the plain order (the other mapping would leave XCDs idle)
  int tn, tm, nb = blockIdx.y;
  if (flat_nb > 0) {
    // batched: the row blocks of ALL batch members are dealt to the XCDs together (49 row blocks per member would give XCD 0 seven
    // and the others six of every member: a 14 % longer tail on that XCD)
    const int r = ((blockIdx.x >> 3) / ntn) * 8 + (blockIdx.x & 7);
    if (r >= flat_nb * ntm) return;
    tn = (blockIdx.x >> 3) % ntn; nb = r / ntm; tm = r - nb * ntm;
  } else if (ntm >= 8) { tn = (blockIdx.x >> 3) % ntn; tm = ((blockIdx.x >> 3) / ntn) * 8 + (blockIdx.x & 7); }
  else { tn = blockIdx.x % ntn; tm = blockIdx.x / ntn; }
  if (tm >= ntm) return;
  const int m0 = tm * BM, n0 = tn * BN;
  A += (long)nb * sA; B += (long)nb * sB; C += (long)nb * sC;
  const int kbeg = blockIdx.z * kper;
  const int kend = min(K, kbeg + kper);
  const int nk = (kend - kbeg + 15) / 16;

  // ---- staging: per-lane source pointers and wave-uniform LDS destinations.
  // K-contiguous: a piece = 16 rows x 64 B, lane = (row lane>>2, slot lane&3); wave w stages pieces w, w+4, ... (A: 16, B: 2 NTW*2).
  // Row-contiguous: A piece = one k row (256 floats), wave w stages k = w, w+4, w+8, w+12; B piece = 256 floats of [k][BN].
  const int rl = lane >> 2, sq = (lane & 3) ^ ((rl >> 2) & 3);   // K-contiguous: row within the piece, source quad of this lane
  const float* asrc;
  if (!AKM) asrc = A + (long)(m0 + 16 * wave + rl) * lda + 4 * sq;
  else asrc = A + (m0 + 4 * lane);
  const bool a_in = AKM ? (m0 + 4 * lane < M) : true;
  const float* bsrc;
  int bk0 = 0;                                                   // row-contiguous B: first k (within the step) of this lane
  if (!BKM) bsrc = B + (long)(n0 + 16 * wave + rl) * ldb + 4 * sq;
  else {
    bsrc = B + n0 + (NTW == 2 ? 4 * (lane & 31) : 4 * (lane & 15));
    bk0 = NTW == 2 ? 2 * wave + (lane >> 5) : 4 * wave + (lane >> 4);
  }
  const bool b_in = BKM ? (n0 + (NTW == 2 ? 4 * (lane & 31) : 4 * (lane & 15)) < N) : true;
  auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
    const int k0 = kbeg + kt * 16;
    float* ad = abuf + buf * ASZ;
    float* bd = bbuf + buf * BSZ;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      if (!AKM) {
        const bool ok = (m0 + 16 * (wave + 4 * t) + rl < M) && (k0 + 4 * sq < kend);
        glds16_opaque(ok ? asrc + (long)(64 * t) * lda + k0 : g_zero_page_g, ad + (wave + 4 * t) * 256);
      } else {
        const int k = k0 + wave + 4 * t;
        glds16_opaque((a_in && k < kend) ? asrc + (long)k * lda : g_zero_page_g, ad + (wave + 4 * t) * BM);
      }
    }
#pragma unroll
    for (int t = 0; t < NTW; t++) {
      if (!BKM) {
        const bool ok = (n0 + 16 * (wave + 4 * t) + rl < N) && (k0 + 4 * sq < kend);
        glds16_opaque(ok ? bsrc + (long)(64 * t) * ldb + k0 : g_zero_page_g, bd + (wave + 4 * t) * 256);
      } else {
        const int k = k0 + bk0 + 8 * t;
        glds16_opaque((b_in && k < kend) ? bsrc + (long)k * ldb : g_zero_page_g,
                      bd + (NTW == 2 ? (2 * wave + 8 * t) * BN : 4 * wave * BN));
      }
    }
  };
  // fragment offsets (floats) of half-step kk = 0, 1 for the K-contiguous layout: row j of a 32-row tile, quad 2kk+h swizzled
  const int swz = (j >> 2) & 3;
  const int fo0 = j * 16 + (((0 + h) ^ swz) << 2), fo1 = j * 16 + (((2 + h) ^ swz) << 2);

  f32x16 acc[4][NTW];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < NTW; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  if (nk > 0) stage(0, 0);
  for (int kt = 0; kt < nk; kt++) {
    // step kt has landed (this wave's pieces: vmcnt(0); the others': the barrier), and nobody still reads the other buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const float* la = abuf + (kt & 1) * ASZ;
    const float* lb = bbuf + (kt & 1) * BSZ;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      float af[4][4], bf[NTW][4];
#pragma unroll
      for (int a = 0; a < 4; a++) {
        if (!AKM) {
          const f32x4 v = *(const f32x4*)(la + (wm * 128 + a * 32) * 16 + (kk ? fo1 : fo0));
          af[a][0] = v[0]; af[a][1] = v[1]; af[a][2] = v[2]; af[a][3] = v[3];
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) af[a][e] = la[(8 * kk + 4 * h + e) * BM + wm * 128 + a * 32 + j];
        }
      }
#pragma unroll
      for (int b = 0; b < NTW; b++) {
        if (!BKM) {
          const f32x4 v = *(const f32x4*)(lb + (wn * 32 * NTW + b * 32) * 16 + (kk ? fo1 : fo0));
          bf[b][0] = v[0]; bf[b][1] = v[1]; bf[b][2] = v[2]; bf[b][3] = v[3];
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) bf[b][e] = lb[(8 * kk + 4 * h + e) * BN + wn * 32 * NTW + b * 32 + j];
        }
      }
      if (relu_ab & 1) {                                           // operand ReLU (PC-DARTS ReLU -> Conv1x1 blocks) on the fragments
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int e = 0; e < 4; e++) af[a][e] = fmaxf(af[a][e], 0.f);
      }
      if (relu_ab & 2) {
#pragma unroll
        for (int b = 0; b < NTW; b++)
#pragma unroll
          for (int e = 0; e < 4; e++) bf[b][e] = fmaxf(bf[b][e], 0.f);
      }
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < NTW; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][e], bf[b][e], acc[a][b], 0, 0, 0);
    }
  }

  float* out = direct ? C : C + (long)blockIdx.z * M * ldc;   // split-K slab [z][M][ldc]
  gemm_store_tiles<4, NTW>(acc, out, ldc, m0 + wm * 128, n0 + wn * 32 * NTW, m0 + BM, n0 + BN, M, N, epi, direct, j, h);
}

// The K-contiguous x K-contiguous case with BUFFER addressing of the LDS-DMA (K % 16 == 0): `buffer_load_dwordx4 ... offen lds` takes a
// wave-uniform 128-bit descriptor (SGPRs), one 32-bit byte offset per lane that is the same for every K step, and the step's byte offset
// as a scalar -- no per-step address arithmetic on the vector ALU and half the address registers of `global_load_lds` (64-bit per lane).
// Rows past M / N are CLAMPED to the last row (their products land in output rows / columns that the epilogue does not store).
template <int NTW>
__global__ __launch_bounds__(256, 2) void gemm_f32_nt_buf_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb,
                                                                 float* __restrict__ C, long ldc, int M, int N, int K, GemmEpi epi, long sA,
                                                                 long sB, long sC, int flat_nb) {
  constexpr int BM = 256, BN = 64 * NTW;
  constexpr int ASZ = BM * 16, BSZ = BN * 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* abuf = smem;                 // [2][ASZ]
  float* bbuf = smem + 2 * ASZ;       // [2][BSZ]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, j = lane & 31, h = lane >> 5;
  const int ntn = (N + BN - 1) / BN, ntm = (M + BM - 1) / BM;
  int tn, tm, nb = blockIdx.y;
  if (flat_nb > 0) {                                              // see gemm_f32_dma_kernel
    const int r = ((blockIdx.x >> 3) / ntn) * 8 + (blockIdx.x & 7);
    if (r >= flat_nb * ntm) return;
    tn = (blockIdx.x >> 3) % ntn; nb = r / ntm; tm = r - nb * ntm;
  } else if (ntm >= 8) { tn = (blockIdx.x >> 3) % ntn; tm = ((blockIdx.x >> 3) / ntn) * 8 + (blockIdx.x & 7); }
  else { tn = blockIdx.x % ntn; tm = blockIdx.x / ntn; }
  if (tm >= ntm) return;
  const int m0 = tm * BM, n0 = tn * BN;
  A += (long)nb * sA + (long)m0 * lda; B += (long)nb * sB + (long)n0 * ldb; C += (long)nb * sC;
  const lct_i32x4 ra = lct_make_rsrc(A), rb = lct_make_rsrc(B);
  const int nk = K / 16;
  const int rl = lane >> 2, sq = (lane & 3) ^ ((rl >> 2) & 3);
  unsigned aoff[4], boff[NTW];
#pragma unroll
  for (int t = 0; t < 4; t++) aoff[t] = (unsigned)(((long)min(16 * (wave + 4 * t) + rl, M - 1 - m0) * lda + 4 * sq) * 4);
#pragma unroll
  for (int t = 0; t < NTW; t++) boff[t] = (unsigned)(((long)min(16 * (wave + 4 * t) + rl, N - 1 - n0) * ldb + 4 * sq) * 4);
  auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)kt * 64u);
    float* ad = abuf + buf * ASZ;
    float* bd = bbuf + buf * BSZ;
#pragma unroll
    for (int t = 0; t < 4; t++) bufdma16(aoff[t], ra, so, ad + (wave + 4 * t) * 256);
#pragma unroll
    for (int t = 0; t < NTW; t++) bufdma16(boff[t], rb, so, bd + (wave + 4 * t) * 256);
  };
  const int swz = (j >> 2) & 3;
  const int fo0 = j * 16 + (((0 + h) ^ swz) << 2), fo1 = j * 16 + (((2 + h) ^ swz) << 2);
  f32x16 acc[4][NTW];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < NTW; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
  if (nk > 0) stage(0, 0);
  for (int kt = 0; kt < nk; kt++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);               // (requesting the first fragments before these six pieces: measured 2 % slower)
    const float* la = abuf + (kt & 1) * ASZ;
    const float* lb = bbuf + (kt & 1) * BSZ;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      f32x4 af[4], bf[NTW];
#pragma unroll
      for (int a = 0; a < 4; a++) af[a] = *(const f32x4*)(la + (wm * 128 + a * 32) * 16 + (kk ? fo1 : fo0));
#pragma unroll
      for (int b = 0; b < NTW; b++) bf[b] = *(const f32x4*)(lb + (wn * 32 * NTW + b * 32) * 16 + (kk ? fo1 : fo0));
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < NTW; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][e], bf[b][e], acc[a][b], 0, 0, 0);
    }
  }
  gemm_store_tiles<4, NTW>(acc, C, ldc, m0 + wm * 128, n0 + wn * 32 * NTW, m0 + BM, n0 + BN, M, N, epi, 1, j, h);
}

__global__ void gemm_splitk_reduce_kernel(const float* __restrict__ ws, int splits, float* __restrict__ C, long ldc,
                                          int M, int N, GemmEpi epi) {
  long total = (long)M * N;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    int n = o % N, m = o / N;
    // four slabs per iteration: independent loads in flight (one at a time made a 32-way split 32 dependent HBM latencies)
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int s = 0;
    for (; s + 3 < splits; s += 4) {
      const float* p = ws + (long)s * total + o;
      v0 += p[0]; v1 += p[total]; v2 += p[2 * total]; v3 += p[3 * total];
    }
    for (; s < splits; s++) v0 += ws[(long)s * total + o];
    const float v = (v0 + v1) + (v2 + v3);
    C[(long)m * ldc + n] = apply_epi(v, m, n, epi);
  }
}

// column sums: out[n] = sum_m X[m][n]  (bias gradients), deterministic two-stage reduction.
// stage 1: block (cb, rb) sums rows [rb*rows_per, ...) of 64 columns -> part[rb][n]; stage 2 sums the parts.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, long ld, int M, int N, int rows_per,
                                                            float* __restrict__ part) {
  __shared__ float red[4][64];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float a = 0.f;
  if (n < N)
  {
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;                           // four rows per iteration: independent loads in flight
    int m = r0 + w;
    for (; m + 12 < r1; m += 16) {
      a += X[(long)m * ld + n]; a1 += X[(long)(m + 4) * ld + n]; a2 += X[(long)(m + 8) * ld + n]; a3 += X[(long)(m + 12) * ld + n];
    }
    for (; m < r1; m += 4) a += X[(long)m * ld + n];
    a = (a + a1) + (a2 + a3);
  }
  red[w][threadIdx.x & 63] = a;
  __syncthreads();
  if (w == 0 && n < N) part[(long)blockIdx.y * N + n] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// stage 2: workgroup = 64 columns x 16 row groups; a group sums its parts eight loads at a time, then the groups in fixed order
// (one thread per column walking up to 2048 parts was a chain of 2048 dependent HBM latencies: ~1 ms for a 64-channel bias gradient)
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ part, int S, int N, float* __restrict__ out) {
  __shared__ float red[16][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  float a = 0.f;
  if (n < N) {
    float v[8];
    int s = g;
    for (; s + 7 * 16 < S; s += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = part[(long)(s + 16 * u) * N + n];
#pragma unroll
      for (int u = 0; u < 8; u++) a += v[u];
    }
    for (; s < S; s += 16) a += part[(long)s * N + n];
  }
  red[g][c] = a;
  __syncthreads();
  if (g == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) t += red[k][c];
    out[n] = t;
  }
}

static int pick_mode(const float* P, long rs, long ks, int K) {
  if (ks == 1) {
    if (rs % 4 == 0 && K % 4 == 0 && ((uintptr_t)P & 15) == 0) return LD_KVEC;
    return LD_KSCALAR;
  }
  if (rs == 1) return LD_MCONTIG;
  return LD_KSCALAR;   // fully general strides
}

static int g_gemm_force_general = 0;                  // tests / A-B: route everything through the general kernel
static thread_local const float* g_fuse_a2 = nullptr;
static thread_local const uint8_t* g_fuse_amask = nullptr;
static thread_local float g_fuse_ascale = 1.f;

typedef void (*gemm_kern_t)(const float*, long, long, const float*, long, long, float*, long, int, int, int, int,
                            GemmEpi, int, int);
int lct_splitk_reduce(const float* ws, int splits, float* C, long ldc, int M, int N, const GemmEpi& epi, hipStream_t st) {
  long total = (long)M * N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, ws, splits, C, ldc, M, N, epi);
  return lct_check_launch("gemm_splitk_reduce");
}

static gemm_kern_t gemm_table[3][3] = {
    {gemm_f32_kernel<0, 0>, gemm_f32_kernel<0, 1>, gemm_f32_kernel<0, 2>},
    {gemm_f32_kernel<1, 0>, gemm_f32_kernel<1, 1>, gemm_f32_kernel<1, 2>},
    {gemm_f32_kernel<2, 0>, gemm_f32_kernel<2, 1>, gemm_f32_kernel<2, 2>}};

extern "C" {

static int plan_splits(int tiles, int K) {
  int splits = 1;
  if (tiles < 192) {
    splits = 512 / tiles;
    int maxs = K / 64;
    if (splits > maxs) splits = maxs;
    int cap = K >= 65536 ? 512 : 32;         // pixel-reduction GEMMs (1x1-conv weight gradients: K = B*H*W) get deep splits
    if (splits > cap) splits = cap;
    if (splits < 1) splits = 1;
  }
  return splits;
}
static int dma_bn(int N) { return N > 64 ? 128 : 64; }

// workspace bytes needed for the split-K slabs of this problem (0 when no split is used): enough for either kernel
long lctvqa_gemm_workspace_bytes(int M, int N, int K) {
  int s1 = plan_splits(cdiv(M, 128) * cdiv(N, 128), K);
  int s2 = plan_splits(cdiv(M, 256) * cdiv(N, dma_bn(N)), K);
  int splits = s1 > s2 ? s1 : s2;
  return splits > 1 ? (long)splits * M * N * (long)sizeof(float) : 0;
}

typedef void (*gemm_dma_kern_t)(const float*, long, const float*, long, float*, long, int, int, int, int, GemmEpi, int, long, long,
                                long, int, int);
static gemm_dma_kern_t gemm_dma_table[2][2][2] = {
    {{gemm_f32_dma_kernel<false, false, 1>, gemm_f32_dma_kernel<false, false, 2>},
     {gemm_f32_dma_kernel<false, true, 1>, gemm_f32_dma_kernel<false, true, 2>}},
    {{gemm_f32_dma_kernel<true, false, 1>, gemm_f32_dma_kernel<true, false, 2>},
     {gemm_f32_dma_kernel<true, true, 1>, gemm_f32_dma_kernel<true, true, 2>}}};

// operand layout the LDS-DMA kernel can stage: 0 = K-contiguous, 1 = row-contiguous, -1 = neither
static int dma_layout(const float* P, long rs, long ks, int rows, int K) {
  if (((uintptr_t)P & 15) != 0 || K % 4 != 0) return -1;
  if (ks == 1 && rs % 4 == 0) return 0;
  if (rs == 1 && ks % 4 == 0 && rows % 4 == 0) return 1;
  return -1;
}

static int launch_gemm_dma(const float* A, long lda, int akm, const float* B, long ldb, int bkm, float* C, long ldc, int M, int N,
                           int K, const GemmEpi& epi, float* workspace, long workspace_bytes, int batch, long sA, long sB, long sC,
                           hipStream_t st, int relu_ab = 0) {
  const int BN = dma_bn(N);
  const int ntm = cdiv(M, 256), ntn = cdiv(N, BN);
  int splits = batch > 1 ? 1 : plan_splits(ntm * ntn, K);
  if (splits > 1) {
    long cap = workspace ? workspace_bytes / ((long)M * N * (long)sizeof(float)) : 0;
    if (cap < splits) splits = cap < 1 ? 1 : (int)cap;           // degrade, still correct
  }
  int kper = ((cdiv(K, splits) + 15) / 16) * 16;
  splits = cdiv(K, kper);
  gemm_dma_kern_t k = gemm_dma_table[akm][bkm][BN == 128];
  const size_t lds = (size_t)(2 * 256 * 16 + 2 * BN * 16) * sizeof(float);
  // batched products whose row-block count is not a multiple of 8: one flat grid over (member, row block), dealt to the XCDs together
  static const int flat_on = [] { const char* e = getenv("LCTVQA_GEMM_FLAT"); return e ? atoi(e) : 1; }();
  const int flat_nb = (flat_on && batch > 1 && (long)batch * ntm >= 8 && (ntm % 8 != 0 || flat_on > 1)) ? batch : 0;
  dim3 grid(flat_nb ? (((long)batch * ntm + 7) / 8) * 8 * ntn : ntm >= 8 ? ((ntm + 7) / 8) * 8 * ntn : ntm * ntn, flat_nb ? 1 : batch, splits);
  static const int buf_on = [] { const char* e = getenv("LCTVQA_GEMM_BUF"); return e ? atoi(e) : 1; }();
  if (buf_on && !akm && !bkm && BN == 128 && splits == 1 && relu_ab == 0 && K % 16 == 0 && (long)256 * lda * 4 < 0x7fffff00L &&
      (long)128 * ldb * 4 < 0x7fffff00L) {
    hipLaunchKernelGGL(gemm_f32_nt_buf_kernel<2>, grid, dim3(256), lds, st, A, lda, B, ldb, C, ldc, M, N, K, epi, sA, sB, sC, flat_nb);
    return lct_check_launch("gemm_f32_nt_buf");
  }
  if (splits == 1) {
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, A, lda, B, ldb, C, ldc, M, N, K, kper, epi, 1, sA, sB, sC, relu_ab, flat_nb);
    return lct_check_launch("gemm_f32_dma");
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, A, lda, B, ldb, workspace, (long)N, M, N, K, kper, epi, 0, 0L, 0L, 0L, relu_ab, 0);
  int rc = lct_check_launch("gemm_f32_dma(splitk)");
  if (rc) return rc;
  return lct_splitk_reduce(workspace, splits, C, ldc, M, N, epi, st);
}

// relu_ab: bit 0 = apply ReLU to A elements as they are loaded, bit 1 = to B elements (PC-DARTS ReLU->Conv1x1
// blocks and their weight gradients, operations.py:27-28,43,58-61)
int lctvqa_gemm_f32_ex(const float* A, long a_ms, long a_ks, const float* B, long b_ns, long b_ks, float* C, long ldc,
                       int M, int N, int K, const float* bias, const float* addend, long ld_add, int act,
                       const uint8_t* mask, long ld_mask, float mask_scale, float* workspace, long workspace_bytes,
                       int relu_ab, void* stream) {
  LCT_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_f32: bad shape %d %d %d", M, N, K);
  LCT_REQUIRE(act >= 0 && act <= 2, "gemm_f32: bad act %d", act);
  hipStream_t st = (hipStream_t)stream;
  GemmEpi epi{nullptr, nullptr, 1.f, bias, addend, ld_add, mask, ld_mask, mask_scale, act};
  if (relu_ab & 4) {      // fused fusion-head prologue requested through lctvqa_fusion_fc1 (operands stashed thread-locally)
    epi.a2 = g_fuse_a2; epi.amask = g_fuse_amask; epi.ascale = g_fuse_ascale;
    relu_ab &= 3;
    LCT_REQUIRE(a_ks == 1, "fused fusion-head prologue needs a K-contiguous A operand");
  }
  if (!epi.a2 && !g_gemm_force_general) {
    const int la = dma_layout(A, a_ms, a_ks, M, K), lb = dma_layout(B, b_ns, b_ks, N, K);
    if (la >= 0 && lb >= 0)
      return launch_gemm_dma(A, la ? a_ks : a_ms, la, B, lb ? b_ks : b_ns, lb, C, ldc, M, N, K, epi, workspace, workspace_bytes, 1, 0, 0,
                             0, st, relu_ab);
  }
  int tiles = cdiv(M, 128) * cdiv(N, 128);
  int splits = plan_splits(tiles, K);
  if (splits > 1 && (workspace == nullptr || workspace_bytes < (long)splits * M * N * (long)sizeof(float))) splits = 1;   // degrade, still correct
  int kper = ((cdiv(K, splits) + 15) / 16) * 16;
  splits = cdiv(K, kper);
  int am = pick_mode(A, a_ms, a_ks, K), bm = pick_mode(B, b_ns, b_ks, K);
  gemm_kern_t k = gemm_table[am][bm];
  if (splits == 1) {
    hipLaunchKernelGGL(k, dim3(tiles, 1, 1), dim3(256), 0, st, A, a_ms, a_ks, B, b_ns, b_ks, C, ldc, M, N, K, kper, epi, 1, relu_ab);
    return lct_check_launch("gemm_f32");
  }
  hipLaunchKernelGGL(k, dim3(tiles, 1, splits), dim3(256), 0, st, A, a_ms, a_ks, B, b_ns, b_ks, workspace, (long)N, M, N, K,
                     kper, epi, 0, relu_ab);
  int rc = lct_check_launch("gemm_f32(splitk)");
  if (rc) return rc;
  return lct_splitk_reduce(workspace, splits, C, ldc, M, N, epi, st);
}

static int colsum_splits(int M, int N) {
  int cap = 2048 / cdiv(N, 64);            // aim at <= ~2048 workgroups
  if (cap < 64) cap = 64;
  int S = cdiv(M, 64);
  if (S > cap) S = cap;
  return S;
}

int lctvqa_gemm_f32(const float* A, long a_ms, long a_ks, const float* B, long b_ns, long b_ks, float* C, long ldc,
                    int M, int N, int K, const float* bias, const float* addend, long ld_add, int act,
                    const uint8_t* mask, long ld_mask, float mask_scale, float* workspace, long workspace_bytes,
                    void* stream) {
  return lctvqa_gemm_f32_ex(A, a_ms, a_ks, B, b_ns, b_ks, C, ldc, M, N, K, bias, addend, ld_add, act, mask, ld_mask, mask_scale,
                            workspace, workspace_bytes, 0, stream);
}

// 1: every product runs on the general (VGPR-staged) kernel, 0: the LDS-DMA kernel where the operand layouts allow (default)
int lctvqa_gemm_force_general(int on) { int old = g_gemm_force_general; g_gemm_force_general = on ? 1 : 0; return old; }

// C[b] = act(A[b] . B[b]^T + bias) for b < batch, operands in either layout the LDS-DMA kernel stages (K-contiguous: X(r,k) =
// X[r*rs + k]; row-contiguous: X(r,k) = X[k*ks + r]; 16-byte aligned, K % 4 == 0, rows % 4 == 0 for a row-contiguous operand),
// element strides sA, sB, sC between the batch members.
int lctvqa_gemm_f32_batched(const float* A, long a_ms, long a_ks, long sA, const float* B, long b_ns, long b_ks, long sB, float* C,
                            long ldc, long sC, int batch, int M, int N, int K, const float* bias, int act, void* stream) {
  LCT_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0 && batch < 65536, "gemm_f32_batched: bad shape %d %d %d x%d", M, N, K, batch);
  const int la = dma_layout(A, a_ms, a_ks, M, K), lb = dma_layout(B, b_ns, b_ks, N, K);
  LCT_REQUIRE(la >= 0 && lb >= 0 && sA % 4 == 0 && sB % 4 == 0,
              "gemm_f32_batched: operands must be 16-byte aligned, K- or row-contiguous, K %% 4 == 0 (rows %% 4 == 0 when row-contiguous)");
  GemmEpi epi{nullptr, nullptr, 1.f, bias, nullptr, 0, nullptr, 0, 1.f, act};
  return launch_gemm_dma(A, la ? a_ks : a_ms, la, B, lb ? b_ks : b_ns, lb, C, ldc, M, N, K, epi, nullptr, 0, batch, sA, sB, sC,
                         (hipStream_t)stream);
}

// the K-contiguous special case (nn.Linear over a stack of inputs)
int lctvqa_gemm_f32_batched_nt(const float* A, long lda, long sA, const float* B, long ldb, long sB, float* C, long ldc, long sC,
                               int batch, int M, int N, int K, const float* bias, int act, void* stream) {
  return lctvqa_gemm_f32_batched(A, lda, 1, sA, B, ldb, 1, sB, C, ldc, sC, batch, M, N, K, bias, act, stream);
}

long lctvqa_colsum_scratch_elems(int M, int N) { return (long)colsum_splits(M, N) * N; }

// Fusion head, first half (reference basic_vqa/models.py:96-100): one GEMM launch computes
//   h1 = mask2*scale2 * tanh( [mask1*scale1 * tanh(img (.) qst)] . W1^T + b1 )
// with the pointwise mul+tanh+dropout applied to the A operand as it is staged (never materialised).
int lctvqa_fusion_fc1(const float* img, const float* qst, const uint8_t* mask1, float scale1, const float* w1, const float* b1,
                      const uint8_t* mask2, float scale2, float* h1, int B, int E, int A, float* workspace, long workspace_bytes,
                      void* stream) {
  g_fuse_a2 = qst; g_fuse_amask = mask1; g_fuse_ascale = scale1;
  int rc = lctvqa_gemm_f32_ex(img, E, 1, w1, E, 1, h1, A, B, A, E, b1, nullptr, 0, 2, mask2, A, scale2, workspace, workspace_bytes, 4,
                              stream);
  g_fuse_a2 = nullptr; g_fuse_amask = nullptr;
  return rc;
}

int lctvqa_colsum_f32(const float* X, long ld, int M, int N, float* out, float* scratch, void* stream) {
  LCT_REQUIRE(M > 0 && N > 0 && scratch != nullptr, "colsum: bad args");
  int S = colsum_splits(M, N);
  int rows_per = cdiv(M, S);
  S = cdiv(M, rows_per);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(N, 64), S), dim3(256), 0, (hipStream_t)stream, X, ld, M, N, rows_per, scratch);
  int rc = lct_check_launch("colsum_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(N, 64)), dim3(1024), 0, (hipStream_t)stream, scratch, S, N, out);
  return lct_check_launch("colsum_finish");
}

}  // extern "C"
