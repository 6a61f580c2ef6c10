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
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = (row < nrows && k + i < kend) ? P[(long)row * rs + (long)(k + i) * ks] : 0.f;
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

struct GemmEpi {
  // fused A-operand prologue of the fusion head (reference basic_vqa/models.py:96-98 feeding fc1):
  //   A(m,k) := amask[m][k] ? ascale * tanh(A(m,k) * a2[m*a_ms + k*a_ks]) : 0      (a2 == null: plain A)
  const float* a2;
  const uint8_t* amask;  // dense [M][K] or null
  float ascale;
  const float* bias;     // [N] or null
  const float* addend;   // [M][ld_add] or null
  long ld_add;
  const uint8_t* mask;   // [M][ld_mask] or null (0/1)
  long ld_mask;
  float mask_scale;
  int act;               // 0 none, 1 relu, 2 tanh
};

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

__device__ __forceinline__ float apply_epi(float v, int m, int n, const GemmEpi& e) {
  if (e.bias) v += e.bias[n];
  if (e.addend) v += e.addend[(long)m * e.ld_add + n];
  if (e.act == 1) v = fmaxf(v, 0.f);
  else if (e.act == 2) v = tanhf(v);
  if (e.mask) v = e.mask[(long)m * e.ld_mask + n] ? v * e.mask_scale : 0.f;
  return v;
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
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int n = n0 + wn * 64 + b * 32 + j;
        if (m < M && n < N) {
          float v = acc[a][b][r];
          if (direct) v = apply_epi(v, m, n, epi);
          out[(long)m * ldc + n] = v;
        }
      }
}

__global__ void gemm_splitk_reduce_kernel(const float* __restrict__ ws, int splits, float* __restrict__ C, long ldc,
                                          int M, int N, GemmEpi epi) {
  long total = (long)M * N;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    int n = o % N, m = o / N;
    float v = 0.f;
    for (int s = 0; s < splits; s++) v += ws[(long)s * total + o];
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
    for (int m = r0 + w; m < r1; m += 4) a += X[(long)m * ld + n];
  red[w][threadIdx.x & 63] = a;
  __syncthreads();
  if (w == 0 && n < N) part[(long)blockIdx.y * N + n] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void colsum_finish_kernel(const float* __restrict__ part, int S, int N, float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float a = 0.f;
  for (int s = 0; s < S; s++) a += part[(long)s * N + n];
  out[n] = a;
}

static int pick_mode(const float* P, long rs, long ks, int K) {
  if (ks == 1) {
    if (rs % 4 == 0 && K % 4 == 0 && ((uintptr_t)P & 15) == 0) return LD_KVEC;
    return LD_KSCALAR;
  }
  if (rs == 1) return LD_MCONTIG;
  return LD_KSCALAR;   // fully general strides
}

static thread_local const float* g_fuse_a2 = nullptr;
static thread_local const uint8_t* g_fuse_amask = nullptr;
static thread_local float g_fuse_ascale = 1.f;

typedef void (*gemm_kern_t)(const float*, long, long, const float*, long, long, float*, long, int, int, int, int,
                            GemmEpi, int, int);
static gemm_kern_t gemm_table[3][3] = {
    {gemm_f32_kernel<0, 0>, gemm_f32_kernel<0, 1>, gemm_f32_kernel<0, 2>},
    {gemm_f32_kernel<1, 0>, gemm_f32_kernel<1, 1>, gemm_f32_kernel<1, 2>},
    {gemm_f32_kernel<2, 0>, gemm_f32_kernel<2, 1>, gemm_f32_kernel<2, 2>}};

extern "C" {

// workspace bytes needed for the split-K slabs of this problem (0 when no split is used)
long lctvqa_gemm_workspace_bytes(int M, int N, int K) {
  int tiles = cdiv(M, 128) * cdiv(N, 128);
  int splits = 1;
  if (tiles < 192) {
    splits = 512 / tiles;
    int maxs = K / 64;
    if (splits > maxs) splits = maxs;
    int cap = K >= 65536 ? 512 : 32;         // pixel-reduction GEMMs (1x1-conv weight gradients: K = B*H*W) get deep splits
    if (splits > cap) splits = cap;
    if (splits < 1) splits = 1;
  }
  return splits > 1 ? (long)splits * M * N * (long)sizeof(float) : 0;
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
  int tiles = cdiv(M, 128) * cdiv(N, 128);
  long need = lctvqa_gemm_workspace_bytes(M, N, K);
  int splits = need ? (int)(need / ((long)M * N * sizeof(float))) : 1;
  if (splits > 1 && (workspace == nullptr || workspace_bytes < need)) splits = 1;   // degrade, still correct
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
  long total = (long)M * N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, workspace, splits, C, ldc, M, N, epi);
  return lct_check_launch("gemm_splitk_reduce");
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
  hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, scratch, S, N, out);
  return lct_check_launch("colsum_finish");
}

}  // extern "C"
