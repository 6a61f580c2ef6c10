// HBM-bound pointwise / gather / reduction kernels of the VQA head and the question encoder.
// All of them are memory-bound: one pass, 16-B accesses where rows allow it, no reshaping into GEMMs.
#include "common.h"

// ------------------------------------------------------------------------------------------
// word2vec gather (+tanh) with the (B,T)->(T,B) transpose fused:  reference basic_vqa/models.py:58-60,
// models_lct.py:101-104 (tanh) and models_lct.py:153-154 (no tanh for generated tokens).
// idx [B][T] int64 -> out [T][B][E]
// ------------------------------------------------------------------------------------------
__global__ void embed_gather_kernel(const long* __restrict__ idx, const float* __restrict__ emb, float* __restrict__ out,
                                    int B, int T, int E, int V, int apply_tanh) {
  const int row = blockIdx.x;            // t*B + b
  const int t = row / B, b = row % B;
  long id = idx[(long)b * T + t];
  if (id < 0 || id >= V) id = 0;         // host validates; stay in bounds regardless
  const float* src = emb + id * E;
  float* dst = out + (long)row * E;
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float v = src[e];
    dst[e] = apply_tanh ? tanhf(v) : v;
  }
}

// backward: demb[idx[b,t]] += dy[t,b,:] * (1 - y^2)   (dense grad, nn.Embedding(sparse=False))
__global__ void embed_scatter_kernel(const long* __restrict__ idx, const float* __restrict__ dy,
                                     const float* __restrict__ y, float* __restrict__ demb, int B, int T, int E, int V,
                                     int apply_tanh) {
  const int row = blockIdx.x;
  const int t = row / B, b = row % B;
  long id = idx[(long)b * T + t];
  if (id < 0 || id >= V) return;
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float g = dy[(long)row * E + e];
    if (apply_tanh) {
      float v = y[(long)row * E + e];
      g *= (1.f - v * v);
    }
    atomicAdd(demb + id * E + e, g);
  }
}

// ------------------------------------------------------------------------------------------
// LSTM cell pointwise (gate order i,f,g,o = nn.LSTM; reference basic_vqa/models.py:52,61).
// pre = xp[b][4H] + hp[b][4H]  (xp already holds b_ih + b_hh)
// ------------------------------------------------------------------------------------------
__global__ void lstm_cell_fwd_kernel(const float* __restrict__ xp, const float* __restrict__ hp,
                                     const float* __restrict__ c_prev, float* __restrict__ gates,
                                     float* __restrict__ c_out, float* __restrict__ h_out, int B, int H) {
  long total = (long)B * H;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    int u = o % H;
    long b = o / H;
    long g0 = b * 4 * H + u;
    float i = sigmoidf_(xp[g0] + hp[g0]);
    float f = sigmoidf_(xp[g0 + H] + hp[g0 + H]);
    float g = tanhf(xp[g0 + 2 * H] + hp[g0 + 2 * H]);
    float oo = sigmoidf_(xp[g0 + 3 * H] + hp[g0 + 3 * H]);
    float c = f * c_prev[o] + i * g;
    gates[g0] = i; gates[g0 + H] = f; gates[g0 + 2 * H] = g; gates[g0 + 3 * H] = oo;
    c_out[o] = c;
    h_out[o] = oo * tanhf(c);
  }
}

// dh = dh_rec (from t+1 through W_hh, may be null) + dh_ext (from heads at this t, may be null)
// dc = dc_next (may be null)
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ dh_rec, const float* __restrict__ dh_ext,
                                     const float* __restrict__ dc_next, const float* __restrict__ gates,
                                     const float* __restrict__ c_prev, const float* __restrict__ c_cur,
                                     float* __restrict__ dgates, float* __restrict__ dc_prev, int B, int H) {
  long total = (long)B * H;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    int u = o % H;
    long b = o / H;
    long g0 = b * 4 * H + u;
    float i = gates[g0], f = gates[g0 + H], g = gates[g0 + 2 * H], oo = gates[g0 + 3 * H];
    float dh = (dh_rec ? dh_rec[o] : 0.f) + (dh_ext ? dh_ext[o] : 0.f);
    float tc = tanhf(c_cur[o]);
    float dc = (dc_next ? dc_next[o] : 0.f) + dh * oo * (1.f - tc * tc);
    dgates[g0] = dc * g * i * (1.f - i);
    dgates[g0 + H] = dc * c_prev[o] * f * (1.f - f);
    dgates[g0 + 2 * H] = dc * i * (1.f - g * g);
    dgates[g0 + 3 * H] = dh * tc * oo * (1.f - oo);
    dc_prev[o] = dc * f;
  }
}

// ------------------------------------------------------------------------------------------
// generic small pointwise ops
// ------------------------------------------------------------------------------------------
// out[b][0:H] = tanh(h[b]), out[b][H:2H] = tanh(c[b])   (models.py:62-65 with num_layers = 1)
__global__ void cat_tanh_kernel(const float* __restrict__ h, const float* __restrict__ c, float* __restrict__ out, int B, int H) {
  long total = (long)B * 2 * H;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    int k = o % (2 * H);
    long b = o / (2 * H);
    out[o] = tanhf(k < H ? h[b * H + k] : c[b * H + k - H]);
  }
}
// dh[b] = dy[b][0:H]*(1-y^2), dc[b] = dy[b][H:2H]*(1-y^2)
__global__ void cat_tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dh,
                                    float* __restrict__ dc, int B, int H) {
  long total = (long)B * 2 * H;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    int k = o % (2 * H);
    long b = o / (2 * H);
    float g = dy[o] * (1.f - y[o] * y[o]);
    if (k < H) dh[b * H + k] = g; else dc[b * H + k - H] = g;
  }
}
// y = tanh(x)
__global__ void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) y[o] = tanhf(x[o]);
}
// dx = dy * (1 - y^2) * (mask ? mask*scale : 1)     [act = 2]
// dx = dy * (y > 0)   * (mask ? mask*scale : 1)     [act = 1]
// dx = dy             * (mask ? mask*scale : 1)     [act = 0]
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const uint8_t* __restrict__ mask,
                               float scale, float* __restrict__ dx, long n, int act) {
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) {
    float g = dy[o];
    float v = y[o];
    if (mask) {
      if (mask[o]) { g *= scale; v = v / scale; } else { g = 0.f; v = 0.f; }   // y holds the post-mask value
    }
    if (act == 2) g *= (1.f - v * v);
    else if (act == 1) g = (v > 0.f) ? g : 0.f;
    dx[o] = g;
  }
}

// z = mask*scale*tanh(a*b)   (reference basic_vqa/models.py:96-98); t = tanh(a*b) kept for backward
__global__ void mul_tanh_drop_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                         const uint8_t* __restrict__ mask, float scale, float* __restrict__ t,
                                         float* __restrict__ z, long n) {
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) {
    float v = tanhf(a[o] * b[o]);
    t[o] = v;
    z[o] = mask ? (mask[o] ? v * scale : 0.f) : v;
  }
}
__global__ void mul_tanh_drop_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ a,
                                         const float* __restrict__ b, const float* __restrict__ t,
                                         const uint8_t* __restrict__ mask, float scale, float* __restrict__ da,
                                         float* __restrict__ db, long n) {
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) {
    float g = dz[o];
    if (mask) g = mask[o] ? g * scale : 0.f;
    g *= (1.f - t[o] * t[o]);
    if (da) da[o] = g * b[o];
    if (db) db[o] = g * a[o];
  }
}

// row-wise L2 normalisation with DETACHED norm (reference basic_vqa/models.py:39-40): y = x/||x||, inv = 1/||x||
__global__ void l2norm_rows_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv, int N) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  float s = 0.f;
  for (int e = threadIdx.x; e < N; e += blockDim.x) { float v = x[row * N + e]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  const float r = 1.f / sqrtf(s);
  if (threadIdx.x == 0) inv[row] = r;
  for (int e = threadIdx.x; e < N; e += blockDim.x) y[row * N + e] = x[row * N + e] * r;
}
__global__ void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, int N, long n) {
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) y[o] = x[o] * s[o / N];
}

// ------------------------------------------------------------------------------------------
// losses:  CrossEntropyLoss (mean) and softXEnt (reference basic_vqa/models.py:7-10,89,115).
// One workgroup per row; writes per-row loss and (optionally) d logits scaled by gscale.
// kind 0: hard labels (labels int64 [M]); kind 1: soft targets (target f32 [M][N]).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xent_rows_kernel(const float* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                       const float* __restrict__ target, long ldt, float* __restrict__ row_loss,
                                                       float* __restrict__ dlogits, long ldd, float gscale, int N, int kind,
                                                       float* __restrict__ dtarget) {
  __shared__ float red[4];
  __shared__ float red2[4];
  const long row = blockIdx.x;
  const float* x = logits + row * ld;
  float mx = -INFINITY;
#pragma unroll 8
  for (int e = threadIdx.x; e < N; e += blockDim.x) mx = fmaxf(mx, x[e]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f, st = 0.f, stx = 0.f;
#pragma unroll 4
  for (int e = threadIdx.x; e < N; e += blockDim.x) {
    float v = x[e];
    se += expf(v - mx);
    if (kind == 1) { float t = target[row * ldt + e]; st += t; stx += t * v; }
  }
  se = wave_sum(se); st = wave_sum(st); stx = wave_sum(stx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = se;
  __syncthreads();
  se = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  if (kind == 1) {
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = st; red2[threadIdx.x >> 6] = stx; }
    __syncthreads();
    st = red[0] + red[1] + red[2] + red[3];
    stx = red2[0] + red2[1] + red2[2] + red2[3];
  }
  const float lse = mx + logf(se);
  long lab = 0;
  if (kind == 0) { lab = labels[row]; if (lab < 0 || lab >= N) lab = 0; }
  if (threadIdx.x == 0) row_loss[row] = (kind == 0) ? (lse - x[lab]) : (st * lse - stx);
  if (dlogits) {
#pragma unroll 4
    for (int e = threadIdx.x; e < N; e += blockDim.x) {
      float p = expf(x[e] - lse);
      float g = (kind == 0) ? (p - (e == lab ? 1.f : 0.f)) : (p * st - target[row * ldt + e]);
      dlogits[row * ldd + e] = g * gscale;
    }
  }
  if (dtarget && kind == 1)                       // d(-sum t*logp)/dt = -(x - lse)
    for (int e = threadIdx.x; e < N; e += blockDim.x) dtarget[row * (long)N + e] = (lse - x[e]) * gscale;
}

// backward of y = softmax(x * inv_temp) per row: dx = inv_temp * y * (dy - sum(dy*y))
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                               float* __restrict__ dx, int N, float inv_temp) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  float s = 0.f;
  for (int e = threadIdx.x; e < N; e += blockDim.x) s += dy[row * N + e] * y[row * N + e];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  for (int e = threadIdx.x; e < N; e += blockDim.x) dx[row * N + e] = inv_temp * y[row * N + e] * (dy[row * N + e] - s);
}

// deterministic sum of n floats scaled (single workgroup): out[0] = scale * sum(x)
__global__ __launch_bounds__(1024) void sum_scale_kernel(const float* __restrict__ x, long n, float scale, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (long o = threadIdx.x; o < n; o += blockDim.x) s += x[o];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += red[i];
    out[0] = t * scale;
  }
}

// row softmax with temperature (pseudo-label softmax: experiment.py:245, architect_lct.py:55)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int N, float inv_temp) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  float mx = -INFINITY;
  for (int e = threadIdx.x; e < N; e += blockDim.x) mx = fmaxf(mx, x[row * N + e] * inv_temp);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f;
  for (int e = threadIdx.x; e < N; e += blockDim.x) se += expf(x[row * N + e] * inv_temp - mx);
  se = wave_sum(se);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = se;
  __syncthreads();
  se = red[0] + red[1] + red[2] + red[3];
  const float r = 1.f / se;
  for (int e = threadIdx.x; e < N; e += blockDim.x) y[row * N + e] = expf(x[row * N + e] * inv_temp - mx) * r;
}

// row argmax (first maximal index, like torch.argmax) -> int64
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, long ld, int N, long* __restrict__ out) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const long row = blockIdx.x;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  const float* xr = x + row * ld;
  int e = threadIdx.x;
  for (; e + 7 * 256 < N; e += 8 * 256) {                       // eight loads in flight, compared in index order (a 17 858-wide row was
    float v[8];                                                  // a chain of 70 dependent HBM latencies per thread)
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = xr[e + 256 * u];
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (v[u] > best || (v[u] == best && e + 256 * u < idx)) { best = v[u]; idx = e + 256 * u; }
  }
  for (; e < N; e += 256) {
    float v = xr[e];
    if (v > best || (v == best && e < idx)) { best = v; idx = e; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    out[row] = idx;
  }
}

// Validation counters of one batch (reference basic_vqa/experiment.py:314-324, misc.py:42-45): for every sample the
// row argmax of the answer logits is compared with its K multiple-choice answers.
//   counts[0] += #samples whose prediction is one of its choices                       ("<unk> accepted", Exp1)
//   counts[1] += the same with a prediction equal to unk_idx replaced by -9999          ("<unk> rejected", Exp2)
// choices: [K][M] int64 (the reference's list of K LongTensors stacked).  pred_out (optional): the argmax per row.
__global__ __launch_bounds__(256) void eval_multi_choice_kernel(const float* __restrict__ x, long ld, int N,
                                                                const long* __restrict__ choices, int K, int M, long unk_idx,
                                                                long* __restrict__ pred_out, long* __restrict__ counts) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const long row = blockIdx.x;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int e = threadIdx.x; e < N; e += blockDim.x) {
    float v = x[row * ld + e];
    if (v > best || (v == best && e < idx)) { best = v; idx = e; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    if (pred_out) pred_out[row] = idx;
    const long p1 = idx, p2 = (idx == unk_idx) ? -9999 : (long)idx;
    int hit1 = 0, hit2 = 0;
    for (int k = 0; k < K; k++) {
      const long c = choices[(long)k * M + row];
      hit1 |= (c == p1);
      hit2 |= (c == p2);
    }
    if (hit1) atomicAdd((unsigned long long*)&counts[0], 1ULL);
    if (hit2) atomicAdd((unsigned long long*)&counts[1], 1ULL);
  }
}

// ------------------------------------------------------------------------------------------
// dropout keep-mask from a counter-based generator (seed, offset): mask[i] = U(i) >= p.
// (PyTorch's CPU Bernoulli stream cannot be reproduced on the GPU; parity tests inject masks.)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 32);
}
__global__ void dropout_mask_kernel(uint8_t* __restrict__ mask, long n, uint64_t seed, uint64_t offset, float p) {
  const uint32_t thr = (uint32_t)(p * 4294967296.0);
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x)
    mask[o] = mix32(seed * 0xD1342543DE82EF95ull + offset + (uint64_t)o) >= thr ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// optimiser: global grad-norm clip (clip_grad_norm_(5), experiment.py:226,256) + Adam
// (model_factory.py:20-21,41-42) over ONE flat parameter arena: 16 B/param read + 12 B/param write.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  const long n4 = n >> 2;
  const f32x4* g4 = (const f32x4*)g;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n4; o += (long)gridDim.x * blockDim.x) {
    f32x4 v = g4[o];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// norm_out[0] = sqrt(sum(part[0:np]) + extra_sq)
__global__ void norm_finish_kernel(const float* __restrict__ part, int np, float* __restrict__ norm_out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += blockDim.x) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) norm_out[0] = sqrtf(red[0] + red[1] + red[2] + red[3]);
}
__device__ __forceinline__ void adam_one(float& pp, float gg, float& mm, float& vv, float coef, float wd, float b1, float b2,
                                         float step, float rs2, float eps) {
  gg *= coef;
  if (wd != 0.f) gg += wd * pp;
  mm = b1 * mm + (1.f - b1) * gg;
  vv = b2 * vv + (1.f - b2) * gg * gg;
  pp = pp - step * mm / (sqrtf(vv) * rs2 + eps);
}
__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, long n, const float* __restrict__ total_norm, float max_norm,
                                 float lr, float b1, float b2, float eps, float bc1, float bc2, float wd) {
  float coef = 1.f;
  if (total_norm) {
    coef = max_norm / (total_norm[0] + 1e-6f);
    coef = coef < 1.f ? coef : 1.f;
  }
  const float step = lr / bc1, rs2 = 1.f / sqrtf(bc2);
  const long n4 = n >> 2;
  f32x4* p4 = (f32x4*)p; const f32x4* g4 = (const f32x4*)g; f32x4* m4 = (f32x4*)m; f32x4* v4 = (f32x4*)v;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n4; o += (long)gridDim.x * blockDim.x) {
    f32x4 pp = p4[o], gg = g4[o], mm = m4[o], vv = v4[o];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float a = pp[k], b = mm[k], c = vv[k];
      adam_one(a, gg[k], b, c, coef, wd, b1, b2, step, rs2, eps);
      pp[k] = a; mm[k] = b; vv[k] = c;
    }
    p4[o] = pp; m4[o] = mm; v4[o] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long o = (n4 << 2) + threadIdx.x;
    float pp = p[o], mm = m[o], vv = v[o];
    adam_one(pp, g[o], mm, vv, coef, wd, b1, b2, step, rs2, eps);
    p[o] = pp; m[o] = mm; v[o] = vv;
  }
}
// y = a*x + y  (architect vector ops on the flat arena: architect_lct.py:112-116,194-207)
__global__ void axpy_kernel(float a, const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) y[o] = fmaf(a, x[o], y[o]);
}
__global__ void scale_kernel(float a, float* __restrict__ y, long n) {
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) y[o] *= a;
}

static inline int ew_blocks(long n, int per = 256) {
  long b = (n + per - 1) / per;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" {

int lctvqa_embed_gather(const long* idx, const float* emb, float* out, int B, int T, int E, int V, int apply_tanh, void* stream) {
  LCT_REQUIRE(B > 0 && T > 0 && E > 0 && V > 0, "embed_gather: bad shape");
  hipLaunchKernelGGL(embed_gather_kernel, dim3(B * T), dim3(128), 0, (hipStream_t)stream, idx, emb, out, B, T, E, V, apply_tanh);
  return lct_check_launch("embed_gather");
}
int lctvqa_embed_scatter_add(const long* idx, const float* dy, const float* y, float* demb, int B, int T, int E, int V,
                             int apply_tanh, void* stream) {
  LCT_REQUIRE(B > 0 && T > 0 && E > 0 && V > 0, "embed_scatter_add: bad shape");
  hipLaunchKernelGGL(embed_scatter_kernel, dim3(B * T), dim3(128), 0, (hipStream_t)stream, idx, dy, y, demb, B, T, E, V, apply_tanh);
  return lct_check_launch("embed_scatter_add");
}
int lctvqa_lstm_cell_fwd(const float* xp, const float* hp, const float* c_prev, float* gates, float* c_out, float* h_out,
                         int B, int H, void* stream) {
  LCT_REQUIRE(B > 0 && H > 0, "lstm_cell_fwd: bad shape");
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(ew_blocks((long)B * H)), dim3(256), 0, (hipStream_t)stream, xp, hp, c_prev, gates, c_out, h_out, B, H);
  return lct_check_launch("lstm_cell_fwd");
}
int lctvqa_lstm_cell_bwd(const float* dh_rec, const float* dh_ext, const float* dc_next, const float* gates,
                         const float* c_prev, const float* c_cur, float* dgates, float* dc_prev, int B, int H, void* stream) {
  LCT_REQUIRE(B > 0 && H > 0, "lstm_cell_bwd: bad shape");
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(ew_blocks((long)B * H)), dim3(256), 0, (hipStream_t)stream, dh_rec, dh_ext, dc_next, gates, c_prev, c_cur, dgates, dc_prev, B, H);
  return lct_check_launch("lstm_cell_bwd");
}
int lctvqa_cat_tanh_fwd(const float* h, const float* c, float* out, int B, int H, void* stream) {
  hipLaunchKernelGGL(cat_tanh_kernel, dim3(ew_blocks((long)B * 2 * H)), dim3(256), 0, (hipStream_t)stream, h, c, out, B, H);
  return lct_check_launch("cat_tanh_fwd");
}
int lctvqa_cat_tanh_bwd(const float* dy, const float* y, float* dh, float* dc, int B, int H, void* stream) {
  hipLaunchKernelGGL(cat_tanh_bwd_kernel, dim3(ew_blocks((long)B * 2 * H)), dim3(256), 0, (hipStream_t)stream, dy, y, dh, dc, B, H);
  return lct_check_launch("cat_tanh_bwd");
}
int lctvqa_tanh_fwd(const float* x, float* y, long n, void* stream) {
  hipLaunchKernelGGL(tanh_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  return lct_check_launch("tanh_fwd");
}
int lctvqa_act_bwd(const float* dy, const float* y, const uint8_t* mask, float scale, float* dx, long n, int act, void* stream) {
  LCT_REQUIRE(act >= 0 && act <= 2, "act_bwd: bad act");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, y, mask, scale, dx, n, act);
  return lct_check_launch("act_bwd");
}
int lctvqa_mul_tanh_drop_fwd(const float* a, const float* b, const uint8_t* mask, float scale, float* t, float* z, long n, void* stream) {
  hipLaunchKernelGGL(mul_tanh_drop_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, mask, scale, t, z, n);
  return lct_check_launch("mul_tanh_drop_fwd");
}
int lctvqa_mul_tanh_drop_bwd(const float* dz, const float* a, const float* b, const float* t, const uint8_t* mask, float scale,
                             float* da, float* db, long n, void* stream) {
  hipLaunchKernelGGL(mul_tanh_drop_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dz, a, b, t, mask, scale, da, db, n);
  return lct_check_launch("mul_tanh_drop_bwd");
}
int lctvqa_l2norm_rows_fwd(const float* x, float* y, float* inv_norm, int M, int N, void* stream) {
  LCT_REQUIRE(M > 0 && N > 0, "l2norm_rows: bad shape");
  hipLaunchKernelGGL(l2norm_rows_fwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, y, inv_norm, N);
  return lct_check_launch("l2norm_rows_fwd");
}
int lctvqa_scale_rows(const float* x, const float* s, float* y, int M, int N, void* stream) {
  hipLaunchKernelGGL(scale_rows_kernel, dim3(ew_blocks((long)M * N)), dim3(256), 0, (hipStream_t)stream, x, s, y, N, (long)M * N);
  return lct_check_launch("scale_rows");
}
// loss_out[0] = loss_scale * sum_rows(row loss);  dlogits = gscale * d(row loss)/d(logits) (may be null)
int lctvqa_xent(const float* logits, long ld, const long* labels, const float* target, long ldt, float* row_loss,
                float* loss_out, float loss_scale, float* dlogits, long ldd, float gscale, float* dtarget, int M, int N, int kind,
                void* stream) {
  LCT_REQUIRE(M > 0 && N > 0 && (kind == 0 || kind == 1), "xent: bad args");
  LCT_REQUIRE(kind == 0 ? labels != nullptr : target != nullptr, "xent: missing labels/target");
  hipLaunchKernelGGL(xent_rows_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, target, ldt, row_loss, dlogits, ldd, gscale, N, kind, dtarget);
  int rc = lct_check_launch("xent_rows");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_loss, (long)M, loss_scale, loss_out);
  return lct_check_launch("xent_sum");
}
int lctvqa_softmax_rows(const float* x, float* y, int M, int N, float inv_temp, void* stream) {
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, y, N, inv_temp);
  return lct_check_launch("softmax_rows");
}
int lctvqa_softmax_rows_bwd(const float* dy, const float* y, float* dx, int M, int N, float inv_temp, void* stream) {
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, dy, y, dx, N, inv_temp);
  return lct_check_launch("softmax_rows_bwd");
}
int lctvqa_argmax_rows(const float* x, long ld, int M, int N, long* out, void* stream) {
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, ld, N, out);
  return lct_check_launch("argmax_rows");
}
int lctvqa_eval_multi_choice(const float* x, long ld, int M, int N, const long* choices, int K, long unk_idx, long* pred_out,
                             long* counts2, void* stream) {
  LCT_REQUIRE(M >= 0 && N > 0 && K >= 0, "eval_multi_choice: bad sizes");
  if (M == 0) return 0;
  hipLaunchKernelGGL(eval_multi_choice_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, ld, N, choices, K, M, unk_idx, pred_out,
                     counts2);
  return lct_check_launch("eval_multi_choice");
}
int lctvqa_dropout_mask(uint8_t* mask, long n, unsigned long long seed, unsigned long long offset, float p, void* stream) {
  LCT_REQUIRE(p >= 0.f && p < 1.f, "dropout_mask: bad p");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, mask, n, (uint64_t)seed, (uint64_t)offset, p);
  return lct_check_launch("dropout_mask");
}
// total_norm_out[0] = ||g||_2 over the flat arena;  scratch: >= 1024 floats
int lctvqa_grad_norm(const float* g, long n, float* scratch, float* total_norm_out, void* stream) {
  LCT_REQUIRE(((uintptr_t)g & 15) == 0, "grad_norm: arena must be 16-B aligned");
  int nb = ew_blocks(n, 2048);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, n, scratch);
  int rc = lct_check_launch("sumsq_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(norm_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, nb, total_norm_out);
  return lct_check_launch("norm_finish");
}
// Adam step over a flat arena; total_norm (device scalar, may be null) enables the fused global-norm clip
int lctvqa_adam_step(float* p, const float* g, float* m, float* v, long n, const float* total_norm, float max_norm,
                     float lr, float b1, float b2, float eps, int step, float weight_decay, void* stream) {
  LCT_REQUIRE(step >= 1, "adam_step: step counts from 1");
  LCT_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: buffers must be 16-B aligned");
  float bc1 = (float)(1.0 - pow((double)b1, (double)step)), bc2 = (float)(1.0 - pow((double)b2, (double)step));
  hipLaunchKernelGGL(adam_clip_kernel, dim3(ew_blocks(n, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, total_norm, max_norm, lr, b1, b2, eps, bc1, bc2, weight_decay);
  return lct_check_launch("adam_step");
}
int lctvqa_axpy(float a, const float* x, float* y, long n, void* stream) {
  hipLaunchKernelGGL(axpy_kernel, dim3(ew_blocks(n, 1024)), dim3(256), 0, (hipStream_t)stream, a, x, y, n);
  return lct_check_launch("axpy");
}
int lctvqa_scale(float a, float* y, long n, void* stream) {
  hipLaunchKernelGGL(scale_kernel, dim3(ew_blocks(n, 1024)), dim3(256), 0, (hipStream_t)stream, a, y, n);
  return lct_check_launch("scale");
}

}  // extern "C"

// y *= a[0] with the scalar read from device memory (autograd grad_output of a scalar loss)
__global__ void scale_dev_kernel(const float* __restrict__ a, const float* __restrict__ x, float* __restrict__ y, long n) {
  const float s = a[0];
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) y[o] = x[o] * s;
}
extern "C" int lctvqa_scale_by_device_scalar(const float* a, const float* x, float* y, long n, void* stream) {
  hipLaunchKernelGGL(scale_dev_kernel, dim3(ew_blocks(n, 1024)), dim3(256), 0, (hipStream_t)stream, a, x, y, n);
  return lct_check_launch("scale_by_device_scalar");
}

// ------------------------------------------------------------------------------------------
// Fused LSTM time step (nn.LSTM, reference basic_vqa/models.py:61, models_lct.py:107-108):
//   pre = xp[b][4H] (input projection + b_ih, precomputed for all t) + h_prev[b][:] . W_hh^T + b_hh
//   i,f,g,o -> c = f*c_prev + i*g ; h = o*tanh(c)
// One launch per step, no split-K slabs, no separate pointwise pass: a workgroup owns 32 batch rows x
// 32 hidden units and ALL FOUR gates of those units (4 accumulator tiles per wave on
// v_mfma_f32_32x32x2_f32), its 4 waves split K, partial tiles meet in LDS and the cell update is the
// epilogue.  h rows and W_hh rows are read as 16-B vectors straight from L2 (W_hh = 4 MB stays L2/MALL
// resident across the T steps).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_step_fused_kernel(const float* __restrict__ xp, const float* __restrict__ h_prev,
                                                              const float* __restrict__ c_prev, const float* __restrict__ w_hh,
                                                              const float* __restrict__ b_hh, float* __restrict__ gates,
                                                              float* __restrict__ c_out, float* __restrict__ h_out, int B, int H) {
  extern __shared__ __attribute__((aligned(16))) float red[];      // [4 waves][4 gates][16 regs][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  const int nut = H / 32;
  const int ut = blockIdx.x % nut, bt = blockIdx.x / nut;
  const int b0 = bt * 32, u0 = ut * 32;
  f32x16 acc[4];
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[g][r] = 0.f;
  const int arow = min(b0 + j, B - 1);                 // clamp: rows >= B are computed but never stored
  const float* ap = h_prev + (long)arow * H;
  const int ngroups = H / 16;
  const float* bp = w_hh + (long)(u0 + j) * H;
  // register double buffer over 16-deep k groups: the loads of group n+1 are in flight under the MFMAs of group n
  f32x4 A0[2], B0[2][4], A1[2], B1[2][4];
  auto ld = [&](int grp, f32x4 (&A)[2], f32x4 (&Bf)[2][4]) {
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      const int kc = grp * 16 + (2 * kk + hh) * 4;
      A[kk] = *(const f32x4*)(ap + kc);
#pragma unroll
      for (int g = 0; g < 4; g++) Bf[kk][g] = *(const f32x4*)(bp + (long)g * H * H + kc);
    }
  };
  auto mm = [&](const f32x4 (&A)[2], const f32x4 (&Bf)[2][4]) {
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int g = 0; g < 4; g++) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[kk][e], Bf[kk][g][e], acc[g], 0, 0, 0);
  };
  int grp = wave;
  if (grp < ngroups) ld(grp, A0, B0);
  while (grp < ngroups) {
    if (grp + 4 < ngroups) ld(grp + 4, A1, B1);
    mm(A0, B0);
    grp += 4;
    if (grp >= ngroups) break;
    if (grp + 4 < ngroups) ld(grp + 4, A0, B0);
    mm(A1, B1);
    grp += 4;
  }
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int r = 0; r < 16; r++) red[((wave * 4 + g) * 16 + r) * 64 + lane] = acc[g][r];
  __syncthreads();
  // wave w finishes accumulator registers 4w..4w+3 of all four gates
#pragma unroll
  for (int rr = 0; rr < 4; rr++) {
    const int r = wave * 4 + rr;
    const int row = b0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
    const int u = u0 + j;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; w++) s += red[((w * 4 + g) * 16 + r) * 64 + lane];
      pre[g] = s;
    }
    if (row < B) {
      const long g0 = (long)row * 4 * H + u;
      const long o = (long)row * H + u;
      const float i = sigmoidf_(pre[0] + xp[g0] + b_hh[u]);
      const float f = sigmoidf_(pre[1] + xp[g0 + H] + b_hh[H + u]);
      const float g = tanhf(pre[2] + xp[g0 + 2 * H] + b_hh[2 * H + u]);
      const float oo = sigmoidf_(pre[3] + xp[g0 + 3 * H] + b_hh[3 * H + u]);
      const float c = f * c_prev[o] + i * g;
      gates[g0] = i; gates[g0 + H] = f; gates[g0 + 2 * H] = g; gates[g0 + 3 * H] = oo;
      c_out[o] = c;
      h_out[o] = oo * tanhf(c);
    }
  }
}

extern "C" int lctvqa_lstm_step_fused_supported(int B, int H) { return (H % 32 == 0 && H >= 32 && B >= 1) ? 1 : 0; }

extern "C" int lctvqa_lstm_step_fused(const float* xp, const float* h_prev, const float* c_prev, const float* w_hh,
                                      const float* b_hh, float* gates, float* c_out, float* h_out, int B, int H, void* stream) {
  LCT_REQUIRE(lctvqa_lstm_step_fused_supported(B, H), "lstm_step_fused: need H %% 32 == 0 (got H=%d)", H);
  LCT_REQUIRE((((uintptr_t)h_prev | (uintptr_t)w_hh) & 15) == 0, "lstm_step_fused: h_prev / w_hh must be 16-B aligned");
  const size_t lds = 4 * 4 * 16 * 64 * sizeof(float);
  hipFuncSetAttribute((const void*)lstm_step_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(lstm_step_fused_kernel, dim3(cdiv(B, 32) * (H / 32)), dim3(256), lds, (hipStream_t)stream, xp, h_prev, c_prev, w_hh,
                     b_hh, gates, c_out, h_out, B, H);
  return lct_check_launch("lstm_step_fused");
}

// ------------------------------------------------------------------------------------------
// Fused BPTT step: dh_rec = dgates_t . W_hh  (the recurrent gradient into step t-1), immediately followed by
// the cell backward of step t-1 in the epilogue (same (row, unit) tile owns everything it needs).
// Workgroup = 32 rows x 32 units, K = 4H split over the 4 waves, partials meet in LDS.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_step_bwd_fused_kernel(const float* __restrict__ dg_t, const float* __restrict__ w_hh,
                                                                  const float* __restrict__ dh_ext, const float* __restrict__ dc_next,
                                                                  const float* __restrict__ gates_p, const float* __restrict__ c_pp,
                                                                  const float* __restrict__ c_p, float* __restrict__ dg_p,
                                                                  float* __restrict__ dc_prev, int B, int H) {
  __shared__ float red[4][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  const int nut = H / 32;
  const int ut = blockIdx.x % nut, bt = blockIdx.x / nut;
  const int b0 = bt * 32, u0 = ut * 32;
  const int K = 4 * H;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  const int arow = min(b0 + j, B - 1);
  const float* ap = dg_t + (long)arow * K;
  // super-groups of 64 k (4 x 16): enough MFMA work (32 per super-group) to cover the L2 latency of the next one
  const int nsg = K / 64;                               // K = 4H, H % 32 == 0  =>  K % 64 == 0
  const float* bp = w_hh + u0 + j;
  f32x4 A0[8], A1[8];
  float B0[8][4], B1[8][4];
  auto ld = [&](int sg, f32x4 (&A)[8], float (&Bf)[8][4]) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int kc = sg * 64 + (q >> 1) * 16 + (2 * (q & 1) + hh) * 4;
      A[q] = *(const f32x4*)(ap + kc);
#pragma unroll
      for (int e = 0; e < 4; e++) Bf[q][e] = bp[(long)(kc + e) * H];
    }
  };
  auto mm = [&](const f32x4 (&A)[8], const float (&Bf)[8][4]) {
#pragma unroll
    for (int q = 0; q < 8; q++)
#pragma unroll
      for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[q][e], Bf[q][e], acc, 0, 0, 0);
  };
  int sg = wave;
  if (sg < nsg) ld(sg, A0, B0);
  while (sg < nsg) {
    if (sg + 4 < nsg) ld(sg + 4, A1, B1);
    mm(A0, B0);
    sg += 4;
    if (sg >= nsg) break;
    if (sg + 4 < nsg) ld(sg + 4, A0, B0);
    mm(A1, B1);
    sg += 4;
  }
#pragma unroll
  for (int r = 0; r < 16; r++) red[wave][r][lane] = acc[r];
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; rr++) {
    const int r = wave * 4 + rr;
    const int row = b0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
    const int u = u0 + j;
    float dh = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
    if (row < B) {
      const long o = (long)row * H + u, g0 = (long)row * 4 * H + u;
      if (dh_ext) dh += dh_ext[o];
      const float i = gates_p[g0], f = gates_p[g0 + H], g = gates_p[g0 + 2 * H], oo = gates_p[g0 + 3 * H];
      const float tc = tanhf(c_p[o]);
      const float dc = (dc_next ? dc_next[o] : 0.f) + dh * oo * (1.f - tc * tc);
      dg_p[g0] = dc * g * i * (1.f - i);
      dg_p[g0 + H] = dc * c_pp[o] * f * (1.f - f);
      dg_p[g0 + 2 * H] = dc * i * (1.f - g * g);
      dg_p[g0 + 3 * H] = dh * tc * oo * (1.f - oo);
      dc_prev[o] = dc * f;
    }
  }
}

// dg_t: dgates of step t [B][4H]; gates_p / c_pp / c_p: activated gates, c_{t-2}... of step t-1 (c_pp = cell state entering
// step t-1, c_p = cell state it produced); dh_ext: external dL/dh_{t-1} (may be NULL); dc_next: dL/dc_{t-1} from step t.
extern "C" int lctvqa_lstm_step_bwd_fused(const float* dg_t, const float* w_hh, const float* dh_ext, const float* dc_next,
                                          const float* gates_p, const float* c_pp, const float* c_p, float* dg_p, float* dc_prev,
                                          int B, int H, void* stream) {
  LCT_REQUIRE(lctvqa_lstm_step_fused_supported(B, H), "lstm_step_bwd_fused: need H %% 32 == 0 (got H=%d)", H);
  LCT_REQUIRE(((uintptr_t)dg_t & 15) == 0, "lstm_step_bwd_fused: dgates must be 16-B aligned");
  hipLaunchKernelGGL(lstm_step_bwd_fused_kernel, dim3(cdiv(B, 32) * (H / 32)), dim3(256), 0, (hipStream_t)stream, dg_t, w_hh, dh_ext,
                     dc_next, gates_p, c_pp, c_p, dg_p, dc_prev, B, H);
  return lct_check_launch("lstm_step_bwd_fused");
}

// ------------------------------------------------------------------------------------------
// Whole-sequence LSTM recurrence (the T dependent steps of nn.LSTM; reference basic_vqa/models.py:61,
// models_lct.py:107-108,140-157).  One C call per direction: W_hh is re-ordered ONCE per sequence into MFMA-fragment
// order, so every weight load of every step is one lane-contiguous 1-KB piece (the per-step kernels above read rows
// of W_hh 16 B at a time, 64 cache lines per load instruction), then the T steps are launched back to back.  The
// exchange of h_t between the workgroups of a step is the kernel boundary: on gfx950 a dependent launch costs
// ~1.5 us, an in-kernel grid barrier with its agent-scope release/acquire 4-7 us (MI355X_MICROARCH.md, price list),
// so a persistent kernel spinning on a barrier per step would be the slower form of the same data flow.
// ------------------------------------------------------------------------------------------
// forward pack: [H/16 unit tiles][H/16 k groups][2 kk][2 tiles][64 lanes][4]; tile 0 = gates (i,f), tile 1 = (g,o);
// lane (j, hh): column j = (gate&1)*16 + unit, k = grp*16 + (2kk+hh)*4 + e
__global__ void lstm_pack_whh_fwd_kernel(const float* __restrict__ w_hh, float* __restrict__ p, int H) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int ng = H / 16;
  if (idx >= (long)H * H) return;                       // 4H*H floats = H*H float4s
  const int lane = idx & 63, tile = (idx >> 6) & 1, kk = (idx >> 7) & 1;
  const long rest = idx >> 8;
  const int grp = rest % ng, ut = rest / ng;
  const int j = lane & 31, hh = lane >> 5;
  const int gate = tile * 2 + (j >> 4), unit = ut * 16 + (j & 15);
  ((f32x4*)p)[idx] = *(const f32x4*)(w_hh + ((long)gate * H + unit) * H + grp * 16 + (2 * kk + hh) * 4);
}

// backward pack: [H/32 unit tiles][4H/64 super-groups][8 q][64 lanes][4]; lane (j, hh): value[e] = W_hh[kc+e][u0+j],
// kc = sg*64 + (q>>1)*16 + (2*(q&1)+hh)*4
__global__ void lstm_pack_whh_bwd_kernel(const float* __restrict__ w_hh, float* __restrict__ p, int H) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)H * H) return;
  const int nsg = 4 * H / 64;
  const int lane = idx & 63, q = (idx >> 6) & 7;
  const long rest = idx >> 9;
  const int sg = rest % nsg, ut = rest / nsg;
  const int j = lane & 31, hh = lane >> 5;
  const int kc = sg * 64 + (q >> 1) * 16 + (2 * (q & 1) + hh) * 4;
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; e++) v[e] = w_hh[(long)(kc + e) * H + ut * 32 + j];
  ((f32x4*)p)[idx] = v;
}

// one time step on the packed weights.  Workgroup = 32 batch rows x 16 hidden units x 4 gates (two 32x32 accumulator
// tiles per wave), the 4 waves split K = H into quarters of NG 16-deep groups; ALL of a wave's operand loads are issued
// before its first MFMA (NG * 6 float4 registers), partial tiles meet in LDS, the cell update is the epilogue.
// bf16 form of the packs (ops.set_compute_dtype('bf16'): the recurrent products on v_mfma_f32_32x32x16_bf16, fp32 accumulation,
// fp32 state and gates): forward [H/16 unit tiles][H/16 k groups][2 tiles][64 lanes][8 bf16], lane (j, hh): k = grp*16 + 8hh + e;
// backward [H/32 unit tiles][4H/64 super-groups][4 k16 groups][64 lanes][8 bf16], value[e] = W_hh[sg*64 + q*16 + 8hh + e][u0 + j]
typedef __bf16 lstm_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ lstm_bf16x8 lstm_cvt8(const f32x4 a, const f32x4 b) {
  return lstm_bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
}
__global__ void lstm_pack_whh_fwd_bf16_kernel(const float* __restrict__ w_hh, lstm_bf16x8* __restrict__ p, int H) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int ng = H / 16;
  if (idx >= (long)H * H / 2) return;                   // 4H*H values, 8 per lane
  const int lane = idx & 63, tile = (idx >> 6) & 1;
  const long rest = idx >> 7;
  const int grp = rest % ng, ut = rest / ng;
  const int j = lane & 31, hh = lane >> 5;
  const int gate = tile * 2 + (j >> 4), unit = ut * 16 + (j & 15);
  const float* src = w_hh + ((long)gate * H + unit) * H + grp * 16 + 8 * hh;
  p[idx] = lstm_cvt8(*(const f32x4*)src, *(const f32x4*)(src + 4));
}
__global__ void lstm_pack_whh_bwd_bf16_kernel(const float* __restrict__ w_hh, lstm_bf16x8* __restrict__ p, int H) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)H * H / 2) return;
  const int nsg = 4 * H / 64;
  const int lane = idx & 63, q = (idx >> 6) & 3;
  const long rest = idx >> 8;
  const int sg = rest % nsg, ut = rest / nsg;
  const int j = lane & 31, hh = lane >> 5;
  const int kc = sg * 64 + q * 16 + 8 * hh;
  lstm_bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; e++) v[e] = (__bf16)w_hh[(long)(kc + e) * H + ut * 32 + j];
  p[idx] = v;
}

template <int NG, bool BF16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void lstm_step2_kernel(const float* __restrict__ xp, const float* __restrict__ h_prev,
                                                         const float* __restrict__ c_prev, const float* __restrict__ wp,
                                                         const float* __restrict__ b_hh, float* __restrict__ gates,
                                                         float* __restrict__ c_out, float* __restrict__ h_out, int B) {
  constexpr int H = 64 * NG, NGRP = H / 16, RSTR = 80;   // RSTR: padded register stride of the exchange buffer
  __shared__ __attribute__((aligned(16))) float red[4 * 2 * 16 * RSTR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  constexpr int NUT = H / 16;
  const int ut = blockIdx.x % NUT, bt = blockIdx.x / NUT;
  const int b0 = bt * 32, u0 = ut * 16;
  const int arow = min(b0 + j, B - 1);                  // rows >= B are computed but never stored
  f32x4 A[NG][2], Bf[NG][2][2];
#pragma unroll
  for (int n = 0; n < NG; n++) {
    const int grp = wave * NG + n;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      if (BF16) {                                       // A[n][0..1] = the lane's 8 consecutive h values, Bf[n][0][t] = 8 bf16 weights
        A[n][kk] = *(const f32x4*)(h_prev + (long)arow * H + grp * 16 + 8 * hh + 4 * kk);
        if (kk == 0) {
#pragma unroll
          for (int t = 0; t < 2; t++) Bf[n][0][t] = *(const f32x4*)(wp + ((((long)ut * NGRP + grp) * 2 + t) * 64 + lane) * 4);
        }
      } else {
        A[n][kk] = *(const f32x4*)(h_prev + (long)arow * H + grp * 16 + (2 * kk + hh) * 4);
#pragma unroll
        for (int t = 0; t < 2; t++)
          Bf[n][kk][t] = *(const f32x4*)(wp + (((((long)ut * NGRP + grp) * 2 + kk) * 2 + t) * 64 + lane) * 4);
      }
    }
  }
  // epilogue operands of this thread's two (row, unit) outputs, requested before the MFMAs
  const int u = tid & 15, rlo = tid >> 4;
  float xin[2][4], bh[4], cp[2];
#pragma unroll
  for (int g = 0; g < 4; g++) bh[g] = b_hh[g * H + u0 + u];
#pragma unroll
  for (int o = 0; o < 2; o++) {
    const int row = min(b0 + rlo + 16 * o, B - 1);
#pragma unroll
    for (int g = 0; g < 4; g++) xin[o][g] = xp[(long)row * 4 * H + g * H + u0 + u];
    cp[o] = c_prev[(long)row * H + u0 + u];
  }
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
  __builtin_amdgcn_sched_barrier(0);                    // keep every load above in front of the first MFMA
  if (BF16) {
#pragma unroll
    for (int n = 0; n < NG; n++) {
      const lstm_bf16x8 a8 = lstm_cvt8(A[n][0], A[n][1]);
#pragma unroll
      for (int t = 0; t < 2; t++)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, __builtin_bit_cast(lstm_bf16x8, Bf[n][0][t]), acc[t], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int n = 0; n < NG; n++)
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
          for (int t = 0; t < 2; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[n][kk][e], Bf[n][kk][t][e], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) red[((wave * 2 + t) * 16 + r) * RSTR + lane] = acc[t][r];
  __syncthreads();
#pragma unroll
  for (int o = 0; o < 2; o++) {
    const int rl = rlo + 16 * o;                        // accumulator row rl = (r&3) + 8*(r>>2) + 4*hh
    const int r = (rl & 3) + 4 * (rl >> 3), rh = (rl >> 2) & 1;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      float sacc = 0.f;
#pragma unroll
      for (int w = 0; w < 4; w++) sacc += red[((w * 2 + (g >> 1)) * 16 + r) * RSTR + rh * 32 + (g & 1) * 16 + u];
      pre[g] = sacc + xin[o][g] + bh[g];
    }
    const int row = b0 + rl;
    if (row < B) {
      const long g0 = (long)row * 4 * H + u0 + u, oo_ = (long)row * H + u0 + u;
      const float i = sigmoidf_(pre[0]), f = sigmoidf_(pre[1]), g = tanhf(pre[2]), og = sigmoidf_(pre[3]);
      const float c = f * cp[o] + i * g;
      gates[g0] = i; gates[g0 + H] = f; gates[g0 + 2 * H] = g; gates[g0 + 3 * H] = og;
      c_out[oo_] = c;
      h_out[oo_] = og * tanhf(c);
    }
  }
}

// BPTT step on the packed (transposed-fragment) weights: as lstm_step_bwd_fused_kernel, with 1-KB weight pieces and the
// operands of two 64-deep super-groups in flight ahead of the MFMAs.
template <bool BF16 = false>
__global__ __launch_bounds__(256) void lstm_step2_bwd_kernel(const float* __restrict__ dg_t, const float* __restrict__ wp,
                                                             const float* __restrict__ dh_ext, const float* __restrict__ dc_next,
                                                             const float* __restrict__ gates_p, const float* __restrict__ c_pp,
                                                             const float* __restrict__ c_p, float* __restrict__ dg_p,
                                                             float* __restrict__ dc_prev, int B, int H) {
  __shared__ float red[4][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  const int nut = H / 32;
  const int ut = blockIdx.x % nut, bt = blockIdx.x / nut;
  const int b0 = bt * 32, u0 = ut * 32;
  const int K = 4 * H, nsg = K / 64;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  const int arow = min(b0 + j, B - 1);
  const float* ap = dg_t + (long)arow * K;
  const float* bp = wp + ((long)ut * nsg * (BF16 ? 4 : 8) * 64 + lane) * 4;
  f32x4 A0[8], A1[8], A2[8], B0[8], B1[8], B2[8];
  auto ld = [&](int sg, f32x4 (&A)[8], f32x4 (&Bf)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (BF16) {                                       // A[2g], A[2g+1] = the lane's 8 consecutive dg values of k16 group g; Bf[g] = 8 bf16 weights
        A[q] = *(const f32x4*)(ap + sg * 64 + (q >> 1) * 16 + 8 * hh + 4 * (q & 1));
        if (q < 4) Bf[q] = *(const f32x4*)(bp + ((long)sg * 4 + q) * 256);
      } else {
        A[q] = *(const f32x4*)(ap + sg * 64 + (q >> 1) * 16 + (2 * (q & 1) + hh) * 4);
        Bf[q] = *(const f32x4*)(bp + ((long)sg * 8 + q) * 256);
      }
    }
  };
  auto mm = [&](const f32x4 (&A)[8], const f32x4 (&Bf)[8]) __attribute__((always_inline)) {
    if (BF16) {
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lstm_cvt8(A[2 * g4], A[2 * g4 + 1]), __builtin_bit_cast(lstm_bf16x8, Bf[g4]), acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int q = 0; q < 8; q++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[q][e], Bf[q][e], acc, 0, 0, 0);
    }
  };
  int sg = wave;                                        // this wave's super-groups: wave, wave + 4, ...
  if (sg < nsg) ld(sg, A0, B0);
  if (sg + 4 < nsg) ld(sg + 4, A1, B1);
  while (sg < nsg) {
    if (sg + 8 < nsg) ld(sg + 8, A2, B2);
    mm(A0, B0);
    sg += 4;
    if (sg >= nsg) break;
    if (sg + 8 < nsg) ld(sg + 8, A0, B0);
    mm(A1, B1);
    sg += 4;
    if (sg >= nsg) break;
    if (sg + 8 < nsg) ld(sg + 8, A1, B1);
    mm(A2, B2);
    sg += 4;
  }
#pragma unroll
  for (int r = 0; r < 16; r++) red[wave][r][lane] = acc[r];
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; rr++) {
    const int r = wave * 4 + rr;
    const int row = b0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
    const int u = u0 + j;
    float dh = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
    if (row < B) {
      const long o = (long)row * H + u, g0 = (long)row * 4 * H + u;
      if (dh_ext) dh += dh_ext[o];
      const float i = gates_p[g0], f = gates_p[g0 + H], g = gates_p[g0 + 2 * H], oo = gates_p[g0 + 3 * H];
      const float tc = tanhf(c_p[o]);
      const float dc = (dc_next ? dc_next[o] : 0.f) + dh * oo * (1.f - tc * tc);
      dg_p[g0] = dc * g * i * (1.f - i);
      dg_p[g0 + H] = dc * c_pp[o] * f * (1.f - f);
      dg_p[g0 + 2 * H] = dc * i * (1.f - g * g);
      dg_p[g0 + 3 * H] = dh * tc * oo * (1.f - oo);
      dc_prev[o] = dc * f;
    }
  }
}

extern "C" long lctvqa_lstm_seq_scratch_elems(int H) { return 4L * H * H; }
extern "C" int lctvqa_lstm_seq_packed(int B, int H) { return (B >= 1 && (H == 64 || H == 128 || H == 256 || H == 512)) ? 1 : 0; }

// h_all / c_all: [T+1][B][H] with slot 0 = the initial state (set by the caller); fills slots 1..T and gates [T][B][4H].
// wpack: lctvqa_lstm_seq_scratch_elems(H) floats of scratch (untouched when the shape runs on the per-step kernels).
static int lstm_seq_fwd_impl(const float* xp, const float* w_hh, const float* b_hh, float* h_all, float* c_all, float* gates,
                             float* wpack, int T, int B, int H, int bf16, void* stream) {
  LCT_REQUIRE(T >= 1 && lctvqa_lstm_step_fused_supported(B, H), "lstm_seq_fwd: need T >= 1, H %% 32 == 0 (got T=%d H=%d)", T, H);
  LCT_REQUIRE((((uintptr_t)h_all | (uintptr_t)w_hh | (uintptr_t)wpack) & 15) == 0, "lstm_seq_fwd: h_all / w_hh / wpack must be 16-B aligned");
  hipStream_t st = (hipStream_t)stream;
  const long BH = (long)B * H;
  if (!lctvqa_lstm_seq_packed(B, H)) {
    for (int t = 0; t < T; t++) {
      int rc = lctvqa_lstm_step_fused(xp + t * 4 * BH, h_all + t * BH, c_all + t * BH, w_hh, b_hh, gates + t * 4 * BH,
                                      c_all + (t + 1) * BH, h_all + (t + 1) * BH, B, H, stream);
      if (rc) return rc;
    }
    return 0;
  }
  if (bf16) hipLaunchKernelGGL(lstm_pack_whh_fwd_bf16_kernel, dim3(cdiv((long)H * H / 2, 256)), dim3(256), 0, st, w_hh, (lstm_bf16x8*)wpack, H);
  else hipLaunchKernelGGL(lstm_pack_whh_fwd_kernel, dim3(cdiv((long)H * H, 256)), dim3(256), 0, st, w_hh, wpack, H);
  const dim3 grid(cdiv(B, 32) * (H / 16));
  for (int t = 0; t < T; t++) {
    const float *x = xp + t * 4 * BH, *hp = h_all + t * BH, *cp = c_all + t * BH;
    float *g = gates + t * 4 * BH, *co = c_all + (t + 1) * BH, *ho = h_all + (t + 1) * BH;
#define LSTM_STEP(NGv)                                                                                                        \
    if (bf16) hipLaunchKernelGGL((lstm_step2_kernel<NGv, true>), grid, dim3(256), 0, st, x, hp, cp, wpack, b_hh, g, co, ho, B); \
    else hipLaunchKernelGGL((lstm_step2_kernel<NGv, false>), grid, dim3(256), 0, st, x, hp, cp, wpack, b_hh, g, co, ho, B);
    switch (H) {
      case 64: LSTM_STEP(1) break;
      case 128: LSTM_STEP(2) break;
      case 256: LSTM_STEP(4) break;
      default: LSTM_STEP(8) break;
    }
#undef LSTM_STEP
  }
  return lct_check_launch("lstm_seq_fwd");
}
extern "C" int lctvqa_lstm_seq_fwd(const float* xp, const float* w_hh, const float* b_hh, float* h_all, float* c_all, float* gates,
                                   float* wpack, int T, int B, int H, void* stream) {
  return lstm_seq_fwd_impl(xp, w_hh, b_hh, h_all, c_all, gates, wpack, T, B, H, 0, stream);
}
// the same recurrence with the products h . W_hh^T on the bf16 matrix cores (h and W_hh rounded to bf16 per step, fp32 accumulation,
// fp32 gates / cell state): the bf16 compute dtype of BASELINE.json's configurations 2-4; shapes outside the packed form run fp32
extern "C" int lctvqa_lstm_seq_fwd_bf16(const float* xp, const float* w_hh, const float* b_hh, float* h_all, float* c_all, float* gates,
                                        float* wpack, int T, int B, int H, void* stream) {
  return lstm_seq_fwd_impl(xp, w_hh, b_hh, h_all, c_all, gates, wpack, T, B, H, 1, stream);
}

// Steps t = T-2 .. 0 of BPTT (step T-1 is the plain cell backward, lctvqa_lstm_cell_bwd, done by the caller into
// dgates[T-1] and dc_buf[0]).  dout: dL/dh_t from the sequence output [T][B][H] or NULL.  dc_buf: [2][B][H]; step i reads
// half i&1 and writes the other, so dL/dc_0's factor ends in half (T-1)&1.
static int lstm_seq_bwd_impl(float* dgates, const float* w_hh, const float* dout, const float* gates, const float* c_all,
                             float* dc_buf, float* wpack, int T, int B, int H, int bf16, void* stream) {
  LCT_REQUIRE(T >= 1 && lctvqa_lstm_step_fused_supported(B, H), "lstm_seq_bwd: need T >= 1, H %% 32 == 0 (got T=%d H=%d)", T, H);
  LCT_REQUIRE((((uintptr_t)dgates | (uintptr_t)wpack) & 15) == 0, "lstm_seq_bwd: dgates / wpack must be 16-B aligned");
  hipStream_t st = (hipStream_t)stream;
  const long BH = (long)B * H;
  const bool packed = lctvqa_lstm_seq_packed(B, H) && T > 1;
  if (packed && bf16) hipLaunchKernelGGL(lstm_pack_whh_bwd_bf16_kernel, dim3(cdiv((long)H * H / 2, 256)), dim3(256), 0, st, w_hh, (lstm_bf16x8*)wpack, H);
  else if (packed) hipLaunchKernelGGL(lstm_pack_whh_bwd_kernel, dim3(cdiv((long)H * H, 256)), dim3(256), 0, st, w_hh, wpack, H);
  for (int t = T - 2, i = 0; t >= 0; t--, i++) {
    const float* dext = dout ? dout + t * BH : nullptr;
    float *dcin = dc_buf + (i & 1) * BH, *dcout = dc_buf + ((i + 1) & 1) * BH;
    if (packed && bf16) {
      hipLaunchKernelGGL(lstm_step2_bwd_kernel<true>, dim3(cdiv(B, 32) * (H / 32)), dim3(256), 0, st, dgates + (t + 1) * 4 * BH, wpack, dext,
                         dcin, gates + t * 4 * BH, c_all + t * BH, c_all + (t + 1) * BH, dgates + t * 4 * BH, dcout, B, H);
    } else if (packed) {
      hipLaunchKernelGGL(lstm_step2_bwd_kernel<false>, dim3(cdiv(B, 32) * (H / 32)), dim3(256), 0, st, dgates + (t + 1) * 4 * BH, wpack, dext,
                         dcin, gates + t * 4 * BH, c_all + t * BH, c_all + (t + 1) * BH, dgates + t * 4 * BH, dcout, B, H);
    } else {
      int rc = lctvqa_lstm_step_bwd_fused(dgates + (t + 1) * 4 * BH, w_hh, dext, dcin, gates + t * 4 * BH, c_all + t * BH,
                                          c_all + (t + 1) * BH, dgates + t * 4 * BH, dcout, B, H, stream);
      if (rc) return rc;
    }
  }
  return lct_check_launch("lstm_seq_bwd");
}
extern "C" int lctvqa_lstm_seq_bwd(float* dgates, const float* w_hh, const float* dout, const float* gates, const float* c_all,
                                   float* dc_buf, float* wpack, int T, int B, int H, void* stream) {
  return lstm_seq_bwd_impl(dgates, w_hh, dout, gates, c_all, dc_buf, wpack, T, B, H, 0, stream);
}
extern "C" int lctvqa_lstm_seq_bwd_bf16(float* dgates, const float* w_hh, const float* dout, const float* gates, const float* c_all,
                                        float* dc_buf, float* wpack, int T, int B, int H, void* stream) {
  return lstm_seq_bwd_impl(dgates, w_hh, dout, gates, c_all, dc_buf, wpack, T, B, H, 1, stream);
}

// ------------------------------------------------------------------------------------------
// Greedy question generation (reference models_lct.py:124-157, darts_vqa/vqa_model.py:190-222): L dependent tokens,
//   x_0 = tanh(emb[<start>]),  x_t = emb[pred_{t-1}] (no tanh);  (h,c) = LSTM(x_t, (h,c));  pred_t = argmax(fc(tanh(h)))
// as ONE C call: W_hh is packed once, then 7 launches per token (gather, input projection, recurrent step, tanh, token-head
// GEMM, argmax, token store) are issued back to back with no host work between them -- the Python loop it replaces spends
// ~8 autograd/ctypes round trips (~100 us) per token on ~70 us of kernels.  Same kernels and GEMM configurations as the
// per-token path, so the token ids are identical.
// ------------------------------------------------------------------------------------------
__global__ void fill_long_kernel(long* __restrict__ p, long v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void store_tokens_kernel(const long* __restrict__ idx, long* __restrict__ out, int B, int L, int t) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[(long)b * L + t] = idx[b];
}

extern "C" long lctvqa_gemm_workspace_bytes(int M, int N, int K);
extern "C" int lctvqa_gemm_f32(const float* A, long a_ms, long a_ks, const float* B, long b_ns, long b_ks, float* C, long ldc, int M,
                               int N, int K, const float* bias, const float* addend, long ld_add, int act, const uint8_t* mask,
                               long ld_mask, float mask_scale, float* workspace, long workspace_bytes, void* stream);

static long decode_ws_bytes(int B, int H, int We, int QV) {
  const long a = lctvqa_gemm_workspace_bytes(B, 4 * H, We), b = lctvqa_gemm_workspace_bytes(B, QV, H);
  return ((a > b ? a : b) + 15) & ~15L;
}
// floats of scratch for lctvqa_lstm_decode_greedy
extern "C" long lctvqa_lstm_decode_scratch_elems(int B, int H, int We, int QV) {
  const long BH = (long)B * H;
  return 4L * H * H + (long)B * We + 8L * BH + 2 * BH + BH + (long)B * QV + 2L * B + 64 + decode_ws_bytes(B, H, We, QV) / 4;
}

// h, c: [B][H] in/out (initial state -> state after the last token); tokens [B][L] int64 out; emb [QV][We]; token head
// fc_w [QV][H], fc_b [QV]; gate order and weight shapes of nn.LSTM (w_ih [4H][We], w_hh [4H][H]).
extern "C" int lctvqa_lstm_decode_greedy(const float* emb, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                                         const float* fc_w, const float* fc_b, float* h, float* c, long* tokens, long start_token,
                                         int L, int B, int H, int We, int QV, float* scratch, void* stream) {
  LCT_REQUIRE(L >= 1 && B >= 1 && lctvqa_lstm_step_fused_supported(B, H), "lstm_decode_greedy: need L, B >= 1 and H %% 32 == 0 (H=%d)", H);
  LCT_REQUIRE((((uintptr_t)scratch | (uintptr_t)h | (uintptr_t)w_hh) & 15) == 0, "lstm_decode_greedy: scratch / h / w_hh must be 16-B aligned");
  hipStream_t st = (hipStream_t)stream;
  const long BH = (long)B * H;
  float* wpack = scratch;
  float* xcur = wpack + 4L * H * H;
  float* xp = xcur + (((long)B * We + 3) & ~3L);
  float* gates = xp + 4 * BH;
  float* hn = gates + 4 * BH;
  float* cn = hn + BH;
  float* tout = cn + BH;
  float* logits = tout + BH;
  long* idx = (long*)(logits + (((long)B * QV + 3) & ~3L));
  float* ws = (float*)(idx + B + (B & 1));
  ws = (float*)(((uintptr_t)ws + 15) & ~(uintptr_t)15);
  const long ws_bytes = decode_ws_bytes(B, H, We, QV);
  const bool packed = lctvqa_lstm_seq_packed(B, H);
  if (packed) hipLaunchKernelGGL(lstm_pack_whh_fwd_kernel, dim3(cdiv((long)H * H, 256)), dim3(256), 0, st, w_hh, wpack, H);
  hipLaunchKernelGGL(fill_long_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, idx, start_token, B);
  float *hc = h, *cc = c, *hx = hn, *cx = cn;
  const dim3 sgrid(cdiv(B, 32) * (H / 16));
  for (int t = 0; t < L; t++) {
    int rc = lctvqa_embed_gather(idx, emb, xcur, B, 1, We, QV, t == 0 ? 1 : 0, stream);
    if (rc) return rc;
    rc = lctvqa_gemm_f32(xcur, We, 1, w_ih, We, 1, xp, 4L * H, B, 4 * H, We, b_ih, nullptr, 0, 0, nullptr, 0, 1.f, ws, ws_bytes, stream);
    if (rc) return rc;
    if (packed) {
      switch (H) {
        case 64: hipLaunchKernelGGL(lstm_step2_kernel<1>, sgrid, dim3(256), 0, st, xp, hc, cc, wpack, b_hh, gates, cx, hx, B); break;
        case 128: hipLaunchKernelGGL(lstm_step2_kernel<2>, sgrid, dim3(256), 0, st, xp, hc, cc, wpack, b_hh, gates, cx, hx, B); break;
        case 256: hipLaunchKernelGGL(lstm_step2_kernel<4>, sgrid, dim3(256), 0, st, xp, hc, cc, wpack, b_hh, gates, cx, hx, B); break;
        default: hipLaunchKernelGGL(lstm_step2_kernel<8>, sgrid, dim3(256), 0, st, xp, hc, cc, wpack, b_hh, gates, cx, hx, B); break;
      }
    } else {
      rc = lctvqa_lstm_step_fused(xp, hc, cc, w_hh, b_hh, gates, cx, hx, B, H, stream);
      if (rc) return rc;
    }
    rc = lctvqa_tanh_fwd(hx, tout, BH, stream);
    if (rc) return rc;
    rc = lctvqa_gemm_f32(tout, H, 1, fc_w, H, 1, logits, QV, B, QV, H, fc_b, nullptr, 0, 0, nullptr, 0, 1.f, ws, ws_bytes, stream);
    if (rc) return rc;
    rc = lctvqa_argmax_rows(logits, QV, B, QV, idx, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(store_tokens_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, idx, tokens, B, L, t);
    float* tp = hc; hc = hx; hx = tp;
    tp = cc; cc = cx; cx = tp;
  }
  if (hc != h) {                                        // odd number of tokens: the final state sits in the scratch halves
    hipMemcpyAsync(h, hc, BH * sizeof(float), hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(c, cc, BH * sizeof(float), hipMemcpyDeviceToDevice, st);
  }
  return lct_check_launch("lstm_decode_greedy");
}

// ------------------------------------------------------------------------------------------
// Input pipeline, device side (reference basic_vqa/data_loader_v2.py:53-55,82-86: PIL fromarray -> ToTensor ->
// Normalize(mean,std) per sample on the host): the uint8 HWC image store rows are copied to HBM as they are
// (3 B/pixel over PCIe instead of 12) and converted here:  y[n][c][h][w] = (x[n][h][w][c]/255 - mean[c]) / std[c]
// ------------------------------------------------------------------------------------------
__global__ void image_u8_to_f32_nchw_kernel(const uint8_t* __restrict__ x, float* __restrict__ y, int B, int H, int W, float m0,
                                            float m1, float m2, float s0, float s1, float s2) {
  const long HW = (long)H * W, total = (long)B * HW;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const long n = o / HW, p = o % HW;
    const uint8_t* px = x + o * 3;
    float* dst = y + n * 3 * HW + p;
    dst[0] = ((float)px[0] * (1.f / 255.f) - m0) / s0;
    dst[HW] = ((float)px[1] * (1.f / 255.f) - m1) / s1;
    dst[2 * HW] = ((float)px[2] * (1.f / 255.f) - m2) / s2;
  }
}
extern "C" int lctvqa_image_u8_nhwc_to_f32_nchw(const uint8_t* x, float* y, int B, int H, int W, float m0, float m1, float m2,
                                                float s0, float s1, float s2, void* stream) {
  LCT_REQUIRE(B > 0 && H > 0 && W > 0 && s0 != 0.f && s1 != 0.f && s2 != 0.f, "image_u8_to_f32: bad args");
  hipLaunchKernelGGL(image_u8_to_f32_nchw_kernel, dim3(ew_blocks((long)B * H * W)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W,
                     m0, m1, m2, s0, s1, s2);
  return lct_check_launch("image_u8_to_f32_nchw");
}

// ------------------------------------------------------------------------------------------
// GRU cell (nn.GRU gate order r,z,n) -- the "LSTM/GRU question encoder" variant named by BASELINE.json
// (not present in the reference; oracle = torch.nn.GRU):
//   r = s(xp_r + hp_r)  z = s(xp_z + hp_z)  n = tanh(xp_n + r * hp_n)  h' = (1-z)*n + z*h
// xp = x.W_ih^T + b_ih, hp = h.W_hh^T + b_hh, each [B][3H].  saved[b][4H] = r, z, n, hp_n for the backward.
// ------------------------------------------------------------------------------------------
__global__ void gru_cell_fwd_kernel(const float* __restrict__ xp, const float* __restrict__ hp, const float* __restrict__ h_prev,
                                    float* __restrict__ saved, float* __restrict__ h_out, int B, int H) {
  long total = (long)B * H;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    int u = o % H; long b = o / H;
    long g0 = b * 3 * H + u, s0 = b * 4 * H + u;
    float r = sigmoidf_(xp[g0] + hp[g0]);
    float z = sigmoidf_(xp[g0 + H] + hp[g0 + H]);
    float hn = hp[g0 + 2 * H];
    float n = tanhf(xp[g0 + 2 * H] + r * hn);
    saved[s0] = r; saved[s0 + H] = z; saved[s0 + 2 * H] = n; saved[s0 + 3 * H] = hn;
    h_out[o] = (1.f - z) * n + z * h_prev[o];
  }
}
// dh = dh_rec + dh_ext;  writes dxp [B][3H], dhp [B][3H] and dh_carry = dh * z (the direct path to h_prev)
__global__ void gru_cell_bwd_kernel(const float* __restrict__ dh_rec, const float* __restrict__ dh_ext,
                                    const float* __restrict__ saved, const float* __restrict__ h_prev, float* __restrict__ dxp,
                                    float* __restrict__ dhp, float* __restrict__ dh_carry, int B, int H) {
  long total = (long)B * H;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    int u = o % H; long b = o / H;
    long g0 = b * 3 * H + u, s0 = b * 4 * H + u;
    float r = saved[s0], z = saved[s0 + H], n = saved[s0 + 2 * H], hn = saved[s0 + 3 * H];
    float dh = (dh_rec ? dh_rec[o] : 0.f) + (dh_ext ? dh_ext[o] : 0.f);
    float dn = dh * (1.f - z) * (1.f - n * n);
    float dz = dh * (h_prev[o] - n) * z * (1.f - z);
    float dr = dn * hn * r * (1.f - r);
    dxp[g0] = dr; dxp[g0 + H] = dz; dxp[g0 + 2 * H] = dn;
    dhp[g0] = dr; dhp[g0 + H] = dz; dhp[g0 + 2 * H] = dn * r;
    dh_carry[o] = dh * z;
  }
}
extern "C" int lctvqa_gru_cell_fwd(const float* xp, const float* hp, const float* h_prev, float* saved, float* h_out, int B, int H,
                                   void* stream) {
  LCT_REQUIRE(B > 0 && H > 0, "gru_cell_fwd: bad shape");
  hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(ew_blocks((long)B * H)), dim3(256), 0, (hipStream_t)stream, xp, hp, h_prev, saved, h_out, B, H);
  return lct_check_launch("gru_cell_fwd");
}
extern "C" int lctvqa_gru_cell_bwd(const float* dh_rec, const float* dh_ext, const float* saved, const float* h_prev, float* dxp,
                                   float* dhp, float* dh_carry, int B, int H, void* stream) {
  LCT_REQUIRE(B > 0 && H > 0, "gru_cell_bwd: bad shape");
  hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(ew_blocks((long)B * H)), dim3(256), 0, (hipStream_t)stream, dh_rec, dh_ext, saved, h_prev,
                     dxp, dhp, dh_carry, B, H);
  return lct_check_launch("gru_cell_bwd");
}

// ---- multi-tensor copy: one launch for a whole state_dict (the architects' twin models: `_construct_model_from_theta` /
// `load_state_dict` per unrolled model, reference pcdarts/architect_lct.py:142-155, architect.py:80-95, 2151 tensors for the
// PC-DARTS examiner).  The caller cuts the tensors into chunks of at most 64 KiB; table = [n][3] x 8 bytes (src, dst, bytes) in HBM.
__global__ __launch_bounds__(256) void multi_copy_kernel(const unsigned long* __restrict__ table, int n) {
  for (int c = blockIdx.x; c < n; c += gridDim.x) {
    const char* s = (const char*)table[3 * c];
    char* d = (char*)table[3 * c + 1];
    const long bytes = (long)table[3 * c + 2];
    if ((((unsigned long)s | (unsigned long)d | (unsigned long)bytes) & 15) == 0) {
      const f32x4* s4 = (const f32x4*)s;
      f32x4* d4 = (f32x4*)d;
      for (long i = threadIdx.x; i < (bytes >> 4); i += blockDim.x) d4[i] = s4[i];
    } else if ((((unsigned long)s | (unsigned long)d | (unsigned long)bytes) & 3) == 0) {
      const float* s1 = (const float*)s;
      float* d1 = (float*)d;
      for (long i = threadIdx.x; i < (bytes >> 2); i += blockDim.x) d1[i] = s1[i];
    } else {
      for (long i = threadIdx.x; i < bytes; i += blockDim.x) d[i] = s[i];
    }
  }
}
extern "C" int lctvqa_multi_copy(const void* table, int n_chunks, void* stream) {
  LCT_REQUIRE(table != nullptr && n_chunks > 0, "multi_copy: empty table");
  const int blocks = n_chunks < 4096 ? n_chunks : 4096;
  hipLaunchKernelGGL(multi_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned long*)table, n_chunks);
  return lct_check_launch("multi_copy");
}
