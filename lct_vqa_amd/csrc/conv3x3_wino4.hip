// Winograd F(4x4,3x3) convolution for the VGG19 trunk on gfx950, fp32 in / fp32 accumulate, NHWC.
//
// Same operator as conv3x3.hip / conv3x3_wino.hip (nn.Conv2d(Cin,Cout,3,padding=1) + bias + ReLU [+ MaxPool2d(2,2)],
// reference basic_vqa/models.py:36 via torchvision vgg19.features) with the larger minimal-filtering tile
//     Y(4x4) = A^T [ sum_cin (G g G^T)(6x6) (.) (B^T d B)(6x6) ] A          (Lavin & Gray 2016, F(4x4,3x3))
// = 36 multiplies per 4x4 output tile and (cin,cout) pair instead of 144: 4x fewer MFMA FLOPs than the direct
// algorithm (F(2x2,3x3): 2.25x).  All arithmetic is fp32; the transforms use the small integers / fractions of the
// published matrices, which costs about one decimal digit: measured 3-4e-6 max error relative to the layer's largest
// output on the trunk's shapes (direct fp32: 2-3e-7; tools/sim_f43.py), far inside the 1e-3 logit gate.
//
// Structure (one workgroup = 8 waves, two per SIMD, 256 registers each):
//  * a workgroup owns 32 tiles (TY x TX tiles of 4x4 output pixels) x 64 output channels = 72 accumulator tiles
//    (36 transformed-domain GEMMs xi*6+nu x two 32-channel groups; 32 tiles x 32 channels = 16 registers each), dealt out
//    8 to each of the six waves that also carry a transform column and 12 to the two that do not; the second wave of a
//    SIMD fills the first one's issue gaps;
//  * K loop in 8-channel chunks: raw halo patch HBM->LDS by LDS-DMA (planar per channel quad, stacked rows with one
//    shared zero row per image boundary); input transform B^T d B by waves 0..5: thread = (tile, channel quad), wave =
//    output column nu, so every wave runs straight-line code for its column; V[36][2 quads][32 tiles][4] double
//    buffered, the transform of chunk c+1 cut into slices pinned between the MFMAs of chunk c; weights U = G g G^T
//    straight from L2 into a 4-slot register ring refilled right behind the MFMAs (counted vmcnt, bare barrier);
//  * epilogue: the 36 accumulators of a (tile, channel) live in 4 waves: they are exchanged through LDS in 4 rounds
//    (8 (channel group, register) units = 73.7 KB per round), each wave output-transforms A^T M A one unit per round,
//    adds bias, ReLU, optional 2x2 max-pool in registers, NHWC stores 128 B contiguous per half wave.
#include "common.h"

__device__ __attribute__((aligned(64))) float g_zero_page_w4[64];

// U[cout/32][Cin/8][36 xi*6+nu][2 quads][32 cout][4 ch] = G g G^T.   flip_t: weights of the dgrad convolution
// (g'[ci][co][ky][kx] = g[co][ci][2-ky][2-kx]; Cout/Cin then name the dgrad conv's own output/input channels).
__global__ void wino4_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, int flip_t) {
  const long total = (long)Cout * Cin * 36;
  const int NCH = Cin / 8;
  const float Gm[6][3] = {{0.25f, 0.f, 0.f},
                          {-1.f / 6, -1.f / 6, -1.f / 6},
                          {-1.f / 6, 1.f / 6, -1.f / 6},
                          {1.f / 24, 1.f / 12, 1.f / 6},
                          {1.f / 24, -1.f / 12, 1.f / 6},
                          {0.f, 0.f, 1.f}};
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int e = o & 3, j = (o >> 2) & 31, h = (o >> 7) & 1;
    const long rest = o >> 8;
    const int xn = rest % 36;
    const long rest2 = rest / 36;
    const int chunk = rest2 % NCH, grp = rest2 / NCH;
    const int co = grp * 32 + j, ci = chunk * 8 + h * 4 + e;
    const int xi = xn / 6, nu = xn % 6;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      float t = 0.f;                                           // (g G^T)[a][nu]
#pragma unroll
      for (int b = 0; b < 3; b++) {
        const float gv = flip_t ? w[((long)ci * Cout + co) * 9 + (2 - a) * 3 + (2 - b)] : w[((long)co * Cin + ci) * 9 + a * 3 + b];
        t += gv * Gm[nu][b];
      }
      acc += Gm[xi][a] * t;
    }
    u[o] = acc;
  }
}

template <int TY, int TX, int GMAX, bool POOL>
__global__ __launch_bounds__(512, 1) void conv3x3_wino4_kernel(const float* __restrict__ x, const float* __restrict__ up,
                                                               const float* __restrict__ bias, float* __restrict__ y, int Btot,
                                                               int H, int W, int Cin, int Cout, int relu, int G, int Hp, int Wp) {
  // Hp, Wp: H, W rounded up to the tile grid (4 rows, TW columns).  Rows H..Hp-1 / columns W..Wp-1 read as zeros and their
  // outputs are dropped, so any map size works (14x14 runs as 16x16: 77 % useful).
  constexpr int NTILE = TY * TX;
  static_assert(NTILE <= 32, "patch holds too many tiles");
  constexpr int TH = 4 * TY, TW = 4 * TX, RS = TW + 2;
  constexpr int HRMAX = TH + 2 + GMAX;
  constexpr int PMAX = HRMAX * RS;
  constexpr int AROUNDS = (PMAX + 63) / 64;
  constexpr int PLANE = AROUNDS * 64 * 4;                       // floats per channel-quad plane of the raw patch
  constexpr int RAWSZ = 2 * PLANE;                              // 8 channels
  constexpr int VSZ = 36 * 256;                                 // floats per V buffer: [36][2 quads][32 tiles][4]
  constexpr int NPC = (2 * AROUNDS + 7) / 8;                    // LDS-DMA pieces per wave and chunk

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* V = smem;                    // [2][VSZ]   (re-used as the exchange buffer of the epilogue)
  float* raw = smem + 2 * VSZ;        // [2][RAWSZ]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int NT = Cout / 64;
  // workgroups are dealt to the 8 XCDs round-robin: the NT workgroups reading one input patch share an XCD (its L2)
  const int CT = Wp / TW;
  const int MTALL = ((Btot * Hp + TH - 1) / TH) * CT;
#ifdef W4_XCD_BAND
  // each XCD walks a contiguous band of patches in row-major order: horizontally and vertically adjacent patches (which share
  // halo columns / rows) run on the same XCD close in time, so the shared lines are served by its L2
  const int n_tile = (blockIdx.x >> 3) % NT, m_tile = (blockIdx.x & 7) * ((MTALL + 7) / 8) + (blockIdx.x >> 3) / NT;
  if ((blockIdx.x >> 3) / NT >= (MTALL + 7) / 8 || m_tile >= MTALL) return;
#else
  const int n_tile = (blockIdx.x >> 3) % NT, m_tile = ((blockIdx.x >> 3) / NT) * 8 + (blockIdx.x & 7);
  if (m_tile >= MTALL) return;
#endif
  const int ct = m_tile % CT, rt = m_tile / CT;
  const int R0 = rt * TH, x0 = ct * TW;
  const int n0img = R0 / Hp, y0 = R0 % Hp;                      // stacked PADDED rows: R = n * Hp + y
  const int NCH = Cin / 8;
  const int PLIVE = (TH + 2 + G) * RS;

  // ---- raw patch staging, this wave takes pieces pc = wave, wave + 8, ...
  // RAWPX: raw[pixel][8 channels]; a piece = 32 halo pixels x both channel quads: lane = (pixel lane>>1, quad lane&1) reads 32
  // contiguous bytes per pixel with its neighbour, so a DMA instruction touches 32 cache lines instead of 64 (-2..-4 % on the 56x56
  // and 28x28 layers; same-process A/B).  Planar raw[quad][pixel][4] (piece = quad * AROUNDS + round, 64 pixels of one quad) is kept
  // for the 32x16-pixel patch, whose transform reads lose more to the 32-byte pixel stride than its DMA gains (+1..2 %).
  constexpr bool RAWPX = !(TY == 8 && TX == 4);
  int aoff[NPC], pdst[NPC];
#pragma unroll
  for (int k = 0; k < NPC; k++) {
    const int pc = wave + 8 * k;
    const int r = pc % AROUNDS;
    const int q = RAWPX ? (lane & 1) : pc / AROUNDS;
    const int p = RAWPX ? pc * 32 + (lane >> 1) : r * 64 + lane;
    const int hr = p / RS, hx = p % RS;
    const int xx = x0 - 1 + hx;
    int yv = y0 - 1 + hr;
    int n = n0img;
    while (yv > Hp) { yv -= (Hp + 1); n++; }
    const bool ok = (p < PLIVE) && (yv >= 0) && (yv < H) && (n < Btot) && (xx >= 0) && (xx < W);
    aoff[k] = ok ? (((n * H + yv) * W + xx) * Cin + q * 4) : -1;
    if (RAWPX) pdst[k] = (pc < 2 * AROUNDS && pc * 32 < PLIVE) ? pc * 256 : -1;     // wave-uniform
    else pdst[k] = (pc < 2 * AROUNDS && r * 64 < PLIVE) ? ((pc / AROUNDS) * PLANE + r * 256) : -1;
  }
  auto stage_raw = [&](int chunk, float* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NPC; k++) {
      if (pdst[k] >= 0) {
        const float* src = (aoff[k] >= 0) ? (x + (long)aoff[k] + chunk * 8) : g_zero_page_w4;
        glds16_opaque(src, dst + pdst[k]);
      }
    }
  };

  // ---- input transform: thread = (tile tt, channel quad tc), wave = output column nu (waves 6, 7: none)
  const int tt = lane & 31, tc = lane >> 5;
  constexpr int PXS = RAWPX ? 8 : 4;                            // floats between neighbouring pixels of the raw patch
  int tbase = RAWPX ? tc * 4 : tc * PLANE;
  if (tt < NTILE) {
    const int ty = tt / TX, tx = tt % TX;
    const int g = min((y0 + 4 * ty) / Hp, G);                    // image boundaries (= shared zero rows) above this tile;
                                                                // tiles past the last image (never stored) stay inside the patch
    tbase += ((4 * ty + g) * RS + 4 * tx) * PXS;
  }
  const int vwoff = tc * 128 + tt * 4;                          // + (xi*6+nu)*256
  f32x4 Tc[6];                                                  // (d B)[y][nu], y = 0..5
  // row yy of the 6x6 patch: the pixels column NU of d B needs, combined with that column of B (= row of B^T).  Load and
  // combine sit in one slice: the LDS latency in between is covered by the SIMD's other wave.
  auto trow = [&](const float* rw, int yy, int nu) __attribute__((always_inline)) {
    const float* p0 = rw + tbase + yy * RS * PXS;
    f32x4 r;
    if (nu == 0 || nu == 5) {                                                   // 4 d0 - 5 d2 + d4  |  4 d1 - 5 d3 + d5
      const float* p = p0 + (nu == 0 ? 0 : PXS);
      const f32x4 l0 = *(const f32x4*)(p), l1 = *(const f32x4*)(p + 2 * PXS), l2 = *(const f32x4*)(p + 4 * PXS);
      r = 4.f * l0 - 5.f * l1 + l2;
    } else {
      const f32x4 l0 = *(const f32x4*)(p0 + PXS), l1 = *(const f32x4*)(p0 + 2 * PXS), l2 = *(const f32x4*)(p0 + 3 * PXS), l3 = *(const f32x4*)(p0 + 4 * PXS);
      if (nu == 1) r = (l3 - 4.f * l1) + (l2 - 4.f * l0);                       // (d4 - 4 d2) + (d3 - 4 d1)
      else if (nu == 2) r = (l3 - 4.f * l1) - (l2 - 4.f * l0);
      else if (nu == 3) r = (l3 - l1) + 2.f * (l2 - l0);                        // (d4 - d2) + 2 (d3 - d1)
      else r = (l3 - l1) - 2.f * (l2 - l0);
    }
    asm volatile("" : "+v"(r));                                 // keep the slice where it is written
    Tc[yy] = r;
  };
  // column pass of column nu, slice s = 0..3: xi = 0 | 1,2 | 3,4 | 5
  auto tcol = [&](float* Vd, int nu, int s) __attribute__((always_inline)) {
    float* v = Vd + nu * 256 + vwoff;
    if (s == 0) {
      *(f32x4*)(v) = 4.f * Tc[0] - 5.f * Tc[2] + Tc[4];
    } else if (s == 1) {
      const f32x4 a = Tc[4] - 4.f * Tc[2], b = Tc[3] - 4.f * Tc[1];
      *(f32x4*)(v + 1 * 6 * 256) = a + b;
      *(f32x4*)(v + 2 * 6 * 256) = a - b;
    } else if (s == 2) {
      const f32x4 c = Tc[4] - Tc[2], e = 2.f * (Tc[3] - Tc[1]);
      *(f32x4*)(v + 3 * 6 * 256) = c + e;
      *(f32x4*)(v + 4 * 6 * 256) = c - e;
    } else {
      *(f32x4*)(v + 5 * 6 * 256) = 4.f * Tc[1] - 5.f * Tc[3] + Tc[5];
    }
  };
  auto transform_all = [&](const float* rw, float* Vd, int nu) __attribute__((always_inline)) {
    if (nu > 5) return;
#pragma unroll
    for (int yy = 0; yy < 6; yy++) {
      trow(rw, yy, nu);
    }
#pragma unroll
    for (int s = 0; s < 4; s++) tcol(Vd, nu, s);
  };
  // transform work of the main loop cut into slices pinned between the MFMAs: slot k = 2 * step + i (i = 0, 1) of a phase of
  // ntw steps.  The four column slices take the last four slots, the six row slices the odd slots before them (ntw >= 8), or
  // slots 1 3 5 7 8 9 (ntw = 7), 2..7 (ntw = 6).
  auto tslice = [&](const float* rwn, float* Vn, int q, int i, int nu, int ntw) __attribute__((always_inline)) {
    if (nu > 5) return;
    const int k = 2 * q + i, k0 = 2 * ntw - 4;
    if (k >= k0) { tcol(Vn, nu, k - k0); return; }
    int r = -1;
    if (ntw >= 8) r = (k & 1) && k <= 11 ? (k - 1) / 2 : -1;
    else if (ntw == 7) r = k == 1 ? 0 : k == 3 ? 1 : k == 5 ? 2 : k >= 7 ? k - 4 : -1;
    else r = k >= 2 ? k - 2 : -1;
    if (r >= 0) trow(rwn, r, nu);
  };

  // -DW4_DIAG: per-wave s_memtime stamps (ticks parked at the end-of-phase wait + barrier, K-loop ticks, epilogue ticks)
  // written to y instead of the result; read by tools/diag_wino4.py
#ifdef W4_DIAG
  long dg_wait = 0, dg_t = 0;
  const long dg_start = __builtin_amdgcn_s_memtime();
#define W4_DIAG_T0 dg_t = __builtin_amdgcn_s_memtime();
#define W4_DIAG_T1 dg_wait += __builtin_amdgcn_s_memtime() - dg_t;
#else
#define W4_DIAG_T0
#define W4_DIAG_T1
#endif
#define W4_SB() __builtin_amdgcn_sched_barrier(0)
  const long UCH = 36 * 256;                                    // floats of packed weights per (32-channel group, chunk)

  // ---- main loop, uneven split.  The 72 accumulator tiles t = cg * 36 + (xi*6+nu) of the workgroup are dealt out as
  // 8 to each of waves 0..5 (which also carry a transform column) and 12 to waves 6 and 7 (which do not): measured with
  // the even 9/9 split, waves 6/7 spent 31-37 % of the K loop parked at the phase barrier while the SIMDs carrying two
  // transform waves set the pace (-DW4_DIAG).  Tile ranges: wave w < 6: [8w, 8w+8), wave 6: [48, 60), wave 7: [60, 72).
  f32x16 acc[13];
#pragma unroll
  for (int q = 0; q < 13; q++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
  const float* ubg0 = up + (long)(n_tile * 2) * NCH * UCH + (h * 32 + j) * 4;     // channel group cg = 0 / 1 of this n_tile
  const float* ubg1 = ubg0 + (long)NCH * UCH;
#ifndef W4_GLOBALW   // (-DW4_GLOBALW: the global_load form, for the A/B -- same process, 8 layers at B = 256: 30.65 vs 30.30 ms)
  // the packed weights through BUFFER loads: a wave-uniform descriptor on this n_tile's two channel groups, the lane's 16 bytes as the only
  // vector offset (loop-invariant), and (channel group, chunk, tile) as a SCALAR byte offset -- no 64-bit vector address arithmetic in the
  // loop (the global_load form pays a v_add_co / v_addc pair on every fourth load)
  typedef int w4_i32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int w4_u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void*)(up + (long)(n_tile * 2) * NCH * UCH), 0,
                                                                         (int)(2L * NCH * UCH * 4), 0x00020000);
  const int uvo = (h * 32 + j) * 16;
  const int ucg1 = (int)((long)NCH * UCH * 4);
#define W4_LDW(CG1, CH, XN) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, uvo, ((CG1) ? ucg1 : 0) + (int)((long)(CH) * UCH * 4) + (XN) * 1024, 0))
#else
#define W4_LDW(CG1, CH, XN) (*(const f32x4*)(((CG1) ? ubg1 : ubg0) + (long)(CH) * UCH + (XN) * 256))
#endif
  const float* va0 = V + h * 128 + j * 4;
  // accumulator tiles per wave (sum 72).  Default 8 8 8 8 | 9 9 11 11 (SIMDs 17 17 19 19).
#ifndef W4_N0
#define W4_N0 8
#define W4_N1 8
#define W4_N2 8
#define W4_N3 8
#define W4_N4 9
#define W4_N5 9
#define W4_N6 11
#define W4_N7 11
#endif
#define W4_T1 (W4_N0)
#define W4_T2 (W4_T1 + W4_N1)
#define W4_T3 (W4_T2 + W4_N2)
#define W4_T4 (W4_T3 + W4_N3)
#define W4_T5 (W4_T4 + W4_N4)
#define W4_T6 (W4_T5 + W4_N5)
#define W4_T7 (W4_T6 + W4_N6)
  static_assert(W4_T7 + W4_N7 == 72, "the eight waves must cover the 72 accumulator tiles");
  // ring size of the weight registers for a wave with N steps: every load needs at least three steps of lead (see below)
#define W4_RING(N) ((N) == 6 ? 3 : ((N) % 4 == 1 || (N) == 10) ? 5 : 4)   // (round 4: 5 slots for every wave measured +2 %: 256 registers, spills)
  const int t0w = wave == 0 ? 0 : wave == 1 ? W4_T1 : wave == 2 ? W4_T2 : wave == 3 ? W4_T3 : wave == 4 ? W4_T4 : wave == 5 ? W4_T5
                  : wave == 6 ? W4_T6 : W4_T7;
  const int ringw = wave == 0 ? W4_RING(W4_N0) : wave == 1 ? W4_RING(W4_N1) : wave == 2 ? W4_RING(W4_N2) : wave == 3 ? W4_RING(W4_N3)
                    : wave == 4 ? W4_RING(W4_N4) : wave == 5 ? W4_RING(W4_N5) : wave == 6 ? W4_RING(W4_N6) : W4_RING(W4_N7);
  // weights: ring of R register slots, step s uses slot s % R; right behind its MFMAs the slot is refilled with the weights of
  // its next user: step s+R of this chunk, or (s was the slot's last user) step s % R of the next chunk.  R = 4 gives at least
  // three steps of lead for 7, 8, 11 and 12 steps per phase; 9, 10 and 13 steps need R = 5 (with R = 4 step 8 and the next
  // chunk's step 0 would share a slot back to back), 6 steps R = 3.
  f32x4 Bw[5];
  f32x4 Aa, Ab;
#define W4_XN(T0, S) (((T0) + (S)) % 36)
#define W4_UB(T0, S) ((((T0) + (S)) / 36) ? ubg1 : ubg0)
#define W4_MF(S, ACUR, E, R) acc[S] = __builtin_amdgcn_mfma_f32_32x32x2f32(ACUR[E], Bw[(S) % (R)][E], acc[S], 0, 0, 0)
  // one tile: 4 MFMAs (8 channels); the other work is pinned between them in source order
#define W4_STEP(S, ACUR, ANEXT, NXT, NU, T0, NTW, R)                                                   \
  if ((S) < (NTW)) {                                                                                  \
    W4_MF(S, ACUR, 0, R);                                                                             \
    if ((S) + 1 < (NTW)) ANEXT = *(const f32x4*)(vc + W4_XN(T0, (S) + 1) * 256);                       \
    W4_SB();                                                                                          \
    W4_MF(S, ACUR, 1, R);                                                                             \
    if (NXT) tslice(rwn, Vn, S, 0, NU, NTW);                                             \
    W4_SB();                                                                                          \
    W4_MF(S, ACUR, 2, R);                                                                             \
    if (NXT) tslice(rwn, Vn, S, 1, NU, NTW);                                             \
    W4_SB();                                                                                          \
    W4_MF(S, ACUR, 3, R);                                                                             \
    if ((S) + (R) < (NTW)) {                                                                          \
      Bw[(S) % (R)] = W4_LDW(((T0) + (S) + (R)) / 36, chunk, W4_XN(T0, (S) + (R)));                       \
    } else if (NXT) {                                                                                 \
      Bw[(S) % (R)] = W4_LDW(((T0) + (S) % (R)) / 36, chunk + 1, W4_XN(T0, (S) % (R)));                   \
    }                                                                                                 \
    W4_SB();                                                                                          \
  }
#define W4_PHASE(NXT, NU, T0, NTW, R)                                                                 \
  {                                                                                                   \
    const float* vc = va0 + (chunk & 1) * VSZ;                                                        \
    float* Vn = V + ((chunk + 1) & 1) * VSZ;                                                          \
    const float* rwn = raw + ((chunk + 1) & 1) * RAWSZ;                                               \
    Aa = *(const f32x4*)(vc + W4_XN(T0, 0) * 256);                                                    \
    W4_STEP(0, Aa, Ab, NXT, NU, T0, NTW, R) W4_STEP(1, Ab, Aa, NXT, NU, T0, NTW, R) W4_STEP(2, Aa, Ab, NXT, NU, T0, NTW, R)     \
    W4_STEP(3, Ab, Aa, NXT, NU, T0, NTW, R) W4_STEP(4, Aa, Ab, NXT, NU, T0, NTW, R) W4_STEP(5, Ab, Aa, NXT, NU, T0, NTW, R)     \
    W4_STEP(6, Aa, Ab, NXT, NU, T0, NTW, R) W4_STEP(7, Ab, Aa, NXT, NU, T0, NTW, R) W4_STEP(8, Aa, Ab, NXT, NU, T0, NTW, R)     \
    W4_STEP(9, Ab, Aa, NXT, NU, T0, NTW, R) W4_STEP(10, Aa, Ab, NXT, NU, T0, NTW, R) W4_STEP(11, Ab, Aa, NXT, NU, T0, NTW, R)   \
    W4_STEP(12, Aa, Ab, NXT, NU, T0, NTW, R)                                                          \
  }
  // End of a phase: the raw patch of chunk+2 (LDS-DMA issued at the start of the phase) must have landed; the NTW weight
  // loads issued after it stay in flight.  Bare barrier (not __syncthreads(), whose fence would drain them); lgkmcnt(0)
  // retires the transform's LDS stores.
#define W4_MAIN(NU, T0, NTW, R)                                                                        \
  {                                                                                                   \
    int chunk = 0;                                                                                    \
    for (; chunk < NCH - 1; chunk++) {                                                                \
      W4_PHASE(true, NU, T0, NTW, R)                                                                   \
      W4_DIAG_T0                                                                                      \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NTW) : "memory");                \
      W4_DIAG_T1                                                                                      \
      if (chunk + 3 < NCH) stage_raw(chunk + 3, raw + ((chunk + 1) & 1) * RAWSZ);      \
    }                                                                                                 \
    W4_PHASE(false, NU, T0, NTW, R)                                                                    \
  }

  stage_raw(0, raw);
  if (NCH > 1) stage_raw(1, raw + RAWSZ);
#pragma unroll
  for (int sl = 0; sl < 5; sl++) {
    const int t = t0w + sl;
    if (sl < ringw) Bw[sl] = W4_LDW(t >= 36, 0, t % 36);
  }
  // builtin (not asm): the compiler's waitcnt bookkeeping then sees the prologue loads retired and derives the counted
  // waits of the loop from the loop's own issue order.  0x0F70 = vmcnt(0), expcnt/lgkmcnt untouched.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  transform_all(raw, V, wave);
  __syncthreads();
  if (NCH > 2) stage_raw(2, raw);

  switch (wave) {
    case 0: W4_MAIN(0, 0, W4_N0, W4_RING(W4_N0)) break;
    case 1: W4_MAIN(1, W4_T1, W4_N1, W4_RING(W4_N1)) break;
    case 2: W4_MAIN(2, W4_T2, W4_N2, W4_RING(W4_N2)) break;
    case 3: W4_MAIN(3, W4_T3, W4_N3, W4_RING(W4_N3)) break;
    case 4: W4_MAIN(4, W4_T4, W4_N4, W4_RING(W4_N4)) break;
    case 5: W4_MAIN(5, W4_T5, W4_N5, W4_RING(W4_N5)) break;
    case 6: W4_MAIN(6, W4_T6, W4_N6, W4_RING(W4_N6)) break;
    default: W4_MAIN(6, W4_T7, W4_N7, W4_RING(W4_N7)) break;
  }
#undef W4_MAIN
#undef W4_PHASE
#undef W4_STEP
#undef W4_MF
#undef W4_SB

#ifdef W4_DIAG
  const long dg_loop_end = __builtin_amdgcn_s_memtime();
#endif
  // ---- epilogue: exchange the 36 accumulators of each (tile, channel) through LDS, 4 rounds of 8 (channel group,
  // register) units; wave w output-transforms unit w of a round
  float* xch = V;                                               // [8 units][36][64 lanes] floats = 73.7 KB = both V buffers
  const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
  for (int rnd = 0; rnd < 4; rnd++) {
    __syncthreads();                                            // V (first round) / the previous round's reads are done
#define W4_EPI_WRITE(T0, NTW)                                                                         \
    _Pragma("unroll") for (int sq = 0; sq < (NTW); sq++) {                                             \
      float* d = xch + (((((T0) + sq) / 36) * 4) * 36 + ((T0) + sq) % 36) * 64 + lane;                 \
      _Pragma("unroll") for (int rr = 0; rr < 4; rr++) d[rr * 36 * 64] = acc[sq][4 * rnd + rr];        \
    }
    switch (wave) {
      case 0: W4_EPI_WRITE(0, W4_N0) break;
      case 1: W4_EPI_WRITE(W4_T1, W4_N1) break;
      case 2: W4_EPI_WRITE(W4_T2, W4_N2) break;
      case 3: W4_EPI_WRITE(W4_T3, W4_N3) break;
      case 4: W4_EPI_WRITE(W4_T4, W4_N4) break;
      case 5: W4_EPI_WRITE(W4_T5, W4_N5) break;
      case 6: W4_EPI_WRITE(W4_T6, W4_N6) break;
      default: W4_EPI_WRITE(W4_T7, W4_N7) break;
    }
#undef W4_EPI_WRITE
    __syncthreads();
    {
      const int unit = wave, c = unit >> 2, rr = unit & 3;
      const int r = 4 * rnd + rr;
      const float* s = xch + (unit * 36) * 64 + lane;
      // R[xi][jj] = sum_nu M[xi][nu] A[nu][jj]  (6 x 4), then Y[i][jj] = sum_xi A^T[i][xi] R[xi][jj]
      float Rm[6][4];
#pragma unroll
      for (int xi = 0; xi < 6; xi++) {
        const float m0 = s[(xi * 6 + 0) * 64], m1 = s[(xi * 6 + 1) * 64], m2 = s[(xi * 6 + 2) * 64], m3 = s[(xi * 6 + 3) * 64],
                    m4 = s[(xi * 6 + 4) * 64], m5 = s[(xi * 6 + 5) * 64];
        const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
        Rm[xi][0] = m0 + s1 + s2;
        Rm[xi][1] = d1 + 2.f * d2;
        Rm[xi][2] = s1 + 4.f * s2;
        Rm[xi][3] = d1 + 8.f * d2 + m5;
      }
      const int co = (n_tile * 2 + c) * 32 + j;
      const float bv = bias ? bias[co] : 0.f;
      float Y[4][4];
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        const float s1 = Rm[1][jj] + Rm[2][jj], d1 = Rm[1][jj] - Rm[2][jj], s2 = Rm[3][jj] + Rm[4][jj], d2 = Rm[3][jj] - Rm[4][jj];
        Y[0][jj] = Rm[0][jj] + s1 + s2 + bv;
        Y[1][jj] = d1 + 2.f * d2 + bv;
        Y[2][jj] = s1 + 4.f * s2 + bv;
        Y[3][jj] = d1 + 8.f * d2 + Rm[5][jj] + bv;
      }
      if (relu) {
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < 4; b++) Y[a][b] = fmaxf(Y[a][b], 0.f);
      }
      const int tile = (r & 3) + 8 * (r >> 2) + 4 * h;
#ifdef W4_DIAG
      if (tile < NTILE && relu == 12345) {                      // diagnostic build: the stamps go to y instead
#else
      if (tile < NTILE) {
#endif
        const int ty = tile / TX, tx = tile % TX;
        // stacked padded row R0 + 4 ty -> (image, row): at most G boundaries inside a patch, no runtime division
        int n = n0img, yy = y0 + 4 * ty;
        while (yy >= Hp) { yy -= Hp; n++; }
        const int xx = x0 + 4 * tx;
        if (n < Btot) {
          // one 64-bit base per lane and round; the 16 (4 pooled) store offsets are wave-uniform
          if (POOL) {
            float* yb = y + (((long)n * Ho + (yy >> 1)) * Wo + (xx >> 1)) * Cout + co;
            const int rs = Wo * Cout;
            const bool inside = (yy >> 1) + 1 < Ho && (xx >> 1) + 1 < Wo;        // false only on the padded rim (14x14 as 16x16)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
              for (int b = 0; b < 2; b++) {
                const float v = fmaxf(fmaxf(Y[2 * a][2 * b], Y[2 * a][2 * b + 1]), fmaxf(Y[2 * a + 1][2 * b], Y[2 * a + 1][2 * b + 1]));
                if (inside || ((yy >> 1) + a < Ho && (xx >> 1) + b < Wo)) yb[a * rs + b * Cout] = v;
              }
          } else {
            float* yb = y + (((long)n * H + yy) * W + xx) * Cout + co;
            const int rs = W * Cout;
            if (yy + 3 < H && xx + 3 < W) {
#pragma unroll
              for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) yb[a * rs + b * Cout] = Y[a][b];
            } else {
#pragma unroll
              for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++)
                  if (yy + a < H && xx + b < W) yb[a * rs + b * Cout] = Y[a][b];
            }
          }
        }
      }
    }
  }
#ifdef W4_DIAG
  __syncthreads();
  if (lane == 0) {
    const long dg_end = __builtin_amdgcn_s_memtime();
    float* dg = y + ((long)blockIdx.x * 8 + wave) * 4;
    dg[0] = (float)dg_wait; dg[1] = (float)(dg_loop_end - dg_start); dg[2] = (float)(dg_end - dg_loop_end); dg[3] = (float)wave;
  }
#endif
}

static int w4_max_boundaries(int H, int TH, long Rtot) {
  int best = 0;
  long lim = Rtot < (long)H * TH * 2 ? Rtot : (long)H * TH * 2;
  for (long R0 = 0; R0 < lim; R0 += TH) {
    long last = R0 + TH - 1 < Rtot - 1 ? R0 + TH - 1 : Rtot - 1;
    int g = (int)(last / H - R0 / H);
    if (g > best) best = g;
  }
  return best;
}

// patch (TY x TX tiles of 4x4 pixels): 1 = (4,8) 16x32 px, 2 = (8,4) 32x16, 3 = (16,2) 64x8, 4 = (32,1) 128x4.
// The width is padded to a multiple of the patch width: pick the patch with the least padding, the widest on a tie.
static int w4_tw(int sel) { return sel == 1 ? 32 : sel == 2 ? 16 : sel == 3 ? 8 : 4; }
static int w4_th(int sel) { return sel == 1 ? 16 : sel == 2 ? 32 : sel == 3 ? 64 : 128; }
static int w4_gmax(int sel) { return sel == 4 ? 5 : 2; }
static int w4_select(int W) {
  int best = 0, best_wp = 1 << 30;
  for (int sel = 1; sel <= 4; sel++) {
    const int tw = w4_tw(sel), wp = (W + tw - 1) / tw * tw;
    if (wp < best_wp) { best_wp = wp; best = sel; }
  }
  return best;
}
static int w4_hp(int H) { return (H + 3) / 4 * 4; }
static int w4_wp(int W, int sel) { const int tw = w4_tw(sel); return (W + tw - 1) / tw * tw; }

template <int TY, int TX, int GMAX>
static int launch_wino4(const float* x, const float* up, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                        int relu, int pool, int G, hipStream_t st) {
  constexpr int RS = 4 * TX + 2;
  constexpr int AROUNDS = ((4 * TY + 2 + GMAX) * RS + 63) / 64;
  constexpr int RAWSZ = 2 * AROUNDS * 64 * 4;
  constexpr int VSZ = 36 * 256;
  size_t lds = (size_t)(2 * VSZ + 2 * RAWSZ) * sizeof(float);
  const int Hp = w4_hp(H), Wp = (W + 4 * TX - 1) / (4 * TX) * (4 * TX);
  const int mtiles = cdiv((long)B * Hp, 4 * TY) * (Wp / (4 * TX));
  dim3 grid(((mtiles + 7) / 8) * 8 * (Cout / 64));
  if (pool) {
    auto k = conv3x3_wino4_kernel<TY, TX, GMAX, true>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(512), lds, st, x, up, bias, y, B, H, W, Cin, Cout, relu, G, Hp, Wp);
  } else {
    auto k = conv3x3_wino4_kernel<TY, TX, GMAX, false>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(512), lds, st, x, up, bias, y, B, H, W, Cin, Cout, relu, G, Hp, Wp);
  }
  return lct_check_launch("conv3x3_wino4");
}

extern "C" {

long lctvqa_conv3x3_wino4_packed_elems(int Cout, int Cin) { return (long)Cout * Cin * 36; }

int lctvqa_conv3x3_wino4_supported(int B, int H, int W, int Cin, int Cout, int pool) {
  if (Cin % 8 || Cin < 16 || Cout % 64 || B < 1 || H < 2 || W < 2) return 0;
  if (pool && ((H | W) & 1)) return 0;
  const int sel = w4_select(W);
  const int Hp = w4_hp(H);
  // useful share of the padded tile grid: below 3/4 the F(2x2,3x3) kernel (no padding for even sizes) is the better choice
  if ((long)H * W * 4 < (long)Hp * w4_wp(W, sel) * 3) return 0;
  if (w4_max_boundaries(Hp, w4_th(sel), (long)B * Hp) > w4_gmax(sel)) return 0;
  if ((long)B * H * W * Cin >= (1L << 31)) return 0;
  return 1;
}

int lctvqa_conv3x3_wino4_pack_weights(const float* w_oihw, float* u_packed, int Cout, int Cin, int flip_transpose, void* stream) {
  LCT_REQUIRE(Cin % 8 == 0 && Cout % 32 == 0, "wino4_pack_weights: need Cin%%8==0, Cout%%32==0 (got %d,%d)", Cin, Cout);
  long total = (long)Cout * Cin * 36;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(wino4_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, u_packed, Cout, Cin, flip_transpose);
  return lct_check_launch("wino4_pack");
}

int lctvqa_conv3x3_wino4_nhwc(const float* x, const float* u_packed, const float* bias, float* y, int B, int H, int W, int Cin,
                              int Cout, int relu, int pool, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  LCT_REQUIRE(lctvqa_conv3x3_wino4_supported(B, H, W, Cin, Cout, pool), "conv3x3_wino4: unsupported shape B=%d H=%d W=%d Cin=%d Cout=%d",
              B, H, W, Cin, Cout);
  const int sel = w4_select(W);
  const int G = w4_max_boundaries(w4_hp(H), w4_th(sel), (long)B * w4_hp(H));
  switch (sel) {
    case 1: return launch_wino4<4, 8, 2>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 2: return launch_wino4<8, 4, 2>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 3: return launch_wino4<16, 2, 2>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    case 4: return launch_wino4<32, 1, 5>(x, u_packed, bias, y, B, H, W, Cin, Cout, relu, pool, G, st);
    default: break;
  }
  lct_set_error("conv3x3_wino4: no tiling");
  return LCT_EUNSUPPORTED;
}

}  // extern "C"
