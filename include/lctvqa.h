/* lctvqa.h -- C ABI of the MI355X (gfx950) kernels behind the VQA train-step hot path.
 *
 * Library: lct_vqa_amd/_C/liblctvqa_hip.so (built by `python -m lct_vqa_amd.build`).
 *
 * The reference (aahamed/LCT-VQA) is pure Python/PyTorch: it has no FFI of its own.  The
 * boundary a maintainer binds is therefore the set of tensor ops that torch.nn dispatches for
 * the path named in BASELINE.json; every entry point below names the reference line whose
 * ATen/cuDNN/cuBLAS dispatch it replaces.  Conventions:
 *   - all pointers are DEVICE pointers (hipMalloc'ed / torch CUDA tensors' data_ptr()),
 *     fp32 unless stated; `long` is 64-bit (torch.int64 index tensors);
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); calls are asynchronous;
 *   - return 0 on success, <0 for invalid arguments (LCT_E*), >0 = hipError_t; the message
 *     is available from lctvqa_last_error() (thread local);
 *   - no call allocates, frees or synchronises: workspaces are caller-owned (graph-capturable).
 * See INTEGRATION.md for the ctypes binding used by lct_vqa_amd and for the binding a
 * maintainer of the reference would add.
 */
#ifndef LCTVQA_H
#define LCTVQA_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LCTVQA_OK 0
#define LCTVQA_EINVAL (-1)
#define LCTVQA_EUNSUPPORTED (-2)
#define LCTVQA_EWORKSPACE (-3)

int lctvqa_version(void);
const char* lctvqa_last_error(void);
int lctvqa_device_count(void);

/* ---- VGG19 trunk: torchvision vgg19.features/avgpool as called at basic_vqa/models.py:36,
 *      models_lct.py:61-63, darts_vqa/vqa_model.py:28-32.  Activations are NHWC. ---------- */

/* elements of the packed weight buffer for a Cout x Cin x 3 x 3 filter bank */
long lctvqa_conv3x3_packed_elems(int Cout, int Cin);
/* 1 if the MFMA kernel tiles this shape, 0 if the direct fallback kernel is used */
int lctvqa_conv3x3_uses_mfma(int B, int H, int W, int Cin, int Cout, int pool);
/* OIHW parameter -> [Cout/32][(Cin/16)*9][4][32][4] step tiles (run every forward: parameters
 * are read by pointer each call, see architect_lct.py:142-155,194-207) */
int lctvqa_conv3x3_pack_weights(const float* w_oihw, float* w_packed, int Cout, int Cin, void* stream);
/* nn.Conv2d(Cin,Cout,3,padding=1) + ReLU (+ MaxPool2d(2,2) when pool) on NHWC activations.
 * y: [B,H,W,Cout] or [B,H/2,W/2,Cout].  pool_tmp ([B,H,W,Cout]) is only used by the fallback. */
int lctvqa_conv3x3_nhwc(const float* x, const float* w_oihw, const float* w_packed, const float* bias, float* y,
                        float* pool_tmp, int B, int H, int W, int Cin, int Cout, int relu, int pool, void* stream);
/* Winograd F(2x2,3x3) variant of the same operator (fp32, 2.25x fewer MFMA FLOPs; differs from the direct kernel by
 * ordinary fp32 rounding).  u_packed: lctvqa_conv3x3_wino_packed_elems floats from lctvqa_conv3x3_wino_pack_weights
 * (flip_transpose=1 packs the weights of the dgrad convolution). */
long lctvqa_conv3x3_wino_packed_elems(int Cout, int Cin);
int lctvqa_conv3x3_wino_supported(int B, int H, int W, int Cin, int Cout, int pool);
int lctvqa_conv3x3_wino_pack_weights(const float* w_oihw, float* u_packed, int Cout, int Cin, int flip_transpose, void* stream);
int lctvqa_conv3x3_wino_nhwc(const float* x, const float* u_packed, const float* bias, float* y, int B, int H, int W, int Cin,
                             int Cout, int relu, int pool, void* stream);
/* Winograd F(4x4,3x3) variant (fp32, 4x fewer MFMA FLOPs than direct; H, W multiples of 4; differs from the direct kernel by
 * fp32 rounding of the larger transforms, ~4e-6 of the layer's largest output).  Same calling convention as the F(2x2,3x3)
 * entry points above; u_packed holds lctvqa_conv3x3_wino4_packed_elems floats. */
long lctvqa_conv3x3_wino4_packed_elems(int Cout, int Cin);
int lctvqa_conv3x3_wino4_supported(int B, int H, int W, int Cin, int Cout, int pool);
int lctvqa_conv3x3_wino4_pack_weights(const float* w_oihw, float* u_packed, int Cout, int Cin, int flip_transpose, void* stream);
int lctvqa_conv3x3_wino4_nhwc(const float* x, const float* u_packed, const float* bias, float* y, int B, int H, int W, int Cin,
                              int Cout, int relu, int pool, void* stream);
/* first layer: NCHW image [B,3,H,W] (the layout the reference's loaders produce,
 * data_loader_v2.py:82-86) -> NHWC [B,H,W,Cout], conv3x3 + bias + ReLU */
int lctvqa_conv3x3_first_nchw(const float* x_nchw, const float* w_oihw, const float* bias, float* y_nhwc, int B,
                              int H, int W, int Cout, int relu, void* stream);
int lctvqa_maxpool2x2_nhwc(const float* x, float* y, int B, int H, int W, int C, void* stream);
/* AdaptiveAvgPool2d((7,7)) + torch.flatten(x,1) in NCHW feature order: NHWC -> [B, C*49] */
int lctvqa_adaptive_avgpool7_flatten(const float* x_nhwc, float* y, int B, int H, int W, int C, void* stream);

/* ---- backward of the VGG19 conv blocks (trainable trunk: models_lct.py:62-63, pretrained=False) ------- */
/* weights of the dgrad convolution (run lctvqa_conv3x3_nhwc on dZ with these; Cin/Cout swap roles) */
int lctvqa_conv3x3_pack_weights_dgrad(const float* w_oihw, float* w_packed, int Cout, int Cin, void* stream);
/* dZ = relu'(act) * unpool(dout): act [B,H,W,C] post-ReLU; dout [B,H/2,W/2,C] if pool else [B,H,W,C] */
int lctvqa_relu_pool_bwd_nhwc(const float* dout, const float* act, float* dz, int B, int H, int W, int C, int pool, void* stream);
int lctvqa_adaptive_avgpool7_flatten_bwd(const float* dy, float* dx_nhwc, int B, int H, int W, int C, void* stream);
/* dW (OIHW) = sum_pixels dz[p][co] * x[p+tap][ci];  x, dz NHWC; scratch: ..._scratch_elems floats */
long lctvqa_conv3x3_wgrad_scratch_elems(int B, int H, int W, int Cin, int Cout);
int lctvqa_conv3x3_wgrad_nhwc(const float* x, const float* dz, float* dw_oihw, float* scratch, int B, int H, int W, int Cin,
                              int Cout, void* stream);
/* the same gradient in the Winograd domain, F(3x3,4x4): a quarter of the multiply-adds as 36 batched fp32 GEMMs over the tile axis
 * (models_lct.py:62-63 trainable VGG19 trunk; the ResNet bottleneck's 3x3).  _supported: 1 where it beats the direct kernel (Cin, Cout
 * >= 128 and one of them >= 256, map at least 3/4 of its 4x4 tile grid, >= 256 tiles); workspace in floats. */
int lctvqa_conv3x3_wino4_wgrad_supported(int B, int H, int W, int Cin, int Cout);
long lctvqa_conv3x3_wino4_wgrad_workspace_elems(int B, int H, int W, int Cin, int Cout);
int lctvqa_conv3x3_wino4_wgrad(const float* x, const float* dz, float* dw_oihw, float* workspace, long workspace_elems, int B, int H,
                               int W, int Cin, int Cout, void* stream);
/* F(4x4,3x3) forward as three kernels (input transform, 36 batched fp32 GEMMs, output transform + bias + ReLU + pool): the split form
 * of conv3x3_wino4_nhwc, built as an A/B for the 28x28 512-channel layers (models.py:35-36) */
long lctvqa_conv3x3_wino4_split_workspace_elems(int B, int H, int W, int Cin, int Cout);
int lctvqa_conv3x3_wino4_split_pack_weights(const float* w_oihw, float* U, int Cout, int Cin, void* stream);
int lctvqa_conv3x3_wino4_split_nhwc(const float* x, const float* U, const float* bias, float* y, float* workspace, long workspace_elems,
                                    int B, int H, int W, int Cin, int Cout, int relu, int pool, void* stream);
long lctvqa_conv3x3_first_wgrad_scratch_elems(int B, int H, int W, int Cout);
int lctvqa_conv3x3_first_wgrad(const float* x_nchw, const float* dz_nhwc, float* dw, float* db, float* scratch, int B, int H,
                               int W, int Cout, void* stream);

/* ---- GEMM: every nn.Linear on the path (models.py:29,37,53,66,80,81; models_lct.py:87-88)
 *      and both of its backward products, via operand strides.
 *      C[m][n] = mask*mask_scale*act(sum_k A[m*a_ms+k*a_ks]*B[n*b_ns+k*b_ks] + bias[n] + addend[m][n])
 *      act: 0 none, 1 relu, 2 tanh.  bias/addend/mask may be NULL. ------------------------ */
long lctvqa_gemm_workspace_bytes(int M, int N, int K);
int lctvqa_gemm_f32(const float* A, long a_ms, long a_ks, const float* B, long b_ns, long b_ks, float* C, long ldc,
                    int M, int N, int K, const float* bias, const float* addend, long ld_add, int act,
                    const uint8_t* mask, long ld_mask, float mask_scale, float* workspace, long workspace_bytes,
                    void* stream);
/* same with ReLU fused into the operand loads: relu_ab bit 0 -> A, bit 1 -> B (PC-DARTS ReLU->Conv1x1 blocks,
 * pcdarts/operations.py:27-28,43,58-61, and their weight gradients) */
int lctvqa_gemm_f32_ex(const float* A, long a_ms, long a_ks, const float* B, long b_ns, long b_ks, float* C, long ldc,
                       int M, int N, int K, const float* bias, const float* addend, long ld_add, int act,
                       const uint8_t* mask, long ld_mask, float mask_scale, float* workspace, long workspace_bytes,
                       int relu_ab, void* stream);
/* Both entry points above run on the LDS-DMA kernel (256 x 128 x 16 tiles, operands HBM -> LDS by global_load_lds) whenever both
 * operands are 16-byte aligned and K- or row-contiguous with K % 4 == 0 (operand ReLU included: applied to the fragments); everything
 * else takes the general VGPR-staged kernel.  lctvqa_gemm_force_general(1) routes every product through the general kernel (A/B, tests);
 * returns the previous setting. */
int lctvqa_gemm_force_general(int on);
/* C[b] = act(A[b] . B[b]^T + bias) for b < batch on the LDS-DMA kernel: operands K-contiguous (X(r,k) = X[r*rs + k]) or row-contiguous
 * (X(r,k) = X[k*ks + r]), 16-byte aligned, K % 4 == 0; element strides sA, sB, sC between batch members (the per-(xi,nu) products of the
 * Winograd-domain weight gradient; nn.Linear over a stack of inputs).  _nt: both operands K-contiguous. */
int lctvqa_gemm_f32_batched(const float* A, long a_ms, long a_ks, long sA, const float* B, long b_ns, long b_ks, long sB, float* C,
                            long ldc, long sC, int batch, int M, int N, int K, const float* bias, int act, void* stream);
int lctvqa_gemm_f32_batched_nt(const float* A, long lda, long sA, const float* B, long ldb, long sB, float* C, long ldc, long sC,
                               int batch, int M, int N, int K, const float* bias, int act, void* stream);
/* fusion head, first half (models.py:96-100) in ONE GEMM launch: the pointwise mul+tanh+dropout(mask1) is applied to the A
 * operand while it is staged: h1 = mask2*scale2*tanh([mask1*scale1*tanh(img.qst)] . W1^T + b1); masks dense uint8 or NULL */
int lctvqa_fusion_fc1(const float* img, const float* qst, const uint8_t* mask1, float scale1, const float* w1, const float* b1,
                      const uint8_t* mask2, float scale2, float* h1, int B, int E, int A, float* workspace, long workspace_bytes,
                      void* stream);
/* out[n] = sum_m X[m][n]  (bias gradients); scratch: lctvqa_colsum_scratch_elems(M,N) floats */
long lctvqa_colsum_scratch_elems(int M, int N);
int lctvqa_colsum_f32(const float* X, long ld, int M, int N, float* out, float* scratch, void* stream);

/* ---- question encoder: nn.Embedding + tanh + nn.LSTM (models.py:58-61, models_lct.py:101-108) */
/* idx [B][T] int64 -> out [T][B][E] = (tanh) emb[idx]  (transpose(0,1) fused) */
int lctvqa_embed_gather(const long* idx, const float* emb, float* out, int B, int T, int E, int V, int apply_tanh, void* stream);
/* demb[idx[b,t]] += dy[t,b,:] * (apply_tanh ? 1-y^2 : 1)   (dense grad; demb pre-zeroed) */
int lctvqa_embed_scatter_add(const long* idx, const float* dy, const float* y, float* demb, int B, int T, int E, int V,
                             int apply_tanh, void* stream);
/* one LSTM time step, gate order i,f,g,o.  pre-activations = xp + hp, each [B][4H] */
int lctvqa_lstm_cell_fwd(const float* xp, const float* hp, const float* c_prev, float* gates, float* c_out, float* h_out,
                         int B, int H, void* stream);
/* fused time step: gates = xp + h_prev.W_hh^T + b_hh -> cell update, one launch, no split-K / pointwise passes */
int lctvqa_lstm_step_fused_supported(int B, int H);
int lctvqa_lstm_step_fused(const float* xp, const float* h_prev, const float* c_prev, const float* w_hh,
                           const float* b_hh, float* gates, float* c_out, float* h_out, int B, int H, void* stream);
/* fused BPTT step: dh_rec = dgates_t . W_hh, then the cell backward of step t-1 in the epilogue */
int lctvqa_lstm_step_bwd_fused(const float* dg_t, const float* w_hh, const float* dh_ext, const float* dc_next,
                               const float* gates_p, const float* c_pp, const float* c_p, float* dg_p, float* dc_prev,
                               int B, int H, void* stream);
int lctvqa_lstm_cell_bwd(const float* dh_rec, const float* dh_ext, const float* dc_next, const float* gates,
                         const float* c_prev, const float* c_cur, float* dgates, float* dc_prev, int B, int H, void* stream);
/* whole-sequence recurrence (the T dependent steps of nn.LSTM, models.py:61, models_lct.py:107-108): W_hh is re-ordered
 * once per call into MFMA-fragment order (wpack: lctvqa_lstm_seq_scratch_elems(H) floats), then the T steps are launched
 * back to back.  lctvqa_lstm_seq_packed(B,H) tells whether the shape runs on the packed kernels (H = 64/128/256/512) or
 * falls back to the per-step kernels above.  fwd: h_all/c_all [T+1][B][H], slot 0 = initial state; fills slots 1..T and
 * gates [T][B][4H].  bwd: steps T-2..0 (the caller has run lctvqa_lstm_cell_bwd for step T-1 into dgates[T-1] and
 * dc_buf[0]); dout [T][B][H] or NULL; dc_buf [2][B][H], result in half (T-1)&1 */
long lctvqa_lstm_seq_scratch_elems(int H);
int lctvqa_lstm_seq_packed(int B, int H);
int lctvqa_lstm_seq_fwd(const float* xp, const float* w_hh, const float* b_hh, float* h_all, float* c_all, float* gates,
                        float* wpack, int T, int B, int H, void* stream);
int lctvqa_lstm_seq_bwd(float* dgates, const float* w_hh, const float* dout, const float* gates, const float* c_all,
                        float* dc_buf, float* wpack, int T, int B, int H, void* stream);
/* the same two calls with the recurrent products h . W_hh^T and dg . W_hh on the bf16 matrix cores (operands rounded to bf16 per step,
 * fp32 accumulation, fp32 gates and cell state): the bf16 compute dtype of BASELINE.json configs[2-4]; the greedy decode stays fp32 */
int lctvqa_lstm_seq_fwd_bf16(const float* xp, const float* w_hh, const float* b_hh, float* h_all, float* c_all, float* gates,
                        float* wpack, int T, int B, int H, void* stream);
int lctvqa_lstm_seq_bwd_bf16(float* dgates, const float* w_hh, const float* dout, const float* gates, const float* c_all,
                        float* dc_buf, float* wpack, int T, int B, int H, void* stream);
/* greedy question generation (models_lct.py:124-157, darts_vqa/vqa_model.py:190-222) as one call: x_0 = tanh(emb[start]),
 * x_t = emb[pred_{t-1}]; (h,c) = LSTM step; pred_t = argmax(fc(tanh(h))).  h, c [B][H] in/out; tokens [B][L] int64 out;
 * scratch: lctvqa_lstm_decode_scratch_elems(B,H,We,QV) floats */
long lctvqa_lstm_decode_scratch_elems(int B, int H, int We, int QV);
int lctvqa_lstm_decode_greedy(const float* emb, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                              const float* fc_w, const float* fc_b, float* h, float* c, long* tokens, long start_token, int L,
                              int B, int H, int We, int QV, float* scratch, void* stream);
/* GRU cell (nn.GRU, gates r,z,n): the "LSTM/GRU" question-encoder variant of BASELINE.json (not in the reference;
 * oracle = torch.nn.GRU).  xp/hp: [B][3H] projections incl. their biases; saved [B][4H] = r,z,n,hp_n */
int lctvqa_gru_cell_fwd(const float* xp, const float* hp, const float* h_prev, float* saved, float* h_out, int B, int H,
                        void* stream);
int lctvqa_gru_cell_bwd(const float* dh_rec, const float* dh_ext, const float* saved, const float* h_prev, float* dxp,
                        float* dhp, float* dh_carry, int B, int H, void* stream);
/* tanh(cat(h,c)) of models.py:62-65 (num_layers = 1) and its backward */
int lctvqa_cat_tanh_fwd(const float* h, const float* c, float* out, int B, int H, void* stream);
int lctvqa_cat_tanh_bwd(const float* dy, const float* y, float* dh, float* dc, int B, int H, void* stream);
int lctvqa_tanh_fwd(const float* x, float* y, long n, void* stream);
/* backward of the GEMM epilogue: dx = dy * mask*scale * act'(.) with y = the epilogue's output */
int lctvqa_act_bwd(const float* dy, const float* y, const uint8_t* mask, float scale, float* dx, long n, int act, void* stream);

/* ---- fusion head: mul -> tanh -> dropout (models.py:96-98) and image-feature L2 norm (models.py:39-40) */
int lctvqa_mul_tanh_drop_fwd(const float* a, const float* b, const uint8_t* mask, float scale, float* t, float* z, long n, void* stream);
int lctvqa_mul_tanh_drop_bwd(const float* dz, const float* a, const float* b, const float* t, const uint8_t* mask, float scale,
                             float* da, float* db, long n, void* stream);
int lctvqa_l2norm_rows_fwd(const float* x, float* y, float* inv_norm, int M, int N, void* stream);
int lctvqa_scale_rows(const float* x, const float* s, float* y, int M, int N, void* stream);

/* ---- losses: nn.CrossEntropyLoss (kind 0) and softXEnt (kind 1), models.py:7-10,89,115-125.
 *      loss_out[0] = loss_scale * sum_rows;  dlogits (optional) = gscale * d(row loss)/d logits */
/*      dtarget (optional, kind 1, dense [M][N]) = gscale * d(row loss)/d target = gscale * -(log_softmax) */
int lctvqa_xent(const float* logits, long ld, const long* labels, const float* target, long ldt, float* row_loss,
                float* loss_out, float loss_scale, float* dlogits, long ldd, float gscale, float* dtarget, int M, int N, int kind,
                void* stream);
/* F.softmax(x * inv_temp, dim=1)  (experiment.py:245, architect_lct.py:55) */
int lctvqa_softmax_rows(const float* x, float* y, int M, int N, float inv_temp, void* stream);
int lctvqa_softmax_rows_bwd(const float* dy, const float* y, float* dx, int M, int N, float inv_temp, void* stream);
/* torch.argmax(x, 1) (first maximal index) -> int64  (models_lct.py:161, experiment.py:218) */
int lctvqa_argmax_rows(const float* x, long ld, int M, int N, long* out, void* stream);
/* validation counters (basic_vqa/experiment.py:314-324 + misc.py:42-45 num_correct): row argmax of the answer logits x[M,N]
 * against K multiple-choice answers choices[K][M] (int64).  counts2[0] += #hits with "<unk>" accepted, counts2[1] +=
 * #hits with a prediction == unk_idx replaced by -9999.  counts2 accumulates (zero it per epoch); pred_out may be null. */
int lctvqa_eval_multi_choice(const float* x, long ld, int M, int N, const long* choices, int K, long unk_idx, long* pred_out,
                             long* counts2, void* stream);
/* nn.Dropout keep mask from a counter-based generator: mask[i] = U(seed, offset+i) >= p */
int lctvqa_dropout_mask(uint8_t* mask, long n, unsigned long long seed, unsigned long long offset, float p, void* stream);

/* ---- PC-DARTS search-cell primitives on NHWC activations (pcdarts/operations.py:4-104,
 *      pcdarts/model_search.py:11-91).  x dense [B,H,W,C] unless a row stride ld* is given. -------------- */
int lctvqa_slice_channels(const float* x, long ldx, int c0, float* y, int Cn, long P, void* stream);
int lctvqa_slice_channels_bwd(const float* dy, int Cn, int c0, float* dx, int C, long P, void* stream);
int lctvqa_copy_into_channels(const float* x, int Cn, float* y, long ldy, int c0, long P, void* stream);
/* depthwise KxK conv (groups=C), w [C][K][K], optional ReLU on the input (ReLU->dwconv of SepConv/DilConv) */
int lctvqa_dwconv_fwd(const float* x, const float* w, float* y, int B, int H, int W, int C, int K, int stride, int pad, int dil,
                      int relu_in, void* stream);
int lctvqa_dwconv_bwd_input(const float* dy, const float* w, const float* x, float* dx, int B, int H, int W, int C, int K,
                            int stride, int pad, int dil, int relu_in, void* stream);
long lctvqa_dwconv_bwd_weight_scratch_elems(int B, int H, int W, int C, int K, int stride, int pad, int dil);
int lctvqa_dwconv_bwd_weight(const float* dy, const float* x, float* dw, float* scratch, int B, int H, int W, int C, int K,
                             int stride, int pad, int dil, int relu_in, void* stream);
/* 1x1-conv weight gradient for small channel counts (Cout*Cin <= 4096): dw[Cout][Cin] = sum_p dy[p][Cout] x relu?(x[p][Cin]),
 * P = B*H*W pixels (operations.py:28,42,57,61 backward; replaces a P-deep 128x128 GEMM tile that is >98 % padding) */
/* 1x1 conv with few channels (Cin, Cout in {4,8,16,32,64}; operations.py:28,42,57,61), thread-per-pixel instead of a GEMM tile:
 * transpose = 0: out[P][Cout] = relu_in?(in[P][Cin]) . w[Cout][Cin]^T;   transpose = 1 (input gradient): out[P][Cin] =
 * in[P][Cout] . w[Cout][Cin], multiplied by (relu_mask_x[P][Cin] > 0) when relu_mask_x is given */
int lctvqa_pwconv_small_supported(long P, int Cout, int Cin);
int lctvqa_pwconv_small(const float* in, const float* w, const float* relu_mask_x, float* out, long P, int Cout, int Cin, int relu_in,
                        int transpose, void* stream);
long lctvqa_pwconv_wgrad_small_scratch_elems(long P, int Cout, int Cin);
int lctvqa_pwconv_wgrad_small_supported(long P, int Cout, int Cin);
int lctvqa_pwconv_wgrad_small(const float* dy, const float* x, float* dw, float* scratch, long P, int Cout, int Cin, int relu_in,
                              void* stream);
/* 3x3 pooling pad 1: kind 0 = MaxPool2d, 1 = AvgPool2d(count_include_pad=False) */
int lctvqa_pool3x3_fwd(const float* x, float* y, int B, int H, int W, int C, int stride, int kind, void* stream);
int lctvqa_pool3x3_bwd(const float* dy, const float* x, float* dx, int B, int H, int W, int C, int stride, int kind, void* stream);
/* y = relu?(x[:, off::2, off::2, :])  (FactorizedReduce taps) and the joint backward of both taps */
int lctvqa_subsample2_fwd(const float* x, float* y, int B, int H, int W, int C, int off, int relu, void* stream);
int lctvqa_subsample2_bwd(const float* dy0, const float* dy1, const float* x, float* dx, int B, int H, int W, int C, int relu,
                          void* stream);
/* BatchNorm2d over P = B*H*W rows of C channels (row strides ldx/ldy): batch statistics when training (running estimates
 * and the int64 num_batches_tracked counter updated in the same call when given), else running statistics */
long lctvqa_bn_scratch_doubles(long P, int C);
int lctvqa_batchnorm_fwd(const float* x, long ldx, float* y, long ldy, long P, int C, const float* gamma, const float* beta,
                         float* run_mean, float* run_var, long* batches_tracked, float* mean, float* invstd, double* scratch,
                         int training, float momentum, float eps, void* stream);
int lctvqa_batchnorm_bwd(const float* dy, long lddy, const float* x, long ldx, const float* mean, const float* invstd,
                         const float* gamma, float* dx, float* dgamma, float* dbeta, float* sums, double* scratch, long P, int C,
                         int training, void* stream);
/* MixedOp tail: out (+)= beta * channel_shuffle(cat(sum_k w_k branch_k, bypass(x)), 4)  (model_search.py:41-55,87) */
int lctvqa_mixed_combine_fwd(const float* const* branches, int nbr, const float* wts, const float* beta, const float* x, int B,
                             int Ho, int Wo, int C, int reduce, float* out, long ldo, int accumulate, void* stream);
int lctvqa_mixed_combine_bwd_branch(const float* dout, long ldd, const float* wts, int k, const float* beta, float* dbr, long P, int Cq,
                                    void* stream);
int lctvqa_mixed_combine_bwd_x(const float* dout, long ldd, const float* beta, const float* x, float* dx, int B, int H, int W, int C,
                               int reduce, void* stream);
long lctvqa_mixed_combine_scratch_elems(int B, int Ho, int Wo);
int lctvqa_mixed_combine_bwd_scalars(const float* dout, long ldd, const float* const* branches, int nbr, const float* wts,
                                     const float* beta, const float* x, int B, int Ho, int Wo, int C, int reduce, float* dw,
                                     float* dbeta, float* scratch, void* stream);

/* ---- bf16 compute path of a FROZEN VGG19 trunk (BASELINE.json's bf16 configurations; the reference is fp32 throughout,
 *      basic_vqa/models.py:23-42): bf16 NHWC activations and packed weights, fp32 accumulation on
 *      v_mfma_f32_32x32x16_bf16, fused bias + ReLU (+ 2x2 max pool): csrc/conv3x3_bf16.hip ---- */
int lctvqa_conv3x3_bf16_supported(int B, int H, int W, int Cin, int Cout, int pool);
int lctvqa_conv3x3_bf16_pack_weights(const float* w_oihw, void* w_packed /* Cout*Cin*9 bf16 */, int Cout, int Cin, void* stream);
/* packed weights of the data-gradient convolution (channels transposed, taps flipped) of a layer with OIHW weights [Cout][Cin][3][3] */
int lctvqa_conv3x3_bf16_pack_weights_dgrad(const float* w_oihw, void* w_packed, int Cout, int Cin, void* stream);
int lctvqa_conv3x3_bf16_nhwc(const void* x_bf16, const void* w_packed, const float* bias, void* y_bf16, int B, int H, int W, int Cin,
                             int Cout, int relu, int pool, void* stream);
int lctvqa_conv3x3_first_nchw_bf16(const float* x_nchw, const float* w_oihw, const float* bias, void* y_nhwc_bf16, int B, int H, int W,
                                   int Cout, int relu, void* stream);
/* the same layer on the matrix cores for Cout = 64 (implicit GEMM, K = 9 (c,ky) rows x 8): image and weights rounded to bf16 */
int lctvqa_conv3x3_first_nchw_bf16_mfma(const float* x_nchw, const float* w_oihw, const float* bias, void* y_nhwc, int B, int H, int W,
                                        int relu, void* stream);
int lctvqa_adaptive_avgpool7_flatten_bf16(const void* x_nhwc_bf16, float* y, int B, int H, int W, int C, void* stream);

/* ---- frozen ResNet-152 trunk (BASELINE.json configs[2]; not in the reference): what the VGG / GEMM
 *      kernels do not already cover -- BatchNorm (running statistics) folded into the convolution in front of it,
 *      the 7x7 / stride-2 stem (fp32 NCHW -> NHWC, bias + ReLU), global average pool: csrc/resnet.hip ---- */
int lctvqa_fold_bn(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, float eps, int Cout,
                   long K, float* w_out, float* shift, void* stream);
int lctvqa_conv7x7_s2_stem(const float* x_nchw, const float* w_oihw, const float* bias, float* y_nhwc, int B, int H, int W, int relu,
                           void* stream);
int lctvqa_global_avgpool_nhwc(const float* x, float* y, int B, int HW, int C, void* stream);
/* BatchNorm2d over dense NHWC rows [P][C] for any C % 4 == 0, fused with the ReLU / residual add that follows it in a bottleneck:
 * y = relu?(bn(x) (+ residual)); batch statistics (running estimates + counter updated) when training, else running statistics.
 * Backward: g = dy * (y_out > 0) when y_out is given, dres = g, dx / dgamma / dbeta as nn.BatchNorm2d's. */
int lctvqa_bn_wide_blocks(long P);
int lctvqa_bn_wide_fwd(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var, long* tracked,
                       const float* residual, int relu, float* y, float* mean, float* invstd, double* part, long P, int C, int training,
                       float momentum, float eps, void* stream);
int lctvqa_bn_wide_bwd(const float* dy, const float* y_out, const float* x, const float* mean, const float* invstd, const float* gamma,
                       float* dx, float* dres, float* dgamma, float* dbeta, float* sums, double* part, long P, int C, int training,
                       void* stream);
/* SyncBN forms of the two calls above (data-parallel training of a BatchNorm trunk: SURVEY.md section 8e): bn_wide_sums leaves the
 * per-channel fp64 sums [C][2] of THIS rank ((sum x, sum x^2), or with bwd (sum g, sum g xhat)) for the caller's all-reduce; the
 * _from_sums calls finish from sums over `count` elements per channel (all ranks').  part: lctvqa_bn_wide_blocks(P) * C * 2 doubles. */
int lctvqa_bn_wide_sums(const float* x, const float* dy, const float* y_out, const float* mean, const float* invstd, long P, int C, int bwd,
                        double* part, double* sums64, void* stream);
int lctvqa_bn_wide_fwd_from_sums(const float* x, const double* sums64, double count, const float* gamma, const float* beta, float* run_mean,
                                 float* run_var, long* tracked, const float* residual, int relu, float* y, float* mean, float* invstd, long P,
                                 int C, float momentum, float eps, void* stream);
int lctvqa_bn_wide_bwd_from_sums(const float* dy, const float* y_out, const float* x, const float* mean, const float* invstd, const float* gamma,
                                 const double* sums_local, const double* sums_global, double count, float* dx, float* dres, float* dgamma,
                                 float* dbeta, float* sums, long P, int C, void* stream);
int lctvqa_global_avgpool_nhwc_bwd(const float* dy, float* dx, int B, int HW, int C, void* stream);
/* weight gradient of the 7x7 / stride-2 stem: per-workgroup partials [blocks][147 (c,ky,kx)][64 cout] */
int lctvqa_conv7x7_s2_stem_wgrad_blocks(int B, int H);
int lctvqa_conv7x7_s2_stem_wgrad(const float* x_nchw, const float* dz_nhwc, float* part, int B, int H, int W, void* stream);
int lctvqa_conv7x7_s2_stem_wgrad_bf16(const float* x_nchw, const void* dz_nhwc_bf16, float* part, int B, int H, int W, void* stream);
/* dst[n][oy+y][ox+x][:] = relu?(src[n][sy+y][sx+x][:]) for an h x w window of NHWC tensors (pad a small map / crop it back) */
int lctvqa_copy_window_nhwc(const float* src, float* dst, int B, int Hs, int Ws, int Hd, int Wd, int C, int sy, int sx, int oy, int ox,
                            int h, int w, int relu, void* stream);

/* ---- bf16 GEMM (fp32 accumulation, v_mfma_f32_32x32x16_bf16) and the bf16 helpers of the bf16 ResNet trunk: csrc/gemm_bf16.hip.
 *      C[M][N] = act(A[M][K] . B[N][K]^T + bias[N] + addend[M][N]); A, B bf16 rows (strides in elements); C / addend bf16 when
 *      out_bf16 else fp32 (addend may alias C); act 0 none / 1 ReLU ---- */
int lctvqa_gemm_bf16_supported(long M, int N, int K);
int lctvqa_gemm_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, long M, int N, int K, const float* bias,
                     const void* addend, long ld_add, int act, int out_bf16, void* stream);
/* ---- the heads under the bf16 compute dtype (BASELINE.json's bf16 configurations; fp32 master weights, fp32 accumulation, fp32
 *      outputs).  What they replace: every nn.Linear product of reference basic_vqa/models.py:37,66,99-102 and models_lct.py:116,120
 *      and the LSTM / GRU input projections (models.py:61), forward and both backward products, as NT GEMMs on bf16 copies:
 *      lctvqa_cvt_pad_bf16 makes the (zero-padded, optionally transposed) bf16 operand, lctvqa_gemm_bf16x multiplies:
 *        C[M][N] (fp32) = mask * mask_scale * act(A[M][K] . B[Np][K]^T + bias[n] + addend[m][n]),  act 0 none / 1 ReLU / 2 tanh,
 *        K % 32 == 0 (zero padded), B padded to Np rows (Np % 64 == 0 >= N); few-tile problems split K into fp32 slabs in workspace.
 *      lctvqa_linear_xent_bf16_{fwd,bwd}: the LCT token head Linear(H, QV) over all tokens fused with its cross entropy
 *      (models_lct.py:118-120, 253-255): the [tokens, QV] logits never reach HBM.  fwd: loss = mean_m(lse[m] - logit[m][labels[m]]);
 *      bwd: dl = (softmax - onehot) * g[0] / M as bf16 [M][Np] and transposed [Np][ld_dlT >= 256-padded M] (the A operands of
 *      dX = dl . W and dW = dl^T . X), dbpart [2 * ceil(M/256)][Np] partial column sums (bias gradient).  Np % 128 == 0. ---- */
long lctvqa_gemm_bf16x_workspace_bytes(long M, int N, int Np, int K);
int lctvqa_gemm_bf16x(const void* A, long lda, const void* B, long ldb, float* C, long ldc, long M, int N, int Np, int K, const float* bias,
                      const float* addend, long ld_add, int act, const uint8_t* mask, long ld_mask, float mask_scale, float* workspace,
                      long workspace_bytes, void* stream);
int lctvqa_cvt_pad_bf16(const float* src, long ld_src, int rows, int cols, void* dst, long ld_dst, int rows_pad, int cols_pad, int transpose,
                        void* stream);
int lctvqa_cvt_pad_bf16_both(const float* src, long ld_src, int rows, int cols, void* plain, long ld_plain, void* dst_t, long ld_t,
                             int rows_pad, int cols_pad, void* stream);
long lctvqa_linear_xent_bf16_part_elems(long M, int Np);
int lctvqa_linear_xent_bf16_fwd(const void* X, long ldx, const void* W, long ldw, const float* bias, const long* labels, long M, int N,
                                int Np, int K, float* part, float* picked, float* lse, float* loss, void* stream);
int lctvqa_linear_xent_bf16_bwd(const void* X, long ldx, const void* W, long ldw, const float* bias, const long* labels, const float* lse,
                                const float* g, long M, int N, int Np, int K, void* dl, long ld_dl, void* dlT, long ld_dlT, float* dbpart,
                                void* stream);
int lctvqa_cvt_f32_to_bf16(const float* x, void* y, long n, void* stream);
int lctvqa_cvt_bf16_to_f32(const void* x, float* y, long n, void* stream);
int lctvqa_maxpool3x3_s2_bf16(const void* x, void* y, int B, int H, int W, int C, void* stream);
int lctvqa_subsample2_bf16(const void* x, void* y, int B, int H, int W, int C, void* stream);
int lctvqa_global_avgpool_bf16(const void* x, float* y, int B, int HW, int C, void* stream);
int lctvqa_conv7x7_s2_stem_bf16(const float* x_nchw, const float* w_oihw, const float* bias, void* y_nhwc_bf16, int B, int H, int W, int relu,
                                void* stream);
/* the same stem on the matrix cores (implicit GEMM, K = 3 x 7 rows x 8, v_mfma_f32_32x32x16_bf16): image and weights rounded to bf16,
 * fp32 accumulation; _supported: even H, W and at most 128 output pixels per row */
int lctvqa_conv7x7_s2_stem_bf16_mfma_supported(int B, int H, int W);
int lctvqa_conv7x7_s2_stem_bf16_mfma(const float* x_nchw, const float* w_oihw, const float* shift, void* y_nhwc_bf16, int B, int H, int W,
                                     int relu, void* stream);

/* One PC-DARTS MixedOp edge as fused tile kernels (model_search.py:22-55 + the seven live primitives of operations.py:4-20
 * on x[..., :C/4], their BatchNorms, the alpha-weighted sum, bypass, channel shuffle and the beta-weighted accumulation
 * into the node's sum, model_search.py:87): csrc/mixed_edge.hip.  w = 14 device pointers: depthwise weights [Cq][K][K] of
 * sep3.a, sep5.a, dil3, dil5, sep3.b, sep5.b; their 1x1 weights [Cq][Cq]; FactorizedReduce conv_1 / conv_2 [Cq/2][Cq] (stride-2
 * edges, else NULL).  bn = 27 pointers: running_mean[9], running_var[9], num_batches_tracked[9] by slot (0 max-pool, 1 avg-pool,
 * 2 skip, 3 sep3.a, 4 sep5.a, 5 dil3, 6 dil5, 7 sep3.b, 8 sep5.b; NULL = none).  pre [9][P][Cq] receives every pre-BatchNorm
 * tensor (saved for the backward), mean / invstd [9][Cq] the statistics used; part = fp64 scratch, grid*7*Cq*2 + 51*Cq doubles
 * (the tail holds the forward kernels' regrouped depthwise weights).
 * wsel [7] = softmax(alpha) of the live primitives, beta [1].  Backward: dx [B,H,W,C] (written, or added to with accumulate_dx = 1:
 * the one gradient buffer of a state that several edges read), dwsel [7], dbeta [1], wgrad = flat weight gradients in the order of w.
 * x1d (both directions, may be NULL): a dense copy [B,H,W,C/4] of x[..., :C/4] (lctvqa_slice_channels), worth making once for a state
 * that several edges read -- the kernels otherwise read a 16..64-byte slice of every NHWC row of x.
 * need_wgrad = 0: dx, dwsel, dbeta only -- the conv stages skip their weight-gradient work and wgrad is not written (the finite-difference
 * passes of the architects, architect_vqa.py:75-88, differentiate w.r.t. alpha / beta only). */
int lctvqa_mixed_edge_supported(int B, int H, int W, int C, int stride);
int lctvqa_mixed_edge_grid(int B, int H, int W, int C, int stride);
int lctvqa_mixed_edge_wgrad_elems(int C);
long lctvqa_mixed_edge_bwd_scratch_floats(int B, int H, int W, int C, int stride);
int lctvqa_mixed_edge_fwd(const float* x, int B, int H, int W, int C, int stride, const void* const* w, void* const* bn,
                          float* pre, float* mean, float* invstd, double* part, const float* wsel, const float* beta,
                          float* out, long ldo, int accumulate, int training, float momentum, float eps, const float* x1d, void* stream);
int lctvqa_mixed_edge_bwd(const float* dout, long ldd, const float* x, int B, int H, int W, int C, int stride, const void* const* w,
                          const float* pre, const float* mean, const float* invstd, const float* wsel, const float* beta, float* dx,
                          float* dwsel, float* dbeta, float* wgrad, float* scratch, int training, int accumulate_dx, const float* x1d,
                          int need_wgrad, void* stream);
/* The same pair with the nine pre-BatchNorm tensors STORED as bf16 (pre = [9][P][Cq] bf16, P*Cq a multiple of 8; everything else as
 * above, arithmetic and statistics fp32): a value is rounded to bf16 before it is stored and before it enters its BatchNorm's batch
 * statistics, so forward normalisation, backward x-hat and the ReLU masks of the sep_conv second stages all see the stored numbers.
 * Selected by darts_ops under the bf16 compute dtype (BASELINE.json configs[4] "bf16"); csrc/mixed_edge_bf16.hip. */
int lctvqa_mixed_edge_fwd_bf16pre(const float* x, int B, int H, int W, int C, int stride, const void* const* w, void* const* bn,
                                  unsigned short* pre, float* mean, float* invstd, double* part, const float* wsel, const float* beta,
                                  float* out, long ldo, int accumulate, int training, float momentum, float eps, const float* x1d,
                                  void* stream);
int lctvqa_mixed_edge_bwd_bf16pre(const float* dout, long ldd, const float* x, int B, int H, int W, int C, int stride,
                                  const void* const* w, const unsigned short* pre, const float* mean, const float* invstd,
                                  const float* wsel, const float* beta, float* dx, float* dwsel, float* dbeta, float* wgrad,
                                  float* scratch, int training, int accumulate_dx, const float* x1d, int need_wgrad, void* stream);

/* ---- bf16-native TRAINABLE convolution trunk (BASELINE.json configs[2] "bf16 ... encoder trainable"; trainable-trunk semantics of
 *      basic_vqa/models_lct.py:56-63): bf16 NHWC activations and activation gradients, fp32 master weights / statistics / weight
 *      gradients, fp32 accumulation on v_mfma_f32_32x32x16_bf16: csrc/train_bf16.hip ----
 * conv_gemm_bf16: C[M][N] bf16 = sum_tap sum_ci A[row + tap][ci] * B[n][tap * Cin + ci] (taps 1: a 1x1 convolution / plain NT
 * product; taps 9: a 3x3 convolution, padding 1, on rows that are the pixels of [.,H,W] images), then out = mask > 0 ? out : 0
 * (bf16 mask or NULL: the ReLU backward), out += addend (bf16 or NULL, may alias C).  stats (or NULL): per-slab column sums
 * (sum, sum of squares) of the fp32 result, [stats_slabs][2][N]: the BatchNorm statistics of the layer's output. */
int lctvqa_conv_gemm_bf16_supported(long M, int N, int Cin, int taps);
int lctvqa_conv_gemm_bf16_stats_slabs(long M, int N, int Cin, int taps);
int lctvqa_conv_gemm_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, long M, int N, int Cin, int taps, int H, int W,
                          const void* addend, long ld_add, const void* mask, long ld_mask, float* stats, void* stream);
/* weight gradient dW (+)= sum_p dz[p][co] * x[p + tap][ci] (bf16 operands [P][ld], both pixel-strided: hardware transpose reads);
 * dW fp32 [Cout][Cin] (taps 1) or OIHW [Cout][Cin][3][3] (taps 9); split over the pixels into fp32 slabs in `workspace` */
long lctvqa_wgrad_bf16_workspace_bytes(long P, int Cout, int Cin, int taps, int H, int W);
int lctvqa_wgrad_bf16(const void* dz, long ldz, const void* x, long ldx, long P, int Cout, int Cin, int taps, int H, int W, float* dW,
                      int accumulate, float* workspace, long workspace_bytes, void* stream);
/* fp32 OIHW [Cout][Cin][3][3] -> bf16 [Cout][tap][Cin] (the B operand of conv_gemm_bf16 with taps 9); dgrad: [Cin][tap][Cout], taps flipped */
int lctvqa_pack3x3_taps_bf16(const float* w_oihw, void* dst, int Cout, int Cin, int dgrad, void* stream);
/* BatchNorm2d on bf16 rows [P][C] (C a power of two in 64..2048).  Forward: partial sums [blocks][2][C] (bn_bf16_stats, or the
 * producer's conv_gemm_bf16 stats) -> optional colreduce_f32 to fp64 (and, under SyncBN, an all-reduce of it) -> fwd_final: coef
 * [4][C] = mean, invstd, scale, shift (+ running statistics) -> apply: y = relu?(z scale + shift (+ res)).  Backward: bwd_sums
 * (sum g, sum g z with g = dy * (ymask > 0); g optionally stored) -> bwd_final: dgamma, dbeta, bcoef [3][C] -> bwd_apply. */
int lctvqa_bn_bf16_blocks(long P, int C);
int lctvqa_bn_bf16_stats(const void* z, long P, int C, float* part, void* stream);
int lctvqa_bn_bf16_bwd_sums(const void* dy, const void* ymask, const void* z, void* g_out, long P, int C, float* part, void* stream);
int lctvqa_colreduce_f32(const float* in, int n, int W2, int chunks, double* out, void* stream);
int lctvqa_bn_bf16_fwd_final(const void* part, int part_f64, int n, int C, double count, float eps, float momentum, const float* gamma,
                             const float* beta, float* run_mean, float* run_var, long* tracked, int training, float* coef, void* stream);
int lctvqa_bn_bf16_bwd_final(const void* loc, int loc_f64, int n, const double* glob, int C, double count, const float* coef, const float* gamma,
                             int training, float* dgamma, float* dbeta, int accumulate, float* bcoef, void* stream);
int lctvqa_bn_bf16_apply(const void* z, const void* res, void* y, const float* coef, long P, int C, int relu, void* stream);
int lctvqa_bn_bf16_bwd_apply(const void* dy, const void* ymask, const void* z, void* dz, const float* bcoef, long P, int C, void* stream);
/* MaxPool2d(3,2,1) recording the winning window position (first maximum in scan order), its backward, the backward of a 2x
 * subsample (zeros at the odd pixels) and of the global average pool (fp32 dy -> bf16 dx) */
int lctvqa_maxpool3x3_s2_bf16_idx(const void* x, void* y, void* idx_u8, int B, int H, int W, int C, void* stream);
int lctvqa_maxpool3x3_s2_bf16_bwd(const void* dy, const void* idx_u8, void* dx, int B, int H, int W, int C, void* stream);
int lctvqa_upsample2_zero_bf16(const void* src, void* dst, int B, int H, int W, int C, void* stream);
int lctvqa_global_avgpool_bf16_bwd(const float* dy, void* dx, int B, int HW, int C, void* stream);

/* pieces of a trainable VGG19 trunk on bf16 activations (basic_vqa/models_lct.py:56-63): MaxPool2d(2,2); the backward of ReLU
 * (-> MaxPool2d(2,2)) from the layer's post-ReLU output (first maximum of a window, like PyTorch); the backward of
 * AdaptiveAvgPool2d((7,7)) + flatten (fp32 dy -> bf16 dx) */
/* the 7x7 / stride-2 stem's input windows as bf16 rows [B*Ho*Wo][192] (k = (c, ky, kx) < 147, zero-padded): its weight gradient is
 * then dz^T . cols on lctvqa_wgrad_bf16 */
int lctvqa_stem_im2col_bf16(const float* x_nchw, void* cols, int B, int H, int W, void* stream);
int lctvqa_maxpool2x2_bf16(const void* x, void* y, int B, int H, int W, int C, void* stream);
int lctvqa_relu_pool_bwd_bf16(const void* d, const void* act, void* dz, int B, int H, int W, int C, int pool, void* stream);
int lctvqa_adaptive_avgpool7_flatten_bwd_bf16(const float* dy, void* dx_nhwc_bf16, int B, int H, int W, int C, void* stream);
/* ---- input pipeline, device side (data_loader_v2.py:53-55,82-86: ToTensor + Normalize done per sample on the host):
 *      uint8 HWC rows of the image store -> fp32 NCHW, y = (x/255 - mean[c]) / std[c] ---------------------------- */
int lctvqa_image_u8_nhwc_to_f32_nchw(const uint8_t* x, float* y, int B, int H, int W, float m0, float m1, float m2,
                                     float s0, float s1, float s2, void* stream);

/* ---- optimiser on a flat parameter arena: clip_grad_norm_(5) + Adam (experiment.py:226-227,
 *      model_factory.py:20-21) and the architects' vector ops (architect_lct.py:112-116) ---- */
int lctvqa_grad_norm(const float* g, long n, float* scratch /* >=1024 floats */, float* total_norm_out, void* stream);
int lctvqa_adam_step(float* p, const float* g, float* m, float* v, long n, const float* total_norm /* or NULL */,
                     float max_norm, float lr, float b1, float b2, float eps, int step, float weight_decay, void* stream);
/* one launch copying a whole list of tensors (the architects' twin models: load_state_dict / _construct_model_from_theta per unrolled
 * model, pcdarts/architect_lct.py:142-155, architect.py:80-95 - 2151 tensors for the PC-DARTS examiner): table in HBM = n_chunks x
 * {src pointer, dst pointer, byte count} (three 8-byte words each), chunks of at most 64 KiB, built once by the caller */
int lctvqa_multi_copy(const void* table, int n_chunks, void* stream);
int lctvqa_axpy(float a, const float* x, float* y, long n, void* stream);
int lctvqa_scale(float a, float* y, long n, void* stream);
/* y = x * a[0], scalar read from device memory (grad_output of a scalar loss; no host sync) */
int lctvqa_scale_by_device_scalar(const float* a, const float* x, float* y, long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LCTVQA_H */
