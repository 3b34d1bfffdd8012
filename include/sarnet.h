/* sarnet.h -- C-ABI of libsarnet_hip.so: the MI355X (gfx950) kernels of the SAR-Net train-step hot path.
 *
 * The reference (pika-online/AESRC2020) has no FFI of its own: every op below replaces a third-party
 * Keras/TF/cuDNN op that the cited reference line *invokes*.  Citations are file:line under /root/reference.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory owned by the caller (no allocation, no
 *     synchronisation, no host<->device copies inside the library); launches go to the caller's `stream`,
 *     so a sequence of calls can be captured into a HIP graph.
 *   - return 0 on success, SAR_EINVAL (-1) for a rejected argument/shape, SAR_ELAUNCH (-2) if HIP refused.
 *   - tensors are fp32, row-major; images NHWC; conv kernels HWIO; Dense (in,out); GRU kernel (in,3H) /
 *     recurrent (H,3H) / bias (6H) in Keras z,r,h order -- i.e. the reference's own weight layouts.
 *   - "accumulate" outputs add into the destination (gradient fan-in); parameter gradients always accumulate.
 */
#ifndef SARNET_H_
#define SARNET_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* sarStream_t; /* == hipStream_t */

#define SAR_OK 0
#define SAR_EINVAL (-1)
#define SAR_ELAUNCH (-2)
enum { SAR_ACT_NONE = 0, SAR_ACT_TANH = 1, SAR_ACT_RELU = 2 };
enum { SAR_HEAD_SPHEREFACE = 0, SAR_HEAD_COSFACE = 1, SAR_HEAD_ARCFACE = 2 };

/* ---- Dense / GEMM: keras Dense at model.py:35-42 (DS), 150, 163; GRU projections model.py:44-50; 1x1
 * assignment conv model.py:89-105.  C[M,N] (+)= act(op(A) op(B) + bias).  a_kc: A is (M,K) row-major else (K,M);
 * b_kc: B is (N,K) row-major else (K,N).  splitk>1 accumulates atomically (act must be NONE). */
int sar_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb,
                 int ldc, int a_kc, int b_kc, int act, int accumulate, int splitk, sarStream_t stream);
/* Backward of Dense w.r.t. its parameters in one launch: dW (in,out) += X^T dY and db (out) += column sums of dY
 * (db nullable); X (R,in) row-major with stride lda, dY (R,out) with stride ldb.  Both accumulate. */
int sar_gemm_wgrad_f32(const float* X, const float* dY, float* dW, float* db, int in_dim, int out_dim, int R, int lda,
                       int ldb, int ldc, int splitk, sarStream_t stream);

/* Up to four problems of equal shape, layouts and flags in ONE launch (grid.z = batch x split-K slices): the input projections of
 * the two directions of Bidirectional(CuDNNGRU) (model.py:44-50) and the weight gradients of its four kernels -- ~1 us of matrix
 * time each, so separate launches only queue behind one another.  Host struct of device pointers; entries >= nbatch are ignored. */
typedef struct sarGemmBatch { const float* A[4]; const float* B[4]; float* C[4]; const float* bias[4]; } sarGemmBatch;
int sar_gemm_batched_f32(const sarGemmBatch* b, int nbatch, int M, int N, int K, int lda, int ldb, int ldc, int a_kc, int b_kc,
                         int act, int accumulate, int splitk, sarStream_t stream);
/* A[i] = X_i (R,in), B[i] = dY_i (R,out), C[i] = dW_i (+=), bias[i] = db_i (+=, nullable): sar_gemm_wgrad_f32, batched */
int sar_gemm_wgrad_batched_f32(const sarGemmBatch* b, int nbatch, int in_dim, int out_dim, int R, int lda, int ldb, int ldc,
                               int splitk, sarStream_t stream);

/* ---- ResNet convolutions: Conv2D at resnet.py:39-42, 60-63, 82-87, 113-117 (TF SAME / VALID padding given as
 * explicit pad_top/pad_left).  scale/shift (nullable) fold the preceding BatchNormalization+ReLU
 * (resnet.py:22-26) into the operand load: P(x)=relu(x*scale+shift).  residual (nullable) is the Add() of
 * resnet.py:89.  Cin, Cout must be multiples of 16. */
int sar_conv2d_fwd_f32(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                       const float* residual, float* y, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH,
                       int KW, int stride, int pad_top, int pad_left, sarStream_t stream);
/* sar_conv2d_fwd_f32 followed by the training-mode BatchNormalization statistics of its OUTPUT for the _bn_relu that
 * consumes it (resnet.py:22-26): identical results to sar_conv2d_fwd_f32 + sar_bn_stats_f32(y, gamma_o, ...).  Without
 * split-K the sums come out of the conv epilogue (no second pass over y).  sums_o = that BatchNorm's workspace. */
int sar_conv2d_fwd_bnstats_f32(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                               const float* residual, float* y, int B, int H, int W, int Cin, int Ho, int Wo, int Cout,
                               int KH, int KW, int stride, int pad_top, int pad_left, const float* gamma_o,
                               const float* beta_o, double* sums_o, float* mean_o, float* rstd_o, float* scale_o,
                               float* shift_o, float* moving_mean_o, float* moving_var_o, float eps, float momentum,
                               int unbiased_moving_var, sarStream_t stream);
/* Weight planes for the split-precision kernels: one launch converts the kernels of many convolutions into bf16 (hi, lo)
 * planes.  `table` = device array of n rows of 8 x int64: {w (fp32 HWIO), wt_hi, wt_lo, wd_hi, wd_lo, K = KH*KW*Cin, Cout,
 * first_block}; wt_* = [Cout][K] (forward), wd_* = [K][Cout] (dgrad); K % 32 == 0, Cout % 64 == 0; first_block = running sum
 * of (K/32)*(Cout/64); total_blocks = that sum over all rows.  Run once per optimiser step (the weights change). */
int sar_conv_weight_split_bf16(const void* table, int n, int total_blocks, sarStream_t stream);
/* sar_conv2d_fwd_f32 with optional companions: wt_hi/wt_lo (nullable) = the kernel's planes from the call above;
 * gamma_o ... (nullable as a group) = the BatchNormalization consuming y, as in sar_conv2d_fwd_bnstats_f32. */
int sar_conv2d_fwd_ex_f32(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                          const float* residual, float* y, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH,
                          int KW, int stride, int pad_top, int pad_left, const void* wt_hi, const void* wt_lo,
                          const float* gamma_o, const float* beta_o, double* sums_o, float* mean_o, float* rstd_o,
                          float* scale_o, float* shift_o, float* moving_mean_o, float* moving_var_o, float eps,
                          float momentum, int unbiased_moving_var, sarStream_t stream);
/* sar_conv2d_dgrad_f32 with the kernel's wd_hi/wd_lo planes (nullable). */
int sar_conv2d_dgrad_ex_f32(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Ho, int Wo,
                            int Cout, int KH, int KW, int stride, int pad_top, int pad_left, int accumulate,
                            const void* wd_hi, const void* wd_lo, sarStream_t stream);
int sar_conv2d_dgrad_f32(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Ho, int Wo,
                         int Cout, int KH, int KW, int stride, int pad_top, int pad_left, int accumulate,
                         sarStream_t stream);
/* dbias (nullable): bias gradient = column sums of dy, accumulated by the same launch */
int sar_conv2d_wgrad_f32(const float* x, const float* dy, const float* scale, const float* shift, float* dw,
                         float* dbias, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_top,
                         int pad_left, sarStream_t stream);
/* Diagnostics: what the last sar_conv2d_wgrad_f32 call launched -- out4 = {pixel slices, XCD-grouped grid 0/1, split-precision
 * kernel 0/1, rows per slice}.  Host-only. */
int sar_dbg_last_wgrad_launch(int* out4);
/* Matrix-core math of the 3x3/1x1 convolutions with >= 64 output (fwd) / input (dgrad) channels:
 * 0 = fp32 MFMA (bit-exact fp32 fma chains); 1 = bf16x3 split precision (default): x = hi + lo in bf16,
 * A*B ~= Ah*Bh + Ah*Bl + Al*Bh on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, ~2^-17 relative error per product.
 * Process-wide host-side setting read at launch time (also: env SAR_CONV_MATH=fp32). */
int sar_set_conv_math(int mode);
int sar_get_conv_math(void);
/* Weight gradients of many Dense / GRU kernels in ONE launch (bf16-MFMA GEMM mode only): problem i is sar_gemm_wgrad_f32(X, dY, dW,
 * db, in_dim, out_dim, R, lda, ldb, ldc, splitk).  sar_gemm_wgrad_group_plan runs on the host: it checks that every problem is
 * eligible for the bf16 path (in_dim % 64 == 0; out_dim, R multiples of 64, or of 4 with out_dim, R >= 64; 16-byte aligned rows) and
 * writes n table rows of 64 bytes + the grid size; SAR_EINVAL = launch them one by one instead.  The caller uploads the table once;
 * sar_gemm_wgrad_group_f32 launches it (the operands only have to be valid when it runs: the host defers the weight gradients of a
 * backward pass and launches the group behind its last input gradient, model.py:340-366 compile/fit has no such ordering). */
typedef struct sarGemmWgrad {
    const float* X; const float* dY; float* dW; float* db;
    int in_dim, out_dim, R, lda, ldb, ldc, splitk, pad_;   /* pad_: 0, or (sar_gemm_wgrad_group_scaled_f32) this problem's dY runs under gscale * 2^-pad_ */
} sarGemmWgrad;
int sar_gemm_wgrad_group_plan(const sarGemmWgrad* probs, int n, void* table_out, int* total_blocks);
int sar_gemm_wgrad_group_f32(const void* table_dev, int n, int total_blocks, sarStream_t stream);
/* The same launch with the BACKWARD arithmetic of the split-precision mode (round 5): gscale > 0 -- one pass on IEEE fp16 copies of the
 * operands, every dY times gscale * 2^-pad_ (powers of two: gscale = the static gradient scale of the 2-byte trunk, 64 x batch; pad_ = the
 * problem's own reduction, for layers whose gradient rows are per utterance, not per frame) while it is staged, dW and db in true units,
 * db summed from the fp32 dY; gscale == 0: sar_gemm_wgrad_group_f32. */
int sar_gemm_wgrad_group_scaled_f32(const void* table_dev, int n, int total_blocks, float gscale, sarStream_t stream);
/* Input gradients of Dense / GRU-projection layers (model.py:35-42 DS, :46-57 BIGRU: the backward TF derives for them), one or two
 * products into one tensor in ONE launch of the 2-byte MFMA GEMM: C (M,N) = [C +] A B + A2 B2.  A, A2 (M, Kseg) row-major (lda); B, B2
 * stored (N, Kseg) when b_kc (a Dense kernel as it lies: dX = dY W^T), else (Kseg, N); A2 = B2 = NULL: one product (the BiGRU's two
 * directions are the two).  scale_a > 0: single-pass IEEE fp16 operands, A times scale_a and B times scale_b (powers of two) while staged,
 * the sums times 1 / (scale_a scale_b); scale_a == 0: the process-wide GEMM arithmetic.  Only whole 64 x 64 x 64 tiles, 16-byte aligned
 * operands, GEMM math bf16 / bf16x3 -- otherwise SAR_EINVAL (the caller runs sar_gemm_f32 per product). */
int sar_gemm_pair_f32(const float* A, const float* B, const float* A2, const float* B2, float* C, int M, int N, int Kseg, int lda,
                      int ldb, int ldc, int b_kc, int accumulate, float scale_a, float scale_b, sarStream_t stream);
/* GEMM arithmetic of sar_gemm_f32 / sar_gemm_wgrad_f32: 0 = fp32 MFMA (default), 1 = operands rounded to bf16 while they are
 * staged, v_mfma_f32_32x32x16_bf16, fp32 accumulate (problems whose M, N, K are multiples of 64; others keep fp32).  The
 * "bf16" mode of BASELINE configs[1]; fp32 tensors at the ABI either way.  Host-side, process-wide (env SAR_GEMM_MATH=bf16). */
int sar_set_gemm_math(int mode);
int sar_get_gemm_math(void);
/* 7x7/2 stem on the 1-channel fbank image: resnet.py:173 / 191.  C0 <= 64, C0 % 4 == 0. */
int sar_stem_conv7x7_fwd_f32(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Ho,
                             int Wo, int C0, int pad_top, int pad_left, sarStream_t stream);
/* sar_stem_conv7x7_fwd_f32 + sar_bn_stats_f32 of its output in one call: the per-channel sums come out of the conv epilogue
 * (sums = the BatchNorm workspace), so the 131 MB stem map (B=64) is not read a second time for its statistics. */
int sar_stem_conv7x7_fwd_bnstats_f32(const float* x, const float* w, const float* bias, float* y, int B, int H, int W,
                                     int Ho, int Wo, int C0, int pad_top, int pad_left, const float* gamma,
                                     const float* beta, double* sums, float* mean, float* rstd, float* scale, float* shift,
                                     float* moving_mean, float* moving_var, float eps, float momentum,
                                     int unbiased_moving_var, sarStream_t stream);
int sar_stem_conv7x7_wgrad_f32(const float* x, const float* dy, float* dw, float* db, int B, int H, int W, int Ho,
                               int Wo, int C0, int pad_top, int pad_left, sarStream_t stream);
/* BN+ReLU (resnet.py:43) fused with MaxPooling2D 3x3/2 'same' (resnet.py:174,192). idx = argmax window slot. */
int sar_bn_relu_maxpool_fwd_f32(const float* x, const float* scale, const float* shift, float* out, uint8_t* idx,
                                int B, int H, int W, int C, int Ho, int Wo, int pad_top, int pad_left,
                                sarStream_t stream);
int sar_maxpool_bn_relu_bwd_f32(const float* dpool, const float* out, const uint8_t* idx, const float* x,
                                const float* mean, const float* rstd, const float* scale, const float* shift, double* sums,
                                float* dx, int B, int H, int W, int C, int Ho, int Wo, int pad_top, int pad_left,
                                sarStream_t stream);
/* Stem weight gradient without the full-resolution gradient tensor: dw += patches(x)^T * d/dy[maxpool(relu(bn(y)))], the
 * pooling/BatchNorm backward evaluated per tile inside the kernel (resnet.py:173-174; nothing but this weight gradient reads
 * that tensor: the stem is the first layer).  Call sar_maxpool_bn_relu_bwd_f32 with dx = NULL first: it leaves the two
 * BatchNorm sums in `sums`. */
int sar_stem_conv7x7_wgrad_poolbwd_f32(const float* x, const float* y, const float* dpool, const float* out,
                                       const uint8_t* idx, const float* mean, const float* rstd, const float* scale,
                                       const double* sums, float* dw, float* db, int B, int H, int W, int Ho, int Wo, int C0,
                                       int pad_top, int pad_left, int Hp, int Wp, int pool_pad_top, int pool_pad_left,
                                       sarStream_t stream);

/* ---- bf16 trunk (BASELINE configs[1] "bf16"): the same ResNet layers (resnet.py:22-125) on 2-byte NHWC activations and
 * activation gradients; fp32 master weights, BatchNorm statistics and MFMA accumulation.  `void*` tensors are bf16.
 * BatchNorm statistics travel through per-layer bins of fp32 atomics: NB = sar_bn_bins() rows of [sum | sum of squares]
 * (forward) or [sum dz | sum dz*xhat] (backward), zeroed by the caller once per step; every consumer sums the rows itself,
 * so there are no finalize / combine / apply launches. */
typedef struct sarBnRef {      /* one BatchNormalization (resnet.py:25) as a consumer kernel sees it; host struct of device pointers */
    const float* gamma; const float* beta;
    const float* bins;         /* forward sums [NB][2][C] of the BN input, or NULL: take mean/rstd/scale/shift from `saved` */
    const float* bbins;        /* backward sums [NB][2][C] */
    float* saved;              /* [4][C]: mean, rstd, scale = gamma*rstd, shift = beta - mean*scale */
    float* moving_mean; float* moving_var;   /* updated (momentum, Bessel-corrected variance) when `write` and bins != NULL */
    float* dgamma; float* dbeta;             /* += when `write` in a backward consumer (nullable) */
    float inv_n, unbias, eps, momentum;      /* 1/rows, rows/(rows-1) */
    int write;                 /* this launch owns the side effects (block 0 performs them) */
    float ginv;                /* dgamma / dbeta += sums * ginv: 1 / (the gradient scale the sums in bbins carry); 1 in the bf16 trunk */
    int det;                   /* 1: `bins` holds int64 [NB][2][C] in units of 2^-28 (deterministic statistics: producers add integers, so the
                                * sums do not depend on block order or on how the batch is tiled; sar_conv2d_* epi 3, sar_stem16_stats det) */
    int pad_;
} sarBnRef;
int sar_bn_bins(void);
/* Kernels of many convolutions (fp32 HWIO) -> the bf16 fragment streams sar_conv2d_bf16 reads, one launch.  table = device
 * array of n rows of 8 x int64: {w, wf (forward stream), wd (input-gradient stream: transposed, taps flipped), KH*KW, Cin,
 * Cout, first_block, 0}; stream sizes: wf = (Cout/32)*(KH*KW*Cin/16 + 8)*1024 bytes, wd = (Cin/32)*(KH*KW*Cout/16 + 8)*1024;
 * blocks per row = ceil((wf + wd bytes)/16/256), first_block = running sum.  Once per optimiser step. */
int sar_conv_pack_weights_bf16(const void* table, int n, int total_blocks, sarStream_t stream);
/* sar_zero_ranges(zero_table, nz) and sar_conv_pack_weights_bf16(table, n, total_blocks) in one launch (round 3): the two launches a
 * train step of the 2-byte trunk begins with. */
int sar_conv_pack_weights_zero_bf16(const void* table, int n, int total_blocks, const void* zero_table, int nz, sarStream_t stream);
/* Generic bf16 convolution, forward (resnet.py:39-42, 60-63, 82-87, 113-117) and input gradient in one kernel:
 *   y[B*Ho*Wo][Cout] = sum_taps T(src)[padded position] * wpk (+ bias) (+ res)
 * fill 0: T = identity; 1: T = relu(bn(.)) of fbn (statistics from fbn->bins, or fbn->saved when bins is NULL: inference /
 * recompute); 2: T = BatchNorm+ReLU backward of (da = src, y = src2) through fbn (sums from fbn->bbins); 3: the same + src3 --
 * the gradient a residual block hands to the block in front of it, bn1_bwd(da1, x) + shortcut gradient (resnet.py:96-125), formed
 * in the fill of that block's conv2 input-gradient launch instead of by a BatchNorm-apply launch of its own.
 * side (nullable, shape of src): T(src) written back to HBM -- the plain operand of the matching weight gradient.
 * epi 0: none; 1: sum and sum of squares of y into e_bins (the statistics of the BatchNorm that consumes y); 2: the two
 * BatchNorm-backward sums of y (= da) against (e_y = that BN's input, e_saved) into e_bins; 3: as 1, deterministic: e_bins is int64
 * [NB][2][C], every stored value and its square converted to 2^-28 fixed point and added as integers (sarBnRef.det consumers).
 * Forward: d = 1, s = stride, pt/pl = TF pads, wpk = wf.  Input gradient: src = dy, Hs/Ws = its size, d = stride, s = 1,
 * Ho/Wo = the forward INPUT size, pt = K-1-pad_top, pl = K-1-pad_left, wpk = wd.  K in {1,3}; Cin, Cout powers of two in
 * [32, 512]. */
int sar_conv2d_bf16(const void* src, const void* src2, const void* src3, const void* wpk, const float* bias, const void* res, void* y, void* side,
                    int B, int Hs, int Ws, int Cin, int Ho, int Wo, int Cout, int K, int s, int d, int pt, int pl, int fill,
                    const sarBnRef* fbn, int epi, float* e_bins, const void* e_y, const float* e_saved, sarStream_t stream);
/* TWO independent convolutions in ONE launch (round 3): a = a 3x3 convolution, b = a 1x1 convolution of the same channel classes -- the first
 * conv of a shape-changing residual block and its shortcut (resnet.py:82-87, 113-117: same input, different outputs), or their input
 * gradients.  sarConvArgs holds the arguments of one sar_conv2d_bf16 call.  Returns SAR_EINVAL and launches NOTHING when the two do not run
 * on instantiations that pair up (the caller then issues them one after the other). */
typedef struct sarConvArgs {
    const void *src, *src2, *src3, *wpk; const float* bias; const void* res; void* y; void* side;
    int B, Hs, Ws, Cin, Ho, Wo, Cout, K, s, d, pt, pl, fill; const sarBnRef* fbn; int epi; float* e_bins; const void* e_y; const float* e_saved;
    void* side2; float side_scale;   /* sar_conv2d_x3 only (below); NULL / 0 elsewhere */
} sarConvArgs;
int sar_conv2d_pair_bf16(const sarConvArgs* a, const sarConvArgs* b, sarStream_t stream);
/* Plan workspace of sar_conv2d_bf16 / sar_conv2d_pair_bf16 (round 6; resnet.py:22-125 -- every conv of the trunk is launched once per step with
 * the same shape).  What a block derives from the launch's SHAPE alone -- the fill's per-thread piece, the tile's row table, the patch position of
 * every output pixel: ~2 000 cycles of integer divisions in front of the first load of the fill, by every wave of every block of every launch --
 * is written ONCE per shape into `buf` (by a small kernel on the stream of the launch that first meets the shape, in front of it) and read back
 * with three loads at kernel entry.  The library does not allocate: the caller lends `buf` (device memory, 256-byte aligned, `bytes` long; ~100 KB
 * per conv shape at B = 64, a few MB per model) until the next call of this function or the end of the process -- captured graphs keep pointers
 * into it.  Without a workspace, or once it is full, launches run the instantiations that derive everything themselves (results identical).
 * The tables of a shape are complete when the launch that first met the shape has run: launches of that shape on OTHER streams must be ordered behind
 * it (the package runs every conv of a model on one stream).  buf = NULL, bytes = 0 detaches.  One per build of the trunk kernels (…_f16: sar_conv2d_f16 / _f16a; …_x3: sar_conv2d_x3).  Not thread-safe. */
int sar_conv_plan_workspace_bf16(void* buf, long bytes);
int sar_conv_plan_workspace_f16(void* buf, long bytes);
int sar_conv_plan_workspace_x3(void* buf, long bytes);
/* Introspection (bench.py, tools): the template arguments <WN, NKC, VT, KG, WMX, BR> of the conv_bf16_kernel instantiation the most
 * recent sar_conv2d_bf16 launch of this process ran -- the name rocprofv3 reports it under.  Not thread-safe. */
int sar_conv2d_last_class_bf16(int* out6);
/* Kernel gradients (resnet.py:39-42, 60-63, 82-87, 113-117 under Keras' backward pass) of MANY convolutions in one launch, on
 * PLAIN bf16 operands: x = the `side` tensor of the forward conv (or the raw input of a shortcut / first conv), dy = the
 * gradient tensor (for a conv that feeds a BatchNorm: the `side` tensor of its input-gradient launch).
 *   dw[K][K][Cin][Cout] += x^T (*) dy per tap;  dbias (nullable) += column sums of dy.
 * They are off the critical path of the backward pass, so the trunk collects them and runs them once per backward part.
 * (1) sar_wgrad_plan_bf16 -- host only, no device access -- turns n descriptors of equal K (1 or 3) into the two device tables of
 * one launch: convs_out (n rows of 128 bytes) and items_out (<= max_items rows of 32 bytes: one block each = one convolution, a
 * tile of 32*tile_ci input x 32*tile_co output channels, a run of pixel chunks -- 128 pixels for 3x3, 64 for 1x1 (round 6; 64 for
 * both until then: a chunk is one barrier, one round of DMA issue and one position table of the block) --; longest first; runs are cut
 * so that the launch has about target_blocks items), the item count, and launch4 = {dynamic LDS bytes, positions of the patch ring,
 * chunk buffers, tile_ci << 4 | tile_co | (128-pixel chunks ? 0x100 : 0)}.  tile_ci, tile_co in {1, 2} (3x3; 1x1: both 1): 64-wide tiles halve the bytes per MFMA where the
 * layer has 64+ channels.  Plan stride-2 convolutions (4x larger patch ring) and the tile classes separately.
 * (2) The caller uploads both tables once; sar_conv2d_wgrad_batch_bf16 launches them. */
typedef struct sarWgradConv {
    const void* x; const void* dy; float* dw; float* dbias;
    int B, H, W, Cin, Ho, Wo, Cout, K, s, pt, pl;          /* geometry as the forward conv (s = stride, pt/pl = TF pads) */
    float ginv;                /* dw, dbias += result * ginv: 1 / (the gradient scale dy carries); 0 is read as 1 */
} sarWgradConv;
/* Flush (round 3): an item that shares its gradient tile with other items stores its raw partial tile into its own slab of a caller-owned
 * workspace (plain 16-byte stores) and a reduction launched by the same call sums a tile's slabs in the plan's fixed order and adds the
 * result into dw -- no atomics on dw, a deterministic sum.  The plan therefore also returns tiles_out (rows of 32 bytes, at most one per
 * (convolution, channel tile)), slices_out (ints, at most one per item) and launch4[4..6] = {reduction tiles, slabs, floats per slab};
 * launch4 has EIGHT ints.  The caller uploads both tables and allocates slab = launch4[5] * launch4[6] floats. */
int sar_wgrad_plan_bf16(const sarWgradConv* convs, int n, int target_blocks, int tile_ci, int tile_co, void* convs_out,
                        void* items_out, int max_items, int* n_items, int* launch4, void* tiles_out, int* slices_out);
int sar_conv2d_wgrad_batch_bf16(const void* convs_dev, const void* items_dev, int n_items, int K, const int* launch4, const void* tiles_dev,
                                const int* slices_dev, float* slab, sarStream_t stream);
/* The reductions of up to four batched launches in ONE launch (round 3): run the batches with launch4[7] = 1 (they then leave their partial
 * tiles in their slabs) and describe each here by its tables, n_tiles = launch4[4], K and the tile class it was planned with.  K + 256 (round 5):
 * the sums are WRITTEN (dw = ...) instead of added -- every gradient tile has exactly one reduction block; the train step, whose dw is zero at that
 * point, saves the read half of the read-modify-write. */
typedef struct sarWgradReduce { const void* convs_dev; const void* tiles_dev; const int* slices_dev; const float* slab; int n_tiles, K, tile_ci, tile_co; } sarWgradReduce;
int sar_wgrad_reduce_multi_bf16(const sarWgradReduce* plans, int n, sarStream_t stream);
/* out (bf16) = BatchNorm(+ReLU) backward of dact (bf16, or fp32 when dact_f32) through bn (input x, sums bn->bbins) (+ add) */
int sar_bn_bwd_apply_bf16(const void* dact, int dact_f32, const void* x, const sarBnRef* bn, const void* add, void* out, long R,
                          int C, int relu, float out_scale, sarStream_t stream);   /* out_scale multiplies the result (1 in the bf16 trunk) */
/* bbins += the two BatchNorm(+ReLU) backward sums for a gradient that does not come out of a conv epilogue (the final BN) */
int sar_bn_bwd_reduce_bf16(const void* dact, int dact_f32, const void* x, const float* saved, float* bbins, long R, int C, int relu,
                           sarStream_t stream);
/* out (fp32) = [relu](bn(x)) for bf16 x: the trunk's final BN+ReLU (resnet.py:178-181) feeding the fp32 sequence layers */
int sar_bn_apply_bf16_f32(const void* x, const sarBnRef* bn, float* out, long R, int C, int relu, sarStream_t stream);
int sar_cast_f32_bf16(const float* x, void* y, long n, sarStream_t stream);   /* n % 8 == 0 */
int sar_cast_bf16_f32(const void* x, float* y, long n, sarStream_t stream);
/* The stem of the bf16 trunk -- Conv2D 7x7/2 'same' (Cin = 1) -> BatchNormalization -> ReLU -> MaxPooling2D 3x3/2 'same',
 * resnet.py:173-174 / 191-192 (_conv_bn_relu, resnet.py:34-45) -- without the full-resolution map (131 MB in fp32 at B=64): the
 * conv is recomputed on bf16 MFMA wherever its output is needed.  x (B,T,D) fp32 (the (B,T,80,1) input), w (7,7,1,C0) fp32 HWIO,
 * C0 in {32, 64}; Ho/Wo, pad_top/pad_left = the conv's TF-SAME geometry, Hp/Wp, pool_pad_* = the pool's.  pooled / g_pooled
 * (B,Hp,Wp,C0) bf16; idx (same shape, u8) = window slot kh*3+kw of the first maximum.
 *   forward : sar_stem16_stats (training: sum / sum of squares of conv+bias into the BatchNorm's bins) -> sar_stem16_pool_fwd (bn
 *             like a consumer of the bf16 trunk: statistics from bn->bins, or bn->saved when bins is NULL = inference).
 *   backward: sar_stem16_bwd_reduce (the two BatchNorm-backward sums into bn->bbins, from g_pooled / pooled / idx alone; it also
 *             resets the tile counter at the head of `work`, the buffer sar_stem16_wgrad gets) ->
 *             sar_stem16_wgrad (dw += kernel gradient; owns dgamma / dbeta when bn->write; the bias gradient is exactly zero and
 *             not computed; work = sar_stem16_wgrad_workspace_bytes(C0) bytes of scratch).  No input gradient: first layer.
 *             Round 6: the BatchNorm-backward gradient at the conv is dy = A (.) dz + E (.) z + F (dz = the pooled gradient scattered to
 *             its maxima, z = P w the conv output, P = the im2col image, A / E / F per channel from the BatchNorm's sums), so
 *             dW = P^T dy = A (.) (P^T dz) + E (.) ((P^T P) w) + (P^T 1) (x) F: one pass builds P^T dz and the 64 x 64 Gram matrix P^T P per
 *             tile on MFMA WITHOUT recomputing the conv (tap 7 of the 8 x 8 tap grid is a zero kernel row and carries 1.0, so P^T 1 is a
 *             Gram column); a finisher sums the per-block parts and applies A, E, F.  work holds the per-block parts:
 *             64 + 512 * (64*C0 + 64*64) * 4 bytes.  SAR_STEM_WG2=0 selects the round-5 recompute form. */
int sar_stem16_stats(const float* x, const float* w, const float* bias, float* bins, int B, int T, int D, int Ho, int Wo, int C0,
                     int pad_top, int pad_left, int det, sarStream_t stream);   /* det: bins is int64 fixed point (sarBnRef.det) */
int sar_stem16_pool_fwd(const float* x, const float* w, const float* bias, const sarBnRef* bn, void* pooled, uint8_t* idx, int B, int T, int D,
                        int Ho, int Wo, int C0, int pad_top, int pad_left, int Hp, int Wp, int pool_pad_top, int pool_pad_left, sarStream_t stream);
int sar_stem16_bwd_reduce(const void* g_pooled, const void* pooled, const uint8_t* idx, const float* x, const float* w, const float* bias,
                          const sarBnRef* bn, float* work, int B, int T, int D, int Ho, int Wo, int C0, int pad_top, int pad_left, int Hp, int Wp,
                          int pool_pad_top, int pool_pad_left, sarStream_t stream);
size_t sar_stem16_wgrad_workspace_bytes(int C0);
int sar_stem16_wgrad(const float* x, const float* w, const float* bias, const void* g_pooled, const void* pooled, const uint8_t* idx,
                     const sarBnRef* bn, float* dw, float* work, int B, int T, int D, int Ho, int Wo, int C0, int pad_top, int pad_left,
                     int Hp, int Wp, int pool_pad_top, int pool_pad_left, sarStream_t stream);

/* ---- fp16 trunk (BASELINE configs[4] "fp16"; also the better-conditioned 2-byte mode for configs[1]): the SAME kernels compiled
 * with IEEE fp16 as the 2-byte type (csrc/trunk16.h: SAR_HALF) -- v_mfma_f32_32x32x16_f16, same bytes and matrix rate as bf16,
 * 2^-11 instead of 2^-8 relative rounding per stored value.  Every `void*` tensor of the functions below is fp16; arguments and
 * semantics are those of the _bf16 function of the same name (sar_wgrad_plan_bf16, sar_bn_bins and
 * sar_stem16_wgrad_workspace_bytes are host-side and shared).  fp16 has 5 exponent bits, so activation GRADIENTS travel through
 * the trunk multiplied by a static scale S: the caller passes out_scale = S where the gradient enters (sar_bn_bwd_apply_f16 at the
 * final BatchNorm, resnet.py:178) and ginv = 1/S wherever parameter gradients leave (sarBnRef.ginv, sarWgradConv.ginv); everything
 * in between is linear in the gradient. */
int sar_conv_pack_weights_f16(const void* table, int n, int total_blocks, sarStream_t stream);
int sar_conv_pack_weights_zero_f16(const void* table, int n, int total_blocks, const void* zero_table, int nz, sarStream_t stream);
int sar_conv2d_f16(const void* src, const void* src2, const void* src3, const void* wpk, const float* bias, const void* res, void* y, void* side,
                   int B, int Hs, int Ws, int Cin, int Ho, int Wo, int Cout, int K, int s, int d, int pt, int pl, int fill,
                   const sarBnRef* fbn, int epi, float* e_bins, const void* e_y, const float* e_saved, sarStream_t stream);
int sar_conv2d_last_class_f16(int* out6);
int sar_conv2d_pair_f16(const sarConvArgs* a, const sarConvArgs* b, sarStream_t stream);
/* The MIXED BACKWARD of the split-precision mode (round 5; the reference trains in fp32, model.py:197 -- what replaces its Conv2D input gradients,
 * resnet.py:39-42,60-63,82-87,113-117, and the BatchNormalization backward in front of them, resnet.py:22-26): sar_conv2d_f16 / sar_conv2d_pair_f16
 * with the two tensors of the FORWARD pass a gradient launch reads -- src2 (the BatchNorm input of the fill's backward transform, fill 2 / 3) and e_y
 * (the BatchNorm input of the epilogue's backward sums, epi 2) -- as the FP32 tensors sar_conv2d_x3 left in HBM; src, src3, res, y, side are fp16
 * gradients under the static scale, wpk the fp16 input-gradient stream (sar_conv_pack_weights_x3 with bit 0 of the table row's last word set).
 * Forward outputs keep the split-precision arithmetic (north_star's 1e-3); tools/grad_recipe_emulator.py: the flat parameter gradient moves from
 * 6.8e-4 (split-precision input gradients, fp16 weight-gradient operands) to 9.6e-4 of the exact one. */
int sar_conv2d_f16a(const void* src, const void* src2, const void* src3, const void* wpk, const float* bias, const void* res, void* y, void* side,
                    int B, int Hs, int Ws, int Cin, int Ho, int Wo, int Cout, int K, int s, int d, int pt, int pl, int fill,
                    const sarBnRef* fbn, int epi, float* e_bins, const void* e_y, const float* e_saved, sarStream_t stream);
int sar_conv2d_pair_f16a(const sarConvArgs* a, const sarConvArgs* b, sarStream_t stream);
int sar_wgrad_reduce_multi_f16(const sarWgradReduce* plans, int n, sarStream_t stream);
int sar_conv2d_wgrad_batch_f16(const void* convs_dev, const void* items_dev, int n_items, int K, const int* launch4, const void* tiles_dev,
                               const int* slices_dev, float* slab, sarStream_t stream);
int sar_bn_bwd_apply_f16(const void* dact, int dact_f32, const void* x, const sarBnRef* bn, const void* add, void* out, long R,
                         int C, int relu, float out_scale, sarStream_t stream);
int sar_bn_bwd_reduce_f16(const void* dact, int dact_f32, const void* x, const float* saved, float* bbins, long R, int C, int relu,
                          sarStream_t stream);
int sar_bn_apply_f16_f32(const void* x, const sarBnRef* bn, float* out, long R, int C, int relu, sarStream_t stream);
int sar_cast_f32_f16(const float* x, void* y, long n, sarStream_t stream);
int sar_cast_f16_f32(const void* x, float* y, long n, sarStream_t stream);
int sar_stem16_stats_f16(const float* x, const float* w, const float* bias, float* bins, int B, int T, int D, int Ho, int Wo, int C0,
                         int pad_top, int pad_left, int det, sarStream_t stream);
int sar_stem16_pool_fwd_f16(const float* x, const float* w, const float* bias, const sarBnRef* bn, void* pooled, uint8_t* idx, int B, int T, int D,
                            int Ho, int Wo, int C0, int pad_top, int pad_left, int Hp, int Wp, int pool_pad_top, int pool_pad_left, sarStream_t stream);
int sar_stem16_bwd_reduce_f16(const void* g_pooled, const void* pooled, const uint8_t* idx, const float* x, const float* w, const float* bias,
                              const sarBnRef* bn, float* work, int B, int T, int D, int Ho, int Wo, int C0, int pad_top, int pad_left, int Hp, int Wp,
                              int pool_pad_top, int pool_pad_left, sarStream_t stream);
int sar_stem16_wgrad_f16(const float* x, const float* w, const float* bias, const void* g_pooled, const void* pooled, const uint8_t* idx,
                         const sarBnRef* bn, float* dw, float* work, int B, int T, int D, int Ho, int Wo, int C0, int pad_top, int pad_left,
                         int Hp, int Wp, int pool_pad_top, int pool_pad_left, sarStream_t stream);

/* ---- split-precision trunk (round 4; the fp32-class mode: north_star's "logits within 1e-3" of the reference's fp32 graph,
 * model.py:197, resnet.py:113-122): the SAME kernels compiled with -DSAR_X3 (csrc/trunk16.h).  Every `void*` activation / gradient
 * tensor of the functions below is an FP32 tensor (r4 v9; before: a pair of bf16 planes, the same bytes); the conv fill splits every
 * transformed operand into hi = bf16(x) and lo = bf16(x - hi) for the LDS planes the MFMAs read, a product block is three MFMAs
 * (lo*hi + hi*lo + hi*hi), weights are packed as interleaved hi / lo fragments.  Arguments and semantics are those of the _bf16 function
 * of the same name, except:
 *   sar_conv2d_x3: `side` is ONE fp16 tensor = fp16(T(src) * side_scale) -- the operand of the weight gradient, which this mode runs
 *     through sar_conv2d_wgrad_batch_f16 (gradient operands carry the fp16 build's static scale, sarWgradConv.ginv undoes it) -- and
 *     side2 (nullable) receives T(src) as fp32 (the materialised block gradient of fill 3, which further convs read);
 *   sar_bn_bwd_apply_x3: `out` (nullable since round 5) is fp32 in TRUE units, out16 (nullable) = fp16(result * out_scale);
 *   sar_conv_pack_weights_x3: bit 0 of a table row's last word = the input-gradient stream of that convolution is a single plane of fp16
 *     fragments (what sar_conv2d_f16a reads) instead of hi / lo bf16 pairs;
 *   sar_cast_f32_x3: y = x (a copy), y16 (nullable) = fp16(x * scale). */
int sar_conv_pack_weights_x3(const void* table, int n, int total_blocks, sarStream_t stream);
int sar_conv_pack_weights_zero_x3(const void* table, int n, int total_blocks, const void* zero_table, int nz, sarStream_t stream);
int sar_conv2d_x3(const void* src, const void* src2, const void* src3, const void* wpk, const float* bias, const void* res, void* y, void* side,
                  int B, int Hs, int Ws, int Cin, int Ho, int Wo, int Cout, int K, int s, int d, int pt, int pl, int fill,
                  const sarBnRef* fbn, int epi, float* e_bins, const void* e_y, const float* e_saved, void* side2, float side_scale, sarStream_t stream);
int sar_conv2d_last_class_x3(int* out6);
int sar_conv2d_pair_x3(const sarConvArgs* a, const sarConvArgs* b, sarStream_t stream);
int sar_bn_bwd_apply_x3(const void* dact, int dact_f32, const void* x, const sarBnRef* bn, const void* add, void* out, void* out16, long R,
                        int C, int relu, float out_scale, sarStream_t stream);
int sar_bn_bwd_reduce_x3(const void* dact, int dact_f32, const void* x, const float* saved, float* bbins, long R, int C, int relu,
                         sarStream_t stream);
int sar_bn_apply_x3_f32(const void* x, const sarBnRef* bn, float* out, long R, int C, int relu, sarStream_t stream);
int sar_cast_f32_x3(const float* x, void* y, void* y16, float scale, long n, sarStream_t stream);
int sar_cast_x3_f32(const void* x, float* y, long n, sarStream_t stream);
/* The recompute-form stem in split precision: the 7x7 conv is evaluated with hi / lo planes of the input patch and of the kernel fragments
 * (three MFMAs per block) wherever its output is needed.  pooled: fp32; pooled_h (nullable): the same map as ONE fp16 tensor, the
 * operand of the first residual block's weight gradients.  sar_stem16_bwd_reduce_x3 / sar_stem16_wgrad_x3 (round 5): g_pooled is an FP16 tensor
 * times the static scale 1 / bn->ginv -- what the mixed backward's last input-gradient launches (sar_conv2d_f16a) write -- exactly as in the fp16
 * build; the final product dW = im2col^T dy runs single-pass on fp16 operands; bn->ginv undoes the scale in dgamma / dbeta and dW. */
int sar_stem16_stats_x3(const float* x, const float* w, const float* bias, float* bins, int B, int T, int D, int Ho, int Wo, int C0,
                        int pad_top, int pad_left, int det, sarStream_t stream);
int sar_stem16_pool_fwd_x3(const float* x, const float* w, const float* bias, const sarBnRef* bn, void* pooled, void* pooled_h, uint8_t* idx, int B, int T, int D,
                           int Ho, int Wo, int C0, int pad_top, int pad_left, int Hp, int Wp, int pool_pad_top, int pool_pad_left, sarStream_t stream);
int sar_stem16_bwd_reduce_x3(const void* g_pooled, const void* pooled, const uint8_t* idx, const float* x, const float* w, const float* bias,
                             const sarBnRef* bn, float* work, int B, int T, int D, int Ho, int Wo, int C0, int pad_top, int pad_left, int Hp, int Wp,
                             int pool_pad_top, int pool_pad_left, sarStream_t stream);
/* sar_stem16_bwd_reduce_x3 reading the FP16 copy of the pooled map (pooled_h of sar_stem16_pool_fwd_x3) instead of the fp32 map: the sums are
 * gradient-class (round 5: the mixed backward), the fp32 map was 33 of the pass's 47 MB. */
int sar_stem16_bwd_reduce_x3h(const void* g_pooled, const void* pooled_h, const uint8_t* idx, const float* x, const float* w, const float* bias,
                              const sarBnRef* bn, float* work, int B, int T, int D, int Ho, int Wo, int C0, int pad_top, int pad_left, int Hp, int Wp,
                              int pool_pad_top, int pool_pad_left, sarStream_t stream);
int sar_stem16_wgrad_x3(const float* x, const float* w, const float* bias, const void* g_pooled, const void* pooled, const uint8_t* idx,
                        const sarBnRef* bn, float* dw, float* work, int B, int T, int D, int Ho, int Wo, int C0, int pad_top, int pad_left,
                        int Hp, int Wp, int pool_pad_top, int pool_pad_left, sarStream_t stream);

/* ---- BatchNormalization, training mode, eps 1e-3, momentum .99: resnet.py:25; model.py:29-30,286,289.
 * x is viewed (R, C).  sums = workspace of sar_bn_workspace_bytes(C): fp64 totals[2C] (sum dz, sum dz*xhat after a
 * backward) followed by per-block fp32 partials; two-stage reduction, no atomics, no zeroing required. */
size_t sar_bn_workspace_bytes(int C);
int sar_bn_stats_f32(const float* x, const float* gamma, const float* beta, double* sums, float* mean, float* rstd,
                     float* scale, float* shift, float* moving_mean, float* moving_var, long R, int C, float eps,
                     float momentum, int unbiased_moving_var, sarStream_t stream);
/* Second half of sar_bn_stats_f32 for a producer that wrote `rows` partial rows [2C] (sum | sum of squares) itself. */
int sar_bn_finalize_f32(const float* part, int rows, const float* gamma, const float* beta, float* mean, float* rstd,
                        float* scale, float* shift, float* moving_mean, float* moving_var, long R, int C, float eps,
                        float momentum, int unbiased_moving_var, sarStream_t stream);
int sar_bn_apply_f32(const float* x, const float* scale, const float* shift, float* y, long R, int C, int relu,
                     sarStream_t stream);
/* Inference mode (Keras learning phase 0, used by predict/evaluate: train.py:42 validation_data, model.py:386):
 * y = gamma*(x-moving_mean)/sqrt(moving_var+eps)+beta.  _fold produces the per-channel scale/shift the conv prologues
 * consume; _infer applies it to an (R, C) tensor of any C. */
int sar_bn_fold_f32(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                    float* scale, float* shift, int C, float eps, sarStream_t stream);
int sar_bn_infer_f32(const float* x, const float* gamma, const float* beta, const float* moving_mean,
                     const float* moving_var, float* y, long R, int C, float eps, int relu, sarStream_t stream);
int sar_bn_bwd_f32(const float* dact, const float* x, const float* scale, const float* shift, const float* mean,
                   const float* rstd, double* sums, const float* add, float* dx, float* dgamma, float* dbeta, long R,
                   int C, int relu, sarStream_t stream);
int sar_bn_param_grads_f32(const double* sums, float* dgamma, float* dbeta, int C, sarStream_t stream);
int sar_bn1d_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                     float* moving_mean, float* moving_var, int B, int C, float eps, float momentum,
                     sarStream_t stream);
int sar_bn1d_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                     float* dx, float* dgamma, float* dbeta, int B, int C, sarStream_t stream);

/* ---- LayerNormalization (keras_layer_normalization, eps = K.epsilon()^2): model.py:4,32-33 */
/* DS(hidden_dim, 'tanh') -> LN in ONE launch (round 6; model.py:253-254 CNN_LIN / CNN_LIN_LN, 265-266 CTC_DS / CTC_DS_LN, 279-280 AR_DS / AR_DS_LN;
 * DS = model.py:35-42, LN = model.py:32-33): act_out (R,256) = tanh(x W + bias), y (R,256) = LayerNormalization(act_out) with gamma / beta / eps,
 * mean / rstd (R) as sar_layernorm_fwd_f32 writes them -- both tensors and both vectors are what the backward launches read.  x (R,K) and W (K,256)
 * row-major and 16-byte aligned, N = 256, K % 64 == 0, GEMM arithmetic bf16x3 (split precision); anything else: SAR_EINVAL, and the caller runs
 * sar_gemm_f32 + sar_layernorm_fwd_f32. */
int sar_dense_tanh_ln_f32(const float* x, const float* W, const float* bias, const float* gamma, const float* beta, float* act_out, float* y,
                          float* mean, float* rstd, long R, int K, int N, float eps, sarStream_t stream);
int sar_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                          long R, int C, float eps, sarStream_t stream);
int sar_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                          float* dx, float* dgamma, float* dbeta, long R, int C, int accumulate, sarStream_t stream);
/* LayerNormalization backward + the backward of the activation in front of it (DS(..., 'tanh') -> LN, model.py:253-254, 265-266,
 * 279-280): x = act(z) is the LayerNorm's input, dx = d/dz.  act: SAR_ACT_NONE / SAR_ACT_TANH. */
int sar_layernorm_bwd_act_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                              float* dx, float* dgamma, float* dbeta, long R, int C, int accumulate, int act, sarStream_t stream);

/* ---- Bidirectional(CuDNNGRU), merge_mode='concat': model.py:44-50.  xa_* = x*kernel + bias[:3H] (B,L,3H);
 * out (B,L,2H) = [fwd | bwd]; gates (2,B,L,4,H), hprev (2,B,L,H) are saved for BPTT; UT_* = U^T (3H,H);
 * dai/dah (2,B,L,3H) are the input-side / recurrent-side pre-activation gradients. */
int sar_bigru_fwd_f32(const float* xa_fwd, const float* xa_bwd, const float* U_fwd, const float* U_bwd,
                      const float* bh_fwd, const float* bh_bwd, float* out, float* gates, float* hprev, int B, int L,
                      int H, sarStream_t stream);
int sar_bigru_bwd_f32(const float* dout, const float* gates, const float* hprev, const float* UT_fwd,
                      const float* UT_bwd, float* dai, float* dah, int B, int L, int H, sarStream_t stream);
/* Any hidden_dim (round 4; the kernels above end at 256 units): ONE time step s of both directions.  The caller computes ha (2,B,3H) = h U with
 * sar_gemm_batched_f32 and calls sar_bigru_step_fwd_f32 (h (2,B,H): the running state, zero in front of step 0); backward, for s = L-1 .. 0:
 * sar_bigru_step_bwd_f32 (dh (2,B,H) in / out, dstep (2,B,3H) out), then dh += dstep U^T by sar_gemm_batched_f32 with accumulate.  Same saved
 * tensors as sar_bigru_fwd_f32.  Launch-bound (two launches per step); model.py:204-224 takes any hidden_dim, so does this. */
int sar_bigru_step_fwd_f32(const float* xa_fwd, const float* xa_bwd, const float* ha, const float* bh_fwd, const float* bh_bwd, float* h,
                           float* out, float* gates, float* hprev, int B, int L, int H, int s, sarStream_t stream);
int sar_bigru_step_bwd_f32(const float* dout, const float* gates, const float* hprev, float* dh, float* dai, float* dah, float* dstep,
                           int B, int L, int H, int s, sarStream_t stream);
/* The recurrence of the "bf16" mode at hidden_dim 256 (model.py:44-50, train.py default): the recurrent kernels are rounded to
 * bf16 and packed once per optimiser step (pack: 2 directions x 2 orientations x 3H x H bf16 = 1.5 MB), then stay on chip for all
 * L steps -- registers + LDS of 16-row workgroups, v_mfma_f32_16x16x32_bf16 -- instead of streaming from L2 every step.  State,
 * gates, saved tensors and accumulation are fp32; same tensors and layouts as sar_bigru_fwd_f32 / sar_bigru_bwd_f32 (no
 * transposed kernels needed), except that `gates` (2*B*L*4*H floats) is opaque: what sar_bigru_fwd_bf16 wrote is for
 * sar_bigru_bwd_bf16 of the same B (the four-row kernels of B <= 512 keep it unit-major).  H must be 256. */
int sar_bigru_pack_bf16(const float* U_fwd, const float* U_bwd, void* pack, int H, sarStream_t stream);
int sar_bigru_fwd_bf16(const float* xa_fwd, const float* xa_bwd, const void* pack, const float* bh_fwd, const float* bh_bwd,
                       float* out, float* gates, float* hprev, int B, int L, int H, sarStream_t stream);
int sar_bigru_bwd_bf16(const float* dout, const float* gates, const float* hprev, const void* pack, float* dai, float* dah,
                       int B, int L, int H, sarStream_t stream);
/* The same recurrence with IEEE fp16 as the two MFMA operands' type (csrc/gru_mfma.hip compiled with -DSAR_GRU_HALF; the fp32-class
 * "bf16x3" mode): |U| <= 1 and |h| < 1, so fp16's 11-bit mantissa costs 1.3e-4 on the accent probabilities where bf16 costs 1.1e-3
 * (tools/recipe_emulator.py) and U still fits on chip in one pass.  BPTT's pre-activation gradients enter the MFMA times `gscale` (a power
 * of two, e.g. 64 x batch; the product is divided by it again) because fp16 has five exponent bits. */
int sar_bigru_pack_f16(const float* U_fwd, const float* U_bwd, void* pack, int H, sarStream_t stream);
int sar_bigru_fwd_f16(const float* xa_fwd, const float* xa_bwd, const void* pack, const float* bh_fwd, const float* bh_bwd,
                      float* out, float* gates, float* hprev, int B, int L, int H, sarStream_t stream);
int sar_bigru_bwd_f16(const float* dout, const float* gates, const float* hprev, const void* pack, float* dai, float* dah,
                      int B, int L, int H, float gscale, sarStream_t stream);
int sar_bigru_last_fwd_f32(const float* out, float* y, int B, int L, int H, sarStream_t stream); /* seq=False, model.py:128 */
int sar_bigru_last_bwd_f32(const float* dy, float* dout, int B, int L, int H, sarStream_t stream);

/* ---- K.ctc_batch_cost(labels, y_pred, input_length, label_length): model.py:62-64.  labels float32 (B,maxU). */
size_t sar_ctc_workspace_bytes(int B, int L, int max_label_len);
int sar_ctc_batch_cost_fwd_f32(const float* probs, const float* labels, const int* in_len, const int* lab_len,
                               float* loss, void* workspace, size_t workspace_bytes, int B, int L, int V,
                               int max_label_len, sarStream_t stream);
int sar_ctc_batch_cost_bwd_f32(const float* probs, const float* labels, const int* in_len, const int* lab_len,
                               const float* loss, const float* dloss, const void* workspace, size_t workspace_bytes,
                               float* dprobs, int B, int L, int V, int max_label_len, sarStream_t stream);
int sar_ctc_greedy_decode_f32(const float* probs, const int* in_len, int* decoded, int* decoded_len, int B, int L,
                              int V, sarStream_t stream); /* K.ctc_decode(greedy=True), model.py:385-389 */

/* ---- VladPooling.call: VLAD.py:26-49.  feat (B,L,D), score (B,L,K+G), centers (K+G,D) -> out (B,K*D). */
int sar_vlad_fwd_f32(const float* feat, const float* score, const float* centers, float* out, float* A, float* inv,
                     int B, int L, int D, int K, int G, sarStream_t stream);
int sar_vlad_bwd_f32(const float* dout, const float* feat, const float* A, const float* centers, const float* out,
                     const float* inv, float* dfeat, float* dscore, float* dcenters, int B, int L, int D, int K, int G,
                     int accumulate_dfeat, sarStream_t stream);

/* ---- SphereFace/CosFace/ArcFace.call: losses.py:28-47, 74-91, 121-144 */
int sar_margin_head_fwd_f32(const float* x, const float* y, const float* W, float* probs, float* cosv, float* inv_x,
                            float* inv_w, int B, int D, int A, int mode, float s, float m, sarStream_t stream);
int sar_margin_head_bwd_f32(const float* dprobs, const float* probs, const float* cosv, const float* x, const float* y,
                            const float* W, const float* inv_x, const float* inv_w, float* dx, float* dW, float* dwn_ws,
                            int B, int D, int A, int mode, float s, float m, int accumulate_dx, sarStream_t stream);
/* The training step of a margin head in two launches instead of six: forward, categorical_crossentropy(y_true, probs) and the backward
 * of both per sample in one wave, then the column-normalisation backward of W.  y = the one-hot labels the margin is applied to (the
 * layer's second input, losses.py:38-47), y_true = the loss target (model.py:354), dloss (B) = d total / d loss_b; dx written or
 * accumulated, dW +=.  inv_w: float[A] scratch; dwn_ws: float[D*A], ZERO on entry and zero again on return. */
int sar_margin_head_train_f32(const float* x, const float* y, const float* y_true, const float* dloss, const float* W,
                              float* probs, float* loss, float* inv_w, float* dx, float* dW, float* dwn_ws, int B, int D,
                              int A, int mode, float s, float m, int accumulate_dx, sarStream_t stream);
/* ---- The accent classifier in one launch: AR_CF_DS1 = Dense(relu) -> AR_CF_DS2 = Dense(relu) -> y_accent = Dense(softmax)
 * (model.py:150-152 via DS, model.py:35-42) + categorical_crossentropy(y_true, y_accent) (model.py:350) + their backward.  x (B,H);
 * W1 (H,D1), W2 (D1,D2), W3 (D2,A) with biases; h1 (B,D1), h2 (B,D2), probs (B,A) are the three layers' outputs.  y_true NULL:
 * forward only; dloss (B, = loss weight / B) NULL: + loss (B); both: + backward -- dW / db of the three layers accumulate, dx (B,H) is
 * written.  D1, D2 <= 64, A <= 32 (the reference: 64 / 64 / 8); larger shapes take the per-layer entry points. */
int sar_accent_classifier_f32(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                              const float* b3, const float* y_true, const float* dloss, float* h1, float* h2, float* probs,
                              float* loss, float* dW1, float* db1, float* dW2, float* db2, float* dW3, float* db3, float* dx,
                              int B, int H, int D1, int D2, int A, sarStream_t stream);
/* ---- Both heads of the accent branch in one launch (training step): the accent classifier above (y_true, dloss required) AND the
 * margin head `mode` (losses.py:28-47 / 74-91 / 121-144; s, m as sar_margin_head_train_f32) with kernel Wm (H,A) on the same embedding
 * rows x: my = its one-hot second input, myt = its loss target (model.py:354), mdl (B) its loss seed -> mprobs (B,A), mloss (B),
 * minvw (A) scratch, dWm += ; dx (B,H) = the SUM of the two heads' input gradients (written).  mws: float[H*A + 1], ZERO before the
 * first call and zero again on return.  Only the sizes of the margin head's register-cached form: A in {4, 8}, H <= 256, Wm 16-byte
 * aligned -- otherwise SAR_EINVAL and the caller runs sar_accent_classifier_f32 + sar_margin_head_train_f32. */
int sar_accent_margin_train_f32(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                                const float* b3, const float* y_true, const float* dloss, float* h1, float* h2, float* probs,
                                float* loss, float* dW1, float* db1, float* dW2, float* db2, float* dW3, float* db3, float* dx,
                                const float* my, const float* myt, const float* mdl, const float* Wm, float* mprobs, float* mloss,
                                float* minvw, float* dWm, float* mws, int B, int H, int D1, int D2, int A, int mode, float s,
                                float m, sarStream_t stream);
/* ---- circle_loss: losses.py:157-172 ; Keras categorical_crossentropy: model.py:350,354 */
int sar_circle_loss_fwd_f32(const float* y, const float* s, float* loss, int B, int A, float gamma, float margin, sarStream_t stream);
int sar_circle_loss_bwd_f32(const float* y, const float* s, const float* dloss, float* ds, int B, int A, float gamma, float margin, sarStream_t stream);
int sar_cce_fwd_f32(const float* y, const float* p, float* loss, int B, int A, sarStream_t stream);
int sar_cce_bwd_f32(const float* y, const float* p, const float* dloss, float* dp, int B, int A, sarStream_t stream);

/* ---- small helpers */
int sar_act_bwd_f32(const float* dy, const float* y, float* dz, long n, int act, sarStream_t stream);
int sar_softmax_fwd_f32(const float* z, float* p, int rows, int N, sarStream_t stream);
int sar_softmax_bwd_f32(const float* dp, const float* p, float* dz, int rows, int N, sarStream_t stream);
int sar_colsum_f32(const float* x, float* out, long R, int C, sarStream_t stream);                /* out += */
int sar_axpy_f32(const float* x, float* y, float a, long n, sarStream_t stream);
int sar_fill_f32(float* y, float v, long n, sarStream_t stream);
/* zero n buffers in one launch: table = device array of n rows {pointer (16-byte aligned), number of 32-bit words} as 2 x int64 */
int sar_zero_ranges(const void* table, int n, sarStream_t stream);
/* copy n buffers in one launch: rows {destination, source, number of 32-bit words} as 3 x int64, 16-byte aligned pointers */
int sar_copy_ranges(const void* table, int n, sarStream_t stream);
int sar_avgpool1d_fwd_f32(const float* x, float* y, int B, int L, int C, sarStream_t stream);     /* model.py:125 */
int sar_avgpool1d_bwd_f32(const float* dy, float* dx, int B, int L, int C, int accumulate, sarStream_t stream);
int sar_l2norm_rows_fwd_f32(const float* x, float* v, float* inv, int R, int C, sarStream_t stream); /* model.py:162 */
int sar_l2norm_rows_bwd_f32(const float* dv, const float* v, const float* inv, float* dx, int R, int C, int accumulate, sarStream_t stream);
int sar_transpose_f32(const float* x, float* y, int R, int C, sarStream_t stream);

/* ---- Gradient exchange of the data-parallel step (the reference: keras.utils.multi_gpu_model inside one process, model.py:193-194;
 * here one process per GPU): sum-all-reduce of a bucket of the flat gradient buffer over RCCL / xGMI.  RCCL is bound at run time
 * (the copy the process already has, else /opt/rocm's): without it these return SAR_ELAUNCH, everything else in the library works.
 * id128: 128 bytes made by rank 0 (sar_comm_unique_id) and handed to every rank by the launcher's own channel; sar_comm_init is
 * collective and binds the communicator to the CURRENT HIP device; sar_allreduce_bucket_f32 is enqueued on `stream` (in place). */
int sar_comm_unique_id(void* out128);
int sar_comm_init(void** comm, const void* id128, int nranks, int rank);
int sar_allreduce_bucket_f32(void* comm, float* buf, long count, sarStream_t stream);
/* The same bucket in half the bytes (optional; DataParallel: SAR_DP_GRAD_DTYPE=bf16): buf is rounded to bf16 into scratch16 (count
 * 2-byte elements), summed over the ranks by RCCL in bf16, and widened back into buf (the fp32 master gradient).  A bandwidth /
 * precision trade: every hop of the ring rounds its partial sum to 8 mantissa bits. */
int sar_allreduce_bucket_bf16(void* comm, float* buf, void* scratch16, long count, sarStream_t stream);
int sar_comm_destroy(void* comm);

/* ---- Adam(lr, decay=2e-4) + l2 regularisers + unit_norm: model.py:197, 35-48, 163.
 * state: device block {float lr, decay, beta1, beta2, eps, lr_t; int iterations; float grad_scale}. */
int sar_adam_prepare(void* state, sarStream_t stream);
int sar_adam_l2_step_f32(float* params, const float* grads, float* m, float* v, const float* chunk_l2,
                         const void* state, float* l2_loss, long n, sarStream_t stream);
/* sar_adam_prepare + sar_adam_l2_step_f32 in one launch (round 3): every block derives lr_t from the iteration count it finds, the last
 * block to finish advances it.  ticket: EIGHT zero-initialised 32-bit words of the caller's: [0] the block ticket (zero again when the launch
 * has finished); [1] set by the caller to switch the loss-scaling guard on (the modes with fp16 operands: fp16 trunk, split-precision mode);
 * [2] the count of steps skipped so far; [3] the flag sar_grads_check_f32 sets when any gradient element is not finite (cleared by the update
 * launch); [4] the count of steps skipped IN A ROW (an applied step resets it: what "diverged" is decided on); [5..7] spare.
 * With the guard on and the flag set the WHOLE update is skipped: weights, m, v and the iteration count stay as they are -- standard
 * loss-scaling semantics; the host halves the gradient scale when it sees the counter move (model.SARModel). */
int sar_grads_check_f32(const float* grads, void* ticket, long n, sarStream_t stream);
int sar_adam_l2_update_f32(float* params, const float* grads, float* m, float* v, const float* chunk_l2, void* state, void* ticket,
                           float* l2_loss, long n, sarStream_t stream);
/* ---- data feeder: utils.py:35-36 feat_norm (sklearn MinMaxScaler.fit_transform per utterance: every
 * feature column scaled to [0,1] over ALL frames of the utterance; constant column -> 0) + utils.py:39-46 feat_reshape
 * (truncate / zero-pad to T frames).  frames = the batch's utterances back to back (sum_n, D) f32; offsets (B+1) i32 frame
 * offsets; out (B, T, D) f32 = x_data (utils.py:102).  Same float32 operation order as sklearn (bit-exact for f32 input). */
int sar_feat_minmax_pad_f32(const float* frames, const int* offsets, float* out, int B, int T, int D, sarStream_t stream);

/* out[0] = sum over the flat parameter buffer of chunk_l2[i/256] * p[i]^2: the kernel/bias_regularizer terms Keras adds
 * to `loss` (model.py:35-48) for evaluate(), where no optimiser step runs. */
int sar_l2_penalty_f32(const float* params, const float* chunk_l2, float* out, long n, sarStream_t stream);
int sar_unit_norm_f32(float* w, int D, int A, sarStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SARNET_H_ */
