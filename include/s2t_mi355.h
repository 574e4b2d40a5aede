/* libs2t_mi355.so -- C ABI of the MI355X (gfx950) ASR-training hot path.
 *
 * The reference (guangkun0818/speech2text) has no native layer: its hot path is
 * Python calling torch / torchaudio / k2 ops.  Every entry point below replaces
 * one of those call sites (cited per function, paths relative to the reference
 * tree) and is what a maintainer would bind with ctypes from the same Python
 * module.
 *
 * Conventions: all pointers are DEVICE pointers owned by the caller; fp32 data,
 * int64 ("long") lengths / labels / ranges exactly as the reference's tensors;
 * `stream` is a hipStream_t; nothing allocates or synchronises; the return value
 * is 0 on success, a hipError_t (>0) from the launch, or -1 for invalid sizes.
 */
#ifndef S2T_MI355_H_
#define S2T_MI355_H_

#ifdef __cplusplus
extern "C" {
#endif

/* ---- frontend: dataset/frontend/frontend.py:85-94 (kaldi fbank), + optional
 * model/layer/global_cmvn.py:30-38 fused.  pcm [batch][pcm_stride], num_samples
 * [batch]; out [batch][max_frames][num_mel] (frames past an utterance's own
 * count are the batch padding value 0, CMVN'd if CMVN is fused); out_frames
 * [batch] (may be NULL).  Host-built tables: window400 = povey window,
 * twiddle512 = exp(-2*pi*i*m/512) as (re,im), compact mel filterbank. */
int s2t_fbank_f32(const float* pcm, long pcm_stride, const long* num_samples, int batch,
                  const float* window400, const float* twiddle512, const int* mel_off,
                  const int* mel_k0, const float* mel_w, int nnz, int num_mel, float eps,
                  float scale_in, const float* cmvn_mean, const float* cmvn_istd, float* out,
                  int max_frames, long* out_frames, void* stream);

/* ---- CTC: model/loss/ctc_loss.py:35-41 (log_softmax + nn.CTCLoss).
 * logits [B][T][V] batch-major; targets [B][tgt_stride]; loss_per_utt [B] (nll,
 * or 0 where infinite and zero_infinity); grad_logits [B][T][V] =
 * grad_scale[b] * d nll_b / d logits (NULL to skip).  Umax (padded label width) may be at most
 * S2T_CTC_MAX_LABELS: the 2U+1 lattice states of an utterance live in one workgroup's LDS
 * (128 threads x up to 16 states each); a wider batch returns -4. */
#define S2T_CTC_MAX_LABELS 1023
long s2t_ctc_workspace_floats(int B, int T, int Umax);
int s2t_ctc_loss_fwd_bwd(const float* logits, const long* targets, long tgt_stride,
                         const long* in_len, const long* tgt_len, int B, int T, int V, int Umax,
                         int blank, int zero_infinity, const float* grad_scale, float* workspace,
                         float* loss_per_utt, float* grad_logits, void* stream);

/* ---- RNN-T lattice.  k2 call sites: model/joiner/joiner.py:100-123,
 * model/loss/pruned_rnnt_loss.py:39-48; torchaudio: model/loss/rnnt_loss.py:42-44.
 * am [B][T][C], lm [B][S+1][C], symbols [B][S], boundary [B][4]=(0,0,S_b,T_b),
 * px [B][S][T+1], py [B][S+1][T], p [B][S+1][T+1], ranges [B][T][R]. */
int s2t_fill_f32(float* p, long n, float v, void* stream);
int s2t_rnnt_row_exp(const float* x, long rows, int C, float* probs, float* rowmax, void* stream);
int s2t_rnnt_simple_pxpy(const float* am, const float* lm, const float* am_max,
                         const float* lm_max, const float* nrm, const long* symbols,
                         const long* boundary, int B, int S, int T, int C, int blank, float* px,
                         float* py, void* stream);
int s2t_rnnt_simple_w(const float* dpx, const float* dpy, const float* nrm, const float* gscale,
                      int B, int S, int T, float* W, void* stream);
int s2t_rnnt_simple_bwd(const float* am_probs, const float* lm_probs, const float* G_am,
                        const float* G_lm, const float* dpx, const float* dpy,
                        const float* gscale, const long* symbols, int B, int S, int T, int C,
                        int blank, float* d_am, float* d_lm, int accumulate, void* stream);
int s2t_mutual_info_fwd(const float* px, const float* py, const long* boundary, int B, int S,
                        int T, float* p, float* ans, void* stream);
int s2t_mutual_info_bwd(const float* px, const float* py, const long* boundary, const float* p,
                        const float* ans_grad, int B, int S, int T, float* px_grad,
                        float* py_grad, void* stream);
int s2t_rnnt_prune_ranges(const float* px_grad, const float* py_grad, const long* boundary, int B,
                          int S, int T, int s_range, long* ranges, void* stream);
/* fused joiner: logits[b,t,i,:] = act(am[b,t,:] + lm[b,ranges[b,t,0]+i,:]) is
 * never materialised; act: 0 = relu, 1 = tanh.  lse [B][T][R]. */
int s2t_rnnt_pruned_fwd(const float* am, const float* lm, const long* ranges,
                        const long* symbols, const long* boundary, int B, int S, int T, int C,
                        int R, int blank, int act, float* px, float* py, float* lse, void* stream);
int s2t_rnnt_pruned_bwd(const float* am, const float* lm, const long* ranges,
                        const long* symbols, const float* lse, const float* dpx, const float* dpy,
                        const float* gscale, int B, int S, int T, int C, int R, int blank, int act,
                        float* d_am, float* d_lm, int accumulate, void* stream);
/* materialised lattice: logits [B][T][R][V]; ranges NULL => full lattice, R = S+1 */
int s2t_rnnt_lattice_fwd(const float* logits, const long* ranges, const long* symbols,
                         const long* boundary, int B, int S, int T, int V, int R, int blank,
                         float* px, float* py, float* lse, void* stream);
int s2t_rnnt_lattice_bwd(const float* logits, const long* ranges, const long* symbols,
                         const float* lse, const float* dpx, const float* dpy,
                         const float* gscale, int B, int S, int T, int V, int R, int blank,
                         float* d_logits, void* stream);


/* ---- zipformer streaming ops (model/layer/scaling.py: Swoosh :1340-1509, BiasNorm
 * :347-476, Balancer backward :741-789 in closed form). */
int s2t_swoosh_fwd(const float* x, float* y, long n, float offset, float constant, void* stream);
int s2t_swoosh_bwd(const float* x, const float* g, float* d, long n, float offset, void* stream);
int s2t_biasnorm_fwd(const float* x, const float* bias, const float* log_scale, long rows, int D,
                     float* y, float* scales, void* stream);
/* s2t_biasnorm_fwd / _bwd with the (B,T,D) -> (T,B,D) transposition of the zipformer frontend's output
 * (model/encoder/zipformer.py:183 `x.transpose(0, 1)` on the output of Conv2dSubsampling.out_norm,
 * model/layer/subsampling.py:262) riding in the pass: x, dx are batch-major (row b T + t), y and g
 * time-major (row t B + b); scales stay in x's row order. */
int s2t_biasnorm_fwd_tb(const float* x, const float* bias, const float* log_scale, int T, int B, int D,
                        float* y, float* scales, void* stream);
int s2t_biasnorm_bwd_tb(const float* x, const float* bias, const float* scales, const float* g, int T,
                        int B, int D, float* dx, float* dbias, float* dls, void* stream);
/* BiasNorm + the layer's bypass as ONE pass each way (the end of Zipformer2EncoderLayer.forward,
 * model/encoder/zipformer.py:1330-1337): out = orig + (x * scales[row] - orig) * bypass_scale[c]
 * (* fm[row % B, c] when fm is given: the stack's feature mask); scales[row] = exp(log_scale) /
 * rms(x - bias) is written for backward, the normalised tensor itself is not.  bwd: g' = g * fm;
 * d_orig = g' (1 - bypass_scale); dx = BiasNorm's backward of g' bypass_scale; d_bypass_scale, dbias,
 * dls are ADDED to.  Equals s2t_biasnorm_fwd + s2t_bypass_fwd[_mask] and s2t_bypass_bwd[_mask] +
 * s2t_biasnorm_bwd. */
int s2t_norm_bypass_fwd(const float* x, const float* bias, const float* log_scale, const float* orig,
                        const float* bypass_scale, const float* fm, int B, long rows, int D, float* out,
                        float* scales, void* stream);
int s2t_norm_bypass_bwd(const float* x, const float* bias, const float* scales, const float* orig,
                        const float* bypass_scale, const float* g, const float* fm, int B, long rows, int D,
                        float* dx, float* d_orig, float* d_bypass_scale, float* dbias, float* dls,
                        void* stream);
int s2t_biasnorm_bwd(const float* x, const float* bias, const float* scales, const float* g,
                     long rows, int D, float* dx, float* dbias, float* dls, void* stream);
/* The whole Balancer backward (model/layer/scaling.py:741-789) in two launches and no fills:
 * out = g + |g| * (a'[c] + b'[c] x); the column statistics of x are accumulated by the first
 * launch, every workgroup of the second derives the per-channel coefficients from them itself.
 * x / g / out are row-strided (ldx / ldg / ldo floats; out == g allowed).  `workspace`:
 * s2t_balancer_bwd_workspace_floats() floats holding two alternating accumulators, zeroed ONCE
 * when allocated; the caller passes parity = 0, 1, 0, 1, ... on successive calls (each call
 * clears the accumulator of the next) and keeps all calls on one stream.  C <= 1024.
 * act_off >= 0: g is the gradient w.r.t. Swoosh(x) (offset act_off: 4 = SwooshL, 1 = SwooshR) and
 * is taken through the activation first -- the hidden Balancer of FeedforwardModule
 * (zipformer.py:1573-1593) and the Swoosh backward in one pass; act_off < 0: plain Balancer. */
long s2t_balancer_bwd_workspace_floats(void);
int s2t_balancer_bwd(const float* x, long ldx, const float* g, long ldg, long rows, int C,
                     float min_mean, float max_mean, float min_rms, float max_rms,
                     float grad_scale, float* out, long ldo, float* workspace, int parity,
                     float act_off, void* stream);
/* the two passes of s2t_balancer_bwd as separate calls: the column statistics (stats: 2 * 1024
 * floats, zeroed by the caller) may be taken where x is produced -- forward pass, side stream --
 * and only the update runs on the data-gradient chain */
int s2t_balancer_stats(const float* x, long ldx, long rows, int C, float* stats, void* stream);
int s2t_balancer_apply(const float* x, long ldx, const float* g, long ldg, long rows, int C,
                       float min_mean, float max_mean, float min_rms, float max_rms,
                       float grad_scale, float* out, long ldo, const float* stats, float act_off,
                       void* stream);


/* ---- zipformer convolution module core (model/encoder/zipformer.py:2672-2690 +
 * model/layer/scaling.py:622-681), time-major.  u (T,B,ld): x = u[..., 0:C], gate
 * pre-activation = u[..., gate_off:gate_off+C] (gate_off < 0: no gate); mask (B,T) bytes,
 * 1 = padded frame (may be NULL); wc (C,(K+1)/2), bc (C): causal taps (NULL for a plain
 * depthwise conv); wk (C,K), bk (C): chunkwise/plain taps; scale (2,C,K) edge scales or NULL;
 * chunk = chunk size in frames (>= T: one chunk).  y (T,B,C).  Backward: du (T,B,2C) (or
 * (T,B,C) without gate) is written; the parameter gradients are ACCUMULATED (zero them). */
int s2t_zipconv_fwd(const float* u, long ld, int gate_off, const unsigned char* mask, int T, int B,
                    int C, int K, int chunk, const float* wc, const float* bc, const float* wk,
                    const float* bk, const float* scale, float* y, void* stream);
/* s2t_zipconv_fwd that also writes y_act = Swoosh(y) (act_kind 1 = SwooshL, 2 = SwooshR): the
 * activation between the depthwise conv and out_proj (model/encoder/zipformer.py:2700-2703) leaves
 * with the conv's output tile instead of re-reading y. */
int s2t_zipconv_fwd_act(const float* u, long ld, int gate_off, const unsigned char* mask, int T, int B,
                        int C, int K, int chunk, const float* wc, const float* bc, const float* wk,
                        const float* bk, const float* scale, float* y, float* y_act, int act_kind,
                        void* stream);
int s2t_zipconv_bwd(const float* u, long ld, int gate_off, const unsigned char* mask, int T, int B,
                    int C, int K, int chunk, const float* wc, const float* wk, const float* bk,
                    const float* scale, const float* dy, float* du, float* dwc, float* dbc,
                    float* dwk, float* dbk, float* dscale, float* workspace, void* stream);
/* the two halves of s2t_zipconv_bwd as separate calls: the parameter gradients only feed the
 * optimizer, so a caller may run them on another stream (ordered after the producers of u / dy;
 * u, dy and workspace alive until that stream is joined) while du stays on the chain */
int s2t_zipconv_bwd_data(const float* u, long ld, int gate_off, const unsigned char* mask, int T,
                         int B, int C, int K, int chunk, const float* wc, const float* wk,
                         const float* bk, const float* scale, const float* dy, float* du,
                         void* stream);
int s2t_zipconv_bwd_params(const float* u, long ld, int gate_off, const unsigned char* mask, int T,
                           int B, int C, int K, int chunk, const float* wc, const float* wk,
                           const float* bk, const float* scale, const float* dy, float* dwc,
                           float* dbc, float* dwk, float* dbk, float* dscale, float* workspace,
                           void* stream);
long s2t_zipconv_bwd_workspace_floats(int T, int B, int C, int K);


/* ---- relative-position attention weights (model/encoder/zipformer.py:1966-2066).
 * qkp (T,B,H*(2*qd+pd)) = in_proj output [q | k | p]; pos (2T-1, H*pd) = linear_pos(pos_emb)
 * or NULL (position term skipped); kpm (B,T) bytes, 1 = padded key; amask (T,T) bytes,
 * 1 = masked (either may be NULL); W (H,B,T,T) softmax weights.  Backward: the gradient
 * of W is passed materialised (dW) and/or factored -- dW0 (B,T,T) for head 0 and up to two
 * (dO_c, V_c) pairs (T,B,H*dv_c) whose outer products are contracted on the fly, so the
 * consumers' (H,B,T,T) gradients are never written; with factors, delta_ws (H,B,T) must hold
 * sum_j W*dW on entry (delta_given=1).  dqkp (T,B,Dp) and dpos fully written (dpos is cleared
 * by the entry point before the partial sums are added; the caller need not zero it). */
int s2t_relpos_attn_fwd(const float* qkp, const float* pos, const unsigned char* kpm,
                        const unsigned char* amask, int T, int B, int H, int qd, int pd, float* W,
                        void* stream);
/* As s2t_relpos_attn_fwd, and *pen_flag (device-visible, e.g. pinned host memory; the caller zeroes
 * it) is set to 1.0f when any raw score -- q.k + p.pos, before masking -- exceeds pen_limit in
 * absolute value: whether the reference's penalize_abs_values_gt(scores, limit=25, penalty=1e-4)
 * (zipformer.py:2010-2026, scaling.py:905-935) contributes a gradient on this call.  Returns -3
 * when the shape is outside the MFMA kernel's range (the only one that reports the limit). */
int s2t_relpos_attn_fwd_flag(const float* qkp, const float* pos, const unsigned char* kpm,
                             const unsigned char* amask, int T, int B, int H, int qd, int pd,
                             float* W, float pen_limit, float* pen_flag, void* stream);
int s2t_relpos_attn_bwd(const float* qkp, const float* pos, const unsigned char* kpm,
                        const unsigned char* amask, int T, int B, int H, int qd, int pd,
                        const float* W, const float* dW, const float* dW0, const float* dO1,
                        const float* V1, int dv1, const float* dO2, const float* V2, int dv2,
                        int delta_given, float* delta_ws, float* dqkp, float* dpos,
                        float* workspace, void* stream);
/* floats of `workspace` (per-(b,h,query block) partial sums of dpos, reduced in a second pass
 * instead of contended atomics); may be NULL when pos is NULL. */
long s2t_relpos_attn_bwd_workspace_floats(int T, int B, int H, int pd);
/* attention apply (model/encoder/zipformer.py:2269): transpose=0: out[i] = sum_j W[i,j] v[j];
 * transpose=1: out[j] = sum_i W[i,j] v[i] (gradient w.r.t. the values).  v,out (T,B,H*dv). */
int s2t_attn_apply(const float* W, const float* v, int T, int B, int H, int dv, int transpose,
                   float* out, void* stream);


/* ---- weight / bias gradient of the linear layers (torch.nn.Linear under loss.backward();
 * model/encoder/zipformer.py:1573-1593 FeedforwardModule, 1966-1992 in_proj, scaling.py
 * ScaledLinear): dW (N,M) = g^T a, db (N) = column sums of g (db may be NULL), for g (R,N) and
 * a (R,M) with row strides ldg/lda (floats), R >> N,M.  The rows are split over the chip and
 * the partial sums land in `workspace` (s2t_linear_wgrad_workspace_floats) before a reduce
 * pass; accumulate=1 adds to dW/db instead of overwriting.  N, M, ldg, lda must be even and
 * the bases 8-byte aligned (-1 otherwise). */
long s2t_linear_wgrad_workspace_floats(int R, int N, int M);
int s2t_linear_wgrad(const float* g, long ldg, const float* a, long lda, int R, int N, int M,
                     float* dW, float* db, int accumulate, float* workspace, void* stream);

/* ---- Whiten gradient shaping (model/layer/scaling.py:949-1095).  Forward, when the module
 * fires: xtx (C,C) = x^T x and colsum (C) (both accumulated by the TN mode of s2t_gemm_f32 over
 * (x, x)) -> per-group centred covariance cov (G,cg,cg), mean (C), scal = [mean diag,
 * sum cov^2 / C, denom, metric]; the metric is also stored to host_metric (pinned host memory,
 * may be NULL) so the host can read it after an event without draining the stream.  xtx and
 * colsum are CONSUMED: the kernel leaves them zeroed, ready for the next accumulating GEMM (no
 * fill launch per call); `workspace` = 4 + 2*C floats, zeroed once when allocated (a
 * self-resetting ticket + per-row partial sums).  Backward: dcov (C,C, block diagonal) =
 * d metric / d cov, bias (C) = -mean . dcov, sums[2] zeroed; the caller forms pg = x dcov + bias
 * with a plain GEMM; s2t_whiten_apply writes out = g + pg * grad_scale * |g| / (|pg| + 1e-20). */
int s2t_whiten_metric(float* xtx, float* colsum, long n, int G, int cg, float* cov, float* mean,
                      float* scal, float* host_metric, float* workspace, void* stream);
/* The round-6 form of the backward: s2t_whiten_prep = s2t_whiten_dcov's dcov / bias, and sums64 = the
 * [2][64] partial-sum slots of the two norms zeroed -- everything that depends on x only, so it runs in
 * FORWARD on the statistics' stream, followed there by s2t_x3p_split of dcov into a per-site piece
 * buffer.  Backward is then s2t_gemm_x3p_sq (pg = x dcov + bias from dcov's pieces, with ||g||^2 and
 * ||pg||^2 taken in its epilogue: the norms are those of the pg actually computed, as the reference
 * takes them) and s2t_whiten_combine64 (out = g + pg * grad_scale ||g|| / (||pg|| + 1e-20)).
 * Late round 6: pg does not depend on the gradient either -- s2t_gemm_x3p_sq with other == NULL (only
 * ||pg||^2 is taken) runs in FORWARD after the split, on the statistics' stream; backward's chain is
 * s2t_sumsq64 (||g||^2 into row 0 of the same slots) + s2t_whiten_combine64. */
int s2t_whiten_prep(const float* cov, const float* mean, const float* scal, int G, int cg, float* dcov,
                    float* bias, float* sums64, void* stream);
int s2t_gemm_x3p_sq(const float* A, long lda, const unsigned short* Bp, int N, int K, float* C, long ldc,
                    int M, const float* bias, const float* other, long ld_other, float* sums, int tile,
                    void* stream);
int s2t_sumsq64(const float* g, long numel, float* sums64, void* stream);
int s2t_whiten_combine64(const float* g, const float* pg, long numel, float grad_scale,
                         const float* sums64, float* out, void* stream);
int s2t_whiten_dcov(const float* cov, const float* mean, const float* scal, int G, int cg,
                    float* dcov, float* bias, float* sums, void* stream);
int s2t_whiten_apply(const float* g, const float* pg, long numel, float grad_scale, float* sums,
                     float* out, void* stream);
/* limit_param_value backward (model/layer/scaling.py:1153-1190): out = g with the sign flipped
 * where (g > 0 and x < lo), then where (g < 0 and x > hi). */
int s2t_limit_param_grad(const float* x, const float* g, float lo, float hi, long n, float* out,
                         void* stream);

/* ---- BEST-RQ labels (model/ssl/best_rq.py:168-217,259-294): stack 9 taps (kernel (3,3),
 * stride (2,2)) -> projection (F*9 -> D) -> nearest code (cosine == euclidean on normalised
 * vectors), label = index + 1, first index on ties.  fp64 internally; bit-exact integer output. */
int s2t_bestrq_labels(const float* feats, int B, int T, int F, const float* proj, int D,
                      const float* codebooks, int ncb, int K, int T2, long* labels, void* stream);


/* col2im of the 3x3 convolutions of Conv2dSubsampling (model/layer/subsampling.py:184-229, the
 * data gradient of nn.Conv2d under loss.backward()) on channel-last data, as one gather pass:
 * dc (B,Ho,Wo,3,3,C) = dy . W^T per patch -> dx (B,H,W,C), Ho = (H-3)/sh+1, Wo = (W-3)/sw+1. */
int s2t_col2im3x3_nhwc(const float* dc, int B, int H, int W, int C, int Ho, int Wo, int sh, int sw,
                       float* dx, void* stream);

/* First subsampling convolution as a direct stencil: Conv2d(1, CO, 3, padding=(0, pw)) of
 * Conv2dSubsampling (model/layer/subsampling.py:184-229) on x (B,H,W) -> y (B,H-2,W+2pw-2,CO),
 * weights in nn.Conv2d's (CO,1,3,3) layout.  mode 0: y = conv(x) + bias; mode 1: dw (CO*9) and
 * db (CO) ACCUMULATED from (x, g) (caller zeroes them); mode 2: dx (B,H,W) from (g, w).  Returns -2
 * for channel counts other than the reference's 8 (the caller then uses im2col + GEMM). */
int s2t_conv3x3_c1(int mode, const float* x, const float* w, const float* bias, const float* g, int B,
                   int H, int W, int pw, int CO, float* y, float* dw, float* db, float* dx,
                   void* stream);

/* Second subsampling convolution as direct kernels: Conv2d(8, 32, 3, stride 2) of
 * Conv2dSubsampling (model/layer/subsampling.py:184-229) on x (B,H,W,8) -> y (B,(H-3)/2+1,
 * (W-3)/2+1,32), weights in nn.Conv2d's (CO,CI,3,3) layout.  mode 0: y = conv(x) + bias; mode 2:
 * dx (B,H,W,8) from (g, w) (the weight gradient stays a TN GEMM over the patch matrix).  Returns -2
 * for other channel counts (the caller then uses im2col + GEMM). */
int s2t_conv3x3_s2(int mode, const float* x, const float* w, const float* bias, const float* g, int B,
                   int H, int W, int CI, int CO, float* y, float* dx, void* stream);

/* ---- 3x3 convolution products with the patch matrix read in place (no im2col copy in HBM):
 * the third subsampling conv and the weight gradients of the second and third
 * (model/layer/subsampling.py:226-243 nn.Conv2d(8,32,3,stride=2), nn.Conv2d(32,128,3,stride=(1,2))).
 * x (B,H,W,C) channel-last, R = B*Ho*Wo output positions, C % 4 == 0 and CO % 4 == 0 (-2 otherwise).
 *   mode 0: out[R,CO] = patches(x)[R,9C] . w2[CO,9C]^T (+ bias[CO]);  w2_or_g = w2 in (cout,kh,kw,cin) order
 *   mode 2: out[CO,9C] += g[R,CO]^T . patches(x)[R,9C],  db[CO] += column sums of g (db may be NULL);
 *           w2_or_g = g; out / db zeroed (or holding the running gradient) by the caller.
 * bf16x3 matrix-core arithmetic: fp32-level error. */
int s2t_conv3x3_gemm(int mode, const float* x, int B, int H, int W, int C, int sh, int sw, int CO,
                     const float* w2_or_g, const float* bias, float* out, float* db, void* stream);

/* ---- channel-last depthwise conv2d of the zipformer frontend (ConvNeXt 7x7,
 * model/layer/subsampling.py:47-53,121).  x,y (N,H,W,C); wgt (C,KH,KW); "same" zero padding.
 * flip=1 applies the flipped taps (= backward data).  wgrad writes dw (C,KH,KW) and db (C). */
int s2t_dwconv2d_nhwc_fwd(const float* x, const float* wgt, const float* bias, int N, int H, int W,
                          int C, int KH, int KW, int flip, float* y, void* stream);
/* same with y = conv(x) + add (add has y's layout; 7x7 only, -2 otherwise): ConvNeXt's residual
 * gradient (subsampling.py:57 `bypass + x`) added inside the backward-data pass. */
int s2t_dwconv2d_nhwc_fwd_add(const float* x, const float* wgt, const float* bias, const float* add,
                              int N, int H, int W, int C, int KH, int KW, int flip, float* y,
                              void* stream);
long s2t_dwconv2d_wgrad_workspace_floats(int N, int H, int C, int KH, int KW);
int s2t_dwconv2d_nhwc_wgrad(const float* x, const float* dy, int N, int H, int W, int C, int KH,
                            int KW, float* workspace, float* dw, float* db, void* stream);

/* ---- fused multi-tensor ScaledAdam + grad-norm clip + zero_grad over one flat fp32 buffer
 * (optimizer/scaled_adam.py:408-527 _get_clipping_scale, :563-736 _step_one_batch / _size_update
 * / _step / _step_scalar; the trainer's gradient_clip_val of config/training/<task>.yaml `trainer:`).
 * The flat buffer holds every tensor padded to 16 bytes; a host-built chunk table cuts each
 * tensor into chunks of s2t_optim_chunk_elems() elements: chunk_off/len/seg [nchunks],
 * seg_chunk_begin [nseg+1], seg_len [nseg].  s2t_seg_stats writes partial [nchunks][3] =
 * (sum g^2, sum p g, sum p^2); s2t_scaled_adam_coef (one workgroup, one param group = tensors
 * [seg_lo,seg_hi)) turns them into per-tensor coefficients segc [nseg][s2t_optim_segc_floats()]
 * and advances the group state (param_rms, scale_exp_avg_sq [nseg_g], scale_grads [P][nseg_g],
 * model_norms [period], fstate [3] = threshold / last norm / last clip factor, istate [3] =
 * has_threshold / num_clipped / non-finite-median flag); s2t_scaled_adam_apply updates
 * p, delta, exp_avg_sq in place and stores the clipped gradient (or zeros if zero_grad).
 * skip (s2t_scaled_adam_coef, s2t_adam_apply): NULL, or a DEVICE flag: != 0 makes the step a no-op
 * on parameters and optimizer state (the gradient is still cleared) -- the data-parallel reducer's
 * "this step's gradient was dropped on every rank" flag (speech2text_amd/ddp.py), no host sync. */
int s2t_optim_chunk_elems(void);
int s2t_optim_segc_floats(void);
int s2t_seg_stats(const float* p, const float* g, const int* chunk_off, const int* chunk_len,
                  int nchunks, float* partial, void* stream);
int s2t_scaled_adam_coef(const float* partial, const int* seg_chunk_begin, const int* seg_len,
                         int nchunks_all, int seg_lo, int seg_hi, float lr, float beta1,
                         float beta2, float eps, float scalar_lr_scale, float param_min_rms,
                         float param_max_rms, float scalar_max, float clip_val,
                         float clipping_scale, int step, int size_update_period,
                         int clipping_update_period, float bc2, float bc2_size, float beta2c,
                         float* param_rms, float* scale_exp_avg_sq, float* scale_grads,
                         float* model_norms, float* fstate, int* istate, float* segstat,
                         float* segc, const float* skip, void* stream);
int s2t_scaled_adam_apply(float* p, float* g, float* delta, float* exp_avg_sq,
                          const int* chunk_off, const int* chunk_len, const int* chunk_seg,
                          int nchunks, const float* segc, int zero_grad, void* stream);

/* ---- fp32 MFMA GEMM family with fused prologue / epilogue (csrc/gemm.hip) for the dense layers:
 * nn.Linear / ScaledLinear / ActivationDropoutAndLinear of model/encoder/zipformer.py:1924-2695,
 * model/layer/scaling.py:1512-1583 and their backward passes.
 *   mode 0 (NT): C[M,N]  = pro_a(A[M,K]) B[N,K]^T  (+ bias[N]) (* act'(act_src)) (+ resid) (+ C)
 *   mode 1 (NN): C[M,N]  = A[M,K] B[K,N]            same epilogue
 *   mode 2 (TN): C[M,N] += A[K,M]^T pro_b(B[K,N])   atomically (split over K); colsum[M] += sum_k A
 * act kinds: 0 none, 1 SwooshL, 2 SwooshR.  Leading dimensions in floats; A and B 16-byte
 * aligned with lda, ldb % 4 == 0 (and K % 4 == 0 for modes 0/1), else -2 (caller falls back to
 * a library GEMM). */
int s2t_gemm_f32(int mode, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                 int M, int N, int K, const float* bias, const float* resid, long ldr,
                 const float* act_src, long lds, int act_kind, int pro_a, int pro_b, float* colsum,
                 int accumulate, void* stream);

/* `batch` independent fp32 products in one launch (bf16x3 matrix cores), X_b = X + b * sX floats:
 * mode 0: C_b[M,N] = A_b[M,K] . B_b[N,K]^T; mode 1: C_b = A_b[M,K] . B_b[K,N];
 * mode 2: C_b[M,N] += A_b[K,M]^T . B_b[K,N] (fp32 atomics: zero C first).  Replaces torch.bmm in
 * the nonlinear attention (reference model/encoder/zipformer.py:2438-2483: attn_weights[0] @ x and
 * its two gradients).  -2 = alignment rules of s2t_gemm_f32 not met (keep the library). */
int s2t_gemm_f32_batched(int mode, const float* A, long lda, long sA, const float* B, long ldb, long sB,
                         float* C, long ldc, long sC, int M, int N, int K, int batch, void* stream);

/* s2t_gemm_f32 modes 0 / 1 (bias only) that also ADDS sums[0] += ||other||_F^2 and sums[1] += ||C||_F^2,
 * other = an (M, N) matrix with rows ld_other apart: the two norms Whiten's backward combines
 * (reference model/layer/scaling.py:1024-1027: x_grad + |x_grad| / |penalty_grad| * grad_scale *
 * penalty_grad), taken while C = x dcov + bias leaves the accumulators instead of by a pass over both
 * tensors.  s2t_whiten_combine is the update that follows (s2t_whiten_apply without its own sums
 * pass).  -2: operands outside the 16-byte epilogue's rules (run s2t_gemm_f32 + s2t_whiten_apply). */
int s2t_gemm_f32_sq(int mode, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                    int M, int N, int K, const float* bias, const float* other, long ld_other,
                    float* sums, void* stream);
int s2t_whiten_combine(const float* g, const float* pg, long numel, float grad_scale, const float* sums,
                       float* out, void* stream);

/* s2t_gemm_f32 modes 0 (NT) / 1 (NN) with the block tile chosen by the caller: tile = "tm tn"
 * digits for a (64 tm) x (64 tn) tile, one of 11 12 21 22 23; 0 = the dispatcher's choice. */
int s2t_gemm_f32_tiled(int mode, const float* A, long lda, const float* B, long ldb, float* C,
                       long ldc, int M, int N, int K, const float* bias, const float* resid,
                       long ldr, int tile, void* stream);

/* Arithmetic of the TN (weight-gradient) products, s2t_gemm_f32 mode 2 / s2t_gemm_tn_grouped /
 * s2t_gemm_xtx: 1 (default, or S2T_TN_X3=1) = both fp32 operands split exactly into three bf16
 * pieces, six v_mfma_f32_32x32x16_bf16 products per 16-deep step, fp32 accumulation (fp32-level
 * error); 0 = v_mfma_f32_32x32x2_f32.  set >= 0 selects, set < 0 queries; returns the mode. */
int s2t_tn_x3(int set);
/* Kernel form of the bf16-split TN products (round 5): 1 (default, or S2T_TN_W=1) = wave-specialised
 * 128x128 / 128x192 / 192x128 tiles (producer waves load 16-byte runs and split, consumer waves run
 * the MFMAs; aligned non-symmetric problems: s2t_gemm_f32 mode 2, s2t_gemm_tn_grouped, and
 * s2t_conv3x3_gemm mode 2 with its implicit patch operand); 0 = the all-waves form (every wave splits
 * what it staged and multiplies it: 64x64 tiles under six products, 128x128 under three).  Same
 * arithmetic, different summation order.  With no setting (S2T_TN_W unset, or set == 2) the form
 * follows the weight-gradient class's arithmetic (s2t_gemm_arith_of(2)): six products -> 1, three -> 0
 * (round 6: measured per step; the implicit-patch product stays on the W form).  set = 0 / 1
 * forces, set = 2 returns to automatic, set < 0 queries; returns the form now in effect. */
int s2t_tn_w(int set);
/* the same switch for the NT / NN products of s2t_gemm_f32 (default 1) */
int s2t_nn_x3(int set);

/* x^T x for the Whiten statistics (model/layer/scaling.py:949-1012): xtx (C,C, ldc) += x^T x and
 * colsum (C) += column sums of x (R,C, ldx), restricted to what the per-group covariance needs:
 * the 64x64 tiles on or above the diagonal that contain a pair of channels of the same group of
 * cg channels.  Tiles below the diagonal are NOT written (s2t_whiten_metric mirrors them). */
int s2t_gemm_xtx(const float* x, long ldx, int R, int C, int cg, float* xtx, long ldc,
                 float* colsum, void* stream);

/* The weight-gradient GEMMs of one layer in ONE launch (mode TN of s2t_gemm_f32, 64x64 tiles):
 * for each problem  C[M,N] += A[K,M]^T . B[K,N]  and  colsum[m] += sum_k A[k][m]  (colsum may be
 * NULL), both scaled by `alpha` (the conformer's 0.5 feed-forward residual weight rides here
 * instead of in a scaling pass over the gradient).  A = the gradient of a Linear's output (rows x out_features), B = its input
 * (rows x in_features), C / colsum = the weight / bias gradient views of the flat gradient buffer
 * (what loss.backward() leaves in .grad for every nn.Linear of
 * model/encoder/zipformer.py:1095-1221).  `probs` is a HOST array of n entries. */
typedef struct S2tTnProblem {
  const float* A;
  long lda;
  const float* B;
  long ldb;
  float* C;
  long ldc;
  int M, N, K;
  float* colsum;
  float alpha;
} S2tTnProblem;
int s2t_gemm_tn_grouped(int n, const S2tTnProblem* probs, void* stream);

/* ---- BEST-RQ SSL heads: fused log-softmax + smoothed-target loss (model/loss/kl_divergence.py:
 * 36-76, model/loss/cross_entropy.py:38-69; call site task_factory/ssl_task.py:140-158).
 * logits [rows][K]; labels [rows]; target t_c = t_other (c != label) / t_label; row_loss[r] =
 * c0 - sum_c t_c log_softmax(scale x)_c and lse[r]; backward grad[r][c] = row_weight[r] * scale *
 * (softmax_c - t_c)  (row_weight = upstream gradient * mask / mask.sum()). */
int s2t_smoothed_nll_fwd(const float* logits, const long* labels, long rows, int K, float scale,
                         float t_other, float t_label, float c0, float* row_loss, float* lse,
                         void* stream);
int s2t_smoothed_nll_bwd(const float* logits, const long* labels, const float* lse,
                         const float* row_weight, long rows, int K, float scale, float t_other,
                         float t_label, float* grad, void* stream);

/* ---- validation-time greedy decoders (model/decoding.py:51-82 CtcGreedyDecoding, :196-271
 * RnntGreedyDecoding; batch_search :27-48 calls them per utterance).  One launch per batch.
 * s2t_ctc_greedy: logits [B][T][V] -> tokens [B][T] (first out_len[b] valid), argmax ties take
 * the first index.  s2t_rnnt_greedy_stateless: am [B][T][V] = joiner enc_proj(encoder_out);
 * stateless predictor parameters (embedding [S][E], depthwise conv [E][ctx], linear [D][E]+[D]),
 * joiner pre_proj [V][D]+[V]; act 0 relu / 1 tanh; at most max_token_step+1 symbols per frame
 * as the reference loop; tokens [B][max_out]. */
int s2t_ctc_greedy(const float* logits, const long* lengths, int B, int T, int V, int blank,
                   long* tokens, long* out_len, void* stream);
int s2t_rnnt_greedy_stateless(const float* am, const long* lengths, const float* emb,
                              const float* conv_w, const float* lin_w, const float* lin_b,
                              const float* pre_w, const float* pre_b, int B, int T, int V, int E,
                              int D, int ctx, int act, int max_token_step, int max_out, int blank,
                              long* tokens, long* out_len, void* stream);

/* ---- batched on-device augmentation + collate next to the fbank kernel
 * (dataset/frontend/data_augmentation.py:13-56 AddNoise, :59-118 MixFeats, :150-196 SpecAugment;
 * dataset/utils.py:182-202 batch()).  Random decisions are made on the host as the reference
 * does; each op is one launch over the padded batch.
 * s2t_row_energy: out[b] = sum over the len[b]*D valid elements of exp(x) (mode 0) or x^2 (mode 1).
 * s2t_mix: mode 0 MixFeats on log-mel (T,D) rows, mode 1 AddNoise on PCM (D = 1); the noise is
 *   indexed (start[b] + t) mod nlen[b] (= the reference's repeat + slice).
 * s2t_specaug: zero the nt time spans and nf frequency spans (start, end) of each utterance.
 * s2t_pad_rows: packed rows + element offsets [B+1] -> zero-padded (B, Lmax, D). */
int s2t_row_energy(const float* x, long stride, const long* len, int B, int D, int mode, float* out,
                   void* stream);
int s2t_mix(const float* src, long sstride, const long* slen, const float* noise, long nstride,
            const long* nlen, const long* start, const float* snr, const float* src_e,
            const float* noise_e, int B, long rows_max, int D, int mode, float max_gain_db,
            float* out, void* stream);
int s2t_specaug(float* feats, int B, int T, int F, const int* tspan, int nt, const int* fspan,
                int nf, void* stream);
int s2t_pad_rows(const float* packed, const long* offsets, int B, long Lmax, int D, float* out,
                 void* stream);

/* ---- plain dense GEMMs of the Linear layers through hipBLASLt's C API (csrc/gemm_lib.hip), with
 * bias and residual / accumulation in the epilogue (model/encoder/zipformer.py:1095-1221
 * `src = src + module(src)`, and the data gradients under loss.backward()).
 * mode 0: D[M,N] = X[M,K] W[N,K]^T (+ bias[N]) (+ beta C[M,N]);  mode 1: D[M,K] = X[M,N] W[N,K] (+ beta C).
 * Row-major, leading dimensions in floats; C may be NULL (beta ignored) or alias D.  Returns -2 when
 * the library offers no algorithm for the shape (the caller then uses torch.matmul). */
int s2t_linear_lt(int mode, const float* X, long ldx, const float* W, long ldw, const float* bias,
                  const float* C, long ldc, float beta, float* D, long ldd, int M, int N, int K,
                  void* workspace, long ws_bytes, void* stream);
/* plans made (one per exact shape, table capped) and buckets {mode, half-octave of M, N, K,
 * bias} whose candidates were timed (capped by S2T_LT_TUNE_MAX, default 192) so far. */
int s2t_linear_lt_stats(int* plans, int* timed);
/* s2t_linear_lt times OUR NT / NN kernel (s2t_gemm_f32, bf16x3 form, bias + residual epilogue) as one
 * more candidate of a bucket and keeps it where it beats the library's best by > 8 % twice
 * (S2T_LT_OWN=0 disables); launches it has served so far: */
long s2t_linear_lt_own_calls(void);

/* ---- fused glue of the zipformer layer (csrc/zip_glue.hip), rows of C channels, time-major.
 * bypass (model/encoder/zipformer.py:1523-1555): out = orig + (src - orig) * scale[c]; backward
 * writes d_orig, d_src and ACCUMULATES d_scale[c] (caller zeroes it).
 * nonlinear attention (zipformer.py:2438-2483): u (T,B,3C) = in_proj(x) = [s | x | y];
 * gate_fwd: xs (B,T,C) = x * tanh(s) (batch-major for the head-0 weights @ x bmm);
 * out_fwd: o (T,B,C) = z (B,T,C) * y;  out_bwd: dz (B,T,C) = g * y, du[..,2C:3C] = g * z;
 * gate_bwd: du[..,0:C] = dxs * x * (1 - tanh(s)^2), du[..,C:2C] = dxs * tanh(s). */
int s2t_bypass_fwd(const float* orig, const float* src, const float* scale, long rows, int C,
                   float* out, void* stream);
int s2t_bypass_bwd(const float* orig, const float* src, const float* scale, const float* g,
                   long rows, int C, float* d_orig, float* d_src, float* d_scale, void* stream);
/* layer-executor helpers (speech2text_amd/zip_layer.py): bypass backward whose d_orig also takes
 * the gradient already collected for orig (acc_in); grad += d with limit_param_value's sign flip
 * (model/layer/scaling.py:1153-1190) applied to d when `limit`; the softmax-backward row constants
 * delta[h,b,i] of the attention weights from its deferred consumers (pairs (dO,O) of the two
 * SelfAttention modules, dW0 (B,T,T) of the head-0 consumer), one launch. */
int s2t_bypass_bwd_acc(const float* orig, const float* src, const float* scale, const float* g,
                       const float* acc_in, long rows, int C, float* d_orig, float* d_src,
                       float* d_scale, void* stream);
int s2t_attn_delta_pairs(const float* W, const float* dW0, const float* dO1, const float* O1,
                         int dv1, const float* dO2, const float* O2, int dv2, int T, int B, int H,
                         float* delta, void* stream);
/* bypass with the stack's per-utterance feature mask (zipformer.py:1095-1113, `output * feature_mask`
 * after every layer) folded in: out = (orig + (src - orig) * scale[c]) * fm[b, c], rows ordered
 * (t, b); the backward multiplies the incoming gradient by fm on the fly. */
int s2t_bypass_fwd_mask(const float* orig, const float* src, const float* scale, const float* fm,
                        int B, long rows, int C, float* out, void* stream);
int s2t_bypass_bwd_mask(const float* orig, const float* src, const float* scale, const float* g,
                        const float* fm, int B, long rows, int C, float* d_orig, float* d_src,
                        float* d_scale, void* stream);
/* SimpleDownsample (zipformer.py:1653-1695): out (ceil(T/ds),B,C) = sum_k w[k] src[min(tt*ds+k, T-1)]
 * with w = softmax(bias) supplied by the caller (ds <= 8); the backward writes d_src (T,B,C) and
 * ACCUMULATES dw[ds] (caller zeroes it; the softmax backward stays with the caller).
 * SimpleUpsample + the out_combiner BypassModule of a downsampled stack (zipformer.py:1253-1283,
 * 1698-1719): out[t] = orig[t] + (src[t / up] - orig[t]) * scale[c] with src (ceil(T/up),B,C) --
 * the upsampled tensor is never materialised; the backward writes d_orig, d_src and ACCUMULATES
 * d_scale[C] (caller zeroes it). */
int s2t_downsample_fwd(const float* src, const float* w, int ds, int T, int B, int C, float* out,
                       void* stream);
int s2t_downsample_bwd(const float* src, const float* w, const float* g, int ds, int T, int B, int C,
                       float* d_src, float* dw, void* stream);
/* the same pair with the OUTPUT (and its gradient g) batch-major, (B, ceil(T/ds), C): the encoder's
 * final x.transpose(0, 1) (model/encoder/zipformer.py:199) rides in the pass that writes / reads it. */
int s2t_downsample_fwd_bt(const float* src, const float* w, int ds, int T, int B, int C, float* out,
                          void* stream);
int s2t_downsample_bwd_bt(const float* src, const float* w, const float* g, int ds, int T, int B, int C,
                          float* d_src, float* dw, void* stream);
int s2t_bypass_up_fwd(const float* orig, const float* src, const float* scale, int up, int T, int B,
                      int C, float* out, void* stream);
int s2t_bypass_up_bwd(const float* orig, const float* src, const float* scale, const float* g, int up,
                      int T, int B, int C, float* d_orig, float* d_src, float* d_scale, void* stream);
/* Up to 8 small parameters in one launch: grad[e] += d[e] (limit_param_value's sign flip,
 * scaling.py:1153-1190, applied to d first where `limit`), then d[e] = 0 -- the accumulators the
 * layer's backward kernels add into are handed back clean.  `items` is a HOST array. */
typedef struct S2tCommit {
  const float* x;
  float* d;
  float* grad;
  float lo, hi;
  int limit;
  long n;
} S2tCommit;
int s2t_param_grad_commit_n(int n, const S2tCommit* items, void* stream);
int s2t_nonlin_gate_fwd(const float* u, int T, int B, int C, float* xs, void* stream);
int s2t_nonlin_out_fwd(const float* z, const float* u, int T, int B, int C, float* o, void* stream);
int s2t_nonlin_out_bwd(const float* g, const float* z, const float* u, int T, int B, int C, float* dz,
                       float* du, void* stream);
int s2t_nonlin_gate_bwd(const float* dxs, const float* u, int T, int B, int C, float* du,
                        void* stream);

/* ---- conformer block (torchaudio.models.Conformer as called at model/encoder/conformer.py:
 * 170-178,193; block structure in csrc/conf_elem.hip).  All (rows, C) tensors row-major fp32,
 * rows ordered (t, b), C % 4 == 0, C <= 1024, 16-byte aligned.
 * s2t_layernorm_fwd: nn.LayerNorm.  y != NULL: the input is x + alpha * y (the layer's residual
 * sums "0.5 * ffn(x) + x" / "x + module(x)"), written to xsum as well.  stats [rows][2] = (mean,
 * rstd) for the backward.
 * s2t_layernorm_bwd: dx = LayerNorm'(dy) (+ resid, the residual branch's gradient); the
 * per-workgroup sums of d gamma / d beta go to `partial` (s2t_layernorm_bwd_partial_floats(rows, C)
 * floats).  s2t_layernorm_param_grad folds the partials of n LayerNorms of width C in one launch:
 * dgamma[C] += ..., dbeta[C] += ... (the parameters' views of the flat gradient buffer); `items`
 * is a HOST array. */
int s2t_layernorm_fwd(const float* x, const float* y, float alpha, const float* gamma,
                      const float* beta, long rows, int C, float eps, float* xsum, float* out,
                      float* stats, void* stream);
typedef struct S2tLnFold {
  const float* partial;
  long rows;
  float* dgamma;
  float* dbeta;
} S2tLnFold;
long s2t_layernorm_bwd_partial_floats(long rows, int C);
int s2t_layernorm_bwd(const float* x, const float* stats, const float* gamma, const float* dy,
                      const float* resid, long rows, int C, float* dx, float* partial,
                      void* stream);
int s2t_layernorm_param_grad(int n, const S2tLnFold* items, int C, void* stream);
/* nn.SiLU of the feed-forward modules: a = h * sigmoid(h); dh = scale * da * silu'(h) (dh may
 * alias da; scale carries the layer's 0.5 feed-forward residual weight). */
int s2t_silu_fwd(const float* h, long n, float* a, void* stream);
int s2t_silu_bwd(const float* h, const float* da, long n, float scale, float* dh, void* stream);
/* nn.Dropout sites of the conformer block (torchaudio.models.Conformer: after the feed-forward
 * SiLU, after each module's last Linear / pointwise conv, model/encoder/conformer.py:170-178).  The
 * keep decision of element i is a stateless hash of (seed, i) -- the one s2t_mhsa_* uses for the
 * attention probabilities -- so backward regenerates the forward's mask; kept values are scaled by
 * 1 / (1 - p).  s2t_dropout_add: out = x + alpha * drop(y) (x may be NULL: the masked gradient);
 * s2t_silu_drop_fwd: a = drop(silu(h)); s2t_silu_drop_bwd: dh = scale * da * mask * silu'(h). */
int s2t_dropout_add(const float* x, const float* y, long n, float alpha, float p,
                    unsigned long long seed, float* out, void* stream);
int s2t_silu_drop_fwd(const float* h, long n, float p, unsigned long long seed, float* a,
                      void* stream);
int s2t_silu_drop_bwd(const float* h, const float* da, long n, float scale, float p,
                      unsigned long long seed, float* dh, void* stream);
/* nn.BatchNorm1d (training mode: statistics over all rows, biased variance for the normalisation,
 * running_mean / running_var (unbiased) / num_batches_tracked updated; running_* may be NULL)
 * followed by nn.SiLU, conv module of the conformer block.  save_mean / save_rstd [C] feed the
 * backward, which ACCUMULATES dgamma / dbeta.  workspace: s2t_bn_workspace_floats(C) floats. */
#define S2T_BN_PARTIALS 128
long s2t_bn_workspace_floats(int C);
int s2t_bn_silu_fwd(const float* x, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, long* num_batches,
                    long rows, int C, float* y, float* save_mean, float* save_rstd,
                    float* workspace, void* stream);
/* evaluation mode (running statistics): y = silu((x - mean) * rstd * gamma + beta) */
int s2t_bn_silu_apply(const float* x, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, long rows, int C, float* y, void* stream);
int s2t_bn_silu_bwd(const float* x, const float* ds, const float* save_mean,
                    const float* save_rstd, const float* gamma, const float* beta, long rows, int C,
                    float* dx, float* dgamma, float* dbeta, float* workspace, void* stream);
/* nn.MultiheadAttention core (no positional term): o = softmax(q k^T * scale, keys >= lens[b]
 * masked) v per (b, head).  q / k / v are the column blocks [qoff | koff | voff] + h * dh of the
 * in-projection output qkv (T*B rows (t, b), row stride ld); o (T*B, ldo) column h * dh;
 * lse [B][H][T] row log-sum-exp kept for the backward.  dh in {16, 32, 64}.
 * dropout_p > 0: dropout on the attention probabilities (nn.MultiheadAttention(dropout=p),
 * training); the mask is a hash of (seed, b, h, q, k) that the backward regenerates from the
 * same seed -- it is never stored.
 * s2t_mhsa_bwd: dqkv (same layout as qkv) from d_o; delta [B][H][T] is scratch. */
int s2t_mhsa_fwd(const float* qkv, long ld, int qoff, int koff, int voff, const long* lens, int T,
                 int B, int H, int dh, float scale, float dropout_p, unsigned long seed, float* o,
                 long ldo, float* lse, void* stream);
int s2t_mhsa_bwd(const float* qkv, long ld, int qoff, int koff, int voff, const long* lens, int T,
                 int B, int H, int dh, float scale, float dropout_p, unsigned long seed,
                 const float* o, const float* d_o, long ldo, const float* lse, float* delta,
                 float* dqkv, void* stream);

/* ---- first convolution of the conformer Subsampling (model/encoder/conformer.py:47-60,114-126):
 * Conv2d(1, C, 3, stride 2) + ReLU on x (B, T, F) -> out (B, T1, F1, C) channel-last, T1 =
 * (T-3)/2+1, F1 = (F-3)/2+1; w (C, 9), bias (C).  s2t_conv1_relu_wgrad: dw (C, 9) and db (C) are
 * ACCUMULATED from d_out (gradient w.r.t. out, same layout); the ReLU mask is recomputed from x.
 * C % 4 == 0, 256 % (C/4) == 0.  workspace: s2t_conv1_relu_workspace_floats(C) floats. */
long s2t_conv1_relu_workspace_floats(int C);
int s2t_conv1_relu_fwd(const float* x, const float* w, const float* bias, int B, int T, int F,
                       int C, float* out, void* stream);
int s2t_conv1_relu_wgrad(const float* x, const float* w, const float* bias, const float* d_out,
                         int B, int T, int F, int C, float* dw, float* db, float* workspace,
                         void* stream);

/* ---- Adam / AdamW on the flat parameter / gradient buffers (torch.optim.Adam / AdamW, amsgrad
 * off: the conformer YAMLs' `optimizer: type: "AdamW"`, optimizer/optim_setup.py:364-385) with the
 * trainer's grad-norm clip and zero_grad folded in: s2t_seg_stats -> s2t_clip_coef (out[0] =
 * min(1, clip / (||g|| + 1e-6)), out[1] = ||g||; clip <= 0: 1) -> s2t_adam_apply.  Chunk tables as
 * for ScaledAdam; group q owns the chunks below groups[q].chunk_hi that no earlier group owns;
 * chunks past the last group are only zeroed.  `groups` is a HOST array. */
#define S2T_ADAM_MAX_GROUPS 8
typedef struct S2tAdamGroup {
  int chunk_hi;
  float lr, beta1, beta2, eps, weight_decay;
  float bias_correction1;         /* 1 - beta1^step */
  float sqrt_bias_correction2;    /* sqrt(1 - beta2^step) */
  int decoupled;                  /* 1: AdamW, 0: Adam (L2 added to the gradient) */
} S2tAdamGroup;
int s2t_clip_coef(const float* partial, int nchunks, float clip_val, float* out, void* stream);
int s2t_adam_apply(float* p, float* g, float* exp_avg, float* exp_avg_sq, const int* chunk_off,
                   const int* chunk_len, int nchunks, int ngroups, const S2tAdamGroup* groups,
                   const float* coef, int zero_grad, const float* skip, void* stream);

/* ---- layer-norm LSTM layer of the RNN-T predictor (model/predictor/lstm_predictor.py:28-109 ->
 * torchaudio 0.13.1 _Predictor / _CustomLSTM), whole sequence per launch, one workgroup per
 * utterance (csrc/lstm.hip).  gx (T,B,4H) = x2g(x); wp_t = p2g.weight transposed (H,4H) for the
 * forward, wp = p2g.weight (4H,H) for the backward; g_* / c_* = g_norm / c_norm weight and bias
 * (all four NULL: no layer norm); h0 / c0 (B,H) or NULL = zeros.  The forward writes hs (T,B,H),
 * the final state hT / cT (B,H) and keeps ghat (T,B,4H), chat (T,B,H), rstd (T,B,2) for the
 * backward, which writes dgx (T,B,4H) (gradient w.r.t. the raw gates: the weight gradients of
 * x2g / p2g are TN GEMMs over its rows) and ACCUMULATES the four LayerNorm parameter gradients.
 * H <= 1024. */
int s2t_lnlstm_fwd(const float* gx, const float* wp_t, const float* g_gamma, const float* g_beta,
                   const float* c_gamma, const float* c_beta, const float* h0, const float* c0,
                   int T, int B, int H, float eps, float* hs, float* ghat, float* chat,
                   float* rstd, float* hT, float* cT, void* stream);
int s2t_lnlstm_bwd(const float* wp, const float* g_gamma, const float* g_beta,
                   const float* c_gamma, const float* c_beta, const float* c0, int T, int B, int H,
                   const float* ghat, const float* chat, const float* rstd, const float* dhs,
                   float* dgx, float* d_g_gamma, float* d_g_beta, float* d_c_gamma,
                   float* d_c_beta, void* stream);

/* ---- StatelessPredictor: embedding + depthwise context convolution in one pass
 * (reference model/predictor/stateless_predictor.py:27-105: nn.Embedding -> nn.Conv1d(D, D, K,
 * groups=D, bias=False) on the blank-left-padded labels).  tokens (B,L) int32 (values clamped to
 * [0,V)), emb (V,D), w (D,K) = the conv weight (D,1,K), K <= 8:
 *   out[b,u,d] = sum_k w[d,k] emb[tokens[b,u+k], d],  out (B, L-K+1, D).
 * bwd: g (B, L-K+1, D); d_emb (V,D) and d_w (D,K) are ADDED to (fp32 atomics; the caller
 * zeroes them); either may be NULL. */
int s2t_predictor_ctx_fwd(const int* tokens, const float* emb, const float* w, int B, int L, int K,
                          int D, int V, float* out, void* stream);
int s2t_predictor_ctx_bwd(const int* tokens, const float* emb, const float* w, const float* g, int B,
                          int L, int K, int D, int V, float* d_emb, float* d_w, void* stream);

/* ---- fp32 GEMM on the bf16 matrix cores with the weight operand split ahead of time
 * (csrc/gemm_x3p.hip): the forward (y = x W^T + b) and data-gradient (dx = g W) products of the
 * layers' Linears (reference model/encoder/zipformer.py:1924-1975,2372-2378,2643-2695,
 * model/layer/scaling.py:1512-1583).  The three bf16 pieces of a logical matrix Bm[n][k]
 * (= src[n*ld + k], or src[k*ld + n] when transposed) are stored fragment-major
 * [ceil(N/32)][2 ceil(K/32)][3][64 lanes][8], zero-padded: s2t_x3p_plane_elems(N,K) bf16.
 * s2t_x3p_split: every matrix of a model in ONE launch -- tab = DEVICE array of n S2tPlaneDesc
 *   (src_off: floats from base; dst_off: bf16 from dst; blk_begin: first block of the descriptor,
 *   ascending, descriptor d owns s2t_x3p_split_blocks(N,K) blocks), total_blocks = their sum.
 * s2t_gemm_x3p: C[M,N] = A[M,K] . Bm^T (+ bias[N]) (* act'(act_src[M,N])) (+ resid[M,N])
 *   (+ resid_b[M,N]) and optionally C2 = act2(C); act kinds 1 = SwooshL, 2 = SwooshR; act2 = 3:
 *   C2 = C + resid_b instead (C itself without resid_b: a module's output and the residual
 *   stream after it from one launch).  K % 8 == 0, N % 4 == 0, rows
 *   16-byte aligned (else -2).  tile = 0 (from the shape) | 11 | 12 | 21 | 22: block tile
 *   (64 tm) x (64 tn); + 100 w: w persistent workgroups per CU;
 *   2000 + tm tn: the form that moves the weight pieces global -> LDS directly, + 200 (two-piece
 *   arithmetic only): 32-deep barrier intervals; 3000 + tm tn (22 | 21 | 12), + 200 (22 | 12): the
 *   same form on 8-wave workgroups, block tile (128 tm) x (64 tn), two-piece arithmetic, no Balancer
 *   epilogue (measured, not in the default plan: DESIGN 3i).  -2: a tile code the current
 *   arithmetic has not. */
/* ---- the arithmetic of every bf16 matrix-core GEMM of the library (s2t_gemm_x3p*, the weight-gradient /
 * NT / NN / batched kernels of csrc/gemm.hip): pieces per fp32 operand.
 *   3 ("bf16x3/6"): x = p0 + p1 + p2 exactly, six piece products per term: fp32-level error
 *       (<= 2e-6 of the result's max against fp64);
 *   2 ("bf16x2/3"): x ~ p0 + p1 (2^-18 relative), the products a0 b0 + a0 b1 + a1 b0: ~ 2^-17 per
 *       term -- torch's float32 matmul precision "high"; the reference trains under the looser
 *       "medium" (reference build_task.py:79, inference.py:58).
 * The arithmetic is a POLICY over four classes of product -- 0 F forward (y = x W^T, conv forward,
 * W0 @ x), 1 D data gradient (dx = g W and the batched products of backward), 2 W weight gradient
 * (dW += g^T x: every TN launch), 3 S statistics (Whiten's x^T x and its penalty product x dcov) --
 * consulted PER CALL: s2t_gemm_arith_set's value (2 | 3) for every class if one is pinned (0 =
 * unpinned); else the environment's S2T_GEMM_ARITH_F / _D / _W / _S for that class, else
 * S2T_GEMM_ARITH for all ("2" | "3" | "bf16x2" | "bf16x3" | "bf16x2/3" | "bf16x3/6"), else the
 * built-in default of the class.  s2t_gemm_arith_of(cls) = that value (cls outside 0..3: the base
 * value).  Entry points whose class is implied take it (TN launches: W; s2t_gemm_xtx: S; the 3x3
 * conv forward: F); s2t_gemm_x3p*, s2t_gemm_f32 (NT / NN) and s2t_gemm_f32_batched run in the class
 * the calling THREAD declared last with s2t_gemm_class_set(cls) (returns the previous one; -1 = none:
 * the base value) -- s2t_gemm_arith() is that class's value.  The weights' piece image (s2t_x3p_split)
 * is the same for both arithmetics: two-piece launches read its first two pieces. */
int s2t_gemm_arith(void);
int s2t_gemm_arith_of(int cls);
int s2t_gemm_class_set(int cls);
int s2t_gemm_arith_set(int arith);
typedef struct {
  long src_off;
  long dst_off;
  int N, K, ld, transposed;
  unsigned blk_begin;
  int pad_;
} S2tPlaneDesc;
long s2t_x3p_plane_elems(int N, int K);
long s2t_x3p_split_blocks(int N, int K);
int s2t_x3p_split(const float* base, const void* tab, int n, int total_blocks, unsigned short* dst,
                  void* stream);
int s2t_gemm_x3p(const float* A, long lda, const unsigned short* Bp, int N, int K, float* C, long ldc,
                 int M, const float* bias, const float* resid, long ldr, const float* act_src,
                 long ld_act, int act_kind, float* C2, long ldc2, int act2, const float* resid_b,
                 long ldrb, int tile, void* stream);
/* s2t_gemm_x3p for a data gradient through an activation with the Balancer on the activation's input
 * in the epilogue (the hidden Balancer of FeedforwardModule / ConvolutionModule / ConvNeXt,
 * model/encoder/zipformer.py:2372-2378, 2643-2695, model/layer/subsampling.py:106-132; update rule
 * model/layer/scaling.py:741-789): C = act'(act_src) (A Bm^T) (+ resid), then C += |C| (a[c] + b[c]
 * act_src) with a, b derived from bal_stats = 4096 floats: column sums [0..N) and sums of squares
 * [1024..1024+N) of act_src over its M rows (s2t_balancer_stats into a zeroed buffer); the call writes
 * the per-column a, b into [2048..) and [3072..) with one small launch before the product (N <= 1024).
 * s2t_balancer_coef is that small launch as its own call, for a caller that takes the statistics early
 * (forward pass, side stream): it then passes tile | S2T_X3P_BAL_COEF_READY.
 * Replaces the separate s2t_balancer_apply pass over the (M, N) gradient.  -2: shape / tile outside the
 * kernel's rules. */
#define S2T_X3P_BAL_COEF_READY (1 << 20)   /* or-ed into `tile`: s2t_balancer_coef already filled [2048..4096) */
int s2t_balancer_coef(float* bal_stats, int N, long rows, float min_mean, float max_mean, float min_rms,
                      float max_rms, float grad_scale, void* stream);
int s2t_gemm_x3p_bal(const float* A, long lda, const unsigned short* Bp, int N, int K, float* C, long ldc,
                     int M, const float* resid, long ldr, const float* act_src, long ld_act, int act_kind,
                     int tile, float* bal_stats, float min_mean, float max_mean, float min_rms,
                     float max_rms, float grad_scale, void* stream);

/* s2t_gemm_x3p with IMPLICIT operands: C = A' Bm^T (+ bias[N]) where row r of A' is `nseg` (<= 4) segments
 * of `seg` (multiple of 16) contiguous floats of the buffer A, segment s at amap(r) + segoff[s], and row
 * r of C lies at cmap(r) (cmap NULL: plain rows of ldc floats); K = seg * nseg; Bp = pieces of Bm (N, K).
 * map(r) = base + b sb + i sh + j sw for r = (b, i, j), b = r / hw, i = (r % hw) / w, j = r % w (all
 * offsets in floats, multiples of 4).  c_elems: elements of the mapped C buffer.  Serves the 3x3 /
 * stride-2 convolution of the conformer's Subsampling (model/encoder/conformer.py:47-57, 114-126)
 * without a patch matrix: forward (3 segments of 3 C floats per output position of the channel-last
 * map) and the data gradient (one launch per input-pixel parity class gathering 1 / 2 / 2 / 4 taps of
 * the zero-bordered output gradient and writing every second pixel), and the zipformer frontend's 32 -> 128
 * stride-(1, 2) convolution (model/layer/subsampling.py:277-319): forward and the data gradient by column
 * parity (even columns: one run of 2 Cout floats per kh, odd: Cout).  tile: 22 (default) 21 12 11; + 200:
 * 32-deep barrier intervals (two-piece arithmetic, seg a multiple of 32: else -2) -- a row's 128 bytes per
 * interval are one cache line fetched once. */
typedef struct S2tRowMap {
  int hw, w;
  long sb, sh, sw, base;
} S2tRowMap;
int s2t_gemm_x3p_map(const float* A, const S2tRowMap* amap, int seg, int nseg, const long* segoff,
                     const unsigned short* Bp, int N, float* C, long ldc, const S2tRowMap* cmap,
                     long c_elems, int M, const float* bias, int tile, void* stream);

/* ---- side stream for work off the critical path (csrc/streams.hip): the weight-gradient GEMMs
 * of backward overlap the data-gradient chain.  s2t_side_stream returns the library-owned stream;
 * s2t_stream_order(from, to) makes later work on `to` wait for the work enqueued so far on `from`. */
void* s2t_side_stream(void);
int s2t_stream_order(void* from, void* to);

/* ---- kernel-attached timing (bench.py's roofline figure).  s2t_prof_pair_arm hands a created
 * (start, stop) HIP event pair to the NEXT s2t_gemm_x3p launch of the calling thread, which is then
 * issued with hipExtLaunchKernelGGL: the pair holds the kernel's own begin / end times (what rocprof
 * reports), not the times of marker packets recorded around it.  s2t_prof_pair_consumed: 1 if a
 * launch took the armed pair, 0 if it was still armed (either way it is disarmed afterwards).
 * s2t_prof_pair_ms waits for the stop event and returns the elapsed milliseconds in *ms. */
/* the same for a SAMPLE of launches, counted inside s2t_gemm_x3p itself (so the launches the native
 * layer executor issues are sampled like the Python call sites'): every `every`-th launch between
 * begin and end carries its own event pair; end waits for them and reports how many were timed, their
 * total milliseconds and the algorithmic bytes / flops of exactly those launches
 * (4 M (N + K + operands N) + 2 P N K bytes for P pieces, 2 M N K flops).  s2t_gemm_x3p_calls: launches so far. */
int s2t_x3p_sample_begin(int every);
int s2t_x3p_sample_end(long* launches, double* total_ms, double* bytes, double* flops);
/* of the sample closed last: 4 M (N + K) + 2 P N K summed -- A, C and the weight pieces only, without the
 * epilogue operands and second outputs the figure above includes */
int s2t_x3p_sample_min_bytes(double* bytes_min);
long s2t_gemm_x3p_calls(void);
long s2t_linear_lt_calls(void);
int s2t_prof_pair_create(void** start, void** stop);
int s2t_prof_pair_arm(void* start, void* stop);
int s2t_prof_pair_consumed(void);
int s2t_prof_pair_ms(void* start, void* stop, float* ms);
int s2t_prof_pair_destroy(void* start, void* stop);


/* ---- element-wise helpers of the layer executor's library-GEMM path (csrc/zip_elem.hip):
 * out = a + b (out may alias an operand; 16-byte aligned); the parity sequence 0, 1, 0, ... of
 * s2t_balancer_bwd's two alternating accumulators, shared by every caller of the process. */
int s2t_add_f32(const float* a, const float* b, float* out, long n, void* stream);
int s2t_balancer_next_parity(void);

/* ---- batch of dense row-major fp32 products through hipBLASLt's strided-batch API (csrc/gemm_lib.hip):
 * the shapes of the nonlinear attention's products (model/encoder/zipformer.py:2468-2473 and their
 * gradients) that s2t_gemm_f32_batched refuses (rows that are not 16-byte multiples: T = 495, 62)
 * and the a^T b form.  mode 0: C_b[M,N] = A_b[M,K] B_b[N,K]^T; 1: C_b = A_b[M,K] B_b[K,N];
 * 2: C_b[M,N] = A_b[K,M]^T B_b[K,N].  Batch strides = rows * cols.  -2: no library algorithm. */
int s2t_bmm_lt(int mode, const float* A, const float* B, float* C, int batch, int M, int N, int K,
               void* workspace, long ws_bytes, void* stream);

/* ---- native per-layer executor (csrc/zip_layer.hip): ONE call per Zipformer2EncoderLayer forward and
 * one per backward issues every launch of the layer (model/encoder/zipformer.py:909-1338 under
 * loss.backward(); gradient shaping model/layer/scaling.py:741-789, 994-1028, 1153-1190).  The Python
 * side (speech2text_amd/zip_native.py) draws the layer's random decisions where the reference draws
 * them and passes them as `dec`; C++ only reads them.
 *   desc  : the layer's shapes, parameter / gradient addresses (views of the FlatStore), bf16 piece
 *           addresses of its weights (planes.py; NULL = library path) and module constants;
 *   call  : this call's shapes, inputs, masks, decisions, persistent scratch, switches;
 *   state : s2t_zip_layer_state_bytes() bytes of HOST memory, written by fwd, read by bwd;
 *   ws    : device workspace, >= s2t_zip_layer_ws_floats(desc, call, backward) floats; the forward's
 *           must stay alive (and untouched) until the backward has been enqueued.
 * Decision indices (dec[]): 0 balance_keys, 1 whiten_keys, 2 use positional scores, 3 score penalty
 * drawn; feed-forward i (hidden balancer, out whiten, post balancer): 4-6, 15-17, 23-25; nonlinear
 * attention (balancer, whiten1, whiten2, post balancer) 7-10; self_attn whiten 11, 19; conv module
 * (balancer1, balancer2, whiten) 12-14, 20-22; limit_param_value of bypass_mid 18, norm.log_scale 27,
 * bypass 28; layer balancer1 26, balancer2 29, whiten 30.
 * Whiten sites (s2t_zip_layer_info(state, 0, site) after bwd: 1 penalty applied, 0 below the limit,
 * -1 did not fire): 0 keys, 1 ff1, 2 nonlin whiten1, 3 nonlin whiten2, 4 self_attn1, 5 conv1, 6 ff2,
 * 7 self_attn2, 8 conv2, 9 ff3, 10 layer output.
 * Returns 0; -4 workspace too small; -5 a product's shape bucket has no plan entry yet (nothing was
 * launched: run the Python executor for this call, it times the shapes); bwd returns 1 when the
 * score penalty of this call is non-zero (it stopped before the attention-weights backward: write
 * dqkp / dpos -- addresses from s2t_zip_layer_info -- and call again with phase 2). */
#define S2T_ZL_NDEC 32
#define S2T_ZL_NWHITEN 11
typedef struct S2tZlLin {          /* nn.Linear, weight (N,K) row-major */
  const float* w;
  const float* b;                  /* NULL: no bias */
  float* gw;
  float* gb;
  const unsigned short* pf;        /* pieces for x W^T, or NULL */
  const unsigned short* pb;        /* pieces for g W, or NULL */
  int N, K;
} S2tZlLin;
typedef struct S2tZlBal {          /* Balancer.cfg(): scaling.py:792-902 */
  float min_mean, max_mean, min_rms, max_rms, grad_scale;
} S2tZlBal;
typedef struct S2tZlWh {           /* Whiten: scaling.py:1031-1095 */
  int groups;
  float limit, grad_scale;
} S2tZlWh;
typedef struct S2tZlFf {
  S2tZlLin in, out;
  S2tZlBal hidden;
  S2tZlWh out_wh;
  S2tZlBal post;                   /* balancer_ff2 / balancer_ff3 of the layer (unused for ff1) */
} S2tZlFf;
typedef struct S2tZlSa {
  S2tZlLin in, out;
  S2tZlWh wh;
} S2tZlSa;
typedef struct S2tZlConv {
  S2tZlLin in, out;
  S2tZlBal bal1, bal2;
  S2tZlWh wh;
  int K, causal;
  const float *wc, *bc, *wk, *bk, *scale;
  float *gwc, *gbc, *gwk, *gbk, *gscale;
} S2tZlConv;
typedef struct S2tZlNa {
  S2tZlLin in, out;
  S2tZlBal bal;
  S2tZlWh wh1, wh2;
  S2tZlBal post;                   /* balancer_na of the layer */
} S2tZlNa;
typedef struct S2tZlParam {
  const float* x;
  float* grad;
  float lo, hi;
} S2tZlParam;
typedef struct S2tZipLayerDesc {
  int D, H, qd, pd, pos_dim;
  S2tZlLin attn_in, attn_pos;
  S2tZlBal bal_keys;
  S2tZlWh wh_keys;
  S2tZlFf ff[3];
  S2tZlNa na;
  S2tZlSa sa[2];
  S2tZlConv cv[2];
  S2tZlParam byp_mid, byp, norm_bias, norm_ls;
  S2tZlBal bal1, bal2;
  S2tZlWh wh_out;
} S2tZipLayerDesc;
typedef struct S2tZlWhScratch {    /* persistent scratch of the Whiten statistics, per channel count */
  int C;
  float* acc;                      /* (C+1, C), zeroed once */
  float* ws;                       /* 4 + 2 C floats, zeroed once */
  const void* tab;                 /* S2tPlaneDesc of a (C,C) matrix for mode 1, or NULL */
  unsigned short* buf;             /* its piece buffer, or NULL */
  int blocks;
} S2tZlWhScratch;
typedef struct S2tZipLayerCall {
  int T, B, chunk_size;
  const float* x0;                 /* (T*B, D) layer input */
  const float* pos;                /* (2T-1, pos_dim) positional embedding */
  const unsigned char* k8;         /* (B,T) key padding mask or NULL */
  const unsigned char* a8;         /* (T,T) attention mask or NULL */
  const float* fm;                 /* (B,D) feature mask or NULL */
  float* out;                      /* forward: (T*B, D) layer output */
  const float* g;                  /* backward: gradient w.r.t. out */
  float* gx;                       /* backward: gradient w.r.t. x0 */
  int dec[S2T_ZL_NDEC];
  float* bal_ws;                   /* s2t_balancer_bwd_workspace_floats(), zeroed once */
  float* layer_acc;                /* 3 D + 4 floats, zeroed once (per D) */
  S2tZlWhScratch wh[8];
  int nwh;
  void* lt_ws;
  long lt_ws_bytes;
  int x3p_on, x3p_tile;
  float x3p_margin;
  int whiten_x3p;                  /* Whiten backward: 2 = dcov + its pieces in forward, product on the pre-split-weight kernel (round 6); 0 = the NN kernel, three launches */
  int conv_w_side, conv_fused, stats_side, wgrad_side, bmm_own;
  int bal_epi;                     /* hidden Balancers in the dgrad epilogue (s2t_gemm_x3p_bal) */
  int whiten_sq;                   /* Whiten's norms in the x dcov product's epilogue (s2t_gemm_f32_sq) */
  int bal_fwd_side;                /* firing Balancers' column statistics taken in forward on the side stream (round 6) */
  int whiten_fwd_pg;               /* Whiten's penalty product x dcov taken in forward on the statistics' stream (round 6) */
} S2tZipLayerCall;
long s2t_zip_layer_state_bytes(void);
long s2t_zip_layer_ws_floats(const S2tZipLayerDesc* desc, const S2tZipLayerCall* call, int backward);
int s2t_zip_layer_plans_missing(const S2tZipLayerDesc* desc, int T, int B);
int s2t_zip_layer_fwd(const S2tZipLayerDesc* desc, const S2tZipLayerCall* call, void* state, float* ws,
                      long ws_floats, void* stream, void* side);
int s2t_zip_layer_bwd(const S2tZipLayerDesc* desc, const S2tZipLayerCall* call, void* state, float* ws,
                      long ws_floats, int phase, void* stream, void* side);
long s2t_zip_layer_info(const void* state, int what, int idx);
void* s2t_zip_layer_error(void);
/* the timings zip_kernels.lt_matmul took for a shape bucket {mode, half-octave of the rows, N, K} of a
 * Linear: library ms, own-kernel ms (< 0: none) and its tile -- what the executor's plan is made of.
 * mode = (0 forward | 1 data gradient) | (the arithmetic the bucket was timed under, s2t_gemm_arith()) << 4 */
int s2t_zl_plan_put(int mode, int half_oct, int N, int K, double t_lib_ms, double t_own_ms, int tile);
int s2t_zl_plan_clear(void);
long s2t_zl_plan_count(void);

#ifdef __cplusplus
}
#endif
#endif /* S2T_MI355_H_ */
