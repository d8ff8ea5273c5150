/* leod_hip.h -- C ABI of libleod_hip.so: the hand-written gfx950 (MI355X / CDNA4) kernels behind the
 * LEOD hot path (RVT recurrent backbone fwd/bwd, YOLOX head + SimOTA + losses, pseudo-label NMS).
 *
 * The reference (Wuziyi616/LEOD) is 100 % Python and has no FFI; each entry point below replaces the
 * ATen / torchvision call sequence of the cited reference lines (paths relative to the reference
 * root).  The reference-side binding is the ctypes stub in leod_amd/_lib.py.
 *
 * Conventions
 *   - plain C symbols, raw DEVICE pointers, explicit sizes; no torch types, no global state;
 *   - every call only enqueues work on `stream` (hipStream_t passed as void*): it never allocates,
 *     never synchronises and is re-entrant across streams; scratch is caller-provided;
 *   - return 0 on success, <0 on error: -1 bad argument, -2 launch failure, -3 unsupported shape;
 *   - activations are fp32 "rows": a row is one token/pixel of a channels-last (NHWC) map,
 *     M = B*H*W rows, contiguous channels.  "+=" outputs accumulate (parameter gradients).
 */
#ifndef LEOD_HIP_H
#define LEOD_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef void* leod_stream_t; /* hipStream_t */

const char* leod_version(void);

/* Precision of the contractions (process-wide; set before the first step).
 * 0: fp32 end to end -- v_mfma_f32_16x16x4_f32, bitwise an fp32 fmaf chain, the mode the parity tests pin against the fp32 oracle.
 * 2 ("16f"): the reference's mixed precision (Lightning precision=16 = fp16 autocast + GradScaler, train.py:236-243): the FORWARD
 *    contractions (GEMM, conv, attention, ConvLSTM, stem) take fp16 operands for v_mfma_f32_16x16x16_f16 / 16x16x32_f16 with fp32
 *    accumulation, and the 16-bit tensors the forward pass stores (qkv, attention output, MLP hidden, LSTM gates) are fp16; the GRADIENT
 *    contractions take bf16 operands (fp32's exponent range: no loss scaler) and gradient rows (dqkv, du, dO, gate gradients) are bf16.
 * 1 ("bf16", Lightning bf16-mixed): bf16 operands and bf16 rows in both directions.
 * In both 16-bit modes LayerNorm / BatchNorm statistics, softmax, the residual stream, LSTM state, SimOTA cost, losses and the optimiser
 * stay fp32.  Wherever a comment below says "precision mode bf16" it means "modes 1 and 2"; the parameters called *_bf16 that describe
 * FORWARD-stored activation rows (qkv, the attention output) carry fp16 rows in mode 2. */
int leod_set_precision(int mode);
int leod_get_precision(void);

/* ---- backbone: MaxViT block pieces (models/layers/maxvit/maxvit.py) ------------------------------ */

/* out[M,N] = LN(x)[M,K] W[N,K]^T + bias (LN skipped when ln_w == NULL); out_act (optional) = gelu_erf(out);
 * stats_out (optional) [M,2] = (mean, rstd).  Replaces norm1->qkv (maxvit.py:267,347) and norm2->fc1->GELU (:110-118,269). */
int leod_ln_linear_fwd(const float* x, long ldx, const float* ln_w, const float* ln_b, float eps, const float* W,
                       const float* bias, float* out, float* out_act, float* stats_out, int M, int N, int K,
                       leod_stream_t stream);

/* t = a[M,K] W[N,K]^T + bias ; tout (optional) = t ; out = res + gamma * t.
 * Replaces proj / fc2 followed by LayerScale and the residual add (maxvit.py:51-53,268-269,353). */
int leod_linear_lsres_fwd(const float* a, const float* W, const float* bias, const float* gamma, const float* res,
                          float* out, float* tout, int M, int N, int K, leod_stream_t stream);

/* Partition attention core on the interleaved qkv rows [M,3C] of an NHWC map (head h owns columns
 * [h*3d,(h+1)*3d) = q|k|v): out[M,C] = softmax(q k^T d^-1/2) v within each window (window=1) or grid
 * (window=0) partition of ph x pw tokens; lse (optional) [M,heads].  Replaces window/grid_partition + SelfAttentionCl
 * core + *_reverse (maxvit.py:252-265,273-304,347-352). */
int leod_partition_attn_fwd(const float* qkv, float* out, float* lse, int B, int H, int W, int C, int heads, int ph,
                            int pw, int window, int qkv_bf16, leod_stream_t stream);
/* dqkv[M,3C] from dout[M,C]; dsum [M,heads] scratch. */
int leod_partition_attn_bwd(const float* qkv, const float* dout, const float* lse, float* dsum, float* dqkv, int B,
                            int H, int W, int C, int heads, int ph, int pw, int window, int qkv_bf16, int dqkv_bf16,
                            leod_stream_t stream);
/* Precision mode bf16: q, k, v only ever enter bf16 MFMAs and dqkv's consumers feed bf16 MFMAs, so both tensors may live in HBM as
 * bf16 (qkv_bf16 / dqkv_bf16 above: the pointers then address bf16 elements) where the LDS attention kernels cover the geometry --
 * 1 from this query; leod_ln_linear_bf16_fwd produces the bf16 qkv rows. */
int leod_partition_attn_16bit_ok(int B, int H, int W, int C, int heads, int ph, int pw);
/* Bit 1 of qkv_bf16 in the two attention calls above: the attention output `out` (forward) is written / its gradient `dout` (backward)
 * is read as bf16 rows [M,C] -- valid where leod_partition_attn_o16_ok returns 1 (bf16-tile kernels).  leod_attn_block_o16_ok adds the
 * conditions of the block's other consumers (leod_linear_lsres_bf16_fwd, bit 1 of dy_bf16 in leod_linear_dgrad = dx written as bf16 rows,
 * bit 1 of dy_bf16 in leod_linear_wgrad = x holds bf16 rows): the reference's autocast holds both tensors in 16 bits as well
 * (maxvit.py:185-270 under train.py:236-243). */
int leod_partition_attn_o16_ok(int B, int H, int W, int C, int heads, int ph, int pw);
int leod_attn_block_o16_ok(int B, int H, int W, int C, int heads, int ph, int pw);
int leod_linear_lsres_bf16_fwd(const void* a16, const float* W, const float* bias, const float* gamma, const float* res, float* out,
                               int M, int N, int K, leod_stream_t stream);
/* out16[M,N] = bf16(LN(x) W^T + bias), stats_out [M,2]; -3 unless the row-streaming kernel covers (M, N, K) in precision mode bf16. */
int leod_ln_linear_bf16_fwd(const float* x, const float* ln_w, const float* ln_b, float eps, const float* W, const float* bias,
                            void* out16, float* stats_out, int M, int N, int K, leod_stream_t stream);

/* Precision mode bf16, stages 1-2 of the MLP (maxvit.py:110-118): the hidden pre-activation u = LN(x) W1^T + b1 is stored ONCE,
 * as fp16 [M,N] (as the reference does under autocast, train.py:236-243; clamped to the fp16 range); the three consumers below
 * apply GELU / GELU' while loading it.  leod_ln_linear_gelu16_fwd returns -3 where the row-streaming kernel does not cover (M, N, K): callers then use
 * leod_ln_linear_fwd and its fp32 (u, gelu(u)) pair.  u16: device pointer to fp16 elements. */
int leod_ln_linear_gelu16_fwd(const float* x, const float* ln_w, const float* ln_b, float eps, const float* W, const float* bias,
                              void* u16, float* stats_out, int M, int N, int K, leod_stream_t stream);
int leod_linear_lsres_gelu16_fwd(const void* u16, const float* W, const float* bias, const float* gamma, const float* res, float* out,
                                 int M, int N, int K, leod_stream_t stream);
int leod_linear_dgrad_gelu16(const float* dy, const float* kscale, const float* W, const void* u16, void* dx, int M, int N, int K,
                             int out_bf16, leod_stream_t stream);
int leod_linear_wgrad_gelu16(const float* dy, long lddy, const void* u16, float* dW, float* dbias, int M, int N, int K,
                             leod_stream_t stream);

/* n <= 4 Linear weight gradients of ONE row count M in one preparation, one contraction and one reduce launch (LDS-DMA kernel with a problem
 * table): dW_k [N_k, K_k] += dy_k^T X_k, dbias_k += colsum(dy_k).  The four weight gradients of an attention block, which its backward issues
 * together (maxvit.py:110-118,252-270).  Host arrays of length n.  dy_fmt: 0 fp32 rows, 1 bf16 rows; x_fmt: 0 fp32 rows, 1 fp32 rows through
 * LayerNorm (stats_k [M,2], ln_w_k, ln_b_k), 2 fp16 pre-activation through GELU, 3 bf16 rows, 4 fp16 rows; dense rows.
 * -3: not coverable (precision mode f32, M outside 8192 .. 60 000 / 400 000 or not a multiple of 64, widths of different tile classes):
 * nothing was launched, run the problems singly. */
int leod_linear_wgrad_group(int n, const void* const* dy, const int* dy_fmt, const void* const* x, const int* x_fmt,
                            const float* const* stats, const float* const* ln_w, const float* const* ln_b, float* const* dW,
                            float* const* dbias, int M, const int* N, const int* K, leod_stream_t stream);

/* The whole MLP of a MaxViT block in one row-streaming launch (maxvit.py:110-118, 268-269), precision mode bf16, K = 48 / H = 192 (stage 1
 * of RVT-S / -T), M >= 16384:  z[M,K] = y + g2 * (gelu(LN(y) W1^T + b1) W2^T + b2) with the hidden in registers (csrc/k_mlp.hip).
 * u16 [M,H] fp16 + stats [M,2] (both or neither): what the backward pass reads back (training).  -3: not covered -- the caller runs
 * leod_ln_linear_gelu16_fwd + leod_linear_lsres_gelu16_fwd. */
int leod_mlp_fwd_fused(const float* y, const float* ln_w, const float* ln_b, float eps, const float* W1, const float* b1, const float* W2,
                       const float* b2, const float* g2, float* out, void* u16, float* stats, int M, int H, int K, leod_stream_t stream);

/* Backward of that MLP along the activation path, one launch: u is recomputed from y, du = ((dz g2) W2) gelu'(u) is written as bf16 rows
 * (du16 [M,H], optional: the fc1 weight gradient reads it), dy = dz + LayerNorm-backward(du W1), dgamma / dbeta [K] += (norm2).
 * stats [M,2] = the (mean, rstd) of the forward pass.  Replaces leod_linear_dgrad_gelu16 + leod_linear_dgrad_lnbwd where it applies
 * (same coverage as leod_mlp_fwd_fused; -3 otherwise). */
int leod_mlp_bwd_dgrad_fused(const float* dz, const float* y, const float* stats, const float* ln_w, const float* ln_b, const float* W1,
                             const float* b1, const float* W2, const float* g2, float* dy, void* du16, float* dgamma, float* dbeta, int M,
                             int H, int K, leod_stream_t stream);

/* Fused ConvLSTM cell, DWSConvLSTM2d.forward with dws_conv=False (models/layers/rnn.py:37-70):
 * gates = [x|h_prev] W[4C,2C]^T + b, (f,i,o)=sigmoid, g=tanh, c=f*c_prev+i*g, h=o*tanh(c).
 * h_prev/c_prev NULL = zero state; gates_out (optional) [M,4,C] post-activation gates. */
int leod_convlstm_fwd(const float* x, const float* h_prev, const float* c_prev, const float* W, const float* bias,
                      float* h_out, float* c_out, float* gates_out, int M, int C, leod_stream_t stream);
/* dgates[M,4,C] (pre-activation), dc_prev (optional) from dh + dh2 (both optional: gradient of h_t from the layers
 * above and from timestep t+1) and dc_next (optional). */
int leod_convlstm_gates_bwd(const float* dh, const float* dh2, const float* dc_next, const float* gates, const float* c_prev,
                            const float* c_t, float* dgates, float* dc_prev, int M, int C, leod_stream_t stream);

/* The ConvLSTM recurrence of a whole sequence in ONE launch per direction (same cell as leod_convlstm_fwd, unrolled over the L
 * timesteps of modules/detection.py:188-226): one workgroup carries 16 rows through all T timesteps with its slice of the weights
 * resident in registers.  leod_convlstm_seq_mode(C): 1 = xin is x_seq [T,M,C] (fused [x|h] contraction), 2 / 3 = xin is the
 * time-batched projection gx [T,M,4C] = x W_x^T + b computed by the caller, 0 = not available for this C / precision mode
 * (callers then loop leod_convlstm_fwd).  hbuf, cbuf [T+1,M,C]: slot 0 = incoming state (zero_state: taken as zeros, not read),
 * slots 1..T are written; gates_out [T,M,4,C] optional -- without it (inference: no backward pass will read the history) only slot T of cbuf
 * is written. */
int leod_convlstm_seq_mode(int C);
/* mode 3 (C = 256 / 384 in precision mode bf16: the weight slice of a wave does not fit its registers): like mode 2, and the waves
 * stream their MFMA B fragments from a fragment-ordered bf16 copy of W_h -- leod_convlstm_seq_pack writes it (once per step, shared by
 * forward and backward) into a buffer of leod_convlstm_seq_pack_bytes(C) bytes, passed as wpack (NULL in modes 1 / 2). */
long leod_convlstm_seq_pack_bytes(int C);
int leod_convlstm_seq_pack(const float* W, void* wpack, int C, leod_stream_t stream);
int leod_convlstm_seq_fwd(const float* xin, int x_is_projection, float* hbuf, float* cbuf, const float* W, const float* bias,
                          float* gates_out, const void* wpack, int M, int C, int T, int zero_state, int gates16, leod_stream_t stream);
/* Backward through time of the same: dh_seq [T,M,C] (optional) gradients of every h_t from above, dc_last [M,C] (optional);
 * writes dgates_out [T,M,4C] (pre-activation; dx = dgates W_x and the weight gradient are one GEMM each over all T*M rows)
 * and optionally dh0 / dc0 [M,C].  LEOD_ERR_UNSUPPORTED (-3) when the weight slice does not fit the registers. */
int leod_convlstm_seq_bwd(const float* dh_seq, const float* dc_last, const float* gates, const float* cbuf, const float* W,
                          float* dgates_out, float* dh0, float* dc0, const void* wpack, int M, int C, int T, int zero_state,
                          int gates16, leod_stream_t stream);
/* gates16 (both calls above; precision mode bf16 only, leod_convlstm_seq_gates16_ok(C) != 0): gates_out / gates point to fp16 storage in
 * the sequence kernels' own lane-linear layout (T x ceil(M / 16) * 16 x 4C halfs, written and read by these two calls only), dgates_out to
 * bf16 rows [T][M][4C] -- the reference's autocast holds the gates in 16 bits as well (rnn.py:58-62 under train.py:236-243). */
int leod_convlstm_seq_gates16_ok(int C);

/* dy_bf16 (here and in leod_linear_wgrad / leod_linear_dgrad_lnbwd): dy points to bf16 elements -- in precision mode bf16 the wide
 * gradients du (out_bf16 of leod_linear_dgrad_gelu16) and dqkv are stored as the bf16 their consumers feed to the MFMAs anyway.
 * dx (=|+=) (dy[M,N]*kscale[N]) W[N,K] ; optional: multiply by gelu'(aux_u[M,K]); route columns >= nsplit to dx2;
 * colsum[K] += column sums of the result; dres [M,K] (optional, leading dimension lddx): dx = dres + result (the second gradient source of a
 * residual branch, so that no separate add kernel runs).  Autograd of the Linear layers above. */
int leod_linear_dgrad(const float* dy, long lddy, const float* kscale, const float* W, float* dx, long lddx, float* dx2,
                      long lddx2, int nsplit, const float* aux_u, float* colsum, int accumulate, const float* dres, int M, int N,
                      int K, int dy_bf16, leod_stream_t stream);
/* Workspace of the weight-gradient calls below for launches on `stream` (precision mode bf16: the per-workgroup partial tiles that a second
 * kernel adds up -- 1024 workgroups adding to the same dW addresses with atomics were 130 us of a 200 us launch): leod_workspace_bytes()
 * bytes, 16-byte aligned, owned by the caller and kept alive until replaced (ws == NULL withdraws it).  Without a registered workspace
 * the library allocates one per stream on first use, or falls back to the atomic epilogue while the stream is being captured.
 * (No counterpart in the reference: torch's autograd owns the cuBLAS workspaces there.) */
long leod_workspace_bytes(void);
int leod_set_workspace(void* ws, long bytes, leod_stream_t stream);
/* dW[N,K] += dy^T X ; dbias[N] += colsum(dy) ; X = x, LN(x) (stats, ln_w, ln_b) or [x | x2] (K1 = cols of x). */
int leod_linear_wgrad(const float* dy, long lddy, const float* x, long ldx, const float* stats, const float* ln_w,
                      const float* ln_b, const float* x2, long ldx2, int K1, float* dW, float* dbias, int M, int N,
                      int K, int dy_bf16, leod_stream_t stream);

/* LayerNorm over channels (eps 1e-5), maxvit.py:172-178 and its autograd. */
int leod_layernorm_fwd(const float* x, const float* w, const float* b, float* y, float* stats, int M, int C, float eps,
                       leod_stream_t stream);
int leod_layernorm_bwd(const float* dn, const float* x, const float* stats, const float* w, const float* dres, float* dx,
                       float* dw, float* db, int M, int C, float eps, leod_stream_t stream);
/* LayerScale autograd: dt = gamma*dz ; dgamma += sum_m dz*t (maxvit.py:45-53). */
int leod_layerscale_bwd(const float* dz, const float* t, const float* gamma, float* dt, float* dgamma, int M, int C,
                        leod_stream_t stream);
/* LayerScale autograd without the stored pre-scale tensor t = h W^T + b (maxvit.py:45-53, :268-269): from the UN-scaled
 * weight gradient G[N,K] = dz^T h and s[N] = colsum(dz) of the Linear in front of the LayerScale:
 * dW += diag(gamma) G ; db += gamma*s ; dgamma[n] += sum_k W[n,k] G[n,k] + b[n] s[n]. */
int leod_layerscale_finalize(const float* W, const float* b, const float* gamma, const float* G, const float* s, float* dW,
                             float* db, float* dgamma, int N, int K, leod_stream_t stream);

/* ---- convolutions (implicit GEMM, NHWC) --------------------------------------------------------- */

/* Stem conv of stage 1 straight from the raw NCHW event tensor (uint8 or fp32, H x W unpadded; reads outside are
 * the zero padding of utils/padding.py:32-58 up to Hp x Wp): ConvDownsampling_Cf2Cl.conv, maxvit.py:160-176. */
int leod_stem_conv_fwd(const void* x, int x_is_u8, const float* w, float* y, int B, int Cin, int H, int W, int Hp, int Wp,
                       int N, int ks, int stride, int pad, leod_stream_t stream);
int leod_stem_conv_wgrad(const float* dy, const void* x, int x_is_u8, float* dw, int B, int Cin, int H, int W, int Hp,
                         int Wp, int N, int ks, int stride, int pad, leod_stream_t stream);
/* y = conv(x NHWC, w[N,Cin,ks,ks]) (+bias); colstats (optional) [stat_rep,2,N] double += (sum, sumsq) for training BatchNorm,
 * spread over stat_rep replicas (power of two, 0/1 = one copy; the caller zero-fills, leod_bn_silu_fwd folds them);
 * bn_w != NULL: eval BatchNorm folded + SiLU (network_blocks.py:29-54; yolo_pafpn.py:109-140; yolo_head.py:208-222).
 * wpack (optional scratch, N*Cin*ks*ks floats): the call first writes a K-contiguous copy of w there and contracts
 * against that (the native [N][Cin][ks][ks] layout strides every weight float4 over 36 bytes).  wpack_valid != 0: wpack already
 * holds the packed copy this very call (same weights, geometry, direction, precision mode) wrote earlier and w has not changed since --
 * the pack launch is skipped (the weights change once per optimiser step, the PAFPN / head convs run 40 packs per step otherwise). */
int leod_conv_nhwc_fwd(const float* x, const float* w, const float* bias, float* y, double* colstats, int stat_rep, const float* bn_w,
                       const float* bn_b, const float* bn_rm, const float* bn_rv, float bn_eps, int B, int H, int W,
                       int Cin, int N, int ks, int stride, int pad, float* wpack, int wpack_valid, leod_stream_t stream);
int leod_conv_nhwc_dgrad(const float* dy, const float* w, float* dx, int accumulate, int B, int H, int W, int Cin, int N,
                         int ks, int stride, int pad, float* wpack, int wpack_valid, leod_stream_t stream);
/* dw[N,Cin,ks,ks] += weight gradient (dbias optional).  ws: scratch of leod_conv_nhwc_wgrad_workspace_floats(...) floats (may be
 * NULL when that is 0; with a workspace the 3x3 / stride-1 gradients of the bf16 mode are reduced without atomics). */
long leod_conv_nhwc_wgrad_workspace_floats(int B, int H, int W, int Cin, int N, int ks, int stride, int pad, int has_bias);
int leod_conv_nhwc_wgrad(const float* dy, const float* x, float* dw, float* dbias, float* ws, int B, int H, int W, int Cin, int N,
                         int ks, int stride, int pad, leod_stream_t stream);

/* Grouped 3x3 / stride-1 / pad-1 convolutions: n <= 8 independent problems of ONE (Cin, Cout) geometry in one launch -- the convs of equal
 * depth in the cls / reg towers of the three head levels (yolo_head.py:61-145,208-222).  All array arguments are HOST arrays of length n
 * (device pointers / sizes per problem); wpack[k] / wpack_valid[k] as in leod_conv_nhwc_fwd.  forward: y_k = conv(x_k, w_k) with the
 * BatchNorm (sum, sumsq) of y_k into colstats[k] ([stat_rep[k]][2][Cout] doubles, zeroed; NULL: none).  dgrad: dx_k (+)= the input
 * gradient from dy_k [B,H,W,N]; problems of one call must write different dx buffers.  -3: not coverable, run the problems singly. */
int leod_conv3x3_group_supported(int n, const int* H, const int* W, int Cin, int Cout);   /* 1: the forward call covers these maps (dgrad: ask with (Cout, Cin)) */
int leod_conv3x3_group_fwd(int n, const float* const* x, const float* const* w, float* const* y, double* const* colstats,
                           const int* stat_rep, void* const* wpack, const int* wpack_valid, const int* B, const int* H, const int* W,
                           int Cin, int Cout, leod_stream_t stream);
int leod_conv3x3_group_dgrad(int n, const float* const* dy, const float* const* w, float* const* dx, const int* accumulate,
                             void* const* wpack, const int* wpack_valid, const int* B, const int* H, const int* W, int Cin, int N,
                             leod_stream_t stream);
/* weight gradients of the same problems: dw_k [N,Cin,3,3] += ... in one launch + one reduce launch; ws[k]: scratch of
 * leod_conv3x3_group_wgrad_workspace_floats(B[k], H[k], W[k], Cin, N) floats (0: not coverable -> -3 from the call) */
long leod_conv3x3_group_wgrad_workspace_floats(int B, int H, int W, int Cin, int N);
int leod_conv3x3_group_wgrad(int n, const float* const* dy, const float* const* x, float* const* dw, float* const* ws, const int* B,
                             const int* H, const int* W, int Cin, int N, leod_stream_t stream);

/* Depthwise convolution (groups == channels) on NHWC maps, w[C,1,ks,ks]: the depthwise half of DWConv (network_blocks.py:57-76, selected
 * by `depthwise` at yolo_pafpn.py:37 / yolo_head.py:52) and conv3x3_dws of the ConvLSTM (models/layers/rnn.py:20-30,50-55).  Same epilogue
 * options as leod_conv_nhwc_fwd: +bias, colstats [stat_rep,2,C] double += (sum, sumsq), or eval BatchNorm folded + SiLU.  C % 4 == 0.
 * dgrad: dx[B,H,W,C] (+= when accumulate) from dy[B,Ho,Wo,C]; wgrad: dw[C,1,ks,ks] += , dbias[C] += (optional).  fp32 in every mode. */
int leod_dwconv_nhwc_fwd(const float* x, const float* w, const float* bias, float* y, double* colstats, int stat_rep, const float* bn_w,
                         const float* bn_b, const float* bn_rm, const float* bn_rv, float bn_eps, int B, int H, int W, int C,
                         int ks, int stride, int pad, leod_stream_t stream);
int leod_dwconv_nhwc_dgrad(const float* dy, const float* w, float* dx, int accumulate, int B, int H, int W, int C, int ks,
                           int stride, int pad, leod_stream_t stream);
int leod_dwconv_nhwc_wgrad(const float* dy, const float* x, float* dw, float* dbias, int B, int H, int W, int C, int ks,
                           int stride, int pad, leod_stream_t stream);

/* BatchNorm2d (batch statistics) + SiLU on rows and its autograd (network_blocks.py:47-51).  `count` = rows that
 * entered colstats/sums.  SyncBatchNorm: with count_dev (device scalar) the row count is count * count_dev[0] -- pass count =
 * rows per image and count_dev = images over all ranks (one all-reduced scalar per step), colstats/sums all-reduced.
 * sums of the backward pair: zero-filled double [rep][2][N]; the reduce kernel's workgroups spread their closing atomics over the rep
 * copies (rep >= 1, chosen by the caller from the row count), the apply kernel folds them. */
int leod_bn_silu_fwd(const float* z, const double* colstats, int stat_rep, const float* w, const float* b, float* y, float* save_mean,
                     float* save_rstd, float* run_mean, float* run_var, int M, int N, double count,
                     const double* count_dev, float eps, float momentum, leod_stream_t stream);
int leod_bn_silu_bwd_reduce(const float* dy, const float* z, const float* mean, const float* rstd, const float* w,
                            const float* b, double* sums, int rep, int M, int N, int lddy, leod_stream_t stream);
int leod_bn_silu_bwd_apply(const float* dy, const float* z, const float* mean, const float* rstd, const float* w,
                           const float* b, const double* sums, int rep, float* dz, float* dw, float* db, int M, int N, double count,
                           const double* count_dev, int lddy, leod_stream_t stream);

/* The three BatchNorm + SiLU launches above for n <= 8 layers of ONE channel count at once (the layers of equal depth over the head levels and
 * branches): every array argument is a HOST array of length n holding what the single call takes for layer k.  run_mean / run_var /
 * count_dev / lddy / dw / db may be NULL as arrays (none for any layer) or hold NULL / 0 entries. */
int leod_bn_silu_fwd_group(int n, const float* const* z, const double* const* colstats, const int* stat_rep, const float* const* w,
                           const float* const* b, float* const* y, float* const* save_mean, float* const* save_rstd, float* const* run_mean,
                           float* const* run_var, const int* M, int N, const double* count, const double* const* count_dev, float eps,
                           const float* momentum, leod_stream_t stream);
int leod_bn_silu_bwd_reduce_group(int n, const float* const* dy, const float* const* z, const float* const* mean, const float* const* rstd,
                                  const float* const* w, const float* const* b, double* const* sums, const int* rep, const int* M, int N,
                                  const int* lddy, leod_stream_t stream);
int leod_bn_silu_bwd_apply_group(int n, const float* const* dy, const float* const* z, const float* const* mean, const float* const* rstd,
                                 const float* const* w, const float* const* b, double* const* sums, const int* rep, float* const* dz,
                                 float* const* dw, float* const* db, const int* M, int N, const double* count,
                                 const double* const* count_dev, const int* lddy, leod_stream_t stream);

/* ---- YOLOX head tail (models/detection/yolox/models/yolo_head.py) --------------------------------- */

/* cls/reg/obj 1x1 prediction convs of one level + grid decode (:216-222,289-332): out_train = decoded boxes + logits,
 * out_infer = decoded boxes + sigmoid, both [B, A, 5+nc] written at anchor offset a0. */
int leod_head_pred_fwd(const float* cls_feat, const float* reg_feat, const float* cls_w, const float* cls_b,
                       const float* reg_w, const float* reg_b, const float* obj_w, const float* obj_b, float* out_train,
                       float* out_infer, int B, int h, int w, int Hd, int nc, int stride, int a0, int A, leod_stream_t stream);
/* gscale (optional): device scalar multiplied into d_raw (the seed gradient of the loss). */
int leod_head_pred_bwd(const float* d_raw, const float* cls_feat, const float* reg_feat, const float* cls_w,
                       const float* reg_w, const float* obj_w, float* d_cls_feat, float* d_reg_feat, float* d_cls_w,
                       float* d_cls_b, float* d_reg_w, float* d_reg_b, float* d_obj_w, float* d_obj_b, const float* gscale,
                       int B, int h, int w, int Hd, int nc, int a0, int A, leod_stream_t stream);

/* SimOTA (get_assignments / get_assignments_w_ignore / simota_matching, :606-774, :974-1148) for a whole batch:
 * outputs [B,A,5+nc] decoded boxes + logits, labels [B,Nmax,7] = (cls,cx,cy,w,h,obj,cls_conf) zero padded.
 * totals[3] int (caller-zeroed): sum num_fg, sum num_gt, status bit0 = some gt had no candidate anchor.
 * No limit on Nmax (boxes per frame) or A: the per-gt and per-candidate arrays live in LDS when they fit (Nmax <= 128,
 * candidates <= 10240) and in the workspace otherwise.  workspace = leod_simota_workspace_floats(B, Nmax, A) floats. */
long leod_simota_workspace_floats(int B, int Nmax, int A);
int leod_simota_assign(const float* outputs, const float* labels, float* workspace, unsigned char* fg_mask,
                       unsigned char* ignore_mask, int* matched_row, int* matched_valid_idx, float* pred_iou,
                       int* num_fg_img, int* totals, int B, int Nmax, int nc, int nlv, const int* hs, const int* ws,
                       const int* strides, float ignore_label, leod_stream_t stream);
/* Loss assembly (:547-597, losses.py:18-85) + gradient wrt the raw prediction-conv outputs (d_raw, optional).
 * sums[3] double caller-zeroed; losses[6] = loss, iou_loss, conf_loss, cls_loss, l1_loss(0), num_fg/num_gt. */
int leod_yolox_loss(const float* outputs, const float* labels, const unsigned char* fg_mask,
                    const unsigned char* ignore_mask, const int* matched_row, const float* pred_iou, const int* totals,
                    double* sums, float* losses, float* d_raw, int B, int Nmax, int nc, int nlv, const int* hs,
                    const int* ws, const int* strides, int focal, float reg_weight, float obj_weight, float cls_weight,
                    float grad_scale, leod_stream_t stream);
/* leod_yolox_loss under ``bbox_loss_weighting`` (_get_bbox_loss_weight, yolo_head.py:358-381; normalisation :550-553 / :928-931):
 * label_w [B,Nmax] = the configured expression of each label row's obj / cls / obj*cls confidence; the IoU and class terms of a
 * foreground anchor and their gradients are scaled by label_w[its box] / (mean of that over the batch's foreground anchors).
 * wsum[1] double caller-zeroed (the sum behind the mean). */
int leod_yolox_loss_weighted(const float* outputs, const float* labels, const unsigned char* fg_mask,
                             const unsigned char* ignore_mask, const int* matched_row, const float* pred_iou, const int* totals,
                             const float* label_w, double* wsum, double* sums, float* losses, float* d_raw, int B, int Nmax, int nc,
                             int nlv, const int* hs, const int* ws, const int* strides, int focal, float reg_weight,
                             float obj_weight, float cls_weight, float grad_scale, leod_stream_t stream);
/* ``ignore_bg_k`` (_get_highest_score_mask, yolo_head.py:335-356, applied at :541-542): marks in ignore_mask [B,A] the
 * int(#background anchors * k) highest objectness logits (outputs[...,4]) among each image's background anchors (fg_mask == 0), so
 * that leod_yolox_loss leaves them out of the objectness term.  Does nothing when any label row carries ignore_label: such a
 * batch takes the reference's get_losses_w_ignore, which has no such step.  0 < k <= 1; ties at the threshold go to the lowest
 * anchor indices (torch.topk leaves that order unspecified). */
int leod_bg_topk_ignore(const float* outputs, const float* labels, const unsigned char* fg_mask, unsigned char* ignore_mask,
                        int B, int Nmax, int A, int nc, double k, float ignore_label, leod_stream_t stream);

/* ---- post-processing (models/detection/yolox/utils/boxes.py:32-86; modules/utils/ssod.py:40-188) -- */

/* postprocess + torchvision-semantics batched NMS for B images at once.  nc>0: pred [B,A,5+nc] (cx,cy,w,h,obj,cls..),
 * boxes rewritten IN PLACE to xyxy; nc==0: pred [B,A,7] already (xyxy,obj,cls_conf,cls_id) (TTA merge).
 * det_out [B,max_det,7] in NMS order, det_cnt[B].  One workgroup per image keeps the candidates in LDS: all A anchors when
 * they fit (A <= 5040: Gen1, Gen4), else 4096 candidates; an image with more boxes above conf_thre than that is sorted and
 * suppressed in its slice of ``workspace`` (leod_postprocess_nms_workspace_bytes(B, A) bytes, 0 when never needed) -- the
 * reference has no candidate limit (boxes.py:53-80).  workspace == NULL: such an image reports det_cnt = -1. */
long leod_postprocess_nms_workspace_bytes(int B, int A);
int leod_postprocess_nms(float* pred, float* det_out, int* det_cnt, void* workspace, int B, int A, int nc, float conf_thre,
                         float nms_thre, int class_agnostic, int max_det, int vanilla_limit, leod_stream_t stream);
/* pred2label + filter_pred_boxes: det -> labels [B,max_det,8] = (0,x,y,w,h,cls,cls_conf,obj), lab_cnt[B]. */
int leod_pseudo_filter(const float* det, const int* det_cnt, float* lab, int* lab_cnt, int B, int max_det,
                       const float* obj_thr, const float* cls_thr, int nthr, int filter_boxes, float frame_w,
                       float frame_h, leod_stream_t stream);

/* ---- optimiser / input ----------------------------------------------------------------------------- */

/* value-clip + AdamW over flat buffers (modules/detection.py:485-518; train.py:236-237). g is scaled/clipped in place.
 * hp_dev (optional) = device float[4] {lr, 1-beta1^step, sqrt(1-beta2^step), grad_scale} overriding the host scalars, so a
 * captured hipGraph can be replayed with a new learning rate every step. */
int leod_adamw_clip_step(float* p, float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int step, float clip_value, float grad_scale, const float* hp_dev,
                         leod_stream_t stream);
/* bf16 shadow of a flat fp32 parameter buffer (the reference keeps ONE copy of the weights and lets autocast convert them per op,
 * train.py:236-243 precision=16; here the Linear kernels of precision mode bf16 read a 16-bit copy made once per optimiser step).
 * leod_set_weight_shadow registers shadow16 (n bf16 values, caller-owned) for the n floats at base (shadow16 == NULL withdraws it);
 * leod_weight_shadow_refresh rounds the registered buffers whose shadow is stale (force != 0: all of them, flags untouched -- for a
 * launch that is being recorded into a step plan / hipGraph) and returns the number of launches; leod_adamw_clip_step on a registered
 * buffer and leod_weight_shadow_invalidate mark shadows stale (stale shadows are never read); leod_weight_shadow_pin(1) makes the
 * launchers read the shadows regardless of the flags while a step is being recorded behind a forced refresh. */
int leod_set_weight_shadow(const float* base, long n, void* shadow16);
/* fp16 copy (n values, caller-owned, 8-byte aligned; NULL withdraws it) of a buffer registered above: leod_weight_shadow_refresh writes it
 * next to the bf16 copy in precision mode 2 ("16f"), whose forward GEMMs read it (the gradient GEMMs keep reading the bf16 copy).
 * LEOD_ERR_ARG if base is not registered. */
int leod_set_weight_shadow_f16(const float* base, void* shadow_f16);
int leod_weight_shadow_refresh(int force, leod_stream_t stream);
int leod_weight_shadow_invalidate(void);
int leod_weight_shadow_pin(int on);
/* Options of the attention block that no shipped config enables (the shipped block is fused into the GEMM kernels and never comes here).
 * act: 0 gelu 1 silu / swish 2 relu 3 sigmoid 4 tanh 5 relu6 6 leaky_relu 7 elu 8 hard_sigmoid 9 hard_swish 10 mish 11 selu 12 celu 13 hard_mish
 * (`mlp_activation`, models/layers/maxvit/maxvit.py:357-365).  gated != 0 (GLU.forward, maxvit.py:80-82): p [M, 2 * inner] = (a | g),
 * h [M, inner] = a * act(g); gated == 0: p [M, inner], h = act(p).  The backward writes dp (same shape as p) from dh.  inner % 4 == 0. */
int leod_act_glu_fwd(const float* p, float* h, long M, int inner, int act, int gated, leod_stream_t stream);
int leod_act_glu_bwd(const float* p, const float* dh, float* dp, long M, int inner, int act, int gated, leod_stream_t stream);
/* Token masking of the first stage (recurrent_backbone/maxvit_rnn.py:190-192): x[m][:] = token where mask[m] (in place; x [M, C], mask
 * [M] bytes, token [C]); backward: dtoken[c] += sum of the masked rows of dx, which are then zeroed (in place). */
int leod_token_mask_fwd(float* x, const unsigned char* mask, const float* token, long M, int C, leod_stream_t stream);
int leod_token_mask_bwd(float* dx, const unsigned char* mask, float* dtoken, long M, int C, leod_stream_t stream);
/* out[i] = (i == idx) ? *g : 0 for i < n <= 64 (device scalar g): the gradient of one entry of the six-entry loss vector as a zero-padded vector. */
int leod_onehot_scale(const float* g, float* out, int n, int idx, leod_stream_t stream);
/* dst[0..3] = (a,b,c,d) on the device (launch-time scalars for a replayed hipGraph). */
int leod_set_scalars4(float* dst, float a, float b, float c, float d, leod_stream_t stream);
/* Recurrent-state plumbing as one launch per call (host arrays of up to 16 entries; pointers and sizes 16-byte aligned):
 * leod_rows_masked_zero -- tensors[k][b, :] = 0 where mask[b] (B rows of row_bytes[k] each): RNNStates.reset over the (h, c) of all
 *   stages, reference modules/utils/detection.py:60-75 (`t[mask] = 0` per tensor);
 * leod_copy_multi -- dst[k][:] = src[k][:]: the initial (h, c) of a stage into slot 0 of its sequence buffers
 *   (reference models/layers/rnn.py:53-60 takes them as h_and_c_previous). */
int leod_rows_masked_zero(void* const* tensors, const long* row_bytes, int n, const unsigned char* mask, int B, leod_stream_t stream);
int leod_copy_multi(void* const* dst, const void* const* src, const long* nbytes, int n, leod_stream_t stream);
/* Channel concatenation of two NHWC maps, the first optionally upsampled x2 (nearest) on the way: out [B,H,W,Ca+Cb] from a [B,H>>up,W>>up,Ca]
 * and b [B,H,W,Cb] -- torch.cat([upsample(a), b], 1) of the PAFPN top-down path (models/detection/yolox_extension/models/yolo_pafpn.py:113-123)
 * and the cat of CSPLayer (models/detection/yolox/models/network_blocks.py:160-166) as one launch; _bwd: da (summed over the 2 x 2 copies when
 * up = 1) and db from dout.  fp32, Ca % 4 == Cb % 4 == 0, up in {0, 1}. */
int leod_cat2_up_fwd(const float* a, const float* b, float* out, int B, int H, int W, int Ca, int Cb, int up, leod_stream_t stream);
int leod_cat2_up_bwd(const float* dout, float* da, float* db, int B, int H, int W, int Ca, int Cb, int up, leod_stream_t stream);
/* dst[idx[j], :] += src[j, :] for j < nsel, rows of row_floats floats (% 4 == 0), idx unique and in [0, nrows_dst): the gradient of the labelled
 * frames added into the gradient of a stage's output map (the `selected_indices` gather of BackboneFeatureSelector, modules/utils/detection.py:120-157,
 * differentiated).  No atomics: the indices of one call must be distinct. */
int leod_rows_index_add(float* dst, const float* src, const long* idx, int nsel, long row_floats, int nrows_dst, leod_stream_t stream);
/* hflip TTA input of the pseudo-label pass (modules/pseudo_labeler.py:469-470: cat([ev, flip(ev, -1)], batch dim)) in one pass:
 * frames[t] = [B, rows_per_sample, W] bytes (T device pointers in a HOST array), out = [T, 2B, rows_per_sample, W]:
 * out[t, b] = frames[t][b], out[t, B + b, r, x] = frames[t][b, r, W - 1 - x]. */
int leod_stack_hflip_u8(const void* const* frames, int T, void* out, int B, long rows_per_sample, int W, leod_stream_t stream);
/* StackedHistogram.construct (data/utils/representations.py:78-123): int64 events -> uint8 [2*bins,H,W]. */
int leod_voxelize_u8(const long* x, const long* y, const long* pol, const long* t, long n_events, int* counts_ws,
                     unsigned char* out, int bins, int H, int W, int count_cutoff, int fastmode, leod_stream_t stream);
/* MixedDensityEventStack.construct (data/utils/representations.py:132-221): int64 events (time sorted) -> int8 [bins,H,W]: polarity sums
 * (2*pol-1) in the logarithmic time bin floor(max(bins - log(t_norm)/log(1/2), 0)), running sum over the bins in int8 arithmetic, clamp to
 * +-count_cutoff (count_cutoff < 0 = none, <= 127).  counts_ws: bins*H*W int32 of scratch. */
int leod_mixed_density_i8(const long* x, const long* y, const long* pol, const long* t, long n_events, int* counts_ws,
                          signed char* out, int bins, int H, int W, int count_cutoff, leod_stream_t stream);

/* On-device spatial augmentation of uint8 event representations src/dst [T,B,C,H,W] (data/utils/augmentor.py:216-331,
 * 390-401): per batch sample b, params[b] = {hflip, mode (0 none, 1 zoom-in, 2 zoom-out), x0, y0, win_h, win_w, tflip}
 * (7 ints); flip first, then the zoom with ATen's nearest-exact index rule; zoom-out leaves zeros outside the pasted
 * window.  tflip != 0 additionally applies time_flip_data (data/genx_utils/sequence_base.py:207-227): frames in reverse
 * order and the 2*bins channel planes of each frame reversed. */
int leod_augment_u8(const unsigned char* src, unsigned char* dst, const int* params, int T, int B, int C, int H, int W,
                    leod_stream_t stream);

/* dgrad of a Linear fused with the LayerNorm backward of its producer (x -> norm -> Linear, maxvit.py:267-269,110-118):
 * dx[M,K] = LN-backward(dy[M,N] @ W[N,K]) (+ dres), dgamma[K] += sum dn*xhat, dbeta[K] += sum dn, with x[M,K] the LayerNorm
 * input and stats[M,2] its saved (mean, rstd).  Stage-1 shapes only (K = 48, N = 144 / 192, M >= 16384): returns -3
 * (unsupported shape) otherwise and the caller runs leod_linear_dgrad + leod_layernorm_bwd. */
int leod_linear_dgrad_lnbwd(const float* dy, const float* W, const float* x, const float* stats, const float* ln_w,
                            const float* dres, float* dx, float* dgamma, float* dbeta, int M, int N, int K, int dy_bf16,
                            leod_stream_t stream);

/* ---- launch plans (no counterpart in the reference: its answer to launch overhead is torch.compile(mode='reduce-overhead') = CUDA
 * graphs, config/model/maxvit_yolox/default.yaml:8-11, modules/detection.py:43-44) -----------------------------------------------
 * A plan replays a stream-captured hipGraph as plain stream launches from one C loop: hip_graph (hipGraph_t of a finished capture:
 * kernel / memset / 1-D memcpy nodes) is sorted topologically, its chains become lanes (lane 0 = the stream given to
 * leod_plan_launch, lanes 1 .. max_lanes-1 = streams owned by the plan), edges between lanes become event record / wait pairs.
 * ~3 us of host time per kernel, parallel branches stay parallel (hipGraphLaunch on ROCm 7.2: 8 us per node single-stream, and as
 * slow as eager Python launches once the graph forks).  The graph must outlive the plan (kernel argument blocks are borrowed).
 * leod_plan_create returns a handle > 0 or a negative error (-3: a node type a plan cannot replay; leod_plan_last_error() names it).
 * The library's own weight-pack kernels (conv3_pack_kernel, lstm_pack_kernel: they read parameters only) are taken out of the chain the capture
 * put them in and run on lane 1 from the start of the plan (LEOD_PLAN_HOIST=0: left where they were captured).
 * leod_plan_info: info[9] = kernels, memsets, memcpys, empty nodes, lanes, events, cross-lane waits, ops, collectives (kernel nodes that are
 * leod_comm_allreduce calls recorded during the capture: the replay issues the all-reduce on the op's lane). */
long leod_plan_create(void* hip_graph, int max_lanes);
/* Address ranges (start address as a long, length in bytes; n of them, copied) of PERSISTENT weight-pack buffers.  leod_plan_create moves a
 * weight-pack kernel (conv3_pack_kernel / lstm_pack_kernel) to the start of the plan -- dropping the stream-order edges of its capture --
 * only when the buffer it writes lies inside one of these ranges: a pack into a temporary of the captured step keeps its captured order
 * (that temporary may share its address with another temporary of the same capture). */
int leod_plan_set_hoist_ranges(const long* starts, const long* bytes, int n);
int leod_plan_launch(long plan, leod_stream_t stream);
/* The step's input tensor without a copy: every kernel of the plan that reads it (the stem convolution and its weight gradient register
 * themselves) and was captured with a pointer inside [captured_base, captured_base + bytes) reads new_base + the same offset from the next
 * launch on.  Returns the number of kernels re-pointed (0: none in this plan -- copy into the captured buffer instead), < 0 on error.  The
 * new buffer must stay alive and unchanged until the launches reading it have completed (the backward plan's stem weight gradient is the
 * last reader) -- what the reference's autograd requires of the event tensor too (modules/detection.py:196-207). */
int leod_plan_rebase_input(long plan, const void* captured_base, long bytes, const void* new_base);
/* leod_plan_launch without its closing join: `stream` does not wait for the plan's side lanes; leod_plan_join(plan, stream) makes it wait
 * later (before the plan is launched again and before anything reads what the side lanes wrote). */
int leod_plan_launch_nojoin(long plan, leod_stream_t stream);
int leod_plan_join(long plan, leod_stream_t stream);
int leod_plan_info(long plan, int* info);
int leod_plan_destroy(long plan);
int leod_plan_dump(long plan, const char* path); /* debug listing: lane, kernel, events per op in launch order */
const char* leod_plan_last_error(void);

/* ---- data-parallel exchange: the library's own RCCL communicator (csrc/k_comm.hip) ---------------------------------
 * The all-reduces of a data-parallel step -- SyncBatchNorm statistics of the detection head, gradient buckets (reference: Lightning's
 * DDP strategy with sync_batchnorm, train.py:131-133,247) -- enqueued on the caller's stream like kernels.  RCCL is resolved from the
 * process at run time (dlopen); without it every call returns -3.  Rank 0 draws an id and hands its 128 bytes to the other ranks (the
 * host side uses torch.distributed for that); leod_comm_init is collective and binds the CURRENT device.  One communicator per process. */
int leod_comm_unique_id(char* id128);
int leod_comm_init(const char* id128, int rank, int world);
int leod_comm_world(void);                                  /* ranks of the communicator, 0 = none */
/* buf[count] <- element-wise sum over ranks, in place.  dtype: 0 float, 1 double, 2 bf16.  On a stream that is being captured the call
 * records a marker kernel instead; a launch plan made of that capture (leod_plan_create) issues the all-reduce at that position of every
 * replay -- a hipGraph launched as such would NOT. */
int leod_comm_allreduce(void* buf, long count, int dtype, leod_stream_t stream);
int leod_comm_destroy(void);
const char* leod_comm_last_error(void);

/* ---- host-side C++ of the path (no GPU involved) ------------------------------------------------------------------
 * Tracking post-filter of the pseudo-label loop: linear-velocity tracklets, confidence-ordered greedy IoU association,
 * short-tracklet removal and in-painting of missed detections.  Replaces modules/tracking/linear.py:10-292,
 * modules/tracking/utils.py:7-96 and EventSeqData._track (modules/pseudo_labeler.py:201-258).
 * boxes [N,5] float32 (cx,cy,w,h,class) of the labelled frames concatenated in frame order; is_gt [N] (uint8);
 * frame_idx [F] strictly increasing; counts [F] boxes per labelled frame.  remove [N] (uint8) out: 1 = box on a finished
 * non-GT tracklet with fewer than min_track_len hits.  inpaint != 0: predicted boxes of the kept tracklets at their
 * missed frames -> inp_frame [inp_cap], inp_box [inp_cap,5], *n_inp (needed count; rc -3 if inp_cap is too small). */
int leod_track_filter(const float* boxes, const unsigned char* is_gt, const int* frame_idx, const int* counts, int F,
                      int img_h, int img_w, int min_track_len, double min_conf, double iou_threshold, double q,
                      unsigned char* remove, int inpaint, int* inp_frame, float* inp_box, int inp_cap, int* n_inp);

/* Detection evaluation of the validation / test loop: the COCO bbox protocol the reference delegates to pycocotools'
 * COCOeval / detectron2's COCOeval_opt (utils/evaluation/prophesee/metrics/coco_eval.py:121-139) over the image windows
 * built by the +-50 ms time matching (coco_eval.py:49-97).  gt_box [G,4] / dt_box [D,4] = (x,y,w,h) float32 rows of all
 * images concatenated, gt_off / dt_off [n_img+1] first row of every image, *_cls class ids in [0,n_cat), dt_score the
 * class confidences.  iou_thrs [T], rec_thrs [R] as np.linspace gives them.  Out: precision [T,R,n_cat,4,3] and recall
 * [T,n_cat,4,3] (area ranges all/small/medium/large, maxDets 1/10/100), -1 where COCOeval leaves them undefined. */
int leod_coco_eval(const float* gt_box, const int* gt_cls, const int* gt_off, const float* dt_box, const int* dt_cls,
                   const float* dt_score, const int* dt_off, int n_img, int n_cat, const double* iou_thrs, int T,
                   const double* rec_thrs, int R, double* precision, double* recall);

#ifdef __cplusplus
}
#endif
#endif /* LEOD_HIP_H */
