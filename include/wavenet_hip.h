/*
 * wavenet_hip.h -- C ABI of libwavenet_hip.so (MI355X / gfx950).
 *
 * The reference (musyoku/wavenet) has no native boundary: its hot path sits behind a Python
 * class API and every FLOP is executed by Chainer.  This header is the boundary the reference
 * would bind if it had one: each entry point replaces the Chainer call sequence of the cited
 * reference lines.  wavenet_amd/{wavenet,faster_wavenet}.py call it through ctypes and keep the
 * reference's Python face (class / method / Params names) on top.
 *
 * Conventions (all entry points)
 *   - return 0 on success; <0 on failure (WN_EARG bad argument, WN_ESHAPE unsupported shape,
 *     WN_EHIP a HIP runtime error, WN_ETIMEOUT a device-side wait that gave up).  wn_last_error() gives a thread-local
 *     message.
 *   - every tensor pointer is a DEVICE pointer owned by the caller; nothing is retained after the
 *     call returns except by decoder handles, which copy what they need at create / load time.
 *   - no allocation, no synchronisation: work is enqueued on `stream` (a hipStream_t; NULL = the
 *     default stream) and the call returns immediately.  Scratch memory (operand images of the split-product GEMMs,
 *     per-workgroup partial tiles, embedding-gradient tables) comes from the caller through WnExec.
 *   - no mutable process state: the arithmetic of the channel GEMMs is an argument of every call that has one (WnExec).
 *   - activations are float32, time-major / channel-minor:  x[b][t][c]  == the reference's
 *     (B, C, 1, T) tensor in channels-last memory format.
 *   - convolution weights keep the reference's element order: a (Cout, Cin, 1, fw) or
 *     (Cout, Cin, fw, 1) Chainer W (wavenet.py:418-424) is the same memory as W[o][c][k], tap
 *     k = 0 the OLDEST sample.  1x1 weights are W[o][c].  A NULL bias pointer means "no bias".
 *   - Z is the reference's zero prefix (columns t < Z of a d > 1 dilated conv are exactly 0, no
 *     bias; wavenet.py:303-340).  The host computes it: Z = max(0, (fw-1)d - pad).  Pass 0 for the
 *     textbook convolution.
 */
#ifndef WAVENET_HIP_H
#define WAVENET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the functions declared between this push and the pop at the end of the
 * header are its ONLY dynamic symbols (tests/test_host_cpu.py holds `nm -D` to this list). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define WN_ABI_VERSION 5
#define WN_OK      0
#define WN_EARG   -1
#define WN_ESHAPE -2
#define WN_EHIP   -3
#define WN_ETIMEOUT -4   /* ABI 4: a device-side wait between co-resident workgroups gave up (wn_decoder_status) */

#define WN_ACT_NONE 0
#define WN_ACT_RELU 1   /* wavenet.py:588 */
#define WN_ACT_ELU  2   /* faster_wavenet.py:108 */

#define WN_MAX_SRC 64   /* sources per wn_skip_sum_* launch; longer lists are chunked by the caller */

int wn_abi_version(void);
const char* wn_last_error(void);

/* Per-call execution options of the entry points that contain a channel GEMM (the skip sum, the head convolutions, the
 * layers of widths other than 32/32/2, their backward) or need scratch memory.  The reference computes these contractions
 * in fp32 (cuDNN / cuBLAS under Chainer); here
 *   WN_GEMM_FP32    fp32-input MFMA                                        (exact fp32 products)
 *   WN_GEMM_BF16X3  every operand split into three bf16 parts, six products (fp32-accurate; what NULL selects)
 *   WN_GEMM_BF16    operands rounded to bf16 once, fp32 accumulation
 *   WN_GEMM_FP16X2  the skip contraction and its two backward GEMMs: operands scaled by a power of two and split into two
 *                   fp16 parts, three products (fp32-accurate to 2^-21 at half the matrix work of BF16X3).  tanh*sigmoid
 *                   outputs use a fixed scale (|z| <= 1); weights and gradients the power of two that fits their absolute
 *                   maximum, measured on the device by one small extra pass each (no host synchronisation; ABI 3: the
 *                   weights too -- a fixed 2^8 used to saturate weights above ~254).  The fused 32-channel layer kernels
 *                   run the same split products with per-tile scales taken from the wave's own maximum.  The head
 *                   convolutions keep the BF16X3 split.
 * Storage and accumulation stay fp32 in all of them; under FP32 / BF16X3 / BF16 the fused 32-channel layer kernels multiply
 * in exact fp32 (v_mfma_f32_32x32x2_f32).
 * ws / ws_bytes: device scratch, at least wn_exec_workspace_bytes() for the model and batch; its contents are dead when the
 * call's kernels have run, so ONE buffer per stream serves every call on that stream (never one buffer for two streams).
 * flags (ABI 3; the library reads NO environment variable and keeps no switch of its own -- what used to be
 * WAVENET_HIP_FORCE_GENERIC / _NO_FUSED_WIDE / _FWD_T1_MIN_BLOCKS inside the .so are per-call fields here):
 *   WN_EXEC_FORCE_GENERIC   every kernel of the call from the any-shape correctness path (generic_kernels.hip), fp32
 *   WN_EXEC_NO_FUSED_WIDE   the 128/128-channel bf16-operand layer forward as two launches instead of one (diagnostic)
 *   WN_EXEC_NO_FWD_GROUPS, WN_EXEC_NO_PIPELINED_GEMM, WN_EXEC_NO_MULTI_LAYER_BWD, WN_EXEC_BF16_MULTI_LAYER_BWD   see the defines
 * fwd_t1_min_blocks: launch size (workgroups of four 32-column tiles) from which the fused 32-channel layer forward takes its
 * one-tile-per-wave form; 0 = the library's default (512: every CU gets two to four workgroups), n > 0 = n (1 = always:
 * parity tests of that kernel at small sizes), < 0 = never.
 * plan (ABI 5): a step plan or NULL ("step plan" at the end of this header).
 * ex == NULL means { WN_GEMM_BF16X3, 0, NULL, 0, 0 }: fine for calls that need no scratch, WN_EARG (with the byte count)
 * otherwise. */
enum { WN_GEMM_FP32 = 0, WN_GEMM_BF16X3 = 1, WN_GEMM_BF16 = 2, WN_GEMM_FP16X2 = 3 };
#define WN_EXEC_FORCE_GENERIC 1u
#define WN_EXEC_NO_FUSED_WIDE 2u
#define WN_EXEC_NO_FWD_GROUPS 4u   /* fp16x2 stack forward: every layer its own launch (no k_layer_fwd_h2_grp); same results,
                                      bit for bit -- A/B timing and the parity tests of the per-layer kernel */
#define WN_EXEC_NO_PIPELINED_GEMM 8u /* fp16x2 skip contractions: the older kernels (k_colgemm_b3, k_wgrad_b3w) instead of
                                        k_colgemm_h2q / k_wgrad_h2p; same results, bit for bit -- A/B timing, parity tests */
#define WN_EXEC_NO_MULTI_LAYER_BWD 16u /* fp16x2 chained stack backward: one launch per layer instead of the multi-layer
                                          launch (k_layer_bwd_chain_multi).  That launch has NO grid barrier: co-resident
                                          workgroups follow one dataflow word per 32-column tile ("layers completed"),
                                          (V, U) rotate through three buffer pairs, the deal of tiles to waves rotates from
                                          layer to layer.  Results agree with the per-layer launches to fp32 summation order
                                          (which wave sums which tiles; ~1e-7) and are bit-reproducible from run to run */
#define WN_EXEC_BF16_MULTI_LAYER_BWD 32u /* bf16 storage (wn16_stack_bwd): layers L-2 .. 1 of the layer backward in ONE launch
                                            (k16_bwd_multi: the same dataflow words, k16_gate_bwd / k16_dx tile code unchanged)
                                            instead of two launches per layer.  Same results, bit for bit.  OPT-IN: measured
                                            equal to the per-layer launches within 1 %, and it needs every
                                            workgroup resident -- the library falls back by itself when the static occupancy
                                            query says they would not be.  CUs held by OTHER work at launch time are not
                                            covered by that query: a dataflow wait that then gives up (2^18 polls) poisons
                                            the layer's dWp with a NaN, so the gradient norm is not finite and wn_adam_step
                                            skips the step (WaveNet.last_update_applied() == False) -- never a silent update */
typedef struct WnExec {
    int precision;
    unsigned flags;
    void* ws;
    size_t ws_bytes;
    int fwd_t1_min_blocks;
    int reserved;                     /* 0 */
    void* plan;                       /* ABI 5: a step plan (wn_plan_create) or NULL -- see "step plan" below */
} WnExec;
/* 1 if the fused MFMA kernels cover this residual-layer shape (a pure shape query; a call with WN_EXEC_FORCE_GENERIC
 * runs the generic kernels whatever this says), 0 if the generic / wide paths run */
int wn_layer_fast_path(int Cr, int Cd, int fw);

/* ---- A10: first causal layer on integer tokens (data.py:61-68 one-hot + wavenet.py:298-301) ----
 * out[b,t,:] = sum_k W[:, idx[b, t-(fw-1-k)], k]  (+bias), taps with t-(fw-1-k) < 0 contribute 0.
 * Equals DilatedConvolution1D(d=1) applied to onehot_pixel_image(idx).                        */
int wn_embed_fwd(const int32_t* idx, const float* W, const float* bias, float* out,
                 int B, int T, int Q, int C, int fw, void* stream);
/* dW[o][q][k] += sum over (b,t) with idx[b,t-(fw-1-k)] == q of dout[b,t,o]; dbias += sum dout. */
int wn_embed_bwd(const int32_t* idx, const float* dout, float* dW, float* dbias,
                 int B, int T, int Q, int C, int fw, const WnExec* ex, void* stream);

/* ---- A5: dense dilated causal convolution, any shape (DilatedConvolution1D.__call__,
 * wavenet.py:294-342):  out[b,t,o] = sum_k sum_c W[o,c,k] x[b, t-(fw-1-k)d, c] + bias[o] for
 * t >= Z, 0 for t < Z.  Used for dense (non one-hot) inputs and for extra causal layers.       */
int wn_conv_fwd(const float* x, const float* W, const float* bias, float* out,
                int B, int T, int Cin, int Cout, int fw, int d, int Z, void* stream);
/* dx (overwritten, may be NULL), dW / dbias (accumulated, may be NULL). */
int wn_conv_bwd(const float* x, const float* W, const float* dout, float* dx, float* dW, float* dbias,
                int B, int T, int Cin, int Cout, int fw, int d, int Z, void* stream);

/* ---- A7: fused residual layer (ResidualConvLayer.__call__, wavenet.py:358-368) -------------
 *   a = conv(x; Wf,bf,d,Z)  g = conv(x; Wg,bg,d,Z)   z = tanh(a) * sigmoid(g)
 *   out = Wp z + bp + x
 * z (B,T,Cd) is always written: the skip projection Ws z is deferred to wn_skip_sum_fwd (one
 * contraction over all layers instead of L read-modify-write passes over a (B,T,Cs) tensor).
 * f_save / g_save (B,T,Cd), when non-NULL, receive tanh(a) and sigmoid(g) for wn_layer_bwd.    */
int wn_layer_fwd(const float* x,
                 const float* Wf, const float* bf, const float* Wg, const float* bg,
                 const float* Wp, const float* bp,
                 float* out, float* z, float* f_save, float* g_save,
                 int B, int T, int Cr, int Cd, int fw, int d, int Z, const WnExec* ex, void* stream);

/* Backward of one layer (Chainer autograd through wavenet.py:358-368).
 *   dz = Wp^T dout + dz_skip      da = dz g (1-f^2)      dg = dz f g (1-g)     (0 for t < Z)
 *   dWp += dout z^T  dbp += dout   dWf_k += da x[t-(fw-1-k)d]^T  dbf += da   (same for g)
 *   dx[t] = dout[t] + sum_k (Wf_k^T da + Wg_k^T dg)[t + (fw-1-k)d]
 * dz_skip (B,T,Cd) is this layer's slice of Ws^T dskip (wn_skip_sum_bwd_dz); NULL = none.
 * dout NULL = zero (the last layer's residual output is discarded, train_audio/train.py:72).
 * dab_ws is caller-provided scratch of wn_layer_bwd_workspace_floats() floats ((da,dg) for every
 * column, plus per-workgroup partial weight-gradient tiles on the MFMA path).  dWp/dbp may be
 * NULL (last layer).                                                                            */
size_t wn_layer_bwd_workspace_floats(int B, int T, int Cr, int Cd, int fw);
int wn_layer_bwd(const float* x, const float* f, const float* g,
                 const float* Wf, const float* Wg, const float* Wp,
                 const float* dout, const float* dz_skip,
                 float* dx, float* dWf, float* dbf, float* dWg, float* dbg, float* dWp, float* dbp,
                 float* dab_ws,
                 int B, int T, int Cr, int Cd, int fw, int d, int Z, const WnExec* ex, void* stream);

/* ---- 1x1 convolution with the activation the reference applies BEFORE it -------------------
 * out[n,:] = W act(x[n,:]) + b.   Head layers (wavenet.py:587-590: relu then conv; elu in
 * faster_wavenet.py:107-110), projection_block / projection_softmax with WN_ACT_NONE.           */
int wn_pointwise_fwd(const float* x, const float* W, const float* bias, float* out,
                     int N, int Cin, int Cout, int act, const WnExec* ex, void* stream);
/* dx[n,:] = act'(x[n,:]) * (W^T dout[n,:]) (overwritten; NULL to skip);  dW += dout act(x)^T;
 * dbias += sum dout. */
int wn_pointwise_bwd(const float* x, const float* W, const float* dout, float* dx, float* dW,
                     float* dbias, int N, int Cin, int Cout, int act, const WnExec* ex, void* stream);

/* ---- A11: the skip sum, deferred:  skip[b,t,:] = sum_l (Ws_l z_l[b, t_off+t, :] + bs_l) ------
 * (wavenet.py:574-582: sum_skip_connections += projection_softmax, all blocks, all layers).
 * z[l] is (B,T,cd[l]); skip is (B,Tw,Cs) for columns t_off .. t_off+Tw-1 (the harness keeps
 * only the last train_width columns, train_audio/train.py:73).  accumulate != 0 adds to skip.  */
int wn_skip_sum_fwd(int L, const float* const* z, const float* const* Ws, const float* const* bs,
                    const int* cd, float* skip, int B, int T, int t_off, int Tw, int Cs,
                    int accumulate, const WnExec* ex, void* stream);
/* dz[l][b,t,:] = Ws_l^T dskip[b,t-t_off,:] for t >= t_off, 0 before (dz[l] is (B,T,cd[l])). */
int wn_skip_sum_bwd_dz(int L, const float* const* Ws, const int* cd, const float* dskip,
                       float* const* dz, int B, int T, int t_off, int Tw, int Cs, const WnExec* ex, void* stream);
/* dWs[l] += dskip^T z_l (Cs x cd[l]);  dbs[l] += sum dskip  (either table entry may be NULL). */
int wn_skip_sum_bwd_dw(int L, const float* const* z, const int* cd, const float* dskip,
                       float* const* dWs, float* const* dbs, int B, int T, int t_off, int Tw, int Cs,
                       const WnExec* ex, void* stream);

/* ---- A11 + A15 as one call each: the reference's per-layer Python loop (wavenet.py:572-582 and
 * Chainer's backward over it) executed inside the library on one stream.                        */
typedef struct WnStackDesc {
    int n_layers, Cr, Cs, fw;              /* n_layers = blocks * layers per block                */
    const int* cd;                         /* host, per layer: gate width                          */
    const int* dilation;                   /* host, per layer: fw ** layer_index (wavenet.py:429)  */
    /* host arrays (n_layers) of device pointers; bias tables or entries may be NULL             */
    const float* const* Wf; const float* const* bf; const float* const* Wg; const float* const* bg;
    const float* const* Wp; const float* const* bp; const float* const* Ws; const float* const* bs;
} WnStackDesc;
/* xs (n_layers,B,T,Cr) receives every layer's output (xs[n_layers-1] is the stack output); z, f, g
 * are layer-major (layer l at offset sum_{i<l} B*T*cd[i]); f/g NULL for inference; skip (B,T-t_off,Cs)
 * may be NULL.  The zero prefix is derived from T per layer when compat_zero_prefix != 0.
 * window_only != 0 (training, train_audio/train.py:72-73 keeps skip[t_off:] only and discards the residual
 * output): columns that cannot influence skip[t_off:] -- further below t_off than the layers above a layer
 * reach -- are not computed; xs / z / f / g are left untouched there and wn_stack_bwd must then be called
 * with dout == NULL.  Loss and gradients are unchanged.                                              */
/* 1 if wn_stack_bwd needs tanh saved (f); 0 if every layer runs on the fused 32-channel kernels without conv / projection
 * biases: then f may be NULL in both calls (only z and sigmoid are kept; tanh = z / sigmoid is recovered by the backward). */
int wn_stack_saves_tanh(const WnStackDesc* d, const WnExec* ex);
int wn_stack_fwd(const WnStackDesc* d, const float* x, float* xs, float* z, float* f, float* g,
                 float* skip, int B, int T, int t_off, int compat_zero_prefix, int window_only, const WnExec* ex, void* stream);
size_t wn_stack_bwd_workspace_bytes(const WnStackDesc* d, int B, int T);
/* Upper bound of the WnExec scratch any call on this model needs at batch (B, T): head_channels = softmax_conv_channels
 * (n entries), causal_channels / causal_fw describe the first causal layer (its gradient tables).  d may be NULL. */
size_t wn_exec_workspace_bytes(const WnStackDesc* d, int Q, int causal_channels, int causal_fw, const int* head_channels,
                               int n_head_channels, int B, int T);
/* dout: gradient of the last layer's output (NULL = unused, train_audio/train.py:72); dskip
 * (B,T-t_off,Cs): gradient of the skip sum (NULL = unused); dx (B,T,Cr) may be NULL.  Gradient
 * tables are host arrays of device pointers, accumulated into (the flat gradient arena).          */
int wn_stack_bwd(const WnStackDesc* d, const float* x, const float* xs, const float* z, const float* f,
                 const float* g, const float* dout, const float* dskip, float* dx,
                 float* const* dWf, float* const* dbf, float* const* dWg, float* const* dbg,
                 float* const* dWp, float* const* dbp, float* const* dWs, float* const* dbs,
                 float* ws, size_t ws_bytes, int B, int T, int t_off, int compat_zero_prefix, const WnExec* ex, void* stream);

/* ---- softmax over the channel axis (wavenet.py:592) and A14 (wavenet.py:597-617) ------------ */
int wn_softmax_fwd(const float* logits, float* prob, int N, int Q, void* stream);
/* loss[0] = sum over rows of -log softmax(logits[n])[target[n]] / n_norm (loss points at WN_XENT_LOSS_WORDS floats: the
 * rest holds per-workgroup sums that one workgroup adds in a fixed order -- no float atomics); dlogits (may be NULL) =
 * (softmax - onehot) / n_norm.  Rows are b*Tw + t, as after the reference's transpose(0,3,2,1) + reshape.  A row whose
 * target is -1 is ignored (no loss, zero gradient) as chainer.functions.softmax_cross_entropy does; so is any other target
 * outside [0, Q) -- nothing is read out of bounds.  n_norm = the number of rows that count (Chainer: labels != -1);
 * n_norm == 0 means N; n_norm < 0 (ABI 3): counted on the device from the labels (rows with a label in [0, Q), at least
 * 1), for targets the host never saw -- one extra small launch, no synchronisation.                                 */
#define WN_XENT_LOSS_WORDS 2056
int wn_softmax_xent(const float* logits, const int32_t* target, float* loss, float* dlogits, int N, int Q,
                    int64_t n_norm, void* stream);
/* ABI 4.  The LAST head convolution and the loss in ONE launch (wavenet.py:584-593 with apply_softmax = False followed by
 * wavenet.py:597-617): logits = W act(x) + b are formed and consumed on the chip -- they never reach memory --, dlogits (N, Cout)
 * receives d loss / d logits for an upstream gradient of 1 (what wn_softmax_xent writes), loss as for wn_softmax_xent
 * (WN_XENT_LOSS_WORDS floats, n_norm with the same meaning).  Covered: WN_GEMM_FP16X2, Cout = 256, Cin a multiple of 32
 * and N <= 253,952 rows -- one 128-row workgroup per partial-sum slot of `loss` -- (wn_head_xent_supported, which takes N since
 * ABI 5; WN_ESHAPE otherwise: run wn_pointwise_fwd + wn_softmax_xent, which has no row limit).  The input has no known range: every
 * 32-channel chunk of a wave's 32 columns is scaled by the power of two that fits the wave's own maximum before the fp16 split
 * (error <= 2^-21 per product as elsewhere under FP16X2).  Backward: wn_pointwise_bwd(x, W, dlogits, ...) as after the two calls. */
int wn_head_xent_supported(int64_t N, int Cin, int Cout, const WnExec* ex);
int wn_head_xent(const float* x, const float* W, const float* bias, const int32_t* target, float* loss, float* dlogits,
                 int N, int Cin, int Cout, int act, int64_t n_norm, const WnExec* ex, void* stream);

/* x[i] *= *scale_dev (a device scalar), and nothing at all when *scale_dev == 1: the backward of the loss node
 * (chainer's softmax_cross_entropy backward multiplies by the upstream gradient, which is 1 for `loss.backward()`).  */
int wn_scale_by_dev(float* x, const float* scale_dev, int64_t n, void* stream);

/* ---- layout conversion at the boundary: (B,C,1,T) T-contiguous <-> (B,T,C) ------------------ */
int wn_nchw_to_btc(const float* src, float* dst, int B, int C, int T, void* stream);
int wn_btc_to_nchw(const float* src, float* dst, int B, int C, int T, void* stream);

/* ---- A16/A17: queue-cached autoregressive decoder (faster_wavenet.py:50-113) -----------------
 * Persistent per-layer state in HBM: a ring of (fw-1)*d input columns per residual layer (and
 * per extra causal layer) instead of the reference's full-window caches that are rolled every
 * step; the head runs on the newest column only.                                               */
#define WN_DECODER_ONE_WORKGROUP 64u
typedef struct WnDecoderDesc {
    int Q, fw_causal, n_causal, fw, n_blocks, n_layers;  /* n_layers per block; dilation fw^l */
    int Cr, Cs, n_head;
    const int* causal_channels;       /* host, n_causal entries                                */
    const int* cd;                    /* host, n_layers entries (per layer index in a block)    */
    const int* head_channels;         /* host, n_head+1 entries: Cs, ..., Q                     */
    /* device pointers to weights in creation order (wavenet.py:461-472); biases may be NULL   */
    const float* const* causal_W; const float* const* causal_b;                 /* n_causal    */
    const float* const* Wf; const float* const* bf;                             /* blocks*layers */
    const float* const* Wg; const float* const* bg;
    const float* const* Wp; const float* const* bp;
    const float* const* Ws; const float* const* bs;
    const float* const* head_W; const float* const* head_b;                     /* n_head      */
    int head_act;                     /* WN_ACT_ELU for FasterWaveNet, WN_ACT_RELU for WaveNet  */
    unsigned flags;                   /* WN_EXEC_FORCE_GENERIC: never the specialised 32/256-channel decode kernel;
                                         WN_DECODER_ONE_WORKGROUP: wn_decoder_run of that kernel on ONE workgroup instead of
                                         nine (the chain | eight workgroups of skip rows and their share of the logits):
                                         same products, another summation order for the skip rows and the logits --
                                         probabilities agree to ~1e-7, sampled tokens can differ at a near-tie of the
                                         cumulative distribution; each form is deterministic */
} WnDecoderDesc;

int wn_decoder_create(void** handle, const WnDecoderDesc* desc, void* stream);
int wn_decoder_destroy(void* handle);
/* re-copy the weights (after an optimiser step) */
int wn_decoder_update_weights(void* handle, const WnDecoderDesc* desc, void* stream);
/* Seed the rings from a full-window forward (faster_wavenet.py:13-47): tokens (W) are the window's
 * tokens, causal_out[i] is (1,W,C_i) and layer_in[j] (1,W,Cr) is the INPUT of residual layer j
 * (blocks*layers entries), all produced by the ordinary forward kernels over the same window.  */
int wn_decoder_load_state(void* handle, const int32_t* tokens, int W,
                          const float* const* causal_out, const float* const* layer_in, void* stream);
/* One step (FasterWaveNet._forward_one_step): consume `token` (the newest sample), advance all
 * rings, write the probabilities (or logits if apply_softmax == 0) of the next sample to prob (Q). */
int wn_decoder_step(void* handle, int32_t token, float* prob, int apply_softmax, void* stream);
/* n steps entirely on device (train_audio/generate.py:24-43 with --fast): each step consumes the
 * previous token, computes p, draws with numpy's algorithm from uniforms[i] (float64 cumsum,
 * normalise, searchsorted right) and appends.  first_token is the newest token of the window the
 * state was loaded from plus one draw, i.e. the token emitted by the prefill step.  out_tokens (n)
 * receives the n emitted tokens; prob_trace (n*Q) is optional.                                  */
int wn_decoder_run(void* handle, int32_t first_token, const double* uniforms, int n,
                   int32_t* out_tokens, float* prob_trace, void* stream);
/* ABI 4.  n_handles independent utterances in ONE launch, nine workgroups each (generate.py:9-60 is batch 1 with a strict
 * sample-to-sample dependency, wavenet.py:286,290,354: a single utterance can use 9 of the GPU's 256 CUs -- the rest can only
 * run OTHER utterances; SURVEY 8(e) "replicas only", on one GPU).  Every handle is a decoder of its own (wn_decoder_create +
 * wn_decoder_load_state: its rings, its packed weights), all of the same model shape, none created with
 * WN_DECODER_ONE_WORKGROUP; uniforms[u] (n doubles), out_tokens[u] (n) and the optional prob_traces[u] (n * Q; the array
 * itself may be NULL) are device pointers held in HOST arrays, first_tokens is a host array.  At most wn_decoder_batch_max()
 * (28) utterances, and 9 * n_handles workgroups must fit the device's CUs (WN_ESHAPE otherwise); n >= 2.  The groups share
 * nothing: an utterance's tokens are those of its own wn_decoder_run with the same uniforms, bit for bit.  same_weights != 0 is
 * the caller's word that every handle was created from (and updated with) the SAME weights: all utterances then read handle 0's
 * packed weights (one copy through the L2s instead of n_handles; each keeps its own state) -- same tokens, higher rate.
 * wn_decoder_status(handle) reports per utterance as after wn_decoder_run. */
int wn_decoder_batch_max(void);
int wn_decoder_run_batch(void* const* handles, int n_handles, const int32_t* first_tokens, const double* const* uniforms, int n,
                         int32_t* const* out_tokens, float* const* prob_traces, int same_weights, void* stream);
/* ABI 4.  wn_decoder_run's default form runs on nine workgroups that hand values to each other through device memory and
 * therefore must all be resident.  The library uses the one-workgroup kernel by itself on a device with fewer than nine
 * CUs; what it cannot know in advance -- other work holding the CUs for seconds -- ends in a wait that GIVES UP after
 * ~1-2 s: no trap, no hang, the launch runs to its end, the HIP context survives, but the tokens of that run are void.
 * wn_decoder_status synchronises `stream` and returns WN_OK, or WN_ETIMEOUT when the last wn_decoder_run on this handle
 * gave up a wait (re-create the handle with WN_DECODER_ONE_WORKGROUP and decode again). */
int wn_decoder_status(void* handle, void* stream);
/* categorical draw with numpy's algorithm for n independent rows (generate.py:39) */
int wn_sample_categorical(const float* prob, const double* uniforms, int32_t* out, int n, int Q,
                          void* stream);

/* ---- A2 on the device (optional path; data.py:18-23 and 37-43) --------------------------------------
 * Table lookups: lut65536[v + 32768] is the token of the int16 sample v, table[q] the sample value of token q;
 * both tables are built on the host with the reference's float64 formulas (wavenet_amd/data.py), so the device
 * results are the host's bit for bit.  Tokens outside [0, Q) are clamped on decode.                        */
int wn_mulaw_encode_pcm16(const int16_t* pcm, const int32_t* lut65536, int32_t* tokens, int64_t n, void* stream);
int wn_mulaw_decode(const int32_t* tokens, const float* table, float* out, int64_t n, int Q, void* stream);

/* ---- the step either side of backward ------------------------- */
/* out[0] = sum (grad*grad_mult + weight_decay*param)^2: the squared norm GradientClipping sees after the
 * WeightDecay hook (wavenet.py:175-199, 477-480).  out points at WN_SQNORM_WORDS floats: out[0] is ASSIGNED (no
 * zeroing by the caller), the rest holds per-workgroup partial sums that one workgroup adds in a fixed order -- no
 * float atomics, the norm (and with it the clipping rate) is bit-reproducible.  param may be NULL when weight_decay == 0. */
#define WN_SQNORM_WORDS 1040
int wn_sqnorm(const float* grad, const float* param, int64_t n, float grad_mult, float weight_decay,
              float* out, void* stream);
/* Chainer Adam with the reference's hooks folded in, in hook order:
 *   g = grad*grad_mult + wd*param;   g *= min(1, clip/sqrt(*sqnorm))  (sqnorm != NULL, clip > 0)
 *   m += (1-b1)(g-m); v += (1-b2)(g^2-v); param -= lr_t * m / (sqrt(v)+eps).
 * lr_t = alpha*sqrt(1-b2^t)/(1-b1^t) is computed by the host.
 * ABI 4: when sqnorm is given (clip > 0) and *sqnorm is not finite, the WHOLE update is skipped -- param, m and v keep
 * their values (every update rule below does the same).  A void gradient (a backward whose multi-layer launch gave up
 * a wait and flagged it with a NaN, an overflow) then costs one step instead of the optimiser state; under data
 * parallelism the all-reduce carries the NaN to every rank and every rank skips the same step.  The host can read the
 * word back whenever it likes (wavenet_amd: WaveNet.last_update_applied()).                        */
int wn_adam_step(float* param, const float* grad, float* m, float* v, int64_t n,
                 float lr_t, float beta1, float beta2, float eps, float weight_decay,
                 const float* sqnorm, float clip, float grad_mult, void* stream);
/* Eve (wavenet.py:10-79, the reference's own optimizer class and its only device code, the elementwise kernel at
 * 57-65): Adam's moments with the denominator d * sqrt(v) + eps.  d is the loss-feedback scalar the host maintains
 * (wavenet.py:27-44); hooks as in wn_adam_step.                                                                   */
int wn_eve_step(float* param, const float* grad, float* m, float* v, int64_t n,
                float lr_t, float beta1, float beta2, float eps, float d, float weight_decay,
                const float* sqnorm, float clip, float grad_mult, void* stream);
/* The same update with the step size read from device memory (*lr_t_dev) at execution time: a training step
 * captured once into a hipGraph is replayed with a fresh bias-corrected step size by writing that scalar.  */
int wn_adam_step_dev(float* param, const float* grad, float* m, float* v, int64_t n,
                     const float* lr_t_dev, float beta1, float beta2, float eps, float weight_decay,
                     const float* sqnorm, float clip, float grad_mult, void* stream);
/* The other update rules get_optimizer() names (wavenet.py:81-97; chainer.optimizers.SGD / MomentumSGD / AdaGrad /
 * AdaDelta / NesterovAG / RMSprop as Chainer publishes them), behind the same hooks as wn_adam_step:
 *   SGD          p -= lr g                                  MomentumSGD  v = hyper v - lr g; p += v
 *   AdaGrad      h += g^2; p -= lr g/(sqrt(h)+eps)          NesterovAG   v = hyper v - lr g; p += hyper^2 v - (1+hyper) lr g
 *   RMSprop      ms += (1-hyper)(g^2-ms); p -= lr g/(sqrt(ms)+eps)
 *   AdaDelta     msg += (1-hyper)(g^2-msg); dx = sqrt((msdx+eps)/(msg+eps)) g; msdx += (1-hyper)(dx^2-msdx); p -= dx
 * s1 = v / h / ms / msg, s2 = msdx (AdaDelta only, else NULL); hyper = momentum / alpha / rho.  lr_dev != NULL: the
 * learning rate is read from device memory at execution time (graph replay), `lr` is ignored.                       */
enum { WN_RULE_SGD = 0, WN_RULE_MOMENTUM_SGD = 1, WN_RULE_ADAGRAD = 2, WN_RULE_ADADELTA = 3, WN_RULE_NESTEROV = 4,
       WN_RULE_RMSPROP = 5 };
int wn_rule_step(int rule, float* param, const float* grad, float* s1, float* s2, int64_t n, float lr,
                 const float* lr_dev, float hyper, float eps, float weight_decay, const float* sqnorm, float clip,
                 float grad_mult, void* stream);

/* ======================================================================================================
 * bf16-storage path (BASELINE.json configs[4]: 128 residual / dilation channels, 512 skip channels): activations are
 * bfloat16 in HBM (uint16_t = raw bf16 bits, same (B, T, C) layout), every contraction accumulates in fp32 on
 * v_mfma_f32_32x32x16_bf16, the weights stay fp32 (master copy, gradients, optimiser) and are repacked into bf16
 * matrix-core operand images once per step.  Same reference operations as the fp32 entry points above
 * (ResidualConvLayer.__call__ wavenet.py:358-368, forward_residual_block 572-582, forward_softmax_block 584-593 and
 * their backward).  Covers: Cr = Cd = 128, filter width 2, Cs a multiple of 256, an even number of layers (<= 48), no
 * conv / projection biases (the reference default); wn16_supported() says whether a stack qualifies.
 * The forward keeps only each layer's output and z; tanh / sigmoid are recomputed by the backward.
 * ====================================================================================================== */
int wn16_supported(const WnStackDesc* d);
/* bf16 elements of the packed operand images of a stack (per-layer images + the skip matrix and its transpose) */
size_t wn16_pack_elems(const WnStackDesc* d);
int wn16_pack_stack(const WnStackDesc* d, uint16_t* pack, void* stream);
/* A10 on tokens, bf16 output (filter width 2) */
int wn16_embed_fwd(const int32_t* idx, const float* W, const float* bias, uint16_t* out, int B, int T, int Q, int C,
                   int fw, void* stream);
/* its gradient (the backward of data.py:61-68 + wavenet.py:298-301 on tokens): dW (C,Q,2) += one-hot(tokens)^T dout on
 * the matrix cores, dbias (C) += column sums or NULL; dout (B,T,128) bf16; 128 channels, 256 token values, filter width 2 */
size_t wn16_embed_bwd_workspace_bytes(int B, int T);
int wn16_embed_bwd(const int32_t* idx, const uint16_t* dout, float* dW, float* dbias, int B, int T, int Q, int C, int fw,
                   void* ws, size_t ws_bytes, void* stream);
int wn16_cvt_to_bf16(const float* src, uint16_t* dst, int64_t n, void* stream);     /* n % 8 == 0 */
int wn16_cvt_to_f32(const uint16_t* src, float* dst, int64_t n, void* stream);
/* A11: xs (L,B,T,128) every layer's output, z (L,B,T,128), skip (B,T-t_off,Cs) or NULL */
int wn16_stack_fwd(const WnStackDesc* d, const uint16_t* pack, const uint16_t* x, uint16_t* xs, uint16_t* z,
                   uint16_t* skip, int B, int T, int t_off, int compat_zero_prefix, void* stream);
size_t wn16_stack_bwd_workspace_bytes(const WnStackDesc* d, int B, int T, int t_off);
/* A15: dout must be NULL (train_audio/train.py:72 discards the stack's residual output), dskip (B,T-t_off,Cs),
 * dx (B,T,128) or NULL; fp32 gradients accumulated.  flags (ABI 4): WN_EXEC_* bits; WN_EXEC_BF16_MULTI_LAYER_BWD selects the
 * one-launch form of the layer backward (see the define). */
int wn16_stack_bwd(const WnStackDesc* d, const uint16_t* pack, const uint16_t* x, const uint16_t* xs, const uint16_t* z,
                   const uint16_t* dout, const uint16_t* dskip, uint16_t* dx, float* const* dWf, float* const* dWg,
                   float* const* dWp, float* const* dWs, void* ws, size_t ws_bytes, int B, int T, int t_off,
                   int compat_zero_prefix, unsigned flags, void* stream);
/* A12: Wb (Cout,Cin) and WbT (Cin,Cout) are the bf16 images of W; out is bf16 or fp32 (out_f32) */
int wn16_pack_pointwise(const float* W, uint16_t* Wb, uint16_t* WbT, int Cout, int Cin, void* stream);
int wn16_pointwise_fwd(const uint16_t* x, const uint16_t* Wb, const float* bias, void* out, int out_f32, int64_t N,
                       int Cin, int Cout, int act, void* stream);
/* ws (ABI 3; wn16_pointwise_bwd_workspace_bytes, may be NULL): with it the weight and bias gradients are sums of
 * per-workgroup partials added in a fixed order -- bit-reproducible; without it they accumulate through float atomics */
size_t wn16_pointwise_bwd_workspace_bytes(int64_t N, int Cout);
int wn16_pointwise_bwd(const uint16_t* x, const uint16_t* WbT, const uint16_t* dout, const float* dout_f32,
                       uint16_t* dout_scratch, uint16_t* dx, float* dW, float* dbias, int64_t N, int Cin, int Cout,
                       int act, void* ws, size_t ws_bytes, void* stream);

/* ---- step plan (ABI 5): the weight-only preparation of a training step in two launches -------------------------------
 * Between the optimiser step and the next one the weights do not change, but every entry point that holds a channel GEMM
 * prepares its operand images per call: zero a range word, measure max |W|, split -- three launch-floor kernels per GEMM --
 * and the stack packs its fp16 x 2 layer images, zeroes the multi-layer backward's dataflow words, measures max |dskip| ...:
 * ~16 of the 63 kernels of BASELINE config 2's step, ~0.1 ms of its 2.8.  A plan hoists them (wavenet.py:515-519 is the step):
 *   wn_plan_create(&plan, dev_mem, bytes)     caller-owned device memory (256-byte aligned; 16 MB covers config 2), no hipMalloc
 *   wn_plan_record(plan)                      then run ONE step whose WnExec carry .plan = plan: it runs as without a plan and
 *                                             every preparation that depends on weights only is registered (launcher arguments)
 *   wn_plan_finish(plan, stream)              lays the images out and uploads the job table (blocking; not under capture)
 *   wn_plan_prepare(plan, zero, n, stream)    FIRST call of every later step: all images, range words and plan-owned words in
 *                                             two launches; also zeroes `zero[0..n)` (the gradient arena: n % 4 == 0, 16-byte
 *                                             aligned; NULL / 0: nothing)
 * A READY plan makes an entry point skip its own preparation when the plan holds it (same weight pointers, same form) --
 * results are bit-identical to the per-call preparation -- so a call that carries one ASSERTS that wn_plan_prepare ran on the
 * same stream since the weights last changed.  Weight POINTERS are the keys: a plan belongs to one model.  Plan-owned words:
 * the dataflow words of the multi-layer backward, and the range word of the head's dx, filled by the GEMM that writes dx and
 * read by the dz contraction instead of a 100 MB pass over dskip.  A plan is a host object for ONE thread at a time;
 * wn_plan_record on a finished plan forgets its jobs and lays the images out anew at the next finish -- launches captured
 * into a graph against the old layout are void from then on (one plan per captured graph: wavenet_amd.TrainStepGraph).
 * wn_plan_stats: out[0..7] = state (0 idle, 1 recording,
 * 2 ready), weight images, layer images (layers), plan-owned words, device bytes used, prepare calls, look-ups served,
 * look-ups not served. */
int wn_plan_create(void** plan, void* dev_mem, size_t dev_bytes);
int wn_plan_destroy(void* plan);
int wn_plan_record(void* plan);
int wn_plan_finish(void* plan, void* stream);
int wn_plan_prepare(void* plan, float* zero, int64_t zero_floats, void* stream);
int wn_plan_stats(void* plan, int64_t* out8);

/* ---- measurement aid (bench.py): per-entry-point HIP-event timing on the caller's stream ------- */
int wn_prof_enable(int on);                    /* 1: clear + start recording, 0: stop               */
/* "name calls total_ms min_ms max_ms" per line into buf; returns the bytes needed (synchronises). */
int wn_prof_report(char* buf, int buflen);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* WAVENET_HIP_H */
