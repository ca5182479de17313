// Internal host-side launchers shared between translation units of libwavenet_hip.so.
#pragma once
#include "wn_common.hpp"

namespace wn {

// ---- generic_kernels.hip (any shape) --------------------------------------------------------
int generic_embed_fwd(const int32_t*, const float*, const float*, float*, int, int, int, int, int, hipStream_t);
int generic_embed_bwd(const int32_t*, const float*, float*, float*, int, int, int, int, int, hipStream_t);
int generic_conv_fwd(const float*, const float*, const float*, float*, int, int, int, int, int, int, int,
                     hipStream_t);
int generic_conv_bwd(const float*, const float*, const float*, float*, float*, float*, int, int, int, int, int,
                     int, int, hipStream_t);
int generic_layer_fwd(const float* x, const float* Wf, const float* bf, const float* Wg, const float* bg,
                      const float* Wp, const float* bp, float* out, float* z, float* fs, float* gs, int B,
                      int T, int Cr, int Cd, int fw, int d, int Z, hipStream_t s);
int generic_layer_bwd(const float* x, const float* f, const float* g, const float* Wf, const float* Wg,
                      const float* Wp, const float* dout, const float* dzs, float* dx, float* dWf, float* dbf,
                      float* dWg, float* dbg, float* dWp, float* dbp, float* dab, int B, int T, int Cr, int Cd,
                      int fw, int d, int Z, hipStream_t s);
int generic_pointwise_fwd(const float*, const float*, const float*, float*, long long, int, int, int,
                          hipStream_t);
int generic_pointwise_bwd(const float*, const float*, const float*, float*, float*, float*, long long, int, int,
                          int, hipStream_t);
int generic_skip_sum_fwd(int L, const float* const* z, const float* const* Ws, const float* const* bs,
                         const int* cd, float* skip, int B, int T, int t_off, int Tw, int Cs, int accumulate,
                         hipStream_t s);
int generic_skip_bwd_dz(int L, const float* const* Ws, const int* cd, const float* dskip, float* const* dz,
                        int B, int T, int t_off, int Tw, int Cs, hipStream_t s);
int generic_skip_bwd_dw(int L, const float* const* z, const int* cd, const float* dskip, float* const* dWs,
                        float* const* dbs, int B, int T, int t_off, int Tw, int Cs, hipStream_t s);
int generic_colsum(const float* A, int nB, int nT, int tmin, int lda, int M, float* out, hipStream_t s);
int generic_softmax(const float*, float*, long long, int, hipStream_t);
int generic_softmax_xent(const float*, const int32_t*, float*, float*, long long, int, long long n_norm, hipStream_t);
// the two small launches around a loss kernel that leaves its per-workgroup sums in loss[kXentPart + workgroup]: the device-side
// count of the rows that count (n_norm < 0; *ncnt = its workgroups) and the fixed-order sum into loss[0]
int generic_xent_count(const int32_t* target, long long N, int Q, float* loss, int* ncnt, hipStream_t s);
int generic_xent_final(float* loss, int nblocks, long long n_norm, int ncnt, hipStream_t s);
int generic_transpose(const float* src, float* dst, int batch, int R, int Cc, hipStream_t s);
int generic_sample(const float*, const double*, int32_t*, int, int, hipStream_t);
int generic_mulaw_encode_pcm16(const int16_t* pcm, const int32_t* lut, int32_t* tok, long long n, hipStream_t s);
int generic_mulaw_decode(const int32_t* tok, const float* table, float* out, long long n, int Q, hipStream_t s);
int generic_sqnorm(const float* g, const float* p, long long n, float gmult, float wd, float* out, hipStream_t s);
// ---- the current call's WnExec, handed down the host-side call tree through a thread-local (set by the entry point for
// the duration of the call: nothing survives it, calls on different threads do not see each other) ----------------
struct ExecScope {
    const WnExec* prev;
    explicit ExecScope(const WnExec* ex);
    ~ExecScope();
};
int gemm_mode();                       // WN_GEMM_* of the current call (WN_GEMM_FP32 under WN_EXEC_FORCE_GENERIC)
bool exec_flag(unsigned f);            // WnExec.flags of the current call
int exec_fwd_t1_min_blocks();          // WnExec.fwd_t1_min_blocks with the default filled in
bool layer_fast_path(int Cr, int Cd, int fw);   // fused 32-channel kernels for this shape in the current call
void* exec_scratch(size_t bytes, const char* what);   // the caller's scratch; NULL + error text when it is too small
bool exec_has_scratch(size_t bytes);                    // whether the current call brought that much
// Device word holding the bits of max |x[i]| (a positive float orders like an unsigned): one pass per array and entry-point
// call, shared by the launchers below it (the word lives in the last 256 bytes of the caller's scratch); NULL + error text
// when there is no scratch.  Used by the fp16 split (WN_GEMM_FP16X2) to scale operands whose range is not known in advance.
const unsigned* exec_absmax(const float* x, long long n, hipStream_t s);
// a zeroed word of the same kind for a maximum the caller accumulates itself (atomicMax over several arrays); *fresh says
// whether the word is new in this entry-point call (then it has been zeroed on the stream and must be filled) or was
// handed out for the same key before
unsigned* exec_word(const void* key, bool* fresh, hipStream_t s);
// ---- the step plan (plan.hip, ABI 5): weight-only preparation hoisted to the start of a training step -------------------
struct StepPlan;
struct CGArgs;
StepPlan* exec_plan();                 // WnExec.plan of the current call, or NULL
// launch_colgemm_b3: true + the prepared image / range word when a READY plan holds this launch's weight tiles (a recording
// plan registers the job and returns false)
bool plan_split_image(const CGArgs& a, int mode, int mtiles, int cps, int nchunks, int one, size_t bytes,
                      const __bf16** img, const unsigned** wmax);
const void* plan_layer_h2_images(int L, const float* const* Wf, const float* const* Wg, const float* const* Wp);
unsigned* plan_sync_words(int nwords);                  // the multi-layer backward's dataflow words, zeroed by wn_plan_prepare
unsigned* plan_xmax_producer();                         // word that receives max |out| of a GEMM (zeroed by wn_plan_prepare)
void plan_xmax_written(const void* out);                // ... called by the launcher whose kernel really fills it
const unsigned* plan_xmax_consumer(const void* x);      // ... for the call that needs the range of the same array
int generic_absmax(const float* x, long long n, unsigned* slot, hipStream_t s);
int generic_zero_word(unsigned* w, hipStream_t s);        // by a kernel: see generic_kernels.hip
int generic_scale_by_dev(float* x, const float* sdev, long long n, hipStream_t s);
int generic_rule(int rule, float* p, const float* g, float* s1, float* s2, long long n, float lr, float hy, float eps,
                 float wd, const float* sqnorm, float clip, float gmult, const float* lr_dev, hipStream_t s);
int generic_adam(float* p, const float* g, float* m, float* v, long long n, float lr_t, float b1, float b2,
                 float eps, float wd, const float* sqnorm, float clip, float gmult, const float* lr_dev,
                 float dscale, hipStream_t s);

// ---- mfma_layer.hip: fp32-MFMA fused residual layer, Cr = Cd = 32, fw = 2 -------------------
bool mfma_layer_supported(int Cr, int Cd, int fw);
size_t mfma_layer_h2_image_bytes(int L);
int mfma_layer_pack_h2(int L, const float* const* Wf, const float* const* Wg, const float* const* Wp, void* img,
                       hipStream_t s);
bool mfma_layer_fwd_h2_ok(int B, int T, int t_live);
int mfma_layer_fwd_h2(const float* x, const void* img, int l, float* out, float* z, float* fs, float* gs, int B, int T,
                      int d, int Z, int t_live, hipStream_t s);
int mfma_layer_fwd_group_len(const int* dil, int l0, int L);   // layers from l0 on that one group launch can chain
int mfma_layer_fwd_h2_group(const float* x, const void* img, int l0, int nl, float* const* outs, float* const* zs,
                            float* const* fs, float* const* gs, const int* dil, const int* Zs, int B, int T, hipStream_t s);
int mfma_layer_fwd(const float* x, const float* Wf, const float* bf, const float* Wg, const float* bg,
                   const float* Wp, const float* bp, float* out, float* z, float* fs, float* gs, int B, int T,
                   int d, int Z, int t_live, hipStream_t s);

// ---- mfma_layer_bwd.hip: backward of the same shape; bias gradients come from the scratch -----
int mfma_layer_bwd(const float* x, const float* f, const float* g, const float* Wf, const float* Wg,
                   const float* Wp, const float* dout, const float* dzs, float* dx, float* dWf, float* dWg,
                   float* dWp, float* dab, int B, int T, int d, int Z, hipStream_t s);
size_t mfma_layer_bwd_extra_ws_floats();
// Chained backward of the fused layer (mfma_layer_bwd.hip).  dout[t] = Vin[t] + Uin[t + dU], rows below vu_t0 taken as
// 0; dzs (may be NULL) is dz_skip for columns t >= dz_t0; columns below t_live are not computed.
int mfma_layer_bwd_chain(const float* x, const float* f, const float* g, const float* Wf, const float* Wg,
                         const float* Wp, const float* Vin, const float* Uin, int dU, int vu_t0, const float* dzs,
                         int dz_t0, float* Vout, float* Uout, float* part, int B, int T, int d, int Z, int t_live,
                         int* nwg, hipStream_t s, bool from_z = false);   // from_z: `f` holds z, tanh = z / sigmoid
int mfma_chain_multi_max_layers();
int mfma_layer_bwd_chain_multi(int n, const int* layer, const float* const* Wf, const float* const* Wg,
                               const float* const* Wp, const int* d, const int* Z, const int* t_live, const int* vu_t0,
                               const int* dU, const float* x0, const float* xs, const float* z, const float* g,
                               const float* dz, float* const* V, float* const* U, float* part, size_t part_stride,
                               unsigned* sync, int B, int T, int dz_t0, int* nwg, hipStream_t s, bool sync_zeroed = false);
size_t mfma_chain_multi_sync_words(int B, int T);
size_t mfma_chain_part_floats();
int mfma_chain_reduce_all(const float* part, int L, const int* nwg, float* const* dWf, float* const* dWg,
                          float* const* dWp, hipStream_t s, const float* V = nullptr, const float* U = nullptr,
                          float* dx = nullptr, int B = 0, int T = 0, int dU = 0, int vu_t0 = 0);
int mfma_chain_combine(const float* V, const float* U, float* dx, int B, int T, int dU, int vu_t0, hipStream_t s);
int generic_layer_bwd_biases(const float* dab, const float* dout, float* dbf, float* dbg, float* dbp, int B,
                             int T, int Cr, int Cd, int Z, hipStream_t s);

// ---- wide_layer.hip: residual layer for any Cr, Cd multiple of 32 and any fw, composed from the channel GEMMs
bool wide_layer_supported(int Cr, int Cd, int fw);
int wide_layer_fwd(const float* x, const float* Wf, const float* bf, const float* Wg, const float* bg, const float* Wp,
                   const float* bp, float* out, float* z, float* fs, float* gs, int B, int T, int Cr, int Cd, int fw,
                   int d, int Z, hipStream_t s);
int wide_layer_bwd(const float* x, const float* f, const float* g, const float* Wf, const float* Wg, const float* Wp,
                   const float* dout, const float* dzs, float* dx, float* dWf, float* dbf, float* dWg, float* dbg,
                   float* dWp, float* dbp, float* ws, int B, int T, int Cr, int Cd, int fw, int d, int Z,
                   hipStream_t s, const float* z = nullptr);   // z = f g when the caller still has it (dWp reads one tensor)
bool wide_layer_in_use(int Cr, int Cd, int fw);               // api.hip: the per-layer entry points route this shape to wide_layer_*

// ---- mfma_gemm.hip: fp32-MFMA channel GEMMs over time columns (all widths multiples of 32) ---
bool mfma_skip_supported(int L, const int* cd, int Cs);
int mfma_skip_sum_fwd(int L, const float* const* z, const float* const* Ws, const float* const* bs, const int* cd,
                      float* skip, int B, int T, int t_off, int Tw, int Cs, int accumulate, hipStream_t s);
int mfma_skip_bwd_dz(int L, const float* const* Ws, const int* cd, const float* dskip, float* const* dz, int B,
                     int T, int t_off, int Tw, int Cs, bool window_only, hipStream_t s);
bool mfma_pointwise_supported(int Cin, int Cout);
int mfma_head_xent(const float* x, const float* W, const float* bias, const int32_t* target, float* loss, float* dlogits,
                   long long N, int Cin, int Cout, int act, long long n_norm, int ncnt, hipStream_t s);
int mfma_pointwise_fwd(const float* x, const float* W, const float* bias, float* out, long long N, int Cin,
                       int Cout, int act, hipStream_t s);
int mfma_pointwise_bwd_dx(const float* x, const float* W, const float* dout, float* dx, long long N, int Cin,
                          int Cout, int act, hipStream_t s);

int mfma_skip_bwd_dw(int L, const float* const* z, const int* cd, const float* dskip, float* const* dWs, int B, int T,
                     int t_off, int Tw, int Cs, hipStream_t s);
// dbias != NULL: dbias[o] += sum_n dout[n][o] is taken along where the kernel can (then *dbias_done = true)
int mfma_pointwise_bwd_dw(const float* x, const float* dout, float* dW, long long N, int Cin, int Cout, int act,
                          float* dbias, bool* dbias_done, hipStream_t s);

}  // namespace wn
