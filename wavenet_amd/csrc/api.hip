// extern "C" surface of libwavenet_hip.so: argument checks, then dispatch to the MFMA kernels for
// the shapes they cover and to the generic kernels for everything else.  No allocation, no sync.
#include <string.h>

#include "wn_kernels.hpp"

namespace wn {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static thread_local const WnExec* g_exec = nullptr;
static thread_local int g_exec_depth = 0;
static thread_local const void* g_absmax_key[64];          // one per word of the scratch tail (kExecTail / 4)
static thread_local int g_absmax_n = 0;
static constexpr size_t kExecTail = 256;                  // bytes at the end of the scratch kept for the absmax words
ExecScope::ExecScope(const WnExec* ex) : prev(g_exec) { g_exec = ex; ++g_exec_depth; }
ExecScope::~ExecScope() {
    g_exec = prev;
    if (--g_exec_depth == 0) g_absmax_n = 0;              // the outermost entry point returns: nothing survives the call
}
static bool force_generic();
int gemm_mode() {
    if (force_generic()) return WN_GEMM_FP32;
    const int m = g_exec ? g_exec->precision : WN_GEMM_BF16X3;
    return (m < WN_GEMM_FP32 || m > WN_GEMM_FP16X2) ? WN_GEMM_BF16X3 : m;
}
StepPlan* exec_plan() { return g_exec ? reinterpret_cast<StepPlan*>(g_exec->plan) : nullptr; }
const unsigned* exec_absmax(const float* x, long long n, hipStream_t s) {
    // a READY step plan: the GEMM that wrote x left max |x| in a plan-owned word (the head's dx = dskip): no pass over x
    if (const unsigned* w = plan_xmax_consumer(x)) return w;
    if (!g_exec || !g_exec->ws || g_exec->ws_bytes < kExecTail) {
        set_error("this call needs WnExec scratch (the fp16 split scales its operands by their measured range)");
        return nullptr;
    }
    unsigned* slots = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(g_exec->ws) + g_exec->ws_bytes - kExecTail);
    for (int i = 0; i < g_absmax_n; ++i)
        if (g_absmax_key[i] == x) return slots + i;
    if (g_absmax_n >= 64) { set_error("exec_absmax: more than 64 range words in one call"); return nullptr; }
    if (generic_absmax(x, n, slots + g_absmax_n, s) != WN_OK) return nullptr;
    g_absmax_key[g_absmax_n] = x;
    return slots + g_absmax_n++;
}
unsigned* exec_word(const void* key, bool* fresh, hipStream_t s) {
    *fresh = false;
    if (!g_exec || !g_exec->ws || g_exec->ws_bytes < kExecTail) {
        set_error("this call needs WnExec scratch (the fp16 split scales its operands by their measured range)");
        return nullptr;
    }
    unsigned* slots = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(g_exec->ws) + g_exec->ws_bytes - kExecTail);
    for (int i = 0; i < g_absmax_n; ++i)
        if (g_absmax_key[i] == key) return slots + i;
    if (g_absmax_n >= 64) { set_error("exec_word: more than 64 range words in one call"); return nullptr; }
    if (generic_zero_word(slots + g_absmax_n, s) != WN_OK) return nullptr;      // (a kernel, not a memset node: generic_kernels.hip)
    g_absmax_key[g_absmax_n] = key;
    *fresh = true;
    return slots + g_absmax_n++;
}
bool exec_has_scratch(size_t bytes) { return g_exec && g_exec->ws && g_exec->ws_bytes >= bytes + kExecTail; }
void* exec_scratch(size_t bytes, const char* what) {
    bytes += kExecTail;
    if (!g_exec || !g_exec->ws || g_exec->ws_bytes < bytes) {
        set_error("this call needs %zu bytes of WnExec scratch for %s (got %zu): size it with wn_exec_workspace_bytes()",
                  bytes, what, g_exec && g_exec->ws ? g_exec->ws_bytes : (size_t)0);
        return nullptr;
    }
    return g_exec->ws;
}
// WnExec.flags of the current call (no process state: the library reads no environment variable)
bool exec_flag(unsigned f) { return g_exec && (g_exec->flags & f) != 0; }
static bool force_generic() { return exec_flag(WN_EXEC_FORCE_GENERIC); }
int exec_fwd_t1_min_blocks() {
    const int v = g_exec ? g_exec->fwd_t1_min_blocks : 0;
    return v == 0 ? 512 : v;
}
// the fused 32-channel kernels for this shape, unless the call pins the generic path
bool layer_fast_path(int Cr, int Cd, int fw) { return !force_generic() && mfma_layer_supported(Cr, Cd, fw); }
}  // namespace wn

using namespace wn;

#define POS(x) WN_CHECK_ARG((x) > 0, "%s: %s must be positive (got %d)", __func__, #x, (int)(x))
#define NN(p) WN_CHECK_ARG((p) != nullptr, "%s: %s is NULL", __func__, #p)

extern "C" {

int wn_abi_version(void) { return WN_ABI_VERSION; }
const char* wn_last_error(void) { return g_err; }

int wn_layer_fast_path(int Cr, int Cd, int fw) { return mfma_layer_supported(Cr, Cd, fw) ? 1 : 0; }

int wn_embed_fwd(const int32_t* idx, const float* W, const float* bias, float* out, int B, int T, int Q, int C,
                 int fw, void* stream) {
    wn::ProfScope prof__("wn_embed_fwd", stream);
    NN(idx); NN(W); NN(out); POS(B); POS(T); POS(Q); POS(C); POS(fw);
    return generic_embed_fwd(idx, W, bias, out, B, T, Q, C, fw, as_stream(stream));
}

int wn_embed_bwd(const int32_t* idx, const float* dout, float* dW, float* dbias, int B, int T, int Q, int C,
                 int fw, const WnExec* ex, void* stream) {
    wn::ExecScope exec__(ex);
    WN_CHECK_ARG(!ex || (ex->precision >= WN_GEMM_FP32 && ex->precision <= WN_GEMM_FP16X2), "%s: WnExec.precision must be 0 .. 3", __func__);
    wn::ProfScope prof__("wn_embed_bwd", stream);
    NN(idx); NN(dout); NN(dW); POS(B); POS(T); POS(Q); POS(C); POS(fw);
    return generic_embed_bwd(idx, dout, dW, dbias, B, T, Q, C, fw, as_stream(stream));
}

int wn_conv_fwd(const float* x, const float* W, const float* bias, float* out, int B, int T, int Cin, int Cout,
                int fw, int d, int Z, void* stream) {
    wn::ProfScope prof__("wn_conv_fwd", stream);
    NN(x); NN(W); NN(out); POS(B); POS(T); POS(Cin); POS(Cout); POS(fw); POS(d);
    WN_CHECK_ARG(Z >= 0, "wn_conv_fwd: Z < 0");
    return generic_conv_fwd(x, W, bias, out, B, T, Cin, Cout, fw, d, Z, as_stream(stream));
}

int wn_conv_bwd(const float* x, const float* W, const float* dout, float* dx, float* dW, float* dbias, int B,
                int T, int Cin, int Cout, int fw, int d, int Z, void* stream) {
    wn::ProfScope prof__("wn_conv_bwd", stream);
    NN(x); NN(W); NN(dout); POS(B); POS(T); POS(Cin); POS(Cout); POS(fw); POS(d);
    WN_CHECK_ARG(Z >= 0, "wn_conv_bwd: Z < 0");
    return generic_conv_bwd(x, W, dout, dx, dW, dbias, B, T, Cin, Cout, fw, d, Z, as_stream(stream));
}

int wn_layer_fwd(const float* x, const float* Wf, const float* bf, const float* Wg, const float* bg,
                 const float* Wp, const float* bp, float* out, float* z, float* f_save, float* g_save, int B,
                 int T, int Cr, int Cd, int fw, int d, int Z, const WnExec* ex, void* stream) {
    wn::ExecScope exec__(ex);
    WN_CHECK_ARG(!ex || (ex->precision >= WN_GEMM_FP32 && ex->precision <= WN_GEMM_FP16X2), "%s: WnExec.precision must be 0 .. 3", __func__);
    wn::ProfScope prof__("wn_layer_fwd", stream);
    NN(x); NN(Wf); NN(Wg); NN(Wp); NN(out); NN(z);
    POS(B); POS(T); POS(Cr); POS(Cd); POS(fw); POS(d);
    WN_CHECK_ARG(Z >= 0, "wn_layer_fwd: Z < 0");
    WN_CHECK_ARG((f_save == nullptr) == (g_save == nullptr), "wn_layer_fwd: f_save and g_save go together");
    WN_CHECK_ARG(out != x, "wn_layer_fwd: out must not alias x (taps read x[t-d])");
    if (layer_fast_path(Cr, Cd, fw))
        return mfma_layer_fwd(x, Wf, bf, Wg, bg, Wp, bp, out, z, f_save, g_save, B, T, d, Z, 0, as_stream(stream));
    if (!force_generic() && wide_layer_supported(Cr, Cd, fw) && (f_save || Cd <= Cr))
        return wide_layer_fwd(x, Wf, bf, Wg, bg, Wp, bp, out, z, f_save, g_save, B, T, Cr, Cd, fw, d, Z,
                              as_stream(stream));
    return generic_layer_fwd(x, Wf, bf, Wg, bg, Wp, bp, out, z, f_save, g_save, B, T, Cr, Cd, fw, d, Z,
                             as_stream(stream));
}

size_t wn_layer_bwd_workspace_floats(int B, int T, int Cr, int Cd, int fw) {
    if (B <= 0 || T <= 0 || Cd <= 0) return 0;
    size_t n = (size_t)B * T * 2 * Cd;
    if (wn_layer_fast_path(Cr, Cd, fw)) n += mfma_layer_bwd_extra_ws_floats();
    return n;
}

}  // extern "C"
namespace wn {
bool wide_layer_in_use(int Cr, int Cd, int fw) {
    return !layer_fast_path(Cr, Cd, fw) && !force_generic() && wide_layer_supported(Cr, Cd, fw);
}
}  // namespace wn
extern "C" {

int wn_layer_bwd(const float* x, const float* f, const float* g, const float* Wf, const float* Wg,
                 const float* Wp, const float* dout, const float* dz_skip, float* dx, float* dWf, float* dbf,
                 float* dWg, float* dbg, float* dWp, float* dbp, float* dab_ws, int B, int T, int Cr, int Cd,
                 int fw, int d, int Z, const WnExec* ex, void* stream) {
    wn::ExecScope exec__(ex);
    WN_CHECK_ARG(!ex || (ex->precision >= WN_GEMM_FP32 && ex->precision <= WN_GEMM_FP16X2), "%s: WnExec.precision must be 0 .. 3", __func__);
    wn::ProfScope prof__("wn_layer_bwd", stream);
    NN(x); NN(f); NN(g); NN(Wf); NN(Wg); NN(Wp); NN(dab_ws);
    POS(B); POS(T); POS(Cr); POS(Cd); POS(fw); POS(d);
    WN_CHECK_ARG(Z >= 0, "wn_layer_bwd: Z < 0");
    WN_CHECK_ARG(dout || dz_skip, "wn_layer_bwd: both dout and dz_skip are NULL");
    if (layer_fast_path(Cr, Cd, fw)) {
        int rc = mfma_layer_bwd(x, f, g, Wf, Wg, Wp, dout, dz_skip, dx, dWf, dWg, dWp, dab_ws, B, T, d, Z,
                                as_stream(stream));
        if (rc) return rc;
        return generic_layer_bwd_biases(dab_ws, dout, dbf, dbg, dbp, B, T, Cr, Cd, Z, as_stream(stream));
    }
    if (!force_generic() && wide_layer_supported(Cr, Cd, fw))
        return wide_layer_bwd(x, f, g, Wf, Wg, Wp, dout, dz_skip, dx, dWf, dbf, dWg, dbg, dWp, dbp, dab_ws, B, T, Cr, Cd,
                              fw, d, Z, as_stream(stream));
    return generic_layer_bwd(x, f, g, Wf, Wg, Wp, dout, dz_skip, dx, dWf, dbf, dWg, dbg, dWp, dbp, dab_ws, B, T,
                             Cr, Cd, fw, d, Z, as_stream(stream));
}

int wn_pointwise_fwd(const float* x, const float* W, const float* bias, float* out, int N, int Cin, int Cout,
                     int act, const WnExec* ex, void* stream) {
    wn::ExecScope exec__(ex);
    WN_CHECK_ARG(!ex || (ex->precision >= WN_GEMM_FP32 && ex->precision <= WN_GEMM_FP16X2), "%s: WnExec.precision must be 0 .. 3", __func__);
    wn::ProfScope prof__("wn_pointwise_fwd", stream);
    NN(x); NN(W); NN(out); POS(N); POS(Cin); POS(Cout);
    WN_CHECK_ARG(act >= WN_ACT_NONE && act <= WN_ACT_ELU, "wn_pointwise_fwd: bad act %d", act);
    if (!force_generic() && mfma_pointwise_supported(Cin, Cout))
        return mfma_pointwise_fwd(x, W, bias, out, N, Cin, Cout, act, as_stream(stream));
    return generic_pointwise_fwd(x, W, bias, out, N, Cin, Cout, act, as_stream(stream));
}

int wn_pointwise_bwd(const float* x, const float* W, const float* dout, float* dx, float* dW, float* dbias,
                     int N, int Cin, int Cout, int act, const WnExec* ex, void* stream) {
    wn::ExecScope exec__(ex);
    WN_CHECK_ARG(!ex || (ex->precision >= WN_GEMM_FP32 && ex->precision <= WN_GEMM_FP16X2), "%s: WnExec.precision must be 0 .. 3", __func__);
    wn::ProfScope prof__("wn_pointwise_bwd", stream);
    NN(x); NN(W); NN(dout); POS(N); POS(Cin); POS(Cout);
    WN_CHECK_ARG(act >= WN_ACT_NONE && act <= WN_ACT_ELU, "wn_pointwise_bwd: bad act %d", act);
    if (dx && !force_generic() && mfma_pointwise_supported(Cin, Cout)) {
        int rc = mfma_pointwise_bwd_dx(x, W, dout, dx, N, Cin, Cout, act, as_stream(stream));
        if (rc) return rc;
        dx = nullptr;
    }
    if (dW && !force_generic() && mfma_pointwise_supported(Cin, Cout)) {
        bool bias_done = false;
        int rc = mfma_pointwise_bwd_dw(x, dout, dW, N, Cin, Cout, act, dbias, &bias_done, as_stream(stream));
        if (rc) return rc;
        dW = nullptr;
        if (bias_done) dbias = nullptr;
    }
    return generic_pointwise_bwd(x, W, dout, dx, dW, dbias, N, Cin, Cout, act, as_stream(stream));
}

static int check_skip(const char* fn, int L, int B, int T, int t_off, int Tw, int Cs) {
    WN_CHECK_ARG(L > 0 && B > 0 && T > 0 && Tw > 0 && Cs > 0, "%s: non-positive size", fn);
    WN_CHECK_ARG(t_off >= 0 && t_off + Tw <= T, "%s: columns [%d,%d) outside [0,%d)", fn, t_off, t_off + Tw, T);
    return WN_OK;
}

int wn_skip_sum_fwd(int L, const float* const* z, const float* const* Ws, const float* const* bs, const int* cd,
                    float* skip, int B, int T, int t_off, int Tw, int Cs, int accumulate, const WnExec* ex, void* stream) {
    wn::ExecScope exec__(ex);
    WN_CHECK_ARG(!ex || (ex->precision >= WN_GEMM_FP32 && ex->precision <= WN_GEMM_FP16X2), "%s: WnExec.precision must be 0 .. 3", __func__);
    wn::ProfScope prof__("wn_skip_sum_fwd", stream);
    NN(z); NN(Ws); NN(cd); NN(skip);
    int rc = check_skip("wn_skip_sum_fwd", L, B, T, t_off, Tw, Cs);
    if (rc) return rc;
    for (int l = 0; l < L; ++l) WN_CHECK_ARG(z[l] && Ws[l] && cd[l] > 0, "wn_skip_sum_fwd: bad source %d", l);
    if (!force_generic() && mfma_skip_supported(L, cd, Cs))
        return mfma_skip_sum_fwd(L, z, Ws, bs, cd, skip, B, T, t_off, Tw, Cs, accumulate, as_stream(stream));
    return generic_skip_sum_fwd(L, z, Ws, bs, cd, skip, B, T, t_off, Tw, Cs, accumulate, as_stream(stream));
}

int wn_skip_sum_bwd_dz(int L, const float* const* Ws, const int* cd, const float* dskip, float* const* dz, int B,
                       int T, int t_off, int Tw, int Cs, const WnExec* ex, void* stream) {
    wn::ExecScope exec__(ex);
    WN_CHECK_ARG(!ex || (ex->precision >= WN_GEMM_FP32 && ex->precision <= WN_GEMM_FP16X2), "%s: WnExec.precision must be 0 .. 3", __func__);
    wn::ProfScope prof__("wn_skip_sum_bwd_dz", stream);
    NN(Ws); NN(cd); NN(dskip); NN(dz);
    int rc = check_skip("wn_skip_sum_bwd_dz", L, B, T, t_off, Tw, Cs);
    if (rc) return rc;
    for (int l = 0; l < L; ++l) WN_CHECK_ARG(dz[l] && Ws[l] && cd[l] > 0, "wn_skip_sum_bwd_dz: bad entry %d", l);
    bool fast = !force_generic() && Cs % 32 == 0;
    for (int l = 0; l < L && fast; ++l) fast = cd[l] % 32 == 0;
    if (fast) return mfma_skip_bwd_dz(L, Ws, cd, dskip, dz, B, T, t_off, Tw, Cs, false, as_stream(stream));
    return generic_skip_bwd_dz(L, Ws, cd, dskip, dz, B, T, t_off, Tw, Cs, as_stream(stream));
}

int wn_skip_sum_bwd_dw(int L, const float* const* z, const int* cd, const float* dskip, float* const* dWs,
                       float* const* dbs, int B, int T, int t_off, int Tw, int Cs, const WnExec* ex, void* stream) {
    wn::ExecScope exec__(ex);
    WN_CHECK_ARG(!ex || (ex->precision >= WN_GEMM_FP32 && ex->precision <= WN_GEMM_FP16X2), "%s: WnExec.precision must be 0 .. 3", __func__);
    wn::ProfScope prof__("wn_skip_sum_bwd_dw", stream);
    NN(z); NN(cd); NN(dskip);
    int rc = check_skip("wn_skip_sum_bwd_dw", L, B, T, t_off, Tw, Cs);
    if (rc) return rc;
    bool fast = !force_generic() && Cs % 32 == 0 && dWs;
    for (int l = 0; l < L && fast; ++l) fast = cd[l] % 32 == 0;
    if (fast) {
        rc = mfma_skip_bwd_dw(L, z, cd, dskip, dWs, B, T, t_off, Tw, Cs, as_stream(stream));
        if (rc) return rc;
        dWs = nullptr;
    }
    return generic_skip_bwd_dw(L, z, cd, dskip, dWs, dbs, B, T, t_off, Tw, Cs, as_stream(stream));
}

int wn_softmax_fwd(const float* logits, float* prob, int N, int Q, void* stream) {
    wn::ProfScope prof__("wn_softmax_fwd", stream);
    NN(logits); NN(prob); POS(N); POS(Q);
    return generic_softmax(logits, prob, N, Q, as_stream(stream));
}

int wn_softmax_xent(const float* logits, const int32_t* target, float* loss, float* dlogits, int N, int Q,
                    int64_t n_norm, void* stream) {
    wn::ProfScope prof__("wn_softmax_xent", stream);
    NN(logits); NN(target); NN(loss); POS(N); POS(Q);
    return generic_softmax_xent(logits, target, loss, dlogits, N, Q, n_norm, as_stream(stream));
}

int wn_head_xent_supported(int64_t N, int Cin, int Cout, const WnExec* ex) {
    wn::ExecScope exec__(ex);
    return (N > 0 && (N + 127) / 128 <= wn::kXentBlocks && !force_generic() && gemm_mode() == WN_GEMM_FP16X2 && Cout == 256 && Cin > 0 && Cin % 32 == 0 && mfma_pointwise_supported(Cin, Cout)) ? 1 : 0;
}

int wn_head_xent(const float* x, const float* W, const float* bias, const int32_t* target, float* loss, float* dlogits,
                 int N, int Cin, int Cout, int act, int64_t n_norm, const WnExec* ex, void* stream) {
    wn::ExecScope exec__(ex);
    WN_CHECK_ARG(!ex || (ex->precision >= WN_GEMM_FP32 && ex->precision <= WN_GEMM_FP16X2), "%s: WnExec.precision must be 0 .. 3", __func__);
    wn::ProfScope prof__("wn_head_xent", stream);
    NN(x); NN(W); NN(target); NN(loss); NN(dlogits); POS(N); POS(Cin); POS(Cout);
    WN_CHECK_ARG(act >= WN_ACT_NONE && act <= WN_ACT_ELU, "wn_head_xent: bad act %d", act);
    WN_CHECK_SHAPE(wn_head_xent_supported(N, Cin, Cout, ex), "wn_head_xent: needs WN_GEMM_FP16X2, 256 outputs, a multiple of 32 "
                                                          "inputs and at most 253,952 rows (run wn_pointwise_fwd + wn_softmax_xent instead)");
    hipStream_t s = as_stream(stream);
    const long long nn = n_norm > 0 ? n_norm : (n_norm == 0 ? N : -1);
    int ncnt = 0, rc;
    if (nn < 0 && (rc = generic_xent_count(target, N, Cout, loss, &ncnt, s))) return rc;
    if ((rc = mfma_head_xent(x, W, bias, target, loss, dlogits, N, Cin, Cout, act, nn, ncnt, s))) return rc;
    return generic_xent_final(loss, (int)((N + 127) / 128), nn, ncnt, s);
}

int wn_nchw_to_btc(const float* src, float* dst, int B, int C, int T, void* stream) {
    wn::ProfScope prof__("wn_nchw_to_btc", stream);
    NN(src); NN(dst); POS(B); POS(C); POS(T);
    return generic_transpose(src, dst, B, C, T, as_stream(stream));
}

int wn_btc_to_nchw(const float* src, float* dst, int B, int C, int T, void* stream) {
    wn::ProfScope prof__("wn_btc_to_nchw", stream);
    NN(src); NN(dst); POS(B); POS(C); POS(T);
    return generic_transpose(src, dst, B, T, C, as_stream(stream));
}

int wn_sample_categorical(const float* prob, const double* uniforms, int32_t* out, int n, int Q, void* stream) {
    wn::ProfScope prof__("wn_sample_categorical", stream);
    NN(prob); NN(uniforms); NN(out); POS(n); POS(Q);
    return generic_sample(prob, uniforms, out, n, Q, as_stream(stream));
}

int wn_mulaw_encode_pcm16(const int16_t* pcm, const int32_t* lut65536, int32_t* tokens, int64_t n, void* stream) {
    NN(pcm); NN(lut65536); NN(tokens);
    WN_CHECK_ARG(n > 0, "wn_mulaw_encode_pcm16: n <= 0");
    return generic_mulaw_encode_pcm16(pcm, lut65536, tokens, n, as_stream(stream));
}

int wn_mulaw_decode(const int32_t* tokens, const float* table, float* out, int64_t n, int Q, void* stream) {
    NN(tokens); NN(table); NN(out);
    WN_CHECK_ARG(n > 0 && Q > 0, "wn_mulaw_decode: n <= 0 or Q <= 0");
    return generic_mulaw_decode(tokens, table, out, n, Q, as_stream(stream));
}

int wn_sqnorm(const float* grad, const float* param, int64_t n, float grad_mult, float weight_decay, float* out,
              void* stream) {
    wn::ProfScope prof__("wn_sqnorm", stream);
    NN(grad); NN(out);
    WN_CHECK_ARG(n > 0, "wn_sqnorm: n <= 0");
    WN_CHECK_ARG(weight_decay == 0.f || param, "wn_sqnorm: weight decay needs param");
    return generic_sqnorm(grad, param, n, grad_mult, weight_decay, out, as_stream(stream));
}

int wn_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1,
                 float beta2, float eps, float weight_decay, const float* sqnorm, float clip, float grad_mult,
                 void* stream) {
    wn::ProfScope prof__("wn_adam_step", stream);
    NN(param); NN(grad); NN(m); NN(v);
    WN_CHECK_ARG(n > 0, "wn_adam_step: n <= 0");
    return generic_adam(param, grad, m, v, n, lr_t, beta1, beta2, eps, weight_decay, sqnorm, clip, grad_mult,
                        nullptr, 1.f, as_stream(stream));
}

int wn_eve_step(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                float eps, float d, float weight_decay, const float* sqnorm, float clip, float grad_mult,
                void* stream) {
    wn::ProfScope prof__("wn_adam_step", stream);
    NN(param); NN(grad); NN(m); NN(v);
    WN_CHECK_ARG(n > 0 && d > 0.f, "wn_eve_step: n <= 0 or d <= 0");
    return generic_adam(param, grad, m, v, n, lr_t, beta1, beta2, eps, weight_decay, sqnorm, clip, grad_mult,
                        nullptr, d, as_stream(stream));
}

int wn_adam_step_dev(float* param, const float* grad, float* m, float* v, int64_t n, const float* lr_t_dev,
                     float beta1, float beta2, float eps, float weight_decay, const float* sqnorm, float clip,
                     float grad_mult, void* stream) {
    wn::ProfScope prof__("wn_adam_step", stream);
    NN(param); NN(grad); NN(m); NN(v); NN(lr_t_dev);
    WN_CHECK_ARG(n > 0, "wn_adam_step_dev: n <= 0");
    return generic_adam(param, grad, m, v, n, 0.f, beta1, beta2, eps, weight_decay, sqnorm, clip, grad_mult,
                        lr_t_dev, 1.f, as_stream(stream));
}

int wn_rule_step(int rule, float* param, const float* grad, float* s1, float* s2, int64_t n, float lr,
                 const float* lr_dev, float hyper, float eps, float weight_decay, const float* sqnorm, float clip,
                 float grad_mult, void* stream) {
    wn::ProfScope prof__("wn_adam_step", stream);
    NN(param); NN(grad);
    WN_CHECK_ARG(n > 0, "wn_rule_step: n <= 0");
    WN_CHECK_ARG(rule >= WN_RULE_SGD && rule <= WN_RULE_RMSPROP, "wn_rule_step: unknown rule");
    WN_CHECK_ARG(rule == WN_RULE_SGD || s1, "wn_rule_step: this rule needs s1");
    WN_CHECK_ARG(rule != WN_RULE_ADADELTA || s2, "wn_rule_step: AdaDelta needs s2");
    return generic_rule(rule, param, grad, s1, s2, n, lr, hyper, eps, weight_decay, sqnorm, clip, grad_mult, lr_dev,
                        as_stream(stream));
}

int wn_scale_by_dev(float* x, const float* scale_dev, int64_t n, void* stream) {
    wn::ProfScope prof__("wn_softmax_xent", stream);
    NN(x); NN(scale_dev);
    WN_CHECK_ARG(n > 0, "wn_scale_by_dev: n <= 0");
    return generic_scale_by_dev(x, scale_dev, n, as_stream(stream));
}

// Upper bound of the scratch any entry point asks for on this model (see the three users: launch_colgemm_b3's split
// weight image, launch_wgrad_b3w's partial tiles, generic_embed_bwd's per-block tables).
size_t wn_exec_workspace_bytes(const WnStackDesc* d, int Q, int causal_channels, int causal_fw, const int* head_channels,
                               int n_head_channels, int B, int T) {
    size_t mk = 0;                                          // largest (rows x contraction) product of a channel GEMM
    size_t nprob = 8;
    auto upd = [&](size_t m, size_t k) { if (m * k > mk) mk = m * k; };
    if (d && d->n_layers > 0 && d->cd) {
        size_t sum_cd = 0, max_cd = 0;
        for (int l = 0; l < d->n_layers; ++l) { sum_cd += d->cd[l]; if ((size_t)d->cd[l] > max_cd) max_cd = d->cd[l]; }
        upd(d->Cs, sum_cd);                                 // skip sum, dz of all layers
        upd(2 * max_cd, (size_t)d->fw * d->Cr);             // wide layers: gate GEMM (filter and gate rows interleaved)
        upd(d->Cr, 2 * max_cd * d->fw);                     // their dx
        upd(max_cd, d->Cr);
        nprob = sum_cd / 32 + d->Cs / 32 + 8;
    }
    for (int i = 0; i + 1 < n_head_channels; ++i) {
        upd(head_channels[i + 1], head_channels[i]);
        nprob += head_channels[i] / 32 + head_channels[i + 1] / 32;
    }
    const size_t image = 12 * mk + (64u << 10);             // 6 KB per 32 x 32 tile, twice in gate mode, + the 128/128 Wp image
    const size_t nB = B > 0 ? (size_t)B : 1;
    const size_t parts = (256 + nB * ((nprob + 7) / 8) + 64) * (8 * 8 * 16 * 64) * sizeof(float);
    // embedding gradient: 256 per-workgroup tables (+ the bias sums, + two padded token planes of the one-hot form)
    const size_t tables = causal_channels > 0
                              ? (size_t)257 * ((size_t)Q * causal_fw * causal_channels + causal_channels) * sizeof(float) +
                                    (size_t)2 * nB * ((size_t)(T > 0 ? T : 0) + 128) * sizeof(int32_t) + 1024
                              : 0;
    // bias gradients: one row of column sums per 256-row chunk (at most 2,048 chunks)
    size_t maxm = Q > 0 ? (size_t)Q : 0;
    if (d && d->n_layers > 0) maxm = maxm > (size_t)d->Cs ? maxm : (size_t)d->Cs;
    for (int i = 0; i < n_head_channels; ++i) maxm = maxm > (size_t)head_channels[i] ? maxm : (size_t)head_channels[i];
    const size_t colsums = (size_t)2048 * (maxm > 512 ? maxm : 512) * sizeof(float);
    size_t need = image > parts ? image : parts;
    if (tables > need) need = tables;
    if (colsums > need) need = colsums;
    return ((need + 4095) & ~(size_t)4095) + 4096;
}

}  // extern "C"
