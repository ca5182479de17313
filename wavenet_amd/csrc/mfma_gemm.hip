// Channel GEMMs over time columns on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   colgemm:  out[n][m] (+)= sum_src sum_k W_src[m][k] * act(X_src[row(n)][k])  (+ sum_src bias_src[m])
//
// used for  (1) the deferred skip sum  sum_l Ws_l z_l  (A11, wavenet.py:574-582: 40 sources of K=32,
// M=256), (2) the head's 1x1 convs (A12, wavenet.py:587-590: relu/elu -> conv), (3) their input
// gradients dx = act'(x) * W^T dout (W read transposed), (4) dz_l = Ws_l^T dskip for all layers in
// one launch ("multi-problem": one source, one 32-wide output per layer).
//
// Tiling: a workgroup = 4 waves = 128 columns x (MT*32) outputs; every wave owns 32 columns and all
// MT output tiles (MT*16 accumulator registers).  The contraction runs in chunks of 32 channels:
// the W chunk [MT*32][32] is staged in LDS already in MFMA A-operand order (one ds_read_b128 feeds
// four MFMAs), the X chunk goes straight from HBM to registers as four float4 per lane using the
// same channel permutation as mfma_layer.hip:  k-step s, lane half h  <->  channel (s&3)+8(s>>2)+4h.
// With that permutation a float4 of four consecutive channels lands in one 16-byte LDS slot, and
// the accumulator layout equals the store layout (float4 stores along the channel axis).
#include "wn_kernels.hpp"
#include "mfma_gemm.hpp"

namespace wn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int cg_ch(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

// MODE 0: multi-source (sum over sources, one output);  MODE 1: one problem per workgroup (M = MT*32 rows);
// MODE 2: MT problems of 32 rows each per workgroup (all share X): X is streamed once per MT problems.
// activation as a template parameter: a runtime switch per element sits inside the chunk loop (see mfma_gemm_b3.hip)
template <int ACT>
__device__ __forceinline__ float cg_act(float x) {
    if (ACT == WN_ACT_RELU) return x > 0.f ? x : 0.f;
    if (ACT == WN_ACT_ELU) return x > 0.f ? x : expm1f(x);
    return x;
}

template <int MT, int MODE, int ACT>
__global__ __launch_bounds__(256, 2) void k_colgemm(CGArgs a) {
    constexpr bool MP = MODE != 0;
    __shared__ __attribute__((aligned(16))) float Alds[MT * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int prob = MODE == 1 ? blockIdx.y : (MODE == 2 ? blockIdx.y * MT : 0);
    const int m0 = MP ? 0 : blockIdx.y * (MT * 32);
    const long long n = ((long long)blockIdx.x * 4 + wave) * 32 + j;
    const bool nvalid = n < a.N;
    long long rb0 = 0;                 // first source row of this column's clip
    int rbase = -(1 << 30);            // row inside the clip before the per-source shift (invalid column: far out)
    long long no = n;                  // output row of this column
    if (nvalid) {
        long long b = n / a.rows_out_per_b;
        rbase = (int)(n - b * a.rows_out_per_b) + a.off;
        rb0 = b * a.rows_src_per_b;
        if (a.out_rows_per_b) no = b * a.out_rows_per_b + a.out_row0 + (n - b * a.rows_out_per_b);
    }
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // fill mapping: thread -> (row i of the tile, 4-channel group c4)
    const int fi = tid & 31, fc4 = tid >> 5;
    const int fq = fc4 >> 1, fh = fc4 & 1;
    float ms = 0.f;                                        // mask of the chunk in xr (set by issue)
    float ms_next = 0.f;

    if (!MP) {
        for (int src = 0; src < a.nsrc; ++src)
            if (a.bias[src]) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][r] += a.bias[src][m0 + mt * 32 + cg_ch(r, h)];
            }
    }

    // Software pipeline over the (source, 32-channel chunk) sequence: the global loads of chunk c+1 (the W
    // slice for LDS and this wave's X columns) are issued BEFORE the MFMAs of chunk c and land behind them.
    float4 wr[MT], xr[4];
    auto issue = [&](int src, int k0) {
        const float* __restrict__ Xb = a.X[src];
        const int K = a.K[src];
        const int rs = rbase + a.soff[src];
        const bool rv = rs >= 0 && rs < a.rows_src_per_b;
        const long long srowc = rb0 + (rv ? rs : 0);       // clamped: loads are unconditional, values masked
        ms_next = rv ? 1.f : 0.f;
#pragma unroll
        for (int it = 0; it < MT; ++it) {
            const float* wp;
            if (MODE == 2) {
                const int pi = prob + it < a.nprob ? prob + it : a.nprob - 1;
                wp = a.W[pi] + (long long)fi * a.wsm[pi] + (long long)(k0 + 4 * fc4) * a.wsk;
            } else {
                const int wi = MODE == 1 ? prob : src;
                wp = a.W[wi] + (long long)(m0 + it * 32 + fi) * a.wsm[wi] + (long long)(k0 + 4 * fc4) * a.wsk;
            }
            if (a.wsk == 1) {
                wr[it] = *reinterpret_cast<const float4*>(wp);
            } else {
                wr[it].x = wp[0]; wr[it].y = wp[a.wsk]; wr[it].z = wp[2 * (long long)a.wsk]; wr[it].w = wp[3 * (long long)a.wsk];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) xr[q] = *reinterpret_cast<const float4*>(Xb + srowc * K + k0 + 8 * q + 4 * h);
    };
    int src = 0, k0 = 0;
    issue(0, 0);
    while (true) {
        __syncthreads();                                     // the previous chunk's MFMAs are done with Alds
#pragma unroll
        for (int it = 0; it < MT; ++it)
            *reinterpret_cast<float4*>(&Alds[(((it * 4 + fq) * 64) + fi + 32 * fh) * 4]) = wr[it];
        ms = ms_next;
        float xb[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            xb[4 * q + 0] = cg_act<ACT>(xr[q].x) * ms; xb[4 * q + 1] = cg_act<ACT>(xr[q].y) * ms;
            xb[4 * q + 2] = cg_act<ACT>(xr[q].z) * ms; xb[4 * q + 3] = cg_act<ACT>(xr[q].w) * ms;
        }
        __syncthreads();
        k0 += 32;
        if (k0 >= a.K[src]) { k0 = 0; ++src; }
        const bool more = src < a.nsrc;
        if (more) issue(src, k0);                            // in flight during the MFMAs below
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float4 a4 = *reinterpret_cast<const float4*>(&Alds[((mt * 4 + q) * 64 + lane) * 4]);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, xb[4 * q + 0], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, xb[4 * q + 1], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, xb[4 * q + 2], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, xb[4 * q + 3], acc[mt], 0, 0, 0);
            }
        }
        if (!more) break;
    }
    if (!nvalid) return;
    float* __restrict__ orow = a.out[MODE == 2 ? 0 : prob] + no * a.ldo + m0 + 4 * h;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (MODE == 2) {
            if (prob + mt >= a.nprob) break;
            orow = a.out[prob + mt] + no * a.ldo + 4 * h - mt * 32;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = make_float4(acc[mt][4 * q], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]);
            float* p = orow + mt * 32 + 8 * q;
            if (a.gate_x) {
                float4 gx = *reinterpret_cast<const float4*>(a.gate_x + no * a.ldo + m0 + 4 * h + mt * 32 + 8 * q);
                v.x *= act_grad(gx.x, a.gate_act); v.y *= act_grad(gx.y, a.gate_act);
                v.z *= act_grad(gx.z, a.gate_act); v.w *= act_grad(gx.w, a.gate_act);
            }
            if (a.residual && MODE == 0) {
                const float4 rr = *reinterpret_cast<const float4*>(a.residual + no * a.ldo + m0 + 4 * h + mt * 32 + 8 * q);
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            if (a.accumulate) {
                float4 o = *reinterpret_cast<const float4*>(p);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *reinterpret_cast<float4*>(p) = v;
        }
    }
}

template <bool MP>
static int launch_colgemm(CGArgs& a, int nprob, hipStream_t s) {
    const int M = a.M;
    a.nprob = nprob;
    if (gemm_b3_enabled()) {
        int rc = launch_colgemm_b3(a, MP ? (M == 32 ? 2 : 1) : 0, nprob, s);
        if (rc != WN_ESHAPE) return rc;                 // WN_ESHAPE = shape not covered: use the exact-fp32 kernels
    }
    if (MP && M == 32 && nprob > 1) {          // 32-row problems: 8 per workgroup, X streamed once per 8
        dim3 grid(cdiv(a.N, 128), cdiv(nprob, 8));
        if (a.act != WN_ACT_NONE) { wn::set_error("colgemm: multi-problem mode takes no activation"); return WN_EARG; }
        hipLaunchKernelGGL((k_colgemm<8, 2, WN_ACT_NONE>), grid, dim3(256), 0, s, a);
        WN_LAUNCH_CHECK();
        return WN_OK;
    }
    int mt = (M % 256 == 0) ? 8 : (M % 128 == 0) ? 4 : (M % 64 == 0) ? 2 : 1;
    if (MP) mt = M / 32;      // multi-problem: the whole (small) M in one workgroup
    dim3 grid(cdiv(a.N, 128), MP ? nprob : M / (mt * 32));
    constexpr int MODE = MP ? 1 : 0;
#define CG_LAUNCH(MT_)                                                                                              \
    do {                                                                                                            \
        if (a.act == WN_ACT_RELU) hipLaunchKernelGGL((k_colgemm<MT_, MODE, WN_ACT_RELU>), grid, dim3(256), 0, s, a);     \
        else if (a.act == WN_ACT_ELU) hipLaunchKernelGGL((k_colgemm<MT_, MODE, WN_ACT_ELU>), grid, dim3(256), 0, s, a);  \
        else hipLaunchKernelGGL((k_colgemm<MT_, MODE, WN_ACT_NONE>), grid, dim3(256), 0, s, a);                     \
    } while (0)
    switch (mt) {
        case 8: CG_LAUNCH(8); break;
        case 4: CG_LAUNCH(4); break;
        case 2: CG_LAUNCH(2); break;
        case 1: CG_LAUNCH(1); break;
        default: wn::set_error("colgemm: unsupported M=%d", M); return WN_ESHAPE;
    }
#undef CG_LAUNCH
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int launch_colgemm_multi(CGArgs& a, hipStream_t s) { return launch_colgemm<false>(a, 1, s); }

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- entry points used by api.hip -------------------------------------------------------------
bool mfma_skip_supported(int L, const int* cd, int Cs) {
    if (Cs % 32) return false;
    for (int l = 0; l < L; ++l)
        if (cd[l] % 32) return false;
    return true;
}

int mfma_skip_sum_fwd(int L, const float* const* z, const float* const* Ws, const float* const* bs, const int* cd,
                      float* skip, int B, int T, int t_off, int Tw, int Cs, int accumulate, hipStream_t s) {
    for (int l0 = 0; l0 < L; l0 += WN_MAX_SRC) {
        CGArgs a{};
        a.nsrc = (L - l0 < WN_MAX_SRC) ? L - l0 : WN_MAX_SRC;
        for (int l = 0; l < a.nsrc; ++l) {
            a.X[l] = z[l0 + l]; a.W[l] = Ws[l0 + l]; a.bias[l] = bs ? bs[l0 + l] : nullptr;
            a.K[l] = cd[l0 + l]; a.wsm[l] = cd[l0 + l];
            if (!aligned16(a.X[l]) || !aligned16(a.W[l])) { wn::set_error("skip_sum: pointers must be 16-byte aligned"); return WN_EARG; }
        }
        a.out[0] = skip; a.wsk = 1; a.M = Cs; a.ldo = Cs; a.N = (long long)B * Tw;
        a.rows_out_per_b = Tw; a.rows_src_per_b = T; a.off = t_off;
        a.act = WN_ACT_NONE; a.gate_x = nullptr; a.gate_act = 0;
        a.accumulate = (accumulate || l0 > 0) ? 1 : 0;
        a.h2_ok = 1;                                     // z = tanh * sigmoid lies in [-1, 1]
        int rc = launch_colgemm<false>(a, 1, s);
        if (rc) return rc;
    }
    return WN_OK;
}

// window_only: compute (and write) columns t >= t_off only; the rows below are left untouched
int mfma_skip_bwd_dz(int L, const float* const* Ws, const int* cd, const float* dskip, float* const* dz, int B,
                     int T, int t_off, int Tw, int Cs, bool window_only, hipStream_t s) {
    // every layer is cd/32 problems of 32 output rows sharing X = dskip; batches of <= WN_MAX_SRC problems with
    // one row stride (ldo = cd) per launch
    int l = 0, j = 0;                 // next layer, next 32-row tile inside it
    while (l < L) {
        CGArgs a{};
        const int width = cd[l];
        int np = 0;
        while (l < L && cd[l] == width && np < WN_MAX_SRC) {
            a.W[np] = Ws[l] + 32 * j; a.wsm[np] = 1; a.out[np] = dz[l] + 32 * j; ++np;
            if (++j == width / 32) { j = 0; ++l; }
        }
        a.nsrc = 1; a.X[0] = dskip; a.K[0] = Cs; a.bias[0] = nullptr;
        a.wsk = width;                                    // W[m][k] = Ws[k][m]
        a.M = 32; a.ldo = width;
        if (window_only) {
            a.N = (long long)B * Tw; a.rows_out_per_b = Tw; a.rows_src_per_b = Tw; a.off = 0;
            a.out_rows_per_b = T; a.out_row0 = t_off;
        } else {
            a.N = (long long)B * T; a.rows_out_per_b = T; a.rows_src_per_b = Tw; a.off = -t_off;
        }
        a.act = WN_ACT_NONE; a.gate_x = nullptr; a.accumulate = 0;
        if (gemm_mode() == WN_GEMM_FP16X2 && !(a.xmax_dev = exec_absmax(dskip, (long long)B * Tw * Cs, s))) return WN_EARG;
        int rc = launch_colgemm<true>(a, np, s);
        if (rc) return rc;
    }
    return WN_OK;
}

bool mfma_pointwise_supported(int Cin, int Cout) { return Cin % 32 == 0 && Cout % 32 == 0; }

int mfma_pointwise_fwd(const float* x, const float* W, const float* bias, float* out, long long N, int Cin,
                       int Cout, int act, hipStream_t s) {
    CGArgs a{};
    a.nsrc = 1; a.X[0] = x; a.W[0] = W; a.bias[0] = bias; a.K[0] = Cin; a.wsm[0] = Cin; a.wsk = 1;
    a.out[0] = out; a.M = Cout; a.ldo = Cout; a.N = N;
    a.rows_out_per_b = (int)(N < (1ll << 30) ? N : (1ll << 30)); a.rows_src_per_b = a.rows_out_per_b; a.off = 0;
    if (N >= (1ll << 30)) { wn::set_error("pointwise: N too large"); return WN_ESHAPE; }
    a.act = act; a.gate_x = nullptr; a.accumulate = 0;
    // (the head keeps the six-term split under WN_GEMM_FP16X2: its input is bounded only by the weights, and measuring the
    // range -- a 100 MB pass per call -- costs more than the fp16 split saves on a contraction this small: 0.095 -> 0.111 ms)
    return launch_colgemm<false>(a, 1, s);
}

// The last head convolution and the loss in one launch (fp16 split, 256 outputs): dlogits = d loss / d (W act(x) + b), the
// workgroups' loss sums in loss[kXentPart ..] (the caller finalises).  WN_ESHAPE when the shape or the arithmetic mode is not
// covered: the caller then runs the convolution and the loss as two calls.
int mfma_head_xent(const float* x, const float* W, const float* bias, const int32_t* target, float* loss, float* dlogits,
                   long long N, int Cin, int Cout, int act, long long n_norm, int ncnt, hipStream_t s) {
    if (Cout != 256 || Cin % 32 || N >= (1ll << 30) || gemm_mode() != WN_GEMM_FP16X2) return WN_ESHAPE;
    CGArgs a{};
    a.nsrc = 1; a.X[0] = x; a.W[0] = W; a.bias[0] = bias; a.K[0] = Cin; a.wsm[0] = Cin; a.wsk = 1;
    a.out[0] = dlogits; a.M = Cout; a.ldo = Cout; a.N = N;
    a.rows_out_per_b = (int)N; a.rows_src_per_b = (int)N; a.off = 0;
    a.act = act; a.gate_x = nullptr; a.accumulate = 0;
    a.xent_target = target; a.xent_loss = loss; a.xent_n_norm = n_norm; a.xent_ncnt = ncnt;
    return launch_colgemm_b3(a, 6, 1, s);
}

// dx[n][c] = act'(x[n][c]) * sum_o W[o][c] dout[n][o]
int mfma_pointwise_bwd_dx(const float* x, const float* W, const float* dout, float* dx, long long N, int Cin,
                          int Cout, int act, hipStream_t s) {
    CGArgs a{};
    a.nsrc = 1; a.X[0] = dout; a.W[0] = W; a.bias[0] = nullptr; a.K[0] = Cout; a.wsm[0] = 1; a.wsk = Cin;
    a.out[0] = dx; a.M = Cin; a.ldo = Cin; a.N = N;
    if (N >= (1ll << 30)) { wn::set_error("pointwise: N too large"); return WN_ESHAPE; }
    a.rows_out_per_b = (int)N; a.rows_src_per_b = (int)N; a.off = 0;
    a.act = WN_ACT_NONE; a.gate_x = (act == WN_ACT_NONE) ? nullptr : x; a.gate_act = act; a.accumulate = 0;
    // step plan: the next consumer of dx (the dz contraction of the skip path, whose fp16 split needs max |dskip|) finds the
    // range in a plan-owned word instead of making a pass over the array (exec_absmax -> plan_xmax_consumer)
    a.outmax_dev = plan_xmax_producer();
    return launch_colgemm<false>(a, 1, s);
}

}  // namespace wn

// =============================================================================================
// wgrad:  dW_p[m][k] += sum_n A[n][m] * act(B_p[row(n)][k])      (contraction over time columns)
//
// used for dWs_l = dskip^T z_l of all layers in one launch (A15; 40 "problems" sharing A) and for the
// head convs' dW = dout^T act(x) (problems = 32-wide column tiles of x).  The MFMA K dimension is
// TIME: a workgroup stages 32 rows of A (all M channels) in LDS, every wave owns one problem (one
// 32-wide tile of B, read straight from HBM in channel-on-lane order) and all MT row tiles of dW.
// Each workgroup covers a slab of rows of one clip and leaves with one set of float atomics.
// =============================================================================================
namespace wn {


template <int MT, bool HAS_B2, int ACT>
__global__ __launch_bounds__(256, 2) void k_wgrad_mfma(WGArgs a) {
    __shared__ __attribute__((aligned(16))) float Alds[32 * MT * 32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int p = blockIdx.y * 4 + wv;
    const bool active = p < a.nprob;
    const int m0 = blockIdx.z * (MT * 32);
    const int b = blockIdx.x / a.wgs_per_b;
    const int r_begin = (blockIdx.x - b * a.wgs_per_b) * a.rows_per_wg;
    const int r_end = min(a.rows_A_per_b, r_begin + a.rows_per_wg);
    const float* __restrict__ Ab = a.A + ((long long)b * a.rows_A_per_b) * a.lda + m0;
    // inactive waves read problem 0 (masked): every load below is unconditional -- a runtime condition around a load
    // makes hipcc branch and drain vmcnt per element (see k_wgrad_b3w in mfma_gemm_b3.hip)
    const float* __restrict__ Bb = a.Bp[active ? p : 0] + ((long long)b * a.rows_B_per_b + a.off) * a.ldb + j;
    const float* __restrict__ B2b = HAS_B2 ? a.B2p[active ? p : 0] + ((long long)b * a.rows_B_per_b + a.off) * a.ldb + j : Bb;
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // software pipeline over 32-row chunks: the loads of chunk c+1 are issued before the MFMAs of chunk c
    float4 ar[MT];
    float br[16];
    auto issue = [&](int r0) {
#pragma unroll
        for (int it = 0; it < MT; ++it) {
            const int idx = it * 256 + tid;                 // float4 index inside the [32][MT*32] chunk
            const int row = idx / (MT * 8), c4 = idx - row * (MT * 8);
            const int rr = r0 + row < r_end ? r0 + row : r_end - 1;              // clamped row, masked below
            float4 v = *reinterpret_cast<const float4*>(Ab + (long long)rr * a.lda + 4 * c4);
            const float m = r0 + row < r_end ? 1.f : 0.f;
            v.x *= m; v.y *= m; v.z *= m; v.w *= m;
            ar[it] = v;
        }
        float raw[16], raw2[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int r = r0 + 2 * s + h;
            int rc = r < r_end ? r : r_end - 1;                              // clamped row, masked value
            if (rc + a.off < 0) rc = -a.off;
            if (rc + a.off >= a.rows_B_per_b) rc = a.rows_B_per_b - 1 - a.off;
            raw[s] = Bb[(long long)rc * a.ldb];
            if (HAS_B2) raw2[s] = B2b[(long long)rc * a.ldb];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int r = r0 + 2 * s + h;
            const int rb = r + a.off;
            const bool ok = active && r < r_end && rb >= 0 && rb < a.rows_B_per_b;
            float v = cg_act<ACT>(raw[s]);
            if (HAS_B2) v *= raw2[s];
            br[s] = ok ? v : 0.f;
        }
    };
    if (r_begin < r_end) issue(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += 32) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < MT; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx / (MT * 8), c4 = idx - row * (MT * 8);
            *reinterpret_cast<float4*>(&Alds[row * (MT * 32) + 4 * c4]) = ar[it];
        }
        float bv[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) bv[s] = br[s];
        __syncthreads();
        if (r0 + 32 < r_end) issue(r0 + 32);                 // in flight during the MFMAs below
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float av = Alds[(2 * s + h) * (MT * 32) + mt * 32 + j];
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[s], acc[mt], 0, 0, 0);
            }
        }
    }
    if (!active) return;
    float* __restrict__ o = a.out[p];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            atomicAdd(o + (long long)(m0 + mt * 32 + cg_ch(r, h)) * a.ldo + (long long)j * (a.osk ? a.osk : 1), acc[mt][r]);
}

int launch_wgrad(WGArgs& a, int M, hipStream_t s);
static int launch_wgrad_mfma(WGArgs& a, int M, hipStream_t s) { return launch_wgrad(a, M, s); }
int launch_wgrad(WGArgs& a, int M, hipStream_t s) {
    int mt = (M % 256 == 0) ? 8 : (M % 128 == 0) ? 4 : (M % 64 == 0) ? 2 : 1;
    // row slabs: enough workgroups to fill the chip, few enough that the atomics stay small
    int groups = cdiv(a.nprob, 4) * (M / (mt * 32));
    int want = 1024 / (groups > 0 ? groups : 1);
    if (want < 1) want = 1;
    int per_b = cdiv(want, a.nB);
    int rows = cdiv(a.rows_A_per_b, per_b);
    rows = ((rows + 31) / 32) * 32;
    if (rows < 256) rows = 256;
    a.rows_per_wg = rows;
    a.wgs_per_b = cdiv(a.rows_A_per_b, rows);
    dim3 grid(a.nB * a.wgs_per_b, cdiv(a.nprob, 4), M / (mt * 32));
    if (gemm_b3_enabled() && M == 256 && a.nprob >= 8) return launch_wgrad_b3w(a, s);
    if (gemm_b3_enabled() && M % 256 == 0 && a.nprob >= 8) {      // config 5's 512 skip channels: one wide launch per 256 rows of A
        for (int m0 = 0; m0 < M; m0 += 256) {
            WGArgs h = a;
            h.A = a.A + m0;
            for (int q = 0; q < a.nprob; ++q) h.out[q] = a.out[q] + (long long)m0 * a.ldo;
            h.colsum = a.colsum ? a.colsum + m0 : nullptr;
            h.colsum_done = 0;
            int rc = launch_wgrad_b3w(h, s);
            if (rc) return rc;
            // Whether the wide kernel takes the column sums depends on the arithmetic mode only (six-term form: yes; one-term
            // bf16 / fp16x2: no), so every 256-row slice must answer alike.  "None did" is fine: colsum_done stays 0 and the
            // caller sums the bias gradient itself.  A mix would leave dbias half written.
            if (a.colsum && m0 > 0 && h.colsum_done != a.colsum_done) {
                wn::set_error("wgrad: the column sums were taken for a part of the rows only");
                return WN_EARG;
            }
            a.colsum_done = h.colsum_done;
        }
        return WN_OK;
    }
    if (gemm_b3_enabled()) {          // bf16x3: at most 4 row tiles per workgroup (register budget), more groups in z
        const int mt3 = mt > 4 ? 4 : mt;
        grid.z = M / (mt3 * 32);
        return launch_wgrad_b3(a, mt3, grid, s);
    }
    bool any_b2 = false, all_b2 = true;
    for (int q = 0; q < a.nprob; ++q) { any_b2 |= a.B2p[q] != nullptr; all_b2 &= a.B2p[q] != nullptr; }
    if (any_b2 != all_b2) { wn::set_error("wgrad: the B2 factor must be given for all problems or for none"); return WN_EARG; }
#define WGF_LAUNCH(MT_, B2_)                                                                                       \
    do {                                                                                                           \
        if (a.act == WN_ACT_RELU) hipLaunchKernelGGL((k_wgrad_mfma<MT_, B2_, WN_ACT_RELU>), grid, dim3(256), 0, s, a);   \
        else if (a.act == WN_ACT_ELU) hipLaunchKernelGGL((k_wgrad_mfma<MT_, B2_, WN_ACT_ELU>), grid, dim3(256), 0, s, a); \
        else hipLaunchKernelGGL((k_wgrad_mfma<MT_, B2_, WN_ACT_NONE>), grid, dim3(256), 0, s, a);                  \
    } while (0)
#define WG_LAUNCH(MT_)                                                                               \
    do {                                                                                             \
        if (any_b2) WGF_LAUNCH(MT_, true);                                                           \
        else WGF_LAUNCH(MT_, false);                                                                 \
    } while (0)
    switch (mt) {
        case 8: WG_LAUNCH(8); break;
        case 4: WG_LAUNCH(4); break;
        case 2: WG_LAUNCH(2); break;
        default: WG_LAUNCH(1); break;
    }
#undef WG_LAUNCH
#undef WGF_LAUNCH
    WN_LAUNCH_CHECK();
    return WN_OK;
}

// dWs[l][cs][cd] += sum dskip[b,t',cs] z_l[b,t_off+t',cd]   (cd any multiple of 32: one problem per 32 columns)
int mfma_skip_bwd_dw(int L, const float* const* z, const int* cd, const float* dskip, float* const* dWs, int B, int T,
                     int t_off, int Tw, int Cs, hipStream_t s) {
    int l = 0, j = 0;
    while (l < L) {
        WGArgs a{};
        a.A = dskip; a.lda = Cs; a.nprob = 0;
        const int width = cd[l];
        while (l < L && cd[l] == width && a.nprob < WN_MAX_SRC) {
            if (dWs[l]) { a.Bp[a.nprob] = z[l] + 32 * j; a.B2p[a.nprob] = nullptr; a.out[a.nprob] = dWs[l] + 32 * j; ++a.nprob; }
            if (++j == width / 32) { j = 0; ++l; }
        }
        if (a.nprob == 0) continue;
        a.ldb = width; a.ldo = width; a.osk = 1;
        a.nB = B; a.rows_A_per_b = Tw; a.rows_B_per_b = T; a.off = t_off; a.act = WN_ACT_NONE;
        if (gemm_mode() == WN_GEMM_FP16X2) {             // A = dskip: measured range; B = z in [-1, 1]
            if (!(a.amax_dev = exec_absmax(dskip, (long long)B * Tw * Cs, s))) return WN_EARG;
            a.h2 = 1;
        }
        int rc = launch_wgrad(a, Cs, s);
        if (rc) return rc;
    }
    return WN_OK;
}

// dW[o][c] += sum_n dout[n][o] act(x[n][c])
int mfma_pointwise_bwd_dw(const float* x, const float* dout, float* dW, long long N, int Cin, int Cout, int act,
                          float* dbias, bool* dbias_done, hipStream_t s) {
    if (N >= (1ll << 30)) { wn::set_error("pointwise dW: N too large"); return WN_ESHAPE; }
    if (dbias_done) *dbias_done = false;
    for (int c0 = 0; c0 < Cin; c0 += 32 * WN_MAX_SRC) {
        WGArgs a{};
        a.A = dout; a.lda = Cout;
        a.colsum = (c0 == 0 && dbias_done) ? dbias : nullptr;     // the column sums of dout: once, with the first group
        a.nprob = 0;
        for (int c = c0; c < Cin && a.nprob < WN_MAX_SRC; c += 32) {
            a.Bp[a.nprob] = x + c; a.out[a.nprob] = dW + c; ++a.nprob;
        }
        a.ldb = Cin; a.ldo = Cin;
        a.nB = 1; a.rows_A_per_b = (int)N; a.rows_B_per_b = (int)N; a.off = 0; a.act = act;
        int rc = launch_wgrad_mfma(a, Cout, s);
        if (rc) return rc;
        if (a.colsum_done && dbias_done) *dbias_done = true;
    }
    return WN_OK;
}

}  // namespace wn
