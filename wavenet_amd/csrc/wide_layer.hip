// Residual layer for channel widths the fused 32/32/2 kernels do not cover -- any Cr, Cd that are
// multiples of 32 and any filter width (BASELINE config 5's 128/512, the reference's default 128/32,
// _tests_' filter width 3 ...): the same maths as ResidualConvLayer.__call__ (wavenet.py:358-368) and its
// backward, composed from the matrix-core channel GEMMs of mfma_gemm*.hip plus two elementwise kernels.
//
//   forward   a = sum_k Wf_k x[t-(fw-1-k)d] (+bf)   g likewise        multi-source GEMM, one source per tap
//             z = tanh(a) sigmoid(g), zero prefix                      k_gate
//             out = Wp z + bp + x                                       GEMM with a residual epilogue
//   backward  dz = Wp^T dout + dz_skip                                  GEMM on the transposed view of Wp
//             da = dz g (1-f^2), dg = dz f g (1-g), zero prefix         k_gate_bwd
//             dx = dout + sum_k Wf_k^T da[t+(fw-1-k)d] + Wg_k^T dg[..]  multi-source GEMM, 2 fw sources
//             dWp, dWf_k, dWg_k                                         weight-gradient GEMMs (contraction over time)
//             biases                                                    column sums
#include <stdlib.h>

#include "mfma_gemm.hpp"

namespace wn {

__global__ void k_wide_gate(float* __restrict__ a, float* __restrict__ g, float* __restrict__ z, float* __restrict__ fs,
                            float* __restrict__ gs, long long n4, int T, int Cd, int Z) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // float4 index
    if (i >= n4) return;
    const int t = (int)((i * 4 / Cd) % T);
    float4 av = reinterpret_cast<const float4*>(a)[i], gv = reinterpret_cast<const float4*>(g)[i];
    if (t < Z) { av = make_float4(0, 0, 0, 0); gv = av; }                      // reference zero prefix
    const float4 f = make_float4(fast_tanh(av.x), fast_tanh(av.y), fast_tanh(av.z), fast_tanh(av.w));
    const float4 s = make_float4(fast_sigmoid(gv.x), fast_sigmoid(gv.y), fast_sigmoid(gv.z), fast_sigmoid(gv.w));
    reinterpret_cast<float4*>(z)[i] = make_float4(f.x * s.x, f.y * s.y, f.z * s.z, f.w * s.w);
    if (fs) { reinterpret_cast<float4*>(fs)[i] = f; reinterpret_cast<float4*>(gs)[i] = s; }
}

// ldo4 = output row stride in float4 (Cd/4: da and dg are separate (n, Cd) arrays; Cd/2: they are the two halves of one
// (n, 2 Cd) array and dg = da + Cd)
__global__ void k_wide_gate_bwd(const float* __restrict__ dz, const float* __restrict__ f, const float* __restrict__ g,
                                float* __restrict__ da, float* __restrict__ dg, long long n4, int T, int Cd, int Z,
                                int ldo4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const long long row = i * 4 / Cd;
    const long long io = row * ldo4 + (i - row * (Cd / 4));
    const int t = (int)(row % T);
    float4 d = reinterpret_cast<const float4*>(dz)[i];
    if (t < Z) d = make_float4(0, 0, 0, 0);
    const float4 fv = reinterpret_cast<const float4*>(f)[i], gv = reinterpret_cast<const float4*>(g)[i];
    reinterpret_cast<float4*>(da)[io] = make_float4(d.x * gv.x * (1.f - fv.x * fv.x), d.y * gv.y * (1.f - fv.y * fv.y),
                                                   d.z * gv.z * (1.f - fv.z * fv.z), d.w * gv.w * (1.f - fv.w * fv.w));
    reinterpret_cast<float4*>(dg)[io] = make_float4(d.x * fv.x * gv.x * (1.f - gv.x), d.y * fv.y * gv.y * (1.f - gv.y),
                                                   d.z * fv.z * gv.z * (1.f - gv.z), d.w * fv.w * gv.w * (1.f - gv.w));
}

bool wide_layer_supported(int Cr, int Cd, int fw) { return Cr % 32 == 0 && Cd % 32 == 0 && 2 * fw <= WN_MAX_SRC; }

static void base_args(CGArgs& a, int B, int T) {
    a.N = (long long)B * T; a.rows_out_per_b = T; a.rows_src_per_b = T; a.off = 0;
    a.act = WN_ACT_NONE; a.gate_x = nullptr; a.gate_act = 0; a.residual = nullptr; a.accumulate = 0;
}

// one dilated conv: out[b,t,:] = sum_k W[:, :, k] x[b, t-(fw-1-k)d, :] + bias
static int conv_gemm(const float* x, const float* W, const float* bias, float* out, int B, int T, int Cin, int Cout,
                     int fw, int d, hipStream_t s) {
    CGArgs a{};
    base_args(a, B, T);
    a.nsrc = fw;
    for (int k = 0; k < fw; ++k) {
        a.X[k] = x; a.K[k] = Cin; a.W[k] = W + k; a.wsm[k] = Cin * fw; a.soff[k] = -(fw - 1 - k) * d;
        a.bias[k] = k == 0 ? bias : nullptr;
    }
    a.wsk = fw; a.M = Cout; a.ldo = Cout; a.out[0] = out;
    return launch_colgemm_multi(a, s);
}

int wide_layer_fwd(const float* x, const float* Wf, const float* bf, const float* Wg, const float* bg, const float* Wp,
                   const float* bp, float* out, float* z, float* fs, float* gs, int B, int T, int Cr, int Cd, int fw,
                   int d, int Z, hipStream_t s) {
    // pre-activations go where f / sigmoid(g) will live (training) or into z / out (inference, needs Cd <= Cr)
    int rc;
    if (gemm_b3_enabled() && Cd % 64 == 0) {
        // one launch for both convolutions and the gate: the rows of Wf and Wg interleaved tile by tile, tanh / sigmoid /
        // product in the epilogue -- x is read once per tap, the pre-activations never reach memory
        CGArgs a{};
        base_args(a, B, T);
        a.nsrc = fw;
        for (int k = 0; k < fw; ++k) {
            a.X[k] = x; a.K[k] = Cr; a.W[k] = Wf + k; a.W2[k] = Wg + k; a.wsm[k] = Cr * fw; a.soff[k] = -(fw - 1 - k) * d;
            a.bias[k] = k == 0 ? bf : nullptr; a.bias2[k] = k == 0 ? bg : nullptr;
        }
        a.wsk = fw; a.M = Cd; a.ldo = Cd; a.out[0] = z;
        a.gate_z = z; a.gate_f = fs; a.gate_s = gs; a.gate_Z = Z;
        const bool no_fused = exec_flag(WN_EXEC_NO_FUSED_WIDE);                          // diagnostic switch
        if (!no_fused && gemm_mode() == 2 && Cr == 128 && Cd == 128) {
            // bf16 operands, 128/128 channels (config 5): the residual projection runs in the same kernel, on z taken from
            // registers -- one launch per layer, z is written but never read back
            a.ldo = Cr; a.out[0] = out; a.residual = x; a.proj_W = Wp; a.proj_bias = bp;
            return launch_colgemm_b3(a, 5, 1, s);
        }
        if ((rc = launch_colgemm_b3(a, 3, 1, s))) return rc;
        CGArgs b{};
        base_args(b, B, T);
        b.nsrc = 1; b.X[0] = z; b.K[0] = Cd; b.W[0] = Wp; b.wsm[0] = Cd; b.wsk = 1; b.bias[0] = bp;
        b.M = Cr; b.ldo = Cr; b.out[0] = out; b.residual = x;
        return launch_colgemm_multi(b, s);
    }
    float* abuf = fs ? fs : z;
    float* gbuf = gs ? gs : out;
    if (!gs && Cd > Cr) { wn::set_error("wide_layer_fwd: inference needs Cd <= Cr"); return WN_ESHAPE; }
    rc = conv_gemm(x, Wf, bf, abuf, B, T, Cr, Cd, fw, d, s);
    if (rc) return rc;
    rc = conv_gemm(x, Wg, bg, gbuf, B, T, Cr, Cd, fw, d, s);
    if (rc) return rc;
    const long long n4 = (long long)B * T * Cd / 4;
    hipLaunchKernelGGL(k_wide_gate, dim3(cdiv(n4, 256)), dim3(256), 0, s, abuf, gbuf, z, fs, gs, n4, T, Cd, Z);
    WN_LAUNCH_CHECK();
    CGArgs a{};
    base_args(a, B, T);
    a.nsrc = 1; a.X[0] = z; a.K[0] = Cd; a.W[0] = Wp; a.wsm[0] = Cd; a.wsk = 1; a.bias[0] = bp;
    a.M = Cr; a.ldo = Cr; a.out[0] = out; a.residual = x;
    return launch_colgemm_multi(a, s);
}

// dW[o][c][k] += sum_n A[n][o] * x[n - (fw-1-k)d][c]
static int conv_wgrad(const float* A, const float* x, float* dW, int B, int T, int Cin, int Cout, int fw, int d,
                      hipStream_t s) {
    for (int k = 0; k < fw; ++k)
        for (int c0 = 0; c0 < Cin; c0 += 32 * WN_MAX_SRC) {
            WGArgs a{};
            a.A = A; a.lda = Cout; a.nprob = 0;
            for (int c = c0; c < Cin && a.nprob < WN_MAX_SRC; c += 32) {
                a.Bp[a.nprob] = x + c; a.B2p[a.nprob] = nullptr; a.out[a.nprob] = dW + (long long)c * fw + k; ++a.nprob;
            }
            a.ldb = Cin; a.ldo = Cin * fw; a.osk = fw;
            a.nB = B; a.rows_A_per_b = T; a.rows_B_per_b = T; a.off = -(fw - 1 - k) * d; a.act = WN_ACT_NONE;
            int rc = launch_wgrad(a, Cout, s);
            if (rc) return rc;
        }
    return WN_OK;
}

// Cd == 128 (config 5): da and dg live side by side in ONE (n, 256) array, so that
//   * dz = Wp^T dout + dz_skip and the gate derivative are one launch (GEMM with the gate-backward epilogue),
//   * all 2 fw weight-gradient contractions of the layer (dWf_k, dWg_k: 256 rows of [da | dg] against the 32-channel
//     slices of x at fw tap shifts) are ONE launch of the wide 256-row block, which reads [da | dg] once instead of
//     2 fw times.
static int wide_layer_bwd_256(const float* x, const float* f, const float* g, const float* Wf, const float* Wg,
                              const float* Wp, const float* dout, const float* dzs, float* dx, float* dWf, float* dbf,
                              float* dWg, float* dbg, float* dWp, float* dbp, float* ws, int B, int T, int Cr, int Cd,
                              int fw, int d, int Z, const float* z, hipStream_t s) {
    const long long n = (long long)B * T;
    float* dadg = ws;                // (B,T,2 Cd)
    int rc;
    if (dout) {
        CGArgs a{};
        base_args(a, B, T);
        a.nsrc = 1; a.X[0] = dout; a.K[0] = Cr; a.W[0] = Wp; a.wsm[0] = 1; a.wsk = Cd; a.bias[0] = nullptr;
        a.M = Cd; a.ldo = Cd; a.out[0] = dadg; a.residual = dzs;
        a.gate_f = const_cast<float*>(f); a.gate_s = const_cast<float*>(g); a.gate_z = dadg; a.gate_Z = Z;
        if ((rc = launch_colgemm_b3(a, 4, 1, s))) return rc;
    } else {
        const long long n4 = n * Cd / 4;
        hipLaunchKernelGGL(k_wide_gate_bwd, dim3(cdiv(n4, 256)), dim3(256), 0, s, dzs, f, g, dadg, dadg + Cd, n4, T, Cd, Z,
                           Cd / 2);
        WN_LAUNCH_CHECK();
    }
    if (dx) {                        // dx = dout + sum_k Wf_k^T da[t+(fw-1-k)d] + Wg_k^T dg[t+(fw-1-k)d]
        CGArgs a{};
        base_args(a, B, T);
        a.nsrc = 2 * fw;
        a.ldx = 2 * Cd;
        for (int k = 0; k < fw; ++k)
            for (int w = 0; w < 2; ++w) {
                const int i = 2 * k + w;
                a.X[i] = dadg + w * Cd; a.K[i] = Cd; a.W[i] = (w ? Wg : Wf) + k; a.wsm[i] = fw;
                a.soff[i] = (fw - 1 - k) * d; a.bias[i] = nullptr;
            }
        a.wsk = Cr * fw; a.M = Cr; a.ldo = Cr; a.out[0] = dx; a.residual = dout;
        if ((rc = launch_colgemm_multi(a, s))) return rc;
    }
    {                                // dWf[o][c][k] += sum da[n][o] x[n-(fw-1-k)d][c];  dWg likewise from the dg half
        WGArgs a{};
        a.A = dadg; a.lda = 2 * Cd; a.nprob = 0; a.m_split = Cd;
        for (int k = 0; k < fw; ++k)
            for (int c = 0; c < Cr; c += 32) {
                const int q = a.nprob++;
                a.Bp[q] = x + c; a.B2p[q] = nullptr; a.offp[q] = -(fw - 1 - k) * d;
                a.out[q] = dWf + (long long)c * fw + k; a.out2[q] = dWg + (long long)c * fw + k;
            }
        a.ldb = Cr; a.ldo = Cr * fw; a.osk = fw;
        a.nB = B; a.rows_A_per_b = T; a.rows_B_per_b = T; a.off = 0; a.act = WN_ACT_NONE;
        if ((rc = launch_wgrad_b3w(a, s))) return rc;
    }
    if (dWp && dout) {               // dWp[cr][cd] += sum dout[n][cr] * (f g)[n][cd]
        WGArgs a{};
        a.A = dout; a.lda = Cr; a.nprob = 0;
        for (int c = 0; c < Cd; c += 32) {       // z itself when the caller still has it (one tensor instead of f and g)
            a.Bp[a.nprob] = (z ? z : f) + c; a.B2p[a.nprob] = z ? nullptr : g + c; a.out[a.nprob] = dWp + c; ++a.nprob;
        }
        a.ldb = Cd; a.ldo = Cd; a.osk = 1;
        a.nB = B; a.rows_A_per_b = T; a.rows_B_per_b = T; a.off = 0; a.act = WN_ACT_NONE;
        if ((rc = launch_wgrad(a, Cr, s))) return rc;
    }
    if (dbf && (rc = generic_colsum(dadg, B, T, 0, 2 * Cd, Cd, dbf, s))) return rc;
    if (dbg && (rc = generic_colsum(dadg + Cd, B, T, 0, 2 * Cd, Cd, dbg, s))) return rc;
    if (dbp && dout && (rc = generic_colsum(dout, B, T, 0, Cr, Cr, dbp, s))) return rc;
    return WN_OK;
}

int wide_layer_bwd(const float* x, const float* f, const float* g, const float* Wf, const float* Wg, const float* Wp,
                   const float* dout, const float* dzs, float* dx, float* dWf, float* dbf, float* dWg, float* dbg,
                   float* dWp, float* dbp, float* ws, int B, int T, int Cr, int Cd, int fw, int d, int Z,
                   hipStream_t s, const float* z) {
    const long long n = (long long)B * T;
    if (gemm_b3_enabled() && 2 * Cd == 256 && Cr % 32 == 0 && (Cr / 32) * fw <= 8 && dWf && dWg)
        return wide_layer_bwd_256(x, f, g, Wf, Wg, Wp, dout, dzs, dx, dWf, dbf, dWg, dbg, dWp, dbp, ws, B, T, Cr, Cd, fw, d,
                                  Z, z, s);
    float* da = ws;                  // (B,T,Cd): dz first, then da in place
    float* dg = ws + n * Cd;         // (B,T,Cd)
    int rc;
    const float* dz = dzs;
    if (dout) {                      // dz = Wp^T dout + dz_skip
        CGArgs a{};
        base_args(a, B, T);
        a.nsrc = 1; a.X[0] = dout; a.K[0] = Cr; a.W[0] = Wp; a.wsm[0] = 1; a.wsk = Cd; a.bias[0] = nullptr;
        a.M = Cd; a.ldo = Cd; a.out[0] = da; a.residual = dzs;
        if ((rc = launch_colgemm_multi(a, s))) return rc;
        dz = da;
    }
    const long long n4 = n * Cd / 4;
    hipLaunchKernelGGL(k_wide_gate_bwd, dim3(cdiv(n4, 256)), dim3(256), 0, s, dz, f, g, da, dg, n4, T, Cd, Z, Cd / 4);
    WN_LAUNCH_CHECK();
    if (dx) {                        // dx = dout + sum_k Wf_k^T da[t+(fw-1-k)d] + Wg_k^T dg[t+(fw-1-k)d]
        CGArgs a{};
        base_args(a, B, T);
        a.nsrc = 2 * fw;
        for (int k = 0; k < fw; ++k)
            for (int w = 0; w < 2; ++w) {
                const int i = 2 * k + w;
                a.X[i] = w ? dg : da; a.K[i] = Cd; a.W[i] = (w ? Wg : Wf) + k; a.wsm[i] = fw;
                a.soff[i] = (fw - 1 - k) * d; a.bias[i] = nullptr;
            }
        a.wsk = Cr * fw; a.M = Cr; a.ldo = Cr; a.out[0] = dx; a.residual = dout;
        if ((rc = launch_colgemm_multi(a, s))) return rc;
    }
    if (dWf && (rc = conv_wgrad(da, x, dWf, B, T, Cr, Cd, fw, d, s))) return rc;
    if (dWg && (rc = conv_wgrad(dg, x, dWg, B, T, Cr, Cd, fw, d, s))) return rc;
    if (dWp && dout) {               // dWp[cr][cd] += sum dout[n][cr] * (f g)[n][cd]
        for (int c0 = 0; c0 < Cd; c0 += 32 * WN_MAX_SRC) {
            WGArgs a{};
            a.A = dout; a.lda = Cr; a.nprob = 0;
            for (int c = c0; c < Cd && a.nprob < WN_MAX_SRC; c += 32) {
                a.Bp[a.nprob] = (z ? z : f) + c; a.B2p[a.nprob] = z ? nullptr : g + c; a.out[a.nprob] = dWp + c; ++a.nprob;
            }
            a.ldb = Cd; a.ldo = Cd; a.osk = 1;
            a.nB = B; a.rows_A_per_b = T; a.rows_B_per_b = T; a.off = 0; a.act = WN_ACT_NONE;
            if ((rc = launch_wgrad(a, Cr, s))) return rc;
        }
    }
    if (dbf && (rc = generic_colsum(da, B, T, 0, Cd, Cd, dbf, s))) return rc;     // da, dg are already 0 for t < Z
    if (dbg && (rc = generic_colsum(dg, B, T, 0, Cd, Cd, dbg, s))) return rc;
    if (dbp && dout && (rc = generic_colsum(dout, B, T, 0, Cr, Cr, dbp, s))) return rc;
    return WN_OK;
}

}  // namespace wn
