// Shape-generic fp32 kernels: correct for every Params the reference accepts (any channel widths,
// any filter width, biases on or off).  One thread per output element, weights through L1/L2.
// The shapes BASELINE.json names take the MFMA kernels in mfma_*.hip instead; these are the
// fallback for everything else and the on-GPU cross-check of the fast kernels.
#include <cstdlib>

#include "wn_common.hpp"

namespace wn {
void* exec_scratch(size_t bytes, const char* what);
bool exec_has_scratch(size_t bytes);       // api.hip: the current call's WnExec scratch

static constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------------
// A10  embedding form of the first causal layer
// ---------------------------------------------------------------------------------------------
__global__ void k_embed_fwd(const int32_t* __restrict__ idx, const float* __restrict__ W,
                            const float* __restrict__ bias, float* __restrict__ out,
                            int B, int T, int Q, int C, int fw) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)B * T * C;
    if (i >= total) return;
    int c = (int)(i % C);
    long long bt = i / C;
    int t = (int)(bt % T);
    int b = (int)(bt / T);
    float acc = bias ? bias[c] : 0.f;
    for (int k = 0; k < fw; ++k) {
        int ts = t - (fw - 1 - k);
        if (ts < 0) continue;
        int q = idx[(long long)b * T + ts];
        acc += W[((long long)c * Q + q) * fw + k];
    }
    out[i] = acc;
}

// Same gather for the common case C % 4 == 0, fw <= 2: one thread = four channels of one column, 32-bit index maths,
// the (at most two) tokens read once per column group.  W is (C, Q, fw): the four channels are Q*fw floats apart.
__global__ void k_embed_fwd4(const int32_t* __restrict__ idx, const float* __restrict__ W,
                             const float* __restrict__ bias, float* __restrict__ out,
                             int ncol, int T, int Q, int C, int fw) {
    const int c4n = C >> 2;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (unsigned)ncol * c4n) return;
    const unsigned col = i / c4n;
    const int c = (int)(i - col * c4n) * 4;
    const int t = (int)(col % (unsigned)T);
    const int q1 = idx[col];                                       // tap fw-1 reads the current token
    const int q0 = (fw == 2 && t > 0) ? idx[col - 1] : -1;         // tap 0 the previous one (nothing before the clip)
    float4 acc = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const long long cs = (long long)Q * fw;
    const float* w1 = W + (long long)c * cs + (long long)q1 * fw + (fw - 1);
    acc.x += w1[0]; acc.y += w1[cs]; acc.z += w1[2 * cs]; acc.w += w1[3 * cs];
    if (q0 >= 0) {
        const float* w0 = W + (long long)c * cs + (long long)q0 * fw;
        acc.x += w0[0]; acc.y += w0[cs]; acc.z += w0[2 * cs]; acc.w += w0[3 * cs];
    }
    *reinterpret_cast<float4*>(out + (long long)col * C + c) = acc;
}

// dW accumulated through an LDS table [Q*fw][C] per block, then flushed with global atomics.
// The block's columns are one contiguous range of dout; a thread takes kEmbU elements per round and issues all of
// their loads (values and tokens) before the first LDS atomic -- a loop of "load, then atomics on it" runs one
// memory latency per element (this kernel took 0.24 ms per step that way; the data is 17 MB).
static constexpr int kEmbU = 8;
// 16 waves per block: with one block per CU (the table fills LDS) 4 waves had 32 loads in flight per SIMD lane group and
// the kernel ran at 6 cycles per element and CU -- latency, not LDS atomics or HBM (17 MB), set its time
static constexpr int kEmbThreads = 1024;
template <int FW>
__global__ void k_embed_bwd_lds(const int32_t* __restrict__ idx, const float* __restrict__ dout,
                                float* __restrict__ ws, int has_bias,
                                int B, int T, int Q, int C, int fw_rt, int cols_per_block, int ldc) {
    // C is the width of this block's channel slice (blockIdx.y selects it), ldc the row stride of dout: a table for all
    // 128 channels of config 5 (256 KB) does not fit in LDS, four 32-channel slices do
    extern __shared__ __attribute__((aligned(16))) float tab[];   // [(q*fw+k)][c], then [C] bias
    const int fw = FW > 0 ? FW : fw_rt;
    const int ntab = Q * fw * C;
    float* btab = tab + ntab;
    for (int i = threadIdx.x; i < ntab + C; i += blockDim.x) tab[i] = 0.f;
    __syncthreads();
    const long long col0 = (long long)blockIdx.x * cols_per_block;
    const long long ncol = (long long)B * T;
    const int ncb = (int)(ncol - col0 < cols_per_block ? ncol - col0 : cols_per_block);
    const int work = ncb * C;
    const float* __restrict__ src = dout + col0 * ldc + (long long)blockIdx.y * C;
    const int b0 = (int)(col0 / T), t00 = (int)(col0 - (long long)b0 * T);
    for (int base = threadIdx.x; base < work; base += blockDim.x * kEmbU) {
        float g[kEmbU];
        int cc[kEmbU], qq[kEmbU][FW > 0 ? FW : 1];
#pragma unroll
        for (int u = 0; u < kEmbU; ++u) {
            const int e = base + u * blockDim.x;
            const int ec = e < work ? e : work - 1;                // clamped load, masked value
            const int colr = ec / C;
            cc[u] = ec - colr * C;
            g[u] = e < work ? src[(long long)colr * ldc + cc[u]] : 0.f;
            const int tt = t00 + colr;
            const int b = b0 + tt / T, t = tt - (tt / T) * T;
            if (FW > 0) {
#pragma unroll
                for (int k = 0; k < (FW > 0 ? FW : 1); ++k) {
                    const int ts = t - (FW - 1 - k);
                    qq[u][k] = ts >= 0 ? idx[(long long)b * T + ts] : -1;
                }
            } else {
                qq[u][0] = b * T + t;                              // runtime filter width: tokens are read below
            }
        }
#pragma unroll
        for (int u = 0; u < kEmbU; ++u) {
            if (base + u * blockDim.x >= work) break;
            if (has_bias) atomicAdd(&btab[cc[u]], g[u]);
            if (FW > 0) {
#pragma unroll
                for (int k = 0; k < (FW > 0 ? FW : 1); ++k)
                    if (qq[u][k] >= 0) atomicAdd(&tab[(qq[u][k] * FW + k) * C + cc[u]], g[u]);
            } else {
                const int bt = qq[u][0], t = bt % T;
                for (int k = 0; k < fw; ++k) {
                    const int ts = t - (fw - 1 - k);
                    if (ts >= 0) atomicAdd(&tab[(idx[bt - t + ts] * fw + k) * C + cc[u]], g[u]);
                }
            }
        }
    }
    __syncthreads();
    // the block's table leaves with plain coalesced stores; k_embed_bwd_reduce sums the tables (256 blocks adding
    // 16k entries each into the same 64 KB with global atomics took 0.2 ms)
    float* __restrict__ o = ws + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * (ntab + C);
    for (int i = threadIdx.x; i < ntab + C; i += blockDim.x) o[i] = tab[i];
}

__global__ void k_embed_bwd_reduce(const float* __restrict__ ws, int nblk, int Q, int C, int fw,
                                   float* __restrict__ dW, float* __restrict__ dbias) {
    const int ntab = Q * fw * C;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntab + C) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = 0;
    for (; b + 4 <= nblk; b += 4) {
        a0 += ws[(long long)(b + 0) * (ntab + C) + i]; a1 += ws[(long long)(b + 1) * (ntab + C) + i];
        a2 += ws[(long long)(b + 2) * (ntab + C) + i]; a3 += ws[(long long)(b + 3) * (ntab + C) + i];
    }
    for (; b < nblk; ++b) a0 += ws[(long long)b * (ntab + C) + i];
    const float v = (a0 + a1) + (a2 + a3);
    if (i < ntab) {
        const int c = i % C, qk = i / C;                   // qk = q*fw + k
        dW[(long long)c * Q * fw + qk] += v;               // sole writer of this element
    } else if (dbias) {
        dbias[i - ntab] += v;
    }
}

__global__ void k_embed_bwd_atomic(const int32_t* __restrict__ idx, const float* __restrict__ dout,
                                   float* __restrict__ dW, float* __restrict__ dbias,
                                   int B, int T, int Q, int C, int fw) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)B * T * C;
    if (i >= total) return;
    int c = (int)(i % C);
    long long bt = i / C;
    int t = (int)(bt % T);
    int b = (int)(bt / T);
    float g = dout[i];
    if (dbias) atomicAdd(&dbias[c], g);
    for (int k = 0; k < fw; ++k) {
        int ts = t - (fw - 1 - k);
        if (ts < 0) continue;
        int q = idx[(long long)b * T + ts];
        atomicAdd(&dW[((long long)c * Q + q) * fw + k], g);
    }
}

// ---------------------------------------------------------------------------------------------
// A5  dense dilated causal convolution
// ---------------------------------------------------------------------------------------------
__global__ void k_conv_fwd(const float* __restrict__ x, const float* __restrict__ W,
                           const float* __restrict__ bias, float* __restrict__ out,
                           int B, int T, int Cin, int Cout, int fw, int d, int Z) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)B * T * Cout;
    if (i >= total) return;
    int o = (int)(i % Cout);
    long long bt = i / Cout;
    int t = (int)(bt % T);
    int b = (int)(bt / T);
    if (t < Z) { out[i] = 0.f; return; }
    float acc = bias ? bias[o] : 0.f;
    const float* wrow = W + (long long)o * Cin * fw;
    for (int k = 0; k < fw; ++k) {
        int ts = t - (fw - 1 - k) * d;
        if (ts < 0) continue;
        const float* xr = x + ((long long)b * T + ts) * Cin;
        for (int c = 0; c < Cin; ++c) acc += wrow[c * fw + k] * xr[c];
    }
    out[i] = acc;
}

__global__ void k_conv_bwd_dx(const float* __restrict__ W, const float* __restrict__ dout,
                              float* __restrict__ dx, int B, int T, int Cin, int Cout, int fw, int d,
                              int Z) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)B * T * Cin;
    if (i >= total) return;
    int c = (int)(i % Cin);
    long long bt = i / Cin;
    int t = (int)(bt % T);
    int b = (int)(bt / T);
    float acc = 0.f;
    for (int k = 0; k < fw; ++k) {
        int to = t + (fw - 1 - k) * d;
        if (to >= T || to < Z) continue;
        const float* gr = dout + ((long long)b * T + to) * Cout;
        for (int o = 0; o < Cout; ++o) acc += W[((long long)o * Cin + c) * fw + k] * gr[o];
    }
    dx[i] = acc;
}

// ---------------------------------------------------------------------------------------------
// generic weight-gradient reduction:  dW[m*sm + k*sk] += sum_b sum_t A(b,t,m) * Bf(b,t,k)
//   A(b,t,m)  = A[b*a_bs + (a_t0+t)*lda + m]
//   Bf(b,t,k) = act(Bm[b*b_bs + (b_t0+t)*ldb + k]) (* B2[same]) ; 0 when b_t0+t < 0
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* A; long long a_bs; int a_t0; int lda;
    const float* Bm; const float* B2; long long b_bs; int b_t0; int ldb;
    int act;
    int nB, tmin, nT, M, K;
    float* dW; int sm, sk;
    int t_chunk;
};

__global__ void k_wgrad(WgradArgs a) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;    // element m*K + k
    int nchunk = (a.nT - a.tmin + a.t_chunk - 1) / a.t_chunk;
    int b = blockIdx.y / nchunk;
    int ch = blockIdx.y % nchunk;
    if (e >= a.M * a.K) return;
    int m = e / a.K, k = e % a.K;
    int t0 = a.tmin + ch * a.t_chunk;
    int t1 = min(a.nT, t0 + a.t_chunk);
    float acc = 0.f;
    for (int t = t0; t < t1; ++t) {
        int tb = a.b_t0 + t;
        if (tb < 0) continue;
        long long bi = (long long)b * a.b_bs + (long long)tb * a.ldb + k;
        float bv = a.Bm[bi];
        if (a.B2) bv *= a.B2[bi];
        bv = act_apply(bv, a.act);
        acc += a.A[(long long)b * a.a_bs + (long long)(a.a_t0 + t) * a.lda + m] * bv;
    }
    atomicAdd(&a.dW[(long long)m * a.sm + (long long)k * a.sk], acc);
}

static int launch_wgrad(WgradArgs a, hipStream_t s) {
    if (a.nT <= a.tmin || a.nB <= 0) return WN_OK;
    a.t_chunk = 512;
    int nchunk = (a.nT - a.tmin + a.t_chunk - 1) / a.t_chunk;
    dim3 grid(cdiv((long long)a.M * a.K, kThreads), a.nB * nchunk);
    hipLaunchKernelGGL(k_wgrad, grid, dim3(kThreads), 0, s, a);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

// column sums:  out[m] += sum_b sum_{t in [tmin,nT)} A(b,t,m)
// part != NULL: block (x, y) leaves its sums in part[y][m] and k_colsum_reduce adds the blocks in index order (no float
// atomics: bit-reproducible bias gradients); part == NULL (no WnExec scratch): one atomic per block and column.
__global__ void k_colsum(const float* __restrict__ A, long long a_bs, int a_t0, int lda, int nB,
                         int tmin, int nT, int M, float* __restrict__ out, int t_chunk, float* __restrict__ part_out) {
    // blockDim = (64 columns, 4 row lanes)
    __shared__ float part[4][64];
    int m = blockIdx.x * 64 + threadIdx.x;
    int nchunk = (nT - tmin + t_chunk - 1) / t_chunk;
    int b = blockIdx.y / nchunk, ch = blockIdx.y % nchunk;
    int t0 = tmin + ch * t_chunk, t1 = min(nT, t0 + t_chunk);
    float acc = 0.f;
    if (m < M)
        for (int t = t0 + threadIdx.y; t < t1; t += 4) acc += A[(long long)b * a_bs + (long long)(a_t0 + t) * lda + m];
    part[threadIdx.y][threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.y == 0 && m < M) {
        const float v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (part_out) part_out[(long long)blockIdx.y * M + m] = v;
        else atomicAdd(&out[m], v);
    }
}
// blockDim = (64 columns, 16 row lanes): lane y adds rows y, y + 16, ... in order, then the 16 lane sums are added in
// index order -- a fixed summation tree, whatever the launch looks like
__global__ void k_colsum_reduce(const float* __restrict__ part, int ny, int M, float* __restrict__ out) {
    __shared__ float red[16][64];
    const int m = blockIdx.x * 64 + threadIdx.x;
    float s = 0.f;
    if (m < M)
        for (int y = threadIdx.y; y < ny; y += 16) s += part[(long long)y * M + m];
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && m < M) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        out[m] += t;
    }
}

// (the wide weight-gradient kernel's column sums are reduced by the same tree inside k_wgrad_b3w_reduce: mfma_gemm_b3.hip)

static int launch_colsum(const float* A, long long a_bs, int a_t0, int lda, int nB, int tmin, int nT,
                         int M, float* out, hipStream_t s) {
    if (nT <= tmin || nB <= 0) return WN_OK;
    int t_chunk = 256;
    int nchunk = (nT - tmin + t_chunk - 1) / t_chunk;
    while ((long long)nB * nchunk > 2048) { t_chunk *= 2; nchunk = (nT - tmin + t_chunk - 1) / t_chunk; }
    dim3 grid(cdiv(M, 64), nB * nchunk);
    float* part = exec_has_scratch((size_t)grid.y * M * sizeof(float))
                      ? reinterpret_cast<float*>(exec_scratch((size_t)grid.y * M * sizeof(float), "column-sum partials"))
                      : nullptr;
    hipLaunchKernelGGL(k_colsum, grid, dim3(64, 4), 0, s, A, a_bs, a_t0, lda, nB, tmin, nT, M, out, t_chunk, part);
    if (part) hipLaunchKernelGGL(k_colsum_reduce, dim3(cdiv(M, 64)), dim3(64, 16), 0, s, part, (int)grid.y, M, out);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

// ---------------------------------------------------------------------------------------------
// A7  residual layer, generic
// ---------------------------------------------------------------------------------------------
__global__ void k_gate_fwd(const float* __restrict__ x, const float* __restrict__ Wf,
                           const float* __restrict__ bf, const float* __restrict__ Wg,
                           const float* __restrict__ bg, float* __restrict__ z,
                           float* __restrict__ fs, float* __restrict__ gs,
                           int B, int T, int Cr, int Cd, int fw, int d, int Z) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)B * T * Cd;
    if (i >= total) return;
    int o = (int)(i % Cd);
    long long bt = i / Cd;
    int t = (int)(bt % T);
    int b = (int)(bt / T);
    float a = 0.f, g = 0.f;
    if (t >= Z) {
        a = bf ? bf[o] : 0.f;
        g = bg ? bg[o] : 0.f;
        const float* wf = Wf + (long long)o * Cr * fw;
        const float* wg = Wg + (long long)o * Cr * fw;
        for (int k = 0; k < fw; ++k) {
            int ts = t - (fw - 1 - k) * d;
            if (ts < 0) continue;
            const float* xr = x + ((long long)b * T + ts) * Cr;
            for (int c = 0; c < Cr; ++c) {
                float xv = xr[c];
                a += wf[c * fw + k] * xv;
                g += wg[c * fw + k] * xv;
            }
        }
    }
    float f = fast_tanh(a), s = fast_sigmoid(g);
    z[i] = f * s;
    if (fs) fs[i] = f;
    if (gs) gs[i] = s;
}

__global__ void k_proj_res_fwd(const float* __restrict__ x, const float* __restrict__ z,
                               const float* __restrict__ Wp, const float* __restrict__ bp,
                               float* __restrict__ out, long long N, int Cr, int Cd) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Cr) return;
    int o = (int)(i % Cr);
    long long n = i / Cr;
    float acc = bp ? bp[o] : 0.f;
    const float* zr = z + n * Cd;
    const float* w = Wp + (long long)o * Cd;
    for (int c = 0; c < Cd; ++c) acc += w[c] * zr[c];
    out[i] = acc + x[i];
}

// dab[b,t,0:Cd] = da, dab[b,t,Cd:2Cd] = dg
__global__ void k_gate_bwd(const float* __restrict__ f, const float* __restrict__ g,
                           const float* __restrict__ Wp, const float* __restrict__ dout,
                           const float* __restrict__ dzs, float* __restrict__ dab,
                           int B, int T, int Cr, int Cd, int Z) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)B * T * Cd;
    if (i >= total) return;
    int c = (int)(i % Cd);
    long long bt = i / Cd;
    int t = (int)(bt % T);
    float da = 0.f, dg = 0.f;
    if (t >= Z) {
        float dz = dzs ? dzs[i] : 0.f;
        if (dout) {
            const float* gr = dout + bt * Cr;
            for (int o = 0; o < Cr; ++o) dz += Wp[(long long)o * Cd + c] * gr[o];
        }
        float fv = f[i], gv = g[i];
        da = dz * gv * (1.f - fv * fv);
        dg = dz * fv * gv * (1.f - gv);
    }
    dab[bt * 2 * Cd + c] = da;
    dab[bt * 2 * Cd + Cd + c] = dg;
}

__global__ void k_layer_bwd_dx(const float* __restrict__ Wf, const float* __restrict__ Wg,
                               const float* __restrict__ dout, const float* __restrict__ dab,
                               float* __restrict__ dx, int B, int T, int Cr, int Cd, int fw, int d) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)B * T * Cr;
    if (i >= total) return;
    int c = (int)(i % Cr);
    long long bt = i / Cr;
    int t = (int)(bt % T);
    int b = (int)(bt / T);
    float acc = dout ? dout[i] : 0.f;
    for (int k = 0; k < fw; ++k) {
        int to = t + (fw - 1 - k) * d;
        if (to >= T) continue;
        const float* r = dab + ((long long)b * T + to) * 2 * Cd;   // zero for to < Z already
        for (int o = 0; o < Cd; ++o) {
            acc += Wf[((long long)o * Cr + c) * fw + k] * r[o];
            acc += Wg[((long long)o * Cr + c) * fw + k] * r[Cd + o];
        }
    }
    dx[i] = acc;
}

// ---------------------------------------------------------------------------------------------
// 1x1 convolution with pre-activation
// ---------------------------------------------------------------------------------------------
__global__ void k_pointwise_fwd(const float* __restrict__ x, const float* __restrict__ W,
                                const float* __restrict__ bias, float* __restrict__ out,
                                long long N, int Cin, int Cout, int act) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Cout) return;
    int o = (int)(i % Cout);
    long long n = i / Cout;
    float acc = bias ? bias[o] : 0.f;
    const float* xr = x + n * Cin;
    const float* w = W + (long long)o * Cin;
    for (int c = 0; c < Cin; ++c) acc += w[c] * act_apply(xr[c], act);
    out[i] = acc;
}

__global__ void k_pointwise_bwd_dx(const float* __restrict__ x, const float* __restrict__ W,
                                   const float* __restrict__ dout, float* __restrict__ dx,
                                   long long N, int Cin, int Cout, int act) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Cin) return;
    int c = (int)(i % Cin);
    long long n = i / Cin;
    const float* gr = dout + n * Cout;
    float acc = 0.f;
    for (int o = 0; o < Cout; ++o) acc += W[(long long)o * Cin + c] * gr[o];
    dx[i] = acc * act_grad(x[i], act);
}

// ---------------------------------------------------------------------------------------------
// A11 deferred skip sum
// ---------------------------------------------------------------------------------------------
struct SkipArgs {
    const float* z[WN_MAX_SRC];
    const float* Ws[WN_MAX_SRC];
    const float* bs[WN_MAX_SRC];
    int cd[WN_MAX_SRC];
    int L;
};

__global__ void k_skip_sum_fwd(SkipArgs a, float* __restrict__ skip, int B, int T, int t_off, int Tw,
                               int Cs, int accumulate) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)B * Tw * Cs;
    if (i >= total) return;
    int o = (int)(i % Cs);
    long long bt = i / Cs;
    int t = (int)(bt % Tw);
    int b = (int)(bt / Tw);
    float acc = accumulate ? skip[i] : 0.f;
    for (int l = 0; l < a.L; ++l) {
        int cd = a.cd[l];
        const float* zr = a.z[l] + ((long long)b * T + t_off + t) * cd;
        const float* w = a.Ws[l] + (long long)o * cd;
        float s = a.bs[l] ? a.bs[l][o] : 0.f;
        for (int c = 0; c < cd; ++c) s += w[c] * zr[c];
        acc += s;
    }
    skip[i] = acc;
}

__global__ void k_skip_bwd_dz(const float* __restrict__ Ws, const float* __restrict__ dskip,
                              float* __restrict__ dz, int B, int T, int t_off, int Tw, int Cs, int cd) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)B * T * cd;
    if (i >= total) return;
    int c = (int)(i % cd);
    long long bt = i / cd;
    int t = (int)(bt % T);
    int b = (int)(bt / T);
    float acc = 0.f;
    if (t >= t_off && t < t_off + Tw) {
        const float* gr = dskip + ((long long)b * Tw + (t - t_off)) * Cs;
        for (int o = 0; o < Cs; ++o) acc += Ws[(long long)o * cd + c] * gr[o];
    }
    dz[i] = acc;
}

// ---------------------------------------------------------------------------------------------
// softmax / cross entropy: one wave per row
// ---------------------------------------------------------------------------------------------
// full-wave reductions on the DPP path (row shifts + row broadcasts + one readlane) instead of six
// ds_bpermute round trips; the result is uniform (broadcast through an SGPR)
__device__ __forceinline__ float dpp_mov(float old, float v, int which) {
    int r;
    switch (which) {
        case 1: r = __builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x111, 0xf, 0xf, false); break;
        case 2: r = __builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x112, 0xf, 0xf, false); break;
        case 4: r = __builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x114, 0xf, 0xf, false); break;
        case 8: r = __builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x118, 0xf, 0xf, false); break;
        case 15: r = __builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x142, 0xa, 0xf, false); break;  // row_bcast:15
        default: r = __builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x143, 0xc, 0xf, false); break;  // row_bcast:31
    }
    return __int_as_float(r);
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    const float ninf = -INFINITY;
    v = fmaxf(v, dpp_mov(ninf, v, 1));
    v = fmaxf(v, dpp_mov(ninf, v, 2));
    v = fmaxf(v, dpp_mov(ninf, v, 4));
    v = fmaxf(v, dpp_mov(ninf, v, 8));
    v = fmaxf(v, dpp_mov(ninf, v, 15));
    v = fmaxf(v, dpp_mov(ninf, v, 31));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_mov(0.f, v, 1);
    v += dpp_mov(0.f, v, 2);
    v += dpp_mov(0.f, v, 4);
    v += dpp_mov(0.f, v, 8);
    v += dpp_mov(0.f, v, 15);
    v += dpp_mov(0.f, v, 31);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void k_softmax(const float* __restrict__ logits, float* __restrict__ prob, long long N, int Q) {
    long long row = (long long)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    int lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* r = logits + row * Q;
    float m = -INFINITY;
    for (int q = lane; q < Q; q += 64) m = fmaxf(m, r[q]);
    m = wave_max(m);
    float s = 0.f;
    for (int q = lane; q < Q; q += 64) s += expf(r[q] - m);
    s = wave_sum(s);
    float inv = 1.f / s;
    for (int q = lane; q < Q; q += 64) prob[row * Q + q] = expf(r[q] - m) * inv;
}

// loss[0] = (sum of the per-block sums, in block order) / n_norm: no float atomics, the loss is bit-reproducible
// (kXentPart, kXentCnt, kXentBlocks: wn_kernels.hpp)
// n_norm < 0: the number of rows that count (labels in [0, Q)) is taken on the device: k_xent_count leaves one INTEGER per
// workgroup in the last kXentCnt words of the loss buffer and every reader adds them (one load per lane and a wave
// reduction; integer sums do not depend on the order).  No memset, no atomics: with one zeroed word + an integer atomic
// per wave, the FIRST replay of a captured bf16x3 training step read a garbage count (the loss buffer is allocated from
// the graph's pool during capture; eager calls and later replays were right).  The single block of 1,024 threads this
// replaces took 45 us at config 2 (98,320 labels) -- 1.4 % of the training step for a count.
__global__ void k_xent_count(const int32_t* __restrict__ target, long long N, int Q, float* __restrict__ loss) {
    __shared__ int red[16];
    int c = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const int t = target[i];
        c += (t >= 0 && t < Q) ? 1 : 0;
    }
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int sum = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += red[w];
        reinterpret_cast<int*>(loss + kXentPart + kXentBlocks)[blockIdx.x] = sum;
    }
}
// every lane of the calling wave gets the count (at least 1); ncnt = workgroups of k_xent_count (<= kXentCnt = 64)
__device__ __forceinline__ float xent_count(const float* loss, int ncnt) {
    const int lane = threadIdx.x & 63;
    int c = lane < ncnt ? reinterpret_cast<const int*>(loss + kXentPart + kXentBlocks)[lane] : 0;
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
    return (float)(c > 0 ? c : 1);
}
__global__ void k_xent_final(float* __restrict__ loss, int nb, long long n_norm, int ncnt) {      // 256 threads, fixed tree
    __shared__ float red[4];
    const float nrm = n_norm < 0 ? xent_count(loss, ncnt) : (float)n_norm;
    float acc = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) acc += loss[kXentPart + i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) / nrm;
}

__global__ void k_softmax_xent(const float* __restrict__ logits, const int32_t* __restrict__ target,
                               float* __restrict__ loss, float* __restrict__ dlogits, long long N, int Q, long long n_norm,
                               int ncnt) {
    // one wave per row; rows of up to 256 logits live in registers (one float4 per lane).  Waves stride
    // over rows so that the loss leaves with ONE atomic per block: thousands of adds to one address
    // would serialise at ~13 ns each.
    int lane = threadIdx.x & 63;
    __shared__ float part[16];
    float rl_acc = 0.f;
    const float invN = 1.f / (n_norm < 0 ? xent_count(loss, ncnt) : (float)n_norm);
    for (long long row = (long long)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64; row < N;
         row += (long long)gridDim.x * (blockDim.x / 64)) {
        float rl = 0.f;
        const float* r = logits + row * Q;
        const int tg = target[row];
        if (tg < 0 || tg >= Q) {
            // Chainer's softmax_cross_entropy ignores label -1 (no loss, no gradient, not counted in the mean: n_norm); any
            // other label outside [0, Q) is treated the same way here instead of reading out of bounds
            if (dlogits)
                for (int q = lane; q < Q; q += 64) dlogits[row * Q + q] = 0.f;
            continue;
        }
        if (Q <= 256 && (Q & 3) == 0) {
            const bool on = 4 * lane < Q;
            float4 v = on ? *reinterpret_cast<const float4*>(r + 4 * lane) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            float m = wave_max_dpp(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            float4 e = make_float4(__expf(v.x - m), __expf(v.y - m), __expf(v.z - m), __expf(v.w - m));
            float s = wave_sum_dpp(e.x + e.y + e.z + e.w);
            const float tv = r[tg];                       // uniform address: one scalar-ish load
            rl = m + logf(s) - tv;
            if (dlogits && on) {
                float inv = invN / s;
                float4 d = make_float4(e.x * inv, e.y * inv, e.z * inv, e.w * inv);
                if ((tg >> 2) == lane) {
                    if ((tg & 3) == 0) d.x -= invN; else if ((tg & 3) == 1) d.y -= invN;
                    else if ((tg & 3) == 2) d.z -= invN; else d.w -= invN;
                }
                *reinterpret_cast<float4*>(dlogits + row * Q + 4 * lane) = d;
            }
        } else {
            float m = -INFINITY;
            for (int q = lane; q < Q; q += 64) m = fmaxf(m, r[q]);
            m = wave_max(m);
            float s = 0.f;
            for (int q = lane; q < Q; q += 64) s += __expf(r[q] - m);
            s = wave_sum(s);
            rl = m + logf(s) - r[tg];
            if (dlogits) {
                float inv = 1.f / s;
                for (int q = lane; q < Q; q += 64)
                    dlogits[row * Q + q] = (__expf(r[q] - m) * inv - (q == tg ? 1.f : 0.f)) * invN;
            }
        }
        rl_acc += rl;
    }
    if (lane == 0) part[threadIdx.x / 64] = rl_acc;
    __syncthreads();
    if (threadIdx.x == 0) {                            // this block's sum; k_xent_final adds the blocks in index order
        float s = 0.f;
        for (int w = 0; w < (int)(blockDim.x / 64); ++w) s += part[w];
        loss[kXentPart + blockIdx.x] = s;
    }
}

// x *= *s, skipped entirely when *s == 1 (every block reads the scalar and leaves): the upstream gradient of a loss is 1
// unless the caller scaled the loss, and multiplying 8.4 M dlogits by it cost a 33 us pass per step.
__global__ void k_scale_by_dev(float* __restrict__ x, const float* __restrict__ s, long long n4, long long n) {
    const float v = *s;
    if (v == 1.f) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 t = reinterpret_cast<float4*>(x)[i];
        t.x *= v; t.y *= v; t.z *= v; t.w *= v;
        reinterpret_cast<float4*>(x)[i] = t;
    }
    if (blockIdx.x == 0)
        for (long long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) x[i] *= v;
}

// ---------------------------------------------------------------------------------------------
// layout conversion (B,C,T) <-> (B,T,C) through a padded 32x32 LDS tile
// ---------------------------------------------------------------------------------------------
__global__ void k_transpose(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc) {
    // src is [batch][R][Cc] -> dst [batch][Cc][R]
    __shared__ float tile[32][33];
    long long base = (long long)blockIdx.z * R * Cc;
    int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int r = r0 + j, c = c0 + threadIdx.x;
        if (r < R && c < Cc) tile[j][threadIdx.x] = src[base + (long long)r * Cc + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int c = c0 + j, r = r0 + threadIdx.x;
        if (r < R && c < Cc) dst[base + (long long)c * R + r] = tile[threadIdx.x][j];
    }
}

// ---------------------------------------------------------------------------------------------
// numpy-compatible categorical draw (generate.py:39)
// ---------------------------------------------------------------------------------------------
__global__ void k_sample(const float* __restrict__ prob, const double* __restrict__ u,
                         int32_t* __restrict__ out, int n, int Q) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = prob + (long long)i * Q;
    double tot = 0.0;
    for (int q = 0; q < Q; ++q) tot += (double)p[q];
    double c = 0.0, uu = u[i];
    int idx = Q;
    for (int q = 0; q < Q; ++q) {
        c += (double)p[q];
        if (c / tot > uu) { idx = q; break; }
    }
    out[i] = idx;
}

// ---------------------------------------------------------------------------------------------
// optimiser step
// ---------------------------------------------------------------------------------------------
// Two launches, no atomics: block b leaves its sum in out[kNormPart + b]; one block then adds the partial sums in index
// order and ASSIGNS out[0].  (A float atomicAdd per block made the norm -- hence the clipping rate, hence every weight --
// depend on the order in which blocks retired: the last bit of a training step differed from run to run.)
static constexpr int kNormPart = 8, kNormBlocks = 1024;
__global__ void k_sqnorm(const float* __restrict__ g, const float* __restrict__ p, long long n,
                         float gmult, float wd, float* __restrict__ out) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i] * gmult;
        if (p && wd != 0.f) gi += wd * p[i];
        acc += gi * gi;
    }
    acc = wave_sum(acc);
    __shared__ float part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x / 64] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < (int)(blockDim.x / 64); ++w) s += part[w];
        out[kNormPart + blockIdx.x] = s;
    }
}
__global__ void k_sqnorm_final(float* __restrict__ out, int nb) {      // 256 threads, fixed tree
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) acc += out[kNormPart + i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, long long n, float lr_t, float b1, float b2, float eps,
                       float wd, const float* __restrict__ sqnorm, float clip, float gmult,
                       const float* __restrict__ lr_dev, float dscale) {
    // hook order of wavenet.py:477-480: WeightDecay first, then GradientClipping on the result
    if (lr_dev) lr_t = *lr_dev;          // captured-graph replays: the step size comes from device memory
    float rate = 1.f;
    if (sqnorm && clip > 0.f) {
        float nrm = sqrtf(*sqnorm);
        // A gradient whose global norm is not finite is no gradient: the whole update is skipped -- weights, m and v stay as
        // they are -- instead of writing NaN into the optimiser state for good.  (Chainer would apply it; no caller can want
        // that.  It is what makes a void backward recoverable: the multi-layer launches flag a wait that gave up with a NaN
        // in a weight gradient, the all-reduce carries it to every rank, and every rank skips the same step.)
        if (!(nrm < 3.0e38f)) return;
        if (nrm > 0.f && clip / nrm < 1.f) rate = clip / nrm;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i] * gmult;
        float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        gi *= rate;
        float mi = m[i], vi = v[i];
        mi += (1.f - b1) * (gi - mi);
        vi += (1.f - b2) * (gi * gi - vi);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - lr_t * mi / (dscale * sqrtf(vi) + eps);      // dscale = 1: Adam; Eve's d otherwise (wavenet.py:57-65)
    }
}

// The other update rules get_optimizer() names (wavenet.py:81-97), as Chainer publishes them, behind the same two hooks.
//   0 SGD          p -= lr g
//   1 MomentumSGD  v = mu v - lr g;  p += v
//   2 AdaGrad      h += g^2;  p -= lr g / (sqrt(h) + eps)
//   3 AdaDelta     msg += (1-rho)(g^2 - msg);  dx = sqrt((msdx+eps)/(msg+eps)) g;  msdx += (1-rho)(dx^2 - msdx);  p -= dx
//   4 NesterovAG   v = mu v - lr g;  p += mu^2 v - (1+mu) lr g
//   5 RMSprop      ms += (1-alpha)(g^2 - ms);  p -= lr g / (sqrt(ms) + eps)
template <int RULE>
__global__ void k_rule(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ s1,
                       float* __restrict__ s2, long long n, float lr, float hy, float eps, float wd,
                       const float* __restrict__ sqnorm, float clip, float gmult, const float* __restrict__ lr_dev) {
    if (lr_dev) lr = *lr_dev;
    float rate = 1.f;
    if (sqnorm && clip > 0.f) {
        float nrm = sqrtf(*sqnorm);
        if (!(nrm < 3.0e38f)) return;      // non-finite gradient norm: the update is skipped (see k_adam)
        if (nrm > 0.f && clip / nrm < 1.f) rate = clip / nrm;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i] * gmult;
        float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        gi *= rate;
        if (RULE == 0) {
            pi -= lr * gi;
        } else if (RULE == 1) {
            const float v = hy * s1[i] - lr * gi;
            s1[i] = v;
            pi += v;
        } else if (RULE == 2) {
            const float h = s1[i] + gi * gi;
            s1[i] = h;
            pi -= lr * gi / (sqrtf(h) + eps);
        } else if (RULE == 3) {
            float msg = s1[i], msdx = s2[i];
            msg += (1.f - hy) * (gi * gi - msg);
            const float dx = sqrtf((msdx + eps) / (msg + eps)) * gi;
            msdx += (1.f - hy) * (dx * dx - msdx);
            s1[i] = msg; s2[i] = msdx;
            pi -= dx;
        } else if (RULE == 4) {
            const float v = hy * s1[i] - lr * gi;
            s1[i] = v;
            pi += hy * hy * v - (1.f + hy) * lr * gi;
        } else {
            float ms = s1[i];
            ms += (1.f - hy) * (gi * gi - ms);
            s1[i] = ms;
            pi -= lr * gi / (sqrtf(ms) + eps);
        }
        p[i] = pi;
    }
}

// ---------------------------------------------------------------------------------------------
// A2 on the device: table lookups (the 65,536-entry encode table and the Q-entry decode table are built on the host
// with the reference's float64 formulas, data.py:18-23 / 37-43, so the device results are theirs bit for bit)
// ---------------------------------------------------------------------------------------------
__global__ void k_mulaw_encode_pcm16(const int16_t* __restrict__ pcm, const int32_t* __restrict__ lut,
                                     int32_t* __restrict__ tok, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        tok[i] = lut[(int)pcm[i] + 32768];
}
__global__ void k_mulaw_decode(const int32_t* __restrict__ tok, const float* __restrict__ table, float* __restrict__ out,
                               long long n, int Q) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int q = tok[i];
        q = q < 0 ? 0 : (q >= Q ? Q - 1 : q);
        out[i] = table[q];
    }
}

}  // namespace wn

using namespace wn;

// =============================================================================================
// generic host entry points (called by the dispatching C ABI in api.hip)
// =============================================================================================
namespace wn {

int generic_embed_fwd(const int32_t* idx, const float* W, const float* bias, float* out, int B, int T,
                      int Q, int C, int fw, hipStream_t s) {
    long long total = (long long)B * T * C;
    if (C % 4 == 0 && fw <= 2 && total / 4 < (1ll << 31) && (long long)B * T < (1ll << 31)) {
        const long long n = total / 4;
        hipLaunchKernelGGL(k_embed_fwd4, dim3(cdiv(n, kThreads)), dim3(kThreads), 0, s, idx, W, bias, out, B * T, T, Q, C, fw);
    } else {
        hipLaunchKernelGGL(k_embed_fwd, dim3(cdiv(total, kThreads)), dim3(kThreads), 0, s, idx, W, bias, out,
                           B, T, Q, C, fw);
    }
    WN_LAUNCH_CHECK();
    return WN_OK;
}

}  // namespace wn
namespace w16 {   // w16_gemm.hip: the table gradient as a one-hot contraction on the matrix cores
size_t embed_bwd_ws_bytes(int B, int T, int C);
int embed_bwd_mfma(const int32_t* idx, const __bf16* dx, const float* dx_f32, int C, float* dW, float* dbias, int B, int T,
                   void* ws, hipStream_t s);
}
namespace wn {

int generic_embed_bwd(const int32_t* idx, const float* dout, float* dW, float* dbias, int B, int T,
                      int Q, int C, int fw, hipStream_t s) {
    // filter width 2, 256 token values, 32 / 64 / 128 channels (every BASELINE config): matrix cores, fixed summation
    // order (bit-reproducible); anything else, or no scratch: per-block tables in LDS
    if (fw == 2 && Q == 256 && (C == 32 || C == 64 || C == 128) && exec_has_scratch(w16::embed_bwd_ws_bytes(B, T, C))) {
        void* ws = exec_scratch(w16::embed_bwd_ws_bytes(B, T, C), "the embedding-gradient partial tables");
        return w16::embed_bwd_mfma(idx, nullptr, dout, C, dW, dbias, B, T, ws, s);
    }
    int Cs = C;                                            // channel slice whose table fits in LDS
    while (((size_t)Q * fw * Cs + Cs) * sizeof(float) > 150 * 1024 && Cs % 2 == 0 && Cs > 8) Cs /= 2;
    size_t lds = ((size_t)Q * fw * Cs + Cs) * sizeof(float);
    long long ncol = (long long)B * T;
    if (lds <= 150 * 1024 && C % Cs == 0) {
        const int nsl = C / Cs;
        long long nb = ncol < 256 * 64 ? (ncol + 63) / 64 : 256;
        if (nsl > 1 && nb > 64) nb = nb / nsl < 64 ? 64 : nb / nsl;       // about one block per CU over all slices
        while ((ncol + nb - 1) / nb * Cs >= (1ll << 30)) nb *= 2;         // 32-bit element indices inside a block
        int cpb = (int)((ncol + nb - 1) / nb);
        int nblk = (int)((ncol + cpb - 1) / cpb);
        const int nent = Q * fw * Cs + Cs;
        float* ws = reinterpret_cast<float*>(exec_scratch((size_t)nsl * nblk * nent * sizeof(float),
                                                          "the per-block embedding-gradient tables"));
        if (!ws) return WN_EARG;
#define EMB_LAUNCH(FW)                                                                                        \
    do {                                                                                                      \
        static bool attr_set = false;                                                                         \
        if (!attr_set) {                                                                                      \
            WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_embed_bwd_lds<FW>),                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));              \
            attr_set = true;                                                                                  \
        }                                                                                                     \
        hipLaunchKernelGGL(k_embed_bwd_lds<FW>, dim3(nblk, nsl), dim3(kEmbThreads), lds, s, idx, dout, ws, dbias ? 1 : 0, B, T, \
                           Q, Cs, fw, cpb, C);                                                                \
    } while (0)
        if (fw == 1) EMB_LAUNCH(1);
        else if (fw == 2) EMB_LAUNCH(2);
        else if (fw == 3) EMB_LAUNCH(3);
        else EMB_LAUNCH(0);
#undef EMB_LAUNCH
        WN_LAUNCH_CHECK();
        for (int sl = 0; sl < nsl; ++sl)
            hipLaunchKernelGGL(k_embed_bwd_reduce, dim3(cdiv(nent, kThreads)), dim3(kThreads), 0, s,
                               ws + (size_t)sl * nblk * nent, nblk, Q, Cs, fw, dW + (size_t)sl * Cs * Q * fw,
                               dbias ? dbias + sl * Cs : nullptr);
    } else {
        long long total = ncol * C;
        hipLaunchKernelGGL(k_embed_bwd_atomic, dim3(cdiv(total, kThreads)), dim3(kThreads), 0, s, idx, dout,
                           dW, dbias, B, T, Q, C, fw);
    }
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int generic_conv_fwd(const float* x, const float* W, const float* bias, float* out, int B, int T, int Cin,
                     int Cout, int fw, int d, int Z, hipStream_t s) {
    long long total = (long long)B * T * Cout;
    hipLaunchKernelGGL(k_conv_fwd, dim3(cdiv(total, kThreads)), dim3(kThreads), 0, s, x, W, bias, out, B, T,
                       Cin, Cout, fw, d, Z);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

// dW[o][c][k] += sum_{t>=Z} dout[b,t,o] x[b,t-(fw-1-k)d,c]
static int conv_dw(const float* x, const float* dout, int ld_dout, float* dW, int B, int T, int Cin, int Cout,
                   int fw, int d, int Z, hipStream_t s) {
    for (int k = 0; k < fw; ++k) {
        WgradArgs a{};
        a.A = dout; a.a_bs = (long long)T * ld_dout; a.a_t0 = 0; a.lda = ld_dout;
        a.Bm = x; a.B2 = nullptr; a.b_bs = (long long)T * Cin; a.b_t0 = -(fw - 1 - k) * d; a.ldb = Cin;
        a.act = WN_ACT_NONE;
        a.nB = B; a.tmin = Z; a.nT = T; a.M = Cout; a.K = Cin;
        a.dW = dW + k; a.sm = Cin * fw; a.sk = fw;
        int rc = launch_wgrad(a, s);
        if (rc) return rc;
    }
    return WN_OK;
}

int generic_conv_bwd(const float* x, const float* W, const float* dout, float* dx, float* dW, float* dbias,
                     int B, int T, int Cin, int Cout, int fw, int d, int Z, hipStream_t s) {
    if (dx) {
        long long total = (long long)B * T * Cin;
        hipLaunchKernelGGL(k_conv_bwd_dx, dim3(cdiv(total, kThreads)), dim3(kThreads), 0, s, W, dout, dx, B, T,
                           Cin, Cout, fw, d, Z);
        WN_LAUNCH_CHECK();
    }
    if (dW) {
        int rc = conv_dw(x, dout, Cout, dW, B, T, Cin, Cout, fw, d, Z, s);
        if (rc) return rc;
    }
    if (dbias) return launch_colsum(dout, (long long)T * Cout, 0, Cout, B, Z, T, Cout, dbias, s);
    return WN_OK;
}

int generic_layer_fwd(const float* x, const float* Wf, const float* bf, const float* Wg, const float* bg,
                      const float* Wp, const float* bp, float* out, float* z, float* fs, float* gs, int B,
                      int T, int Cr, int Cd, int fw, int d, int Z, hipStream_t s) {
    long long tot = (long long)B * T * Cd;
    hipLaunchKernelGGL(k_gate_fwd, dim3(cdiv(tot, kThreads)), dim3(kThreads), 0, s, x, Wf, bf, Wg, bg, z, fs,
                       gs, B, T, Cr, Cd, fw, d, Z);
    WN_LAUNCH_CHECK();
    long long N = (long long)B * T;
    hipLaunchKernelGGL(k_proj_res_fwd, dim3(cdiv(N * Cr, kThreads)), dim3(kThreads), 0, s, x, z, Wp, bp, out,
                       N, Cr, Cd);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int generic_layer_bwd(const float* x, const float* f, const float* g, const float* Wf, const float* Wg,
                      const float* Wp, const float* dout, const float* dzs, float* dx, float* dWf, float* dbf,
                      float* dWg, float* dbg, float* dWp, float* dbp, float* dab, int B, int T, int Cr, int Cd,
                      int fw, int d, int Z, hipStream_t s) {
    long long tot = (long long)B * T * Cd;
    hipLaunchKernelGGL(k_gate_bwd, dim3(cdiv(tot, kThreads)), dim3(kThreads), 0, s, f, g, Wp, dout, dzs, dab, B,
                       T, Cr, Cd, Z);
    WN_LAUNCH_CHECK();
    if (dx) {
        long long n = (long long)B * T * Cr;
        hipLaunchKernelGGL(k_layer_bwd_dx, dim3(cdiv(n, kThreads)), dim3(kThreads), 0, s, Wf, Wg, dout, dab, dx,
                           B, T, Cr, Cd, fw, d);
        WN_LAUNCH_CHECK();
    }
    int rc;
    if (dWf && (rc = conv_dw(x, dab, 2 * Cd, dWf, B, T, Cr, Cd, fw, d, Z, s))) return rc;
    if (dWg && (rc = conv_dw(x, dab + Cd, 2 * Cd, dWg, B, T, Cr, Cd, fw, d, Z, s))) return rc;
    if (dbf && (rc = launch_colsum(dab, (long long)T * 2 * Cd, 0, 2 * Cd, B, Z, T, Cd, dbf, s))) return rc;
    if (dbg && (rc = launch_colsum(dab + Cd, (long long)T * 2 * Cd, 0, 2 * Cd, B, Z, T, Cd, dbg, s))) return rc;
    if (dWp) {   // dWp[cr][cd] += dout z^T,  z = f*g
        WgradArgs a{};
        a.A = dout; a.a_bs = (long long)T * Cr; a.a_t0 = 0; a.lda = Cr;
        a.Bm = f; a.B2 = g; a.b_bs = (long long)T * Cd; a.b_t0 = 0; a.ldb = Cd;
        a.act = WN_ACT_NONE; a.nB = B; a.tmin = 0; a.nT = T; a.M = Cr; a.K = Cd;
        a.dW = dWp; a.sm = Cd; a.sk = 1;
        if ((rc = launch_wgrad(a, s))) return rc;
    }
    if (dbp && (rc = launch_colsum(dout, (long long)T * Cr, 0, Cr, B, 0, T, Cr, dbp, s))) return rc;
    return WN_OK;
}

// out[m] += sum over rows t in [tmin, nT) of every clip of A[b][t][m]  (dense rows of width lda)
int generic_colsum(const float* A, int nB, int nT, int tmin, int lda, int M, float* out, hipStream_t s) {
    return launch_colsum(A, (long long)nT * lda, 0, lda, nB, tmin, nT, M, out, s);
}

// bias gradients of a residual layer from the (da,dg) scratch: dbf += sum da, dbg += sum dg, dbp += sum dout
int generic_layer_bwd_biases(const float* dab, const float* dout, float* dbf, float* dbg, float* dbp, int B,
                             int T, int Cr, int Cd, int Z, hipStream_t s) {
    int rc;
    if (dbf && (rc = launch_colsum(dab, (long long)T * 2 * Cd, 0, 2 * Cd, B, Z, T, Cd, dbf, s))) return rc;
    if (dbg && (rc = launch_colsum(dab + Cd, (long long)T * 2 * Cd, 0, 2 * Cd, B, Z, T, Cd, dbg, s))) return rc;
    if (dbp && dout && (rc = launch_colsum(dout, (long long)T * Cr, 0, Cr, B, 0, T, Cr, dbp, s))) return rc;
    return WN_OK;
}

int generic_pointwise_fwd(const float* x, const float* W, const float* bias, float* out, long long N, int Cin,
                          int Cout, int act, hipStream_t s) {
    hipLaunchKernelGGL(k_pointwise_fwd, dim3(cdiv(N * Cout, kThreads)), dim3(kThreads), 0, s, x, W, bias, out, N,
                       Cin, Cout, act);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int generic_pointwise_bwd(const float* x, const float* W, const float* dout, float* dx, float* dW, float* dbias,
                          long long N, int Cin, int Cout, int act, hipStream_t s) {
    if (dx) {
        hipLaunchKernelGGL(k_pointwise_bwd_dx, dim3(cdiv(N * Cin, kThreads)), dim3(kThreads), 0, s, x, W, dout,
                           dx, N, Cin, Cout, act);
        WN_LAUNCH_CHECK();
    }
    int rc;
    if (dW) {
        WgradArgs a{};
        a.A = dout; a.a_bs = 0; a.a_t0 = 0; a.lda = Cout;
        a.Bm = x; a.B2 = nullptr; a.b_bs = 0; a.b_t0 = 0; a.ldb = Cin;
        a.act = act; a.nB = 1; a.tmin = 0; a.nT = (int)N; a.M = Cout; a.K = Cin;
        a.dW = dW; a.sm = Cin; a.sk = 1;
        if ((rc = launch_wgrad(a, s))) return rc;
    }
    if (dbias && (rc = launch_colsum(dout, 0, 0, Cout, 1, 0, (int)N, Cout, dbias, s))) return rc;
    return WN_OK;
}

int generic_skip_sum_fwd(int L, const float* const* z, const float* const* Ws, const float* const* bs,
                         const int* cd, float* skip, int B, int T, int t_off, int Tw, int Cs, int accumulate,
                         hipStream_t s) {
    long long total = (long long)B * Tw * Cs;
    for (int l0 = 0; l0 < L; l0 += WN_MAX_SRC) {
        SkipArgs a{};
        a.L = (L - l0 < WN_MAX_SRC) ? L - l0 : WN_MAX_SRC;
        for (int l = 0; l < a.L; ++l) {
            a.z[l] = z[l0 + l]; a.Ws[l] = Ws[l0 + l]; a.bs[l] = bs ? bs[l0 + l] : nullptr; a.cd[l] = cd[l0 + l];
        }
        hipLaunchKernelGGL(k_skip_sum_fwd, dim3(cdiv(total, kThreads)), dim3(kThreads), 0, s, a, skip, B, T,
                           t_off, Tw, Cs, (accumulate || l0 > 0) ? 1 : 0);
        WN_LAUNCH_CHECK();
    }
    return WN_OK;
}

int generic_skip_bwd_dz(int L, const float* const* Ws, const int* cd, const float* dskip, float* const* dz,
                        int B, int T, int t_off, int Tw, int Cs, hipStream_t s) {
    for (int l = 0; l < L; ++l) {
        long long total = (long long)B * T * cd[l];
        hipLaunchKernelGGL(k_skip_bwd_dz, dim3(cdiv(total, kThreads)), dim3(kThreads), 0, s, Ws[l], dskip, dz[l],
                           B, T, t_off, Tw, Cs, cd[l]);
        WN_LAUNCH_CHECK();
    }
    return WN_OK;
}

int generic_skip_bwd_dw(int L, const float* const* z, const int* cd, const float* dskip, float* const* dWs,
                        float* const* dbs, int B, int T, int t_off, int Tw, int Cs, hipStream_t s) {
    int rc;
    for (int l = 0; l < L; ++l) {
        if (dWs && dWs[l]) {
            WgradArgs a{};
            a.A = dskip; a.a_bs = (long long)Tw * Cs; a.a_t0 = 0; a.lda = Cs;
            a.Bm = z[l]; a.B2 = nullptr; a.b_bs = (long long)T * cd[l]; a.b_t0 = t_off; a.ldb = cd[l];
            a.act = WN_ACT_NONE; a.nB = B; a.tmin = 0; a.nT = Tw; a.M = Cs; a.K = cd[l];
            a.dW = dWs[l]; a.sm = cd[l]; a.sk = 1;
            if ((rc = launch_wgrad(a, s))) return rc;
        }
        if (dbs && dbs[l] &&
            (rc = launch_colsum(dskip, (long long)Tw * Cs, 0, Cs, B, 0, Tw, Cs, dbs[l], s)))
            return rc;
    }
    return WN_OK;
}

int generic_softmax(const float* logits, float* prob, long long N, int Q, hipStream_t s) {
    hipLaunchKernelGGL(k_softmax, dim3(cdiv(N, 4)), dim3(256), 0, s, logits, prob, N, Q);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int generic_xent_count(const int32_t* target, long long N, int Q, float* loss, int* ncnt_out, hipStream_t s) {
    int ncnt = cdiv(N, 2048);
    ncnt = ncnt > kXentCnt ? kXentCnt : (ncnt < 1 ? 1 : ncnt);
    hipLaunchKernelGGL(k_xent_count, dim3(ncnt), dim3(1024), 0, s, target, N, Q, loss);
    WN_LAUNCH_CHECK();
    *ncnt_out = ncnt;
    return WN_OK;
}
int generic_xent_final(float* loss, int nblocks, long long n_norm, int ncnt, hipStream_t s) {
    hipLaunchKernelGGL(k_xent_final, dim3(1), dim3(256), 0, s, loss, nblocks, n_norm, ncnt);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int generic_softmax_xent(const float* logits, const int32_t* target, float* loss, float* dlogits, long long N,
                         int Q, long long n_norm, hipStream_t s) {
    int blocks = cdiv(N, 4);
    if (blocks > kXentBlocks) blocks = kXentBlocks;
    // n_norm > 0: that many rows count; 0: all N; < 0: counted on the device (labels in [0, Q)), one small launch
    const long long nn = n_norm > 0 ? n_norm : (n_norm == 0 ? N : -1);
    int ncnt = 0;
    if (nn < 0) {
        ncnt = cdiv(N, 2048);
        ncnt = ncnt > kXentCnt ? kXentCnt : (ncnt < 1 ? 1 : ncnt);
        hipLaunchKernelGGL(k_xent_count, dim3(ncnt), dim3(1024), 0, s, target, N, Q, loss);
    }
    hipLaunchKernelGGL(k_softmax_xent, dim3(blocks), dim3(256), 0, s, logits, target, loss, dlogits, N, Q, nn, ncnt);
    hipLaunchKernelGGL(k_xent_final, dim3(1), dim3(256), 0, s, loss, blocks, nn, ncnt);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

// bits of max |x[i]| (0 for an empty or all-zero array): block maxima, one atomic per block
__global__ void k_absmax(const float* __restrict__ x, long long n4, long long n, unsigned* __restrict__ slot) {
    float m = 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {       // four 16-byte loads in flight per thread
        const float4 v0 = reinterpret_cast<const float4*>(x)[i], v1 = reinterpret_cast<const float4*>(x)[i + stride];
        const float4 v2 = reinterpret_cast<const float4*>(x)[i + 2 * stride], v3 = reinterpret_cast<const float4*>(x)[i + 3 * stride];
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))),
                           fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w)))));
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), fmaxf(fabsf(v2.z), fabsf(v2.w))),
                           fmaxf(fmaxf(fabsf(v3.x), fabsf(v3.y)), fmaxf(fabsf(v3.z), fabsf(v3.w)))));
    }
    for (; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0)
        for (long long i = 4 * n4 + threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    __shared__ float part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, part[w]);
        atomicMax(slot, __float_as_uint(m));
    }
}
// One word = 0, by a KERNEL.  hipMemsetAsync under stream capture becomes a memset node, and with graph replays launched back
// to back a memset node in front of a kernel node was observed NOT to be ordered before it (the
// multi-layer backward spun on a counter that still held the previous replay's value).  Every word a following kernel
// accumulates into with atomicMax is therefore cleared by a kernel node.
__global__ void k_zero_word(unsigned* w) { *w = 0u; }
int generic_zero_word(unsigned* w, hipStream_t s) {
    hipLaunchKernelGGL(k_zero_word, dim3(1), dim3(1), 0, s, w);
    WN_LAUNCH_CHECK();
    return WN_OK;
}
int generic_absmax(const float* x, long long n, unsigned* slot, hipStream_t s) {
    if (int rc = generic_zero_word(slot, s)) return rc;
    if (n <= 0) return WN_OK;
    if (reinterpret_cast<uintptr_t>(x) & 15) { wn::set_error("absmax: the array must be 16-byte aligned"); return WN_EARG; }
    // one atomicMax per block, and atomics on ONE address retire at ~13 ns each: 2,048 blocks of 256 threads spent 27 of the
    // pass's 34 us (100 MB) queueing there; 256 blocks of 1,024 threads keep 16 MB in flight and queue for 3 us
    long long blocks = (n / 4 + 1023) / 1024;
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_absmax, dim3((unsigned)blocks), dim3(1024), 0, s, x, n / 4, n, slot);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int generic_transpose(const float* src, float* dst, int batch, int R, int Cc, hipStream_t s) {
    dim3 grid(cdiv(Cc, 32), cdiv(R, 32), batch);
    hipLaunchKernelGGL(k_transpose, grid, dim3(32, 8), 0, s, src, dst, R, Cc);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int generic_sample(const float* prob, const double* u, int32_t* out, int n, int Q, hipStream_t s) {
    hipLaunchKernelGGL(k_sample, dim3(cdiv(n, 64)), dim3(64), 0, s, prob, u, out, n, Q);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int generic_mulaw_encode_pcm16(const int16_t* pcm, const int32_t* lut, int32_t* tok, long long n, hipStream_t s) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_mulaw_encode_pcm16, dim3(blocks), dim3(256), 0, s, pcm, lut, tok, n);
    WN_LAUNCH_CHECK();
    return WN_OK;
}
int generic_mulaw_decode(const int32_t* tok, const float* table, float* out, long long n, int Q, hipStream_t s) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_mulaw_decode, dim3(blocks), dim3(256), 0, s, tok, table, out, n, Q);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int generic_sqnorm(const float* g, const float* p, long long n, float gmult, float wd, float* out,
                   hipStream_t s) {
    int blocks = (int)((n + 1023) / 1024);
    if (blocks > kNormBlocks) blocks = kNormBlocks;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_sqnorm, dim3(blocks), dim3(256), 0, s, g, p, n, gmult, wd, out);
    hipLaunchKernelGGL(k_sqnorm_final, dim3(1), dim3(256), 0, s, out, blocks);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int generic_adam(float* p, const float* g, float* m, float* v, long long n, float lr_t, float b1, float b2,
                 float eps, float wd, const float* sqnorm, float clip, float gmult, const float* lr_dev,
                 float dscale, hipStream_t s) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, s, p, g, m, v, n, lr_t, b1, b2, eps, wd, sqnorm, clip,
                       gmult, lr_dev, dscale);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int generic_rule(int rule, float* p, const float* g, float* s1, float* s2, long long n, float lr, float hy, float eps,
                 float wd, const float* sqnorm, float clip, float gmult, const float* lr_dev, hipStream_t s) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
#define RULE_LAUNCH(R) hipLaunchKernelGGL(k_rule<R>, dim3(blocks), dim3(256), 0, s, p, g, s1, s2, n, lr, hy, eps, wd, \
                                          sqnorm, clip, gmult, lr_dev)
    switch (rule) {
        case 0: RULE_LAUNCH(0); break;
        case 1: RULE_LAUNCH(1); break;
        case 2: RULE_LAUNCH(2); break;
        case 3: RULE_LAUNCH(3); break;
        case 4: RULE_LAUNCH(4); break;
        default: RULE_LAUNCH(5); break;
    }
#undef RULE_LAUNCH
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int generic_scale_by_dev(float* x, const float* sdev, long long n, hipStream_t s) {
    const long long n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n / 4 : 0;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_scale_by_dev, dim3(blocks), dim3(256), 0, s, x, sdev, n4, n);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

}  // namespace wn
