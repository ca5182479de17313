// Queue-cached autoregressive decoder (FasterWaveNet._forward_one_step, faster_wavenet.py:50-113,
// driven by train_audio/generate.py:24-43).
//
// What the reference keeps per layer is a full-window cache (1,C,1,W) that it rolls by one column
// every step (faster_wavenet.py:72-73, 90-96) although only columns -1 and -1-d are ever read
// (wavenet.py:286,290,354).  Here each layer keeps a ring of its last (fw-1)*d INPUT columns in HBM
// (0.52 MB in total at 4 x 10 layers of 32 channels), the head runs on the newest column only, and
// the whole sample loop -- network step, softmax, numpy-compatible categorical draw, feedback of
// the drawn token -- runs inside ONE persistent workgroup, so a 16,000-sample utterance is one
// launch.  Sample n+1 depends on sample n through all layers, so there is nothing to spread
// over other CUs without paying a cross-CU hand-off per layer (MI355X_MICROARCH price list).
//
// Weights are re-packed once per handle into transposed form (consecutive threads read consecutive
// addresses): gate  WfgT[k*Cr+c][o2]  (o2 < 2Cd: filter then gate),  projections  WpsT[cd][o]
// (o < Cr+Cs: residual rows then skip rows), head  WhT[ci][co], first causal layer W0t[q][k][c].
#include <new>
#include <vector>

#include "wn_kernels.hpp"
#include "decoder_types.hpp"

namespace wn {

static constexpr int kDecThreads = 512;

struct Decoder {
    DecMeta meta{};
    std::vector<DecCausal> causal;
    std::vector<DecLayer> layers;
    std::vector<DecHead> heads;
    std::vector<int> causal_ch, cd, head_ch;
    float* arena = nullptr;       // packed weights + rings
    long long arena_floats = 0;
    int* tok_ring = nullptr;      // fwc-1 previous tokens
    void* dmeta = nullptr;        // device copies of causal/layers/heads tables
    DecCausal* d_causal = nullptr;
    DecLayer* d_layers = nullptr;
    DecHead* d_heads = nullptr;
    long long step = 0;           // index of the next column to be consumed
    size_t lds_bytes = 0;
    float* fastP = nullptr;       // packed weights of the specialised kernel (decoder_fast.hip), or NULL
    int head_bias_off = -1;
    bool three_wgs = true;        // wn_decoder_run on nine workgroups (decoder_fast.hip); WN_DECODER_ONE_WORKGROUP clears it
    bool ran_multi = false;       // a nine-workgroup run has been launched (its error entry is meaningful)
    unsigned long long src_key = 0;   // hash of the caller's weight POINTERS at the last pack: two handles packed from the same
                                      // model carry the same key (wn_decoder_run_batch's same_weights check)
};

// the shape decoder_fast.hip is written for (BASELINE.json config 4 with the reference's default biases)
static bool fast_shape(const WnDecoderDesc* d) {
    if (d->flags & WN_EXEC_FORCE_GENERIC) return false;
    if (!(d->Q == 256 && d->n_causal == 1 && d->fw_causal == 2 && d->fw == 2 && d->Cr == 32 && d->Cs == 256 &&
          d->n_head == 1 && d->head_channels[0] == 256 && d->head_channels[1] == 256 &&
          d->n_blocks * d->n_layers <= 128))
        return false;
    if (d->causal_b && d->causal_b[0]) return false;
    for (int l = 0; l < d->n_layers; ++l)
        if (d->cd[l] != 32) return false;
    for (int j = 0; j < d->n_blocks * d->n_layers; ++j)
        if ((d->bf && d->bf[j]) || (d->bg && d->bg[j]) || (d->bp && d->bp[j]) || (d->bs && d->bs[j])) return false;
    return true;
}

// ---- packing ---------------------------------------------------------------------------------
// dst[(k*Cin + c)*ostride + ooff + o] = src[(o*Cin + c)*fw + k]
__global__ void k_pack_conv_T(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int fw,
                              int ostride, int ooff) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cout * Cin * fw) return;
    int k = i % fw;
    int c = (i / fw) % Cin;
    int o = i / (fw * Cin);
    dst[((long long)k * Cin + c) * ostride + ooff + o] = src[i];
}
// first causal layer: dst[(q*fw + k)*C + c] = src[(c*Q + q)*fw + k]
__global__ void k_pack_embed(const float* __restrict__ src, float* __restrict__ dst, int C, int Q, int fw) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * Q * fw) return;
    int k = i % fw;
    int q = (i / fw) % Q;
    int c = i / (fw * Q);
    dst[((long long)q * fw + k) * C + c] = src[i];
}
__global__ void k_copy_off(const float* __restrict__ src, float* __restrict__ dst, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
// ring[(col mod D)*C + c] = src[col*C + c] for col in [W-D, W) (zero where col < 0)
__global__ void k_load_ring(const float* __restrict__ src, float* __restrict__ ring, int W, int D, int C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * C) return;
    int c = i % C;
    int m = i / C;                       // 0 .. D-1: column W-D+m
    long long col = (long long)W - D + m;
    float v = col >= 0 ? src[col * C + c] : 0.f;
    long long slot = ((col % D) + D) % D;
    ring[slot * C + c] = v;
}
__global__ void k_load_tok_ring(const int32_t* __restrict__ tokens, int* __restrict__ ring, int W, int D) {
    int m = threadIdx.x;
    if (m >= D) return;
    long long col = (long long)W - D + m;
    long long slot = ((col % D) + D) % D;
    ring[slot] = col >= 0 ? tokens[col] : 0;
}

// ---- the step loop ---------------------------------------------------------------------------
__device__ __forceinline__ int pmod(long long a, int D) {
    long long r = a % D;
    return (int)(r < 0 ? r + D : r);
}
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
    for (int o = 32; o > 0; o >>= 1) {
        float w = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, w) : v + w;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}

__global__ __launch_bounds__(kDecThreads) void k_decode(
    DecMeta M, const DecCausal* __restrict__ causal, const DecLayer* __restrict__ layers,
    const DecHead* __restrict__ heads, float* __restrict__ arena, int* __restrict__ tok_ring, long long n0,
    int nsteps, int first_token, const double* __restrict__ uniforms, int32_t* __restrict__ out_tokens,
    float* __restrict__ prob_out, int prob_stride, int apply_softmax, int do_sample) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, NT = blockDim.x;
    float* xcur = sm;                    // [maxc] current column of the residual stream / causal stack
    float* xnew = xcur + M.maxc;         // [maxc]
    float* ab = xnew + M.maxc;           // [2*maxc] gate pre-activations
    float* zz = ab + 2 * M.maxc;         // [maxc]
    float* skip = zz + M.maxc;           // [maxc]
    float* hbuf = skip + M.maxc;         // [maxc]
    float* red = hbuf + M.maxc;          // [16]
    __shared__ int s_token;
    if (tid == 0) s_token = first_token;
    __syncthreads();

    for (int it = 0; it < nsteps; ++it) {
        const long long n = n0 + it;
        const int token = s_token;
        // ---- causal layer 0: two (fwc) gathered rows of the packed table ---------------------
        {
            const DecCausal L = causal[0];
            const int Dc = M.fwc - 1;
            for (int c = tid; c < L.cout; c += NT) {
                float acc = L.b >= 0 ? arena[L.b + c] : 0.f;
                for (int k = 0; k < M.fwc; ++k) {
                    int m = M.fwc - 1 - k;      // age of the tap
                    int q = m == 0 ? token : tok_ring[pmod(n - m, Dc)];
                    acc += arena[L.w + ((long long)q * M.fwc + k) * L.cout + c];
                }
                xcur[c] = acc;
            }
            __syncthreads();
            if (tid == 0 && Dc > 0) tok_ring[pmod(n, Dc)] = token;
        }
        // ---- further causal layers (d = 1), generic ------------------------------------------
        for (int i = 1; i < M.ncausal; ++i) {
            const DecCausal L = causal[i];
            const int Dc = M.fwc - 1;
            for (int o = tid; o < L.cout; o += NT) {
                float acc = L.b >= 0 ? arena[L.b + o] : 0.f;
                for (int k = 0; k < M.fwc; ++k) {
                    int m = M.fwc - 1 - k;
                    const float* src = m == 0 ? nullptr : arena + L.ring + (long long)pmod(n - m, Dc) * L.cin;
                    for (int c = 0; c < L.cin; ++c) {
                        float xv = m == 0 ? xcur[c] : src[c];
                        acc += arena[L.w + ((long long)k * L.cin + c) * L.cout + o] * xv;
                    }
                }
                xnew[o] = acc;
            }
            __syncthreads();
            if (Dc > 0)
                for (int c = tid; c < L.cin; c += NT) arena[L.ring + (long long)pmod(n, Dc) * L.cin + c] = xcur[c];
            __syncthreads();
            for (int o = tid; o < L.cout; o += NT) xcur[o] = xnew[o];
            __syncthreads();
        }
        for (int o = tid; o < M.Cs; o += NT) skip[o] = 0.f;
        // ---- residual layers ------------------------------------------------------------------
        for (int j = 0; j < M.nlayers; ++j) {
            const DecLayer L = layers[j];
            const int D = (M.fw - 1) * L.d;
            const int n2 = 2 * L.cd;
            // gate: ab[o2] = b + sum_k sum_c WfgT[k*Cr+c][o2] x[n-(fw-1-k)d][c]
            for (int o = tid; o < n2; o += NT) {
                float acc = L.bfg >= 0 ? arena[L.bfg + o] : 0.f;
                for (int k = 0; k < M.fw; ++k) {
                    int m = M.fw - 1 - k;
                    const float* w = arena + L.wfg + (long long)k * M.Cr * n2 + o;
                    if (m == 0) {
                        for (int c = 0; c < M.Cr; ++c) acc += w[(long long)c * n2] * xcur[c];
                    } else {
                        const float* src = arena + L.ring + (long long)pmod(n - (long long)m * L.d, D) * M.Cr;
                        for (int c = 0; c < M.Cr; ++c) acc += w[(long long)c * n2] * src[c];
                    }
                }
                ab[o] = acc;
            }
            __syncthreads();
            for (int o = tid; o < L.cd; o += NT) zz[o] = fast_tanh(ab[o]) * fast_sigmoid(ab[L.cd + o]);
            // the ring slot of column n is the one that held column n-(fw-1)d: all reads are done
            if (D > 0)
                for (int c = tid; c < M.Cr; c += NT) arena[L.ring + (long long)pmod(n, D) * M.Cr + c] = xcur[c];
            __syncthreads();
            // projections: rows [0,Cr) residual, rows [Cr,Cr+Cs) skip
            const int no = M.Cr + M.Cs;
            for (int o = tid; o < no; o += NT) {
                float acc = L.bps >= 0 ? arena[L.bps + o] : 0.f;
                const float* w = arena + L.wps + o;
                for (int c = 0; c < L.cd; ++c) acc += w[(long long)c * no] * zz[c];
                if (o < M.Cr) xnew[o] = acc + xcur[o];
                else skip[o - M.Cr] += acc;
            }
            __syncthreads();
            for (int c = tid; c < M.Cr; c += NT) xcur[c] = xnew[c];
            __syncthreads();
        }
        // ---- head on the newest column only ---------------------------------------------------
        float* hin = skip;
        float* hout = hbuf;
        for (int i = 0; i < M.nhead; ++i) {
            const DecHead H = heads[i];
            for (int o = tid; o < H.cout; o += NT) {
                float acc = H.b >= 0 ? arena[H.b + o] : 0.f;
                const float* w = arena + H.w + o;
                for (int c = 0; c < H.cin; ++c) acc += w[(long long)c * H.cout] * act_apply(hin[c], M.head_act);
                hout[o] = acc;
            }
            __syncthreads();
            float* t = hin; hin = hout; hout = t;
        }
        // hin holds the Q logits
        if (apply_softmax) {
            float m = -INFINITY;
            for (int q = tid; q < M.Q; q += NT) m = fmaxf(m, hin[q]);
            m = block_reduce(m, true, red);
            float s = 0.f;
            for (int q = tid; q < M.Q; q += NT) s += expf(hin[q] - m);
            s = block_reduce(s, false, red);
            float inv = 1.f / s;
            __syncthreads();
            for (int q = tid; q < M.Q; q += NT) hin[q] = expf(hin[q] - m) * inv;
            __syncthreads();
        }
        if (prob_out)
            for (int q = tid; q < M.Q; q += NT) prob_out[(long long)it * prob_stride + q] = hin[q];
        if (do_sample) {
            if (tid == 0) {
                // numpy legacy choice: float64 cumsum, divide by the total, first index with cdf > u
                double tot = 0.0;
                for (int q = 0; q < M.Q; ++q) tot += (double)hin[q];
                double c = 0.0, u = uniforms[it];
                int idx = M.Q;
                for (int q = 0; q < M.Q; ++q) {
                    c += (double)hin[q];
                    if (c / tot > u) { idx = q; break; }
                }
                if (idx >= M.Q) idx = M.Q - 1;
                s_token = idx;
                out_tokens[it] = idx;
            }
        }
        __syncthreads();
    }
}

// ---- host side -------------------------------------------------------------------------------
static int validate(const WnDecoderDesc* d) {
    WN_CHECK_ARG(d, "decoder: desc is NULL");
    WN_CHECK_ARG(d->Q > 0 && d->fw_causal > 0 && d->n_causal > 0 && d->fw > 0 && d->n_blocks > 0 &&
                     d->n_layers > 0 && d->Cr > 0 && d->Cs > 0 && d->n_head > 0,
                 "decoder: non-positive size in desc");
    WN_CHECK_ARG(d->causal_channels && d->cd && d->head_channels && d->causal_W && d->Wf && d->Wg && d->Wp &&
                     d->Ws && d->head_W,
                 "decoder: NULL table in desc");
    WN_CHECK_ARG(d->causal_channels[d->n_causal - 1] == d->Cr, "decoder: last causal width != Cr");
    WN_CHECK_ARG(d->head_channels[0] == d->Cs && d->head_channels[d->n_head] == d->Q,
                 "decoder: head channels must run Cs ... Q");
    WN_CHECK_ARG(d->head_act == WN_ACT_RELU || d->head_act == WN_ACT_ELU || d->head_act == WN_ACT_NONE,
                 "decoder: bad head_act");
    return WN_OK;
}

static inline long long align64(long long x) { return (x + 63) & ~63ll; }

static int pack_weights(Decoder* D, const WnDecoderDesc* d, hipStream_t s) {
    const int T = 256;
    {   // FNV-1a over every weight / bias pointer of the description, in a fixed order
        unsigned long long h = 1469598103934665603ull;
        auto mix = [&](const void* p) { h = (h ^ (unsigned long long)(uintptr_t)p) * 1099511628211ull; };
        for (int i = 0; i < d->n_causal; ++i) { mix(d->causal_W[i]); mix(d->causal_b ? d->causal_b[i] : nullptr); }
        for (int j = 0; j < (int)D->layers.size(); ++j) {
            mix(d->Wf[j]); mix(d->Wg[j]); mix(d->Wp[j]); mix(d->Ws[j]);
            mix(d->bf ? d->bf[j] : nullptr); mix(d->bg ? d->bg[j] : nullptr); mix(d->bp ? d->bp[j] : nullptr); mix(d->bs ? d->bs[j] : nullptr);
        }
        for (int i = 0; i < d->n_head; ++i) { mix(d->head_W[i]); mix(d->head_b ? d->head_b[i] : nullptr); }
        D->src_key = h;
    }
    for (int i = 0; i < d->n_causal; ++i) {
        const DecCausal& L = D->causal[i];
        int n = L.cout * L.cin * d->fw_causal;
        WN_CHECK_ARG(d->causal_W[i], "decoder: causal_W[%d] NULL", i);
        if (i == 0)
            hipLaunchKernelGGL(k_pack_embed, dim3(cdiv(n, T)), dim3(T), 0, s, d->causal_W[i], D->arena + L.w, L.cout,
                               L.cin, d->fw_causal);
        else
            hipLaunchKernelGGL(k_pack_conv_T, dim3(cdiv(n, T)), dim3(T), 0, s, d->causal_W[i], D->arena + L.w,
                               L.cout, L.cin, d->fw_causal, L.cout, 0);
        if (L.b >= 0)
            hipLaunchKernelGGL(k_copy_off, dim3(cdiv(L.cout, T)), dim3(T), 0, s, d->causal_b[i], D->arena + L.b,
                               L.cout);
    }
    for (int j = 0; j < (int)D->layers.size(); ++j) {
        const DecLayer& L = D->layers[j];
        int cd = L.cd, Cr = d->Cr, Cs = d->Cs;
        WN_CHECK_ARG(d->Wf[j] && d->Wg[j] && d->Wp[j] && d->Ws[j], "decoder: layer %d has a NULL weight", j);
        int n = cd * Cr * d->fw;
        hipLaunchKernelGGL(k_pack_conv_T, dim3(cdiv(n, T)), dim3(T), 0, s, d->Wf[j], D->arena + L.wfg, cd, Cr, d->fw,
                           2 * cd, 0);
        hipLaunchKernelGGL(k_pack_conv_T, dim3(cdiv(n, T)), dim3(T), 0, s, d->Wg[j], D->arena + L.wfg, cd, Cr, d->fw,
                           2 * cd, cd);
        hipLaunchKernelGGL(k_pack_conv_T, dim3(cdiv(Cr * cd, T)), dim3(T), 0, s, d->Wp[j], D->arena + L.wps, Cr, cd, 1,
                           Cr + Cs, 0);
        hipLaunchKernelGGL(k_pack_conv_T, dim3(cdiv(Cs * cd, T)), dim3(T), 0, s, d->Ws[j], D->arena + L.wps, Cs, cd, 1,
                           Cr + Cs, Cr);
        if (L.bfg >= 0) {
            hipLaunchKernelGGL(k_copy_off, dim3(cdiv(cd, T)), dim3(T), 0, s, d->bf[j], D->arena + L.bfg, cd);
            hipLaunchKernelGGL(k_copy_off, dim3(cdiv(cd, T)), dim3(T), 0, s, d->bg[j], D->arena + L.bfg + cd, cd);
        }
        if (L.bps >= 0) {
            hipLaunchKernelGGL(k_copy_off, dim3(cdiv(Cr, T)), dim3(T), 0, s, d->bp[j], D->arena + L.bps, Cr);
            hipLaunchKernelGGL(k_copy_off, dim3(cdiv(Cs, T)), dim3(T), 0, s, d->bs[j], D->arena + L.bps + Cr, Cs);
        }
    }
    for (int i = 0; i < d->n_head; ++i) {
        const DecHead& H = D->heads[i];
        WN_CHECK_ARG(d->head_W[i], "decoder: head_W[%d] NULL", i);
        hipLaunchKernelGGL(k_pack_conv_T, dim3(cdiv(H.cin * H.cout, T)), dim3(T), 0, s, d->head_W[i], D->arena + H.w,
                           H.cout, H.cin, 1, H.cout, 0);
        if (H.b >= 0)
            hipLaunchKernelGGL(k_copy_off, dim3(cdiv(H.cout, T)), dim3(T), 0, s, d->head_b[i], D->arena + H.b, H.cout);
    }
    WN_LAUNCH_CHECK();
    if (D->fastP) return decode_fast_pack(d, D->fastP, s);
    return WN_OK;
}

}  // namespace wn

using namespace wn;

extern "C" {

int wn_decoder_create(void** handle, const WnDecoderDesc* d, void* stream) {
    WN_CHECK_ARG(handle, "wn_decoder_create: handle is NULL");
    int rc = validate(d);
    if (rc) return rc;
    Decoder* D = new (std::nothrow) Decoder();
    WN_CHECK_ARG(D, "wn_decoder_create: out of host memory");
    DecMeta& M = D->meta;
    M.Q = d->Q; M.fwc = d->fw_causal; M.ncausal = d->n_causal; M.fw = d->fw;
    M.nlayers = d->n_blocks * d->n_layers; M.Cr = d->Cr; M.Cs = d->Cs; M.nhead = d->n_head;
    M.head_act = d->head_act;
    long long off = 0;
    int maxc = d->Q > d->Cs ? d->Q : d->Cs;
    if (d->Cr > maxc) maxc = d->Cr;
    // layout of the arena
    int cin = d->Q;
    for (int i = 0; i < d->n_causal; ++i) {
        DecCausal L{};
        L.cin = cin; L.cout = d->causal_channels[i];
        if (L.cout > maxc) maxc = L.cout;
        L.w = (int)off; off = align64(off + (long long)L.cin * L.cout * d->fw_causal);
        bool hb = d->causal_b && d->causal_b[i];
        L.b = hb ? (int)off : -1; if (hb) off = align64(off + L.cout);
        L.ring = (int)off;
        if (i > 0) off = align64(off + (long long)(d->fw_causal - 1) * L.cin);
        D->causal.push_back(L);
        cin = L.cout;
    }
    for (int b = 0; b < d->n_blocks; ++b) {
        int dil = 1;
        for (int l = 0; l < d->n_layers; ++l) {
            int j = b * d->n_layers + l;
            DecLayer L{};
            L.cd = d->cd[l]; L.d = dil;
            if (2 * L.cd > 2 * maxc) maxc = L.cd;
            L.wfg = (int)off; off = align64(off + (long long)d->fw * d->Cr * 2 * L.cd);
            bool hb = d->bf && d->bf[j];
            WN_CHECK_ARG(hb == (d->bg && d->bg[j]), "decoder: bf/bg must both be present or absent");
            L.bfg = hb ? (int)off : -1; if (hb) off = align64(off + 2 * L.cd);
            L.wps = (int)off; off = align64(off + (long long)L.cd * (d->Cr + d->Cs));
            bool hp = d->bp && d->bp[j];
            WN_CHECK_ARG(hp == (d->bs && d->bs[j]), "decoder: bp/bs must both be present or absent");
            L.bps = hp ? (int)off : -1; if (hp) off = align64(off + d->Cr + d->Cs);
            L.ring = (int)off; off = align64(off + (long long)(d->fw - 1) * dil * d->Cr);
            D->layers.push_back(L);
            dil *= d->fw;
        }
    }
    for (int i = 0; i < d->n_head; ++i) {
        DecHead H{};
        H.cin = d->head_channels[i]; H.cout = d->head_channels[i + 1];
        if (H.cout > maxc) maxc = H.cout;
        H.w = (int)off; off = align64(off + (long long)H.cin * H.cout);
        bool hb = d->head_b && d->head_b[i];
        H.b = hb ? (int)off : -1; if (hb) off = align64(off + H.cout);
        D->heads.push_back(H);
    }
    if (off >= (1ll << 31)) { delete D; wn::set_error("decoder: arena too large"); return WN_ESHAPE; }
    M.maxc = maxc;
    D->arena_floats = off;
    D->lds_bytes = (size_t)(7 * maxc + 16) * sizeof(float);
    if (D->lds_bytes > 150 * 1024) { delete D; wn::set_error("decoder: channel width %d too large", maxc); return WN_ESHAPE; }
    hipStream_t s = as_stream(stream);
    size_t tab = D->causal.size() * sizeof(DecCausal) + D->layers.size() * sizeof(DecLayer) +
                 D->heads.size() * sizeof(DecHead);
#define DEC_HIP(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) { wn::set_error("%s: %s", #e, hipGetErrorString(e__)); wn_decoder_destroy(D); return WN_EHIP; } } while (0)
    DEC_HIP(hipMalloc(&D->arena, (size_t)off * sizeof(float)));
    DEC_HIP(hipMemsetAsync(D->arena, 0, (size_t)off * sizeof(float), s));
    DEC_HIP(hipMalloc(&D->tok_ring, sizeof(int) * (d->fw_causal > 1 ? d->fw_causal - 1 : 1)));
    DEC_HIP(hipMemsetAsync(D->tok_ring, 0, sizeof(int) * (d->fw_causal > 1 ? d->fw_causal - 1 : 1), s));
    DEC_HIP(hipMalloc(&D->dmeta, tab));
    char* p = (char*)D->dmeta;
    D->d_causal = (DecCausal*)p; p += D->causal.size() * sizeof(DecCausal);
    D->d_layers = (DecLayer*)p; p += D->layers.size() * sizeof(DecLayer);
    D->d_heads = (DecHead*)p;
    DEC_HIP(hipMemcpyAsync(D->d_causal, D->causal.data(), D->causal.size() * sizeof(DecCausal), hipMemcpyHostToDevice, s));
    DEC_HIP(hipMemcpyAsync(D->d_layers, D->layers.data(), D->layers.size() * sizeof(DecLayer), hipMemcpyHostToDevice, s));
    DEC_HIP(hipMemcpyAsync(D->d_heads, D->heads.data(), D->heads.size() * sizeof(DecHead), hipMemcpyHostToDevice, s));
    DEC_HIP(hipStreamSynchronize(s));      // the host vectors above must outlive the copies
    if (fast_shape(d)) {
        DEC_HIP(hipMalloc(&D->fastP, decode_fast_pack_floats(M.nlayers) * sizeof(float)));
        D->head_bias_off = D->heads[0].b;
        D->three_wgs = !(d->flags & WN_DECODER_ONE_WORKGROUP);
    }
#undef DEC_HIP
    rc = pack_weights(D, d, s);
    if (rc) { wn_decoder_destroy(D); return rc; }
    *handle = D;
    return WN_OK;
}

int wn_decoder_destroy(void* handle) {
    Decoder* D = (Decoder*)handle;
    if (!D) return WN_OK;
    if (D->arena) (void)hipFree(D->arena);
    if (D->fastP) (void)hipFree(D->fastP);
    if (D->tok_ring) (void)hipFree(D->tok_ring);
    if (D->dmeta) (void)hipFree(D->dmeta);
    delete D;
    return WN_OK;
}

int wn_decoder_update_weights(void* handle, const WnDecoderDesc* d, void* stream) {
    Decoder* D = (Decoder*)handle;
    WN_CHECK_ARG(D, "wn_decoder_update_weights: NULL handle");
    int rc = validate(d);
    if (rc) return rc;
    WN_CHECK_ARG(d->n_blocks * d->n_layers == D->meta.nlayers && d->Cr == D->meta.Cr && d->Cs == D->meta.Cs &&
                     d->Q == D->meta.Q && d->n_causal == D->meta.ncausal && d->n_head == D->meta.nhead,
                 "wn_decoder_update_weights: topology differs from the handle's");
    return pack_weights(D, d, as_stream(stream));
}

int wn_decoder_load_state(void* handle, const int32_t* tokens, int W, const float* const* causal_out,
                          const float* const* layer_in, void* stream) {
    Decoder* D = (Decoder*)handle;
    WN_CHECK_ARG(D && tokens && layer_in && W > 0, "wn_decoder_load_state: bad argument");
    hipStream_t s = as_stream(stream);
    const DecMeta& M = D->meta;
    if (M.fwc > 1) hipLaunchKernelGGL(k_load_tok_ring, dim3(1), dim3(64), 0, s, tokens, D->tok_ring, W, M.fwc - 1);
    for (int i = 1; i < M.ncausal; ++i) {
        WN_CHECK_ARG(causal_out && causal_out[i - 1], "wn_decoder_load_state: causal_out[%d] NULL", i - 1);
        int Dd = M.fwc - 1, C = D->causal[i].cin;
        if (Dd > 0)
            hipLaunchKernelGGL(k_load_ring, dim3(cdiv((long long)Dd * C, 256)), dim3(256), 0, s, causal_out[i - 1],
                               D->arena + D->causal[i].ring, W, Dd, C);
    }
    for (int j = 0; j < M.nlayers; ++j) {
        WN_CHECK_ARG(layer_in[j], "wn_decoder_load_state: layer_in[%d] NULL", j);
        int Dd = (M.fw - 1) * D->layers[j].d;
        if (Dd > 0)
            hipLaunchKernelGGL(k_load_ring, dim3(cdiv((long long)Dd * M.Cr, 256)), dim3(256), 0, s, layer_in[j],
                               D->arena + D->layers[j].ring, W, Dd, M.Cr);
    }
    WN_LAUNCH_CHECK();
    D->step = W;
    return WN_OK;
}

int wn_decoder_step(void* handle, int32_t token, float* prob, int apply_softmax, void* stream) {
    Decoder* D = (Decoder*)handle;
    WN_CHECK_ARG(D && prob, "wn_decoder_step: bad argument");
    WN_CHECK_ARG(token >= 0 && token < D->meta.Q, "wn_decoder_step: token %d outside [0,%d)", token, D->meta.Q);
    WN_CHECK_ARG(D->step + 1 < (1ll << 31), "wn_decoder_step: step counter overflow");
    if (D->fastP) {
        int rc = decode_fast_launch(D->fastP, D->meta.nlayers, D->head_bias_off >= 0 ? D->arena + D->head_bias_off : nullptr,
                                    D->arena + D->causal[0].w, D->d_layers, D->arena, D->tok_ring, D->step, 1, (int)token,
                                    nullptr, nullptr, prob, D->meta.Q, apply_softmax, 0, D->meta.head_act, false,
                                    as_stream(stream));
        if (rc) return rc;
        D->step += 1;
        return WN_OK;
    }
    hipLaunchKernelGGL(k_decode, dim3(1), dim3(kDecThreads), D->lds_bytes, as_stream(stream), D->meta, D->d_causal,
                       D->d_layers, D->d_heads, D->arena, D->tok_ring, D->step, 1, (int)token,
                       (const double*)nullptr, (int32_t*)nullptr, prob, D->meta.Q, apply_softmax, 0);
    WN_LAUNCH_CHECK();
    D->step += 1;
    return WN_OK;
}

int wn_decoder_run(void* handle, int32_t first_token, const double* uniforms, int n, int32_t* out_tokens,
                   float* prob_trace, void* stream) {
    Decoder* D = (Decoder*)handle;
    WN_CHECK_ARG(D && uniforms && out_tokens && n > 0, "wn_decoder_run: bad argument");
    WN_CHECK_ARG(first_token >= 0 && first_token < D->meta.Q, "wn_decoder_run: token outside [0,Q)");
    WN_CHECK_ARG(D->step + n < (1ll << 31), "wn_decoder_run: step counter overflow");
    if (D->fastP) {
        int rc = decode_fast_launch(D->fastP, D->meta.nlayers, D->head_bias_off >= 0 ? D->arena + D->head_bias_off : nullptr,
                                    D->arena + D->causal[0].w, D->d_layers, D->arena, D->tok_ring, D->step, n,
                                    (int)first_token, uniforms, out_tokens, prob_trace, D->meta.Q, 1, 1,
                                    D->meta.head_act, D->three_wgs, as_stream(stream));
        if (rc) return rc;
        D->ran_multi = D->three_wgs && n > 1;
        D->step += n;
        return WN_OK;
    }
    hipLaunchKernelGGL(k_decode, dim3(1), dim3(kDecThreads), D->lds_bytes, as_stream(stream), D->meta, D->d_causal,
                       D->d_layers, D->d_heads, D->arena, D->tok_ring, D->step, n, (int)first_token, uniforms,
                       out_tokens, prob_trace, D->meta.Q, 1, 1);
    WN_LAUNCH_CHECK();
    D->step += n;
    return WN_OK;
}

int wn_decoder_batch_max(void) { return kDecMaxBatch; }

int wn_decoder_run_batch(void* const* handles, int n_handles, const int32_t* first_tokens, const double* const* uniforms, int n,
                         int32_t* const* out_tokens, float* const* prob_traces, int same_weights, void* stream) {
    WN_CHECK_ARG(handles && first_tokens && uniforms && out_tokens && n_handles >= 1 && n > 0, "wn_decoder_run_batch: bad argument");
    WN_CHECK_SHAPE(n_handles <= kDecMaxBatch, "wn_decoder_run_batch: at most %d utterances per launch", kDecMaxBatch);
    const float* P[kDecMaxBatch]; const float* hb[kDecMaxBatch]; const float* E[kDecMaxBatch]; const DecLayer* ly[kDecMaxBatch];
    float* ar[kDecMaxBatch]; int* tr[kDecMaxBatch]; long long n0[kDecMaxBatch]; int ft[kDecMaxBatch];
    Decoder* D0 = (Decoder*)handles[0];
    for (int u = 0; u < n_handles; ++u) {
        Decoder* D = (Decoder*)handles[u];
        WN_CHECK_ARG(D && uniforms[u] && out_tokens[u], "wn_decoder_run_batch: NULL handle / uniforms / out_tokens of utterance %d", u);
        for (int v = 0; v < u; ++v) WN_CHECK_ARG(handles[v] != handles[u], "wn_decoder_run_batch: handle %d given twice", u);
        WN_CHECK_SHAPE(D->fastP && D->three_wgs, "wn_decoder_run_batch: needs the specialised decoder (config 4's shape) "
                                                 "without WN_DECODER_ONE_WORKGROUP");
        WN_CHECK_SHAPE(D->meta.nlayers == D0->meta.nlayers && D->meta.head_act == D0->meta.head_act && D->meta.Q == D0->meta.Q &&
                           D->meta.Cr == D0->meta.Cr && D->meta.Cs == D0->meta.Cs && D->meta.fw == D0->meta.fw &&
                           D->layers.size() == D0->layers.size(),
                       "wn_decoder_run_batch: the utterances' models differ (layers, channels, Q or head activation)");
        for (size_t l = 0; l < D->layers.size(); ++l)
            WN_CHECK_SHAPE(D->layers[l].d == D0->layers[l].d && D->layers[l].cd == D0->layers[l].cd,
                           "wn_decoder_run_batch: layer %d of utterance %d has another dilation / width than utterance 0's", (int)l, u);
        // same_weights = 1 reads ONE copy of the packed weights (handle 0's) for every utterance: only sound when every
        // handle was packed from the same model -- the same weight pointers at its last create / update_weights
        WN_CHECK_ARG(!same_weights || D->src_key == D0->src_key,
                     "wn_decoder_run_batch: same_weights = 1 but utterance %d's handle was packed from other weights than utterance 0's", u);
        WN_CHECK_ARG(first_tokens[u] >= 0 && first_tokens[u] < D->meta.Q, "wn_decoder_run_batch: token outside [0,Q)");
        WN_CHECK_ARG(D->step + n < (1ll << 31), "wn_decoder_run_batch: step counter overflow");
        P[u] = D->fastP; hb[u] = D->head_bias_off >= 0 ? D->arena + D->head_bias_off : nullptr;
        E[u] = D->arena + D->causal[0].w; ly[u] = D->d_layers; ar[u] = D->arena; tr[u] = D->tok_ring; n0[u] = D->step;
        ft[u] = (int)first_tokens[u];
    }
    WN_CHECK_SHAPE(decode_fast_batch_ok(D0->meta.nlayers, n_handles, n),
                   "wn_decoder_run_batch: %d utterances x 9 workgroups must all be resident on the device, and n >= 2", n_handles);
    const int rc = decode_fast_launch_batch(n_handles, P, D0->meta.nlayers, hb, E, ly, ar, tr, n0, n, ft, uniforms, out_tokens,
                                            prob_traces, D0->meta.Q, D0->meta.head_act, same_weights != 0, as_stream(stream));
    if (rc) return rc;
    for (int u = 0; u < n_handles; ++u) {
        Decoder* D = (Decoder*)handles[u];
        D->ran_multi = true;
        D->step += n;
    }
    return WN_OK;
}

int wn_decoder_status(void* handle, void* stream) {
    Decoder* D = (Decoder*)handle;
    WN_CHECK_ARG(D, "wn_decoder_status: NULL handle");
    if (!D->fastP || !D->ran_multi) {                   // nothing that could have given up has run
        WN_HIP(hipStreamSynchronize(as_stream(stream)));
        return WN_OK;
    }
    int gave_up = 0;
    const int rc = decode_fast_status(D->fastP, D->meta.nlayers, as_stream(stream), &gave_up);
    if (rc) return rc;
    if (gave_up) {
        wn::set_error("wn_decoder_run: a wait between the nine workgroups gave up (they were not all resident); the tokens of "
                      "that run are void -- re-create the decoder with WN_DECODER_ONE_WORKGROUP");
        return WN_ETIMEOUT;
    }
    return WN_OK;
}

}  // extern "C"
