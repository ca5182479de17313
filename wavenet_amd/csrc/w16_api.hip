// extern "C" surface and orchestration of the bf16-storage path (BASELINE config 5: 128 residual / dilation channels,
// 512 skip channels, bf16 activations in HBM, fp32 accumulation, fp32 master weights and gradients).
// Same division of labour as stack.hip: the reference's per-layer Python loop (wavenet.py:572-582) and Chainer's
// backward over it run here, on one stream, from caller-provided buffers only (no allocation, no synchronisation).
#include <vector>

#include "w16_gemm.hpp"
#include "wn_kernels.hpp"

using namespace w16;

namespace {

int zero_prefix(int T, int d, int fw) {                  // wavenet.py:303-340
    if (d == 1) return 0;
    int pad = ((-T) % d + d) % d;
    const int height = (T + pad) / d;
    if (height < fw) pad += (fw - height) * d;
    const int z = (fw - 1) * d - pad;
    return z > 0 ? z : 0;
}

const char* unsupported(const WnStackDesc* d) {
    if (!d || d->n_layers < 1 || !d->cd || !d->dilation || !d->Wf || !d->Wg || !d->Wp || !d->Ws) return "NULL descriptor table";
    if (d->Cr != 128 || d->fw != 2) return "needs 128 residual channels and filter width 2";
    if (d->Cs % 256 || d->Cs < 256) return "needs a multiple of 256 skip channels";
    if (d->n_layers > kMaxProb16 || (d->n_layers & 1)) return "needs an even number of layers, at most 48";
    for (int l = 0; l < d->n_layers; ++l) {
        if (d->cd[l] != 128) return "needs 128 dilation channels in every layer";
        if ((d->bf && d->bf[l]) || (d->bg && d->bg[l]) || (d->bp && d->bp[l]) || (d->bs && d->bs[l]))
            return "convolution / projection biases are not supported (the reference's default has none)";
    }
    return nullptr;
}

size_t pack_elems(const WnStackDesc* d) {
    const size_t L = d->n_layers;
    return L * kLayerImg + 2 * (size_t)d->Cs * L * 128;     // layer images, skip matrix [Cs][128 L], dz matrix [128 L][Cs]
}

struct BwdWs { bf16* dzs; bf16* dadg; bf16* dxb[2]; float* parts; float* wgparts; unsigned* sync; size_t bytes; };
BwdWs carve(const WnStackDesc* d, int B, int T, int t_off, char* ws) {
    const size_t L = d->n_layers, n = (size_t)B * T, nw = (size_t)B * (T - t_off);
    BwdWs r{};
    size_t o = 0;
    auto take = [&](size_t bytes) { char* p = ws ? ws + o : nullptr; o += (bytes + 255) & ~(size_t)255; return p; };
    r.dzs = reinterpret_cast<bf16*>(take(L * nw * 128 * 2));
    r.dadg = reinterpret_cast<bf16*>(take(L * n * 256 * 2));
    r.dxb[0] = reinterpret_cast<bf16*>(take(n * 128 * 2));
    r.dxb[1] = reinterpret_cast<bf16*>(take(n * 128 * 2));
    r.parts = reinterpret_cast<float*>(take(L * (size_t)dx_grid(B, T) * 128 * 128 * 4));
    r.wgparts = reinterpret_cast<float*>(take(kWgPartBytes));        // per-workgroup blocks of the time contractions
    r.sync = reinterpret_cast<unsigned*>(take(bwd_multi_sync_words(B, T) * sizeof(unsigned)));   // dataflow words of k16_bwd_multi
    r.bytes = o;
    return r;
}

}  // namespace

#define W16_CHECK_DESC(d)                                                            \
    do {                                                                             \
        const char* why__ = unsupported(d);                                          \
        if (why__) { wn::set_error("%s: bf16 storage %s", __func__, why__); return WN_ESHAPE; } \
    } while (0)

extern "C" {

int wn16_supported(const WnStackDesc* d) { return unsupported(d) == nullptr ? 1 : 0; }

size_t wn16_pack_elems(const WnStackDesc* d) { return unsupported(d) ? 0 : pack_elems(d); }

int wn16_pack_stack(const WnStackDesc* d, uint16_t* pack, void* stream) {
    W16_CHECK_DESC(d);
    WN_CHECK_ARG(pack, "wn16_pack_stack: pack is NULL");
    hipStream_t s = wn::as_stream(stream);
    const int L = d->n_layers;
    bf16* img = reinterpret_cast<bf16*>(pack);
    int rc = pack_layers(L, d->Wf, d->Wg, d->Wp, img, s);
    if (rc) return rc;
    bf16* skipW = img + (size_t)L * kLayerImg;
    bf16* dzW = skipW + (size_t)d->Cs * L * 128;
    if ((rc = pack_mat(L, d->Ws, skipW, d->Cs, 128, 0, s))) return rc;
    return pack_mat(L, d->Ws, dzW, d->Cs, 128, 1, s);
}

int wn16_embed_fwd(const int32_t* idx, const float* W, const float* bias, uint16_t* out, int B, int T, int Q, int C,
                   int fw, void* stream) {
    wn::ProfScope prof__("wn16_embed_fwd", stream);
    WN_CHECK_ARG(idx && W && out && B > 0 && T > 0 && Q > 0, "wn16_embed_fwd: bad argument");
    WN_CHECK_SHAPE(fw == 2 && C % 8 == 0, "wn16_embed_fwd: needs filter width 2 and a multiple of 8 channels");
    return embed_fwd16(idx, W, bias, reinterpret_cast<bf16*>(out), B, T, Q, C, wn::as_stream(stream));
}

size_t wn16_embed_bwd_workspace_bytes(int B, int T) { return B > 0 && T > 0 ? embed_bwd16_ws_bytes(B, T) : 0; }
int wn16_embed_bwd(const int32_t* idx, const uint16_t* dout, float* dW, float* dbias, int B, int T, int Q, int C, int fw,
                   void* ws, size_t ws_bytes, void* stream) {
    wn::ProfScope prof__("wn16_embed_bwd", stream);
    WN_CHECK_ARG(idx && dout && dW && ws && B > 0 && T > 0, "wn16_embed_bwd: bad argument");
    WN_CHECK_SHAPE(fw == 2 && C == 128 && Q == 256, "wn16_embed_bwd: needs filter width 2, 128 channels and 256 token values");
    WN_CHECK_ARG(ws_bytes >= embed_bwd16_ws_bytes(B, T), "wn16_embed_bwd: workspace too small");
    return embed_bwd16(idx, reinterpret_cast<const bf16*>(dout), dW, dbias, B, T, ws, wn::as_stream(stream));
}

int wn16_cvt_to_bf16(const float* src, uint16_t* dst, int64_t n, void* stream) {
    WN_CHECK_ARG(src && dst && n > 0, "wn16_cvt_to_bf16: bad argument");
    return cvt_f2b(src, reinterpret_cast<bf16*>(dst), n, wn::as_stream(stream));
}
int wn16_cvt_to_f32(const uint16_t* src, float* dst, int64_t n, void* stream) {
    WN_CHECK_ARG(src && dst && n > 0, "wn16_cvt_to_f32: bad argument");
    return cvt_b2f(reinterpret_cast<const bf16*>(src), dst, n, wn::as_stream(stream));
}

int wn16_stack_fwd(const WnStackDesc* d, const uint16_t* pack, const uint16_t* x, uint16_t* xs, uint16_t* z,
                   uint16_t* skip, int B, int T, int t_off, int compat_zero_prefix, void* stream) {
    W16_CHECK_DESC(d);
    WN_CHECK_ARG(pack && x && xs && z && B > 0 && T > 0 && t_off >= 0 && t_off < T, "wn16_stack_fwd: bad argument");
    hipStream_t s = wn::as_stream(stream);
    const int L = d->n_layers;
    const size_t n = (size_t)B * T;
    const bf16* img = reinterpret_cast<const bf16*>(pack);
    const bf16* in = reinterpret_cast<const bf16*>(x);
    bf16* xsb = reinterpret_cast<bf16*>(xs);
    bf16* zb = reinterpret_cast<bf16*>(z);
    int rc;
    {
        wn::ProfScope prof__("wn16_layer_fwd", stream);
        for (int l = 0; l < L; ++l) {
            const int Z = compat_zero_prefix ? zero_prefix(T, d->dilation[l], 2) : 0;
            bf16* out = xsb + (size_t)l * n * 128;
            if ((rc = fwd_layer(in, img + (size_t)l * kLayerImg, out, zb + (size_t)l * n * 128, B, T, d->dilation[l], Z, s)))
                return rc;
            in = out;
        }
    }
    if (!skip) return WN_OK;
    wn::ProfScope prof__("wn16_skip_sum_fwd", stream);
    CG16 a{};
    for (int l = 0; l < L; ++l) { a.X[l] = zb + (size_t)l * n * 128; a.shift[l] = 0; }
    a.nsrc = L; a.ksrc = 128; a.ldx = 128; a.x_rows_per_b = T; a.x_row0 = t_off;
    a.W = img + (size_t)L * kLayerImg; a.M = d->Cs; a.K = L * 128;
    a.B = B; a.rows_per_b = T - t_off; a.out = skip; a.out_f32 = 0; a.ldo = d->Cs; a.ob_stride = 0; a.ob_col = 128;
    return launch_cgemm(a, s);
}

size_t wn16_stack_bwd_workspace_bytes(const WnStackDesc* d, int B, int T, int t_off) {
    if (unsupported(d) || B <= 0 || T <= 0 || t_off < 0 || t_off >= T) return 0;
    return carve(d, B, T, t_off, nullptr).bytes;
}

int wn16_stack_bwd(const WnStackDesc* d, const uint16_t* pack, const uint16_t* x, const uint16_t* xs, const uint16_t* z,
                   const uint16_t* dout, const uint16_t* dskip, uint16_t* dx, float* const* dWf, float* const* dWg,
                   float* const* dWp, float* const* dWs, void* ws, size_t ws_bytes, int B, int T, int t_off,
                   int compat_zero_prefix, unsigned flags, void* stream) {
    W16_CHECK_DESC(d);
    WN_CHECK_ARG(pack && x && xs && z && dWf && dWg && dWp && ws && B > 0 && T > 0, "wn16_stack_bwd: bad argument");
    WN_CHECK_ARG(dskip, "wn16_stack_bwd: dskip is NULL (the loss reaches the stack through the skip sum)");
    WN_CHECK_ARG(t_off >= 0 && t_off < T, "wn16_stack_bwd: t_off outside [0,T)");
    WN_CHECK_ARG(ws_bytes >= wn16_stack_bwd_workspace_bytes(d, B, T, t_off), "wn16_stack_bwd: workspace too small");
    hipStream_t s = wn::as_stream(stream);
    const int L = d->n_layers, Cs = d->Cs, Tw = T - t_off;
    const size_t n = (size_t)B * T, nw = (size_t)B * Tw;
    const bf16* img = reinterpret_cast<const bf16*>(pack);
    const bf16* xb = reinterpret_cast<const bf16*>(x);
    const bf16* xsb = reinterpret_cast<const bf16*>(xs);
    const bf16* zb = reinterpret_cast<const bf16*>(z);
    const bf16* dsk = reinterpret_cast<const bf16*>(dskip);
    BwdWs w = carve(d, B, T, t_off, reinterpret_cast<char*>(ws));
    int rc;
    if (dsk) {
        {   // dz_skip of every layer: one GEMM, M = 128 L, each 128-row block is a layer's (B, Tw, 128) slab
            wn::ProfScope prof__("wn16_skip_sum_bwd_dz", stream);
            CG16 a{};
            a.X[0] = dsk; a.shift[0] = 0; a.nsrc = 1; a.ksrc = Cs; a.ldx = Cs; a.x_rows_per_b = Tw; a.x_row0 = 0;
            a.W = img + (size_t)L * kLayerImg + (size_t)Cs * L * 128; a.M = L * 128; a.K = Cs;
            a.B = B; a.rows_per_b = Tw; a.out = w.dzs; a.out_f32 = 0; a.ldo = 128; a.ob_stride = (long long)nw * 128; a.ob_col = 0;
            if ((rc = launch_cgemm(a, s))) return rc;
        }
        if (dWs) {   // dWs_l[cs][cd] += sum dskip[t][cs] z_l[t_off + t][cd]: problems = (256 rows of cs) x (a pair of layers)
            wn::ProfScope prof__("wn16_skip_sum_bwd_dw", stream);
            WG16 a{};
            a.lda = Cs; a.ldb = 128; a.nB = B; a.R = Tw; a.a_rpb = Tw; a.a_r0 = 0; a.b_rpb = T; a.b_r0 = t_off;
            a.os_m = 128; a.os_n = 1; a.relu_b = 0; a.part = w.wgparts;
            int np = 0;
            for (int q = 0; q < L / 2; ++q)                   // the problems that share a layer pair's z are neighbours
                for (int mb = 0; mb < Cs / 256; ++mb) {
                    WG16Prob& p = a.prob[np++];
                    p.A = dsk + 256 * mb;
                    for (int nh = 0; nh < 2; ++nh) {
                        const int l = 2 * q + nh;
                        p.Bh[nh] = zb + (size_t)l * n * 128; p.shift[nh] = 0;
                        for (int mh = 0; mh < 2; ++mh) p.out[mh][nh] = dWs[l] ? dWs[l] + (size_t)(256 * mb + 128 * mh) * 128 : nullptr;
                    }
                    if (np == kMaxProb16) { if ((rc = launch_wgrad16(a, np, s))) return rc; np = 0; }
                }
            if (np && (rc = launch_wgrad16(a, np, s))) return rc;
        }
    }
    const int nwg = dx_grid(B, T);
    const long long part_stride = (long long)nwg * 128 * 128;
    std::vector<int> Zl(L), live_g(L), live_x(L), zero_g(L);    // per layer: zero prefix, live ranges (GateP / DxP in w16_layer.hip)
    const bf16* gout = reinterpret_cast<const bf16*>(dout);
    std::vector<float*> dWp_eff(L, nullptr);
    if (gout) {
        // off the training path (train_audio/train.py:72 discards the stack's residual output): the top layer's dWp would
        // need its own contraction, every other dWp comes out of the dx kernel of the layer above
        wn::set_error("wn16_stack_bwd: a gradient through the stack's residual output is not supported in bf16 storage");
        return WN_ESHAPE;
    }
    {
        wn::ProfScope prof__("wn16_layer_bwd", stream);
        // Dead columns (as in the fp32 stack, stack.hip): the loss reaches the stack through skip[t_off:] only, so layer l
        // receives gradient at columns t >= t_off - (reach of the layers above it) and nowhere else.  Tiles wholly below that
        // load and compute nothing; they store the zeros their readers (the dx kernel, the deferred weight-gradient launch,
        // the layer below) expect.  12.5 % of the sample-layers at config 5's window.
        int t_live = t_off;
        for (int l = L - 1; l >= 0; --l) {
            const int dl = d->dilation[l];
            Zl[l] = compat_zero_prefix ? zero_prefix(T, dl, 2) : 0;
            live_g[l] = (t_live / 32) * 32;
            zero_g[l] = (t_live / 64) * 64;                   // the deferred weight-gradient launch reads [da | dg] from here on (64-row chunks)
            t_live = t_live - dl > 0 ? t_live - dl : 0;       // the layer below (and this layer's dx): one more dilation of reach
            live_x[l] = (t_live / 32) * 32;
        }
        // on request, layers L - 2 .. 1 in ONE launch (k16_bwd_multi: per-tile dataflow words instead of eighty launch
        // boundaries; the top layer (no dout) and layer 0 (no layer below, caller's dx) keep their own launches).  Opt-in:
        // bit-identical and, measured, no faster than the per-layer launches
        const bool multi = (flags & WN_EXEC_BF16_MULTI_LAYER_BWD) && L >= 4 && bwd_multi_ok(B, T);
        for (int l = L - 1; l >= 0; --l) {
            if (multi && l == L - 2) {
                if ((rc = bwd_multi(xb, xsb, zb, img, w.dzs, w.dadg, w.dxb[0], w.dxb[1], w.parts, part_stride, w.sync,
                                    d->dilation, Zl.data(), live_g.data(), live_x.data(), zero_g.data(), L - 2, 1, B, T, t_off,
                                    s)))
                    return rc;
                for (int k = L - 3; k >= 0; --k) dWp_eff[k] = dWp[k];
                gout = w.dxb[1];                            // dx of layer 1
                l = 1;
                continue;
            }
            const bf16* in = l == 0 ? xb : xsb + (size_t)(l - 1) * n * 128;
            const int dl = d->dilation[l];
            bf16* dadg = w.dadg + (size_t)l * n * 256;
            if ((rc = gate_bwd_layer(in, img + (size_t)l * kLayerImg, gout, dsk ? w.dzs + (size_t)l * nw * 128 : nullptr,
                                     t_off, dadg, B, T, dl, Zl[l], live_g[l], zero_g[l], s)))
                return rc;
            bf16* gin = l == 0 ? reinterpret_cast<bf16*>(dx) : w.dxb[l & 1];
            if (!gin) break;                               // l == 0 and the caller does not want dx
            // dx_l[t] = dout[t] + [Wf1;Wg1]^T dab[t] + [Wf0;Wg0]^T dab[t + d]; it is the dout of layer l - 1, whose
            // projection gradient dWp_{l-1} += dx_l z_{l-1}^T is taken in the same pass
            const bf16* zprev = l > 0 ? zb + (size_t)(l - 1) * n * 128 : nullptr;
            if ((rc = dx_layer(dadg, img + (size_t)l * kLayerImg, gout, zprev, gin,
                               l > 0 ? w.parts + (size_t)(l - 1) * part_stride : nullptr, B, T, dl, live_x[l], live_g[l],
                               l == 0 ? 1 : 0 /* the embedding backward reads every row of layer 0's dx */, s)))
                return rc;
            if (l > 0) dWp_eff[l - 1] = dWp[l - 1];
            gout = gin;
        }
    }
    {   // conv weight gradients of every layer in one launch: A = [da | dg] (256), B = (x[t - d], x[t]) -> taps 0, 1
        wn::ProfScope prof__("wn16_conv_wgrad", stream);
        WG16 a{};
        a.lda = 256; a.ldb = 128; a.nB = B; a.R = T; a.a_rpb = T; a.a_r0 = 0; a.b_rpb = T; a.b_r0 = 0;
        a.os_m = 256; a.os_n = 2; a.relu_b = 0; a.part = w.wgparts;
        for (int l = 0; l < L; ++l) {
            WG16Prob& p = a.prob[l];
            const bf16* in = l == 0 ? xb : xsb + (size_t)(l - 1) * n * 128;
            p.A = w.dadg + (size_t)l * n * 256;
            p.r_lo = zero_g[l];                               // [da | dg] is zero (and unwritten) below: not read
            for (int nh = 0; nh < 2; ++nh) {
                p.Bh[nh] = in; p.shift[nh] = nh == 0 ? -d->dilation[l] : 0;
                p.out[0][nh] = dWf[l] + nh;               // W[o][c][k]: (o * 128 + c) * 2 + k
                p.out[1][nh] = dWg[l] + nh;
            }
        }
        if ((rc = launch_wgrad16(a, L, s))) return rc;
    }
    return reduce_parts(w.parts, part_stride, nwg, 128 * 128, dWp_eff.data(), L, s);
}

// ---- head 1x1 convolutions (wavenet.py:584-593): out = W relu(x) + b ------------------------------------------------
int wn16_pack_pointwise(const float* W, uint16_t* Wb, uint16_t* WbT, int Cout, int Cin, void* stream) {
    WN_CHECK_ARG(W && Wb && WbT && Cout > 0 && Cin > 0, "wn16_pack_pointwise: bad argument");
    WN_CHECK_SHAPE(Cout % 8 == 0 && Cin % 8 == 0, "wn16_pack_pointwise: channel counts must be multiples of 8");
    hipStream_t s = wn::as_stream(stream);
    int rc = pack_mat(1, &W, reinterpret_cast<bf16*>(Wb), Cout, Cin, 0, s);
    if (rc) return rc;
    return pack_mat(1, &W, reinterpret_cast<bf16*>(WbT), Cout, Cin, 1, s);
}

int wn16_pointwise_fwd(const uint16_t* x, const uint16_t* Wb, const float* bias, void* out, int out_f32, int64_t N,
                       int Cin, int Cout, int act, void* stream) {
    wn::ProfScope prof__("wn16_pointwise_fwd", stream);
    WN_CHECK_ARG(x && Wb && out && N > 0, "wn16_pointwise_fwd: bad argument");
    WN_CHECK_SHAPE(Cin % 128 == 0 && Cout % 128 == 0 && Cin / 128 <= kMaxSrc16, "wn16_pointwise_fwd: channels must be multiples of 128");
    WN_CHECK_SHAPE(act == WN_ACT_NONE || act == WN_ACT_RELU, "wn16_pointwise_fwd: relu or none");
    WN_CHECK_SHAPE(N < (1ll << 31), "wn16_pointwise_fwd: too many rows");
    CG16 a{};
    a.X[0] = reinterpret_cast<const bf16*>(x); a.shift[0] = 0; a.nsrc = 1; a.ksrc = Cin; a.ldx = Cin;
    a.x_rows_per_b = (int)N; a.x_row0 = 0;
    a.W = reinterpret_cast<const bf16*>(Wb); a.M = Cout; a.K = Cin; a.B = 1; a.rows_per_b = (int)N;
    a.out = out; a.out_f32 = out_f32 ? 1 : 0; a.ldo = Cout; a.ob_stride = 0; a.ob_col = 128; a.bias = bias;
    a.relu_x = act == WN_ACT_RELU ? 1 : 0;
    return launch_cgemm(a, wn::as_stream(stream));
}

// dx = act'(x) (W^T dout) (bf16, may be NULL), dW += dout^T act(x), dbias += sum dout.  dout arrives either as bf16
// (dout) or as fp32 (dout_f32, e.g. d loss / d logits): the latter is rounded into dout_scratch (N x Cout bf16) first.
size_t wn16_pointwise_bwd_workspace_bytes(int64_t N, int Cout) {
    // the weight-gradient contraction's per-workgroup blocks + the bias gradient's per-chunk column sums (2,048 chunks at most)
    return kWgPartBytes + (size_t)2048 * (Cout > 512 ? Cout : 512) * sizeof(float) + 4096 + 256;
}

int wn16_pointwise_bwd(const uint16_t* x, const uint16_t* WbT, const uint16_t* dout, const float* dout_f32,
                       uint16_t* dout_scratch, uint16_t* dx, float* dW, float* dbias, int64_t N, int Cin, int Cout,
                       int act, void* ws, size_t ws_bytes, void* stream) {
    wn::ProfScope prof__("wn16_pointwise_bwd", stream);
    WN_CHECK_ARG(!ws || ws_bytes >= wn16_pointwise_bwd_workspace_bytes(N, Cout), "wn16_pointwise_bwd: workspace too small");
    WN_CHECK_ARG(x && WbT && (dout || (dout_f32 && dout_scratch)) && N > 0, "wn16_pointwise_bwd: bad argument");
    WN_CHECK_SHAPE(Cin % 256 == 0 && Cout % 256 == 0 && Cout / 128 <= kMaxSrc16, "wn16_pointwise_bwd: channels must be multiples of 256");
    WN_CHECK_SHAPE(act == WN_ACT_NONE || act == WN_ACT_RELU, "wn16_pointwise_bwd: relu or none");
    WN_CHECK_SHAPE(N < (1ll << 31) && N * Cout % 8 == 0, "wn16_pointwise_bwd: row count");
    hipStream_t s = wn::as_stream(stream);
    int rc;
    const bf16* g = reinterpret_cast<const bf16*>(dout);
    if (!g) {
        if ((rc = cvt_f2b(dout_f32, reinterpret_cast<bf16*>(dout_scratch), N * Cout, s))) return rc;
        g = reinterpret_cast<const bf16*>(dout_scratch);
    }
    const bf16* xb = reinterpret_cast<const bf16*>(x);
    if (dx) {
        CG16 a{};
        a.X[0] = g; a.shift[0] = 0; a.nsrc = 1; a.ksrc = Cout; a.ldx = Cout; a.x_rows_per_b = (int)N; a.x_row0 = 0;
        a.W = reinterpret_cast<const bf16*>(WbT); a.M = Cin; a.K = Cout; a.B = 1; a.rows_per_b = (int)N;
        a.out = dx; a.out_f32 = 0; a.ldo = Cin; a.ob_stride = 0; a.ob_col = 128;
        if (act == WN_ACT_RELU) { a.ep = 2; a.extra = xb; a.lde = Cin; }
        if ((rc = launch_cgemm(a, s))) return rc;
    }
    if (dW) {
        WG16 a{};
        a.lda = Cout; a.ldb = Cin; a.nB = 1; a.R = (int)N; a.a_rpb = (int)N; a.a_r0 = 0; a.b_rpb = (int)N; a.b_r0 = 0;
        a.os_m = Cin; a.os_n = 1; a.relu_b = act == WN_ACT_RELU ? 1 : 0;
        a.part = reinterpret_cast<float*>(ws);              // NULL: float atomics (the order of the slabs is then not defined)
        int np = 0;
        for (int mb = 0; mb < Cout / 256; ++mb)
            for (int nb = 0; nb < Cin / 256; ++nb) {
                WG16Prob& p = a.prob[np++];
                p.A = g + 256 * mb;
                for (int nh = 0; nh < 2; ++nh) {
                    p.Bh[nh] = xb + 256 * nb + 128 * nh; p.shift[nh] = 0;
                    for (int mh = 0; mh < 2; ++mh) p.out[mh][nh] = dW + (size_t)(256 * mb + 128 * mh) * Cin + 256 * nb + 128 * nh;
                }
                if (np == kMaxProb16) { if ((rc = launch_wgrad16(a, np, s))) return rc; np = 0; }
            }
        if (np && (rc = launch_wgrad16(a, np, s))) return rc;
    }
    if (dbias) {
        if (!dout_f32) { wn::set_error("wn16_pointwise_bwd: the bias gradient is summed from the fp32 gradient"); return WN_EARG; }
        // with a workspace the column sums leave per-chunk partials that one kernel adds in a fixed order (no atomics)
        WnExec ex{WN_GEMM_BF16, 0u, ws ? reinterpret_cast<char*>(ws) + kWgPartBytes : nullptr,
                  ws ? ws_bytes - kWgPartBytes : (size_t)0, 0, 0};
        wn::ExecScope scope__(ws ? &ex : nullptr);
        if ((rc = wn::generic_colsum(dout_f32, 1, (int)N, 0, Cout, Cout, dbias, s))) return rc;
    }
    return WN_OK;
}

}  // extern "C"
