// Whole residual stack in one call (WaveNet.forward_residual_block, wavenet.py:572-582, and its
// backward): the per-layer loop of the reference runs here, in C++, on one stream, so the host
// enqueues a 40-layer stack with two calls instead of ~250.  Pure orchestration: every kernel is
// reached through the per-op entry points of this library.
#include <vector>

#include "wn_kernels.hpp"

namespace wn {
static int zero_prefix(int T, int d, int fw) {           // wavenet.py:303-340
    if (d == 1) return 0;
    int pad = ((-T) % d + d) % d;
    int height = (T + pad) / d;
    if (height < fw) pad += (fw - height) * d;
    int z = (fw - 1) * d - pad;
    return z > 0 ? z : 0;
}
// every layer on the fused 32-channel kernels and no conv / projection bias: the stack backward takes the chained path, which
// reads z and sigmoid only (tanh = z / sigmoid)
static bool chain_capable(const WnStackDesc* d) {
    for (int l = 0; l < d->n_layers; ++l)
        if (!layer_fast_path(d->Cr, d->cd[l], d->fw) || (d->bf && d->bf[l]) || (d->bg && d->bg[l]) || (d->bp && d->bp[l]))
            return false;
    return true;
}
static int check_desc(const WnStackDesc* d) {
    WN_CHECK_ARG(d, "stack: desc is NULL");
    WN_CHECK_ARG(d->n_layers > 0 && d->Cr > 0 && d->Cs > 0 && d->fw > 0, "stack: non-positive size");
    WN_CHECK_ARG(d->cd && d->dilation && d->Wf && d->Wg && d->Wp && d->Ws, "stack: NULL table");
    return WN_OK;
}
}  // namespace wn

using namespace wn;

extern "C" {

int wn_stack_saves_tanh(const WnStackDesc* d, const WnExec* ex) {
    wn::ExecScope exec__(ex);
    return (check_desc(d) == WN_OK && chain_capable(d)) ? 0 : 1;
}

size_t wn_stack_bwd_workspace_bytes(const WnStackDesc* d, int B, int T) {
    if (!d || B <= 0 || T <= 0) return 0;
    size_t n = (size_t)B * T, tot = 0;
    int maxcd = 0;
    for (int l = 0; l < d->n_layers; ++l) { tot += n * d->cd[l]; if (d->cd[l] > maxcd) maxcd = d->cd[l]; }
    size_t lw = 0;                  // per-layer scratch: (da, dg) + the MFMA path's partial weight-gradient tiles
    for (int l = 0; l < d->n_layers; ++l) {
        size_t w = wn_layer_bwd_workspace_floats(B, T, d->Cr, d->cd[l], d->fw);
        if (w > lw) lw = w;
    }
    if (lw < n * 2 * maxcd) lw = n * 2 * maxcd;
    tot += lw;
    tot += 2 * n * d->Cr;           // ping-pong gradient of the residual stream
    bool fast = true;               // chained MFMA path: every layer keeps its partial weight-gradient tiles until the end
    for (int l = 0; l < d->n_layers && fast; ++l) fast = wn_layer_fast_path(d->Cr, d->cd[l], d->fw) != 0;
    // + the multi-layer launch's dataflow words (rounded to 256 B) and its third (V, U) pair
    if (fast) tot += (size_t)d->n_layers * mfma_chain_part_floats() + ((mfma_chain_multi_sync_words(B, T) + 63) / 64) * 64 + 2 * n * d->Cr;
    return tot * sizeof(float);
}

int wn_stack_fwd(const WnStackDesc* d, const float* x, float* xs, float* z, float* f, float* g, float* skip,
                 int B, int T, int t_off, int compat_zero_prefix, int window_only, const WnExec* ex, void* stream) {
    wn::ExecScope exec__(ex);
    int rc = check_desc(d);
    if (rc) return rc;
    WN_CHECK_ARG(x && xs && z && B > 0 && T > 0, "wn_stack_fwd: bad argument");
    WN_CHECK_ARG(!f || g, "wn_stack_fwd: f without g");
    const bool g_only = g && !f;
    WN_CHECK_ARG(!g_only || chain_capable(d), "wn_stack_fwd: this stack's backward needs tanh saved (wn_stack_saves_tanh)");
    WN_CHECK_ARG(t_off >= 0 && t_off < T, "wn_stack_fwd: t_off outside [0,T)");
    const size_t n = (size_t)B * T;
    const int L = d->n_layers;
    std::vector<const float*> zp(L);
    size_t zoff = 0;
    const float* in = x;
    // window_only: only what skip[t_off:] depends on is computed.  Layer l's columns below
    //     live[l] = 32 * floor((live[l+1] - (fw-1) d_{l+1}) / 32),   live[L-1] = 32 * floor(t_off / 32)
    // cannot reach the window through the layers above (and the backward never reads them: its own live ranges start
    // at or above these); xs, z, f, g are left untouched there.  Only for stacks that run on the fused MFMA kernels.
    std::vector<int> live(L, 0);
    if (window_only && skip && t_off > 0) {
        // the same condition under which wn_stack_bwd takes the chained path (which honours these ranges): fused
        // layers without conv / projection biases
        bool fast = true;
        for (int l = 0; l < L && fast; ++l)
            fast = layer_fast_path(d->Cr, d->cd[l], d->fw) && !(d->bf && d->bf[l]) && !(d->bg && d->bg[l]) &&
                   !(d->bp && d->bp[l]);
        if (fast) {
            live[L - 1] = (t_off / 32) * 32;
            for (int l = L - 2; l >= 0; --l) {
                int v = live[l + 1] - (d->fw - 1) * d->dilation[l + 1];
                live[l] = v > 0 ? (v / 32) * 32 : 0;
            }
        }
    }
    // fp16 x 2 split products (WN_GEMM_FP16X2): 32-channel layers without conv / projection biases run the f16-MFMA form
    // of the fused kernel on weight images split once for the whole stack (into the call's scratch: the skip
    // contraction re-uses it after the last layer)
    const void* h2img = nullptr;
    if (gemm_mode() == WN_GEMM_FP16X2 && L <= 64 && d->Cr == 32 && d->fw == 2 &&
        exec_has_scratch(mfma_layer_h2_image_bytes(L))) {
        bool ok = true;
        for (int l = 0; l < L && ok; ++l)
            ok = d->cd[l] == 32 && !(d->bf && d->bf[l]) && !(d->bg && d->bg[l]) && !(d->bp && d->bp[l]);
        if (ok) {
            // (a READY step plan built the images at the start of the step: plan.hip)
            h2img = plan_layer_h2_images(L, d->Wf, d->Wg, d->Wp);
            if (!h2img) {
                void* img = exec_scratch(mfma_layer_h2_image_bytes(L), "the fp16 x 2 layer weight images");
                if ((rc = mfma_layer_pack_h2(L, d->Wf, d->Wg, d->Wp, img, as_stream(stream)))) return rc;
                h2img = img;
            }
        }
    }
    {
        wn::ProfGroup prof_layers__("wn_layer_fwd", stream);     // one bracket around the L launches
        for (int l = 0; l < L; ++l) {
            float* out = xs + (size_t)l * n * d->Cr;
            int Z = compat_zero_prefix ? zero_prefix(T, d->dilation[l], d->fw) : 0;
            // consecutive layers whose dilations add up to <= 32 (d = 1 .. 16 of every block): ONE launch, the inner layers'
            // inputs never leave the chip (k_layer_fwd_h2_grp)
            // (fp16 x 2 only.  The same group on exact fp32 MFMA -- k_layer_fwd_f32_grp, round 6 -- was built, bit-identical to
            // the per-layer launches, and 3.5 % SLOWER on the bf16x3 step (3.844 against 3.712 ms, same box): 80 fp32 MFMAs per
            // tile-layer are MFMA-bound at the per-layer kernel's 16 waves per CU already, and the halo tile adds 12.5 %)
            int ng = (h2img && !exec_flag(WN_EXEC_NO_FWD_GROUPS) && mfma_layer_fwd_h2_ok(B, T, 0))
                         ? mfma_layer_fwd_group_len(d->dilation, l, L) : 0;
            for (int k = 0; k < ng; ++k)
                if (live[l + k] > 0) ng = 0;
            if (ng >= 2) {
                float* outs[8]; float* zs[8]; float* fsv[8]; float* gsv[8]; int Zs[8];
                size_t zo = zoff;
                for (int k = 0; k < ng; ++k) {
                    outs[k] = xs + (size_t)(l + k) * n * d->Cr;
                    zs[k] = z + zo; fsv[k] = f ? f + zo : nullptr; gsv[k] = g ? g + zo : nullptr;
                    Zs[k] = compat_zero_prefix ? zero_prefix(T, d->dilation[l + k], d->fw) : 0;
                    zp[l + k] = z + zo;
                    zo += n * d->cd[l + k];
                }
                {
                    wn::ProfScope prof__("wn_layer_fwd", stream);
                    rc = mfma_layer_fwd_h2_group(in, h2img, l, ng, outs, zs, f ? fsv : nullptr, g ? gsv : nullptr,
                                                 d->dilation + l, Zs, B, T, as_stream(stream));
                }
                if (rc) return rc;
                zoff = zo;
                in = outs[ng - 1];
                l += ng - 1;
                continue;
            }
            if (h2img && mfma_layer_fwd_h2_ok(B, T, live[l])) {
                wn::ProfScope prof__("wn_layer_fwd", stream);
                rc = mfma_layer_fwd_h2(in, h2img, l, out, z + zoff, f ? f + zoff : nullptr, g ? g + zoff : nullptr, B, T,
                                       d->dilation[l], Z, live[l], as_stream(stream));
            } else if (live[l] > 0 || g_only) {
                wn::ProfScope prof__("wn_layer_fwd", stream);
                rc = mfma_layer_fwd(in, d->Wf[l], d->bf ? d->bf[l] : nullptr, d->Wg[l], d->bg ? d->bg[l] : nullptr,
                                    d->Wp[l], d->bp ? d->bp[l] : nullptr, out, z + zoff, f ? f + zoff : nullptr,
                                    g ? g + zoff : nullptr, B, T, d->dilation[l], Z, live[l], as_stream(stream));
            } else {
                rc = wn_layer_fwd(in, d->Wf[l], d->bf ? d->bf[l] : nullptr, d->Wg[l], d->bg ? d->bg[l] : nullptr, d->Wp[l],
                                  d->bp ? d->bp[l] : nullptr, out, z + zoff, f ? f + zoff : nullptr, g ? g + zoff : nullptr,
                                  B, T, d->Cr, d->cd[l], d->fw, d->dilation[l], Z, ex, stream);
            }
            if (rc) return rc;
            zp[l] = z + zoff;
            zoff += n * d->cd[l];
            in = out;
        }
    }
    if (skip)
        return wn_skip_sum_fwd(L, zp.data(), d->Ws, d->bs, d->cd, skip, B, T, t_off, T - t_off, d->Cs, 0, ex, stream);
    return WN_OK;
}

int wn_stack_bwd(const WnStackDesc* d, const float* x, const float* xs, const float* z, const float* f,
                 const float* g, const float* dout, const float* dskip, float* dx,
                 float* const* dWf, float* const* dbf, float* const* dWg, float* const* dbg, float* const* dWp,
                 float* const* dbp, float* const* dWs, float* const* dbs, float* ws, size_t ws_bytes, int B, int T,
                 int t_off, int compat_zero_prefix, const WnExec* ex, void* stream) {
    wn::ExecScope exec__(ex);
    int rc = check_desc(d);
    if (rc) return rc;
    WN_CHECK_ARG(x && xs && z && g && ws && dWf && dWg && dWp, "wn_stack_bwd: NULL argument");
    // ---- chained path: every layer on the MFMA kernels and no conv / projection bias GRADIENTS asked for.  It is the only
    // path that recovers tanh from z / sigmoid, so f may be NULL exactly when it is taken (a desc without biases but with
    // non-NULL dbf / dbg / dbp tables takes the per-layer path, which reads f).
    bool chain = true;
    for (int l = 0; l < d->n_layers && chain; ++l)
        chain = layer_fast_path(d->Cr, d->cd[l], d->fw) && !(dbf && dbf[l]) && !(dbg && dbg[l]) && !(dbp && dbp[l]);
    WN_CHECK_ARG(f || (chain && chain_capable(d)),
                 "wn_stack_bwd: this stack's backward needs tanh saved (wn_stack_saves_tanh; bias-gradient tables select the "
                 "per-layer path, which reads f)");
    WN_CHECK_ARG(dout || dskip, "wn_stack_bwd: no incoming gradient");
    WN_CHECK_ARG(ws_bytes >= wn_stack_bwd_workspace_bytes(d, B, T), "wn_stack_bwd: workspace too small");
    const size_t n = (size_t)B * T;
    const int L = d->n_layers;
    int maxcd = 0;
    std::vector<const float*> zp(L);
    std::vector<float*> dzp(L);
    std::vector<size_t> off(L);
    size_t zoff = 0;
    for (int l = 0; l < L; ++l) { off[l] = zoff; zp[l] = z + zoff; dzp[l] = ws + zoff; zoff += n * d->cd[l]; if (d->cd[l] > maxcd) maxcd = d->cd[l]; }
    float* dab = ws + zoff;
    size_t lw = n * 2 * maxcd;
    for (int l = 0; l < L; ++l) {
        size_t w = wn_layer_bwd_workspace_floats(B, T, d->Cr, d->cd[l], d->fw);
        if (w > lw) lw = w;
    }
    float* gbuf[2] = {dab + lw, dab + lw + n * d->Cr};
    const int Tw = T - t_off;
    if (dskip) {
        if (chain && d->Cs % 32 == 0) {
            // the chained layer kernels take dz as 0 below t_off and never read it there: only the loss window is computed
            wn::ProfScope prof__("wn_skip_sum_bwd_dz", stream);
            rc = mfma_skip_bwd_dz(L, d->Ws, d->cd, dskip, dzp.data(), B, T, t_off, Tw, d->Cs, true, as_stream(stream));
        } else {
            rc = wn_skip_sum_bwd_dz(L, d->Ws, d->cd, dskip, dzp.data(), B, T, t_off, Tw, d->Cs, ex, stream);
        }
        if (rc) return rc;
        if (dWs) {
            rc = wn_skip_sum_bwd_dw(L, zp.data(), d->cd, dskip, dWs, dbs, B, T, t_off, Tw, d->Cs, ex, stream);
            if (rc) return rc;
        }
    }
    if (chain) {
        // the (da,dg) scratch is not needed: its room (past the per-layer kernel's own partial-tile area) and the two
        // ping-pong buffers together hold the split gradient (V, U) of two consecutive layers; every layer's partial
        // weight-gradient tiles go to their own slot behind them
        float* parts = gbuf[1] + n * d->Cr;          // L x mfma_chain_part_floats(): summed once, after the last layer
        float* vu = dab + mfma_layer_bwd_extra_ws_floats();
        std::vector<float*> dWp_eff(L);
        std::vector<int> nwg(L);
        // three (V, U) pairs, layer l writes pair l % 3 and reads pair (l + 1) % 3: with two, a layer rewrites the rows the layer
        // above it READ, and the multi-layer launch would have to hold a tile back until the readers of the previous layer
        // are through; with three the readers are two layers back
        unsigned* sync = reinterpret_cast<unsigned*>(parts + (size_t)L * mfma_chain_part_floats());
        float* vu3 = reinterpret_cast<float*>(sync) + ((mfma_chain_multi_sync_words(B, T) + 63) / 64) * 64;
        // a READY step plan owns the dataflow words and zeroed them at the start of the step (no k_chain_zero_sync launch)
        bool sync_zeroed = false;
        if (unsigned* ps = plan_sync_words((int)mfma_chain_multi_sync_words(B, T))) { sync = ps; sync_zeroed = true; }
        float* Vb[3] = {vu, vu + n * d->Cr, vu3};
        float* Ub[3] = {vu + 2 * n * d->Cr, vu + 3 * n * d->Cr, vu3 + n * d->Cr};
        const float* Vin = dout;
        const float* Uin = nullptr;
        int dU = 0;
        // Dead columns: with no gradient through the stack's residual output, layer l receives gradient only at columns
        // t >= t_off - (reach of the layers above it) -- everything below is exactly zero and is neither computed nor
        // stored.  t_live is rounded down to a tile; the layer below is told where the written rows start.
        int t_live = dout ? 0 : t_off;               // first column of the current layer that can carry gradient
        int vu_t0 = 0;                               // first written row of Vin / Uin
        wn::ProfScope prof__("wn_layer_bwd", stream);        // one bracket: 40 layer kernels + the tile reduction
        // fp16 x 2, z + sigmoid saved, loss on the skip sum: every layer that has V, U and dz_skip inputs (all but the top
        // one) runs in ONE launch of co-resident workgroups following per-tile dataflow words (k_layer_bwd_chain_multi: no grid
        // barrier); the per-layer loop below
        // then only collects their parameters
        const bool multi = gemm_mode() != WN_GEMM_BF16 && f == nullptr && dskip && d->Cs % 32 == 0 && L >= 3 &&
                           L - 1 <= mfma_chain_multi_max_layers() && !exec_flag(WN_EXEC_NO_MULTI_LAYER_BWD);
        std::vector<int> m_layer, m_d, m_Z, m_live, m_vu_t0, m_dU;
        std::vector<const float*> m_Wf, m_Wg, m_Wp;
        for (int l = L - 1; l >= 0; --l) {
            const float* in = l == 0 ? x : xs + (size_t)(l - 1) * n * d->Cr;
            int Z = compat_zero_prefix ? zero_prefix(T, d->dilation[l], d->fw) : 0;
            const int live = (t_live / 32) * 32;
            if (multi && Vin && Uin) {
                m_layer.push_back(l); m_d.push_back(d->dilation[l]); m_Z.push_back(Z); m_live.push_back(live);
                m_vu_t0.push_back(vu_t0); m_dU.push_back(dU);
                m_Wf.push_back(d->Wf[l]); m_Wg.push_back(d->Wg[l]); m_Wp.push_back(d->Wp[l]);
            } else {
                rc = mfma_layer_bwd_chain(in, f ? f + off[l] : z + off[l], g + off[l], d->Wf[l], d->Wg[l], d->Wp[l], Vin, Uin,
                                          dU, vu_t0, dskip ? dzp[l] : nullptr, (dskip && d->Cs % 32 == 0) ? t_off : 0,
                                          Vb[l % 3], Ub[l % 3], parts + (size_t)l * mfma_chain_part_floats(), B, T,
                                          d->dilation[l], Z, live, &nwg[l], as_stream(stream), f == nullptr);
                if (rc) return rc;
            }
            dWp_eff[l] = (Vin || Uin) ? dWp[l] : nullptr;
            Vin = Vb[l % 3]; Uin = Ub[l % 3]; dU = d->dilation[l];
            vu_t0 = live;
            t_live = t_live - (d->fw - 1) * d->dilation[l];      // the layer below: one more dilation of reach
            if (t_live < 0) t_live = 0;
        }
        if (!m_layer.empty()) {
            int grid = 0;
            rc = mfma_layer_bwd_chain_multi((int)m_layer.size(), m_layer.data(), m_Wf.data(), m_Wg.data(), m_Wp.data(),
                                            m_d.data(), m_Z.data(), m_live.data(), m_vu_t0.data(), m_dU.data(), x, xs, z, g,
                                            ws /* dz of layer l at ws + l n 32 */, Vb, Ub, parts, mfma_chain_part_floats(),
                                            sync, B, T, t_off, &grid, as_stream(stream), sync_zeroed);
            if (rc == WN_ESHAPE) {                  // not every workgroup would be resident on this device: one launch per layer after all
                wn::set_error("");
                for (size_t i = 0; i < m_layer.size(); ++i) {
                    const int l = m_layer[i];
                    const float* in = l == 0 ? x : xs + (size_t)(l - 1) * n * d->Cr;
                    rc = mfma_layer_bwd_chain(in, z + off[l], g + off[l], d->Wf[l], d->Wg[l], d->Wp[l], Vb[(l + 1) % 3],
                                              Ub[(l + 1) % 3], m_dU[i], m_vu_t0[i], dzp[l], t_off, Vb[l % 3], Ub[l % 3],
                                              parts + (size_t)l * mfma_chain_part_floats(), B, T, m_d[i], m_Z[i], m_live[i],
                                              &nwg[l], as_stream(stream), true);
                    if (rc) return rc;
                }
                grid = 0;
            }
            if (rc) return rc;
            if (grid)
                for (int l : m_layer) nwg[l] = grid;
        }
        // (+ dx = V + U[t + dU], the stack's input gradient, in the same launch)
        return mfma_chain_reduce_all(parts, L, nwg.data(), dWf, dWg, dWp_eff.data(), as_stream(stream), Vin, Uin, dx, B, T, dU, vu_t0);
    }
    const float* gout = dout;
    for (int l = L - 1; l >= 0; --l) {
        const float* in = l == 0 ? x : xs + (size_t)(l - 1) * n * d->Cr;
        float* gin = (l == 0) ? dx : gbuf[l & 1];
        int Z = compat_zero_prefix ? zero_prefix(T, d->dilation[l], d->fw) : 0;
        if (wide_layer_in_use(d->Cr, d->cd[l], d->fw)) {      // the stack still holds z = f g: the projection gradient reads it
            wn::ProfScope prof__("wn_layer_bwd", stream);
            rc = wide_layer_bwd(in, f + off[l], g + off[l], d->Wf[l], d->Wg[l], d->Wp[l], gout, dskip ? dzp[l] : nullptr,
                                gin, dWf[l], dbf ? dbf[l] : nullptr, dWg[l], dbg ? dbg[l] : nullptr,
                                gout ? dWp[l] : nullptr, (gout && dbp) ? dbp[l] : nullptr, dab, B, T, d->Cr, d->cd[l],
                                d->fw, d->dilation[l], Z, as_stream(stream), z + off[l]);
        } else
        rc = wn_layer_bwd(in, f + off[l], g + off[l], d->Wf[l], d->Wg[l], d->Wp[l], gout, dskip ? dzp[l] : nullptr, gin,
                          dWf[l], dbf ? dbf[l] : nullptr, dWg[l], dbg ? dbg[l] : nullptr, gout ? dWp[l] : nullptr,
                          (gout && dbp) ? dbp[l] : nullptr, dab, B, T, d->Cr, d->cd[l], d->fw, d->dilation[l], Z, ex, stream);
        if (rc) return rc;
        gout = gin;
        if (!gin) break;          // l == 0 and the caller does not want dx
    }
    return WN_OK;
}

}  // extern "C"
