// Fused residual-layer kernels of the bf16-storage path (BASELINE config 5; ResidualConvLayer.__call__, wavenet.py:358-368,
// and Chainer's backward through it), Cr = Cd = 128, filter width 2.
//
//   k16_fwd       x (bf16) -> out = Wp z + x (bf16), z = tanh(Wf * x) sigmoid(Wg * x) (bf16).  One kernel per layer.
//   k16_gate_bwd  recomputes tanh / sigmoid from x (nothing but z is saved by the forward), dz = Wp^T dout + dz_skip,
//                 [da | dg] = dz (g (1 - f^2) | f g (1 - g)) (bf16), and the projection's weight gradient
//                 dWp += dout z^T (fp32 partial tiles per workgroup).
// Both are weight-stationary: every wave keeps its slice of the layer's weights in registers as MFMA A operands for the
// whole launch (8 waves x 16 gate channels: the filter rows and the gate rows of a channel sit in the SAME 32-row MFMA
// tile, rows r and r + 16, i.e. accumulator registers r and r + 8 of one lane -- the gate needs no data movement), and
// the workgroup streams 256-byte-row time tiles through LDS by LDS-DMA, double buffered, two waves per SIMD so that one
// wave's gate arithmetic runs under the other's MFMAs.  HBM-bound by design: per sample-layer the forward reads 256 B
// and writes 512 B; the gate backward reads 768 B and writes 512 B.
#include "w16_gemm.hpp"
#include "wn_kernels.hpp"

namespace w16 {

using wn::fast_sigmoid;
using wn::fast_tanh;

// ---------------------------------------------------------------------------------------------
// forward.  LDS: xold[2], xcur[2] (64 x 256 B each), z tile, out tile  = 96 KB
// ---------------------------------------------------------------------------------------------
static constexpr int kFT = 64;                         // time columns per tile
static constexpr int kFTileB = kFT * 256;              // 16 KB
static constexpr int kFwdLds = 6 * kFTileB;

__global__ __launch_bounds__(512, 2) void k16_fwd(const bf16* __restrict__ x, const bf16* __restrict__ convA,
                                                   const bf16* __restrict__ projA, bf16* __restrict__ out,
                                                   bf16* __restrict__ z, int B, int T, int d, int Z, int tiles_per_b,
                                                   int ntiles) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    auto xold = [&](int buf) { return lds + buf * kFTileB; };
    auto xcur = [&](int buf) { return lds + (2 + buf) * kFTileB; };
    char* zt = lds + 4 * kFTileB;
    char* ot = lds + 5 * kFTileB;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    int first, stride, last;
    tile_range(ntiles, first, stride, last);
    if (first >= last) return;

    // A operands: conv (16 k-steps: 0..7 tap 0 = x[t-d], 8..15 tap 1 = x[t]) and this wave's projection tile
    bf16x8 cA[16], pA[8];
#pragma unroll
    for (int s = 0; s < 16; ++s) cA[s] = *reinterpret_cast<const bf16x8*>(convA + ((w * 16 + s) * 64 + lane) * 8);
    const int mt = w & 3, nt2 = w >> 2;
#pragma unroll
    for (int s = 0; s < 8; ++s) pA[s] = *reinterpret_cast<const bf16x8*>(projA + ((mt * 8 + s) * 64 + lane) * 8);

    // the operand loads are waited for HERE: left pending, the compiler would place its s_waitcnt vmcnt(0) at their first
    // use inside the loop, where it would also drain the LDS-DMA prefetch of every iteration
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" ::"v"(cA[s]));
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" ::"v"(pA[s]));

    auto issue = [&](int tile, int buf) {
        const int b = tile / tiles_per_b;
        const int t0 = (tile - b * tiles_per_b) * kFT;
        const bf16* xb = x + (long long)b * T * 128;
        dma_pieces(xcur(buf), lane, 2 * w, 1, 2, [&](int r) {
            const int t = t0 + r < T ? t0 + r : T - 1;
            return xb + (long long)t * 128;
        });
        dma_pieces(xold(buf), lane, 2 * w, 1, 2, [&](int r) {
            int t = t0 + r < T ? t0 + r : T - 1;
            t = t - d >= 0 ? t - d : 0;
            return xb + (long long)t * 128;
        });
    };

    issue(first, 0);
    bool full_prev = false;                            // the previous tile's 4 stores were issued unconditionally
    int it = 0;
    for (int tile = first; tile < last; tile += stride, ++it) {
        const int buf = it & 1;
        const int b = tile / tiles_per_b;
        const int t0 = (tile - b * tiles_per_b) * kFT;
        // this tile's 4 DMA pieces are older than the previous tile's stores: leave those in flight
        if (full_prev) wait_vm<4>(); else wait_vm<0>();
        barrier();
        if (tile + stride < last) issue(tile + stride, buf ^ 1);
        if (t0 < d) {
            // rows whose tap-0 sample lies before the clip start read as 0 (wavenet.py:298-301 pads with zeros)
            for (int r = w; r < kFT; r += 8)
                if (t0 + r < d) *reinterpret_cast<unsigned*>(xold(buf) + r * 256 + lane * 4) = 0u;
            barrier();
        }
        // ---- both dilated convolutions for this wave's 16 gate channels, 64 columns ----
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
            const int row = nt * 32 + j;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const bf16x8 bv = frag_row(s < 8 ? xold(buf) : xcur(buf), row, s & 7, h);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cA[s], bv, acc[nt], 0, 0, 0);
            }
        }
        // ---- gate: registers r (filter) and r + 8 (gate) of a lane are the same channel ----
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int row = nt * 32 + j;
            const bool live = t0 + row >= Z;           // the reference's zero prefix: a = g = 0 there, so z = 0
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float zz[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float f = fast_tanh(live ? acc[nt][4 * q + e] : 0.f);
                    const float g = fast_sigmoid(live ? acc[nt][8 + 4 * q + e] : 0.f);
                    zz[e] = f * g;
                }
                *reinterpret_cast<bf16x4*>(zt + toff(row, 2 * w + q) + 8 * h) = pack4(zz[0], zz[1], zz[2], zz[3]);
            }
        }
        barrier();
        // ---- residual projection: out tile (32 channels mt) x (32 columns nt2) per wave, + x ----
        {
            f32x16 ao;
#pragma unroll
            for (int r = 0; r < 16; ++r) ao[r] = 0.f;
            const int row = nt2 * 32 + j;
#pragma unroll
            for (int s = 0; s < 8; ++s)
                ao = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pA[s], frag_row(zt, row, s, h), ao, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = toff(row, 4 * mt + q) + 8 * h;
                const bf16x4 xv = *reinterpret_cast<const bf16x4*>(xcur(buf) + o);
                *reinterpret_cast<bf16x4*>(ot + o) = pack4(ao[4 * q] + (float)xv[0], ao[4 * q + 1] + (float)xv[1],
                                                           ao[4 * q + 2] + (float)xv[2], ao[4 * q + 3] + (float)xv[3]);
            }
        }
        barrier();
        // ---- whole 256-byte rows leave: 4 rows per wave instruction, two pieces of each tile per wave ----
        const bool full = t0 + kFT <= T;
        bf16* zb = z + ((long long)b * T + t0) * 128;
        bf16* ob = out + ((long long)b * T + t0) * 128;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int p = 2 * w + pp;
            const int r = 4 * p + (lane >> 4);
            const int c = (lane & 15) ^ key(r);
            const u32x4 vz = *reinterpret_cast<const u32x4*>(zt + p * 1024 + lane * 16);
            const u32x4 vo = *reinterpret_cast<const u32x4*>(ot + p * 1024 + lane * 16);
            if (full || t0 + r < T) {
                *reinterpret_cast<u32x4*>(zb + r * 128 + c * 8) = vz;
                *reinterpret_cast<u32x4*>(ob + r * 128 + c * 8) = vo;
            }
        }
        full_prev = full;
    }
}

// ---------------------------------------------------------------------------------------------
// gate backward.  32-column tiles.  LDS: xold[2], xcur[2], dout[2], dzs[2] (8 KB each) + da, dg, z tiles = 88 KB
// ---------------------------------------------------------------------------------------------
static constexpr int kGT = 32;
static constexpr int kGTileB = kGT * 256;              // 8 KB
static constexpr int kGateLds = 11 * kGTileB;
static constexpr int kDwpPart = 128 * 128;             // floats per workgroup partial of dWp

template <bool HAS_DO, bool HAS_DZ>
__global__ __launch_bounds__(512, 2) void k16_gate_bwd(
    const bf16* __restrict__ x, const bf16* __restrict__ convA, const bf16* __restrict__ dzA,
    const bf16* __restrict__ dout, const bf16* __restrict__ dzs, int dz_t0, bf16* __restrict__ dadg,
    float* __restrict__ dwp_part, int B, int T, int d, int Z, int tiles_per_b, int ntiles) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    auto xold = [&](int buf) { return lds + buf * kGTileB; };
    auto xcur = [&](int buf) { return lds + (2 + buf) * kGTileB; };
    auto dot = [&](int buf) { return lds + (4 + buf) * kGTileB; };
    auto dzt = [&](int buf) { return lds + (6 + buf) * kGTileB; };
    char* dat = lds + 8 * kGTileB;
    char* dgt = lds + 9 * kGTileB;
    char* zt = lds + 10 * kGTileB;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int Tw = T - dz_t0;
    int first, stride, last;
    tile_range(ntiles, first, stride, last);

    bf16x8 cA[16], zA[8];
#pragma unroll
    for (int s = 0; s < 16; ++s) cA[s] = *reinterpret_cast<const bf16x8*>(convA + ((w * 16 + s) * 64 + lane) * 8);
    if (HAS_DO) {
#pragma unroll
        for (int s = 0; s < 8; ++s) zA[s] = *reinterpret_cast<const bf16x8*>(dzA + ((w * 8 + s) * 64 + lane) * 8);
    }
    // dWp[cr][cd] partial: wave w owns rows cr 32 (w & 3) .. + 31, columns cd 64 (w >> 2) .. + 63
    f32x16 wp[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { wp[0][r] = 0.f; wp[1][r] = 0.f; }

#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" ::"v"(cA[s]));
    if (HAS_DO) {
#pragma unroll
        for (int s = 0; s < 8; ++s) asm volatile("" ::"v"(zA[s]));
    }

    auto issue = [&](int tile, int buf) {
        const int b = tile / tiles_per_b;
        const int t0 = (tile - b * tiles_per_b) * kGT;
        const bf16* xb = x + (long long)b * T * 128;
        dma_pieces(xcur(buf), lane, w, 1, 1, [&](int r) {
            const int t = t0 + r < T ? t0 + r : T - 1;
            return xb + (long long)t * 128;
        });
        dma_pieces(xold(buf), lane, w, 1, 1, [&](int r) {
            int t = t0 + r < T ? t0 + r : T - 1;
            t = t - d >= 0 ? t - d : 0;
            return xb + (long long)t * 128;
        });
        if (HAS_DO) {
            const bf16* db = dout + (long long)b * T * 128;
            dma_pieces(dot(buf), lane, w, 1, 1, [&](int r) {
                const int t = t0 + r < T ? t0 + r : T - 1;
                return db + (long long)t * 128;
            });
        }
        if (HAS_DZ) {
            const bf16* zb = dzs + (long long)b * Tw * 128;       // dz_skip exists for the loss window only
            dma_pieces(dzt(buf), lane, w, 1, 1, [&](int r) {
                int t = t0 + r < T ? t0 + r : T - 1;
                t = t - dz_t0 >= 0 ? t - dz_t0 : 0;
                return zb + (long long)t * 128;
            });
        }
    };
    constexpr int kStores = 2;                          // da and dg pieces per wave and tile

    if (first < last) issue(first, 0);
    bool full_prev = false;
    int it = 0;
    for (int tile = first; tile < last; tile += stride, ++it) {
        const int buf = it & 1;
        const int b = tile / tiles_per_b;
        const int t0 = (tile - b * tiles_per_b) * kGT;
        if (full_prev) wait_vm<kStores>(); else wait_vm<0>();
        barrier();
        if (tile + stride < last) issue(tile + stride, buf ^ 1);
        const bool fix_old = t0 < d;
        const bool fix_do = HAS_DO && t0 + kGT > T;       // rows beyond the clip must not reach dWp
        if (fix_old || fix_do) {
            for (int r = w; r < kGT; r += 8) {
                if (fix_old && t0 + r < d) *reinterpret_cast<unsigned*>(xold(buf) + r * 256 + lane * 4) = 0u;
                if (fix_do && t0 + r >= T) *reinterpret_cast<unsigned*>(dot(buf) + r * 256 + lane * 4) = 0u;
            }
            barrier();
        }
        const int t = t0 + j;
        // ---- recompute the gate pre-activations of this wave's 16 channels ----
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cA[s], frag_row(s < 8 ? xold(buf) : xcur(buf), j, s & 7, h), acc,
                                                          0, 0, 0);
        // ---- dz = Wp^T dout + dz_skip (registers 0..7; rows 16..31 of the A tile are zero) ----
        f32x16 dz;
#pragma unroll
        for (int r = 0; r < 16; ++r) dz[r] = 0.f;
        if (HAS_DZ) {
            if (t0 + kGT > dz_t0) {
                const bool in = t >= dz_t0;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const bf16x4 v = *reinterpret_cast<const bf16x4*>(dzt(buf) + toff(j, 2 * w + q) + 8 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dz[4 * q + e] = in ? (float)v[e] : 0.f;
                }
            }
        }
        if (HAS_DO) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                dz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zA[s], frag_row(dot(buf), j, s, h), dz, 0, 0, 0);
        }
        // ---- gate forward + backward, elementwise ----
        const bool live = t >= Z && t < T;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float da[4], dg[4], zz[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f = fast_tanh(live ? acc[4 * q + e] : 0.f);
                const float g = fast_sigmoid(live ? acc[8 + 4 * q + e] : 0.f);
                const float dzv = live ? dz[4 * q + e] : 0.f;
                da[e] = dzv * g * (1.f - f * f);
                dg[e] = dzv * f * g * (1.f - g);
                zz[e] = f * g;
            }
            const int o = toff(j, 2 * w + q) + 8 * h;
            *reinterpret_cast<bf16x4*>(dat + o) = pack4(da[0], da[1], da[2], da[3]);
            *reinterpret_cast<bf16x4*>(dgt + o) = pack4(dg[0], dg[1], dg[2], dg[3]);
            if (HAS_DO) *reinterpret_cast<bf16x4*>(zt + o) = pack4(zz[0], zz[1], zz[2], zz[3]);
        }
        barrier();
        // ---- dWp += dout z^T over this tile's 32 columns (contraction over time: transposed LDS reads) ----
        if (HAS_DO) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 a = frag_tr(dot(buf), 16 * ks, 32 * (w & 3), lane);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const bf16x8 bz = frag_tr(zt, 16 * ks, 64 * (w >> 2) + 32 * n, lane);
                    wp[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bz, wp[n], 0, 0, 0);
                }
            }
        }
        // ---- [da | dg] rows leave whole: 512-byte rows, da in the first half ----
        const bool full = t0 + kGT <= T;
        {
            const int r = 4 * w + (lane >> 4);
            const int c = (lane & 15) ^ key(r);
            const u32x4 va = *reinterpret_cast<const u32x4*>(dat + w * 1024 + lane * 16);
            const u32x4 vg = *reinterpret_cast<const u32x4*>(dgt + w * 1024 + lane * 16);
            bf16* o = dadg + ((long long)b * T + t0 + r) * 256 + c * 8;
            if (full || t0 + r < T) {
                *reinterpret_cast<u32x4*>(o) = va;
                *reinterpret_cast<u32x4*>(o + 128) = vg;
            }
        }
        full_prev = full;
    }
    if (HAS_DO) {
        // this workgroup's partial of dWp: [cr][cd] fp32, summed over workgroups by k16_reduce_parts
        float* o = dwp_part + (long long)blockIdx.x * kDwpPart;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                o[(32 * (w & 3) + acc_row(r, h)) * 128 + 64 * (w >> 2) + 32 * n + j] = wp[n][r];
    }
}

// dW[e] += sum over workgroups of part[wg][e]  (fixed order: deterministic).  blockIdx.y = layer.
struct ReduceArgs { float* dW[kMaxProb16]; };
__global__ void k16_reduce_parts(const float* __restrict__ part, long long layer_stride, int nwg, int n, ReduceArgs dW) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float* o = dW.dW[blockIdx.y];
    if (!o) return;
    const float* p = part + (long long)blockIdx.y * layer_stride + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = 0;
    for (; g + 4 <= nwg; g += 4) {
        s0 += p[(long long)g * n]; s1 += p[(long long)(g + 1) * n];
        s2 += p[(long long)(g + 2) * n]; s3 += p[(long long)(g + 3) * n];
    }
    for (; g < nwg; ++g) s0 += p[(long long)g * n];
    o[e] += (s0 + s1) + (s2 + s3);
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static int grid_for(int ntiles) {
    int g = ntiles < 256 ? ntiles : 256;                 // one 512-thread workgroup per CU
    if (g >= 8) g &= ~7;
    return g;
}

int fwd_layer(const bf16* x, const bf16* img, bf16* out, bf16* z, int B, int T, int d, int Z, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k16_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kFwdLds));
        attr = true;
    }
    const int tiles_per_b = (T + kFT - 1) / kFT;
    const int ntiles = B * tiles_per_b;
    hipLaunchKernelGGL(k16_fwd, dim3(grid_for(ntiles)), dim3(512), kFwdLds, s, x, img, img + kConvA, out, z, B, T, d, Z,
                       tiles_per_b, ntiles);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int gate_bwd_grid(int B, int T) { return grid_for(B * ((T + kGT - 1) / kGT)); }

int gate_bwd_layer(const bf16* x, const bf16* img, const bf16* dout, const bf16* dzs, int dz_t0, bf16* dadg,
                   float* dwp_part, int B, int T, int d, int Z, hipStream_t s) {
    const int tiles_per_b = (T + kGT - 1) / kGT;
    const int ntiles = B * tiles_per_b;
    const int grid = grid_for(ntiles);
#define GB_LAUNCH(DO, DZ)                                                                                              \
    do {                                                                                                               \
        static bool attr = false;                                                                                      \
        if (!attr) {                                                                                                   \
            WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k16_gate_bwd<DO, DZ>),                            \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kGateLds));                         \
            attr = true;                                                                                               \
        }                                                                                                              \
        hipLaunchKernelGGL((k16_gate_bwd<DO, DZ>), dim3(grid), dim3(512), kGateLds, s, x, img, img + kConvA + kProjA,  \
                           dout, dzs, dz_t0, dadg, dwp_part, B, T, d, Z, tiles_per_b, ntiles);                         \
    } while (0)
    if (dout && dzs) GB_LAUNCH(true, true);
    else if (dout) GB_LAUNCH(true, false);
    else if (dzs) GB_LAUNCH(false, true);
    else { wn::set_error("gate_bwd_layer: no incoming gradient"); return WN_EARG; }
#undef GB_LAUNCH
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int reduce_parts(const float* part, long long layer_stride, int nwg, int n, float* const* dW, int L, hipStream_t s) {
    if (L > kMaxProb16) { wn::set_error("w16: more than %d layers", kMaxProb16); return WN_ESHAPE; }
    ReduceArgs a{};
    for (int l = 0; l < L; ++l) a.dW[l] = dW[l];
    hipLaunchKernelGGL(k16_reduce_parts, dim3((n + 255) / 256, L), dim3(256), 0, s, part, layer_stride, nwg, n, a);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

}  // namespace w16
