// The split of a channel GEMM's weight tiles into MFMA A-operand images (bf16 x 3, bf16 x 1, fp16 x 2): shared by the GEMM
// launchers (mfma_gemm_b3.hip: k_split_w, once per launch) and by the step plan (plan.hip: every image of a training step in
// two launches).
#pragma once
#include "mfma_gemm.hpp"

namespace wn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

static constexpr int kTileElems = 2 * 3 * 64 * 8;          // bf16 elements of one tile image: [ks][comp][lane][8]
static constexpr int kTileBytes = kTileElems * 2;          // 6144


// fp16 two-way split (WN_GEMM_FP16X2): x S = h + m with h, m in fp16 (11 significant bits each, |x S - h - m| <= 2^-22 |x S|
// while m is a normal number), products wh xh + (wh xm + wm xh): the dropped wm xm term is 2^-22 relative, so the result
// is fp32-accurate like the six-term bf16 split at HALF the matrix instructions.  fp16's narrow exponent range is what
// restricts it: operands are scaled by a power of two (exact) and must stay below 65504 after scaling, which is known for
// the forward contractions (z = tanh * sigmoid in [-1, 1]; weights; the skip sum in front of the head) and is not for
// gradients, whose magnitude follows the batch size and any loss scaling -- those keep the bf16 split.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
static constexpr float kH2ScaleW = 256.f;        // weights without a measured range (never used by the launchers below)
static constexpr float kH2ScaleX = 16384.f;      // z = tanh * sigmoid in [-1, 1] (the only operand with a static range): 2^14,
                                                 // so that values down to ~1e-8 keep their two parts (with 2^4 a residual
                                                 // stream 4,096 times smaller than usual lost them: 1e-3 relative in the skip sum)
// power-of-two scale that brings max |x| just below 2^14 (mx_dev = bits of max |x|); `fixed` when the range is static
__device__ __forceinline__ float h2_scale_of(float m) {
    if (!(m > 0.f) || !(m < 3e38f)) return 1.f;
    int e;
    (void)frexpf(m, &e);                                  // m = f 2^e, f in [0.5, 1)
    return ldexpf(1.f, 14 - e);
}
__device__ __forceinline__ float h2_scale(const unsigned* mx_dev, float fixed) {
    if (!mx_dev) return fixed;
    return h2_scale_of(__uint_as_float(*mx_dev));
}
__device__ __forceinline__ void split2h(float x, _Float16& h, _Float16& m) {
    x = fminf(fmaxf(x, -65000.f), 65000.f);
    h = (_Float16)x;
    m = (_Float16)(x - (float)h);
}

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// ---- weight image: tile (chunk c, m-tile t) at img + (c*mtiles + t)*kTileElems -----------------------------
//   element [ks][comp][lane = i + 32*hh][j]  =  comp-part of  W_tile[i][16*ks + 8*hh + j]
// mode 0: chunk c = (source, 32-wide k slice), m-tile t = rows 32t.. of that source's W[m][k]
// mode 2: m-tile t = problem t (32 rows), chunk c = k slice of W[t]
// one != 0 (one-term products): only the h parts are stored, 2 KB per tile instead of 6
// WMAX: only the largest |w| of the launch's weight tiles, into *a.wmax_dev (atomicMax of the bits: a positive float orders
// like an unsigned) -- the fp16 split scales the weights by the power of two that brings that maximum just below 2^14
// A: anything with the fields W[], W2[], wsm[], wsk (CGArgs; the step plan's SplitJob).  WMAX: returns (thread 0 only) the
// largest |w| of the tile and writes nothing; else splits the tile into img -- the fp16 form (one == 3) scales by the power of
// two h2sw that the caller derived from the launch's maximum (h2_scale / h2_scale_of).
template <bool WMAX, class A>
__device__ __forceinline__ float split_w_tile(const A& a, int mode, int mtiles, int chunks_per_src, __bf16* __restrict__ img, int one,
                                              int tile, float h2sw) {
    // one: 1 = h parts only (one-term bf16), 3 = fp16 two-way split of W * h2sw, 0 = bf16 three-way split
    const int c = tile / mtiles, t = tile - c * mtiles;
    const int i = threadIdx.x & 31, c4 = threadIdx.x >> 5;           // row, group of 4 consecutive k
    const float* W;
    int wsm, k0;
    if (mode == 0 || mode == 4 || mode == 6) {
        const int src = c / chunks_per_src;
        k0 = (c - src * chunks_per_src) * 32;
        W = a.W[src] + (long long)(t * 32) * a.wsm[src];
        wsm = a.wsm[src];
    } else if (mode == 3 || mode == 5) {   // m-tile 2i = filter rows 32i.., m-tile 2i+1 = gate rows 32i..
        const int src = c / chunks_per_src;
        k0 = (c - src * chunks_per_src) * 32;
        W = ((t & 1) ? a.W2[src] : a.W[src]) + (long long)((t >> 1) * 32) * a.wsm[src];
        wsm = a.wsm[src];
    } else {
        k0 = c * 32;
        W = a.W[t];
        wsm = a.wsm[t];
    }
    const float* wp = W + (long long)i * wsm + (long long)(k0 + 4 * c4) * a.wsk;
    float w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = wp[(long long)e * a.wsk];
    if (WMAX) {
        float mw = fmaxf(fmaxf(fabsf(w[0]), fabsf(w[1])), fmaxf(fabsf(w[2]), fabsf(w[3])));
        for (int o = 32; o >= 1; o >>= 1) mw = fmaxf(mw, __shfl_xor(mw, o));
        __shared__ float red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mw;
        __syncthreads();
        return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
    bf16x4 h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        __bf16 hh, mm, ll;
        split3(w[e], hh, mm, ll);
        h[e] = hh; m[e] = mm; l[e] = ll;
    }
    const int ks = c4 >> 2, hh = (c4 >> 1) & 1, jo = 4 * (c4 & 1);
    if (one == 3) {
        f16x4 fh, fm;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 a0, a1;
            split2h(w[e] * h2sw, a0, a1);
            fh[e] = a0; fm[e] = a1;
        }
        __bf16* d = img + (long long)tile * (kTileElems * 2 / 3) + (i + 32 * hh) * 8 + jo;
        *reinterpret_cast<f16x4*>(d + (ks * 2 + 0) * 512) = fh;
        *reinterpret_cast<f16x4*>(d + (ks * 2 + 1) * 512) = fm;
        return 0.f;
    }
    if (one) {
        __bf16* d = img + (long long)tile * (kTileElems / 3) + (i + 32 * hh) * 8 + jo;
        *reinterpret_cast<bf16x4*>(d + ks * 512) = h;
        return 0.f;
    }
    __bf16* d = img + (long long)tile * kTileElems + (i + 32 * hh) * 8 + jo;
    *reinterpret_cast<bf16x4*>(d + (ks * 3 + 0) * 512) = h;
    *reinterpret_cast<bf16x4*>(d + (ks * 3 + 1) * 512) = m;
    *reinterpret_cast<bf16x4*>(d + (ks * 3 + 2) * 512) = l;
    return 0.f;
}

}  // namespace wn
