// fp16 x 2 weight images of the fused 32-channel layer kernels (mfma_layer.hip), shared with the step plan (plan.hip).
#pragma once
#include "h2_ops.hpp"

namespace wn {

static constexpr int kH2ImgBytes = 20 * 1024;
static constexpr int kH2ImgStride = kH2ImgBytes + 256;      // + the inverse scale

struct PackH2Args { const float* Wf[64]; const float* Wg[64]; const float* Wp[64]; };
// one layer's image, by one workgroup of 256 threads (k_layer_pack_h2: one launch for a stack; plan.hip: inside the step's
// preparation launch)
template <class A>
__device__ __forceinline__ void layer_pack_h2_block(const A& a, char* __restrict__ img_all, int l) {
    const float* Wf = a.Wf[l];
    const float* Wg = a.Wg[l];
    const float* Wp = a.Wp[l];
    char* img = img_all + (size_t)l * kH2ImgStride;
    float4 s_wf[2], s_wg[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        s_wf[k] = reinterpret_cast<const float4*>(Wf)[threadIdx.x + k * 256];
        s_wg[k] = reinterpret_cast<const float4*>(Wg)[threadIdx.x + k * 256];
    }
    const float4 s_wp = reinterpret_cast<const float4*>(Wp)[threadIdx.x];
    float mw = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        mw = fmaxf(mw, fmaxf(fmaxf(fabsf(s_wf[k].x), fabsf(s_wf[k].y)), fmaxf(fabsf(s_wf[k].z), fabsf(s_wf[k].w))));
        mw = fmaxf(mw, fmaxf(fmaxf(fabsf(s_wg[k].x), fabsf(s_wg[k].y)), fmaxf(fabsf(s_wg[k].z), fabsf(s_wg[k].w))));
    }
    mw = fmaxf(mw, fmaxf(fmaxf(fabsf(s_wp.x), fabsf(s_wp.y)), fmaxf(fabsf(s_wp.z), fabsf(s_wp.w))));
    mw = lb_wave_max(mw);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mw;
    __syncthreads();
    mw = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sw, iw;
    lb_pow2_scale(mw, sw, iw);
    if (threadIdx.x == 0) *reinterpret_cast<float*>(img + kH2ImgBytes) = iw;
    auto put = [&](int mt, int kidx, int jj, float v) {          // kidx: the contraction channel, jj: the output row
        const int hh = (kidx >> 2) & 1, ks = kidx >> 4, e = (kidx & 3) + 4 * ((kidx >> 3) & 1);
        const float xs = v * sw;
        const _Float16 hv = (_Float16)xs;
        const _Float16 mv = (_Float16)(xs - (float)hv);
        char* dst = img + ((mt * 2 + ks) * 2) * 1024 + (hh * 32 + jj) * 16 + e * 2;
        *reinterpret_cast<_Float16*>(dst) = hv;
        *reinterpret_cast<_Float16*>(dst + 1024) = mv;
    };
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e0 = (threadIdx.x + k * 256) * 4;                 // flat index into W[cd][cr][tap]
        const int cd = e0 >> 6, cr = (e0 >> 1) & 31;
        put(0, cr, cd, s_wf[k].x); put(1, cr, cd, s_wf[k].y); put(0, cr + 1, cd, s_wf[k].z); put(1, cr + 1, cd, s_wf[k].w);
        put(2, cr, cd, s_wg[k].x); put(3, cr, cd, s_wg[k].y); put(2, cr + 1, cd, s_wg[k].z); put(3, cr + 1, cd, s_wg[k].w);
    }
    {
        const int e0 = threadIdx.x * 4;                               // flat index into Wp[cr][cd]
        const int cr = e0 >> 5, cd = e0 & 31;
        put(4, cd, cr, s_wp.x); put(4, cd + 1, cr, s_wp.y); put(4, cd + 2, cr, s_wp.z); put(4, cd + 3, cr, s_wp.w);
    }
}

}  // namespace wn
