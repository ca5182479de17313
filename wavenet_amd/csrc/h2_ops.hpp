// fp16 x 2 split-product helpers shared by the fused 32-channel layer kernels (mfma_layer.hip, mfma_layer_bwd.hip).
// An fp32 value v, scaled by a power of two s so that |v s| < 2^15, is carried as two fp16 parts h + m (22 bits); a
// product of two such values is the three terms m h' + h m' + h h' of v_mfma_f32_32x32x16_f16, exact to 2^-21.
#pragma once
#include <hip/hip_runtime.h>

namespace wn {

typedef float h2_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
struct H2Op { h16x8 h[2], m[2]; };
__device__ __forceinline__ float lb_dpp(float v, int which) {
    int r;
    switch (which) {
        case 1: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false); break;
        case 2: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false); break;
        case 4: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, false); break;
        case 8: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, false); break;
        case 15: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false); break;
        default: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false); break;
    }
    return __int_as_float(r);
}
// max over the wave of a non-negative value, uniform
__device__ __forceinline__ float lb_wave_max(float v) {
    v = fmaxf(v, lb_dpp(v, 1));
    v = fmaxf(v, lb_dpp(v, 2));
    v = fmaxf(v, lb_dpp(v, 4));
    v = fmaxf(v, lb_dpp(v, 8));
    v = fmaxf(v, lb_dpp(v, 15));
    v = fmaxf(v, lb_dpp(v, 31));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// power of two s with s * mx in [2^14, 2^15) (mx = 0 or denormal: the largest finite power of two), and 1 / s
__device__ __forceinline__ void lb_pow2_scale(float mx, float& s, float& inv) {
    const int e = (__float_as_int(mx) >> 23) & 0xff;           // biased exponent of mx
    int f = 268 - e;                                           // 127 + 14 - (e - 127)
    f = f > 254 ? 254 : (f < 1 ? 1 : f);
    s = __int_as_float(f << 23);
    inv = __int_as_float((254 - f) << 23);                     // f = 254: 0 (the tile is all zeros or denormals)
}
// x s = h + m for 16 values.  The residual m = x s - h is ONE instruction per value, v_fma_mix{lo,hi}_f16 (fp32 x fp32 - fp16
// -> fp16: s is a power of two, so x s is exact and the fused form equals the two-step one bit for bit); written in C,
// hipcc's SLP pass turns the residual into "convert h back to fp32, packed subtract, convert again": six instructions per
// pair of values instead of four -- a third of all the vector instructions of a split, in kernels that are bound by
// vector-instruction issue.  One instruction per asm statement and no `volatile`: the scheduler moves them as freely as
// any other VALU instruction (the four-instruction asm chains of an earlier attempt could not be interleaved with the
// MFMAs and lost 5 %).
#ifndef WN_SPLIT_C_ONLY
__device__ __forceinline__ void lb_split16(const float (&v)[16], float s, H2Op& o) {
    typedef unsigned h2_u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        h2_u32x4 mp;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float a = v[8 * ks + 2 * p], b = v[8 * ks + 2 * p + 1];
            typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
            h16x2_t hp;
            hp[0] = (_Float16)(a * s);
            hp[1] = (_Float16)(b * s);
            o.h[ks][2 * p] = hp[0];
            o.h[ks][2 * p + 1] = hp[1];
            const unsigned hu = __builtin_bit_cast(unsigned, hp);
            unsigned m;
            asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(m) : "v"(a), "v"(s), "v"(hu));
            asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(m) : "v"(b), "v"(s), "v"(hu));
            mp[p] = m;
        }
        o.m[ks] = __builtin_bit_cast(h16x8, mp);
    }
}
#else
__device__ __forceinline__ void lb_split16(const float (&v)[16], float s, H2Op& o) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xs = v[8 * ks + e] * s;
            const _Float16 hh = (_Float16)xs;
            o.h[ks][e] = hh;
            o.m[ks][e] = (_Float16)(xs - (float)hh);
        }
}
#endif

// LDS-DMA of one 1 KB piece (global_load_lds_dwordx4: lane L's 16 bytes at `src` land at lds_dst + 16 L; lds_dst is
// wave-uniform) as inline asm.  With the builtin, hipcc knows that LDS is written asynchronously and puts `s_waitcnt
// vmcnt(0)` in front of every later LDS read that may alias it -- which also waits for every store issued since.  The asm
// form is opaque to that pass: the kernel's own counted `s_waitcnt vmcnt(N)` (vector-memory operations retire in issue
// order) is then the only wait, so every read of the destination must sit behind one, and a barrier that is meant to
// publish the data needs an explicit wait in front of it (the compiler no longer drains vmcnt there by itself).
__device__ __forceinline__ void lds_dma16(const void* src, void* lds_dst) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
}

#if defined(WN_SC_MODE) && WN_SC_MODE == 3
#define WN_SC_BITS "sc0 sc1"
#define WN_SC_AUX 17
#else
#define WN_SC_BITS "sc1"
#define WN_SC_AUX 16
#endif
// The same request with the sc1 bit: coherent at agent scope (the line is fetched from memory, not from a possibly stale copy
// in this XCD's L2).  For data another XCD wrote earlier in the SAME launch with st16_sc1.
__device__ __forceinline__ void lds_dma16_sc1(const void* src, void* lds_dst) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off " WN_SC_BITS ::"v"(src), "s"(m0v) : "memory", "m0");
}
// One dword per lane into LDS (lane L's word at lds_dst + 4 L), sc1: a way to request words whose VALUE is needed later
// without giving the compiler a register to wait for -- the kernel's own counted s_waitcnt covers the request.
__device__ __forceinline__ void lds_dma4_sc1(const void* src, void* lds_dst) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off sc1" ::"v"(src), "s"(m0v) : "memory", "m0");
}
// 16-byte store written through to memory at agent scope (sc1): visible to every XCD once the wave's vmcnt has counted it.
typedef float h2_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_sc1(float* dst, float a, float b, float c, float d) {
    const h2_f32x4 v = {a, b, c, d};
    // s_nop 1: the two wait states a VALU write of the data registers needs behind a store of more than 8 bytes (gfx940+);
    // hipcc's hazard recogniser does not look inside inline asm
    asm volatile("global_store_dwordx4 %0, %1, off " WN_SC_BITS "\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}

}  // namespace wn
