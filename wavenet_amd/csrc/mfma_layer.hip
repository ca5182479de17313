// Fused residual layer forward on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), for the shape
// BASELINE.json's configs 2-4 use: Cr = Cd = 32, filter width 2.  Reference op sequence replaced:
// ResidualConvLayer.__call__ (wavenet.py:358-368) = 2 x DilatedConvolution1D.__call__
// (wavenet.py:294-342) + tanh * sigmoid + projection_block + residual add.
//
// One wave owns a tile of 32 time columns and computes  D[channel][time] = W[channel][k] X[k][time]
// with time on the MFMA's N (lane) axis.  The contraction order is permuted so that everything
// stays in registers, no LDS and no cross-lane traffic:
//
//   k-step s (0..15), lane half h (lane>>5)  <->  channel ch(s,h) = (s&3) + 8(s>>2) + 4h
//
// With that order (a) a lane's 16 B-operand values of x[t] are four float4 loads (channels
// 8q+4h .. 8q+4h+3), (b) the accumulator register r of lane (j,h) holds output channel ch(r,h) of
// column j -- the same map -- so the gate output z is directly the B operand of the projection
// MFMA, x[t] is directly its C input (the residual add is free), and out / z / f / g are stored as
// float4 with the addresses of the loads.  Weights are read once per wave in their reference
// layout W[o][c][k] (tap pairs are adjacent, so two float4 give both taps of four channels).
//
// Per tile: 64 + 16 MFMAs (5120 SIMD cycles); 2 KB + 2 KB of x read, 4 KB out + 4 KB z written.
#include <cstdlib>

#include "h2_ops.hpp"
#include "layer_pack.hpp"
#include "wn_kernels.hpp"

namespace wn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int ch_of(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

static constexpr int kWRow = 66;     // LDS row stride of W[i][.][.] (64 floats + 2: rows land on different banks)
static constexpr int kPRow = 33;     // LDS row stride of Wp[i][.]

template <int SAVE, bool HAS_BIAS>
__global__ __launch_bounds__(256, 2) void k_layer_fwd_mfma32(
    const float* __restrict__ x, const float* __restrict__ Wf, const float* __restrict__ bf,
    const float* __restrict__ Wg, const float* __restrict__ bg, const float* __restrict__ Wp,
    const float* __restrict__ bp, float* __restrict__ out, float* __restrict__ zout,
    float* __restrict__ fout, float* __restrict__ gout, int B, int T, int d, int Z, int tile_lo, int tiles_per_b,
    int ntiles) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 31;      // time column inside the tile (B/D operand), weight row (A operand)
    const int h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Tile order: workgroups are dealt to the 8 XCDs round-robin, so XCD k = blockIdx % 8 is given the k-th contiguous
    // eighth of the tiles: x[t-d] of a tile is then a row range that a neighbouring CU of the SAME XCD fetches as its
    // x[t] in the same round -- an L2 hit instead of a second trip to HBM.
    int first, stride, last;                        // this wave's tiles: first, first + stride, ... < last
    if ((gridDim.x & 7) == 0) {
        const int per_xcd = (ntiles + 7) >> 3;
        const int xcd = blockIdx.x & 7;
        stride = (gridDim.x >> 3) * 4;
        first = xcd * per_xcd + (blockIdx.x >> 3) * 4 + wv;
        last = (xcd + 1) * per_xcd < ntiles ? (xcd + 1) * per_xcd : ntiles;
    } else {
        stride = gridDim.x * 4;
        first = blockIdx.x * 4 + wv;
        last = ntiles;
    }

    // ---- A operands: lane (i=j, h), step s holds W[i][ch(s,h)] ------------------------------
    // The workgroup stages the layer's weights in LDS with coalesced 16-byte loads (rows padded to 66 / 33 floats) and
    // every wave picks its operands from there.  Fetching them per wave straight from global memory -- 20 float4
    // loads whose lanes sit 256 B apart, 32+ cache lines per instruction -- kept the texture addresser busy for
    // ~14,000 cycles per wave (6 us of a 30 us kernel, measured with s_memtime stamps).
    __shared__ __attribute__((aligned(16))) float wlds[2 * 32 * kWRow + 32 * kPRow];
    float4 sa[2], sc[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        sa[k] = *reinterpret_cast<const float4*>(Wf + (threadIdx.x + 256 * k) * 4);    // flat W[32][32][2]
        sc[k] = *reinterpret_cast<const float4*>(Wg + (threadIdx.x + 256 * k) * 4);
    }
    const float4 sp = *reinterpret_cast<const float4*>(Wp + threadIdx.x * 4);          // flat Wp[32][32]
    // biases (off by default in the reference, wavenet.py:116-117) are re-read per tile in accumulator
    // layout -- register r of lane (.,h) is channel ch(r,h) -- instead of occupying 48 registers

    // x[t] and x[t-d] of a tile: unconditional loads from clamped rows, masked afterwards (a "cond ? load : 0"
    // makes hipcc branch around every load and drain vmcnt per element)
    auto load_tile = [&](int tile, float (&xc)[16], float (&xo)[16]) {
        const int b = tile / tiles_per_b;
        const int t = (tile_lo + tile - b * tiles_per_b) * 32 + j;
        const bool valid = t < T;
        const int tc = valid ? t : T - 1;
        const long long rowc = ((long long)b * T + tc) * 32 + 4 * h;
        const long long rowo = ((long long)b * T + (tc - d >= 0 ? tc - d : 0)) * 32 + 4 * h;
        const float mc = valid ? 1.f : 0.f, mo = (valid && t - d >= 0) ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(x + rowc + 8 * q);
            const float4 o = *reinterpret_cast<const float4*>(x + rowo + 8 * q);
            xc[4 * q + 0] = v.x * mc; xc[4 * q + 1] = v.y * mc; xc[4 * q + 2] = v.z * mc; xc[4 * q + 3] = v.w * mc;
            xo[4 * q + 0] = o.x * mo; xo[4 * q + 1] = o.y * mo; xo[4 * q + 2] = o.z * mo; xo[4 * q + 3] = o.w * mo;
        }
    };

    float xc[16], xo[16], xcn[16], xon[16];
    if (first < last) load_tile(first, xc, xo);          // in flight while the weights settle in LDS
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = (threadIdx.x + 256 * k) * 4;
        float* lf = wlds + (e >> 6) * kWRow + (e & 63);
        float* lg = lf + 32 * kWRow;
        *reinterpret_cast<float2*>(lf) = make_float2(sa[k].x, sa[k].y);
        *reinterpret_cast<float2*>(lf + 2) = make_float2(sa[k].z, sa[k].w);
        *reinterpret_cast<float2*>(lg) = make_float2(sc[k].x, sc[k].y);
        *reinterpret_cast<float2*>(lg + 2) = make_float2(sc[k].z, sc[k].w);
    }
    {
        const int e = threadIdx.x * 4;
        float* lp = wlds + 2 * 32 * kWRow + (e >> 5) * kPRow + (e & 31);
        lp[0] = sp.x; lp[1] = sp.y; lp[2] = sp.z; lp[3] = sp.w;
    }
    __syncthreads();
    float wf0[16], wf1[16], wg0[16], wg1[16], wp[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // W[i][c][k], c = 8q+4h .. +3, k = 0,1  ->  8 consecutive floats of row i
        const float* pf = wlds + j * kWRow + 16 * q + 8 * h;
        const float* pg = pf + 32 * kWRow;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float2 a = *reinterpret_cast<const float2*>(pf + 2 * m);
            const float2 c = *reinterpret_cast<const float2*>(pg + 2 * m);
            wf0[4 * q + m] = a.x; wf1[4 * q + m] = a.y;
            wg0[4 * q + m] = c.x; wg1[4 * q + m] = c.y;
            wp[4 * q + m] = wlds[2 * 32 * kWRow + j * kPRow + 8 * q + 4 * h + m];
        }
    }
    for (int tile = first; tile < last; tile += stride) {
        const int b = tile / tiles_per_b;
        const int t = (tile_lo + tile - b * tiles_per_b) * 32 + j;
        const bool valid = t < T;
        const long long row = ((long long)b * T + t) * 32 + 4 * h;
        // the next tile's columns are fetched while this tile computes
        const bool more = tile + stride < last;
        if (more) load_tile(tile + stride, xcn, xon);
        f32x16 aa, ag;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            aa[r] = (HAS_BIAS && bf) ? bf[ch_of(r, h)] : 0.f;
            ag[r] = (HAS_BIAS && bg) ? bg[ch_of(r, h)] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            aa = __builtin_amdgcn_mfma_f32_32x32x2f32(wf0[s], xo[s], aa, 0, 0, 0);
            ag = __builtin_amdgcn_mfma_f32_32x32x2f32(wg0[s], xo[s], ag, 0, 0, 0);
            aa = __builtin_amdgcn_mfma_f32_32x32x2f32(wf1[s], xc[s], aa, 0, 0, 0);
            ag = __builtin_amdgcn_mfma_f32_32x32x2f32(wg1[s], xc[s], ag, 0, 0, 0);
        }
        const bool live = t >= Z;       // reference zero prefix: conv outputs (and bias) are 0 there
        float zz[16], ff[16], gg[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = live ? aa[r] : 0.f;
            float g = live ? ag[r] : 0.f;
            ff[r] = fast_tanh(a);
            gg[r] = fast_sigmoid(g);
            zz[r] = ff[r] * gg[r];
        }
        f32x16 ao;
#pragma unroll
        for (int r = 0; r < 16; ++r) ao[r] = xc[r] + ((HAS_BIAS && bp) ? bp[ch_of(r, h)] : 0.f);
#pragma unroll
        for (int s = 0; s < 16; ++s) ao = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[s], zz[s], ao, 0, 0, 0);
        if (valid) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<float4*>(out + row + 8 * q) =
                    make_float4(ao[4 * q], ao[4 * q + 1], ao[4 * q + 2], ao[4 * q + 3]);
                *reinterpret_cast<float4*>(zout + row + 8 * q) =
                    make_float4(zz[4 * q], zz[4 * q + 1], zz[4 * q + 2], zz[4 * q + 3]);
                if (SAVE == 1)
                    *reinterpret_cast<float4*>(fout + row + 8 * q) =
                        make_float4(ff[4 * q], ff[4 * q + 1], ff[4 * q + 2], ff[4 * q + 3]);
                if (SAVE >= 1)
                    *reinterpret_cast<float4*>(gout + row + 8 * q) =
                        make_float4(gg[4 * q], gg[4 * q + 1], gg[4 * q + 2], gg[4 * q + 3]);
            }
        }
        if (more) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { xc[r] = xcn[r]; xo[r] = xon[r]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// One tile per wave, four waves per SIMD.  Same maths and data layout as k_layer_fwd_mfma32; what changes is the
// occupancy: the A operands are read from LDS at each MFMA instead of living in 80 registers, so a wave fits in 128
// registers, 16 waves share a CU, and a wave's load latency, gate arithmetic and stores run under the MFMAs of three
// others instead of under one sibling's.  No tile loop, no prefetch, no tile-count quantisation: a workgroup is four
// consecutive tiles.
// ---------------------------------------------------------------------------------------------
template <int SAVE, bool HAS_BIAS>
__global__ __launch_bounds__(256, 4) void k_layer_fwd_mfma32_t1(
    const float* __restrict__ x, const float* __restrict__ Wf, const float* __restrict__ bf,
    const float* __restrict__ Wg, const float* __restrict__ bg, const float* __restrict__ Wp,
    const float* __restrict__ bp, float* __restrict__ out, float* __restrict__ zout,
    float* __restrict__ fout, float* __restrict__ gout, int B, int T, int d, int Z, int tile_lo, int tiles_per_b,
    int ntiles) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 31;
    const int h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD k = blockIdx % 8 gets the k-th contiguous eighth of the workgroups (see k_layer_fwd_mfma32)
    int wg = blockIdx.x;
    if ((gridDim.x & 7) == 0) wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int tile = wg * 4 + wv;
    const bool tvalid = tile < ntiles;
    const int tl = tvalid ? tile : ntiles - 1;
    const int b = tl / tiles_per_b;
    const int t = (tile_lo + tl - b * tiles_per_b) * 32 + j;
    const bool valid = tvalid && t < T;
    const int tc = t < T ? t : T - 1;

    __shared__ __attribute__((aligned(16))) float wlds[2 * 32 * kWRow + 32 * kPRow];
    float4 sa[2], sc[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        sa[k] = *reinterpret_cast<const float4*>(Wf + (threadIdx.x + 256 * k) * 4);
        sc[k] = *reinterpret_cast<const float4*>(Wg + (threadIdx.x + 256 * k) * 4);
    }
    const float4 sp = *reinterpret_cast<const float4*>(Wp + threadIdx.x * 4);
    // this tile's columns: unconditional loads from clamped rows, masked afterwards
    float xc[16], xo[16];
    {
        const long long rowc = ((long long)b * T + tc) * 32 + 4 * h;
        const long long rowo = ((long long)b * T + (tc - d >= 0 ? tc - d : 0)) * 32 + 4 * h;
        const float mc = valid ? 1.f : 0.f, mo = (valid && t - d >= 0) ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(x + rowc + 8 * q);
            const float4 o = *reinterpret_cast<const float4*>(x + rowo + 8 * q);
            xc[4 * q + 0] = v.x * mc; xc[4 * q + 1] = v.y * mc; xc[4 * q + 2] = v.z * mc; xc[4 * q + 3] = v.w * mc;
            xo[4 * q + 0] = o.x * mo; xo[4 * q + 1] = o.y * mo; xo[4 * q + 2] = o.z * mo; xo[4 * q + 3] = o.w * mo;
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = (threadIdx.x + 256 * k) * 4;
        float* lf = wlds + (e >> 6) * kWRow + (e & 63);
        float* lg = lf + 32 * kWRow;
        *reinterpret_cast<float2*>(lf) = make_float2(sa[k].x, sa[k].y);
        *reinterpret_cast<float2*>(lf + 2) = make_float2(sa[k].z, sa[k].w);
        *reinterpret_cast<float2*>(lg) = make_float2(sc[k].x, sc[k].y);
        *reinterpret_cast<float2*>(lg + 2) = make_float2(sc[k].z, sc[k].w);
    }
    {
        const int e = threadIdx.x * 4;
        float* lp = wlds + 2 * 32 * kWRow + (e >> 5) * kPRow + (e & 31);
        lp[0] = sp.x; lp[1] = sp.y; lp[2] = sp.z; lp[3] = sp.w;
    }
    __syncthreads();

    f32x16 aa, ag;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        aa[r] = (HAS_BIAS && bf) ? bf[ch_of(r, h)] : 0.f;
        ag[r] = (HAS_BIAS && bg) ? bg[ch_of(r, h)] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        // W[i = j][c = ch(s,h)][k = 0,1]: one 8-byte LDS read gives both taps
        const float* pf = wlds + j * kWRow + 16 * (s >> 2) + 8 * h + 2 * (s & 3);
        const float2 a = *reinterpret_cast<const float2*>(pf);
        const float2 c = *reinterpret_cast<const float2*>(pf + 32 * kWRow);
        aa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xo[s], aa, 0, 0, 0);
        ag = __builtin_amdgcn_mfma_f32_32x32x2f32(c.x, xo[s], ag, 0, 0, 0);
        aa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xc[s], aa, 0, 0, 0);
        ag = __builtin_amdgcn_mfma_f32_32x32x2f32(c.y, xc[s], ag, 0, 0, 0);
    }
    const bool live = t >= Z;
    const long long row = ((long long)b * T + t) * 32 + 4 * h;
    float zz[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float f4[4], g4[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int r = 4 * q + m;
            f4[m] = fast_tanh(live ? aa[r] : 0.f);
            g4[m] = fast_sigmoid(live ? ag[r] : 0.f);
            zz[r] = f4[m] * g4[m];
        }
        if (valid) {                                 // f, g and z leave as soon as they exist: registers stay under 128
            if (SAVE == 1) *reinterpret_cast<float4*>(fout + row + 8 * q) = make_float4(f4[0], f4[1], f4[2], f4[3]);
            if (SAVE >= 1) *reinterpret_cast<float4*>(gout + row + 8 * q) = make_float4(g4[0], g4[1], g4[2], g4[3]);
            *reinterpret_cast<float4*>(zout + row + 8 * q) = make_float4(zz[4 * q], zz[4 * q + 1], zz[4 * q + 2], zz[4 * q + 3]);
        }
    }
    f32x16 ao;
#pragma unroll
    for (int r = 0; r < 16; ++r) ao[r] = xc[r] + ((HAS_BIAS && bp) ? bp[ch_of(r, h)] : 0.f);
#pragma unroll
    for (int s = 0; s < 16; ++s)
        ao = __builtin_amdgcn_mfma_f32_32x32x2f32(wlds[2 * 32 * kWRow + j * kPRow + ch_of(s, h)], zz[s], ao, 0, 0, 0);
    if (valid) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(out + row + 8 * q) = make_float4(ao[4 * q], ao[4 * q + 1], ao[4 * q + 2], ao[4 * q + 3]);
    }
}

// ---------------------------------------------------------------------------------------------
// The same layer on fp16 x 2 split products (WnExec.precision == WN_GEMM_FP16X2, no conv / projection biases):
// 30 v_mfma_f32_32x32x16_f16 per tile (960 SIMD cycles) instead of 80 fp32 MFMAs (5,120).  Nothing about the data
// layout changes: a lane's 16 channels of x[t] (ch(s, h), s = 0..15) are an f16 operand pair as they stand (k-step ks,
// element e <-> s = 8 ks + e), the accumulator register r still holds channel ch(r, h), so z feeds the projection
// from registers and x[t] is added to its result in place.  x is scaled per tile by the power of two that brings the
// wave's own maximum below 2^15, z = tanh sigmoid by 2^14; the weights arrive as pre-split images in A-operand order
// (k_layer_pack_h2: one launch for all layers of a stack, scale = a power of two from the layer's largest weight).
// Image of a layer: 1 KB per (matrix-tap mt, k-step ks, part): lane (j, hh) element e =
//     mt 0..3 (Wf tap 0, Wf tap 1, Wg tap 0, Wg tap 1):  W[cd = j][cr = ch(8 ks + e, hh)][tap]
//     mt 4 (Wp):                                         Wp[cr = j][cd = ch(8 ks + e, hh)]
// followed by 1 / scale.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_layer_pack_h2(PackH2Args a, char* __restrict__ img_all) {
    layer_pack_h2_block(a, img_all, (int)blockIdx.x);
}

template <int SAVE>
__global__ __launch_bounds__(256, 4) void k_layer_fwd_h2_t1(
    const float* __restrict__ x, const char* __restrict__ img_g, float* __restrict__ out, float* __restrict__ zout,
    float* __restrict__ fout, float* __restrict__ gout, int B, int T, int d, int Z, int tile_lo, int tiles_per_b,
    int ntiles) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 31;
    const int h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int wg = blockIdx.x;
    if ((gridDim.x & 7) == 0) wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int tile = wg * 4 + wv;
    const bool tvalid = tile < ntiles;
    const int tl = tvalid ? tile : ntiles - 1;
    const int b = tl / tiles_per_b;
    const int t = (tile_lo + tl - b * tiles_per_b) * 32 + j;
    const bool valid = tvalid && t < T;
    const int tc = t < T ? t : T - 1;

    __shared__ __attribute__((aligned(16))) char img[kH2ImgBytes];
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t stage[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) stage[k] = reinterpret_cast<const u32x4_t*>(img_g)[threadIdx.x + 256 * k];
    const float w_inv = *reinterpret_cast<const float*>(img_g + kH2ImgBytes);
    float xc[16], xo[16];
    {
        const long long rowc = ((long long)b * T + tc) * 32 + 4 * h;
        const long long rowo = ((long long)b * T + (tc - d >= 0 ? tc - d : 0)) * 32 + 4 * h;
        const float mc = valid ? 1.f : 0.f, mo = (valid && t - d >= 0) ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(x + rowc + 8 * q);
            const float4 o = *reinterpret_cast<const float4*>(x + rowo + 8 * q);
            xc[4 * q + 0] = v.x * mc; xc[4 * q + 1] = v.y * mc; xc[4 * q + 2] = v.z * mc; xc[4 * q + 3] = v.w * mc;
            xo[4 * q + 0] = o.x * mo; xo[4 * q + 1] = o.y * mo; xo[4 * q + 2] = o.z * mo; xo[4 * q + 3] = o.w * mo;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) reinterpret_cast<u32x4_t*>(img)[threadIdx.x + 256 * k] = stage[k];
    __syncthreads();

    float mx = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) mx = fmaxf(mx, fmaxf(fabsf(xc[s]), fabsf(xo[s])));
    mx = lb_wave_max(mx);
    float sx, ix;
    lb_pow2_scale(mx, sx, ix);
    H2Op oc, oo;
    lb_split16(xc, sx, oc);
    lb_split16(xo, sx, oo);
    f32x16 aa, ag;
#pragma unroll
    for (int r = 0; r < 16; ++r) { aa[r] = 0.f; ag[r] = 0.f; }
    const char* ib = img + lane * 16;
    auto frag = [&](int mt, int ks, int part) {
        return *reinterpret_cast<const h16x8*>(ib + ((mt * 2 + ks) * 2 + part) * 1024);
    };
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int tap = 0; tap < 2; ++tap) {
            const H2Op& ob = tap == 0 ? oo : oc;                     // tap 0 multiplies x[t - d]
            const h16x8 fh = frag(tap, ks, 0), fm = frag(tap, ks, 1);
            const h16x8 gh = frag(2 + tap, ks, 0), gm = frag(2 + tap, ks, 1);
            aa = __builtin_amdgcn_mfma_f32_32x32x16_f16(fm, ob.h[ks], aa, 0, 0, 0);
            ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(gm, ob.h[ks], ag, 0, 0, 0);
            aa = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, ob.m[ks], aa, 0, 0, 0);
            ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, ob.m[ks], ag, 0, 0, 0);
            aa = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, ob.h[ks], aa, 0, 0, 0);
            ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, ob.h[ks], ag, 0, 0, 0);
        }
    }
    const float uc = ix * w_inv;
    const bool live = t >= Z;
    const long long row = ((long long)b * T + t) * 32 + 4 * h;
    float zz[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float f4[4], g4[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int r = 4 * q + m;
            f4[m] = fast_tanh(live ? aa[r] * uc : 0.f);
            g4[m] = fast_sigmoid(live ? ag[r] * uc : 0.f);
            zz[r] = f4[m] * g4[m];
        }
        if (valid) {
            if (SAVE == 1) *reinterpret_cast<float4*>(fout + row + 8 * q) = make_float4(f4[0], f4[1], f4[2], f4[3]);
            if (SAVE >= 1) *reinterpret_cast<float4*>(gout + row + 8 * q) = make_float4(g4[0], g4[1], g4[2], g4[3]);
            *reinterpret_cast<float4*>(zout + row + 8 * q) = make_float4(zz[4 * q], zz[4 * q + 1], zz[4 * q + 2], zz[4 * q + 3]);
        }
    }
    H2Op oz;
    lb_split16(zz, 16384.f, oz);
    f32x16 ao;
#pragma unroll
    for (int r = 0; r < 16; ++r) ao[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const h16x8 ph = frag(4, ks, 0), pm = frag(4, ks, 1);
        ao = __builtin_amdgcn_mfma_f32_32x32x16_f16(pm, oz.h[ks], ao, 0, 0, 0);
        ao = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, oz.m[ks], ao, 0, 0, 0);
        ao = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, oz.h[ks], ao, 0, 0, 0);
    }
    const float up = w_inv * (1.f / 16384.f);
    if (valid) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(out + row + 8 * q) =
                make_float4(fmaf(ao[4 * q], up, xc[4 * q]), fmaf(ao[4 * q + 1], up, xc[4 * q + 1]),
                            fmaf(ao[4 * q + 2], up, xc[4 * q + 2]), fmaf(ao[4 * q + 3], up, xc[4 * q + 3]));
    }
}

// ---------------------------------------------------------------------------------------------
// A GROUP of consecutive small-dilation layers in one launch (fp16 x 2 form).  With per-layer launches every layer reads
// its input back from HBM (128 of the 512 bytes a column costs in training form) and pays a launch's ramp: every wave
// of the grid loads, computes and stores at the same time, so loads and stores never overlap.  Layers whose dilations add
// up to at most 32 (d = 1, 2, 4, 8, 16 of every block) can be chained inside a workgroup instead: a workgroup owns 8
// consecutive tiles plus the tile to their left (the halo, recomputed: 12.5 % more arithmetic, no extra HBM reads beyond
// the first layer's), one tile per wave, 9 waves.  (Eight main tiles, not seven: config 2's 4,096 tiles are then 512
// workgroups = exactly the two per CU that fit -- with seven, 592 workgroups ran as one full round plus a round of 80,
// and the launch took as long as the five it replaced.)  Per layer a wave
//   1. puts its tile of the layer's input (registers: the previous layer's output, or the group's input from memory)
//      into the workgroup's slab in LDS (time-major rows, 16-byte chunk c of row r at position c ^ (r & 7)),
//   2. after a barrier takes x[t - d] from the slab (its own rows and the left neighbour's), second barrier,
//   3. runs the layer exactly as k_layer_fwd_h2_t1 does (same MFMAs, same scales, same stores of z / sigmoid / out) and
//      keeps out in registers as the next layer's x[t].
// The halo tile's results are wrong to the left of column t0 - 32 + (sum of the dilations so far) and are read only to
// the right of it; it stores nothing.  The layer's weight image comes from the stack's packed images by LDS-DMA, one
// layer ahead, into the other of two buffers (counted waits: the pieces are issued BEFORE the layer's stores).
// LDS: slab 36 KB + 2 x 20 KB of images = 76 KB, two workgroups per CU (18 waves: at most 96 registers each).
// ---------------------------------------------------------------------------------------------
static constexpr int kGrpTiles = 8;                      // main tiles per workgroup (+ 1 halo tile)
static constexpr int kGrpWaves = kGrpTiles + 1;
static constexpr int kGrpMaxLayers = 8;
struct GrpArgs {
    const float* x;                                       // input of the group's first layer
    float* out[kGrpMaxLayers]; float* z[kGrpMaxLayers]; float* f[kGrpMaxLayers]; float* g[kGrpMaxLayers];
    int d[kGrpMaxLayers], Z[kGrpMaxLayers];
    const char* img;                                      // first layer's image; consecutive layers kH2ImgStride apart
    int nl, B, T, tiles_per_b, wgs_per_b;
};

template <int SAVE>
__global__ __launch_bounds__(64 * kGrpWaves, 5) void k_layer_fwd_h2_grp(GrpArgs a) {
    __shared__ __attribute__((aligned(16))) char imgs[2][kH2ImgBytes];
    __shared__ __attribute__((aligned(16))) float slab[kGrpWaves * 1024];
    const int lane = threadIdx.x & 63;
    const int j = lane & 31;
    const int h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD k = blockIdx % 8 gets the k-th contiguous eighth of the workgroups (neighbouring slabs share an L2)
    int wg = blockIdx.x;
    if ((gridDim.x & 7) == 0) wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int b = wg / a.wgs_per_b;
    const int ti = (wg - b * a.wgs_per_b) * kGrpTiles - 1 + wv;          // this wave's tile of clip b (-1: left of the clip)
    const int T = a.T;
    const int t = ti * 32 + j;
    const bool tile_ok = ti >= 0 && ti < a.tiles_per_b;                  // wave-uniform: the tile exists
    const bool valid = tile_ok && t < T;
    const bool writer = wv > 0 && tile_ok;                               // the halo tile stores nothing
    const int tc = t < 0 ? 0 : (t < T ? t : T - 1);
    const long long row = ((long long)b * T + tc) * 32 + 4 * h;

    // weight image of the first layer (all waves: 20 pieces of 1 KB), and this tile of the group's input
    auto dma_image = [&](int l, int buf) {
        const char* src = a.img + (size_t)l * kH2ImgStride + lane * 16;
        for (int piece = wv; piece < kH2ImgBytes / 1024; piece += kGrpWaves)
            lds_dma16(src + piece * 1024, imgs[buf] + piece * 1024);
    };
    dma_image(0, 0);
    float xc[16];
    {
        const float mc = valid ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(a.x + row + 8 * q);
            xc[4 * q + 0] = v.x * mc; xc[4 * q + 1] = v.y * mc; xc[4 * q + 2] = v.z * mc; xc[4 * q + 3] = v.w * mc;
        }
    }
    float* const my_rows = slab + wv * 1024 + j * 32;
    for (int l = 0; l < a.nl; ++l) {
        const int d = a.d[l];
        // 1. this tile of the layer's input -> slab
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(my_rows + (((2 * q + h) ^ (j & 7)) << 2)) =
                make_float4(xc[4 * q], xc[4 * q + 1], xc[4 * q + 2], xc[4 * q + 3]);
        // the image of this layer (requested a layer ago, before that layer's stores: everything but the youngest
        // `nstores` operations has retired) and the slab rows are published by the barrier
        if (l == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (writer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (2 + (SAVE >= 1 ? 1 : 0) + (SAVE == 1 ? 1 : 0))) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // the image's inverse scale, through the SCALAR cache.  (As a plain load it was a vector-memory load hipcc waited for
        // with s_waitcnt vmcnt(0) a dozen instructions later -- right behind the NEXT image's requests, which have a whole
        // layer to land, and this layer's stores.)
        float w_inv;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(w_inv) : "s"(a.img + (size_t)l * kH2ImgStride + kH2ImgBytes) : "memory");
        __syncthreads();
        // 2. x[t - d]: slab row (32 wv + j - d); rows left of the slab (halo tile only) and columns before the clip are 0
        float xo[16];
        {
            const int r = wv * 32 + j - d;
            const bool in = r >= 0 && t - d >= 0 && valid;
            const int rr = r < 0 ? 0 : r;
            const float* src = slab + rr * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 o = *reinterpret_cast<const float4*>(src + (((2 * q + h) ^ (rr & 7)) << 2));
                xo[4 * q + 0] = in ? o.x : 0.f; xo[4 * q + 1] = in ? o.y : 0.f;
                xo[4 * q + 2] = in ? o.z : 0.f; xo[4 * q + 3] = in ? o.w : 0.f;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                                 // every wave has its x[t - d]: the slab may be rewritten
        // the next layer's image goes out now, ahead of this layer's stores
        if (l + 1 < a.nl) dma_image(l + 1, (l + 1) & 1);
        // 3. the layer (k_layer_fwd_h2_t1's arithmetic)
        const char* img = imgs[l & 1];
        float mx = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) mx = fmaxf(mx, fmaxf(fabsf(xc[s]), fabsf(xo[s])));
        mx = lb_wave_max(mx);
        float sx, ix;
        lb_pow2_scale(mx, sx, ix);
        H2Op oc, oo;
        lb_split16(xc, sx, oc);
        lb_split16(xo, sx, oo);
        f32x16 aa, ag;
#pragma unroll
        for (int r = 0; r < 16; ++r) { aa[r] = 0.f; ag[r] = 0.f; }
        const char* ib = img + lane * 16;
        auto frag = [&](int mt, int ks, int part) {
            return *reinterpret_cast<const h16x8*>(ib + ((mt * 2 + ks) * 2 + part) * 1024);
        };
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int tap = 0; tap < 2; ++tap) {
                const H2Op& ob = tap == 0 ? oo : oc;                     // tap 0 multiplies x[t - d]
                const h16x8 fh = frag(tap, ks, 0), fm = frag(tap, ks, 1);
                const h16x8 gh = frag(2 + tap, ks, 0), gm = frag(2 + tap, ks, 1);
                aa = __builtin_amdgcn_mfma_f32_32x32x16_f16(fm, ob.h[ks], aa, 0, 0, 0);
                ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(gm, ob.h[ks], ag, 0, 0, 0);
                aa = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, ob.m[ks], aa, 0, 0, 0);
                ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, ob.m[ks], ag, 0, 0, 0);
                aa = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, ob.h[ks], aa, 0, 0, 0);
                ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, ob.h[ks], ag, 0, 0, 0);
            }
        }
        const float uc = ix * w_inv;
        const bool live = t >= a.Z[l];
        const bool st = writer && valid;
        float zz[16];
        float* const zout = a.z[l];
        float* const gout = a.g[l];
        float* const fout = a.f[l];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float f4[4], g4[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int r = 4 * q + m;
                f4[m] = fast_tanh(live ? aa[r] * uc : 0.f);
                g4[m] = fast_sigmoid(live ? ag[r] * uc : 0.f);
                zz[r] = f4[m] * g4[m];
            }
            if (writer) {                                                // (wave-uniform; the lanes beyond T are masked)
                if (st) {
                    if (SAVE == 1) *reinterpret_cast<float4*>(fout + row + 8 * q) = make_float4(f4[0], f4[1], f4[2], f4[3]);
                    if (SAVE >= 1) *reinterpret_cast<float4*>(gout + row + 8 * q) = make_float4(g4[0], g4[1], g4[2], g4[3]);
                    *reinterpret_cast<float4*>(zout + row + 8 * q) = make_float4(zz[4 * q], zz[4 * q + 1], zz[4 * q + 2], zz[4 * q + 3]);
                }
            }
        }
        H2Op oz;
        lb_split16(zz, 16384.f, oz);
        f32x16 ao;
#pragma unroll
        for (int r = 0; r < 16; ++r) ao[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const h16x8 ph = frag(4, ks, 0), pm = frag(4, ks, 1);
            ao = __builtin_amdgcn_mfma_f32_32x32x16_f16(pm, oz.h[ks], ao, 0, 0, 0);
            ao = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, oz.m[ks], ao, 0, 0, 0);
            ao = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, oz.h[ks], ao, 0, 0, 0);
        }
        const float up = w_inv * (1.f / 16384.f);
        const float mv = valid ? 1.f : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) xc[r] = fmaf(ao[r], up, xc[r]) * mv;     // the layer's output = the next layer's x[t]
        if (writer) {
            float* const oout = a.out[l];
            if (st) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(oout + row + 8 * q) = make_float4(xc[4 * q], xc[4 * q + 1], xc[4 * q + 2], xc[4 * q + 3]);
            }
        }
    }
}

// how many of the layers l0, l0 + 1, ... (dilations dil[]) one group launch can take: their dilations add up to <= 32
int mfma_layer_fwd_group_len(const int* dil, int l0, int L) {
    int n = 0, sum = 0;
    while (l0 + n < L && n < kGrpMaxLayers && sum + dil[l0 + n] <= 32) { sum += dil[l0 + n]; ++n; }
    return n;
}
// layers l0 .. l0 + nl - 1 of a packed stack in one launch (see k_layer_fwd_h2_grp); outs / zs / fs / gs are per layer
int mfma_layer_fwd_h2_group(const float* x, const void* img, int l0, int nl, float* const* outs, float* const* zs,
                            float* const* fs, float* const* gs, const int* dil, const int* Zs, int B, int T,
                            hipStream_t s) {
    WN_CHECK_ARG(nl >= 1 && nl <= kGrpMaxLayers, "mfma_layer_fwd_h2_group: 1..%d layers", kGrpMaxLayers);
    GrpArgs a{};
    a.x = x;
    a.img = reinterpret_cast<const char*>(img) + (size_t)l0 * kH2ImgStride;
    a.nl = nl; a.B = B; a.T = T;
    a.tiles_per_b = (T + 31) / 32;
    a.wgs_per_b = (a.tiles_per_b + kGrpTiles - 1) / kGrpTiles;
    for (int l = 0; l < nl; ++l) {
        a.out[l] = outs[l]; a.z[l] = zs[l]; a.f[l] = fs ? fs[l] : nullptr; a.g[l] = gs ? gs[l] : nullptr;
        a.d[l] = dil[l]; a.Z[l] = Zs[l];
    }
    const long long blocks = (long long)B * a.wgs_per_b;
    WN_CHECK_SHAPE(blocks < (1ll << 31), "mfma_layer_fwd_h2_group: too many workgroups");
    const dim3 grid((unsigned)blocks), block(64 * kGrpWaves);
    if (fs && fs[0]) hipLaunchKernelGGL(k_layer_fwd_h2_grp<1>, grid, block, 0, s, a);
    else if (gs && gs[0]) hipLaunchKernelGGL(k_layer_fwd_h2_grp<2>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(k_layer_fwd_h2_grp<0>, grid, block, 0, s, a);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

size_t mfma_layer_h2_image_bytes(int L) { return (size_t)L * kH2ImgStride; }

// images of L layers (Cr = Cd = 32, fw = 2) into img: one launch
int mfma_layer_pack_h2(int L, const float* const* Wf, const float* const* Wg, const float* const* Wp, void* img,
                       hipStream_t s) {
    WN_CHECK_SHAPE(L >= 1 && L <= 64, "mfma_layer_pack_h2: 1..64 layers per call");
    PackH2Args a{};
    for (int l = 0; l < L; ++l) { a.Wf[l] = Wf[l]; a.Wg[l] = Wg[l]; a.Wp[l] = Wp[l]; }
    hipLaunchKernelGGL(k_layer_pack_h2, dim3(L), dim3(256), 0, s, a, reinterpret_cast<char*>(img));
    WN_LAUNCH_CHECK();
    return WN_OK;
}

// layer l of a packed stack; only the one-tile-per-wave form exists (small launches take the fp32 kernel)
// launches of at least this many workgroups (4 tiles each) take the one-tile-per-wave kernels; smaller ones the looping
// fp32 kernel.  WnExec.fwd_t1_min_blocks overrides the threshold (1 = always, used by the parity tests; < 0 = never).
static int t1_min_blocks() { return exec_fwd_t1_min_blocks(); }
bool mfma_layer_fwd_h2_ok(int B, int T, int t_live) {
    const int tile_lo = t_live > 0 ? t_live / 32 : 0;
    const long long nt = (long long)B * ((T + 31) / 32 - tile_lo);
    return t1_min_blocks() > 0 && nt > 0 && (nt + 3) / 4 >= t1_min_blocks() && nt < (1ll << 31);
}
int mfma_layer_fwd_h2(const float* x, const void* img, int l, float* out, float* z, float* fs, float* gs, int B, int T,
                      int d, int Z, int t_live, hipStream_t s) {
    const int tile_lo = t_live > 0 ? t_live / 32 : 0;
    const int tiles_per_b = (T + 31) / 32 - tile_lo;
    const int ntiles = B * tiles_per_b;
    const int blocks = (ntiles + 3) / 4;
    const char* im = reinterpret_cast<const char*>(img) + (size_t)l * kH2ImgStride;
#define FWDH_LAUNCH(SAVE)                                                                                        \
    hipLaunchKernelGGL((k_layer_fwd_h2_t1<SAVE>), dim3(blocks), dim3(256), 0, s, x, im, out, z, fs, gs, B, T, d, Z, \
                       tile_lo, tiles_per_b, ntiles)
    if (fs) FWDH_LAUNCH(1);
    else if (gs) FWDH_LAUNCH(2);
    else FWDH_LAUNCH(0);
#undef FWDH_LAUNCH
    WN_LAUNCH_CHECK();
    return WN_OK;
}

bool mfma_layer_supported(int Cr, int Cd, int fw) { return Cr == 32 && Cd == 32 && fw == 2; }

// fs / gs: where tanh / sigmoid are saved for the backward: both (any backward), gs only (the chained stack backward,
// which recovers tanh = z / sigmoid), or neither (inference)
int mfma_layer_fwd(const float* x, const float* Wf, const float* bf, const float* Wg, const float* bg,
                   const float* Wp, const float* bp, float* out, float* z, float* fs, float* gs, int B, int T,
                   int d, int Z, int t_live, hipStream_t s) {
    // columns below t_live (a multiple of 32) are not computed: tiles_per_b counts the live tiles of a clip
    const int tile_lo = t_live > 0 ? t_live / 32 : 0;
    const int tiles_per_b = (T + 31) / 32 - tile_lo;
    WN_CHECK_ARG(tiles_per_b > 0, "mfma_layer_fwd: no live column");
    const long long nt = (long long)B * tiles_per_b;
    WN_CHECK_SHAPE(nt < (1ll << 31), "mfma_layer_fwd: too many tiles");
    const int ntiles = (int)nt;
    int blocks = (ntiles + 3) / 4;
    const bool hb = bf || bg || bp;
    // enough workgroups to give every CU two to four of them (8-16 waves): one tile per wave; otherwise the looping kernel.
    // WnExec.fwd_t1_min_blocks overrides the threshold (1 = always, used by the parity tests; < 0 = never).
    const int t1_min = t1_min_blocks();
    if (t1_min > 0 && blocks >= t1_min) {
#define FWD1_LAUNCH(SAVE, BIAS)                                                                              \
    hipLaunchKernelGGL((k_layer_fwd_mfma32_t1<SAVE, BIAS>), dim3(blocks), dim3(256), 0, s, x, Wf, bf, Wg, bg, Wp, bp, \
                       out, z, fs, gs, B, T, d, Z, tile_lo, tiles_per_b, ntiles)
        if (fs && hb) FWD1_LAUNCH(1, true);
        else if (fs) FWD1_LAUNCH(1, false);
        else if (gs && hb) FWD1_LAUNCH(2, true);
        else if (gs) FWD1_LAUNCH(2, false);
        else if (hb) FWD1_LAUNCH(0, true);
        else FWD1_LAUNCH(0, false);
#undef FWD1_LAUNCH
        WN_LAUNCH_CHECK();
        return WN_OK;
    }
    if (blocks > 512) blocks = 512;          // 256 CUs x 2 resident workgroups; waves stride over tiles
#define FWD_LAUNCH(SAVE, BIAS)                                                                               \
    hipLaunchKernelGGL((k_layer_fwd_mfma32<SAVE, BIAS>), dim3(blocks), dim3(256), 0, s, x, Wf, bf, Wg, bg, Wp, bp, \
                       out, z, fs, gs, B, T, d, Z, tile_lo, tiles_per_b, ntiles)
    if (fs && hb) FWD_LAUNCH(1, true);
    else if (fs) FWD_LAUNCH(1, false);
    else if (gs && hb) FWD_LAUNCH(2, true);
    else if (gs) FWD_LAUNCH(2, false);
    else if (hb) FWD_LAUNCH(0, true);
    else FWD_LAUNCH(0, false);
#undef FWD_LAUNCH
    WN_LAUNCH_CHECK();
    return WN_OK;
}

}  // namespace wn
