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
    // WAVENET_HIP_FWD_T1_MIN_BLOCKS overrides the threshold (1 = always, used by the parity tests; 0 = never).
    static const int t1_min = getenv("WAVENET_HIP_FWD_T1_MIN_BLOCKS") ? atoi(getenv("WAVENET_HIP_FWD_T1_MIN_BLOCKS")) : 512;
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
