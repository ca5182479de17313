// bf16x3 variant of the channel GEMM (same contract as k_colgemm in mfma_gemm.hip).
//
// fp32-input MFMA runs at the fp32 vector rate (157 TF); bf16 MFMA runs 16x faster.  Every fp32 operand is
// split into three bf16 parts  x = h + m + l  (h = bf16(x), m = bf16(x-h), l = bf16(x-h-m); |x-h-m-l| <=
// 2^-27 |x|) and a product w*x is evaluated as the six terms of magnitude >= 2^-18:
//        wh*xh + (wh*xm + wm*xh) + (wm*xm + wh*xl + wl*xh)
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms are
// <= 3 * 2^-27 relative, i.e. below fp32 rounding (2^-24): the result is as accurate as the fp32 MFMA path
// (the parity tests hold the same 1e-4 logit / 1e-4 relative-gradient bars) at 16/6 = 2.7x its matrix rate.
//
// The weights are split ONCE per launch by k_split_w into an image that is already in MFMA A-operand order
// (tile = 32 rows x 32 k x {h,m,l} = 6 KB), so the per-chunk LDS fill is a plain copy done by LDS-DMA
// (global_load_lds, no staging registers) into a double buffer: the copy of chunk c+1 lands while chunk c
// computes.  X columns are split in registers right after their (prefetched) float4 loads.
#include "mfma_gemm.hpp"
#include "split_w.hpp"
#include "h2_ops.hpp"
#include <type_traits>

namespace wn {

bool gemm_b3_enabled() { return gemm_mode() >= 1; }
static bool one_term() { return gemm_mode() == 2; }
static bool half2_mode() { return gemm_mode() == 3; }

// the weight tiles of one launch (see split_w.hpp); WMAX: only the largest |w|, into *a.wmax_dev (atomicMax of the bits: a
// positive float orders like an unsigned)
template <bool WMAX>
__global__ void k_split_w(CGArgs a, int mode, int mtiles, int chunks_per_src, __bf16* __restrict__ img, int one) {
    if (WMAX) {
        const float mw = split_w_tile<true>(a, mode, mtiles, chunks_per_src, img, one, (int)blockIdx.x, 1.f);
        if (threadIdx.x == 0) atomicMax(const_cast<unsigned*>(a.wmax_dev), __float_as_uint(mw));
    } else {
        split_w_tile<false>(a, mode, mtiles, chunks_per_src, img, one, (int)blockIdx.x, one == 3 ? h2_scale(a.wmax_dev, kH2ScaleW) : 1.f);
    }
}

__device__ __forceinline__ int b3_ch(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- whole-row global access for tiles in accumulator layout ------------------------------------------------------
// A wave's 32-column x 32-channel tile sits in registers as four float4 per lane: lane (j, h), piece q = channels
// 8q + 4h .. +3 of column j, i.e. 16-byte chunk 2q + h of that column's 128-byte row.  Moved straight between registers and
// memory, one instruction touches 32 bytes of each of 32 rows (measured: 2.7 TB/s for the stores and 3.1 TB/s for the
// loads of the gate-backward epilogue).  Through a 4 KB per-wave LDS patch (chunk c of row r in slot c ^ (r & 7):
// conflict-free in both directions) one instruction moves eight whole 128-byte rows.
struct RowMap { long long row[4]; bool ok[4]; };          // global row / validity of tile row 8 it + lane / 8

__device__ __forceinline__ RowMap row_map(long long no, bool nvalid, int lane) {
    RowMap m;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int src = it * 8 + (lane >> 3);
        const int lo = __shfl((int)(no & 0xffffffffll), src), hi = __shfl((int)(no >> 32), src);
        m.row[it] = ((long long)hi << 32) | (unsigned)lo;
        m.ok[it] = __shfl(nvalid ? 1 : 0, src) != 0;
    }
    return m;
}
__device__ __forceinline__ float* patch_tile(float* patch, int j, int h, int q) {
    return patch + j * 32 + (((2 * q + h) ^ (j & 7)) << 2);
}
__device__ __forceinline__ float* patch_rows(float* patch, int lane, int it) {
    const int r = it * 8 + (lane >> 3), c = lane & 7;
    return patch + r * 32 + ((c ^ (r & 7)) << 2);
}
// global (whole rows) -> registers in tile layout; `stride` floats between rows, `col` = first channel of the tile
__device__ __forceinline__ void rows_load(const float* __restrict__ g, long long stride, int col, const RowMap& m, int lane,
                                          float4 (&v)[4]) {
#pragma unroll
    for (int it = 0; it < 4; ++it)
        v[it] = *reinterpret_cast<const float4*>(g + (m.ok[it] ? m.row[it] : 0) * stride + col + ((lane & 7) << 2));
}
__device__ __forceinline__ void rows_to_tile(float* patch, int lane, const float4 (&v)[4], float4 (&t)[4]) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int it = 0; it < 4; ++it) *reinterpret_cast<float4*>(patch_rows(patch, lane, it)) = v[it];
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = *reinterpret_cast<const float4*>(patch_tile(patch, j, h, q));
}
__device__ __forceinline__ void tile_store_rows(float* patch, int lane, const float4 (&t)[4], float* __restrict__ g,
                                                long long stride, int col, const RowMap& m) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(patch_tile(patch, j, h, q)) = t[q];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const float4 v = *reinterpret_cast<const float4*>(patch_rows(patch, lane, it));
        if (m.ok[it]) *reinterpret_cast<float4*>(g + m.row[it] * stride + col + ((lane & 7) << 2)) = v;
    }
}

// MT m-tiles per workgroup (one per wave for the LDS-DMA fill: MT == 4 waves)
// The activation on X is a template parameter: as a runtime switch it put ~80 branches and ~270 scalar instructions (and
// an inlined expm1f per element) into every 32-deep chunk of the contraction, next to 48 MFMAs.
template <int ACT>
__device__ __forceinline__ float act_apply_t(float x) {
    if (ACT == WN_ACT_RELU) return x > 0.f ? x : 0.f;
    if (ACT == WN_ACT_ELU) return x > 0.f ? x : expm1f(x);
    return x;
}

// MT = 8 (one-term mode only: 128 accumulator registers, 16 KB of LDS per buffer) halves the number of times X is
// re-read when there are 8 or more m-tiles (skip sum, dz, the gate-mode layer GEMM).
// TM: terms of a product -- 6 (bf16 x 3), 1 (bf16), 3 (fp16 x 2: modes 0 and 2 only)
template <int MODE, int ACT, int TM, int MT>
__global__ __launch_bounds__(256, MT == 8 ? 2 : 3) void k_colgemm_b3(CGArgs a, const __bf16* __restrict__ img, int mtiles,
                                                                     int nchunks, int chunks_per_src,
                                                                     const __bf16* __restrict__ img2) {
    constexpr bool ONE = TM == 1;
    constexpr bool H2 = TM == 3;
    const float h2sx = H2 ? h2_scale(a.xmax_dev, kH2ScaleX) : 1.f;
    const float h2sw = H2 ? h2_scale(a.wmax_dev, kH2ScaleW) : 1.f;
    static_assert(MT == 4 || (MT == 8 && (ONE || H2)), "8 m-tiles per workgroup: one-term or fp16-split products");
    static_assert(!H2 || MODE == 0 || MODE == 2 || MODE == 6, "the fp16 split serves the plain contractions only");
    static_assert(MODE != 6 || (H2 && MT == 8), "head + loss mode: fp16 split, all eight m-tiles (256 logits) in one workgroup");
    // MODE 6 (head + loss): X -- relu(skip sum), no known range -- is scaled PER CHUNK by a power of two taken from the wave's own
    // maximum (as the fused layer kernels do), so no range pass and no overflow fallback exist; a chunk's products leave the matrix
    // core in their own scale and join the fp32 logits with one fma per element.
    constexpr bool XENT = MODE == 6;
    constexpr int TB = ONE ? kTileBytes / 3 : (H2 ? kTileBytes * 2 / 3 : kTileBytes);   // bytes of one tile image
    __shared__ __attribute__((aligned(16))) char lds[2 * MT * TB];              // double-buffered tile images
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    // Workgroup -> (column block bx, m-tile group by).  The `ny` groups of one column block read the same X: they get
    // linear ids xcd + 8 * by (+ a multiple of 8 ny), i.e. the SAME XCD at consecutive dispatch slots, so that X comes from
    // HBM once and from that XCD's L2 for the others (with by as the slow grid index the re-reads were HBM reads: the skip
    // sum fetched 1.38 GB for 0.5 GB of z, dz 1.02 GB for 0.1 GB of dskip -- both ran at the HBM limit, 5.2-5.4 TB/s).
    int bx, by;
    {
        const int ny = (mtiles + MT - 1) / MT;
        const int nx = gridDim.x / ny;
        const int id = blockIdx.x;
        const int full = nx & ~7;
        if (id < full * ny) {
            const int g = id / (8 * ny), rem = id - g * 8 * ny;
            bx = g * 8 + (rem & 7);
            by = rem >> 3;
        } else {
            const int t = id - full * ny;
            by = t % ny;
            bx = full + t / ny;
        }
    }
    const int t0 = by * MT;                                // first m-tile (mode 2: first problem) of this workgroup
    const long long n = ((long long)bx * 4 + wave) * 32 + j;
    const bool nvalid = n < a.N;
    long long rb0 = 0;                 // first source row of this column's clip
    int rbase = -(1 << 30);            // row inside the clip before the per-source shift (invalid column: far out)
    long long no = n;                  // output row of this column
    if (nvalid) {
        long long b = n / a.rows_out_per_b;
        rbase = (int)(n - b * a.rows_out_per_b) + a.off;
        rb0 = b * a.rows_src_per_b;
        if (a.out_rows_per_b) no = b * a.out_rows_per_b + a.out_row0 + (n - b * a.rows_out_per_b);
    }
    float ms = 0.f, ms_next = 0.f;     // masks of the chunk being computed / being fetched

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    if (MODE == 3 || MODE == 5) {
        for (int src = 0; src < a.nsrc; ++src)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float* bb = (mt & 1) ? a.bias2[src] : a.bias[src];
                if (bb && t0 + mt < mtiles) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][r] += bb[((t0 + mt) >> 1) * 32 + b3_ch(r, h)];
                }
            }
    }
    if (MODE == 0) {
        for (int src = 0; src < a.nsrc; ++src)
            if (a.bias[src]) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (t0 + mt < mtiles) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[mt][r] += a.bias[src][(t0 + mt) * 32 + b3_ch(r, h)] * (H2 ? h2sw * h2sx : 1.f);
                    }
            }
    }

    // this wave copies m-tiles t0 + wave (+ 4) of a chunk: 1 KB LDS-DMA pieces
    auto dma_chunk = [&](int c, int buf) {
#pragma unroll
        for (int tt = 0; tt < MT / 4; ++tt) {
            const int tl = wave + 4 * tt;
            const int my_tile = (t0 + tl < mtiles) ? t0 + tl : mtiles - 1;
            const char* src = reinterpret_cast<const char*>(img) + ((long long)c * mtiles + my_tile) * TB + lane * 16;
            char* dst = lds + (buf * MT + tl) * TB;
#pragma unroll
            for (int q = 0; q < TB / 1024; ++q)
                __builtin_amdgcn_global_load_lds(src + q * 1024, (__attribute__((address_space(3))) void*)(dst + q * 1024),
                                                 16, 0, 0);
        }
    };
    float4 xr[4];
    auto load_x = [&](int c) {
        const int src = (MODE == 0 || MODE >= 3) ? c / chunks_per_src : 0;          // (mode 6 = mode 0 here)
        const int k0 = ((MODE == 0 || MODE >= 3) ? c - src * chunks_per_src : c) * 32;
        const int rs = rbase + a.soff[src];
        const bool rv = rs >= 0 && rs < a.rows_src_per_b;
        ms_next = rv ? 1.f : 0.f;
        const float* __restrict__ Xb = a.X[src] + (rb0 + (rv ? rs : 0)) * (a.ldx ? a.ldx : a.K[src]) + k0 + 8 * h;
        xr[0] = *reinterpret_cast<const float4*>(Xb);             // k-step 0: channels 8h .. 8h+7
        xr[1] = *reinterpret_cast<const float4*>(Xb + 4);
        xr[2] = *reinterpret_cast<const float4*>(Xb + 16);        // k-step 1: channels 16+8h .. 16+8h+7
        xr[3] = *reinterpret_cast<const float4*>(Xb + 20);
    };

    dma_chunk(0, 0);
    load_x(0);
    for (int c = 0; c < nchunks; ++c) {
        // split this chunk's X columns (the loads were issued one iteration ago)
        ms = ms_next;
        bf16x8 xh[2], xm[2], xl[2];
        float csx = 1.f, cix = 1.f;                       // MODE 6: this chunk's scale (a power of two) and its inverse
        if constexpr (XENT) {
            float mx = 0.f;
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(act_apply_t<ACT>(xr[i4].x)), fabsf(act_apply_t<ACT>(xr[i4].y))),
                                     fmaxf(fabsf(act_apply_t<ACT>(xr[i4].z)), fabsf(act_apply_t<ACT>(xr[i4].w)))));
            mx = lb_wave_max(mx * ms);
            lb_pow2_scale(mx, csx, cix);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float v[8] = {xr[2 * ks].x, xr[2 * ks].y, xr[2 * ks].z, xr[2 * ks].w,
                                xr[2 * ks + 1].x, xr[2 * ks + 1].y, xr[2 * ks + 1].z, xr[2 * ks + 1].w};
            if (H2) {
                f16x8 fh, fm;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 a0, a1;
                    split2h(act_apply_t<ACT>(v[e]) * (ms * (XENT ? csx : h2sx)), a0, a1);
                    fh[e] = a0; fm[e] = a1;
                }
                xh[ks] = __builtin_bit_cast(bf16x8, fh);
                xm[ks] = __builtin_bit_cast(bf16x8, fm);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    __bf16 hh, mm, ll;
                    split3(act_apply_t<ACT>(v[e]) * ms, hh, mm, ll);
                    xh[ks][e] = hh; xm[ks][e] = mm; xl[ks][e] = ll;
                }
            }
        }
        __syncthreads();                 // vmcnt(0): chunk c's image has landed; every wave is done with chunk c-1
        if (c + 1 < nchunks) {           // next chunk: image into the other buffer, X into registers
            dma_chunk(c + 1, (c + 1) & 1);
            load_x(c + 1);
        }
        const char* Ab = lds + ((c & 1) * MT) * TB + lane * 16;
        // The A fragments of m-tile mt + 1 are requested while the MFMAs of m-tile mt run.  (Left to itself hipcc sinks every
        // ds_read next to its first use: "read, s_waitcnt lgkmcnt(0), three MFMAs" 16 times per chunk, one exposed LDS
        // latency per 96 cycles of matrix work -- 48 % matrix utilisation by PMC.  With an LDS-DMA in the loop every wait
        // is lgkmcnt(0), so the order has to be: wait for tile mt's fragments (the empty asm is that use), THEN request
        // tile mt + 1's, then multiply.)
        constexpr bool PIPE = ONE || H2;                 // six terms at three workgroups per CU: no registers for a second set
        bf16x8 fr[PIPE ? 2 : 1][2][3];
        auto ldA = [&](int mt, bf16x8 (&f)[2][3]) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* p = Ab + mt * TB + ks * (TB / 2);
                f[ks][0] = *reinterpret_cast<const bf16x8*>(p);
                if (!ONE) f[ks][1] = *reinterpret_cast<const bf16x8*>(p + 1024);
                if (!ONE && !H2) f[ks][2] = *reinterpret_cast<const bf16x8*>(p + 2048);
            }
        };
        if (PIPE) ldA(0, fr[0]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            bf16x8 (&f)[2][3] = fr[PIPE ? (mt & 1) : 0];
            if (PIPE) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    asm volatile("" ::"v"(f[ks][0]));
                    if (!ONE) asm volatile("" ::"v"(f[ks][1]));
                }
                if (mt + 1 < MT) ldA(mt + 1, fr[PIPE ? ((mt + 1) & 1) : 0]);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                ldA(mt, f);
            }
            f32x16 tacc;                                   // MODE 6: the chunk's product of this m-tile, in the chunk's scale
            if constexpr (XENT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tacc[r] = 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 ah = f[ks][0];
                const bf16x8 am = ONE ? f[ks][0] : f[ks][1];
                const bf16x8 al = (ONE || H2) ? f[ks][0] : f[ks][2];
                if (H2) {                                  // fp16 two-way split: wm xh + wh xm + wh xh
                    const f16x8 fah = __builtin_bit_cast(f16x8, ah), fam = __builtin_bit_cast(f16x8, am);
                    const f16x8 fxh = __builtin_bit_cast(f16x8, xh[ks]), fxm = __builtin_bit_cast(f16x8, xm[ks]);
                    if constexpr (XENT) {
                        tacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fam, fxh, tacc, 0, 0, 0);
                        tacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fxm, tacc, 0, 0, 0);
                        tacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fxh, tacc, 0, 0, 0);
                        continue;
                    }
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fam, fxh, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fxm, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fxh, acc[mt], 0, 0, 0);
                    continue;
                }
                // smallest terms first
                if (!ONE) {
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh[ks], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl[ks], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xm[ks], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xh[ks], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xm[ks], acc[mt], 0, 0, 0);
                }
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[ks], acc[mt], 0, 0, 0);
            }
            if constexpr (XENT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][r] = fmaf(tacc[r], cix, acc[mt][r]);
            }
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (MODE == 5 && MT == 8) {
        // ---- the whole 128/128 layer: gate epilogue, then out = Wp z + bp + x with z taken from registers --------------
        // z leaves phase 1 in accumulator layout (lane (j, h): channels (r & 3) + 8 (r >> 2) + 4 h of each 32-channel tile);
        // the second contraction wants it as B operands (lane (j, h2): 8 consecutive channels 16 ks + 8 h2 ..).  Packed to
        // bf16 pairs, one v_permlane32_swap per dword pair does that exchange between lanes j and j + 32 -- no LDS.
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const bool live = nvalid && (rbase - a.off >= a.gate_Z);
        __syncthreads();                                            // every wave is done with the last gate chunk
        float* patch5 = reinterpret_cast<float*>(lds) + wave * 1024;
        const RowMap rm5 = row_map(no, nvalid, lane);
        bf16x8 zb[4][2];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            unsigned pk[8];
            float4 tz[4], tf[4], ts[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float f4[4], s4[4], z4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f4[e] = fast_tanh(live ? acc[2 * pr][4 * q + e] : 0.f);
                    s4[e] = fast_sigmoid(live ? acc[2 * pr + 1][4 * q + e] : 0.f);
                    z4[e] = f4[e] * s4[e];
                }
                tz[q] = make_float4(z4[0], z4[1], z4[2], z4[3]);
                tf[q] = make_float4(f4[0], f4[1], f4[2], f4[3]);
                ts[q] = make_float4(s4[0], s4[1], s4[2], s4[3]);
                bf16x2 p0, p1;
                p0[0] = (__bf16)z4[0]; p0[1] = (__bf16)z4[1]; p1[0] = (__bf16)z4[2]; p1[1] = (__bf16)z4[3];
                pk[2 * q] = __builtin_bit_cast(unsigned, p0);
                pk[2 * q + 1] = __builtin_bit_cast(unsigned, p1);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * ks], pk[4 * ks + 2], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * ks + 1], pk[4 * ks + 3], false, false);
                u32x4 v;
                v[0] = s0[0]; v[1] = s1[0]; v[2] = s0[1]; v[3] = s1[1];
                zb[pr][ks] = __builtin_bit_cast(bf16x8, v);
            }
            // z, f, s leave as whole rows through this wave's patch (in the image buffers, free since the barrier above)
            tile_store_rows(patch5, lane, tz, a.gate_z, a.M, pr * 32, rm5);
            if (a.gate_f) {
                tile_store_rows(patch5, lane, tf, a.gate_f, a.M, pr * 32, rm5);
                tile_store_rows(patch5, lane, ts, a.gate_s, a.M, pr * 32, rm5);
            }
        }
        __syncthreads();                                            // every wave is done with its patch
        {   // Wp's image: 16 tiles of 2 KB (chunk-major), four per wave
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int tile = wave * 4 + tt;
                const char* src = reinterpret_cast<const char*>(img2) + tile * TB + lane * 16;
                char* dst = lds + tile * TB;
#pragma unroll
                for (int q = 0; q < TB / 1024; ++q)
                    __builtin_amdgcn_global_load_lds(src + q * 1024, (__attribute__((address_space(3))) void*)(dst + q * 1024),
                                                     16, 0, 0);
            }
        }
        f32x16 acc2[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[mt][r] = a.proj_bias ? a.proj_bias[mt * 32 + b3_ch(r, h)] : 0.f;
        __syncthreads();                                            // (drains vmcnt) the image has landed
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 aw = *reinterpret_cast<const bf16x8*>(lds + (c * 4 + mt) * TB + ks * (TB / 2) + lane * 16);
                    acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw, zb[c][ks], acc2[mt], 0, 0, 0);
                }
        __syncthreads();                                            // every wave is done with Wp's image: patches again
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            float4 t[4], v[4], u[4];
            rows_load(a.residual, a.ldo, mt * 32, rm5, lane, v);
            rows_to_tile(patch5, lane, v, u);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                t[q] = make_float4(acc2[mt][4 * q] + u[q].x, acc2[mt][4 * q + 1] + u[q].y, acc2[mt][4 * q + 2] + u[q].z,
                                   acc2[mt][4 * q + 3] + u[q].w);
            tile_store_rows(patch5, lane, t, a.out[0], a.ldo, mt * 32, rm5);
        }
        return;
    }
    if (H2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] *= 1.f / (h2sw * (XENT ? 1.f : h2sx));      // exact: a power of two
    }
    __syncthreads();                                               // the image buffers become the waves' 4 KB row patches
    float* patch = reinterpret_cast<float*>(lds) + wave * 1024;
    const RowMap rm = row_map(no, nvalid, lane);
    if constexpr (XENT) {
        // ---- head + loss: acc[mt][r] (+ bias) = logit 32 mt + b3_ch(r, h) of column j; lanes j and j + 32 hold 128 logits each.
        // Softmax cross-entropy against the column's label as k_softmax_xent computes it (loss row = m + log sum exp(l - m) - l_t;
        // a label outside [0, 256) -- Chainer's ignore label -1 -- gives no loss and a zero gradient); the logits themselves
        // never reach memory: out[0] receives d loss / d logits = (softmax - onehot) / (rows that count).
        const float* bb = a.bias[0];
        if (bb) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][r] += bb[mt * 32 + b3_ch(r, h)];
        }
        const int tg = nvalid ? a.xent_target[no] : -1;
        const bool counts = tg >= 0 && tg < 32 * MT;
        float m = -INFINITY, lt = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                m = fmaxf(m, acc[mt][r]);
                lt = (mt * 32 + b3_ch(r, h) == tg) ? acc[mt][r] : lt;
            }
        m = fmaxf(m, __shfl_xor(m, 32));
        lt += __shfl_xor(lt, 32);                          // the partner lane holds 0 for a label that is not among its channels
        float ssum = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[mt][r] = __expf(acc[mt][r] - m);
                ssum += acc[mt][r];
            }
        ssum += __shfl_xor(ssum, 32);
        // the rows that count: given by the host, or counted on the device from the labels (k_xent_count's integer partials)
        float cnt;
        if (a.xent_n_norm < 0) {
            int cn = lane < a.xent_ncnt ? reinterpret_cast<const int*>(a.xent_loss + kXentPart + kXentBlocks)[lane] : 0;
            for (int o = 32; o >= 1; o >>= 1) cn += __shfl_xor(cn, o);
            cnt = (float)(cn > 0 ? cn : 1);
        } else {
            cnt = (float)a.xent_n_norm;
        }
        const float invN = 1.f / cnt;
        const float sc = counts ? invN / ssum : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float4 t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float d4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q + e;
                    d4[e] = acc[mt][r] * sc - ((counts && mt * 32 + b3_ch(r, h) == tg) ? invN : 0.f);
                }
                t[q] = make_float4(d4[0], d4[1], d4[2], d4[3]);
            }
            tile_store_rows(patch, lane, t, a.out[0], a.ldo, mt * 32, rm);
        }
        // this workgroup's loss sum: columns in lane order (lanes 0..31 carry a column each), waves in order -- a fixed tree
        float rl = (counts && h == 0) ? m + __logf(ssum) - lt : 0.f;
        for (int o = 32; o >= 1; o >>= 1) rl += __shfl_xor(rl, o);
        __syncthreads();                                            // every wave is done with its patch
        float* red = reinterpret_cast<float*>(lds);
        if (lane == 0) red[wave] = rl;
        __syncthreads();
        if (tid == 0) a.xent_loss[kXentPart + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        return;
    }
    if (MODE == 3) {
        const bool live = rbase - a.off >= a.gate_Z;               // rbase - off = t of this column
#pragma unroll
        for (int pr = 0; pr < MT / 2; ++pr) {
            if (t0 + 2 * pr >= mtiles) break;
            const int col = ((t0 >> 1) + pr) * 32;
            float4 tf[4], ts[4], tz[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float f4[4], s4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f4[e] = fast_tanh(live ? acc[2 * pr][4 * q + e] : 0.f);
                    s4[e] = fast_sigmoid(live ? acc[2 * pr + 1][4 * q + e] : 0.f);
                }
                tf[q] = make_float4(f4[0], f4[1], f4[2], f4[3]);
                ts[q] = make_float4(s4[0], s4[1], s4[2], s4[3]);
                tz[q] = make_float4(f4[0] * s4[0], f4[1] * s4[1], f4[2] * s4[2], f4[3] * s4[3]);
            }
            tile_store_rows(patch, lane, tz, a.gate_z, a.M, col, rm);
            if (a.gate_f) {
                tile_store_rows(patch, lane, tf, a.gate_f, a.M, col, rm);
                tile_store_rows(patch, lane, ts, a.gate_s, a.M, col, rm);
            }
        }
        return;
    }
    if (MODE == 4) {
        const bool live = rbase - a.off >= a.gate_Z;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (t0 + mt >= mtiles) break;
            const int col = (t0 + mt) * 32;
            float4 vf[4], vs[4], vr[4], tf[4], ts[4], tr[4], da[4], dg[4];
            rows_load(a.gate_f, a.M, col, rm, lane, vf);                // f, s, dz_skip: row stride M
            rows_load(a.gate_s, a.M, col, rm, lane, vs);
            if (a.residual) rows_load(a.residual, a.M, col, rm, lane, vr);
            rows_to_tile(patch, lane, vf, tf);
            rows_to_tile(patch, lane, vs, ts);
            if (a.residual) rows_to_tile(patch, lane, vr, tr);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 dz = make_float4(acc[mt][4 * q], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]);
                if (a.residual) { dz.x += tr[q].x; dz.y += tr[q].y; dz.z += tr[q].z; dz.w += tr[q].w; }
                if (!live) dz = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 f4 = tf[q], s4 = ts[q];
                da[q] = make_float4(dz.x * s4.x * (1.f - f4.x * f4.x), dz.y * s4.y * (1.f - f4.y * f4.y),
                                    dz.z * s4.z * (1.f - f4.z * f4.z), dz.w * s4.w * (1.f - f4.w * f4.w));
                dg[q] = make_float4(dz.x * f4.x * s4.x * (1.f - s4.x), dz.y * f4.y * s4.y * (1.f - s4.y),
                                    dz.z * f4.z * s4.z * (1.f - s4.z), dz.w * f4.w * s4.w * (1.f - s4.w));
            }
            tile_store_rows(patch, lane, da, a.gate_z, 2 * a.M, col, rm);            // [da | dg]: row stride 2 M
            tile_store_rows(patch, lane, dg, a.gate_z, 2 * a.M, a.M + col, rm);
        }
        return;
    }
    float omax = 0.f;                                              // MODE 0 with a.outmax_dev: max |out| of this wave's tiles
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (t0 + mt >= mtiles) break;
        float* __restrict__ og = (MODE == 2) ? a.out[t0 + mt] : a.out[0];
        const int col = (MODE == 2) ? 0 : (t0 + mt) * 32;
        float4 t[4], v[4], u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = make_float4(acc[mt][4 * q], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]);
        if (MODE == 0 && a.gate_x) {
            rows_load(a.gate_x, a.ldo, col, rm, lane, v);
            rows_to_tile(patch, lane, v, u);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                t[q].x *= act_grad(u[q].x, a.gate_act); t[q].y *= act_grad(u[q].y, a.gate_act);
                t[q].z *= act_grad(u[q].z, a.gate_act); t[q].w *= act_grad(u[q].w, a.gate_act);
            }
        }
        if (MODE == 0 && a.residual) {
            rows_load(a.residual, a.ldo, col, rm, lane, v);
            rows_to_tile(patch, lane, v, u);
#pragma unroll
            for (int q = 0; q < 4; ++q) { t[q].x += u[q].x; t[q].y += u[q].y; t[q].z += u[q].z; t[q].w += u[q].w; }
        }
        if (a.accumulate) {
            rows_load(og, a.ldo, col, rm, lane, v);
            rows_to_tile(patch, lane, v, u);
#pragma unroll
            for (int q = 0; q < 4; ++q) { t[q].x += u[q].x; t[q].y += u[q].y; t[q].z += u[q].z; t[q].w += u[q].w; }
        }
        if (MODE == 0 && a.outmax_dev && nvalid) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                omax = fmaxf(omax, fmaxf(fmaxf(fabsf(t[q].x), fabsf(t[q].y)), fmaxf(fabsf(t[q].z), fabsf(t[q].w))));
        }
        tile_store_rows(patch, lane, t, og, a.ldo, col, rm);
    }
    if (MODE == 0 && a.outmax_dev) {
        // the range of the output for a later call of the step (step plan: the head's dx is the dz contraction's operand): one
        // atomic per wave, and none once the word already holds a larger value (bits of a non-negative float order like an
        // unsigned; a NaN -- bits above every finite value -- sticks, as it must)
        for (int o = 32; o >= 1; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
        if (lane == 0) {
            const unsigned bits = __float_as_uint(omax);
            if (bits > __hip_atomic_load(a.outmax_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.outmax_dev, bits);
        }
    }
}

// =============================================================================================
// k_colgemm_h2q: the fp16-split multi-source contraction (mode 0, eight m-tiles per workgroup: the skip sum) with every
// fetch three HALF-chunks ahead.  DESIGN.md ("what bounds the skip contractions") has the measurements behind each choice:
//   * the loop of k_colgemm_b3 waits ~2 us per chunk for requests issued one chunk ago (and its __syncthreads() drains
//     vmcnt, so the prefetch cannot be deepened there);
//   * one eight-wave workgroup per CU with a deeper prefetch runs all its waves in lock step behind one barrier: here a
//     workgroup is four waves (one per SIMD, 128 columns) and a CU holds two, so that one group's barrier / request phase
//     runs under the other's MFMAs;
//   * a step is HALF a chunk (one 16-deep k-step: 24 MFMAs per wave, 16 KB of image); the image sits in a ring of four such
//     buffers (64 KB per workgroup), X in a ring of four register pairs:
//       step s:  wait (image(s), X(s+1) landed) -> barrier -> request image(s+3) -> 24 MFMAs of half-chunk s on operand
//                set s & 1, the split of X(s+1) into set (s+1) & 1 spread over them -> request X(s+4) into X(s)'s registers
//   * requests are inline asm (invisible to hipcc's counter pass) and retire in issue order; between X(s+1) and the wait of
//     step s lie image(s+1), X(s+2), image(s+2), X(s+3) = 12 requests: `s_waitcnt vmcnt(12)`.  Requests past the end re-fetch
//     the last half-chunk (uniform counts).  nchunks must be even (the loop is unrolled by four: static register indices);
//   * every request is `scalar base + 32-bit lane offset`: with 64-bit vector address arithmetic, a v_readfirstlane + M0
//     write per request and the per-source pointer / shift / stride fetched from the kernel arguments behind
//     `s_waitcnt lgkmcnt(0)`, stamps showed 350-1,100 cycles for issuing the four image requests and 600-1,300 for the two X
//     requests of a 770-cycle half-chunk.  The lane offsets are fixed for the launch, the bases advance by a uniform stride
//     per chunk (the launcher checks that the sources are equally spaced arrays of one chunk each with one row shift -- the
//     layers' z of a stack -- or one array), M0 comes from scalar arithmetic.
// Same results as k_colgemm_b3<0, ., 3, 8>, bit for bit (same products in the same order); config 2's skip sum 0.255 ->
// 0.242 ms.  `dz` (mode 2: 8 chunks per workgroup, prologue and epilogue dominate) measured equal and stays on the older kernel.
// =============================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void asm_load16(f32x4& dst, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
static constexpr int kH2qTB = kTileBytes * 2 / 3;            // 4 KB: one m-tile of an fp16-split chunk ([ks][plane][lane][8])
static constexpr int kH2qHalf = kH2qTB / 2;                  // 2 KB: one k-step of it
static constexpr int kH2qLds = 4 * 8 * kH2qHalf;             // ring of four half-chunk images

__device__ __forceinline__ void asm_load16s(f32x4& dst, unsigned voff, const char* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// two 1 KB pieces of one image tile half: lane offset voff, uniform base sbase, LDS destination m0v (+ 1 KB for the second)
__device__ __forceinline__ void asm_dma2(unsigned voff, const char* sbase, unsigned m0v) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
                 :: "v"(voff), "s"(sbase), "s"(m0v) : "memory", "m0");
}

// x_stride: bytes from chunk c's X to chunk c + 1's (the launcher's check: equally spaced sources of one chunk each, or
// one source: 128)
template <int MODE, int ACT>
__global__ __launch_bounds__(256, 2) void k_colgemm_h2q(CGArgs a, const __bf16* __restrict__ img, int mtiles, int nchunks,
                                                        long long x_stride) {
    constexpr int MT = 8;
    static_assert(MODE == 0 || MODE == 2, "plain contractions only");
    extern __shared__ __attribute__((aligned(16))) char ldsq[];
    const float h2sx = h2_scale(a.xmax_dev, kH2ScaleX);
    const float h2sw = h2_scale(a.wmax_dev, kH2ScaleW);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    int bx, by;
    {   // the m-tile groups of one column block on the same XCD at consecutive slots (as in k_colgemm_b3)
        const int ny = (mtiles + MT - 1) / MT;
        const int nx = gridDim.x / ny;
        const int id = blockIdx.x;
        const int full = nx & ~7;
        if (id < full * ny) {
            const int g = id / (8 * ny), rem = id - g * 8 * ny;
            bx = g * 8 + (rem & 7);
            by = rem >> 3;
        } else {
            const int t = id - full * ny;
            by = t % ny;
            bx = full + t / ny;
        }
    }
    const int t0 = by * MT;
    const long long n = ((long long)bx * 4 + wave) * 32 + j;
    const bool nvalid = n < a.N;
    long long rb0 = 0;
    int rbase = -(1 << 30);
    long long no = n;
    if (nvalid) {
        long long b = n / a.rows_out_per_b;
        rbase = (int)(n - b * a.rows_out_per_b) + a.off;
        rb0 = b * a.rows_src_per_b;
        if (a.out_rows_per_b) no = b * a.out_rows_per_b + a.out_row0 + (n - b * a.rows_out_per_b);
    }
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    if (MODE == 0) {
        for (int src = 0; src < a.nsrc; ++src)
            if (a.bias[src]) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (t0 + mt < mtiles) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][r] += a.bias[src][(t0 + mt) * 32 + b3_ch(r, h)] * (h2sw * h2sx);
                    }
            }
    }
    const int nhalf = 2 * nchunks;
    // this wave copies the half-chunk pieces of m-tiles t0 + wave and t0 + wave + 4: four 1 KB requests
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ldsq;
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    unsigned tile_off[2];              // lane offsets into a chunk's image: this wave's two m-tiles
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
        tile_off[tt] = (unsigned)((t0 + swave + 4 * tt < mtiles) ? t0 + swave + 4 * tt : mtiles - 1) * kH2qTB + lane * 16;
    const long long img_stride = (long long)mtiles * kH2qTB;          // bytes from chunk c's image to chunk c + 1's
    auto dma_half = [&](int s) {
        const int ss = s < nhalf ? s : nhalf - 1;
        const char* base = reinterpret_cast<const char*>(img) + (long long)(ss >> 1) * img_stride + (ss & 1) * kH2qHalf;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
            asm_dma2(tile_off[tt], base, lds0 + ((s & 3) * MT + swave + 4 * tt) * kH2qHalf);
    };
    f32x4 xr[4][2];                    // X(s) in xr[s & 3]: 8 consecutive k of this lane's column
    // one row shift for all sources: the lane's row, its validity and its byte offset are fixed for the launch
    const int rs0 = rbase + a.soff[0];
    const bool rv0 = rs0 >= 0 && rs0 < a.rows_src_per_b;
    const float msk0 = rv0 ? 1.f : 0.f;
    const unsigned x_off = (unsigned)(((rb0 + (rv0 ? rs0 : 0)) * (a.ldx ? a.ldx : a.K[0]) + 8 * h) * 4);
    const char* x0 = reinterpret_cast<const char*>(a.X[0]);
    auto load_x = [&](int s, auto slot_tag) {
        constexpr int slot = decltype(slot_tag)::value;
        const int ss = s < nhalf ? s : nhalf - 1;
        const char* base = x0 + (long long)(ss >> 1) * x_stride + (ss & 1) * 64;
        asm_load16s(xr[slot][0], x_off, base);
        asm_load16s(xr[slot][1], x_off + 16, base);
    };
    f16x8 xh[2], xm[2];                // operand sets by half-chunk parity
    auto split_one = [&](const f32x4 (&raw)[2], float ms, int e, f16x8& oh, f16x8& om) {
        const float v = raw[e >> 2][e & 3];
        _Float16 a0, a1;
        split2h(act_apply_t<ACT>(v) * (ms * h2sx), a0, a1);
        oh[e] = a0; om[e] = a1;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    load_x(0, I0{});
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(xr[0][0]), "+v"(xr[0][1]) : : "memory");
#pragma unroll
    for (int e = 0; e < 8; ++e) split_one(xr[0], msk0, e, xh[0], xm[0]);
    asm volatile("" : "+v"(xh[0]), "+v"(xm[0]));
    dma_half(0); load_x(1, I1{});
    dma_half(1); load_x(2, I2{});
    dma_half(2); load_x(3, I3{});

    auto step = [&](int s, auto p_tag) {
        constexpr int p = decltype(p_tag)::value;          // s & 3
        constexpr int set = p & 1, nxt = 1 - set;
        constexpr int rn = (p + 1) & 3;                     // raw slot of X(s + 1)
        asm volatile("s_waitcnt vmcnt(12)" : "+v"(xr[rn][0]), "+v"(xr[rn][1]) : : "memory");
        const float ms = msk0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // image(s) complete; every wave is done with half-chunk s - 1
        dma_half(s + 3);
        const char* Ab = ldsq + (p * MT) * kH2qHalf + lane * 16;
        f16x8 fr[2][2];
        auto ldA = [&](int mt, f16x8 (&f)[2]) {
            f[0] = *reinterpret_cast<const f16x8*>(Ab + mt * kH2qHalf);
            f[1] = *reinterpret_cast<const f16x8*>(Ab + mt * kH2qHalf + 1024);
        };
        ldA(0, fr[0]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f16x8 (&f)[2] = fr[mt & 1];
            asm volatile("" ::"v"(f[0]));
            asm volatile("" ::"v"(f[1]));
            if (mt + 1 < MT) ldA(mt + 1, fr[(mt + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            split_one(xr[rn], ms, mt, nxt ? xh[1] : xh[0], nxt ? xm[1] : xm[0]);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], xh[set], acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], xm[set], acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], xh[set], acc[mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the split values are complete before the raw registers of X(s) (split a step ago) take the next request
        asm volatile("" : "+v"(xh[nxt]), "+v"(xm[nxt]));
        load_x(s + 4, p_tag);
    };
    for (int s = 0; s < nhalf; s += 4) {
        step(s, I0{});
        step(s + 1, I1{});
        step(s + 2, I2{});
        step(s + 3, I3{});
    }
    // the requests past the end are still in flight (into registers that stay allocated and into the ring)
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(xr[0][0]), "+v"(xr[0][1]), "+v"(xr[1][0]), "+v"(xr[1][1]), "+v"(xr[2][0]), "+v"(xr[2][1]),
                   "+v"(xr[3][0]), "+v"(xr[3][1])
                 :
                 : "memory");
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] *= 1.f / (h2sw * h2sx);                 // exact: a power of two
    __syncthreads();                                               // the ring becomes the waves' 4 KB row patches
    float* patch = reinterpret_cast<float*>(ldsq) + wave * 1024;
    const RowMap rm = row_map(no, nvalid, lane);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (t0 + mt >= mtiles) break;
        float* __restrict__ og = (MODE == 2) ? a.out[t0 + mt] : a.out[0];
        const int col = (MODE == 2) ? 0 : (t0 + mt) * 32;
        float4 t[4], v[4], u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = make_float4(acc[mt][4 * q], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]);
        if (MODE == 0 && a.gate_x) {
            rows_load(a.gate_x, a.ldo, col, rm, lane, v);
            rows_to_tile(patch, lane, v, u);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                t[q].x *= act_grad(u[q].x, a.gate_act); t[q].y *= act_grad(u[q].y, a.gate_act);
                t[q].z *= act_grad(u[q].z, a.gate_act); t[q].w *= act_grad(u[q].w, a.gate_act);
            }
        }
        if (MODE == 0 && a.residual) {
            rows_load(a.residual, a.ldo, col, rm, lane, v);
            rows_to_tile(patch, lane, v, u);
#pragma unroll
            for (int q = 0; q < 4; ++q) { t[q].x += u[q].x; t[q].y += u[q].y; t[q].z += u[q].z; t[q].w += u[q].w; }
        }
        if (a.accumulate) {
            rows_load(og, a.ldo, col, rm, lane, v);
            rows_to_tile(patch, lane, v, u);
#pragma unroll
            for (int q = 0; q < 4; ++q) { t[q].x += u[q].x; t[q].y += u[q].y; t[q].z += u[q].z; t[q].w += u[q].w; }
        }
        tile_store_rows(patch, lane, t, og, a.ldo, col, rm);
    }
}

int launch_colgemm_b3(CGArgs& a, int mode, int nprob, hipStream_t s) {
    if (mode != 0 && mode != 2 && mode != 3 && mode != 4 && mode != 5 && mode != 6) return WN_ESHAPE;
    if (mode == 6 && !(half2_mode() && a.M == 256 && a.nsrc == 1 && a.xent_target && a.xent_loss && a.out[0] &&
                       cdiv(a.N, 128) <= kXentBlocks))
        return WN_ESHAPE;
    int mtiles, nchunks, cps;
    if (mode == 0 || mode >= 3) {
        if (a.M % 32) return WN_ESHAPE;
        if ((mode == 3 || mode == 5) && a.nsrc > WN_GATE_TAPS) return WN_ESHAPE;
        if (mode == 5 && !(one_term() && a.M == 128 && a.ldo == 128)) return WN_ESHAPE;
        for (int i = 0; i < a.nsrc; ++i)
            if (a.K[i] != a.K[0] || a.K[i] % 32) return WN_ESHAPE;
        mtiles = ((mode == 3 || mode == 5) ? 2 : 1) * a.M / 32;
        cps = a.K[0] / 32;
        nchunks = a.nsrc * cps;
    } else {
        if (a.M != 32 || a.K[0] % 32) return WN_ESHAPE;
        mtiles = nprob;
        cps = a.K[0] / 32;
        nchunks = cps;
    }
    const bool one = one_term();
    // fp16 split: plain contractions whose operands the caller declared range-safe (forward activations); else six terms
    const bool h2 = half2_mode() && ((a.h2_ok || a.xmax_dev) && (mode == 0 || mode == 2) || mode == 6);
    const size_t bytes = (size_t)nchunks * mtiles * (one ? kTileBytes / 3 : (h2 ? kTileBytes * 2 / 3 : kTileBytes));
    const size_t bytes2 = mode == 5 ? (size_t)16 * (kTileBytes / 3) : 0;        // Wp's image behind the gate image
    // a READY step plan holds this launch's image (and its range word) already: wn_plan_prepare built them at the start of the
    // step; a recording plan notes the job, and this call prepares its own image as ever (plan.hip)
    const __bf16* pimg = nullptr; const unsigned* pwmax = nullptr;
    const bool planned = plan_split_image(a, mode, mtiles, cps, nchunks, one ? 1 : (h2 ? 3 : 0), bytes, &pimg, &pwmax);
    __bf16* img = planned ? const_cast<__bf16*>(pimg)
                          : reinterpret_cast<__bf16*>(exec_scratch(bytes + bytes2, "the split weight image"));
    if (!img) return WN_EARG;
    if (planned) {
        a.wmax_dev = pwmax;
    } else {
    if (h2) {
        // the weights' own range: |w| <= 2^7 was an assumption (a weight above ~254 saturated the fp16 parts silently); one
        // pass of the same grid over the weight tiles measures it, per entry-point call and weight set
        bool fresh = false;
        // keyed by the first weight array AND the launch's form: a second launch of the same call that starts at the same
        // array but covers another weight set (more sources, another mode) must measure its own maximum
        unsigned* wm = exec_word(reinterpret_cast<const char*>(a.W[0]) + (((unsigned)mode & 7u) << 8 | ((unsigned)a.nsrc & 255u)), &fresh, s);
        if (!wm) return WN_EARG;
        a.wmax_dev = wm;
        if (fresh) hipLaunchKernelGGL(k_split_w<true>, dim3(nchunks * mtiles), dim3(256), 0, s, a, mode, mtiles, cps, img, 3);
    }
    hipLaunchKernelGGL(k_split_w<false>, dim3(nchunks * mtiles), dim3(256), 0, s, a, mode, mtiles, cps, img, one ? 1 : (h2 ? 3 : 0));
    }
    if (mode == 5) {
        if (!a.proj_W || !a.residual || !a.gate_z || a.act != WN_ACT_NONE || mtiles != 8) {
            wn::set_error("colgemm_b3: fused-layer mode needs Wp, the residual input, z and 128 gate channels");
            return WN_EARG;
        }
        CGArgs b{};
        b.nsrc = 1; b.W[0] = a.proj_W; b.wsm[0] = 128; b.wsk = 1; b.K[0] = 128; b.M = 128;
        __bf16* img2 = reinterpret_cast<__bf16*>(reinterpret_cast<char*>(img) + bytes);
        hipLaunchKernelGGL(k_split_w<false>, dim3(16), dim3(256), 0, s, b, 0, 4, 4, img2, 1);
        hipLaunchKernelGGL((k_colgemm_b3<5, WN_ACT_NONE, 1, 8>), dim3(cdiv(a.N, 128)), dim3(256), 0, s, a,
                           (const __bf16*)img, mtiles, nchunks, cps, (const __bf16*)img2);
        WN_LAUNCH_CHECK();
        return WN_OK;
    }
    // (mode 2 with X STATIONARY -- k_dz_xs, round 6: a wave splits its 32 columns of dskip once and walks all 40 layers' tiles,
    // weight tiles through a 3/4-slot LDS ring with counted waits, the row stores left in flight -- was built, bit-identical, and
    // measured: HBM fetch 179 -> 131 MB, the call - 2 ... - 6 % stand-alone, the training step +- 0 (2.888 against 2.882 ms, same
    // box); without its stores it still took 180 of 225 us against 84 us of MFMA time: neither the writes nor the re-split bound
    // this contraction.  Not kept.  DESIGN.md, round 6.)
    const bool mt8 = (one || h2) && mtiles >= 8;
    dim3 grid(cdiv(a.N, 128) * cdiv(mtiles, mt8 ? 8 : 4));
    if (mode == 6) {
#define X_LAUNCH(ACT_) hipLaunchKernelGGL((k_colgemm_b3<6, ACT_, 3, 8>), grid, dim3(256), 0, s, a, (const __bf16*)img, mtiles,   \
                                          nchunks, cps, (const __bf16*)nullptr)
        if (a.act == WN_ACT_RELU) X_LAUNCH(WN_ACT_RELU);
        else if (a.act == WN_ACT_ELU) X_LAUNCH(WN_ACT_ELU);
        else X_LAUNCH(WN_ACT_NONE);
#undef X_LAUNCH
        WN_LAUNCH_CHECK();
        return WN_OK;
    }
#define CG_LAUNCH(MODE_, ACT_)                                                                                          \
    do {                                                                                                                \
        if (mt8) hipLaunchKernelGGL((k_colgemm_b3<MODE_, ACT_, 1, 8>), grid, dim3(256), 0, s, a, (const __bf16*)img,    \
                                    mtiles, nchunks, cps, (const __bf16*)nullptr);                                      \
        else if (one) hipLaunchKernelGGL((k_colgemm_b3<MODE_, ACT_, 1, 4>), grid, dim3(256), 0, s, a,                   \
                                         (const __bf16*)img, mtiles, nchunks, cps, (const __bf16*)nullptr);             \
        else hipLaunchKernelGGL((k_colgemm_b3<MODE_, ACT_, 6, 4>), grid, dim3(256), 0, s, a, (const __bf16*)img,        \
                                mtiles, nchunks, cps, (const __bf16*)nullptr);                                          \
    } while (0)
#define CG_LAUNCH_H2(MODE_, ACT_)                                                                                       \
    do {                                                                                                                \
        if (mt8) hipLaunchKernelGGL((k_colgemm_b3<MODE_, ACT_, 3, 8>), grid, dim3(256), 0, s, a, (const __bf16*)img,    \
                                    mtiles, nchunks, cps, (const __bf16*)nullptr);                                      \
        else hipLaunchKernelGGL((k_colgemm_b3<MODE_, ACT_, 3, 4>), grid, dim3(256), 0, s, a, (const __bf16*)img,        \
                                mtiles, nchunks, cps, (const __bf16*)nullptr);                                          \
    } while (0)
    // the pipelined kernel takes sources that are equally spaced arrays of one chunk each with one row shift (the z of a
    // stack's layers), or a single source
    long long x_stride = 128;
    bool lean = h2 && mt8 && mode == 0 && (nchunks & 1) == 0 && !exec_flag(WN_EXEC_NO_PIPELINED_GEMM) &&
                (a.N / a.rows_out_per_b + 1) * (long long)a.rows_src_per_b * (a.ldx ? a.ldx : a.K[0]) * 4 < (1ll << 32);   // 32-bit lane offsets
    if (lean && mode == 0 && a.nsrc > 1) {
        x_stride = (const char*)a.X[1] - (const char*)a.X[0];
        lean = cps == 1;
        for (int i = 1; i < a.nsrc && lean; ++i)
            lean = a.soff[i] == a.soff[0] && a.K[i] == a.K[0] && (const char*)a.X[i] - (const char*)a.X[i - 1] == x_stride;
    } else if (lean && mode == 0) {
        lean = a.nsrc == 1;
    }
    if (lean) {
        static bool attr = false;
        if (!attr) {
#define Q_ATTR(MODE_, ACT_)                                                                                   \
    WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_colgemm_h2q<MODE_, ACT_>),                      \
                               hipFuncAttributeMaxDynamicSharedMemorySize, kH2qLds))
            Q_ATTR(0, WN_ACT_NONE); Q_ATTR(0, WN_ACT_RELU); Q_ATTR(0, WN_ACT_ELU);
#undef Q_ATTR
            attr = true;
        }
#define Q_LAUNCH(MODE_, ACT_)                                                                                 \
    hipLaunchKernelGGL((k_colgemm_h2q<MODE_, ACT_>), grid, dim3(256), kH2qLds, s, a, (const __bf16*)img, mtiles, nchunks, x_stride)
        if (a.act == WN_ACT_RELU) Q_LAUNCH(0, WN_ACT_RELU);
        else if (a.act == WN_ACT_ELU) Q_LAUNCH(0, WN_ACT_ELU);
        else Q_LAUNCH(0, WN_ACT_NONE);
#undef Q_LAUNCH
        WN_LAUNCH_CHECK();
        return WN_OK;
    }
    if (h2) {
        if (mode == 2) CG_LAUNCH_H2(2, WN_ACT_NONE);
        else if (a.act == WN_ACT_RELU) CG_LAUNCH_H2(0, WN_ACT_RELU);
        else if (a.act == WN_ACT_ELU) CG_LAUNCH_H2(0, WN_ACT_ELU);
        else CG_LAUNCH_H2(0, WN_ACT_NONE);
        WN_LAUNCH_CHECK();
        return WN_OK;
    }
#undef CG_LAUNCH_H2
    if (mode == 3) {
        if (a.act != WN_ACT_NONE || !a.gate_z) { wn::set_error("colgemm_b3: gate mode takes no activation and needs gate_z"); return WN_EARG; }
        CG_LAUNCH(3, WN_ACT_NONE);
    } else if (mode == 4) {
        if (a.act != WN_ACT_NONE || !a.gate_z || !a.gate_f || !a.gate_s) { wn::set_error("colgemm_b3: gate-backward mode needs f, s and the output"); return WN_EARG; }
        CG_LAUNCH(4, WN_ACT_NONE);
    } else if (mode == 0) {
        // (k_colgemm_b3's epilogue is the one that fills a.outmax_dev: only now may the plan hand the word to a consumer)
        if (a.outmax_dev) plan_xmax_written(a.out[0]);
        if (a.act == WN_ACT_RELU) CG_LAUNCH(0, WN_ACT_RELU);
        else if (a.act == WN_ACT_ELU) CG_LAUNCH(0, WN_ACT_ELU);
        else if (a.act == WN_ACT_NONE) CG_LAUNCH(0, WN_ACT_NONE);
        else { wn::set_error("colgemm_b3: unknown activation %d", a.act); return WN_EARG; }
    } else {
        if (a.act != WN_ACT_NONE) { wn::set_error("colgemm_b3: multi-problem mode takes no activation"); return WN_EARG; }
        CG_LAUNCH(2, WN_ACT_NONE);
    }
#undef CG_LAUNCH
    WN_LAUNCH_CHECK();
    return WN_OK;
}

// =============================================================================================
// bf16x3 weight gradient:  dW_p[m][k] += sum_n A[n][m] * act(B_p[row(n)][k])   (contraction over time rows)
// MFMA K = time (16 rows per instruction), so both operands are needed "8 consecutive rows per lane":
//   * A (shared by the 4 waves): every thread builds whole lane fragments -- 8 rows of one channel, eight
//     coalesced dword loads -- splits them into h/m/l and stores each with ONE ds_write_b128 into the same
//     A-operand image layout as above; raw loads of chunk c+1 are in flight during the MFMAs of chunk c;
//   * B (one 32-channel tile per wave): eight dword loads per lane per k-step, split in registers.
// =============================================================================================
template <int MT, bool HAS_B2, int ACT, bool ONE>
__global__ __launch_bounds__(256, 2) void k_wgrad_b3(WGArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[MT * kTileBytes];          // one 32-row chunk of A, split
    constexpr int NF = MT / 2 > 0 ? MT / 2 : 1;             // A fragments per thread per chunk (MT*128 / 256)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int p = blockIdx.y * 4 + wv;
    const bool active = p < a.nprob;
    const int m0 = blockIdx.z * (MT * 32);
    const int b = blockIdx.x / a.wgs_per_b;
    const int r_begin = (blockIdx.x - b * a.wgs_per_b) * a.rows_per_wg;
    const int r_end = min(a.rows_A_per_b, r_begin + a.rows_per_wg);
    const float* __restrict__ Ab = a.A + ((long long)b * a.rows_A_per_b) * a.lda + m0;
    const float* __restrict__ Bb = a.Bp[active ? p : 0] + ((long long)b * a.rows_B_per_b + a.off) * a.ldb + j;
    const float* __restrict__ B2b = a.B2p[active ? p : 0] ? a.B2p[active ? p : 0] + ((long long)b * a.rows_B_per_b + a.off) * a.ldb + j : nullptr;
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // loads only, never under a runtime condition (see k_wgrad_b3w): activation, the B2 factor and the edge masks are
    // applied at split time on registers
    float ar[NF][8], br[2][8], b2r[2][8];
    const float* __restrict__ B2s = HAS_B2 ? B2b : Bb;
    const float amask = active ? 1.f : 0.f;
    // per-thread fragment coordinates are chunk-invariant
    int fm[NF], fg[NF];
#pragma unroll
    for (int it = 0; it < NF; ++it) {
        const int f = it * 256 + tid;                        // fragment: channel m, row group g = 2*ks + hh
        fm[it] = f % (MT * 32);
        fg[it] = f / (MT * 32);
    }
    auto issue = [&](int r0) {
        const bool full = r0 + 32 <= r_end && r0 + a.off >= 0 && r0 + 31 + a.off < a.rows_B_per_b;
        if (full) {
#pragma unroll
            for (int it = 0; it < NF; ++it) {
                const float* ap = Ab + (long long)(r0 + 8 * (fg[it] & 3)) * a.lda + fm[it];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) ar[it][jj] = ap[(long long)jj * a.lda];
            }
            const float* bp = Bb + (long long)(r0 + 8 * h) * a.ldb;
            const float* b2p = B2s + (long long)(r0 + 8 * h) * a.ldb;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    br[ks][jj] = bp[(long long)(16 * ks + jj) * a.ldb];
                    if (HAS_B2) b2r[ks][jj] = b2p[(long long)(16 * ks + jj) * a.ldb];
                }
            return;
        }
#pragma unroll
        for (int it = 0; it < NF; ++it)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int r = r0 + 8 * (fg[it] & 3) + jj;
                const int rc = r < r_end ? r : r_end - 1;    // clamped row, masked at split time
                ar[it][jj] = Ab[(long long)rc * a.lda + fm[it]];
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int r = r0 + 16 * ks + 8 * h + jj;
                int rc = r < r_end ? r : r_end - 1;
                if (rc + a.off < 0) rc = -a.off;
                if (rc + a.off >= a.rows_B_per_b) rc = a.rows_B_per_b - 1 - a.off;
                br[ks][jj] = Bb[(long long)rc * a.ldb];
                if (HAS_B2) b2r[ks][jj] = B2s[(long long)rc * a.ldb];
            }
    };
    if (r_begin < r_end) issue(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += 32) {
        // split this chunk's operands (loads were issued one iteration ago)
        bf16x8 ah[NF], am[NF], al[NF], bh[2], bm[2], bl[2];
        const bool edge = !(r0 + 32 <= r_end && r0 + a.off >= 0 && r0 + 31 + a.off < a.rows_B_per_b);
#pragma unroll
        for (int it = 0; it < NF; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float av = ar[it][e];
                if (edge) av *= (r0 + 8 * (fg[it] & 3) + e) < r_end ? 1.f : 0.f;
                __bf16 x0, x1, x2;
                split3(av, x0, x1, x2);
                ah[it][e] = x0; am[it][e] = x1; al[it][e] = x2;
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float bv = act_apply_t<ACT>(br[ks][e]) * amask;
                if (HAS_B2) bv *= b2r[ks][e];
                if (edge) {
                    const int rb = r0 + 16 * ks + 8 * h + e;
                    bv *= (rb < r_end && rb + a.off >= 0 && rb + a.off < a.rows_B_per_b) ? 1.f : 0.f;
                }
                __bf16 x0, x1, x2;
                split3(bv, x0, x1, x2);
                bh[ks][e] = x0; bm[ks][e] = x1; bl[ks][e] = x2;
            }
        __syncthreads();                                     // the previous chunk's MFMAs are done with the image
#pragma unroll
        for (int it = 0; it < NF; ++it) {
            const int f = it * 256 + tid;
            const int m = f % (MT * 32), g = f / (MT * 32);
            if (MT >= 2 || g < 4) {
                char* d = lds + (m >> 5) * kTileBytes + ((g >> 1) * 3) * 1024 + ((m & 31) + 32 * (g & 1)) * 16;
                *reinterpret_cast<bf16x8*>(d) = ah[it];
                *reinterpret_cast<bf16x8*>(d + 1024) = am[it];
                *reinterpret_cast<bf16x8*>(d + 2048) = al[it];
            }
        }
        __syncthreads();
        if (r0 + 32 < r_end) issue(r0 + 32);                 // in flight during the MFMAs below
        const char* Al = lds + lane * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* q = Al + mt * kTileBytes + ks * 3 * 1024;
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(q);
                const bf16x8 xm = *reinterpret_cast<const bf16x8*>(q + 1024);
                const bf16x8 xl = *reinterpret_cast<const bf16x8*>(q + 2048);
                if (!ONE) {
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh[ks], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl[ks], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bm[ks], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bh[ks], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bm[ks], acc[mt], 0, 0, 0);
                }
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh[ks], acc[mt], 0, 0, 0);
            }
        }
    }
    if (!active) return;
    float* __restrict__ o = a.out[p];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            atomicAdd(o + (long long)(m0 + mt * 32 + b3_ch(r, h)) * a.ldo + (long long)j * (a.osk ? a.osk : 1), acc[mt][r]);
}

// ---------------------------------------------------------------------------------------------
// Wide weight-gradient block: 8 waves, all 256 rows of A (8 m-tiles) x 8 problems of 32 columns per workgroup.
// k_wgrad_b3<4> (4 waves, 128 x 128) spends 75 % of its wave-cycles waiting (PMC: SQ_WAIT_ANY; the matrix pipe is 17 %
// busy): two barriers per 32-row chunk and ~260 split/pack VALU instructions per wave against 48 MFMAs.  Here a chunk
// of A is split ONCE per 256 x 256 block (16 values per thread), the image is double buffered in LDS (96 KB, one
// barrier per chunk), and every wave runs 96 MFMAs per chunk on its own problem's B values.
// Grid: x = batch * row slabs, y = ceil(nprob / 8); M == 256.
// ---------------------------------------------------------------------------------------------
template <bool HAS_B2, int ACT, int TM>
__global__ __launch_bounds__(512, 1) void k_wgrad_b3w(WGArgs a) {
    constexpr int MT = 8;
    constexpr bool ONE = TM == 1;
    constexpr bool H2 = TM == 3;
    const float h2sa = H2 ? h2_scale(a.amax_dev, 1.f) : 1.f;
    const float h2sb = H2 ? h2_scale(a.bmax_dev, kH2ScaleX) : 1.f;
    extern __shared__ __attribute__((aligned(16))) char ldsw[];               // 2 x MT x 6 KB
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int p = blockIdx.y * 8 + wv;
    const bool active = p < a.nprob;
    const int off = a.off + a.offp[active ? p : 0];            // this wave's problem may carry its own tap shift
    const int b = blockIdx.x / a.wgs_per_b;
    const int r_begin = (blockIdx.x - b * a.wgs_per_b) * a.rows_per_wg;
    const int r_end = min(a.rows_A_per_b, r_begin + a.rows_per_wg);
    const float* __restrict__ Ab = a.A + ((long long)b * a.rows_A_per_b) * a.lda;
    const float* __restrict__ Bb = a.Bp[active ? p : 0] + ((long long)b * a.rows_B_per_b + off) * a.ldb + j;
    const float* __restrict__ B2b = a.B2p[active ? p : 0] ? a.B2p[active ? p : 0] + ((long long)b * a.rows_B_per_b + off) * a.ldb + j : nullptr;
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // A fragments of this thread: channel fm, row groups g = 2*ks + hh for (ks, hh) = (it, tid >> 8)
    const int fm = tid & 255;
    const int fhh = tid >> 8;
    // Loads only -- no activation, no masks, no runtime-conditional loads: a uniform `if` around a load (the optional
    // B2 factor, `active ? load : 0`, the activation switch) makes hipcc branch and drain vmcnt PER ELEMENT, which
    // serialised the sixteen B loads of every chunk (0.44 ms of this kernel's 0.76 ms at config 2).  Rows are clamped
    // in the edge chunk; everything else happens at split time on registers.
    float ar[2][8], br[2][8], b2r[2][8];
    const float* __restrict__ B2s = HAS_B2 ? B2b : Bb;
    auto issue = [&](int r0) {
        const bool full = r0 + 32 <= r_end && r0 + off >= 0 && r0 + 31 + off < a.rows_B_per_b;
        if (full) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float* ap = Ab + (long long)(r0 + 16 * ks + 8 * fhh) * a.lda + fm;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) ar[ks][jj] = ap[(long long)jj * a.lda];
            }
            const float* bp = Bb + (long long)(r0 + 8 * h) * a.ldb;
            const float* b2p = B2s + (long long)(r0 + 8 * h) * a.ldb;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    br[ks][jj] = bp[(long long)(16 * ks + jj) * a.ldb];
                    if (HAS_B2) b2r[ks][jj] = b2p[(long long)(16 * ks + jj) * a.ldb];
                }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int r = r0 + 16 * ks + 8 * fhh + jj;
                const int rc = r < r_end ? r : r_end - 1;
                ar[ks][jj] = Ab[(long long)rc * a.lda + fm];
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int r = r0 + 16 * ks + 8 * h + jj;
                int rc = r < r_end ? r : r_end - 1;
                if (rc + off < 0) rc = -off;
                if (rc + off >= a.rows_B_per_b) rc = a.rows_B_per_b - 1 - off;
                br[ks][jj] = Bb[(long long)rc * a.ldb];
                if (HAS_B2) b2r[ks][jj] = B2s[(long long)rc * a.ldb];
            }
    };
    const float amask = active ? 1.f : 0.f;
    float csum = 0.f;                    // six-term form: sum over this slab's rows of A[.][fm] (rows 8 fhh .. of every 16)
    if (r_begin < r_end) issue(r_begin);
    int c = 0;
    for (int r0 = r_begin; r0 < r_end; r0 += 32, ++c) {
        char* buf = ldsw + (c & 1) * (MT * kTileBytes);
        bf16x8 bh[2], bm[2], bl[2];
        const bool edge = !(r0 + 32 <= r_end && r0 + off >= 0 && r0 + 31 + off < a.rows_B_per_b);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah, am, al;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float av = ar[ks][e];
                float bv = act_apply_t<ACT>(br[ks][e]) * amask;
                if (HAS_B2) bv *= b2r[ks][e];
                if (edge) {                              // rows beyond the slab / outside B's clip contribute nothing
                    const int ra = r0 + 16 * ks + 8 * fhh + e;
                    const int rb = r0 + 16 * ks + 8 * h + e;
                    av *= ra < r_end ? 1.f : 0.f;
                    bv *= (rb < r_end && rb + off >= 0 && rb + off < a.rows_B_per_b) ? 1.f : 0.f;
                }
                if (TM == 6 && !HAS_B2) csum += av;
                if (H2) {                                // fp16 two-way split of the scaled operands (bit patterns travel as bf16x8)
                    _Float16 y0, y1;
                    split2h(av * h2sa, y0, y1);
                    ah[e] = __builtin_bit_cast(__bf16, y0); am[e] = __builtin_bit_cast(__bf16, y1);
                    split2h(bv * h2sb, y0, y1);
                    bh[ks][e] = __builtin_bit_cast(__bf16, y0); bm[ks][e] = __builtin_bit_cast(__bf16, y1);
                } else {
                    __bf16 x0, x1, x2;
                    split3(av, x0, x1, x2);
                    ah[e] = x0; am[e] = x1; al[e] = x2;
                    split3(bv, x0, x1, x2);
                    bh[ks][e] = x0; bm[ks][e] = x1; bl[ks][e] = x2;
                }
            }
            // image: tile fm>>5, [ks][comp][lane = (fm & 31) + 32*fhh][8]
            char* d = buf + (fm >> 5) * kTileBytes + (ks * 3) * 1024 + ((fm & 31) + 32 * fhh) * 16;
            *reinterpret_cast<bf16x8*>(d) = ah;
            *reinterpret_cast<bf16x8*>(d + 1024) = am;
            if (!H2) *reinterpret_cast<bf16x8*>(d + 2048) = al;
        }
        __syncthreads();                 // the image of this chunk is complete (the buffer written next was read two chunks ago)
        if (r0 + 32 < r_end) issue(r0 + 32);                 // in flight during the MFMAs below
        const char* Al = buf + lane * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* q = Al + mt * kTileBytes + ks * 3 * 1024;
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(q);
                const bf16x8 xm = *reinterpret_cast<const bf16x8*>(q + 1024);
                if (H2) {
                    const f16x8 fah = __builtin_bit_cast(f16x8, xh), fam = __builtin_bit_cast(f16x8, xm);
                    const f16x8 fbh = __builtin_bit_cast(f16x8, bh[ks]), fbm = __builtin_bit_cast(f16x8, bm[ks]);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fam, fbh, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fbm, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fbh, acc[mt], 0, 0, 0);
                    continue;
                }
                const bf16x8 xl = *reinterpret_cast<const bf16x8*>(q + 2048);
                if (!ONE) {
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh[ks], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl[ks], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bm[ks], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bh[ks], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bm[ks], acc[mt], 0, 0, 0);
                }
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh[ks], acc[mt], 0, 0, 0);
            }
        }
    }
    if (TM == 6 && !HAS_B2) {
        // [slab][row half][channel]: k_colsum_reduce adds the 2 x slabs rows in index order
        if (a.colsum_part && blockIdx.y == 0) a.colsum_part[((long long)blockIdx.x * 2 + fhh) * 256 + fm] = csum;
    }
    if (!active) return;
    if (H2) {
        const float inv = 1.f / (h2sa * h2sb);               // exact: powers of two
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] *= inv;
    }
    // [workgroup][wave = problem][mt][r][lane]: whole 256-byte rows per store instruction, no atomics; k_wgrad_b3w_reduce
    // sums the workgroups' tiles.  (256 workgroups adding 64 K values each into the SAME 64 K addresses with float atomics
    // were 98 of this kernel's 155 us at config 5's layer shapes.)
    float* __restrict__ o = a.part + (((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wv) * (MT * 16 * 64) + lane;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[(mt * 16 + r) * 64] = acc[mt][r];
}

// ---------------------------------------------------------------------------------------------
// k_wgrad_h2p: the wide weight-gradient block for the fp16-split dWs launch (M == 256 rows of A = dskip, B = z of one
// layer per wave), restructured after its in-kernel stamps: a 32-row chunk of k_wgrad_b3w<., ., 3> cost a wave 8,000
// cycles -- split + image stores 1,600, barrier 1,100-2,000, issuing the 32 dword loads (64-bit address arithmetic each)
// 1,250-2,350, the 48 MFMAs 3,050 (two waves share a SIMD's pipe) -- with all eight waves in lock step behind the one
// barrier, i.e. the matrix pipe idle through 62 % of the chunk.  Here
//   * the raw A chunk (32 rows x 256 floats = 32 KB) arrives by LDS-DMA into a two-deep staging ring and the raw B rows
//     (16 dwords per lane) into two register sets, both requested TWO chunks ahead (right after the barrier that frees
//     their buffers): scalar bases + fixed lane offsets for the DMA, scalar row offsets + one select per B load;
//   * the split of chunk c + 1 (A: staging -> fp16 planes of the next image buffer; B: registers -> the other operand
//     set) rides in the MFMA stream of chunk c, four values per m-tile;
//   * one LDS-only barrier per chunk; the requests of chunk c + 1 are a whole chunk old when they are waited for
//     (`s_waitcnt vmcnt(0)` in front of the barrier: chunk c + 2 is requested behind it).
// Same products in the same order as k_wgrad_b3w<false, ACT, 3>: bit-identical partial tiles.  Requirements (the
// launcher checks): fp16 split with a measured A range, no B2 factor, lda == 256, 32-bit byte offsets into a B clip.
// LDS: two 32 KB images ([tile][ks][plane][lane][8] fp16) + two 32 KB staging buffers = 128 KB, one workgroup per CU.
// ---------------------------------------------------------------------------------------------
static constexpr int kWpTile = 4096, kWpImg = 8 * kWpTile, kWpStage = 32 * 1024, kWpLds = 2 * kWpImg + 2 * kWpStage;

template <int ACT>
__global__ __launch_bounds__(512, 1) void k_wgrad_h2p(WGArgs a) {
    constexpr int MT = 8;
    const float h2sa = h2_scale(a.amax_dev, 1.f);
    const float h2sb = h2_scale(a.bmax_dev, kH2ScaleX);
    extern __shared__ __attribute__((aligned(16))) char ldsp2[];
    char* const img0 = ldsp2;                                  // image buffers: img0 + (c & 1) * kWpImg
    char* const stg0 = ldsp2 + 2 * kWpImg;                     // staging:       stg0 + (c & 1) * kWpStage
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int p = blockIdx.y * 8 + wv;
    const bool active = p < a.nprob;
    const int off = a.off + a.offp[active ? p : 0];
    const int b = blockIdx.x / a.wgs_per_b;
    const int r_begin = (blockIdx.x - b * a.wgs_per_b) * a.rows_per_wg;
    const int r_end = min(a.rows_A_per_b, r_begin + a.rows_per_wg);
    const int nch = (r_end - r_begin + 31) / 32;
    const char* Abase = reinterpret_cast<const char*>(a.A + ((long long)b * a.rows_A_per_b) * 256);          // uniform
    const char* Bbase = reinterpret_cast<const char*>(a.Bp[active ? p : 0] + ((long long)b * a.rows_B_per_b + off) * a.ldb);
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ldsp2;
    const unsigned a_voff = lane * 16;                          // a staging row: lane L moves channels 4 L .. 4 L + 3
    const int ldb4 = a.ldb * 4;
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    const int fm = tid & 255, fhh = tid >> 8;
    const float amask = active ? 1.f : 0.f;
    float brr[2][2][8];                 // raw B of chunks c + 1 / c + 2 (set = chunk parity): [ks][jj] = row 16 ks + 8 h + jj
    f16x8 bh[2][2], bm[2][2];           // B operands by chunk parity

    // requests of chunk cc (clamped to the slab's last chunk): 4 LDS-DMA rows of A (inline asm: scalar base, fixed lane
    // offset, M0 from scalar arithmetic) + 16 dwords of B as ORDINARY loads -- the compiler must know that brr is in flight:
    // as asm outputs it copied them to other registers at the loop header, before they had landed (intermittently wrong
    // sums at full size).  Rows past the slab's end are clamped in scalar arithmetic (two candidates per load, the lane's
    // row half selects) and masked at split time.
    auto issue = [&](int cc, auto set_tag) {
        constexpr int set = decltype(set_tag)::value;
        const int c = cc < nch ? cc : nch - 1;
        const int r0 = r_begin + 32 * c;
        const int rlast = r_end - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {                           // this wave's four rows of the chunk (clamped at the slab's end)
            const int row = min(r0 + 4 * wv + k, rlast);
            const char* sb = Abase + (long long)row * 1024;
            const unsigned m0v = lds_base + 2 * kWpImg + set * kWpStage + (4 * wv + k) * 1024;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(a_voff), "s"(sb), "s"(m0v) : "memory", "m0");
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int row = r0 + 16 * ks + jj;
                const unsigned o0 = (unsigned)(min(row, rlast) * ldb4), o1 = (unsigned)(min(row + 8, rlast) * ldb4);   // scalar
                const unsigned vo = (h ? o1 : o0) + (unsigned)(j * 4);
                brr[set][ks][jj] = *reinterpret_cast<const float*>(Bbase + vo);
            }
    };
    // split of two A values and two B values of chunk cc (e0, e0 + 1 of k-step ks) into A fragments fa / B operand set
    auto split4 = [&](int cc, auto set_tag, int ks, int e0, f16x8 (&fah)[2], f16x8 (&fam)[2]) {
        constexpr int set = decltype(set_tag)::value;
        const int r0 = r_begin + 32 * cc;                       // NOT clamped: a chunk past the slab's end is all zeros
        const char* st = stg0 + set * kWpStage;
#pragma unroll
        for (int e = e0; e < e0 + 2; ++e) {
            const int ra = r0 + 16 * ks + 8 * fhh + e, rb = r0 + 16 * ks + 8 * h + e;
            float av = *reinterpret_cast<const float*>(st + (16 * ks + 8 * fhh + e) * 1024 + fm * 4);
            float bv = act_apply_t<ACT>(brr[set][ks][e]) * amask;
            av = ra < r_end ? av : 0.f;
            bv = rb < r_end ? bv : 0.f;
            _Float16 y0, y1;
            split2h(av * h2sa, y0, y1);
            fah[ks][e] = y0; fam[ks][e] = y1;
            split2h(bv * h2sb, y0, y1);
            bh[set][ks][e] = y0; bm[set][ks][e] = y1;
        }
    };
    auto store_frag = [&](int set, int ks, const f16x8& fah, const f16x8& fam) {
        char* d = img0 + set * kWpImg + (fm >> 5) * kWpTile + ks * 2048 + ((fm & 31) + 32 * fhh) * 16;
        *reinterpret_cast<f16x8*>(d) = fah;
        *reinterpret_cast<f16x8*>(d + 1024) = fam;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    if (nch > 0) {
        // chunk 0: requested, landed, published, split outside the MFMA stream; chunk 1 is requested in between
        issue(0, I0{});
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0) (a builtin: the compiler's load bookkeeping follows it)
        __builtin_amdgcn_s_barrier();
        issue(1, I1{});
        {
            f16x8 fah[2], fam[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int e0 = 0; e0 < 8; e0 += 2) split4(0, I0{}, ks, e0, fah, fam);
                store_frag(0, ks, fah[ks], fam[ks]);
            }
        }
    }
    auto step = [&](int c, auto set_tag) {
        constexpr int set = decltype(set_tag)::value;          // c & 1
        constexpr int nxt = 1 - set;
        using NXT = std::integral_constant<int, nxt>;
        // chunk c + 1's requests are a chunk old; image(c) and (for the others) this wave's rows of staging(c + 1) are complete
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): the loads the compiler tracks AND the LDS-DMA it cannot see
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(c + 2, set_tag);                                  // staging / raw set of chunk c: split a chunk ago by everyone
        const char* Al = img0 + set * kWpImg + lane * 16;
        f16x8 fr[2][2][2];
        auto ldA = [&](int mt, f16x8 (&f)[2][2]) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f[ks][0] = *reinterpret_cast<const f16x8*>(Al + mt * kWpTile + ks * 2048);
                f[ks][1] = *reinterpret_cast<const f16x8*>(Al + mt * kWpTile + ks * 2048 + 1024);
            }
        };
        f16x8 fah[2], fam[2];
        ldA(0, fr[0]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f16x8 (&f)[2][2] = fr[mt & 1];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                asm volatile("" ::"v"(f[ks][0]));
                asm volatile("" ::"v"(f[ks][1]));
            }
            if (mt + 1 < MT) ldA(mt + 1, fr[(mt + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            split4(c + 1, NXT{}, mt >> 2, (mt & 3) * 2, fah, fam);
            if ((mt & 3) == 3) store_frag(nxt, mt >> 2, fah[mt >> 2], fam[mt >> 2]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[ks][1], bh[set][ks], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[ks][0], bm[set][ks], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[ks][0], bh[set][ks], acc[mt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Two steps per trip, no branch between them: the raw B registers are written by requests the compiler cannot see, and a
    // control-flow merge inside the ring would let it copy them while they are in flight (seen: wrong sums at odd chunk
    // counts).  An odd slab runs one chunk past its end, whose rows are masked to zero by the split.
    for (int c = 0; c < nch; c += 2) {
        step(c, I0{});
        step(c + 1, I1{});
    }
    // requests past the slab's end are still in flight (staging that nobody reads any more)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (!active) return;
    const float inv = 1.f / (h2sa * h2sb);                       // exact: powers of two
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] *= inv;
    float* __restrict__ o = a.part + (((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wv) * (MT * 16 * 64) + lane;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[(mt * 16 + r) * 64] = acc[mt][r];
}

// Sum of the partial tiles of k_wgrad_b3w: one thread per output element, the workgroups' tiles of one (problem, mt, r)
// are 256-byte rows `stride` floats apart.  dW += sum (sole writer of its element).
__global__ void k_wgrad_b3w_reduce(WGArgs a, int nwg_x) {
    if ((int)blockIdx.y == a.nprob) {
        // the extra grid row (launched only with a column-sum request): the bias gradient's 2 x nwg_x partial rows, in the fixed
        // tree of k_colsum_reduce (generic_kernels.hip) -- row lane y adds rows y, y + 16, ... in order, the 16 lane sums are added
        // in index order -- so the result is that kernel's bit for bit; it used to be a launch of its own (one workgroup per 64
        // columns, 11 us at the launch floor's mercy)
        __shared__ float red[16][16];
        const int c = threadIdx.x & 15, y = threadIdx.x >> 4;
        const int m = blockIdx.x * 16 + c;
        if (m >= 256) return;                                   // (whole blocks: blockIdx.x >= 16)
        const int ny = 2 * nwg_x;
        float sacc = 0.f;
        for (int r = y; r < ny; r += 16) sacc += a.colsum_part[(long long)r * 256 + m];
        red[y][c] = sacc;
        __syncthreads();
        if (y == 0) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][c];
            a.colsum[m] += t;
        }
        return;
    }
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // (mt, r) of this wave: 0 .. 127
    const int p = blockIdx.y;                                              // problem
    const int y = p >> 3, wv = p & 7;
    const long long stride = 8ll * 8 * 16 * 64;                             // one workgroup's tiles
    const float* __restrict__ src = a.part + (((long long)y * nwg_x) * 8 + wv) * (8 * 16 * 64) + e * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = 0;
    for (; w + 4 <= nwg_x; w += 4) {
        s0 += src[(w + 0) * stride]; s1 += src[(w + 1) * stride]; s2 += src[(w + 2) * stride]; s3 += src[(w + 3) * stride];
    }
    for (; w < nwg_x; ++w) s0 += src[w * stride];
    const int mt = e >> 4, r = e & 15, j = lane & 31, h = lane >> 5;
    const bool second = a.m_split > 0 && mt * 32 >= a.m_split;
    float* __restrict__ o = second ? a.out2[p] : a.out[p];
    const int mrow = mt * 32 - (second ? a.m_split : 0);
    o[(long long)(mrow + b3_ch(r, h)) * a.ldo + (long long)j * (a.osk ? a.osk : 1)] += (s0 + s1) + (s2 + s3);
}

// M == 256 and at least 8 problems: the wide block.  Picks its own row slabs (one workgroup per CU).
int launch_wgrad_b3w(WGArgs& a_io, hipStream_t s) {
    WGArgs a = a_io;
    static bool attr_set = false;
    if (!attr_set) {
#define W_ATTR(B2_, ACT_)                                                                               \
    WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_b3w<B2_, ACT_, 6>),                 \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * kTileBytes));        \
    WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_b3w<B2_, ACT_, 3>),                 \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * kTileBytes));        \
    WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_b3w<B2_, ACT_, 1>),                 \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * kTileBytes))
        W_ATTR(false, WN_ACT_NONE); W_ATTR(false, WN_ACT_RELU); W_ATTR(false, WN_ACT_ELU);
        W_ATTR(true, WN_ACT_NONE); W_ATTR(true, WN_ACT_RELU); W_ATTR(true, WN_ACT_ELU);
#undef W_ATTR
        attr_set = true;
    }
    const int gy = (a.nprob + 7) / 8;
    int slabs = 256 / gy;                                    // workgroups in flight: one per CU
    if (slabs < 1) slabs = 1;
    int per_b = slabs / a.nB;                                // never more workgroups than CUs: a second round doubles the time
    if (per_b < 1) per_b = 1;
    int rows = (a.rows_A_per_b + per_b - 1) / per_b;
    rows = ((rows + 31) / 32) * 32;
    if (rows < 256) rows = 256;
    a.rows_per_wg = rows;
    a.wgs_per_b = (a.rows_A_per_b + rows - 1) / rows;
    bool any_b2 = false, all_b2 = true;
    for (int q = 0; q < a.nprob; ++q) { any_b2 |= a.B2p[q] != nullptr; all_b2 &= a.B2p[q] != nullptr; }
    if (any_b2 != all_b2) { wn::set_error("wgrad_b3w: the B2 factor must be given for all problems or for none"); return WN_EARG; }
    const dim3 grid(a.nB * a.wgs_per_b, gy);
    // The result leaves as per-workgroup partial tiles (plain coalesced stores) that k_wgrad_b3w_reduce sums in a fixed
    // order.  With few 32-row chunks per workgroup (config 5's per-layer gradients: 16) the alternative -- 64 K float
    // atomics per workgroup into the same addresses -- cost more than the contraction; with long slabs (config 2's dWs: 85
    // chunks) the two cost the same (the partial tiles are ~60 MB of extra traffic) and this form is deterministic.
    const size_t part_bytes = (size_t)grid.x * grid.y * 8 * (8 * 16 * 64) * sizeof(float);
    const bool one = one_term();
    const bool h2 = half2_mode() && a.h2 && a.amax_dev && !any_b2;
    // the bias gradient rides along in the six-term form (the head convolutions): 2 x slabs rows of 256 column sums
    const bool cs = a.colsum && !one && !h2 && !any_b2;
    const size_t cs_bytes = cs ? (size_t)grid.x * 2 * 256 * sizeof(float) : 0;
    a.part = reinterpret_cast<float*>(exec_scratch(part_bytes + cs_bytes, "the weight-gradient partial tiles"));
    if (!a.part) return WN_EARG;
    a.colsum_part = cs ? a.part + part_bytes / sizeof(float) : nullptr;
    // the pipelined kernel clamps B's rows to the end of the slab only: every row a chunk can reach must lie inside B's clip,
    // i.e. a non-negative shift that keeps A's rows inside B's and no per-problem shift (the skip path's shape); anything else
    // takes k_wgrad_b3w, which masks rows outside the clip
    bool clip_ok = a.off >= 0 && (long long)a.rows_A_per_b + a.off <= a.rows_B_per_b;
    for (int q = 0; q < a.nprob && clip_ok; ++q) clip_ok = a.offp[q] == 0;
    if (h2 && clip_ok && a.lda == 256 && (long long)a.rows_B_per_b * a.ldb * 4 < (1ll << 31) && !exec_flag(WN_EXEC_NO_PIPELINED_GEMM)) {
        static bool attr_p = false;
        if (!attr_p) {
            WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_h2p<WN_ACT_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, kWpLds));
            WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_h2p<WN_ACT_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, kWpLds));
            WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_h2p<WN_ACT_ELU>), hipFuncAttributeMaxDynamicSharedMemorySize, kWpLds));
            attr_p = true;
        }
        if (a.act == WN_ACT_RELU) hipLaunchKernelGGL((k_wgrad_h2p<WN_ACT_RELU>), grid, dim3(512), kWpLds, s, a);
        else if (a.act == WN_ACT_ELU) hipLaunchKernelGGL((k_wgrad_h2p<WN_ACT_ELU>), grid, dim3(512), kWpLds, s, a);
        else hipLaunchKernelGGL((k_wgrad_h2p<WN_ACT_NONE>), grid, dim3(512), kWpLds, s, a);
        WN_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_wgrad_b3w_reduce, dim3(128 / 4, a.nprob), dim3(256), 0, s, a, (int)grid.x);
        WN_LAUNCH_CHECK();
        return WN_OK;
    }
#define W_LAUNCH(B2_, ACT_)                                                                                     \
    do {                                                                                                        \
        if (one) hipLaunchKernelGGL((k_wgrad_b3w<B2_, ACT_, 1>), grid, dim3(512), 2 * 8 * kTileBytes, s, a);     \
        else if (h2) hipLaunchKernelGGL((k_wgrad_b3w<B2_, ACT_, 3>), grid, dim3(512), 2 * 8 * kTileBytes, s, a); \
        else hipLaunchKernelGGL((k_wgrad_b3w<B2_, ACT_, 6>), grid, dim3(512), 2 * 8 * kTileBytes, s, a);         \
    } while (0)
#define W_LAUNCH_A(B2_)                                           \
    do {                                                          \
        if (a.act == WN_ACT_RELU) W_LAUNCH(B2_, WN_ACT_RELU);     \
        else if (a.act == WN_ACT_ELU) W_LAUNCH(B2_, WN_ACT_ELU);  \
        else W_LAUNCH(B2_, WN_ACT_NONE);                          \
    } while (0)
    if (any_b2) W_LAUNCH_A(true);
    else W_LAUNCH_A(false);
#undef W_LAUNCH_A
#undef W_LAUNCH
    WN_LAUNCH_CHECK();
    // (+ one grid row for the column sums' partial rows when they were asked for: see the kernel)
    hipLaunchKernelGGL(k_wgrad_b3w_reduce, dim3(128 / 4, a.nprob + (cs ? 1 : 0)), dim3(256), 0, s, a, (int)grid.x);
    WN_LAUNCH_CHECK();
    if (cs) a_io.colsum_done = 1;
    return WN_OK;
}

int launch_wgrad_b3(const WGArgs& a, int mt, dim3 grid, hipStream_t s) {
    bool any_b2 = false, all_b2 = true;
    for (int q = 0; q < a.nprob; ++q) { any_b2 |= a.B2p[q] != nullptr; all_b2 &= a.B2p[q] != nullptr; }
    if (any_b2 != all_b2) { wn::set_error("wgrad_b3: the B2 factor must be given for all problems or for none"); return WN_EARG; }
    const bool one = one_term();
#define WG_LAUNCH_T(MT_, B2_, ACT_)                                                                     \
    do {                                                                                                \
        if (one) hipLaunchKernelGGL((k_wgrad_b3<MT_, B2_, ACT_, true>), grid, dim3(256), 0, s, a);      \
        else hipLaunchKernelGGL((k_wgrad_b3<MT_, B2_, ACT_, false>), grid, dim3(256), 0, s, a);         \
    } while (0)
#define WG_LAUNCH_A(MT_, B2_)                                                                           \
    do {                                                                                                \
        if (a.act == WN_ACT_RELU) WG_LAUNCH_T(MT_, B2_, WN_ACT_RELU);                                   \
        else if (a.act == WN_ACT_ELU) WG_LAUNCH_T(MT_, B2_, WN_ACT_ELU);                                \
        else WG_LAUNCH_T(MT_, B2_, WN_ACT_NONE);                                                        \
    } while (0)
#define WG_LAUNCH(MT_)                                                                              \
    do {                                                                                            \
        if (any_b2) WG_LAUNCH_A(MT_, true);                                                         \
        else WG_LAUNCH_A(MT_, false);                                                               \
    } while (0)
    switch (mt) {                        // at most 4 row tiles per workgroup here (8 would spill: 128 accumulator registers)
        case 4: WG_LAUNCH(4); break;
        case 2: WG_LAUNCH(2); break;
        case 1: WG_LAUNCH(1); break;
        default: wn::set_error("wgrad_b3: unsupported tile count %d", mt); return WN_ESHAPE;
    }
#undef WG_LAUNCH_T
#undef WG_LAUNCH_A
#undef WG_LAUNCH
    WN_LAUNCH_CHECK();
    return WN_OK;
}

}  // namespace wn
