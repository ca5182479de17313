// Fused residual-layer kernels of the bf16-storage path (BASELINE config 5; ResidualConvLayer.__call__, wavenet.py:358-368,
// and Chainer's backward through it), Cr = Cd = 128, filter width 2.
//
//   k16_fwd       x (bf16) -> out = Wp z + x (bf16), z = tanh(Wf * x) sigmoid(Wg * x) (bf16).  One kernel per layer.
//   k16_gate_bwd  recomputes tanh / sigmoid from x (nothing but z is saved by the forward), dz = Wp^T dout + dz_skip,
//                 [da | dg] = dz (g (1 - f^2) | f g (1 - g)) (bf16).
//   k16_dx        dx[t] = dout[t] + [Wf1;Wg1]^T dab[t] + [Wf0;Wg0]^T dab[t + d] (bf16) -- the gradient the layer below
//                 receives -- and, with that layer's z, its projection weight gradient dWp += dx z^T (fp32 partials).
// All three are weight-stationary: every wave keeps its slice of the layer's weights in registers as MFMA A operands for
// the whole launch and the workgroup streams 32-column time tiles (256-byte rows) through LDS by LDS-DMA.
//
// Shape of the first two (measured on config 5, ablations in DESIGN.md): per 64 columns and SIMD the layer needs ~2,600
// cycles of matrix pipe and ~2,800 cycles of VALU issue (tanh / sigmoid: four transcendentals per gate element).  One
// 8-wave workgroup per CU with barriers between the phases kept all waves in lockstep -- everybody on the matrix pipe,
// then everybody on the VALU -- and the phases ADDED (28 us of compute per layer against 19 us of memory time).  Now a
// workgroup is 4 waves (one per SIMD, 32 gate channels each: the filter rows and the gate rows of a channel are the same
// accumulator register of two MFMA tiles) and TWO workgroups share a CU: they are not synchronised with each other, so one
// workgroup's gate arithmetic runs under the other's MFMAs and stores.
#include <stdlib.h>
#include <type_traits>

#include "w16_gemm.hpp"
#include "wn_kernels.hpp"

namespace w16 {

using wn::fast_sigmoid;
using wn::fast_tanh;

// Diagnostic build (-DWN16_STAMPS): lane 0 of wave 0 of two workgroups records s_memtime at the phase boundaries of its
// first tiles into a device array that wn16_debug_stamps() copies out.  Never compiled into the product library.
#ifdef WN16_STAMPS
__device__ unsigned long long g_stamps[3 * 2 * 16 * 16];   // [kernel: fwd, gate_bwd, dx][first / last workgroup][tile][stamp]
#define STAMP_K(kern, k)                                                                                  \
    do {                                                                                                  \
        if (threadIdx.x == 0 && it < 16 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))              \
            g_stamps[(((kern) * 2 + (blockIdx.x != 0)) * 16 + it) * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define STAMP_K(kern, k) do { } while (0)
#endif
#define STAMP(k) STAMP_K(0, k)

// z = tanh(a) sigmoid(g) with ONE reciprocal: (1 - e^{-2a}) / ((1 + e^{-2a}) (1 + e^{-g})) -- two exponentials, one reciprocal and
// eight plain instructions instead of two of each and fourteen (fast_tanh's small-|a| series included): the forward is bound
// by VALU issue, and the result is rounded to bf16 (8 bits) right away.  a is clamped at -30 (tanh is -1 to
// the last fp32 bit long before; e^{60} stays finite so that inf / inf cannot arise); a huge e^{-g} makes the
// denominator inf and z the 0 it should be.
__device__ __forceinline__ float gate_z(float a, float g) {
    const float e2 = __builtin_amdgcn_exp2f(fmaxf(a, -30.f) * -2.8853900817779268f);
    const float eg = __builtin_amdgcn_exp2f(g * -1.4426950408889634f);
    return (1.f - e2) * __builtin_amdgcn_rcpf((1.f + e2) * (1.f + eg));
}

// tanh(a) and sigmoid(g) with ONE reciprocal (the backward needs both): r = 1 / ((1 + e^{-2a}) (1 + e^{-g})),
// sigmoid = (1 + e^{-2a}) r, tanh = (1 - e^{-2a}) (1 + e^{-g}) r.  g is clamped at -80 so that the product stays finite
// (sigmoid(-80) = 1.8e-35: a gradient factor of 0 to bf16 either way); a as in gate_z.
__device__ __forceinline__ void gate_fg(float a, float g, float& f, float& s) {
    const float e2 = __builtin_amdgcn_exp2f(fmaxf(a, -30.f) * -2.8853900817779268f);
    const float eg = __builtin_amdgcn_exp2f(fmaxf(g, -80.f) * -1.4426950408889634f);
    const float p2 = 1.f + e2, pg = 1.f + eg;
    const float r = __builtin_amdgcn_rcpf(p2 * pg);
    s = p2 * r;
    f = (1.f - e2) * (pg * r);
}

static constexpr int kLT = 32;                          // time columns per tile
static constexpr int kLTileB = kLT * 256;               // 8 KB

// ---------------------------------------------------------------------------------------------
// forward.  256 threads; LDS: (xold, xcur) x 2 buffers + z tile + out tile = 48 KB, + the projection's operand image
// (32 KB, read at each use: keeping it in registers left none to prefetch the column operands with, and every
// ds_read_b128 -> s_waitcnt -> MFMA pair ran back to back: 16 exposed LDS latencies per tile)
// ---------------------------------------------------------------------------------------------
static constexpr int kFwdLds = 6 * kLTileB + kProjA * 2;

__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void k16_fwd(const bf16* __restrict__ x, const bf16* __restrict__ convA,
                                                   const bf16* __restrict__ projA, bf16* __restrict__ out,
                                                   bf16* __restrict__ z, int B, int T, int d, int Z, int tiles_per_b,
                                                   int ntiles) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    auto xold = [&](int buf) { return lds + buf * kLTileB; };
    auto xcur = [&](int buf) { return lds + (2 + buf) * kLTileB; };
    char* zt = lds + 4 * kLTileB;
    char* ot = lds + 5 * kLTileB;
    char* wpl = lds + 6 * kLTileB;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    int first, stride, last;
    tile_range(ntiles, first, stride, last);
    if (first >= last) return;

    // A operands: filter rows and gate rows of channels 32 w .. 32 w + 31 (16 k-steps: 0..7 tap 0 = x[t-d], 8..15 tap 1
    // = x[t]); this wave's slice of the projection image goes to LDS
    bf16x8 fA[16], gA[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        fA[s] = *reinterpret_cast<const bf16x8*>(convA + (((w * 2 + 0) * 16 + s) * 64 + lane) * 8);
        gA[s] = *reinterpret_cast<const bf16x8*>(convA + (((w * 2 + 1) * 16 + s) * 64 + lane) * 8);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) W16_DMA16(projA + ((w * 8 + s) * 64 + lane) * 8, wpl + (w * 8 + s) * 1024);
    // the operand loads are waited for HERE: left pending, the compiler would place its s_waitcnt vmcnt(0) at their first
    // use inside the loop, where it would also drain the LDS-DMA prefetch of every iteration
#pragma unroll
    for (int s = 0; s < 16; ++s) { asm volatile("" ::"v"(fA[s])); asm volatile("" ::"v"(gA[s])); }

    // per-lane LDS offsets, the same for every tile
    int foff[8], qoff[4];
#pragma unroll
    for (int s = 0; s < 8; ++s) foff[s] = toff(j, 2 * s + h);       // operand of k-step s: row j, chunk 2 s + h
#pragma unroll
    for (int q = 0; q < 4; ++q) qoff[q] = toff(j, 4 * w + q) + 8 * h;   // accumulator registers 4 q .. 4 q + 3 of this wave
    const int prow = lane >> 4;                                     // row piece / chunk of the whole-row copies
    int pcol[2];
    unsigned loff[2];                                               // byte offset of this lane's 16 bytes inside a tile of rows
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        pcol[pp] = ((lane & 15) ^ key(4 * (2 * w + pp) + prow)) * 8;
        loff[pp] = (unsigned)(((4 * (2 * w + pp) + prow) * 128 + pcol[pp]) * 2);
    }
    const unsigned lds0 = lds_addr_of(lds);

    auto issue = [&](int tile, int buf) {
        const int b = tile / tiles_per_b;
        const int t0 = (tile - b * tiles_per_b) * kLT;
        const bf16* xb = x + (long long)b * T * 128;
        if (t0 + kLT <= T && t0 >= d) {                  // interior tile: no clamping; uniform base + fixed lane offsets
            const bf16* bc = xb + (long long)t0 * 128;
            const bf16* bo = bc - (long long)d * 128;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int p = 2 * w + pp;
                dma16_s(bc, loff[pp], lds0 + (2 + buf) * kLTileB + p * 1024);
                dma16_s(bo, loff[pp], lds0 + buf * kLTileB + p * 1024);
            }
            return;
        }
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
        const int t0 = (tile - b * tiles_per_b) * kLT;
        // this tile's 4 DMA pieces are older than the previous tile's 4 stores: those stay in flight
        STAMP(0);
        if (full_prev) wait_vm<4>(); else wait_vm<0>();
        STAMP(1);
        barrier();
        STAMP(2);
        if (tile + stride < last) issue(tile + stride, buf ^ 1);
        STAMP(3);
        if (t0 < d) {
            // rows whose tap-0 sample lies before the clip start read as 0 (wavenet.py:298-301 pads with zeros)
            for (int r = w; r < kLT; r += 4)
                if (t0 + r < d) *reinterpret_cast<unsigned*>(xold(buf) + r * 256 + lane * 4) = 0u;
            barrier();
        }
        // ---- both dilated convolutions for this wave's 32 channels; column operands fetched four k-steps ahead ----
        f32x16 af, ag;
#pragma unroll
        for (int r = 0; r < 16; ++r) { af[r] = 0.f; ag[r] = 0.f; }
        {
            const char* t0p = xold(buf);
            const char* t1p = xcur(buf);
            bf16x8 bq[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) bq[s] = *reinterpret_cast<const bf16x8*>(t0p + foff[s]);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const bf16x8 bv = bq[s & 3];
                if (s + 4 < 16) bq[s & 3] = *reinterpret_cast<const bf16x8*>((s + 4 < 8 ? t0p : t1p) + foff[(s + 4) & 7]);
                af = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fA[s], bv, af, 0, 0, 0);
                ag = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gA[s], bv, ag, 0, 0, 0);
            }
        }
        STAMP(4);
        // ---- gate ----
        if (t0 >= Z) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float zz[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) zz[e] = gate_z(af[4 * q + e], ag[4 * q + e]);
                *reinterpret_cast<bf16x4*>(zt + qoff[q]) = pack4(zz[0], zz[1], zz[2], zz[3]);
            }
        } else {                                       // the reference's zero prefix: a = g = 0 there, so z = 0
            const bool live = t0 + j >= Z;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float zz[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    zz[e] = live ? gate_z(af[4 * q + e], ag[4 * q + e]) : 0.f;
                *reinterpret_cast<bf16x4*>(zt + qoff[q]) = pack4(zz[0], zz[1], zz[2], zz[3]);
            }
        }
        STAMP(5);
        barrier();
        STAMP(6);
        // ---- residual projection: out rows 32 w .. of all 32 columns, + x ----
        {
            f32x16 ao;
#pragma unroll
            for (int r = 0; r < 16; ++r) ao[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s)
                ao = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    *reinterpret_cast<const bf16x8*>(wpl + ((w * 8 + s) * 64 + lane) * 16),
                    *reinterpret_cast<const bf16x8*>(zt + foff[s]), ao, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x4 xv = *reinterpret_cast<const bf16x4*>(xcur(buf) + qoff[q]);
                *reinterpret_cast<bf16x4*>(ot + qoff[q]) = pack4(ao[4 * q] + (float)xv[0], ao[4 * q + 1] + (float)xv[1],
                                                                 ao[4 * q + 2] + (float)xv[2], ao[4 * q + 3] + (float)xv[3]);
            }
        }
        STAMP(7);
        barrier();
        STAMP(8);
        // ---- whole 256-byte rows leave: 4 rows per wave instruction, two pieces of each tile per wave ----
        const bool full = t0 + kLT <= T;
        bf16* zb = z + ((long long)b * T + t0) * 128;
        bf16* ob = out + ((long long)b * T + t0) * 128;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int p = 2 * w + pp;
            const int r = 4 * p + prow;
            const u32x4 vz = *reinterpret_cast<const u32x4*>(zt + p * 1024 + lane * 16);
            const u32x4 vo = *reinterpret_cast<const u32x4*>(ot + p * 1024 + lane * 16);
            if (full || t0 + r < T) {
                st16_wt_s(zb, loff[pp], vz);
                st16_wt_s(ob, loff[pp], vo);
            }
        }
        full_prev = full;
        STAMP(9);
    }
}

// ---------------------------------------------------------------------------------------------
// gate backward.  512 threads, one workgroup per CU: wave w owns gate channels 16 w .. 16 w + 15 -- the filter rows and the
// gate rows of a channel sit in the SAME 32-row MFMA tile (rows r and r + 16 = accumulator registers r and r + 8 of one
// lane), dz comes from a tile whose rows 16..31 are zero.  LDS: (xold, xcur, dout, dzs) x 2 buffers + da, dg tiles = 80 KB.
// (A 4-wave / two-workgroup form like the forward's needs 128 + 32 operand registers per wave and spilled: 65 us against
// 47 us per layer.)
// ---------------------------------------------------------------------------------------------
static constexpr int kGateLds = 10 * kLTileB;

template <bool HAS_DO, bool HAS_DZ>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void k16_gate_bwd(
    const bf16* __restrict__ x, const bf16* __restrict__ convA, const bf16* __restrict__ dzA,
    const bf16* __restrict__ dout, const bf16* __restrict__ dzs, int dz_t0, bf16* __restrict__ dadg, int B, int T, int d,
    int Z, int tiles_per_b, int ntiles, int t_live) {
    // t_live (a multiple of 32): no gradient reaches this layer's columns below it (they are further from the loss window than
    // the layers above can see).  Such tiles load and compute nothing: they store the zeros their readers expect.
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    auto xold = [&](int buf) { return lds + buf * kLTileB; };
    auto xcur = [&](int buf) { return lds + (2 + buf) * kLTileB; };
    auto dot = [&](int buf) { return lds + (4 + buf) * kLTileB; };
    auto dzt = [&](int buf) { return lds + (6 + buf) * kLTileB; };
    char* dat = lds + 8 * kLTileB;
    char* dgt = lds + 9 * kLTileB;
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

#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" ::"v"(cA[s]));
    if (HAS_DO) {
#pragma unroll
        for (int s = 0; s < 8; ++s) asm volatile("" ::"v"(zA[s]));
    }

#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" ::"v"(cA[s]));
    if (HAS_DO) {
#pragma unroll
        for (int s = 0; s < 8; ++s) asm volatile("" ::"v"(zA[s]));
    }

    // this wave's piece of a 32-row tile: rows 4 w .. 4 w + 3; the lane's 16 bytes at a fixed offset from the tile's first row
    const unsigned off128 = (unsigned)((4 * w + (lane >> 4)) * 256 + (((lane & 15) ^ key(4 * w + (lane >> 4))) << 4));
    const unsigned lds0 = lds_addr_of(lds);
    auto issue = [&](int tile, int buf) {
        const int b = tile / tiles_per_b;
        const int t0 = (tile - b * tiles_per_b) * kLT;
        if (t0 + kLT <= t_live) return;                    // a dead tile
        if (t0 + kLT <= T && t0 >= d && (!HAS_DZ || t0 >= dz_t0)) {
            // interior tile (nothing to clamp): uniform base + the fixed lane offset, no 64-bit address arithmetic per
            // request -- the four requests cost a wave ~520 cycles of a ~5,900-cycle tile in the clamped form below
            const bf16* bc = x + ((long long)b * T + t0) * 128;
            dma16_s(bc, off128, lds0 + (2 + buf) * kLTileB + w * 1024);
            dma16_s(bc - (long long)d * 128, off128, lds0 + buf * kLTileB + w * 1024);
            if (HAS_DO) dma16_s(dout + ((long long)b * T + t0) * 128, off128, lds0 + (4 + buf) * kLTileB + w * 1024);
            if (HAS_DZ) dma16_s(dzs + ((long long)b * Tw + (t0 - dz_t0)) * 128, off128, lds0 + (6 + buf) * kLTileB + w * 1024);
            return;
        }
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
        const int t0 = (tile - b * tiles_per_b) * kLT;
        STAMP_K(1, 0);
        if (full_prev) wait_vm<kStores>(); else wait_vm<0>();
        STAMP_K(1, 1);
        barrier();
        STAMP_K(1, 2);
        if (tile + stride < last) issue(tile + stride, buf ^ 1);
        STAMP_K(1, 3);
        if (t0 + kLT <= t_live) {                          // dead tile (workgroup-uniform): [da | dg] = 0, two stores per wave as below
            const int r = 4 * w + (lane >> 4);
            const int c = (lane & 15) ^ key(r);
            bf16* o = dadg + ((long long)b * T + t0 + r) * 256 + c * 8;
            const u32x4 zero4 = {0u, 0u, 0u, 0u};
            st16_wt(o, zero4);
            st16_wt(o + 128, zero4);
            full_prev = true;
            continue;
        }
        const bool fix_old = t0 < d;
        const bool fix_do = false;
        if (fix_old || fix_do) {
            for (int r = w; r < kLT; r += 8) {
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
        {
            // every column operand of the phase is requested before the first MFMA: paired one by one (read -> wait -> MFMA) the
            // sixteen LDS latencies lay end to end (1,300 cycles for 512 cycles of matrix work, in-kernel stamps)
            bf16x8 fr[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) fr[s] = frag_row(s < 8 ? xold(buf) : xcur(buf), j, s & 7, h);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cA[s], fr[s], acc, 0, 0, 0);
        }
        STAMP_K(1, 4);
        // ---- dz = Wp^T dout + dz_skip (registers 0..7; rows 16..31 of the A tile are zero) ----
        f32x16 dz;
#pragma unroll
        for (int r = 0; r < 16; ++r) dz[r] = 0.f;
        if (HAS_DZ) {
            if (t0 + kLT > dz_t0) {
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
            bf16x8 fr[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) fr[s] = frag_row(dot(buf), j, s, h);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 8; ++s) dz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zA[s], fr[s], dz, 0, 0, 0);
        }
        STAMP_K(1, 5);
        // ---- gate forward + backward, elementwise ----
        const bool live = t >= Z && t < T;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float da[4], dg[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float f, g;
                gate_fg(live ? acc[4 * q + e] : 0.f, live ? acc[8 + 4 * q + e] : 0.f, f, g);
                const float dzv = live ? dz[4 * q + e] : 0.f;
                da[e] = dzv * g * (1.f - f * f);
                dg[e] = dzv * f * g * (1.f - g);
            }
            const int o = toff(j, 2 * w + q) + 8 * h;
            *reinterpret_cast<bf16x4*>(dat + o) = pack4(da[0], da[1], da[2], da[3]);
            *reinterpret_cast<bf16x4*>(dgt + o) = pack4(dg[0], dg[1], dg[2], dg[3]);
        }
        STAMP_K(1, 6);
        barrier();
        STAMP_K(1, 7);
        // ---- [da | dg] rows leave whole: 512-byte rows, da in the first half ----
        const bool full = t0 + kLT <= T;
        {
            const int r = 4 * w + (lane >> 4);
            const int c = (lane & 15) ^ key(r);
            const u32x4 va = *reinterpret_cast<const u32x4*>(dat + w * 1024 + lane * 16);
            const u32x4 vg = *reinterpret_cast<const u32x4*>(dgt + w * 1024 + lane * 16);
            bf16* o = dadg + ((long long)b * T + t0 + r) * 256 + c * 8;
            if (full || t0 + r < T) {
                st16_wt(o, va);
                st16_wt(o + 128, vg);
            }
        }
        full_prev = full;
        STAMP_K(1, 8);
    }
}

// ---------------------------------------------------------------------------------------------
// dx + dWp.  512 threads, one workgroup per CU (no VALU-heavy phase here).  Wave (mt = w & 3, kh = w >> 2): output rows
// 32 mt .., contraction half kh (0: dab[t] with the tap-1 weights, 1: dab[t + d] with the tap-0 weights); the halves
// meet in an fp32 LDS patch.  LDS: (da, dg, da', dg', dout, z) x 2 buffers = 96 KB + 16 KB exchange + 8 KB dx tile
// ---------------------------------------------------------------------------------------------
static constexpr int kDxLds = 12 * kLTileB + 16384 + kLTileB;
static constexpr int kDwpPart = 128 * 128;             // floats per workgroup partial of dWp

template <bool HAS_DO, bool HAS_Z>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void k16_dx(const bf16* __restrict__ dadg, const bf16* __restrict__ dxA,
                                                  const bf16* __restrict__ dout, const bf16* __restrict__ zprev,
                                                  bf16* __restrict__ dx, float* __restrict__ dwp_part, int B, int T, int d,
                                                  int tiles_per_b, int ntiles, int t_live) {
    // t_live (a multiple of 32): dx is exactly zero below it (no gradient reaches those columns); such tiles store zeros
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    auto tile_at = [&](int buf, int which) { return lds + (buf * 6 + which) * kLTileB; };   // da dg da' dg' dout z
    float* xch = reinterpret_cast<float*>(lds + 12 * kLTileB);
    char* dxt = lds + 12 * kLTileB + 16384;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int mt = w & 3, kh = w >> 2;
    int first, stride, last;
    tile_range(ntiles, first, stride, last);

    bf16x8 A[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) A[s] = *reinterpret_cast<const bf16x8*>(dxA + (((mt * 2 + kh) * 16 + s) * 64 + lane) * 8);
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" ::"v"(A[s]));
    // dWp[cr][cd] partial: wave w owns rows cr 32 (w & 3) .. + 31, columns cd 64 (w >> 2) .. + 63
    f32x16 wp[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { wp[0][r] = 0.f; wp[1][r] = 0.f; }

    const unsigned off128 = (unsigned)((4 * w + (lane >> 4)) * 256 + (((lane & 15) ^ key(4 * w + (lane >> 4))) << 4));
    const unsigned off256 = off128 + (unsigned)((4 * w + (lane >> 4)) * 256);          // the same piece of 512-byte [da | dg] rows
    const unsigned lds0 = lds_addr_of(lds);
    auto issue = [&](int tile, int buf) {
        const int b = tile / tiles_per_b;
        const int t0 = (tile - b * tiles_per_b) * kLT;
        if (t0 + kLT <= t_live) return;                    // a dead tile
        if (t0 + kLT + d <= T) {                           // interior tile: uniform bases + fixed lane offsets (as in k16_gate_bwd)
            const bf16* a0 = dadg + ((long long)b * T + t0) * 256;
            const bf16* a1 = a0 + (long long)d * 256;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                dma16_s(a0 + 128 * half, off256, lds0 + (buf * 6 + half) * kLTileB + w * 1024);
                dma16_s(a1 + 128 * half, off256, lds0 + (buf * 6 + 2 + half) * kLTileB + w * 1024);
            }
            if (HAS_DO) dma16_s(dout + ((long long)b * T + t0) * 128, off128, lds0 + (buf * 6 + 4) * kLTileB + w * 1024);
            if (HAS_Z) dma16_s(zprev + ((long long)b * T + t0) * 128, off128, lds0 + (buf * 6 + 5) * kLTileB + w * 1024);
            return;
        }
        const bf16* ab = dadg + (long long)b * T * 256;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            dma_pieces(tile_at(buf, half), lane, w, 1, 1, [&](int r) {
                const int t = t0 + r < T ? t0 + r : T - 1;
                return ab + (long long)t * 256 + 128 * half;
            });
            dma_pieces(tile_at(buf, 2 + half), lane, w, 1, 1, [&](int r) {
                int t = t0 + r + d;
                t = t < T ? t : T - 1;
                return ab + (long long)t * 256 + 128 * half;
            });
        }
        if (HAS_DO) {
            const bf16* db = dout + (long long)b * T * 128;
            dma_pieces(tile_at(buf, 4), lane, w, 1, 1, [&](int r) {
                const int t = t0 + r < T ? t0 + r : T - 1;
                return db + (long long)t * 128;
            });
        }
        if (HAS_Z) {
            const bf16* zb = zprev + (long long)b * T * 128;
            dma_pieces(tile_at(buf, 5), lane, w, 1, 1, [&](int r) {
                const int t = t0 + r < T ? t0 + r : T - 1;
                return zb + (long long)t * 128;
            });
        }
    };

    if (first < last) issue(first, 0);
    bool full_prev = false;
    int it = 0;
    for (int tile = first; tile < last; tile += stride, ++it) {
        const int buf = it & 1;
        const int b = tile / tiles_per_b;
        const int t0 = (tile - b * tiles_per_b) * kLT;
        STAMP_K(2, 0);
        if (full_prev) wait_vm<1>(); else wait_vm<0>();  // one dx store per wave and tile
        STAMP_K(2, 1);
        barrier();
        STAMP_K(2, 2);
        if (tile + stride < last) issue(tile + stride, buf ^ 1);
        STAMP_K(2, 3);
        if (t0 + kLT <= t_live) {                          // dead tile (workgroup-uniform): dx = 0, one store per wave as below
            const int r = 4 * w + (lane >> 4);
            const int c = (lane & 15) ^ key(r);
            const u32x4 zero4 = {0u, 0u, 0u, 0u};
            st16_wt(dx + ((long long)b * T + t0 + r) * 128 + c * 8, zero4);
            full_prev = true;
            continue;
        }
        if (t0 + kLT + d > T) {                          // dab[t + d] beyond the clip end contributes nothing
            for (int r = w; r < kLT; r += 8)
                if (t0 + r + d >= T) {
                    *reinterpret_cast<unsigned*>(tile_at(buf, 2) + r * 256 + lane * 4) = 0u;
                    *reinterpret_cast<unsigned*>(tile_at(buf, 3) + r * 256 + lane * 4) = 0u;
                }
            barrier();
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            bf16x8 fr[16];                               // all sixteen operands requested before the first MFMA (as in k16_gate_bwd)
#pragma unroll
            for (int s = 0; s < 16; ++s) fr[s] = frag_row(tile_at(buf, 2 * kh + (s >> 3)), j, s & 7, h);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s], fr[s], acc, 0, 0, 0);
        }
        STAMP_K(2, 4);
        // the two contraction halves meet: wave kh keeps accumulator quarters 2 kh, 2 kh + 1 and hands the other two to its
        // partner (before, one half handed over everything and idled through the other's 1,200-cycle combine)
        auto hand_over = [&](auto KH) {                   // (constant register indices: kh is uniform but not a constant)
            constexpr int kk = decltype(KH)::value;
#pragma unroll
            for (int r = 0; r < 8; ++r) xch[(mt * 16 + 8 * (1 - kk) + r) * 64 + lane] = acc[8 * (1 - kk) + r];
        };
        auto combine = [&](auto KH) {
            constexpr int kk = decltype(KH)::value;
            const bool valid = t0 + j < T;               // rows beyond the clip must not reach dWp
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                constexpr int q0 = 2 * kk;
                const int q = q0 + qq;
                const int o = toff(j, 4 * mt + q) + 8 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[4 * q + e] + xch[(mt * 16 + 4 * q + e) * 64 + lane];
                if (HAS_DO) {
                    const bf16x4 xv = *reinterpret_cast<const bf16x4*>(tile_at(buf, 4) + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)xv[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = valid ? v[e] : 0.f;
                *reinterpret_cast<bf16x4*>(dxt + o) = pack4(v[0], v[1], v[2], v[3]);
            }
        };
        if (kh == 0) hand_over(std::integral_constant<int, 0>{}); else hand_over(std::integral_constant<int, 1>{});
        barrier();
        STAMP_K(2, 5);
        if (kh == 0) combine(std::integral_constant<int, 0>{}); else combine(std::integral_constant<int, 1>{});
        STAMP_K(2, 6);
        barrier();
        STAMP_K(2, 7);
        if (HAS_Z) {
            // dWp += dx z^T over this tile's 32 columns (contraction over time: transposed LDS reads)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 a = frag_tr(dxt, 16 * ks, 32 * (w & 3), lane);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const bf16x8 bz = frag_tr(tile_at(buf, 5), 16 * ks, 64 * (w >> 2) + 32 * n, lane);
                    wp[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bz, wp[n], 0, 0, 0);
                }
            }
        }
        STAMP_K(2, 8);
        const bool full = t0 + kLT <= T;
        {
            const int r = 4 * w + (lane >> 4);
            const int c = (lane & 15) ^ key(r);
            const u32x4 v = *reinterpret_cast<const u32x4*>(dxt + w * 1024 + lane * 16);
            if (full || t0 + r < T) st16_wt(dx + ((long long)b * T + t0 + r) * 128 + c * 8, v);
        }
        full_prev = full;
        STAMP_K(2, 9);
    }
    if (HAS_Z) {
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
static int grid_for(int ntiles, int per_cu) {
    int g = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
    if (g >= 8) g &= ~7;
    return g;
}

int fwd_layer(const bf16* x, const bf16* img, bf16* out, bf16* z, int B, int T, int d, int Z, hipStream_t s) {
    const int tiles_per_b = (T + kLT - 1) / kLT;
    const int ntiles = B * tiles_per_b;
    static bool attr = false;
    if (!attr) {
        WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k16_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kFwdLds));
        attr = true;
    }
    hipLaunchKernelGGL(k16_fwd, dim3(grid_for(ntiles, 2)), dim3(256), kFwdLds, s, x, img, img + kConvA, out, z, B, T, d, Z,
                       tiles_per_b, ntiles);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int gate_bwd_layer(const bf16* x, const bf16* img, const bf16* dout, const bf16* dzs, int dz_t0, bf16* dadg, int B, int T,
                   int d, int Z, int t_live, hipStream_t s) {
    const int tiles_per_b = (T + kLT - 1) / kLT;
    const int ntiles = B * tiles_per_b;
    const int grid = grid_for(ntiles, 1);
#define GB_LAUNCH(DO, DZ)                                                                                              \
    do {                                                                                                               \
        static bool attr = false;                                                                                      \
        if (!attr) {                                                                                                   \
            WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k16_gate_bwd<DO, DZ>),                            \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kGateLds));                         \
            attr = true;                                                                                               \
        }                                                                                                              \
        hipLaunchKernelGGL((k16_gate_bwd<DO, DZ>), dim3(grid), dim3(512), kGateLds, s, x, img + kOffConvA8,           \
                           img + kOffDzA8, dout, dzs, dz_t0, dadg, B, T, d, Z, tiles_per_b, ntiles, t_live);          \
    } while (0)
    if (dout && dzs) GB_LAUNCH(true, true);
    else if (dout) GB_LAUNCH(true, false);
    else if (dzs) GB_LAUNCH(false, true);
    else { wn::set_error("gate_bwd_layer: no incoming gradient"); return WN_EARG; }
#undef GB_LAUNCH
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int dx_grid(int B, int T) { return grid_for(B * ((T + kLT - 1) / kLT), 1); }

// dx = dout + conv^T(dadg); zprev (the z of the layer BELOW, may be NULL) -> that layer's dWp partial tiles
int dx_layer(const bf16* dadg, const bf16* img, const bf16* dout, const bf16* zprev, bf16* dx, float* dwp_part, int B,
             int T, int d, int t_live, hipStream_t s) {
    const int tiles_per_b = (T + kLT - 1) / kLT;
    const int ntiles = B * tiles_per_b;
    const int grid = dx_grid(B, T);
#define DX_LAUNCH(DO, ZZ)                                                                                              \
    do {                                                                                                               \
        static bool attr = false;                                                                                      \
        if (!attr) {                                                                                                   \
            WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k16_dx<DO, ZZ>),                                  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kDxLds));                           \
            attr = true;                                                                                               \
        }                                                                                                              \
        hipLaunchKernelGGL((k16_dx<DO, ZZ>), dim3(grid), dim3(512), kDxLds, s, dadg, img + kOffDxA, dout,                \
                           zprev, dx, dwp_part, B, T, d, tiles_per_b, ntiles, t_live);                                \
    } while (0)
    if (dout && zprev) DX_LAUNCH(true, true);
    else if (dout) DX_LAUNCH(true, false);
    else if (zprev) DX_LAUNCH(false, true);
    else DX_LAUNCH(false, false);
#undef DX_LAUNCH
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

#ifdef WN16_STAMPS
int debug_stamps(unsigned long long* dst, int n) {
    if (n > 3 * 2 * 16 * 16) n = 3 * 2 * 16 * 16;
    WN_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), n * sizeof(unsigned long long)));
    return WN_OK;
}
#endif

}  // namespace w16

#ifdef WN16_STAMPS
extern "C" int wn16_debug_stamps(unsigned long long* dst, int n) { return w16::debug_stamps(dst, n); }
#endif
