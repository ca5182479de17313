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
#ifdef WN16_MSTAMPS
// k16_bwd_multi: per phase (2 per layer) and workgroup: s_memtime at the phase's start, cycles wave 0 spent in dep_wait, polls
__device__ unsigned long long g_mstamps[100][256][4];
#endif

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
        // (inline asm as the interior tiles' requests: one builtin LDS-DMA anywhere in the loop and every LDS wait hipcc emits
        // in it is lgkmcnt(0) -- fragments in flight for the next k-step are waited for with the ones being consumed)
        dma_pieces<false, true>(xcur(buf), lane, 2 * w, 1, 2, [&](int r) {
            const int t = t0 + r < T ? t0 + r : T - 1;
            return xb + (long long)t * 128;
        });
        dma_pieces<false, true>(xold(buf), lane, 2 * w, 1, 2, [&](int r) {
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

// State of the per-tile dataflow words of k16_bwd_multi (the whole layer backward in ONE launch, below).  sync[0]: set when
// a wait gave up (the launch's results are void), sync[1]: a word that always reads "done", sync[2 + tile]: the number of
// gate-backward phases tile `tile` (b * tiles_per_b + X) has completed in this launch.
struct MSync {
    unsigned* sync;
    unsigned done;                      // gate phase: the value to publish for a finished tile; dx phase: the value a tile it reads must show
    unsigned dep_lds;                   // LDS byte address of this wave's 64 request words
    const unsigned* dep_ptr_lds;        // the same words as a pointer
    unsigned* cnt;                      // LDS: ring of four arrival counters (waves whose stores of a tile have been counted)
    bool gave_up;
#ifdef WN16_MSTAMPS
    unsigned long long dbg_cycles, dbg_polls, dbg_events;
#endif
};
static constexpr int kSyncHead = 2;
// this lane's request word, read as inline asm: a C++ read of LDS that the builtin LDS-DMA of the clamped-tile path "may alias"
// gets an s_waitcnt vmcnt(0) from hipcc in front of it -- which drained the previous tile's store in every loop body of the
// dx phase (+ 0.8 us per tile, in-kernel stamps); the loop's own counted wait covers the request that filled the word
__device__ __forceinline__ unsigned dep_word(const MSync& ms, int lane) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ms.dep_lds + 4u * (unsigned)lane) : "memory");
    return v;
}

// Live ranges (all multiples of 32; the host computes them, w16_api.hip).  The loss reaches the stack through skip[t_off:]
// only, so layer l receives gradient at columns t >= t_off - (reach of the layers above) and nowhere else:
//   gate phase: columns below t_live carry no gradient.  Tiles wholly below t_zero (<= t_live, a multiple of 64: the deferred
//               weight-gradient launch reads [da | dg] in 64-row chunks starting there) are NOT TOUCHED -- nothing loaded,
//               nothing stored; the dead tile between t_zero and t_live, if any, stores zeros.
//   dx phase:   dx is exactly zero below t_live (= the gate bound of the layer below); such tiles are not touched unless
//               zero_dead (layer 0: the embedding backward reads every row).  [da | dg](t) and dout(t) are zero below t_gate
//               (this layer's gate bound: rows nobody wrote) -- they are not read there, their LDS tiles are zero-filled.
struct GateP {
    const bf16* x; const bf16* convA; const bf16* dzA; const bf16* dout; const bf16* dzs; int dz_t0; bf16* dadg;
    int B, T, d, Z, tiles_per_b, ntiles, t_live, t_zero;
};
struct DxP {
    const bf16* dadg; const bf16* dxA; const bf16* dout; const bf16* zprev; bf16* dx; float* dwp_part;
    int B, T, d, tiles_per_b, ntiles, t_live, t_gate, zero_dead;
};

// what every request of a wave needs: its piece of a 32-row tile is rows 4 w .. 4 w + 3, the lane's 16 bytes at a fixed offset
// from the tile's first row (off128: 256-byte rows, off256: the same piece of 512-byte [da | dg] rows)
struct WaveC { int lane, w; unsigned off128, off256, lds0; };
__device__ __forceinline__ WaveC wave_consts(const char* lds) {
    WaveC c;
    c.lane = threadIdx.x & 63;
    c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = 4 * c.w + (c.lane >> 4);
    c.off128 = (unsigned)(r * 256 + (((c.lane & 15) ^ key(r)) << 4));
    c.off256 = c.off128 + (unsigned)(r * 256);
    c.lds0 = lds_addr_of(lds);
    return c;
}

// LDS maps, in 8 KB tiles.  The per-layer kernels pack their tiles (gate 10, dx 15).  k16_bwd_multi gives each phase a FIRST
// buffer no other phase touches -- gate tiles 0..3, dx tiles 4..9 -- so that the first tile of the next phase can be requested
// while the last tile of the current one is still being worked on; everything else shares tiles 10..18 (the phases never
// overlap beyond that first request); tile 19 holds the request words and the arrival counters.
template <bool MULTI> struct GateMap {
    static __device__ __forceinline__ int xold(int b) { return MULTI ? (b ? 10 : 0) : b; }
    static __device__ __forceinline__ int xcur(int b) { return MULTI ? (b ? 11 : 1) : 2 + b; }
    static __device__ __forceinline__ int dot(int b) { return MULTI ? (b ? 12 : 2) : 4 + b; }
    static __device__ __forceinline__ int dzt(int b) { return MULTI ? (b ? 13 : 3) : 6 + b; }
    static constexpr int dat = MULTI ? 14 : 8, dgt = MULTI ? 15 : 9;
};
template <bool MULTI> struct DxMap {
    static __device__ __forceinline__ int at(int b, int which) { return (MULTI ? (b ? 10 : 4) : 6 * b) + which; }   // da dg da' dg' dout z
    static constexpr int xch = MULTI ? 16 : 12, dxt = MULTI ? 18 : 14;
};
static constexpr int kMultiWords = 19;                 // tile 19: 8 x 64 request words, then the arrival counters
static constexpr int kMultiLds = 19 * kLTileB + 4096;

// the requests of one gate-backward tile (this wave's piece of x[t], x[t - d], dout, dz_skip)
template <bool HAS_DO, bool HAS_DZ, bool MULTI>
__device__ __forceinline__ void gate_issue(char* lds, const GateP& P, const WaveC& c, int tile, int buf) {
    using M = GateMap<MULTI>;
    if constexpr (MULTI) tile = __builtin_amdgcn_readfirstlane(tile);   // (uniform; behind a wait loop hipcc no longer proves it)
    const int T = P.T, d = P.d, dz_t0 = P.dz_t0, Tw = P.T - P.dz_t0, lane = c.lane, w = c.w;
    const int b = tile / P.tiles_per_b;
    const int t0 = (tile - b * P.tiles_per_b) * kLT;
    if (t0 + kLT <= P.t_live) return;                    // a dead tile
    if (t0 + kLT <= T && t0 >= d && (!HAS_DZ || t0 >= dz_t0)) {
        // interior tile (nothing to clamp): uniform base + the fixed lane offset, no 64-bit address arithmetic per
        // request -- the four requests cost a wave ~520 cycles of a ~5,900-cycle tile in the clamped form below
        const bf16* bc = P.x + ((long long)b * T + t0) * 128;
        dma16_s(bc, c.off128, c.lds0 + M::xcur(buf) * kLTileB + w * 1024);
        dma16_s(bc - (long long)d * 128, c.off128, c.lds0 + M::xold(buf) * kLTileB + w * 1024);
        if (HAS_DO) dma16_sx<MULTI>(P.dout + ((long long)b * T + t0) * 128, c.off128, c.lds0 + M::dot(buf) * kLTileB + w * 1024);
        if (HAS_DZ) dma16_s(P.dzs + ((long long)b * Tw + (t0 - dz_t0)) * 128, c.off128, c.lds0 + M::dzt(buf) * kLTileB + w * 1024);
        return;
    }
    const bf16* xb = P.x + (long long)b * T * 128;
    dma_pieces<false, true>(lds + M::xcur(buf) * kLTileB, lane, w, 1, 1, [&](int r) {
        const int t = t0 + r < T ? t0 + r : T - 1;
        return xb + (long long)t * 128;
    });
    dma_pieces<false, true>(lds + M::xold(buf) * kLTileB, lane, w, 1, 1, [&](int r) {
        int t = t0 + r < T ? t0 + r : T - 1;
        t = t - d >= 0 ? t - d : 0;
        return xb + (long long)t * 128;
    });
    if (HAS_DO) {
        const bf16* db = P.dout + (long long)b * T * 128;
        dma_pieces<MULTI, true>(lds + M::dot(buf) * kLTileB, lane, w, 1, 1, [&](int r) {
            const int t = t0 + r < T ? t0 + r : T - 1;
            return db + (long long)t * 128;
        });
    }
    if (HAS_DZ) {
        const bf16* zb = P.dzs + (long long)b * Tw * 128;       // dz_skip exists for the loss window only
        dma_pieces<false, true>(lds + M::dzt(buf) * kLTileB, lane, w, 1, 1, [&](int r) {
            int t = t0 + r < T ? t0 + r : T - 1;
            t = t - dz_t0 >= 0 ? t - dz_t0 : 0;
            return zb + (long long)t * 128;
        });
    }
}

// the requests of one dx tile (this wave's piece of [da | dg](t), [da | dg](t + d), dout, z of the layer below)
template <bool HAS_DO, bool HAS_Z, bool MULTI>
__device__ __forceinline__ void dx_issue(char* lds, const DxP& P, const WaveC& c, int tile, int buf) {
    using M = DxMap<MULTI>;
    if constexpr (MULTI) tile = __builtin_amdgcn_readfirstlane(tile);
    const int T = P.T, d = P.d, lane = c.lane, w = c.w;
    const int b = tile / P.tiles_per_b;
    const int t0 = (tile - b * P.tiles_per_b) * kLT;
    if (t0 + kLT <= P.t_live) return;                    // a dead tile
    const bool tlive = t0 + kLT > P.t_gate;            // the tile's own rows of [da | dg] and dout exist (else: zeros, dx_phase fills them in)
    if (t0 + kLT + d <= T) {                           // interior tile: uniform bases + fixed lane offsets (as in gate_issue)
        const bf16* a0 = P.dadg + ((long long)b * T + t0) * 256;
        const bf16* a1 = a0 + (long long)d * 256;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (tlive) dma16_sx<MULTI>(a0 + 128 * half, c.off256, c.lds0 + M::at(buf, half) * kLTileB + w * 1024);
            dma16_sx<MULTI>(a1 + 128 * half, c.off256, c.lds0 + M::at(buf, 2 + half) * kLTileB + w * 1024);
        }
        if (HAS_DO && tlive) dma16_sx<MULTI>(P.dout + ((long long)b * T + t0) * 128, c.off128, c.lds0 + M::at(buf, 4) * kLTileB + w * 1024);
        if (HAS_Z) dma16_s(P.zprev + ((long long)b * T + t0) * 128, c.off128, c.lds0 + M::at(buf, 5) * kLTileB + w * 1024);
        return;
    }
    const bf16* ab = P.dadg + (long long)b * T * 256;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (tlive) dma_pieces<MULTI, true>(lds + M::at(buf, half) * kLTileB, lane, w, 1, 1, [&](int r) {
            const int t = t0 + r < T ? t0 + r : T - 1;
            return ab + (long long)t * 256 + 128 * half;
        });
        dma_pieces<MULTI, true>(lds + M::at(buf, 2 + half) * kLTileB, lane, w, 1, 1, [&](int r) {
            int t = t0 + r + d;
            t = t < T ? t : T - 1;
            return ab + (long long)t * 256 + 128 * half;
        });
    }
    if (HAS_DO && tlive) {
        const bf16* db = P.dout + (long long)b * T * 128;
        dma_pieces<MULTI, true>(lds + M::at(buf, 4) * kLTileB, lane, w, 1, 1, [&](int r) {
            const int t = t0 + r < T ? t0 + r : T - 1;
            return db + (long long)t * 128;
        });
    }
    if (HAS_Z) {
        const bf16* zb = P.zprev + (long long)b * T * 128;
        dma_pieces<false, true>(lds + M::at(buf, 5) * kLTileB, lane, w, 1, 1, [&](int r) {
            const int t = t0 + r < T ? t0 + r : T - 1;
            return zb + (long long)t * 128;
        });
    }
}

// MULTI: the dataflow words of the [da | dg] tiles a dx tile reads from ANOTHER workgroup (rows t0 + d .. t0 + 31 + d: lanes
// 0, 1; every other lane, and every word that imposes nothing -- a dead tile, rows beyond the clip -- reads the word that
// always says "done")
__device__ __forceinline__ const unsigned* dx_dep_ptr(const DxP& P, const MSync& ms, int lane, int tile) {
    const unsigned* p = ms.sync + 1;
    const int b = tile / P.tiles_per_b;
    const int t0 = (tile - b * P.tiles_per_b) * kLT;
    if (t0 + kLT > P.t_live) {
        int q = -1;
        if (lane == 0) q = (t0 + P.d) >> 5;
        else if (lane == 1) q = (t0 + kLT - 1 + P.d) >> 5;
        if (q >= 0 && q < P.tiles_per_b) p = ms.sync + kSyncHead + b * P.tiles_per_b + q;
    }
    return p;
}

// This wave's stores of the gate tile of loop body `idx` have been counted by vmcnt: the wave whose arrival is the eighth
// publishes the tile's word (one sc1 store).  No barrier is involved, so a wave may arrive whenever its own counted wait
// allows -- in particular BEFORE it starts spinning on somebody else's word (a workgroup never waits while it sits on
// finished, unpublished tiles: no cycle).
__device__ __forceinline__ void gate_arrive(MSync& ms, int lane, int idx, int tile) {
    if (lane == 0) {
        unsigned* cn = ms.cnt + (idx & 3);
        const unsigned old = __hip_atomic_fetch_add(cn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (old == 7u) {
            __hip_atomic_store(cn, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(ms.sync + kSyncHead + tile, ms.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// what a phase does for the phase that FOLLOWS it in k16_bwd_multi: request() in the second-to-last loop body (the words the
// follower's first tile depends on), issue() in the last one (the follower's first tile itself, if those words allow)
struct NoTail {
    __device__ __forceinline__ void request() {}
    __device__ __forceinline__ bool issue() { return false; }
};

// One layer's gate backward over this workgroup's tiles.  MULTI (k16_bwd_multi): dout was written earlier in the same launch
// (by this workgroup's own dx phase of the layer above, write-through) and is requested with sc1; a finished tile's word is
// published once every wave's vmcnt has counted its stores of the tile -- two loop bodies later, where the body's own counted
// wait says so (gate_arrive); the phase ends WITHOUT draining its stores: the last two tiles arrive at the start of the dx
// phase.  pre_issued: the first tile's requests are already in flight (the previous phase's tail).  Returns whether the
// tail issued the follower's first tile.
template <bool HAS_DO, bool HAS_DZ, bool MULTI, class Tail>
__device__ __forceinline__ bool gate_phase(char* lds, const GateP& P, MSync& ms, bool pre_issued, Tail& tail) {
    using M = GateMap<MULTI>;
    const bf16* __restrict__ convA = P.convA; const bf16* __restrict__ dzA = P.dzA; bf16* __restrict__ dadg = P.dadg;
    const int dz_t0 = P.dz_t0, T = P.T, d = P.d, Z = P.Z, tiles_per_b = P.tiles_per_b, ntiles = P.ntiles, t_live = P.t_live;
    // t_live (a multiple of 32): no gradient reaches this layer's columns below it (they are further from the loss window than
    // the layers above can see).  Such tiles load and compute nothing: they store the zeros their readers expect.
    auto xold = [&](int buf) { return lds + M::xold(buf) * kLTileB; };
    auto xcur = [&](int buf) { return lds + M::xcur(buf) * kLTileB; };
    auto dot = [&](int buf) { return lds + M::dot(buf) * kLTileB; };
    auto dzt = [&](int buf) { return lds + M::dzt(buf) * kLTileB; };
    char* dat = lds + M::dat * kLTileB;
    char* dgt = lds + M::dgt * kLTileB;
    const WaveC c = wave_consts(lds);
    const int lane = c.lane, w = c.w;
    const int j = lane & 31, h = lane >> 5;
    int first, stride, last;
    tile_range(ntiles, first, stride, last);
    const int n = first < last ? (last - first + stride - 1) / stride : 0;     // this workgroup's tiles
    auto issue = [&](int tile, int buf) { gate_issue<HAS_DO, HAS_DZ, MULTI>(lds, P, c, tile, buf); };
    constexpr int kStores = 2;                          // da and dg pieces per wave and tile

    // the first tile's operands are requested BEFORE this wave's weights (register-resident A operands) so that the two
    // round trips overlap; the weights are waited for here (left pending, the compiler's s_waitcnt vmcnt(0) would sit at their
    // first use inside the loop and drain the LDS-DMA prefetch of every iteration)
    if (n > 0 && !(MULTI && pre_issued)) issue(first, 0);
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
    bool full_prev = false;
    bool tail_issued = false;
    int it = 0;
    for (int tile = first; tile < last; tile += stride, ++it) {
        const int buf = it & 1;
        const int b = tile / tiles_per_b;
        const int t0 = (tile - b * tiles_per_b) * kLT;
        STAMP_K(1, 0);
        if (full_prev) wait_vm<kStores>(); else wait_vm<0>();
        STAMP_K(1, 1);
        // in flight now: at most this wave's stores of the previous tile -- those of the tile before it have been counted
        if constexpr (MULTI) { if (it >= 2) gate_arrive(ms, lane, it - 2, tile - 2 * stride); }
        barrier();
        STAMP_K(1, 2);
        if (tile + stride < last) issue(tile + stride, buf ^ 1);
        if constexpr (MULTI) {
            if (n >= 3 && it == n - 2) tail.request();
            if (n >= 3 && it == n - 1) tail_issued = tail.issue();
        }
        STAMP_K(1, 3);
        if (t0 + kLT <= t_live) {                          // dead tile (workgroup-uniform)
            if (t0 + kLT <= P.t_zero) {                    // ... that nobody reads: not touched
                full_prev = false;                         // (no stores: the next body's wait covers its requests in full)
                continue;
            }
            const int r = 4 * w + (lane >> 4);             // [da | dg] = 0, two stores per wave as below
            const int cc = (lane & 15) ^ key(r);
            bf16* o = dadg + ((long long)b * T + t0 + r) * 256 + cc * 8;
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
            const int cc = (lane & 15) ^ key(r);
            const u32x4 va = *reinterpret_cast<const u32x4*>(dat + w * 1024 + lane * 16);
            const u32x4 vg = *reinterpret_cast<const u32x4*>(dgt + w * 1024 + lane * 16);
            bf16* o = dadg + ((long long)b * T + t0 + r) * 256 + cc * 8;
            if (full || t0 + r < T) {
                st16_wt(o, va);
                st16_wt(o + 128, vg);
            }
        }
        full_prev = full;
        STAMP_K(1, 8);
    }
    if constexpr (MULTI) {
        // end of the phase.  With three tiles or more nothing is drained: the follower's first tile reads rows this workgroup
        // stored in body 0 (counted since body 2), and the last two tiles arrive at the follower's start.  LDS: the barrier
        // below orders this phase's last reads before the follower's first writes of the shared tiles.
        if (n < 3) wait_vm<0>();
        barrier();
    }
    return tail_issued;
}

template <bool HAS_DO, bool HAS_DZ>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void k16_gate_bwd(
    const bf16* __restrict__ x, const bf16* __restrict__ convA, const bf16* __restrict__ dzA,
    const bf16* __restrict__ dout, const bf16* __restrict__ dzs, int dz_t0, bf16* __restrict__ dadg, int B, int T, int d,
    int Z, int tiles_per_b, int ntiles, int t_live, int t_zero) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const GateP P{x, convA, dzA, dout, dzs, dz_t0, dadg, B, T, d, Z, tiles_per_b, ntiles, t_live, t_zero};
    MSync ms{};
    NoTail nt;
    gate_phase<HAS_DO, HAS_DZ, false>(lds, P, ms, false, nt);
}

// ---------------------------------------------------------------------------------------------
// dx + dWp.  512 threads, one workgroup per CU (no VALU-heavy phase here).  Wave (mt = w & 3, kh = w >> 2): output rows
// 32 mt .., contraction half kh (0: dab[t] with the tap-1 weights, 1: dab[t + d] with the tap-0 weights); the halves
// meet in an fp32 LDS patch.  LDS: (da, dg, da', dg', dout, z) x 2 buffers = 96 KB + 16 KB exchange + 8 KB dx tile
// ---------------------------------------------------------------------------------------------
static constexpr int kDxLds = 12 * kLTileB + 16384 + kLTileB;
static constexpr int kDwpPart = 128 * 128;             // floats per workgroup partial of dWp

// One layer's dx (+ the projection gradient of the layer below) over this workgroup's tiles.  MULTI (k16_bwd_multi): [da | dg]
// and dout were written earlier in the same launch (write-through) and are requested with sc1; a tile's requests go out only
// when the words of the [da | dg] tiles it reads -- rows t0 + d .. t0 + 31 + d, another workgroup's -- show this layer's gate
// phase done.  The words of the tile after next are requested into LDS one loop body ahead (dma4_sc1, covered by the body's
// own counted wait), so the test costs no memory round trip unless a tile really is late.  The phase starts by letting the
// last two tiles of the gate phase arrive (this wave's weight loads have drained vmcnt by then) and ends without a drain.
template <bool HAS_DO, bool HAS_Z, bool MULTI, class Tail>
__device__ __forceinline__ bool dx_phase(char* lds, const DxP& P, MSync& ms, bool pre_issued, Tail& tail) {
    using M = DxMap<MULTI>;
    const bf16* __restrict__ dxA = P.dxA; bf16* __restrict__ dx = P.dx; float* __restrict__ dwp_part = P.dwp_part;
    const int T = P.T, d = P.d, tiles_per_b = P.tiles_per_b, ntiles = P.ntiles, t_live = P.t_live;
    // t_live (a multiple of 32): dx is exactly zero below it (no gradient reaches those columns); such tiles store zeros
    auto tile_at = [&](int buf, int which) { return lds + M::at(buf, which) * kLTileB; };   // da dg da' dg' dout z
    float* xch = reinterpret_cast<float*>(lds + M::xch * kLTileB);
    char* dxt = lds + M::dxt * kLTileB;
    const WaveC c = wave_consts(lds);
    const int lane = c.lane, w = c.w;
    const int j = lane & 31, h = lane >> 5;
    const int mt = w & 3, kh = w >> 2;
    int first, stride, last;
    tile_range(ntiles, first, stride, last);
    const int n = first < last ? (last - first + stride - 1) / stride : 0;     // this workgroup's tiles

    // dWp[cr][cd] partial: wave w owns rows cr 32 (w & 3) .. + 31, columns cd 64 (w >> 2) .. + 63
    f32x16 wp[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { wp[0][r] = 0.f; wp[1][r] = 0.f; }

    auto issue = [&](int tile, int buf) { dx_issue<HAS_DO, HAS_Z, MULTI>(lds, P, c, tile, buf); };
    auto dep_request_lds = [&](int tile) {
        if constexpr (MULTI) dma4_sc1(dx_dep_ptr(P, ms, lane, tile), ms.dep_lds);
    };
    auto dep_wait = [&](int tile, unsigned v) {
        if constexpr (MULTI) {
            unsigned spins = 0;
#ifdef WN16_MSTAMPS
            const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
            // (every polled value is USED before the loop can be left: a load still pending at the exit would make hipcc put
            // s_waitcnt vmcnt(0) in front of the next write of that register -- the next body's dep_word -- and with it
            // drain the store every body leaves in flight on purpose)
            asm volatile("" ::"v"(v));                   // (the first poll too: also on the path that has given up)
            bool late = !ms.gave_up && __builtin_amdgcn_ballot_w64(v < ms.done) != 0ull;
            while (late) {
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(dx_dep_ptr(P, ms, lane, tile), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                late = __builtin_amdgcn_ballot_w64(v < ms.done) != 0ull;
                if (++spins > (1u << 18)) {              // never hang the GPU: give up for good and say so (the results are void)
                    ms.gave_up = true;
                    if (lane == 0) __hip_atomic_store(ms.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    late = false;
                }
            }
#ifdef WN16_MSTAMPS
            if (spins) { ms.dbg_cycles += __builtin_amdgcn_s_memtime() - w0; ms.dbg_polls += spins; ms.dbg_events += 1; }
#endif
        }
    };

    // per-layer launch: the first tile's requests, then this wave's A operands (the two round trips overlap).  MULTI: the
    // operands first -- their wait drains vmcnt, so the gate phase's last two tiles can arrive (and only then may this wave
    // spin on a word) --, then the first tile unless the gate phase's tail has already requested it
    if (!MULTI && n > 0) issue(first, 0);
    bf16x8 A[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) A[s] = *reinterpret_cast<const bf16x8*>(dxA + (((mt * 2 + kh) * 16 + s) * 64 + lane) * 8);
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" ::"v"(A[s]));
    if constexpr (MULTI) {
        wait_vm<0>();                                    // (already drained by the operands' wait; says so to the compiler too)
        if (n >= 2) gate_arrive(ms, lane, n - 2, first + (n - 2) * stride);
        if (n >= 1) gate_arrive(ms, lane, n - 1, first + (n - 1) * stride);
        if (n > 0) {
            if (!pre_issued) {
                dep_wait(first, __hip_atomic_load(dx_dep_ptr(P, ms, lane, first), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                issue(first, 0);
            }
            if (n > 1) dep_request_lds(first + stride);
        }
    }
    bool full_prev = false;
    bool tail_issued = false;
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
        if (tile + stride < last) {
            if constexpr (MULTI) {
                // the words requested a body ago have landed (they are older than the store the wait above left in flight)
                dep_wait(tile + stride, dep_word(ms, lane));
            }
            issue(tile + stride, buf ^ 1);
            if constexpr (MULTI) { if (tile + 2 * stride < last) dep_request_lds(tile + 2 * stride); }
        }
        if constexpr (MULTI) {
            if (n >= 3 && it == n - 1) tail_issued = tail.issue();
        }
        STAMP_K(2, 3);
        if (t0 + kLT <= t_live) {                          // dead tile (workgroup-uniform)
            if (!P.zero_dead) {                            // ... that nobody reads: not touched
                full_prev = false;
                continue;
            }
            const int r = 4 * w + (lane >> 4);             // dx = 0, one store per wave as below
            const int cc = (lane & 15) ^ key(r);
            const u32x4 zero4 = {0u, 0u, 0u, 0u};
            st16_wt(dx + ((long long)b * T + t0 + r) * 128 + cc * 8, zero4);
            full_prev = true;
            continue;
        }
        {
            // rows that are zero by construction and were never written (or never requested): dab[t + d] beyond the clip end
            // and below this layer's gate bound; the tile's own [da | dg] and dout when it lies below the gate bound
            const bool hi_fix = t0 + kLT + d > T, lo_fix = t0 + d < P.t_gate, own_dead = t0 + kLT <= P.t_gate;
            if (hi_fix || lo_fix || own_dead) {
                for (int r = w; r < kLT; r += 8) {
                    if (t0 + r + d >= T || t0 + r + d < P.t_gate) {
                        *reinterpret_cast<unsigned*>(tile_at(buf, 2) + r * 256 + lane * 4) = 0u;
                        *reinterpret_cast<unsigned*>(tile_at(buf, 3) + r * 256 + lane * 4) = 0u;
                    }
                    if (own_dead) {
                        *reinterpret_cast<unsigned*>(tile_at(buf, 0) + r * 256 + lane * 4) = 0u;
                        *reinterpret_cast<unsigned*>(tile_at(buf, 1) + r * 256 + lane * 4) = 0u;
                        if (HAS_DO) *reinterpret_cast<unsigned*>(tile_at(buf, 4) + r * 256 + lane * 4) = 0u;
                    }
                }
                barrier();
            }
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
            const int cc = (lane & 15) ^ key(r);
            const u32x4 v = *reinterpret_cast<const u32x4*>(dxt + w * 1024 + lane * 16);
            if (full || t0 + r < T) st16_wt(dx + ((long long)b * T + t0 + r) * 128 + cc * 8, v);
        }
        full_prev = full;
        STAMP_K(2, 9);
    }
    if (HAS_Z) {
        // this workgroup's partial of dWp: [cr][cd] fp32, summed over workgroups by k16_reduce_parts
        float* o = dwp_part + (long long)blockIdx.x * kDwpPart;
        if constexpr (MULTI) {
            // a dataflow wait that gave up means this launch's results are void (tiles were consumed before they were produced):
            // make that LOUD, like the fp32 chain does (mfma_layer_bwd.hip, acc_to_lds) -- a NaN in this layer's dWp, so the
            // gradient norm is not finite and wn_adam_step skips the step (ABI 4's rule) instead of applying it silently.
            // gave_up is sticky: every layer after the wait that failed is poisoned.  (sync[0] = 1 says the same.)
            if (ms.gave_up) wp[0][0] = __builtin_nanf("");
        }
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                o[(32 * (w & 3) + acc_row(r, h)) * 128 + 64 * (w >> 2) + 32 * n + j] = wp[n][r];
    }
    if constexpr (MULTI) {
        // end of the phase (see gate_phase): this workgroup's dx rows are the dout of its own next gate phase, whose first
        // tile reads what body 0 stored
        if (n < 3) wait_vm<0>();
        barrier();
    }
    return tail_issued;
}

template <bool HAS_DO, bool HAS_Z>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void k16_dx(const bf16* __restrict__ dadg, const bf16* __restrict__ dxA,
                                                  const bf16* __restrict__ dout, const bf16* __restrict__ zprev,
                                                  bf16* __restrict__ dx, float* __restrict__ dwp_part, int B, int T, int d,
                                                  int tiles_per_b, int ntiles, int t_live, int t_gate, int zero_dead) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const DxP P{dadg, dxA, dout, zprev, dx, dwp_part, B, T, d, tiles_per_b, ntiles, t_live, t_gate, zero_dead};
    MSync ms{};
    NoTail nt;
    dx_phase<HAS_DO, HAS_Z, false>(lds, P, ms, false, nt);
}

// ---------------------------------------------------------------------------------------------
// The layer backward of stack layers l_hi .. l_lo (descending) in ONE launch of co-resident workgroups: gate_phase and
// dx_phase -- the tile code of k16_gate_bwd / k16_dx unchanged -- alternate per layer, each workgroup keeps the tiles
// tile_range() deals it for the whole launch, and nothing resembling a grid barrier exists.  What a phase reads that the
// launch itself wrote:
//   gate(l), tile X:  dout[X] = dx(l + 1)[X]                     -- this workgroup's own earlier phase (its stores are counted
//                                                                   by the vmcnt(0) that ends every phase);
//   dx(l), tile X:    [da | dg](l)[X]                            -- the same;
//                     [da | dg](l) rows 32 X + d .. 32 X + 31 + d -- ANOTHER workgroup's tiles: one word per tile in device
//                     memory holds the number of gate phases the tile has completed, a tile's requests go out when the
//                     words it depends on show this layer's phase (MSync, dep_wait);
//   dx(l) WRITES dxb[l & 1], which gate(l + 1) and dx(l + 1) of the SAME tile read (as dout) -- both precede dx(l) of that
//                     tile in this workgroup's own program order, so the two buffers need no test.
// gate phases wait for nothing another workgroup does and dx phases only for gate phases, so there is no cycle; every
// workgroup must be RESIDENT (the launcher checks the occupancy of the device the launch goes to).  All data that crosses
// workgroups inside the launch is stored write-through (st16_wt, sc1) and requested with sc1.  Same tiles, same per-tile
// arithmetic, same per-workgroup dWp partial tiles as the per-layer launches with the same grid: bit-identical results.
// ---------------------------------------------------------------------------------------------
struct BwdMultiArgs {
    const bf16* x0; const bf16* xs; const bf16* z; const bf16* img; const bf16* dzs;
    bf16* dadg; bf16* dxb[2]; float* parts; long long part_stride; unsigned* sync;
    long long n, nw;                                     // B T, B (T - dz_t0)
    int d[kMaxProb16], Z[kMaxProb16], live_gate[kMaxProb16], live_dx[kMaxProb16], zero_gate[kMaxProb16];   // by stack layer
    int l_hi, l_lo, B, T, dz_t0, tiles_per_b, ntiles;
};

// the gate phase's tail: the dx phase's first tile (its words requested one body earlier; not issued if they do not yet
// show this layer's gate phase -- the dx phase then waits for them itself, after the last two gate tiles have arrived)
struct DxTail {
    char* lds; const DxP* P; const WaveC* c; MSync* ms; int first;
    __device__ __forceinline__ void request() { dma4_sc1(dx_dep_ptr(*P, *ms, c->lane, first), ms->dep_lds); }
    __device__ __forceinline__ bool issue() {
        const unsigned v = dep_word(*ms, c->lane);
        if (__builtin_amdgcn_ballot_w64(v < ms->done) != 0ull) return false;
        dx_issue<true, true, true>(lds, *P, *c, first, 0);
        return true;
    }
};
// the dx phase's tail: the first tile of the NEXT layer's gate phase (it reads nothing another workgroup writes)
struct GateTail {
    char* lds; const GateP* P; const WaveC* c; int first; bool active;
    __device__ __forceinline__ void request() {}
    __device__ __forceinline__ bool issue() {
        if (!active) return false;
        gate_issue<true, true, true>(lds, *P, *c, first, 0);
        return true;
    }
};

__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void k16_bwd_multi(const BwdMultiArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const WaveC c = wave_consts(lds);
    MSync ms;
    ms.sync = a.sync;
    ms.gave_up = false;
    ms.dep_lds = lds_addr_of(lds + kMultiWords * kLTileB + c.w * 256);
    ms.dep_ptr_lds = reinterpret_cast<const unsigned*>(lds + kMultiWords * kLTileB + c.w * 256);
    ms.cnt = reinterpret_cast<unsigned*>(lds + kMultiWords * kLTileB + 2048);
    if (threadIdx.x < 4) ms.cnt[threadIdx.x] = 0u;
    barrier();
    int first, stride, last;
    tile_range(a.ntiles, first, stride, last);
    auto gate_args = [&](int l) {
        const bf16* in = uniform_ptr(l == 0 ? a.x0 : a.xs + (long long)(l - 1) * a.n * 128);
        const bf16* img = uniform_ptr(a.img + (long long)l * kLayerImg);
        return GateP{in, img + kOffConvA8, img + kOffDzA8, uniform_ptr(a.dxb[(l + 1) & 1]),
                     uniform_ptr(a.dzs + (long long)l * a.nw * 128), a.dz_t0, uniform_ptr(a.dadg + (long long)l * a.n * 256),
                     a.B, a.T, a.d[l], a.Z[l], a.tiles_per_b, a.ntiles, a.live_gate[l], a.zero_gate[l]};
    };
    bool pre = false;
    for (int l = a.l_hi; l >= a.l_lo; --l) {
        ms.done = (unsigned)(a.l_hi - l + 1);
        const GateP g = gate_args(l);
        const DxP x{g.dadg, uniform_ptr(a.img + (long long)l * kLayerImg) + kOffDxA, g.dout,
                    uniform_ptr(a.z + (long long)(l - 1) * a.n * 128), uniform_ptr(a.dxb[l & 1]),
                    a.parts + (long long)(l - 1) * a.part_stride, a.B, a.T, a.d[l], a.tiles_per_b, a.ntiles, a.live_dx[l],
                    a.live_gate[l], 0};
        DxTail dt{lds, &x, &c, &ms, first};
#ifdef WN16_MSTAMPS
        const int ph = 2 * (a.l_hi - l);
        if (threadIdx.x == 0 && blockIdx.x < 256) g_mstamps[ph][blockIdx.x][0] = __builtin_amdgcn_s_memtime();
#endif
        const bool pre_dx = gate_phase<true, true, true>(lds, g, ms, pre, dt);
        const GateP gn = gate_args(l > a.l_lo ? l - 1 : l);
        GateTail gt{lds, &gn, &c, first, l > a.l_lo};
#ifdef WN16_MSTAMPS
        if (threadIdx.x == 0 && blockIdx.x < 256) { g_mstamps[ph + 1][blockIdx.x][0] = __builtin_amdgcn_s_memtime(); g_mstamps[ph + 1][blockIdx.x][3] = pre_dx; }
        ms.dbg_cycles = ms.dbg_polls = ms.dbg_events = 0;
#endif
        pre = dx_phase<true, true, true>(lds, x, ms, pre_dx, gt);
#ifdef WN16_MSTAMPS
        if (threadIdx.x == 0 && blockIdx.x < 256) {
            g_mstamps[ph + 1][blockIdx.x][1] = ms.dbg_cycles; g_mstamps[ph + 1][blockIdx.x][2] = (ms.dbg_events << 32) | ms.dbg_polls;
            g_mstamps[ph + 2][blockIdx.x][0] = __builtin_amdgcn_s_memtime();
        }
#endif
    }
}

__global__ void k16_zero_sync(unsigned* sync, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) sync[i] = i == 1 ? 0xffffffffu : 0u;
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
                   int d, int Z, int t_live, int t_zero, hipStream_t s) {
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
                           img + kOffDzA8, dout, dzs, dz_t0, dadg, B, T, d, Z, tiles_per_b, ntiles, t_live, t_zero);  \
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
             int T, int d, int t_live, int t_gate, int zero_dead, hipStream_t s) {
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
                           zprev, dx, dwp_part, B, T, d, tiles_per_b, ntiles, t_live, t_gate, zero_dead);             \
    } while (0)
    if (dout && zprev) DX_LAUNCH(true, true);
    else if (dout) DX_LAUNCH(true, false);
    else if (zprev) DX_LAUNCH(false, true);
    else DX_LAUNCH(false, false);
#undef DX_LAUNCH
    WN_LAUNCH_CHECK();
    return WN_OK;
}

size_t bwd_multi_sync_words(int B, int T) { return (size_t)kSyncHead + (size_t)B * ((T + kLT - 1) / kLT); }

// 1 if `grid` workgroups of k16_bwd_multi are all resident on the device the current stream belongs to (they wait for
// each other's tiles).  Asked per device: the occupancy of the kernel's 512 threads / kMultiLds bytes x the device's CUs.
static bool bwd_multi_resident(int grid) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    static int capacity[64];                            // per device; 0 = not asked yet (a benign race: every thread writes the same value)
    if (!capacity[dev]) {
        int n_cu = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k16_bwd_multi), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kMultiLds) != hipSuccess) return false;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k16_bwd_multi, 512, kMultiLds) != hipSuccess) return false;
        capacity[dev] = n_cu * per_cu > 0 ? n_cu * per_cu : -1;
    }
    return grid <= capacity[dev];
}

// 1 if layers l_hi .. l_lo can go into one launch at this size (the caller falls back to per-layer launches otherwise)
int bwd_multi_ok(int B, int T) {
    const int ntiles = B * ((T + kLT - 1) / kLT);
    return ntiles >= 8 && bwd_multi_resident(grid_for(ntiles, 1)) ? 1 : 0;
}

int bwd_multi(const bf16* x0, const bf16* xs, const bf16* z, const bf16* img, const bf16* dzs, bf16* dadg, bf16* dxb0,
              bf16* dxb1, float* parts, long long part_stride, unsigned* sync, const int* d, const int* Z,
              const int* live_gate, const int* live_dx, const int* zero_gate, int l_hi, int l_lo, int B, int T, int dz_t0,
              hipStream_t s) {
    if (l_hi >= kMaxProb16 || l_lo < 1 || l_hi < l_lo) { wn::set_error("w16 bwd_multi: layers %d..%d", l_hi, l_lo); return WN_EARG; }
    BwdMultiArgs a{};
    a.x0 = x0; a.xs = xs; a.z = z; a.img = img; a.dzs = dzs; a.dadg = dadg; a.dxb[0] = dxb0; a.dxb[1] = dxb1;
    a.parts = parts; a.part_stride = part_stride; a.sync = sync;
    a.tiles_per_b = (T + kLT - 1) / kLT;
    a.ntiles = B * a.tiles_per_b;
    a.n = (long long)B * T; a.nw = (long long)B * (T - dz_t0);
    for (int l = l_lo; l <= l_hi; ++l) {
        a.d[l] = d[l]; a.Z[l] = Z[l]; a.live_gate[l] = live_gate[l]; a.live_dx[l] = live_dx[l]; a.zero_gate[l] = zero_gate[l];
    }
    a.l_hi = l_hi; a.l_lo = l_lo; a.B = B; a.T = T; a.dz_t0 = dz_t0;
    const int grid = grid_for(a.ntiles, 1);
    if (!bwd_multi_resident(grid)) { wn::set_error("w16 bwd_multi: %d workgroups are not all resident", grid); return WN_ESHAPE; }
    // the words are zeroed by a KERNEL (a memset node is not ordered before the launch when graph replays follow each other
    // without a host synchronisation: DESIGN.md, round 4)
    const int nsync = (int)bwd_multi_sync_words(B, T);
    hipLaunchKernelGGL(k16_zero_sync, dim3((nsync + 255) / 256), dim3(256), 0, s, sync, nsync);
    WN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k16_bwd_multi, dim3(grid), dim3(512), kMultiLds, s, a);
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

#ifdef WN16_MSTAMPS
extern "C" __attribute__((visibility("default"))) int wn16_debug_mstamps(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(w16::g_mstamps), sizeof(unsigned long long) * 100 * 256 * 4);
}
#endif
#ifdef WN16_STAMPS
extern "C" __attribute__((visibility("default"))) int wn16_debug_stamps(unsigned long long* dst, int n) { return w16::debug_stamps(dst, n); }
#endif
