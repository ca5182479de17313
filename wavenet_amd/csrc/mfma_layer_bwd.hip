// Backward of the fused residual layer on the fp32 matrix cores, Cr = Cd = 32, filter width 2
// (Chainer autograd through ResidualConvLayer.__call__, wavenet.py:358-368).
//
// Pass 1 (k_layer_bwd_p1), one wave per tile of 32 time columns:
//     dz = Wp^T dout + dz_skip          D[cd][t], time on lanes, same channel permutation as forward
//     da = dz g (1 - f^2),  dg = dz f g (1 - g)   (0 in the reference's zero prefix t < Z)
//     da, dg -> scratch (B,T,64) for pass 2
//   and the weight gradients, where the MFMA contraction runs over TIME: the tile's da / dg are
//   transposed through a per-wave LDS patch (channel on lanes), x[t], x[t-d], dout, f*g are read
//   straight from HBM/L2 in channel-on-lane order (one coalesced 128-byte row per half wave):
//     dWf_k += da x[t-(1-k)d]^T,  dWg_k += dg x[t-(1-k)d]^T,  dWp += dout z^T
//   Five 32x32 accumulators stay in registers over all tiles of a wave, are summed over the
//   workgroup's waves in LDS and leave as one set of float atomics per workgroup.
// Pass 2 (k_layer_bwd_p2):  dx[t] = dout[t] + [Wf1;Wg1]^T dab[t] + [Wf0;Wg0]^T dab[t+d]
//   -- the forward kernel's structure with transposed weights (64 MFMAs per tile).
#include <cstdlib>
#include <cstring>

#include "h2_ops.hpp"
#include "wn_kernels.hpp"

namespace wn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int bch(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

static constexpr int kPad = 36;                 // LDS row stride (floats) of a transposed 32x32 patch
static constexpr int kWaves = 4;
static constexpr int kPartFloats = 5 * 16 * 64;   // five 32x32 accumulator tiles per workgroup
static constexpr int kMaxBlocks = 512;

template <bool HAS_DO, bool HAS_DZ>
__global__ __launch_bounds__(256, 2) void k_layer_bwd_p1(
    const float* __restrict__ x, const float* __restrict__ f, const float* __restrict__ g,
    const float* __restrict__ Wp, const float* __restrict__ dout, const float* __restrict__ dzs,
    float* __restrict__ dab, float* __restrict__ part, int B, int T, int d, int Z, int tiles_per_b, int ntiles) {
    __shared__ __attribute__((aligned(16))) float lds[kWaves * 2 * 32 * kPad + 5 * 16 * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    float* lda = lds + wv * (2 * 32 * kPad);
    float* ldg = lda + 32 * kPad;
    float* red = lds + kWaves * 2 * 32 * kPad;
    const int wave = blockIdx.x * kWaves + wv;
    const int nwaves = gridDim.x * kWaves;
    float wpT[16];                                   // A operand of dz: lane (i=cd,h), step s: Wp[ch(s,h)][i]
#pragma unroll
    for (int s = 0; s < 16; ++s) wpT[s] = Wp[bch(s, h) * 32 + j];

    f32x16 aWf0, aWf1, aWg0, aWg1, aWp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { aWf0[r] = 0.f; aWf1[r] = 0.f; aWg0[r] = 0.f; aWg1[r] = 0.f; aWp[r] = 0.f; }

    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int b = tile / tiles_per_b;
        const int t0 = (tile - b * tiles_per_b) * 32;
        const int t = t0 + j;
        const bool valid = t < T;
        const long long rowc = ((long long)b * T + (valid ? t : T - 1)) * 32 + 4 * h;   // clamped: loads are unconditional
        f32x16 acc;
        float ff[16], gg[16], dob[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 z4 = make_float4(0, 0, 0, 0), o4 = z4;
            const float4 f4 = *reinterpret_cast<const float4*>(f + rowc + 8 * q);
            const float4 g4 = *reinterpret_cast<const float4*>(g + rowc + 8 * q);
            if (HAS_DZ) z4 = *reinterpret_cast<const float4*>(dzs + rowc + 8 * q);
            if (HAS_DO) o4 = *reinterpret_cast<const float4*>(dout + rowc + 8 * q);
            acc[4 * q] = z4.x; acc[4 * q + 1] = z4.y; acc[4 * q + 2] = z4.z; acc[4 * q + 3] = z4.w;
            ff[4 * q] = f4.x; ff[4 * q + 1] = f4.y; ff[4 * q + 2] = f4.z; ff[4 * q + 3] = f4.w;
            gg[4 * q] = g4.x; gg[4 * q + 1] = g4.y; gg[4 * q + 2] = g4.z; gg[4 * q + 3] = g4.w;
            dob[4 * q] = o4.x; dob[4 * q + 1] = o4.y; dob[4 * q + 2] = o4.z; dob[4 * q + 3] = o4.w;
        }
        if (HAS_DO) {
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wpT[s], dob[s], acc, 0, 0, 0);
        }
        const bool live = valid && t >= Z;
        float da[16], dg[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float dz = live ? acc[r] : 0.f;
            da[r] = dz * gg[r] * (1.f - ff[r] * ff[r]);
            dg[r] = dz * ff[r] * gg[r] * (1.f - gg[r]);
        }
        // scratch for pass 2 and the transposed LDS patches for the weight gradients
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 a4 = make_float4(da[4 * q], da[4 * q + 1], da[4 * q + 2], da[4 * q + 3]);
            float4 g4 = make_float4(dg[4 * q], dg[4 * q + 1], dg[4 * q + 2], dg[4 * q + 3]);
            if (valid) {
                float* p = dab + ((long long)b * T + t) * 64 + 8 * q + 4 * h;
                *reinterpret_cast<float4*>(p) = a4;
                *reinterpret_cast<float4*>(p + 32) = g4;
            }
            *reinterpret_cast<float4*>(lda + j * kPad + 8 * q + 4 * h) = a4;
            *reinterpret_cast<float4*>(ldg + j * kPad + 8 * q + 4 * h) = g4;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // contraction over the tile's 32 time columns: step s covers columns 2s and 2s+1.  ALL operand
        // loads are issued first (straight-line code, no branch in between), then the MFMAs run, so
        // the L2 round trips overlap instead of queueing one behind the other.
        float bxc[16], bxo[16], ado[16], bfz[16], bgz[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {                                        // all 80 (32 without dout) loads in flight
            const int tt = t0 + 2 * u + h;
            const int ttc = tt < T ? tt : T - 1;                              // clamped rows, masked values
            const long long r0 = ((long long)b * T + ttc) * 32 + j;
            const long long r1 = ((long long)b * T + (ttc - d >= 0 ? ttc - d : 0)) * 32 + j;
            bxc[u] = x[r0];
            bxo[u] = x[r1];
            if (HAS_DO) { ado[u] = dout[r0]; bfz[u] = f[r0]; bgz[u] = g[r0]; }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int tt = t0 + 2 * s + h;
            const float mv = tt < T ? 1.f : 0.f, mo = (tt < T && tt - d >= 0) ? 1.f : 0.f;
            const float a_da = lda[(2 * s + h) * kPad + j];
            const float a_dg = ldg[(2 * s + h) * kPad + j];
            const float b_xc = bxc[s] * mv, b_xo = bxo[s] * mo;
            aWf1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_da, b_xc, aWf1, 0, 0, 0);
            aWf0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_da, b_xo, aWf0, 0, 0, 0);
            aWg1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_dg, b_xc, aWg1, 0, 0, 0);
            aWg0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_dg, b_xo, aWg0, 0, 0, 0);
            if (HAS_DO) aWp = __builtin_amdgcn_mfma_f32_32x32x2f32(ado[s] * mv, bfz[s] * bgz[s], aWp, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- sum the five accumulators over the workgroup's waves, then one set of atomics ---------
    for (int w = 1; w < kWaves; ++w) {
        __syncthreads();
        if (wv == w) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                red[(0 * 16 + r) * 64 + lane] = aWf0[r]; red[(1 * 16 + r) * 64 + lane] = aWf1[r];
                red[(2 * 16 + r) * 64 + lane] = aWg0[r]; red[(3 * 16 + r) * 64 + lane] = aWg1[r];
                red[(4 * 16 + r) * 64 + lane] = aWp[r];
            }
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                aWf0[r] += red[(0 * 16 + r) * 64 + lane]; aWf1[r] += red[(1 * 16 + r) * 64 + lane];
                aWg0[r] += red[(2 * 16 + r) * 64 + lane]; aWg1[r] += red[(3 * 16 + r) * 64 + lane];
                aWp[r] += red[(4 * 16 + r) * 64 + lane];
            }
        }
    }
    if (wv == 0) {
        // this workgroup's five partial tiles leave with plain coalesced stores: part[wg][tile][r][lane].
        // (512 workgroups adding atomically into the same 20 KB cost ~35 us per layer; k_layer_bwd_reduce
        // sums the partials instead.)
        float* __restrict__ o = part + (long long)blockIdx.x * kPartFloats + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o[(0 * 16 + r) * 64] = aWf0[r]; o[(1 * 16 + r) * 64] = aWf1[r];
            o[(2 * 16 + r) * 64] = aWg0[r]; o[(3 * 16 + r) * 64] = aWg1[r];
            o[(4 * 16 + r) * 64] = aWp[r];
        }
    }
}

// =============================================================================================
// Chained backward (no pass 2, no (da,dg) scratch).  The gradient of the residual stream is kept in the
// split form    dout_l[t] = V[t] + U[t + dU]    where the layer above (dilation dU) wrote
//     V = dout + [Wf1;Wg1]^T dab      (tap 1 reads x[t])        U = [Wf0;Wg0]^T dab    (tap 0 reads x[t-d])
// so the kernel reads its dout on the fly from two tensors, does everything k_layer_bwd_p1 does, and
// additionally runs the two transposed-weight GEMMs on the (da,dg) it still holds in registers (their
// accumulator layout IS the MFMA B-operand layout) to emit its own V and U.  Per layer this removes the pass-2
// kernel, the 256 B/column scratch write and its two re-reads.
//
// Structures tried on the way, all measured on config 2 (8 x 16,384 columns; DESIGN.md section 5 has the numbers):
//   1. 4 waves x 2 workgroups per CU, operands loaded straight from memory in MFMA layout, weight-gradient operands
//      re-read channel-on-lanes: 64 us per layer (96 scalar loads per tile queue behind the V/U stores; 165 MB fetched
//      for 117 MB of distinct data);
//   2. one wave per SIMD, every tensor through LDS-DMA exactly once, whole-row stores: 49 us;
//   3. producer / consumer wave pairs sharing a SIMD: 47 us -- a wave that streams MFMAs starves its sibling's VALU
//      issue (11,000 cycles for 300 VALU instructions next to the consumer's MFMA stream);
//   4. (this one) one wave per SIMD, software-pipelined: 45 us.
// =============================================================================================
static constexpr int kCWaves = 4;
#ifdef WN_BWD_STAMPS
__device__ unsigned long long g_bwd_stamps[1024 * 8];
#define BST(v) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); } while (0)
#endif
static constexpr int kCWaveFloats = 8192;                       // two groups of four 4 KB slots per wave
static constexpr int kCWFloats = 2048 + 2048 + 1024;            // Wf, Wg, Wp
static constexpr int kCLdsBytes = (kCWFloats + kCWaves * kCWaveFloats + kCWaves * 64) * 4;   // + 256 B per wave: dataflow words (MULTI)
static constexpr int kCMaxBlocks = 256;

#define WN_LDS_DMA16(src, dst) \
    __builtin_amdgcn_global_load_lds((src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
#define WN_LDS_DMA16_SC1(src, dst) \
    __builtin_amdgcn_global_load_lds((src), (__attribute__((address_space(3))) void*)(dst), 16, 0, WN_SC_AUX)

// ---- fp16 x 2 split products for the weight-gradient contractions (H2W) ------------------------------------------------
// The 80 fp32 MFMAs per tile that contract over time (dWf, dWg: [da; dg] x [x[t-d]; x[t]], dWp: dout x z) are half of the
// kernel's matrix time (5,120 of 10,240 cycles per tile).  Their operands already sit in registers with 16 time steps
// per lane (rows 2 s + h), which is exactly an f16 MFMA operand pair (k-step ks, element e <-> s = 8 ks + e): each fp32
// value v s is split into two fp16 parts h + m (22 bits), a product is the three terms m h' + h m' + h h' of
// v_mfma_f32_32x32x16_f16 (32 cycles for K = 16 against 64 for K = 2): 30 MFMAs = 960 cycles per tile.  The scale s is a
// power of two chosen PER TILE from the wave's own maximum (gradients: max over da, dg, dout; inputs: max over x[t],
// x[t-d]; z = tanh sigmoid: 2^14), so that nothing overflows fp16 and the parts lost to fp16's subnormals are below
// 2^-24 of the tile's maximum; a tile's product leaves the matrix core in its own scale and joins the running fp32
// accumulator with one fma per element.  Selected by WnExec.precision == WN_GEMM_FP16X2; any other precision keeps
// the exact-fp32 MFMAs.
__device__ __forceinline__ void lb_h2_product(const H2Op& a, const H2Op& b, float u, f32x16& acc) {
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                           // smallest terms first
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.m[ks], b.h[ks], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h[ks], b.m[ks], t, 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h[ks], b.h[ks], t, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = fmaf(t[r], u, acc[r]);
}

// ---------------------------------------------------------------------------------------------
// k_layer_bwd_chainsp: one workgroup of 4 waves per CU (one wave per SIMD, 512 registers, 148 KB of LDS), software-
// pipelined: the loop body holds the 80 weight-gradient MFMAs of tile n (operands already in registers) AND the whole
// first half of tile n+1 (LDS reads, dz chain, gate, V/U MFMAs, patch writes) in one basic block, so the compiler's
// scheduler interleaves the VALU / LDS work of one tile between the MFMAs of the other.
// Data movement -- every tensor crosses the memory pipeline exactly once and in whole 128-byte rows:
//   * f, g, V, U (time on lanes): LDS-DMA (global_load_lds: lane L of piece k fetches 16 bytes of row 8k + L/8, 1 KB
//     contiguous per instruction, no registers), one tile ahead, into the wave's other slot group.  The tiles are
//     stored XOR-swizzled (16-byte chunk c of row r at position c ^ (r & 7), done by permuting the SOURCE addresses),
//     so that the float4 read of lane (j,h) -- row j, chunks 2q+h -- is bank-conflict free;
//   * x[t], x[t-d] (channel on lanes, weight gradients only): plain loads, a half wave = one 128-byte row, one tile
//     ahead in registers;  dz: float4 per lane (time on lanes, nothing to transpose);
//   * V, U results: written to the (consumed) V/U slots, read back row-wise, stored as whole rows;
//   * da, dg, dout, z = f g: left as swizzled patches in the slots the tile's inputs came in, taken into registers
//     (channel on lanes) at the top of the next body;
//   * weights: raw Wf/Wg/Wp copied to LDS with coalesced float4 loads; Wf[cd][cr][0..1] is one ds_read_b64 that
//     yields the A operands of both the V (tap 1) and the U (tap 0) MFMA.
// Vector-memory operations retire in issue order, so "it has landed" is an s_waitcnt vmcnt(N) with N = the number of
// operations issued after it (the eight V/U stores of a complete tile).
// Tile order: workgroups are dealt to the 8 XCDs round-robin, so XCD k = blockIdx % 8 gets the k-th contiguous eighth
// of the tiles (x[t-d] and U[t+dU] of a tile are rows a neighbouring CU of the same XCD fetches in the same round).
// ---------------------------------------------------------------------------------------------
// ---- several layers in ONE launch (MULTI) ------------------------------------------------------------------------------
// A launch gives a wave four tiles: the weight images, the first tile's fetch burst (every workgroup of the grid at once),
// the half-empty pipeline of the first and last tile and the partial-tile reduction are a third of it.  The multi-layer
// form keeps the tile code and walks the layers of a stack (entries 0, 1, ... = stack layers layer[0] > layer[1] > ...)
// inside one launch of co-resident workgroups (one per CU: the LDS footprint guarantees it).
//
// Synchronisation is DATAFLOW, per tile, not a grid barrier.  (A grid barrier was built first and measured: counter in
// device memory, one arrival per workgroup and layer.  With agent-scope fences around it -- buffer_wbl2 / buffer_inv sc1,
// 32 per XCD and layer -- the layer backward went 1.35 -> 2.03 ms; without cache maintenance, (V, U) travelling with the
// sc1 bit instead, 1.35 -> 1.51 ms = +4 us per layer, while the same launch with every barrier open ran 0.17 ms FASTER
// than the per-layer launches: a barrier costs ~9 us per layer, mostly waiting for the slowest of 256 workgroups before
// any workgroup may request the next layer's (V, U).)  Every tile of the gradient stream has a word in device memory
// holding the number of entries it has completed.  Entry e's tile X (rows 32 X ..) may
//   * READ  V[X] and U[X + dU/32 (+1)] of entry e - 1 once those tiles have completed e - 1, i.e. show >= e, and
//   * WRITE its own V, U -- (V, U) rotate through THREE buffer pairs: entry e writes the pair entry e - 2 read, so the
//     write-after-read distance is two entries -- once the READERS of those rows, tiles X and X - dU'/32 (-1) of the entry
//     that read them, show that entry done,
// so one test, "five words >= e", covers both; tiles the previous entry did not process (below its live range, outside the
// clip) impose nothing.  A wave publishes a tile (one sc1 store of e + 1) when the tile's stores have been counted by
// vmcnt -- at the top of the next loop body, whose wait already guarantees it -- and requests the words of the tile it
// will fetch NEXT one body ahead, so that the test costs no memory round trip.  Waits only ever point at earlier
// entries, so there is no cycle; in steady state nothing waits at all (the words a tile needs were written two bodies
// ago by waves that walk the same tile sequence).  (V, U) are stored and loaded with the sc1 bit (written through to
// memory / fetched from memory: st16_sc1, lds_dma16_sc1, aux = 16 of the builtin): the only data that crosses XCDs
// inside the launch.  Everything of the NEXT layer that the forward pass wrote -- weights, z, sigmoid, dz_skip, x of the
// first tile -- is requested at the layer boundary before the finished layer's partial tile is reduced.  The partial
// weight-gradient tiles leave per layer exactly as in the per-layer form (same sums, same order: results are
// bit-identical when both forms use the same grid).
static constexpr int kChainMaxL = 48;
struct ChainArgs {
    const float* Wf[kChainMaxL]; const float* Wg[kChainMaxL]; const float* Wp[kChainMaxL];     // per entry of the launch
    int d[kChainMaxL], Z[kChainMaxL], tile_lo[kChainMaxL], vu_t0[kChainMaxL], dU[kChainMaxL], rot[kChainMaxL];
    int layer[kChainMaxL];             // stack index of entry i (descending): selects x, z, sigmoid, dz, the V/U parity, the partial tiles
    const float* x0;                   // input of stack layer 0
    const float* xs;                   // (L, B, T, 32): layer l's input is xs[l - 1]
    const float* z; const float* g;    // (L, B, T, 32)
    const float* dz;                   // (L, B, T, 32); rows below dz_t0 are not read
    float* V[3]; float* U[3];          // three pairs in rotation: layer l reads [(l + 1) % 3], writes [l % 3]
    float* part; long long part_stride;
    unsigned* sync;                    // [0]: set if a wait gave up (results void), [1]: always "done", [2 + b * tiles_all + X]: entries
                                       // tile X of clip b has completed; all zeroed by k_chain_zero_sync before the launch
    int n, B, T, dz_t0;
};
static constexpr int kChainSyncHead = 2;
#ifdef WN_MULTI_STAMPS
__device__ unsigned long long g_multi_stamps[kChainMaxL + 1][256][4];     // [entry][workgroup][wave]: s_memtime when the wave finished the entry's tiles
__device__ unsigned long long g_multi_spins[kChainMaxL + 1][256][4];      // cycles spent inside dep_wait (<< 20) | number of reloads
__device__ unsigned long long g_multi_seg[256][4][8];                     // cycles per segment of the layer boundary, summed over entries
#define MST(v) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); } while (0)
#endif

// FROM_Z: the forward saved z and sigmoid only; `f` points at z and tanh is recovered as z / sigmoid (z = tanh * sigmoid was
// rounded once in fp32, so the quotient is tanh to ~1.2e-7 relative; where sigmoid underflowed, da and dg are 0 anyway).
template <bool HAS_DO, bool HAS_U, bool HAS_DZ, bool FROM_Z, bool H2W, bool MULTI>
__device__ __forceinline__ void chain_body(
    const float* __restrict__ x, const float* __restrict__ f, const float* __restrict__ g,
    const float* __restrict__ Wp, const float* __restrict__ Wf, const float* __restrict__ Wg,
    const float* __restrict__ Vin, const float* __restrict__ Uin, int dU, int vu_t0,
    const float* __restrict__ dzs, int dz_t0, float* __restrict__ Vout, float* __restrict__ Uout,
    float* __restrict__ part, int B, int T, int d, int Z, int tile_lo, int tiles_per_b, int ntiles,
    const ChainArgs* __restrict__ ma) {
    // tiles_per_b counts the LIVE tiles of a clip: tile k of the grid is tile tile_lo + k % tiles_per_b of clip
    // k / tiles_per_b.  Columns below 32 * tile_lo cannot receive gradient (they are further from the loss window than
    // the layers above reach): they are not computed, and Vin / Uin rows below vu_t0 (which the layer above did not
    // write for the same reason) are taken as 0 without being trusted.
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float* lWf = dyn;
    float* lWg = dyn + 2048;
    float* lWp = dyn + 4096;
    float* wbase = dyn + kCWFloats;
#ifdef WN_BWD_STAMPS
    unsigned long long st_k0 = 0, st_k1 = 0, st_k2 = 0, st_k3 = 0, st_k4 = 0, st_loop_end = 0;
    BST(st_k0);
#endif
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    constexpr int NW = kCWaves;
    constexpr bool PL = H2W;
    // two slot groups per wave (software pipeline)
    float* pbase = wbase + wv * kCWaveFloats;   // group k: pbase + 4096 k = {f | dout, g | z, V | da, U | dg}
    // Tile order.  A wave walks ids first, first + stride, ... < last; id -> (clip b, tile X of the clip) is
    //     b = cb0 + id % nb,   X = tiles_all - 1 - id / nb
    // i.e. clips interleaved, time running BACKWARDS from the end of the clip.  The live tiles of a layer (X >= tile_lo) are
    // then the ids below nb * tiles_per_b whatever tile_lo is: a tile belongs to the same wave in every layer (MULTI: what a
    // tile waits for was finished by waves walking the same sequence, two or more loop bodies earlier), and the tiles a
    // tile's U rows come from (larger X) precede it.  With the workgroups dealt to the 8 XCDs round-robin (blockIdx % 8)
    // and B a multiple of 8, XCD k owns clips k B/8 .. (nb = B/8): x[t - d] and U[t + dU] of a tile are rows a neighbouring
    // CU of the same XCD fetches in the same round.  Otherwise nb = B and the ids are dealt to all waves in turn.
    const int tiles_all = (T + 31) / 32;
    int first, stride, last;
    int nb, cb0;
    if ((gridDim.x & 7) == 0 && (B & 7) == 0) {
        nb = B >> 3;
        cb0 = (blockIdx.x & 7) * nb;
        stride = (gridDim.x >> 3) * NW;
        first = (blockIdx.x >> 3) * NW + wv;
    } else {
        nb = B;
        cb0 = 0;
        stride = gridDim.x * NW;
        first = blockIdx.x * NW + wv;
    }
    auto decode = [&](int id, int& b, int& t0) {
        const int r = id / nb;
        b = cb0 + (id - r * nb);
        t0 = (tiles_all - 1 - r) * 32;
    };
    // MULTI: the ids are dealt to the waves ROTATED by rot[entry].  A layer's live tiles are rarely a whole number per wave
    // (config 2: 385 .. 512 live tiles per clip on 128 waves = 3 or 4 each): with a fixed deal the same waves would carry the
    // extra tile in every layer and everybody else would wait for them at the next wide dependency; the host advances
    // rot by each entry's surplus, so the extra tiles walk around the waves and every wave does the AVERAGE number of
    // tiles over the launch -- which the dataflow synchronisation (no barrier) lets it turn into time.
    const int first0 = first;
    int li = 0;                        // MULTI: the current entry of the launch's table
    auto partition = [&]() {
        last = nb * tiles_per_b;
        if constexpr (MULTI) {
            first = first0 - ma->rot[li];
            if (first < 0) first += stride;
        }
    };
    // MULTI: entry i of the launch's table becomes the current layer (uniform: scalar loads from the kernel arguments)
    int dUp = 0, lo_p = 0;             // dU and first live tile of the entry TWO back: its tiles read the rows of the (V, U) pair
                                       // the current entry rewrites (three pairs in rotation)
    auto set_layer = [&](int i) {
        const int l = ma->layer[i];
        dUp = i > 1 ? ma->dU[i - 2] : 0;
        lo_p = i > 1 ? ma->tile_lo[i - 2] : 0;
        const long long lo = (long long)l * B * T * 32;
        x = l == 0 ? ma->x0 : ma->xs + (lo - (long long)B * T * 32);
        f = ma->z + lo;
        g = ma->g + lo;
        dzs = ma->dz + lo;
        Wp = ma->Wp[i]; Wf = ma->Wf[i]; Wg = ma->Wg[i];
        Vin = ma->V[(l + 1) % 3]; Uin = ma->U[(l + 1) % 3];
        Vout = ma->V[l % 3]; Uout = ma->U[l % 3];
        part = ma->part + (long long)l * ma->part_stride;
        dU = ma->dU[i]; vu_t0 = ma->vu_t0[i]; d = ma->d[i]; Z = ma->Z[i];
        tile_lo = ma->tile_lo[i];
        tiles_per_b = (T + 31) / 32 - tile_lo;
        ntiles = B * tiles_per_b;
    };
    if constexpr (MULTI) set_layer(0);
    partition();
    const int lr = lane >> 3, lp = lane & 7;
    // ---- MULTI: the per-tile dataflow words (see "several layers in ONE launch" above)
    bool gave_up = false;
    // request the five words tile `tile` of the current entry depends on (lanes 0..4; every other lane, and every word that
    // imposes nothing, reads as "done"); the value is tested later by dep_wait
    auto dep_ptr = [&](int tile) -> const unsigned* {
        const unsigned* ptr = ma->sync + 1;                           // the word that always reads "done"
        if (li > 0) {
            int b, t0;
            decode(tile, b, t0);
            const int X = t0 >> 5;                                 // tile index inside the clip
            const int k = dU >> 5, kp = dUp >> 5;
            int q = -1, lo = vu_t0 >> 5;                           // lo: first tile the previous entry processed
            if (lane == 0) q = X;                                  // V[X]
            else if (lane == 1) q = X + k;                         // U rows 32 X + dU ..
            else if (lane == 2) q = (dU & 31) ? X + k + 1 : -1;
            else if (lane == 3 && li > 1) { q = X - kp; lo = lo_p; }                          // who read U rows 32 X .. of the pair this
            else if (lane == 4 && li > 1) { q = (dUp & 31) ? X - kp - 1 : -1; lo = lo_p; }    // tile rewrites, two entries back
            if (q >= lo && q < tiles_all) ptr = ma->sync + kChainSyncHead + b * tiles_all + q;
        }
        return ptr;
    };
    // lanes 0..2 must show the previous entry done (>= li), lanes 3, 4 the entry before it (>= li - 1)
    const unsigned dep_slack = (lane == 3 || lane == 4) ? 1u : 0u;
    auto dep_request = [&](int tile) -> unsigned {
        unsigned v = 0xffffffffu;
        if constexpr (MULTI) {
            if (li > 0) v = __hip_atomic_load(dep_ptr(tile), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return v;
    };
    // the same request through LDS (lds_dma4_sc1): for the tile loop, where a register destination would make hipcc wait for
    // every vector-memory operation issued since -- the eight (V, U) stores the loop deliberately leaves in flight included.
    // The loop's own s_waitcnt vmcnt(8) at the top of the next body covers the request; dep_take reads the words then.
    unsigned* const dep_lds = reinterpret_cast<unsigned*>(wbase + kCWaves * kCWaveFloats) + wv * 64;
    auto dep_request_lds = [&](int tile) {
        if constexpr (MULTI) lds_dma4_sc1(dep_ptr(tile), dep_lds);
    };
    // (inline asm with its own wait: a plain LDS read here "may alias" the loop's LDS-DMA requests in hipcc's eyes and gets
    // s_waitcnt vmcnt(0) in front of it -- the x rows the body has just asked for and the (V, U) stores included)
    auto dep_take = [&]() -> unsigned {
        unsigned v = 0xffffffffu;
        if constexpr (MULTI)
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(v) : "v"((unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)dep_lds + 4u * (unsigned)lane) : "memory");
        return v;
    };
#ifdef WN_MULTI_STAMPS
    unsigned long long dbg_spin = 0;
#endif
    auto dep_wait = [&](int tile, unsigned v) {
        if constexpr (MULTI) {
            unsigned spins = 0;
#ifdef WN_MULTI_STAMPS
            unsigned long long w0; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w0) :: "memory");
#endif
            // a word still shows an earlier entry?  Every polled value is USED before the loop can be left, and the loop's
            // condition is a scalar: with `while (ballot(v < ..))` the header's compare may be fed by the poll of the back edge,
            // so hipcc puts s_waitcnt vmcnt(0) in front of it ON EVERY PASS -- also the first, whose v came from LDS -- and
            // the x rows fetch_x has just requested (and the eight (V, U) stores) are waited for with nothing to overlap them
            bool late = !gave_up && __builtin_amdgcn_ballot_w64(v < (unsigned)li - dep_slack) != 0ull;
            while (late) {
                __builtin_amdgcn_s_sleep(1);
                v = dep_request(tile);
                late = __builtin_amdgcn_ballot_w64(v < (unsigned)li - dep_slack) != 0ull;
                if (++spins > (1u << 18)) {          // never hang the GPU: give up for good, flag it, the results are void
                    gave_up = true;
                    if (lane == 0) __hip_atomic_store(ma->sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    late = false;
                }
            }
#ifdef WN_MULTI_STAMPS
            if (spins) { unsigned long long w1; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w1) :: "memory");
                         dbg_spin += ((w1 - w0) << 20) | spins; }
#endif
        }
    };
    auto publish = [&](int tile) {
        if constexpr (MULTI) {
            int b, t0;
            decode(tile, b, t0);
            const int X = t0 >> 5;
            if (lane == 0)
                __hip_atomic_store(ma->sync + kChainSyncHead + b * tiles_all + X, (unsigned)(li + 1), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    unsigned dep_1 = 0xffffffffu;                             // requested words of the entry's second tile
    int pub = 0;                                              // next tile of this wave to publish

    // which: 1 = z (or tanh), sigmoid and dz_skip; 2 = V, U; 3 = everything.  ASM: the LDS-DMA as inline asm (h2_ops.hpp): hipcc
    // then puts no vmcnt(0) in front of later LDS reads, so the requests stay in flight across the layer boundary's LDS
    // work -- the caller waits for them itself
    auto fetch_some = [&](int tile, float* grp, float4 (&dz4)[4], const int which, const bool use_asm) {
        int b, t0;
        decode(tile, b, t0);
        auto dma = [&](const float* src, float* dst) {
            if (use_asm) lds_dma16(src, dst);
            else WN_LDS_DMA16(src, dst);
        };
        auto dma_vu = [&](const float* src, float* dst) {       // MULTI: (V, U) were written by other XCDs in this launch
            if constexpr (MULTI) {
                if (use_asm) lds_dma16_sc1(src, dst);
                else WN_LDS_DMA16_SC1(src, dst);
            } else {
                dma(src, dst);
            }
        };
        if (t0 + 32 + (HAS_U ? dU : 0) <= T) {
            // interior tile (wave-uniform test): one base address per tensor, the four pieces are 1 KB apart
            const long long o = ((long long)b * T + t0 + lr) * 32 + ((lp ^ lr) << 2);
            const float* pf = f + o;
            const float* pg = g + o;
            const float* pv = Vin + o;
            const float* pu = Uin + o + (long long)dU * 32;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (which & 1) {
                    dma(pf + k * 256, grp + k * 256);
                    dma(pg + k * 256, grp + 1024 + k * 256);
                }
                if ((which & 2) && HAS_DO) dma_vu(pv + k * 256, grp + 2048 + k * 256);
                if ((which & 2) && HAS_U) dma_vu(pu + k * 256, grp + 3072 + k * 256);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 8 * k + lr;
                const int tt = t0 + r;
                const int ttc = tt < T ? tt : T - 1;
                const int ttu = ttc + dU < T ? ttc + dU : T - 1;
                const long long o = ((long long)b * T + ttc) * 32 + ((lp ^ (r & 7)) << 2);
                if (which & 1) {
                    dma(f + o, grp + k * 256);
                    dma(g + o, grp + 1024 + k * 256);
                }
                if ((which & 2) && HAS_DO) dma_vu(Vin + o, grp + 2048 + k * 256);
                if ((which & 2) && HAS_U) dma_vu(Uin + ((long long)b * T + ttu) * 32 + ((lp ^ (r & 7)) << 2), grp + 3072 + k * 256);
            }
        }
        if (HAS_DZ && (which & 1)) {
            // dz_skip exists for columns t >= dz_t0 only (the loss window): tiles below it load nothing, the tile
            // that straddles dz_t0 selects (the memory below dz_t0 is uninitialised)
            if (t0 + 32 > dz_t0) {
                const int t = t0 + j;
                const long long rowc = ((long long)b * T + (t < T ? t : T - 1)) * 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) dz4[q] = *reinterpret_cast<const float4*>(dzs + rowc + 8 * q);
                if (t0 < dz_t0) {
                    const bool in = t >= dz_t0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        dz4[q].x = in ? dz4[q].x : 0.f; dz4[q].y = in ? dz4[q].y : 0.f;
                        dz4[q].z = in ? dz4[q].z : 0.f; dz4[q].w = in ? dz4[q].w : 0.f;
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) dz4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto fetch_a = [&](int tile, float* grp, float4 (&dz4)[4]) { fetch_some(tile, grp, dz4, 3, false); };
    auto fetch_x = [&](int tile, float (&xc)[16], float (&xo)[16]) {
        int b, t0;
        decode(tile, b, t0);
        if (t0 - d >= 0 && t0 + 32 <= T) {               // interior tile: rows 2s+h are 256 B apart from one base
            if constexpr (PL) {                          // rows 16 (s >> 3) + 8 h + (s & 7): the k order of the tr reads
                const float* pc = x + ((long long)b * T + t0 + 8 * h) * 32 + j;
                const float* po = pc - (long long)d * 32;
#pragma unroll
                for (int s = 0; s < 16; ++s) { xc[s] = pc[(16 * (s >> 3) + (s & 7)) * 32]; xo[s] = po[(16 * (s >> 3) + (s & 7)) * 32]; }
            } else {
                const float* pc = x + ((long long)b * T + t0 + h) * 32 + j;
                const float* po = pc - (long long)d * 32;
#pragma unroll
                for (int s = 0; s < 16; ++s) { xc[s] = pc[s * 64]; xo[s] = po[s * 64]; }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int tt = t0 + (PL ? 16 * (s >> 3) + 8 * h + (s & 7) : 2 * s + h);
                const int ttc = tt < T ? tt : T - 1;
                const int tto = ttc - d >= 0 ? ttc - d : 0;
                xc[s] = x[((long long)b * T + ttc) * 32 + j];
                xo[s] = x[((long long)b * T + tto) * 32 + j];
            }
        }
    };

    f32x16 aWf0, aWf1, aWg0, aWg1, aWp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { aWf0[r] = 0.f; aWf1[r] = 0.f; aWg0[r] = 0.f; aWg1[r] = 0.f; aWp[r] = 0.f; }

    // first half of a tile: everything up to the patches in its slot group; the V/U rows come back in registers
    // (vv, uu) and are stored by the caller.  `grp` holds the tile's f, g, V, U (landed).
    float w_inv = 1.f;                                 // H2W: 1 / (the weight images' power-of-two scale)
    // PL (H2W, four waves): a tile's dout, z, da, dg leave phase_a already split, as fp16 planes (h at +0, m at +2 KB of the
    // 4 KB slot, 64-byte rows, the 8-byte piece c of row r at position c ^ ((r >> 2) & 7)), and the weight-gradient
    // operands are taken from them TRANSPOSED by ds_read_b64_tr_b16 -- no second split, no transposing scalar reads.
    float pa_ig = 0.f, pa_id = 0.f;                    // 1 / scale of (da, dg) and of dout of the tile phase_a saw last
    float wg_ig = 0.f, wg_id = 0.f;                    // ... of the tile whose weight gradients come next
    float vv[16], uu[16];
    auto phase_a = [&](int tile, float* grp, const float4 (&dz4)[4]) {
        float* tf = grp;
        float* tg = grp + 1024;
        float* tv = grp + 2048;
        float* tu = grp + 3072;
        int b, t0;
        decode(tile, b, t0);
        const int t = t0 + j;
        const bool valid = t < T;
        const bool vin = t >= vu_t0;                                  // select, never multiply: unwritten rows may hold anything
        const bool uin = HAS_U && valid && t + dU < T && t + dU >= vu_t0;
        f32x16 acc;
        float ff[16], gg[16], dob[16];
        H2Op od, oa, og;                                 // H2W: dout, da, dg split (kept for the planes)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = j * 32 + (((2 * q + h) ^ (j & 7)) << 2);
            float4 z4 = make_float4(0, 0, 0, 0), o4 = z4;
            const float4 f4 = *reinterpret_cast<const float4*>(tf + o);
            const float4 g4 = *reinterpret_cast<const float4*>(tg + o);
            if (HAS_DZ) z4 = dz4[q];
            if (HAS_DO) {
                const float4 v4 = *reinterpret_cast<const float4*>(tv + o);
                o4.x = vin ? v4.x : 0.f; o4.y = vin ? v4.y : 0.f; o4.z = vin ? v4.z : 0.f; o4.w = vin ? v4.w : 0.f;
            }
            if (HAS_U) {
                const float4 u4 = *reinterpret_cast<const float4*>(tu + o);
                o4.x += uin ? u4.x : 0.f; o4.y += uin ? u4.y : 0.f; o4.z += uin ? u4.z : 0.f; o4.w += uin ? u4.w : 0.f;
            }
            acc[4 * q] = z4.x; acc[4 * q + 1] = z4.y; acc[4 * q + 2] = z4.z; acc[4 * q + 3] = z4.w;
            ff[4 * q] = f4.x; ff[4 * q + 1] = f4.y; ff[4 * q + 2] = f4.z; ff[4 * q + 3] = f4.w;
            gg[4 * q] = g4.x; gg[4 * q + 1] = g4.y; gg[4 * q + 2] = g4.z; gg[4 * q + 3] = g4.w;
            dob[4 * q] = o4.x; dob[4 * q + 1] = o4.y; dob[4 * q + 2] = o4.z; dob[4 * q + 3] = o4.w;
        }
        if (HAS_DO || HAS_U) {
            if constexpr (H2W) {
                // dz += Wp^T dout: the lane's 16 channels of dout are an f16 operand pair as they stand (k-step ks, element
                // e <-> s = 8 ks + e); Wp's image in LDS is pre-split in the same channel order
                float md = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) md = fmaxf(md, fabsf(dob[s]));
                md = lb_wave_max(md);
                float sd, id;
                lb_pow2_scale(md, sd, id);
                pa_id = id;
                lb_split16(dob, sd, od);
                f32x16 t;
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = 0.f;
                const char* ip = reinterpret_cast<const char*>(dyn) + 16384 + lane * 16;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const h16x8 ah = *reinterpret_cast<const h16x8*>(ip + (ks * 2 + 0) * 1024);
                    const h16x8 am = *reinterpret_cast<const h16x8*>(ip + (ks * 2 + 1) * 1024);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, od.h[ks], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, od.m[ks], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, od.h[ks], t, 0, 0, 0);
                }
                const float u = id * w_inv;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaf(t[r], u, acc[r]);
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lWp[bch(s, h) * 32 + j], dob[s], acc, 0, 0, 0);
            }
        }
        const bool live = valid && t >= Z;
        float da[16], dg[16], zz[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float dz = live ? acc[r] : 0.f;
            if (FROM_Z) {
                zz[r] = ff[r];                                            // the slot held z
                ff[r] = gg[r] > 1e-30f ? zz[r] * __builtin_amdgcn_rcpf(gg[r]) : 0.f;
            } else {
                zz[r] = ff[r] * gg[r];
            }
            da[r] = dz * gg[r] * (1.f - ff[r] * ff[r]);
            dg[r] = dz * ff[r] * gg[r] * (1.f - gg[r]);
        }
        f32x16 v1, u0;
        if constexpr (H2W) {
            // V = dout + [Wf1; Wg1]^T [da; dg], U = [Wf0; Wg0]^T [da; dg]: 24 f16 MFMAs for 64 fp32 ones
            float mg = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) mg = fmaxf(mg, fmaxf(fabsf(da[s]), fabsf(dg[s])));
            mg = lb_wave_max(mg);
            float sg, ig;
            lb_pow2_scale(mg, sg, ig);
            pa_ig = ig;
            lb_split16(da, sg, oa);
            lb_split16(dg, sg, og);
            f32x16 tv, tu;
#pragma unroll
            for (int r = 0; r < 16; ++r) { tv[r] = 0.f; tu[r] = 0.f; }
            const char* ib = reinterpret_cast<const char*>(dyn) + lane * 16;      // image (mat, tap, ks, part) at 1 KB steps
#pragma unroll
            for (int mat = 0; mat < 2; ++mat) {
                const H2Op& ob = mat == 0 ? oa : og;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const h16x8 a0h = *reinterpret_cast<const h16x8*>(ib + ((((mat * 2 + 0) * 2 + ks) * 2) + 0) * 1024);
                    const h16x8 a0m = *reinterpret_cast<const h16x8*>(ib + ((((mat * 2 + 0) * 2 + ks) * 2) + 1) * 1024);
                    const h16x8 a1h = *reinterpret_cast<const h16x8*>(ib + ((((mat * 2 + 1) * 2 + ks) * 2) + 0) * 1024);
                    const h16x8 a1m = *reinterpret_cast<const h16x8*>(ib + ((((mat * 2 + 1) * 2 + ks) * 2) + 1) * 1024);
                    tv = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1m, ob.h[ks], tv, 0, 0, 0);
                    tu = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0m, ob.h[ks], tu, 0, 0, 0);
                    tv = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, ob.m[ks], tv, 0, 0, 0);
                    tu = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, ob.m[ks], tu, 0, 0, 0);
                    tv = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, ob.h[ks], tv, 0, 0, 0);
                    tu = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, ob.h[ks], tu, 0, 0, 0);
                }
            }
            const float u = ig * w_inv;
#pragma unroll
            for (int r = 0; r < 16; ++r) { v1[r] = fmaf(tv[r], u, dob[r]); u0[r] = tu[r] * u; }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { v1[r] = dob[r]; u0[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float2 wf = *reinterpret_cast<const float2*>(lWf + (bch(s, h) * 32 + j) * 2);
                const float2 wg = *reinterpret_cast<const float2*>(lWg + (bch(s, h) * 32 + j) * 2);
                v1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.y, da[s], v1, 0, 0, 0);
                u0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.x, da[s], u0, 0, 0, 0);
                v1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wg.y, dg[s], v1, 0, 0, 0);
                u0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wg.x, dg[s], u0, 0, 0, 0);
            }
        }
        // V, U through the (consumed) V/U slots into row order
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = j * 32 + (((2 * q + h) ^ (j & 7)) << 2);
            *reinterpret_cast<float4*>(tv + o) = make_float4(v1[4 * q], v1[4 * q + 1], v1[4 * q + 2], v1[4 * q + 3]);
            *reinterpret_cast<float4*>(tu + o) = make_float4(u0[4 * q], u0[4 * q + 1], u0[4 * q + 2], u0[4 * q + 3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(tv + k * 256 + lane * 4);
            const float4 c = *reinterpret_cast<const float4*>(tu + k * 256 + lane * 4);
            vv[4 * k] = a.x; vv[4 * k + 1] = a.y; vv[4 * k + 2] = a.z; vv[4 * k + 3] = a.w;
            uu[4 * k] = c.x; uu[4 * k + 1] = c.y; uu[4 * k + 2] = c.z; uu[4 * k + 3] = c.w;
        }
        if constexpr (PL) {
            // the four arrays as fp16 planes (see PL).  Lane (j, h) owns row j; k-step ks of an H2Op holds channels
            // 16 ks + 4 h + 0..3 (dwords 0, 1) and 16 ks + 8 + 4 h + 0..3 (dwords 2, 3): 8-byte pieces 4 ks + 2 grp + h.
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            H2Op oz;
            lb_split16(zz, 16384.f, oz);
            char* const rowb = reinterpret_cast<char*>(grp) + j * 64;
            const int key = (j >> 2) & 7;
            auto put_planes = [&](int slot, const H2Op& o, bool zero) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4_t hv = __builtin_bit_cast(u32x4_t, o.h[ks]), mv = __builtin_bit_cast(u32x4_t, o.m[ks]);
                    if (zero) { hv = u32x4_t{0u, 0u, 0u, 0u}; mv = hv; }
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int pos = ((4 * ks + 2 * g2 + h) ^ key) * 8;
                        *reinterpret_cast<u32x2_t*>(rowb + slot * 4096 + pos) = u32x2_t{hv[2 * g2], hv[2 * g2 + 1]};
                        *reinterpret_cast<u32x2_t*>(rowb + slot * 4096 + 2048 + pos) = u32x2_t{mv[2 * g2], mv[2 * g2 + 1]};
                    }
                }
            };
            if (HAS_DO || HAS_U) { put_planes(0, od, !valid); put_planes(1, oz, false); }
            put_planes(2, oa, false);
            put_planes(3, og, false);
            return;
        }
        // the four patches: dout (0 beyond T), z, da, dg
        const float mvj = valid ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = j * 32 + (((2 * q + h) ^ (j & 7)) << 2);
            *reinterpret_cast<float4*>(tf + o) = make_float4(dob[4 * q] * mvj, dob[4 * q + 1] * mvj, dob[4 * q + 2] * mvj, dob[4 * q + 3] * mvj);
            *reinterpret_cast<float4*>(tg + o) = make_float4(zz[4 * q], zz[4 * q + 1], zz[4 * q + 2], zz[4 * q + 3]);
            *reinterpret_cast<float4*>(tv + o) = make_float4(da[4 * q], da[4 * q + 1], da[4 * q + 2], da[4 * q + 3]);
            *reinterpret_cast<float4*>(tu + o) = make_float4(dg[4 * q], dg[4 * q + 1], dg[4 * q + 2], dg[4 * q + 3]);
        }
    };
    // returns true when the tile is complete and the eight stores were issued unconditionally (they can then be left in
    // flight across the next s_waitcnt vmcnt)
    auto store_vu = [&](int tile) -> bool {
        int b, t0;
        decode(tile, b, t0);
        const bool full = t0 + 32 <= T;
        auto st = [&](float* p, const float* v) {
            if constexpr (MULTI) st16_sc1(p, v[0], v[1], v[2], v[3]);     // read by other XCDs within this launch
            else *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        };
        if (full) {
            float* pv = Vout + ((long long)b * T + t0 + lr) * 32 + ((lp ^ lr) << 2);
            float* pu = Uout + ((long long)b * T + t0 + lr) * 32 + ((lp ^ lr) << 2);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                st(pv + k * 256, vv + 4 * k);
                st(pu + k * 256, uu + 4 * k);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 8 * k + lr;
                if (t0 + r < T) {
                    const long long o = ((long long)b * T + t0 + r) * 32 + ((lp ^ (r & 7)) << 2);
                    st(Vout + o, vv + 4 * k);
                    st(Uout + o, uu + 4 * k);
                }
            }
        }
        return full;
    };
    struct WOps { float a_da[16], a_dg[16], a_do[16], b_z[16], b_xc[16], b_xo[16]; };
    auto take = [&](int tile, const float* grp, const float (&xc)[16], const float (&xo)[16], WOps& w) {
        int b, t0;
        decode(tile, b, t0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int r = 2 * s + h;
            const int tt = t0 + r;
            const int po = r * 32 + ((((j >> 2) ^ (r & 7)) << 2) | (j & 3));
            w.a_do[s] = grp[po]; w.b_z[s] = grp[1024 + po]; w.a_da[s] = grp[2048 + po]; w.a_dg[s] = grp[3072 + po];
            w.b_xc[s] = xc[s] * (tt < T ? 1.f : 0.f);
            w.b_xo[s] = xo[s] * ((tt < T && tt - d >= 0) ? 1.f : 0.f);
        }
    };
    struct WOpsT { H2Op da, dg, dov, z; };
    auto take_t = [&](const float* grp, WOpsT& w) {
        const char* base = reinterpret_cast<const char*>(grp);
        const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        auto frag = [&](int slot, int plane, int ks) {
            const int rb = 16 * ks + 8 * (g >> 1);
            const int c = 4 * (g & 1) + p;                           // logical 8-byte piece: channels 16 (g & 1) + 4 p ..
            const int r0 = rb + q, r1 = rb + 4 + q;
            typedef short s16x4_t __attribute__((ext_vector_type(4)));
            typedef short s16x8_t __attribute__((ext_vector_type(8)));
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(
                base + slot * 4096 + plane * 2048 + r0 * 64 + ((c ^ ((r0 >> 2) & 7)) * 8)));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(
                base + slot * 4096 + plane * 2048 + r1 * 64 + ((c ^ ((r1 >> 2) & 7)) * 8)));
            s16x8_t v;
            v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
            return __builtin_bit_cast(h16x8, v);
        };
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (HAS_DO || HAS_U) {
                w.dov.h[ks] = frag(0, 0, ks); w.dov.m[ks] = frag(0, 1, ks);
                w.z.h[ks] = frag(1, 0, ks); w.z.m[ks] = frag(1, 1, ks);
            }
            w.da.h[ks] = frag(2, 0, ks); w.da.m[ks] = frag(2, 1, ks);
            w.dg.h[ks] = frag(3, 0, ks); w.dg.m[ks] = frag(3, 1, ks);
        }
    };
    // x[t], x[t-d] of the tile (rows in tr order), masked, split with the wave's own scale; then the five products
    auto wgrad_t = [&](int tile, const WOpsT& w, const float (&xc)[16], const float (&xo)[16]) {
        int b, t0;
        decode(tile, b, t0);
        float bxc[16], bxo[16];
        float mx = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int tt = t0 + 16 * (s >> 3) + 8 * h + (s & 7);
            bxc[s] = xc[s] * (tt < T ? 1.f : 0.f);
            bxo[s] = xo[s] * ((tt < T && tt - d >= 0) ? 1.f : 0.f);
            mx = fmaxf(mx, fmaxf(fabsf(bxc[s]), fabsf(bxo[s])));
        }
        mx = lb_wave_max(mx);
        float sx, ix;
        lb_pow2_scale(mx, sx, ix);
        H2Op oxc, oxo;
        lb_split16(bxc, sx, oxc);
        lb_split16(bxo, sx, oxo);
        const float u = wg_ig * ix;
        lb_h2_product(w.da, oxc, u, aWf1);
        lb_h2_product(w.da, oxo, u, aWf0);
        lb_h2_product(w.dg, oxc, u, aWg1);
        lb_h2_product(w.dg, oxo, u, aWg0);
        if (HAS_DO || HAS_U) lb_h2_product(w.dov, w.z, wg_id * (1.f / 16384.f), aWp);
    };
    auto wgrad = [&](const WOps& w) {
        if constexpr (H2W) {
            float mg = 0.f, mx = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                mg = fmaxf(fmaxf(mg, fabsf(w.a_da[s])), fmaxf(fabsf(w.a_dg[s]), (HAS_DO || HAS_U) ? fabsf(w.a_do[s]) : 0.f));
                mx = fmaxf(mx, fmaxf(fabsf(w.b_xc[s]), fabsf(w.b_xo[s])));
            }
            mg = lb_wave_max(mg);
            mx = lb_wave_max(mx);
            float sg, ig, sx, ix;
            lb_pow2_scale(mg, sg, ig);
            lb_pow2_scale(mx, sx, ix);
            H2Op oda, odg, oxc, oxo;
            lb_split16(w.a_da, sg, oda);
            lb_split16(w.a_dg, sg, odg);
            lb_split16(w.b_xc, sx, oxc);
            lb_split16(w.b_xo, sx, oxo);
            const float u = ig * ix;
            lb_h2_product(oda, oxc, u, aWf1);
            lb_h2_product(oda, oxo, u, aWf0);
            lb_h2_product(odg, oxc, u, aWg1);
            lb_h2_product(odg, oxo, u, aWg0);
            if (HAS_DO || HAS_U) {
                H2Op odo, oz;
                lb_split16(w.a_do, sg, odo);
                lb_split16(w.b_z, 16384.f, oz);
                lb_h2_product(odo, oz, ig * (1.f / 16384.f), aWp);
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            aWf1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w.a_da[s], w.b_xc[s], aWf1, 0, 0, 0);
            aWf0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w.a_da[s], w.b_xo[s], aWf0, 0, 0, 0);
            aWg1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w.a_dg[s], w.b_xc[s], aWg1, 0, 0, 0);
            aWg0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w.a_dg[s], w.b_xo[s], aWg0, 0, 0, 0);
            if (HAS_DO || HAS_U) aWp = __builtin_amdgcn_mfma_f32_32x32x2f32(w.a_do[s], w.b_z[s], aWp, 0, 0, 0);
        }
    };

    // The five accumulator tiles of a wave -> LDS and the workgroup's partial tile -> memory in ONE layout,
    //     float index ((k * 4 + r / 4) * 64 + lane) * 4 + r % 4        (k: Wf0, Wf1, Wg0, Wg1, Wp; r: accumulator register),
    // so that both the LDS traffic and the partial tile's stores are 128-bit (k_layer_bwd_reduce_all decodes the same layout)
    auto acc_to_lds = [&](float* red) {
        if constexpr (MULTI) {
            // a wait that gave up means this launch's results are void: make that LOUD -- a NaN in the layer's weight gradient
            // (the next loss is NaN) instead of a silently wrong step.  (sync[0] says the same to whoever looks.)
            if (gave_up) aWf0[0] = __builtin_nanf("");
        }
        const f32x16* accs[5] = {&aWf0, &aWf1, &aWg0, &aWg1, &aWp};
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(red + ((k * 4 + q) * 64 + lane) * 4) =
                    make_float4((*accs[k])[4 * q], (*accs[k])[4 * q + 1], (*accs[k])[4 * q + 2], (*accs[k])[4 * q + 3]);
    };
    // every thread adds the four waves' copies of its 20 elements in a fixed order and stores them
    auto sum_to_part = [&](const float* rb, int wave_stride, float* __restrict__ o) {
#pragma unroll
        for (int i = 0; i < kPartFloats / 4 / (64 * NW); ++i) {
            const int e4 = (threadIdx.x + i * 64 * NW) * 4;
            const float4 a = *reinterpret_cast<const float4*>(rb + e4);
            const float4 b = *reinterpret_cast<const float4*>(rb + wave_stride + e4);
            const float4 c = *reinterpret_cast<const float4*>(rb + 2 * wave_stride + e4);
            const float4 d4 = *reinterpret_cast<const float4*>(rb + 3 * wave_stride + e4);
            *reinterpret_cast<float4*>(o + e4) = make_float4((a.x + b.x) + (c.x + d4.x), (a.y + b.y) + (c.y + d4.y),
                                                             (a.z + b.z) + (c.z + d4.z), (a.w + b.w) + (c.w + d4.w));
        }
    };
    // ---- prologue: weights -> LDS, first tile's operands, its first half -----------------------------------
    constexpr int kThreads = 64 * NW;
    constexpr int NK = 512 / kThreads;                  // float4 pieces of Wf (and of Wg) per thread
    float4 s_wf[NK], s_wg[NK];
    float4 s_wp;
    auto load_weights = [&]() {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            s_wf[k] = reinterpret_cast<const float4*>(Wf)[threadIdx.x + k * kThreads];
            s_wg[k] = reinterpret_cast<const float4*>(Wg)[threadIdx.x + k * kThreads];
        }
        s_wp = reinterpret_cast<const float4*>(Wp)[threadIdx.x & 255];
    };
    float4 dza[4], dzb[4];
    float xc[16], xo[16];
    bool any = first < last;
    bool stores_in_flight = false;                     // the last eight vector-memory operations are unconditional V/U stores
    // The layer's weight images from the registers load_weights() filled.  H2W: `between` runs after the weight loads have
    // been waited for and before the LDS-only barrier of the maximum exchange -- what it requests travels under the split.
    auto build_images = [&](auto&& between) {
        if constexpr (H2W) {
            // fp16 x 2 images of the three weight matrices in A-operand order: 1 KB per (matrix, tap, k-step, part), lane
            // (j, h) element e = W[cd = bch(8 ks + e, h)][cr = j][tap] (Wp: [cr = bch(..)][cd = j]) scaled by one power of two
            // taken from the largest weight of the layer (every workgroup sees all of them: 20 values per thread)
            float mw = 0.f;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                mw = fmaxf(mw, fmaxf(fmaxf(fabsf(s_wf[k].x), fabsf(s_wf[k].y)), fmaxf(fabsf(s_wf[k].z), fabsf(s_wf[k].w))));
                mw = fmaxf(mw, fmaxf(fmaxf(fabsf(s_wg[k].x), fabsf(s_wg[k].y)), fmaxf(fabsf(s_wg[k].z), fabsf(s_wg[k].w))));
            }
            mw = fmaxf(mw, fmaxf(fmaxf(fabsf(s_wp.x), fabsf(s_wp.y)), fmaxf(fabsf(s_wp.z), fabsf(s_wp.w))));
            mw = lb_wave_max(mw);
            float* red = wbase + kCWaves * kCWaveFloats - 8;   // last 32 bytes of the slot area: free until tile data lands there
            if (lane == 0) red[wv] = mw;
            // the first tile's fetch goes out only now, behind the weight loads (which the maximum above has waited for), and
            // the exchange of the four maxima uses an LDS-only barrier: splitting and scattering the images (~2 k cycles) runs
            // under the fetch's HBM round trip instead of behind it (a __syncthreads() here would drain it)
            between();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            mw = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            float sw;
            lb_pow2_scale(mw, sw, w_inv);
            char* img = reinterpret_cast<char*>(dyn);
            auto put = [&](int mat_tap, int kidx, int jj, float v) {      // kidx: the contraction channel, jj: the output row
                const int hh = (kidx >> 2) & 1, ks = kidx >> 4, e = (kidx & 3) + 4 * ((kidx >> 3) & 1);
                const float xs = v * sw;
                const _Float16 hv = (_Float16)xs;
                const _Float16 mv = (_Float16)(xs - (float)hv);
                char* dst = img + ((mat_tap * 2 + ks) * 2) * 1024 + (hh * 32 + jj) * 16 + e * 2;
                *reinterpret_cast<_Float16*>(dst) = hv;
                *reinterpret_cast<_Float16*>(dst + 1024) = mv;
            };
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int e0 = (threadIdx.x + k * kThreads) * 4;        // flat index into W[cd][cr][tap]
                const int cd = e0 >> 6, cr = (e0 >> 1) & 31;
                put(0, cd, cr, s_wf[k].x); put(1, cd, cr, s_wf[k].y); put(0, cd, cr + 1, s_wf[k].z); put(1, cd, cr + 1, s_wf[k].w);
                put(2, cd, cr, s_wg[k].x); put(3, cd, cr, s_wg[k].y); put(2, cd, cr + 1, s_wg[k].z); put(3, cd, cr + 1, s_wg[k].w);
            }
            if (threadIdx.x < 256) {
                const int e0 = threadIdx.x * 4;                           // flat index into Wp[cr][cd]
                const int cr = e0 >> 5, cd = e0 & 31;
                put(4, cr, cd, s_wp.x); put(4, cr, cd + 1, s_wp.y); put(4, cr, cd + 2, s_wp.z); put(4, cr, cd + 3, s_wp.w);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                reinterpret_cast<float4*>(lWf)[threadIdx.x + k * kThreads] = s_wf[k];
                reinterpret_cast<float4*>(lWg)[threadIdx.x + k * kThreads] = s_wg[k];
            }
            if (threadIdx.x < 256) reinterpret_cast<float4*>(lWp)[threadIdx.x] = s_wp;
        }
    };
    load_weights();
    if (!H2W && any) { fetch_a(first, pbase, dza); fetch_x(first, xc, xo); }
    build_images([&]() { if (any) { fetch_a(first, pbase, dza); fetch_x(first, xc, xo); } });
    __syncthreads();                                   // (also drains vmcnt: the first tile has landed)
#ifdef WN_BWD_STAMPS
    unsigned long long st_wait = 0, st_take = 0, st_body = 0, st_n = 0, st_t0 = 0, st_t1 = 0;
    BST(st_k1);
#endif
    int tile = first;
    int it = 0;
#ifdef WN_MULTI_STAMPS
    if constexpr (MULTI) { unsigned long long tt; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory");
          if (lane == 0 && blockIdx.x < 256) g_multi_stamps[0][blockIdx.x][wv] = tt; }
#endif
    for (;;) {                                         // MULTI: one pass per layer of the launch
    if (any) {
        if (first + stride < last) {
            dep_wait(first + stride, dep_1);
            fetch_a(first + stride, pbase + 4096, dzb);
        }
        if (MULTI && first + 2 * stride < last) dep_request_lds(first + 2 * stride);
        phase_a(first, pbase, dza);
        stores_in_flight = store_vu(first);
    }
    pub = first;
    // Loop over the tiles that have a successor (one basic block: no branch between the MFMAs of the two tiles, and the
    // accumulators flow through a single path -- an if/else around the weight-gradient MFMAs made the compiler copy all
    // 80 accumulator registers every iteration); the last tile's weight gradients follow the loop.
    it = 0;
    tile = first;
#ifdef WN_BWD_STAMPS
    BST(st_t0);
#endif
    for (; tile + stride < last; tile += stride, ++it) {
        float* grp = pbase + (it & 1) * 4096;          // patches of `tile`
        float* ngrp = pbase + ((it + 1) & 1) * 4096;   // f, g, V, U of tile + stride
        // what the previous body fetched (this tile's x, the next tile's f, g, V, U, dz) has had a whole body to land;
        // its V/U stores were issued last and may stay in flight
#ifdef WN_BWD_STAMPS
        unsigned long long c0, c1, c2, c3;
        BST(c0);
#endif
        if (stores_in_flight) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (MULTI) {                             // the stores of the tile before `tile` have been counted
            if (pub < tile) { publish(pub); pub += stride; }
        }
#ifdef WN_BWD_STAMPS
        BST(c1);
#endif
        WOps w;
        WOpsT wt;
        float xcw[16], xow[16];                           // this tile's x (the registers are fetched into again below)
        if constexpr (PL) {
            take_t(grp, wt);
            wg_ig = pa_ig; wg_id = pa_id;
#pragma unroll
            for (int s = 0; s < 16; ++s) { xcw[s] = xc[s]; xow[s] = xo[s]; }
        } else {
            take(tile, grp, xc, xo, w);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the patches are in registers: their slots are free
#ifdef WN_BWD_STAMPS
        BST(c2);
#endif
#pragma unroll
        for (int q = 0; q < 4; ++q) dza[q] = dzb[q];
        if constexpr (!PL) {
            // (dzb came from plain loads a body ago: hipcc waits for them with vmcnt(0) where it copies them -- keep that in front
            // of the x requests below, which it otherwise issues first and then waits for with everything else)
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(dza[q].x), "+v"(dza[q].y), "+v"(dza[q].z), "+v"(dza[q].w));
            __builtin_amdgcn_sched_barrier(0);
        }
        fetch_x(tile + stride, xc, xo);
        if (tile + 2 * stride < last) {
            dep_wait(tile + 2 * stride, dep_take());
            fetch_a(tile + 2 * stride, grp, dzb);
        }
        if (MULTI && tile + 3 * stride < last) dep_request_lds(tile + 3 * stride);
        // one basic block: 80 weight-gradient MFMAs of `tile` and the first half of the next tile
        if constexpr (PL) wgrad_t(tile, wt, xcw, xow);
        else wgrad(w);
        phase_a(tile + stride, ngrp, dza);
        stores_in_flight = store_vu(tile + stride);
#ifdef WN_BWD_STAMPS
        BST(c3);
        st_wait += c1 - c0; st_take += c2 - c1; st_body += c3 - c2; ++st_n;
#endif
    }
#ifdef WN_BWD_STAMPS
    BST(st_loop_end);
#endif
    if (any) {
        if (stores_in_flight) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (MULTI) {                             // the last tile but one: before the last weight-gradient products, not after
            if (pub < tile) { publish(pub); pub += stride; }
        }
        if constexpr (PL) {
            WOpsT wt;
            take_t(pbase + (it & 1) * 4096, wt);
            wg_ig = pa_ig; wg_id = pa_id;
            wgrad_t(tile, wt, xc, xo);
        } else {
            WOps w;
            take(tile, pbase + (it & 1) * 4096, xc, xo, w);
            wgrad(w);
        }
    }
    if constexpr (!MULTI) {
        break;
    } else {
        if (li + 1 >= ma->n) break;
        // ---- layer boundary (workgroup-local: no grid barrier).  (1) this layer's V / U stores have been counted: its last
        // tiles are published; (2) everything of the next layer that the forward pass wrote is requested: weights, and of its
        // first tile z, sigmoid (LDS-DMA as inline asm into group 0's first two slots), dz_skip and x (registers), together
        // with the dataflow words of its first two tiles; (3) the finished layer's accumulators -> this wave's own region of
        // LDS (the V / U slots of group 0 and group 1: dead), one LDS-only barrier, fixed-order sum, partial tile; (4) the next
        // layer's weight images; (5) the first tile's V, U as soon as its words allow.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef WN_MULTI_STAMPS
        { unsigned long long tt; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory");
          if (lane == 0 && blockIdx.x < 256) { g_multi_stamps[li + 1][blockIdx.x][wv] = tt; g_multi_spins[li][blockIdx.x][wv] = dbg_spin; }
          dbg_spin = 0; }
#endif
        if (any) {
            while (pub <= tile) { publish(pub); pub += stride; }
        }
#ifdef WN_MULTI_STAMPS
        unsigned long long sg0, sg1, sg2, sg3, sg4, sg5; MST(sg0);
#endif
        float* const part_done = part;
        ++li;
        set_layer(li);
        partition();
        any = first < last;
        load_weights();
        unsigned dep_0 = 0xffffffffu;
        dep_1 = 0xffffffffu;
        if (any) {
            dep_0 = dep_request(first);
            if (first + stride < last) dep_1 = dep_request(first + stride);
            fetch_some(first, pbase, dza, 1, true);
            fetch_x(first, xc, xo);
        }
        acc_to_lds(pbase + 2048);
#pragma unroll
        for (int r = 0; r < 16; ++r) { aWf0[r] = 0.f; aWf1[r] = 0.f; aWg0[r] = 0.f; aWg1[r] = 0.f; aWp[r] = 0.f; }
#ifdef WN_MULTI_STAMPS
        MST(sg1);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        sum_to_part(wbase + 2048, kCWaveFloats, part_done + (long long)blockIdx.x * kPartFloats);
#ifdef WN_MULTI_STAMPS
        MST(sg2);
#endif
        build_images([]() {});
#ifdef WN_MULTI_STAMPS
        MST(sg3);
#endif
        if (any) {
            dep_wait(first, dep_0);
            fetch_some(first, pbase, dza, 2, true);
        }
#ifdef WN_MULTI_STAMPS
        MST(sg4);
#endif
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // images published, first tile landed
#ifdef WN_MULTI_STAMPS
        MST(sg5);
        if (lane == 0 && blockIdx.x < 256) {
            unsigned long long* q = g_multi_seg[blockIdx.x][wv];
            q[0] += sg1 - sg0; q[1] += sg2 - sg1; q[2] += sg3 - sg2; q[3] += sg4 - sg3; q[4] += sg5 - sg4; q[5] += 1;
        }
#endif
    }
    }
#ifdef WN_BWD_STAMPS
    BST(st_k2);
    {
        st_t1 = st_loop_end;
        const int gw = blockIdx.x * NW + wv;
        if (lane == 0 && gw < 1024) {
            g_bwd_stamps[gw * 8 + 0] = st_wait; g_bwd_stamps[gw * 8 + 1] = st_take; g_bwd_stamps[gw * 8 + 2] = st_body;
            g_bwd_stamps[gw * 8 + 3] = st_n; g_bwd_stamps[gw * 8 + 4] = st_t1 - st_t0;
            g_bwd_stamps[gw * 8 + 5] = st_k1 - st_k0; g_bwd_stamps[gw * 8 + 6] = st_t0 - st_k1; g_bwd_stamps[gw * 8 + 7] = st_k2 - st_loop_end;
        }
    }
#endif
    // ---- the four waves' accumulators -> LDS (the slot groups are dead), ONE barrier, then every thread adds them in a fixed
    // order for its 20 elements and writes the workgroup's partial tile with coalesced stores.  (The two-level tree it
    // replaces cost four barriers and left the 80 stores of the tile to one wave.)
    __syncthreads();                                   // every wave is done with its slot groups
    acc_to_lds(wbase + wv * kPartFloats);
    __syncthreads();
    sum_to_part(wbase, kPartFloats, part + (long long)blockIdx.x * kPartFloats);
}

template <bool HAS_DO, bool HAS_U, bool HAS_DZ, bool FROM_Z, bool H2W>
__global__ __launch_bounds__(256, 1) void k_layer_bwd_chainsp(
    const float* __restrict__ x, const float* __restrict__ f, const float* __restrict__ g,
    const float* __restrict__ Wp, const float* __restrict__ Wf, const float* __restrict__ Wg,
    const float* __restrict__ Vin, const float* __restrict__ Uin, int dU, int vu_t0,
    const float* __restrict__ dzs, int dz_t0, float* __restrict__ Vout, float* __restrict__ Uout,
    float* __restrict__ part, int B, int T, int d, int Z, int tile_lo, int tiles_per_b, int ntiles) {
    chain_body<HAS_DO, HAS_U, HAS_DZ, FROM_Z, H2W, false>(x, f, g, Wp, Wf, Wg, Vin, Uin, dU, vu_t0, dzs, dz_t0, Vout, Uout, part,
                                                          B, T, d, Z, tile_lo, tiles_per_b, ntiles, nullptr);
}

// Layers a.layer[0] > a.layer[1] > ... of a stack in one launch (see MULTI above); every layer has V, U and dz_skip inputs.
template <bool FROM_Z, bool H2W>
__global__ __launch_bounds__(256, 1) void k_layer_bwd_chain_multi(const ChainArgs a) {
    chain_body<true, true, true, FROM_Z, H2W, true>(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0,
                                                    nullptr, a.dz_t0, nullptr, nullptr, nullptr, a.B, a.T, 1, 0, 0, 1, 1, &a);
}

// dx[t] = V[t] + U[t + dU]   (the split gradient of the stack input, materialised once at the bottom)
__global__ void k_chain_combine(const float* __restrict__ V, const float* __restrict__ U, float* __restrict__ dx,
                                int B, int T, int dU, int vu_t0);

// dW += sum over workgroups of the partial tiles.  Thread = one element of the five tiles; blockIdx.y
// splits the workgroup range so that enough loads are in flight; kRedParts light atomics per address.
static constexpr int kRedParts = 16;
__global__ void k_layer_bwd_reduce(const float* __restrict__ part, int nwg, float* __restrict__ dWf,
                                   float* __restrict__ dWg, float* __restrict__ dWp) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;      // 0 .. 5119: tile*1024 + r*64 + lane
    if (e >= kPartFloats) return;
    const int per = (nwg + kRedParts - 1) / kRedParts;
    const int w0 = blockIdx.y * per, w1 = min(nwg, w0 + per);
    float acc = 0.f;
#pragma unroll 8
    for (int w = w0; w < w1; ++w) acc += part[(long long)w * kPartFloats + e];
    if (w1 <= w0) return;
    const int tile = e >> 10, r = (e >> 6) & 15, lane = e & 63;
    const int j = lane & 31, i = bch(r, lane >> 5);           // D layout: element [row bch(r,h)][column j]
    if (tile < 4) {                                           // dW[cd=i][cr=j][k]
        float* dW = tile < 2 ? dWf : dWg;
        if (dW) atomicAdd(dW + (i * 32 + j) * 2 + (tile & 1), acc);
    } else if (dWp) {
        atomicAdd(dWp + i * 32 + j, acc);                     // dWp[cr=i][cd=j]
    }
}

// The same sum for up to kRedAllMax layers in one launch (blockIdx.z = layer): the stack's chained backward keeps
// every layer's partial tiles and reduces them all at the end, instead of 40 small launches between the layer kernels.
static constexpr int kRedAllMax = 64;
struct RedAllArgs { float* dWf[kRedAllMax]; float* dWg[kRedAllMax]; float* dWp[kRedAllMax]; int nwg[kRedAllMax]; };
// Deterministic: a block owns 64 elements; its four waves each sum a quarter of the workgroups' tiles (whole 256-byte rows
// per load), the quarters are added in a fixed order through LDS, and the owner adds the total to dW without an atomic.
// (gridDim.z = layers + kCombineSlices when the stack's input gradient is wanted: the extra z-slices materialise dx = V + U[t + dU]
// -- the k_chain_combine pass, grid-stride -- under the reduction instead of behind it as a launch of its own; with ONE slice of
// 80 blocks the pass took 66 us and the launch with it: it needs the blocks of a launch of its own)
struct CombineArgs { const float* V; const float* U; float* dx; int B, T, dU, vu_t0; };
static constexpr int kCombineSlices = 24;               // x 80 blocks x 256 threads: two float4 per thread at config 2
__device__ __forceinline__ void chain_combine_at(const CombineArgs& c, long long i) {
    const long long col = i >> 3;
    const int t = (int)(col % c.T);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);          // rows below vu_t0 were not written: exact zeros
    if (t >= c.vu_t0) v = reinterpret_cast<const float4*>(c.V)[i];
    if (t + c.dU < c.T && t + c.dU >= c.vu_t0) {
        const float4 u = reinterpret_cast<const float4*>(c.U)[i + (long long)c.dU * 8];
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    reinterpret_cast<float4*>(c.dx)[i] = v;
}
__global__ void k_layer_bwd_reduce_all(const float* __restrict__ part, long long layer_stride, RedAllArgs a, int nlayers,
                                       CombineArgs cmb) {
    if ((int)blockIdx.z >= nlayers) {                        // kCombineSlices z-slices of the grid: enough blocks for a 50 MB pass
        const long long n4 = (long long)cmb.B * cmb.T * 8;
        const long long nthr = (long long)(gridDim.z - nlayers) * gridDim.x * blockDim.x;
        for (long long i = ((long long)(blockIdx.z - nlayers) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += nthr)
            chain_combine_at(cmb, i);
        return;
    }
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, sp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;                    // kPartFloats is a multiple of 64
    const int l = blockIdx.z;
    const int nwg = a.nwg[l];
    const float* __restrict__ p = part + (long long)l * layer_stride + e;
    const int per = (nwg + 3) / 4;
    const int w0 = sp * per, w1 = min(nwg, w0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = w0;
    for (; w + 4 <= w1; w += 4) {
        s0 += p[(long long)(w + 0) * kPartFloats]; s1 += p[(long long)(w + 1) * kPartFloats];
        s2 += p[(long long)(w + 2) * kPartFloats]; s3 += p[(long long)(w + 3) * kPartFloats];
    }
    for (; w < w1; ++w) s0 += p[(long long)w * kPartFloats];
    red[sp][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sp != 0) return;
    const float acc = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int tile = e >> 10, x = e & 1023;                  // partial-tile layout: ((tile * 4 + r / 4) * 64 + lane) * 4 + r % 4
    const int r = 4 * (x >> 8) + (x & 3), ln = (x >> 2) & 63;
    const int j = ln & 31, i = bch(r, ln >> 5);
    if (tile < 4) {
        float* dW = tile < 2 ? a.dWf[l] : a.dWg[l];
        if (dW) dW[(i * 32 + j) * 2 + (tile & 1)] += acc;
    } else if (a.dWp[l]) {
        a.dWp[l][i * 32 + j] += acc;
    }
}

__global__ __launch_bounds__(256, 2) void k_layer_bwd_p2(
    const float* __restrict__ Wf, const float* __restrict__ Wg, const float* __restrict__ dout,
    const float* __restrict__ dab, float* __restrict__ dx, int B, int T, int d, int tiles_per_b, int ntiles) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    // A operands: lane (i=cr,h), step s: W[cd=ch(s,h)][cr=i][k]
    float wf0[16], wf1[16], wg0[16], wg1[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float2 a = *reinterpret_cast<const float2*>(Wf + (bch(s, h) * 32 + j) * 2);
        const float2 c = *reinterpret_cast<const float2*>(Wg + (bch(s, h) * 32 + j) * 2);
        wf0[s] = a.x; wf1[s] = a.y; wg0[s] = c.x; wg1[s] = c.y;
    }
    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int b = tile / tiles_per_b;
        const int t = (tile - b * tiles_per_b) * 32 + j;
        const bool valid = t < T;
        const bool has_new = valid && (t + d) < T;
        const long long row = ((long long)b * T + t) * 32 + 4 * h;
        const int tc = valid ? t : T - 1;                                     // clamped rows, masked values
        const long long rowc = ((long long)b * T + tc) * 32 + 4 * h;
        const long long drow = ((long long)b * T + tc) * 64 + 4 * h;
        const long long nrow = ((long long)b * T + (tc + d < T ? tc + d : T - 1)) * 64 + 4 * h;
        const float mn = has_new ? 1.f : 0.f;
        f32x16 acc;
        float a0[16], g0[16], a1[16], g1[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 o = make_float4(0, 0, 0, 0);
            if (dout) o = *reinterpret_cast<const float4*>(dout + rowc + 8 * q);          // uniform condition
            const float4 va = *reinterpret_cast<const float4*>(dab + drow + 8 * q);
            const float4 vg = *reinterpret_cast<const float4*>(dab + drow + 32 + 8 * q);
            float4 na = *reinterpret_cast<const float4*>(dab + nrow + 8 * q);
            float4 ng = *reinterpret_cast<const float4*>(dab + nrow + 32 + 8 * q);
            na.x *= mn; na.y *= mn; na.z *= mn; na.w *= mn;
            ng.x *= mn; ng.y *= mn; ng.z *= mn; ng.w *= mn;
            acc[4 * q] = o.x; acc[4 * q + 1] = o.y; acc[4 * q + 2] = o.z; acc[4 * q + 3] = o.w;
            a0[4 * q] = va.x; a0[4 * q + 1] = va.y; a0[4 * q + 2] = va.z; a0[4 * q + 3] = va.w;
            g0[4 * q] = vg.x; g0[4 * q + 1] = vg.y; g0[4 * q + 2] = vg.z; g0[4 * q + 3] = vg.w;
            a1[4 * q] = na.x; a1[4 * q + 1] = na.y; a1[4 * q + 2] = na.z; a1[4 * q + 3] = na.w;
            g1[4 * q] = ng.x; g1[4 * q + 1] = ng.y; g1[4 * q + 2] = ng.z; g1[4 * q + 3] = ng.w;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf1[s], a0[s], acc, 0, 0, 0);   // tap 1 reads x[t]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wg1[s], g0[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf0[s], a1[s], acc, 0, 0, 0);   // tap 0 of column t+d reads x[t]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wg0[s], g1[s], acc, 0, 0, 0);
        }
        if (valid) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(dx + row + 8 * q) =
                    make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
    }
}

int mfma_layer_bwd(const float* x, const float* f, const float* g, const float* Wf, const float* Wg,
                   const float* Wp, const float* dout, const float* dzs, float* dx, float* dWf, float* dWg,
                   float* dWp, float* dab, int B, int T, int d, int Z, hipStream_t s) {
    const int tiles_per_b = (T + 31) / 32;
    const long long nt = (long long)B * tiles_per_b;
    WN_CHECK_SHAPE(nt < (1ll << 31), "mfma_layer_bwd: too many tiles");
    const int ntiles = (int)nt;
    int blocks = (ntiles + 3) / 4;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    float* part = dab + (size_t)B * T * 64;          // workspace tail: blocks x kPartFloats partial tiles
#define P1_LAUNCH(DO, DZ)                                                                                  \
    hipLaunchKernelGGL((k_layer_bwd_p1<DO, DZ>), dim3(blocks), dim3(256), 0, s, x, f, g, Wp, dout, dzs, dab, part, \
                       B, T, d, Z, tiles_per_b, ntiles)
    if (dout && dzs) P1_LAUNCH(true, true);
    else if (dout) P1_LAUNCH(true, false);
    else P1_LAUNCH(false, true);
#undef P1_LAUNCH
    WN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_layer_bwd_reduce, dim3(kPartFloats / 256, kRedParts), dim3(256), 0, s, part, blocks, dWf, dWg,
                       dout ? dWp : (float*)nullptr);
    WN_LAUNCH_CHECK();
    if (dx) {
        hipLaunchKernelGGL(k_layer_bwd_p2, dim3(blocks), dim3(256), 0, s, Wf, Wg, dout, dab, dx, B, T, d,
                           tiles_per_b, ntiles);
        WN_LAUNCH_CHECK();
    }
    return WN_OK;
}

size_t mfma_layer_bwd_extra_ws_floats() { return (size_t)kMaxBlocks * kPartFloats; }

// One layer of the chained backward.  Vin/Uin (either may be NULL): dout[t] = Vin[t] + Uin[t + dU] with rows below
// vu_t0 taken as 0; columns below t_live are not computed (no gradient reaches them).  Returns the number of
// workgroups (= partial weight-gradient tiles written to `part`) through *nwg.
int mfma_layer_bwd_chain(const float* x, const float* f, const float* g, const float* Wf, const float* Wg,
                         const float* Wp, const float* Vin, const float* Uin, int dU, int vu_t0, const float* dzs,
                         int dz_t0, float* Vout, float* Uout, float* part, int B, int T, int d, int Z, int t_live,
                         int* nwg, hipStream_t s, bool from_z) {
    const bool h2w = gemm_mode() == WN_GEMM_FP16X2;           // weight-gradient contractions on fp16 x 2 split products
    const int tiles_all = (T + 31) / 32;
    const int tile_lo = t_live > 0 ? t_live / 32 : 0;
    const int tiles_per_b = tiles_all - tile_lo;
    const long long nt = (long long)B * tiles_per_b;
    WN_CHECK_SHAPE(nt < (1ll << 31), "mfma_layer_bwd_chain: too many tiles");
    WN_CHECK_ARG(tiles_per_b > 0, "mfma_layer_bwd_chain: no live column");
    WN_CHECK_ARG(Vin || Uin || dzs, "mfma_layer_bwd_chain: no incoming gradient");
    const int ntiles = (int)nt;
    const int nw = kCWaves;
    int blocks = (ntiles + nw - 1) / nw;
    if (blocks > kCMaxBlocks) blocks = kCMaxBlocks;
    if (nwg) *nwg = blocks;
#define CH_LAUNCH3(DO, UU, DZ, FZ, HW)                                                                            \
    do {                                                                                                           \
        static bool attr_set = false;                                                                              \
        if (!attr_set) {                                                                                           \
            WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer_bwd_chainsp<DO, UU, DZ, FZ, HW>),      \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kCLdsBytes));                   \
            attr_set = true;                                                                                       \
        }                                                                                                          \
        hipLaunchKernelGGL((k_layer_bwd_chainsp<DO, UU, DZ, FZ, HW>), dim3(blocks), dim3(256), kCLdsBytes, s,       \
                           x, f, g, Wp, Wf, Wg, Vin, Uin, dU, vu_t0, dzs, dz_t0, Vout, Uout, part, B, T, d, Z,       \
                           tile_lo, tiles_per_b, ntiles);                                                          \
    } while (0)
#define CH_LAUNCH2(DO, UU, DZ, FZ)                            \
    do {                                                      \
        if (h2w) CH_LAUNCH3(DO, UU, DZ, FZ, true);            \
        else CH_LAUNCH3(DO, UU, DZ, FZ, false);               \
    } while (0)
#define CH_LAUNCH(DO, UU, DZ)                                 \
    do {                                                      \
        if (from_z) CH_LAUNCH2(DO, UU, DZ, true);             \
        else CH_LAUNCH2(DO, UU, DZ, false);                   \
    } while (0)
    const int key = (Vin ? 4 : 0) | (Uin ? 2 : 0) | (dzs ? 1 : 0);
    switch (key) {
        case 7: CH_LAUNCH(true, true, true); break;
        case 6: CH_LAUNCH(true, true, false); break;
        case 5: CH_LAUNCH(true, false, true); break;
        case 4: CH_LAUNCH(true, false, false); break;
        case 3: CH_LAUNCH(false, true, true); break;
        case 2: CH_LAUNCH(false, true, false); break;
        default: CH_LAUNCH(false, false, true); break;
    }
#undef CH_LAUNCH
#undef CH_LAUNCH2
#undef CH_LAUNCH3
    WN_LAUNCH_CHECK();
    return WN_OK;
}

// Entries 0 .. n-1 (stack layers layer[0] > layer[1] > ...) in ONE launch of co-resident workgroups that follow per-tile
// dataflow words -- no grid barrier -- (k_layer_bwd_chain_multi).  The form reading z and sigmoid (FROM_Z), in the arithmetic of the call
// (fp16 x 2 split products, or exact fp32 MFMA under WN_GEMM_BF16X3 / WN_GEMM_FP32: round 6), every layer with V, U
// and dz_skip inputs.  `sync` points at mfma_chain_multi_sync_words(B, T) words of device memory (zeroed here).  *nwg receives the number of partial tiles
// every layer writes (= the grid).
__global__ void k_chain_zero_sync(unsigned* sync, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) sync[i] = i == 1 ? 0xffffffffu : 0u;
}
size_t mfma_chain_multi_sync_words(int B, int T) { return (size_t)kChainSyncHead + (size_t)B * ((T + 31) / 32); }
int mfma_chain_multi_max_layers() { return kChainMaxL; }
int mfma_layer_bwd_chain_multi(int n, const int* layer, const float* const* Wf, const float* const* Wg,
                               const float* const* Wp, const int* d, const int* Z, const int* t_live, const int* vu_t0,
                               const int* dU, const float* x0, const float* xs, const float* z, const float* g,
                               const float* dz, float* const* V, float* const* U, float* part, size_t part_stride,
                               unsigned* sync, int B, int T, int dz_t0, int* nwg, hipStream_t s, bool sync_zeroed) {
    WN_CHECK_ARG(n >= 1 && n <= kChainMaxL, "mfma_layer_bwd_chain_multi: 1..%d layers", kChainMaxL);
    const bool h2w = gemm_mode() == WN_GEMM_FP16X2;           // else: every product on exact fp32 MFMA (bf16x3 / fp32 modes)
    const int tiles_all = (T + 31) / 32;
    ChainArgs a{};
    int blocks = 1;
    for (int i = 0; i < n; ++i) {
        a.Wf[i] = Wf[i]; a.Wg[i] = Wg[i]; a.Wp[i] = Wp[i];
        a.d[i] = d[i]; a.Z[i] = Z[i]; a.vu_t0[i] = vu_t0[i]; a.dU[i] = dU[i]; a.layer[i] = layer[i];
        a.tile_lo[i] = t_live[i] > 0 ? t_live[i] / 32 : 0;
        const long long nt = (long long)B * (tiles_all - a.tile_lo[i]);
        WN_CHECK_SHAPE(nt < (1ll << 31), "mfma_layer_bwd_chain_multi: too many tiles");
        WN_CHECK_ARG(nt > 0, "mfma_layer_bwd_chain_multi: no live column");
        const int bl = (int)((nt + kCWaves - 1) / kCWaves);
        if (bl > blocks) blocks = bl;
    }
    if (blocks > kCMaxBlocks) blocks = kCMaxBlocks;
    // every workgroup must be RESIDENT (they wait for each other's tiles).  Asked PER DEVICE (the stream's): the occupancy of
    // this kernel's 256 threads / kCLdsBytes of LDS x the device's CUs -- not a count cached for the first device a process
    // happened to use.  What the query cannot see (another process on the GPU, a CU mask) is caught at run time: a wait that
    // does not end gives up after ~0.3 s, flags sync[0] and poisons the layer's weight gradient with a NaN, and the optimiser
    // kernels skip a step whose gradient norm is not finite (wn_adam_step, ABI 4) -- a violated assumption costs time and one
    // step, never the GPU and never the training state.
    {
        int dev = 0;
        WN_HIP(hipGetDevice(&dev));
        static int capacity[2][64];                     // per arithmetic and device; 0 = not asked yet (a benign race: every thread writes the same value)
        if (dev < 0 || dev >= 64) { wn::set_error("mfma_layer_bwd_chain_multi: device index %d", dev); return WN_ESHAPE; }
        int& cap = capacity[h2w ? 1 : 0][dev];
        if (!cap) {
            int n_cu = 0, per_cu = 0;
            WN_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
            const void* fn = h2w ? reinterpret_cast<const void*>(k_layer_bwd_chain_multi<true, true>)
                                 : reinterpret_cast<const void*>(k_layer_bwd_chain_multi<true, false>);
            WN_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kCLdsBytes));
            if (h2w) WN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_layer_bwd_chain_multi<true, true>, 256, kCLdsBytes));
            else WN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_layer_bwd_chain_multi<true, false>, 256, kCLdsBytes));
            cap = n_cu * per_cu > 0 ? n_cu * per_cu : -1;
        }
        if (blocks > cap) {
            wn::set_error("mfma_layer_bwd_chain_multi: %d workgroups, %d resident on device %d", blocks, cap, dev);
            return WN_ESHAPE;
        }
    }
    {   // the rotation of the deal (see `partition` in chain_body): entry i + 1 starts where entry i's surplus tiles ended
        const bool xcd = (blocks & 7) == 0 && (B & 7) == 0;
        const int nbx = xcd ? B >> 3 : B;
        const int stride = (xcd ? blocks >> 3 : blocks) * kCWaves;
        int rot = 0;
        for (int i = 0; i < n; ++i) {
            a.rot[i] = rot;
            rot = (int)((rot + (long long)nbx * (tiles_all - a.tile_lo[i])) % stride);
        }
    }
    a.x0 = x0; a.xs = xs; a.z = z; a.g = g; a.dz = dz;
    for (int k = 0; k < 3; ++k) { a.V[k] = V[k]; a.U[k] = U[k]; }
    a.part = part; a.part_stride = (long long)part_stride; a.sync = sync;
    a.n = n; a.B = B; a.T = T; a.dz_t0 = dz_t0;
    if (nwg) *nwg = blocks;
    // the words are zeroed by a KERNEL: a memset node in front of the launch (hipMemsetAsync under stream capture) was not
    // ordered before it when graph replays follow each other without a host synchronisation -- the launch then saw the
    // previous replay's words, every wait was open, and the layers raced on the nearly-right (V, U) of the step before
    // (fast, and wrong in the fourth digit of the loss after a hundred steps)
    // (sync_zeroed: the words are a step plan's, zeroed by wn_plan_prepare -- a kernel too, at the start of the same graph)
    const int nsync = (int)mfma_chain_multi_sync_words(B, T);
    if (!sync_zeroed) hipLaunchKernelGGL(k_chain_zero_sync, dim3(cdiv(nsync, 256)), dim3(256), 0, s, sync, nsync);
    WN_LAUNCH_CHECK();
    if (h2w) hipLaunchKernelGGL((k_layer_bwd_chain_multi<true, true>), dim3(blocks), dim3(256), kCLdsBytes, s, a);
    else hipLaunchKernelGGL((k_layer_bwd_chain_multi<true, false>), dim3(blocks), dim3(256), kCLdsBytes, s, a);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

size_t mfma_chain_part_floats() { return (size_t)kCMaxBlocks * kPartFloats; }

// Sum the partial tiles of L layers (layer l: nwg[l] tiles at part + l * mfma_chain_part_floats()) into their weight
// gradients.  dWp[l] == NULL: that layer had no gradient through its output (the top layer of the stack).
__global__ void k_chain_combine(const float* __restrict__ V, const float* __restrict__ U, float* __restrict__ dx,
                                int B, int T, int dU, int vu_t0) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // float4 index
    if (i >= (long long)B * T * 8) return;
    chain_combine_at(CombineArgs{V, U, dx, B, T, dU, vu_t0}, i);
}

// dx != NULL: also materialise dx = V + U[t + dU] (the stack's input gradient) in the same launch (stacks of <= kRedAllMax layers)
int mfma_chain_reduce_all(const float* part, int L, const int* nwg, float* const* dWf, float* const* dWg,
                          float* const* dWp, hipStream_t s, const float* V, const float* U, float* dx, int B, int T,
                          int dU, int vu_t0) {
    const bool fold = dx != nullptr && L <= kRedAllMax;
    for (int l0 = 0; l0 < L; l0 += kRedAllMax) {
        const int n = L - l0 < kRedAllMax ? L - l0 : kRedAllMax;
        RedAllArgs a{};
        for (int l = 0; l < n; ++l) {
            a.dWf[l] = dWf[l0 + l]; a.dWg[l] = dWg[l0 + l]; a.dWp[l] = dWp[l0 + l]; a.nwg[l] = nwg[l0 + l];
        }
        const CombineArgs c{V, U, dx, B, T, dU, vu_t0};
        hipLaunchKernelGGL(k_layer_bwd_reduce_all, dim3(kPartFloats / 64, 1, n + (fold ? kCombineSlices : 0)), dim3(256), 0, s,
                           part + (size_t)l0 * mfma_chain_part_floats(), (long long)mfma_chain_part_floats(), a, n, c);
        WN_LAUNCH_CHECK();
    }
    if (dx && !fold) return mfma_chain_combine(V, U, dx, B, T, dU, vu_t0, s);
    return WN_OK;
}

int mfma_chain_combine(const float* V, const float* U, float* dx, int B, int T, int dU, int vu_t0, hipStream_t s) {
    const long long n4 = (long long)B * T * 8;
    hipLaunchKernelGGL(k_chain_combine, dim3(cdiv(n4, 256)), dim3(256), 0, s, V, U, dx, B, T, dU, vu_t0);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

}  // namespace wn

#ifdef WN_MULTI_STAMPS
extern "C" __attribute__((visibility("default"))) int wn_debug_multi_stamps(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wn::g_multi_stamps), sizeof(unsigned long long) * (wn::kChainMaxL + 1) * 256 * 4);
}
extern "C" __attribute__((visibility("default"))) int wn_debug_multi_seg(unsigned long long* dst, int zero) {
    if (zero) { static unsigned long long z[256 * 4 * 8]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(wn::g_multi_seg), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wn::g_multi_seg), sizeof(unsigned long long) * 256 * 4 * 8);
}
extern "C" __attribute__((visibility("default"))) int wn_debug_multi_spins(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wn::g_multi_spins), sizeof(unsigned long long) * (wn::kChainMaxL + 1) * 256 * 4);
}
#endif
#ifdef WN_BWD_STAMPS
extern "C" __attribute__((visibility("default"))) int wn_debug_bwd_stamps(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wn::g_bwd_stamps), sizeof(unsigned long long) * 1024 * 8);
}
#endif
