// bf16-storage path (BASELINE config 5: 128 residual / dilation channels, 512 skip channels, bf16 activations in HBM,
// fp32 accumulation on v_mfma_f32_32x32x16_bf16).  Device-side building blocks shared by the kernels of w16_*.hip.
//
// LDS tile format used by every kernel here: a tile is R rows of 256 bytes (128 bf16 channels of one time column),
// 16-byte chunk c (0..15) of row r stored at chunk position c ^ key(r), key(r) = ((r & 3) << 2) | ((r >> 2) & 3).
// With that key both access kinds are bank-conflict free (cdna_hip_programming.md T10, image (b)):
//   * row reads  -- ds_read_b128 of lane (j, h): row j, chunk 2s + h = the MFMA operand "8 consecutive k of row j";
//   * transposed reads -- ds_read_b64_tr_b16: four rows x 16 channels delivered channel-per-lane (operands of the
//     weight-gradient GEMMs, whose contraction runs over time = tile rows).
// Tiles are filled by LDS-DMA (global_load_lds, 16 bytes per lane, 4 rows = 1 KB per wave instruction); the swizzle is
// applied to the SOURCE address (the LDS side of an LDS-DMA is lane-linear).
#pragma once
#include "wn_common.hpp"

namespace w16 {

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define W16_DMA16(src, dst) \
    __builtin_amdgcn_global_load_lds((const void*)(src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
// the same request with the sc1 bit (aux = 16): fetched from memory, never from a stale copy in this CU's L1 or this XCD's
// L2 -- for rows another workgroup wrote EARLIER IN THE SAME LAUNCH with st16_wt (k16_bwd_multi)
#define W16_DMA16_SC1(src, dst) \
    __builtin_amdgcn_global_load_lds((const void*)(src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 16)

// 16-byte store of streamed output (activations, gradients: written once, far more per launch than the 32 MB of L2):
// write-through ("sc1").  A plain store leaves the line dirty in this XCD's L2 and the launch ends with the write-back of
// whatever is still there; written through, the bytes leave while the kernel computes.  Measured on config 5 (same box,
// alternating libraries): forward 1.45 -> 1.23 ms for its 40 launches with the two stores of k16_fwd alone, and the
// launches that follow a forward got faster too.  W16_ST_MODE: 0 plain, 1 sc1 (default), 3 nt.
#ifndef W16_ST_MODE
#define W16_ST_MODE 1
#endif
__device__ __forceinline__ void st16_wt(void* dst, u32x4 v) {
#if W16_ST_MODE == 1
    // s_nop 1: a store of more than 8 bytes reads its data registers late -- a VALU instruction that overwrites them needs two
    // wait states after it (gfx940+).  hipcc's hazard recogniser inserts them behind its own stores, not behind inline asm:
    // without the s_nop the ragged-tile path of k16_fwd computed the next address into v[0:1] right behind a store of v[0:3].
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
#elif W16_ST_MODE == 3
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
#else
    *reinterpret_cast<u32x4*>(dst) = v;
#endif
}

// a pointer the program knows to be wave-uniform, as the SGPR pair the `"s"` operands below need: for pointers the compiler
// cannot prove uniform (picked by a loop variable).  Use it where the pointer is MADE, not next to the request: a vector-memory
// instruction must not read an SGPR within five wait states of the v_readfirstlane that wrote it, and hipcc does not look
// inside inline asm (tests/test_isa_cpu.py checks the library's disassembly for exactly that)
template <class P>
__device__ __forceinline__ P* uniform_ptr(P* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (P*)(((unsigned long long)hi << 32) | lo);
}

// The same store as `uniform base (SGPR pair) + 32-bit lane offset`: no 64-bit address arithmetic in vector registers
__device__ __forceinline__ void st16_wt_s(const void* base, unsigned off, u32x4 v) {
#if W16_ST_MODE == 1
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
#else
    st16_wt(const_cast<char*>(reinterpret_cast<const char*>(base)) + off, v);
#endif
}
// LDS-DMA of 16 bytes per lane from `uniform base + 32-bit lane offset` to the LDS byte address `lds_addr` (uniform; lane L's
// bytes land at lds_addr + 16 L).  M0 carries the LDS address; the s_nop covers the M0 write -> LDS-DMA hazard.
// (M0 is declared clobbered: the kernels mix this form with the builtin, whose M0 initialisation the compiler emits and may
// otherwise hoist or merge across an asm block it believes leaves M0 alone.)
__device__ __forceinline__ void dma16_s(const void* base, unsigned off, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds_addr) : "memory", "m0");
}
template <bool SC1>
__device__ __forceinline__ void dma16_sx(const void* base, unsigned off, unsigned lds_addr) {
    if constexpr (SC1)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" ::"v"(off), "s"(base), "s"(lds_addr) : "memory", "m0");
    else
        dma16_s(base, off, lds_addr);
}
// one dword per lane into LDS (lane L's word at lds_addr + 4 L), sc1: words whose VALUE is wanted later, requested without
// giving the compiler a register to wait for -- the kernel's own counted s_waitcnt covers the request
__device__ __forceinline__ void dma4_sc1(const void* src, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off sc1" ::"v"(src), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p;
}

__device__ __forceinline__ int key(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
// byte offset of 16-byte chunk c of row r inside a tile
__device__ __forceinline__ int toff(int r, int c) { return (r << 8) + ((c ^ key(r)) << 4); }

__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void barrier() {      // LDS-only barrier: never drains vmcnt (LDS-DMA stays in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Fill `npieces` 4-row pieces (first piece p0, stride pstride between this wave's pieces) of a tile from global rows.
// rowptr(r) must return the (clamped, always valid) global address of channel 0 of tile row r.
// ASM (k16_bwd_multi): the request as inline asm with a per-lane 64-bit address.  With the builtin hipcc knows that LDS is
// written asynchronously and puts s_waitcnt vmcnt(0) in front of every later LDS access that "may alias" -- inline asm with a
// memory clobber included --, which also waits for every store issued since; the asm form leaves the kernel's own counted
// waits as the only ones (every read of the destination must then sit behind one).
template <bool SC1 = false, bool ASM = false, class RowPtr>
__device__ __forceinline__ void dma_pieces(char* tile, int lane, int p0, int pstride, int npieces, RowPtr rowptr) {
#pragma unroll
    for (int i = 0; i < npieces; ++i) {
        const int p = p0 + i * pstride;
        const int r = 4 * p + (lane >> 4);
        const int c = (lane & 15) ^ key(r);
        if constexpr (ASM) {
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_addr_of(tile + p * 1024));
            const void* src = rowptr(r) + c * 8;
            if constexpr (SC1)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" ::"v"(src), "s"(m0v) : "memory", "m0");
            else
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
        } else {
            if constexpr (SC1) W16_DMA16_SC1(rowptr(r) + c * 8, tile + p * 1024);
            else W16_DMA16(rowptr(r) + c * 8, tile + p * 1024);
        }
    }
}

// MFMA B operand (or A operand: same map) of lane (j, h) for k-step s of a 128-channel row tile: row j, chunk 2 s + h
// A 16-byte LDS read through an explicit address-space-3 pointer.  (Note on waits: with an LDS-DMA anywhere in a loop,
// hipcc's wait-count pass emits `s_waitcnt lgkmcnt(0)` for every LDS wait of that loop -- it treats the DMA as a flat
// access that may return out of order with LDS reads -- so a prefetched fragment set in flight is waited for together
// with the one being consumed; the pointer's address space makes no difference to that.)
__device__ __forceinline__ bf16x8 lds_read16(const char* p) {
    return *(const __attribute__((address_space(3))) bf16x8*)(p);
}
__device__ __forceinline__ bf16x8 frag_row(const char* tile, int row, int s, int h) {
    return lds_read16(tile + toff(row, 2 * s + h));
}

// Transposed operand: lane (r = lane & 31, hh = lane >> 5) receives channel c0 + r of the 8 consecutive tile rows
// t0 + 8 hh .. + 7 (= the 8 k values 8 hh + j of a 16-row k-step starting at row t0).  Two ds_read_b64_tr_b16: each
// 16-lane group reads a block of 4 rows x 16 channels; lane 4 q + p of a group supplies the address of row q,
// channels 4 p .. 4 p + 3 of the block (8 bytes).
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int t0, int c0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int cb = c0 + 16 * (g & 1);                 // first channel of this group's block
    const int rb = t0 + 8 * (g >> 1);                 // first row of the group's two blocks
    const int chunk = (cb >> 3) + (p >> 1);
    const int sub = 8 * (p & 1);
    const int r0 = rb + q, r1 = rb + 4 + q;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + toff(r0, chunk) + sub));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + toff(r1, chunk) + sub));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
}

// relu on a packed bf16 operand (sign bit set -> 0)
__device__ __forceinline__ bf16x8 relu8(bf16x8 v) {
    u32x4 u = __builtin_bit_cast(u32x4, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned neg = (u[i] >> 15) & 0x00010001u;
        u[i] &= ~(neg * 0xFFFFu);
    }
    return __builtin_bit_cast(bf16x8, u);
}

__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
    bf16x4 v;
    v[0] = (bf16)a; v[1] = (bf16)b; v[2] = (bf16)c; v[3] = (bf16)d;
    return v;
}

// accumulator register r of lane half h  <->  row (r & 3) + 8 (r >> 2) + 4 h of the 32-row MFMA tile
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// XCD-aware persistent tile range of a workgroup (workgroups are dealt round-robin to the 8 XCDs: XCD k = blockIdx % 8
// gets the k-th contiguous eighth of the tiles, so neighbouring tiles -- whose halos overlap -- share an L2)
__device__ __forceinline__ void tile_range(int ntiles, int& first, int& stride, int& last) {
    if ((gridDim.x & 7) == 0) {
        const int per_xcd = (ntiles + 7) >> 3;
        const int xcd = blockIdx.x & 7;
        stride = gridDim.x >> 3;
        first = xcd * per_xcd + (blockIdx.x >> 3);
        last = (xcd + 1) * per_xcd < ntiles ? (xcd + 1) * per_xcd : ntiles;
    } else {
        stride = gridDim.x;
        first = blockIdx.x;
        last = ntiles;
    }
}

// ---- sizes of the per-layer weight images (bf16 elements), see k16_pack_layers -------------------------------------
static constexpr int kConvA = 4 * 2 * 16 * 64 * 8;  // forward conv: 4 waves x (filter, gate) x 16 k-steps x 64 lanes x 8
static constexpr int kProjA = 4 * 8 * 64 * 8;       // residual projection: 4 m-tiles x 8 k-steps
static constexpr int kConvA8 = 8 * 16 * 64 * 8;     // gate backward conv: 8 waves x 16 k-steps, tile rows [16 filter; 16 gate]
static constexpr int kDzA8 = 8 * 8 * 64 * 8;        // Wp^T for dz: 8 waves x 8 k-steps (rows 16..31 of each tile are 0)
static constexpr int kDxA = 4 * 2 * 16 * 64 * 8;    // [Wf1;Wg1 | Wf0;Wg0]^T: 4 m-tiles x 2 contraction halves x 16 k-steps
static constexpr int kOffConvA8 = kConvA + kProjA;
static constexpr int kOffDzA8 = kOffConvA8 + kConvA8;
static constexpr int kOffDxA = kOffDzA8 + kDzA8;
static constexpr int kLayerImg = kOffDxA + kDxA;

}  // namespace w16
