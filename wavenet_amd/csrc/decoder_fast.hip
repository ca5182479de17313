// Specialised persistent decode kernel for the shape of BASELINE.json config 4:
//   Q = 256, one causal layer (fw 2), Cr = Cd = 32, Cs = 256, residual fw 2, one head conv 256 -> 256,
//   no biases except the head's (the reference's defaults, wavenet.py:108,116-117,138).
// Same algorithm and state as k_decode in decoder.hip (FasterWaveNet._forward_one_step restated with
// rings instead of rolled windows); what changes is how ONE workgroup is used so that a step is not
// a chain of L2 round trips and wide barriers:
//   * 256 threads = one wave per SIMD;
//   * the 64 KB embedding table of the causal layer sits in LDS for the whole launch;
//   * the head's 256x256 weights live in registers for the whole launch (256 per thread: one wave
//     per SIMD owns the whole 512-entry register file);
//   * a layer's 13,312 weights are 52 per thread, fetched ONE LAYER AHEAD (the loads of layer l+1 are
//     in flight while layer l computes; the in-loop barriers wait for LDS only, never for vmcnt), as
//     is the ring column x[n-d] of the next layer;
//   * the waves are specialised.  Wave 0 alone runs the dependent chain of the 40 layers WITHOUT any barrier and
//     without an LDS round trip on the critical path: lane r owns gate row r (filter rows in lanes 0-31, gate rows in
//     32-63) with its 64 weights in registers, the x[n] it multiplies them with reaches it as 32 scalar broadcasts
//     (v_readlane -> SGPR operand of v_fmac), tanh and sigmoid are ONE exp + rcp sequence over the two half waves,
//     v_permlane32_swap brings filter and gate together, the residual projection is 32 more broadcast-FMAs per lane;
//     the x[n-d] half of the gate (known since the start of the step) comes from LDS broadcast reads off the critical
//     path.  (The previous form -- gate rows over 2 lanes, projection over 4, DPP reductions, z and x through LDS with
//     two workgroup barriers per layer -- took 1,650 cycles per layer.)
//     Wave 1 runs AHEAD of the chain: the x[n-d] half of every gate row depends only on the rings, which are known
//     when the step starts, so it computes all 40 layers' partial sums into LDS (its own copy of those weights, its
//     own layer counter) and the chain wave adds one float per row instead of 16 packed FMAs and 8 broadcast reads.
//     Waves 2-3 run the skip projection (61 % of the MACs) behind the chain from the table of z columns in LDS, two
//     rows per thread accumulated in registers over all 40 layers; they follow the chain through a layer counter in
//     LDS (LDS operations of a wave retire in order, so the counter store after the z store publishes it).
#include "wn_kernels.hpp"
#include "decoder_types.hpp"

namespace wn {

static constexpr int kFT = 256;      // waves 0-1: the dependent chain; waves 2-3: skip rows, one layer behind
static constexpr int kMaxFastLayers = 128;
static constexpr int kGateFloats = 4096, kProjFloats = 2048;
static constexpr int kLayerFloats = kGateFloats + kProjFloats + 8192;     // gate | residual projection (both half waves) | skip projection

__device__ __forceinline__ float dpp_f(float v, int ctrl_sel) {
    int r;
    switch (ctrl_sel) {
        case 101: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true); break;  // quad xor 1
        default: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true); break;   // quad xor 2
    }
    return __int_as_float(r);
}
__device__ __forceinline__ float quad_allsum(float v) {
    v += dpp_f(v, 101);
    v += dpp_f(v, 102);
    return v;
}
// DPP moves: lane i <- lane i-n inside its 16-lane row (row_shr), or the previous row's last lane (row_bcast)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ float wave_allmax(float v) {      // full-wave max, broadcast through an SGPR
    const int ninf = __float_as_int(-INFINITY);
    v = fmaxf(v, __int_as_float(dpp_i<0x111, 0xf>(ninf, __float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<0x112, 0xf>(ninf, __float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<0x114, 0xf>(ninf, __float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<0x118, 0xf>(ninf, __float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<0x142, 0xa>(ninf, __float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<0x143, 0xc>(ninf, __float_as_int(v))));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_allsum(float v) {
    v += __int_as_float(dpp_i<0x111, 0xf>(0, __float_as_int(v)));
    v += __int_as_float(dpp_i<0x112, 0xf>(0, __float_as_int(v)));
    v += __int_as_float(dpp_i<0x114, 0xf>(0, __float_as_int(v)));
    v += __int_as_float(dpp_i<0x118, 0xf>(0, __float_as_int(v)));
    v += __int_as_float(dpp_i<0x142, 0xa>(0, __float_as_int(v)));
    v += __int_as_float(dpp_i<0x143, 0xc>(0, __float_as_int(v)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// inclusive prefix sum over the 64 lanes of a wave, float64 (both halves travel through the same DPP move)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i<CTRL, ROW_MASK>(0, (int)(b & 0xffffffffll));
    const int hi = dpp_i<CTRL, ROW_MASK>(0, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_scan_f64(double v) {
    v += dpp_d<0x111, 0xf>(v);
    v += dpp_d<0x112, 0xf>(v);
    v += dpp_d<0x114, 0xf>(v);
    v += dpp_d<0x118, 0xf>(v);
    v += dpp_d<0x142, 0xa>(v);       // rows 1,3 += last lane of rows 0,2
    v += dpp_d<0x143, 0xc>(v);       // rows 2,3 += lane 31
    return v;
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// ---- packing: every per-lane weight group is a float4 at [plane][lane] (coalesced 16-byte loads) --
// chain wave, lane r = 0..63:  gate row r (0-31 filter channel r, 32-63 gate channel r-32), k = 4j + e over
//                              [x_old(32) | x_cur(32)]                                           (16 planes of 64 lanes)
//                              Wp row o = r % 32, k = 4j + e (both half waves hold the same rows)  (8 planes)
// skip threads t = 0..127:     rows cs = 2t (planes 0-7) and 2t+1 (planes 8-15), k = 4(j%8) + e
__global__ void k_pack_fast_layer(const float* __restrict__ Wf, const float* __restrict__ Wg,
                                  const float* __restrict__ Wp, const float* __restrict__ Ws,
                                  float* __restrict__ dst) {
    const int t = threadIdx.x;     // 0..127
    if (t < 64) {
        const int c = t & 31;
        const float* W = (t >> 5) ? Wg : Wf;
        for (int j = 0; j < 16; ++j)
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * j + e, tap = k >> 5, ch = k & 31;               // [x_old(32) | x_cur(32)]
                dst[(j * 64 + t) * 4 + e] = W[(c * 32 + ch) * 2 + tap];
            }
        for (int j = 0; j < 8; ++j)
            for (int e = 0; e < 4; ++e) dst[kGateFloats + (j * 64 + t) * 4 + e] = Wp[c * 32 + 4 * j + e];
    }
    for (int j = 0; j < 16; ++j)
        for (int e = 0; e < 4; ++e)
            dst[kGateFloats + kProjFloats + (j * 128 + t) * 4 + e] = Ws[(2 * t + (j >> 3)) * 32 + 4 * (j & 7) + e];
}
// head: row q = tid, k = 4j+e  ->  Ph[j][tid][4], j = 0..63
__global__ void k_pack_fast_head(const float* __restrict__ Wh, float* __restrict__ dst) {
    const int tid = threadIdx.x;
    for (int j = 0; j < 64; ++j)
        for (int e = 0; e < 4; ++e) dst[(j * 256 + tid) * 4 + e] = Wh[tid * 256 + 4 * j + e];
}

typedef float f4v __attribute__((ext_vector_type(4)));
struct ChainW { f4v g[8]; f4v p[8]; };            // gate rows, x[n] half | projection rows
struct OldW { float4 g[8]; };                      // gate rows, x[n-d] half (wave 1)
struct SkipW { float4 s[16]; };

// The chain wave's 16 loads of a layer: uniform layer base in an SGPR pair + one of four loop-invariant 32-bit lane
// offsets + an immediate (the immediate reaches 4 KB, the 16 planes of a layer span 16 KB).  Written as asm: from the C
// form hipcc built a 64-bit address per group in vector registers (eleven VALU / carry instructions per layer on the
// chain's critical path).  The loads are invisible to the compiler's counters: chain_run() waits for them itself
// (s_waitcnt vmcnt(0) at the top of the layer that uses them) and re-defines the registers behind that wait.
struct ChainOff { unsigned o[4]; };                // byte offsets 16 lane + 8192 + 4096 k
__device__ __forceinline__ ChainOff chain_offsets(int lane) {
    ChainOff c;
#pragma unroll
    for (int k = 0; k < 4; ++k) c.o[k] = 16u * lane + 8192u + 4096u * k;
    return c;
}
#define WN_CHAIN_LD(dst, off, imm) \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #imm : "=v"(dst) : "v"(off), "s"(b))
__device__ __forceinline__ void load_chain(ChainW& w, const float* __restrict__ P, int l, const ChainOff& c) {
    const float* b = P + (long long)l * kLayerFloats;
    WN_CHAIN_LD(w.g[0], c.o[0], 0); WN_CHAIN_LD(w.g[1], c.o[0], 1024); WN_CHAIN_LD(w.g[2], c.o[0], 2048); WN_CHAIN_LD(w.g[3], c.o[0], 3072);
    WN_CHAIN_LD(w.g[4], c.o[1], 0); WN_CHAIN_LD(w.g[5], c.o[1], 1024); WN_CHAIN_LD(w.g[6], c.o[1], 2048); WN_CHAIN_LD(w.g[7], c.o[1], 3072);
    WN_CHAIN_LD(w.p[0], c.o[2], 0); WN_CHAIN_LD(w.p[1], c.o[2], 1024); WN_CHAIN_LD(w.p[2], c.o[2], 2048); WN_CHAIN_LD(w.p[3], c.o[2], 3072);
    WN_CHAIN_LD(w.p[4], c.o[3], 0); WN_CHAIN_LD(w.p[5], c.o[3], 1024); WN_CHAIN_LD(w.p[6], c.o[3], 2048); WN_CHAIN_LD(w.p[7], c.o[3], 3072);
}
#undef WN_CHAIN_LD
// ... and the wait: everything this wave has in flight has landed, and no use of `w` can be scheduled above this point
__device__ __forceinline__ void chain_landed(ChainW& w) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(w.g[j]), "+v"(w.p[j]));
}
__device__ __forceinline__ void load_old(OldW& w, const float* __restrict__ P, int l, unsigned l4) {
    const float* b = P + (long long)l * kLayerFloats;
#pragma unroll
    for (int j = 0; j < 8; ++j) w.g[j] = *reinterpret_cast<const float4*>(b + j * 256 + l4);
}
__device__ __forceinline__ void load_skip(SkipW& w, const float* __restrict__ P, int l, unsigned t4) {
    const float* b = P + (long long)l * kLayerFloats + kGateFloats + kProjFloats;
#pragma unroll
    for (int j = 0; j < 16; ++j) w.s[j] = *reinterpret_cast<const float4*>(b + j * 512 + t4);
}

// LDS-only barrier: waits for this wave's LDS traffic, not for its outstanding global loads, so the
// one-layer-ahead weight prefetch stays in flight across it (__syncthreads() would drain vmcnt).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct FastLds { const float* xold; float* xcur; float* zall; float* aold; int* ready; int* ready_old; };

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float f4c(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
__device__ __forceinline__ float bcast(float v, int k) {       // lane k's value as a scalar (SGPR) operand
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k));
}

// The chain wave: one residual layer of one step.  xc = x_l[n] (channel lane % 32, valid in every lane); returns
// x_{l+1}[n].  Writes z to zall[l] and x_{l+1}[n] to xcur[l+1] (for the ring update), then publishes the layer.
// Hand-off protocol between the waves of the ONE workgroup of this kernel (all on one CU, all through LDS):
//   producer: data ds_writes ... counter ds_write, in that order in ITS instruction stream;
//   consumer: counter ds_read (spin) ... data ds_reads, in that order in its stream.
// What makes this sufficient is a property of the hardware, not of the HSA memory model: a CU has ONE LDS pipeline
// that executes the DS instructions of all its waves in the order they were issued, and a wave's DS instructions issue
// in program order -- so the data writes are in the array before the counter write is, and a read issued after the
// counter read that saw the new value is served after both.  The fences below are therefore wavefront-scope: they only
// stop the COMPILER from moving DS accesses across the counter access (a workgroup-scope release would add an
// s_waitcnt lgkmcnt(0) on the 40-layer critical path for an ordering the pipeline already gives).  gfx950-specific by
// construction, like everything else in this file; the 256-step golden trace of config 4
// (tests/test_gpu_baseline_configs.py) and the 16,000-step bench run would show a reordering as a token mismatch.
// The spin is bounded: a lost hand-off traps (kernel abort, reported by the next HIP call) instead of hanging the GPU.
__device__ __forceinline__ void wait_count(const int* c, int v) {               // *c >= v, then an acquire at compile level
    int spins = 0;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 24)) __builtin_trap();
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// wave 1: the x[n-d] half of gate row `lane` of layer l -> aold[l][lane]
__device__ __forceinline__ void old_layer(const FastLds& S, const OldW& w, int l, int lane) {
    const float* xo = S.xold + l * 32;                 // same address in every lane: broadcast reads
    f2 A0 = {0.f, 0.f}, A1 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 x4 = *reinterpret_cast<const float4*>(xo + 4 * j);
        A0 = __builtin_elementwise_fma(f2{w.g[j].x, w.g[j].y}, f2{x4.x, x4.y}, A0);
        A1 = __builtin_elementwise_fma(f2{w.g[j].z, w.g[j].w}, f2{x4.z, x4.w}, A1);
    }
    S.aold[l * 64 + lane] = (A0.x + A0.y) + (A1.x + A1.y);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (lane == 0) __hip_atomic_store(S.ready_old, l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The chain wave: one residual layer of one step.  xc = x_l[n] (channel lane % 32, valid in every lane); a_old = the
// x[n-d] half of this lane's gate row (from wave 1).  Returns x_{l+1}[n]; a_old is replaced by the next layer's.
// The next layer's a_old is requested at the TOP of this layer together with wave 1's counter (counter read first, data read
// second: the LDS pipeline serves them in that order, so a counter value >= l + 2 proves the data read saw wave 1's
// write) and looked at only at the bottom: two LDS round trips (the spin's read, then the data's) left the 40-layer
// critical path.  Wave 1 is normally many layers ahead; when it is not, the bounded spin and a second read follow.
// zx != NULL (three-workgroup form): z also goes to the skip workgroup, every lane's value in ONE 8-byte store together with
// the step's sequence number (flag-in-data: the reader spins on the entry itself, so no ordering between a data store and
// a flag store is needed and the chain never waits for a store).
#ifndef XSLEEP
#define XSLEEP 1
#endif
typedef unsigned long long u64;
__device__ __forceinline__ void xput(u64* p, float v, unsigned seq) {
    __hip_atomic_store(p, ((u64)seq << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Cross-workgroup waits never trap and never hang: an entry that does not arrive within ~1-2 s is given up -- the waiting
// thread marks the launch void in the error entry (x_err: read by wn_decoder_status) and from then on waits for nothing
// (`dead`), so every workgroup still runs to the end of the launch and the HIP context survives; the tokens of such a run
// are garbage and the host is told so.  (It happens when the nine workgroups are not all resident: a device with fewer
// free CUs than workgroups, other work holding them for seconds.)
__device__ __forceinline__ float xget(const u64* p, unsigned seq, u64* err, bool& dead) {
    int spins = 0;
    u64 w = 0;
    if (dead) return 0.f;
    while ((unsigned)((w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != seq) {
        __builtin_amdgcn_s_sleep(XSLEEP);
        ++spins;
        // after a while also look at the error entry: when another workgroup has given up, this entry may never come
        if (spins > (1 << 21) || ((spins & 0xfff) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            __hip_atomic_store(err, (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dead = true;
            return 0.f;
        }
    }
    return __uint_as_float((unsigned)w);
}
__device__ __forceinline__ float chain_layer(const FastLds& S, const ChainW& w, float& a_old, int l, const bool more,
                                             int lane, float xc) {
    int c_next = 0;
    float a_next = 0.f;
    if (more) {
        c_next = __hip_atomic_load(S.ready_old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        a_next = S.aold[(l + 1) * 64 + lane];
    }
    // gate row `lane`: 32 scalar broadcasts of x[n], all read before the first use (a v_readlane result needs wait
    // states before a VALU may consume it: batched, the FMAs need no s_nop); packed FMAs (v_pk_fma_f32: two MACs per
    // instruction, the scalar pair as an SGPR operand), two accumulator pairs
    float sx[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) sx[k] = bcast(xc, k);
    f2 A0 = {a_old, 0.f}, A1 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        A0 = __builtin_elementwise_fma(f2{sx[4 * j], sx[4 * j + 1]}, f2{w.g[j].x, w.g[j].y}, A0);
        A1 = __builtin_elementwise_fma(f2{sx[4 * j + 2], sx[4 * j + 3]}, f2{w.g[j].z, w.g[j].w}, A1);
    }
    const float acc = (A0.x + A0.y) + (A1.x + A1.y);
    // tanh (lanes 0-31) and sigmoid (lanes 32-63) as one sequence: 1 - 2/(1 + e^{2a})  |  1/(1 + e^{-g})
    const bool lo = lane < 32;
    const float r = __builtin_amdgcn_rcpf(1.0f + __expf(lo ? 2.0f * acc : -acc));
    const float act = lo ? 1.0f - 2.0f * r : r;
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(act), __float_as_uint(act), false, false);
    const float z = __uint_as_float(sw[0]) * __uint_as_float(sw[1]);          // z[lane % 32] in every lane
    S.zall[l * 32 + (lane & 31)] = z;                  // both half waves hold the same value: no exec-mask branch
    // residual projection row lane % 32
    float sz[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) sz[k] = bcast(z, k);
    f2 P0 = {0.f, 0.f}, P1 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        P0 = __builtin_elementwise_fma(f2{sz[4 * j], sz[4 * j + 1]}, f2{w.p[j].x, w.p[j].y}, P0);
        P1 = __builtin_elementwise_fma(f2{sz[4 * j + 2], sz[4 * j + 3]}, f2{w.p[j].z, w.p[j].w}, P1);
    }
    const float xn = ((P0.x + P0.y) + (P1.x + P1.y)) + xc;
    S.xcur[(l + 1) * 32 + (lane & 31)] = xn;
    // publish: LDS operations of a wave retire in order, so the counter store only has to FOLLOW the z store in the
    // instruction stream (a compiler-level fence; no s_waitcnt on the chain's critical path)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (lane == 0) __hip_atomic_store(S.ready, l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (more) {
        if (__builtin_expect(__builtin_amdgcn_readfirstlane(c_next) < l + 2, 0)) {       // rare: wave 1 has not been here yet
            wait_count(S.ready_old, l + 2);
            a_next = S.aold[(l + 1) * 64 + lane];
        }
        a_old = a_next;
    }
    return xn;
}
__device__ __forceinline__ void wait_layer(const FastLds& S, int l) { wait_count(S.ready, l); }   // z of layers < l is in zall

// skip waves: rows 2t and 2t+1 of Ws_l z_l, accumulated over the layers
__device__ __forceinline__ void skip_layer(const FastLds& S, const SkipW& w, int l, float& s0, float& s1) {
    const float* z = S.zall + l * 32;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 z4 = *reinterpret_cast<const float4*>(z + 4 * j);
        s0 += dot4(w.s[j], z4);
        s1 += dot4(w.s[8 + j], z4);
    }
}

#ifdef WN_DECODE_STAMPS
#define STAMP(k) do { if (tid == 0) stamps[k] = clock64(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

static constexpr int kUnroll = 10;      // layers per trip of the layer loop (one block of the 4 x 10 stack)

// The chain wave's walk over the layers of one step.  Layer l's weights were requested a whole layer ago (an L2 hit lands in
// a fifth of that): the wait in front of the next request costs nothing.
// WHOLE: the stack is a whole number of kUnroll-layer trips (the 4 x 10 stack is): "is there a layer l?" and "is there a
// layer l + 1?" are then compile-time facts for nine layers of ten -- each was three scalar branches on the chain's
// critical path.
template <bool WHOLE>
__device__ __forceinline__ float chain_run(const FastLds& S, const float* __restrict__ P, int nlayers, int lane, float xc,
                                           float a_old, ChainW (&w)[2], const ChainOff& co) {   // w[0]: layer 0's weights, requested by the caller
    for (int l0 = 0; l0 < nlayers; l0 += kUnroll) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int l = l0 + u;
            if (WHOLE || l < nlayers) {
                const bool more = WHOLE && u + 1 < kUnroll ? true : l + 1 < nlayers;
                chain_landed(w[u & 1]);                      // requested a whole layer ago
                if (more) load_chain(w[(u + 1) & 1], P, l + 1, co);   // in flight during layer l
                xc = chain_layer(S, w[u & 1], a_old, l, more, lane, xc);
            }
        }
    }
    return xc;
}

__global__ __launch_bounds__(kFT, 1) void k_decode_fast(
    const float* __restrict__ P, const float* __restrict__ Ph, const float* __restrict__ hbias,
    const float* __restrict__ E, const DecLayer* __restrict__ layers, int nlayers, float* __restrict__ arena,
    int* __restrict__ tok_ring, long long n0, int nsteps, int first_token, const double* __restrict__ uniforms,
    int32_t* __restrict__ out_tokens, float* __restrict__ prob_out, int prob_stride, int apply_softmax,
    int do_sample, int head_act) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Elds = sm;                                   // [256][2][32] embedding table of the causal layer
    float* xold = Elds + 256 * 2 * 32;                  // [L][32]   x_l[n-d] of every layer, fetched at step start
    float* xcur = xold + kMaxFastLayers * 32;           // [L+1][32] x_l[n]: input of layer l (output of l-1)
    float* zall = xcur + (kMaxFastLayers + 1) * 32;     // [L][32]   gate output of every layer of this step
    float* hvec = zall + kMaxFastLayers * 32;           // [256]
    float* lg = hvec + 256;                             // [256] logits / probabilities
    float* red = lg + 256;                              // [16]
    double* cdf = reinterpret_cast<double*>(red + 16);  // [256]
    int* s_tok = reinterpret_cast<int*>(cdf + 256);     // [4]: current token, previous token
    int* ringt = s_tok + 4;                             // [L] ring offset per layer
    int* dmask = ringt + kMaxFastLayers;                // [L] d - 1 (d is a power of two: fw = 2)
    int* ready = dmask + kMaxFastLayers;                // [2] layers of this step whose z is in zall | whose aold is there
    float* aold = reinterpret_cast<float*>(ready + 4);  // [L][64] x[n-d] half of every gate row, from wave 1
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    for (int i = tid; i < 256 * 2 * 32 / 4; i += kFT)
        reinterpret_cast<float4*>(Elds)[i] = reinterpret_cast<const float4*>(E)[i];
    for (int i = tid; i < nlayers; i += kFT) { ringt[i] = layers[i].ring; dmask[i] = layers[i].d - 1; }
    const float hb = hbias ? hbias[tid] : 0.f;
    if (tid == 0) { s_tok[0] = first_token; s_tok[1] = tok_ring[0]; }     // fwc = 2: ring depth 1
    __syncthreads();
    FastLds S{xold, xcur, zall, aold, ready, ready + 1};
    const unsigned t4 = 4u * (tid & 127);
#ifdef WN_DECODE_STAMPS
    long long stamps[8];
#endif

    for (int it = 0; it < nsteps; ++it) {
        const unsigned n = (unsigned)(n0 + it);
        const int token = s_tok[0], tprev = s_tok[1];
        const double u_draw = do_sample ? uniforms[it] : 0.0;      // fetched here, used after the network
        STAMP(0);
        // every layer's x[n-d] was written at least one step ago: fetch them all now, off the layer chain
        for (int i = tid; i < nlayers * 32; i += kFT) {
            const int l = i >> 5;
            xold[i] = arena[ringt[l] + (long long)(n & (unsigned)dmask[l]) * 32 + (i & 31)];
        }
        if (tid < 32) xcur[tid] = Elds[(tprev * 2 + 0) * 32 + tid] + Elds[(token * 2 + 1) * 32 + tid];
        if (tid == 0) {
            __hip_atomic_store(ready, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(ready + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (wv == 0) {
            ChainW w[2];
            const ChainOff co = chain_offsets(lane);
            load_chain(w[0], P, 0, co);
            __syncthreads();
            STAMP(1);
            const float xc0 = xcur[lane & 31];
            wait_count(S.ready_old, 1);
            const float a0 = aold[lane];
            if (nlayers % kUnroll == 0) chain_run<true>(S, P, nlayers, lane, xc0, a0, w, co);
            else chain_run<false>(S, P, nlayers, lane, xc0, a0, w, co);
        } else if (wv == 1) {
            OldW wo[2];
            load_old(wo[0], P, 0, 4u * lane);
            __syncthreads();
            for (int l0 = 0; l0 < nlayers; l0 += kUnroll) {
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const int l = l0 + u;
                    if (l < nlayers) {
                        __builtin_amdgcn_s_waitcnt(0x0F70);          // as in the chain wave: exact counters
                        if (l + 1 < nlayers) load_old(wo[(u + 1) & 1], P, l + 1, 4u * lane);
                        old_layer(S, wo[u & 1], l, lane);
                    }
                }
            }
        } else {
            float skip0 = 0.f, skip1 = 0.f;
            SkipW w[2];
            load_skip(w[0], P, 0, t4);
            __syncthreads();
            for (int l0 = 0; l0 < nlayers; l0 += kUnroll) {
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const int l = l0 + u;
                    if (l < nlayers && l > 0) {
                        // behind the chain: fetch layer l's rows, use layer l-1's (loaded a trip ago) once its z is there
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                        load_skip(w[u & 1], P, l, t4);
                        wait_layer(S, l);
                        skip_layer(S, w[(u + 1) & 1], l - 1, skip0, skip1);
                    }
                }
            }
            wait_layer(S, nlayers);
            if ((nlayers - 1) & 1) skip_layer(S, w[1], nlayers - 1, skip0, skip1);
            else skip_layer(S, w[0], nlayers - 1, skip0, skip1);
            hvec[2 * (tid - 128)] = act_apply(skip0, head_act);
            hvec[2 * (tid - 128) + 1] = act_apply(skip1, head_act);
        }
        STAMP(2);
        // ---- head on the newest column: thread q owns logit q; its 256 weights stream in from L2 -------
        float v;
        {
            const float* Phs = Ph;
            asm volatile("" : "+s"(Phs));                 // keep the loads here (no hoisting above the layer loop)
            const unsigned q4 = 4u * tid;
            float4 wa[16], wb[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) wa[j] = *reinterpret_cast<const float4*>(Phs + j * 1024 + q4);
#pragma unroll
            for (int j = 0; j < 16; ++j) wb[j] = *reinterpret_cast<const float4*>(Phs + (16 + j) * 1024 + q4);
            lds_barrier();                                // the chain has written every x_cur, the skip waves hvec
                                                          // (LDS only: the weight loads stay in flight)
            // this step's x_cur of every layer becomes the newest ring column
            for (int i = tid; i < nlayers * 32; i += kFT) {
                const int l = i >> 5;
                arena[ringt[l] + (long long)(n & (unsigned)dmask[l]) * 32 + (i & 31)] = xcur[i];
            }
            float p0 = hb, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#define HEAD_ACC(W, J0)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 16; j += 4) {                                                   \
        p0 += dot4(W[j], *reinterpret_cast<const float4*>(hvec + 4 * ((J0) + j)));                        \
        p1 += dot4(W[j + 1], *reinterpret_cast<const float4*>(hvec + 4 * ((J0) + j) + 4));                \
        p2 += dot4(W[j + 2], *reinterpret_cast<const float4*>(hvec + 4 * ((J0) + j) + 8));                \
        p3 += dot4(W[j + 3], *reinterpret_cast<const float4*>(hvec + 4 * ((J0) + j) + 12));               \
    }
            HEAD_ACC(wa, 0)
#pragma unroll
            for (int j = 0; j < 16; ++j) wa[j] = *reinterpret_cast<const float4*>(Phs + (32 + j) * 1024 + q4);
            HEAD_ACC(wb, 16)
#pragma unroll
            for (int j = 0; j < 16; ++j) wb[j] = *reinterpret_cast<const float4*>(Phs + (48 + j) * 1024 + q4);
            HEAD_ACC(wa, 32)
            HEAD_ACC(wb, 48)
#undef HEAD_ACC
            v = (p0 + p1) + (p2 + p3);
        }
        if (apply_softmax) {
            float m = wave_allmax(v);
            if (lane == 0) red[wv] = m;
            lds_barrier();
            m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            const float e = expf(v - m);
            float s = wave_allsum(e);
            if (lane == 0) red[4 + wv] = s;
            lds_barrier();
            s = red[4] + red[5] + red[6] + red[7];
            v = e * (1.f / s);
        }
        lg[tid] = v;
#ifndef WN_DECODE_STAMPS
        if (prob_out) prob_out[(long long)it * prob_stride + tid] = v;
#endif
        lds_barrier();
        STAMP(3);
        if (do_sample) {
            // numpy: cdf = cumsum(float64(p)); cdf /= cdf[-1]; first index with cdf > u.  The running sum is a
            // 256-long dependent chain (9.5 k cycles); a parallel scan associates differently, so its cdf may differ
            // from numpy's by a few ulp (<= 256 * 2^-53).  It is therefore used only when no cdf_i / total lies within
            // 1e-12 of u -- then both orders give the same index, provably -- and the exact chain runs otherwise.
            double c = wave_scan_f64((double)lg[tid]);
            if (lane == 63) cdf[wv] = c;                   // wave totals
            lds_barrier();
            {
                const double w0 = cdf[0], w1 = cdf[1], w2 = cdf[2], w3 = cdf[3];
                c += wv > 0 ? w0 : 0.0;
                c += wv > 1 ? w1 : 0.0;
                c += wv > 2 ? w2 : 0.0;
                const double tot = ((w0 + w1) + w2) + w3;
                const double q = c / tot;
                const bool gt = q > u_draw;
                const bool near = fabs(q - u_draw) < 1e-12;
                const unsigned long long bal = __ballot(gt), nb = __ballot(near);
                if (lane == 0) {
                    reinterpret_cast<int*>(red)[8 + wv] = bal ? wv * 64 + __ffsll((long long)bal) - 1 : 256;
                    reinterpret_cast<int*>(red)[12 + wv] = nb ? 1 : 0;
                }
            }
            lds_barrier();
            const int* rr = reinterpret_cast<const int*>(red);
            const bool ambiguous = (rr[12] | rr[13] | rr[14] | rr[15]) != 0;     // uniform over the workgroup
            if (ambiguous) {
                if (tid == 0) {                    // numpy's float64 running sum, in index order
                    double cs = 0.0;
                    for (int i = 0; i < 256; ++i) { cs += (double)lg[i]; cdf[i] = cs; }
                }
                lds_barrier();
                {
                    const double tot = cdf[255];
                    const bool gt = cdf[tid] / tot > u_draw;
                    const unsigned long long bal = __ballot(gt);
                    lds_barrier();
                    if (lane == 0) reinterpret_cast<int*>(red)[8 + wv] = bal ? wv * 64 + __ffsll((long long)bal) - 1 : 256;
                }
                lds_barrier();
            }
            if (tid == 0) {
                const int* r = reinterpret_cast<const int*>(red) + 8;
                int idx = min(min(r[0], r[1]), min(r[2], r[3]));
                if (idx > 255) idx = 255;
                out_tokens[it] = idx;
                s_tok[1] = token;
                s_tok[0] = idx;
            }
        } else if (tid == 0) {
            s_tok[1] = token;
        }
        __syncthreads();                           // full barrier: orders this step's ring stores before the next step's loads
#ifdef WN_DECODE_STAMPS
        STAMP(4);
        if (tid == 0 && prob_out)
            for (int k = 0; k < 4; ++k) prob_out[(long long)it * prob_stride + k] = (float)(stamps[k + 1] - stamps[k]);
#endif
    }
    if (tid == 0) tok_ring[0] = s_tok[1];          // the token before the next one to be consumed
}

// ---------------------------------------------------------------------------------------------
// The same decode on NINE workgroups (nine CUs).  A step of k_decode_fast streams 2.5 MB of fp32 weights into one CU --
// 69.5 k cycles at the 36 B/clk one CU gets out of L2, which is the step time whatever the four waves do (DESIGN.md,
// round 3) -- and runs head, softmax and sampling (18 k cycles) behind the 40-layer chain.  Here
//   workgroup 0      keeps the chain (wave 0), the x[n-d] halves (wave 1), softmax and the numpy-compatible sampler; its
//                    wave 2 follows the chain through the LDS layer counter and publishes every layer's z to the others
//                    (the chain wave itself must not store: vector-memory operations retire in order, and a write-through
//                    store in its queue held back the next layer's weight loads -- 60 k cycles for the chain instead of 49);
//   workgroups 1..8  own 32 skip rows each and keep their slice of ALL layers' skip weights in REGISTERS for the whole
//                    launch (thread = (row, eighth of the 32 gate channels): one float4 per layer, 40 float4): no weight
//                    streaming, a layer costs four FMAs; they poll eight layers of z per memory round trip;
//                    and, having their 32 activated skip values, add those columns' share of EVERY logit (thread q: 32 more
//                    resident weights of head row q) -- workgroup 0 adds the eight shares: no separate head hop.
// Exchange is through `X` in device memory, flag-in-data: every float travels in an 8-byte entry with the step's
// sequence number (xput / xget above), written once and polled by its reader; entries are zeroed by a kernel before the
// launch.  Two hops follow the chain's last layer (z -> partial logits -> workgroup 0).  Skip rows and logits are summed
// in another order than k_decode_fast's (eight partial sums per row added pairwise; eight shares per logit): probabilities agree to rounding
// (~1e-7), not bit for bit; the launch itself is deterministic.  At most 40 layers (the weights must fit the registers).
// ---------------------------------------------------------------------------------------------
static constexpr int kXZ = 0;                                   // X layout (u64 entries): z [L][64] | partial logits [8][256]
static constexpr int kD10MaxL = 40;
static constexpr int kD10Skip = 8;                              // skip workgroups
__device__ __host__ __forceinline__ int x_pl(int nlayers) { return nlayers * 64; }
__device__ __host__ __forceinline__ int x_err(int nlayers) { return nlayers * 64 + 8 * 256 + 8; }   // != 0: a wait gave up, the run is void

// `wg`: this workgroup's role in its utterance's group of nine (0 = chain, 1..8 = skip rows): blockIdx.x for one utterance,
// blockIdx.x % 9 in the batched launch (k_decode_fast3_batch)
__device__ __forceinline__ void decode_fast3_body(
    const int wg, const float* __restrict__ P, const float* __restrict__ Ph, const float* __restrict__ hbias,
    const float* __restrict__ E, const DecLayer* __restrict__ layers, int nlayers, float* __restrict__ arena,
    int* __restrict__ tok_ring, long long n0, int nsteps, int first_token, const double* __restrict__ uniforms,
    int32_t* __restrict__ out_tokens, float* __restrict__ prob_out, int prob_stride, int apply_softmax,
    int do_sample, int head_act, u64* __restrict__ X) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    if (wg >= 1 && wg <= kD10Skip) {
        // ---- 32 skip rows: thread (r = tid / 8, s = tid % 8) holds Ws_l[row][4 s .. 4 s + 3] of every layer ----
        float* zs = sm;                                               // [L][32]
        float* hs = sm + kD10MaxL * 32;                               // [32]
        const int row = 32 * (wg - 1) + (tid >> 3), sl = tid & 7;
        float4 wh[8];                                                 // head row `tid`, the 32 columns of this workgroup's skip rows
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) wh[jj] = *reinterpret_cast<const float4*>(Ph + (8 * (wg - 1) + jj) * 1024 + 4 * tid);
        float4 w[kD10MaxL];
#pragma unroll
        for (int l = 0; l < kD10MaxL; ++l)
            w[l] = l < nlayers ? *reinterpret_cast<const float4*>(P + (long long)l * kLayerFloats + kGateFloats + kProjFloats +
                                                                  ((row & 1) * 8 + sl) * 512 + 4 * (row >> 1))
                               : make_float4(0.f, 0.f, 0.f, 0.f);
        bool dead = false;                                            // a wait of this thread gave up: wait for nothing any more
        for (int it = 0; it < nsteps; ++it) {
            const unsigned seq = (unsigned)(it + 1);
            float acc = 0.f;
#pragma unroll
            for (int l0 = 0; l0 < kD10MaxL; l0 += 8) {
                if (l0 < nlayers) {
                    // a poll is a round trip to memory: the four waves fetch EIGHT layers per round trip (wave w the layers
                    // l0 + 2 w and l0 + 2 w + 1, one per half wave) into a shared table; the chain needs ~0.55 us per layer
                    const int lp = l0 + 2 * wv + (lane >> 5);
                    if (lp < nlayers) zs[lp * 32 + (lane & 31)] = xget(X + kXZ + lp * 64 + (lane & 31), seq, X + x_err(nlayers), dead);
                    lds_barrier();
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (l0 + u < nlayers) acc += dot4(w[l0 + u], *reinterpret_cast<const float4*>(zs + (l0 + u) * 32 + 4 * sl));
                }
            }
            // the row's eight partial sums, pairwise (lanes 8 r .. 8 r + 7 of a row of 16 DPP lanes: xor 1, 2, then 4)
            acc += dpp_f(acc, 101);
            acc += dpp_f(acc, 102);
            acc += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(acc), 0x101F));    // xor 4 inside groups of 32
            // this workgroup's 32 activated skip values -> LDS, then its share of EVERY logit: thread q adds the 32 products
            // of head row q with them (8 float4 of head weights per thread, resident); workgroup 0 adds the eight shares
            if (sl == 0) hs[tid >> 3] = act_apply(acc, head_act);
            lds_barrier();
            float pl = 0.f;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pl += dot4(wh[jj], *reinterpret_cast<const float4*>(hs + 4 * jj));
            xput(X + x_pl(nlayers) + (wg - 1) * 256 + tid, pl, seq);
            lds_barrier();                                            // zs / hs are rewritten by the next step
        }
        return;
    }
    // ---- workgroup 0: the chain, softmax, sampling (k_decode_fast without its skip waves and head) ----
    float* Elds = sm;                                   // [256][2][32] embedding table of the causal layer
    float* xold = Elds + 256 * 2 * 32;
    float* xcur = xold + kMaxFastLayers * 32;
    float* zall = xcur + (kMaxFastLayers + 1) * 32;
    float* hvec = zall + kMaxFastLayers * 32;
    float* lg = hvec + 256;
    float* red = lg + 256;
    double* cdf = reinterpret_cast<double*>(red + 16);
    int* s_tok = reinterpret_cast<int*>(cdf + 256);
    int* ringt = s_tok + 4;
    int* dmask = ringt + kMaxFastLayers;
    int* ready = dmask + kMaxFastLayers;
    float* aold = reinterpret_cast<float*>(ready + 4);
    for (int i = tid; i < 256 * 2 * 32 / 4; i += kFT)
        reinterpret_cast<float4*>(Elds)[i] = reinterpret_cast<const float4*>(E)[i];
    for (int i = tid; i < nlayers; i += kFT) { ringt[i] = layers[i].ring; dmask[i] = layers[i].d - 1; }
    if (tid == 0) { s_tok[0] = first_token; s_tok[1] = tok_ring[0]; }
    const float hb = hbias ? hbias[tid] : 0.f;
    __syncthreads();
    FastLds S{xold, xcur, zall, aold, ready, ready + 1};
    bool dead0 = false;                                   // this thread's wait for the logit shares gave up: wait for nothing any more

    for (int it = 0; it < nsteps; ++it) {
        const unsigned n = (unsigned)(n0 + it);
        const unsigned seq = (unsigned)(it + 1);
        const int token = s_tok[0], tprev = s_tok[1];
        const double u_draw = do_sample ? uniforms[it] : 0.0;
#ifdef WN_DEC3_STAMPS
        long long st0 = clock64();
#endif
        if (it == 0) {                                    // later steps: fetched during the previous step's wait (below)
            for (int i = tid; i < nlayers * 32; i += kFT) {
                const int l = i >> 5;
                xold[i] = arena[ringt[l] + (long long)(n & (unsigned)dmask[l]) * 32 + (i & 31)];
            }
        }
        if (tid < 32) xcur[tid] = Elds[(tprev * 2 + 0) * 32 + tid] + Elds[(token * 2 + 1) * 32 + tid];
        if (tid == 0) {
            __hip_atomic_store(ready, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(ready + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (wv == 0) {
            ChainW w[2];
            const ChainOff co = chain_offsets(lane);
            load_chain(w[0], P, 0, co);
            __syncthreads();
            const float xc0 = xcur[lane & 31];
            wait_count(S.ready_old, 1);
            const float a0 = aold[lane];
            if (nlayers % kUnroll == 0) chain_run<true>(S, P, nlayers, lane, xc0, a0, w, co);
            else chain_run<false>(S, P, nlayers, lane, xc0, a0, w, co);
        } else if (wv == 1) {
            OldW wo[2];
            load_old(wo[0], P, 0, 4u * lane);
            __syncthreads();
            for (int l0 = 0; l0 < nlayers; l0 += kUnroll) {
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const int l = l0 + u;
                    if (l < nlayers) {
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                        if (l + 1 < nlayers) load_old(wo[(u + 1) & 1], P, l + 1, 4u * lane);
                        old_layer(S, wo[u & 1], l, lane);
                    }
                }
            }
        } else if (wv == 2) {
            __syncthreads();
            // the publisher: follows the chain through the layer counter in LDS and hands every layer's z to the other CUs
            for (int l = 0; l < nlayers; ++l) {
                wait_layer(S, l + 1);
                xput(X + kXZ + l * 64 + lane, zall[l * 32 + (lane & 31)], seq);
            }
        } else {
            __syncthreads();
        }
        lds_barrier();                                    // the chain has written every x_cur
#ifdef WN_DEC3_STAMPS
        long long st1 = clock64();
#endif
        for (int i = tid; i < nlayers * 32; i += kFT) {   // this step's x_cur of every layer becomes the newest ring column
            const int l = i >> 5;
            const unsigned dm = (unsigned)dmask[l];
            float* ring = arena + ringt[l] + (i & 31);
            const float xn = xcur[i];
            ring[(long long)(n & dm) * 32] = xn;
            // ... and the NEXT step's x[n + 1 - d] is fetched now, under the wait for the logits: with d = 1 it is the value just
            // stored, otherwise a column written d - 1 steps ago (the chain and wave 1 are through with xold: barrier above)
            xold[i] = dm == 0u ? xn : ring[(long long)((n + 1u) & dm) * 32];
        }
        // logit `tid` = bias + the eight workgroups' shares, added in workgroup order.  The eight entries are requested
        // together (one memory round trip), re-requested together until all carry this step's number
        float v;
        {
            const u64* e = X + x_pl(nlayers) + tid;
            u64 wd[kD10Skip];
            int spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < kD10Skip; ++k) wd[k] = __hip_atomic_load(e + k * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < kD10Skip; ++k) ok = ok && (unsigned)(wd[k] >> 32) == seq;
                if (ok || dead0) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                if (spins > (1 << 21) || ((spins & 0xfff) == 0 &&
                                          __hip_atomic_load(X + x_err(nlayers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    // given up (see xget): the launch is void, say so, and wait for nothing from here on
                    __hip_atomic_store(X + x_err(nlayers), (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dead0 = true;
                }
            }
            v = hb;
#pragma unroll
            for (int k = 0; k < kD10Skip; ++k) v += __uint_as_float((unsigned)wd[k]);
        }
#ifdef WN_DEC3_STAMPS
        long long st2 = clock64();
#endif
        if (apply_softmax) {
            float m = wave_allmax(v);
            if (lane == 0) red[wv] = m;
            lds_barrier();
            m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            const float e = expf(v - m);
            float s = wave_allsum(e);
            if (lane == 0) red[4 + wv] = s;
            lds_barrier();
            s = red[4] + red[5] + red[6] + red[7];
            v = e * (1.f / s);
        }
        lg[tid] = v;
        if (prob_out) prob_out[(long long)it * prob_stride + tid] = v;
        lds_barrier();
        if (do_sample) {
            double c = wave_scan_f64((double)lg[tid]);
            if (lane == 63) cdf[wv] = c;
            lds_barrier();
            {
                const double w0 = cdf[0], w1 = cdf[1], w2 = cdf[2], w3 = cdf[3];
                c += wv > 0 ? w0 : 0.0;
                c += wv > 1 ? w1 : 0.0;
                c += wv > 2 ? w2 : 0.0;
                const double tot = ((w0 + w1) + w2) + w3;
                const double q = c / tot;
                const bool gt = q > u_draw;
                const bool near = fabs(q - u_draw) < 1e-12;
                const unsigned long long bal = __ballot(gt), nb = __ballot(near);
                if (lane == 0) {
                    reinterpret_cast<int*>(red)[8 + wv] = bal ? wv * 64 + __ffsll((long long)bal) - 1 : 256;
                    reinterpret_cast<int*>(red)[12 + wv] = nb ? 1 : 0;
                }
            }
            lds_barrier();
            const int* rr = reinterpret_cast<const int*>(red);
            const bool ambiguous = (rr[12] | rr[13] | rr[14] | rr[15]) != 0;
            if (ambiguous) {
                if (tid == 0) {
                    double cs = 0.0;
                    for (int i = 0; i < 256; ++i) { cs += (double)lg[i]; cdf[i] = cs; }
                }
                lds_barrier();
                {
                    const double tot = cdf[255];
                    const bool gt = cdf[tid] / tot > u_draw;
                    const unsigned long long bal = __ballot(gt);
                    lds_barrier();
                    if (lane == 0) reinterpret_cast<int*>(red)[8 + wv] = bal ? wv * 64 + __ffsll((long long)bal) - 1 : 256;
                }
                lds_barrier();
            }
            if (tid == 0) {
                const int* r = reinterpret_cast<const int*>(red) + 8;
                int idx = min(min(r[0], r[1]), min(r[2], r[3]));
                if (idx > 255) idx = 255;
                out_tokens[it] = idx;
                s_tok[1] = token;
                s_tok[0] = idx;
            }
        } else if (tid == 0) {
            s_tok[1] = token;
        }
        __syncthreads();                           // full barrier: orders this step's ring stores before the next step's loads
#ifdef WN_DEC3_STAMPS
        if (tid == 0) {
            u64* D = X + nlayers * 64 + 8 * 256;
            D[0] += (u64)(st1 - st0); D[1] += (u64)(st2 - st1); D[2] += (u64)(clock64() - st2); D[3] += 1;
        }
#endif
    }
    if (tid == 0) tok_ring[0] = s_tok[1];
}

#ifdef WN_DEC3_STAMPS
static u64* g_dbg_decX = nullptr;        // diagnostic build only: where wn_debug_dec3_stamps finds the last launch's stamps
static int g_dbg_decL = 0;
#endif
__global__ __launch_bounds__(kFT, 1) void k_decode_fast3(
    const float* __restrict__ P, const float* __restrict__ Ph, const float* __restrict__ hbias,
    const float* __restrict__ E, const DecLayer* __restrict__ layers, int nlayers, float* __restrict__ arena,
    int* __restrict__ tok_ring, long long n0, int nsteps, int first_token, const double* __restrict__ uniforms,
    int32_t* __restrict__ out_tokens, float* __restrict__ prob_out, int prob_stride, int apply_softmax,
    int do_sample, int head_act, u64* __restrict__ X) {
    decode_fast3_body((int)blockIdx.x, P, Ph, hbias, E, layers, nlayers, arena, tok_ring, n0, nsteps, first_token, uniforms,
                      out_tokens, prob_out, prob_stride, apply_softmax, do_sample, head_act, X);
}

// N independent utterances in ONE launch: nine workgroups each (the single-GPU form of "replicas only", SURVEY 8(e): batch 1
// has a strict sample-to-sample dependency, so the other 247 CUs can only run OTHER utterances).  Every utterance has its own
// decoder state (rings, token ring, exchange entries, packed weights: a handle each), its own uniforms and outputs; the groups
// share nothing and never wait for each other, so an utterance's tokens are those of its own wn_decoder_run, bit for bit.
struct DecBatchItem {
    const float* P; const float* hbias; const float* E; const DecLayer* layers; float* arena; int* tok_ring;
    long long n0; const double* uniforms; int32_t* out_tokens; float* prob_out; u64* X; int first_token; int pad;
};
struct DecBatchArgs { DecBatchItem it[kDecMaxBatch]; };
__global__ __launch_bounds__(kFT, 1) void k_decode_fast3_batch(const DecBatchArgs a, int nlayers, int nsteps, int prob_stride,
                                                               int head_act) {
    const int u = blockIdx.x / (kD10Skip + 1);
    const DecBatchItem& q = a.it[u];
    decode_fast3_body((int)blockIdx.x - u * (kD10Skip + 1), q.P, q.P + (size_t)nlayers * kLayerFloats, q.hbias, q.E, q.layers,
                      nlayers, q.arena, q.tok_ring, q.n0, nsteps, q.first_token, q.uniforms, q.out_tokens, q.prob_out,
                      prob_stride, 1, 1, head_act, q.X);
}

__global__ void k_decode_zero_x(u64* X, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) X[i] = 0ull;
}

size_t decode_fast_lds_bytes() {
    return (size_t)(256 * 2 * 32 + kMaxFastLayers * 32 + (kMaxFastLayers + 1) * 32 + kMaxFastLayers * 32 + 256 + 256 + 16) * 4 +
           256 * 8 + (4 + 2 * kMaxFastLayers + 4) * 4 + kMaxFastLayers * 64 * 4;
}
size_t decode_fast_pack_floats(int nlayers) {            // weights, then the exchange entries of the three-workgroup form (8 bytes each)
    return (size_t)nlayers * kLayerFloats + 256 * 256 + 2 * ((size_t)nlayers * 64 + 8 * 256 + 16);
}

int decode_fast_pack(const WnDecoderDesc* d, float* dst, hipStream_t s) {
    const int L = d->n_blocks * d->n_layers;
    for (int l = 0; l < L; ++l)
        hipLaunchKernelGGL(k_pack_fast_layer, dim3(1), dim3(128), 0, s, d->Wf[l], d->Wg[l], d->Wp[l], d->Ws[l],
                           dst + (size_t)l * kLayerFloats);
    hipLaunchKernelGGL(k_pack_fast_head, dim3(1), dim3(256), 0, s, d->head_W[0], dst + (size_t)L * kLayerFloats);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int decode_fast_launch(const float* P, int nlayers, const float* hbias, const float* E, const DecLayer* layers,
                       float* arena, int* tok_ring, long long n0, int nsteps, int first_token,
                       const double* uniforms, int32_t* out_tokens, float* prob_out, int prob_stride,
                       int apply_softmax, int do_sample, int head_act, bool three_wgs, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_decode_fast),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)decode_fast_lds_bytes()));
        WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_decode_fast3),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)decode_fast_lds_bytes()));
        attr = true;
    }
    if (three_wgs && nsteps > 1 && nsteps < (1 << 30) && nlayers <= kD10MaxL) {
        // nine workgroups that wait for each other: the device the stream belongs to must be able to hold them at once (asked
        // per device, per call: cheap, and a process may drive several devices).  What the count cannot see -- other work
        // holding the CUs for seconds -- ends in a given-up wait that wn_decoder_status reports (no trap, no hang).
        int dev = 0, n_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < kD10Skip + 1)
            three_wgs = false;
    }
    if (three_wgs && nsteps > 1 && nsteps < (1 << 30) && nlayers <= kD10MaxL) {
        // the exchange entries live behind the packed weights (decode_fast_pack_floats); cleared by a kernel, in stream order
        u64* X = reinterpret_cast<u64*>(const_cast<float*>(P) + (size_t)nlayers * kLayerFloats + 256 * 256);
        const int nx = nlayers * 64 + 8 * 256 + 16;
#ifdef WN_DEC3_STAMPS
        g_dbg_decX = X; g_dbg_decL = nlayers;
#endif
        hipLaunchKernelGGL(k_decode_zero_x, dim3(cdiv(nx, 256)), dim3(256), 0, s, X, nx);
        hipLaunchKernelGGL(k_decode_fast3, dim3(kD10Skip + 1), dim3(kFT), decode_fast_lds_bytes(), s, P,
                           P + (size_t)nlayers * kLayerFloats, hbias, E, layers, nlayers, arena, tok_ring, n0, nsteps,
                           first_token, uniforms, out_tokens, prob_out, prob_stride, apply_softmax, do_sample, head_act, X);
        WN_LAUNCH_CHECK();
        return WN_OK;
    }
    hipLaunchKernelGGL(k_decode_fast, dim3(1), dim3(kFT), decode_fast_lds_bytes(), s, P,
                       P + (size_t)nlayers * kLayerFloats, hbias, E, layers, nlayers, arena, tok_ring, n0, nsteps,
                       first_token, uniforms, out_tokens, prob_out, prob_stride, apply_softmax, do_sample, head_act);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

int decode_fast_batch_ok(int nlayers, int n_utt, int nsteps) {
    if (n_utt < 1 || n_utt > kDecMaxBatch || nsteps < 2 || nsteps >= (1 << 30) || nlayers > kD10MaxL) return 0;
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return 0;
    return n_utt * (kD10Skip + 1) <= n_cu ? 1 : 0;             // every workgroup resident (one per CU by LDS footprint)
}

int decode_fast_launch_batch(int n_utt, const float* const* P, int nlayers, const float* const* hbias, const float* const* E,
                             const DecLayer* const* layers, float* const* arena, int* const* tok_ring, const long long* n0,
                             int nsteps, const int* first_token, const double* const* uniforms, int32_t* const* out_tokens,
                             float* const* prob_out, int prob_stride, int head_act, bool same_weights, hipStream_t s) {
    if (!decode_fast_batch_ok(nlayers, n_utt, nsteps)) {
        wn::set_error("decode batch: %d utterances x 9 workgroups do not fit the device (or fewer than 2 steps)", n_utt);
        return WN_ESHAPE;
    }
    static bool attr = false;
    if (!attr) {
        WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_decode_fast3_batch),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)decode_fast_lds_bytes()));
        attr = true;
    }
    DecBatchArgs a{};
    const int nx = nlayers * 64 + 8 * 256 + 16;
    for (int u = 0; u < n_utt; ++u) {
        DecBatchItem& q = a.it[u];
        // same_weights (the caller's word that every handle was created from the same weights): every utterance reads utterance
        // 0's packed weights, embedding table and head bias -- the 28 chain workgroups then stream ONE 0.96 MB copy per step
        // through the XCDs' L2s instead of 28; the state (rings, token ring, exchange entries) stays per utterance
        const int wu = same_weights ? 0 : u;
        q.P = P[wu]; q.hbias = hbias[wu]; q.E = E[wu]; q.layers = layers[u]; q.arena = arena[u]; q.tok_ring = tok_ring[u];
        q.n0 = n0[u]; q.uniforms = uniforms[u]; q.out_tokens = out_tokens[u]; q.prob_out = prob_out ? prob_out[u] : nullptr;
        q.first_token = first_token[u];
        q.X = reinterpret_cast<u64*>(const_cast<float*>(P[u]) + (size_t)nlayers * kLayerFloats + 256 * 256);
        hipLaunchKernelGGL(k_decode_zero_x, dim3(cdiv(nx, 256)), dim3(256), 0, s, q.X, nx);
    }
    hipLaunchKernelGGL(k_decode_fast3_batch, dim3(n_utt * (kD10Skip + 1)), dim3(kFT), decode_fast_lds_bytes(), s, a, nlayers,
                       nsteps, prob_stride, head_act);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

// 0 = the last nine-workgroup run's exchange completed; 1 = a wait gave up (its tokens are void).  Synchronises the stream.
int decode_fast_status(const float* P, int nlayers, hipStream_t s, int* gave_up) {
    const u64* X = reinterpret_cast<const u64*>(P + (size_t)nlayers * kLayerFloats + 256 * 256);
    u64 w = 0;
    WN_HIP(hipMemcpyAsync(&w, X + x_err(nlayers), sizeof(w), hipMemcpyDeviceToHost, s));
    WN_HIP(hipStreamSynchronize(s));
    *gave_up = w != 0 ? 1 : 0;
    return WN_OK;
}

}  // namespace wn

#ifdef WN_DEC3_STAMPS
extern "C" __attribute__((visibility("default"))) int wn_debug_dec3_stamps(unsigned long long* dst) {
    if (!wn::g_dbg_decX) return -1;
    return (int)hipMemcpy(dst, wn::g_dbg_decX + wn::g_dbg_decL * 64 + 8 * 256, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}
#endif
