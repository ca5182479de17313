// Channel GEMMs of the bf16-storage path (BASELINE config 5): bf16 operands from HBM, fp32 accumulation on
// v_mfma_f32_32x32x16_bf16, LDS-DMA double-buffered tiles in the 256-byte-row format of w16.hpp.
//
//   k16_cgemm   out[n][m] = sum_src sum_k W[m][src, k] X_src[n + shift_src][k]  (+ bias, + residual, x relu mask)
//               128 time columns x 128 output channels per workgroup, 128-deep stages.  Used for the deferred skip sum
//               (wavenet.py:574-582: 40 sources), dz_skip = Ws^T dskip, the head 1x1 convs and their dx
//               (wavenet.py:584-593), and the layer's data gradient dx = dout + W1^T dab[t] + W0^T dab[t + d].
//   k16_wgrad   dW[m][n] += sum_t A[t][m] B[t + shift][n]: contraction over TIME, both operands taken from row-major
//               [t][channel] tiles with the hardware transposing LDS read (ds_read_b64_tr_b16).  256 x 256 outputs per
//               workgroup, time split into slabs over the grid, fp32 atomics at the end of a slab.  Used for the conv
//               weight gradients of all layers in one launch, dWs of all layers, and the head's dW.
#include "w16.hpp"
#include "w16_gemm.hpp"

namespace w16 {

// =============================================================================================
// k16_cgemm
// =============================================================================================
static constexpr int kCTileB = 128 * 256;               // 32 KB: 128 rows
static constexpr int kCgLds = 4 * kCTileB;              // X[2], W[2]

template <bool RELU_X, int EP, bool OUT_F32>             // EP: 0 none, 1 + residual, 2 x (mask > 0)
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void k16_cgemm(CG16 a) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    auto xt = [&](int buf) { return lds + buf * kCTileB; };
    auto wt = [&](int buf) { return lds + (2 + buf) * kCTileB; };
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int wn = w & 3, wm = w >> 2;
    // block -> (column block, m block): m blocks of one column block sit on the same XCD at adjacent slots (they re-read
    // the same X tiles from that XCD's L2)
    int id = blockIdx.x;
    const int nmb = a.M >> 7;
    int nblk, mblk;
    {
        const int nnb = a.n_blocks;
        if ((nnb & 7) == 0) {
            const int xcd = id & 7, slot = id >> 3;
            mblk = slot % nmb;
            nblk = (slot / nmb) * 8 + xcd;
        } else {
            mblk = id % nmb;
            nblk = id / nmb;
        }
    }
    const int bpb = a.blocks_per_b;
    const int b = nblk / bpb;
    const int r0 = (nblk - b * bpb) * 128;
    const int m0 = mblk * 128;
    const int spk = a.ksrc >> 7;                        // stages per source
    const int nst = a.nsrc * spk;

    auto issue = [&](int st, int buf) {
        const int src = st / spk;
        const int k0 = (st - src * spk) * 128;
        const int sh = a.x_row0 + r0 + a.shift[src];
        const bf16* xb = a.X[src] + (long long)b * a.x_rows_per_b * a.ldx + k0;
        const int hi = a.x_rows_per_b - 1;
        dma_pieces(xt(buf), lane, w, 8, 4, [&](int r) {
            int t = sh + r;
            t = t < 0 ? 0 : (t > hi ? hi : t);
            return xb + (long long)t * a.ldx;
        });
        const bf16* wb = a.W + (long long)m0 * a.K + st * 128;
        dma_pieces(wt(buf), lane, w, 8, 4, [&](int r) { return wb + (long long)r * a.K; });
    };

    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    issue(0, 0);
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        wait_vm<0>();
        barrier();
        if (st + 1 < nst) issue(st + 1, buf ^ 1);
        {
            const int src = st / spk;
            const int sh = a.x_row0 + r0 + a.shift[src];
            if (sh < 0 || sh + 127 >= a.x_rows_per_b) {  // rows outside the clip read as 0
                for (int r = w; r < 128; r += 8)
                    if (sh + r < 0 || sh + r >= a.x_rows_per_b)
                        *reinterpret_cast<unsigned*>(xt(buf) + r * 256 + lane * 4) = 0u;
                barrier();
            }
        }
        // all 24 operand fragments of the stage are requested before the first MFMA (left to itself the compiler paired
        // every MFMA with its own ds_read + s_waitcnt lgkmcnt(0): one exposed LDS latency per matrix instruction)
        const int row = 32 * wn + j;
        bf16x8 bv[8], av[2][8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            bv[s] = frag_row(xt(buf), row, s, h);
            av[0][s] = frag_row(wt(buf), 64 * wm + j, s, h);
            av[1][s] = frag_row(wt(buf), 64 * wm + 32 + j, s, h);
        }
        __builtin_amdgcn_sched_barrier(0);               // keep the reads above, the MFMAs below
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (RELU_X) bv[s] = relu8(bv[s]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mt][s], bv[s], acc[mt], 0, 0, 0);
        }
    }
    // ---- epilogue ----
    const long long orow0 = (long long)b * a.rows_per_b + r0;
    const int nrows = a.rows_per_b - r0 < 128 ? a.rows_per_b - r0 : 128;
    const int col0 = mblk * a.ob_col;
    barrier();                                            // every wave is done with the last stage's tiles
    if (EP != 0) {
        const bf16* eb = a.extra + orow0 * a.lde + col0;
        dma_pieces(wt(0), lane, w, 8, 4, [&](int r) { return eb + (long long)(r < nrows ? r : nrows - 1) * a.lde; });
        wait_vm<0>();
        barrier();
    }
    const int row = 32 * wn + j;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mc = 64 * wm + 32 * mt + 8 * q + 4 * h;    // first of 4 consecutive channels inside the block
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[mt][4 * q + e] + (a.bias ? a.bias[m0 + mc + e] : 0.f);
            const int o = toff(row, mc >> 3) + 8 * ((mc >> 2) & 1);
            if (EP != 0) {
                const bf16x4 xv = *reinterpret_cast<const bf16x4*>(wt(0) + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (EP == 1) v[e] += (float)xv[e];
                    else v[e] = (float)xv[e] > 0.f ? v[e] : 0.f;
                }
            }
            if (OUT_F32) {
                if (row < nrows)
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + (long long)mblk * a.ob_stride +
                                               (orow0 + row) * a.ldo + col0 + mc) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                *reinterpret_cast<bf16x4*>(xt(0) + o) = pack4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    if (!OUT_F32) {
        barrier();
        bf16* ob = reinterpret_cast<bf16*>(a.out) + (long long)mblk * a.ob_stride + orow0 * a.ldo + col0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = w + 8 * i;
            const int r = 4 * p + (lane >> 4);
            const int c = (lane & 15) ^ key(r);
            const u32x4 v = *reinterpret_cast<const u32x4*>(xt(0) + p * 1024 + lane * 16);
            if (r < nrows) st16_wt(ob + (long long)r * a.ldo + c * 8, v);
        }
    }
}

static int launch_cgemm256(CG16& a, hipStream_t s);

int launch_cgemm(CG16& a, hipStream_t s) {
    if (a.M % 128 || a.ksrc % 128 || a.nsrc < 1 || a.nsrc > kMaxSrc16 || a.K != a.nsrc * a.ksrc) {
        wn::set_error("w16 cgemm: unsupported shape M=%d K=%d nsrc=%d ksrc=%d", a.M, a.K, a.nsrc, a.ksrc);
        return WN_ESHAPE;
    }
    if (a.ep != 0 && a.ob_stride != 0) { wn::set_error("w16 cgemm: extra operand with split outputs"); return WN_EARG; }
    {
        bool plain = a.ep == 0 && !a.relu_x && !a.out_f32 && a.M % 256 == 0 && a.ksrc % 64 == 0 && a.ob_col % 128 == 0 &&
                     (a.ob_col == 0 || a.ob_col == 128) && (long long)a.B * a.rows_per_b >= 4096;
        // the sources must be equally spaced (they are: one [L][B][T][C] allocation): a pointer fetched from the
        // argument table inside the stage loop is a scalar load on lgkmcnt, and with one in the loop hipcc waits
        // lgkmcnt(0) in front of every MFMA group instead of counting the LDS reads it has in flight
        a.x_src_stride = a.nsrc > 1 ? a.X[1] - a.X[0] : 0;
        for (int i = 0; i < a.nsrc && plain; ++i) plain = a.shift[i] == 0 && a.X[i] - a.X[0] == i * a.x_src_stride;
        if (plain && a.x_row0 >= 0 && a.x_row0 + a.rows_per_b <= a.x_rows_per_b) return launch_cgemm256(a, s);
    }
    a.blocks_per_b = (a.rows_per_b + 127) / 128;
    a.n_blocks = a.B * a.blocks_per_b;
    const int grid = a.n_blocks * (a.M / 128);
#define CGL(RX, EP, F32)                                                                                          \
    do {                                                                                                          \
        static bool attr = false;                                                                                 \
        if (!attr) {                                                                                              \
            WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k16_cgemm<RX, EP, F32>),                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kCgLds));                      \
            attr = true;                                                                                          \
        }                                                                                                         \
        hipLaunchKernelGGL((k16_cgemm<RX, EP, F32>), dim3(grid), dim3(512), kCgLds, s, a);                        \
    } while (0)
    const int key = (a.relu_x ? 8 : 0) | (a.ep << 1) | (a.out_f32 ? 1 : 0);
    switch (key) {
        case 0: CGL(false, 0, false); break;
        case 1: CGL(false, 0, true); break;
        case 2: CGL(false, 1, false); break;
        case 4: CGL(false, 2, false); break;
        case 8: CGL(true, 0, false); break;
        case 9: CGL(true, 0, true); break;
        default: wn::set_error("w16 cgemm: unsupported epilogue combination %d", key); return WN_EARG;
    }
#undef CGL
    WN_LAUNCH_CHECK();
    return WN_OK;
}

// =============================================================================================
// k16_cgemm256: the same contraction for the two large ones (the deferred skip sum, K = 128 L; dz_skip of all layers,
// M = 128 L): 256 columns x 256 output channels per workgroup, 64-deep stages (128-byte tile rows).
// With 128 x 128 blocks these GEMMs ran one 64 KB stage per ~1.8 us of LDS-DMA latency (measured: 0.98 / 0.87 ms, the
// same whether the LDS reads were serialised with the MFMAs or batched): the latency is paid per stage, so a stage now
// carries four times the work (8.4 MFLOP per 64 KB) and the X tile is re-read from L2 half as often.
// Tile format here: rows of 128 bytes, chunk c (0..7) of row r at position c ^ ((r >> 1) & 7) -- two rows share a
// 256-byte bank line, so the 16 rows of a ds_read_b128 lane group hit 16 distinct 16-byte slots.
// =============================================================================================
static constexpr int kBTileB = 256 * 128;               // 32 KB: 256 rows x 64 bf16
static constexpr int kCg256Lds = 4 * kBTileB;           // X[2], W[2]

__device__ __forceinline__ int boff(int r, int c) { return (r << 7) + ((c ^ ((r >> 1) & 7)) << 4); }

__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void k16_cgemm256(CG16 a) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    auto xt = [&](int buf) { return lds + buf * kBTileB; };
    auto wt = [&](int buf) { return lds + (2 + buf) * kBTileB; };
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int wn = w & 1, wm = w >> 1;                   // wave tile: columns 128 wn .., channels 64 wm ..
    const int nmb = a.M >> 8;
    int nblk, mblk;
    {
        const int id = blockIdx.x;
        if ((a.n_blocks & 7) == 0) {
            // An XCD (linear ids = xcd mod 8, dispatched in id order) takes its column blocks in panels of 8 and, inside a
            // panel, the channel blocks one after the other: the ~64 workgroups in flight on its 32 CUs are then 8 channel
            // blocks x the panel's 8 column blocks, so a W slice is fetched once per panel and the panel's X tiles (2 MB) stay
            // in that XCD's L2 until all channel blocks have passed.  (With the channel block as the fast index the XCD cycled
            // through ALL of W -- 5.2 MB for config 5's dz, above its 4 MB of L2 -- once per 3 column blocks: 2.1 GB fetched
            // for a 0.1 GB X.)
            const int xcd = id & 7, slot = id >> 3;
            const int nbx = a.n_blocks >> 3;
            const int panel = slot / (8 * nmb);
            const int pw = nbx - 8 * panel < 8 ? nbx - 8 * panel : 8;
            const int rem = slot - panel * (8 * nmb);
            mblk = rem / pw;
            nblk = (8 * panel + (rem - mblk * pw)) * 8 + xcd;
        } else {
            mblk = id % nmb;
            nblk = id / nmb;
        }
    }
    const int bpb = a.blocks_per_b;
    const int b = nblk / bpb;
    const int r0 = (nblk - b * bpb) * 256;
    const int m0 = mblk * 256;
    const int spk = a.ksrc >> 6;                         // stages per source
    const int nst = a.nsrc * spk;
    const int lr = lane >> 3, lp = lane & 7;
    // The epilogue's arguments are fetched HERE: hipcc otherwise hoists their s_load to just in front of the stage loop, and
    // with a scalar load pending at the loop's entry (scalar loads return out of order) it can no longer count the LDS reads
    // in flight -- it put s_waitcnt lgkmcnt(0) behind the six fresh reads of every stage's first k-step (now lgkmcnt(6)).
    // Measured: no difference in time -- like every other instruction-level change to this kernel.
    asm volatile("" ::"s"(a.out), "s"(a.bias), "s"(a.ldo), "s"(a.ob_stride), "s"(a.ob_col), "s"(a.rows_per_b), "s"(a.x_src_stride));

    // Requests as `scalar base + fixed 32-bit lane offset`: the rows and chunk positions a lane copies do not change from
    // stage to stage, only the uniform k offset (and source) does.  (With a 64-bit pointer per piece computed in vector
    // registers -- row clamp, multiply, add -- and the builtin's v_readfirstlane + M0 write, issuing the eight requests of a
    // stage cost a wave several hundred cycles of its 1,024-cycle MFMA budget: the same finding as in k_colgemm_h2q.)
    unsigned xo[4], wo[4];
    {
        const int sh = a.x_row0 + r0;
        const int hi = a.x_rows_per_b - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = w + 8 * i;
            const int r = 8 * p + lr;
            const int c = lp ^ ((r >> 1) & 7);
            int t = sh + r;
            t = t > hi ? hi : t;
            xo[i] = (unsigned)(((long long)t * a.ldx + c * 8) * 2);
            wo[i] = (unsigned)(((long long)r * a.K + c * 8) * 2);
        }
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
    const char* xbase0 = reinterpret_cast<const char*>(a.X[0] + ((long long)b * a.x_rows_per_b) * a.ldx);
    const char* wbase0 = reinterpret_cast<const char*>(a.W + (long long)m0 * a.K);
    auto issue = [&](int st, int buf) {
        const int src = st / spk;
        const int k0 = (st - src * spk) * 64;
        const char* xb = xbase0 + ((long long)src * a.x_src_stride + k0) * 2;
        const char* wb = wbase0 + (long long)st * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned mx = lds0 + buf * kBTileB + (w + 8 * i) * 1024, mw = mx + 2 * kBTileB;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(xo[i]), "s"(xb), "s"(mx) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(wo[i]), "s"(wb), "s"(mw) : "memory", "m0");
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.f;
    // Fragments are double-buffered in registers: the LDS reads of k-step s + 1 are issued before the MFMAs of k-step s,
    // so a wave's matrix stream never waits for a read it has just issued (with "4 reads, lgkmcnt(0), 4 MFMAs" groups the
    // two waves of a SIMD each exposed ~150 cycles of LDS latency per 128 cycles of MFMA: 35 % matrix utilisation by PMC).
    // The one barrier of a stage sits in front of the LAST k-step's MFMAs: by then that k-step's fragments are in
    // registers, so the stage's buffer is free for the stage after next, and the next stage's buffer has landed, so its
    // first fragments are fetched under those MFMAs -- the stream continues across the stage boundary.
    bf16x8 fa[2][2], fb[2][4];
    auto ld = [&](int buf, int ks, int slot) {
        const char* xb = xt(buf);
        const char* wb = wt(buf);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) fb[slot][nt] = lds_read16(xb + boff(128 * wn + 32 * nt + j, 2 * ks + h));
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) fa[slot][mt] = lds_read16(wb + boff(64 * wm + 32 * mt + j, 2 * ks + h));
    };
    auto mm = [&](int slot) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[slot][mt], fb[slot][nt], acc[nt][mt], 0, 0, 0);
    };
    issue(0, 0);
    wait_vm<0>();
    barrier();
    if (nst > 1) issue(1, 1);
    ld(0, 0, 0);
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            ld(buf, ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(ks & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_vm<0>();
        barrier();
        ld(buf ^ 1, 0, 0);                               // stale data after the last stage: never used
        if (st + 2 < nst) issue(st + 2, buf);
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: two 128-channel halves, each a [256 rows][256 B] tile in the format of w16.hpp, then whole rows ----
    barrier();
    const int nrows = a.rows_per_b - r0 < 256 ? a.rows_per_b - r0 : 256;
    const long long orow0 = (long long)b * a.rows_per_b + r0;
    {
        char* half = lds + (wm >> 1) * (256 * 256);      // this wave's 64 channels lie in half wm >> 1
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int row = 128 * wn + 32 * nt + j;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int mc = 64 * (wm & 1) + 32 * mt + 8 * q + 4 * h;      // channel inside the half
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = acc[nt][mt][4 * q + e] + (a.bias ? a.bias[m0 + 128 * (wm >> 1) + mc + e] : 0.f);
                    *reinterpret_cast<bf16x4*>(half + toff(row, mc >> 3) + 8 * ((mc >> 2) & 1)) = pack4(v[0], v[1], v[2], v[3]);
                }
        }
    }
    barrier();
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int ob = 2 * mblk + hh;                    // 128-channel output block
        bf16* o = reinterpret_cast<bf16*>(a.out) + (long long)ob * a.ob_stride + orow0 * a.ldo + ob * a.ob_col;
        const char* half = lds + hh * (256 * 256);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = w + 8 * i;                     // 4-row piece of the half tile
            const int r = 4 * p + (lane >> 4);
            const int c = (lane & 15) ^ key(r);
            const u32x4 v = *reinterpret_cast<const u32x4*>(half + p * 1024 + lane * 16);
            if (r < nrows) st16_wt(o + (long long)r * a.ldo + c * 8, v);
        }
    }
}

static int launch_cgemm256(CG16& a, hipStream_t s) {
    // the kernel's requests carry 32-bit byte offsets from per-stage scalar bases
    if ((long long)a.x_rows_per_b * a.ldx * 2 >= (1ll << 32) || 256ll * a.K * 2 >= (1ll << 32)) {
        wn::set_error("w16 cgemm256: a clip of %d rows x %d channels exceeds the 32-bit offsets of its requests", a.x_rows_per_b, a.ldx);
        return WN_ESHAPE;
    }
    a.blocks_per_b = (a.rows_per_b + 255) / 256;
    a.n_blocks = a.B * a.blocks_per_b;
    const int grid = a.n_blocks * (a.M / 256);
    static bool attr = false;
    if (!attr) {
        WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k16_cgemm256), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kCg256Lds));
        attr = true;
    }
    hipLaunchKernelGGL(k16_cgemm256, dim3(grid), dim3(512), kCg256Lds, s, a);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

// =============================================================================================
// k16_wgrad
// =============================================================================================
static constexpr int kWT = 64;                           // time rows per stage
static constexpr int kWTileB = kWT * 256;                // 16 KB
static constexpr int kWgLds = 8 * kWTileB;               // (A0, A1, B0, B1) x 2 buffers

template <bool RELU_B>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void k16_wgrad(WG16 a) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int wm = w & 3, wn = w >> 2;
    // (slab, problem) of this workgroup.  Linear ids go round the 8 XCDs, so XCD c is given the c-th eighth of the
    // (slab-major, problem-minor) list: its workgroups stream the SAME rows of neighbouring problems, and operands that
    // neighbouring problems share (dWs: the two dskip halves of a layer pair, one dskip half for all layer pairs) come from
    // HBM once per XCD and from its L2 for the rest.  (In grid order an XCD held every slab of every fifth problem or so:
    // 42 distinct operand slabs per XCD for 30 workgroups, 2.86 GB fetched for 1.1 GB of operands.)
    // The launch is a LIST of (slab, problem) pairs, slab-major -- level s holds the problems that have more than s slabs -- and
    // only as long as the pairs that exist (a.n_items <= 256: every workgroup resident, at most 32 per XCD).
    int slab = 0, p = 0;
    {
        const int total = gridDim.x;
        int item = blockIdx.x;
        if ((total & 7) == 0) item = (item & 7) * (total >> 3) + (item >> 3);
        if (item >= a.n_items) return;                       // (the list is padded to a multiple of 8)
        while (item >= (int)a.level_cnt[slab]) { item -= (int)a.level_cnt[slab]; ++slab; }
        for (int q = 0; q < a.n_prob; ++q)
            if ((int)a.prob[q].nslab > slab) {
                if (item == 0) { p = q; break; }
                --item;
            }
    }
    const WG16Prob& pr = a.prob[p];
    // this workgroup's slab of 64-row chunks.  A problem's rows start at r_lo (a multiple of 64): below it A is zero -- the
    // columns of a layer no gradient reaches -- and nothing is read; problems get slabs in proportion to their rows
    // (launch_wgrad16), so the workgroups of a launch stream about the same number of chunks.
    const int rlo = pr.r_lo;
    const int cpb = (a.R - rlo + kWT - 1) / kWT;
    const int nch = a.nB * cpb;
    const int c_begin = (int)((long long)nch * slab / pr.nslab);
    const int c_end = (int)((long long)nch * (slab + 1) / pr.nslab);
    auto tile = [&](int buf, int which) { return lds + (buf * 4 + which) * kWTileB; };   // which: A0 A1 B0 B1

    // (the requests as inline asm: with the builtin hipcc knows LDS is written behind its back and puts s_waitcnt vmcnt(0)
    // in front of the next LDS read -- the top of the stage loop, right after the requests of the stage after next were
    // issued: the prefetch then overlapped with eight MFMAs instead of a whole stage, 5 us per 64 KB stage)
    auto issue = [&](int c, int buf) {
        const int b = c / cpb;
        const int r0 = rlo + (c - b * cpb) * kWT;
        const bf16* ab = pr.A + ((long long)b * a.a_rpb + a.a_r0) * a.lda;
        const int ahi = a.R - 1;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
            dma_pieces<false, true>(tile(buf, mh), lane, w, 8, 2, [&](int r) {
                const int t = r0 + r < ahi ? r0 + r : ahi;
                return ab + (long long)t * a.lda + 128 * mh;
            });
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            const bf16* bb = pr.Bh[nh] + (long long)b * a.b_rpb * a.ldb;
            const int sh = a.b_r0 + r0 + pr.shift[nh];
            const int bhi = a.b_rpb - 1;
            dma_pieces<false, true>(tile(buf, 2 + nh), lane, w, 8, 2, [&](int r) {
                int t = sh + r;
                t = t < 0 ? 0 : (t > bhi ? bhi : t);
                return bb + (long long)t * a.ldb;
            });
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if (c_begin >= c_end) return;
    // rows that do not exist (the ragged end of a clip, a dilated tap reaching before its start) contribute nothing
    auto fixup = [&](int c, int buf) {
        const int b = c / cpb;
        const int r0 = rlo + (c - b * cpb) * kWT;
        bool fix = r0 + kWT > a.R;
        int sh[2];
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            sh[nh] = a.b_r0 + r0 + pr.shift[nh];
            fix = fix || sh[nh] < 0 || sh[nh] + kWT > a.b_rpb;
        }
        if (fix) {
            for (int r = w; r < kWT; r += 8) {
                if (r0 + r >= a.R) {
                    *reinterpret_cast<unsigned*>(tile(buf, 0) + r * 256 + lane * 4) = 0u;
                    *reinterpret_cast<unsigned*>(tile(buf, 1) + r * 256 + lane * 4) = 0u;
                }
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
                    if (sh[nh] + r < 0 || sh[nh] + r >= a.b_rpb)
                        *reinterpret_cast<unsigned*>(tile(buf, 2 + nh) + r * 256 + lane * 4) = 0u;
            }
            barrier();
        }
    };
    // fragments double-buffered in registers, the stage barrier in front of the last k-step's MFMAs: as in k16_cgemm256
    bf16x8 fa[2][2], fb[2][4];
    auto ld = [&](int buf, int ks, int slot) {
        const char* at = tile(buf, wm >> 1);
        const char* bt = tile(buf, 2 + wn);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) fa[slot][mi] = frag_tr(at, 16 * ks, 64 * (wm & 1) + 32 * mi, lane);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) fb[slot][ni] = frag_tr(bt, 16 * ks, 32 * ni, lane);
    };
    auto mm = [&](int slot) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const bf16x8 bv = RELU_B ? relu8(fb[slot][ni]) : fb[slot][ni];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[slot][mi], bv, acc[mi][ni], 0, 0, 0);
        }
    };
    issue(c_begin, 0);
    wait_vm<0>();
    barrier();
    fixup(c_begin, 0);
    if (c_begin + 1 < c_end) issue(c_begin + 1, 1);
    ld(0, 0, 0);
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            ld(buf, ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(ks & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_vm<0>();
        barrier();
        if (c + 1 < c_end) fixup(c + 1, buf ^ 1);
        ld(buf ^ 1, 0, 0);                               // stale after the last stage: never used
        if (c + 2 < c_end) issue(c + 2, buf);
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (a.part) {
        // this workgroup's block as it sits in the accumulators: [wave][mi][ni][r][lane], 256-byte rows, coalesced
        float* pp = a.part + (size_t)(pr.slab0 + slab) * 65536 + (size_t)w * 8192 + lane;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) pp[((mi * 4 + ni) * 16 + r) * 64] = acc[mi][ni][r];
        return;
    }
    float* ob = pr.out[wm >> 1][wn];
    if (!ob) return;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 64 * (wm & 1) + 32 * mi + acc_row(r, h);
                const int n = 32 * ni + j;
                atomicAdd(ob + (long long)m * a.os_m + (long long)n * a.os_n, acc[mi][ni][r]);
            }
}

// dW += the slabs' blocks of a problem, slab 0 first: one thread per element of the 256 x 256 block, no atomics.  A slab
// whose chunk range is empty has left its block unwritten: the ranges are recomputed here.
__global__ void k16_wgrad_reduce(WG16 a) {
    const int p = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;            // [wave][mi][ni][r][lane]
    const int lane = e & 63, r = (e >> 6) & 15, ni = (e >> 10) & 3, mi = (e >> 12) & 1, w = e >> 13;
    const int wm = w & 3, wn = w >> 2, j = lane & 31, h = lane >> 5;
    const WG16Prob& pr = a.prob[p];
    float* ob = pr.out[wm >> 1][wn];
    if (!ob) return;
    const int cpb = (a.R - pr.r_lo + kWT - 1) / kWT;
    const int nch = a.nB * cpb;
    // (the slab ranges in 32-bit arithmetic: launch_wgrad16 guarantees nch * (slabs + 1) < 2^31 -- the 64-bit divisions this
    // loop used to do per thread and slab were most of its 41 us)
    float sum = 0.f;
    unsigned cb = 0;
    const int slabs = pr.nslab;
    for (int sl = 0; sl < slabs; ++sl) {
        const unsigned ce = (unsigned)nch * (unsigned)(sl + 1) / (unsigned)slabs;
        if (cb < ce) sum += a.part[(size_t)(pr.slab0 + sl) * 65536 + e];
        cb = ce;
    }
    const int m = 64 * (wm & 1) + 32 * mi + acc_row(r, h);
    const int n = 32 * ni + j;
    ob[(long long)m * a.os_m + (long long)n * a.os_n] += sum;
}

int launch_wgrad16(const WG16& a_in, int nprob, hipStream_t s) {
    if (nprob < 1 || nprob > kMaxProb16) { wn::set_error("w16 wgrad: %d problems", nprob); return WN_EARG; }
    WG16 a = a_in;
    // every workgroup resident at once (256 in all: slabs stream in step), dealt to the problems in proportion to their rows
    long long tot = 0;
    int nchp[kMaxProb16];
    for (int p = 0; p < nprob; ++p) {
        int rlo = a.prob[p].r_lo;
        rlo = rlo < 0 ? 0 : (rlo / kWT) * kWT;
        if (rlo >= a.R) rlo = ((a.R - 1) / kWT) * kWT;
        a.prob[p].r_lo = rlo;
        nchp[p] = a.nB * ((a.R - rlo + kWT - 1) / kWT);
        tot += nchp[p];
    }
    int slabs = 0, used = 0;
    for (int p = 0; p < nprob; ++p) {
        int ns = (int)((256ll * nchp[p]) / (tot > 0 ? tot : 1));
        if (ns < 1) ns = 1;
        if (ns > nchp[p]) ns = nchp[p];
        if (ns > 64) ns = 64;
        if ((long long)nchp[p] * (ns + 1) >= (1ll << 31)) { wn::set_error("w16 wgrad: %d chunks x %d slabs overflows", nchp[p], ns); return WN_ESHAPE; }
        a.prob[p].nslab = (unsigned short)ns;
        a.prob[p].slab0 = (unsigned short)used;
        used += ns;
        if (ns > slabs) slabs = ns;
    }
    if (used > 256) { wn::set_error("w16 wgrad: %d workgroups", used); return WN_ESHAPE; }
    static_assert(sizeof(WG16) <= 4096, "WG16 travels as a kernel argument");
    a.n_items = used;
    a.n_prob = nprob;
    for (int sl = 0; sl < 64; ++sl) {
        int cnt = 0;
        for (int p = 0; p < nprob; ++p) cnt += a.prob[p].nslab > sl ? 1 : 0;
        a.level_cnt[sl] = (unsigned short)cnt;
    }
    const int grid = (used + 7) & ~7;
#define WGL(RB)                                                                                                   \
    do {                                                                                                          \
        static bool attr = false;                                                                                 \
        if (!attr) {                                                                                              \
            WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k16_wgrad<RB>),                              \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kWgLds));                      \
            attr = true;                                                                                          \
        }                                                                                                         \
        hipLaunchKernelGGL((k16_wgrad<RB>), dim3(grid), dim3(512), kWgLds, s, a);                                 \
    } while (0)
    if (a.relu_b) WGL(true); else WGL(false);
#undef WGL
    WN_LAUNCH_CHECK();
    if (a.part) {
        hipLaunchKernelGGL(k16_wgrad_reduce, dim3(256, nprob), dim3(256), 0, s, a);
        WN_LAUNCH_CHECK();
    }
    return WN_OK;
}

// =============================================================================================
// weight packing (fp32 master weights -> bf16 operand images), conversions, the token embedding
// =============================================================================================
struct PackLayersArgs { const float* Wf[kMaxProb16]; const float* Wg[kMaxProb16]; const float* Wp[kMaxProb16]; };

// one thread = 8 consecutive bf16 of a layer's image (blockIdx.y = layer)
__global__ void k16_pack_layers(PackLayersArgs a, bf16* __restrict__ img) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= kLayerImg / 8) return;
    const int l = blockIdx.y;
    const float* Wf = a.Wf[l];
    const float* Wg = a.Wg[l];
    const float* Wp = a.Wp[l];
    float v[8];
    int e = g * 8;
    if (e < kConvA) {                                    // [w][gate][s][lane][8]: channel 32 w + r, k = 16 s + 8 h + jj
        const int lane = (e >> 3) & 63, s = (e >> 9) & 15, gate = (e >> 13) & 1, w = e >> 14;
        const int r = lane & 31, h = lane >> 5;
        const float* W = gate ? Wg : Wf;
        const int ch = 32 * w + r;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int k = 16 * s + 8 * h + jj;
            v[jj] = W[(ch * 128 + (k & 127)) * 2 + (k >> 7)];
        }
    } else if (e < kConvA + kProjA) {                    // [mt][s][lane][8]: Wp[32 mt + r][16 s + 8 h + jj]
        e -= kConvA;
        const int lane = (e >> 3) & 63, s = (e >> 9) & 7, mt = e >> 12;
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = Wp[(32 * mt + r) * 128 + 16 * s + 8 * h + jj];
    } else if (e < kOffDzA8) {                           // [w][s][lane][8]: row r of wave w (16 filter, 16 gate rows)
        e -= kOffConvA8;
        const int lane = (e >> 3) & 63, s = (e >> 9) & 15, w = e >> 13;
        const int r = lane & 31, h = lane >> 5;
        const float* W = (r >> 4) ? Wg : Wf;
        const int ch = 16 * w + (r & 15);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int k = 16 * s + 8 * h + jj;
            v[jj] = W[(ch * 128 + (k & 127)) * 2 + (k >> 7)];
        }
    } else if (e < kOffDxA) {                            // [w][s][lane][8]: Wp[k = cr][cd = 16 w + r], rows 16.. zero
        e -= kOffDzA8;
        const int lane = (e >> 3) & 63, s = (e >> 9) & 7, w = e >> 12;
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = r < 16 ? Wp[(16 * s + 8 * h + jj) * 128 + 16 * w + r] : 0.f;
    } else {                                             // [mt][kh][s][lane][8]: rows cr = 32 mt + r, k = 256 kh + 16 s + ..
        e -= kOffDxA;
        const int lane = (e >> 3) & 63, s = (e >> 9) & 15, kh = (e >> 13) & 1, mt = e >> 14;
        const int r = lane & 31, h = lane >> 5;
        const int cr = 32 * mt + r;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int kk = 16 * s + 8 * h + jj;          // 0..127: da channels (Wf), 128..255: dg channels (Wg)
            const float* W = kk < 128 ? Wf : Wg;
            v[jj] = W[((kk & 127) * 128 + cr) * 2 + (kh == 0 ? 1 : 0)];     // half 0 = dab[t] (tap 1), half 1 = dab[t + d] (tap 0)
        }
    }
    bf16x8 o;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) o[jj] = (bf16)v[jj];
    *reinterpret_cast<bf16x8*>(img + (long long)l * kLayerImg + (long long)g * 8) = o;
}

struct PackMatArgs { const float* src[kMaxProb16]; };
// mode 0: dst[m][l * kc + c] = src_l[m * kc + c]        (dst is [M][L kc])
// mode 1: dst[l * kc + c][m] = src_l[m * kc + c]        (dst is [L kc][M])
__global__ void k16_pack_mat(PackMatArgs a, bf16* __restrict__ dst, int L, int M, int kc, int mode) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)L * M * kc / 8;
    if (g >= total) return;
    const long long e = g * 8;
    bf16x8 o;
    if (mode == 0) {
        const int K = L * kc;
        const int m = (int)(e / K), k = (int)(e - (long long)m * K);
        const int l = k / kc, c = k - l * kc;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) o[jj] = (bf16)a.src[l][(long long)m * kc + c + jj];
    } else {
        const int row = (int)(e / M), m = (int)(e - (long long)row * M);
        const int l = row / kc, c = row - l * kc;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) o[jj] = (bf16)a.src[l][(long long)(m + jj) * kc + c];
    }
    *reinterpret_cast<bf16x8*>(dst + e) = o;
}

__global__ void k16_cvt_f2b(const float* __restrict__ src, bf16* __restrict__ dst, long long n8) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n8) return;
    const float4 a = reinterpret_cast<const float4*>(src)[2 * g], b = reinterpret_cast<const float4*>(src)[2 * g + 1];
    bf16x8 o;
    o[0] = (bf16)a.x; o[1] = (bf16)a.y; o[2] = (bf16)a.z; o[3] = (bf16)a.w;
    o[4] = (bf16)b.x; o[5] = (bf16)b.y; o[6] = (bf16)b.z; o[7] = (bf16)b.w;
    reinterpret_cast<bf16x8*>(dst)[g] = o;
}
__global__ void k16_cvt_b2f(const bf16* __restrict__ src, float* __restrict__ dst, long long n8) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n8) return;
    const bf16x8 v = reinterpret_cast<const bf16x8*>(src)[g];
    reinterpret_cast<float4*>(dst)[2 * g] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    reinterpret_cast<float4*>(dst)[2 * g + 1] = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
}

// first causal layer on tokens (data.py:61-68 + wavenet.py:298-301, fw = 2): out[b,t,:] = W[:, idx[t-1], 0] + W[:, idx[t], 1]
__global__ void k16_embed_fwd(const int32_t* __restrict__ idx, const float* __restrict__ W, const float* __restrict__ bias,
                              bf16* __restrict__ out, long long N, int T, int Q, int C) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = C / 8;
    if (g >= N * cg) return;
    const long long n = g / cg;
    const int c0 = (int)(g - n * cg) * 8;
    const int t = (int)(n % T);
    const int q1 = idx[n];
    const int q0 = t > 0 ? idx[n - 1] : -1;
    bf16x8 o;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int c = c0 + jj;
        float v = W[((long long)c * Q + q1) * 2 + 1] + (bias ? bias[c] : 0.f);
        if (q0 >= 0) v += W[((long long)c * Q + q0) * 2];
        o[jj] = (bf16)v;
    }
    *reinterpret_cast<bf16x8*>(out + n * C + c0) = o;
}

// The same gather with the table of a 64-channel slice staged in LDS as [tap][token value][channel] fp32 (2 x Q x 64 x 4 B
// = 128 KB at Q = 256): the form above reads W[c][q][tap] for 8 channels of one thread 2 KB apart, 16 scattered dwords per
// 16 bytes written (67 us at config 5 for 33 MB of output).  Here lane = channel while the slice is loaded (conflict-free
// LDS stores; the 8-byte global reads of a wave hit 64 lines that stay in L1 for the next 15 token values), then a
// thread adds two 32-byte rows per token and writes 16 bytes; a block covers kEmbTok tokens so that the table load
// (128 KB from L2) is paid once per 128 KB of output.
static constexpr int kEmbTok = 1024;
__global__ __launch_bounds__(256) void k16_embed_fwd_lds(const int32_t* __restrict__ idx, const float* __restrict__ W,
                                                         const float* __restrict__ bias, bf16* __restrict__ out,
                                                         long long N, int T, int Q, int C) {
    extern __shared__ __attribute__((aligned(16))) float etab[];          // [2][Q][64]
    const int c0 = blockIdx.y * 64;
    {
        const int c = threadIdx.x & 63;
        const float bc = bias ? bias[c0 + c] : 0.f;
        const float2* wc = reinterpret_cast<const float2*>(W + (long long)(c0 + c) * Q * 2);
        for (int q = threadIdx.x >> 6; q < Q; q += 4) {
            const float2 v = wc[q];
            etab[q * 64 + c] = v.x;                                        // tap 0: the token one step back
            etab[(Q + q) * 64 + c] = v.y + bc;                             // tap 1 (+ bias, added once per output)
        }
    }
    __syncthreads();
    const long long n0 = (long long)blockIdx.x * kEmbTok;
    const int g = threadIdx.x & 7;                                         // 8 channels each
    for (int i = threadIdx.x >> 3; i < kEmbTok; i += 32) {
        const long long n = n0 + i;
        if (n >= N) break;
        const int t = (int)(n % T);
        const int q1 = idx[n];
        const int q0 = t > 0 ? idx[n - 1] : -1;
        const float4* r1 = reinterpret_cast<const float4*>(etab + (Q + q1) * 64 + 8 * g);
        float4 lo = r1[0], hi = r1[1];
        if (q0 >= 0) {
            const float4* r0 = reinterpret_cast<const float4*>(etab + q0 * 64 + 8 * g);
            const float4 a = r0[0], b = r0[1];
            lo.x += a.x; lo.y += a.y; lo.z += a.z; lo.w += a.w;
            hi.x += b.x; hi.y += b.y; hi.z += b.z; hi.w += b.w;
        }
        bf16x8 o;
        o[0] = (bf16)lo.x; o[1] = (bf16)lo.y; o[2] = (bf16)lo.z; o[3] = (bf16)lo.w;
        o[4] = (bf16)hi.x; o[5] = (bf16)hi.y; o[6] = (bf16)hi.z; o[7] = (bf16)hi.w;
        *reinterpret_cast<bf16x8*>(out + n * C + c0 + 8 * g) = o;
    }
}

// =============================================================================================
// k16_embed_bwd: gradient of the first causal layer's table (data.py:61-68 + wavenet.py:298-301, fw = 2) on the matrix
// cores.  dW[c][q][tap] = sum_n dx[n][c] * [token(n - (1 - tap)) == q] is a contraction over time of a one-hot matrix
// (256 x n) with dx (n x 128): the A operand of lane (q, k-half) is GENERATED from eight tokens (compare + select, no
// memory), the B operand is dx read transposed out of LDS exactly as k16_wgrad reads it.  17 GFLOP of bf16 MFMA (7 us of
// matrix time) replace 33.5 M LDS float atomics behind scattered loads (0.19 ms + a bf16 -> fp32 conversion pass).
// Tokens come from two padded planes ([2][B][Tp]: the token one step back / the current token of every row, -1 for the
// first row's missing predecessor and for rows past the end of the clip, which match no q), so every load is
// unconditional and 16-byte aligned.  Wave w: tap w >> 2, token values 64 (w & 3) .. + 63, all
// 128 channels: 8 accumulators.  A workgroup's partial table leaves in [tap][q][c] order (coalesced); the reduce kernel
// sums the partials in a fixed order, transposes into W's [c][q][tap] and takes the bias gradient as the sum over q of
// tap 1 (every sample has exactly one current token).
// =============================================================================================
static constexpr int kEbRows = 64;
static constexpr int kEbLds = 2 * kEbRows * 256;         // two dx stages

__global__ void k16_embed_pad_tokens(const int32_t* __restrict__ idx, int32_t* __restrict__ tokp, int B, int T, int Tp) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * Tp) return;
    const int b = (int)(i / Tp), t = (int)(i - (long long)b * Tp);
    // plane 0: the token one step back (tap 0), plane 1: the current token (tap 1); -1 where the row has no such token
    tokp[i] = (t >= 1 && t < T) ? idx[(long long)b * T + t - 1] : -1;
    tokp[(long long)B * Tp + i] = t < T ? idx[(long long)b * T + t] : -1;
}

__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k16_embed_bwd(const int32_t* __restrict__ tokp, const bf16* __restrict__ dx, float* __restrict__ part, int B, int T,
                   int Tp) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int tap = w >> 2, qw = 64 * (w & 3);
    const int cpb = (T + kEbRows - 1) / kEbRows;
    const int nch = B * cpb;
    const int c_begin = (int)((long long)nch * blockIdx.x / gridDim.x);
    const int c_end = (int)((long long)nch * (blockIdx.x + 1) / gridDim.x);
    auto tile = [&](int buf) { return lds + buf * (kEbRows * 256); };
    auto issue = [&](int c, int buf) {
        const int b = c / cpb;
        const int r0 = (c - b * cpb) * kEbRows;
        const bf16* xb = dx + (long long)b * T * 128;
        const int hi = T - 1;
        dma_pieces(tile(buf), lane, w, 8, 2, [&](int r) {
            const int t = r0 + r < hi ? r0 + r : hi;                 // ragged end: a valid row, its tokens are -1
            return xb + (long long)t * 128;
        });
    };
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 tk[4][2], tkn[4][2];
    auto load_tok = [&](int c, i32x4 (&dst)[4][2]) {
        const int b = c / cpb;
        const int r0 = (c - b * cpb) * kEbRows;
        const int32_t* tp = tokp + ((long long)tap * B + b) * Tp + r0 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            dst[ks][0] = *reinterpret_cast<const i32x4*>(tp + 16 * ks);
            dst[ks][1] = *reinterpret_cast<const i32x4*>(tp + 16 * ks + 4);
        }
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    if (c_begin < c_end) {
        issue(c_begin, 0);
        load_tok(c_begin, tkn);
    }
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { tk[ks][0] = tkn[ks][0]; tk[ks][1] = tkn[ks][1]; }
        wait_vm<0>();
        barrier();
        if (c + 1 < c_end) issue(c + 1, buf ^ 1);
        load_tok(c + 1 < c_end ? c + 1 : c, tkn);
        const char* bt = tile(buf);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 bv[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bv[ni] = frag_tr(bt, 16 * ks, 32 * ni, lane);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int q = qw + 32 * mi + j;
                u32x4 a;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t0 = e < 2 ? tk[ks][0][2 * e] : tk[ks][1][2 * e - 4];
                    const int t1 = e < 2 ? tk[ks][0][2 * e + 1] : tk[ks][1][2 * e - 3];
                    a[e] = (t0 == q ? 0x3F80u : 0u) | (t1 == q ? 0x3F800000u : 0u);     // bf16 1.0 in the low / high half
                }
                const bf16x8 av = __builtin_bit_cast(bf16x8, a);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv[ni], acc[mi][ni], 0, 0, 0);
            }
        }
    }
    float* o = part + (long long)blockIdx.x * (2 * 256 * 128) + (long long)tap * (256 * 128);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                o[(qw + 32 * mi + acc_row(r, h)) * 128 + 32 * ni + j] = acc[mi][ni][r];
}

// dW[c][q][tap] += sum over workgroups (in index order) of part[wg][tap][q][c]; the tap-1 sums also go to bsum[q][c],
// from which k16_embed_bias takes dbias[c] += sum over q (every sample has exactly one current token) -- no atomics.
__global__ __launch_bounds__(256) void k16_embed_bwd_reduce(const float* __restrict__ part, int nwg, int C,
                                                            float* __restrict__ dW, float* __restrict__ bsum) {
    // A block owns 64 consecutive (tap, q, c) values; its four waves each add a quarter of the workgroups' tables and the
    // quarters meet in a fixed order through LDS (64 blocks of 256 single threads walking all nwg tables took 19 us for
    // 8 MB: 64 dependent rounds of loads on a quarter of the CUs).
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, sp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;                    // ((tap * 256 + q) * C + c)
    const long long wgs = 2ll * 256 * C;
    const float* p = part + e;
    const int per = (nwg + 3) / 4;
    const int g0 = sp * per, g1 = g0 + per < nwg ? g0 + per : nwg;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = g0;
    for (; g + 4 <= g1; g += 4) {
        s0 += p[g * wgs]; s1 += p[(g + 1) * wgs];
        s2 += p[(g + 2) * wgs]; s3 += p[(g + 3) * wgs];
    }
    for (; g < g1; ++g) s0 += p[g * wgs];
    red[sp][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sp != 0) return;
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int c = e % C, tq = e / C, q = tq & 255, tap = tq >> 8;
    dW[((long long)c * 256 + q) * 2 + tap] += v;
    if (bsum && tap == 1) bsum[q * C + c] = v;
}
__global__ void k16_embed_bias(const float* __restrict__ bsum, int C, float* __restrict__ dbias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int q = 0; q < 256; ++q) s += bsum[q * C + c];
    dbias[c] += s;
}

// The same contraction for an fp32 gradient (the fp32-storage path: 32 NC channels): dx is split into three bf16 parts
// (h + m + l = x to 2^-24) on its way into LDS, the one-hot operand is exact, so the three MFMAs per tile reproduce the
// fp32 sum to rounding -- and, unlike per-block tables filled with LDS float atomics, in a fixed order.
template <int NC>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k16_embed_bwd_f32(const int32_t* __restrict__ tokp, const float* __restrict__ dx, float* __restrict__ part, int B,
                       int T, int Tp) {
    constexpr int C = 32 * NC;
    __shared__ __attribute__((aligned(1024))) char lds[3 * kEbRows * 256];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int tap = w >> 2, qw = 64 * (w & 3);
    const int cpb = (T + kEbRows - 1) / kEbRows;
    const int nch = B * cpb;
    const int c_begin = (int)((long long)nch * blockIdx.x / gridDim.x);
    const int c_end = (int)((long long)nch * (blockIdx.x + 1) / gridDim.x);
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    float4 xv[NC];
    i32x4 tkn[4][2];
    auto fetch = [&](int c) {
        const int b = c / cpb;
        const int r0 = (c - b * cpb) * kEbRows;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int i = threadIdx.x + 512 * k;
            const int row = i / (C / 4), c4 = i - row * (C / 4);
            const int t = r0 + row < T - 1 ? r0 + row : T - 1;      // ragged end: a valid row, its tokens are -1
            xv[k] = *reinterpret_cast<const float4*>(dx + ((long long)b * T + t) * C + 4 * c4);
        }
        const int32_t* tp = tokp + ((long long)tap * B + b) * Tp + r0 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            tkn[ks][0] = *reinterpret_cast<const i32x4*>(tp + 16 * ks);
            tkn[ks][1] = *reinterpret_cast<const i32x4*>(tp + 16 * ks + 4);
        }
    };
    f32x16 acc[2][NC];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NC; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    if (c_begin < c_end) fetch(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        i32x4 tk[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { tk[ks][0] = tkn[ks][0]; tk[ks][1] = tkn[ks][1]; }
#pragma unroll
        for (int k = 0; k < NC; ++k) {                   // split this stage's values into the three planes
            const int i = threadIdx.x + 512 * k;
            const int row = i / (C / 4), c4 = i - row * (C / 4);
            const float v[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
            bf16x4 ph, pm, pl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bf16 hh = (bf16)v[e];
                const float r1 = v[e] - (float)hh;
                const bf16 mm = (bf16)r1;
                ph[e] = hh; pm[e] = mm; pl[e] = (bf16)(r1 - (float)mm);
            }
            const int o = toff(row, (4 * c4) >> 3) + 2 * ((4 * c4) & 7);
            *reinterpret_cast<bf16x4*>(lds + o) = ph;
            *reinterpret_cast<bf16x4*>(lds + kEbRows * 256 + o) = pm;
            *reinterpret_cast<bf16x4*>(lds + 2 * kEbRows * 256 + o) = pl;
        }
        __syncthreads();
        fetch(c + 1 < c_end ? c + 1 : c);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 bv[3][NC];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int ni = 0; ni < NC; ++ni) bv[pl][ni] = frag_tr(lds + pl * kEbRows * 256, 16 * ks, 32 * ni, lane);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int q = qw + 32 * mi + j;
                u32x4 a;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t0 = e < 2 ? tk[ks][0][2 * e] : tk[ks][1][2 * e - 4];
                    const int t1 = e < 2 ? tk[ks][0][2 * e + 1] : tk[ks][1][2 * e - 3];
                    a[e] = (t0 == q ? 0x3F80u : 0u) | (t1 == q ? 0x3F800000u : 0u);
                }
                const bf16x8 av = __builtin_bit_cast(bf16x8, a);
#pragma unroll
                for (int ni = 0; ni < NC; ++ni) {            // smallest parts first
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv[2][ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv[1][ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv[0][ni], acc[mi][ni], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    float* o = part + (long long)blockIdx.x * (2 * 256 * C) + (long long)tap * (256 * C);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NC; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                o[(qw + 32 * mi + acc_row(r, h)) * C + 32 * ni + j] = acc[mi][ni][r];
}

int pack_layers(int L, const float* const* Wf, const float* const* Wg, const float* const* Wp, bf16* img, hipStream_t s) {
    if (L > kMaxProb16) { wn::set_error("w16: more than %d layers", kMaxProb16); return WN_ESHAPE; }
    PackLayersArgs a{};
    for (int l = 0; l < L; ++l) { a.Wf[l] = Wf[l]; a.Wg[l] = Wg[l]; a.Wp[l] = Wp[l]; }
    hipLaunchKernelGGL(k16_pack_layers, dim3((kLayerImg / 8 + 255) / 256, L), dim3(256), 0, s, a, img);
    WN_LAUNCH_CHECK();
    return WN_OK;
}
int pack_mat(int L, const float* const* src, bf16* dst, int M, int kc, int mode, hipStream_t s) {
    if (L > kMaxProb16) { wn::set_error("w16: more than %d sources", kMaxProb16); return WN_ESHAPE; }
    PackMatArgs a{};
    for (int l = 0; l < L; ++l) a.src[l] = src[l];
    const long long total = (long long)L * M * kc / 8;
    hipLaunchKernelGGL(k16_pack_mat, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, dst, L, M, kc, mode);
    WN_LAUNCH_CHECK();
    return WN_OK;
}
int cvt_f2b(const float* src, bf16* dst, long long n, hipStream_t s) {
    if (n % 8) { wn::set_error("w16 cvt: n %% 8 != 0"); return WN_ESHAPE; }
    hipLaunchKernelGGL(k16_cvt_f2b, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, s, src, dst, n / 8);
    WN_LAUNCH_CHECK();
    return WN_OK;
}
int cvt_b2f(const bf16* src, float* dst, long long n, hipStream_t s) {
    if (n % 8) { wn::set_error("w16 cvt: n %% 8 != 0"); return WN_ESHAPE; }
    hipLaunchKernelGGL(k16_cvt_b2f, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, s, src, dst, n / 8);
    WN_LAUNCH_CHECK();
    return WN_OK;
}
int embed_fwd16(const int32_t* idx, const float* W, const float* bias, bf16* out, int B, int T, int Q, int C,
                hipStream_t s) {
    const long long N = (long long)B * T;
    const size_t lds = (size_t)2 * Q * 64 * sizeof(float);
    if (C % 64 == 0 && lds <= 144 * 1024 && N >= 8 * kEmbTok) {
        static bool attr = false;
        if (!attr) {
            WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k16_embed_fwd_lds),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
            attr = true;
        }
        hipLaunchKernelGGL(k16_embed_fwd_lds, dim3((unsigned)((N + kEmbTok - 1) / kEmbTok), C / 64), dim3(256), lds, s, idx, W,
                           bias, out, N, T, Q, C);
        WN_LAUNCH_CHECK();
        return WN_OK;
    }
    const long long n = N * (C / 8);
    hipLaunchKernelGGL(k16_embed_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, idx, W, bias, out, N, T, Q, C);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

static size_t embed_tok_bytes(int B, int T) {
    const int Tp = (T + kEbRows - 1) / kEbRows * kEbRows;
    return ((size_t)2 * B * Tp * sizeof(int32_t) + 255) & ~(size_t)255;
}
size_t embed_bwd_ws_bytes(int B, int T, int C) {
    return embed_tok_bytes(B, T) + (size_t)256 * 2 * 256 * C * sizeof(float) + (size_t)256 * C * sizeof(float) + 256;
}
size_t embed_bwd16_ws_bytes(int B, int T) { return embed_bwd_ws_bytes(B, T, 128); }

// dx: bf16 (B,T,128) when dx_f32 == NULL, else fp32 (B,T,C) with C = 32, 64, 96 or 128
int embed_bwd_mfma(const int32_t* idx, const bf16* dx, const float* dx_f32, int C, float* dW, float* dbias, int B, int T,
                   void* ws, hipStream_t s) {
    const int Tp = (T + kEbRows - 1) / kEbRows * kEbRows;
    int32_t* tokp = reinterpret_cast<int32_t*>(ws);
    float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + embed_tok_bytes(B, T));
    const long long np = (long long)B * Tp;
    hipLaunchKernelGGL(k16_embed_pad_tokens, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, idx, tokp, B, T, Tp);
    const int nch = B * ((T + kEbRows - 1) / kEbRows);
    const int nwg = nch < 256 ? nch : 256;
    float* bsum = dbias ? part + (size_t)nwg * 2 * 256 * C : nullptr;
    if (!dx_f32) {
        static bool attr = false;
        if (!attr) {
            WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k16_embed_bwd),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kEbLds));
            attr = true;
        }
        hipLaunchKernelGGL(k16_embed_bwd, dim3(nwg), dim3(512), kEbLds, s, tokp, dx, part, B, T, Tp);
    } else if (C == 32) {
        hipLaunchKernelGGL(k16_embed_bwd_f32<1>, dim3(nwg), dim3(512), 0, s, tokp, dx_f32, part, B, T, Tp);
    } else if (C == 64) {
        hipLaunchKernelGGL(k16_embed_bwd_f32<2>, dim3(nwg), dim3(512), 0, s, tokp, dx_f32, part, B, T, Tp);
    } else if (C == 128) {
        hipLaunchKernelGGL(k16_embed_bwd_f32<4>, dim3(nwg), dim3(512), 0, s, tokp, dx_f32, part, B, T, Tp);
    } else {
        wn::set_error("embed_bwd_mfma: %d channels", C);
        return WN_ESHAPE;
    }
    hipLaunchKernelGGL(k16_embed_bwd_reduce, dim3(2 * 256 * C / 64), dim3(256), 0, s, part, nwg, C, dW, bsum);   // 64 values per block
    if (dbias) hipLaunchKernelGGL(k16_embed_bias, dim3((C + 63) / 64), dim3(64), 0, s, bsum, C, dbias);
    WN_LAUNCH_CHECK();
    return WN_OK;
}
int embed_bwd16(const int32_t* idx, const bf16* dx, float* dW, float* dbias, int B, int T, void* ws, hipStream_t s) {
    return embed_bwd_mfma(idx, dx, nullptr, 128, dW, dbias, B, T, ws, s);
}

}  // namespace w16
