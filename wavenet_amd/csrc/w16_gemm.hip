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
            if (r < nrows) *reinterpret_cast<u32x4*>(ob + (long long)r * a.ldo + c * 8) = v;
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
        for (int i = 0; i < a.nsrc && plain; ++i) plain = a.shift[i] == 0;
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
    const int r0 = (nblk - b * bpb) * 256;
    const int m0 = mblk * 256;
    const int spk = a.ksrc >> 6;                         // stages per source
    const int nst = a.nsrc * spk;
    const int lr = lane >> 3, lp = lane & 7;

    auto issue = [&](int st, int buf) {
        const int src = st / spk;
        const int k0 = (st - src * spk) * 64;
        const bf16* xb = a.X[src] + ((long long)b * a.x_rows_per_b) * a.ldx + k0;
        const int sh = a.x_row0 + r0;
        const int hi = a.x_rows_per_b - 1;
        const bf16* wb = a.W + (long long)m0 * a.K + st * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = w + 8 * i;
            const int r = 8 * p + lr;
            const int c = lp ^ ((r >> 1) & 7);
            int t = sh + r;
            t = t > hi ? hi : t;
            W16_DMA16(xb + (long long)t * a.ldx + c * 8, xt(buf) + p * 1024);
            W16_DMA16(wb + (long long)r * a.K + c * 8, wt(buf) + p * 1024);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.f;
    issue(0, 0);
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        wait_vm<0>();
        barrier();
        if (st + 1 < nst) issue(st + 1, buf ^ 1);
        const char* xb = xt(buf);
        const char* wb = wt(buf);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 bv[4], av[2];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bv[nt] = *reinterpret_cast<const bf16x8*>(xb + boff(128 * wn + 32 * nt + j, 2 * s + h));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) av[mt] = *reinterpret_cast<const bf16x8*>(wb + boff(64 * wm + 32 * mt + j, 2 * s + h));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mt], bv[nt], acc[nt][mt], 0, 0, 0);
        }
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
            if (r < nrows) *reinterpret_cast<u32x4*>(o + (long long)r * a.ldo + c * 8) = v;
        }
    }
}

static int launch_cgemm256(CG16& a, hipStream_t s) {
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
    const int p = blockIdx.y;
    const WG16Prob& pr = a.prob[p];
    // this workgroup's slab of 64-row chunks
    const int cpb = (a.R + kWT - 1) / kWT;
    const int nch = a.nB * cpb;
    const int c_begin = (int)((long long)nch * blockIdx.x / gridDim.x);
    const int c_end = (int)((long long)nch * (blockIdx.x + 1) / gridDim.x);
    auto tile = [&](int buf, int which) { return lds + (buf * 4 + which) * kWTileB; };   // which: A0 A1 B0 B1

    auto issue = [&](int c, int buf) {
        const int b = c / cpb;
        const int r0 = (c - b * cpb) * kWT;
        const bf16* ab = pr.A + ((long long)b * a.a_rpb + a.a_r0) * a.lda;
        const int ahi = a.R - 1;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
            dma_pieces(tile(buf, mh), lane, w, 8, 2, [&](int r) {
                const int t = r0 + r < ahi ? r0 + r : ahi;
                return ab + (long long)t * a.lda + 128 * mh;
            });
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            const bf16* bb = pr.Bh[nh] + (long long)b * a.b_rpb * a.ldb;
            const int sh = a.b_r0 + r0 + pr.shift[nh];
            const int bhi = a.b_rpb - 1;
            dma_pieces(tile(buf, 2 + nh), lane, w, 8, 2, [&](int r) {
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

    if (c_begin < c_end) issue(c_begin, 0);
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        wait_vm<0>();
        barrier();
        if (c + 1 < c_end) issue(c + 1, buf ^ 1);
        {
            const int b = c / cpb;
            const int r0 = (c - b * cpb) * kWT;
            bool fix = r0 + kWT > a.R;
            int sh[2];
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
                sh[nh] = a.b_r0 + r0 + pr.shift[nh];
                fix = fix || sh[nh] < 0 || sh[nh] + kWT > a.b_rpb;
            }
            if (fix) {                                   // rows that do not exist contribute nothing
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
        }
        const char* at = tile(buf, wm >> 1);
        const char* bt = tile(buf, 2 + wn);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 av[2], bv[4];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) av[mi] = frag_tr(at, 16 * ks, 64 * (wm & 1) + 32 * mi, lane);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                bv[ni] = frag_tr(bt, 16 * ks, 32 * ni, lane);
                if (RELU_B) bv[ni] = relu8(bv[ni]);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
        }
    }
    if (c_begin >= c_end) return;
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

int launch_wgrad16(const WG16& a, int nprob, hipStream_t s) {
    if (nprob < 1 || nprob > kMaxProb16) { wn::set_error("w16 wgrad: %d problems", nprob); return WN_EARG; }
    const int cpb = (a.R + kWT - 1) / kWT;
    const int nch = a.nB * cpb;
    int slabs = 256 / nprob;                             // every workgroup resident at once: slabs stream in step
    if (slabs < 1) slabs = 1;
    if (slabs > nch) slabs = nch;
#define WGL(RB)                                                                                                   \
    do {                                                                                                          \
        static bool attr = false;                                                                                 \
        if (!attr) {                                                                                              \
            WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k16_wgrad<RB>),                              \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kWgLds));                      \
            attr = true;                                                                                          \
        }                                                                                                         \
        hipLaunchKernelGGL((k16_wgrad<RB>), dim3(slabs, nprob), dim3(512), kWgLds, s, a);                         \
    } while (0)
    if (a.relu_b) WGL(true); else WGL(false);
#undef WGL
    WN_LAUNCH_CHECK();
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
    const long long n = (long long)B * T * (C / 8);
    hipLaunchKernelGGL(k16_embed_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, idx, W, bias, out,
                       (long long)B * T, T, Q, C);
    WN_LAUNCH_CHECK();
    return WN_OK;
}

}  // namespace w16
