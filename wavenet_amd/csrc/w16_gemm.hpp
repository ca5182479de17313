// Host-side descriptors and launchers of the bf16-storage path (w16_*.hip).
#pragma once
#include "w16.hpp"

namespace w16 {

static constexpr int kMaxSrc16 = 64;
static constexpr int kMaxProb16 = 48;

// out[n][m] = sum_src sum_k W[m][src * ksrc + k] * X_src[row(n) + shift[src]][k]   (rows outside the clip read as 0)
struct CG16 {
    const bf16* X[kMaxSrc16];
    int shift[kMaxSrc16];
    int nsrc, ksrc;                  // sources, channels per source (multiple of 128)
    int ldx;                         // row stride of every source (elements)
    int x_rows_per_b, x_row0;        // source row of output row r of clip b: b * x_rows_per_b + x_row0 + r + shift
    const bf16* W;                   // [M][K] row-major bf16, K = nsrc * ksrc
    int M, K;
    int B, rows_per_b;               // output rows: B clips of rows_per_b
    void* out; int out_f32; int ldo; // output row stride (elements)
    long long ob_stride;             // m-block i writes at out + i * ob_stride + row * ldo + i * ob_col
    int ob_col;
    const float* bias;               // [M] or NULL
    int ep;                          // 0 none, 1: out += extra, 2: out = extra > 0 ? out : 0
    const bf16* extra; int lde;      // [rows][..], same rows as out, column offset i * ob_col
    int relu_x;                      // relu applied to X on its way to the matrix core
    int blocks_per_b, n_blocks;      // filled by launch_cgemm
    long long x_src_stride;          // filled by launch_cgemm: X[i] = X[0] + i * x_src_stride (the 256-block kernel)
};
int launch_cgemm(CG16& a, hipStream_t s);

struct WG16Prob {
    const bf16* A;                   // [rows][lda]: 256 channels starting at this pointer
    const bf16* Bh[2];               // two 128-channel operands [rows][ldb]
    int shift[2];                    // row shift of each B half (dilated taps); rows outside the clip read as 0
    float* out[2][2];                // [m half][n half]: element (m, n) at out + m * os_m + n * os_n; NULL = skip
    int r_lo;                        // rows below r_lo (a multiple of 64) are not read: A is zero there (dead columns of a layer)
    unsigned short nslab;            // filled by launch_wgrad16: workgroups (time slabs) of this problem, ~ its share of the rows
    unsigned short slab0;            // filled by launch_wgrad16: index of its first partial block
};
// dW[m][n] += sum over clips b and rows r < R of A[b * a_rpb + a_r0 + r][m] * B[b * b_rpb + b_r0 + r + shift][n]
struct WG16 {
    WG16Prob prob[kMaxProb16];
    int lda, ldb;
    int nB, R, a_rpb, a_r0, b_rpb, b_r0;
    int os_m, os_n;
    int relu_b;
    int n_items, n_prob;             // filled by launch_wgrad16: (slab, problem) pairs of the launch, problems
    unsigned short level_cnt[64];    // filled by launch_wgrad16: problems that have more than s slabs (at most 64 slabs per problem)
    float* part;                     // kWgPartBytes of scratch: every workgroup leaves its 256 x 256 block there and a second
                                     // kernel adds the slabs of a problem in a fixed order (bit-reproducible); NULL: each
                                     // workgroup adds its block to dW with float atomics (order of the slabs not defined)
};
static constexpr size_t kWgPartBytes = (size_t)256 * 256 * 256 * sizeof(float);   // at most 256 workgroups per launch
int launch_wgrad16(const WG16& a, int nprob, hipStream_t s);

int pack_layers(int L, const float* const* Wf, const float* const* Wg, const float* const* Wp, bf16* img, hipStream_t s);
int pack_mat(int L, const float* const* src, bf16* dst, int M, int kc, int mode, hipStream_t s);
int cvt_f2b(const float* src, bf16* dst, long long n, hipStream_t s);
int cvt_b2f(const bf16* src, float* dst, long long n, hipStream_t s);
size_t embed_bwd16_ws_bytes(int B, int T);
size_t embed_bwd_ws_bytes(int B, int T, int C);
int embed_bwd_mfma(const int32_t* idx, const bf16* dx, const float* dx_f32, int C, float* dW, float* dbias, int B, int T,
                   void* ws, hipStream_t s);
int embed_bwd16(const int32_t* idx, const bf16* dx, float* dW, float* dbias, int B, int T, void* ws, hipStream_t s);
int embed_fwd16(const int32_t* idx, const float* W, const float* bias, bf16* out, int B, int T, int Q, int C, hipStream_t s);

// w16_layer.hip
int fwd_layer(const bf16* x, const bf16* img, bf16* out, bf16* z, int B, int T, int d, int Z, hipStream_t s);
// live ranges (multiples of 32; GateP / DxP in w16_layer.hip): columns below t_live receive no gradient; gate tiles below t_zero
// and (unless zero_dead) dx tiles below t_live are not touched at all; [da | dg](t) and dout(t) read as zero below t_gate
int gate_bwd_layer(const bf16* x, const bf16* img, const bf16* dout, const bf16* dzs, int dz_t0, bf16* dadg, int B, int T,
                   int d, int Z, int t_live, int t_zero, hipStream_t s);
int dx_grid(int B, int T);
int dx_layer(const bf16* dadg, const bf16* img, const bf16* dout, const bf16* zprev, bf16* dx, float* dwp_part, int B,
             int T, int d, int t_live, int t_gate, int zero_dead, hipStream_t s);
// stack layers l_hi .. l_lo (>= 1) of the layer backward in ONE launch (k16_bwd_multi): same results as gate_bwd_layer +
// dx_layer per layer, bit for bit; sync: bwd_multi_sync_words(B, T) words of device memory
size_t bwd_multi_sync_words(int B, int T);
int bwd_multi_ok(int B, int T);
int bwd_multi(const bf16* x0, const bf16* xs, const bf16* z, const bf16* img, const bf16* dzs, bf16* dadg, bf16* dxb0,
              bf16* dxb1, float* parts, long long part_stride, unsigned* sync, const int* d, const int* Z,
              const int* live_gate, const int* live_dx, const int* zero_gate, int l_hi, int l_lo, int B, int T, int dz_t0,
              hipStream_t s);
int reduce_parts(const float* part, long long layer_stride, int nwg, int n, float* const* dW_dev, int L, hipStream_t s);

}  // namespace w16
