// Shared between the exact-fp32 channel GEMM (mfma_gemm.hip) and its bf16x3 split-product variant
// (mfma_gemm_b3.hip).
#pragma once
#include "wn_kernels.hpp"

namespace wn {

static constexpr int WN_GATE_TAPS = 8;      // taps of a gate-mode launch (nsrc <= this)

struct CGArgs {
    const float* X[WN_MAX_SRC];      // per source
    const float* W[WN_MAX_SRC];      // per source (multi-source) or per problem (multi-problem)
    const float* bias[WN_MAX_SRC];   // per source, may be NULL
    float* out[WN_MAX_SRC];          // per problem
    int K[WN_MAX_SRC];               // per source; row stride of X_src is K
    int wsm[WN_MAX_SRC];             // W row (m) stride per source / problem
    int wsk;                         // W column (k) stride: 1 = row-major W[m][k], else transposed view
    int nsrc, nprob, M, ldo;
    long long N;                     // output rows
    int rows_out_per_b, rows_src_per_b, off;   // src row = b*rows_src_per_b + (n % rows_out_per_b) + off + soff[src]
    int out_rows_per_b, out_row0;    // out row of column n = b*out_rows_per_b + out_row0 + n % rows_out_per_b; 0, 0 = row n
    int soff[WN_MAX_SRC];            // per-source row shift (dilated taps); rows outside [0, rows_src_per_b) read as 0
    int act;                         // applied to X on load
    const float* gate_x; int gate_act;         // out *= act'(gate_x[n][m])   (dx of a pre-activated conv)
    const float* residual;           // out += residual[n][m] (row stride ldo), may be NULL
    int accumulate;
    // gate mode (bf16x3/bf16 kernel, mode 3): rows of W (filter) and W2 (gate) are interleaved tile by tile, the epilogue
    // writes f = tanh(a), s = sigmoid(g), z = f s (row stride M) and zeroes columns with t < gate_Z (the reference's zero
    // prefix: a = g = 0 there).  gate_f / gate_s may be NULL (inference).
    // gate-backward mode (mode 4): the GEMM result is dz (+ residual = dz_skip); the epilogue reads f = gate_f, s = gate_s
    // (row stride M) and writes da = dz s (1 - f^2) | dg = dz f s (1 - s) side by side into gate_z (row stride 2 M),
    // zero for t < gate_Z.
    const float* W2[WN_GATE_TAPS];
    const float* bias2[WN_GATE_TAPS];
    float* gate_z; float* gate_f; float* gate_s;
    int gate_Z;
    int ldx;                         // row stride of every X source when != 0 (default: K[src])
    // fused-layer mode (mode 5 = gate mode + the residual projection, 128/128 channels, one-term products): out[0] (row
    // stride ldo = 128) = Wp z + proj_bias + residual, Wp's image is the kernel's second image argument
    int h2_ok;                       // X is provably within the fp16 split's static range (z = tanh * sigmoid)
    const unsigned* xmax_dev;        // else: device word with the bits of max |X| (exec_absmax) -> dynamic power-of-two scale
    const unsigned* wmax_dev;        // fp16 split: bits of max |W| over the launch's weight tiles (launch_colgemm_b3 fills it)
    unsigned* outmax_dev;            // mode 0, k_colgemm_b3 only: atomicMax of the bits of max |out| (a step plan's word), or NULL
    const float* proj_W;             // Wp[128][128], row-major
    const float* proj_bias;
    // head + loss mode (mode 6, fp16 split, exactly 8 m-tiles = 256 outputs): the GEMM result + bias are a column's logits; the
    // epilogue takes softmax cross-entropy against xent_target[row] and writes d loss / d logits to out[0] instead of the logits
    // (row stride ldo), the workgroup's loss sum to xent_loss[kXentPart + workgroup] (wn_kernels.hpp)
    const int32_t* xent_target;
    float* xent_loss;
    long long xent_n_norm;           // > 0: rows that count; < 0: counted on the device (xent_ncnt partial counts in xent_loss)
    int xent_ncnt;
};


struct WGArgs {
    const float* A; int lda;
    const float* Bp[WN_MAX_SRC];
    const float* B2p[WN_MAX_SRC];    // optional elementwise factor on B (z = f*g), same addressing; entries may be NULL
    float* out[WN_MAX_SRC];
    int nprob, ldb, ldo;
    int osk;                         // output column stride (1; fw for a conv weight W[o][c][k]); 0 means 1
    int nB, rows_A_per_b, rows_B_per_b, off;   // B row = b*rows_B_per_b + r + off for A row b*rows_A_per_b + r
    int act;
    int rows_per_wg, wgs_per_b;
    // wide block only: per-problem extra row shift of B, and a second output for the rows m >= m_split of A
    // (A = [da | dg]: rows below m_split accumulate into out[p], the others into out2[p] at row m - m_split)
    int offp[WN_MAX_SRC];
    float* out2[WN_MAX_SRC];
    int m_split;
    float* part;                     // wide block: per-workgroup partial tiles go here (plain stores), a second kernel sums them
    // fp16 split (WN_GEMM_FP16X2): amax_dev = bits of max |A| (required); bmax_dev = bits of max |B| or NULL when B is
    // provably in [-1, 1] (z)
    const unsigned* amax_dev; const unsigned* bmax_dev; int h2;
    // wide block, six-term products only: colsum != NULL asks for colsum[m] += sum over rows of A[.][m] (the bias gradient of a
    // 1x1 convolution whose weight gradient this launch is) from the A values the kernel loads anyway; launch_wgrad_b3w sets
    // colsum_done when it took the request (other kernels ignore it), colsum_part is its scratch
    float* colsum; float* colsum_part; int colsum_done;
};

// 1 unless WAVENET_HIP_GEMM=fp32: contractions use three-way bf16 splits (6 bf16 MFMAs per product term)
bool gemm_b3_enabled();      // the current call's precision is bf16x3 or bf16
// mode 0: multi-source, one output; mode 2: nprob problems of 32 rows sharing X.  Returns WN_ESHAPE when
// the shape is not covered (the caller then uses the exact-fp32 kernel).
int launch_colgemm_b3(CGArgs& a, int mode, int nprob, hipStream_t s);

// dW_p[m*ldo + k*osk] += sum_n A[n][m] * act(B_p[row(n)][k]) (* B2_p);  M rows; picks bf16x3 or exact fp32
int launch_wgrad(WGArgs& a, int M, hipStream_t s);
// wide bf16x3 block (M == 256, nprob >= 8)
int launch_wgrad_b3w(WGArgs& a, hipStream_t s);
// one channel GEMM launch (multi-source form); picks bf16x3 or exact fp32
int launch_colgemm_multi(CGArgs& a, hipStream_t s);

// bf16x3 form of k_wgrad_mfma (same grid / arguments)
int launch_wgrad_b3(const WGArgs& a, int mt, dim3 grid, hipStream_t s);

}  // namespace wn
