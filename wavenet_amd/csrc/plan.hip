// The step plan (ABI 5): the preparation work of a training step that depends on nothing but the WEIGHTS -- the split
// weight images of every channel GEMM and their range words, the fp16 x 2 images of the fused layer kernels -- and the
// words / arrays a step needs zeroed, hoisted out of the entry points into TWO launches at the start of the step
// (wn_plan_prepare) instead of ~16 launch-floor kernels spread over it (per GEMM: zero a word, measure max |W|, split;
// k_layer_pack_h2; k_chain_zero_sync; the range pass over dskip; the fill of the gradient arena).  round 6, VERDICT r5 #5.
//
// How a plan comes to know the work: it RECORDS it.  In the recording state every entry point that carries the plan
// (WnExec.plan) runs exactly as without one and, where it prepares something that only depends on weights, registers the
// job (the launcher's own arguments: nothing is described twice).  wn_plan_finish lays the images out in the plan's device
// memory (the caller's: the library allocates nothing) and uploads the job table.  In the READY state the same entry points
// find their image by key (first weight pointer, mode, tile counts) and launch nothing of their own; what a plan does not
// hold is prepared by the entry point as before.  The contract is the caller's: a call that carries a READY plan asserts
// that wn_plan_prepare ran on the same stream after the weights last changed (TrainStepGraph: first node of the graph).
#include <new>
#include <vector>

#include "layer_pack.hpp"
#include "split_w.hpp"
#include "wn_kernels.hpp"

namespace wn {

static constexpr int kPlanMaxSplit = 8, kPlanMaxZero = 4;
enum { kPlanIdle = 0, kPlanRecord = 1, kPlanReady = 2 };

struct SplitJob {                                  // what split_w_tile reads of a CGArgs, + where the image goes
    const float* W[WN_MAX_SRC];
    const float* W2[WN_GATE_TAPS];
    int wsm[WN_MAX_SRC];
    int wsk, mode, mtiles, cps, one, ntiles;
    __bf16* img;
    unsigned* wmax;                                // one == 3: receives the bits of max |W| (what the GEMM kernel unscales by)
    float* partial;                                // one == 3: ntiles per-tile maxima (phase 1 -> phase 2)
};
struct ZeroJob { unsigned* p; int n; int second_ones; };
struct PlanDev {
    int nsplit, nzero, pack_L, pad0;
    int b1_split[kPlanMaxSplit + 1];               // phase 1: first block of split job i (all jobs: maxima, or the split itself)
    int b1_pack, b1_zero, b1_user, b1_total;
    int b2_split[kPlanMaxSplit + 1];               // phase 2: jobs with one == 3 only (the others have no blocks)
    SplitJob split[kPlanMaxSplit];
    ZeroJob zero[kPlanMaxZero];
    PackH2Args pack;
    char* pack_img;
};

struct StepPlan {
    int state = kPlanIdle;
    char* dev = nullptr; size_t dev_bytes = 0, used = 0;
    PlanDev host{};                                // the table as uploaded
    PlanDev* dtab = nullptr;
    size_t img_bytes[kPlanMaxSplit] = {};
    int split_nsrc[kPlanMaxSplit] = {};
    // plan-owned words
    int sync_words = 0; unsigned* sync = nullptr;  // dataflow words of the multi-layer backward
    bool want_xmax = false; unsigned* xmax = nullptr; const void* xmax_src = nullptr;   // range of a GEMM output (producer -> consumer)
    int xmax_writes = 0;
    int prepared = 0, hits = 0, misses = 0;        // statistics (wn_plan_stats)
};

static void* plan_alloc(StepPlan* P, size_t bytes, size_t align = 256) {
    size_t off = (P->used + align - 1) / align * align;
    if (off + bytes > P->dev_bytes) return nullptr;
    P->used = off + bytes;
    return P->dev + off;
}

StepPlan* exec_plan();                             // api.hip: WnExec.plan of the current call

// ---- hooks for the launchers ---------------------------------------------------------------------------------------
static bool same_sources(const SplitJob& j, const CGArgs& a, int n, int mode) {
    for (int i = 0; i < n; ++i)
        if (j.W[i] != a.W[i] || j.wsm[i] != a.wsm[i]) return false;
    if (mode == 3 || mode == 5)
        for (int i = 0; i < n && i < WN_GATE_TAPS; ++i)
            if (j.W2[i] != a.W2[i]) return false;
    return j.wsk == a.wsk;
}

// launch_colgemm_b3: the image of this launch's weight tiles.  READY: true + *img / *wmax when the plan holds it.
// RECORD: registers the job and returns false (the launcher prepares its own image this time).
bool plan_split_image(const CGArgs& a, int mode, int mtiles, int cps, int nchunks, int one, size_t bytes,
                      const __bf16** img, const unsigned** wmax) {
    StepPlan* P = exec_plan();
    if (!P || P->state == kPlanIdle) return false;
    if (!(mode == 0 || mode == 2 || mode == 6)) return false;
    const int nw = (mode == 2) ? mtiles : a.nsrc;                  // entries of W[] the split reads
    if (nw < 1 || nw > WN_MAX_SRC) return false;
    PlanDev& H = P->host;
    for (int i = 0; i < H.nsplit; ++i) {
        const SplitJob& j = H.split[i];
        if (j.mode == mode && j.mtiles == mtiles && j.cps == cps && j.one == one && j.ntiles == nchunks * mtiles &&
            P->split_nsrc[i] == nw && same_sources(j, a, nw, mode)) {
            if (P->state != kPlanReady) return false;               // recorded twice in one pass (same weights, same form): one job
            *img = j.img; *wmax = j.wmax;
            ++P->hits;
            return true;
        }
    }
    if (P->state == kPlanReady) { ++P->misses; return false; }
    if (H.nsplit >= kPlanMaxSplit) return false;
    SplitJob& j = H.split[H.nsplit];
    j = SplitJob{};
    for (int i = 0; i < nw; ++i) { j.W[i] = a.W[i]; j.wsm[i] = a.wsm[i]; }
    for (int i = 0; i < WN_GATE_TAPS; ++i) j.W2[i] = a.W2[i];
    j.wsk = a.wsk; j.mode = mode; j.mtiles = mtiles; j.cps = cps; j.one = one; j.ntiles = nchunks * mtiles;
    P->img_bytes[H.nsplit] = bytes;
    P->split_nsrc[H.nsplit] = nw;
    ++H.nsplit;
    return false;
}

// wn_stack_fwd: the fp16 x 2 images of the stack's fused layer kernels
const void* plan_layer_h2_images(int L, const float* const* Wf, const float* const* Wg, const float* const* Wp) {
    StepPlan* P = exec_plan();
    if (!P || P->state == kPlanIdle || L < 1 || L > 64) return nullptr;
    PlanDev& H = P->host;
    if (H.pack_L == L) {
        bool same = true;
        for (int l = 0; l < L && same; ++l) same = H.pack.Wf[l] == Wf[l] && H.pack.Wg[l] == Wg[l] && H.pack.Wp[l] == Wp[l];
        if (same) {
            if (P->state != kPlanReady) return nullptr;
            ++P->hits;
            return H.pack_img;
        }
    }
    if (P->state == kPlanReady) { ++P->misses; return nullptr; }
    if (H.pack_L) return nullptr;                                   // one stack per plan
    H.pack_L = L;
    for (int l = 0; l < L; ++l) { H.pack.Wf[l] = Wf[l]; H.pack.Wg[l] = Wg[l]; H.pack.Wp[l] = Wp[l]; }
    return nullptr;
}

// wn_stack_bwd: the multi-layer backward's dataflow words, zeroed (word 1 = ~0) by wn_plan_prepare
unsigned* plan_sync_words(int nwords) {
    StepPlan* P = exec_plan();
    if (!P || P->state == kPlanIdle || nwords < 2) return nullptr;
    if (P->state == kPlanRecord) { if (nwords > P->sync_words) P->sync_words = nwords; return nullptr; }
    if (P->sync && nwords <= P->sync_words) { ++P->hits; return P->sync; }
    ++P->misses;
    return nullptr;
}

// a GEMM whose OUTPUT's range a later call of the step needs (the head's dx = dskip): the producer asks for the word
// (zeroed by wn_plan_prepare) and folds max |out| into it; the consumer (exec_absmax) takes it when the array is the same
unsigned* plan_xmax_producer() {
    StepPlan* P = exec_plan();
    if (!P || P->state == kPlanIdle) return nullptr;
    if (P->state == kPlanRecord) { P->want_xmax = true; return nullptr; }
    return P->xmax;
}
void plan_xmax_written(const void* out) {           // the launch that fills the word is going out: `out` is what it describes
    StepPlan* P = exec_plan();
    if (!(P && P->state == kPlanReady && P->xmax)) return;
    // one word, one array: a second producer since wn_plan_prepare (a head of two convolutions) folds ITS maximum into the same
    // word -- an upper bound of either array, but a scale taken from a bound that is too large costs the smaller array its low
    // bits -- so from then on the word describes nothing and the consumer measures its operand itself
    P->xmax_src = (P->xmax_writes++ == 0) ? out : nullptr;
}
const unsigned* plan_xmax_consumer(const void* x) {
    StepPlan* P = exec_plan();
    if (!P || P->state != kPlanReady || !P->xmax || !x || P->xmax_src != x) return nullptr;
    ++P->hits;
    return P->xmax;
}

// ---- the two launches ----------------------------------------------------------------------------------------------
__device__ __forceinline__ int plan_find(const int* first, int n, int b) {
    int i = 0;
    while (i + 1 < n && b >= first[i + 1]) ++i;
    return i;
}

__global__ __launch_bounds__(256) void k_plan_prep1(const PlanDev* __restrict__ T, float4* __restrict__ user, long long user_n4) {
    const int b = blockIdx.x;
    if (b < T->b1_pack) {                                           // a weight tile: its maximum (fp16 split) or the split itself
        const int i = plan_find(T->b1_split, T->nsplit, b);
        const SplitJob& j = T->split[i];
        const int tile = b - T->b1_split[i];
        if (j.one == 3) {
            const float mw = split_w_tile<true>(j, j.mode, j.mtiles, j.cps, j.img, j.one, tile, 1.f);
            if (threadIdx.x == 0) j.partial[tile] = mw;
        } else {
            split_w_tile<false>(j, j.mode, j.mtiles, j.cps, j.img, j.one, tile, 1.f);
        }
    } else if (b < T->b1_zero) {
        layer_pack_h2_block(T->pack, T->pack_img, b - T->b1_pack);
    } else if (b < T->b1_user) {
        const ZeroJob& z = T->zero[b - T->b1_zero];
        for (int i = threadIdx.x; i < z.n; i += 256) z.p[i] = (z.second_ones && i == 1) ? 0xffffffffu : 0u;
    } else {
        const long long i0 = (long long)(b - T->b1_user) * 1024 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = i0 + 256 * k;
            if (i < user_n4) user[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

__global__ __launch_bounds__(256) void k_plan_prep2(const PlanDev* __restrict__ T) {
    const int b = blockIdx.x;
    const int i = plan_find(T->b2_split, T->nsplit, b);
    const SplitJob& j = T->split[i];
    const int tile = b - T->b2_split[i];
    float m = 0.f;
    for (int k = threadIdx.x; k < j.ntiles; k += 256) m = fmaxf(m, j.partial[k]);
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float red2[4];
    if ((threadIdx.x & 63) == 0) red2[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red2[0], red2[1]), fmaxf(red2[2], red2[3]));
    if (tile == 0 && threadIdx.x == 0) *j.wmax = __float_as_uint(m);
    split_w_tile<false>(j, j.mode, j.mtiles, j.cps, j.img, j.one, tile, h2_scale_of(m));
}

}  // namespace wn

using namespace wn;

extern "C" {

int wn_plan_create(void** plan, void* dev_mem, size_t dev_bytes) {
    WN_CHECK_ARG(plan && dev_mem && dev_bytes >= (1u << 16), "wn_plan_create: NULL argument or fewer than 64 KB of device memory");
    WN_CHECK_ARG(((uintptr_t)dev_mem & 255) == 0, "wn_plan_create: the device memory must be 256-byte aligned");
    StepPlan* P = new (std::nothrow) StepPlan();
    WN_CHECK_ARG(P, "wn_plan_create: out of host memory");
    P->dev = reinterpret_cast<char*>(dev_mem);
    P->dev_bytes = dev_bytes;
    *plan = P;
    return WN_OK;
}

int wn_plan_destroy(void* plan) {
    delete reinterpret_cast<StepPlan*>(plan);
    return WN_OK;
}

int wn_plan_record(void* plan) {
    StepPlan* P = reinterpret_cast<StepPlan*>(plan);
    WN_CHECK_ARG(P, "wn_plan_record: NULL plan");
    char* dev = P->dev; const size_t bytes = P->dev_bytes;
    *P = StepPlan();
    P->dev = dev; P->dev_bytes = bytes;
    P->state = kPlanRecord;
    return WN_OK;
}

int wn_plan_finish(void* plan, void* stream) {
    StepPlan* P = reinterpret_cast<StepPlan*>(plan);
    WN_CHECK_ARG(P && P->state == kPlanRecord, "wn_plan_finish: the plan is not recording");
    PlanDev& H = P->host;
    P->used = 0;
    P->dtab = reinterpret_cast<PlanDev*>(plan_alloc(P, sizeof(PlanDev)));
    bool ok = P->dtab != nullptr;
    int b1 = 0, b2 = 0;
    for (int i = 0; i < H.nsplit && ok; ++i) {
        SplitJob& j = H.split[i];
        j.img = reinterpret_cast<__bf16*>(plan_alloc(P, P->img_bytes[i], 1024));
        ok = j.img != nullptr;
        if (ok && j.one == 3) {
            j.wmax = reinterpret_cast<unsigned*>(plan_alloc(P, 256));
            j.partial = reinterpret_cast<float*>(plan_alloc(P, sizeof(float) * (size_t)j.ntiles));
            ok = j.wmax && j.partial;
        }
        H.b1_split[i] = b1; b1 += j.ntiles;
        H.b2_split[i] = b2; b2 += (j.one == 3) ? j.ntiles : 0;
    }
    H.b1_split[H.nsplit] = b1; H.b2_split[H.nsplit] = b2;
    H.b1_pack = b1;
    if (ok && H.pack_L) {
        H.pack_img = reinterpret_cast<char*>(plan_alloc(P, (size_t)H.pack_L * kH2ImgStride, 1024));
        ok = H.pack_img != nullptr;
        b1 += H.pack_L;
    }
    H.b1_zero = b1;
    H.nzero = 0;
    if (ok && P->sync_words) {
        P->sync = reinterpret_cast<unsigned*>(plan_alloc(P, sizeof(unsigned) * (size_t)P->sync_words));
        ok = P->sync != nullptr;
        if (ok) H.zero[H.nzero++] = ZeroJob{P->sync, P->sync_words, 1};
    }
    if (ok && P->want_xmax) {
        P->xmax = reinterpret_cast<unsigned*>(plan_alloc(P, 256));
        ok = P->xmax != nullptr;
        if (ok) H.zero[H.nzero++] = ZeroJob{P->xmax, 1, 0};
    }
    b1 += H.nzero;
    H.b1_user = b1;
    H.b1_total = b1;
    if (!ok) {
        P->state = kPlanIdle;
        wn::set_error("wn_plan_finish: the plan's device memory (%zu bytes) is too small for %d weight images", P->dev_bytes, H.nsplit);
        return WN_EARG;
    }
    // a blocking copy: the table is host memory of this object, and finish runs outside any capture
    WN_HIP(hipStreamSynchronize(as_stream(stream)));
    WN_HIP(hipMemcpy(P->dtab, &H, sizeof(PlanDev), hipMemcpyHostToDevice));
    P->state = kPlanReady;
    return WN_OK;
}

int wn_plan_prepare(void* plan, float* zero, int64_t zero_floats, void* stream) {
    StepPlan* P = reinterpret_cast<StepPlan*>(plan);
    WN_CHECK_ARG(P && P->state == kPlanReady, "wn_plan_prepare: the plan is not ready (wn_plan_record ... wn_plan_finish first)");
    WN_CHECK_ARG(zero_floats >= 0 && (zero_floats == 0 || (zero && zero_floats % 4 == 0 && ((uintptr_t)zero & 15) == 0)),
                 "wn_plan_prepare: the array to zero must be 16-byte aligned and a multiple of four floats long");
    wn::ProfScope prof__("wn_plan_prepare", stream);
    const PlanDev& H = P->host;
    const long long n4 = zero_floats / 4;
    const int user_blocks = (int)((n4 + 1023) / 1024);
    const int g1 = H.b1_total + user_blocks;
    hipStream_t s = as_stream(stream);
    if (g1 > 0) hipLaunchKernelGGL(k_plan_prep1, dim3(g1), dim3(256), 0, s, P->dtab, reinterpret_cast<float4*>(zero), n4);
    const int g2 = H.b2_split[H.nsplit];
    if (g2 > 0) hipLaunchKernelGGL(k_plan_prep2, dim3(g2), dim3(256), 0, s, P->dtab);
    WN_LAUNCH_CHECK();
    P->xmax_src = nullptr;
    P->xmax_writes = 0;
    ++P->prepared;
    return WN_OK;
}

/* out[0..7]: state (0 idle, 1 recording, 2 ready), weight images, layer images, plan-owned words, device bytes used,
 * wn_plan_prepare calls, entry-point look-ups served, look-ups not served */
int wn_plan_stats(void* plan, int64_t* out) {
    StepPlan* P = reinterpret_cast<StepPlan*>(plan);
    WN_CHECK_ARG(P && out, "wn_plan_stats: NULL argument");
    out[0] = P->state; out[1] = P->host.nsplit; out[2] = P->host.pack_L; out[3] = P->sync_words + (P->want_xmax ? 1 : 0);
    out[4] = (int64_t)P->used; out[5] = P->prepared; out[6] = P->hits; out[7] = P->misses;
    return WN_OK;
}

}  // extern "C"
