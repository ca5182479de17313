// Shared host/device helpers for libwavenet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>

#include "../../include/wavenet_hip.h"

namespace wn {

void set_error(const char* fmt, ...);

#define WN_CHECK_ARG(cond, ...)                       \
    do {                                              \
        if (!(cond)) {                                \
            wn::set_error(__VA_ARGS__);               \
            return WN_EARG;                           \
        }                                             \
    } while (0)

#define WN_CHECK_SHAPE(cond, ...)                     \
    do {                                              \
        if (!(cond)) {                                \
            wn::set_error(__VA_ARGS__);               \
            return WN_ESHAPE;                         \
        }                                             \
    } while (0)

#define WN_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess) {                                                            \
            wn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                          __LINE__);                                                        \
            return WN_EHIP;                                                                 \
        }                                                                                   \
    } while (0)

#define WN_LAUNCH_CHECK()                                                                 \
    do {                                                                                  \
        hipError_t e__ = hipGetLastError();                                               \
        if (e__ != hipSuccess) {                                                          \
            wn::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__),     \
                          __FILE__, __LINE__);                                            \
            return WN_EHIP;                                                               \
        }                                                                                 \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// prof.hip: brackets an entry point's kernels with hipEvents when wn_prof_enable(1) is in effect
struct ProfScope {
    int id = 0; hipStream_t s = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr; bool on;
    ProfScope(const char* name, void* stream);
    ~ProfScope();
};
// One bracket around a loop of calls that each have a ProfScope of the same name (the stack's layer loops): the
// launches run back to back inside it, as they do in production, instead of each between its own pair of events.
struct ProfGroup {
    ProfScope scope;
    ProfGroup(const char* name, void* stream);
    ~ProfGroup();
};
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- device math shared by the generic and the MFMA kernels ---------------------------------
// tanh / sigmoid through one v_exp_f32 and one v_rcp_f32 each (__builtin_amdgcn_rcpf: the 1-ulp hardware reciprocal --
// __frcp_rn expands to the ten-instruction correctly-rounded division sequence, 320 VALU instructions per 32-column tile of
// the layer forward); absolute error ~1e-7, which is
// what the 1e-4 logit tolerance of the north star needs (the reference computes them in float32
// too: F.tanh, and F.sigmoid as tanh(x/2)/2+1/2).
__device__ __forceinline__ float fast_tanh(float a) {
    // 1 - 2/(1+e^{2a});  e^{2a} -> inf gives 1, -> 0 gives -1.  The closed form cancels for small |a| (absolute error
    // ~6e-8 whatever a is: 1 % of a = 1e-5 -- seen as a 10 % error of the skip-projection gradients of a net whose
    // residual stream was scaled down 4,096 times); below 1/16 the odd series a - a^3/3 + 2a^5/15 (truncation < 1e-10
    // relative there) keeps the relative error at fp32 rounding.
    const float e = __expf(2.0f * a);
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
    const float a2 = a * a;
    const float p = a * fmaf(a2, fmaf(a2, 0.13333334f, -0.33333334f), 1.0f);
    return fabsf(a) < 0.0625f ? p : t;
}
__device__ __forceinline__ float fast_sigmoid(float a) {
    float e = __expf(-a);
    return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float act_apply(float x, int act) {
    if (act == WN_ACT_RELU) return x > 0.f ? x : 0.f;
    if (act == WN_ACT_ELU) return x > 0.f ? x : expm1f(x);
    return x;
}
// d act(x) / dx
__device__ __forceinline__ float act_grad(float x, int act) {
    if (act == WN_ACT_RELU) return x > 0.f ? 1.f : 0.f;
    if (act == WN_ACT_ELU) return x > 0.f ? 1.f : expf(x);
    return 1.f;
}

// layout of the WN_XENT_LOSS_WORDS floats behind `loss` (generic_kernels.hip): [0] the result, [kXentPart ..] one partial sum per
// workgroup of the loss kernel (at most kXentBlocks), then kXentCnt integer partial counts of the rows that count
static constexpr int kXentPart = 8, kXentCnt = 64, kXentBlocks = 2048 - kXentCnt;
}  // namespace wn
