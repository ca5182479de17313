// vmcnt ordering with LDS-DMA in the queue (gfx950): an older COLD LDS-DMA, then younger hot register loads / stores; wait vmcnt(N younger)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void fill(unsigned* s, long long n) { for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) s[i] = (unsigned)i; }
// MODE 0: cold DMA, NY hot stores, vmcnt(NW), ds_read.   MODE 1: cold DMA, NY hot reg loads (asm, agpr), vmcnt(NW), ds_read
// MODE 2: cold reg load (agpr), NY hot DMAs, then 8 stores, vmcnt(NW): reg landed?
template <int MODE, int NY, int NW>
__global__ void k(const unsigned* __restrict__ src, unsigned* __restrict__ dst, unsigned* __restrict__ bad, int iters, long long stride_words) {
    __shared__ unsigned lds[4][2][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long gw = (long long)blockIdx.x * (blockDim.x >> 6) + wv;
    unsigned nbad = 0;
    const unsigned l0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)&lds[wv][0][0]);
    const unsigned l1 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)&lds[wv][1][0]);
    for (int it = 0; it < iters; ++it) {
        const long long w = ((gw * iters + it) * 64 + lane) * stride_words % (1ll << 28);
        const unsigned* p = src + w;                          // cold
        unsigned* q = dst + (gw * 64 + lane) * 4;             // hot
        const unsigned* hot = src + (gw * 64 + lane);         // hot after the first pass
        lds[wv][0][lane] = 0xdeadbeefu;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        unsigned got;
        if constexpr (MODE == 0 || MODE == 1) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(p), "s"(l0) : "memory", "m0");
            if constexpr (MODE == 0) {
#pragma unroll
                for (int s = 0; s < NY; ++s) asm volatile("global_store_dword %0, %1, off" :: "v"(q + (s & 3)), "v"((unsigned)it) : "memory");
            } else {
                unsigned t[NY > 0 ? NY : 1];
#pragma unroll
                for (int s = 0; s < NY; ++s) asm volatile("global_load_dword %0, %1, off" : "=a"(t[s]) : "v"(hot) : "memory");
#pragma unroll
                for (int s = 0; s < NY; ++s) asm volatile("" :: "a"(t[s]));
            }
            asm volatile("s_waitcnt vmcnt(%2)\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(l0 + 4u * lane), "n"(NW) : "memory");
        } else {
            unsigned v;
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(v) : "v"(0xdeadbeefu));
            asm volatile("global_load_dword %0, %1, off" : "+a"(v) : "v"(p) : "memory");
#pragma unroll
            for (int s = 0; s < NY; ++s)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(hot), "s"(l1) : "memory", "m0");
#pragma unroll
            for (int s = 0; s < 8; ++s) asm volatile("global_store_dword %0, %1, off" :: "v"(q + (s & 3)), "v"((unsigned)it) : "memory");
            asm volatile("s_waitcnt vmcnt(%2)\n\ts_nop 0\n\tv_accvgpr_read_b32 %0, %1" : "=v"(got), "+a"(v) : "n"(NW) : "memory");
        }
        if (got != (unsigned)w) ++nbad;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main() {
    const long long words = 1ll << 28;
    unsigned *src, *dst, *bad;
    CK(hipMalloc(&src, words * 4)); CK(hipMalloc(&dst, 1 << 24)); CK(hipMalloc(&bad, 4));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, src, words);
    CK(hipDeviceSynchronize());
#define RUN(M, NY, NW, name) do { CK(hipMemset(bad, 0, 4)); hipLaunchKernelGGL((k<M, NY, NW>), dim3(1024), dim3(256), 0, 0, src, dst, bad, 1000, 4099ll); \
    CK(hipDeviceSynchronize()); unsigned b; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost)); printf("%-70s %u stale of %lld\n", name, b, 1024ll * 4 * 64 * 1000); } while (0)
    RUN(0, 8, 8, "cold DMA, 8 hot stores, vmcnt(8)");
    RUN(0, 8, 9, "cold DMA, 8 hot stores, vmcnt(9) [must be stale]");
    RUN(1, 8, 8, "cold DMA, 8 hot reg loads, vmcnt(8)");
    RUN(1, 8, 9, "cold DMA, 8 hot reg loads, vmcnt(9) [must be stale]");
    RUN(2, 4, 12, "cold reg load, 4 hot DMAs, 8 stores, vmcnt(12)");
    RUN(2, 4, 13, "cold reg load, 4 hot DMAs, 8 stores, vmcnt(13) [must be stale]");
    RUN(2, 31, 39, "cold reg load, 31 hot DMAs, 8 stores, vmcnt(39)");
    return 0;
}
