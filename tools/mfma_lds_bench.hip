// Inner loop of coarse_scan_kernel in isolation: per k-step one ds_read_b128 (issued PF steps ahead,
// consumed behind a counted lgkmcnt) feeding QB v_mfma_f32_16x16x32_bf16.  ns per k-step per SIMD for
// QB = 2 / 4, PF = 1..8, one or two waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_lds_bench.hip -o tools/bin/mfma_lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));

template <int OFF> __device__ __forceinline__ void rd(f32x4v& r, uint32_t a) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void wt(f32x4v& r) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(N) : "memory");
}
template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(static_cast<F&&>(f)); }
}

template <int QB, int PF, int THREADS, bool NOP>
__global__ __launch_bounds__(THREADS) void k(const float* __restrict__ g, float* __restrict__ out, int iters) {
    constexpr int KS = 24;
    __shared__ __attribute__((aligned(16))) char smem[KS * 1024];
    for (int i = threadIdx.x; i < KS * 256; i += THREADS) reinterpret_cast<float*>(smem)[i] = g[i & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    const uint32_t a0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem +
                        (4 * lr + (lg ^ ((4 - (lr >> 2)) & 3))) * 16;
    bf16x8v q[QB];
    {
        typedef float f32x8v __attribute__((ext_vector_type(8)));
        f32x8v x;
        for (int b = 0; b < QB; ++b) { for (int e = 0; e < 8; ++e) x[e] = g[64 + b * 8 + lane + e]; q[b] = __builtin_convertvector(x, bf16x8v); }
    }
    f32x4v acc[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) acc[b] = f32x4v{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        f32x4v xr[PF + 1];
        sfor<0, PF>([&](auto S) { constexpr int s = decltype(S)::value; rd<s * 1024>(xr[s], a0); });
        sfor<0, KS>([&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (s + PF < KS) rd<(s + PF) * 1024>(xr[(s + PF) % (PF + 1)], a0);
            wt<((KS - 1 - s) < PF ? (KS - 1 - s) : PF)>(xr[s % (PF + 1)]);
            const bf16x8v af = __builtin_bit_cast(bf16x8v, xr[s % (PF + 1)]);
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                if (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[b]) : "v"(af), "v"(q[b]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[b]) : "v"(af), "v"(q[b]));
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc[0]));
    float s = 0;
#pragma unroll
    for (int b = 0; b < QB; ++b) s += acc[b][0] + acc[b][1] + acc[b][2] + acc[b][3];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int QB, int PF, int THREADS, bool NOP>
void run(const float* g, float* out) {
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<QB, PF, THREADS, NOP>), dim3(256), dim3(THREADS), 0, 0, g, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<QB, PF, THREADS, NOP>), dim3(256), dim3(THREADS), 0, 0, g, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 24 * QB * (THREADS / 256);           // MFMAs per SIMD
    printf("QB %d PF %d waves/SIMD %d nop %d : %.2f ns per MFMA on the SIMD (ideal 7.1)\n", QB, PF, THREADS / 256, (int)NOP,
           ms * 1e6 / mf);
}

int main() {
    float *g, *out; (void)hipMalloc(&g, 1 << 16); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMemset(g, 0, 1 << 16);
    run<4, 2, 256, true>(g, out); run<4, 4, 256, true>(g, out); run<4, 8, 256, true>(g, out);
    run<2, 2, 256, true>(g, out); run<2, 4, 256, true>(g, out); run<2, 8, 256, true>(g, out);
    run<2, 2, 512, true>(g, out); run<2, 4, 512, true>(g, out);
    run<4, 4, 256, false>(g, out); run<2, 4, 256, false>(g, out); run<2, 2, 512, false>(g, out);
    return 0;
}
