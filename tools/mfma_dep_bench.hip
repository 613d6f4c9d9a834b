// How far apart must two v_mfma_f32_16x16x32_bf16 on the SAME accumulator be?  ns per MFMA for NB
// independent accumulator chains per wave, with 1 or 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_dep_bench.hip -o /tmp/mfma_dep_bench && /tmp/mfma_dep_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));

template <int NB, int THREADS, bool NOP>
__global__ __launch_bounds__(THREADS) void k(const float* __restrict__ g, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8v a, q[NB];
    {
        typedef float f32x8v __attribute__((ext_vector_type(8)));
        f32x8v x;
        for (int e = 0; e < 8; ++e) x[e] = g[lane + e];
        a = __builtin_convertvector(x, bf16x8v);
        for (int b = 0; b < NB; ++b) { for (int e = 0; e < 8; ++e) x[e] = g[64 + b * 8 + lane + e]; q[b] = __builtin_convertvector(x, bf16x8v); }
    }
    f32x4v acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = f32x4v{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 48 / NB; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[b]) : "v"(a), "v"(q[b]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[b]) : "v"(a), "v"(q[b]));
            }
    }
    asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc[0]));
    float s = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) s += acc[b][0] + acc[b][1] + acc[b][2] + acc[b][3];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int NB, int THREADS, bool NOP>
void run(const float* g, float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NB, THREADS, NOP>), dim3(256), dim3(THREADS), 0, 0, g, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NB, THREADS, NOP>), dim3(256), dim3(THREADS), 0, 0, g, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 48 * (THREADS / 256);          // MFMAs per SIMD
    printf("chains/wave %d  waves/SIMD %d  nop %d : %.2f ns per MFMA on the SIMD (%.1f cyc @2.4GHz)\n", NB, THREADS / 256,
           (int)NOP, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}

int main() {
    float *g, *out; hipMalloc(&g, 1 << 16); hipMalloc(&out, 256 * 512 * 4); hipMemset(g, 0, 1 << 16);
    run<1, 256, false>(g, out); run<2, 256, false>(g, out); run<3, 256, false>(g, out); run<4, 256, false>(g, out); run<8, 256, false>(g, out);
    run<2, 256, true>(g, out); run<4, 256, true>(g, out);
    run<1, 512, false>(g, out); run<2, 512, false>(g, out); run<4, 512, false>(g, out);
    run<2, 512, true>(g, out); run<4, 512, true>(g, out);
    return 0;
}
