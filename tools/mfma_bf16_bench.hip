// Microbenchmark for the inner loop of coarse_scan_kernel: cycles per v_mfma_f32_16x16x32_bf16 with the
// stationary operand in AGPRs / VGPRs, with and without the s_nop guard, the fp32->bf16 converts and
// the LDS fragment reads.   hipcc -O3 --offload-arch=gfx950 tools/mfma_bf16_bench.hip -o /tmp/mfma_bf16_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));

template <bool QA, bool NOP>
__device__ __forceinline__ void mf(f32x4v& acc, const bf16x8v& a, const bf16x8v& q) {
    if constexpr (QA && NOP) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(q));
    else if constexpr (QA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(q));
    else if constexpr (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(q));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(q));
}

// MODE bit0: LDS reads, bit1: converts, bit2: s_nop, bit3: all B operands in VGPR (KS must be small)
template <int KS, int MODE, int NA>
__global__ __launch_bounds__(256) void k(const float* __restrict__ g, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < KS * 512; i += 256) reinterpret_cast<float*>(smem)[i] = g[i];
    __syncthreads();
    bf16x8v qf[4][KS];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f32x8v x;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = g[(b * KS + s) * 8 + e + lane];
            const bf16x8v cv = __builtin_convertvector(x, bf16x8v);
            if (b * KS + s < NA) asm volatile("" : "=a"(qf[b][s]) : "0"(cv));
            else asm volatile("" : "=v"(qf[b][s]) : "0"(cv));
        }
    const int lr = lane & 15, lg = lane >> 4, sw = (lr >> 1) & 7;
    const int off0 = (8 * lr + ((2 * lg) ^ sw)) * 16, off1 = (8 * lr + ((2 * lg + 1) ^ sw)) * 16;
    f32x4v acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f32x4v{0, 0, 0, 0};
    f32x4v x0 = *reinterpret_cast<const f32x4v*>(smem + off0), x1 = *reinterpret_cast<const f32x4v*>(smem + off1);
    for (int it = 0; it < iters; ++it) {
        f32x4v xr[3][2];
        xr[0][0] = x0; xr[0][1] = x1; xr[1][0] = x1; xr[1][1] = x0;
        if (MODE & 1) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                xr[s][0] = *reinterpret_cast<const f32x4v*>(smem + s * 2048 + off0);
                xr[s][1] = *reinterpret_cast<const f32x4v*>(smem + s * 2048 + off1);
            }
        }
        xr[2][0] = x0; xr[2][1] = x1;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if ((MODE & 1) && s + 2 < KS) {
                xr[(s + 2) % 3][0] = *reinterpret_cast<const f32x4v*>(smem + (s + 2) * 2048 + off0);
                xr[(s + 2) % 3][1] = *reinterpret_cast<const f32x4v*>(smem + (s + 2) * 2048 + off1);
            }
            const f32x4v a0 = xr[s % 3][0], a1 = xr[s % 3][1];
            bf16x8v af;
            if (MODE & 2) {
                f32x8v x;
                x[0] = a0[0]; x[1] = a0[1]; x[2] = a0[2]; x[3] = a0[3];
                x[4] = a1[0]; x[5] = a1[1]; x[6] = a1[2]; x[7] = a1[3];
                af = __builtin_convertvector(x, bf16x8v);
            } else {
                af = __builtin_bit_cast(bf16x8v, a0);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b * KS + s < NA) {
                    if (MODE & 4) mf<true, true>(acc[b], af, qf[b][s]); else mf<true, false>(acc[b], af, qf[b][s]);
                } else {
                    if (MODE & 4) mf<false, true>(acc[b], af, qf[b][s]); else mf<false, false>(acc[b], af, qf[b][s]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
    float s = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) s += acc[b][0] + acc[b][1] + acc[b][2] + acc[b][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KS, int MODE, int NA>
void run(const char* name, const float* g, float* out) {
    const int iters = 2000;
    const size_t lds = KS * 2048;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<KS, MODE, NA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KS, MODE, NA>), dim3(256), dim3(256), lds, 0, g, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KS, MODE, NA>), dim3(256), dim3(256), lds, 0, g, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * KS * 4;               // per wave (one wave per SIMD)
    const double ns_per = ms * 1e6 / mfmas;
    const double tf = 256.0 * 4 * mfmas * 16384 / (ms * 1e-3) / 1e12;
    printf("%-44s KS=%2d  %.2f ns/MFMA (%.1f cyc @2.4GHz)  %.0f TFLOP/s\n", name, KS, ns_per, ns_per * 2.4, tf);
}

int main() {
    float *g, *out;
    hipMalloc(&g, 1 << 22); hipMalloc(&out, 1 << 20);
    hipMemset(g, 0, 1 << 22);
    run<8, 0, 32>("mfma only, B in AGPR", g, out);
    run<8, 0, 0>("mfma only, B in VGPR", g, out);
    run<8, 4, 32>("mfma + s_nop 1, B in AGPR", g, out);
    run<8, 2, 32>("mfma + cvt, B in AGPR", g, out);
    run<8, 3, 32>("mfma + cvt + ds_read, B in AGPR", g, out);
    run<8, 7, 32>("mfma + cvt + ds_read + nop, B in AGPR", g, out);
    run<24, 7, 58>("as the kernel: KS=24, 58 frags AGPR", g, out);
    run<24, 3, 58>("KS=24 without the nops", g, out);
    run<24, 0, 58>("KS=24 mfma only", g, out);
    return 0;
}
