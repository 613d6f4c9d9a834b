// Micro-benchmark: v_mfma_f32_32x32x2_f32 issue rate with the operand-feed patterns of the kNN scan.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f32_bench.hip -o /tmp/mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[384 * 36 + (VARIANT == 3 ? 16384 : 0)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (int)(sizeof(lds)/4); i += 512) lds[i] = (float)(i % 7) * 0.001f;
    __syncthreads();
    f32x16 acc[4];
    for (int r = 0; r < 4; ++r) for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
    const float* qrow = lds + (wave * 32 + (lane & 31)) * 36 + 4 * (lane >> 5);
    const float* brow = lds + (256 + (lane & 31)) * 36 + 4 * (lane >> 5);
    float4 av = *(const float4*)qrow;
    float4 bv[4];
    for (int r = 0; r < 4; ++r) bv[r] = *(const float4*)(brow + r * 32 * 36);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (VARIANT >= 1) {
                asm volatile("" ::: "memory");  // keep the LDS reads inside the loop  // (variant 3 = variant 1 with 120 KB LDS -> 1 block per CU)   // re-read fragments from LDS every 16 MFMAs, as the scan does
                av = *(const float4*)(qrow + kk * 8);
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = *(const float4*)(brow + r * 32 * 36 + kk * 8);
            }
            const float af[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float bf = j == 0 ? bv[r].x : j == 1 ? bv[r].y : j == 2 ? bv[r].z : bv[r].w;
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf, acc[r], 0, 0, 0);
                }
        }
        if (VARIANT == 2) __syncthreads();
    }
    float s = 0.f;
    for (int r = 0; r < 4; ++r) for (int e = 0; e < 16; ++e) s += acc[r][e];
    out[blockIdx.x * 512 + tid] = s;
}

template <int V>
void run(const char* name, int blocks, int threads, int iters) {
    float* d; hipMalloc(&d, (size_t)blocks * 512 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double flop = (double)blocks * (threads / 64) * iters * 64.0 * 4096.0;
    printf("%-40s blocks=%d threads=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks, threads, ms, flop / ms / 1e9);
    hipFree(d);
}

int main() {
    run<0>("regs only, 2 waves/SIMD (512 thr)", 256, 512, 2000);
    run<0>("regs only, 1 wave/SIMD (256 thr)", 256, 256, 2000);
    run<1>("ds_read_b128 per 16 MFMA, 2 waves/SIMD", 256, 512, 2000);
    run<1>("ds_read_b128 per 16 MFMA, 1 wave/SIMD", 256, 256, 2000);
    run<2>("+ barrier per 64 MFMA, 2 waves/SIMD", 256, 512, 2000);
    run<1>("ds_read, 2 waves/SIMD, 750 blocks", 750, 512, 24);
    run<1>("ds_read, 2 waves/SIMD, 768 blocks", 768, 512, 24);
    run<1>("ds_read, 2 waves/SIMD, 1024 blocks", 1024, 512, 24);
    run<3>("1 block/CU (big LDS), 256 blocks", 256, 512, 24);
    run<3>("1 block/CU (big LDS), 512 blocks", 512, 512, 24);
    run<3>("1 block/CU (big LDS), 750 blocks", 750, 512, 24);
    run<3>("1 block/CU (big LDS), 768 blocks", 768, 512, 24);
    run<3>("1 block/CU (big LDS), 1024 blocks", 1024, 512, 24);
    return 0;
}
