// aura_zone.hip -- AdditionLinear, the -L1-distance "projection" of the neuromorphic brain zones
// (src/maths/addition_linear.py:42-67):  out[b][o] = -sum_k |x[b][k] - w[o][k]| (+ bias[o]).
// The reference materialises the (B, out, in) difference tensor; here a 64x64 output tile per
// workgroup streams x and w through LDS in 32-deep k-tiles and keeps 4x4 partial sums per lane.
// |a-b| has no matrix-core form, so this is plain VALU work (2 ops per (b,o,k)).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/aura_hip.h"

#pragma clang fp contract(off)

namespace {
constexpr int TB = 64, TO = 64, TK = 32, PAD = 33;

__global__ __launch_bounds__(256) void addition_linear_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ out, int64_t B,
                                                              int64_t IN, int64_t OUT) {
    __shared__ float xs[TB * PAD];
    __shared__ float ws[TO * PAD];
    const int tid = threadIdx.x;
    const int tb = tid >> 4, to = tid & 15;  // 16 x 16 threads, 4 x 4 outputs each
    const int64_t b0 = (int64_t)blockIdx.y * TB, o0 = (int64_t)blockIdx.x * TO;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;

    for (int64_t k0 = 0; k0 < IN; k0 += TK) {
        for (int f = tid; f < TB * TK; f += 256) {
            const int r = f / TK, c = f % TK;
            const int64_t k = k0 + c;
            xs[r * PAD + c] = (b0 + r < B && k < IN) ? x[(b0 + r) * IN + k] : 0.0f;
            ws[r * PAD + c] = (o0 + r < OUT && k < IN) ? w[(o0 + r) * IN + k] : 0.0f;
        }
        __syncthreads();
        const int kmax = (IN - k0) < TK ? (int)(IN - k0) : TK;
        for (int kk = 0; kk < kmax; ++kk) {
            float xv[4], wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[i] = xs[(tb + 16 * i) * PAD + kk];
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = ws[(to + 16 * j) * PAD + kk];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += fabsf(xv[i] - wv[j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t b = b0 + tb + 16 * i;
        if (b >= B) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t o = o0 + to + 16 * j;
            if (o >= OUT) continue;
            float v = -acc[i][j];
            if (bias) v = v + bias[o];
            out[b * OUT + o] = v;
        }
    }
}
}  // namespace

extern "C" int aura_addition_linear(const float* x, const float* weight_patterns, const float* bias,
                                    float* out, int64_t B, int64_t in_features,
                                    int64_t out_features, void* stream) {
    if (B < 0 || in_features <= 0 || out_features <= 0) return AURA_E_INVAL;
    if (B == 0) return AURA_OK;
    if (!x || !weight_patterns || !out) return AURA_E_INVAL;
    const int64_t gy = (B + TB - 1) / TB, gx = (out_features + TO - 1) / TO;
    if (gy > 65535) return AURA_E_INVAL;
    hipLaunchKernelGGL(addition_linear_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, weight_patterns, bias, out, B,
                       in_features, out_features);
    return hipGetLastError() == hipSuccess ? AURA_OK : AURA_E_LAUNCH;
}
