// aura_zone.hip -- AdditionLinear, the -L1-distance "projection" of the neuromorphic brain zones
// (src/maths/addition_linear.py:42-67):  out[b][o] = -sum_k |x[b][k] - w[o][k]| (+ bias[o]).
// The reference materialises the (B, out, in) difference tensor; here a 64x64 output tile per
// workgroup streams x and w through LDS in 32-deep k-tiles and keeps 4x4 partial sums per lane.
// |a-b| has no matrix-core form, so this is plain VALU work (2 ops per (b,o,k)).  The backward pair
// (aura_addition_linear_backward) follows the forward.
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

// ---- backward (what autograd derives from addition_linear.py:50-64: abs -> sign, sign(0) = 0) ----
//   g_x[b][k] = -sum_o g[b][o] sgn(x[b][k] - w[o][k]),   g_w[o][k] = +sum_b g[b][o] sgn(x[b][k] - w[o][k])
// One kernel, two roles: a 64 x 64 output tile (rows r, columns k) whose 4 x 4 per-lane entries stay in registers
// with their own operand (x for g_x, w for g_w), while the reduction index (o resp. b) streams through LDS in
// 32-deep tiles: the gradient tile g[.][.] and the other operand's rows.  sgn(d) g = g with d's sign bit xor-ed in,
// zero where d == 0; VALU work like the forward (no matrix-core form).
constexpr int BR = 64, BK = 64, BT = 32;
template <bool WRT_X>
__global__ __launch_bounds__(256) void addition_linear_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ g, float* __restrict__ out,
                                                                  int64_t B, int64_t IN, int64_t OUT) {
    // WRT_X: rows r = batch rows b, reduction t = outputs o, own operand x[r][k], streamed operand w[t][k], g[r][t]
    // else : rows r = outputs o,   reduction t = batch rows b, own operand w[r][k], streamed operand x[t][k], g[t][r]
    __shared__ float gs[BT * (BR + 1)];                      // [t][r]
    __shared__ float os[BT * (BK + 1)];                      // [t][k]
    const int tid = threadIdx.x, tr = tid >> 4, tk = tid & 15;
    const int64_t R = WRT_X ? B : OUT, T = WRT_X ? OUT : B;
    const int64_t r0 = (int64_t)blockIdx.y * BR, k0 = (int64_t)blockIdx.x * BK;
    const float* const own = WRT_X ? x : w;
    const float* const oth = WRT_X ? w : x;
    float ov[4][4], acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = r0 + tr + 16 * i, k = k0 + tk + 16 * j;
            ov[i][j] = (r < R && k < IN) ? own[r * IN + k] : 0.0f;
            acc[i][j] = 0.0f;
        }
    for (int64_t t0 = 0; t0 < T; t0 += BT) {
        for (int f = tid; f < BT * BR; f += 256) {
            const int t = f / BR, r = f % BR;
            const int64_t tt = t0 + t, rr = r0 + r;
            float v = 0.0f;
            if (tt < T && rr < R) v = WRT_X ? g[rr * OUT + tt] : g[tt * OUT + rr];
            gs[t * (BR + 1) + r] = v;
        }
        for (int f = tid; f < BT * BK; f += 256) {
            const int t = f / BK, c = f % BK;
            const int64_t tt = t0 + t, k = k0 + c;
            os[t * (BK + 1) + c] = (tt < T && k < IN) ? oth[tt * IN + k] : 0.0f;
        }
        __syncthreads();
        const int tmax = (T - t0) < BT ? (int)(T - t0) : BT;
        for (int t = 0; t < tmax; ++t) {
            float gv[4], sv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) gv[i] = gs[t * (BR + 1) + tr + 16 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) sv[j] = os[t * (BK + 1) + tk + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = WRT_X ? ov[i][j] - sv[j] : sv[j] - ov[i][j];     // x - w
                    const uint32_t sg = __float_as_uint(gv[i]) ^ (__float_as_uint(d) & 0x80000000u);
                    acc[i][j] += d == 0.0f ? 0.0f : __uint_as_float(sg);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = r0 + tr + 16 * i, k = k0 + tk + 16 * j;
            if (r < R && k < IN) out[r * IN + k] = WRT_X ? -acc[i][j] : acc[i][j];
        }
}
}  // namespace

extern "C" int aura_addition_linear_backward(const float* x, const float* weight_patterns, const float* g_out,
                                             float* g_x, float* g_w, int64_t B, int64_t in_features,
                                             int64_t out_features, void* stream) {
    if (B < 0 || in_features <= 0 || out_features <= 0) return AURA_E_INVAL;
    if (!x || !weight_patterns || (B > 0 && !g_out)) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t gk = (in_features + BK - 1) / BK;
    if (g_x && B > 0) {
        const int64_t gr = (B + BR - 1) / BR;
        if (gr > 65535) return AURA_E_INVAL;
        hipLaunchKernelGGL((addition_linear_bwd_kernel<true>), dim3((unsigned)gk, (unsigned)gr), dim3(256), 0, s, x,
                           weight_patterns, g_out, g_x, B, in_features, out_features);
    }
    if (g_w) {
        const int64_t gr = (out_features + BR - 1) / BR;
        if (gr > 65535) return AURA_E_INVAL;
        hipLaunchKernelGGL((addition_linear_bwd_kernel<false>), dim3((unsigned)gk, (unsigned)gr), dim3(256), 0, s, x,
                           weight_patterns, g_out, g_w, B, in_features, out_features);   // (B == 0: zeros)
    }
    return hipGetLastError() == hipSuccess ? AURA_OK : AURA_E_LAUNCH;
}

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
