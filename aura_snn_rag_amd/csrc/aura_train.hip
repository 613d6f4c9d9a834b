// aura_train.hip -- surrogate-gradient training kernels and the prosody-modulated GIF loop
// (SURVEY.md section 8f-4), fp32 and bf16, channel-contiguous layout [rows][T][H].
//
//   GIF   forward (training): the loop of gif_neuron.py:54-69 that also saves, per step, the
//         pre-clamp potential a_t and the threshold theta_{t-1} it was computed with;
//   GIF   backward: BPTT through T steps with MultiBitSurrogate's triangular window
//         (gif_neuron.py:16-22), the tensor-bound clamp (gradient reaches theta through the
//         bounds when a value is clamped) and the threshold adaptation recurrence;
//   LIF   one-step forward saving the surrogate input, and backward with
//         LearnableSurrogateGradient (neuron.py:80-108): fast-sigmoid window + slope gradient;
//   ProsodyModulatedGIF forward (prosody_gif.py:69-106): per-(row, t) attention gains scale the
//         input, the effective threshold and the adaptation rate.
// One lane owns VEC adjacent channels; state lives in registers over the whole T loop; HBM-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/aura_hip.h"

#pragma clang fp contract(off)

namespace {

template <int VEC> struct V;
template <> struct V<4> {
    __device__ static void ld(const float* p, float (&x)[4]) { float4 t = *reinterpret_cast<const float4*>(p); x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w; }
    __device__ static void st(float* p, const float (&x)[4]) { *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]); }
};
template <> struct V<1> {
    __device__ static void ld(const float* p, float (&x)[1]) { x[0] = *p; }
    __device__ static void st(float* p, const float (&x)[1]) { *p = x[0]; }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int grid_for(int64_t items) {
    int64_t b = (items + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}
inline int check_launch() { return hipGetLastError() == hipSuccess ? AURA_OK : AURA_E_LAUNCH; }

struct GifP { float decay, Lf, alpha, thr0; };

template <int VEC>
__global__ __launch_bounds__(256) void gif_train_fwd_kernel(GifP p, const float* __restrict__ h,
                                                            float* __restrict__ spikes, float* v_io,
                                                            float* th_io, float* __restrict__ save_a,
                                                            float* __restrict__ save_th, int64_t R,
                                                            int64_t T, int64_t C) {
    const int64_t cv = C / VEC, items = R * cv;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / cv, c0 = (it - row * cv) * VEC;
        float v[VEC], th[VEC];
        V<VEC>::ld(v_io + row * C + c0, v);
        V<VEC>::ld(th_io + row * C + c0, th);
        for (int64_t t = 0; t < T; ++t) {
            const int64_t o = (row * T + t) * C + c0;
            float x[VEC], a[VEC], s[VEC], thp[VEC];
            V<VEC>::ld(h + o, x);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                thp[e] = th[e];
                a[e] = v[e] * p.decay + x[e];
                const float cl = p.Lf * th[e] * 2.0f;
                const float b = fminf(fmaxf(a[e], -cl), cl);
                const float n = b / (th[e] + 1e-6f);
                s[e] = fminf(fmaxf(floorf(n), 0.0f), p.Lf);
                v[e] = b - s[e] * th[e];
                if (p.alpha > 0.0f) th[e] = th[e] + p.alpha * s[e] - p.alpha * (th[e] - p.thr0);
            }
            V<VEC>::st(spikes + o, s);
            V<VEC>::st(save_a + o, a);
            V<VEC>::st(save_th + o, thp);
        }
        V<VEC>::st(v_io + row * C + c0, v);
        V<VEC>::st(th_io + row * C + c0, th);
    }
}

// gv_io / gth_io: in = dL/dv_T, dL/dtheta_T; out = dL/dv_0, dL/dtheta_0
template <int VEC>
__global__ __launch_bounds__(256) void gif_bwd_kernel(GifP p, const float* __restrict__ save_a,
                                                      const float* __restrict__ save_th,
                                                      const float* __restrict__ g_spikes,
                                                      float* __restrict__ g_h, float* gv_io,
                                                      float* gth_io, int64_t R, int64_t T, int64_t C) {
    const int64_t cv = C / VEC, items = R * cv;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / cv, c0 = (it - row * cv) * VEC;
        float gv[VEC], gth[VEC];
        V<VEC>::ld(gv_io + row * C + c0, gv);
        V<VEC>::ld(gth_io + row * C + c0, gth);
        for (int64_t t = T - 1; t >= 0; --t) {
            const int64_t o = (row * T + t) * C + c0;
            float a[VEC], thp[VEC], gs[VEC], gh[VEC];
            V<VEC>::ld(save_a + o, a);
            V<VEC>::ld(save_th + o, thp);
            V<VEC>::ld(g_spikes + o, gs);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float cl = p.Lf * thp[e] * 2.0f;
                const float b = fminf(fmaxf(a[e], -cl), cl);
                const float d = thp[e] + 1e-6f;
                const float n = b / d;
                const float s = fminf(fmaxf(floorf(n), 0.0f), p.Lf);
                float gth_prev = gth[e];
                float gs_tot = gs[e];
                if (p.alpha > 0.0f) {            // theta_t = (thp + alpha*s) - alpha*(thp - thr0)
                    gs_tot = gs_tot + p.alpha * gth[e];
                    gth_prev = gth_prev - p.alpha * gth[e];
                }
                gs_tot = gs_tot - thp[e] * gv[e];   // v_t = b - s*thp
                gth_prev = gth_prev - s * gv[e];
                float gb = gv[e];
                const float dist = fabsf(n - rintf(n));
                const float tri = fminf(fmaxf(1.0f - 2.0f * dist, 0.0f), 1.0f);
                const float sur = (n >= 0.0f && n <= p.Lf + 1.0f) ? tri : 0.0f;
                const float gn = gs_tot * sur;
                gb = gb + gn / d;
                gth_prev = gth_prev + (-gn * b / (d * d));
                float ga = 0.0f, gcl = 0.0f;
                if (a[e] < -cl) gcl = -gb;
                else if (a[e] > cl) gcl = gb;
                else ga = gb;
                gth_prev = gth_prev + gcl * (2.0f * p.Lf);
                gh[e] = ga;
                gv[e] = ga * p.decay;
                gth[e] = gth_prev;
            }
            V<VEC>::st(g_h + o, gh);
        }
        V<VEC>::st(gv_io + row * C + c0, gv);
        V<VEC>::st(gth_io + row * C + c0, gth);
    }
}

// ---- bf16 tensors (the reference under bf16 / autocast): [rows][T][H] of bf16 bit patterns ----
// Forward: the per-op-rounded loop of aura_gif_run's bf16 form (each op rounds its fp32 result to bf16, Python
// scalars are fp32) -- spikes and state are bit-identical to it -- saving a_t and theta_{t-1} as the bf16
// values the reference's graph holds.  Backward: the same BPTT as the fp32 kernel, evaluated in fp32 from
// those saved bf16 values (the forward intermediates b, d, n, s are recomputed WITH their bf16 roundings, so
// the surrogate window and the clamp branches are the reference's), gradients carried in fp32 registers
// across the T steps and rounded to bf16 once, on the way out.  The reference's autograd rounds every
// intermediate gradient to bf16 instead: its result scatters around this one by a few bf16 ulps per step
// (tests/test_gpu_training.py compares both).
__device__ __forceinline__ float rb(float x) { return static_cast<float>(static_cast<__bf16>(x)); }
__device__ __forceinline__ float bf2f(uint16_t u) { return __uint_as_float((uint32_t)u << 16); }
__device__ __forceinline__ uint16_t f2bf(float x) {       // round to nearest even (values already bf16: exact)
    const __bf16 b = static_cast<__bf16>(x);
    return __builtin_bit_cast(uint16_t, b);
}
template <int VEC> struct VB;
template <> struct VB<8> {
    __device__ static void ld(const uint16_t* p, float (&x)[8]) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[2 * i] = __uint_as_float(w[i] << 16); x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    __device__ static void st(uint16_t* p, const float (&x)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f2bf(x[2 * i]) | ((uint32_t)f2bf(x[2 * i + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <> struct VB<1> {
    __device__ static void ld(const uint16_t* p, float (&x)[1]) { x[0] = bf2f(*p); }
    __device__ static void st(uint16_t* p, const float (&x)[1]) { *p = f2bf(x[0]); }
};

// one bf16 GIF step's forward intermediates from (a, theta_prev): b, d, n, s (all bf16 values)
__device__ __forceinline__ void gif_bf16_mid(const GifP& p, float a, float thp, float& b, float& d, float& n, float& s) {
    const float cl = rb(rb(p.Lf * thp) * 2.0f);
    b = fminf(fmaxf(a, -cl), cl);
    d = rb(thp + 1e-6f);
    n = rb(b / d);
    s = fminf(fmaxf(floorf(n), 0.0f), p.Lf);
}

template <int VEC>
__global__ __launch_bounds__(256) void gif_train_fwd_bf16_kernel(GifP p, const uint16_t* __restrict__ h,
                                                                 uint16_t* __restrict__ spikes, uint16_t* v_io,
                                                                 uint16_t* th_io, uint16_t* __restrict__ save_a,
                                                                 uint16_t* __restrict__ save_th, int64_t R,
                                                                 int64_t T, int64_t C) {
    const int64_t cv = C / VEC, items = R * cv;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / cv, c0 = (it - row * cv) * VEC;
        float v[VEC], th[VEC];
        VB<VEC>::ld(v_io + row * C + c0, v);
        VB<VEC>::ld(th_io + row * C + c0, th);
        for (int64_t t = 0; t < T; ++t) {
            const int64_t o = (row * T + t) * C + c0;
            float x[VEC], a[VEC], s[VEC], thp[VEC];
            VB<VEC>::ld(h + o, x);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                thp[e] = th[e];
                a[e] = rb(rb(v[e] * p.decay) + x[e]);
                float b, d, n;
                gif_bf16_mid(p, a[e], th[e], b, d, n, s[e]);
                v[e] = rb(b - rb(s[e] * th[e]));
                if (p.alpha > 0.0f)
                    th[e] = rb(rb(th[e] + rb(p.alpha * s[e])) - rb(p.alpha * rb(th[e] - p.thr0)));
            }
            VB<VEC>::st(spikes + o, s);
            VB<VEC>::st(save_a + o, a);
            VB<VEC>::st(save_th + o, thp);
        }
        VB<VEC>::st(v_io + row * C + c0, v);
        VB<VEC>::st(th_io + row * C + c0, th);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void gif_bwd_bf16_kernel(GifP p, const uint16_t* __restrict__ save_a,
                                                           const uint16_t* __restrict__ save_th,
                                                           const uint16_t* __restrict__ g_spikes,
                                                           uint16_t* __restrict__ g_h, uint16_t* gv_io,
                                                           uint16_t* gth_io, int64_t R, int64_t T, int64_t C) {
    const int64_t cv = C / VEC, items = R * cv;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / cv, c0 = (it - row * cv) * VEC;
        float gv[VEC], gth[VEC];
        VB<VEC>::ld(gv_io + row * C + c0, gv);
        VB<VEC>::ld(gth_io + row * C + c0, gth);
        for (int64_t t = T - 1; t >= 0; --t) {
            const int64_t o = (row * T + t) * C + c0;
            float a[VEC], thp[VEC], gs[VEC], gh[VEC];
            VB<VEC>::ld(save_a + o, a);
            VB<VEC>::ld(save_th + o, thp);
            VB<VEC>::ld(g_spikes + o, gs);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float b, d, n, s;
                gif_bf16_mid(p, a[e], thp[e], b, d, n, s);
                const float cl = rb(rb(p.Lf * thp[e]) * 2.0f);
                float gth_prev = gth[e];
                float gs_tot = gs[e];
                if (p.alpha > 0.0f) {            // theta_t = (thp + alpha*s) - alpha*(thp - thr0)
                    gs_tot = gs_tot + p.alpha * gth[e];
                    gth_prev = gth_prev - p.alpha * gth[e];
                }
                gs_tot = gs_tot - thp[e] * gv[e];   // v_t = b - s*thp
                gth_prev = gth_prev - s * gv[e];
                float gb = gv[e];
                const float dist = fabsf(n - rintf(n));
                const float tri = fminf(fmaxf(1.0f - 2.0f * dist, 0.0f), 1.0f);
                const float sur = (n >= 0.0f && n <= p.Lf + 1.0f) ? tri : 0.0f;
                const float gn = gs_tot * sur;
                gb = gb + gn / d;
                gth_prev = gth_prev + (-gn * b / (d * d));
                float ga = 0.0f, gcl = 0.0f;
                if (a[e] < -cl) gcl = -gb;
                else if (a[e] > cl) gcl = gb;
                else ga = gb;
                gth_prev = gth_prev + gcl * (2.0f * p.Lf);
                gh[e] = ga;
                gv[e] = ga * p.decay;
                gth[e] = gth_prev;
            }
            VB<VEC>::st(g_h + o, gh);
        }
        VB<VEC>::st(gv_io + row * C + c0, gv);
        VB<VEC>::st(gth_io + row * C + c0, gth);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void lif_train_fwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ mem_in,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ thr,
                                                            float* __restrict__ spk,
                                                            float* __restrict__ mem_out,
                                                            float* __restrict__ pre, int64_t B,
                                                            int64_t C) {
    const int64_t cv = C / VEC, items = B * cv;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / cv, c0 = (it - row * cv) * VEC;
        float xv[VEC], m[VEC], bt[VEC], th[VEC], s[VEC], pr[VEC];
        V<VEC>::ld(x + row * C + c0, xv);
        V<VEC>::ld(mem_in + row * C + c0, m);
        V<VEC>::ld(beta + c0, bt);
        V<VEC>::ld(thr + c0, th);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float mm = bt[e] * m[e] + xv[e];
            pr[e] = mm - th[e];
            s[e] = pr[e] > 0.0f ? 1.0f : 0.0f;
            m[e] = mm - s[e] * th[e];
        }
        V<VEC>::st(spk + row * C + c0, s);
        V<VEC>::st(mem_out + row * C + c0, m);
        V<VEC>::st(pre + row * C + c0, pr);
    }
}

// g_in = (g_mem_out + (g_spk - g_mem_out*thr) * slope/(|slope*pre|+1)^2); g_x = g_in;
// g_mem_prev = beta * g_in; raw_slope = -(g_spk - g_mem_out*thr) * |pre|*sign(pre)/(slope*|pre|+1)^2
template <int VEC>
__global__ __launch_bounds__(256) void lif_bwd_kernel(const float* __restrict__ pre,
                                                      const float* __restrict__ g_spk,
                                                      const float* __restrict__ g_mem,
                                                      const float* __restrict__ beta,
                                                      const float* __restrict__ thr,
                                                      const float* __restrict__ slope,
                                                      float* __restrict__ g_x,
                                                      float* __restrict__ g_mem_prev,
                                                      float* __restrict__ raw_slope, int64_t B,
                                                      int64_t C) {
    const int64_t cv = C / VEC, items = B * cv;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / cv, c0 = (it - row * cv) * VEC;
        float pr[VEC], gs[VEC], gm[VEC], bt[VEC], th[VEC], sl[VEC], gx[VEC], gp[VEC], rs[VEC];
        V<VEC>::ld(pre + row * C + c0, pr);
        V<VEC>::ld(g_spk + row * C + c0, gs);
        V<VEC>::ld(g_mem + row * C + c0, gm);
        V<VEC>::ld(beta + c0, bt);
        V<VEC>::ld(thr + c0, th);
        V<VEC>::ld(slope + c0, sl);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float g_s = gs[e] - gm[e] * th[e];         // spk feeds the output and mem_out = mem - spk*thr
            const float den = fabsf(sl[e] * pr[e]) + 1.0f;
            const float g_pre = g_s * (sl[e] / (den * den));
            const float g_m = gm[e] + g_pre;                 // grad of mem' = beta*mem + x
            gx[e] = g_m;
            gp[e] = bt[e] * g_m;
            const float ap = fabsf(pr[e]);
            const float sg = pr[e] > 0.0f ? 1.0f : (pr[e] < 0.0f ? -1.0f : 0.0f);
            const float den2 = sl[e] * ap + 1.0f;
            rs[e] = -g_s * ap * sg / (den2 * den2);
        }
        V<VEC>::st(g_x + row * C + c0, gx);
        V<VEC>::st(g_mem_prev + row * C + c0, gp);
        V<VEC>::st(raw_slope + row * C + c0, rs);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void gif_prosody_kernel(GifP p, float strength,
                                                          const float* __restrict__ h,
                                                          const float* __restrict__ gains,
                                                          float* __restrict__ spikes, float* v_io,
                                                          float* th_io, int64_t R, int64_t T, int64_t C) {
    const int64_t cv = C / VEC, items = R * cv;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / cv, c0 = (it - row * cv) * VEC;
        float v[VEC], th[VEC];
        V<VEC>::ld(v_io + row * C + c0, v);
        V<VEC>::ld(th_io + row * C + c0, th);
        for (int64_t t = 0; t < T; ++t) {
            const int64_t o = (row * T + t) * C + c0;
            float x[VEC], s[VEC];
            V<VEC>::ld(h + o, x);
            const bool mod = gains != nullptr;
            const float g = mod ? gains[row * T + t] : 1.0f;
            const float scale = fminf(fmaxf(1.0f - strength * (g - 1.0f), 0.5f), 1.5f);
            const float a_eff = mod ? p.alpha * g : p.alpha;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float i_t = mod ? x[e] * g : x[e];
                float vv = v[e] * p.decay + i_t;
                const float te = mod ? th[e] * scale : th[e];
                const float cl = p.Lf * te * 2.0f;
                vv = fminf(fmaxf(vv, -cl), cl);
                s[e] = fminf(fmaxf(floorf(vv / te), 0.0f), p.Lf);
                v[e] = vv - s[e] * te;
                if (p.alpha > 0.0f) th[e] = th[e] + a_eff * s[e] - a_eff * (th[e] - p.thr0);
            }
            V<VEC>::st(spikes + o, s);
        }
        V<VEC>::st(v_io + row * C + c0, v);
        V<VEC>::st(th_io + row * C + c0, th);
    }
}

// Training forward of the prosody GIF: as gif_prosody_kernel, also saving the pre-clamp potential a_t and
// the threshold theta_{t-1} of every step (everything else is recomputed in backward).
template <int VEC>
__global__ __launch_bounds__(256) void gif_prosody_train_fwd_kernel(GifP p, float strength,
                                                                    const float* __restrict__ h,
                                                                    const float* __restrict__ gains,
                                                                    float* __restrict__ spikes, float* v_io,
                                                                    float* th_io, float* __restrict__ save_a,
                                                                    float* __restrict__ save_th, int64_t R,
                                                                    int64_t T, int64_t C) {
    const int64_t cv = C / VEC, items = R * cv;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / cv, c0 = (it - row * cv) * VEC;
        float v[VEC], th[VEC];
        V<VEC>::ld(v_io + row * C + c0, v);
        V<VEC>::ld(th_io + row * C + c0, th);
        for (int64_t t = 0; t < T; ++t) {
            const int64_t o = (row * T + t) * C + c0;
            float x[VEC], s[VEC], a[VEC], thp[VEC];
            V<VEC>::ld(h + o, x);
            const bool mod = gains != nullptr;
            const float g = mod ? gains[row * T + t] : 1.0f;
            const float scale = fminf(fmaxf(1.0f - strength * (g - 1.0f), 0.5f), 1.5f);
            const float a_eff = mod ? p.alpha * g : p.alpha;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                thp[e] = th[e];
                const float i_t = mod ? x[e] * g : x[e];
                a[e] = v[e] * p.decay + i_t;
                const float te = mod ? th[e] * scale : th[e];
                const float cl = p.Lf * te * 2.0f;
                const float b = fminf(fmaxf(a[e], -cl), cl);
                s[e] = fminf(fmaxf(floorf(b / te), 0.0f), p.Lf);
                v[e] = b - s[e] * te;
                if (p.alpha > 0.0f) th[e] = th[e] + a_eff * s[e] - a_eff * (th[e] - p.thr0);
            }
            V<VEC>::st(spikes + o, s);
            V<VEC>::st(save_a + o, a);
            V<VEC>::st(save_th + o, thp);
        }
        V<VEC>::st(v_io + row * C + c0, v);
        V<VEC>::st(th_io + row * C + c0, th);
    }
}

// Backward of that loop (BPTT, reverse time).  Per step, with g = gains[row][t], scale = clamp(1 - str (g - 1),
// 0.5, 1.5), te = thp scale, ae = alpha g:
//   theta_t = thp + ae s - ae (thp - thr0);  v_t = b - s te;  s = surrogate(n), n = b / te;
//   b = clamp(a, -2 L te, 2 L te);  a = v_{t-1} decay + x g.
// Besides dL/dh and the initial-state gradients it returns dL/dgains[row][t] (summed over the channels with
// float atomics: g_gains must be zeroed by the caller): through the input gain (x), the threshold scale
// (-strength where the clamp is inactive) and the adaptation rate (alpha).
template <int VEC>
__global__ __launch_bounds__(256) void gif_prosody_bwd_kernel(GifP p, float strength,
                                                              const float* __restrict__ save_a,
                                                              const float* __restrict__ save_th,
                                                              const float* __restrict__ h,
                                                              const float* __restrict__ gains,
                                                              const float* __restrict__ g_spikes,
                                                              float* __restrict__ g_h, float* __restrict__ g_gains,
                                                              float* gv_io, float* gth_io, int64_t R, int64_t T,
                                                              int64_t C) {
    const int64_t cv = C / VEC, items = R * cv;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / cv, c0 = (it - row * cv) * VEC;
        float gv[VEC], gth[VEC];
        V<VEC>::ld(gv_io + row * C + c0, gv);
        V<VEC>::ld(gth_io + row * C + c0, gth);
        for (int64_t t = T - 1; t >= 0; --t) {
            const int64_t o = (row * T + t) * C + c0;
            float a[VEC], thp[VEC], gs[VEC], gh[VEC], x[VEC];
            V<VEC>::ld(save_a + o, a);
            V<VEC>::ld(save_th + o, thp);
            V<VEC>::ld(g_spikes + o, gs);
            V<VEC>::ld(h + o, x);
            const bool mod = gains != nullptr;
            const float g = mod ? gains[row * T + t] : 1.0f;
            const float raw = 1.0f - strength * (g - 1.0f);
            const float scale = fminf(fmaxf(raw, 0.5f), 1.5f);
            const bool scale_live = mod && raw >= 0.5f && raw <= 1.5f;   // torch.clamp passes the gradient at the bounds
            const float ae = mod ? p.alpha * g : p.alpha;
            float gg = 0.0f;                                             // this lane's share of dL/dgains[row][t]
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float te = mod ? thp[e] * scale : thp[e];
                const float cl = p.Lf * te * 2.0f;
                const float b = fminf(fmaxf(a[e], -cl), cl);
                const float n = b / te;
                const float s = fminf(fmaxf(floorf(n), 0.0f), p.Lf);
                float gth_prev = gth[e];
                float gs_tot = gs[e];
                if (p.alpha > 0.0f) {
                    gs_tot = gs_tot + ae * gth[e];
                    gth_prev = gth_prev - ae * gth[e];
                    if (mod) gg += gth[e] * (s - (thp[e] - p.thr0)) * p.alpha;
                }
                gs_tot = gs_tot - te * gv[e];                            // v_t = b - s te
                float gte = -s * gv[e];
                float gb = gv[e];
                const float dist = fabsf(n - rintf(n));
                const float tri = fminf(fmaxf(1.0f - 2.0f * dist, 0.0f), 1.0f);
                const float sur = (n >= 0.0f && n <= p.Lf + 1.0f) ? tri : 0.0f;
                const float gn = gs_tot * sur;
                gb = gb + gn / te;
                gte = gte + (-gn * b / (te * te));
                float ga = 0.0f, gcl = 0.0f;
                if (a[e] < -cl) gcl = -gb;
                else if (a[e] > cl) gcl = gb;
                else ga = gb;
                gte = gte + gcl * (2.0f * p.Lf);
                gth_prev = gth_prev + (mod ? gte * scale : gte);
                if (scale_live) gg += gte * thp[e] * (-strength);
                gh[e] = mod ? ga * g : ga;
                if (mod) gg += ga * x[e];
                gv[e] = ga * p.decay;
                gth[e] = gth_prev;
            }
            V<VEC>::st(g_h + o, gh);
            if (mod && g_gains) atomicAdd(g_gains + row * T + t, gg);
        }
        V<VEC>::st(gv_io + row * C + c0, gv);
        V<VEC>::st(gth_io + row * C + c0, gth);
    }
}

// ---- bf16 prosody GIF (ProsodyModulatedGIF under .bfloat16(): x, state AND gains bf16) ----------------------
// Every op of prosody_gif.py:64-101 rounds its fp32 result to bf16 (Python scalars enter as fp32), so the per
// (row, t) quantities are themselves rounded chains: scale = clamp(rb(1 - rb(str * rb(g - 1))), .5, 1.5),
// ae = rb(alpha g).  SAVE = the recording forward (a_t and theta_{t-1} as the bf16 values the graph holds).
struct ProsodyStep { float g, scale, ae; bool live; };
__device__ __forceinline__ ProsodyStep prosody_step_bf16(const GifP& p, float strength, const uint16_t* gains, int64_t at) {
    ProsodyStep r;
    if (gains == nullptr) { r.g = 1.0f; r.scale = 1.0f; r.ae = p.alpha; r.live = false; return r; }
    r.g = bf2f(gains[at]);
    const float raw = rb(1.0f - rb(strength * rb(r.g - 1.0f)));
    r.scale = fminf(fmaxf(raw, 0.5f), 1.5f);
    r.live = raw >= 0.5f && raw <= 1.5f;
    r.ae = rb(p.alpha * r.g);
    return r;
}
// forward intermediates of one step from (a, theta_prev): te, cl, b, n, s (all bf16 values)
__device__ __forceinline__ void prosody_bf16_mid(const GifP& p, bool mod, float scale, float a, float thp, float& te,
                                                 float& cl, float& b, float& n, float& s) {
    te = mod ? rb(thp * scale) : thp;
    cl = rb(rb(p.Lf * te) * 2.0f);
    b = fminf(fmaxf(a, -cl), cl);
    n = rb(b / te);
    s = fminf(fmaxf(floorf(n), 0.0f), p.Lf);
}

template <int VEC, bool SAVE>
__global__ __launch_bounds__(256) void gif_prosody_fwd_bf16_kernel(GifP p, float strength,
                                                                   const uint16_t* __restrict__ h,
                                                                   const uint16_t* __restrict__ gains,
                                                                   uint16_t* __restrict__ spikes, uint16_t* v_io,
                                                                   uint16_t* th_io, uint16_t* __restrict__ save_a,
                                                                   uint16_t* __restrict__ save_th, int64_t R,
                                                                   int64_t T, int64_t C) {
    const int64_t cv = C / VEC, items = R * cv;
    const bool mod = gains != nullptr;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / cv, c0 = (it - row * cv) * VEC;
        float v[VEC], th[VEC];
        VB<VEC>::ld(v_io + row * C + c0, v);
        VB<VEC>::ld(th_io + row * C + c0, th);
        for (int64_t t = 0; t < T; ++t) {
            const int64_t o = (row * T + t) * C + c0;
            float x[VEC], a[VEC], s[VEC], thp[VEC];
            VB<VEC>::ld(h + o, x);
            const ProsodyStep ps = prosody_step_bf16(p, strength, gains, row * T + t);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                thp[e] = th[e];
                const float i_t = mod ? rb(x[e] * ps.g) : x[e];
                a[e] = rb(rb(v[e] * p.decay) + i_t);
                float te, cl, b, n;
                prosody_bf16_mid(p, mod, ps.scale, a[e], th[e], te, cl, b, n, s[e]);
                v[e] = rb(b - rb(s[e] * te));
                if (p.alpha > 0.0f)
                    th[e] = rb(rb(th[e] + rb(ps.ae * s[e])) - rb(ps.ae * rb(th[e] - p.thr0)));
            }
            VB<VEC>::st(spikes + o, s);
            if (SAVE) { VB<VEC>::st(save_a + o, a); VB<VEC>::st(save_th + o, thp); }
        }
        VB<VEC>::st(v_io + row * C + c0, v);
        VB<VEC>::st(th_io + row * C + c0, th);
    }
}

// BPTT of that loop: the fp32 chain of gif_prosody_bwd_kernel evaluated on the forward's bf16-rounded
// intermediates; state gradients stay in fp32 registers over the T steps, g_h is rounded once on the way out,
// g_gains is accumulated in fp32 (the caller rounds it).
template <int VEC>
__global__ __launch_bounds__(256) void gif_prosody_bwd_bf16_kernel(GifP p, float strength,
                                                                   const uint16_t* __restrict__ save_a,
                                                                   const uint16_t* __restrict__ save_th,
                                                                   const uint16_t* __restrict__ h,
                                                                   const uint16_t* __restrict__ gains,
                                                                   const uint16_t* __restrict__ g_spikes,
                                                                   uint16_t* __restrict__ g_h, float* __restrict__ g_gains,
                                                                   uint16_t* gv_io, uint16_t* gth_io, int64_t R, int64_t T,
                                                                   int64_t C) {
    const int64_t cv = C / VEC, items = R * cv;
    const bool mod = gains != nullptr;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / cv, c0 = (it - row * cv) * VEC;
        float gv[VEC], gth[VEC];
        VB<VEC>::ld(gv_io + row * C + c0, gv);
        VB<VEC>::ld(gth_io + row * C + c0, gth);
        for (int64_t t = T - 1; t >= 0; --t) {
            const int64_t o = (row * T + t) * C + c0;
            float a[VEC], thp[VEC], gs[VEC], gh[VEC], x[VEC];
            VB<VEC>::ld(save_a + o, a);
            VB<VEC>::ld(save_th + o, thp);
            VB<VEC>::ld(g_spikes + o, gs);
            VB<VEC>::ld(h + o, x);
            const ProsodyStep ps = prosody_step_bf16(p, strength, gains, row * T + t);
            float gg = 0.0f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float te, cl, b, n, s;
                prosody_bf16_mid(p, mod, ps.scale, a[e], thp[e], te, cl, b, n, s);
                float gth_prev = gth[e];
                float gs_tot = gs[e];
                if (p.alpha > 0.0f) {
                    gs_tot = gs_tot + ps.ae * gth[e];
                    gth_prev = gth_prev - ps.ae * gth[e];
                    if (mod) gg += gth[e] * (s - (thp[e] - p.thr0)) * p.alpha;
                }
                gs_tot = gs_tot - te * gv[e];
                float gte = -s * gv[e];
                float gb = gv[e];
                const float dist = fabsf(n - rintf(n));
                const float tri = fminf(fmaxf(1.0f - 2.0f * dist, 0.0f), 1.0f);
                const float sur = (n >= 0.0f && n <= p.Lf + 1.0f) ? tri : 0.0f;
                const float gn = gs_tot * sur;
                gb = gb + gn / te;
                gte = gte + (-gn * b / (te * te));
                float ga = 0.0f, gcl = 0.0f;
                if (a[e] < -cl) gcl = -gb;
                else if (a[e] > cl) gcl = gb;
                else ga = gb;
                gte = gte + gcl * (2.0f * p.Lf);
                gth_prev = gth_prev + (mod ? gte * ps.scale : gte);
                if (ps.live) gg += gte * thp[e] * (-strength);
                gh[e] = mod ? ga * ps.g : ga;
                if (mod) gg += ga * x[e];
                gv[e] = ga * p.decay;
                gth[e] = gth_prev;
            }
            VB<VEC>::st(g_h + o, gh);
            if (mod && g_gains) atomicAdd(g_gains + row * T + t, gg);
        }
        VB<VEC>::st(gv_io + row * C + c0, gv);
        VB<VEC>::st(gth_io + row * C + c0, gth);
    }
}

// ---- bf16 LIF (VectorizedLIFNeuron under .bfloat16(): beta, threshold, slope and the input bf16) ------------
// neuron.py:135-139 with each op rounded: mm = rb(rb(beta m) + x); pre = rb(mm - thr); s = pre > 0;
// m' = rb(mm - s thr).  TRAIN: one step that also saves pre (mem_in is left alone); else the T-step loop.
template <int VEC, bool TRAIN>
__global__ __launch_bounds__(256) void lif_bf16_kernel(const uint16_t* __restrict__ x, const uint16_t* mem_in,
                                                       const uint16_t* __restrict__ beta,
                                                       const uint16_t* __restrict__ thr, uint16_t* __restrict__ spk,
                                                       uint16_t* mem_out, uint16_t* __restrict__ pre, int64_t B,
                                                       int64_t T, int64_t C) {
    const int64_t cv = C / VEC, items = B * cv;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / cv, c0 = (it - row * cv) * VEC;
        float m[VEC], bt[VEC], th[VEC];
        VB<VEC>::ld(mem_in + row * C + c0, m);
        VB<VEC>::ld(beta + c0, bt);
        VB<VEC>::ld(thr + c0, th);
        for (int64_t t = 0; t < T; ++t) {
            const int64_t o = (row * T + t) * C + c0;
            float xv[VEC], s[VEC], pr[VEC];
            VB<VEC>::ld(x + o, xv);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float mm = rb(rb(bt[e] * m[e]) + xv[e]);
                pr[e] = rb(mm - th[e]);
                s[e] = pr[e] > 0.0f ? 1.0f : 0.0f;
                m[e] = rb(mm - s[e] * th[e]);
            }
            VB<VEC>::st(spk + o, s);
            if (TRAIN) VB<VEC>::st(pre + o, pr);
        }
        VB<VEC>::st(mem_out + row * C + c0, m);
    }
}

// the fp32 formulas of lif_bwd_kernel on the saved bf16 pre; g_x, g_mem_prev rounded once, raw_slope fp32
template <int VEC>
__global__ __launch_bounds__(256) void lif_bwd_bf16_kernel(const uint16_t* __restrict__ pre,
                                                           const uint16_t* __restrict__ g_spk,
                                                           const uint16_t* __restrict__ g_mem,
                                                           const uint16_t* __restrict__ beta,
                                                           const uint16_t* __restrict__ thr,
                                                           const uint16_t* __restrict__ slope,
                                                           uint16_t* __restrict__ g_x, uint16_t* __restrict__ g_mem_prev,
                                                           float* __restrict__ raw_slope, int64_t B, int64_t C) {
    const int64_t cv = C / VEC, items = B * cv;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / cv, c0 = (it - row * cv) * VEC;
        float pr[VEC], gs[VEC], gm[VEC], bt[VEC], th[VEC], sl[VEC], gx[VEC], gp[VEC];
        VB<VEC>::ld(pre + row * C + c0, pr);
        VB<VEC>::ld(g_spk + row * C + c0, gs);
        VB<VEC>::ld(g_mem + row * C + c0, gm);
        VB<VEC>::ld(beta + c0, bt);
        VB<VEC>::ld(thr + c0, th);
        VB<VEC>::ld(slope + c0, sl);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float g_s = gs[e] - gm[e] * th[e];
            const float den = fabsf(sl[e] * pr[e]) + 1.0f;
            const float g_pre = g_s * (sl[e] / (den * den));
            const float g_m = gm[e] + g_pre;
            gx[e] = g_m;
            gp[e] = bt[e] * g_m;
            const float ap = fabsf(pr[e]);
            const float sg = pr[e] > 0.0f ? 1.0f : (pr[e] < 0.0f ? -1.0f : 0.0f);
            const float den2 = sl[e] * ap + 1.0f;
            raw_slope[row * C + c0 + e] = -g_s * ap * sg / (den2 * den2);
        }
        VB<VEC>::st(g_x + row * C + c0, gx);
        VB<VEC>::st(g_mem_prev + row * C + c0, gp);
    }
}

}  // namespace

#define AURA_VEC_DISPATCH(KERNEL, ITEMS4, ITEMS1, VECOK, ...)                                       \
    do {                                                                                            \
        if (VECOK) hipLaunchKernelGGL((KERNEL<4>), dim3(grid_for(ITEMS4)), dim3(256), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<1>), dim3(grid_for(ITEMS1)), dim3(256), 0, s, __VA_ARGS__);  \
    } while (0)

extern "C" {

int aura_gif_train_forward(const float* h, float* spikes, float* v, float* theta, float* save_a,
                           float* save_theta, float decay, int L, float alpha, float threshold,
                           int64_t rows, int64_t T, int64_t H, void* stream) {
    if (rows < 0 || T < 0 || H < 0 || L < 0) return AURA_E_INVAL;
    if (rows == 0 || H == 0) return AURA_OK;
    if (!v || !theta || (T && (!h || !spikes || !save_a || !save_theta))) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = H % 4 == 0 && aligned16(h) && aligned16(spikes) && aligned16(v) && aligned16(theta) &&
                     aligned16(save_a) && aligned16(save_theta);
    const GifP p{decay, (float)L, alpha, threshold};
    AURA_VEC_DISPATCH(gif_train_fwd_kernel, rows * (H / 4), rows * H, vec, p, h, spikes, v, theta, save_a,
                      save_theta, rows, T, H);
    return check_launch();
}

int aura_gif_backward(const float* save_a, const float* save_theta, const float* g_spikes, float* g_h,
                      float* g_v, float* g_theta, float decay, int L, float alpha, float threshold,
                      int64_t rows, int64_t T, int64_t H, void* stream) {
    if (rows < 0 || T < 0 || H < 0 || L < 0) return AURA_E_INVAL;
    if (rows == 0 || H == 0) return AURA_OK;
    if (!g_v || !g_theta || (T && (!save_a || !save_theta || !g_spikes || !g_h))) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = H % 4 == 0 && aligned16(save_a) && aligned16(save_theta) && aligned16(g_spikes) &&
                     aligned16(g_h) && aligned16(g_v) && aligned16(g_theta);
    const GifP p{decay, (float)L, alpha, threshold};
    AURA_VEC_DISPATCH(gif_bwd_kernel, rows * (H / 4), rows * H, vec, p, save_a, save_theta, g_spikes, g_h,
                      g_v, g_theta, rows, T, H);
    return check_launch();
}

int aura_gif_train_forward_bf16(const uint16_t* h, uint16_t* spikes, uint16_t* v, uint16_t* theta, uint16_t* save_a,
                                uint16_t* save_theta, float decay, int L, float alpha, float threshold,
                                int64_t rows, int64_t T, int64_t H, void* stream) {
    if (rows < 0 || T < 0 || H < 0 || L < 0 || L > 256) return AURA_E_INVAL;   // spike counts up to 256 are exact in bf16
    if (rows == 0 || H == 0) return AURA_OK;
    if (!v || !theta || (T && (!h || !spikes || !save_a || !save_theta))) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = H % 8 == 0 && aligned16(h) && aligned16(spikes) && aligned16(v) && aligned16(theta) &&
                     aligned16(save_a) && aligned16(save_theta);
    const GifP p{decay, (float)L, alpha, threshold};
    if (vec) hipLaunchKernelGGL((gif_train_fwd_bf16_kernel<8>), dim3(grid_for(rows * (H / 8))), dim3(256), 0, s, p, h,
                                spikes, v, theta, save_a, save_theta, rows, T, H);
    else hipLaunchKernelGGL((gif_train_fwd_bf16_kernel<1>), dim3(grid_for(rows * H)), dim3(256), 0, s, p, h, spikes, v,
                            theta, save_a, save_theta, rows, T, H);
    return check_launch();
}

int aura_gif_backward_bf16(const uint16_t* save_a, const uint16_t* save_theta, const uint16_t* g_spikes, uint16_t* g_h,
                           uint16_t* g_v, uint16_t* g_theta, float decay, int L, float alpha, float threshold,
                           int64_t rows, int64_t T, int64_t H, void* stream) {
    if (rows < 0 || T < 0 || H < 0 || L < 0 || L > 256) return AURA_E_INVAL;
    if (rows == 0 || H == 0) return AURA_OK;
    if (!g_v || !g_theta || (T && (!save_a || !save_theta || !g_spikes || !g_h))) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = H % 8 == 0 && aligned16(save_a) && aligned16(save_theta) && aligned16(g_spikes) &&
                     aligned16(g_h) && aligned16(g_v) && aligned16(g_theta);
    const GifP p{decay, (float)L, alpha, threshold};
    if (vec) hipLaunchKernelGGL((gif_bwd_bf16_kernel<8>), dim3(grid_for(rows * (H / 8))), dim3(256), 0, s, p, save_a,
                                save_theta, g_spikes, g_h, g_v, g_theta, rows, T, H);
    else hipLaunchKernelGGL((gif_bwd_bf16_kernel<1>), dim3(grid_for(rows * H)), dim3(256), 0, s, p, save_a, save_theta,
                            g_spikes, g_h, g_v, g_theta, rows, T, H);
    return check_launch();
}

int aura_lif_train_forward(const float* x, const float* mem_in, const float* beta,
                           const float* threshold, float* spikes, float* mem_out, float* pre,
                           int64_t B, int64_t size, void* stream) {
    if (B < 0 || size < 0) return AURA_E_INVAL;
    if (B == 0 || size == 0) return AURA_OK;
    if (!x || !mem_in || !beta || !threshold || !spikes || !mem_out || !pre) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = size % 4 == 0 && aligned16(x) && aligned16(mem_in) && aligned16(beta) &&
                     aligned16(threshold) && aligned16(spikes) && aligned16(mem_out) && aligned16(pre);
    AURA_VEC_DISPATCH(lif_train_fwd_kernel, B * (size / 4), B * size, vec, x, mem_in, beta, threshold,
                      spikes, mem_out, pre, B, size);
    return check_launch();
}

int aura_lif_backward(const float* pre, const float* g_spikes, const float* g_mem, const float* beta,
                      const float* threshold, const float* slope, float* g_x, float* g_mem_prev,
                      float* raw_slope, int64_t B, int64_t size, void* stream) {
    if (B < 0 || size < 0) return AURA_E_INVAL;
    if (B == 0 || size == 0) return AURA_OK;
    if (!pre || !g_spikes || !g_mem || !beta || !threshold || !slope || !g_x || !g_mem_prev || !raw_slope)
        return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = size % 4 == 0 && aligned16(pre) && aligned16(g_spikes) && aligned16(g_mem) &&
                     aligned16(beta) && aligned16(threshold) && aligned16(slope) && aligned16(g_x) &&
                     aligned16(g_mem_prev) && aligned16(raw_slope);
    AURA_VEC_DISPATCH(lif_bwd_kernel, B * (size / 4), B * size, vec, pre, g_spikes, g_mem, beta, threshold,
                      slope, g_x, g_mem_prev, raw_slope, B, size);
    return check_launch();
}

int aura_gif_prosody_run(const float* h, const float* gains, float* spikes, float* v, float* theta,
                         float decay, int L, float alpha, float threshold, float strength,
                         int64_t rows, int64_t T, int64_t H, void* stream) {
    if (rows < 0 || T < 0 || H < 0 || L < 0) return AURA_E_INVAL;
    if (rows == 0 || H == 0) return AURA_OK;
    if (!v || !theta || (T && (!h || !spikes))) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = H % 4 == 0 && aligned16(h) && aligned16(spikes) && aligned16(v) && aligned16(theta);
    const GifP p{decay, (float)L, alpha, threshold};
    AURA_VEC_DISPATCH(gif_prosody_kernel, rows * (H / 4), rows * H, vec, p, strength, h, gains, spikes, v,
                      theta, rows, T, H);
    return check_launch();
}

int aura_gif_prosody_train_forward(const float* h, const float* gains, float* spikes, float* v, float* theta,
                                   float* save_a, float* save_theta, float decay, int L, float alpha,
                                   float threshold, float strength, int64_t rows, int64_t T, int64_t H,
                                   void* stream) {
    if (rows < 0 || T < 0 || H < 0 || L < 0) return AURA_E_INVAL;
    if (rows == 0 || H == 0) return AURA_OK;
    if (!v || !theta || (T && (!h || !spikes || !save_a || !save_theta))) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = H % 4 == 0 && aligned16(h) && aligned16(spikes) && aligned16(v) && aligned16(theta) &&
                     aligned16(save_a) && aligned16(save_theta);
    const GifP p{decay, (float)L, alpha, threshold};
    AURA_VEC_DISPATCH(gif_prosody_train_fwd_kernel, rows * (H / 4), rows * H, vec, p, strength, h, gains, spikes,
                      v, theta, save_a, save_theta, rows, T, H);
    return check_launch();
}

int aura_gif_prosody_backward(const float* save_a, const float* save_theta, const float* h, const float* gains,
                              const float* g_spikes, float* g_h, float* g_gains, float* g_v, float* g_theta,
                              float decay, int L, float alpha, float threshold, float strength, int64_t rows,
                              int64_t T, int64_t H, void* stream) {
    if (rows < 0 || T < 0 || H < 0 || L < 0) return AURA_E_INVAL;
    if (rows == 0 || H == 0) return AURA_OK;
    if (!g_v || !g_theta || (T && (!save_a || !save_theta || !h || !g_spikes || !g_h))) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = H % 4 == 0 && aligned16(save_a) && aligned16(save_theta) && aligned16(h) &&
                     aligned16(g_spikes) && aligned16(g_h) && aligned16(g_v) && aligned16(g_theta);
    const GifP p{decay, (float)L, alpha, threshold};
    AURA_VEC_DISPATCH(gif_prosody_bwd_kernel, rows * (H / 4), rows * H, vec, p, strength, save_a, save_theta, h,
                      gains, g_spikes, g_h, g_gains, g_v, g_theta, rows, T, H);
    return check_launch();
}

#define AURA_VEC8_DISPATCH(KERNEL8, KERNEL1, ROWS, H, VECOK, ...)                                        \
    do {                                                                                                \
        if (VECOK) hipLaunchKernelGGL(KERNEL8, dim3(grid_for((ROWS) * ((H) / 8))), dim3(256), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL1, dim3(grid_for((ROWS) * (H))), dim3(256), 0, s, __VA_ARGS__);    \
    } while (0)

int aura_gif_prosody_run_bf16(const uint16_t* h, const uint16_t* gains, uint16_t* spikes, uint16_t* v, uint16_t* theta,
                              float decay, int L, float alpha, float threshold, float strength, int64_t rows,
                              int64_t T, int64_t H, void* stream) {
    if (rows < 0 || T < 0 || H < 0 || L < 0 || L > 256) return AURA_E_INVAL;
    if (rows == 0 || H == 0) return AURA_OK;
    if (!v || !theta || (T && (!h || !spikes))) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = H % 8 == 0 && aligned16(h) && aligned16(spikes) && aligned16(v) && aligned16(theta);
    const GifP p{decay, (float)L, alpha, threshold};
    uint16_t* none = nullptr;
    AURA_VEC8_DISPATCH((gif_prosody_fwd_bf16_kernel<8, false>), (gif_prosody_fwd_bf16_kernel<1, false>), rows, H, vec, p,
                       strength, h, gains, spikes, v, theta, none, none, rows, T, H);
    return check_launch();
}

int aura_gif_prosody_train_forward_bf16(const uint16_t* h, const uint16_t* gains, uint16_t* spikes, uint16_t* v,
                                        uint16_t* theta, uint16_t* save_a, uint16_t* save_theta, float decay, int L,
                                        float alpha, float threshold, float strength, int64_t rows, int64_t T,
                                        int64_t H, void* stream) {
    if (rows < 0 || T < 0 || H < 0 || L < 0 || L > 256) return AURA_E_INVAL;
    if (rows == 0 || H == 0) return AURA_OK;
    if (!v || !theta || (T && (!h || !spikes || !save_a || !save_theta))) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = H % 8 == 0 && aligned16(h) && aligned16(spikes) && aligned16(v) && aligned16(theta) &&
                     aligned16(save_a) && aligned16(save_theta);
    const GifP p{decay, (float)L, alpha, threshold};
    AURA_VEC8_DISPATCH((gif_prosody_fwd_bf16_kernel<8, true>), (gif_prosody_fwd_bf16_kernel<1, true>), rows, H, vec, p,
                       strength, h, gains, spikes, v, theta, save_a, save_theta, rows, T, H);
    return check_launch();
}

int aura_gif_prosody_backward_bf16(const uint16_t* save_a, const uint16_t* save_theta, const uint16_t* h,
                                   const uint16_t* gains, const uint16_t* g_spikes, uint16_t* g_h, float* g_gains,
                                   uint16_t* g_v, uint16_t* g_theta, float decay, int L, float alpha, float threshold,
                                   float strength, int64_t rows, int64_t T, int64_t H, void* stream) {
    if (rows < 0 || T < 0 || H < 0 || L < 0 || L > 256) return AURA_E_INVAL;
    if (rows == 0 || H == 0) return AURA_OK;
    if (!g_v || !g_theta || (T && (!save_a || !save_theta || !h || !g_spikes || !g_h))) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = H % 8 == 0 && aligned16(save_a) && aligned16(save_theta) && aligned16(h) &&
                     aligned16(g_spikes) && aligned16(g_h) && aligned16(g_v) && aligned16(g_theta);
    const GifP p{decay, (float)L, alpha, threshold};
    AURA_VEC8_DISPATCH((gif_prosody_bwd_bf16_kernel<8>), (gif_prosody_bwd_bf16_kernel<1>), rows, H, vec, p, strength,
                       save_a, save_theta, h, gains, g_spikes, g_h, g_gains, g_v, g_theta, rows, T, H);
    return check_launch();
}

int aura_lif_run_bf16(const uint16_t* x, uint16_t* spikes, uint16_t* mem, const uint16_t* beta,
                      const uint16_t* threshold, int64_t B, int64_t T, int64_t size, void* stream) {
    if (B < 0 || T < 0 || size < 0) return AURA_E_INVAL;
    if (B == 0 || size == 0 || T == 0) return AURA_OK;
    if (!x || !spikes || !mem || !beta || !threshold) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = size % 8 == 0 && aligned16(x) && aligned16(spikes) && aligned16(mem) && aligned16(beta) &&
                     aligned16(threshold);
    uint16_t* none = nullptr;
    const uint16_t* mem_in = mem;
    AURA_VEC8_DISPATCH((lif_bf16_kernel<8, false>), (lif_bf16_kernel<1, false>), B, size, vec, x, mem_in, beta, threshold,
                       spikes, mem, none, B, T, size);
    return check_launch();
}

int aura_lif_train_forward_bf16(const uint16_t* x, const uint16_t* mem_in, const uint16_t* beta,
                                const uint16_t* threshold, uint16_t* spikes, uint16_t* mem_out, uint16_t* pre,
                                int64_t B, int64_t size, void* stream) {
    if (B < 0 || size < 0) return AURA_E_INVAL;
    if (B == 0 || size == 0) return AURA_OK;
    if (!x || !mem_in || !beta || !threshold || !spikes || !mem_out || !pre) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = size % 8 == 0 && aligned16(x) && aligned16(mem_in) && aligned16(beta) &&
                     aligned16(threshold) && aligned16(spikes) && aligned16(mem_out) && aligned16(pre);
    const int64_t one = 1;
    AURA_VEC8_DISPATCH((lif_bf16_kernel<8, true>), (lif_bf16_kernel<1, true>), B, size, vec, x, mem_in, beta, threshold,
                       spikes, mem_out, pre, B, one, size);
    return check_launch();
}

int aura_lif_backward_bf16(const uint16_t* pre, const uint16_t* g_spikes, const uint16_t* g_mem, const uint16_t* beta,
                           const uint16_t* threshold, const uint16_t* slope, uint16_t* g_x, uint16_t* g_mem_prev,
                           float* raw_slope, int64_t B, int64_t size, void* stream) {
    if (B < 0 || size < 0) return AURA_E_INVAL;
    if (B == 0 || size == 0) return AURA_OK;
    if (!pre || !g_spikes || !g_mem || !beta || !threshold || !slope || !g_x || !g_mem_prev || !raw_slope)
        return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = size % 8 == 0 && aligned16(pre) && aligned16(g_spikes) && aligned16(g_mem) &&
                     aligned16(beta) && aligned16(threshold) && aligned16(slope) && aligned16(g_x) &&
                     aligned16(g_mem_prev);
    AURA_VEC8_DISPATCH((lif_bwd_bf16_kernel<8>), (lif_bwd_bf16_kernel<1>), B, size, vec, pre, g_spikes, g_mem, beta,
                       threshold, slope, g_x, g_mem_prev, raw_slope, B, size);
    return check_launch();
}

}  // extern "C"
