/*
 * aura_oracle.c -- scalar C restatement of the reference's hot-path arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/ as an independent checker
 * of the PyTorch-op restatement in oracle/aura_oracle.py (same results bit for bit in fp32 and
 * in per-op-rounded bf16) and by bench.py's cpu_baseline leg ("port", OpenMP over host cores).
 * Never linked into or loaded by the product library.
 *
 * Built with -ffp-contract=off -fno-fast-math: every expression is evaluated in the reference's
 * op order in IEEE fp32 with no fused multiply-add, like the eager PyTorch ops it restates.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* src/base/neuron.py:186-195 */
void oracle_izh_run_nt(const float* I, float* S, float* v, float* u, float a, float b, float c,
                       float d, float dt, int64_t N, int64_t T) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        float vv = v[n], uu = u[n];
        for (int64_t t = 0; t < T; ++t) {
            float i_t = I[n * T + t];
            float dv = 0.04f * vv * vv + 5.0f * vv + 140.0f - uu + i_t;
            vv = vv + dt * dv;
            float du = a * (b * vv - uu);
            uu = uu + dt * du;
            int spk = vv >= 30.0f;
            S[n * T + t] = spk ? 1.0f : 0.0f;
            if (spk) { vv = c; uu = uu + d; }
        }
        v[n] = vv; u[n] = uu;
    }
}

/* src/base/neuron.py:237-247; params = {tau_m,E_L,V_T,Delta_T,R,tau_w,a,b,V_reset,V_spike,dt} */
void oracle_adex_run_nt(const float* I, float* S, float* V, float* w, const float* p, int64_t N,
                        int64_t T) {
    const float tau_m = p[0], E_L = p[1], V_T = p[2], D_T = p[3], R = p[4], tau_w = p[5], a = p[6],
                b = p[7], V_reset = p[8], V_spike = p[9], dt = p[10];
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        float vv = V[n], ww = w[n];
        for (int64_t t = 0; t < T; ++t) {
            float i_t = I[n * T + t];
            float exp_term = D_T * expf((vv - V_T) / D_T);
            float dV = (-(vv - E_L) + exp_term - R * ww + R * i_t) / tau_m;
            vv = vv + dt * dV;
            float dw = (a * (vv - E_L) - ww) / tau_w;
            ww = ww + dt * dw;
            int spk = vv >= V_spike;
            S[n * T + t] = spk ? 1.0f : 0.0f;
            if (spk) { vv = V_reset; ww = ww + b; }
        }
        V[n] = vv; w[n] = ww;
    }
}

/* src/base/neuron.py:135-137 over x[B][T][size] */
void oracle_lif_run(const float* x, float* S, float* mem, const float* beta, const float* thr,
                    int64_t B, int64_t T, int64_t size) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b)
        for (int64_t c = 0; c < size; ++c) {
            float m = mem[b * size + c];
            for (int64_t t = 0; t < T; ++t) {
                m = beta[c] * m + x[(b * T + t) * size + c];
                float spk = (m - thr[c]) > 0.0f ? 1.0f : 0.0f;
                m = m - spk * thr[c];
                S[(b * T + t) * size + c] = spk;
            }
            mem[b * size + c] = m;
        }
}

static inline float bf16_round(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) { u |= 0x00400000u; u &= 0xffff0000u; }  /* quiet NaN */
    else { u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u; }
    memcpy(&x, &u, 4);
    return x;
}
static inline float bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline uint16_t f_to_bf16(float x) { x = bf16_round(x); uint32_t u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); }

#define GIF_STEP(R)                                                                    \
    vv = R(R(vv * decay) + i_t);                                                       \
    {                                                                                  \
        float cl = R(R(Lf * th) * 2.0f);                                               \
        vv = fminf(fmaxf(vv, -cl), cl);                                                \
        float nv = R(vv / R(th + 1e-6f));                                              \
        spike = fminf(fmaxf(floorf(nv), 0.0f), Lf);                                    \
        vv = R(vv - R(spike * th));                                                    \
        if (alpha > 0.0f) th = R(R(th + R(alpha * spike)) - R(alpha * R(th - thr0))); \
    }
#define IDENT(x) (x)

/* src/core/language_zone/gif_neuron.py:56-67, fp32; h[rows][T][H] */
void oracle_gif_run_f32(const float* h, float* S, float* v, float* theta, float decay, int L,
                        float alpha, float thr0, int64_t rows, int64_t T, int64_t H) {
    const float Lf = (float)L;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < H; ++c) {
            float vv = v[r * H + c], th = theta[r * H + c], spike;
            for (int64_t t = 0; t < T; ++t) {
                float i_t = h[(r * T + t) * H + c];
                GIF_STEP(IDENT)
                S[(r * T + t) * H + c] = spike;
            }
            v[r * H + c] = vv; theta[r * H + c] = th;
        }
}

/* same loop on bf16 tensors: every op rounds its fp32 result to bf16 (state dtype follows the
 * input, gif_neuron.py:46-47); Python-float scalars stay fp32 (PyTorch's reduced-float scalar path) */
void oracle_gif_run_bf16(const uint16_t* h, uint16_t* S, uint16_t* v, uint16_t* theta, float decay,
                         int L, float alpha, float thr0, int64_t rows, int64_t T, int64_t H) {
    const float Lf = (float)L;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < H; ++c) {
            float vv = bf16_to_f(v[r * H + c]), th = bf16_to_f(theta[r * H + c]), spike;
            for (int64_t t = 0; t < T; ++t) {
                float i_t = bf16_to_f(h[(r * T + t) * H + c]);
                GIF_STEP(bf16_round)
                S[(r * T + t) * H + c] = f_to_bf16(spike);
            }
            v[r * H + c] = f_to_bf16(vv); theta[r * H + c] = f_to_bf16(th);
        }
}

/* combined scores of src/core/hippocampal.py:272-303 for ONE query against rows [0,N): the
 * reference's per-query cost (bank re-normalised for every query, :278) is kept on purpose --
 * this is the CPU baseline's algorithm, not an optimised one. */
void oracle_knn_scores(const float* bank, const float* meta, const float* q, float now, int64_t N,
                       int64_t D, float* out) {
    double qs = 0.0;
    float qn = 0.0f;
    for (int64_t j = 0; j < D; ++j) qs += (double)q[j] * q[j];
    qn = fmaxf((float)sqrt(qs), 1e-12f);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const float* m = bank + i * D;
        float ms = 0.0f;
        for (int64_t j = 0; j < D; ++j) ms += m[j] * m[j];
        const float mn = fmaxf(sqrtf(ms), 1e-12f);
        float dot = 0.0f;
        for (int64_t j = 0; j < D; ++j) dot += (q[j] / qn) * (m[j] / mn);
        const float age = now - meta[i * 4 + 1];
        const float temporal = expf(-age / 3600.0f);
        out[i] = (0.5f * dot + 0.2f * temporal) * meta[i * 4 + 0];
    }
}

/* top-k of scores[0..N) descending, ties -> lower index (partial selection sort, k small) */
void oracle_topk(const float* scores, int64_t N, int k, float* out_s, int32_t* out_i) {
    uint8_t* taken = (uint8_t*)calloc((size_t)N, 1);
    for (int r = 0; r < k; ++r) {
        int64_t best = -1;
        for (int64_t i = 0; i < N; ++i)
            if (!taken[i] && (best < 0 || scores[i] > scores[best])) best = i;
        out_s[r] = best >= 0 ? scores[best] : -INFINITY;
        out_i[r] = (int32_t)best;
        if (best >= 0) taken[best] = 1;
    }
    free(taken);
}

/* SLEEF 3.x Sleef_expf{8,16}_u10 (the vector form with fused multiply-adds: range reduction by ln 2 in two
 * parts, degree-6 polynomial, ldexp in two halves), restated to find out what the reference's torch.exp is:
 * it equals libtorch_cpu's Sleef_expf16_u10 bit for bit, and torch.exp is NOT it (it is MKL's vmsExp) --
 * tests/test_oracle_known_answers.py.  fmaf() is libm's correctly rounded fused multiply-add. */
static inline float oracle_pow2if(int q) { int32_t i = (int32_t)(q + 0x7f) << 23; float f; memcpy(&f, &i, 4); return f; }
void oracle_sleef_expf_u10(const float* x, float* out, int64_t n) {
    const float R_LN2f = 1.442695040888963407359924681001892137426645954152985934135449406931f;
    const float L2Uf = 0.693145751953125f, L2Lf = 1.428606765330187045e-06f;
    for (int64_t i = 0; i < n; ++i) {
        const float d = x[i];
        const int q = (int)rintf(d * R_LN2f);
        float s = fmaf((float)q, -L2Uf, d);
        s = fmaf((float)q, -L2Lf, s);
        float u = 0.000198527617612853646278381f;
        u = fmaf(u, s, 0.00139304355252534151077271f);
        u = fmaf(u, s, 0.00833336077630519866943359f);
        u = fmaf(u, s, 0.0416664853692054748535156f);
        u = fmaf(u, s, 0.166666671633720397949219f);
        u = fmaf(u, s, 0.5f);
        u = 1.0f + fmaf(s * s, u, s);
        u = u * oracle_pow2if(q >> 1) * oracle_pow2if(q - (q >> 1));
        if (d < -104.0f) u = 0.0f;
        if (d > 100.0f) u = INFINITY;
        out[i] = u;
    }
}
