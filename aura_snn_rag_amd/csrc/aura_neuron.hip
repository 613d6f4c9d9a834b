// aura_neuron.hip -- fused membrane-update loops for gfx950 (MI355X, wave64).
//
// One launch runs the WHOLE T loop of a neuron population with the state (v,u / V,w / mem /
// v,theta) held in registers, so HBM sees each input current once and each output spike once:
//   Izhikevich / AdEx fp32 : 4 B in + 4 B out + 16/T B state per neuron-timestep
//   LIF single step fp32   : 16 B;   GIF fp32: 8 B + 16/T;   GIF bf16: 4 B + 8/T
// (SURVEY.md section 8d).  All of these kernels are HBM-bandwidth bound; there is no matrix work
// here, so no MFMA.  Two data layouts exist in the reference:
//   * time-contiguous   I[N][T]    (IzhikevichNeuron / AdExNeuron 1-D/2-D inputs) -- a lane owns a
//     neuron but HBM wants lanes along t, so each wave transposes 64x32 tiles through LDS;
//   * channel-contiguous x[R][T][C] (3-D inputs, LIF, GIF) -- a lane owns VEC adjacent channels
//     and streams 16-byte vectors down the T axis, fully coalesced with no LDS.
//
// Arithmetic contract: every expression is evaluated in the reference's op order in IEEE fp32
// with NO fused multiply-add (the reference runs unfused eager PyTorch ops and floor()/>= turn a
// 1-ulp difference into a spike).  The file is compiled with -ffp-contract=off and repeats the
// pragma below; division and exp are the correctly rounded / OCML forms.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/aura_hip.h"

#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------------------------------
// Neuron models.  Lane = per-neuron registers (state s0,s1 + per-channel constants).
// ------------------------------------------------------------------------------------------

// src/base/neuron.py:186-195
struct IzhModel {
    float a, b, c, d, dt;
    static constexpr int NS = 2;
    struct Lane { float s0, s1; };
    __device__ __forceinline__ void init(Lane&, int64_t) const {}
    __device__ __forceinline__ float step(Lane& l, float i_t) const {
        float v = l.s0, u = l.s1;
        float dv = 0.04f * v * v + 5.0f * v + 140.0f - u + i_t;
        v = v + dt * dv;
        float du = a * (b * v - u);
        u = u + dt * du;
        bool spk = v >= 30.0f;
        l.s0 = spk ? c : v;
        l.s1 = spk ? u + d : u;
        return spk ? 1.0f : 0.0f;
    }
};

// src/base/neuron.py:237-247
struct AdExModel {
    float tau_m, E_L, V_T, Delta_T, R, tau_w, a, b, V_reset, V_spike, dt;
    static constexpr int NS = 2;
    struct Lane { float s0, s1; };
    __device__ __forceinline__ void init(Lane&, int64_t) const {}
    __device__ __forceinline__ float step(Lane& l, float i_t) const {
        float V = l.s0, w = l.s1;
        // The reference's torch.exp on CPU is Intel MKL's vmsExp (high-accuracy mode; checked bit for bit against
        // libtorch_cpu's export over 10^6 arguments by the known-answer tests) -- closed source, within
        // ~0.51 ulp.  OCML's expf (1 ulp) disagrees with it on ~10 % of the arguments, SLEEF's expf_u10 (restated
        // and checked bit for bit against libtorch's Sleef_expf16_u10 -- but that is not what torch.exp calls) on
        // 9.5 %; the CORRECTLY ROUNDED float exp on 1.07 %, each by one ulp: the closest a portable
        // implementation gets.  exp in fp64 rounded once to fp32 is correctly rounded except where the fp64
        // result sits within 2^-29 of a rounding boundary.  (The loop is HBM-bound; the fp64 polynomial rides
        // in its shadow.)
        float exp_term = Delta_T * (float)exp((double)((V - V_T) / Delta_T));
        float dV = (-(V - E_L) + exp_term - R * w + R * i_t) / tau_m;
        V = V + dt * dV;
        float dw = (a * (V - E_L) - w) / tau_w;
        w = w + dt * dw;
        bool spk = V >= V_spike;
        l.s0 = spk ? V_reset : V;
        l.s1 = spk ? w + b : w;
        return spk ? 1.0f : 0.0f;
    }
};

// src/base/neuron.py:135-137 (forward value of the surrogate, :77)
struct LifModel {
    const float* beta;
    const float* thr;
    static constexpr int NS = 1;
    struct Lane { float s0, s1, beta, thr; };
    __device__ __forceinline__ void init(Lane& l, int64_t c) const {
        l.beta = beta[c];
        l.thr = thr[c];
    }
    __device__ __forceinline__ float step(Lane& l, float x) const {
        float mem = l.beta * l.s0 + x;
        float spk = (mem - l.thr) > 0.0f ? 1.0f : 0.0f;
        l.s0 = mem - spk * l.thr;
        return spk;
    }
};

__device__ __forceinline__ float round_bf16(float x) {
    // round-to-nearest-even through the hardware convert (v_cvt_pk_bf16_f32); NaN stays NaN
    return static_cast<float>(static_cast<__bf16>(x));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two roundings for the price of one convert: v_cvt_pk_bf16_f32 packs both results
__device__ __forceinline__ void round_bf16_pair(float& a, float& b) {
    const f32x2 f = {a, b};
    const f32x2 g = __builtin_convertvector(__builtin_convertvector(f, bf16x2), f32x2);
    a = g.x;
    b = g.y;
}

// src/core/language_zone/gif_neuron.py:56-67.  BF16 = state and every intermediate are bf16
// tensors in the reference, i.e. each op rounds its fp32 result to bf16.
//
// Two exact shortcuts of the bf16 loop (13 roundings per step):
//  * the quotient.  The reference divides two bf16 tensors: RNE_bf16(RN_fp32(v / t)).  v and t carry
//    8-bit significands, so the real quotient x = (mv / mt) 2^e is never a bf16 rounding midpoint
//    (a dyadic mv / mt has at most 8 significant bits) and lies at least 2^-17 (relative) away from
//    every midpoint: |mv 2^s 256 - N mt| >= 1 for odd N, mt <= 255.  Any approximation of x that is
//    good to 2^-18 therefore rounds to the same bf16 value: v * rcp(t) (v_rcp_f32: 1 ulp) replaces
//    the ~10-instruction IEEE division, bit for bit (the whole bf16 GIF suite, and the exhaustive
//    significand-pair proof in the CPU test suite, test_bf16_quotients_are_never_near_a_rounding_midpoint).
//  * POW2L: for L a power of two, L * theta and its doubling are exact in bf16: no rounding.
template <bool BF16, bool POW2L = false>
struct GifModel {
    float decay, Lf, alpha, thr0;
    static constexpr int NS = 2;
    struct Lane { float s0, s1; };
    __device__ __forceinline__ static float r(float x) { return BF16 ? round_bf16(x) : x; }
    __device__ __forceinline__ static void r2(float& a, float& b) { if (BF16) round_bf16_pair(a, b); }
    __device__ __forceinline__ static float quot(float v, float t) {
        return BF16 ? v * __builtin_amdgcn_rcpf(t) : v / t;
    }
    __device__ __forceinline__ void init(Lane&, int64_t) const {}
    // two neurons at once: the same op sequence as step(), roundings paired (bf16 build: 3 VALU
    // ops per two roundings instead of 4).  x*2 of a bf16 value is exact, so r(r(L*theta)*2) is
    // r(L*theta)*2.
    __device__ __forceinline__ void step2(Lane& la, Lane& lb, float ia, float ib, float& sa,
                                          float& sb) const {
        float va = la.s0 * decay, vb = lb.s0 * decay;           r2(va, vb);
        va = va + ia; vb = vb + ib;                               r2(va, vb);
        float ca = Lf * la.s1, cb = Lf * lb.s1;                   if (!POW2L) r2(ca, cb);
        ca = ca * 2.0f; cb = cb * 2.0f;
        va = fminf(fmaxf(va, -ca), ca); vb = fminf(fmaxf(vb, -cb), cb);
        float ta = la.s1 + 1e-6f, tb = lb.s1 + 1e-6f;             r2(ta, tb);
        float na = quot(va, ta), nb = quot(vb, tb);               r2(na, nb);
        sa = fminf(fmaxf(floorf(na), 0.0f), Lf); sb = fminf(fmaxf(floorf(nb), 0.0f), Lf);
        float pa = sa * la.s1, pb = sb * lb.s1;                   r2(pa, pb);
        va = va - pa; vb = vb - pb;                               r2(va, vb);
        float tha = la.s1, thb = lb.s1;
        if (alpha > 0.0f) {
            float ea = alpha * sa, eb = alpha * sb;               r2(ea, eb);
            float ua = tha + ea, ub = thb + eb;                   r2(ua, ub);
            float da = tha - thr0, db = thb - thr0;               r2(da, db);
            da = alpha * da; db = alpha * db;                     r2(da, db);
            tha = ua - da; thb = ub - db;                         r2(tha, thb);
        }
        la.s0 = va; la.s1 = tha;
        lb.s0 = vb; lb.s1 = thb;
    }
    __device__ __forceinline__ float step(Lane& l, float i_t) const {
        float v = l.s0, theta = l.s1;
        v = r(r(v * decay) + i_t);
        float cl = r(r(Lf * theta) * 2.0f);
        v = fminf(fmaxf(v, -cl), cl);
        float nv = r(quot(v, r(theta + 1e-6f)));
        float spike = fminf(fmaxf(floorf(nv), 0.0f), Lf);
        v = r(v - r(spike * theta));
        if (alpha > 0.0f) theta = r(r(theta + r(alpha * spike)) - r(alpha * r(theta - thr0)));
        l.s0 = v;
        l.s1 = theta;
        return spike;
    }
};

// step two neurons: models with a paired form use it, the others take two scalar steps
template <class Model>
__device__ __forceinline__ auto step_pair(const Model& m, typename Model::Lane& a,
                                          typename Model::Lane& b, float xa, float xb, float& sa,
                                          float& sb, int) -> decltype(m.step2(a, b, xa, xb, sa, sb)) {
    m.step2(a, b, xa, xb, sa, sb);
}
template <class Model>
__device__ __forceinline__ void step_pair(const Model& m, typename Model::Lane& a,
                                          typename Model::Lane& b, float xa, float xb, float& sa,
                                          float& sb, long) {
    sa = m.step(a, xa);
    sb = m.step(b, xb);
}
template <class Model, int VEC>
__device__ __forceinline__ void step_vec(const Model& m, typename Model::Lane (&lane)[VEC],
                                         const float (&x)[VEC], float (&spk)[VEC]) {
    if constexpr (VEC % 2 == 0) {
#pragma unroll
        for (int e = 0; e < VEC; e += 2) step_pair(m, lane[e], lane[e + 1], x[e], x[e + 1], spk[e], spk[e + 1], 0);
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) spk[e] = m.step(lane[e], x[e]);
    }
}

// ------------------------------------------------------------------------------------------
// 16-byte vector I/O
// ------------------------------------------------------------------------------------------
template <typename T, int VEC>
struct Io;

template <>
struct Io<float, 4> {
    __device__ __forceinline__ static void load(const float* p, float (&x)[4]) {
        float4 t = *reinterpret_cast<const float4*>(p);
        x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
    }
    __device__ __forceinline__ static void store(float* p, const float (&x)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
    }
    // streaming (touched once) tensors: non-temporal, so they do not evict each other from L2
    typedef float v4f __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ static void load_nt(const float* p, float (&x)[4]) {
        const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
        x[0] = t[0]; x[1] = t[1]; x[2] = t[2]; x[3] = t[3];
    }
    __device__ __forceinline__ static void store_nt(float* p, const float (&x)[4]) {
        const v4f t = {x[0], x[1], x[2], x[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
    }
};
template <>
struct Io<float, 1> {
    __device__ __forceinline__ static void load(const float* p, float (&x)[1]) { x[0] = *p; }
    __device__ __forceinline__ static void store(float* p, const float (&x)[1]) { *p = x[0]; }
    __device__ __forceinline__ static void load_nt(const float* p, float (&x)[1]) { x[0] = __builtin_nontemporal_load(p); }
    __device__ __forceinline__ static void store_nt(float* p, const float (&x)[1]) { __builtin_nontemporal_store(x[0], p); }
};

__device__ __forceinline__ float bf16_bits_to_float(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint32_t float_to_bf16_bits(float x) {
    return __float_as_uint(round_bf16(x)) >> 16;
}

template <>
struct Io<uint16_t, 8> {
    __device__ __forceinline__ static void load(const uint16_t* p, float (&x)[8]) {
        uint4 t = *reinterpret_cast<const uint4*>(p);
        uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[2 * i] = __uint_as_float(w[i] << 16);
            x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ static void store(uint16_t* p, const float (&x)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = float_to_bf16_bits(x[2 * i]) | (float_to_bf16_bits(x[2 * i + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    // bf16 streams keep the default cache policy: non-temporal measured slower here (the bf16 GIF
    // loop is VALU-bound: 0.67 vs 0.54 ms at 8192 x 16 x 3072), while it took the fp32 GIF loop from
    // 5.5 to 6.4 TB/s
    __device__ __forceinline__ static void load_nt(const uint16_t* p, float (&x)[8]) { load(p, x); }
    __device__ __forceinline__ static void store_nt(uint16_t* p, const float (&x)[8]) { store(p, x); }
};
template <>
struct Io<uint16_t, 1> {
    __device__ __forceinline__ static void load_nt(const uint16_t* p, float (&x)[1]) { load(p, x); }
    __device__ __forceinline__ static void store_nt(uint16_t* p, const float (&x)[1]) { store(p, x); }
    __device__ __forceinline__ static void load(const uint16_t* p, float (&x)[1]) {
        x[0] = bf16_bits_to_float(*p);
    }
    __device__ __forceinline__ static void store(uint16_t* p, const float (&x)[1]) {
        *p = static_cast<uint16_t>(float_to_bf16_bits(x[0]));
    }
};

// ------------------------------------------------------------------------------------------
// Channel-contiguous kernel: x[R][T][C] (or [R][C] if TIME_INV) -> out[R][T][C] (or the T-mean
// [R][C] if MEAN_OUT); state st0/st1 [R][C].  One work item = VEC adjacent channels of one row.
// ------------------------------------------------------------------------------------------
constexpr int RTC_UNROLL = 4;

template <class Model, typename T, int VEC, bool TIME_INV, bool MEAN_OUT, bool BF16_MEAN>
__global__ __launch_bounds__(256) void seq_rtc_kernel(Model m, const T* __restrict__ x,
                                                      T* __restrict__ out, T* st0, T* st1,
                                                      int64_t R, int64_t Tn, int64_t C) {
    const int64_t cvecs = C / VEC;
    const int64_t items = R * cvecs;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / cvecs;
        const int64_t c0 = (it - row * cvecs) * VEC;
        typename Model::Lane lane[VEC];
        float s0[VEC], s1[VEC];
        Io<T, VEC>::load(st0 + row * C + c0, s0);
        if (Model::NS > 1) Io<T, VEC>::load(st1 + row * C + c0, s1);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            m.init(lane[e], c0 + e);
            lane[e].s0 = s0[e];
            if (Model::NS > 1) lane[e].s1 = s1[e];
        }
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.0f;

        const T* xp = x + (TIME_INV ? row * C + c0 : row * Tn * C + c0);
        T* op = out + (MEAN_OUT ? row * C + c0 : row * Tn * C + c0);

        if (TIME_INV) {
            float xin[VEC], spk[VEC];
            Io<T, VEC>::load(xp, xin);
            for (int64_t t = 0; t < Tn; ++t) {
                step_vec<Model, VEC>(m, lane, xin, spk);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += spk[e];
                if (!MEAN_OUT) Io<T, VEC>::store_nt(op + t * C, spk);
            }
        } else {
            int64_t t = 0;
            for (; t + RTC_UNROLL <= Tn; t += RTC_UNROLL) {
                float xin[RTC_UNROLL][VEC];
#pragma unroll
                for (int j = 0; j < RTC_UNROLL; ++j) Io<T, VEC>::load_nt(xp + (t + j) * C, xin[j]);
#pragma unroll
                for (int j = 0; j < RTC_UNROLL; ++j) {
                    float spk[VEC];
                    step_vec<Model, VEC>(m, lane, xin[j], spk);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += spk[e];
                    if (!MEAN_OUT) Io<T, VEC>::store_nt(op + (t + j) * C, spk);
                }
            }
            for (; t < Tn; ++t) {
                float xin[VEC], spk[VEC];
                Io<T, VEC>::load_nt(xp + t * C, xin);
                step_vec<Model, VEC>(m, lane, xin, spk);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += spk[e];
                if (!MEAN_OUT) Io<T, VEC>::store_nt(op + t * C, spk);
            }
        }
        if (MEAN_OUT) {
            // spikes.mean(dim=1): sum (exact: small integers) then a true division by T
            // (snn_ffn.py:81); in bf16 the sum is rounded to bf16 before the division.
            float mean[VEC];
            const float tf = static_cast<float>(Tn);
#pragma unroll
            for (int e = 0; e < VEC; ++e) mean[e] = (BF16_MEAN ? round_bf16(acc[e]) : acc[e]) / tf;
            Io<T, VEC>::store(op, mean);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            s0[e] = lane[e].s0;
            if (Model::NS > 1) s1[e] = lane[e].s1;
        }
        Io<T, VEC>::store(st0 + row * C + c0, s0);
        if (Model::NS > 1) Io<T, VEC>::store(st1 + row * C + c0, s1);
    }
}

// ------------------------------------------------------------------------------------------
// Time-contiguous kernel: I[N][T] -> S[N][T]; lane owns neuron (wave*64 + lane).  Each wave
// moves 64 neurons x 32 timesteps through a private LDS tile: HBM side is read/written with
// lanes along t (8 lanes x 16 B per 128-byte row segment), the compute side reads its own row
// as 8 x ds_read_b128.  Row stride 36 floats (144 B) keeps both the 16-byte row-segment writes
// and the per-lane ds_read_b128 of 64 different rows bank-conflict-free (slot = 9*row mod 16 is
// a permutation on each ds_read_b128 lane group) and 16-byte aligned.
// The next chunk's loads are issued before the current chunk is computed (register prefetch).
// ------------------------------------------------------------------------------------------
constexpr int NT_TC = 32;      // timesteps per chunk
constexpr int NT_STRIDE = 36;  // floats per LDS row
constexpr int NT_WAVES = 4;

template <class Model, int VEC>
__global__ __launch_bounds__(64 * NT_WAVES) void seq_nt_kernel(Model m, const float* __restrict__ I,
                                                               float* __restrict__ S, float* st0,
                                                               float* st1, int64_t N, int64_t Tn) {
    constexpr int LPR = NT_TC / VEC;  // lanes per row
    constexpr int RPI = 64 / LPR;     // rows per wave-instruction
    constexpr int NI = 64 / RPI;      // wave-instructions per tile
    __shared__ __attribute__((aligned(16))) float tiles[NT_WAVES][64 * NT_STRIDE];

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    float* tile = tiles[wave];
    const int64_t n0 = ((int64_t)blockIdx.x * NT_WAVES + wave) * 64;
    const int64_t n = n0 + lane;
    const int lrow = lane / LPR;
    const int lcol = (lane % LPR) * VEC;

    typename Model::Lane ln;
    m.init(ln, n);
    ln.s0 = 0.0f;
    ln.s1 = 0.0f;
    if (n < N) {
        ln.s0 = st0[n];
        if (Model::NS > 1) ln.s1 = st1[n];
    }

    float reg[NI][VEC];
    auto load_chunk = [&](int64_t c0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int64_t rn = n0 + i * RPI + lrow;
            const int64_t t = c0 + lcol;
            if (rn < N && t < Tn) {
                Io<float, VEC>::load(I + rn * Tn + t, reg[i]);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) reg[i][e] = 0.0f;
            }
        }
    };

    load_chunk(0);
    for (int64_t c0 = 0; c0 < Tn; c0 += NT_TC) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            Io<float, VEC>::store(tile + (i * RPI + lrow) * NT_STRIDE + lcol, reg[i]);
        if (c0 + NT_TC < Tn) load_chunk(c0 + NT_TC);
        __syncthreads();

        const int tc = (Tn - c0) < NT_TC ? (int)(Tn - c0) : NT_TC;
        float* myrow = tile + lane * NT_STRIDE;
#pragma unroll
        for (int j = 0; j < NT_TC / 4; ++j) {
            if (4 * j < tc) {
                float xin[4], spk[4];
                Io<float, 4>::load(myrow + 4 * j, xin);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    spk[e] = 0.0f;
                    if (4 * j + e < tc) spk[e] = m.step(ln, xin[e]);
                }
                Io<float, 4>::store(myrow + 4 * j, spk);
            }
        }
        __syncthreads();

#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int64_t rn = n0 + i * RPI + lrow;
            const int64_t t = c0 + lcol;
            if (rn < N && t < Tn) {
                float o[VEC];
                Io<float, VEC>::load(tile + (i * RPI + lrow) * NT_STRIDE + lcol, o);
                Io<float, VEC>::store(S + rn * Tn + t, o);
            }
        }
        __syncthreads();
    }
    if (n < N) {
        st0[n] = ln.s0;
        if (Model::NS > 1) st1[n] = ln.s1;
    }
}

// ------------------------------------------------------------------------------------------
// Time-contiguous kernel, wave-tile form (T % 4 == 0): ONE WAVE = 64 neurons x W timesteps with
// W = T for T <= 128 (else 128-step chunks).  For W = T the wave's input is one contiguous
// 64*T*4-byte region of HBM, so every load/store instruction moves 1 KiB of consecutive bytes and
// each cache line is fetched exactly once (the 32-step tiling above re-fetches the lines that a
// 400-byte row shares between chunks: FETCH_SIZE showed 2.1x the algorithmic bytes at T = 100).
// LDS row stride S = W rounded so that S/4 is odd: the per-lane ds_read_b128 of 64 different rows
// then hits 16 distinct 16-byte slots per lane group (conflict free).  Blocks are single waves,
// so no workgroup barrier is needed; 4-6 independent waves per CU overlap their load / compute /
// store phases.
// ------------------------------------------------------------------------------------------
constexpr int NTW_MAXW = 128;

template <class Model>
__global__ __launch_bounds__(64) void seq_ntw_kernel(Model m, const float* __restrict__ I,
                                                     float* __restrict__ Sp, float* st0, float* st1,
                                                     int64_t N, int64_t Tn, int Wmax, int S, int dma) {
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [64][S]
    const int lane = threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int64_t n = n0 + lane;
    const int nrows = (N - n0) < 64 ? (int)(N - n0) : 64;

    typename Model::Lane ln;
    m.init(ln, n);
    ln.s0 = 0.0f;
    ln.s1 = 0.0f;
    if (n < N) {
        ln.s0 = st0[n];
        if (Model::NS > 1) ln.s1 = st1[n];
    }
    float* myrow = tile + lane * S;

    for (int64_t t0 = 0; t0 < Tn; t0 += Wmax) {
        const int W = (Tn - t0) < Wmax ? (int)(Tn - t0) : Wmax;  // multiple of 4
        const int w4 = W >> 2;                                   // float4 per row piece = pieces
        // (row, col4) of this lane's element in piece p: e4 = p*64 + lane; advance incrementally
        const int dq = 64 / w4, dr = 64 % w4;
        int row = lane / w4, c4 = lane % w4;
        // ---- load.  Whole tiles whose LDS image is the HBM image (S == W == T: the wave's 64 rows are one
        //      contiguous 64 T 4-byte region) go HBM -> LDS directly (global_load_lds, 1 KiB per
        //      instruction, lane-linear): all T/4 pieces are in flight at once and no VGPR is staged
        //      (Izhikevich 2^22 x 100: 0.737 -> 0.621 ms, 4.6 -> 5.5 TB/s).
        if ((dma & 1) && nrows == 64) {
            const float* src = I + n0 * Tn + lane * 4;
            for (int p = 0; p < w4; ++p)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p * 256),
                                                 (__attribute__((address_space(3))) void*)(tile + p * 256), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else
        // ---- otherwise: pieces in batches of 8 (32 VGPRs in flight per lane; issuing all 25-32
        //      pieces at once costs 256 VGPRs and measured slower)
        for (int p0 = 0; p0 < w4; p0 += 8) {
            float4 v[8];
            int ra[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ra[j] = row * S + 4 * c4;
                v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p0 + j < w4 && row < nrows)
                    v[j] = *reinterpret_cast<const float4*>(I + (n0 + row) * Tn + t0 + 4 * c4);
                row += dq; c4 += dr;
                if (c4 >= w4) { c4 -= w4; ++row; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (p0 + j < w4) *reinterpret_cast<float4*>(tile + ra[j]) = v[j];
        }
        __syncthreads();   // single-wave block: orders the LDS writes before the row reads
        // ---- compute: this lane's neuron over W steps, spikes written back in place
        for (int j = 0; j < w4; ++j) {
            float xin[4], spk[4];
            Io<float, 4>::load(myrow + 4 * j, xin);
#pragma unroll
            for (int e = 0; e < 4; ++e) spk[e] = m.step(ln, xin[e]);
            Io<float, 4>::store(myrow + 4 * j, spk);
        }
        __syncthreads();
        // ---- store: mirror of the load
        row = lane / w4; c4 = lane % w4;
        for (int p = 0; p < w4; ++p) {
            if (row < nrows) {
                float v[4];
                Io<float, 4>::load(tile + row * S + 4 * c4, v);
                if (dma & 2) Io<float, 4>::store_nt(Sp + (n0 + row) * Tn + t0 + 4 * c4, v);   // written once, never re-read here
                else Io<float, 4>::store(Sp + (n0 + row) * Tn + t0 + 4 * c4, v);
            }
            row += dq; c4 += dr;
            if (c4 >= w4) { c4 -= w4; ++row; }
        }
        __syncthreads();
    }
    if (n < N) {
        st0[n] = ln.s0;
        if (Model::NS > 1) st1[n] = ln.s1;
    }
}

// ------------------------------------------------------------------------------------------
// Host-side launch helpers
// ------------------------------------------------------------------------------------------
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int check_launch() {
    return hipGetLastError() == hipSuccess ? AURA_OK : AURA_E_LAUNCH;
}

inline int rtc_grid(int64_t items) {
    // memory-bound streaming: cap at 256 CUs x 8 blocks and grid-stride the rest
    int64_t blocks = (items + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <class Model>
int launch_nt(const Model& m, const float* I, float* S, float* st0, float* st1, int64_t N,
              int64_t T, hipStream_t s) {
    if (N == 0 || T == 0) return AURA_OK;
    const int64_t blocks = (N + 64 * NT_WAVES - 1) / (64 * NT_WAVES);
    if (blocks > 0x7fffffffLL) return AURA_E_INVAL;
    const bool vec = (T % 4 == 0) && aligned16(I) && aligned16(S);
    static const bool legacy = getenv("AURA_NT_LEGACY") != nullptr;
    // rows that are whole 128-byte lines (T % 32 == 0) stream best through the 32-step tiling
    // (5.3 TB/s at T = 128 vs 3.6 for the wave-tile form); other T re-fetch split lines there
    // (2.9 TB/s at T = 100) and use the contiguous wave-tile form (4.4 TB/s)
    if (vec && !legacy && (T % 32 != 0)) {
        // wave-tile form: W = T if T <= 128, else 128-step chunks; stride with S/4 odd
        const int W = T <= NTW_MAXW ? (int)T : NTW_MAXW;
        const int Sld = ((W / 4) & 1) ? W : W + 4;
        const int64_t wblocks = (N + 63) / 64;
        if (wblocks > 0x7fffffffLL) return AURA_E_INVAL;
        static const bool no_dma = getenv("AURA_NT_NO_DMA") != nullptr;
        static const bool nt_st = getenv("AURA_NT_STORE_NT") != nullptr;   // A/B runs: non-temporal spike stores
        const int dma = ((!no_dma && Sld == W && T <= NTW_MAXW) ? 1 : 0) | (nt_st ? 2 : 0);
        // (Measured dead end, round 3: a persistent form -- single-wave blocks walking tiles with two LDS buffers,
        //  the next tile's LDS-DMA issued before the current tile is computed -- ran 0.705 ms against 0.627 ms
        //  for this one-block-per-tile form at 2^22 x 100: six independent waves per CU in different phases keep
        //  more loads in flight than three double-buffered ones.)
        hipLaunchKernelGGL((seq_ntw_kernel<Model>), dim3((unsigned)wblocks), dim3(64),
                           (size_t)64 * Sld * sizeof(float), s, m, I, S, st0, st1, N, T, W, Sld, dma);
    } else if (vec)
        hipLaunchKernelGGL((seq_nt_kernel<Model, 4>), dim3((unsigned)blocks), dim3(64 * NT_WAVES),
                           0, s, m, I, S, st0, st1, N, T);
    else
        hipLaunchKernelGGL((seq_nt_kernel<Model, 1>), dim3((unsigned)blocks), dim3(64 * NT_WAVES),
                           0, s, m, I, S, st0, st1, N, T);
    return check_launch();
}

template <class Model>
int launch_rtc_f32(const Model& m, const float* x, float* out, float* st0, float* st1, int64_t R,
                   int64_t T, int64_t C, hipStream_t s) {
    if (R == 0 || C == 0) return AURA_OK;
    const bool vec = (C % 4 == 0) && aligned16(x) && aligned16(out) && aligned16(st0) &&
                     (Model::NS == 1 || aligned16(st1));
    if (vec)
        hipLaunchKernelGGL((seq_rtc_kernel<Model, float, 4, false, false, false>),
                           dim3(rtc_grid(R * (C / 4))), dim3(256), 0, s, m, x, out, st0, st1, R, T,
                           C);
    else
        hipLaunchKernelGGL((seq_rtc_kernel<Model, float, 1, false, false, false>),
                           dim3(rtc_grid(R * C)), dim3(256), 0, s, m, x, out, st0, st1, R, T, C);
    return check_launch();
}

template <typename T, int VEC, bool BF16, bool POW2L>
int launch_gif(const GifModel<BF16, POW2L>& m, const void* h, void* out, void* v, void* th, int64_t rows,
               int64_t Tn, int64_t H, int flags, hipStream_t s) {
    const T* hp = static_cast<const T*>(h);
    T* op = static_cast<T*>(out);
    T* vp = static_cast<T*>(v);
    T* tp = static_cast<T*>(th);
    const dim3 g(rtc_grid(rows * (H / VEC))), b(256);
    const bool ti = flags & AURA_GIF_TIME_INVARIANT, mo = flags & AURA_GIF_MEAN_OUT;
#define AURA_GIF_LAUNCH(TI, MO)                                                                  \
    hipLaunchKernelGGL((seq_rtc_kernel<GifModel<BF16, POW2L>, T, VEC, TI, MO, BF16>), g, b, 0, s, m, hp, \
                       op, vp, tp, rows, Tn, H)
    if (ti && mo) AURA_GIF_LAUNCH(true, true);
    else if (ti) AURA_GIF_LAUNCH(true, false);
    else if (mo) AURA_GIF_LAUNCH(false, true);
    else AURA_GIF_LAUNCH(false, false);
#undef AURA_GIF_LAUNCH
    return check_launch();
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char* aura_version(void) { return "aura_hip 0.1.0 gfx950"; }

int aura_izh_run_nt(const float* I, float* spikes, float* v, float* u, float a, float b, float c,
                    float d, float dt, int64_t N, int64_t T, void* stream) {
    if (N < 0 || T < 0) return AURA_E_INVAL;
    if (N && T && (!I || !spikes)) return AURA_E_INVAL;
    if (N && (!v || !u)) return AURA_E_INVAL;
    IzhModel m{a, b, c, d, dt};
    return launch_nt(m, I, spikes, v, u, N, T, static_cast<hipStream_t>(stream));
}

int aura_izh_run_btd(const float* I, float* spikes, float* v, float* u, float a, float b, float c,
                     float d, float dt, int64_t B, int64_t T, int64_t D, void* stream) {
    if (B < 0 || T < 0 || D < 0) return AURA_E_INVAL;
    if (B && D && (!v || !u || (T && (!I || !spikes)))) return AURA_E_INVAL;
    IzhModel m{a, b, c, d, dt};
    return launch_rtc_f32(m, I, spikes, v, u, B, T, D, static_cast<hipStream_t>(stream));
}

static AdExModel adex_from(const float* p) {
    return AdExModel{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10]};
}

int aura_adex_run_nt(const float* I, float* spikes, float* V, float* w, const float* params_host,
                     int64_t N, int64_t T, void* stream) {
    if (N < 0 || T < 0 || !params_host) return AURA_E_INVAL;
    if (N && (!V || !w || (T && (!I || !spikes)))) return AURA_E_INVAL;
    return launch_nt(adex_from(params_host), I, spikes, V, w, N, T,
                     static_cast<hipStream_t>(stream));
}

int aura_adex_run_btd(const float* I, float* spikes, float* V, float* w, const float* params_host,
                      int64_t B, int64_t T, int64_t D, void* stream) {
    if (B < 0 || T < 0 || D < 0 || !params_host) return AURA_E_INVAL;
    if (B && D && (!V || !w || (T && (!I || !spikes)))) return AURA_E_INVAL;
    return launch_rtc_f32(adex_from(params_host), I, spikes, V, w, B, T, D,
                          static_cast<hipStream_t>(stream));
}

int aura_lif_run(const float* x, float* spikes, float* mem, const float* beta,
                 const float* threshold, int64_t B, int64_t T, int64_t size, void* stream) {
    if (B < 0 || T < 0 || size < 0) return AURA_E_INVAL;
    if (B && size && (!mem || !beta || !threshold || (T && (!x || !spikes)))) return AURA_E_INVAL;
    LifModel m{beta, threshold};
    return launch_rtc_f32(m, x, spikes, mem, static_cast<float*>(nullptr), B, T, size,
                          static_cast<hipStream_t>(stream));
}

int aura_gif_run(const void* h, void* out, void* v, void* theta, float decay, int L, float alpha,
                 float threshold, int64_t rows, int64_t T, int64_t H, int dtype, int flags,
                 void* stream) {
    if (rows < 0 || T < 0 || H < 0 || L < 0) return AURA_E_INVAL;
    if (dtype != AURA_DTYPE_F32 && dtype != AURA_DTYPE_BF16) return AURA_E_INVAL;
    if (flags & ~(AURA_GIF_TIME_INVARIANT | AURA_GIF_MEAN_OUT)) return AURA_E_INVAL;
    if (rows == 0 || H == 0) return AURA_OK;
    if (!v || !theta || (T && (!h || !out))) return AURA_E_INVAL;
    if ((flags & AURA_GIF_MEAN_OUT) && T == 0) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool al = aligned16(h) && aligned16(out) && aligned16(v) && aligned16(theta);
    if (dtype == AURA_DTYPE_F32) {
        GifModel<false> m{decay, (float)L, alpha, threshold};
        if (al && H % 4 == 0) return launch_gif<float, 4, false, false>(m, h, out, v, theta, rows, T, H, flags, s);
        return launch_gif<float, 1, false, false>(m, h, out, v, theta, rows, T, H, flags, s);
    }
    // bf16 scalars: the reference multiplies bf16 tensors by Python floats in fp32 opmath
    // (the scalar is NOT rounded to bf16 first), so decay/alpha/threshold stay fp32 here.
    if (L > 0 && (L & (L - 1)) == 0 && al && H % 8 == 0) {   // L = 2^j: L * theta is exact in bf16
        GifModel<true, true> m{decay, (float)L, alpha, threshold};
        return launch_gif<uint16_t, 8, true, true>(m, h, out, v, theta, rows, T, H, flags, s);
    }
    GifModel<true> m{decay, (float)L, alpha, threshold};
    if (al && H % 8 == 0) return launch_gif<uint16_t, 8, true, false>(m, h, out, v, theta, rows, T, H, flags, s);
    return launch_gif<uint16_t, 1, true, false>(m, h, out, v, theta, rows, T, H, flags, s);
}

}  // extern "C"
