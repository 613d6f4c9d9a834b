// aura_knn.hip -- episodic-bank kernels for gfx950 (MI355X): one-shot write, row norms, decay and
// the exact batched cosine-kNN recall (scan + top-k).
//
// Data layout in HBM (all fp32, row-major; names follow HippocampalFormation's buffers,
// src/core/hippocampal.py:90-117):
//   bank     [M][D]   memory_features          meta [M][4] = {strength, timestamp, centroid_id, 0}
//   loc      [M][S]   memory_locations         inv_norm [M] = 1/max(||bank_i||, 1e-12)  (build-side)
//
// Recall = scores(Q x bank) -> per-query top-k.  For a batch of queries the score matrix is a
// true dense contraction (2*nq*N*D FLOP over N*D*4 bytes: 127 FLOP/B at nq = 256), so the scan
// runs on the fp32 matrix cores: v_mfma_f32_32x32x2_f32 is exact fp32 (a k-ordered fmaf chain,
// no xf32/TF32 on gfx950) at 64 FLOP/clk/SIMD.  Each 512-thread workgroup owns a tile of bank
// rows x up to 256 queries, streams both operands through LDS in 32-deep k-tiles (the bank is
// read from HBM exactly once per 256-query block) and applies the reference's combined-score
// epilogue on the accumulators.  For nq <= 32 the same kernel degenerates to an HBM-bound
// batched GEMV (one 32-query MFMA column, 8 waves along the bank rows).
//
// Top-k never materialises nq x N scores on the fast path: a strided sample of bank tiles is
// scored densely first, its exact per-query k-th best is a valid lower bound of the global k-th
// best, and the main scan appends only (score,row) pairs that reach that bound to small
// per-query candidate lists; an LDS radix-select + bitonic sort over those lists gives the exact
// top-k (ties -> lower row).  The result is independent of append order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <set>
#include <utility>
#include <vector>
#include <type_traits>

#include "../../include/aura_hip.h"

#pragma clang fp contract(off)

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v_t __attribute__((ext_vector_type(4)));

inline int check_launch() { return hipGetLastError() == hipSuccess ? AURA_OK : AURA_E_LAUNCH; }
inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel); callable from any
// host thread and for any device of the process
inline int ensure_lds_attr(const void* fn, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return AURA_E_LAUNCH;
    std::lock_guard<std::mutex> g(mu);
    if (done.count({dev, fn})) return AURA_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return AURA_E_LAUNCH;
    done.insert({dev, fn});
    return AURA_OK;
}

// order-preserving float -> uint32 (larger float <=> larger key); -0 < +0, NaN ends up extreme
__device__ __forceinline__ uint32_t ord_key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_unkey(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// ------------------------------------------------------------------------------------------
// Row norms / query norms:  inv[i] = 1 / max(||x_i||, 1e-12)   (F.normalize, hippocampal.py:273,278)
// One wave per row, 16-byte loads.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_inv_norm_kernel(const float* __restrict__ x,
                                                           float* __restrict__ inv, int64_t n,
                                                           int64_t D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* p = x + row * D;
    float s = 0.0f;
    if ((D & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
        for (int64_t i = lane * 4; i < D; i += 256) {
            float4 v = *reinterpret_cast<const float4*>(p + i);
            s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
        }
    } else {
        for (int64_t i = lane; i < D; i += 64) s = fmaf(p[i], p[i], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) inv[row] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
}

// ------------------------------------------------------------------------------------------
// One-shot writes.  Both kernels compute 1/||row|| with row_inv_norm_kernel's arithmetic (same
// per-lane fmaf chain, same butterfly), so a bank's cached norms do not depend on whether they were
// produced by a write or recomputed after a state_dict load: recall is bit-identical either way.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_row_sumsq(const float* __restrict__ p, int64_t D, int lane, bool vec4) {
    float s = 0.0f;
    if (vec4) {
        for (int64_t i = lane * 4; i < D; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(p + i);
            s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
        }
    } else {
        for (int64_t i = lane; i < D; i += 64) s = fmaf(p[i], p[i], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    return s;
}

// Without centroid maintenance: independent rows, one wave per row.
__global__ __launch_bounds__(256) void bank_write_kernel(float* bank, float* loc, float* meta,
                                                         float* inv_norm, const float* feats,
                                                         const int64_t* slots, const float* cur_loc,
                                                         int sdims, float now, float cid,
                                                         int64_t n, int64_t D, int vec4) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t slot = slots[i];
    const float* src = feats + i * D;
    float* dst = bank + slot * D;
    if (vec4) {
        for (int64_t c = lane * 4; c < D; c += 256)
            *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
    } else {
        for (int64_t c = lane; c < D; c += 64) dst[c] = src[c];
    }
    const float s = wave_row_sumsq(src, D, lane, vec4 != 0);
    if (lane == 0) {
        inv_norm[slot] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
        float4 m = make_float4(1.0f, now, cid, 0.0f);
        *reinterpret_cast<float4*>(meta + slot * 4) = m;
    }
    if (lane < sdims) loc[slot * sdims + lane] = cur_loc[lane];
}

// ------------------------------------------------------------------------------------------
// One-shot write WITH the online centroid update of hippocampal.py:218-230.  The update is order
// dependent (row i sees the centroids left by rows < i), so ONE 1024-thread workgroup walks the rows in
// order; per row: the row goes to LDS; wave w scores centroids w, w + 16, ... (a centroid = one coalesced
// 4 D-byte read from L2 across the lanes, four centroids in flight per wave, butterfly reduction); wave 0
// takes the first minimum (torch.argmin) and the running mean of that centroid is updated by all
// threads.  ~1-2 us per row (the first version spread every centroid over 4 threads reading it with a
// 16-byte stride: 38 us per row, 10 ms for a 256-row layer batch).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bank_write_centroid_kernel(
    float* bank, float* loc, float* meta, float* inv_norm, float* centroids, float* counts,
    int eff_k, const float* feats, const int64_t* slots, const float* cur_loc, int sdims, float now,
    int64_t n, int64_t D, int vec4) {
    extern __shared__ __attribute__((aligned(16))) float s_row[];   // [D]
    __shared__ float s_dist[256];
    __shared__ int s_best;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t slot = slots[i];
        const float* src = feats + i * D;
        for (int64_t j = tid; j < D; j += 1024) {
            const float v = src[j];
            s_row[j] = v;
            bank[slot * D + j] = v;
        }
        if (wave == 15) {                                    // norm + metadata shell (centroid id follows)
            const float s = wave_row_sumsq(src, D, lane, vec4 != 0);
            if (lane == 0) inv_norm[slot] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
            if (lane < sdims) loc[slot * sdims + lane] = cur_loc[lane];
        }
        __syncthreads();
        // distances ||c - x||_2: wave w -> centroids w + 16 g
        for (int c0 = wave; c0 < eff_k; c0 += 64) {
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (vec4) {
                for (int64_t j = lane * 4; j < D; j += 256) {
                    const float4 x = *reinterpret_cast<const float4*>(s_row + j);
                    float4 cv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = c0 + 16 * u;
                        cv[u] = c < eff_k ? *reinterpret_cast<const float4*>(centroids + (int64_t)c * D + j)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float d = cv[u].x - x.x; acc[u] = fmaf(d, d, acc[u]);
                        d = cv[u].y - x.y; acc[u] = fmaf(d, d, acc[u]);
                        d = cv[u].z - x.z; acc[u] = fmaf(d, d, acc[u]);
                        d = cv[u].w - x.w; acc[u] = fmaf(d, d, acc[u]);
                    }
                }
            } else {
                for (int64_t j = lane; j < D; j += 64) {
                    const float x = s_row[j];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = c0 + 16 * u;
                        const float d = (c < eff_k ? centroids[(int64_t)c * D + j] : 0.0f) - x;
                        acc[u] = fmaf(d, d, acc[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc[u] += __shfl_xor(acc[u], off);
                const int c = c0 + 16 * u;
                if (lane == 0 && c < eff_k) s_dist[c] = sqrtf(acc[u]);
            }
        }
        __syncthreads();
        if (wave == 0) {                                     // first minimum over c < eff_k, as torch.argmin
            float bd = INFINITY;
            int best = 0x7fffffff;
            for (int c = lane; c < eff_k && c < 256; c += 64) {
                const float d = s_dist[c];
                if (d < bd) { bd = d; best = c; }            // ascending c per lane: keeps the lane's first minimum
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float od = __shfl_xor(bd, off);
                const int ob = __shfl_xor(best, off);
                if (od < bd || (od == bd && ob < best)) { bd = od; best = ob; }
            }
            if (best == 0x7fffffff) best = 0;                // every distance NaN: torch.argmin's answer is moot
            if (lane == 0) {
                s_best = best;
                counts[best] = counts[best] + 1.0f;
                *reinterpret_cast<float4*>(meta + slot * 4) = make_float4(1.0f, now, (float)best, 0.0f);
            }
        }
        __syncthreads();
        const int best = s_best;
        const float eta = 1.0f / fmaxf(counts[best], 1.0f);
        const float one_m = 1.0f - eta;
        float* cp = centroids + (int64_t)best * D;
        for (int64_t j = tid; j < D; j += 1024) cp[j] = one_m * cp[j] + eta * s_row[j];
        __threadfence_block();
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// The same online update with the distances taken out of the serial chain (round 3).  The serial kernel
// above stays as the checker (and serves batches that name a slot twice); results are bit-identical:
//   phase 0  bank_write_kernel: rows, norms, locations, metadata shells -- parallel, as without the index;
//   phase A  online_dist0_kernel: d0[i][c] = distance of row i to centroid c AS THE TABLE STANDS AT BATCH
//            START, with exactly the serial kernel's arithmetic (per-lane fmaf chain, butterfly, sqrtf);
//   phase B  online_assign_kernel: ONE workgroup walks the rows in order.  Centroid c has moved, since
//            batch start, by at most delta[c] = sum of ||step|| over the rows assigned to it so far (a step
//            is eta (x - c): its length is eta times a distance the pass already holds), so the distance
//            the serial kernel would compute now lies in d0 +- (delta[c] + fp slack); for a centroid that
//            has not moved it IS d0, bit for bit.  Row i's candidates are the centroids whose lower bound
//            does not exceed the smallest upper bound: usually one -- the answer, no distance computed --
//            otherwise the candidates (typically 2-4 of 256) are re-scored against the current table with
//            the serial arithmetic and the first minimum wins, as torch.argmin.  Then counts, eta, the
//            running mean (same expression as above), delta.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float centroid_dist_wave(const float* __restrict__ x, const float* __restrict__ c,
                                                    int64_t D, int lane, bool vec4) {
    float acc = 0.0f;
    if (vec4) {
        for (int64_t j = lane * 4; j < D; j += 256) {
            const float4 xv = *reinterpret_cast<const float4*>(x + j);
            const float4 cv = *reinterpret_cast<const float4*>(c + j);
            float d = cv.x - xv.x; acc = fmaf(d, d, acc);
            d = cv.y - xv.y; acc = fmaf(d, d, acc);
            d = cv.z - xv.z; acc = fmaf(d, d, acc);
            d = cv.w - xv.w; acc = fmaf(d, d, acc);
        }
    } else {
        for (int64_t j = lane; j < D; j += 64) {
            const float d = c[j] - x[j];
            acc = fmaf(d, d, acc);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    return sqrtf(acc);
}

constexpr int ONL_CG = 8;                   // centroids per wave task of phase A
constexpr int ONL_CHUNK = 512;              // rows per phase A / phase B pair (workspace: 259 floats per row).  The step
                                            // bounds only grow inside a chunk, so long chunks re-score more and more rows:
                                            // 4096-row chunks ran at half the per-row rate of 512-row ones

__global__ __launch_bounds__(256) void online_dist0_kernel(const float* __restrict__ feats,
                                                           const float* __restrict__ centroids, int eff_k,
                                                           int64_t n, int64_t D, int vec4,
                                                           float* __restrict__ d0,       // [n][256]
                                                           float* __restrict__ xnorm) {  // [n] upper bound of ||x_i||
    const int lane = threadIdx.x & 63;
    const int groups = (eff_k + ONL_CG - 1) / ONL_CG;
    const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= n * groups) return;
    const int64_t i = task / groups;
    const int g = (int)(task - i * groups);
    const float* const x = feats + i * D;
#pragma unroll 2
    for (int u = 0; u < ONL_CG; ++u) {
        const int c = g * ONL_CG + u;
        if (c >= eff_k) break;
        const float d = centroid_dist_wave(x, centroids + (int64_t)c * D, D, lane, vec4 != 0);
        if (lane == 0) d0[i * 256 + c] = d;
    }
    if (g == 0) {
        const float s = wave_row_sumsq(x, D, lane, vec4 != 0);
        if (lane == 0) xnorm[i] = sqrtf(s) * 1.0001f;
    }
}

__global__ __launch_bounds__(1024) void online_assign_kernel(float* meta, float* centroids, float* counts, int eff_k,
                                                             const float* __restrict__ feats,
                                                             const int64_t* __restrict__ slots,
                                                             const float* __restrict__ d0,
                                                             const float* __restrict__ xnorm, int64_t n, int64_t D,
                                                             int vec4, float rel) {
    extern __shared__ __attribute__((aligned(16))) float s_row[];   // [D] (re-scoring only)
    __shared__ float s_delta[256], s_counts[256], s_fresh[256];
    __shared__ int s_cand[256];
    __shared__ int s_ncand, s_best;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 256) {
        s_delta[tid] = 0.0f;
        s_counts[tid] = tid < eff_k ? counts[tid] : 0.0f;
    }
    __syncthreads();
    // what wave 0 does once row i's centroid is known: count, step bound, metadata
    auto commit = [&](int64_t i, int best, float dwin_up) {   // (one lane)
        const float cn = s_counts[best] + 1.0f;
        s_counts[best] = cn;
        const float eta = 1.0f / fmaxf(cn, 1.0f);
        // ||step|| <= eta ||x - c|| + rounding of the mean's three roundings per coordinate:
        // <= 4 u (||c|| + eta ||x||), ||c|| <= ||x|| + ||x - c||
        s_delta[best] += 1.0001f * (eta * dwin_up) + 3e-7f * (2.0f * xnorm[i] + dwin_up);
        meta[slots[i] * 4 + 2] = (float)best;
        s_best = best;
    };
    for (int64_t i = 0; i < n; ++i) {
        if (wave == 0) {
            float lo[4], hi[4], dv[4], dl[4];
            float hmin = INFINITY;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = lane + 64 * u;
                const bool valid = c < eff_k;
                dv[u] = valid ? d0[i * 256 + c] : INFINITY;
                dl[u] = s_delta[c];
                const float m = dl[u] == 0.0f ? 0.0f : dv[u] * rel + dl[u] * 1.0001f;   // unmoved: d0 is the value itself
                lo[u] = valid ? dv[u] - m : INFINITY;
                hi[u] = valid ? dv[u] + m : INFINITY;
                hmin = fminf(hmin, hi[u]);                   // (NaN distances never become candidates, as in the
            }                                                //  serial kernel's `d < best` scan)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) hmin = fminf(hmin, __shfl_xor(hmin, off));
            unsigned long long cm[4];
            int total = 0;
            bool moved = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool cnd = lo[u] <= hmin;
                cm[u] = __ballot(cnd);
                total += (int)__popcll(cm[u]);
                moved = moved || (cnd && dl[u] != 0.0f);
            }
            const bool any_moved = __ballot(moved) != 0ull;
            if (total == 0) {                               // every distance NaN: the serial kernel answers 0
                if (lane == 0) { s_ncand = 1; commit(i, 0, 0.0f); }
            } else if (total == 1) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (lo[u] <= hmin) { s_ncand = 1; commit(i, lane + 64 * u, hi[u]); }
            } else if (!any_moved) {
                // exact ties among centroids that have not moved: their distances are d0 itself -> first minimum
                float bd = INFINITY;
                int best = 0x7fffffff;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (lo[u] <= hmin && dv[u] < bd) { bd = dv[u]; best = lane + 64 * u; }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float od = __shfl_xor(bd, off);
                    const int ob = __shfl_xor(best, off);
                    if (od < bd || (od == bd && ob < best)) { bd = od; best = ob; }
                }
                if (lane == 0) { s_ncand = 1; commit(i, best, bd); }
            } else {
                // candidate list in ascending centroid order (c = 64 u + lane)
                int base = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (lo[u] <= hmin) {
                        const int p = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cm[u] >> 32),
                                                  __builtin_amdgcn_mbcnt_lo((unsigned)cm[u], 0u));
                        s_cand[p] = lane + 64 * u;
                    }
                    base += (int)__popcll(cm[u]);
                }
                if (lane == 0) s_ncand = total;
            }
        }
        __syncthreads();
        const int ncand = s_ncand;
        if (ncand > 1) {                                    // (workgroup-uniform)
            for (int64_t j = tid; j < D; j += 1024) s_row[j] = feats[i * D + j];
            __threadfence_block();                          // earlier rows' centroid stores are visible to every wave
            __syncthreads();
            for (int ci = wave; ci < ncand; ci += 16) {
                const float d = centroid_dist_wave(s_row, centroids + (int64_t)s_cand[ci] * D, D, lane, vec4 != 0);
                if (lane == 0) s_fresh[ci] = d;
            }
            __syncthreads();
            if (wave == 0) {                                // first minimum; the list is in ascending centroid order
                float bd = INFINITY;
                int bi = 0x7fffffff;
                for (int ci = lane; ci < ncand; ci += 64) {
                    const float d = s_fresh[ci];
                    if (d < bd) { bd = d; bi = ci; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float od = __shfl_xor(bd, off);
                    const int ob = __shfl_xor(bi, off);
                    if (od < bd || (od == bd && ob < bi)) { bd = od; bi = ob; }
                }
                // (all re-scored distances NaN cannot be: a candidate had a finite bound)
                if (lane == 0) commit(i, s_cand[bi == 0x7fffffff ? 0 : bi], bd * (1.0f + rel));
            }
            __syncthreads();
        }
        const int best = s_best;
        const float eta = 1.0f / fmaxf(s_counts[best], 1.0f);
        const float one_m = 1.0f - eta;
        float* cp = centroids + (int64_t)best * D;
        const float* xr = feats + i * D;
        for (int64_t j = tid; j < D; j += 1024) cp[j] = one_m * cp[j] + eta * xr[j];
        __threadfence_block();                              // the store is done before this thread touches the row again
        __syncthreads();                                    // s_best / s_ncand / s_counts are rewritten next
    }
    if (tid < eff_k) counts[tid] = s_counts[tid];
}

// ---- phase B, pipelined form (D <= 1024: one centroid element per thread) ----
// The simple form above spends ~2 us per row on three global round trips inside the serial chain: the row's d0
// line, the winning centroid's row, the acknowledgement of its store.  Here nothing in the chain leaves the CU
// when a row's decision is clear (the usual case):
//   * d0[i], x_i and the centroid row of row i's PREDICTED winner (argmin of d0[i], computed by
//     online_pred_kernel) are loaded three rows ahead into registers; the prediction is right whenever the bound
//     decides the row, and a predicted row that an update of the last two iterations has overtaken is replaced
//     by that update's own result, kept in registers (a thread only ever writes element `tid` of a centroid row,
//     so its own loads see its own stores: single-thread coherence; cross-thread reads happen only in the
//     re-scoring branch, behind a fence and a barrier);
//   * wave 0 decides from registers and LDS and publishes (winner, eta) in a two-slot mailbox: ONE barrier per row;
//   * centroid ids go to a dense array (scattered into the metadata by a parallel launch afterwards).
struct OnlSlot { float x, cpre; int cid; float d[4]; };

__global__ __launch_bounds__(256) void online_pred_kernel(const float* __restrict__ d0, int eff_k, int64_t n,
                                                          int32_t* __restrict__ pred) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    float bd = INFINITY;
    int best = 0x7fffffff;
    for (int c = lane; c < eff_k; c += 64) {
        const float d = d0[i * 256 + c];
        if (d < bd) { bd = d; best = c; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_xor(bd, off);
        const int ob = __shfl_xor(best, off);
        if (od < bd || (od == bd && ob < best)) { bd = od; best = ob; }
    }
    if (lane == 0) pred[i] = best == 0x7fffffff ? 0 : best;
}

__global__ __launch_bounds__(256) void online_cid_scatter_kernel(float* meta, const int64_t* __restrict__ slots,
                                                                 const int32_t* __restrict__ cid, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) meta[slots[i] * 4 + 2] = (float)cid[i];
}

__global__ __launch_bounds__(1024) void online_assign_fast_kernel(float* centroids, float* counts, int eff_k,
                                                                  const float* __restrict__ feats,
                                                                  const float* __restrict__ d0,
                                                                  const float* __restrict__ xnorm,
                                                                  const int32_t* __restrict__ pred,
                                                                  int32_t* __restrict__ cid_out, int64_t n, int64_t D,
                                                                  int vec4, float rel) {
    extern __shared__ __attribute__((aligned(16))) float s_row[];   // [D] (re-scoring only)
    __shared__ float s_delta[256], s_counts[256], s_fresh[256], s_xn[ONL_CHUNK];
    __shared__ int s_cand[256], s_pred[ONL_CHUNK];
    __shared__ int s_pub_best[2], s_pub_ncand[2];
    __shared__ float s_pub_eta[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool owner = tid < D;                              // this thread owns element `tid` of every centroid row
    if (tid < 256) {
        s_delta[tid] = 0.0f;
        s_counts[tid] = tid < eff_k ? counts[tid] : 0.0f;
    }
    for (int64_t i = tid; i < n; i += 1024) { s_pred[i] = pred[i]; s_xn[i] = xnorm[i]; }
    __syncthreads();
    auto prefetch = [&](OnlSlot& sl, int64_t row) {
        if (row >= n) return;
        const int p = s_pred[row];
        sl.cid = p;
        if (owner) {
            sl.x = feats[row * D + tid];
            sl.cpre = centroids[(int64_t)p * D + tid];
        }
        if (wave == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = lane + 64 * u;
                sl.d[u] = c < eff_k ? d0[row * 256 + c] : INFINITY;
            }
        }
    };
    auto commit = [&](int64_t i, int best, float dwin_up) {   // (one lane of wave 0)
        const float cn = s_counts[best] + 1.0f;
        s_counts[best] = cn;
        const float eta = 1.0f / fmaxf(cn, 1.0f);
        s_delta[best] += 1.0001f * (eta * dwin_up) + 3e-7f * (2.0f * s_xn[i] + dwin_up);
        cid_out[i] = best;
        s_pub_best[i & 1] = best;
        s_pub_eta[i & 1] = eta;
    };
    int last1_id = -1, last2_id = -1;                         // centroids written in the previous two iterations ...
    float last1_v = 0.0f, last2_v = 0.0f;                     // ... and this thread's element of what was written
    auto step = [&](int64_t i, OnlSlot& sl) {
        if (wave == 0) {
            float lo[4], hi[4], dl[4];
            float hmin = INFINITY;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = lane + 64 * u;
                const bool valid = c < eff_k;
                dl[u] = s_delta[c];
                const float m = dl[u] == 0.0f ? 0.0f : sl.d[u] * rel + dl[u] * 1.0001f;
                lo[u] = valid ? sl.d[u] - m : INFINITY;
                hi[u] = valid ? sl.d[u] + m : INFINITY;
                hmin = fminf(hmin, hi[u]);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) hmin = fminf(hmin, __shfl_xor(hmin, off));
            unsigned long long cm[4];
            int total = 0;
            bool moved = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool cnd = lo[u] <= hmin;
                cm[u] = __ballot(cnd);
                total += (int)__popcll(cm[u]);
                moved = moved || (cnd && dl[u] != 0.0f);
            }
            const bool any_moved = __ballot(moved) != 0ull;
            if (total == 0) {
                if (lane == 0) { s_pub_ncand[i & 1] = 1; commit(i, 0, 0.0f); }
            } else if (total == 1) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (lo[u] <= hmin) { s_pub_ncand[i & 1] = 1; commit(i, lane + 64 * u, hi[u]); }
            } else if (!any_moved) {
                float bd = INFINITY;
                int best = 0x7fffffff;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (lo[u] <= hmin && sl.d[u] < bd) { bd = sl.d[u]; best = lane + 64 * u; }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float od = __shfl_xor(bd, off);
                    const int ob = __shfl_xor(best, off);
                    if (od < bd || (od == bd && ob < best)) { bd = od; best = ob; }
                }
                if (lane == 0) { s_pub_ncand[i & 1] = 1; commit(i, best, bd); }
            } else {
                int base = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (lo[u] <= hmin) {
                        const int p = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cm[u] >> 32),
                                                  __builtin_amdgcn_mbcnt_lo((unsigned)cm[u], 0u));
                        s_cand[p] = lane + 64 * u;
                    }
                    base += (int)__popcll(cm[u]);
                }
                if (lane == 0) s_pub_ncand[i & 1] = total;
            }
        }
        __syncthreads();
        const int ncand = s_pub_ncand[i & 1];
        if (ncand > 1) {                                    // (workgroup-uniform) re-score the candidates
            if (owner) s_row[tid] = sl.x;
            __threadfence_block();                          // every thread's centroid stores are done ...
            __syncthreads();                                // ... before any wave reads whole rows
            for (int ci = wave; ci < ncand; ci += 16) {
                const float d = centroid_dist_wave(s_row, centroids + (int64_t)s_cand[ci] * D, D, lane, vec4 != 0);
                if (lane == 0) s_fresh[ci] = d;
            }
            __syncthreads();
            if (wave == 0) {
                float bd = INFINITY;
                int bi = 0x7fffffff;
                for (int ci = lane; ci < ncand; ci += 64) {
                    const float d = s_fresh[ci];
                    if (d < bd) { bd = d; bi = ci; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float od = __shfl_xor(bd, off);
                    const int ob = __shfl_xor(bi, off);
                    if (od < bd || (od == bd && ob < bi)) { bd = od; bi = ob; }
                }
                if (lane == 0) commit(i, s_cand[bi == 0x7fffffff ? 0 : bi], bd * (1.0f + rel));
            }
            __syncthreads();
        }
        const int best = s_pub_best[i & 1];
        const float eta = s_pub_eta[i & 1];
        const float one_m = 1.0f - eta;
        if (owner) {
            float c_old;
            if (best == last1_id) c_old = last1_v;           // the newest value of that row is still in a register
            else if (best == last2_id) c_old = last2_v;
            else if (best == sl.cid) c_old = sl.cpre;        // prefetched behind every older store of this thread
            else c_old = centroids[(int64_t)best * D + tid]; // prediction missed: one L2 round trip
            const float c_new = one_m * c_old + eta * sl.x;
            centroids[(int64_t)best * D + tid] = c_new;
            last2_v = last1_v; last1_v = c_new;
        }
        last2_id = last1_id; last1_id = best;
        prefetch(sl, i + 3);                                 // (issued behind this iteration's store)
    };
    OnlSlot a, b, c;
    a.cid = b.cid = c.cid = -1;
    prefetch(a, 0); prefetch(b, 1); prefetch(c, 2);
    for (int64_t i = 0; i < n; i += 3) {
        step(i, a);
        if (i + 1 < n) step(i + 1, b);
        if (i + 2 < n) step(i + 2, c);
    }
    __syncthreads();
    if (tid < eff_k) counts[tid] = s_counts[tid];
}

// ---- phase B, decoupled form (round 3, second pass) ----
// As the pipelined form above, without its barrier on the clear rows: wave 0 decides row i, publishes (winner, eta)
// in a mailbox of ONL_WIN entries and a sequence number, applies its own 64 elements of the update and goes on to
// row i + 1; the other fifteen waves wait for the sequence number (an LDS poll), apply theirs and prefetch.  A row
// then costs max(decision, update) instead of decision + barrier + update.  Rows that need re-scoring meet at the
// same barriers as before (every wave reads the same mailbox entry, so all of them take that branch), and one
// barrier per ONL_WIN rows keeps the mailbox from being lapped.
constexpr int ONL_WIN = 8;
#ifndef AURA_ONL_POLL_SLEEP
#define AURA_ONL_POLL_SLEEP 2
#endif
constexpr int ONL_POLL_SLEEP = AURA_ONL_POLL_SLEEP;   // s_sleep units (64 clocks) between two polls of the sequence number
// minimum over the 64 lanes, in every lane: four DPP row rotations (each row of 16 lanes then holds its own
// minimum), one lane read per row, three scalar-operand mins
__device__ __forceinline__ float onl_wave_min(float v) {
#define AURA_ROR_MIN(ctrl)                                                                                    \
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false)))
    AURA_ROR_MIN(0x128); AURA_ROR_MIN(0x124); AURA_ROR_MIN(0x122); AURA_ROR_MIN(0x121);
#undef AURA_ROR_MIN
    const int vi = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}

__global__ __launch_bounds__(1024) void online_assign_fast2_kernel(float* centroids, float* counts, int eff_k,
                                                                  const float* __restrict__ feats,
                                                                  const float* __restrict__ d0,
                                                                  const float* __restrict__ xnorm,
                                                                  const int32_t* __restrict__ pred,
                                                                  int32_t* __restrict__ cid_out, int64_t n, int64_t D,
                                                                  int vec4, float rel) {
    extern __shared__ __attribute__((aligned(16))) float s_row[];   // [D] (re-scoring only)
    __shared__ float s_delta[256], s_counts[256], s_fresh[256], s_xn[ONL_CHUNK];
    __shared__ int s_cand[256], s_pred[ONL_CHUNK];
    __shared__ int s_pub_best[ONL_WIN], s_pub_ncand[ONL_WIN];
    __shared__ float s_pub_eta[ONL_WIN];
    __shared__ int s_seq;                                    // rows published so far (row i's mailbox entry is valid once s_seq > i)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Element e of every centroid row belongs to thread own0 + e.  With D <= 960 wave 0 owns nothing (own0 = 64): its
    // loop is the decision chain alone, the fifteen other waves carry the update.
    const int own0 = D <= 960 ? 64 : 0;
    const int el = tid - own0;                               // this thread's element (if owner)
    const bool owner = el >= 0 && el < D;
    if (tid < 256) {
        s_delta[tid] = 0.0f;
        s_counts[tid] = tid < eff_k ? counts[tid] : 0.0f;
    }
    if (tid == 0) s_seq = 0;
    for (int64_t i = tid; i < n; i += 1024) { s_pred[i] = pred[i]; s_xn[i] = xnorm[i]; }
    __syncthreads();
    auto prefetch = [&](OnlSlot& sl, int64_t row) {
        if (row >= n) return;
        const int p = s_pred[row];
        sl.cid = p;
        if (owner) {
            sl.x = feats[row * D + el];
            sl.cpre = centroids[(int64_t)p * D + el];
        }
        if (wave == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = lane + 64 * u;
                sl.d[u] = c < eff_k ? d0[row * 256 + c] : INFINITY;
            }
        }
    };
    auto commit = [&](int64_t i, int best, float dwin_up) {   // (one lane of wave 0)
        const float cn = s_counts[best] + 1.0f;
        s_counts[best] = cn;
        const float eta = 1.0f / fmaxf(cn, 1.0f);
        s_delta[best] += 1.0001f * (eta * dwin_up) + 3e-7f * (2.0f * s_xn[i] + dwin_up);
        cid_out[i] = best;
        s_pub_best[i & (ONL_WIN - 1)] = best;
        s_pub_eta[i & (ONL_WIN - 1)] = eta;
    };
    int last1_id = -1, last2_id = -1;                         // centroids written in the previous two iterations ...
    float last1_v = 0.0f, last2_v = 0.0f;                     // ... and this thread's element of what was written
    auto step = [&](int64_t i, OnlSlot& sl) {
        if (wave == 0) {
            float lo[4], hi[4], dl[4];
            float hmin = INFINITY;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = lane + 64 * u;
                const bool valid = c < eff_k;
                dl[u] = s_delta[c];
                const float m = dl[u] == 0.0f ? 0.0f : sl.d[u] * rel + dl[u] * 1.0001f;
                lo[u] = valid ? sl.d[u] - m : INFINITY;
                hi[u] = valid ? sl.d[u] + m : INFINITY;
                hmin = fminf(hmin, hi[u]);
            }
            hmin = onl_wave_min(hmin);                       // (DPP rotations + four lane reads: the ds_bpermute butterfly
                                                             //  this replaces was six dependent LDS round trips per row)
            unsigned long long cm[4];
            int total = 0;
            bool moved = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool cnd = lo[u] <= hmin;
                cm[u] = __ballot(cnd);
                total += (int)__popcll(cm[u]);
                moved = moved || (cnd && dl[u] != 0.0f);
            }
            const bool any_moved = __ballot(moved) != 0ull;
            if (total == 0) {
                if (lane == 0) { s_pub_ncand[i & (ONL_WIN - 1)] = 1; commit(i, 0, 0.0f); }
            } else if (total == 1) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (lo[u] <= hmin) { s_pub_ncand[i & (ONL_WIN - 1)] = 1; commit(i, lane + 64 * u, hi[u]); }
            } else if (!any_moved) {
                float bd = INFINITY;
                int best = 0x7fffffff;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (lo[u] <= hmin && sl.d[u] < bd) { bd = sl.d[u]; best = lane + 64 * u; }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float od = __shfl_xor(bd, off);
                    const int ob = __shfl_xor(best, off);
                    if (od < bd || (od == bd && ob < best)) { bd = od; best = ob; }
                }
                if (lane == 0) { s_pub_ncand[i & (ONL_WIN - 1)] = 1; commit(i, best, bd); }
            } else {
                int base = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (lo[u] <= hmin) {
                        const int p = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cm[u] >> 32),
                                                  __builtin_amdgcn_mbcnt_lo((unsigned)cm[u], 0u));
                        s_cand[p] = lane + 64 * u;
                    }
                    base += (int)__popcll(cm[u]);
                }
                if (lane == 0) s_pub_ncand[i & (ONL_WIN - 1)] = total;
            }
            // publish: the entry's fields were written by lanes of this wave, LDS keeps a wave's accesses in order
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) *(volatile int*)&s_seq = (int)(i + 1);
        } else {
            // the other waves follow wave 0's decisions through the mailbox: no barrier on the clear rows
            int spins = 0;
            while (*(volatile int*)&s_seq <= (int)i) {
                __builtin_amdgcn_s_sleep(ONL_POLL_SLEEP);
                if (++spins > (1 << 24)) break;              // (never in a correct run: do not hang the device)
            }
            asm volatile("" ::: "memory");
        }
        const int ncand = *(volatile int*)&s_pub_ncand[i & (ONL_WIN - 1)];
        if (ncand > 1) {                                    // (workgroup-uniform) re-score the candidates
            if (owner) s_row[el] = sl.x;
            __threadfence_block();                          // every thread's centroid stores are done ...
            __syncthreads();                                // ... before any wave reads whole rows
            for (int ci = wave; ci < ncand; ci += 16) {
                const float d = centroid_dist_wave(s_row, centroids + (int64_t)s_cand[ci] * D, D, lane, vec4 != 0);
                if (lane == 0) s_fresh[ci] = d;
            }
            __syncthreads();
            if (wave == 0) {
                float bd = INFINITY;
                int bi = 0x7fffffff;
                for (int ci = lane; ci < ncand; ci += 64) {
                    const float d = s_fresh[ci];
                    if (d < bd) { bd = d; bi = ci; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float od = __shfl_xor(bd, off);
                    const int ob = __shfl_xor(bi, off);
                    if (od < bd || (od == bd && ob < bi)) { bd = od; bi = ob; }
                }
                if (lane == 0) commit(i, s_cand[bi == 0x7fffffff ? 0 : bi], bd * (1.0f + rel));
            }
            __syncthreads();
        }
        const int best = *(volatile int*)&s_pub_best[i & (ONL_WIN - 1)];
        const float eta = *(volatile float*)&s_pub_eta[i & (ONL_WIN - 1)];
        const float one_m = 1.0f - eta;
        if (owner) {
            float c_old;
            if (best == last1_id) c_old = last1_v;           // the newest value of that row is still in a register
            else if (best == last2_id) c_old = last2_v;
            else if (best == sl.cid) c_old = sl.cpre;        // prefetched behind every older store of this thread
            else c_old = centroids[(int64_t)best * D + el];  // prediction missed: one L2 round trip
            const float c_new = one_m * c_old + eta * sl.x;
            centroids[(int64_t)best * D + el] = c_new;
            last2_v = last1_v; last1_v = c_new;
        }
        last2_id = last1_id; last1_id = best;
        prefetch(sl, i + 3);                                 // (issued behind this iteration's store)
        // the mailbox has ONL_WIN entries: one barrier per window keeps wave 0 from lapping the slowest wave
        if (((i + 1) & (ONL_WIN - 1)) == 0) __syncthreads();
    };
    OnlSlot a, b, c;
    a.cid = b.cid = c.cid = -1;
    prefetch(a, 0); prefetch(b, 1); prefetch(c, 2);
    for (int64_t i = 0; i < n; i += 3) {
        step(i, a);
        if (i + 1 < n) step(i + 1, b);
        if (i + 2 < n) step(i + 2, c);
    }
    __syncthreads();
    if (tid < eff_k) counts[tid] = s_counts[tid];
}

__global__ __launch_bounds__(256) void bank_decay_kernel(float* meta, float factor, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) meta[i * 4] *= factor;
}

__global__ __launch_bounds__(256) void bank_gather_kernel(const float* __restrict__ bank,
                                                          const int32_t* __restrict__ idx,
                                                          float* __restrict__ out, int64_t n,
                                                          int64_t D, int64_t rows) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int32_t r = idx[i];
    const bool ok = r >= 0 && r < rows;                      // anything else gathers zeros
    for (int64_t c = lane; c < D; c += 64) out[i * D + c] = ok ? bank[(int64_t)r * D + c] : 0.0f;
}

// ------------------------------------------------------------------------------------------
// Scan kernel
// ------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 512;
constexpr int BK = 32;          // k-tile depth
constexpr int LDS_STRIDE = 36;  // floats per LDS row: 144 B keeps ds_read_b128 of 32 rows conflict-free
constexpr int MODE_DENSE = 0;
constexpr int MODE_FILTER = 1;
constexpr int MODE_ASSIGN = 2;
// per-query candidate counters live on separate 128-byte lines: ~800 atomics per query per
// launch would otherwise serialise 32 queries' traffic on one L2 line
constexpr int CNT_STRIDE = 32;

struct ScanArgs {
    const float* bank;
    const float* inv_norm;
    const float* meta;
    const float* loc;
    const float* queries;  // [nq][D] (this query block's base already applied)
    const float* inv_q;    // [nq]
    const float* q_loc;    // [nq][sdims] or null
    int sdims;
    float now;
    int64_t row_begin, row_end;  // scanned row range
    int64_t D;
    int nq;                 // queries in this block
    // Sampling geometry.  Rows are grouped in LOGICAL tiles of `logical_rows` (128); logical tile
    // L is a sample tile iff L % tile_step == 0 && L / tile_step < n_sample_tiles.
    //   DENSE : workgroup bx scores kernel-tile (bx % sub) of logical tile (bx / sub) * tile_step,
    //           sub = logical_rows / BR; dense column = bx * BR + local row
    //   FILTER: every row that is not in a sample tile
    int tile_step;
    int n_sample_tiles;
    int logical_rows;
    int64_t n_items;        // FILTER v2: number of non-sample 32-row tiles in [row_begin, row_end)
    // DENSE output
    float* dense;           // [nq][dense_ld]; column = blockIdx.x * BR + local row
    int64_t dense_ld;
    float* gmax;            // optional [nq][gmax_ld]: max score of each 32-row group
    int64_t gmax_ld;        //   (group = blockIdx.x * BR/32 + 32-row sub-tile of the workgroup)
    // FILTER output
    const uint32_t* thr;    // [nq] ordered keys
    int32_t* cnt;           // [nq]
    float* cand_scores;     // [nq][cap]
    int32_t* cand_idx;      // [nq][cap]
    int cap;
    // optional centroid-candidate mask: bit c of probe_mask[q][8] set <=> centroid c is probed
    // by query q; rows whose centroid id is not probed are not candidates (hippocampal.py:264-270)
    const uint32_t* probe_mask;
    // ASSIGN mode (k-means): "queries" are centroids; assign_out[row] = argmin_c(|c|^2 - 2 x.c)
    const float* qnorm2;    // [nq]
    int32_t* assign_out;    // [rows]
    int dbg;                // timing ablations (AURA_SCAN_DBG), results invalid when non-zero
};

template <int WQ, int WR, int RT, int MODE, bool VEC4>
__global__ __launch_bounds__(SCAN_THREADS) void knn_scan_kernel(const ScanArgs a) {
    static_assert(WQ * WR == 8, "8 waves");
    constexpr int BQ = WQ * 32;
    constexpr int BR = WR * RT * 32;
    constexpr int NV = (BQ + BR) * (BK / 4);                     // float4 slots per k-tile
    constexpr int NLD = (NV + SCAN_THREADS - 1) / SCAN_THREADS;  // per-thread loads
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;                    // [BQ][LDS_STRIDE]
    float* Bs = smem + BQ * LDS_STRIDE;  // [BR][LDS_STRIDE]

    int64_t row0;
    if (MODE == MODE_DENSE) {
        const int sub = a.logical_rows / BR;   // kernel tiles per logical tile (>= 1)
        row0 = a.row_begin + ((int64_t)blockIdx.x / sub) * a.tile_step * a.logical_rows +
               ((int64_t)blockIdx.x % sub) * BR;
    } else {
        const int64_t tile = blockIdx.x;       // FILTER / ASSIGN: BR == logical_rows
        if (MODE == MODE_FILTER && a.tile_step > 0 && tile % a.tile_step == 0 &&
            tile / a.tile_step < a.n_sample_tiles)
            return;  // scored densely by the sample pass
        row0 = a.row_begin + tile * BR;
    }
    const int q0 = blockIdx.y * BQ;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wq = wave % WQ, wr = wave / WQ;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t D = a.D;

    f32x16 acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.0f;

    float4 pre[NLD];
    auto load_tile = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int f = tid + i * SCAN_THREADS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < NV) {
                const int r = f >> 3, c = (f & 7) * 4;
                const int64_t k = k0 + c;
                const float* src = nullptr;
                if (r < BQ) {
                    if (q0 + r < a.nq) src = a.queries + (int64_t)(q0 + r) * D + k;
                } else {
                    const int64_t row = row0 + (r - BQ);
                    if (row < a.row_end) src = a.bank + row * D + k;
                }
                if (src) {
                    if (VEC4) {
                        if (k < D) v = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (k + 0 < D) v.x = src[0];
                        if (k + 1 < D) v.y = src[1];
                        if (k + 2 < D) v.z = src[2];
                        if (k + 3 < D) v.w = src[3];
                    }
                }
            }
            pre[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int f = tid + i * SCAN_THREADS;
            if (f < NV) {
                const int r = f >> 3, c = (f & 7) * 4;
                *reinterpret_cast<float4*>(smem + r * LDS_STRIDE + c) = pre[i];
            }
        }
    };

    const int64_t KT = (D + BK - 1) / BK;
    load_tile(0);
    for (int64_t kt = 0; kt < KT; ++kt) {
        if (!(a.dbg & 2) || kt == 0) {
            store_tile();
            __syncthreads();
        }
        if (kt + 1 < KT && !(a.dbg & 1)) load_tile((kt + 1) * BK);
        const float* qrow = Qs + (wq * 32 + li) * LDS_STRIDE + 4 * lh;
        const float* brow = Bs + ((wr * RT) * 32 + li) * LDS_STRIDE + 4 * lh;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const float4 av = *reinterpret_cast<const float4*>(qrow + kk * 8);
            float4 bv[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r)
                bv[r] = *reinterpret_cast<const float4*>(brow + r * 32 * LDS_STRIDE + kk * 8);
            const float af[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const float bf = j == 0 ? bv[r].x : j == 1 ? bv[r].y : j == 2 ? bv[r].z : bv[r].w;
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf, acc[r], 0, 0, 0);
                }
            }
        }
        if (!(a.dbg & 2)) __syncthreads();
    }

    // ---- epilogue on the accumulators ------------------------------------------------------
    // lane holds bank row (li) x 16 "queries" {(e&3) + 8*(e>>2) + 4*lh} of its wave's 32
    if (MODE == MODE_ASSIGN) {
        // nearest centroid per bank row: argmin_c (|c|^2 - 2 x.c), ties -> lower c
        float* red_v = smem;                                   // [WQ][BR]
        int* red_c = reinterpret_cast<int*>(smem + WQ * BR);   // [WQ][BR]
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            float bv = INFINITY;
            int bc = 0x7fffffff;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = q0 + wq * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (c < a.nq) {
                    const float val = a.qnorm2[c] - 2.0f * acc[r][e];
                    if (val < bv || (val == bv && c < bc)) { bv = val; bc = c; }
                }
            }
            const float ov = __shfl_xor(bv, 32);
            const int oc = __shfl_xor(bc, 32);
            if (ov < bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }
            if (lh == 0) {
                const int lrow = (wr * RT + r) * 32 + li;
                red_v[wq * BR + lrow] = bv;
                red_c[wq * BR + lrow] = bc;
            }
        }
        __syncthreads();
        for (int t = tid; t < BR; t += SCAN_THREADS) {
            float bv = red_v[t];
            int bc = red_c[t];
#pragma unroll
            for (int w = 1; w < WQ; ++w) {
                const float ov = red_v[w * BR + t];
                const int oc = red_c[w * BR + t];
                if (ov < bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }
            }
            const int64_t row = row0 + t;
            if (row < a.row_end) a.assign_out[row] = bc;
        }
        return;
    }

    // combined score (hippocampal.py:279-303)
    uint32_t* s_mask = reinterpret_cast<uint32_t*>(smem);  // [BQ][8], staged after the k loop
    if (a.probe_mask) {
        for (int t = tid; t < BQ * 8; t += SCAN_THREADS) {
            const int q = q0 + (t >> 3);
            s_mask[t] = q < a.nq ? a.probe_mask[(int64_t)q * 8 + (t & 7)] : 0u;
        }
        __syncthreads();
    }
    float iq[16];
    uint32_t thr[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int q = q0 + wq * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        iq[e] = q < a.nq ? a.inv_q[q] : 0.0f;
        thr[e] = (MODE == MODE_FILTER && q < a.nq) ? a.thr[q] : 0xffffffffu;
    }
    // pass 1: turn every accumulator into its combined score; remember which pairs are candidates
    unsigned long long pass = 0ull;   // bit r*16+e
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int lrow = (wr * RT + r) * 32 + li;
        const int64_t row = row0 + lrow;
        const bool vrow = row < a.row_end;
        float inv_m = 0.f, strength = 0.f, tw = 0.f;
        int cid = -1;
        float lx[4] = {0.f, 0.f, 0.f, 0.f};
        if (vrow) {
            inv_m = a.inv_norm[row];
            const float4 m = *reinterpret_cast<const float4*>(a.meta + row * 4);
            strength = m.x;
            const float age = a.now - m.y;
            tw = 0.2f * expf(-age / 3600.0f);
            cid = (int)m.z;
            if (a.q_loc)
                for (int d = 0; d < a.sdims && d < 4; ++d) lx[d] = a.loc[row * a.sdims + d];
        }
        const bool cid_ok = cid >= 0 && cid < 256;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ql = wq * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const int q = q0 + ql;
            const float sim = acc[r][e] * iq[e] * inv_m;
            float comb = 0.5f * sim;
            if (a.q_loc && q < a.nq) {
                float d2 = 0.0f;
                for (int d = 0; d < a.sdims && d < 4; ++d) {
                    const float df = lx[d] - a.q_loc[(int64_t)q * a.sdims + d];
                    d2 = d2 + df * df;
                }
                comb = comb + 0.3f * (1.0f / (1.0f + sqrtf(d2)));
            }
            comb = (comb + tw) * strength;
            bool cand = vrow && q < a.nq;
            if (a.probe_mask)
                cand = cand && cid_ok && ((s_mask[ql * 8 + (cid >> 5)] >> (cid & 31)) & 1u);
            if (MODE == MODE_DENSE) {
                const float sc = cand ? comb : -INFINITY;
                if (q < a.nq)
                    a.dense[(int64_t)q * a.dense_ld + (int64_t)blockIdx.x * BR + lrow] = sc;
                acc[r][e] = sc;
            } else {
                acc[r][e] = comb;
                if (cand && ord_key(comb) >= thr[e]) pass |= 1ull << (r * 16 + e);
            }
        }
        if (MODE == MODE_DENSE && a.gmax) {
            // max over the 32 rows (lanes li) of this 32-row group, per query
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float m = acc[r][e];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
                const int q = q0 + wq * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (li == 0 && q < a.nq)
                    a.gmax[(int64_t)q * a.gmax_ld + (int64_t)blockIdx.x * (BR / 32) + wr * RT + r] = m;
            }
        }
    }
    if (MODE == MODE_FILTER && pass != 0ull) {
        // pass 2: one slot reservation per (lane, query): up to 16 independent atomics are in
        // flight together instead of one blocking round trip per candidate
        int pos[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int n = 0;
#pragma unroll
            for (int r = 0; r < RT; ++r) n += (int)((pass >> (r * 16 + e)) & 1ull);
            const int q = q0 + wq * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            pos[e] = n > 0 ? atomicAdd(a.cnt + (int64_t)q * CNT_STRIDE, n) : 0;
        }
        // pass 3: write the candidates
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int q = q0 + wq * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            int p = pos[e];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                if ((pass >> (r * 16 + e)) & 1ull) {
                    if (p < a.cap) {
                        a.cand_scores[(int64_t)q * a.cap + p] = acc[r][e];
                        a.cand_idx[(int64_t)q * a.cap + p] =
                            (int32_t)(row0 + (wr * RT + r) * 32 + li);
                    }
                    ++p;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Main scan, v2: persistent FILTER kernel for 256-query blocks.
//   * one 512-thread workgroup per CU walks a balanced, contiguous span of the non-sample 32-row
//     tiles (4 tiles = 128 rows per chunk), so the chip is evenly loaded whatever N is;
//   * LDS is double buffered (2 x 54 KiB): one barrier per 32-deep k-tile; the next k-tile's
//     global loads are in flight during the whole compute phase and are written to the other
//     buffer by the two SIMD partners at DIFFERENT points of their MFMA streams (waves 0-3
//     before, waves 4-7 after their first 16 MFMAs), so one partner always feeds the matrix pipe;
//   * A/B fragments for MFMA group kk+1 are read from LDS before the 16 MFMAs of group kk.
// Arithmetic (k order of every dot product, epilogue) is identical to knn_scan_kernel, so the
// dense and filter paths return bit-identical results.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t filter_tile_of(int64_t j, int step, int ns) {
    // j-th non-sample 32-row tile -> 32-row tile index (sample logical tiles hold 4 of them)
    const int64_t s = (int64_t)step * 4, per = s - 4;
    if (ns > 0 && j < (int64_t)ns * per) return (j / per) * s + 4 + (j % per);
    return (int64_t)ns * s + (j - (int64_t)ns * per);
}

// FAST = no location term and no centroid mask (the common case): those branches and their
// registers are compiled out of the epilogue.
template <bool VEC4, bool FAST>
__global__ __launch_bounds__(SCAN_THREADS) void knn_scan_filter_v2(const ScanArgs a) {
    constexpr int BQ = 256, BRR = 128, RT = 4;
    constexpr int TILE_FLOATS = (BQ + BRR) * LDS_STRIDE;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const buf0 = smem;
    float* const buf1 = smem + TILE_FLOATS;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int64_t D = a.D;
    const int64_t KT = (D + BK - 1) / BK;
    // work space = (256-query block) x (non-sample 32-row tile); a workgroup owns a contiguous
    // span of it, cut into chunks of up to 4 tiles that never straddle two query blocks
    const int64_t nqblk = (a.nq + BQ - 1) / BQ;
    const int64_t total = a.n_items * nqblk;
    const int64_t lo = total * (int64_t)blockIdx.x / gridDim.x;
    const int64_t hi = total * ((int64_t)blockIdx.x + 1) / gridDim.x;
    if (lo >= hi) return;

    // staging slots of this thread: 4 query rows + 2 bank rows (both per chunk)
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const int lds_slot = srow * LDS_STRIDE + scol;

    int64_t c = lo;
    while (c < hi) {
        const int64_t qblk = c / a.n_items, j0 = c - qblk * a.n_items;
        int64_t nt = hi - c;
        if (nt > RT) nt = RT;
        if (nt > a.n_items - j0) nt = a.n_items - j0;
        const int qoff = (int)qblk * BQ;                 // first query of this chunk's block
        c += nt;
        const float* qsrc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = qoff + srow + 64 * i;
            qsrc[i] = q < a.nq ? a.queries + (int64_t)q * D + scol : nullptr;
        }
        int64_t row0[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r)
            row0[r] = (r < nt)
                          ? a.row_begin + filter_tile_of(j0 + r, a.tile_step, a.n_sample_tiles) * 32
                          : a.row_end;
        const float* bsrc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int br = srow + 64 * i;            // 0..127
            const int64_t row = row0[br >> 5] + (br & 31);
            bsrc[i] = row < a.row_end ? a.bank + row * D + scol : nullptr;
        }

        f32x16 acc[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][e] = 0.0f;

        float4 pre[6];
        auto gload = [&](int64_t k0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float* src = i < 4 ? qsrc[i] : bsrc[i - 4];
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int64_t k = k0 + scol;
                if (src) {
                    if (VEC4) {
                        if (k < D) v = *reinterpret_cast<const float4*>(src + k0);
                    } else {
                        if (k + 0 < D) v.x = src[k0 + 0];
                        if (k + 1 < D) v.y = src[k0 + 1];
                        if (k + 2 < D) v.z = src[k0 + 2];
                        if (k + 3 < D) v.w = src[k0 + 3];
                    }
                }
                pre[i] = v;
            }
        };
        auto lstore = [&](float* buf) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                *reinterpret_cast<float4*>(buf + lds_slot + i * 64 * LDS_STRIDE) = pre[i];
        };

        gload(0);
        lstore(buf0);
        if (KT > 1) gload(BK);
        __syncthreads();

        // k loop, specialised on the number of real row tiles in the chunk (NT): a partial chunk
        // issues only NT/4 of the MFMAs, so spans of 11 tiles (4+4+3) cost 11, not 12
        auto kloop = [&](auto nt_tag) {
            constexpr int NT = decltype(nt_tag)::value;
            for (int64_t kt = 0; kt < KT; ++kt) {
                const float* cur = (kt & 1) ? buf1 : buf0;
                float* nxt = (kt & 1) ? buf0 : buf1;
                const bool has1 = kt + 1 < KT, has2 = kt + 2 < KT;
                const float* qrow = cur + (wave * 32 + li) * LDS_STRIDE + 4 * lh;
                const float* brow = cur + (BQ + li) * LDS_STRIDE + 4 * lh;

                if (wave < 4) {                 // early stagers
                    if (has1) lstore(nxt);
                    if (has2) gload((kt + 2) * BK);
                }
                float4 av[2], bv[2][NT];
                av[0] = *reinterpret_cast<const float4*>(qrow);
#pragma unroll
                for (int r = 0; r < NT; ++r)
                    bv[0][r] = *reinterpret_cast<const float4*>(brow + r * 32 * LDS_STRIDE);
#pragma unroll
                for (int kk = 0; kk < BK / 8; ++kk) {
                    const int s0 = kk & 1, s1 = s0 ^ 1;
                    if (kk + 1 < BK / 8) {
                        av[s1] = *reinterpret_cast<const float4*>(qrow + (kk + 1) * 8);
#pragma unroll
                        for (int r = 0; r < NT; ++r)
                            bv[s1][r] = *reinterpret_cast<const float4*>(brow + r * 32 * LDS_STRIDE +
                                                                         (kk + 1) * 8);
                    }
                    const float af[4] = {av[s0].x, av[s0].y, av[s0].z, av[s0].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int r = 0; r < NT; ++r) {
                            const float bf = j == 0 ? bv[s0][r].x : j == 1 ? bv[s0][r].y
                                             : j == 2 ? bv[s0][r].z : bv[s0][r].w;
                            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf, acc[r], 0, 0, 0);
                        }
                    }
                    if (kk == 0 && wave >= 4) {  // late stagers: after their first MFMA group
                        if (has1) lstore(nxt);
                        if (has2) gload((kt + 2) * BK);
                    }
                }
                __syncthreads();
            }
        };
        if (nt == 4) kloop(std::integral_constant<int, 4>{});
        else if (nt == 3) kloop(std::integral_constant<int, 3>{});
        else if (nt == 2) kloop(std::integral_constant<int, 2>{});
        else kloop(std::integral_constant<int, 1>{});

        // ---- epilogue: combined score + candidate append (same arithmetic as knn_scan_kernel) ----
        uint32_t* s_mask = reinterpret_cast<uint32_t*>(smem);
        if (!FAST && a.probe_mask) {
            for (int t = tid; t < BQ * 8; t += SCAN_THREADS) {
                const int q = qoff + (t >> 3);
                s_mask[t] = q < a.nq ? a.probe_mask[(int64_t)q * 8 + (t & 7)] : 0u;
            }
            __syncthreads();
        }
        // per-lane query constants: re-read per chunk (L2 hits) rather than held in 32 VGPRs
        float iq[16], thrf[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int q = qoff + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            iq[e] = q < a.nq ? a.inv_q[q] : 0.0f;
            thrf[e] = q < a.nq ? ord_unkey(a.thr[q]) : __builtin_nanf("");   // same order as the key compare; padding
                                                                              // queries: NaN ("comb >= NaN" never holds, comb may be +inf)
        }
        unsigned long long pass = 0ull;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int64_t row = row0[r] + li;
            const bool vrow = row < a.row_end;
            float inv_m = 0.f, strength = 0.f, tw = 0.f;
            int cid = -1;
            float lx[4] = {0.f, 0.f, 0.f, 0.f};
            if (vrow) {
                inv_m = a.inv_norm[row];
                const float4 m = *reinterpret_cast<const float4*>(a.meta + row * 4);
                strength = m.x;
                const float age = a.now - m.y;
                tw = 0.2f * expf(-age / 3600.0f);
                cid = (int)m.z;
                if (!FAST && a.q_loc)
                    for (int d = 0; d < a.sdims && d < 4; ++d) lx[d] = a.loc[row * a.sdims + d];
            }
            if (FAST) {
                unsigned int bits = 0u;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float sim = acc[r][e] * iq[e] * inv_m;
                    const float comb = (0.5f * sim + tw) * strength;
                    acc[r][e] = comb;
                    bits |= (comb >= thrf[e]) ? (1u << e) : 0u;
                }
                if (!vrow) bits = 0u;
                pass |= (unsigned long long)bits << (r * 16);
            } else {
                const bool cid_ok = cid >= 0 && cid < 256;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int q = qoff + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const float sim = acc[r][e] * iq[e] * inv_m;
                    float comb = 0.5f * sim;
                    if (a.q_loc && q < a.nq) {
                        float d2 = 0.0f;
                        for (int d = 0; d < a.sdims && d < 4; ++d) {
                            const float df = lx[d] - a.q_loc[(int64_t)q * a.sdims + d];
                            d2 = d2 + df * df;
                        }
                        comb = comb + 0.3f * (1.0f / (1.0f + sqrtf(d2)));
                    }
                    comb = (comb + tw) * strength;
                    bool cand = vrow && q < a.nq;
                    if (a.probe_mask)
                        cand = cand && cid_ok && ((s_mask[(q - qoff) * 8 + (cid >> 5)] >> (cid & 31)) & 1u);
                    acc[r][e] = comb;
                    if (cand && comb >= thrf[e]) pass |= 1ull << (r * 16 + e);
                }
            }
        }
        if (pass != 0ull) {
            int pos[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int n = 0;
#pragma unroll
                for (int r = 0; r < RT; ++r) n += (int)((pass >> (r * 16 + e)) & 1ull);
                const int q = qoff + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                pos[e] = n > 0 ? atomicAdd(a.cnt + (int64_t)q * CNT_STRIDE, n) : 0;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int q = qoff + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                int p = pos[e];
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    if ((pass >> (r * 16 + e)) & 1ull) {
                        if (p < a.cap) {
                            a.cand_scores[(int64_t)q * a.cap + p] = acc[r][e];
                            a.cand_idx[(int64_t)q * a.cap + p] = (int32_t)(row0[r] + li);
                        }
                        ++p;
                    }
                }
            }
        }
        if (!FAST && a.probe_mask) __syncthreads();   // s_mask aliases the staging buffers
    }
}

// ------------------------------------------------------------------------------------------
// Centroid probe: for each query the `nprobe` nearest (L2, unnormalised query) of the 256
// centroid rows -> 256-bit mask (+ ids in distance order).  hippocampal.py:261-262: the topk runs
// over ALL 256 buffer rows, also the zero rows beyond centroids_k.
//
// One fused kernel (round 3; rounds 1-2: a VALU distance kernel, 43 us per 2048 queries, and a
// 256-thread argmin kernel, 25 us).  For a fixed query, ||c - q||^2 = ||q||^2 + (||c||^2 - 2 q.c), so the
// ranking key is ||c||^2 - 2 q.c: the query's own norm -- 768 for a Gaussian query, where the key is O(10)
// -- never enters, which makes the key MORE accurate in fp32 than the direct sum of 768 squares.  q.c runs
// on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation).
//   workgroup = 16 queries x 256 centroids, 8 waves; wave w owns centroids [32 w, 32 w + 32) as two
//   16 x 16 output tiles.  Lane (r = lane & 15, h = lane >> 4) loads 16 bytes of query row r and of
//   centroid rows 16 t + r at k = 16 j + 4 h: element m of those loads is the lane's A / B value of the
//   j-th group's m-th MFMA (any k permutation is fine as long as A and B use the same one).  ||c||^2
//   falls out of the same loads (the lane's partial over its k, summed over the four h lanes).
//   Keys go to LDS; then 16 lanes per query hold 16 keys each and run `nprobe` rounds of
//   (local argmin, 4-step butterfly), ties to the lower centroid; the winner leaves through an alive mask.
//   The per-list query lists of the inverted-list recall (lq_cnt / lq_list, optional) are filled here:
//   one returning atomic per probe on the list's counter (zeroed by the caller); the order inside a
//   list is whatever the atomics make it (results do not depend on it).
// ------------------------------------------------------------------------------------------
constexpr int PR_Q = 16;                   // queries per workgroup
constexpr int PR_WAVES = 8;                // waves per workgroup: wave w owns centroids [32 w, 32 w + 32)
constexpr int PR_T = 256 / PR_WAVES / 16;  // 16 x 16 output tiles per wave
constexpr int PR_G = 4;                    // k-groups (16 k each) per software-pipeline stage
constexpr int PR_QLDS_MAX_D = 1024;        // query tile in LDS up to this many columns (16 x 1028 x 4 B = 64 KiB, beside
                                           // 8 x 8 KiB of centroid stages and 16 KiB of keys)
constexpr int PR_KSTRIDE = 260;            // floats per key row in LDS (260 % 32 = 4: the four query groups of a
                                           // wave read different banks)

template <bool QLDS, bool VEC>             // QLDS: the workgroup's 16 queries staged in LDS once (D <= PR_QLDS_MAX_D);
                                           // VEC: 16-byte loads (D % 4 == 0, aligned bases)
__global__ __launch_bounds__(64 * PR_WAVES) void centroid_probe_kernel(const float* __restrict__ centroids,
                                                             const float* __restrict__ queries,
                                                             int64_t D, int nq, int nprobe,
                                                             uint32_t* __restrict__ mask_out,
                                                             int32_t* __restrict__ ids_out,
                                                             int32_t* __restrict__ lq_cnt,
                                                             int32_t* __restrict__ lq_list, int lq_stride) {
    __shared__ float s_key[PR_Q * PR_KSTRIDE];
    extern __shared__ __attribute__((aligned(16))) float s_qt[];   // QLDS: [16][D16 + 4] query tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, h = lane >> 4;
    const int q0 = blockIdx.x * PR_Q;
    const int c0 = wave * (16 * PR_T);
    const int64_t D16 = (D + 15) / 16 * 16;
    const int64_t QS = D16 + 4;                               // row stride: 16 rows x ds_read_b128 hit 64 different banks
    const int qrow = q0 + r < nq ? q0 + r : nq - 1;          // rows beyond nq repeat the last query (never written)
    const float* const qp = queries + (int64_t)qrow * D;
    const float* cp[PR_T];
#pragma unroll
    for (int t = 0; t < PR_T; ++t) cp[t] = centroids + (int64_t)(c0 + 16 * t + r) * D;
    // Branch-free loads: the address is clamped into the row and lanes beyond D are zeroed by a select.  (With
    // `if (k < D)` around the load hipcc built a diamond per load, each with its own s_waitcnt vmcnt(0): every
    // load waited for the one before it -- 73-103 us per 2048 queries whatever the pipeline depth.)
    // (the zeroing is a separate step so that a prefetch can leave it to the point of use: applied at load time
    //  the select waits for the load it belongs to, and the prefetch is no prefetch)
    auto ld_raw = [&](const float* base, int64_t k) -> float4 {
        float4 v;
        if (VEC) {
            const int64_t kk = k < D ? k : D - 4;            // D % 4 == 0, D >= 4
            v = *reinterpret_cast<const float4*>(base + kk);
        } else {
            const int64_t l = D - 1;
            v.x = base[k + 0 < D ? k + 0 : l]; v.y = base[k + 1 < D ? k + 1 : l];
            v.z = base[k + 2 < D ? k + 2 : l]; v.w = base[k + 3 < D ? k + 3 : l];
        }
        return v;
    };
    auto ld_mask = [&](float4 v, int64_t k) -> float4 {
        if (VEC) {
            const bool in = k < D;
            v.x = in ? v.x : 0.0f; v.y = in ? v.y : 0.0f; v.z = in ? v.z : 0.0f; v.w = in ? v.w : 0.0f;
        } else {
            v.x = k + 0 < D ? v.x : 0.0f; v.y = k + 1 < D ? v.y : 0.0f;
            v.z = k + 2 < D ? v.z : 0.0f; v.w = k + 3 < D ? v.w : 0.0f;
        }
        return v;
    };
    auto ld = [&](const float* base, int64_t k) -> float4 { return ld_mask(ld_raw(base, k), k); };
    f32x4v_t acc[PR_T];
    float cn[PR_T];
#pragma unroll
    for (int t = 0; t < PR_T; ++t) { acc[t] = f32x4v_t{0.f, 0.f, 0.f, 0.f}; cn[t] = 0.0f; }
    // one stage = PR_G k-groups: (1 + PR_T) PR_G 16-byte loads per lane in flight while the previous stage's
    // 4 PR_T PR_G MFMAs run; two waves per SIMD cover the rest of the L2 latency (the first version kept one
    // k-group in flight with one wave per SIMD: 103 us per 2048 queries, all of it load latency)
    if (QLDS) {
        // The query rows come from HBM (read once, never in L2): loaded per stage they put an HBM round trip
        // in front of every stage's MFMAs (79 us per 2048 queries).  One burst instead: the whole 16 x D tile,
        // every load in flight at once, zero padded to a multiple of 16 columns.
        for (int64_t i = tid; i < 16 * (D16 / 4); i += 64 * PR_WAVES) {
            const int64_t row = i / (D16 / 4), kc = (i - row * (D16 / 4)) * 4;
            const int qr = q0 + (int)row < nq ? q0 + (int)row : nq - 1;
            *reinterpret_cast<float4*>(s_qt + row * QS + kc) = ld(queries + (int64_t)qr * D, kc);
        }
        __syncthreads();
    }
    constexpr int64_t STAGE = 16 * PR_G;
    if constexpr (QLDS) {
        // Centroid rows through a wave-private LDS stage.  Loaded straight into the MFMA layout, lane (r, h) reads
        // 16 bytes of row r: neighbouring lanes sit 3 KB apart, 64 tag look-ups per load instruction (46 us per
        // 2048 queries, load-issue bound).  Instead a lane loads CONSECUTIVE memory -- 16 lanes cover 256 bytes
        // of one row, an instruction 4 rows -- the wave writes its 32 rows x 64 k to LDS (16-byte chunk j of row
        // w at position j ^ (w & 15): the writes and the MFMA-layout reads below are both conflict-free) and
        // reads its operands back in the MFMA layout.  The stage is private to the wave: no barrier, the LDS
        // pipe keeps a wave's accesses in order.  The next stage's global loads fly during the MFMAs.
        float* const s_b = s_qt + 16 * QS + wave * (32 * 64);
        const int lrow = lane >> 4, lc = lane & 15;          // loader: row lrow + 4 m of the wave's 32, chunk lc
        float4 nb[8];
        auto load_b = [&](int64_t k0) {
#pragma unroll
            for (int m = 0; m < 8; ++m)
                nb[m] = ld_raw(centroids + (int64_t)(c0 + lrow + 4 * m) * D, k0 + 4 * lc);
        };
        auto store_b = [&](int64_t k0) {                     // (columns beyond D are zeroed here, see ld_mask)
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int w = lrow + 4 * m;
                *reinterpret_cast<float4*>(s_b + w * 64 + 4 * (lc ^ (w & 15))) = ld_mask(nb[m], k0 + 4 * lc);
            }
        };
        // Every workgroup walks k in step, and a 64-column slab of the table (256 rows, 3 KB apart) sits on a
        // few L2 channels: wave w starts its walk at stage w and wraps around, so that the eight waves of a
        // workgroup -- of every workgroup -- pull from eight different slabs at any time.  (The k order of a
        // (query, centroid) dot product then depends on the centroid's wave, i.e. on the centroid alone.)
        const int64_t nst = (D + STAGE - 1) / STAGE;
        const int64_t st0 = wave % nst;
        auto stage_k = [&](int64_t it) { const int64_t s_ = st0 + it; return (s_ < nst ? s_ : s_ - nst) * STAGE; };
        float tch = 0.0f;
        load_b(stage_k(0));
#ifdef AURA_PR_EXP_NOMAIN                                    // timing experiments only (tools/build_variant.sh)
        for (int64_t it = 0; it < 1; ++it) {
#else
        for (int64_t it = 0; it < nst; ++it) {
#endif
            const int64_t k0 = stage_k(it);
            store_b(k0);
            asm volatile("" ::"v"(tch));                     // (the touch below is long done: store_b waited for all loads)
            load_b(stage_k(it + 1 < nst ? it + 1 : it));     // unconditional (the last one reloads its own stage, never
                                                             // used): a branch here costs register copies that wait
                                                             // for the loads they were meant to overlap
            {
                // A stage's lines are new to this XCD's L2 when the first workgroup asks for them: ~1.7 us per stage
                // with one stage of look-ahead (measured: the whole kernel takes as long for 256 queries as for
                // 2048).  One dword per 128-byte line of the stage three ahead starts the fill early: 64 lanes = 32
                // rows x 2 lines, one instruction, no register kept.
                const int64_t kt = stage_k(it + 3 < nst ? it + 3 : nst - 1) + (lane & 1) * 32;
                tch = centroids[(int64_t)(c0 + (lane >> 1)) * D + (kt < D ? kt : D - 1)];
            }
#pragma unroll
            for (int g = 0; g < PR_G; ++g) {
                const int64_t k = k0 + 16 * g + 4 * h;
                const float4 av = k < D16 ? *reinterpret_cast<const float4*>(s_qt + r * QS + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < PR_T; ++t) {
                    const int w = 16 * t + r;
                    const float4 bv = *reinterpret_cast<const float4*>(s_b + w * 64 + 4 * ((4 * g + h) ^ (w & 15)));
                    cn[t] = fmaf(bv.x, bv.x, cn[t]); cn[t] = fmaf(bv.y, bv.y, cn[t]);
                    cn[t] = fmaf(bv.z, bv.z, cn[t]); cn[t] = fmaf(bv.w, bv.w, cn[t]);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[t], 0, 0, 0);
                }
            }
        }
    } else {
    // two register stages used in turn (no copies between them: a copy at the end of an iteration would wait
    // for the prefetch it was meant to hide)
    float4 a0[PR_G], b0[PR_G][PR_T], a1[PR_G], b1[PR_G][PR_T];
    auto load_stage = [&](float4 (&aa)[PR_G], float4 (&bb)[PR_G][PR_T], int64_t k0) {
#pragma unroll
        for (int g = 0; g < PR_G; ++g) {
            if (!QLDS) aa[g] = ld(qp, k0 + 16 * g + 4 * h);  // (beyond D: zeros, which add nothing)
#pragma unroll
            for (int t = 0; t < PR_T; ++t) bb[g][t] = ld(cp[t], k0 + 16 * g + 4 * h);
        }
    };
    auto compute = [&](float4 (&aa)[PR_G], const float4 (&bb)[PR_G][PR_T], int64_t k0) {
        if (QLDS) {
#pragma unroll
            for (int g = 0; g < PR_G; ++g) {
                const int64_t k = k0 + 16 * g + 4 * h;
                aa[g] = k < D16 ? *reinterpret_cast<const float4*>(s_qt + r * QS + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int g = 0; g < PR_G; ++g) {
#pragma unroll
            for (int t = 0; t < PR_T; ++t) {
                cn[t] = fmaf(bb[g][t].x, bb[g][t].x, cn[t]); cn[t] = fmaf(bb[g][t].y, bb[g][t].y, cn[t]);
                cn[t] = fmaf(bb[g][t].z, bb[g][t].z, cn[t]); cn[t] = fmaf(bb[g][t].w, bb[g][t].w, cn[t]);
            }
#pragma unroll
            for (int t = 0; t < PR_T; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[g].x, bb[g][t].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[g].y, bb[g][t].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[g].z, bb[g][t].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[g].w, bb[g][t].w, acc[t], 0, 0, 0);
            }
        }
    };
    load_stage(a0, b0, 0);
    for (int64_t k0 = 0; k0 < D; k0 += 2 * STAGE) {
        load_stage(a1, b1, k0 + STAGE);
        compute(a0, b0, k0);
        if (k0 + STAGE >= D) break;
        load_stage(a0, b0, k0 + 2 * STAGE);
        compute(a1, b1, k0 + STAGE);
    }
    }
    // ||c||^2 of centroid c0 + 16 t + r: the four h lanes hold its partials
#pragma unroll
    for (int t = 0; t < PR_T; ++t) {
        cn[t] += __shfl_xor(cn[t], 16);
        cn[t] += __shfl_xor(cn[t], 32);
    }
    // C/D layout: lane -> centroid column r, query rows 4 h + e
#pragma unroll
    for (int t = 0; t < PR_T; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            s_key[(4 * h + e) * PR_KSTRIDE + c0 + 16 * t + r] = fmaf(-2.0f, acc[t][e], cn[t]);
    __syncthreads();
    if (wave >= PR_Q / 4) return;
#ifdef AURA_PR_EXP_NOSELECT
    nprobe = 1;
#endif

    // ---- selection: 16 lanes per query (query 4 wave + h of the workgroup), lane r holds centroids r + 16 m ----
    const int ql = 4 * wave + h;
    const int q = q0 + ql;
    // keys as ordered uint32 (smaller float <=> smaller key); 16-lane all-reduce by DPP row rotations: the first
    // version's shuffles (ds_bpermute) cost 1 us per round, 8 us of a 42 us kernel
    uint32_t kv[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) kv[m] = ord_key(s_key[ql * PR_KSTRIDE + r + 16 * m]);
    auto row_min = [](uint32_t v) -> uint32_t {              // min over the lane's row of 16, result in every lane
        uint32_t o;
        o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); v = o < v ? o : v;   // row_ror:8
        o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false); v = o < v ? o : v;   // row_ror:4
        o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false); v = o < v ? o : v;   // row_ror:2
        o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false); v = o < v ? o : v;   // row_ror:1
        return v;
    };
    uint32_t alive = 0xffffu;
    uint32_t mword = 0u;                                     // lane r < 8: word r of the query's 256-bit mask
    int my_id = -1;                                          // lane r < 8: the r-th nearest centroid
    for (int p = 0; p < nprobe; ++p) {
        uint32_t km = 0xffffffffu;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const uint32_t v = ((alive >> m) & 1u) ? kv[m] : 0xffffffffu;
            km = v < km ? v : km;
        }
        km = row_min(km);
        uint32_t bi_u = 0xffffffffu;                         // lowest centroid among the alive entries that hold km
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const uint32_t c = (((alive >> m) & 1u) && kv[m] == km) ? (uint32_t)(r + 16 * m) : 0xffffffffu;
            bi_u = c < bi_u ? c : bi_u;
        }
        bi_u = row_min(bi_u);
        if (bi_u == 0xffffffffu) break;                      // nothing left (all 256 taken)
        const int bi = (int)bi_u;
        if ((bi & 15) == r) alive &= ~(1u << (bi >> 4));
        if ((bi >> 5) == r) mword |= 1u << (bi & 31);
        if (p == r) my_id = bi;
    }
    if (q < nq && r < 8) {
        mask_out[(int64_t)q * 8 + r] = mword;
        if (my_id >= 0) {
            if (ids_out) ids_out[(int64_t)q * 8 + r] = my_id;
            if (lq_cnt) {
                const int slot = atomicAdd(&lq_cnt[my_id], 1);
                if (slot < lq_stride) lq_list[(int64_t)my_id * lq_stride + slot] = (q << 4) | r;
            }
        }
    }
}

// probe = keys + selection in one launch (dist_ws: unused since round 3, kept in the workspace layout)
inline int launch_probe(const float* centroids, const float* queries, int64_t D, int nq, int nprobe,
                        float* dist_ws, uint32_t* mask_out, int32_t* ids_out, hipStream_t s,
                        int32_t* lq_cnt = nullptr, int32_t* lq_list = nullptr, int lq_stride = 0) {
    (void)dist_ws;
    if (nq <= 0) return AURA_OK;
    const dim3 grid((unsigned)((nq + PR_Q - 1) / PR_Q)), block(64 * PR_WAVES);
    const bool vec = (D & 3) == 0 && ((reinterpret_cast<uintptr_t>(queries) | reinterpret_cast<uintptr_t>(centroids)) & 15) == 0;
#define AURA_PROBE(QL, VC, LDS) hipLaunchKernelGGL((centroid_probe_kernel<QL, VC>), grid, block, LDS, s, centroids, queries, D, nq, \
                                                   nprobe, mask_out, ids_out, lq_cnt, lq_list, lq_stride)
    if (D <= PR_QLDS_MAX_D) {
        const size_t lds = ((size_t)16 * ((D + 15) / 16 * 16 + 4) + (size_t)PR_WAVES * 32 * 64) * sizeof(float);
        const int lds_max = (int)((16 * (PR_QLDS_MAX_D + 4) + PR_WAVES * 32 * 64) * 4);
        if (ensure_lds_attr(reinterpret_cast<const void*>(centroid_probe_kernel<true, true>), lds_max) ||
            ensure_lds_attr(reinterpret_cast<const void*>(centroid_probe_kernel<true, false>), lds_max))
            return AURA_E_LAUNCH;
        if (vec) AURA_PROBE(true, true, lds); else AURA_PROBE(true, false, lds);
    } else {
        if (vec) AURA_PROBE(false, true, 0); else AURA_PROBE(false, false, 0);
    }
#undef AURA_PROBE
    return hipGetLastError() == hipSuccess ? AURA_OK : AURA_E_LAUNCH;
}

// ------------------------------------------------------------------------------------------
// Inverted-list (IVF) recall: the reference's centroid-candidate retrieval
// (hippocampal.py:259-270) done the way the index is meant to be used -- each query touches only
// the rows of its `nprobe` nearest centroids.  The bank rows are grouped by centroid through
// `list_rows` (row ids sorted by centroid id, built by the host module from memory_metadata[:,2]);
// a batch of queries is regrouped BY LIST, so every probed list is streamed from HBM once per
// batch and scored against all the queries that probe it:
//   ivf_prepare_kernel : per-list query lists, per-(query, probe) output offsets, 256-row groups
//   ivf_scan_kernel    : one workgroup = (list, 256-row group, 32-query tile); 8 waves x 32 rows,
//                        queries as the MFMA A operand, same k order and epilogue arithmetic as the
//                        full scan (bit-identical scores); scores land at deterministic slots
//                        cand[q][offset(q, probe) + position in list] -- no atomics
//   topk_select_kernel : exact top-k over each query's slots
// HBM-bound: N*D*4 bytes per batch (every list is probed by some query at nq = 256) against
// 2*nq*nprobe*(N/256)*D useful FLOP.
// ------------------------------------------------------------------------------------------
constexpr int IVF_MAXQ = 2048;     // queries per pass (per prepare / scan launch)
constexpr int IVF_GROUP = 128;     // bank rows per work item (4 waves x 32 rows)
constexpr int IVF_THREADS = IVF_GROUP * 2;

struct IvfArgs {
    const float* bank;
    const float* inv_norm;
    const float* meta;
    const float* queries;     // [nq][D]
    const float* inv_q;       // [nq]
    const int32_t* list_rows; // row ids grouped by centroid
    const int32_t* list_off;  // [257] start of each list inside list_rows
    const int32_t* list_len;  // [256]
    const int32_t* lq_cnt;    // [256] queries probing each list
    const int32_t* lq_list;   // [256][IVF_MAXQ] packed (q << 4 | probe slot)
    const int32_t* qbase;     // [nq][8] output offset of each (query, probe slot)
    const int32_t* item_off;  // [257] prefix over lists of ceil(len/128) * ceil(queries/32)
    int32_t* work_counter;    // [1] next unclaimed work item (zeroed by ivf_prepare_kernel)
    float* cand_scores;       // [nq][cap]
    int32_t* cand_idx;
    int cap;
    float now;
    int64_t D;
    int nq;
};

__global__ __launch_bounds__(256) void ivf_prepare_kernel(const int32_t* __restrict__ probe_ids,
                                                          int nprobe, int nq,
                                                          const int32_t* __restrict__ list_len,
                                                          int32_t* lq_cnt, int32_t* lq_list,
                                                          int32_t* qbase, int32_t* qcnt,
                                                          int32_t* item_off, int32_t* work_counter,
                                                          int cap, int qtile, int32_t* overflow) {
    __shared__ int s_cnt[256];
    __shared__ int s_item[257];
    const int tid = threadIdx.x;
    s_cnt[tid] = 0;
    __syncthreads();
    for (int q = tid; q < nq; q += 256) {
        int running = 0;
        for (int p = 0; p < nprobe; ++p) {
            const int c = probe_ids[q * 8 + p];
            qbase[q * 8 + p] = running;
            running += list_len[c];
            const int slot = atomicAdd(&s_cnt[c], 1);
            lq_list[c * IVF_MAXQ + slot] = (q << 4) | p;
        }
        if (running > cap) {
            running = cap;
            if (overflow) *overflow = 1;
        }
        qcnt[(int64_t)q * CNT_STRIDE] = running;
    }
    __syncthreads();
    lq_cnt[tid] = s_cnt[tid];
    s_item[tid + 1] = ((list_len[tid] + IVF_GROUP - 1) / IVF_GROUP) * ((s_cnt[tid] + qtile - 1) / qtile);
    if (tid == 0) { s_item[0] = 0; *work_counter = 0; }
    __syncthreads();
    if (tid == 0)
        for (int c = 1; c <= 256; ++c) s_item[c] += s_item[c - 1];
    __syncthreads();
    item_off[tid] = s_item[tid];
    if (tid == 0) item_off[256] = s_item[256];
}

// Persistent: workgroups claim (list, 128-row group, QT*32-query tile) items from a global counter
// until none is left; items of one row group with different query tiles are adjacent, so the
// group's rows are re-read from L2, not HBM.
// QT = 1 for small batches (a list is probed by ~nq/32 queries), QT = 2 when lists see >= 64
// queries: twice the MFMA work per staged row tile and half the items.
template <bool VEC4, int QT>
__global__ __launch_bounds__(IVF_THREADS) void ivf_scan_kernel(const IvfArgs a) {
    constexpr int BQ = 32 * QT, BR = IVF_GROUP;
    constexpr int NV = (BQ + BR) * (BK / 4);
    constexpr int NLD = (NV + IVF_THREADS - 1) / IVF_THREADS;   // 5 (QT = 1) or 6 (QT = 2)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;
    float* Bs = smem + BQ * LDS_STRIDE;
    __shared__ int s_q[BQ];      // packed (q << 4 | p) of the tile's queries, -1 = none
    __shared__ int s_rid[BR];    // bank row of each slot, -1 = past the end of the list
    __shared__ int s_item;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int64_t D = a.D;
    const int64_t KT = (D + BK - 1) / BK;
    const int total = a.item_off[256];

    for (;;) {
        if (tid == 0) s_item = atomicAdd(a.work_counter, 1);
        __syncthreads();
        const int item = s_item;
        if (item >= total) break;
        int lo = 0, hi = 256;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.item_off[mid] <= item) lo = mid; else hi = mid;
        }
        const int c = lo;
        const int len = a.list_len[c], nql = a.lq_cnt[c];
        const int nqt = (nql + BQ - 1) / BQ;
        const int local = item - a.item_off[c];
        const int g = local / nqt, qt = local - g * nqt;

        if (tid < BQ) s_q[tid] = (qt * BQ + tid < nql) ? a.lq_list[c * IVF_MAXQ + qt * BQ + tid] : -1;
        if (tid < BR) {
            const int pos = g * BR + tid;
            s_rid[tid] = pos < len ? a.list_rows[a.list_off[c] + pos] : -1;
        }
        __syncthreads();

        const float* src[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int f = tid + i * IVF_THREADS;
            src[i] = nullptr;
            if (f < NV) {
                const int r = f >> 3, col = (f & 7) * 4;
                if (r < BQ) {
                    const int pk = s_q[r];
                    if (pk >= 0) src[i] = a.queries + (int64_t)(pk >> 4) * D + col;
                } else {
                    const int rid = s_rid[r - BQ];
                    if (rid >= 0) src[i] = a.bank + (int64_t)rid * D + col;
                }
            }
        }
        f32x16 acc[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
        // one k-tile of loads in flight per workgroup; several workgroups per CU overlap each
        // other's waits (a second register set of loads in flight measured slower)
        float4 pre[NLD];
        auto gload = [&](int64_t k0) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int f = tid + i * IVF_THREADS;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (f < NV && src[i]) {
                    const int64_t k = k0 + (f & 7) * 4;
                    if (VEC4) {
                        if (k < D) v = *reinterpret_cast<const float4*>(src[i] + k0);
                    } else {
                        if (k + 0 < D) v.x = src[i][k0 + 0];
                        if (k + 1 < D) v.y = src[i][k0 + 1];
                        if (k + 2 < D) v.z = src[i][k0 + 2];
                        if (k + 3 < D) v.w = src[i][k0 + 3];
                    }
                }
                pre[i] = v;
            }
        };
        gload(0);
        for (int64_t kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int f = tid + i * IVF_THREADS;
                if (f < NV) *reinterpret_cast<float4*>(smem + (f >> 3) * LDS_STRIDE + (f & 7) * 4) = pre[i];
            }
            __syncthreads();
            if (kt + 1 < KT) gload((kt + 1) * BK);
            const float* qrow = Qs + li * LDS_STRIDE + 4 * lh;
            const float* brow = Bs + (wave * 32 + li) * LDS_STRIDE + 4 * lh;
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                const float4 bv = *reinterpret_cast<const float4*>(brow + kk * 8);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const float4 av = *reinterpret_cast<const float4*>(qrow + t * 32 * LDS_STRIDE + kk * 8);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[t], 0, 0, 0);
                }
            }
            __syncthreads();
        }
        // epilogue: lane = list slot (wave*32 + li) x 16 queries of the tile
        const int slot = wave * 32 + li;
        const int rid = s_rid[slot];
        if (rid >= 0) {
            const float inv_m = a.inv_norm[rid];
            const float4 m = *reinterpret_cast<const float4*>(a.meta + (int64_t)rid * 4);
            const float tw = 0.2f * expf(-(a.now - m.y) / 3600.0f);
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pk = s_q[t * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh];
                if (pk < 0) continue;
                const int q = pk >> 4, p = pk & 15;
                const float sim = acc[t][e] * a.inv_q[q] * inv_m;
                const float comb = (0.5f * sim + tw) * m.x;
                const int dst = a.qbase[q * 8 + p] + g * BR + slot;
                if (dst < a.cap) {
                    a.cand_scores[(int64_t)q * a.cap + dst] = comb;
                    a.cand_idx[(int64_t)q * a.cap + dst] = rid;
                }
            }
        }
        __syncthreads();   // s_q / s_rid / s_item are rewritten by the next item
    }
}

// ------------------------------------------------------------------------------------------
// k-means helpers (rebuild_centroids, hippocampal.py:345-377)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_norm2_kernel(const float* __restrict__ x,
                                                        float* __restrict__ out, int64_t n,
                                                        int64_t D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float s = 0.0f;
    for (int64_t i = lane; i < D; i += 64) s = fmaf(x[row * D + i], x[row * D + i], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[row] = s;
}

// ------------------------------------------------------------------------------------------
// Sample threshold.  The k-th largest of the per-group maxima of the sample (groups = 32-row
// tiles) is a valid lower bound of the global k-th best score: the k largest group maxima are k
// distinct rows scoring at least that much.  One workgroup per query:
//   1. rank-count the G group maxima in LDS -> thr[q];
//   2. append every sample entry that reaches thr to the query's candidate list and publish the
//      count (so the counters and thresholds need no memset).
// ------------------------------------------------------------------------------------------
constexpr int THR_MAX_GROUPS = 4096;

__global__ __launch_bounds__(256) void sample_threshold_kernel(
    const float* __restrict__ gmax, int64_t gmax_ld, int G, const float* __restrict__ dense,
    int64_t dense_ld, int64_t cols, int blk, int step, int64_t row_begin, int64_t row_end, int k,
    uint32_t* __restrict__ thr_out, int32_t* __restrict__ cnt_out, float* __restrict__ cand_scores,
    int32_t* __restrict__ cand_idx, int cap) {
    __shared__ uint32_t s_g[THR_MAX_GROUPS];
    __shared__ uint32_t s_thr;
    __shared__ int s_cnt;
    const int q = blockIdx.x, tid = threadIdx.x;
    for (int g = tid; g < G; g += 256) s_g[g] = ord_key(gmax[(int64_t)q * gmax_ld + g]);
    if (tid == 0) { s_thr = 0u; s_cnt = 0; }
    __syncthreads();
    if (G >= k) {
        for (int g = tid; g < G; g += 256) {
            const uint32_t mine = s_g[g];
            int rank = 0;   // number of groups ordered before this one (ties -> lower group first)
            for (int j = 0; j < G; ++j) {
                const uint32_t o = s_g[j];
                rank += (o > mine || (o == mine && j < g)) ? 1 : 0;
            }
            if (rank == k - 1) s_thr = mine;
        }
    }
    __syncthreads();
    const uint32_t thr = s_thr;
    const int blk_log2 = 31 - __clz(blk);
    for (int64_t j = tid; j < cols; j += 256) {
        const float sc = dense[(int64_t)q * dense_ld + j];
        const int jj = (int)j;   // cols <= 131072; blk is a power of two (128 or 256)
        const int64_t row = row_begin + (int64_t)(jj >> blk_log2) * blk * step + (jj & (blk - 1));
        if (sc > -INFINITY && row < row_end && ord_key(sc) >= thr) {
            const int p = atomicAdd(&s_cnt, 1);
            if (p < cap) {
                cand_scores[(int64_t)q * cap + p] = sc;
                cand_idx[(int64_t)q * cap + p] = (int32_t)row;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        thr_out[q] = thr;
        cnt_out[(int64_t)q * CNT_STRIDE] = s_cnt;
    }
}

// inverse L2 norms of the queries (one wave per query) + reset of the overflow flag
__global__ __launch_bounds__(256) void query_prep_kernel(const float* __restrict__ x,
                                                         float* __restrict__ inv, int64_t n,
                                                         int64_t D, int32_t* overflow) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (overflow && blockIdx.x == 0 && threadIdx.x == 0) *overflow = 0;
    if (row >= n) return;
    const float* p = x + row * D;
    float s = 0.0f;
    if ((D & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
        for (int64_t i = lane * 4; i < D; i += 256) {
            float4 v = *reinterpret_cast<const float4*>(p + i);
            s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
        }
    } else {
        for (int64_t i = lane; i < D; i += 64) s = fmaf(p[i], p[i], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) inv[row] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
}

// ------------------------------------------------------------------------------------------
// Exact top-k select: one 256-thread workgroup per (chunk, query).  64-bit composite keys
// (ord(score) << 32 | ~idx) make every key unique, so "k-th largest key" is exact and ties go
// to the lower index.  MSB-first 8-bit radix select over LDS-resident keys, then the k winners
// are either appended to a candidate list (unsorted) or bitonic-sorted and written out.
// ------------------------------------------------------------------------------------------
constexpr int SEL_THREADS = 256;
constexpr int SEL_LDS_KEYS = 8192;        // 64 KiB of keys: default chunk (2 workgroups per CU)
constexpr int SEL_LDS_KEYS_HARD = 16384;  // 128 KiB: the most one workgroup can hold
constexpr int SEL_MAX_K = 1024;

struct SelectArgs {
    // source element i of query q, chunk c:
    //   score = src_scores[q*src_qs + (j / src_inner)*src_outer + j % src_inner],  j = c*chunk + i
    //   idx   = src_idx ? src_idx[same] : dense mapping (row_begin + (j / blk)*blk*step + j % blk)
    const float* src_scores;
    const int32_t* src_idx;
    int64_t src_qs, src_inner, src_outer;
    const int32_t* src_cnt;  // per-query element count (clamped to n_max) or null -> n_max
    int64_t n_max;           // elements per query over all chunks
    int chunk;               // elements per chunk (grid.x chunks)
    int blk, step;           // dense mapping
    int64_t row_begin, row_end;
    int k;
    // destination
    int sorted;              // 1: sorted output to dst[q*dst_qs + i]; 0: append block at dst[q*dst_qs + dst_off + c*k + i]
    float* dst_scores;
    int32_t* dst_idx;
    int64_t dst_qs;
    int64_t dst_off;
    int32_t idx_base;
    uint32_t* thr_out;       // optional: atomicMax of the chunk's k-th score key
    int32_t* cnt_out;        // optional: atomicAdd k (append mode)
    int32_t* overflow;       // optional: set when src_cnt > n_max
};

__global__ __launch_bounds__(SEL_THREADS) void topk_select_kernel(const SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];  // [chunk <= SEL_LDS_KEYS]
    __shared__ unsigned int s_hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_remaining;
    __shared__ unsigned long long s_win[SEL_MAX_K];
    __shared__ int s_nwin;
    __shared__ unsigned int s_wsum[SEL_THREADS / 64];
    __shared__ int s_done;

    const int tid = threadIdx.x;
    const int q = blockIdx.y;
    const int c = blockIdx.x;
    int64_t total = a.n_max;
    if (a.src_cnt) {
        const int64_t cnt = a.src_cnt[(int64_t)q * CNT_STRIDE];
        if (cnt > a.n_max) {
            if (tid == 0 && a.overflow) *a.overflow = 1;
        } else {
            total = cnt;
        }
    }
    const int64_t j0 = (int64_t)c * a.chunk;
    int n = (int)((total - j0) < a.chunk ? (total - j0) : a.chunk);
    if (n < 0) n = 0;

    // load keys
    for (int i = tid; i < n; i += SEL_THREADS) {
        const int64_t j = j0 + i;
        const int64_t off = (int64_t)q * a.src_qs + (j / a.src_inner) * a.src_outer + j % a.src_inner;
        float s = a.src_scores[off];
        int64_t idx;
        if (a.src_idx) {
            idx = a.src_idx[off];
        } else {
            idx = a.row_begin + (j / a.blk) * (int64_t)a.blk * a.step + j % a.blk;
            if (idx >= a.row_end) s = -INFINITY;
        }
        // -inf marks "not a candidate" (masked by the centroid probe, out of range, padding).  Such
        // entries get the low word i (< 2^31, below every real row's 0xffffffff - idx) so that ALL
        // keys of a chunk stay distinct: with one shared key, a chunk holding fewer than k
        // candidates made "key >= k-th key" match more than k entries and which of them reached
        // the k output slots was a race (lost real candidates for k >~ 100 with centroid masks).
        if (idx < 0 || s == -INFINITY)
            s_keys[i] = ((unsigned long long)ord_key(-INFINITY) << 32) | (uint32_t)i;
        else
            s_keys[i] = ((unsigned long long)ord_key(s) << 32) | (0xffffffffu - (uint32_t)idx);
    }
    const int k = a.k < n ? a.k : n;  // winners available in this chunk
    if (tid == 0) { s_prefix = 0ull; s_remaining = k; s_nwin = 0; s_done = 0; }
    __syncthreads();

    if (k > 0 && k < n) {
        // radix select of the k-th largest key, 8 bits per pass, MSB first
        const int lane = tid & 63, wave = tid >> 6;
        for (int byte = 7; byte >= 0; --byte) {
            s_hist[tid] = 0;
            __syncthreads();
            const unsigned long long prefix = s_prefix;
            const int rem = s_remaining;
            const int shift = byte * 8;
            for (int i0 = 0; i0 < n; i0 += SEL_THREADS) {
                const int i = i0 + tid;
                bool pending = false;
                unsigned int d = 0;
                if (i < n) {
                    const unsigned long long key = s_keys[i];
                    pending = (byte == 7) || ((key >> (shift + 8)) == (prefix >> (shift + 8)));
                    d = (unsigned int)(key >> shift) & 255u;
                }
                // wave-aggregate the common case of many lanes sharing a digit (top bytes of
                // clustered scores), then fall back to one LDS atomic per remaining lane
#pragma unroll 1
                for (int it = 0; it < 3; ++it) {
                    const unsigned long long active = __ballot(pending);
                    if (!active) break;
                    const int leader = __ffsll((long long)active) - 1;
                    const unsigned int ld = __shfl(d, leader);
                    const unsigned long long same = __ballot(pending && d == ld);
                    if (lane == leader) atomicAdd(&s_hist[ld], (unsigned int)__popcll(same));
                    if (d == ld) pending = false;
                }
                if (pending) atomicAdd(&s_hist[d], 1u);
            }
            __syncthreads();
            // suffix sums over the 256 bins: thread t owns bin t
            const unsigned int h = s_hist[tid];
            unsigned int x = h;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned int y = __shfl_down(x, off);
                if (lane + off < 64) x += y;
            }
            if (lane == 0) s_wsum[wave] = x;
            __syncthreads();
            for (int w = wave + 1; w < SEL_THREADS / 64; ++w) x += s_wsum[w];
            // x = #keys (matching the prefix) with digit >= tid
            if ((int)x >= rem && (int)(x - h) < rem) {
                s_prefix = prefix | ((unsigned long long)tid << shift);
                s_remaining = rem - (int)(x - h);
                // after the last score byte: if every key sharing this score is needed, the index
                // bytes cannot change the answer -> take all keys >= (score << 32)
                s_done = (byte == 4 && (int)h == rem - (int)(x - h)) ? 1 : 0;
            }
            __syncthreads();
            if (s_done) break;
        }
    }
    const unsigned long long kth = (k > 0 && k < n) ? s_prefix : 0ull;
    // collect winners (key >= kth): exactly k of them
    for (int i = tid; i < n; i += SEL_THREADS) {
        const unsigned long long key = s_keys[i];
        if (k > 0 && key >= kth) {
            const int p = atomicAdd(&s_nwin, 1);
            if (p < SEL_MAX_K) s_win[p] = key;
        }
    }
    __syncthreads();

    if (!a.sorted) {
        // append block (unsorted); pad with (-inf, -1) so every chunk contributes exactly a.k
        for (int i = tid; i < a.k; i += SEL_THREADS) {
            float s = -INFINITY;
            int32_t idx = -1;
            if (i < k) {
                const unsigned long long key = s_win[i];
                if ((uint32_t)key >= 0x80000000u) {          // a real row (see the key construction)
                    idx = (int32_t)(0xffffffffu - (uint32_t)key);
                    s = ord_unkey((uint32_t)(key >> 32));
                }
            }
            const int64_t o = (int64_t)q * a.dst_qs + a.dst_off + (int64_t)c * a.k + i;
            a.dst_scores[o] = s;
            a.dst_idx[o] = idx;
        }
        if (tid == 0) {
            if (a.cnt_out) atomicAdd(a.cnt_out + (int64_t)q * CNT_STRIDE, a.k);
            if (a.thr_out && k == a.k && k > 0) atomicMax(a.thr_out + q, (uint32_t)(kth >> 32));
        }
        return;
    }

    // sorted output: bitonic sort (descending) of the winners padded to a power of two
    int P = 1;
    while (P < a.k) P <<= 1;
    for (int i = k + tid; i < P; i += SEL_THREADS) s_win[i] = 0ull;
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < P / 2; t += SEL_THREADS) {
                const int lo = (t / stride) * stride * 2 + (t % stride);
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long x = s_win[lo], y = s_win[hi];
                if (desc ? (x < y) : (x > y)) { s_win[lo] = y; s_win[hi] = x; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < a.k; i += SEL_THREADS) {
        float s = -INFINITY;
        int32_t idx = -1;
        if (i < k) {
            const unsigned long long key = s_win[i];
            if ((uint32_t)key >= 0x80000000u) {              // else: padding / masked / out-of-range column
                idx = (int32_t)(0xffffffffu - (uint32_t)key) + a.idx_base;
                s = ord_unkey((uint32_t)(key >> 32));
            }
        }
        a.dst_scores[(int64_t)q * a.dst_qs + i] = s;
        a.dst_idx[(int64_t)q * a.dst_qs + i] = idx;
    }
}

__global__ void fill_u32_kernel(uint32_t* p, uint32_t v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------
// Host-side search driver
// ------------------------------------------------------------------------------------------
constexpr int QPASS_MAX = 2048;         // most queries handled by one pass of the pipeline
constexpr int64_t DENSE_COLS = 131072;  // dense score columns kept per query (128 MiB / 256 q)
constexpr int SEL_CHUNK = 4096;         // dense columns per select workgroup
constexpr int CAND_CAP_MIN = 8192;      // filter-path candidate slots per query

struct Workspace {
    float* inv_q;        // [qp]            (qp = queries per pass)
    float* eq;           // [qp rounded up to 256] query parts of the two-stage error bound
    uint32_t* thr;       // [qp]
    int32_t* cnt;        // [qp][CNT_STRIDE]
    uint32_t* probe;     // [qp][8]
    float* probe_dist;   // [qp][256]
    float* cand_scores;  // [qp][cap]
    int32_t* cand_idx;   // [qb][cap]
    float* cand2_scores; // [qb][cap2]  (reduce ping-pong)
    int32_t* cand2_idx;
    float* dense;        // [qb][dense cols]
    float* gmax;         // [qb][THR_MAX_GROUPS]
    uint16_t* qhat;      // [qb rounded up to 256][768] bf16 query fragments (two-stage path)
    float4* rowc;        // [N] per-row score constants (two-stage path)
    int32_t* heavy;      // [qp] refine: queries left to the workgroup-per-query kernel
    int cap, cap2;
    int qp;              // queries per pass
    int64_t bytes;
};

inline int cand_cap(int64_t N, int k) {
    // the dense path appends k winners per SEL_CHUNK columns; the filter path expects ~1024
    const int64_t need = ((N + SEL_CHUNK - 1) / SEL_CHUNK + 1) * (int64_t)k;
    const int64_t cap = need > CAND_CAP_MIN ? need : CAND_CAP_MIN;
    return (int)align_up(cap, 64);
}

inline Workspace carve(void* base, int64_t N, int64_t nq, int k) {
    Workspace w;
    char* p = static_cast<char*>(base);
    int64_t off = 0;
    // queries per pass: as many as fit a 1 GiB dense-score buffer, at most QPASS_MAX
    const int64_t cols = N < DENSE_COLS ? align_up(N > 0 ? N : 1, 1024) : DENSE_COLS;
    int64_t qp = nq < QPASS_MAX ? (nq > 0 ? nq : 1) : QPASS_MAX;
    while (qp > 256 && qp * cols * 4 > (int64_t(1) << 30)) qp = (qp / 2 + 255) / 256 * 256;
    const int qb = (int)qp;
    w.qp = qb;
    auto take = [&](int64_t bytes) {
        char* r = p ? p + off : nullptr;
        off += align_up(bytes, 256);
        return r;
    };
    w.cap = cand_cap(N, k);
    w.cap2 = (int)align_up(((int64_t)w.cap + SEL_LDS_KEYS - 1) / SEL_LDS_KEYS * k, 64);
    w.inv_q = reinterpret_cast<float*>(take((int64_t)qb * 4));
    w.eq = reinterpret_cast<float*>(take(((int64_t)qb + 255) / 256 * 256 * 4));
    w.thr = reinterpret_cast<uint32_t*>(take((int64_t)qb * 4));
    w.cnt = reinterpret_cast<int32_t*>(take((int64_t)qb * CNT_STRIDE * 4));
    w.probe = reinterpret_cast<uint32_t*>(take((int64_t)qb * 32));
    w.probe_dist = reinterpret_cast<float*>(take((int64_t)qb * 256 * 4));
    w.cand_scores = reinterpret_cast<float*>(take((int64_t)qb * w.cap * 4));
    w.cand_idx = reinterpret_cast<int32_t*>(take((int64_t)qb * w.cap * 4));
    w.cand2_scores = reinterpret_cast<float*>(take((int64_t)qb * w.cap2 * 4));
    w.cand2_idx = reinterpret_cast<int32_t*>(take((int64_t)qb * w.cap2 * 4));
    w.dense = reinterpret_cast<float*>(take((int64_t)qb * cols * 4));
    w.gmax = reinterpret_cast<float*>(take((int64_t)qb * THR_MAX_GROUPS * 4));
    w.qhat = reinterpret_cast<uint16_t*>(take(((int64_t)qb + 255) / 256 * 256 * 768 * 2));
    w.rowc = reinterpret_cast<float4*>(take((N > 0 ? N : 1) * 16));
    w.heavy = reinterpret_cast<int32_t*>(take(((int64_t)qb + 1) * 4));
    w.bytes = off;
    return w;
}

inline int launch_select(const SelectArgs& a, int64_t nchunks, int nq, hipStream_t s) {
    // the final select keeps 8192 64-bit keys (64 KiB) + 9 KiB static in LDS
    if (ensure_lds_attr(reinterpret_cast<const void*>(topk_select_kernel), SEL_LDS_KEYS_HARD * 8)) return AURA_E_LAUNCH;
    if (a.chunk > SEL_LDS_KEYS_HARD || a.k > SEL_MAX_K) return AURA_E_INVAL;
    hipLaunchKernelGGL(topk_select_kernel, dim3((unsigned)nchunks, (unsigned)nq), dim3(SEL_THREADS),
                       (size_t)a.chunk * 8, s, a);
    return check_launch();
}

// Optional per-launch timing of the main (FILTER) scan with HIP events on the launch stream:
// bench.py turns it on to measure the dominant kernel's duration live (aura_profile_*).
struct ProfileState {
    hipEvent_t* start = nullptr;
    hipEvent_t* stop = nullptr;
    int cap = 0, used = 0;
    bool on = false;
    int64_t rows = 0, nq = 0;   // rows x queries scored by the last profiled main-scan launch
    int kind = 0;               // 0 = fp32 MFMA scan, 1 = bf16 prefilter scan
};
// AURA_CS_DBG: ablation / timing switches of the scan kernels (results are invalid when non-zero); read from the
// environment once, replaceable through aura_debug_cs_flags (A/B runs inside one process: same allocations, same
// clocks -- successive processes on one box differ by +-5 %, more than most of the effects looked for)
int g_cs_dbg = -1;
inline int cs_dbg_flags() {
    if (g_cs_dbg < 0) g_cs_dbg = getenv("AURA_CS_DBG") ? (atoi(getenv("AURA_CS_DBG")) & 0x7fffffff) : 0;
    return g_cs_dbg;
}
ProfileState g_prof;            // one measurement session per PROCESS (bench.py / tests; aura_profile_begin documents it):
                                // events belong to the device that was current at aura_profile_begin

template <int WQ, int WR, int RT>
int launch_scan(const ScanArgs& a_in, int mode, int64_t ntiles_grid, hipStream_t s) {
    ScanArgs a = a_in;
    static const int dbg = getenv("AURA_SCAN_DBG") ? atoi(getenv("AURA_SCAN_DBG")) : 0;
    a.dbg = dbg;
    constexpr int BQ = WQ * 32, BR = WR * RT * 32;
    const size_t lds = (size_t)(BQ + BR) * LDS_STRIDE * sizeof(float);
    const dim3 grid((unsigned)ntiles_grid, (unsigned)((a.nq + BQ - 1) / BQ));
    const bool vec4 = (a.D % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.bank) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(a.queries) & 15) == 0);
#define AURA_SCAN(MODE, V4)                                                                       \
    hipLaunchKernelGGL((knn_scan_kernel<WQ, WR, RT, MODE, V4>), grid, dim3(SCAN_THREADS), lds, s, a)
    if (mode == MODE_DENSE) {
        if (vec4) AURA_SCAN(MODE_DENSE, true); else AURA_SCAN(MODE_DENSE, false);
    } else if (mode == MODE_FILTER) {
        if (vec4) AURA_SCAN(MODE_FILTER, true); else AURA_SCAN(MODE_FILTER, false);
    } else {
        if (WQ != 8) return AURA_E_INVAL;  // assign always uses the 256-"query" geometry
        if (vec4) AURA_SCAN(MODE_ASSIGN, true); else AURA_SCAN(MODE_ASSIGN, false);
    }
#undef AURA_SCAN
    return check_launch();
}

inline int device_cu_count() {
    // per DEVICE (a process that drives several GPUs gets each one's own count)
    static std::mutex mu;
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> g(mu);
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus[dev] = prop.multiProcessorCount;
        if (cus[dev] <= 0) cus[dev] = 256;
    }
    return cus[dev];
}

// persistent FILTER scan for a 256-query block (a.n_items non-sample 32-row tiles)
inline int launch_filter_v2(const ScanArgs& a, hipStream_t s) {
    const size_t lds = (size_t)2 * (256 + 128) * LDS_STRIDE * sizeof(float);
    {
        const void* fns[4] = {reinterpret_cast<const void*>(knn_scan_filter_v2<true, true>),
                              reinterpret_cast<const void*>(knn_scan_filter_v2<true, false>),
                              reinterpret_cast<const void*>(knn_scan_filter_v2<false, true>),
                              reinterpret_cast<const void*>(knn_scan_filter_v2<false, false>)};
        for (const void* f : fns)
            if (ensure_lds_attr(f, (int)lds)) return AURA_E_LAUNCH;
    }
    if (a.n_items <= 0) return AURA_OK;
    int64_t grid = device_cu_count();
    const int64_t chunks = ((a.n_items + 3) / 4) * ((a.nq + 255) / 256);
    if (grid > chunks) grid = chunks;
    const bool vec4 = (a.D % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.bank) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(a.queries) & 15) == 0);
    const bool fast = !a.q_loc && !a.probe_mask;
    const dim3 g((unsigned)grid), b(SCAN_THREADS);
    if (vec4 && fast) hipLaunchKernelGGL((knn_scan_filter_v2<true, true>), g, b, lds, s, a);
    else if (vec4) hipLaunchKernelGGL((knn_scan_filter_v2<true, false>), g, b, lds, s, a);
    else if (fast) hipLaunchKernelGGL((knn_scan_filter_v2<false, true>), g, b, lds, s, a);
    else hipLaunchKernelGGL((knn_scan_filter_v2<false, false>), g, b, lds, s, a);
    return check_launch();
}

// bank rows per workgroup tile for a query block of nq
inline int tile_rows_for(int nq) { return nq <= 64 ? 256 : 128; }
inline int dispatch_scan(const ScanArgs& a, int mode, int64_t ntiles_grid, hipStream_t s) {
    if (a.nq <= 32) return launch_scan<1, 8, 1>(a, mode, ntiles_grid, s);   // 32 q x 256 rows
    if (a.nq <= 64) return launch_scan<2, 4, 2>(a, mode, ntiles_grid, s);   // 64 q x 256 rows
    if (a.nq <= 128) return launch_scan<4, 2, 2>(a, mode, ntiles_grid, s);  // 128 q x 128 rows
    return launch_scan<8, 1, 4>(a, mode, ntiles_grid, s);                   // 256 q x 128 rows
}

#include "aura_knn_coarse.inl"
#include "aura_knn_ivf2.inl"

// coarse_refine_kernel in the geometry that fits the pass (see the comment at the kernel)
inline int launch_refine_for(const RefineArgs& r_in, int nqb, int64_t D, int cus, hipStream_t s, int32_t* heavy) {
    const int64_t Dpad = (D + 31) / 32 * 32;
    RefineArgs r = r_in;
    // Passes of many queries: one WAVE per query first (coarse_refine_wave_kernel); the workgroup-per-query kernel
    // then only works on the queries that one marked heavy.  AURA_RF_WAVE=0 / 1: never / always (A/B runs).
    static const int wave_mode = getenv("AURA_RF_WAVE") ? atoi(getenv("AURA_RF_WAVE")) : -1;
    // (At 8 queries per CU the workgroup-per-query kernel is the faster one: 125 us against 136 + list walk for the
    //  2048-query headline; beyond, its two queries per CU are the limit: 16384 queries 0.9 ms against 0.7.)
    // Beyond 8 queries per CU the one-wave form pays where most queries keep at most 512 candidates -- the shards of a
    // row-sharded bank (125 000 rows x 16 384 queries: 1.59 against 1.75 ms per recall); on a 1 M-row bank a query
    // keeps ~560, half of them overflow to the list and both kernels run (16 384 queries: 3.9 against 3.1 ms).
    const bool use_wave = heavy && !r.dbg_out && (wave_mode == 1 || (wave_mode != 0 && nqb > 8 * cus && r.N <= 300000));
    if (use_wave) {
        const bool big = nqb <= 8 * cus;                      // every query gets a wave at one workgroup per CU
        const int rows_w = big ? 16 : 6;
        const size_t region = (size_t)Dpad * 4 + (size_t)rows_w * (RW_KC + 4) * 4 + (size_t)RW_SURV * 12;
        const size_t lds_w = 8 * region;
        if (lds_w <= 150 * 1024) {
            r.heavy = heavy;
            if (hipMemsetAsync(heavy, 0, 4, s) != hipSuccess) return AURA_E_LAUNCH;
            if (big) {
                if (ensure_lds_attr(reinterpret_cast<const void*>(coarse_refine_wave16_kernel), 150 * 1024)) return AURA_E_LAUNCH;
                hipLaunchKernelGGL(coarse_refine_wave16_kernel, dim3((unsigned)((nqb + 7) / 8)), dim3(RF_THREADS), lds_w, s, r, nqb);
            } else {
                if (ensure_lds_attr(reinterpret_cast<const void*>(coarse_refine_wave6_kernel), 150 * 1024)) return AURA_E_LAUNCH;
                hipLaunchKernelGGL(coarse_refine_wave6_kernel, dim3((unsigned)((nqb + 7) / 8)), dim3(RF_THREADS), lds_w, s, r, nqb);
            }
            if (check_launch()) return AURA_E_LAUNCH;
        }
    }
    const int grid_q = r.heavy ? (nqb < 2 * cus ? nqb : 2 * cus) : nqb;   // heavy list: a small grid walks it
    auto launch_refine = [&](auto rows_tag, auto kc_tag) -> int {
        constexpr int ROWS = decltype(rows_tag)::value, KC = decltype(kc_tag)::value;
        size_t lds = (size_t)8 * ROWS * (KC + 4) * 4 + (size_t)Dpad * 4;
        if (lds < (size_t)RF_CAP * 12) lds = (size_t)RF_CAP * 12;
        if (r.heavy) {
            if (ensure_lds_attr(reinterpret_cast<const void*>(coarse_refine_list_kernel<ROWS, KC>),
                                8 * ROWS * (KC + 4) * 4 + 768 * 4))
                return AURA_E_LAUNCH;
            hipLaunchKernelGGL((coarse_refine_list_kernel<ROWS, KC>), dim3((unsigned)grid_q), dim3(RF_THREADS), lds, s, r);
            return check_launch();
        }
        if (ensure_lds_attr(reinterpret_cast<const void*>(coarse_refine_kernel<ROWS, KC>),
                            8 * ROWS * (KC + 4) * 4 + 768 * 4))
            return AURA_E_LAUNCH;
        hipLaunchKernelGGL((coarse_refine_kernel<ROWS, KC>), dim3((unsigned)grid_q), dim3(RF_THREADS), lds, s, r);
        return check_launch();
    };
    static const int geom = getenv("AURA_RF_GEOM") ? atoi(getenv("AURA_RF_GEOM")) : 0;   // A/B runs
    if (geom == 1) return launch_refine(std::integral_constant<int, 8>{}, std::integral_constant<int, 128>{});
    if (geom == 2) return launch_refine(std::integral_constant<int, 6>{}, std::integral_constant<int, 128>{});
    if (geom == 3) return launch_refine(std::integral_constant<int, 16>{}, std::integral_constant<int, 256>{});
    if (nqb > cus) return launch_refine(std::integral_constant<int, 10>{}, std::integral_constant<int, 192>{});
    return launch_refine(std::integral_constant<int, 16>{}, std::integral_constant<int, 256>{});
}

// AURA_CS_DBG bit 64 (builds with -DAURA_CS_TIMERS=1 only): per-wave phase times of a filter launch (written by
// the kernel into `buf`)
inline void print_cs_phases(hipStream_t s, const float* buf, int cus) {
    (void)hipStreamSynchronize(s);
    std::vector<float> h((size_t)cus * 64);
    (void)hipMemcpy(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost);
    static const char* const names[6] = {"DMA wait", "check+issue", "segment set-up", "mfma loop", "barrier", "write-out+epilogue"};
    for (int role = 0; role < 2; ++role) {
        double sum[6] = {0, 0, 0, 0, 0, 0}, tiles = 0; int waves = 0;
        for (int g = 0; g < cus; ++g)
            for (int wv = role * 4; wv < role * 4 + 4; ++wv) {
                const float* o = &h[((size_t)g * 8 + wv) * 8];
                if (o[6] <= 0) continue;
                for (int i = 0; i < 6; ++i) sum[i] += o[i];
                tiles += o[6]; ++waves;
            }
        if (!waves) continue;
        fprintf(stderr, "[cs phases] waves %d-%d: %d waves, %.1f tiles each;", role * 4, role * 4 + 3, waves, tiles / waves);
        double tot = 0;
        for (int i = 0; i < 6; ++i) {
            fprintf(stderr, " %s %.2f us (%.0f ns/tile);", names[i], sum[i] / waves * 0.01, sum[i] / tiles * 10.0);
            tot += sum[i];
        }
        fprintf(stderr, " total %.2f us\n", tot / waves * 0.01);
    }
    // per-wave busy / barrier times of a few workgroups: is the barrier skew systematic (one wave always last)?
    for (int g = 0; g < cus && g < 4; ++g) {
        fprintf(stderr, "[cs phases] wg %d [dma issue setup mma barrier epi] us per wave:", g);
        for (int wv = 0; wv < 8; ++wv) {
            const float* o = &h[((size_t)g * 8 + wv) * 8];
            fprintf(stderr, " [%.0f %.0f %.0f %.0f %.0f %.0f]", o[0] * 0.01, o[1] * 0.01, o[2] * 0.01, o[3] * 0.01, o[4] * 0.01, o[5] * 0.01);
        }
        fprintf(stderr, "\n");
    }
    // whole-kernel time per workgroup (wave 0), tile-loop time, tiles: spread over the workgroups
    double kmin = 1e30, kmax = 0, ksum = 0, lmin = 1e30, lmax = 0, lsum = 0; int n = 0;
    for (int g = 0; g < cus; ++g) {
        const float* o = &h[(size_t)g * 64];
        if (o[6] <= 0) continue;
        double l = 0; for (int i = 0; i < 6; ++i) l += o[i];
        const double kt = o[7];
        kmin = kt < kmin ? kt : kmin; kmax = kt > kmax ? kt : kmax; ksum += kt;
        lmin = l < lmin ? l : lmin; lmax = l > lmax ? l : lmax; lsum += l; ++n;
    }
    if (n) fprintf(stderr, "[cs phases] per workgroup: kernel time min %.1f mean %.1f max %.1f us; tile loops min %.1f mean %.1f max %.1f us\n",
                   kmin * 0.01, ksum / n * 0.01, kmax * 0.01, lmin * 0.01, lsum / n * 0.01, lmax * 0.01);
}

// AURA_CS_DBG bit 128: the refine kernel's phase times and candidate / survivor counts per query
inline void print_refine_phases(hipStream_t s, const float* buf, int nqb, const char* path) {
    (void)hipStreamSynchronize(s);
    std::vector<float> h((size_t)nqb * 8);
    (void)hipMemcpy(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost);
    static const char* const names[6] = {"load candidates", "select T2", "survivors", "load query", "re-score", "rank+write"};
    double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < nqb; ++q)
        for (int i = 0; i < 8; ++i) { sum[i] += h[(size_t)q * 8 + i]; if (h[(size_t)q * 8 + i] > mx[i]) mx[i] = h[(size_t)q * 8 + i]; }
    fprintf(stderr, "[refine phases, %s] %d queries: candidates mean %.0f max %.0f, survivors mean %.0f max %.0f;", path, nqb,
            sum[6] / nqb, mx[6], sum[7] / nqb, mx[7]);
    for (int i = 0; i < 6; ++i) fprintf(stderr, " %s %.2f us (max %.2f);", names[i], sum[i] / nqb * 0.01, mx[i] * 0.01);
    fprintf(stderr, "\n");
}

// Two-stage recall of one query pass (see aura_knn_coarse.inl).  Returns AURA_OK after queuing
// sample scan -> threshold -> filter scan -> refine; the caller skips the fp32 pipeline.
inline int run_coarse_pass(const float* bank, const uint16_t* bank16, const float* rho, const float* inv_norm,
                           const float* meta, const float* qptr, float now, int64_t N, int64_t D, int nqb, int k,
                           int32_t idx_base, float* out_scores, int32_t* out_idx,
                           const Workspace& w, int32_t* overflow_out, bool reset_flag,
                           const uint32_t* probe_mask, hipStream_t s) {
    int rc;
    // sample: strided 128-row logical tiles (8 coarse tiles each), sized as for the fp32 path
    int64_t sample_rows = (int64_t)k * N / 512;
    static const int64_t sample_cap = getenv("AURA_CS_SAMPLE_ROWS") ? atoll(getenv("AURA_CS_SAMPLE_ROWS")) : 8192;
    int64_t floor_rows = N / 8 < sample_cap ? N / 8 : sample_cap;
    if (floor_rows < (int64_t)k * 48) floor_rows = (int64_t)k * 48;
    if (sample_rows < floor_rows) sample_rows = floor_rows;
    const int64_t ntiles128 = N / 128;                       // whole logical tiles only
    int64_t n_sample = (sample_rows + 127) / 128;
    // a group is a 16-row tile while the sample fits THR_MAX_GROUPS of them (banks up to ~1 M rows at
    // k = 32); beyond, a whole 128-row logical tile -- the k-th largest group maximum certifies k
    // distinct rows either way, and the sample keeps growing with the bank (up to 512 K rows) instead of
    // stopping at 64 K, where a 10 M-row bank's queries collected more candidates than the refine stage holds
    const int gshift = n_sample * 8 > THR_MAX_GROUPS ? 3 : 0;
    if (gshift && n_sample > THR_MAX_GROUPS) n_sample = THR_MAX_GROUPS;
    if (n_sample > ntiles128) n_sample = ntiles128;
    const int tile_step = (int)(ntiles128 / n_sample);
    const int G = gshift ? (int)n_sample : (int)n_sample * 8;
    if (G < k) return AURA_E_INVAL;                          // ruled out by coarse_eligible()

    // per-call preparation: bf16 query fragments + 1/||q|| (+ overflow-flag reset), row constants
    const int KS = D <= 256 ? 8 : (D <= 512 ? 16 : 24);
    const int64_t nq_pad = ((int64_t)nqb + 255) / 256 * 256;
    // error bound of the bf16 scores (aura_knn_coarse.inl): E_fix for the accumulation, E_worst for
    // rows rounded on the fly; with the shadow the rows' and queries' own residual norms
    const float e_fix = aura_e_fix((float)D);
    const float e_cos = 0.0078125f * (1.0f + 0.001953125f) + e_fix;
    // the shadow is only usable when its rows are 16-byte aligned and its error norms are known
    const bool use16 = bank16 && rho && (D & 7) == 0 && (reinterpret_cast<uintptr_t>(bank16) & 15) == 0;
    const int qblocks = (int)(nq_pad / 4);
    hipLaunchKernelGGL(coarse_prep_kernel, dim3((unsigned)(qblocks + (N + 255) / 256)), dim3(256), 0, s,
                       qptr, (int64_t)nqb, nq_pad, D, KS, w.qhat, w.inv_q, w.eq,
                       reset_flag ? overflow_out : nullptr, qblocks, meta, inv_norm, use16 ? rho : nullptr, N, now,
                       e_fix, e_cos, w.rowc);
    if ((rc = check_launch())) return rc;
    CoarseArgs c{};
    c.bank = bank; c.rowc = w.rowc; c.qhat = w.qhat; c.inv_q = w.inv_q; c.eq = w.eq;
    c.bank16 = use16 ? bank16 : nullptr;
    const int cs_dbg = cs_dbg_flags();
    c.dbg = cs_dbg;
    c.N = N; c.D = D; c.nq = nqb;
    c.thr = w.thr; c.cnt = w.cnt; c.cand_scores = w.cand_scores; c.cand_idx = w.cand_idx;
    c.cap = w.cap; c.probe_mask = probe_mask;
    const int cus = device_cu_count();

    c.n_tiles = n_sample * 8; c.tile_step = tile_step; c.n_sample = (int)n_sample;
    c.gmax = w.gmax; c.gmax_ld = THR_MAX_GROUPS; c.gshift = gshift;
    if (gshift && hipMemsetAsync(w.gmax, 0, (size_t)nqb * THR_MAX_GROUPS * 4, s) != hipSuccess) return AURA_E_LAUNCH;
    {
        const int64_t items = (int64_t)n_sample * 8 * ((nqb + 255) / 256);
        if ((rc = dispatch_coarse(c, CS_MODE_SAMPLE, (int)(items < cus ? items : cus), s))) return rc;
    }
    {
        const dim3 tg((unsigned)((nqb + 3) / 4)), tb(256);
#define AURA_THR(PER)                                                                              \
    hipLaunchKernelGGL((coarse_threshold_kernel<PER>), tg, tb, 0, s, w.gmax, (int64_t)THR_MAX_GROUPS, \
                       G, k, nqb, gshift ? 1 : 0, w.thr, w.cnt)
        if (G <= 512) AURA_THR(8);
        else if (G <= 1024) AURA_THR(16);
        else if (G <= 2048) AURA_THR(32);
        else AURA_THR(64);
#undef AURA_THR
    }
    if ((rc = check_launch())) return rc;

    c.n_tiles = (N + CS_ROWS - 1) / CS_ROWS; c.gmax = nullptr; c.gshift = 0;
    static int tm_left = 3;                                  // AURA_CS_DBG bit 64: phase times of the first launches
    const bool tm = CS_TIMERS && (cs_dbg & 64) && tm_left > 0;
    if (tm) {
        c.gmax = w.gmax;
        (void)hipMemsetAsync(w.gmax, 0, (size_t)cus * 8 * 8 * 4, s);
    }
    const bool prof = g_prof.on && g_prof.used < g_prof.cap;
    if (prof) {
        (void)hipEventRecord(g_prof.start[g_prof.used], s);
        g_prof.rows = N; g_prof.nq = nqb; g_prof.kind = c.bank16 ? 2 : 1;
    }
    {
        const int64_t items = c.n_tiles * ((nqb + 255) / 256);
        if ((rc = dispatch_coarse(c, CS_MODE_FILTER, (int)(items < cus ? items : cus), s))) return rc;
    }
    if (prof) (void)hipEventRecord(g_prof.stop[g_prof.used++], s);
    if (tm) {
        --tm_left;
        print_cs_phases(s, w.gmax, cus);
    }

    RefineArgs r{};
    r.bank = bank; r.inv_norm = inv_norm; r.meta = meta; r.queries = qptr; r.inv_q = w.inv_q;
    r.now = now; r.e_cos = e_cos; r.N = N; r.D = D; r.k = k; r.cnt = w.cnt;
    r.rho = use16 ? rho : nullptr; r.eq = w.eq; r.e_fix = e_fix; r.eq_worst = coarse_eq_worst((float)D);
    r.cand_scores = w.cand_scores; r.cand_idx = w.cand_idx; r.cap = w.cap; r.idx_base = idx_base;
    r.out_scores = out_scores; r.out_idx = out_idx; r.overflow = overflow_out;
    static int rtm_left = 3;                                 // AURA_CS_DBG bit 128: refine phase times
    const bool rtm = CS_TIMERS && (cs_dbg & 128) && rtm_left > 0;
    if (rtm) r.dbg_out = w.gmax;
    if ((rc = launch_refine_for(r, nqb, D, cus, s, w.heavy))) return rc;
    if (rtm) {
        --rtm_left;
        print_refine_phases(s, w.gmax, nqb, "full scan");
    }
    return check_launch();
}

}  // namespace

extern "C" {

int aura_bank_row_norms(const float* bank, float* inv_norm, int64_t row0, int64_t n, int64_t D,
                        void* stream) {
    if (n < 0 || D <= 0 || row0 < 0) return AURA_E_INVAL;
    if (n == 0) return AURA_OK;
    if (!bank || !inv_norm) return AURA_E_INVAL;
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), bank + row0 * D, inv_norm + row0, n, D);
    return check_launch();
}

int aura_bank_write(float* bank, float* loc, float* meta, float* inv_norm, float* centroids,
                    float* centroid_counts, int eff_k, const float* feats, const int64_t* slots,
                    const float* cur_loc, int spatial_dims, float now, int64_t n, int64_t D,
                    void* stream) {
    if (n < 0 || D <= 0 || spatial_dims < 0 || spatial_dims > 64) return AURA_E_INVAL;
    if (n == 0) return AURA_OK;
    if (!bank || !loc || !meta || !inv_norm || !feats || !slots || !cur_loc) return AURA_E_INVAL;
    if (reinterpret_cast<uintptr_t>(meta) & 15) return AURA_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int vec4 = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(feats) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(bank) & 15) == 0) &&
                     (!centroids || (reinterpret_cast<uintptr_t>(centroids) & 15) == 0);
    if (centroids) {
        if (!centroid_counts || eff_k <= 0 || eff_k > 256) return AURA_E_INVAL;
        const size_t lds = (size_t)D * sizeof(float);
        if (lds > 150 * 1024) return AURA_E_INVAL;
        if (ensure_lds_attr(reinterpret_cast<const void*>(bank_write_centroid_kernel), 150 * 1024)) return AURA_E_LAUNCH;
        hipLaunchKernelGGL(bank_write_centroid_kernel, dim3(1), dim3(1024), lds, s, bank, loc, meta,
                           inv_norm, centroids, centroid_counts, eff_k, feats, slots, cur_loc,
                           spatial_dims, now, n, D, vec4);
    } else {
        hipLaunchKernelGGL(bank_write_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, bank,
                           loc, meta, inv_norm, feats, slots, cur_loc, spatial_dims, now, -1.0f, n, D, vec4);
    }
    return check_launch();
}

int64_t aura_bank_write_online_workspace_bytes(int64_t n) {
    if (n < 0) return -1;
    const int64_t ch = n < ONL_CHUNK ? (n > 0 ? n : 1) : ONL_CHUNK;
    return align_up(ch * 256 * 4, 256) + 3 * align_up(ch * 4, 256);
}

int aura_bank_write_online(float* bank, float* loc, float* meta, float* inv_norm, float* centroids,
                           float* centroid_counts, int eff_k, const float* feats, const int64_t* slots,
                           const float* cur_loc, int spatial_dims, float now, int64_t n, int64_t D,
                           void* workspace, int64_t workspace_bytes, void* stream) {
    if (n < 0 || D <= 0 || spatial_dims < 0 || spatial_dims > 64) return AURA_E_INVAL;
    if (n == 0) return AURA_OK;
    if (!bank || !loc || !meta || !inv_norm || !feats || !slots || !cur_loc || !centroids || !centroid_counts || !workspace)
        return AURA_E_INVAL;
    if (eff_k <= 0 || eff_k > 256) return AURA_E_INVAL;
    if ((reinterpret_cast<uintptr_t>(meta) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 255)) return AURA_E_ALIGN;
    if (workspace_bytes < aura_bank_write_online_workspace_bytes(n)) return AURA_E_INVAL;
    const size_t lds = (size_t)D * sizeof(float);
    if (lds > 140 * 1024) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int vec4 = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(feats) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(bank) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(centroids) & 15) == 0);
    if (ensure_lds_attr(reinterpret_cast<const void*>(online_assign_kernel), 140 * 1024)) return AURA_E_LAUNCH;
    hipLaunchKernelGGL(bank_write_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, bank, loc, meta, inv_norm,
                       feats, slots, cur_loc, spatial_dims, now, -1.0f, n, D, vec4);
    int rc;
    if ((rc = check_launch())) return rc;
    const int64_t ch = n < ONL_CHUNK ? n : ONL_CHUNK;
    float* const d0 = static_cast<float*>(workspace);
    float* const xnorm = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up(ch * 256 * 4, 256));
    int32_t* const pred = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(xnorm) + align_up(ch * 4, 256));
    int32_t* const cid = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(pred) + align_up(ch * 4, 256));
    static const bool simple = getenv("AURA_ONLINE_SIMPLE") != nullptr;   // A/B runs: the unpipelined phase B
    static const bool v1 = getenv("AURA_ONLINE_V1") != nullptr;         // A/B runs: one barrier per row (the pipelined form)
    const bool fast = D <= 1024 && !simple;
    if (fast && (ensure_lds_attr(reinterpret_cast<const void*>(online_assign_fast_kernel), 8 * 1024) ||
                 ensure_lds_attr(reinterpret_cast<const void*>(online_assign_fast2_kernel), 8 * 1024))) return AURA_E_LAUNCH;
    // fp slack of a computed distance against the exact one, both ways: 2 x (chain of D/64 fmaf + 6 butterfly
    // adds + subtraction, square root) x 2^-24, with a factor 2 in hand
    const float rel = 4.0f * ((float)((D + 63) / 64) + 10.0f) * 5.9604645e-8f;
    const int groups = (eff_k + ONL_CG - 1) / ONL_CG;
    for (int64_t r0 = 0; r0 < n; r0 += ch) {
        const int64_t nr = (n - r0) < ch ? (n - r0) : ch;
        hipLaunchKernelGGL(online_dist0_kernel, dim3((unsigned)((nr * groups + 3) / 4)), dim3(256), 0, s,
                           feats + r0 * D, centroids, eff_k, nr, D, vec4, d0, xnorm);
        if ((rc = check_launch())) return rc;
        if (fast) {
            hipLaunchKernelGGL(online_pred_kernel, dim3((unsigned)((nr + 3) / 4)), dim3(256), 0, s, d0, eff_k, nr, pred);
            if ((rc = check_launch())) return rc;
            if (v1) hipLaunchKernelGGL(online_assign_fast_kernel, dim3(1), dim3(1024), lds, s, centroids, centroid_counts,
                                       eff_k, feats + r0 * D, d0, xnorm, pred, cid, nr, D, vec4, rel);
            else hipLaunchKernelGGL(online_assign_fast2_kernel, dim3(1), dim3(1024), lds, s, centroids, centroid_counts,
                                    eff_k, feats + r0 * D, d0, xnorm, pred, cid, nr, D, vec4, rel);
            if ((rc = check_launch())) return rc;
            hipLaunchKernelGGL(online_cid_scatter_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, s, meta,
                               slots + r0, cid, nr);
        } else {
            hipLaunchKernelGGL(online_assign_kernel, dim3(1), dim3(1024), lds, s, meta, centroids, centroid_counts, eff_k,
                               feats + r0 * D, slots + r0, d0, xnorm, nr, D, vec4, rel);
        }
        if ((rc = check_launch())) return rc;
    }
    return AURA_OK;
}

int aura_bank_decay(float* meta, float rate, int64_t count, void* stream) {
    if (count < 0) return AURA_E_INVAL;
    if (count == 0) return AURA_OK;
    if (!meta) return AURA_E_INVAL;
    hipLaunchKernelGGL(bank_decay_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), meta, 1.0f - rate, count);
    return check_launch();
}

int aura_bank_gather(const float* bank, int64_t rows, const int32_t* idx, float* out, int64_t n, int64_t D,
                     void* stream) {
    if (n < 0 || D <= 0 || rows < 0) return AURA_E_INVAL;
    if (n == 0) return AURA_OK;
    if (!bank || !idx || !out) return AURA_E_INVAL;
    hipLaunchKernelGGL(bank_gather_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), bank, idx, out, n, D, rows);
    return check_launch();
}

int64_t aura_knn_workspace_bytes(int64_t N, int64_t nq, int k) {
    if (N < 0 || nq < 0 || k <= 0) return AURA_E_INVAL;
    return carve(nullptr, N, nq, k).bytes;
}

static int knn_search_impl(const float* bank, const uint16_t* bank_bf16, const float* rho, const float* inv_norm,
                           const float* meta, const float* loc,
                           int spatial_dims, const float* queries, const float* q_loc, float now,
                           int64_t N, int64_t D, int64_t nq, int k, int32_t idx_base, float* out_scores,
                           int32_t* out_idx, void* workspace, int64_t workspace_bytes, int flags,
                           int32_t* overflow_out, const float* centroids, int nprobe, void* stream) {
    if (N < 0 || D <= 0 || nq < 0 || k <= 0 || k > SEL_MAX_K || k > N) return AURA_E_INVAL;
    if (centroids && (nprobe <= 0 || nprobe > 256)) return AURA_E_INVAL;
    if (flags & ~(AURA_KNN_FORCE_DENSE | AURA_KNN_FP32_SCAN)) return AURA_E_INVAL;
    if (nq == 0) return AURA_OK;
    if (!bank || !inv_norm || !meta || !queries || !out_scores || !out_idx || !workspace)
        return AURA_E_INVAL;
    if (q_loc && (!loc || spatial_dims <= 0 || spatial_dims > 4)) return AURA_E_INVAL;
    if (N > 0x7ffffff0LL) return AURA_E_INVAL;
    if (reinterpret_cast<uintptr_t>(meta) & 15) return AURA_E_ALIGN;
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return AURA_E_ALIGN;
    const Workspace w = carve(workspace, N, nq, k);
    if (w.bytes > workspace_bytes) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc;

    for (int64_t qb0 = 0; qb0 < nq; qb0 += w.qp) {
        const int nqb = (int)((nq - qb0) < w.qp ? (nq - qb0) : w.qp);
        const float* qptr = queries + qb0 * D;
        const int br = tile_rows_for(nqb);
        const int64_t ntiles = (N + br - 1) / br;

        if (!coarse_eligible(bank, bank_bf16, qptr, q_loc, centroids, N, D, k, flags)) {
            hipLaunchKernelGGL(query_prep_kernel, dim3((unsigned)((nqb + 3) / 4)), dim3(256), 0, s,
                               qptr, w.inv_q, (int64_t)nqb, D, qb0 == 0 ? overflow_out : nullptr);
            if ((rc = check_launch())) return rc;
        }

        if (coarse_eligible(bank, bank_bf16, qptr, q_loc, centroids, N, D, k, flags)) {
            const uint32_t* pmask = nullptr;
            if (centroids) {
                if ((rc = launch_probe(centroids, qptr, D, nqb, nprobe, w.probe_dist, w.probe, nullptr, s)))
                    return rc;
                pmask = w.probe;
            }
            if ((rc = run_coarse_pass(bank, bank_bf16, rho, inv_norm, meta, qptr, now, N, D, nqb, k, idx_base,
                                      out_scores + qb0 * k, out_idx + qb0 * k, w, overflow_out,
                                      qb0 == 0, pmask, s)))
                return rc;
            continue;
        }

        ScanArgs a{};
        a.bank = bank; a.inv_norm = inv_norm; a.meta = meta; a.loc = loc;
        a.queries = qptr; a.inv_q = w.inv_q;
        a.q_loc = q_loc ? q_loc + qb0 * spatial_dims : nullptr;
        a.sdims = spatial_dims; a.now = now; a.D = D; a.nq = nqb;
        a.row_begin = 0; a.row_end = N;
        a.thr = w.thr; a.cnt = w.cnt; a.cand_scores = w.cand_scores; a.cand_idx = w.cand_idx;
        a.cap = w.cap;
        a.dense = w.dense;
        if (centroids) {
            if ((rc = launch_probe(centroids, qptr, D, nqb, nprobe, w.probe_dist, w.probe, nullptr, s)))
                return rc;
            a.probe_mask = w.probe;
        }

        SelectArgs sel{};   // dense columns -> k winners per SEL_CHUNK appended to the candidates
        sel.src_scores = w.dense; sel.src_idx = nullptr; sel.src_outer = 0; sel.src_cnt = nullptr;
        sel.chunk = SEL_CHUNK; sel.blk = br;
        sel.k = k; sel.sorted = 0;
        sel.dst_scores = w.cand_scores; sel.dst_idx = w.cand_idx; sel.dst_qs = w.cap;
        sel.cnt_out = w.cnt; sel.row_end = N;

        // filter path: expected candidates per query ~ k*N/sample_rows; aim for <= ~512
        //   floor: 8192 rows for large banks, N/8 for small ones (row shards of a multi-GPU bank),
        //   never fewer than 1.5 k groups of 32 rows (the bound is the k-th largest group maximum)
        int64_t sample_rows = (int64_t)k * N / 512;
        int64_t floor_rows = N / 8 < 8192 ? N / 8 : 8192;
        if (floor_rows < (int64_t)k * 48) floor_rows = (int64_t)k * 48;
        if (sample_rows < floor_rows) sample_rows = floor_rows;
        // (growing the sample so that every span loses a tile made the scan 7 % faster but pushed the
        //  sample pass past one workgroup per CU: net slower, 0.449 vs 0.433 ms per step)
        const int64_t n_sample_tiles = (sample_rows + br - 1) / br;
        const bool dense_all = (flags & AURA_KNN_FORCE_DENSE) || n_sample_tiles * 4 > ntiles ||
                               n_sample_tiles * br > DENSE_COLS ||
                               n_sample_tiles * br / 32 > THR_MAX_GROUPS || n_sample_tiles * br / 32 < k;
        int64_t cur_n;  // candidate slots in use (capacity view) after this stage
        const int32_t* cur_cnt;
        if (dense_all) {
            if (hipMemsetAsync(w.cnt, 0, (size_t)w.qp * CNT_STRIDE * 4, s) != hipSuccess) return AURA_E_LAUNCH;
            const int64_t tiles_per_super = DENSE_COLS / br;
            int64_t appended = 0;
            for (int64_t t0 = 0; t0 < ntiles; t0 += tiles_per_super) {
                const int64_t nt = (ntiles - t0) < tiles_per_super ? (ntiles - t0) : tiles_per_super;
                const int64_t cols = nt * br;
                a.row_begin = t0 * br; a.tile_step = 1; a.n_sample_tiles = 0; a.dense_ld = cols;
                a.logical_rows = br;
                if ((rc = dispatch_scan(a, MODE_DENSE, nt, s))) return rc;
                const int64_t nchunks = (cols + SEL_CHUNK - 1) / SEL_CHUNK;
                sel.src_qs = cols; sel.src_inner = cols; sel.n_max = cols; sel.step = 1;
                sel.row_begin = a.row_begin; sel.dst_off = appended * k; sel.thr_out = nullptr;
                if ((rc = launch_select(sel, nchunks, nqb, s))) return rc;
                appended += nchunks;
            }
            cur_n = appended * k;
            cur_cnt = nullptr;  // every chunk appended exactly k (padded) entries
        } else {
            // 1) strided sample of tiles scored densely -> exact k-th best of the sample per
            //    query = a valid lower bound of the global k-th best
            const int tile_step = (int)(ntiles / n_sample_tiles);
            const int64_t cols = n_sample_tiles * br;
            a.tile_step = tile_step; a.n_sample_tiles = (int)n_sample_tiles; a.dense_ld = cols;
            a.logical_rows = br;
            a.gmax = w.gmax; a.gmax_ld = THR_MAX_GROUPS;
            if (nqb > 128) {
                // 32-row kernel tiles: 4x the workgroups of the 128-row geometry, so the short
                // sample pass covers the whole chip instead of a few CUs
                if ((rc = launch_scan<8, 1, 1>(a, MODE_DENSE, n_sample_tiles * 4, s))) return rc;
            } else {
                if ((rc = dispatch_scan(a, MODE_DENSE, n_sample_tiles, s))) return rc;
            }
            a.gmax = nullptr;
            // threshold = k-th largest group maximum; sample entries that reach it seed the lists
            hipLaunchKernelGGL(sample_threshold_kernel, dim3((unsigned)nqb), dim3(256), 0, s, w.gmax,
                               (int64_t)THR_MAX_GROUPS, (int)(cols / 32), w.dense, cols, cols, br,
                               tile_step, (int64_t)0, N, k, w.thr, w.cnt, w.cand_scores, w.cand_idx,
                               w.cap);
            if ((rc = check_launch())) return rc;
            // 2) main scan appends the rows that reach the bound (sample tiles are skipped)
            const bool prof = g_prof.on && g_prof.used < g_prof.cap;
            if (prof) {
                (void)hipEventRecord(g_prof.start[g_prof.used], s);
                g_prof.rows = N - n_sample_tiles * br;
                g_prof.nq = nqb;   // all query blocks of the pass are scored by this one launch
                g_prof.kind = 0;
            }
            static const bool force_v1 = getenv("AURA_SCAN_V1") != nullptr;
            if (nqb > 128 && !force_v1) {
                a.n_items = (N + 31) / 32 - 4 * n_sample_tiles;
                if ((rc = launch_filter_v2(a, s))) return rc;
            } else {
                if ((rc = dispatch_scan(a, MODE_FILTER, ntiles, s))) return rc;
            }
            if (prof) (void)hipEventRecord(g_prof.stop[g_prof.used++], s);
            cur_n = w.cap;
            cur_cnt = w.cnt;
        }

        // reduce the candidate lists until they fit one LDS select, then sort the winners
        const float* cs = w.cand_scores; const int32_t* ci = w.cand_idx; int64_t cqs = w.cap;
        float* os = w.cand2_scores; int32_t* oi = w.cand2_idx; int64_t oqs = w.cap2;
        bool first = true;
        while (cur_n > SEL_LDS_KEYS) {
            const int64_t nch = (cur_n + SEL_LDS_KEYS - 1) / SEL_LDS_KEYS;
            SelectArgs r{};
            r.src_scores = cs; r.src_idx = ci; r.src_qs = cqs; r.src_inner = cqs; r.src_outer = 0;
            r.src_cnt = cur_cnt; r.n_max = cur_n; r.chunk = SEL_LDS_KEYS; r.blk = 1; r.step = 1;
            r.row_begin = 0; r.row_end = N; r.k = k; r.sorted = 0;
            r.dst_scores = os; r.dst_idx = oi; r.dst_qs = oqs; r.dst_off = 0;
            r.overflow = first ? overflow_out : nullptr;
            if (nch * k > oqs) return AURA_E_INVAL;
            if ((rc = launch_select(r, nch, nqb, s))) return rc;
            // ping-pong
            const float* ts = cs; const int32_t* ti = ci; const int64_t tq = cqs;
            cs = os; ci = oi; cqs = oqs;
            os = const_cast<float*>(ts); oi = const_cast<int32_t*>(ti); oqs = tq;
            cur_n = nch * k; cur_cnt = nullptr; first = false;
        }
        SelectArgs fin{};
        fin.src_scores = cs; fin.src_idx = ci; fin.src_qs = cqs; fin.src_inner = cqs; fin.src_outer = 0;
        fin.src_cnt = cur_cnt; fin.n_max = cur_n; fin.chunk = (int)cur_n; fin.blk = 1; fin.step = 1;
        fin.row_begin = 0; fin.row_end = N; fin.k = k; fin.sorted = 1;
        fin.dst_scores = out_scores + qb0 * k; fin.dst_idx = out_idx + qb0 * k; fin.dst_qs = k;
        fin.idx_base = idx_base; fin.overflow = first ? overflow_out : nullptr;
        if ((rc = launch_select(fin, 1, nqb, s))) return rc;
    }
    return AURA_OK;
}

int aura_knn_search_ex(const float* bank, const float* inv_norm, const float* meta, const float* loc,
                       int spatial_dims, const float* queries, const float* q_loc, float now,
                       int64_t N, int64_t D, int64_t nq, int k, int32_t idx_base, float* out_scores,
                       int32_t* out_idx, void* workspace, int64_t workspace_bytes, int flags,
                       int32_t* overflow_out, const float* centroids, int nprobe, void* stream) {
    return knn_search_impl(bank, nullptr, nullptr, inv_norm, meta, loc, spatial_dims, queries, q_loc, now, N, D, nq,
                           k, idx_base, out_scores, out_idx, workspace, workspace_bytes, flags,
                           overflow_out, centroids, nprobe, stream);
}

int aura_knn_search_shadow(const float* bank, const uint16_t* bank_bf16, const float* rho, const float* inv_norm,
                           const float* meta, const float* queries, float now, int64_t N, int64_t D,
                           int64_t nq, int k, int32_t idx_base, float* out_scores, int32_t* out_idx,
                           void* workspace, int64_t workspace_bytes, int flags, int32_t* overflow_out,
                           const float* centroids, int nprobe, void* stream) {
    if (bank_bf16 && !rho) return AURA_E_INVAL;
    return knn_search_impl(bank, bank_bf16, rho, inv_norm, meta, nullptr, 0, queries, nullptr, now, N, D, nq, k,
                           idx_base, out_scores, out_idx, workspace, workspace_bytes, flags, overflow_out,
                           centroids, nprobe, stream);
}

int64_t aura_centroid_probe_workspace_bytes(int64_t nq) {
    if (nq < 0) return -1;
    return align_up(nq * 256 * 4, 256) + align_up(nq * 32, 256) + 256;
}

int64_t aura_knn_ivf2_workspace_bytes(int64_t n_sorted, int64_t nq, int k) {
    if (n_sorted < 0 || nq < 0 || k <= 0) return -1;
    return carve_ivf2(nullptr, n_sorted, nq, k).bytes;
}

int aura_centroid_probe(const float* centroids, const float* queries, int64_t D, int64_t nq, int nprobe,
                        int32_t* ids_out, void* workspace, int64_t workspace_bytes, void* stream) {
    if (D <= 0 || nq < 0 || nq > 0x7fffffffLL / 256 || nprobe <= 0 || nprobe > 8) return AURA_E_INVAL;
    if (nq == 0) return AURA_OK;
    if (!centroids || !queries || !ids_out || !workspace) return AURA_E_INVAL;
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return AURA_E_ALIGN;
    if (workspace_bytes < aura_centroid_probe_workspace_bytes(nq)) return AURA_E_INVAL;
    float* const dist = static_cast<float*>(workspace);
    uint32_t* const mask = reinterpret_cast<uint32_t*>(static_cast<char*>(workspace) + align_up(nq * 256 * 4, 256));
    return launch_probe(centroids, queries, D, (int)nq, nprobe, dist, mask, ids_out, static_cast<hipStream_t>(stream));
}

static int knn_search_ivf2_impl(const float* bank, const float* inv_norm, const float* meta,
                         const uint16_t* sorted_bf16, const float* rho, const int32_t* sorted_rows, const int32_t* pad_off,
                         const int32_t* list_len, const int32_t* lists_flag, int64_t n_sorted, int64_t N,
                         const float* queries, float now,
                         int64_t D, int64_t nq, int k,
                         const float* centroids, int nprobe, int32_t idx_base, float* out_scores,
                         int32_t* out_idx, void* workspace, int64_t workspace_bytes,
                         int32_t* overflow_out, const int32_t* probe_ids, const float4* rowc_cached, void* stream,
                         int stg = 0, int k2 = 0, float* bounds = nullptr, uint32_t* host_word = nullptr,
                         uint32_t host_seq = 0) {
    // stg: 0 = the whole recall; 1 = up to the sampled bounds, written to bounds[nq][2] = {k-th, k2-th largest
    // sampled lower bound}; 2 = from there on (same workspace, untouched in between), every query's threshold
    // first raised to bounds[q] (one float per query: the caller's combination of all shards' stage-1 bounds).
    // Staged calls are single passes (nq <= 8192).
    if (n_sorted <= 0 || N <= 0 || N > 0x7ffffff0LL || D <= 0 || D > 768 || (D & 7) || nq < 0 || k <= 0 ||
        k > COARSE_MAX_K)
        return AURA_E_INVAL;
    if (nprobe <= 0 || nprobe > 8) return AURA_E_INVAL;
    if (nq == 0) return AURA_OK;
    if (!bank || !inv_norm || !meta || !sorted_bf16 || !rho || !sorted_rows || !pad_off || !list_len || !queries || !centroids ||
        !out_scores || !out_idx || !workspace)
        return AURA_E_INVAL;
    if ((reinterpret_cast<uintptr_t>(meta) & 15) || (reinterpret_cast<uintptr_t>(sorted_bf16) & 15) ||
        (reinterpret_cast<uintptr_t>(queries) & 15) || (reinterpret_cast<uintptr_t>(bank) & 15))
        return AURA_E_ALIGN;
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return AURA_E_ALIGN;
    if (n_sorted > 0x7ffffff0LL || (n_sorted & 15)) return AURA_E_INVAL;
    const Ivf2Workspace w = carve_ivf2(workspace, n_sorted, nq, k);
    if (w.bytes > workspace_bytes) return AURA_E_INVAL;
    if (stg != 0 && (nq > w.qp || !bounds || stg < 0 || stg > 4 || k2 < 0 || k2 > k)) return AURA_E_INVAL;
    const bool stage_a = stg <= 1;                           // query prep .. thresholds run in this call (stages 0, 1)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int KS = D <= 256 ? 8 : (D <= 512 ? 16 : 24);
    const float e_fix = aura_e_fix((float)D);                // see aura_knn_coarse.inl
    const int cus = device_cu_count();
    int rc;
    // AURA_IVF2_TRACE: synchronise after every stage and name it on stderr (to localise a device fault)
    static const bool trace = getenv("AURA_IVF2_TRACE") != nullptr;
    auto stage = [&](const char* name) {
        if (trace) {
            const hipError_t e = hipStreamSynchronize(s);
            fprintf(stderr, "[ivf2] %s: %s\n", name, e == hipSuccess ? "ok" : hipGetErrorString(e));
            fflush(stderr);
        }
    };
    const int stiles = ivf2_stiles(n_sorted, k, stg != 0);   // sample tiles per list (fewer when a bound from outside follows)
    static const bool w4 = getenv("AURA_CS_WAVES4") != nullptr;   // A/B runs: one wave per SIMD
    // AURA_IVF_WG4: TWO independent four-wave workgroups per CU, 128 query slots per block (round-3 experiment)
    static const bool wg4 = !w4 && getenv("AURA_IVF_WG4") != nullptr;
    const int bsh = wg4 ? 7 : 8;
    // Relative time of a filter tile in a block of <= 128 queries and in a fuller one (the one- and the
    // two-column-block form of the tile loop): the plan splits tiles x weight evenly over the workgroups.
    // AURA_IVF_W=s,d overrides (tuning runs); the one-wave-per-SIMD form has a single tile loop.
    static int w_sparse = 0, w_dense = 0;
    if (w_sparse == 0) {
        int ws = 5, wd = 7;
        if (const char* e = getenv("AURA_IVF_W")) { if (sscanf(e, "%d,%d", &ws, &wd) != 2 || ws < 1 || wd < 1 || ws > 64 || wd > 64) { ws = 5; wd = 7; } }
        if (w4 || wg4) ws = wd = 1;
        w_dense = wd; w_sparse = ws;
    }
    // The sorted rows' score constants depend on `now` and the bank only: the caller may keep them across
    // calls (rowc_cached, aura_ivf2_row_constants; 31 us per call at 1 M rows otherwise), else one launch per
    // call, not per pass.  (Measured dead end: the same launch on a side stream, forked and joined with events
    // so that it runs beside the probe / plan launches -- 0.80 vs 0.78 ms per 2048-query call, the two event
    // creations and the cross-stream waits cost more than the 35 us they hide.)
    const float4* rowc = rowc_cached;
    if (!rowc) {
        if (stage_a) {
            hipLaunchKernelGGL(ivf2_rowc_kernel, dim3((unsigned)((n_sorted + 255) / 256)), dim3(256), 0, s,
                               meta, rho, sorted_rows, n_sorted, now, (float)D, w.rowc);
            if ((rc = check_launch())) return rc;
        }
        rowc = w.rowc;
    }
    for (int64_t qb0 = 0; qb0 < nq; qb0 += w.qp) {
        const int nqb = (int)((nq - qb0) < w.qp ? (nq - qb0) : w.qp);
        const float* qptr = queries + qb0 * D;
        if (stage_a) {
        // per query: 1/||q||, bf16 fragments, eq; resets of the pass (per-list counters, qslot, the call's flag)
        hipLaunchKernelGGL(ivf2_qprep_kernel, dim3((unsigned)((nqb + 1 + 3) / 4)), dim3(256), 0, s,
                           qptr, (int64_t)nqb, D, KS, w.qhat, w.inv_q, w.eq_q, w.qslot, w.lq_cnt,
                           qb0 == 0 ? overflow_out : nullptr, lists_flag);
        if ((rc = check_launch())) return rc;
        stage("query prep");
        // the probe launch also fills the per-list query lists (lq_cnt / lq_list); probes that the caller
        // already has (a sharded bank computes them once per query, not once per rank) only fill the lists
        if (probe_ids) {
            hipLaunchKernelGGL(ivf2_lists_from_ids_kernel, dim3((unsigned)((nqb * 8 + 256 * LFI_PER - 1) / (256 * LFI_PER))), dim3(256), 0, s,
                               probe_ids + qb0 * 8, nqb, nprobe, w.lq_cnt, w.lq_list);
            if ((rc = check_launch())) return rc;
        } else if ((rc = launch_probe(centroids, qptr, D, nqb, nprobe, w.probe_dist, w.probe, w.probe_ids, s,
                                      w.lq_cnt, w.lq_list, IVF2_MAXQ))) return rc;
        stage("probe");
        hipLaunchKernelGGL(ivf2_plan_kernel, dim3(1), dim3(256), 0, s, w.lq_cnt, pad_off, list_len, w.blk_off,
                           w.blk_list, w.blk_row0, w.blk_stride, w.blk_nq, w.item_off, w.sitem_off, w.nblk,
                           stiles, w_sparse, w_dense, bsh);
        if ((rc = check_launch())) return rc;
        stage("plan");
        const int nslot_blocks = (int)((((int64_t)ivf2_maxblk(w.qp, bsh) << bsh) + 255) / 256);   // 256 slots per launch block
        hipLaunchKernelGGL(ivf2_slots_kernel, dim3((unsigned)nslot_blocks), dim3(256), 0, s,
                           w.lq_cnt, w.lq_list, w.blk_off, w.blk_list, w.nblk, w.eq_q, w.slotq, w.qslot, w.thr,
                           w.eq_slot, bsh);
        if ((rc = check_launch())) return rc;
        stage("slots");
        }

        CoarseArgs c{};
        c.bank = bank; c.bank16 = sorted_bf16; c.rowc = rowc; c.qhat = w.qhat; c.inv_q = w.inv_q;
        c.eq = w.eq_slot; c.qzero = nqb;
        c.N = n_sorted; c.D = D; c.nq = ivf2_maxblk(w.qp, bsh) << bsh;
        c.thr = w.thr; c.cnt = w.cnt; c.cand_scores = w.cand_scores; c.cand_idx = w.cand_idx; c.cap = w.cap;
        c.blk_row0 = w.blk_row0; c.blk_stride = w.blk_stride; c.nblk = w.nblk; c.slotq = w.slotq;
        c.blk_nq = w.blk_nq; c.w_sparse = w_sparse; c.w_dense = w_dense;
        c.gmax = w.gmax; c.gmax_ld = 2 * stiles; c.item_off = w.sitem_off;
        const int cs_dbg = cs_dbg_flags();
        c.dbg = cs_dbg;
        auto launch = [&](int mode) -> int {
            if (wg4) {
                if (KS == 8) return launch_coarse_ivf<8, 4, 2>(c, mode, 2 * cus, s);
                if (KS == 16) return launch_coarse_ivf<16, 4, 2>(c, mode, 2 * cus, s);
                return launch_coarse_ivf<24, 4, 2>(c, mode, 2 * cus, s);
            }
            if (w4) {
                if (KS == 8) return launch_coarse_ivf<8, 4>(c, mode, cus, s);
                if (KS == 16) return launch_coarse_ivf<16, 4>(c, mode, cus, s);
                return launch_coarse_ivf<24, 4>(c, mode, cus, s);
            }
            if (KS == 8) return launch_coarse_ivf<8, 8>(c, mode, cus, s);
            if (KS == 16) return launch_coarse_ivf<16, 8>(c, mode, cus, s);
            return launch_coarse_ivf<24, 8>(c, mode, cus, s);
        };
        if (stage_a) {
        if ((rc = launch(CS_MODE_SAMPLE))) return rc;
        stage("sample scan");
        {
            const dim3 tg((unsigned)((nqb + 3) / 4)), tb(256);
#define AURA_THR2(PER) hipLaunchKernelGGL((ivf2_threshold_kernel<PER>), tg, tb, 0, s, w.gmax, w.qslot, w.blk_list, list_len, \
                                          nprobe, k, nqb, w.thr, w.cnt, stg == 1 ? k2 : 0, stg == 1 ? bounds : nullptr, bsh)
            if (stiles == 8) AURA_THR2(2);
            else if (stiles == 16) AURA_THR2(4);
            else if (stiles == 32) AURA_THR2(8);
            else if (stiles == 64) AURA_THR2(16);
            else if (stiles == 128) AURA_THR2(32);
            else AURA_THR2(64);
#undef AURA_THR2
        }
        if ((rc = check_launch())) return rc;
        stage("threshold");
        if (stg == 1) return AURA_OK;
        } else if (stg != 3) {
            // the caller's bound (e.g. combined over the shards of a row-sharded bank) tightens the thresholds
            hipLaunchKernelGGL(ivf2_raise_thr_kernel, dim3((unsigned)((((int64_t)ivf2_maxblk(w.qp, bsh) << bsh) + 255) / 256)), dim3(256), 0, s,
                               w.slotq, w.nblk, bounds, w.thr, bsh);
            if ((rc = check_launch())) return rc;
            stage("raise thresholds");
        }
        if (stg != 3) {                                      // (stage 3: the candidates of stage 4 are in the workspace)
        c.gmax = nullptr; c.item_off = w.item_off;
        static int tm_left2 = 2;                             // AURA_CS_DBG bit 64: phase times of the first launches
        const bool tm2 = CS_TIMERS && (cs_dbg & 64) && tm_left2 > 0;
        if (tm2) {
            c.gmax = w.gmax;
            (void)hipMemsetAsync(w.gmax, 0, (size_t)cus * 8 * 8 * 4, s);
        }
        const bool prof = g_prof.on && g_prof.used < g_prof.cap;
        if (prof) {
            (void)hipEventRecord(g_prof.start[g_prof.used], s);
            g_prof.rows = n_sorted; g_prof.nq = nqb; g_prof.kind = 3;
        }
        if (trace) (void)hipMemsetAsync(w.cand_idx, 0xff, (size_t)nqb * w.cap * 4, s);   // unwritten slots show up as row -1
        if ((rc = launch(CS_MODE_FILTER))) return rc;
        if (prof) (void)hipEventRecord(g_prof.stop[g_prof.used++], s);
        stage("filter scan");
        if (tm2) {
            --tm_left2;
            print_cs_phases(s, w.gmax, cus);
            std::vector<int32_t> lq(256);
            (void)hipMemcpy(lq.data(), w.lq_cnt, 256 * 4, hipMemcpyDeviceToHost);
            std::vector<int32_t> ll(256);
            (void)hipMemcpy(ll.data(), list_len, 256 * 4, hipMemcpyDeviceToHost);
            long rows_le64 = 0, rows_le128 = 0, rows_le256 = 0, rows_more = 0, reads = 0;
            for (int c2 = 0; c2 < 256; ++c2) {
                const long blocks = (lq[c2] + 255) / 256;
                reads += blocks * ll[c2];
                if (lq[c2] <= 64) rows_le64 += ll[c2]; else if (lq[c2] <= 128) rows_le128 += ll[c2];
                else if (lq[c2] <= 256) rows_le256 += ll[c2]; else rows_more += ll[c2];
            }
            fprintf(stderr, "[ivf2] rows in lists probed by <=64 / <=128 / <=256 / >256 queries: %ld / %ld / %ld / %ld; row reads %ld\n",
                    rows_le64, rows_le128, rows_le256, rows_more, reads);
        }
        }

        RefineArgs r{};
        r.bank = bank; r.inv_norm = inv_norm; r.meta = meta; r.queries = qptr; r.inv_q = w.inv_q;
        r.now = now; r.e_cos = 0.0f; r.N = N; r.D = D; r.k = k; r.cnt = w.cnt;
        r.rho = rho; r.eq = w.eq_q; r.e_fix = e_fix; r.eq_worst = coarse_eq_worst((float)D);
        r.cand_scores = w.cand_scores; r.cand_idx = w.cand_idx; r.cap = w.cap; r.idx_base = idx_base;
        r.out_scores = out_scores + qb0 * k; r.out_idx = out_idx + qb0 * k; r.overflow = overflow_out;
        if (stg == 4) {
            // stage 4 ends here: the candidates' k-th and k2-th largest lower bounds per query, for the caller to
            // combine over its shards; stage 3 re-scores against the combined bound
            hipLaunchKernelGGL(coarse_refine_bounds_kernel, dim3((unsigned)((nqb + RF_THREADS / 64 - 1) / (RF_THREADS / 64))),
                               dim3(RF_THREADS), 0, s, r, nqb, k2, bounds);
            return check_launch();
        }
        if (stg == 3) r.t2_ext = bounds;
        if (trace) {                                         // candidate lists with row ids outside the bank
            (void)hipStreamSynchronize(s);
            std::vector<int32_t> hc((size_t)nqb * CNT_STRIDE), hi((size_t)nqb * w.cap);
            std::vector<float> hu((size_t)nqb * w.cap);
            (void)hipMemcpy(hc.data(), w.cnt, hc.size() * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hi.data(), w.cand_idx, hi.size() * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hu.data(), w.cand_scores, hu.size() * 4, hipMemcpyDeviceToHost);
            int shown = 0; long bad = 0, tot = 0;
            for (int q = 0; q < nqb; ++q) {
                const int n = hc[(size_t)q * CNT_STRIDE] < w.cap ? hc[(size_t)q * CNT_STRIDE] : w.cap;
                tot += n;
                for (int i = 0; i < n; ++i) {
                    const int32_t row = hi[(size_t)q * w.cap + i];
                    if ((uint32_t)row < (uint32_t)N) continue;
                    ++bad;
                    if (shown++ < 40)
                        fprintf(stderr, "[ivf2] invalid candidate: q %d pos %d of %d row %d (0x%08x) U %g\n", q, i, n,
                                row, (unsigned)row, hu[(size_t)q * w.cap + i]);
                }
            }
            fprintf(stderr, "[ivf2] candidates %ld, invalid %ld\n", tot, bad);
        }
        static int rtm2_left = 2;                            // AURA_CS_DBG bit 128: refine phase times
        const bool rtm2 = CS_TIMERS && (cs_dbg & 128) && rtm2_left > 0;
        if (rtm2) r.dbg_out = w.gmax;
        if ((rc = launch_refine_for(r, nqb, D, cus, s, w.heavy))) return rc;
        if (host_word && qb0 + w.qp >= nq) {                 // behind the call's last launch: flag + sequence number to the host
            hipLaunchKernelGGL(ivf2_signal_kernel, dim3(1), dim3(64), 0, s, overflow_out, host_word, host_seq);
            if ((rc = check_launch())) return rc;
        }
        if (rtm2) {
            --rtm2_left;
            print_refine_phases(s, w.gmax, nqb, "inverted lists");
        }
        stage("refine");
    }
    return AURA_OK;
}

int aura_ivf2_row_constants(const float* meta, const float* rho, const int32_t* sorted_rows, int64_t n_sorted,
                            int64_t D, float now, float* row_constants, void* stream) {
    if (n_sorted < 0 || n_sorted > 0x7ffffff0LL || D <= 0) return AURA_E_INVAL;
    if (n_sorted == 0) return AURA_OK;
    if (!meta || !rho || !sorted_rows || !row_constants) return AURA_E_INVAL;
    if ((reinterpret_cast<uintptr_t>(meta) & 15) || (reinterpret_cast<uintptr_t>(row_constants) & 15)) return AURA_E_ALIGN;
    hipLaunchKernelGGL(ivf2_rowc_kernel, dim3((unsigned)((n_sorted + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), meta, rho, sorted_rows, n_sorted, now, (float)D,
                       reinterpret_cast<float4*>(row_constants));
    return check_launch();
}

int aura_knn_search_ivf2(const float* bank, const float* inv_norm, const float* meta,
                         const uint16_t* sorted_bf16, const float* rho, const int32_t* sorted_rows, const int32_t* pad_off,
                         const int32_t* list_len, const int32_t* lists_flag, const float* row_constants,
                         int64_t n_sorted, int64_t N,
                         const float* queries, float now, int64_t D, int64_t nq, int k,
                         const float* centroids, int nprobe, int32_t idx_base, float* out_scores,
                         int32_t* out_idx, void* workspace, int64_t workspace_bytes,
                         int32_t* overflow_out, void* stream) {
    if (reinterpret_cast<uintptr_t>(row_constants) & 15) return AURA_E_ALIGN;
    return knn_search_ivf2_impl(bank, inv_norm, meta, sorted_bf16, rho, sorted_rows, pad_off, list_len, lists_flag,
                                n_sorted, N, queries, now, D, nq, k, centroids, nprobe, idx_base, out_scores, out_idx,
                                workspace, workspace_bytes, overflow_out, nullptr,
                                reinterpret_cast<const float4*>(row_constants), stream);
}

int aura_knn_search_ivf2_probed(const float* bank, const float* inv_norm, const float* meta,
                                const uint16_t* sorted_bf16, const float* rho, const int32_t* sorted_rows,
                                const int32_t* pad_off, const int32_t* list_len, const int32_t* lists_flag,
                                const float* row_constants,
                                int64_t n_sorted, int64_t N, const float* queries, float now, int64_t D, int64_t nq,
                                int k, const float* centroids, int nprobe, const int32_t* probe_ids, int32_t idx_base,
                                float* out_scores, int32_t* out_idx, void* workspace, int64_t workspace_bytes,
                                int32_t* overflow_out, void* stream) {
    if (!probe_ids) return AURA_E_INVAL;
    if (reinterpret_cast<uintptr_t>(row_constants) & 15) return AURA_E_ALIGN;
    return knn_search_ivf2_impl(bank, inv_norm, meta, sorted_bf16, rho, sorted_rows, pad_off, list_len, lists_flag,
                                n_sorted, N, queries, now, D, nq, k, centroids, nprobe, idx_base, out_scores, out_idx,
                                workspace, workspace_bytes, overflow_out, probe_ids,
                                reinterpret_cast<const float4*>(row_constants), stream);
}

int aura_host_word_alloc(void** host_word_out) {
    if (!host_word_out) return AURA_E_INVAL;
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return AURA_E_LAUNCH;
    for (int i = 0; i < 16; ++i) static_cast<volatile uint32_t*>(p)[i] = 0u;
    *host_word_out = p;
    return AURA_OK;
}

// flag + sequence number to a host word behind whatever the stream holds (the staged recall's last stage, a chain of
// passes): the caller polls host_word[1] == host_seq instead of synchronising the stream -- a blocking wait on a
// ~1 ms stream costs the host 1-2 ms on this runtime (interrupt wake-up), the poll costs what the GPU takes
int aura_signal_flag(const int32_t* flag_dev, uint32_t* host_word, uint32_t host_seq, void* stream) {
    if (!host_word) return AURA_E_INVAL;
    hipLaunchKernelGGL(ivf2_signal_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), flag_dev, host_word,
                       host_seq);
    return check_launch();
}

int aura_host_word_free(void* host_word) {
    if (!host_word) return AURA_OK;
    return hipHostFree(host_word) == hipSuccess ? AURA_OK : AURA_E_LAUNCH;
}

int aura_knn_search_ivf2_signal(const float* bank, const float* inv_norm, const float* meta,
                                const uint16_t* sorted_bf16, const float* rho, const int32_t* sorted_rows,
                                const int32_t* pad_off, const int32_t* list_len, const int32_t* lists_flag,
                                const float* row_constants,
                                int64_t n_sorted, int64_t N, const float* queries, float now, int64_t D, int64_t nq,
                                int k, const float* centroids, int nprobe, const int32_t* probe_ids, int32_t idx_base,
                                float* out_scores, int32_t* out_idx, void* workspace, int64_t workspace_bytes,
                                int32_t* overflow_out, uint32_t* host_word, uint32_t host_seq, void* stream) {
    if (!host_word || !overflow_out) return AURA_E_INVAL;
    if (reinterpret_cast<uintptr_t>(row_constants) & 15) return AURA_E_ALIGN;
    return knn_search_ivf2_impl(bank, inv_norm, meta, sorted_bf16, rho, sorted_rows, pad_off, list_len, lists_flag,
                                n_sorted, N, queries, now, D, nq, k, centroids, nprobe, idx_base, out_scores, out_idx,
                                workspace, workspace_bytes, overflow_out, probe_ids,
                                reinterpret_cast<const float4*>(row_constants), stream, 0, 0, nullptr, host_word, host_seq);
}

int aura_knn_search_ivf2_staged(const float* bank, const float* inv_norm, const float* meta,
                                const uint16_t* sorted_bf16, const float* rho, const int32_t* sorted_rows,
                                const int32_t* pad_off, const int32_t* list_len, const int32_t* lists_flag,
                                const float* row_constants,
                                int64_t n_sorted, int64_t N, const float* queries, float now, int64_t D, int64_t nq,
                                int k, const float* centroids, int nprobe, const int32_t* probe_ids, int32_t idx_base,
                                float* out_scores, int32_t* out_idx, void* workspace, int64_t workspace_bytes,
                                int32_t* overflow_out, int stage, int k2, float* bounds, void* stream) {
    if (stage < 1 || stage > 4) return AURA_E_INVAL;        // 1, 2, 4 + 3 (see the header)
    if (reinterpret_cast<uintptr_t>(row_constants) & 15) return AURA_E_ALIGN;
    return knn_search_ivf2_impl(bank, inv_norm, meta, sorted_bf16, rho, sorted_rows, pad_off, list_len, lists_flag,
                                n_sorted, N, queries, now, D, nq, k, centroids, nprobe, idx_base, out_scores, out_idx,
                                workspace, workspace_bytes, overflow_out, probe_ids,
                                reinterpret_cast<const float4*>(row_constants), stream, stage, k2, bounds);
}

int aura_knn_search(const float* bank, const float* inv_norm, const float* meta, const float* loc,
                    int spatial_dims, const float* queries, const float* q_loc, float now,
                    int64_t N, int64_t D, int64_t nq, int k, int32_t idx_base, float* out_scores,
                    int32_t* out_idx, void* workspace, int64_t workspace_bytes, void* stream) {
    return aura_knn_search_ex(bank, inv_norm, meta, loc, spatial_dims, queries, q_loc, now, N, D, nq,
                              k, idx_base, out_scores, out_idx, workspace, workspace_bytes, 0,
                              nullptr, nullptr, 0, stream);
}

// workspace of the IVF path: `cap` candidate slots per query (the caller sizes it from the list
// lengths: the 8 longest lists bound any query's candidates; a larger need sets the overflow flag)
constexpr int IVF_SEL_CHUNK = 2048;

struct IvfWorkspace {
    float* inv_q; int32_t* cnt; uint32_t* probe; float* probe_dist; int32_t* probe_ids; int32_t* qbase;
    int32_t* lq_cnt; int32_t* lq_list; int32_t* item_off; int32_t* work_counter;
    float* cand_scores; int32_t* cand_idx; float* cand2_scores; int32_t* cand2_idx;
    int cap2, qp;
    int64_t bytes;
};

static IvfWorkspace carve_ivf(void* base, int64_t nq, int k, int cap) {
    IvfWorkspace w;
    char* p = static_cast<char*>(base);
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        char* r = p ? p + off : nullptr;
        off += align_up(bytes, 256);
        return r;
    };
    const int64_t qp = nq < IVF_MAXQ ? (nq > 0 ? nq : 1) : IVF_MAXQ;
    w.qp = (int)qp;
    w.inv_q = reinterpret_cast<float*>(take(qp * 4));
    w.cnt = reinterpret_cast<int32_t*>(take(qp * CNT_STRIDE * 4));
    w.probe = reinterpret_cast<uint32_t*>(take(qp * 32));
    w.probe_dist = reinterpret_cast<float*>(take(qp * 256 * 4));
    w.probe_ids = reinterpret_cast<int32_t*>(take(qp * 8 * 4));
    w.qbase = reinterpret_cast<int32_t*>(take(qp * 8 * 4));
    w.lq_cnt = reinterpret_cast<int32_t*>(take(256 * 4));
    w.lq_list = reinterpret_cast<int32_t*>(take((int64_t)256 * IVF_MAXQ * 4));
    w.item_off = reinterpret_cast<int32_t*>(take(257 * 4));
    w.work_counter = reinterpret_cast<int32_t*>(take(16));
    w.cand_scores = reinterpret_cast<float*>(take(qp * cap * 4));
    w.cand_idx = reinterpret_cast<int32_t*>(take(qp * cap * 4));
    w.cap2 = (int)align_up((int64_t)(cap / IVF_SEL_CHUNK) * k, 64);
    w.cand2_scores = reinterpret_cast<float*>(take(qp * w.cap2 * 4));
    w.cand2_idx = reinterpret_cast<int32_t*>(take(qp * w.cap2 * 4));
    w.bytes = off;
    return w;
}

static bool ivf_cap_ok(int k, int cap) {
    return cap >= IVF_SEL_CHUNK && cap % IVF_SEL_CHUNK == 0 &&
           (int64_t)(cap / IVF_SEL_CHUNK) * k <= SEL_LDS_KEYS_HARD;
}

int64_t aura_knn_ivf_workspace_bytes(int64_t nq, int k, int cap) {
    if (nq < 0 || k <= 0 || k > SEL_MAX_K || !ivf_cap_ok(k, cap)) return AURA_E_INVAL;
    return carve_ivf(nullptr, nq, k, cap).bytes;
}

int aura_knn_search_ivf(const float* bank, const float* inv_norm, const float* meta,
                        const float* queries, float now, int64_t N, int64_t D, int64_t nq, int k,
                        const float* centroids, int nprobe, const int32_t* list_rows,
                        const int32_t* list_off, const int32_t* list_len, int cap, int32_t idx_base,
                        float* out_scores, int32_t* out_idx, void* workspace,
                        int64_t workspace_bytes, int32_t* overflow_out, void* stream) {
    if (N <= 0 || D <= 0 || nq < 0 || k <= 0 || k > SEL_MAX_K) return AURA_E_INVAL;
    if (nprobe <= 0 || nprobe > 8 || !ivf_cap_ok(k, cap)) return AURA_E_INVAL;
    if (nq == 0) return AURA_OK;
    if (!bank || !inv_norm || !meta || !queries || !centroids || !list_rows || !list_off ||
        !list_len || !out_scores || !out_idx || !workspace)
        return AURA_E_INVAL;
    if (reinterpret_cast<uintptr_t>(meta) & 15) return AURA_E_ALIGN;
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return AURA_E_ALIGN;
    const IvfWorkspace w = carve_ivf(workspace, nq, k, cap);
    if (w.bytes > workspace_bytes) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc;
    const bool vec4 = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(bank) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(queries) & 15) == 0);
    // persistent grid: 6 workgroups of 256 threads fit a CU (LDS 23 KiB, 78 VGPRs)
    const dim3 grid((unsigned)(device_cu_count() * 6));

    for (int64_t qb0 = 0; qb0 < nq; qb0 += w.qp) {
        const int nqb = (int)((nq - qb0) < w.qp ? (nq - qb0) : w.qp);
        const float* qptr = queries + qb0 * D;
        // expected queries per list = nqb * nprobe / 256: 64-query tiles once that reaches ~48
        const int qt = (int64_t)nqb * nprobe >= 48 * 256 ? 2 : 1;
        hipLaunchKernelGGL(query_prep_kernel, dim3((unsigned)((nqb + 3) / 4)), dim3(256), 0, s, qptr,
                           w.inv_q, (int64_t)nqb, D, qb0 == 0 ? overflow_out : nullptr);
        if ((rc = check_launch())) return rc;
        if ((rc = launch_probe(centroids, qptr, D, nqb, nprobe, w.probe_dist, w.probe, w.probe_ids, s)))
            return rc;
        hipLaunchKernelGGL(ivf_prepare_kernel, dim3(1), dim3(256), 0, s, w.probe_ids, nprobe, nqb,
                           list_len, w.lq_cnt, w.lq_list, w.qbase, w.cnt, w.item_off, w.work_counter,
                           cap, qt * 32, overflow_out);
        if ((rc = check_launch())) return rc;

        IvfArgs a{};
        a.bank = bank; a.inv_norm = inv_norm; a.meta = meta; a.queries = qptr; a.inv_q = w.inv_q;
        a.list_rows = list_rows; a.list_off = list_off; a.list_len = list_len;
        a.lq_cnt = w.lq_cnt; a.lq_list = w.lq_list; a.qbase = w.qbase;
        a.item_off = w.item_off; a.work_counter = w.work_counter;
        a.cand_scores = w.cand_scores; a.cand_idx = w.cand_idx; a.cap = cap;
        a.now = now; a.D = D; a.nq = nqb;
        const bool prof = g_prof.on && g_prof.used < g_prof.cap;
        if (prof) {
            (void)hipEventRecord(g_prof.start[g_prof.used], s);
            g_prof.rows = N; g_prof.nq = nqb;
        }
        const size_t lds = (size_t)(32 * qt + IVF_GROUP) * LDS_STRIDE * sizeof(float);
        if (qt == 2) {
            if (vec4) hipLaunchKernelGGL((ivf_scan_kernel<true, 2>), grid, dim3(IVF_THREADS), lds, s, a);
            else hipLaunchKernelGGL((ivf_scan_kernel<false, 2>), grid, dim3(IVF_THREADS), lds, s, a);
        } else {
            if (vec4) hipLaunchKernelGGL((ivf_scan_kernel<true, 1>), grid, dim3(IVF_THREADS), lds, s, a);
            else hipLaunchKernelGGL((ivf_scan_kernel<false, 1>), grid, dim3(IVF_THREADS), lds, s, a);
        }
        if ((rc = check_launch())) return rc;
        if (prof) (void)hipEventRecord(g_prof.stop[g_prof.used++], s);

        // exact top-k of each query's slots (count = total length of its probed lists), in two
        // levels: 2048-slot chunks keep their k best (many small workgroups, 16 KiB of LDS each),
        // then one sorted select over the chunk winners
        const int64_t nch = cap / IVF_SEL_CHUNK;
        SelectArgs r{};
        r.src_scores = w.cand_scores; r.src_idx = w.cand_idx; r.src_qs = cap; r.src_inner = cap;
        r.src_outer = 0; r.src_cnt = w.cnt; r.n_max = cap; r.chunk = IVF_SEL_CHUNK; r.blk = 1; r.step = 1;
        r.row_begin = 0; r.row_end = N; r.k = k; r.sorted = 0;
        r.dst_scores = w.cand2_scores; r.dst_idx = w.cand2_idx; r.dst_qs = w.cap2; r.dst_off = 0;
        if ((rc = launch_select(r, nch, nqb, s))) return rc;
        SelectArgs fin{};
        fin.src_scores = w.cand2_scores; fin.src_idx = w.cand2_idx; fin.src_qs = w.cap2;
        fin.src_inner = w.cap2; fin.src_outer = 0;
        fin.src_cnt = nullptr; fin.n_max = nch * k; fin.chunk = (int)(nch * k); fin.blk = 1; fin.step = 1;
        fin.row_begin = 0; fin.row_end = N; fin.k = k; fin.sorted = 1;
        fin.dst_scores = out_scores + qb0 * k; fin.dst_idx = out_idx + qb0 * k; fin.dst_qs = k;
        fin.idx_base = idx_base;
        if ((rc = launch_select(fin, 1, nqb, s))) return rc;
    }
    return AURA_OK;
}

int aura_topk_merge(const float* in_scores, const int32_t* in_idx, int S, int64_t nq, int k,
                    float* out_scores, int32_t* out_idx, void* stream) {
    if (S <= 0 || nq < 0 || k <= 0 || k > SEL_MAX_K) return AURA_E_INVAL;
    if (nq == 0) return AURA_OK;
    if (!in_scores || !in_idx || !out_scores || !out_idx) return AURA_E_INVAL;
    if ((int64_t)S * k > SEL_LDS_KEYS) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int64_t q0 = 0; q0 < nq; q0 += 32768) {
        const int64_t nb = (nq - q0) < 32768 ? (nq - q0) : 32768;
        SelectArgs m{};
        m.src_scores = in_scores + q0 * k; m.src_idx = in_idx + q0 * k;
        m.src_qs = k; m.src_inner = k; m.src_outer = nq * k;
        m.src_cnt = nullptr; m.n_max = (int64_t)S * k; m.chunk = S * k;
        m.blk = 1; m.step = 1; m.row_begin = 0; m.row_end = 0x7fffffff;
        m.k = k; m.sorted = 1; m.dst_scores = out_scores + q0 * k; m.dst_idx = out_idx + q0 * k;
        m.dst_qs = k; m.idx_base = 0;
        int rc = launch_select(m, 1, (int)nb, s);
        if (rc) return rc;
    }
    return AURA_OK;
}

int aura_kmeans_assign(const float* bank, const float* centroids, float* cnorm2_ws,
                       int32_t* assign_out, int64_t N, int64_t D, int k, void* stream) {
    if (N < 0 || D <= 0 || k <= 0 || k > 256) return AURA_E_INVAL;
    if (N == 0) return AURA_OK;
    if (!bank || !centroids || !cnorm2_ws || !assign_out) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(row_norm2_kernel, dim3((unsigned)((k + 3) / 4)), dim3(256), 0, s, centroids,
                       cnorm2_ws, (int64_t)k, D);
    int rc = check_launch();
    if (rc) return rc;
    ScanArgs a{};
    a.bank = bank; a.queries = centroids; a.D = D; a.nq = k;
    a.row_begin = 0; a.row_end = N; a.tile_step = 1;
    a.qnorm2 = cnorm2_ws; a.assign_out = assign_out;
    return launch_scan<8, 1, 4>(a, MODE_ASSIGN, (N + 127) / 128, s);
}

// shader clock as the kernels see it: s_memtime ticks (shader clock) per s_memrealtime tick (100 MHz), one wave
// spinning for `spin_us` microseconds.  Tuning hook (tools/ab_headline.py prints it beside the timings: the filter
// launch's time is bimodal per process).
__global__ void clock_probe_kernel(float* out, int spin_ticks) {
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    uint64_t r1 = r0;
    while ((int64_t)(r1 - r0) < (int64_t)spin_ticks) { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = (float)((double)(c1 - c0) / (double)(r1 - r0) * 100.0);   // MHz
}

int aura_debug_clock_mhz(float* out_dev, int spin_us, void* stream) {
    if (!out_dev || spin_us < 1 || spin_us > 100000) return AURA_E_INVAL;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out_dev, spin_us * 100);
    return check_launch();
}

int aura_debug_cs_flags(int flags) {
    const int old = cs_dbg_flags();
    if (flags >= 0) g_cs_dbg = flags;
    return old;
}

int aura_profile_begin(int max_launches) {
    if (max_launches <= 0 || max_launches > 65536) return AURA_E_INVAL;
    if (g_prof.cap < max_launches) {
        for (int i = 0; i < g_prof.cap; ++i) { (void)hipEventDestroy(g_prof.start[i]); (void)hipEventDestroy(g_prof.stop[i]); }
        delete[] g_prof.start; delete[] g_prof.stop;
        g_prof.start = new hipEvent_t[max_launches];
        g_prof.stop = new hipEvent_t[max_launches];
        for (int i = 0; i < max_launches; ++i) {
            if (hipEventCreate(&g_prof.start[i]) != hipSuccess || hipEventCreate(&g_prof.stop[i]) != hipSuccess)
                return AURA_E_LAUNCH;
        }
        g_prof.cap = max_launches;
    }
    g_prof.used = 0;
    g_prof.on = true;
    return AURA_OK;
}

int aura_profile_end(float* ms_out_host, int max_out) {
    g_prof.on = false;
    if (!ms_out_host || max_out < 0) return AURA_E_INVAL;
    int n = g_prof.used < max_out ? g_prof.used : max_out;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(g_prof.stop[i]) != hipSuccess) return AURA_E_LAUNCH;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, g_prof.start[i], g_prof.stop[i]) != hipSuccess) return AURA_E_LAUNCH;
        ms_out_host[i] = ms;
    }
    return n;
}

int aura_profile_last_scan_kind(void) { return g_prof.kind; }

int aura_profile_last_scan(int64_t* rows_out, int64_t* nq_out) {
    if (!rows_out || !nq_out) return AURA_E_INVAL;
    *rows_out = g_prof.rows;
    *nq_out = g_prof.nq;
    return AURA_OK;
}

}  // extern "C"
