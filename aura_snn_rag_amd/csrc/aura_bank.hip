// aura_bank.hip -- maintenance kernels of the episodic bank for gfx950 (MI355X): the bf16 shadow
// rows the two-stage recall streams (plain and list-sorted, with their per-row rounding-error
// norms), incremental upkeep of the inverted lists after writes, and the centroid rebuild's means
// as a segmented reduction over rows grouped by cluster (rebuild_centroids,
// src/core/hippocampal.py:345-377).  All HBM-bound streaming / gather kernels, no MFMA.
//
// Shadow rows.  shadow[r] = bf16(bank[r] * inv_norm[r]): the NORMALISED row rounded to bf16, so
// the prefilter's accumulator is the cosine itself.  Beside it rho[r] is an upper bound of
//     || bf16(r_hat) - r_hat_true ||_2          (r_hat_true = bank[r] / ||bank[r]||, norm 1)
// computed from the row's actual rounding residual (typically 0.0017 against the worst case
// 2^-8 = 0.0039): the data-dependent half of the two-stage recall's error bound, see
// aura_knn_coarse.inl.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/aura_hip.h"

#pragma clang fp contract(off)

namespace {

#include "aura_rowc.inl"

typedef float f32x8b __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8b __attribute__((ext_vector_type(8)));

inline int check_launch_b() { return hipGetLastError() == hipSuccess ? AURA_OK : AURA_E_LAUNCH; }

// One wave converts one row: dst[j] = bf16(src[j] * inv) for j < D (D % 8 == 0, 16-byte chunks);
// returns (on every lane) rho = 1.001 * ||bf16(x) - x||_2 + (D/2 + 3) 2^-24, x = fl(src * inv).
// The second term covers fl(src * inv) against src / ||src|| (rounding of the product and of
// inv_norm's own sum / sqrt / division); the factor the rounding of this very reduction.
__device__ __forceinline__ float shadow_convert_row(const float* __restrict__ src, float inv,
                                                    uint16_t* __restrict__ dst, int64_t D, int lane) {
    float e2 = 0.0f;
    for (int64_t c = lane; c < D / 8; c += 64) {
        const float4 u = *reinterpret_cast<const float4*>(src + 8 * c);
        const float4 w = *reinterpret_cast<const float4*>(src + 8 * c + 4);
        f32x8b x;
        x[0] = u.x * inv; x[1] = u.y * inv; x[2] = u.z * inv; x[3] = u.w * inv;
        x[4] = w.x * inv; x[5] = w.y * inv; x[6] = w.z * inv; x[7] = w.w * inv;
        const bf16x8b b = __builtin_convertvector(x, bf16x8b);
        *reinterpret_cast<bf16x8b*>(dst + 8 * c) = b;
        const f32x8b back = __builtin_convertvector(b, f32x8b);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = back[e] - x[e];          // exact: both within a factor 2 of each other
            e2 = fmaf(d, d, e2);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off);
    return 1.001f * sqrtf(e2) + (0.5f * (float)D + 3.0f) * 5.9604645e-8f;
}

__device__ __forceinline__ void shadow_zero_row(uint16_t* __restrict__ dst, int64_t D, int lane) {
    for (int64_t c = lane; c < D / 8; c += 64)
        *reinterpret_cast<uint4*>(dst + 8 * c) = make_uint4(0u, 0u, 0u, 0u);
}

// shadow[r] / rho[r] for r in slots[0..n) or [row0, row0 + n); one wave per row
__global__ __launch_bounds__(256) void bank_shadow_kernel(const float* __restrict__ bank,
                                                          const float* __restrict__ inv_norm,
                                                          uint16_t* __restrict__ shadow,
                                                          float* __restrict__ rho,
                                                          const int64_t* __restrict__ slots, int64_t row0,
                                                          int64_t n, int64_t D) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
        const int64_t row = slots ? slots[i] : row0 + i;
        const float r = shadow_convert_row(bank + row * D, inv_norm[row], shadow + row * D, D, lane);
        if (lane == 0) rho[row] = r;
    }
}

// list-sorted shadow: sorted row i holds the shadow row of bank row sorted_rows[i] (zeros for -1);
// also refreshes rho[row] and, when given, the reverse map pos_of_row[row] = i
__global__ __launch_bounds__(256) void bank_shadow_sorted_kernel(const float* __restrict__ bank,
                                                                 const float* __restrict__ inv_norm,
                                                                 const int32_t* __restrict__ sorted_rows,
                                                                 uint16_t* __restrict__ out,
                                                                 float* __restrict__ rho,
                                                                 int32_t* __restrict__ pos_of_row,
                                                                 int64_t n_sorted, int64_t D) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n_sorted; i += (int64_t)gridDim.x * 4) {
        const int32_t row = sorted_rows[i];
        if (row < 0) {
            shadow_zero_row(out + i * D, D, lane);
            continue;
        }
        const float r = shadow_convert_row(bank + (int64_t)row * D, inv_norm[row], out + i * D, D, lane);
        if (lane == 0) {
            rho[row] = r;
            if (pos_of_row) pos_of_row[row] = (int32_t)i;
        }
    }
}

// Incremental upkeep of the inverted lists after a write of n DISTINCT bank rows `slots`: the row's old
// entry (if any) becomes a hole (sorted_rows = -1: scanned as padding), and the row is appended to the
// list of its centroid id meta[slot][2] (none if < 0) inside the list's slack.  One wave per row.
// The host guarantees the slack suffices (it counts appended rows and re-packs in time); a list that is
// full nevertheless sets *flag and drops the row from the lists.
__global__ __launch_bounds__(256) void ivf2_append_kernel(const float* __restrict__ bank,
                                                          const float* __restrict__ inv_norm,
                                                          const float* __restrict__ meta,
                                                          const int64_t* __restrict__ slots, int64_t n,
                                                          int64_t D, uint16_t* __restrict__ sorted_bf16,
                                                          int32_t* __restrict__ sorted_rows,
                                                          const int32_t* __restrict__ pad_off,
                                                          int32_t* __restrict__ list_len,
                                                          int32_t* __restrict__ pos_of_row,
                                                          float* __restrict__ rho, int32_t* __restrict__ flag,
                                                          float4* __restrict__ rowc, float rowc_now) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t slot = slots[i];
    int pos = -1;
    if (lane == 0) {
        const int old = pos_of_row[slot];
        if (old >= 0) {
            sorted_rows[old] = -1;
            if (rowc) rowc[old] = ivf2_row_constants(-1, meta, rho, rowc_now, (float)D);
        }
        const int c = (int)meta[slot * 4 + 2];
        if (c >= 0 && c < 256) {
            const int p = atomicAdd(&list_len[c], 1);
            if (p < pad_off[c + 1] - pad_off[c]) {
                pos = pad_off[c] + p;
                sorted_rows[pos] = (int32_t)slot;
            } else {
                atomicSub(&list_len[c], 1);
                atomicOr(flag, 1);
            }
        }
        pos_of_row[slot] = pos;
    }
    pos = __shfl(pos, 0);
    if (pos < 0) return;
    const float r = shadow_convert_row(bank + slot * D, inv_norm[slot], sorted_bf16 + (int64_t)pos * D, D, lane);
    if (lane == 0) {
        rho[slot] = r;
        // the caller's cached score constants (aura_ivf2_row_constants) follow the new entry
        if (rowc) {
            const float4 m = *reinterpret_cast<const float4*>(meta + slot * 4);
            rowc[pos] = coarse_row_constants(m, 0.0f, &r, rowc_now, aura_e_fix((float)D), 0.0f, coarse_eq_worst((float)D),
                                             __int_as_float((int32_t)slot));
        }
    }
}

// ------------------------------------------------------------------------------------------
// Centroid means as a segmented reduction (rebuild_centroids' masked means, hippocampal.py:358-363).
// Rows arrive grouped by cluster: order[seg_off[c] .. seg_off[c+1]) are the rows of cluster c.
// Stage 1: item = (cluster, chunk of KM_SEG consecutive rows of its segment); one wave sums its rows
//   in order (the feature dim across lanes, 16-byte loads: every row is one contiguous 4 D-byte read)
//   into partial[item][D].  Stage 2: one wave per (cluster, 256-column slice) adds the cluster's
//   partials in chunk order and divides by the count.  Fixed summation order -> reproducible means;
//   the bank is read once: N D 4 bytes + 2 N D 4 / KM_SEG of partials.
// ------------------------------------------------------------------------------------------
constexpr int KM_SEG = 64;

// s_pref[c] = first item of cluster c, s_pref[k] = number of items (k <= 256, 256 threads)
__device__ __forceinline__ void km_item_prefix(const int32_t* __restrict__ seg_off, int k, int* s_pref) {
    const int tid = threadIdx.x;
    int v = 0;
    if (tid < k) v = (seg_off[tid + 1] - seg_off[tid] + KM_SEG - 1) / KM_SEG;
    s_pref[tid + 1] = v;
    if (tid == 0) s_pref[0] = 0;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {              // inclusive scan of s_pref[1..256]
        const int add = tid >= off ? s_pref[tid + 1 - off] : 0;
        __syncthreads();
        s_pref[tid + 1] += add;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void kmeans_partial_kernel(const float* __restrict__ bank,
                                                             const int32_t* __restrict__ order,
                                                             const int32_t* __restrict__ seg_off,
                                                             float* __restrict__ partial, int64_t D, int k) {
    __shared__ int s_pref[257];
    km_item_prefix(seg_off, k, s_pref);
    const int lane = threadIdx.x & 63;
    const int n_items = s_pref[k];
    for (int item = blockIdx.x * 4 + (threadIdx.x >> 6); item < n_items; item += gridDim.x * 4) {
        int lo = 0, hi = k;                                 // cluster c with s_pref[c] <= item < s_pref[c+1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_pref[mid] <= item) lo = mid; else hi = mid;
        }
        const int c = lo;
        const int beg = seg_off[c] + (item - s_pref[c]) * KM_SEG;
        const int end = seg_off[c + 1];
        const int cnt = (end - beg) < KM_SEG ? (end - beg) : KM_SEG;
        const int32_t rid = lane < cnt ? order[beg + lane] : 0;
        float* const out = partial + (int64_t)item * D;
        for (int64_t col = (int64_t)lane * 4; col < D; col += 256) {
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
            int r = 0;
            for (; r + 4 <= cnt; r += 4) {                  // four rows in flight; fixed association
                const float4 x0 = *reinterpret_cast<const float4*>(bank + (int64_t)__builtin_amdgcn_readlane(rid, r) * D + col);
                const float4 x1 = *reinterpret_cast<const float4*>(bank + (int64_t)__builtin_amdgcn_readlane(rid, r + 1) * D + col);
                const float4 x2 = *reinterpret_cast<const float4*>(bank + (int64_t)__builtin_amdgcn_readlane(rid, r + 2) * D + col);
                const float4 x3 = *reinterpret_cast<const float4*>(bank + (int64_t)__builtin_amdgcn_readlane(rid, r + 3) * D + col);
                a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
                a1.x += x1.x; a1.y += x1.y; a1.z += x1.z; a1.w += x1.w;
                a2.x += x2.x; a2.y += x2.y; a2.z += x2.z; a2.w += x2.w;
                a3.x += x3.x; a3.y += x3.y; a3.z += x3.z; a3.w += x3.w;
            }
            for (; r < cnt; ++r) {
                const float4 x0 = *reinterpret_cast<const float4*>(bank + (int64_t)__builtin_amdgcn_readlane(rid, r) * D + col);
                a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
            }
            float4 s;
            s.x = (a0.x + a1.x) + (a2.x + a3.x); s.y = (a0.y + a1.y) + (a2.y + a3.y);
            s.z = (a0.z + a1.z) + (a2.z + a3.z); s.w = (a0.w + a1.w) + (a2.w + a3.w);
            *reinterpret_cast<float4*>(out + col) = s;
        }
    }
}

__global__ __launch_bounds__(256) void kmeans_reduce_kernel(const float* __restrict__ partial,
                                                            const int32_t* __restrict__ seg_off,
                                                            float* __restrict__ centroids, int64_t D, int k,
                                                            int sums_only) {
    __shared__ int s_pref[257];
    km_item_prefix(seg_off, k, s_pref);
    const int lane = threadIdx.x & 63;
    const int slices = (int)((D + 255) / 256);
    for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w < k * slices; w += gridDim.x * 4) {
        const int c = w / slices;
        const int64_t col = (int64_t)(w - c * slices) * 256 + lane * 4;
        const int len = seg_off[c + 1] - seg_off[c];
        if (col >= D) continue;
        if (len <= 0) {                                     // empty clusters keep their centroid (:362-363);
            if (sums_only) *reinterpret_cast<float4*>(centroids + (int64_t)c * D + col) = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;                                       // as a partial sum an empty cluster is zero
        }
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int it = s_pref[c]; it < s_pref[c + 1]; ++it) {
            const float4 x = *reinterpret_cast<const float4*>(partial + (int64_t)it * D + col);
            s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
        const float inv = sums_only ? 1.0f : (float)len;
        *reinterpret_cast<float4*>(centroids + (int64_t)c * D + col) =
            make_float4(s.x / inv, s.y / inv, s.z / inv, s.w / inv);
    }
}

// meta[i][2] = assign[i]; counts[c] = rows of cluster c (hippocampal.py:370-376)
__global__ __launch_bounds__(256) void kmeans_commit_kernel(const int32_t* __restrict__ assign,
                                                            const int32_t* __restrict__ seg_off,
                                                            float* __restrict__ meta,
                                                            float* __restrict__ counts, int64_t N, int k) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) meta[i * 4 + 2] = (float)assign[i];
    if (blockIdx.x == 0 && counts && (int)threadIdx.x < k)
        counts[threadIdx.x] = (float)(seg_off[threadIdx.x + 1] - seg_off[threadIdx.x]);
}

}  // namespace

extern "C" {

int aura_bank_shadow_update(const float* bank, const float* inv_norm, uint16_t* bank_bf16, float* rho,
                            const int64_t* slots, int64_t row0, int64_t n, int64_t D, void* stream) {
    if (n < 0 || D <= 0 || (D & 7) || row0 < 0) return AURA_E_INVAL;
    if (n == 0) return AURA_OK;
    if (!bank || !inv_norm || !bank_bf16 || !rho) return AURA_E_INVAL;
    if ((reinterpret_cast<uintptr_t>(bank) & 15) || (reinterpret_cast<uintptr_t>(bank_bf16) & 15)) return AURA_E_ALIGN;
    int64_t blocks = (n + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(bank_shadow_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       bank, inv_norm, bank_bf16, rho, slots, row0, n, D);
    return check_launch_b();
}

int aura_bank_shadow_sorted(const float* bank, const float* inv_norm, const int32_t* sorted_rows,
                            uint16_t* sorted_bf16, float* rho, int32_t* pos_of_row, int64_t n_sorted, int64_t D,
                            void* stream) {
    if (n_sorted < 0 || D <= 0 || (D & 7)) return AURA_E_INVAL;
    if (n_sorted == 0) return AURA_OK;
    if (!bank || !inv_norm || !sorted_rows || !sorted_bf16 || !rho) return AURA_E_INVAL;
    if ((reinterpret_cast<uintptr_t>(bank) & 15) || (reinterpret_cast<uintptr_t>(sorted_bf16) & 15)) return AURA_E_ALIGN;
    int64_t blocks = (n_sorted + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(bank_shadow_sorted_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), bank, inv_norm, sorted_rows, sorted_bf16, rho, pos_of_row,
                       n_sorted, D);
    return check_launch_b();
}

int aura_ivf2_append(const float* bank, const float* inv_norm, const float* meta, const int64_t* slots,
                     int64_t n, int64_t D, uint16_t* sorted_bf16, int32_t* sorted_rows, const int32_t* pad_off,
                     int32_t* list_len, int32_t* pos_of_row, float* rho, int32_t* flag, float* row_constants,
                     float row_constants_now, void* stream) {
    if (n < 0 || D <= 0 || (D & 7)) return AURA_E_INVAL;
    if (reinterpret_cast<uintptr_t>(row_constants) & 15) return AURA_E_ALIGN;
    if (n == 0) return AURA_OK;
    if (!bank || !inv_norm || !meta || !slots || !sorted_bf16 || !sorted_rows || !pad_off || !list_len ||
        !pos_of_row || !rho || !flag)
        return AURA_E_INVAL;
    if ((reinterpret_cast<uintptr_t>(bank) & 15) || (reinterpret_cast<uintptr_t>(sorted_bf16) & 15)) return AURA_E_ALIGN;
    hipLaunchKernelGGL(ivf2_append_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), bank, inv_norm, meta, slots, n, D, sorted_bf16, sorted_rows,
                       pad_off, list_len, pos_of_row, rho, flag, reinterpret_cast<float4*>(row_constants),
                       row_constants_now);
    return check_launch_b();
}

int64_t aura_kmeans_means_workspace_bytes(int64_t N, int64_t D, int k) {
    if (N < 0 || D <= 0 || k <= 0 || k > 256) return -1;
    return ((N + KM_SEG - 1) / KM_SEG + k + 1) * D * 4;
}

int aura_kmeans_segment_means(const float* bank, const int32_t* order, const int32_t* seg_off, float* centroids,
                              void* workspace, int64_t workspace_bytes, int64_t N, int64_t D, int k,
                              int sums_only, void* stream) {
    if (N < 0 || N > 0x7ffffff0LL || D <= 0 || (D & 3) || k <= 0 || k > 256) return AURA_E_INVAL;
    if (N == 0) return AURA_OK;
    if (!bank || !order || !seg_off || !centroids || !workspace) return AURA_E_INVAL;
    if ((reinterpret_cast<uintptr_t>(bank) & 15) || (reinterpret_cast<uintptr_t>(centroids) & 15) ||
        (reinterpret_cast<uintptr_t>(workspace) & 15))
        return AURA_E_ALIGN;
    if (workspace_bytes < aura_kmeans_means_workspace_bytes(N, D, k)) return AURA_E_INVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t max_items = (N + KM_SEG - 1) / KM_SEG + k;
    int64_t blocks = (max_items + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(kmeans_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, s, bank, order, seg_off,
                       static_cast<float*>(workspace), D, k);
    int rc = check_launch_b();
    if (rc) return rc;
    const int64_t waves = (int64_t)k * ((D + 255) / 256);
    hipLaunchKernelGGL(kmeans_reduce_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s,
                       static_cast<const float*>(workspace), seg_off, centroids, D, k, sums_only);
    return check_launch_b();
}

int aura_kmeans_commit(const int32_t* assign, const int32_t* seg_off, float* meta, float* counts, int64_t N,
                       int k, void* stream) {
    if (N < 0 || k <= 0 || k > 256) return AURA_E_INVAL;
    if (!assign || !seg_off || !meta) return AURA_E_INVAL;
    const int64_t blocks = N > 0 ? (N + 255) / 256 : 1;
    hipLaunchKernelGGL(kmeans_commit_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       assign, seg_off, meta, counts, N, k);
    return check_launch_b();
}

}  // extern "C"
