// aura_rowc.inl -- per-row score constants of the two-stage recall, shared by aura_knn.hip (prep kernels,
// the cached constants of the inverted lists) and aura_bank.hip (aura_ivf2_append keeps a cached table
// current).  See aura_knn_coarse.inl's header for the bound these constants implement.
#pragma once

// E_fix = 2 D 2^-24 + 1e-5: fp32 accumulation of D terms in either pipe + association of the score formulas
__host__ __device__ __forceinline__ float aura_e_fix(float D) { return 2.0f * D * 5.9604645e-8f + 1e-5f; }

// EQ_WORST bounds the query part of the error for any query (round-to-nearest: ||e_q|| <= 2^-8 ||q_hat||)
__host__ __device__ __forceinline__ float coarse_eq_worst(float D) {
    return (1.001f * 0.00390625f * 1.00001f + (0.5f * D + 3.0f) * 5.9604645e-8f) * 1.0078125f;
}

// Per-row score constants {A, B_up, B_lo, w} of the two-stage path (w: centroid id / bank row id).
//   rho != NULL (normalised bf16 shadow rows): A = 0.5 strength, the row's error part from rho[row];
//   rho == NULL (fp32 rows rounded on the fly): A = 0.5 strength / ||row||, worst-case error e_worst.
__device__ __forceinline__ float4 coarse_row_constants(const float4 m, float inv_norm_row, const float* rho_row,
                                                       float now, float e_fix, float e_worst, float eq_worst,
                                                       float w) {
    const float strength = m.x;
    const float tw = 0.2f * expf(-(now - m.y) / 3600.0f);
    float A, err;
    if (rho_row) {
        A = 0.5f * strength;
        err = 0.5f * fabsf(strength) * (*rho_row + e_fix);
        if (strength < 0.0f) err += fabsf(strength) * eq_worst;     // 2 |A| eq_worst
    } else {
        A = 0.5f * inv_norm_row * strength;
        err = 0.5f * e_worst * fabsf(strength);
    }
    return make_float4(A, tw * strength + err, tw * strength - err, w);
}

// constants of a sorted row of the inverted lists: .w = the bank row id's bits; padding / holes can never
// reach a threshold
__device__ __forceinline__ float4 ivf2_row_constants(int32_t row, const float* __restrict__ meta,
                                                     const float* __restrict__ rho, float now, float D) {
    if (row < 0) return make_float4(0.0f, -INFINITY, -INFINITY, __int_as_float(-1));
    const float4 m = *reinterpret_cast<const float4*>(meta + (int64_t)row * 4);
    return coarse_row_constants(m, 0.0f, rho + row, now, aura_e_fix(D), 0.0f, coarse_eq_worst(D), __int_as_float(row));
}
