// aura_knn_coarse.inl -- two-stage exact recall: bf16 matrix-core prefilter + fp32 re-scoring.
// Included by aura_knn.hip (same translation unit: shares ord_key, CNT_STRIDE, the workspace).
//
// Why: the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate) bounds the exact scan
// at ~0.35 ms for 256 queries x 100 k rows.  A scan in bf16 is bound by reading the bank once from
// HBM instead.  Results stay those of the fp32 path because the bf16 score is only used as a
// bound.  With q_hat, r_hat the normalised query / row and b_q = bf16(q_hat), b_r = bf16(r_hat):
//     b_q.b_r - q_hat.r_hat = q_hat.e_r + e_q.r_hat + e_q.e_r,   e_x = b_x - x_hat
//     |cos_bf16 - cos_fp32| <= rho_r + rho_q + rho_r rho_q + E_fix,   rho_x >= ||e_x||_2
// (Cauchy-Schwarz on unit vectors; E_fix = 2 D 2^-24 + 1e-5 covers the fp32 accumulation of D
// terms in either pipe and the association of the score formulas).  Round-to-nearest bf16 has
// unit roundoff 2^-8, so the worst case is rho = 2^-8 each: E_worst = 2^-7 (1 + 2^-9) + E_fix.
//   * bf16-row kernels (SRC16): the shadow holds the NORMALISED rows and aura_bank.hip records
//     every row's actual rho_r beside it (typically 0.0017 at D = 768); rho_q is measured per
//     query by the prep kernel -> a rigorous bound about half the worst case.  eq = rho_q (1 + 2^-7)
//     >= rho_q (1 + rho_r) enters the scan as the accumulators' initial value (+eq for U, -eq for
//     L), the row's part (rho_r + E_fix) through its constants:
//         U = A (t + eq) + B_up,  L = A (t - eq) + B_lo,  A = 0.5 strength,
//         B = 0.2 exp(-(now - ts)/3600) strength +- 0.5 |strength| (rho_r + E_fix)
//     (rows with a negative strength get 2 |A| eq_worst on top: A (t + eq) then under-states U).
//   * fp32-row kernels (no shadow): rows are rounded on the fly, nothing is known about their
//     residual: E_worst for every pair (A = 0.5 strength / ||row||, accumulators start at 0).
// So  L <= exact score <= U  for every (query, row) pair:
//   1. coarse_scan<SAMPLE>: group maxima of L over a strided sample of 16-row groups;
//      sample_threshold_kernel: T = k-th largest group maximum  (k distinct rows score >= T);
//   2. coarse_scan<FILTER> over every row: rows with U >= T go to the query's candidate list;
//   3. coarse_refine_kernel, one workgroup per query: T2 = k-th largest L of the candidates
//      (>= T, still a lower bound of the k-th best exact score); survivors U >= T2 are re-scored
//      with the fp32 path's own arithmetic (same fmaf order as the MFMA chain, same epilogue) and
//      the top k of those exact scores are returned, ties to the lower row -- bit-identical to
//      the fp32 scan whenever the candidate lists fit (else the overflow flag sends the caller
//      to the fp32 path).
//
// coarse_scan design (one 512-thread workgroup per CU, persistent over a contiguous span of
// 16-row tiles):
//   * queries are STATIONARY: wave w keeps its 32 queries as bf16 MFMA B-fragments in registers
//     for the whole launch (2 x KS x 4 VGPRs, KS = ceil(D/32) <= 24) -- no query traffic at all;
//   * bank rows stream HBM -> LDS as fp32 by global_load_lds (16 B per lane, full 128-B lines,
//     no VGPR staging) into a 3-slot ring: two tiles (2 x 48 KB) are in flight per CU while a
//     third is consumed; one s_barrier per tile, counted s_waitcnt vmcnt so the prefetch spans it;
//   * every wave reads the whole tile from LDS (conflict-free XOR-swizzled image, swizzle applied
//     on the global address), converts to bf16 in registers and issues 2 KS
//     v_mfma_f32_16x16x32_bf16 per tile;
//   * the steady-state loop has no ordinary global load and no returning atomic (hipcc would wait
//     vmcnt(0) for those and drain the prefetch): per-row constants arrive through the same
//     global_load_lds ring and candidates are buffered in LDS and flushed per span.
// Traffic: the bank once (N D 4 bytes) + 20 B of row constants per row.

#include "aura_rowc.inl"

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));

constexpr int CS_THREADS = 256;           // 4 waves (one per SIMD, 512 registers each), 64 queries per wave
constexpr int CS_QB = 4;                 // 16-query MFMA column blocks per wave
constexpr int CS_ROWS = 16;              // bank rows per tile
constexpr int CS_SLOTS = 3;
constexpr int IVF2_MAXBLK_C = 1024;      // >= IVF2_MAXBLK (aura_knn_ivf2.inl; checked there): blocks of one inverted-list pass at most
// bf16-row 8-wave kernels (not the probe-mask form, whose masks need the LDS) allocate twice the ring:
// their row-split form (blocks of at most 128 queries) streams two 16-row tiles per step
template <bool SRC16, bool MASKED, int NW> constexpr int cs_lds_slots() { return (SRC16 && NW == 8 && !MASKED) ? 2 * CS_SLOTS : CS_SLOTS; }
constexpr int CS_AUX_BYTES = 256;        // per slot: row constants (16 rows x 16 B, one 4-byte LDS-DMA per lane)
constexpr int CS_BUF = 1024;             // candidate entries buffered per workgroup (two halves of 512)
constexpr int CS_FLUSH_MIN = 128;        // a stable half is written out once it holds this many
constexpr int CS_MODE_SAMPLE = 0, CS_MODE_FILTER = 1;
constexpr int64_t COARSE_MIN_ROWS = 8192;
constexpr int COARSE_MAX_K = 256;
constexpr int RF_THREADS = 512;
constexpr int RF_CAP = 4096;             // candidates per query the refine kernel holds in LDS
constexpr int RF_SURV = 1024;            // survivors re-scored per query at most
// re-scoring geometry (template parameters of coarse_refine_kernel): survivors per wave and round x
// k-chunk.  <16, 256>: 133 KB of LDS, one workgroup per CU, 3 chunks -- passes with at most one query
// per CU.  <10, 192>: 66 KB, two workgroups per CU, 4 chunks -- larger passes (e.g. the row-sharded
// multi-GPU layout, where a rank refines every rank's queries).  Survivors are dealt round-robin so
// the usual ~70 keep all 8 waves busy in either geometry.

struct CoarseArgs {
    const float* bank;
    const uint16_t* bank16;  // optional bf16 shadow of the bank (SRC16 kernels), row-major [N][D]
    const float4* rowc;      // [N] per-row score constants {A, B_up, B_lo, centroid id} (coarse_prep_kernel)
    const uint16_t* qhat;    // bf16 query fragments, [nq/256][4 waves][4 blocks][KS][64 lanes][8]
    const float* inv_q;      // [nq]
    const float* eq;         // [nq] (IVF: [block slot]) query part of the error bound; SRC16 kernels only
    int64_t N, D;
    int nq;
    int64_t n_tiles;         // 16-row tiles this launch walks (per 256-query block)
    int tile_step, n_sample; // SAMPLE: tile j -> logical 128-row tile (j/8)*tile_step, sub-tile j%8
    float* gmax;             // SAMPLE out: [nq][gmax_ld], group = tile j >> gshift
    int64_t gmax_ld;
    int gshift;              // full scan, large banks: 3 = a group is a whole 128-row logical tile; the eight 16-row
                             // tiles max-combine their ORDERED KEYS (atomic, gmax zeroed by the caller)
    const uint32_t* thr;     // FILTER in: [nq] ordered keys
    int32_t* cnt;            // [nq][CNT_STRIDE]
    float* cand_scores;      // [nq][cap]: U
    int32_t* cand_idx;
    int cap;
    // optional centroid-candidate mask (MASKED kernels): bit c of probe_mask[q][8] set <=> query q
    // probes centroid c; a row is a candidate only if its centroid id (rowc.w) is probed
    // (src/core/hippocampal.py:259-270)
    const uint32_t* probe_mask;
    // IVF kernels (inverted lists over a LIST-SORTED bf16 shadow, every list padded to 16 rows): the
    // "query blocks" are (list, 256 probing queries) blocks, the rows of block B are the contiguous
    // sorted rows blk_row0[B] .. + 16 blk_tiles[B]; `thr` / `gmax` are indexed by block slot
    // (B * 256 + slot), slotq maps a slot to its query (-1: unused), rowc.w holds the ORIGINAL row
    // id's bits.  item_off[B] = first work item (tile) of block B, item_off[nblk] = total; all on
    // the device so the host never syncs.
    const int32_t* blk_row0;
    const int32_t* blk_stride; // SAMPLE: sample tile j of block B is the block's tile j * blk_stride[B]
    const int32_t* item_off;
    const int32_t* nblk;     // [1]
    const int32_t* slotq;    // [nblk * 256]
    const int32_t* blk_nq;   // [nblk] query slots in use in block B (the rest of its 256 are padding)
    int qzero;               // IVF: qhat holds ONE fragment set per QUERY ([q][KS][4 k-groups][8], written once per
                             // query whatever the number of lists it probes); a lane finds its slot's query through
                             // slotq, unused slots read the all-zero entry qzero
    int w_sparse, w_dense;   // IVF filter: item_off counts tiles x these weights (block of <= 128 / more queries)
    int dbg;                 // AURA_CS_DBG timing ablations (results invalid when non-zero)
};

__device__ __forceinline__ void glds16(const float* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const float* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

// MFMA with the stationary (query) operand in an accumulation register: 4 waves per workgroup own
// 512 registers each, but hipcc only places MFMA B operands in the 256 architectural VGPRs, so
// the 384 registers of query fragments are pinned by hand: the first CS_QA fragments live in
// AGPRs (the hardware reads srcB from either file), the rest in VGPRs.  Hazards the compiler
// would cover for its own MFMAs are covered by the s_nop statements around the groups.
#ifndef AURA_CS_PF16
#define AURA_CS_PF16 4
#endif
constexpr int CS_PF16 = AURA_CS_PF16;       // bf16-row kernels: fragment reads run this many k-steps ahead
constexpr int CS_QA8 = 30;  // 8-wave kernels (48 fragments, 8 accumulators per wave)
#ifndef AURA_CS_WFLUSH_DIV
#define AURA_CS_WFLUSH_DIV 2
#endif
constexpr int CS_WFLUSH_DIV = AURA_CS_WFLUSH_DIV;   // a wave writes a stable half out once it holds capacity / this
constexpr int CS_QA = 60;   // fragments (4 registers each) kept in AGPRs, next to the 16 accumulators
// LDS accesses of the steady-state loop that are NOT the MFMA fragments go through inline asm:
// hipcc puts "s_waitcnt vmcnt(0)" in front of an ordinary LDS access it cannot separate from an
// outstanding LDS-DMA (global_load_lds) -- which would drain the two-tile prefetch every tile.
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ void lds_read4x16(uint32_t addr, float4& r0, float4& r1, float4& r2, float4& r3) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\t"
                 "ds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "memory");
}
__device__ __forceinline__ void lds_read4x16_nowait(uint32_t addr, f32x4v& r0, f32x4v& r1, f32x4v& r2, f32x4v& r3) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\t"
                 "ds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "memory");
}
// fragment reads of the MFMA loop: issued without a wait, consumed behind a COUNTED wait (LDS
// operations return in order; n = LDS operations issued after the one being consumed).  hipcc's own
// placement was "s_waitcnt lgkmcnt(0)" right behind a freshly issued read every PF+1 k-steps: the
// full LDS latency, exposed, five times per tile.
template <int OFF>
__device__ __forceinline__ void lds_read16_nowait(f32x4v& r, uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
}
// (the consumed registers are tied to the wait so that no use -- and no copy -- can move above it;
// never pass the same object twice: the second operand would be a copy taken BEFORE the wait)
template <int N>
__device__ __forceinline__ void lds_wait_n(f32x4v& r0, f32x4v& r1) {
    static_assert(N >= 0 && N <= 15, "lgkmcnt field");
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait_n(f32x4v& r0) {
    static_assert(N >= 0 && N <= 15, "lgkmcnt field");
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r0) : "n"(N) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void cs_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cs_static_for<I + 1, N>(static_cast<F&&>(f));
    }
}
__device__ __forceinline__ void lds_wait4(f32x4v& r0, f32x4v& r1, f32x4v& r2, f32x4v& r3) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)::"memory");
}
__device__ __forceinline__ void lds_write_i32(uint32_t addr, int v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_read3(uint32_t addr, uint32_t& a0, uint32_t& a1, uint32_t& a2) {
    asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:4\n\tds_read_b32 %2, %3 offset:8\n\t"
                 "s_waitcnt lgkmcnt(0)" : "=&v"(a0), "=&v"(a1), "=&v"(a2) : "v"(addr) : "memory");
}
// returning global atomic add whose result is NOT waited for here (the caller counts vmcnt itself:
// an ordinary atomicAdd would make hipcc wait vmcnt(0) and drain the LDS-DMA prefetch)
__device__ __forceinline__ void gatomic_inc_nowait(int32_t* p, int& ret) {
    asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=&v"(ret) : "v"(p), "v"(1) : "memory");
}
__device__ __forceinline__ int lds_add_rtn(uint32_t addr, int v) {
    int r;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr), "v"(v) : "memory");
    return r;
}
__device__ __forceinline__ int lds_read_i32(uint32_t addr) {
    int r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ void lds_write3(uint32_t addr, uint32_t a0, uint32_t a1, uint32_t a2) {
    asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4\n\tds_write_b32 %0, %3 offset:8"
                 :: "v"(addr), "v"(a0), "v"(a1), "v"(a2) : "memory");
}

// XOR swizzles of the tile image (chunk slot = chunk ^ swz(row)), chosen for the lane groups in which
// gfx950 services a wave's ds_read_b128 -- {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59},
// {36-43,48-51,60-63} (MI355X_MICROARCH.md, LDS table), NOT four runs of 16 consecutive lanes: with the
// MFMA A layout (lane -> row lane&15, k-group lane>>4) a group holds rows 0-3 and 12-15 of one k-group
// and rows 4-11 of the next, and all 16 lanes must land in different 16-byte bank quads.
//   bf16 image, 4 chunks per row : swz = (4 - row/4) & 3          = 0,3,2,1 for rows 0-3,4-7,8-11,12-15
//   fp32 image, 8 chunks per row : swz = (m & 1) | (m & 4 ? 6 : 0), m = row/2   = 0,1,0,1,6,7,6,7
// (AURA_CS_EXP_OLDSWZ: the first version's row/4 and row/2, conflict-free only for consecutive lanes.)
__device__ __forceinline__ uint32_t cs_swz16(uint32_t row) {
#ifdef AURA_CS_EXP_OLDSWZ
    return (row >> 2) & 3;
#else
    return (4u - (row >> 2)) & 3u;
#endif
}
__device__ __forceinline__ uint32_t cs_swz32(uint32_t row) {
#ifdef AURA_CS_EXP_OLDSWZ
    return (row >> 1) & 7;
#else
    const uint32_t m = row >> 1;
    return (m & 1u) | ((m & 4u) ? 6u : 0u);
#endif
}

template <bool QA, bool LAST, bool PAD, bool COND = false>
__device__ __forceinline__ void mfma_bf16_q(f32x4v& acc, const bf16x8v& af, const bf16x8v& q, int on = 1) {
    // COND: the MFMA runs only when `on` (wave-uniform, in an SGPR) is non-zero -- a scalar compare and
    // branch inside the statement, so the compiler sees one definition of the accumulator either way.
    if constexpr (COND) {
#define AURA_SKIP_IN "s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 .Laura_skip%=\n\t"
#define AURA_SKIP_OUT "\n.Laura_skip%=:"
#define AURA_NOP1C "s_nop 1\n\t"
#define AURA_MFMAC "v_mfma_f32_16x16x32_bf16 "
        if constexpr (PAD) {
            if constexpr (QA && LAST)
                asm volatile(AURA_SKIP_IN AURA_NOP1C AURA_MFMAC "%0, %1, %2, %0\n\ts_nop 7\n\ts_nop 4" AURA_SKIP_OUT : "+a"(acc) : "v"(af), "a"(q), "s"(on) : "scc");
            else if constexpr (QA)
                asm volatile(AURA_SKIP_IN AURA_NOP1C AURA_MFMAC "%0, %1, %2, %0" AURA_SKIP_OUT : "+a"(acc) : "v"(af), "a"(q), "s"(on) : "scc");
            else if constexpr (LAST)
                asm volatile(AURA_SKIP_IN AURA_NOP1C AURA_MFMAC "%0, %1, %2, %0\n\ts_nop 7\n\ts_nop 4" AURA_SKIP_OUT : "+a"(acc) : "v"(af), "v"(q), "s"(on) : "scc");
            else
                asm volatile(AURA_SKIP_IN AURA_NOP1C AURA_MFMAC "%0, %1, %2, %0" AURA_SKIP_OUT : "+a"(acc) : "v"(af), "v"(q), "s"(on) : "scc");
        } else {
            if constexpr (QA && LAST)
                asm volatile(AURA_SKIP_IN AURA_MFMAC "%0, %1, %2, %0\n\ts_nop 7\n\ts_nop 4" AURA_SKIP_OUT : "+a"(acc) : "v"(af), "a"(q), "s"(on) : "scc");
            else if constexpr (QA)
                asm volatile(AURA_SKIP_IN AURA_MFMAC "%0, %1, %2, %0" AURA_SKIP_OUT : "+a"(acc) : "v"(af), "a"(q), "s"(on) : "scc");
            else if constexpr (LAST)
                asm volatile(AURA_SKIP_IN AURA_MFMAC "%0, %1, %2, %0\n\ts_nop 7\n\ts_nop 4" AURA_SKIP_OUT : "+a"(acc) : "v"(af), "v"(q), "s"(on) : "scc");
            else
                asm volatile(AURA_SKIP_IN AURA_MFMAC "%0, %1, %2, %0" AURA_SKIP_OUT : "+a"(acc) : "v"(af), "v"(q), "s"(on) : "scc");
        }
        return;
    }
    // PAD: s_nop 1 in front = the two wait states between a VALU write of an operand register and the
    // MFMA reading it (hipcc pads nothing inside inline asm).  The fp32-row kernels need it: their A
    // fragment comes out of v_cvt_pk_bf16_f32 right before.  The bf16-row kernels run without: the A
    // fragment comes from ds_read_b128 behind an s_waitcnt, the query fragments are never rewritten and
    // the accumulator set-up is several instructions away -- tools/check_mfma_hazards.py (a CPU test)
    // verifies exactly that on the generated code of every build (0.6 ns of the 8.2 ns per MFMA).
    // LAST (the tile's final k-step): 12 wait states behind the MFMA, INSIDE the statement, because
    // the compiler may read or move the accumulator right after it (it has no idea this is an MFMA).
    // AURA_CS_EXP_*: timing experiments only (tools/build_variant.sh; results are wrong with them)
#if defined(AURA_CS_EXP_NONOP)
#define AURA_NOP1 ""
#else
#define AURA_NOP1 "s_nop 1\n\t"
#endif
#ifdef AURA_CS_EXP_NOMFMA
#define AURA_MFMA "; "
#else
#define AURA_MFMA "v_mfma_f32_16x16x32_bf16 "
#endif
    if constexpr (PAD) {
        if constexpr (QA && LAST)
            asm volatile(AURA_NOP1 AURA_MFMA "%0, %1, %2, %0\n\ts_nop 7\n\ts_nop 4" : "+a"(acc) : "v"(af), "a"(q));
        else if constexpr (QA)
            asm volatile(AURA_NOP1 AURA_MFMA "%0, %1, %2, %0" : "+a"(acc) : "v"(af), "a"(q));
        else if constexpr (LAST)
            asm volatile(AURA_NOP1 AURA_MFMA "%0, %1, %2, %0\n\ts_nop 7\n\ts_nop 4" : "+a"(acc) : "v"(af), "v"(q));
        else
            asm volatile(AURA_NOP1 AURA_MFMA "%0, %1, %2, %0" : "+a"(acc) : "v"(af), "v"(q));
    } else {
        if constexpr (QA && LAST)
            asm volatile(AURA_MFMA "%0, %1, %2, %0\n\ts_nop 7\n\ts_nop 4" : "+a"(acc) : "v"(af), "a"(q));
        else if constexpr (QA)
            asm volatile(AURA_MFMA "%0, %1, %2, %0" : "+a"(acc) : "v"(af), "a"(q));
        else if constexpr (LAST)
            asm volatile(AURA_MFMA "%0, %1, %2, %0\n\ts_nop 7\n\ts_nop 4" : "+a"(acc) : "v"(af), "v"(q));
        else
            asm volatile(AURA_MFMA "%0, %1, %2, %0" : "+a"(acc) : "v"(af), "v"(q));
    }
}

// Query part of the error bound from the rounding residual e2 = sum (bf16(x) - x)^2 of the normalised
// query: eq = rho_q (1 + 2^-7), rho_q = 1.001 sqrt(e2) + (D/2 + 3) 2^-24 (see aura_bank.hip for the
// second term).  EQ_WORST bounds it for any query (round-to-nearest: ||e_q|| <= 2^-8 ||q_hat||).
__device__ __forceinline__ float coarse_eq_from_e2(float e2, float D) {
    return (1.001f * sqrtf(e2) + (0.5f * D + 3.0f) * 5.9604645e-8f) * 1.0078125f;
}
// (coarse_eq_worst, coarse_row_constants: aura_rowc.inl)

// Per-call preparation for the two-stage path, one launch:
//   blocks [0, qblocks): 4 queries each (one wave per query): 1/||q|| (as query_prep_kernel), the
//     query as bf16 MFMA B-fragments in the order coarse_scan_kernel's waves load them
//     ([256-query block][wave][16-query block][k-step][lane][8], zero padded: a fragment is one
//     coalesced 1-KiB wave load) and eq[q], the query's part of the error bound;
//   blocks [qblocks, ...): 256 rows each: the row's share of the combined score (see the header).
//     (Association differs from the exact epilogue by a few ulp: covered by the 1e-5 in E_fix.)
__global__ __launch_bounds__(256) void coarse_prep_kernel(const float* __restrict__ x, int64_t nq,
                                                          int64_t nq_pad, int64_t D, int KS,
                                                          uint16_t* __restrict__ qhat,
                                                          float* __restrict__ inv, float* __restrict__ eq_out,
                                                          int32_t* overflow,
                                                          int qblocks, const float* __restrict__ meta,
                                                          const float* __restrict__ inv_norm,
                                                          const float* __restrict__ rho,
                                                          int64_t N, float now, float e_fix, float e_worst,
                                                          float4* __restrict__ rowc) {
    if ((int)blockIdx.x >= qblocks) {
        const int64_t row = ((int64_t)blockIdx.x - qblocks) * 256 + threadIdx.x;
        if (row >= N) return;
        const float4 m = *reinterpret_cast<const float4*>(meta + row * 4);
        rowc[row] = coarse_row_constants(m, inv_norm[row], rho ? rho + row : nullptr, now, e_fix, e_worst,
                                         coarse_eq_worst((float)D), m.z);   // .w: centroid id
        return;
    }
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (overflow && blockIdx.x == 0 && threadIdx.x == 0) *overflow = 0;
    if (q >= nq_pad) return;
    const int64_t qblk = q >> 8;
    const int wq = (int)(q >> 6) & 3, b = (int)(q >> 4) & 3, lr = (int)q & 15;
    uint16_t* const base = qhat + ((((qblk * 4 + wq) * 4 + b) * KS) * 64 + lr) * 8;
    // 1/||q|| with query_prep_kernel's arithmetic (the re-scored results must not depend on the path)
    float s = 0.0f;
    if (q < nq)
        for (int64_t i = lane * 4; i < D; i += 256) {
            const float4 u = *reinterpret_cast<const float4*>(x + q * D + i);
            s = fmaf(u.x, u.x, s); s = fmaf(u.y, u.y, s); s = fmaf(u.z, u.z, s); s = fmaf(u.w, u.w, s);
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float iqv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    if (lane == 0 && q < nq) inv[q] = iqv;
    // fragments hold the NORMALISED query: the scan's accumulator is the cosine numerator / ||q||
    float e2 = 0.0f;
    for (int c = lane; c < KS * 4; c += 64) {             // chunk c: k = 8c .. 8c+7 = k-step c/4, lg c%4
        f32x8v v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.0f;
        const int64_t k0 = 8 * (int64_t)c;
        if (q < nq && k0 < D) {
            const float4 u = *reinterpret_cast<const float4*>(x + q * D + k0);      // D % 4 == 0
            v[0] = u.x * iqv; v[1] = u.y * iqv; v[2] = u.z * iqv; v[3] = u.w * iqv;
            if (k0 + 4 < D) {
                const float4 w = *reinterpret_cast<const float4*>(x + q * D + k0 + 4);
                v[4] = w.x * iqv; v[5] = w.y * iqv; v[6] = w.z * iqv; v[7] = w.w * iqv;
            }
        }
        const bf16x8v bv = __builtin_convertvector(v, bf16x8v);
        *reinterpret_cast<bf16x8v*>(base + ((int64_t)(c >> 2) * 64 + (c & 3) * 16) * 8) = bv;
        const f32x8v back = __builtin_convertvector(bv, f32x8v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = back[e] - v[e]; e2 = fmaf(d, d, e2); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off);
    if (lane == 0 && eq_out) eq_out[q] = q < nq ? coarse_eq_from_e2(e2, (float)D) : 0.0f;
}

// T[q] = k-th largest of the G group maxima: one wave per query, PER keys per lane in registers;
// the key is built bit by bit from the top (largest T with count(keys >= T) >= k).  Also zeroes
// the candidate counters.
template <int PER>
__global__ __launch_bounds__(256) void coarse_threshold_kernel(const float* __restrict__ gmax,
                                                               int64_t gmax_ld, int G, int k, int nq, int keys,   // keys: gmax holds ordered keys
                                                               uint32_t* __restrict__ thr_out,
                                                               int32_t* __restrict__ cnt_out) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    uint32_t key[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int g = i * 64 + lane;
        const float v = g < G ? gmax[(int64_t)q * gmax_ld + g] : 0.0f;
        key[i] = g < G ? (keys ? __float_as_uint(v) : ord_key(v)) : 0u;
    }
    uint32_t T = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = T | (1u << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) c += __popcll(__ballot(key[i] >= cand));
        if (c >= k) T = cand;
    }
    if (lane == 0) {
        thr_out[q] = T;
        cnt_out[(int64_t)q * CNT_STRIDE] = 0;
    }
}

// SRC16 = rows come from the bf16 shadow of the bank (half the HBM bytes, half the LDS-DMA writes,
// one ds_read_b128 per k-step and no convert); otherwise from the fp32 bank.
// MASKED = centroid-candidate mode: the 256 queries' probe masks (8 KB) sit in LDS behind the candidate
// buffer (only the bf16-row variant has the room) and gate every (query, row) pair in the epilogue.
// IVF = inverted-list mode (see CoarseArgs): same scan, the work items are (block, tile) pairs.
// NW = waves per workgroup: 4 (one per SIMD, 64 queries and 512 registers each) or, bf16 rows only,
// 8 (two per SIMD, 32 queries and 256 registers each: the second wave fills the first one's stalls).
// QBT = 16-query column blocks per wave.  NW = 4 with QBT = 2 (round 3, inverted lists only): a workgroup of four
// waves holds 128 query slots in 256 registers per wave and half the LDS -- TWO independent workgroups per CU, so
// that one's MFMA phase overlaps the other's wait / issue / epilogue without a barrier between them.
// Per-wave phase timers of the filter launch (AURA_CS_DBG bit 64) exist only in builds with -DAURA_CS_TIMERS=1
// (tools/build_variant.sh): s_memrealtime is a scalar-memory instruction, and with it in the tile loop hipcc puts
// "s_waitcnt lgkmcnt(0)" wherever its destination registers are reused -- which also waits for the loop's
// inline-asm LDS reads, at a place that moves with every change of the register allocation (measured round 3:
// the same source +-8 % on the launch).
#ifndef AURA_CS_TIMERS
#define AURA_CS_TIMERS 0
#endif
constexpr bool CS_TIMERS = AURA_CS_TIMERS != 0;

template <int NW, int QBT> constexpr int cs_min_waves() { return (NW == 4 && QBT == 2) ? 2 : 1; }
template <int NW, int QBT> constexpr int cs_cand_buf() { return (NW == 4 && QBT == 2) ? 512 : CS_BUF; }
template <int KS, int MODE, bool SRC16, bool MASKED, bool IVF = false, int NW = 4, int QBT = 16 / NW>
__global__ __launch_bounds__(64 * NW, (cs_min_waves<NW, QBT>())) void coarse_scan_kernel(const CoarseArgs a) {
    static_assert(!MASKED || SRC16, "the probe masks need the LDS the bf16 rows leave free");
    static_assert(!IVF || (SRC16 && !MASKED), "inverted lists run over the sorted bf16 shadow");
    static_assert(NW == 4 || (NW == 8 && SRC16), "8 waves: bf16 rows only");
    static_assert(KS % NW == 0, "pieces are dealt over the waves");
    constexpr int THREADS = 64 * NW;
    constexpr int QB = QBT;                                // 16-query MFMA column blocks per wave
    constexpr int BLKQ = NW * 16 * QB;                     // query slots per block (256; 128 in the two-workgroup form)
    static_assert(BLKQ == 256 || (IVF && BLKQ == 128), "a query block is 256 slots (128: inverted lists, NW = 4, QBT = 2)");
    constexpr int CBUF = cs_cand_buf<NW, QBT>();           // candidate entries buffered per workgroup
    constexpr int QA = QB == 4 ? CS_QA : CS_QA8;           // fragments pinned in AGPRs
    constexpr int STEP_BYTES = SRC16 ? 1024 : 2048;        // one k-step (32 k) of 16 rows
    constexpr int TILE_BYTES = KS * STEP_BYTES;
    constexpr int SLOT_BYTES = TILE_BYTES + CS_AUX_BYTES;
    constexpr int NP = SRC16 ? KS / NW : KS / 2;           // bank pieces (1 KiB each) per wave and tile
    constexpr int GL = NP + 1;                             // global_load_lds per wave and tile
    constexpr bool SP32 = SRC16 && NW == 8 && !MASKED;     // has the row-split form (see run_segment)
    constexpr int LSLOTS = cs_lds_slots<SRC16, MASKED, NW>();   // 16-row slots in LDS
    constexpr int NSLOT = CS_SLOTS;                        // ring slots
    extern __shared__ __attribute__((aligned(16))) char csmem[];
    // candidate buffer [NW][2][WCAP][3] follows the slots (addressed through buf_addr)

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, lr = lane & 15, lg = lane >> 4;
    const uint32_t D = (uint32_t)a.D;

    const int64_t nqblk = (a.nq + 255) / 256;
    const int ivf_nblk = IVF ? a.nblk[0] : 0;
    const int64_t total = IVF ? (int64_t)a.item_off[ivf_nblk] : a.n_tiles * nqblk;
    int64_t lo = total * (int64_t)blockIdx.x / gridDim.x;
    int64_t hi = total * ((int64_t)blockIdx.x + 1) / gridDim.x;
    if (IVF && MODE == CS_MODE_SAMPLE && ivf_nblk <= (int)gridDim.x && !(a.dbg & 32768)) {
        // Sample pass over inverted lists: every block has the same number of sample tiles, and a workgroup's
        // set-up for a block (slot tables, 48 query fragments per lane, first tiles: three dependent latencies,
        // ~10 us) is as long as four of its eight steps -- with no more blocks than workgroups each workgroup takes
        // ONE whole block instead of the two halves an even split of the tiles gives it.
        const int g = (int)blockIdx.x;
        lo = g < ivf_nblk ? (int64_t)a.item_off[g] : 0;
        hi = g < ivf_nblk ? (int64_t)a.item_off[g + 1] : 0;
    }

    // Reader offsets inside a k-step block (lane = row lr, k-group lg), XOR-swizzled so that the 16
    // lanes of a ds_read_b128 phase hit 16 different bank quads:
    //   fp32 image: 8 chunks of 16 B per row (128 B); the lane reads chunks 2 lg and 2 lg + 1
    //   bf16 image: 4 chunks of 16 B per row ( 64 B); the lane reads chunk lg
    const int sw = SRC16 ? cs_swz16(lr) : cs_swz32(lr);
    const int off0 = SRC16 ? (4 * lr + (lg ^ sw)) * 16 : (8 * lr + ((2 * lg) ^ sw)) * 16;
    const int off1 = SRC16 ? 0 : (8 * lr + ((2 * lg + 1) ^ sw)) * 16;
    // Loader roles.  The image is swizzled by choosing WHICH 16-byte chunk a lane fetches (LDS-DMA
    // writes lane-linear).  A wave's pieces are 256 bytes apart in the row either way, so one offset
    // register serves them all.  Columns beyond D re-read the row's last 16 bytes: finite data times
    // the queries' zero padding (ragged D only; clamped per piece).
    //   fp32: piece m = wave + 4 i -> k-step m>>1, row half m&1 (fixed per wave); lane -> row
    //         8 half + lane/8, chunk (lane&7) ^ swizzle of the 128-byte line
    //   bf16: piece m = wave + 4 i -> k-step m; lane -> row lane/4, chunk (lane&3) ^ swizzle of the
    //         64-byte half line
    constexpr uint32_t ESZ = SRC16 ? 2 : 4, ECH = SRC16 ? 8 : 4;    // element bytes, elements per chunk
    constexpr int PSTEP = SRC16 ? 64 * NW : 256;           // bytes between a wave's pieces inside a row
    const bool full_k = D == (uint32_t)(KS * 32);
    const char* const src = SRC16 ? reinterpret_cast<const char*>(a.bank16) : reinterpret_cast<const char*>(a.bank);

    int64_t ivf_row0 = 0;                                  // IVF: first sorted row of the current block
    int64_t ivf_step = 16;                                 // IVF: rows between consecutive work tiles
    auto tile_row0 = [&](int64_t j) -> int64_t {
        if (IVF) return ivf_row0 + j * ivf_step;
        if (MODE == CS_MODE_SAMPLE) return ((j >> 3) * a.tile_step) * 128 + (j & 7) * 16;
        return j * 16;
    };
    auto issue = [&](int64_t j, int slot) {
        char* const sb = csmem + slot * SLOT_BYTES +
                         (SRC16 ? wave * 1024 : (wave & 1) * 1024 + (wave >> 1) * 2048);
        // 8-wave kernels recompute the lane's loader role per call (256 registers per wave: values kept
        // live across the fragment loads get spilled, and a scratch reload waits for vmcnt(0), which
        // drains the prefetch)
        uint32_t ln = (uint32_t)lane;
        if (NW == 8) asm volatile("" : "+v"(ln));
        const uint32_t ld_rho = SRC16 ? ln >> 2 : 8 * (wave & 1) + (ln >> 3);
        const uint32_t ld_cch = SRC16 ? (ln & 3) ^ cs_swz16(ld_rho) : (ln & 7) ^ cs_swz32(ld_rho);
        const uint32_t ld_kf0 = (SRC16 ? 32 * wave : 32 * (wave >> 1)) + ECH * ld_cch;   // elements; piece i: + PSTEP B
        const uint32_t voff0 = (ld_rho * D + ld_kf0) * ESZ;             // bytes from the tile's first row
        const int64_t r0 = tile_row0(j);
        const char* base = src + r0 * (int64_t)D * ESZ;
        if (r0 + CS_ROWS > a.N) {                          // last, partial tile: clamp the row
            const uint32_t last = (uint32_t)(a.N - 1 - r0);
            if (ld_rho > last) base -= (int64_t)(ld_rho - last) * D * ESZ;
        }
        if (full_k) {
#pragma unroll
            for (int i = 0; i < NP; ++i)
                glds16(reinterpret_cast<const float*>(base + voff0 + PSTEP * i), sb + i * (1024 * NW));
        } else {
            uint32_t kf0 = ld_kf0;
            asm volatile("" : "+v"(kf0));                  // recompute per call: hoisting these
#pragma unroll                                             // addresses out of the tile loop spills
            for (int i = 0; i < NP; ++i) {
                uint32_t kf = kf0 + (PSTEP / ESZ) * i;
                if (kf >= D) kf = D - ECH;
                glds16(reinterpret_cast<const float*>(base + (ld_rho * D + kf) * ESZ), sb + i * (1024 * NW));
            }
        }
        if (wave == 0) {   // row constants: one piece per tile, from wave 0 (its vmcnt waits count one more)
            int64_t row = r0 + (lane >> 2);                // 16 rows x 16 B: lane l carries dword l & 3 of row l >> 2
            if (row >= a.N) row = a.N - 1;
            glds4(reinterpret_cast<const float*>(a.rowc + row) + (lane & 3), csmem + slot * SLOT_BYTES + TILE_BYTES);
        }
    };

    // Candidate buffer: every wave owns a region of two halves of WCAP entries (q, row, U) and keeps
    // the halves' fill counts in scalar registers -- no LDS atomics, no shared counters, no cross-wave
    // ordering.  Tile t appends to half t&1; the other half is stable during tile t and is written out
    // (by its own wave) once it holds WFLUSH entries; entries beyond WCAP go straight to the lists.
    constexpr int WCAP = CBUF / (2 * NW);                  // 64 (8 waves; 4 waves x 32 queries) or 128 (4 waves x 64 queries)
    constexpr int EPL = WCAP / 64;                         // entries per lane and half in a write-out
    constexpr int WFLUSH = WCAP / CS_WFLUSH_DIV;
    const uint32_t cs_base = lds_addr(csmem);
    const uint32_t buf_addr = cs_base + LSLOTS * SLOT_BYTES;
    const uint32_t wreg_addr = buf_addr + wave * (2 * WCAP * 12);
    const uint32_t mask_addr = buf_addr + CBUF * 12;              // [256][8] probe masks (MASKED) / [BLKQ] slot -> query (IVF)
    // [256] query parts of the error bound (SRC16), signed for the mode: kept in LDS, not in registers --
    // two more live VGPRs across the tile loop made the 24-k-step 8-wave kernels spill inside it
    constexpr int EQ_OFF = CBUF * 12 + (MASKED ? 256 * 32 : (IVF ? BLKQ * 4 : 0));
    const uint32_t eq_addr = buf_addr + EQ_OFF;
    int wc[2] = {0, 0};                                    // fill counts of this wave's halves (wave-uniform)
    // span-end write-out of both halves: every slot reservation is issued before any is waited for
    // (up to 4 entries per lane), then one wait, then the stores
    auto flush_all = [&]() {
        int pos[4] = {-1, -1, -1, -1};
        uint32_t eq[4], er[4], eu[4];
#pragma unroll
        for (int u = 0; u < 2 * EPL; ++u) {
            const int i = lane + (u % EPL) * 64, h = u / EPL;
            const int nh = wc[h] < WCAP ? wc[h] : WCAP;
            pos[u] = -1;
            if (i < nh) {
                lds_read3(wreg_addr + (h * WCAP + i) * 12, eq[u], er[u], eu[u]);
                gatomic_inc_nowait(a.cnt + (int64_t)eq[u] * CNT_STRIDE, pos[u]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pos[0]), "+v"(pos[1]), "+v"(pos[2]), "+v"(pos[3])::"memory");
#pragma unroll
        for (int u = 0; u < 2 * EPL; ++u) {
            const int i = lane + (u % EPL) * 64, h = u / EPL;
            const int nh = wc[h] < WCAP ? wc[h] : WCAP;
            if (i < nh && pos[u] < a.cap) {
                a.cand_scores[(int64_t)eq[u] * a.cap + pos[u]] = __uint_as_float(eu[u]);
                a.cand_idx[(int64_t)eq[u] * a.cap + pos[u]] = (int32_t)er[u];
            }
        }
        wc[0] = 0; wc[1] = 0;
    };

    const uint32_t t_kernel0 = (CS_TIMERS && MODE == CS_MODE_FILTER && (a.dbg & 64)) ? (uint32_t)__builtin_amdgcn_s_memrealtime() : 0u;
    int64_t c = lo;
    int blk_cur = -1;                                      // IVF: block of the previous segment
    while (c < hi) {
        const uint32_t t_seg0 = (CS_TIMERS && MODE == CS_MODE_FILTER && (a.dbg & 64)) ? (uint32_t)__builtin_amdgcn_s_memrealtime() : 0u;
        int64_t qblk, j0, seg;
        if (IVF) {
            // The span [lo, hi) is in work units: item_off counts tiles x the block's weight (filter
            // scan; 1 otherwise).  Tile j of block B belongs to the workgroup whose span holds
            // item_off[B] + j * weight.
            int blo;
            if (blk_cur < 0) {                              // block B with item_off[B] <= c < item_off[B+1]:
                // B = number of prefixes item_off[1 .. nblk-1] that are <= c (the prefixes never decrease);
                // five loads in flight and five ballots instead of a chain of nine dependent loads
                int n_le = 0;
#pragma unroll
                for (int i = 0; i < (IVF2_MAXBLK_C + 63) / 64; ++i) {
                    const int idx = i * 64 + lane;
                    const bool le = idx >= 1 && idx < ivf_nblk && (int64_t)a.item_off[idx] <= c;
                    n_le += (int)__popcll(__ballot(le));
                }
                blo = __builtin_amdgcn_readfirstlane(n_le);
            } else {
                blo = blk_cur + 1;                          // the previous segment ran to the end of its block
            }
            blk_cur = blo;
            qblk = blo;
            const int64_t base = a.item_off[blo], next = a.item_off[blo + 1];
            const int64_t wgt = (MODE == CS_MODE_FILTER) ? (a.blk_nq[blo] <= 128 ? a.w_sparse : a.w_dense) : 1;
            const int64_t c_end = hi < next ? hi : next;
            j0 = (c - base + wgt - 1) / wgt;
            seg = (c_end - base + wgt - 1) / wgt - j0;
            c = c_end;
            if (seg <= 0) continue;                         // (workgroup-uniform)
            ivf_row0 = a.blk_row0[blo];
            ivf_step = MODE == CS_MODE_SAMPLE ? 16 * (int64_t)a.blk_stride[blo] : 16;   // sample: spread over the list
        } else {
            qblk = c / a.n_tiles; j0 = c - qblk * a.n_tiles;
            seg = (hi - c) < (a.n_tiles - j0) ? (hi - c) : (a.n_tiles - j0);
            c += seg;
        }
        // Query slots in use in this block.  The interval between two barriers is ONE wave's serial work
        // on a tile (DMA issue, MFMAs, epilogue), and a block of at most 128 queries -- the common case at 8
        // probes of 256 lists -- leaves half the waves without a query.  Such a block runs the ROW-SPLIT
        // form of the loop (run_segment<.., 2>): a step streams TWO 16-row tiles (the ring holds 3 x 2
        // slots), waves w and w + 4 hold the same 32 query slots, and wave w + 4 works on the second tile.
        // One barrier, one wait and one round of DMA issue then serve 32 rows, and twice the bytes are in
        // flight per CU.  Waves without a query (or, in an odd last step, without a tile) skip MFMAs and
        // epilogue; they still issue their share of the LDS-DMA and meet the barrier.
        const int n_used = IVF ? a.blk_nq[qblk] : (a.nq - (int)qblk * 256);

        auto run_segment = [&](auto NB_, auto RS_) {
        constexpr int NBc = decltype(NB_)::value;           // column blocks (16 queries) per wave
        // RS: 1 = one 16-row tile per step; 2 = row-split form (two tiles per step, waves w and w + 4 share
        // the query slots and take one tile each); 3 = two tiles per step, every wave works on both (blocks
        // of more than 128 queries in the kernels with the doubled ring: one barrier, one wait and one burst
        // of DMA issue per 32 rows instead of per 16)
        constexpr int RS = decltype(RS_)::value;
        constexpr bool SPLIT = RS == 2;
        constexpr int TPS = RS == 1 ? 1 : 2;                // tiles per step
        const int qg = SPLIT ? (wave & 3) : wave;           // query group: slots [qg * 16 QB, + 16 QB) of the block
        const int rh = SPLIT ? (wave >> 2) : 0;             // row-split: which tile of the step this wave works on
        const int qoff = (int)qblk * BLKQ + qg * (16 * QB); // this wave's first query (IVF: block slot)
        int my_cnt = n_used - qg * (16 * QB);
        my_cnt = my_cnt < 0 ? 0 : (my_cnt > 16 * NBc ? 16 * NBc : my_cnt);
        // column blocks this wave runs (wave-uniform, in a scalar register)
        const int nb = __builtin_amdgcn_readfirstlane(NBc == 2 ? (my_cnt + 15) / 16 : (my_cnt > 0 ? NBc : 0));
        const bool wave_active = nb > 0;
        // ---- stationary operand: the (normalised) queries of this wave as bf16 B-fragments ----
        // one coalesced 16-byte load per lane and fragment, all in flight at once, landing in
        // their final registers
        bf16x8v qf[NBc][KS];
        float thrf[NBc];
        int qid[NBc];                                       // IVF filter: the queries in this lane's slots (16 b + lr)
        if (MASKED) {                                       // this block's probe masks -> LDS
            uint32_t* const s_mask = reinterpret_cast<uint32_t*>(csmem + LSLOTS * SLOT_BYTES + CBUF * 12);
            for (int i = tid; i < 256 * 8; i += THREADS) {
                const int64_t q = qblk * 256 + (i >> 3);
                s_mask[i] = q < a.nq ? a.probe_mask[q * 8 + (i & 7)] : 0u;
            }
            __syncthreads();
        }
        // The block's slot tables -- the query part of the error bound (the accumulators start at +eq (FILTER: U) /
        // -eq (SAMPLE: L)), the thresholds, the slots' queries -- are three independent loads: issued together and
        // waited for once (one global latency in front of the fragment loads instead of three; the fragment
        // addresses need the queries, and the thresholds' wait must not sit between the fragment loads).
        float eqv = 0.0f;
        uint32_t key[NBc];
        if (SRC16 && tid < BLKQ) {
            const int64_t q = qblk * BLKQ + tid;
            eqv = a.eq[q < a.nq ? q : a.nq - 1];
        }
#pragma unroll
        for (int b = 0; b < NBc; ++b) {
            const int q = qoff + 16 * b + lr;
            key[b] = MODE == CS_MODE_FILTER ? a.thr[q < a.nq ? q : a.nq - 1] : 0u;
        }
#pragma unroll
        for (int b = 0; b < NBc; ++b) qid[b] = IVF ? a.slotq[qoff + 16 * b + lr] : 0;
        if (SRC16) {
            float* const s_eq = reinterpret_cast<float*>(csmem + LSLOTS * SLOT_BYTES + EQ_OFF);
            if (tid < BLKQ) s_eq[tid] = MODE == CS_MODE_FILTER ? eqv : -eqv;
            __syncthreads();
        }
#pragma unroll
        for (int b = 0; b < NBc; ++b)                       // padding columns: NaN, "U >= NaN" never holds (U may be +inf)
            thrf[b] = MODE == CS_MODE_FILTER ? (qoff + 16 * b + lr < a.nq ? ord_unkey(key[b]) : __builtin_nanf("")) : INFINITY;
#pragma unroll
        for (int b = 0; b < NBc; ++b) {                     // (a column block without a query is never multiplied)
            // full scan: fragments laid out by slot, a fragment is one coalesced 1-KiB wave load; inverted lists:
            // one fragment set per query, the lane's 16 bytes of k-step s sit at [query][s][lg] (64 B per query
            // and k-step, all KS steps of a query contiguous)
            const uint16_t* qp = IVF ? a.qhat + ((int64_t)(qid[b] >= 0 ? qid[b] : a.qzero) * (KS * 4) + lg) * 8
                                     : a.qhat + (((qblk * 16 + qg * QB + b) * KS) * 64 + lane) * 8;
            constexpr int QSTEP = IVF ? 32 : 512;           // uint16 elements between k-steps
            if (b < nb || NBc != 2) {
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    qf[b][s] = *reinterpret_cast<const bf16x8v*>(qp + (int64_t)s * QSTEP);
            } else {
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int e = 0; e < 8; ++e) qf[b][s][e] = 0;
            }
        }
        // the first two tiles start streaming while the fragments are still in flight: the single
        // wait in front of the first pin below then covers fragments and tiles alike
        // step T of the segment = tile j0 + T (RS 1) / tiles j0 + 2T and j0 + 2T + 1 (RS 2; an odd last
        // step streams its one tile twice, so that every step has the same number of pieces in flight)
        const int64_t n_steps = TPS == 2 ? (seg + 1) / 2 : seg;
        auto issue_step = [&](int64_t T, int ss) {
            if (TPS == 2) {
                issue(j0 + 2 * T, 2 * ss);
                issue(j0 + (2 * T + 1 < seg ? 2 * T + 1 : seg - 1), 2 * ss + 1);
            } else {
                issue(j0 + T, ss);
            }
        };
        issue_step(0, 0);
        if (n_steps > 1) issue_step(1, 1);
#pragma unroll
        for (int b = 0; b < NBc; ++b) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                // pin each fragment in its register file (see mfma_bf16_q)
                if (b * KS + s < QA) asm volatile("" : "=a"(qf[b][s]) : "0"(qf[b][s]));
                else asm volatile("" : "=v"(qf[b][s]) : "0"(qf[b][s]));
            }
        }
        // ordinary loads are all consumed here; until the span ends only LDS-DMA is in flight
        // (plus the rare flush)

        // (Measured dead end for the 8-wave kernels: waves 4-7 running one tile behind -- epilogue of
        // tile t-1 first, then tile t's MFMAs, so that the two waves of a SIMD are in opposite phases --
        // was 2-5 us SLOWER than letting both run in step: 65.6-68.8 vs 63.6 us at config 2.)
        f32x4v acc[NBc];
        f32x4v rcv[4];                                       // constants of rows 4 lg .. 4 lg + 3
        int slot = 0;
        const int64_t n_int = n_steps;
        // AURA_CS_DBG bit 64: per-wave phase times (100 MHz ticks) into a.gmax (FILTER launches only)
        const bool tm = CS_TIMERS && MODE == CS_MODE_FILTER && (a.dbg & 64) && a.gmax != nullptr;
        uint32_t tacc[6] = {0u, 0u, 0u, 0u, 0u, 0u};       // DMA wait, check+issue, segment set-up, mma (+ late issue), barrier, write-out + epilogue
        auto stamp = [&]() -> uint32_t { return tm ? (uint32_t)__builtin_amdgcn_s_memrealtime() : 0u; };
        tacc[2] = stamp() - t_seg0;
        for (int64_t t = 0; t < n_int; ++t) {
            const uint32_t ts0 = stamp();
            const int par = (int)(t & 1);
            // The half tile t-1 appended to is this wave's own and stable from here on.  Once it holds enough
            // entries it is written out in two steps that never stall the stream: the slot reservations
            // (returning atomics, a 1-2 us round trip) are issued HERE, in front of the wait for this step's
            // tiles and the barrier, and consumed after the MFMA loop with a counted wait: the barrier
            // wait, the DMA issue and the MFMA loop all run in the atomics' shadow (issued behind the
            // barrier they had the MFMA loop alone; measured gain of the move: 1 % of the launch).
            int fl_n = 0, fl_pos0 = 0, fl_pos1 = 0;
            if (MODE == CS_MODE_FILTER) {
                const int nbo = wc[par ^ 1];                 // (wave-uniform, in scalar registers)
                if (nbo >= WFLUSH && (a.dbg & 2048)) {       // timing experiment: drop the half instead of writing it out
                    wc[par ^ 1] = 0;
                } else if (nbo >= WFLUSH) {
                    fl_n = nbo < WCAP ? nbo : WCAP;
                    const uint32_t eb = wreg_addr + (par ^ 1) * (WCAP * 12);
                    if (lane < fl_n)
                        gatomic_inc_nowait(a.cnt + (int64_t)lds_read_i32(eb + lane * 12) * CNT_STRIDE, fl_pos0);
                    if (EPL > 1 && lane + 64 < fl_n)
                        gatomic_inc_nowait(a.cnt + (int64_t)lds_read_i32(eb + (lane + 64) * 12) * CNT_STRIDE,
                                           fl_pos1);
                }
            }
            // this tile's pieces have landed; the next tile's (NP per wave, + the row constants on wave 0)
            // and the reservation just issued (one instruction when EPL == 1) stay in flight
            if (t + 1 >= n_steps) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else if (EPL == 1 && fl_n > 0) {
                if (wave == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(TPS * GL + 1) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(TPS * NP + 1) : "memory");
            }
            else if (wave == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(TPS * GL) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(TPS * NP) : "memory");
            const uint32_t ts0b = stamp();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const uint32_t ts1 = stamp();
            // this wave's eq values: read here, consumed by the accumulator set-up behind the DMA issue
            uint32_t eqr[NBc];
            if (SRC16) {
                uint32_t ln2 = (uint32_t)lane;
                asm volatile("" : "+v"(ln2));                // recomputed per tile (see issue())
                const uint32_t ea = eq_addr + (uint32_t)(qg * (16 * QB)) * 4u + (ln2 & 15u) * 4u;
#pragma unroll
                for (int b = 0; b < NBc; ++b)
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(eqr[b]) : "v"(ea), "n"(64 * b) : "memory");
            }
            // Waves 4-7 of an 8-wave workgroup issue their share of step t + 2 BEHIND their MFMA loop (AURA_CS_DBG
            // bit 4096 puts it back in front): an LDS-DMA instruction blocks its wave until the texture addresser
            // takes it (0.4-0.8 us per step for the workgroup's 48 one-KiB requests); with the two waves of a SIMD
            // in opposite orders one of them multiplies while the other one is blocked (-2.5 % of the launch,
            // tools/ab_headline.py 0 4096).
            const bool issue_late = NW == 8 && wave >= 4 && !(a.dbg & 4096);
            if (!issue_late && t + 2 < n_steps && !(a.dbg & 4)) issue_step(t + 2, (slot + 2) % NSLOT);
            int cslot = 0;                                          // the 16-row slot this wave computes on
            int64_t ctile = 0;                                      // ... = tile j0 + ctile of the block
            bool work = false;
            auto set_tile = [&](int u) {                            // u-th tile of the step (RS 3); row-split: this wave's
                const int w = SPLIT ? rh : u;
                cslot = TPS == 2 ? 2 * slot + w : slot;
                ctile = TPS == 2 ? 2 * t + w : t;
                work = wave_active && ctile < seg;
            };
            set_tile(0);

            auto mma = [&]() {
            if (SRC16) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(eqr[0])::"memory");
#pragma unroll
                for (int b = 1; b < NBc; ++b) asm volatile("" : "+v"(eqr[b])::"memory");
            }
#pragma unroll
            for (int b = 0; b < NBc; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[b][e] = SRC16 ? __uint_as_float(eqr[b]) : 0.0f;
            // fragment reads run two k-steps ahead of the MFMAs (one wave per SIMD: nothing else
            // hides the LDS latency)
            // k-steps the reads run ahead (8 waves: registers are short, the other wave hides the latency)
            constexpr int PF = (SRC16 && NW == 4 && QB == 4) ? CS_PF16 : 2;
            f32x4v xr[PF + 1][2];
            constexpr int RP = SRC16 ? 1 : 2;                // LDS reads per k-step
            constexpr int S_RC = KS >= 3 ? KS - 3 : 0;       // the row constants are fetched behind this step
            // (8-wave kernels: the lane's reader offset is recomputed per call, like the loader role in issue() --
            //  kept live across the tile loop it was spilled, and its reload waits for vmcnt(0))
            uint32_t o0 = (uint32_t)off0, lgo = (uint32_t)lg * 64u;
            if (NW == 8 && SRC16) {
                uint32_t ln3 = (uint32_t)lane;
                asm volatile("" : "+v"(ln3));
                const uint32_t lr3 = ln3 & 15u, lg3 = ln3 >> 4;
                o0 = (4u * lr3 + (lg3 ^ (uint32_t)cs_swz16((int)lr3))) * 16u;
                lgo = lg3 * 64u;
            }
            const uint32_t a0 = cs_base + cslot * SLOT_BYTES + o0, a1 = cs_base + cslot * SLOT_BYTES + off1;
            if (!(a.dbg & 1)) {
            cs_static_for<0, (PF < KS ? PF : KS)>([&](auto S) {
                constexpr int s = decltype(S)::value;
                lds_read16_nowait<s * STEP_BYTES>(xr[s][0], a0);
                if constexpr (!SRC16) lds_read16_nowait<s * STEP_BYTES>(xr[s][1], a1);
            });
            cs_static_for<0, KS>([&](auto S) {
                constexpr int s = decltype(S)::value;
#ifndef AURA_CS_EXP_NOREAD
                if constexpr (s + PF < KS) {
                    lds_read16_nowait<(s + PF) * STEP_BYTES>(xr[(s + PF) % (PF + 1)][0], a0);
                    if constexpr (!SRC16) lds_read16_nowait<(s + PF) * STEP_BYTES>(xr[(s + PF) % (PF + 1)][1], a1);
                }
#endif
                // younger than this step's reads: the reads of the next min(PF, KS-1-s) steps and, behind
                // step S_RC, the four row-constant reads
                constexpr int YOUNGER = RP * ((KS - 1 - s) < PF ? (KS - 1 - s) : PF) + (s > S_RC ? 4 : 0);
                if constexpr (SRC16) lds_wait_n<YOUNGER>(xr[s % (PF + 1)][0]);
                else lds_wait_n<YOUNGER>(xr[s % (PF + 1)][0], xr[s % (PF + 1)][1]);
                bf16x8v af;
                if constexpr (SRC16) {
                    af = __builtin_bit_cast(bf16x8v, xr[s % (PF + 1)][0]);
                } else {
                    const f32x4v x0 = xr[s % (PF + 1)][0], x1 = xr[s % (PF + 1)][1];
                    f32x8v x;
                    x[0] = x0[0]; x[1] = x0[1]; x[2] = x0[2]; x[3] = x0[3];
                    x[4] = x1[0]; x[5] = x1[1]; x[6] = x1[2]; x[7] = x1[3];
                    af = __builtin_convertvector(x, bf16x8v);
                }
                cs_static_for<0, NBc>([&](auto Bc) {
                    constexpr int b = decltype(Bc)::value;
                    // the second column block of a wave runs only when it holds a query (the skip is a
                    // scalar branch inside the statement, so the accumulators keep one definition)
                    constexpr bool CND = NBc == 2 && b == 1;
                    constexpr bool INA = b * KS + s < QA;
                    mfma_bf16_q<INA, s == KS - 1, !SRC16, CND>(acc[b], af, qf[b][s], nb - 1);
                });
                // the epilogue's row constants are fetched behind the last two k-steps
                if constexpr (s == S_RC)
                    lds_read4x16_nowait(cs_base + cslot * SLOT_BYTES + TILE_BYTES + lgo, rcv[0], rcv[1], rcv[2], rcv[3]);
                __builtin_amdgcn_sched_barrier(0);
            });
            } else {
                lds_read4x16_nowait(cs_base + cslot * SLOT_BYTES + TILE_BYTES + lgo, rcv[0], rcv[1], rcv[2], rcv[3]);
            }
            lds_wait4(rcv[0], rcv[1], rcv[2], rcv[3]);
            __builtin_amdgcn_sched_barrier(0);
            };
            auto epi = [&](int64_t tt) {
            // ---- epilogue: rows 4 lg + e of the tile, queries qoff + 16 b + lr ----
            const int64_t r0 = tile_row0(j0 + tt);
            const int rows_left = (int)((a.N - r0) < CS_ROWS ? (a.N - r0) : CS_ROWS);
            float gm[NBc];
            float uv[4 * NBc];                                // U of the tile's pairs (kept out of the AGPR accumulators)
            unsigned bits = 0u;
            if (!(a.dbg & 2)) {
#pragma unroll
            for (int b = 0; b < NBc; ++b) gm[b] = -INFINITY;
            unsigned allow = 0xffffu;                        // bit 4 b + e: pair passes the probe mask
            if (MASKED) {
                // 16 mask words (4 rows x 4 query blocks) by inline-asm LDS reads, one wait
                uint32_t mw[4 * QB];
                int cidv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int cid = (int)rcv[e][3];
                    cidv[e] = (cid >= 0 && cid < 256) ? cid : -1;
#pragma unroll
                    for (int b = 0; b < QB; ++b)
                        asm volatile("ds_read_b32 %0, %1" : "=v"(mw[b * 4 + e])
                                     : "v"(mask_addr + (uint32_t)((qg * (16 * QB) + 16 * b + lr) * 32 + ((cidv[e] < 0 ? 0 : cidv[e]) >> 5) * 4))
                                     : "memory");
                }
                if constexpr (QB == 4)
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(mw[0]), "+v"(mw[1]), "+v"(mw[2]), "+v"(mw[3]), "+v"(mw[4]), "+v"(mw[5]),
                                   "+v"(mw[6]), "+v"(mw[7]), "+v"(mw[8]), "+v"(mw[9]), "+v"(mw[10]), "+v"(mw[11]),
                                   "+v"(mw[12]), "+v"(mw[13]), "+v"(mw[14]), "+v"(mw[15])::"memory");
                else
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(mw[0]), "+v"(mw[1]), "+v"(mw[2]), "+v"(mw[3]), "+v"(mw[4]), "+v"(mw[5]),
                                   "+v"(mw[6]), "+v"(mw[7])::"memory");
                allow = 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int b = 0; b < QB; ++b)
                        allow |= (cidv[e] >= 0 && ((mw[b * 4 + e] >> (cidv[e] & 31)) & 1u)) ? (1u << (b * 4 + e)) : 0u;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4v rc = rcv[e];
                const bool vrow = 4 * lg + e < rows_left;
#pragma unroll
                for (int b = 0; b < NBc; ++b) {
                    const bool ok = b < nb && vrow && ((allow >> (b * 4 + e)) & 1u);
                    if (MODE == CS_MODE_SAMPLE) {
                        gm[b] = fmaxf(gm[b], ok ? acc[b][e] * rc[0] + rc[2] : -INFINITY);
                    } else {
                        const float up = acc[b][e] * rc[0] + rc[1];
                        uv[b * 4 + e] = up;
                        bits |= (ok && up >= thrf[b]) ? (1u << (b * 4 + e)) : 0u;   // thrf = NaN: no query in this column
                    }
                }
            }
            }
            if (MODE == CS_MODE_SAMPLE) {
#pragma unroll
                for (int b = 0; b < NBc; ++b) {
                    float mx = gm[b];
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    const int q = qoff + 16 * b + lr;
                    if (IVF) {
                        // inverted lists: 8-row groups (rows 0-7 = lg 0,1; rows 8-15 = lg 2,3), two per
                        // tile, so that a query probing a single list still has 64 groups for its bound
                        if ((lg & 1) == 0 && q < a.nq)
                            a.gmax[(int64_t)q * a.gmax_ld + 2 * (j0 + tt) + (lg >> 1)] = mx;
                    } else {
                        mx = fmaxf(mx, __shfl_xor(mx, 32));
                        if (lg == 0 && q < a.nq) {
                            if (a.gshift)
                                atomicMax(reinterpret_cast<unsigned int*>(a.gmax) + (int64_t)q * a.gmax_ld + ((j0 + tt) >> a.gshift),
                                          ord_key(mx));
                            else
                                a.gmax[(int64_t)q * a.gmax_ld + (j0 + tt)] = mx;
                        }
                    }
                }
            } else if (!(a.dbg & 32)) {
                // Candidate append, one entry per lane and round: a lane emits its lowest pending
                // (query block, row) pair; the round's slots follow the wave's own (scalar) fill count.
                // Most lanes have nothing and most of the others exactly one pair, so a wave-tile
                // usually takes a single round (the 16-way per-bit branching this replaces cost
                // more than the MFMA loop's converts).
                unsigned rem = bits;
                for (;;) {
                    const unsigned long long m = __ballot(rem != 0u);
                    if (m == 0ull) break;
                    const int base = wc[par];
                    wc[par] = base + (int)__popcll(m);
                    if (rem != 0u) {
                        const int idx = __ffs(rem) - 1;      // = 4 b + e
                        float u = uv[0];
#pragma unroll
                        for (int i = 1; i < 4 * NBc; ++i) u = idx == i ? uv[i] : u;
                        const int p = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                          __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        int q = qoff + 16 * (idx >> 2) + lr;
                        int row = (int)r0 + 4 * lg + (idx & 3);
                        if (IVF) {                           // slot -> query, sorted row -> bank row
                            q = qid[0];
#pragma unroll
                            for (int b = 1; b < NBc; ++b) q = (idx >> 2) == b ? qid[b] : q;
                            float rb = rcv[0][3];
                            rb = (idx & 3) == 1 ? rcv[1][3] : rb;
                            rb = (idx & 3) == 2 ? rcv[2][3] : rb;
                            rb = (idx & 3) == 3 ? rcv[3][3] : rb;
                            row = __float_as_int(rb);
                        }
                        if (p < WCAP) {
                            lds_write3(wreg_addr + par * (WCAP * 12) + p * 12, (uint32_t)q, (uint32_t)row,
                                       __float_as_uint(u));
                        } else {
                            // burst beyond the buffer (e.g. a run of fresh rows that every query
                            // wants): straight to the query's list; slow (drains the prefetch) but exact
                            const int gp = atomicAdd(a.cnt + (int64_t)q * CNT_STRIDE, 1);
                            if (gp < a.cap) {
                                a.cand_scores[(int64_t)q * a.cap + gp] = u;
                                a.cand_idx[(int64_t)q * a.cap + gp] = row;
                            }
                        }
                        rem &= rem - 1u;
                    }
                }
            }
            };
            const uint32_t ts2 = stamp();
            const uint32_t ts3 = ts2;
            if (work) mma();
            if (issue_late && t + 2 < n_steps && !(a.dbg & 4)) issue_step(t + 2, (slot + 2) % NSLOT);
            const uint32_t ts4 = stamp();
            if (MODE == CS_MODE_FILTER && fl_n > 0) {       // second step of the write-out (wave-uniform)
                if (t + 2 >= n_steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TPS * GL) : "memory");   // reservations are older
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TPS * NP) : "memory");
                const uint32_t eb = wreg_addr + (par ^ 1) * (WCAP * 12);
#pragma unroll
                for (int u = 0; u < EPL; ++u) {
                    const int i = lane + u * 64;
                    const int pos = u == 0 ? fl_pos0 : fl_pos1;
                    if (i < fl_n && pos < a.cap) {
                        uint32_t eq, er, eu;
                        lds_read3(eb + i * 12, eq, er, eu);
                        a.cand_scores[(int64_t)eq * a.cap + pos] = __uint_as_float(eu);
                        a.cand_idx[(int64_t)eq * a.cap + pos] = (int32_t)er;
                    }
                }
                wc[par ^ 1] = 0;
            }

            if (work) epi(ctile);
            if (RS == 3) {                                          // the step's second tile, same wave
                set_tile(1);
                if (work) { mma(); epi(ctile); }
            }
            if (tm) {
                const uint32_t ts6 = stamp();
                tacc[0] += ts0b - ts0; tacc[1] += ts2 - ts1;
                tacc[3] += ts4 - ts3; tacc[4] += ts1 - ts0b; tacc[5] += ts6 - ts4;
            }
            slot = (slot + 1) % NSLOT;
        }
        if (tm && lane == 0) {
            float* const o = a.gmax + ((int64_t)blockIdx.x * 8 + wave) * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) o[i] += (float)tacc[i];
            o[6] += (float)seg;
        }
        };
        if (SP32 && QB == 2 && n_used <= 128 && !(a.dbg & 256)) run_segment(std::integral_constant<int, QB>{}, std::integral_constant<int, 2>{});
        else if (SP32 && QB == 2 && !(a.dbg & 1024)) run_segment(std::integral_constant<int, QB>{}, std::integral_constant<int, 3>{});
        else run_segment(std::integral_constant<int, QB>{}, std::integral_constant<int, 1>{});
        // segment end: the stream has drained and every wave is done with the block's LDS tables.  Buffered
        // candidates carry their query and bank row, so they stay where they are until the buffer fills or
        // the workgroup's span ends.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (MODE == CS_MODE_FILTER) {                           // span end: both halves go out
        if (wc[0] > 0 || wc[1] > 0) flush_all();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    if (CS_TIMERS && MODE == CS_MODE_FILTER && (a.dbg & 64) && a.gmax != nullptr && lane == 0)   // whole-kernel time of this wave
        a.gmax[((int64_t)blockIdx.x * 8 + wave) * 8 + 7] += (float)((uint32_t)__builtin_amdgcn_s_memrealtime() - t_kernel0);
}

// ------------------------------------------------------------------------------------------
// Refine: one workgroup per query.
// ------------------------------------------------------------------------------------------
struct RefineArgs {
    const float* bank;
    const float* inv_norm;
    const float* meta;
    const float* queries;
    const float* inv_q;
    float now, e_cos;           // e_cos: worst-case bound, used when rho == NULL (fp32-row prefilter)
    const float* rho;           // [N] per-row error norms of the bf16 shadow (NULL: no shadow)
    const float* eq;            // [nq] query parts of the bound (with rho)
    float e_fix, eq_worst;
    int64_t N, D;
    int k;
    const int32_t* cnt;
    const float* cand_scores;   // U
    const int32_t* cand_idx;
    int cap;
    int32_t idx_base;
    float* out_scores;          // [nq][k]
    int32_t* out_idx;
    int32_t* overflow;
    float* dbg_out;             // AURA_CS_DBG bit 128: [nq][8] phase times (100 MHz ticks), n, S
    const float* t2_ext;        // optional [nq]: a lower bound of every query's k-th best score known from OUTSIDE this
                                // bank (the shards of a row-sharded bank combine theirs, coarse_refine_bounds_kernel):
                                // candidates with U below it cannot be in the caller's top k and are not re-scored
    int32_t* heavy;             // optional [1 + nq]: heavy[0] = number of queries coarse_refine_wave_kernel left to the
                                // workgroup-per-query kernel (zero before that launch), heavy[1..] = those queries
};

// ------------------------------------------------------------------------------------------
// Refine, one WAVE per query (round 3) -- for passes of many queries.  The workgroup-per-query kernel below
// spends ~11 of its ~28 us per query in phases that are pure latency for 512 threads (two dependent gathers for
// the candidates' error terms, four radix passes with 16 barriers for T2, the query load) and holds at most two
// queries per CU.  Here a wave keeps its query's candidates in registers (up to RW_CAND per lane), finds T2 by a
// 32-step ballot search, re-scores its survivors RW_ROWS at a time through a private LDS stage with the SAME
// arithmetic (fmaf order of the MFMA chain, same epilogue expressions: bit-identical scores), ranks and writes.
// No workgroup barrier anywhere: 16 queries per CU in flight, each one's latencies hidden by the other fifteen.
// A query with more than 64 RW_CAND candidates or more than RW_SURV survivors goes on the `heavy` list and is left
// to the kernel below, launched right after with a small grid that walks that list (usually empty).
// ------------------------------------------------------------------------------------------
constexpr int RW_SURV = 128;             // survivors per query
constexpr int RW_KC = 192;               // k-chunk of the private stage; RW_ROWS survivors per round: 16 (one workgroup
                                         // = 8 queries per CU: a pass of at most 8 queries per CU, where a query's
                                         // ~60 survivors should take few rounds) or 6 (two workgroups per CU)

template <int RW_ROWS, int RW_CAND>      // RW_CAND: candidates per lane (64 RW_CAND per query at most)
__device__ __forceinline__ void refine_wave_body(const RefineArgs& a, int nq) {
    extern __shared__ __attribute__((aligned(16))) char rwmem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int q = blockIdx.x * 8 + wave;
    if (q >= nq) return;
    const int64_t D = a.D;
    const int64_t Dpad = (D + 31) / 32 * 32;
    constexpr int RSTRIDE = RW_KC + 4;
    const size_t region = (size_t)Dpad * 4 + (size_t)RW_ROWS * RSTRIDE * 4 + (size_t)RW_SURV * 12;
    char* const base = rwmem + (size_t)wave * region;
    float* const s_q = reinterpret_cast<float*>(base);
    float* const s_rows = s_q + Dpad;
    unsigned long long* const s_key = reinterpret_cast<unsigned long long*>(s_rows + RW_ROWS * RSTRIDE);
    int32_t* const s_surv = reinterpret_cast<int32_t*>(s_key + RW_SURV);

    const int n = a.cnt[(int64_t)q * CNT_STRIDE];
    const int capn = a.cap < RF_CAP ? a.cap : RF_CAP;
    if (n > capn || n > RW_CAND * 64) {                      // (overflowing lists are the other kernel's to report)
        if (lane == 0) a.heavy[1 + atomicAdd(a.heavy, 1)] = q;
        return;
    }
    const int nj = (n + 63) >> 6;                            // registers in use (wave-uniform)
    const float eqq = a.rho ? a.eq[q] : 0.0f;
    float cu[RW_CAND];
    int32_t cr[RW_CAND];
    uint32_t cl[RW_CAND];
    // two phases, so that every load of a phase is in flight at once (a load of the error terms behind each
    // candidate's own load would be nj round trips in a row)
#pragma unroll
    for (int j = 0; j < RW_CAND; ++j) {
        cu[j] = -INFINITY; cr[j] = 0; cl[j] = 0u;
        const int i = j * 64 + lane;
        if (j < nj && i < n) {
            cu[j] = a.cand_scores[(int64_t)q * a.cap + i];
            cr[j] = a.cand_idx[(int64_t)q * a.cap + i];
        }
    }
#pragma unroll
    for (int j = 0; j < RW_CAND; ++j) {
        const int i = j * 64 + lane;
        if (j < nj && i < n) {
            if ((uint32_t)cr[j] >= (uint32_t)a.N) {          // never a valid row: fail loudly (bit 4), do not fault
                if (a.overflow) atomicOr(a.overflow, 16);
                cr[j] = 0; cu[j] = -INFINITY;
            }
            const float strength = a.meta[(int64_t)cr[j] * 4];
            float err = 0.5f * a.e_cos * fabsf(strength);
            if (a.rho)
                err = 0.5f * fabsf(strength) * (a.rho[cr[j]] + a.e_fix + (strength < 0.0f ? 2.0f * a.eq_worst : eqq));
            cl[j] = ord_key(cu[j] - 2.0f * err);
        }
    }
    // ---- T2 = k-th largest L (everything survives if n < k) ----
    uint32_t t2 = 0u;
    if (n >= a.k) {
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = t2 | (1u << bit);
            int c = 0;
#pragma unroll
            for (int j = 0; j < RW_CAND; ++j)
                if (j < nj) c += (int)__popcll(__ballot(j * 64 + lane < n && cl[j] >= cand));
            if (c >= a.k) t2 = cand;
        }
    }
    if (a.t2_ext) { const uint32_t e = ord_key(a.t2_ext[q]); t2 = e > t2 ? e : t2; }
    // ---- survivors: U >= T2, compacted in candidate order ----
    int S = 0;
#pragma unroll
    for (int j = 0; j < RW_CAND; ++j) {
        if (j < nj) {
            const bool pass = j * 64 + lane < n && ord_key(cu[j]) >= t2;
            const unsigned long long m = __ballot(pass);
            if (pass) {
                const int p = S + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (p < RW_SURV) s_surv[p] = cr[j];
            }
            S += (int)__popcll(m);
        }
    }
    if (S > RW_SURV) {
        if (lane == 0) a.heavy[1 + atomicAdd(a.heavy, 1)] = q;
        return;
    }
    // (under an external bound a query whose candidates were all pruned HAD candidates: the flag is for n == 0)
    if (lane == 0 && S == 0 && (n == 0 || !a.t2_ext) && a.overflow) atomicOr(a.overflow, AURA_KNN_FLAG_NO_CANDIDATES);   // (not an overflow)
    // ---- the query, zero padded to Dpad ----
    for (int64_t i = lane; i < Dpad; i += 64) s_q[i] = i < D ? a.queries[(int64_t)q * D + i] : 0.0f;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float iq = a.inv_q[q];
    const bool ld_lane = lane * 4 < RW_KC;                   // lanes that move a row chunk
    // ---- exact re-scoring, RW_ROWS survivors per round; k order of the fp32 scan (knn_scan_filter_v2): inside
    //      each group of 8 consecutive k the MFMA chain visits 0,4,1,5,2,6,3,7; D padded with zeros to 32 ----
    for (int b0 = 0; b0 < S; b0 += RW_ROWS) {
        const int cntw = (S - b0) < RW_ROWS ? (S - b0) : RW_ROWS;
        const float* rowp[RW_ROWS];
#pragma unroll
        for (int r = 0; r < RW_ROWS; ++r)
            rowp[r] = (r < cntw && ld_lane) ? a.bank + (int64_t)s_surv[b0 + r] * D + lane * 4 : nullptr;
        float4 pre[RW_ROWS];
        auto fetch = [&](int64_t k0) {
#pragma unroll
            for (int r = 0; r < RW_ROWS; ++r) {
                pre[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rowp[r] && k0 + lane * 4 < D) pre[r] = *reinterpret_cast<const float4*>(rowp[r] + k0);
            }
        };
        float acc = 0.0f;
        fetch(0);
        for (int64_t k0 = 0; k0 < Dpad; k0 += RW_KC) {
            const int kc = (int)((Dpad - k0) < RW_KC ? (Dpad - k0) : RW_KC);
            if (ld_lane) {
#pragma unroll
                for (int r = 0; r < RW_ROWS; ++r)
                    *reinterpret_cast<float4*>(s_rows + r * RSTRIDE + lane * 4) = pre[r];
            }
            if (k0 + RW_KC < Dpad) fetch(k0 + RW_KC);        // next chunk flies during the chain
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < RW_ROWS) {
                const float* rp = s_rows + lane * RSTRIDE;
                const float* qp = s_q + k0;
#pragma unroll 4
                for (int kk = 0; kk < kc; kk += 8) {
                    const float4 b0v = *reinterpret_cast<const float4*>(rp + kk);
                    const float4 b1v = *reinterpret_cast<const float4*>(rp + kk + 4);
                    const float4 q0 = *reinterpret_cast<const float4*>(qp + kk);
                    const float4 q1 = *reinterpret_cast<const float4*>(qp + kk + 4);
                    acc = fmaf(q0.x, b0v.x, acc); acc = fmaf(q1.x, b1v.x, acc);
                    acc = fmaf(q0.y, b0v.y, acc); acc = fmaf(q1.y, b1v.y, acc);
                    acc = fmaf(q0.z, b0v.z, acc); acc = fmaf(q1.z, b1v.z, acc);
                    acc = fmaf(q0.w, b0v.w, acc); acc = fmaf(q1.w, b1v.w, acc);
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (lane < cntw) {
            const int si = b0 + lane;
            const int32_t row = s_surv[si];
            const float inv_m = a.inv_norm[row];
            const float4 m = *reinterpret_cast<const float4*>(a.meta + (int64_t)row * 4);
            const float tw = 0.2f * expf(-(a.now - m.y) / 3600.0f);
            const float sim = acc * iq * inv_m;
            const float comb = (0.5f * sim + tw) * m.x;
            s_key[si] = ((unsigned long long)ord_key(comb) << 32) | (uint32_t)(~(uint32_t)row);
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- rank the survivors' exact keys (all distinct) and write the top k, sorted ----
    for (int i = lane; i < S; i += 64) {
        const unsigned long long mine = s_key[i];
        int rank = 0;
        for (int j2 = 0; j2 < S; ++j2) rank += s_key[j2] > mine ? 1 : 0;
        if (rank < a.k) {
            a.out_scores[(int64_t)q * a.k + rank] = ord_unkey((uint32_t)(mine >> 32));
            a.out_idx[(int64_t)q * a.k + rank] = (int32_t)(~(uint32_t)mine) + a.idx_base;
        }
    }
    for (int i = S + lane; i < a.k; i += 64) {               // fewer than k rows can score
        a.out_scores[(int64_t)q * a.k + i] = -INFINITY;
        a.out_idx[(int64_t)q * a.k + i] = -1;
    }
}

// 16 survivors per round, up to 1024 candidates: one workgroup (8 queries) per CU, up to 256 registers per lane;
// 6 per round, up to 512 candidates: two workgroups per CU, so at most 128 registers
__global__ __launch_bounds__(RF_THREADS) void coarse_refine_wave16_kernel(const RefineArgs a, int nq) {
    refine_wave_body<16, 16>(a, nq);
}
__global__ __launch_bounds__(RF_THREADS, 4) void coarse_refine_wave6_kernel(const RefineArgs a, int nq) {
    refine_wave_body<6, 8>(a, nq);
}

// ---- the candidates' own bounds, for a row-sharded bank (round 3): per query T_k and T_k2, the k-th and the
//      k2-th largest LOWER bound L = U - 2 err among this bank's candidates (k2 = ceil(k / shards)).  The shards
//      combine them -- max over shards of T_k, min over shards of T_k2: S disjoint shards x k2 rows each are k rows
//      -- into a lower bound of the GLOBAL k-th best score, far tighter than the sampled one of stage 1 (it comes
//      from the filtered candidates, i.e. from exact bounds of the best rows), and the refine launches re-score
//      only candidates whose U reaches it (RefineArgs::t2_ext): ~k / S + gap survivors per shard instead of k + gap.
//      One wave per query, candidates in registers as in refine_wave_body; a list too long for that (more than
//      512 candidates) reports -inf, the neutral element of both reductions.  out [nq][2]. ----
__global__ __launch_bounds__(RF_THREADS) void coarse_refine_bounds_kernel(const RefineArgs a, int nq, int k2,
                                                                          float* __restrict__ out) {
    constexpr int RW_CAND = 8;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int q = blockIdx.x * (RF_THREADS / 64) + wave;
    if (q >= nq) return;
    const int n = a.cnt[(int64_t)q * CNT_STRIDE];
    const int capn = a.cap < RF_CAP ? a.cap : RF_CAP;
    float tk = -INFINITY, tk2 = -INFINITY;
    if (n <= capn && n <= RW_CAND * 64 && n > 0) {
        const int nj = (n + 63) >> 6;
        const float eqq = a.rho ? a.eq[q] : 0.0f;
        float cu[RW_CAND];
        int32_t cr[RW_CAND];
        uint32_t cl[RW_CAND];
#pragma unroll
        for (int j = 0; j < RW_CAND; ++j) {
            cu[j] = -INFINITY; cr[j] = 0; cl[j] = 0u;
            const int i = j * 64 + lane;
            if (j < nj && i < n) {
                cu[j] = a.cand_scores[(int64_t)q * a.cap + i];
                cr[j] = a.cand_idx[(int64_t)q * a.cap + i];
            }
        }
#pragma unroll
        for (int j = 0; j < RW_CAND; ++j) {
            const int i = j * 64 + lane;
            if (j < nj && i < n) {
                if ((uint32_t)cr[j] >= (uint32_t)a.N) { cr[j] = 0; cu[j] = -INFINITY; }   // (reported by the refine launch)
                const float strength = a.meta[(int64_t)cr[j] * 4];
                float err = 0.5f * a.e_cos * fabsf(strength);
                if (a.rho)
                    err = 0.5f * fabsf(strength) * (a.rho[cr[j]] + a.e_fix + (strength < 0.0f ? 2.0f * a.eq_worst : eqq));
                cl[j] = ord_key(cu[j] - 2.0f * err);
            }
        }
        auto kth = [&](int kk) -> float {                    // kk-th largest L, -inf if there are fewer
            if (kk <= 0 || n < kk) return -INFINITY;
            uint32_t t = 0u;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t cand = t | (1u << bit);
                int c = 0;
#pragma unroll
                for (int j = 0; j < RW_CAND; ++j)
                    if (j < nj) c += (int)__popcll(__ballot(j * 64 + lane < n && cl[j] >= cand));
                if (c >= kk) t = cand;
            }
            return ord_unkey(t);
        };
        tk = kth(a.k);
        tk2 = kth(k2);
    }
    if (lane == 0) { out[(int64_t)q * 2] = tk; out[(int64_t)q * 2 + 1] = tk2; }
}

template <int RF_ROWS, int RF_KC>
__device__ __forceinline__ void refine_one_query(const RefineArgs& a, const int q) {
    extern __shared__ __attribute__((aligned(16))) char rsmem[];
    // phase A: candidate arrays; phase B (aliases A): per-wave row chunks + the query
    float* const s_u = reinterpret_cast<float*>(rsmem);                       // [RF_CAP]
    uint32_t* const s_l = reinterpret_cast<uint32_t*>(rsmem + RF_CAP * 4);    // [RF_CAP] ord_key(L)
    int32_t* const s_i = reinterpret_cast<int32_t*>(rsmem + RF_CAP * 8);      // [RF_CAP]
    __shared__ int32_t s_surv[RF_SURV];
    __shared__ unsigned long long s_key[RF_SURV];
    __shared__ int s_hist[256];
    __shared__ int s_wsum[4];
    __shared__ int s_ns, s_sel_k;
    __shared__ uint32_t s_prefix;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int64_t D = a.D;
    const bool tm = CS_TIMERS && a.dbg_out != nullptr;     // (phase timers: -DAURA_CS_TIMERS=1 builds only, see CS_TIMERS)
    uint32_t tst[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
    auto stamp = [&](int i) { if (tm) tst[i] = (uint32_t)__builtin_amdgcn_s_memrealtime(); };
    stamp(0);
    int n = a.cnt[(int64_t)q * CNT_STRIDE];
    bool ovf = false;
    const int capn = a.cap < RF_CAP ? a.cap : RF_CAP;
    int ovf_bits = 0;
    if (n > capn) { ovf = true; ovf_bits |= 4; n = capn; }

    const float eqq = a.rho ? a.eq[q] : 0.0f;
    for (int i = tid; i < n; i += RF_THREADS) {
        float u = a.cand_scores[(int64_t)q * a.cap + i];
        int32_t r = a.cand_idx[(int64_t)q * a.cap + i];
        if ((uint32_t)r >= (uint32_t)a.N) {              // never a valid row: fail loudly (bit 4), do not fault
            if (a.overflow) atomicOr(a.overflow, 16);
            r = 0; u = -INFINITY;
        }
        // the pair's error as the scan kernels formed it (see the header): L = U - 2 err
        const float strength = a.meta[(int64_t)r * 4];
        float err = 0.5f * a.e_cos * fabsf(strength);
        if (a.rho)
            err = 0.5f * fabsf(strength) * (a.rho[r] + a.e_fix + (strength < 0.0f ? 2.0f * a.eq_worst : eqq));
        s_u[i] = u;
        s_i[i] = r;
        s_l[i] = ord_key(u - 2.0f * err);
    }
    if (tid == 0) { s_ns = 0; s_prefix = 0u; s_sel_k = a.k; }
    __syncthreads();
    stamp(1);

    // ---- T2 = k-th largest L: MSB-first 8-bit radix select, the digit found by a parallel suffix
    //      scan of the 256-bin histogram; everything survives if n < k ----
    uint32_t t2 = 0u;
    if (n >= a.k) {
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) s_hist[tid] = 0;
            __syncthreads();
            const uint32_t prefix = s_prefix;
            const int need = s_sel_k;
            const uint32_t himask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
            for (int i = tid; i < n; i += RF_THREADS) {
                const uint32_t key = s_l[i];
                if ((key & himask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255], 1);
            }
            __syncthreads();
            int h = 0, v = 0;
            if (tid < 256) {
                h = s_hist[tid];
                v = h;                                   // -> sum of bins tid .. end of this wave
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_down(v, off);
                    if (lane + off < 64) v += o;
                }
                if (lane == 0) s_wsum[wave] = v;
            }
            __syncthreads();
            if (tid < 256) {
                for (int w2 = wave + 1; w2 < 4; ++w2) v += s_wsum[w2];   // keys with digit >= tid
                if (v >= need && v - h < need) {          // exactly one bin satisfies this
                    s_sel_k = need - (v - h);
                    s_prefix = prefix | ((uint32_t)tid << shift);
                }
            }
            __syncthreads();
        }
        t2 = s_prefix;
    }
    if (a.t2_ext) { const uint32_t e = ord_key(a.t2_ext[q]); t2 = e > t2 ? e : t2; }
    stamp(2);
    // ---- survivors: U >= T2 ----
    for (int i = tid; i < n; i += RF_THREADS) {
        if (ord_key(s_u[i]) >= t2) {
            const int p = atomicAdd(&s_ns, 1);
            if (p < RF_SURV) s_surv[p] = s_i[i];
        }
    }
    __syncthreads();
    int S = s_ns;
    if (S > RF_SURV) { ovf = true; ovf_bits |= 8; S = RF_SURV; }
    if (ovf && tid == 0 && a.overflow) atomicOr(a.overflow, ovf_bits);   // which list overflowed
    if (S == 0 && (n == 0 || !a.t2_ext) && tid == 0 && a.overflow) atomicOr(a.overflow, AURA_KNN_FLAG_NO_CANDIDATES);   // (not an overflow)
    __syncthreads();     // candidate arrays are dead from here: rsmem is reused below

    // ---- exact re-scoring: rounds of 8 x RF_ROWS survivors, survivor base + 8 r + w -> wave w, slot r ----
    // k order of the fp32 scan (knn_scan_filter_v2): inside each group of 8 consecutive k the
    // MFMA chain visits 0,4,1,5,2,6,3,7; D is padded with zeros to a multiple of 32.
    constexpr int RSTRIDE = RF_KC + 4;                       // floats; 16-B aligned, conflict-free
    float* const s_rows = reinterpret_cast<float*>(rsmem) + wave * (RF_ROWS * RSTRIDE);
    float* const s_q = reinterpret_cast<float*>(rsmem) + 8 * (RF_ROWS * RSTRIDE);   // [Dpad]
    const int64_t Dpad = (D + 31) / 32 * 32;
    stamp(3);
    for (int64_t i = tid; i < Dpad; i += RF_THREADS) s_q[i] = i < D ? a.queries[(int64_t)q * D + i] : 0.0f;
    __syncthreads();
    stamp(4);
    const float iq = a.inv_q[q];
    const bool ld_lane = lane * 4 < RF_KC;                   // lanes that move a row chunk
    for (int base = 0; base < S; base += 8 * RF_ROWS) {
        int cntw = (S - base - wave + 7) / 8;                // survivors of this wave in this round
        cntw = cntw < 0 ? 0 : (cntw > RF_ROWS ? RF_ROWS : cntw);
        if (cntw <= 0) continue;                             // wave-uniform
        const float* rowp[RF_ROWS];
#pragma unroll
        for (int r = 0; r < RF_ROWS; ++r)
            rowp[r] = (r < cntw && ld_lane) ? a.bank + (int64_t)s_surv[base + 8 * r + wave] * D + lane * 4 : nullptr;
        float4 pre[RF_ROWS];
        auto fetch = [&](int64_t k0) {                       // chunk [k0, k0 + RF_KC): lane -> 16 bytes
#pragma unroll
            for (int r = 0; r < RF_ROWS; ++r) {
                pre[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rowp[r] && k0 + lane * 4 < D) pre[r] = *reinterpret_cast<const float4*>(rowp[r] + k0);
            }
        };
        float acc = 0.0f;
        fetch(0);
        for (int64_t k0 = 0; k0 < Dpad; k0 += RF_KC) {
            const int kc = (int)((Dpad - k0) < RF_KC ? (Dpad - k0) : RF_KC);
            if (ld_lane) {
#pragma unroll
                for (int r = 0; r < RF_ROWS; ++r)
                    *reinterpret_cast<float4*>(s_rows + r * RSTRIDE + lane * 4) = pre[r];
            }
            if (k0 + RF_KC < Dpad) fetch(k0 + RF_KC);        // next chunk flies during the chain
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < RF_ROWS) {
                const float* rp = s_rows + lane * RSTRIDE;
                const float* qp = s_q + k0;
                // (measured: running the LDS reads four groups ahead of the chain changes nothing -- the
                // phase is bound by the gather of the survivors' fp32 rows, 256 x 71 x 3 KB = 56 MB of
                // random 1-KB pieces per launch at ~4 TB/s; AURA_CS_DBG=128 prints the phase times)
#pragma unroll 4
                for (int kk = 0; kk < kc; kk += 8) {
                    const float4 b0 = *reinterpret_cast<const float4*>(rp + kk);
                    const float4 b1 = *reinterpret_cast<const float4*>(rp + kk + 4);
                    const float4 q0 = *reinterpret_cast<const float4*>(qp + kk);
                    const float4 q1 = *reinterpret_cast<const float4*>(qp + kk + 4);
                    acc = fmaf(q0.x, b0.x, acc); acc = fmaf(q1.x, b1.x, acc);
                    acc = fmaf(q0.y, b0.y, acc); acc = fmaf(q1.y, b1.y, acc);
                    acc = fmaf(q0.z, b0.z, acc); acc = fmaf(q1.z, b1.z, acc);
                    acc = fmaf(q0.w, b0.w, acc); acc = fmaf(q1.w, b1.w, acc);
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (lane < cntw) {
            const int si = base + 8 * lane + wave;
            const int32_t row = s_surv[si];
            const float inv_m = a.inv_norm[row];
            const float4 m = *reinterpret_cast<const float4*>(a.meta + (int64_t)row * 4);
            const float tw = 0.2f * expf(-(a.now - m.y) / 3600.0f);
            const float sim = acc * iq * inv_m;
            const float comb = (0.5f * sim + tw) * m.x;
            s_key[si] = ((unsigned long long)ord_key(comb) << 32) | (uint32_t)(~(uint32_t)row);
        }
    }
    __syncthreads();
    stamp(5);
    // ---- rank the survivors' exact keys (all distinct) and write the top k, sorted ----
    for (int i = tid; i < S; i += RF_THREADS) {
        const unsigned long long mine = s_key[i];
        int rank = 0;
        for (int j2 = 0; j2 < S; ++j2) rank += s_key[j2] > mine ? 1 : 0;
        if (rank < a.k) {
            a.out_scores[(int64_t)q * a.k + rank] = ord_unkey((uint32_t)(mine >> 32));
            a.out_idx[(int64_t)q * a.k + rank] = (int32_t)(~(uint32_t)mine) + a.idx_base;
        }
    }
    for (int i = S + tid; i < a.k; i += RF_THREADS) {       // fewer than k rows can score (never when k <= N)
        a.out_scores[(int64_t)q * a.k + i] = -INFINITY;
        a.out_idx[(int64_t)q * a.k + i] = -1;
    }
    if (tm) {
        __syncthreads();
        stamp(6);
        if (tid == 0) {
            float* const o = a.dbg_out + (int64_t)q * 8;
            for (int i = 0; i < 6; ++i) o[i] = (float)(tst[i + 1] - tst[i]);
            o[6] = (float)n; o[7] = (float)S;
        }
    }
}

// one workgroup per query
template <int RF_ROWS, int RF_KC>
__global__ __launch_bounds__(RF_THREADS) void coarse_refine_kernel(const RefineArgs a) {
    refine_one_query<RF_ROWS, RF_KC>(a, (int)blockIdx.x);
}

// a small grid walks the list of queries coarse_refine_wave_kernel left over (a.heavy; usually empty)
template <int RF_ROWS, int RF_KC>
__global__ __launch_bounds__(RF_THREADS) void coarse_refine_list_kernel(const RefineArgs a) {
    const int n_items = a.heavy[0];
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        refine_one_query<RF_ROWS, RF_KC>(a, a.heavy[1 + item]);
        __syncthreads();                                     // the shared arrays are reused
    }
}


inline bool coarse_eligible(const float* bank, const uint16_t* bank16, const float* queries, const float* loc_q,
                            const float* centroids, int64_t N, int64_t D, int k, int flags) {
    static const bool off = getenv("AURA_KNN_NO_COARSE") != nullptr;
    if (off || (flags & (AURA_KNN_FORCE_DENSE | AURA_KNN_FP32_SCAN))) return false;
    if (loc_q) return false;
    // the centroid-candidate mode keeps its probe masks in the LDS only the bf16-row kernels have free
    if (centroids && !(bank16 && (D & 7) == 0 && (reinterpret_cast<uintptr_t>(bank16) & 15) == 0)) return false;
    if (N < COARSE_MIN_ROWS || D > 768 || (D & 3) || k > COARSE_MAX_K) return false;
    if ((reinterpret_cast<uintptr_t>(bank) & 15) || (reinterpret_cast<uintptr_t>(queries) & 15)) return false;
    return true;
}

template <int KS, bool SRC16, bool MASKED, int NW = 4>
inline int launch_coarse(const CoarseArgs& a, int mode, int grid, hipStream_t s) {
    const size_t lds = (size_t)cs_lds_slots<SRC16, MASKED, NW>() * (KS * (SRC16 ? 1024 : 2048) + CS_AUX_BYTES) + (size_t)CS_BUF * 12 +
                       (MASKED ? 256 * 32 : 0) + (SRC16 ? 256 * 4 : 0);
    if (ensure_lds_attr(reinterpret_cast<const void*>(coarse_scan_kernel<KS, CS_MODE_SAMPLE, SRC16, MASKED, false, NW>), (int)lds) ||
        ensure_lds_attr(reinterpret_cast<const void*>(coarse_scan_kernel<KS, CS_MODE_FILTER, SRC16, MASKED, false, NW>), (int)lds))
        return AURA_E_LAUNCH;
    if (mode == CS_MODE_SAMPLE)
        hipLaunchKernelGGL((coarse_scan_kernel<KS, CS_MODE_SAMPLE, SRC16, MASKED, false, NW>), dim3(grid), dim3(64 * NW), lds, s, a);
    else
        hipLaunchKernelGGL((coarse_scan_kernel<KS, CS_MODE_FILTER, SRC16, MASKED, false, NW>), dim3(grid), dim3(64 * NW), lds, s, a);
    return check_launch();
}

inline int dispatch_coarse(const CoarseArgs& a, int mode, int grid, hipStream_t s) {
    const int64_t ks = (a.D + 31) / 32;
    // bf16-row kernels run 8 waves per workgroup (two per SIMD) unless AURA_CS_WAVES4 is set (A/B runs)
    static const bool w4 = getenv("AURA_CS_WAVES4") != nullptr;
    if (a.bank16 && a.probe_mask) {
        if (!w4) {
            if (ks <= 8) return launch_coarse<8, true, true, 8>(a, mode, grid, s);
            if (ks <= 16) return launch_coarse<16, true, true, 8>(a, mode, grid, s);
            return launch_coarse<24, true, true, 8>(a, mode, grid, s);
        }
        if (ks <= 8) return launch_coarse<8, true, true>(a, mode, grid, s);
        if (ks <= 16) return launch_coarse<16, true, true>(a, mode, grid, s);
        return launch_coarse<24, true, true>(a, mode, grid, s);
    }
    if (a.probe_mask) return AURA_E_INVAL;                   // masked mode needs the bf16 rows
    if (a.bank16) {
        if (!w4) {
            if (ks <= 8) return launch_coarse<8, true, false, 8>(a, mode, grid, s);
            if (ks <= 16) return launch_coarse<16, true, false, 8>(a, mode, grid, s);
            return launch_coarse<24, true, false, 8>(a, mode, grid, s);
        }
        if (ks <= 8) return launch_coarse<8, true, false>(a, mode, grid, s);
        if (ks <= 16) return launch_coarse<16, true, false>(a, mode, grid, s);
        return launch_coarse<24, true, false>(a, mode, grid, s);
    }
    if (ks <= 8) return launch_coarse<8, false, false>(a, mode, grid, s);
    if (ks <= 16) return launch_coarse<16, false, false>(a, mode, grid, s);
    return launch_coarse<24, false, false>(a, mode, grid, s);
}
