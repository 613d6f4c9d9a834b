// aura_knn_ivf2.inl -- centroid-index (inverted-list) recall through the two-stage scan.
// Included by aura_knn.hip after aura_knn_coarse.inl and the IVF section (same translation unit).
//
// The reference restricts a query to the rows of its `nprobe` nearest centroids
// (src/core/hippocampal.py:259-270).  aura_knn_search_ivf streams every probed list once per batch
// on the fp32 matrix pipe and writes ALL candidate scores; here the same restriction runs on the
// bf16 prefilter + fp32 re-scoring machinery of aura_knn_coarse.inl:
//   * the caller keeps a LIST-SORTED bf16 shadow of the bank (rows grouped by centroid, every list
//     padded to a multiple of 16 rows; `sorted_rows[i]` = bank row of sorted row i, -1 for pads),
//     rebuilt whenever its inverted lists are (aura_bank_shadow_sorted);
//   * a batch is regrouped BY LIST (ivf_prepare_kernel): block B = (list, up to 256 of the queries
//     that probe it).  The block's queries are the scan's stationary bf16 fragments and the list's
//     contiguous sorted rows stream past them, so a list is read once per batch whatever the batch
//     size (coarse_scan_kernel<..., IVF>);
//   * thresholds: a query's T is the k-th largest 8-row group maximum of L over IVF2_STILES tiles
//     spread evenly over each of its lists (k distinct candidate rows score >= T; spread because
//     rows inside a list are in write order and the score has a recency term);
//   * candidates (U >= T) land in per-query lists with their ORIGINAL row ids and go through
//     coarse_refine_kernel unchanged: tightened, re-scored in fp32 from the fp32 bank, ranked.
// Results are bit-identical to aura_knn_search_ivf and to the masked scans (same candidate sets,
// same fp32 arithmetic, ties to the lower row).

constexpr int IVF2_STILES = 8;             // sample tiles (of 16 rows) per list, at least
constexpr int IVF2_STILES_MAX = 256;       // ... and at most
// Sample tiles per list for a sorted shadow of n_sorted rows (slack included) and top-k: a power of two in [8, 256],
//   * at least a twelfth of an average list, so that the k-th best of the sample sits near rank 8..12 k of the
//     probed rows whatever the bank's size -- with a fixed 32 tiles a 10 M-row bank's queries collected more
//     candidates than the refine stage holds and every call fell back to the fp32 lists.  (At 1 M rows 64 tiles
//     instead of 32 cost 35 us more in the sample scan and threshold launches and saved 15 in the refine.)
//   * at least k / 8 (a query's eight lists then hold 16 x tiles >= 2 k sampled 8-row groups: the k-th largest
//     group maximum exists with room to spare; k = 256 -> 32 tiles, the floor of rounds 1-3a for every k);
//   * no more than that for SHORT lists: with the old floor of 32 a 125 000-row shard of the 8-rank layout (30 tiles
//     per list) was sampled in full -- the sample scan did the filter scan's work once more (109 of a pass's 480 us).
//     (Only where a bound from outside prunes the refine -- the staged recall of a sharded bank, `exchanged`: a
//     bank on its own pays for the looser threshold in the refine, 1.53 -> 2.05 ms at 125 000 rows x 16 384 queries,
//     and keeps the floor of 32.)
static inline int ivf2_stiles(int64_t n_sorted, int k, bool exchanged = false) {
    static const int forced = getenv("AURA_IVF_STILES") ? atoi(getenv("AURA_IVF_STILES")) : 0;   // tuning runs: 8 .. 256
    int st = exchanged ? IVF2_STILES : 32;
    while (st < IVF2_STILES_MAX && 8 * st < k) st *= 2;
    if (forced == 8 || forced == 16 || forced == 32 || forced == 64 || forced == 128 || forced == 256) return forced > st ? forced : st;
    const int64_t avg_tiles = n_sorted / 16 / 256;
    while (st < IVF2_STILES_MAX && (int64_t)st * 12 < avg_tiles) st *= 2;
    return st;
}
constexpr int IVF2_MAXQ = 8192;            // queries per pass: every probed list is streamed once per 256 of the
                                           // queries that probe it, so large batches (bulk recall, the all-gathered
                                           // query blocks of a sharded bank) want long passes; workspace grows with
                                           // min(nq, IVF2_MAXQ) (2048 queries: as in r01)
// A block holds 1 << bsh query slots: 256, or 128 in the two-workgroups-per-CU form of the scan (AURA_IVF_WG4).
// Blocks of a pass of qp queries at most: sum over the lists of ceil(count / B) <= 256 + 8 qp / B.
static inline int ivf2_maxblk(int64_t qp, int bsh = 8) { return 256 + (int)((qp * 8 + (1 << bsh) - 1) >> bsh); }
constexpr int IVF2_MAXBLK = 256 + IVF2_MAXQ * 8 / 128;
static_assert(IVF2_MAXBLK <= IVF2_MAXBLK_C, "coarse_scan_kernel's block search covers IVF2_MAXBLK_C prefixes");

// queries probing list c: the per-list query list holds IVF2_MAXQ entries, a counter beyond that (caller-made
// probe ids that repeat a list) is clamped everywhere it is read
__device__ __forceinline__ int ivf2_list_queries(const int32_t* __restrict__ lq_cnt, int c) {
    const int n = lq_cnt[c];
    return n < IVF2_MAXQ ? n : IVF2_MAXQ;
}

// ---- plan: blocks, their row ranges and the work-item prefixes (one workgroup) ----
__global__ __launch_bounds__(256) void ivf2_plan_kernel(const int32_t* __restrict__ lq_cnt,
                                                        const int32_t* __restrict__ pad_off,   // [257] first sorted row of each list
                                                        const int32_t* __restrict__ list_len,  // [256] entries in use (holes included)
                                                        int32_t* blk_off,    // [257] first block of each list
                                                        int32_t* blk_list,   // [MAXBLK] list of block B
                                                        int32_t* blk_row0,   // [MAXBLK]
                                                        int32_t* blk_stride, // [MAXBLK] tiles between sample tiles
                                                        int32_t* blk_nq,     // [MAXBLK] query slots in use
                                                        int32_t* item_off,   // [MAXBLK + 1] filter work (tiles x block weight)
                                                        int32_t* sitem_off,  // [MAXBLK + 1] sample items
                                                        int32_t* nblk,
                                                        int stiles,          // sample tiles per list (ivf2_stiles)
                                                        int w_sparse, int w_dense,     // cost of a tile of a block of <= 128 / more queries
                                                        int bsh) {           // log2 of the query slots per block (8 or 7)
    __shared__ int s_nb[257];
    const int tid = threadIdx.x;
    const int my_cnt = ivf2_list_queries(lq_cnt, tid);
    const int BQ = 1 << bsh;
    s_nb[tid + 1] = (my_cnt + BQ - 1) >> bsh;
    if (tid == 0) s_nb[0] = 0;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {              // inclusive scan of s_nb[1..256]
        const int add = tid >= off ? s_nb[tid + 1 - off] : 0;
        __syncthreads();
        s_nb[tid + 1] += add;
        __syncthreads();
    }
    blk_off[tid] = s_nb[tid];
    if (tid == 0) { blk_off[256] = s_nb[256]; nblk[0] = s_nb[256]; }
    const int tiles = (list_len[tid] + 15) / 16;
    for (int b = s_nb[tid]; b < s_nb[tid + 1]; ++b) {
        blk_list[b] = tid;
        blk_row0[b] = pad_off[tid];
        const int left = my_cnt - (b - s_nb[tid]) * BQ;
        blk_nq[b] = left < BQ ? left : BQ;
        blk_stride[b] = tiles > stiles ? tiles / stiles : 1;   // sample tiles j * stride, j < stiles
    }
    __syncthreads();
    // work prefixes: list c contributes (tiles of c) x (weights of its blocks) filter work and
    // (blocks of c) x min(tiles, STILES) sample items; exclusive scan over the lists in LDS.  The filter
    // scan splits the WORK evenly over its workgroups, and a tile of a block of at most 128 queries
    // (one column block per wave, see coarse_scan_kernel) takes less time than one of a fuller block:
    // all but the last block of a list hold 256 queries.
    __shared__ int s_it[257], s_st[257];
    const int nb = s_nb[tid + 1] - s_nb[tid];
    const int stl = tiles < stiles ? tiles : stiles;
    const int last_left = my_cnt - (nb - 1) * BQ;
    const int w_last = last_left <= 128 ? w_sparse : w_dense;
    s_it[tid + 1] = nb > 0 ? tiles * ((nb - 1) * w_dense + w_last) : 0; s_st[tid + 1] = nb * stl;
    if (tid == 0) { s_it[0] = 0; s_st[0] = 0; }
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int a_it = tid >= off ? s_it[tid + 1 - off] : 0;
        const int a_st = tid >= off ? s_st[tid + 1 - off] : 0;
        __syncthreads();
        s_it[tid + 1] += a_it; s_st[tid + 1] += a_st;
        __syncthreads();
    }
    for (int b = s_nb[tid], j = 0; b < s_nb[tid + 1]; ++b, ++j) {
        item_off[b] = s_it[tid] + j * tiles * w_dense;
        sitem_off[b] = s_st[tid] + j * stl;
    }
    if (tid == 0) { item_off[s_nb[256]] = s_it[256]; sitem_off[s_nb[256]] = s_st[256]; }
}

// ---- per-list query lists from probe ids the caller already has (aura_knn_search_ivf2_probed) ----
// A workgroup takes 8192 ids: ranks inside the workgroup come from an LDS histogram (returning LDS atomics), ONE
// global atomic per list and workgroup reserves the block of list entries.  (One returning global atomic per id --
// 65 536 of them on 256 counters for a pass of 8192 queries -- took 45 us, 8 % of a shard's share at 8 ranks.)
constexpr int LFI_PER = 32;                                // ids per thread
__global__ __launch_bounds__(256) void ivf2_lists_from_ids_kernel(const int32_t* __restrict__ ids,   // [nq][8]
                                                                  int nq, int nprobe, int32_t* lq_cnt,
                                                                  int32_t* __restrict__ lq_list) {
    __shared__ int s_hist[256], s_base[256];
    const int tid = threadIdx.x;
    s_hist[tid] = 0;
    __syncthreads();
    const int t0 = blockIdx.x * (256 * LFI_PER);
    int cl[LFI_PER], rk[LFI_PER];
#pragma unroll
    for (int u = 0; u < LFI_PER; ++u) {
        const int t = t0 + u * 256 + tid;                  // (coalesced: consecutive threads, consecutive ids)
        const int q = t >> 3, p = t & 7;
        int c = -1;
        if (q < nq && p < nprobe) {
            c = ids[t];
            if ((unsigned)c >= 256u) c = -1;               // not a centroid row: the probe is dropped
            for (int p2 = 0; p2 < p && c >= 0; ++p2)       // a list named twice by one query is scanned once
                if (ids[q * 8 + p2] == c) c = -1;
        }
        cl[u] = c;
        rk[u] = c >= 0 ? atomicAdd(&s_hist[c], 1) : 0;
    }
    __syncthreads();
    s_base[tid] = s_hist[tid] > 0 ? atomicAdd(&lq_cnt[tid], s_hist[tid]) : 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < LFI_PER; ++u) {
        if (cl[u] >= 0) {
            const int t = t0 + u * 256 + tid;
            const int slot = s_base[cl[u]] + rk[u];
            if (slot < IVF2_MAXQ) lq_list[(int64_t)cl[u] * IVF2_MAXQ + slot] = ((t >> 3) << 4) | (t & 7);
        }
    }
}

// ---- per-query prep (one wave per query): 1/||q|| (query_prep_kernel's arithmetic), the normalised query as
//      bf16 fragments [q][KS][4 k-groups][8] -- ONE set per query, whatever the number of lists it probes; the
//      scan's lanes find their slot's query through slotq (rounds 1-2 wrote one set per (query, probe) slot:
//      25 MB per 2048 queries, 31 us) -- its part eq of the error bound, and the pass's resets: qslot = -1,
//      the per-list counters, the call's flag, the all-zero entry nq that unused slots read ----
__global__ __launch_bounds__(256) void ivf2_qprep_kernel(const float* __restrict__ x, int64_t nq, int64_t D, int KS,
                                                         uint16_t* __restrict__ qfrag, float* __restrict__ inv,
                                                         float* __restrict__ eq_q, int32_t* __restrict__ qslot,
                                                         int32_t* __restrict__ lq_cnt, int32_t* overflow,
                                                         const int32_t* __restrict__ lists_flag) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0) {
        lq_cnt[threadIdx.x] = 0;
        // reset of the call's flag; bit 7: aura_ivf2_append dropped a row because a list had no slack left
        if (overflow && threadIdx.x == 0) *overflow = (lists_flag && *lists_flag) ? AURA_KNN_FLAG_LISTS_STALE : 0;
    }
    if (q > nq) return;
    uint16_t* const base = qfrag + q * (int64_t)KS * 32;
    if (q == nq) {                                          // the zero entry
        for (int c = lane; c < KS * 4; c += 64) *reinterpret_cast<uint4*>(base + c * 8) = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    float s = 0.0f;
    for (int64_t i = lane * 4; i < D; i += 256) {          // 1/||q|| with query_prep_kernel's arithmetic
        const float4 u = *reinterpret_cast<const float4*>(x + q * D + i);
        s = fmaf(u.x, u.x, s); s = fmaf(u.y, u.y, s); s = fmaf(u.z, u.z, s); s = fmaf(u.w, u.w, s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float iqv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    float e2 = 0.0f;
    for (int c = lane; c < KS * 4; c += 64) {              // chunk c: k = 8c .. 8c+7 = k-step c/4, k-group c%4
        f32x8v v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.0f;
        const int64_t k0 = 8 * (int64_t)c;
        if (k0 < D) {
            const float4 u = *reinterpret_cast<const float4*>(x + q * D + k0);
            v[0] = u.x * iqv; v[1] = u.y * iqv; v[2] = u.z * iqv; v[3] = u.w * iqv;
            if (k0 + 4 < D) {
                const float4 w = *reinterpret_cast<const float4*>(x + q * D + k0 + 4);
                v[4] = w.x * iqv; v[5] = w.y * iqv; v[6] = w.z * iqv; v[7] = w.w * iqv;
            }
        }
        const bf16x8v bv = __builtin_convertvector(v, bf16x8v);
        *reinterpret_cast<bf16x8v*>(base + c * 8) = bv;
        const f32x8v back = __builtin_convertvector(bv, f32x8v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = back[e] - v[e]; e2 = fmaf(d, d, e2); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off);
    if (lane == 0) {
        inv[q] = iqv;
        eq_q[q] = coarse_eq_from_e2(e2, (float)D);          // query part of the error bound (coarse_prep_kernel)
    }
    if (lane < 8) qslot[q * 8 + lane] = -1;                 // probes without a slot (dropped ids) stay -1
}

// ---- per block slot (one thread each): slot <-> query maps, +inf thresholds, the slot's eq ----
__global__ __launch_bounds__(256) void ivf2_slots_kernel(const int32_t* __restrict__ lq_cnt,
                                                         const int32_t* __restrict__ lq_list,   // [256][IVF2_MAXQ]
                                                         const int32_t* __restrict__ blk_off,
                                                         const int32_t* __restrict__ blk_list,
                                                         const int32_t* __restrict__ nblk,
                                                         const float* __restrict__ eq_q,
                                                         int32_t* __restrict__ slotq, int32_t* __restrict__ qslot,
                                                         uint32_t* __restrict__ thr, float* __restrict__ eq_slot, int bsh) {
    const int64_t vs = (int64_t)blockIdx.x * 256 + threadIdx.x;            // virtual slot = (B << bsh) + slot
    const int B = (int)(vs >> bsh);
    if (B >= nblk[0]) return;
    const int list = blk_list[B];
    const int ls = ((B - blk_off[list]) << bsh) + (int)(vs & ((1 << bsh) - 1));   // slot inside the list's query list
    int q = -1, p = 0;
    if (ls < ivf2_list_queries(lq_cnt, list)) {
        const int packed = lq_list[(int64_t)list * IVF2_MAXQ + ls];
        q = packed >> 4; p = packed & 15;
    }
    slotq[vs] = q;
    // used slot: ord_key(+inf), nothing passes until the threshold kernel lowers it.  Unused slot: the key of a
    // NaN -- "U >= NaN" is false for EVERY U.  (+inf is not enough: a row stamped in the future of `now` has
    // U = +inf, passes "+inf >= +inf", and the candidate of query -1 lands in front of the lists: corrupted
    // counters, then a device fault.  Found by a bank whose seeding and recall clocks disagreed.)
    thr[vs] = q >= 0 ? 0xff800000u : 0xffc00000u;
    eq_slot[vs] = q >= 0 ? eq_q[q] : 0.0f;                  // padding columns stay finite
    if (q >= 0) qslot[(int64_t)q * 8 + p] = (int32_t)vs;
}

// ---- per sorted row: the score constants (bank row id's bits in .w).  They depend on the bank and on `now`
//      only: aura_ivf2_row_constants lets the caller keep them across calls (aura_ivf2_append updates the
//      entries it touches), otherwise they are recomputed per call into the workspace ----
__global__ __launch_bounds__(256) void ivf2_rowc_kernel(const float* __restrict__ meta, const float* __restrict__ rho,
                                                        const int32_t* __restrict__ sorted_rows, int64_t Npad,
                                                        float now, float D, float4* __restrict__ rowc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Npad) return;
    rowc[i] = ivf2_row_constants(sorted_rows[i], meta, rho, now, D);
}

// ---- thresholds: one wave per query over the group maxima of its nprobe lists' sample tiles ----
template <int PER>                                         // keys per lane = stiles / 4
__global__ __launch_bounds__(256) void ivf2_threshold_kernel(const float* __restrict__ gmax,       // [slots][2 stiles]
                                                             const int32_t* __restrict__ qslot,    // [nq][8]
                                                             const int32_t* __restrict__ blk_list,
                                                             const int32_t* __restrict__ list_len,
                                                             int nprobe, int k, int nq,
                                                             uint32_t* __restrict__ thr,
                                                             int32_t* __restrict__ cnt_out,
                                                             int k2, float* __restrict__ bounds,     // staged recall: bounds[q] = {T_k, T_k2}
                                                             int bsh) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    // lane -> probe lane>>3, 8-row sample groups PER (lane&7) .. +PER-1 (two groups per sample tile)
    constexpr int STL = PER * 4;
    const int p = lane >> 3;
    uint32_t key[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) key[i] = 0u;
    int vs = -1;
    if (p < nprobe) vs = qslot[(int64_t)q * 8 + p];       // -1: the probe was dropped (id out of range / repeated)
    if (vs >= 0) {
        const int list = blk_list[vs >> bsh];
        const int tiles = (list_len[list] + 15) / 16;
        const int groups = 2 * (tiles < STL ? tiles : STL);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int g = PER * (lane & 7) + i;
            if (g < groups) key[i] = ord_key(gmax[(int64_t)vs * (2 * STL) + g]);
        }
    }
    uint32_t T = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = T | (1u << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) c += __popcll(__ballot(key[i] >= cand));
        if (c >= k) T = cand;
    }
    // fewer than k sampled groups (short lists): every real candidate row must pass, the padding
    // rows (U = -inf) must not
    // ... and a k-th largest maximum that belongs to a pad-only group (-inf) must not become the bound
    // either: "U >= -inf" would admit the padding rows, whose row id is -1
    if (T < 0x00800000u) T = 0x00800000u;                  // ord_key(-FLT_MAX): real rows pass, pads (-inf) do not
    if (vs >= 0 && (lane & 7) == 0) thr[vs] = T;
    if (lane == 0) cnt_out[(int64_t)q * CNT_STRIDE] = 0;
    if (bounds) {
        // k2-th largest sampled lower bound: at least k2 DISTINCT rows of this bank score >= it, so over S
        // disjoint shards the minimum of the shards' values is a lower bound of the (S k2)-th best score
        uint32_t T2 = T;
        if (k2 > 0 && k2 < k) {
            T2 = 0u;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t cand = T2 | (1u << bit);
                int c = 0;
#pragma unroll
                for (int i = 0; i < PER; ++i) c += __popcll(__ballot(key[i] >= cand));
                if (c >= k2) T2 = cand;
            }
            if (T2 < 0x00800000u) T2 = 0x00800000u;
        }
        if (lane == 0) { bounds[(int64_t)q * 2] = ord_unkey(T); bounds[(int64_t)q * 2 + 1] = ord_unkey(T2); }
    }
}

// Completion word (aura_knn_search_ivf2_signal): launched behind the call's last kernel, so the flag is final.
// (The first version counted finished workgroups inside the refine kernel: 2048 atomics on one address cost the
// launch 80 us.)
__global__ void ivf2_signal_kernel(const int32_t* __restrict__ flag, volatile uint32_t* host_word, uint32_t seq) {
    if (threadIdx.x == 0) {
        host_word[0] = (uint32_t)(flag ? *flag : 0);
        __threadfence_system();
        host_word[1] = seq;
        __threadfence_system();
    }
}

// staged recall, stage 2: thr[slot] = max(thr[slot], bound of the slot's query) (ordered keys: larger = larger)
__global__ __launch_bounds__(256) void ivf2_raise_thr_kernel(const int32_t* __restrict__ slotq,
                                                             const int32_t* __restrict__ nblk,
                                                             const float* __restrict__ bound_q,
                                                             uint32_t* __restrict__ thr, int bsh) {
    const int64_t vs = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if ((int)(vs >> bsh) >= nblk[0]) return;
    const int q = slotq[vs];
    if (q < 0) return;
    const float b = bound_q[q];
    if (!(b == b) || b == INFINITY) return;                 // never trust a NaN / +inf bound
    uint32_t kq = ord_key(b);
    if (kq < 0x00800000u) kq = 0x00800000u;
    if (kq > thr[vs]) thr[vs] = kq;
}

struct Ivf2Workspace {
    // shared with the lists / coarse paths
    float* inv_q; uint32_t* probe; float* probe_dist; int32_t* probe_ids;
    int32_t* lq_cnt; int32_t* lq_list; int32_t* qbase; int32_t* item_off_old; int32_t* work_counter;
    int32_t* cnt; float* cand_scores; int32_t* cand_idx;
    // two-stage inverted lists
    int32_t* blk_off; int32_t* blk_list; int32_t* blk_row0; int32_t* blk_stride; int32_t* blk_nq; int32_t* item_off; int32_t* sitem_off; int32_t* nblk;
    int32_t* slotq; int32_t* qslot; uint32_t* thr; float* gmax; uint16_t* qhat; float4* rowc;
    float* eq_slot; float* eq_q; int32_t* heavy;
    int cap; int qp; int64_t bytes;
};

static Ivf2Workspace carve_ivf2(void* base, int64_t Npad, int64_t nq, int k) {
    Ivf2Workspace w;
    char* p = static_cast<char*>(base);
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        char* r = p ? p + off : nullptr;
        off += align_up(bytes, 256);
        return r;
    };
    int64_t qp = nq < IVF2_MAXQ ? (nq > 0 ? nq : 1) : IVF2_MAXQ;
    w.qp = (int)qp;
    const int64_t mb = ivf2_maxblk(qp, 8);                 // mb * 256 query slots (the most either block size needs)
    const int64_t mbb = ivf2_maxblk(qp, 7);                // blocks (the most either block size needs)
    w.cap = RF_CAP;                                         // refine holds at most this many per query
    w.inv_q = reinterpret_cast<float*>(take(qp * 4));
    w.probe = reinterpret_cast<uint32_t*>(take(qp * 32));
    w.probe_dist = reinterpret_cast<float*>(take(qp * 256 * 4));
    w.probe_ids = reinterpret_cast<int32_t*>(take(qp * 8 * 4));
    w.lq_cnt = reinterpret_cast<int32_t*>(take(256 * 4));
    w.lq_list = reinterpret_cast<int32_t*>(take((int64_t)256 * IVF2_MAXQ * 4));
    w.qbase = reinterpret_cast<int32_t*>(take(qp * 8 * 4));
    w.item_off_old = reinterpret_cast<int32_t*>(take(257 * 4));
    w.work_counter = reinterpret_cast<int32_t*>(take(256));
    w.cnt = reinterpret_cast<int32_t*>(take(qp * CNT_STRIDE * 4));
    w.cand_scores = reinterpret_cast<float*>(take(qp * w.cap * 4));
    w.cand_idx = reinterpret_cast<int32_t*>(take(qp * w.cap * 4));
    w.blk_off = reinterpret_cast<int32_t*>(take(257 * 4));
    w.blk_list = reinterpret_cast<int32_t*>(take(mbb * 4));
    w.blk_row0 = reinterpret_cast<int32_t*>(take(mbb * 4));
    w.blk_stride = reinterpret_cast<int32_t*>(take(mbb * 4));
    w.blk_nq = reinterpret_cast<int32_t*>(take(mbb * 4));
    w.item_off = reinterpret_cast<int32_t*>(take((mbb + 1) * 4));
    w.sitem_off = reinterpret_cast<int32_t*>(take((mbb + 1) * 4));
    w.nblk = reinterpret_cast<int32_t*>(take(256));
    w.slotq = reinterpret_cast<int32_t*>(take(mb * 256 * 4));
    w.qslot = reinterpret_cast<int32_t*>(take(qp * 8 * 4));
    w.thr = reinterpret_cast<uint32_t*>(take(mb * 256 * 4));
    w.eq_slot = reinterpret_cast<float*>(take(mb * 256 * 4));
    w.eq_q = reinterpret_cast<float*>(take(qp * 4));
    w.gmax = reinterpret_cast<float*>(take(mb * 256 * 2 * ivf2_stiles(Npad, k) * 4));
    w.qhat = reinterpret_cast<uint16_t*>(take((qp + 1) * 768 * 2));   // one fragment set per query + the zero entry
    w.rowc = reinterpret_cast<float4*>(take((Npad > 0 ? Npad : 1) * 16));
    w.heavy = reinterpret_cast<int32_t*>(take((qp + 1) * 4));
    w.bytes = off;
    return w;
}

template <int KS, int NW, int QBT = 16 / NW>
inline int launch_coarse_ivf(const CoarseArgs& a, int mode, int grid, hipStream_t s) {
    constexpr int BLKQ = NW * 16 * QBT;
    const size_t lds = (size_t)cs_lds_slots<true, false, NW>() * (KS * 1024 + CS_AUX_BYTES) + (size_t)cs_cand_buf<NW, QBT>() * 12 +
                       BLKQ * 4 + BLKQ * 4;
    if (ensure_lds_attr(reinterpret_cast<const void*>(coarse_scan_kernel<KS, CS_MODE_SAMPLE, true, false, true, NW, QBT>), (int)lds) ||
        ensure_lds_attr(reinterpret_cast<const void*>(coarse_scan_kernel<KS, CS_MODE_FILTER, true, false, true, NW, QBT>), (int)lds))
        return AURA_E_LAUNCH;
    if (mode == CS_MODE_SAMPLE)
        hipLaunchKernelGGL((coarse_scan_kernel<KS, CS_MODE_SAMPLE, true, false, true, NW, QBT>), dim3(grid), dim3(64 * NW), lds, s, a);
    else
        hipLaunchKernelGGL((coarse_scan_kernel<KS, CS_MODE_FILTER, true, false, true, NW, QBT>), dim3(grid), dim3(64 * NW), lds, s, a);
    return check_launch();
}
