/*
 * aura_hip.h -- C ABI of libaura_hip.so: the MI355X (gfx950) implementation of Aura's
 * SNN-timestep + episodic-retrieval hot path.
 *
 * The reference (auralmn/aura-snn-rag) is 100 % Python/PyTorch and has no FFI of its own
 * (SURVEY.md section 0); the boundary it exposes for this path is the nn.Module API listed in
 * SURVEY.md section 8b.  Each entry point below names the reference method whose arithmetic it
 * replaces (file:line relative to the upstream checkout).  The Python classes in
 * aura_snn_rag_amd/ keep the reference's signatures and call these through ctypes
 * (INTEGRATION.md shows the binding a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless it says "host";
 *   - sizes are element counts, int64_t; tensors are dense row-major;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); every call is
 *     asynchronous on that stream and never synchronises or allocates;
 *   - the return value is 0 on success, a negative AURA_E_* code otherwise; nothing throws;
 *   - all arithmetic is IEEE fp32 without FMA contraction in the neuron loops, so results are
 *     bit-identical to the reference's unfused PyTorch op sequence.
 */
#ifndef AURA_HIP_H
#define AURA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AURA_OK 0
#define AURA_E_INVAL (-1)   /* bad argument (null pointer, negative size, unsupported value) */
#define AURA_E_LAUNCH (-2)  /* HIP reported a launch error */
#define AURA_E_ALIGN (-3)   /* pointer alignment requirement violated */

#define AURA_DTYPE_F32 0
#define AURA_DTYPE_BF16 1

/* gif flags */
#define AURA_GIF_TIME_INVARIANT 1 /* h is [rows, H]: the same current at every timestep */
#define AURA_GIF_MEAN_OUT 2       /* out is [rows, H] = mean over T of the spikes */

/* Library identification: returns a static string "aura_hip <version> gfx950". */
const char* aura_version(void);

/* ---------------------------------------------------------------------------------------
 * Spiking-neuron membrane loops
 * ------------------------------------------------------------------------------------- */

/* Izhikevich Euler loop, time-contiguous layout I[N][T] -> spikes[N][T]; v,u [N] in/out.
 * Replaces IzhikevichNeuron._jit_step_loop, src/base/neuron.py:181-196 (2-D / 1-D inputs of
 * forward_sequence, :162-167). */
int aura_izh_run_nt(const float* I, float* spikes, float* v, float* u, float a, float b, float c,
                    float d, float dt, int64_t N, int64_t T, void* stream);

/* Same loop, channel-contiguous layout I[B][T][D] -> spikes[B][T][D]; state index b*D+d.
 * Replaces the 3-D branch of forward_sequence (permute/reshape, src/base/neuron.py:158-160,
 * 176-178) without materialising the permuted copy. */
int aura_izh_run_btd(const float* I, float* spikes, float* v, float* u, float a, float b, float c,
                     float d, float dt, int64_t B, int64_t T, int64_t D, void* stream);

/* AdEx Euler loop; params = the 11-float buffer of src/base/neuron.py:207
 * {tau_m,E_L,V_T,Delta_T,R,tau_w,a,b,V_reset,V_spike,dt} passed BY VALUE from the host.
 * Replaces AdExNeuron._jit_step_loop, src/base/neuron.py:233-248. */
int aura_adex_run_nt(const float* I, float* spikes, float* V, float* w, const float* params_host,
                     int64_t N, int64_t T, void* stream);
int aura_adex_run_btd(const float* I, float* spikes, float* V, float* w, const float* params_host,
                      int64_t B, int64_t T, int64_t D, void* stream);

/* Vectorised LIF over x[B][T][size] (T = 1 is the single step of VectorizedLIFNeuron.forward,
 * src/base/neuron.py:131-139; T > 1 is the per-t loop of EnhancedSpikingNeuron.forward,
 * src/base/snn_brain_zones.py:73-79).  beta, threshold: [size]; mem: [B][size] in/out. */
int aura_lif_run(const float* x, float* spikes, float* mem, const float* beta,
                 const float* threshold, int64_t B, int64_t T, int64_t size, void* stream);

/* GIF membrane loop (decay, clamp, divide, floor-spike, soft reset, threshold adaptation).
 * h: currents after the neuron's nn.Linear, [rows][T][H] (or [rows][H] with
 * AURA_GIF_TIME_INVARIANT); out: spikes [rows][T][H] (or the T-mean [rows][H] with
 * AURA_GIF_MEAN_OUT); v, theta: [rows][H] in/out; dtype of h/out/v/theta = AURA_DTYPE_*.
 * In bf16 every op rounds to bf16, as the reference does (state dtype follows the input,
 * gif_neuron.py:46-47).  Replaces GIFNeuron.forward's loop,
 * src/core/language_zone/gif_neuron.py:54-69 (+ MultiBitSurrogate.forward :11-13) and the
 * spikes.mean(dim=1) readout of SNNFFN.forward, snn_ffn.py:81. */
int aura_gif_run(const void* h, void* out, void* v, void* theta, float decay, int L, float alpha,
                 float threshold, int64_t rows, int64_t T, int64_t H, int dtype, int flags,
                 void* stream);

/* ---------------------------------------------------------------------------------------
 * Episodic bank (HippocampalFormation)
 * ------------------------------------------------------------------------------------- */

/* inv_norm[i] = 1 / max(||bank[row0+i]||_2, 1e-12) for i in [0, n): the per-row factor of
 * F.normalize(active_feats, dim=1), src/core/hippocampal.py:278, hoisted out of the query. */
int aura_bank_row_norms(const float* bank, float* inv_norm, int64_t row0, int64_t n, int64_t D,
                        void* stream);

/* One-shot write of n rows: bank[slots[i]] = feats[i]; loc[slots[i]] = cur_loc;
 * meta[slots[i]] = {1, now, cid, 0}; inv_norm[slots[i]] refreshed.  With centroids != NULL
 * each row is assigned to its nearest centroid among the first `eff_k` (L2), the centroid's
 * running mean and count are updated IN ROW ORDER (c <- (1-1/n)c + (1/n)x), and cid is stored;
 * otherwise cid = -1.  slots: int64 [n] device.  Replaces create_episodic_memory's tensor
 * work, src/core/hippocampal.py:211-232. */
int aura_bank_write(float* bank, float* loc, float* meta, float* inv_norm, float* centroids,
                    float* centroid_counts, int eff_k, const float* feats, const int64_t* slots,
                    const float* cur_loc, int spatial_dims, float now, int64_t n, int64_t D,
                    void* stream);

/* aura_bank_write with centroids != NULL for batches whose slots are all DISTINCT, same results bit for
 * bit: the distances leave the serial chain -- every row is scored against the table as it stands at
 * batch start in parallel, then one workgroup walks the rows in order and re-scores (with the serial
 * arithmetic) only the centroids that earlier rows of the batch moved close enough to matter
 * (hippocampal.py:218-230 is order dependent: row i sees the centroids left by rows < i).
 * workspace: aura_bank_write_online_workspace_bytes(n) bytes, 256-byte aligned. */
int64_t aura_bank_write_online_workspace_bytes(int64_t n);
int aura_bank_write_online(float* bank, float* loc, float* meta, float* inv_norm, float* centroids,
                           float* centroid_counts, int eff_k, const float* feats, const int64_t* slots,
                           const float* cur_loc, int spatial_dims, float now, int64_t n, int64_t D,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* meta[i][0] *= (1 - rate) for i < count.  Replaces decay_memories, hippocampal.py:334. */
int aura_bank_decay(float* meta, float rate, int64_t count, void* stream);

/* Workspace size (bytes) aura_knn_search needs for (N, nq, k). */
int64_t aura_knn_workspace_bytes(int64_t N, int64_t nq, int k);

/* Exact batched recall: for each of nq queries the top-k rows of bank[0..N) by the reference's
 * combined score (0.5*cos + 0.3*spatial + 0.2*exp(-(now-ts)/3600)) * strength, descending,
 * ties -> lower row index.  q_loc == NULL means "no location" (spatial = 0).
 * Outputs: out_scores [nq][k] fp32, out_idx [nq][k] int32 (row index + idx_base).
 * k <= min(N, 1024).  workspace: aura_knn_workspace_bytes(N, nq, k) bytes of HBM.
 * Replaces retrieve_similar_memories steps 1-5, src/core/hippocampal.py:272-307, for a batch
 * of queries (the reference is called once per batch row,
 * memory_augmented_layer.py:113-121). */
int aura_knn_search(const float* bank, const float* inv_norm, const float* meta, const float* loc,
                    int spatial_dims, const float* queries, const float* q_loc, float now,
                    int64_t N, int64_t D, int64_t nq, int k, int32_t idx_base, float* out_scores,
                    int32_t* out_idx, void* workspace, int64_t workspace_bytes, void* stream);

/* As aura_knn_search with extras: centroids != NULL restricts each query's candidates to the rows
 * whose centroid id (meta[.][2]) is among the `nprobe` nearest (L2, unnormalised query) of the
 * 256 centroid rows -- the candidate selection of retrieve_similar_memories,
 * src/core/hippocampal.py:259-270 (a query left without candidates returns idx -1 everywhere and
 * the caller falls back to the full scan, :269-270); flags (AURA_KNN_FORCE_DENSE scores every row densely instead
 * of using the sampled-threshold filter; same results, used by tests) and overflow_out (device
 * int32, reset by every call and set non-zero if a candidate list overflowed -- bit 0: filter
 * list of the fp32 scan, bit 2 / bit 3: candidate / survivor list of the two-stage path, bit 4: a
 * candidate row id outside [0, N) reached the re-scoring stage (an internal invariant broke; the row
 * is dropped instead of dereferenced) -- the caller must then re-run with AURA_KNN_FORCE_DENSE; may
 * be NULL). */
/* overflow_out bit 6: some query of a two-stage call ended with NO candidate row at all (its probed
 * centroids own no rows): its outputs are -inf / -1 and the caller applies the reference's full-scan
 * fallback (hippocampal.py:269-270).  Not an overflow: the other queries' results are complete. */
#define AURA_KNN_FLAG_NO_CANDIDATES 64
/* overflow_out bit 7 (aura_knn_search_ivf2): the caller's lists_flag was set -- aura_ivf2_append dropped
 * a row because its list had no free entry left: the lists must be re-packed and the call repeated. */
#define AURA_KNN_FLAG_LISTS_STALE 128
#define AURA_KNN_FORCE_DENSE 1
/* AURA_KNN_FP32_SCAN: score every row on the fp32 matrix pipe.  Without it, large banks
 * (>= 8192 rows, D <= 768, D % 4 == 0, no location term / centroid mask, k <= 256) are first
 * filtered by a bf16 scan whose error is bounded rigorously (round-to-nearest bf16 of both
 * operands: |cos_bf16 - cos_fp32| <= rho_row + rho_query + rho_row rho_query + 2 D 2^-24 + 1e-5
 * with rho <= 2^-8 the relative L2 rounding residuals), and only rows that can still reach the top
 * k are re-scored in fp32 with the same arithmetic: results are bit-identical either way. */
#define AURA_KNN_FP32_SCAN 2
int aura_knn_search_ex(const float* bank, const float* inv_norm, const float* meta,
                       const float* loc, int spatial_dims, const float* queries,
                       const float* q_loc, float now, int64_t N, int64_t D, int64_t nq, int k,
                       int32_t idx_base, float* out_scores, int32_t* out_idx, void* workspace,
                       int64_t workspace_bytes, int flags, int32_t* overflow_out,
                       const float* centroids, int nprobe, void* stream);

/* Inverted-list (IVF) form of the centroid-candidate recall: identical results to
 * aura_knn_search_ex(..., centroids, nprobe) without a location term, but each probed list is
 * streamed once per pass of up to 2048 queries and only against the queries that probe it.
 * list_rows: row ids of bank[0..N) grouped by centroid id (int32 [N], rows with centroid id < 0
 * first); list_off [257]: start of each list in list_rows (list_off[256] = N); list_len [256].
 * The host module derives the three arrays from meta[.][2] after writes / rebuilds.  nprobe <= 8.
 * cap: candidate slots per query, a multiple of 2048 with (cap/2048)*k <= 16384; the sum of the
 * nprobe longest lists never overflows it.  A query whose lists hold more rows sets *overflow_out
 * (the caller falls back to the masked full scan).
 * workspace: aura_knn_ivf_workspace_bytes(nq, k, cap) bytes.
 * Replaces retrieve_similar_memories steps 0-5, src/core/hippocampal.py:259-307. */
int64_t aura_knn_ivf_workspace_bytes(int64_t nq, int k, int cap);
int aura_knn_search_ivf(const float* bank, const float* inv_norm, const float* meta,
                        const float* queries, float now, int64_t N, int64_t D, int64_t nq, int k,
                        const float* centroids, int nprobe, const int32_t* list_rows,
                        const int32_t* list_off, const int32_t* list_len, int cap,
                        int32_t idx_base, float* out_scores, int32_t* out_idx, void* workspace,
                        int64_t workspace_bytes, int32_t* overflow_out, void* stream);

/* Measurement hooks (bench.py): between aura_profile_begin(max) and aura_profile_end, every
 * launch of the main scan kernel inside aura_knn_search[_ex] is bracketed by HIP events recorded
 * on the launch stream.  aura_profile_end synchronises those events, writes up to max_out
 * per-launch durations in milliseconds to the HOST array and returns how many it wrote
 * (negative AURA_E_* on error). */
int aura_profile_begin(int max_launches);
/* Tuning hook (tools/ab_headline.py): replaces the AURA_CS_DBG ablation flags of the scan kernels for the
 * following calls (flags < 0: only read) and returns the previous value.  Non-zero flags switch kernel phases
 * off or on for timing experiments; results are only valid with 0, the default. */
int aura_debug_cs_flags(int flags);
/* Tuning hook: the shader clock in MHz as a kernel sees it (s_memtime ticks per 100-MHz s_memrealtime tick over
 * spin_us microseconds, 1..100000), written to out_dev[0] (device float). */
int aura_debug_clock_mhz(float* out_dev, int spin_us, void* stream);
int aura_profile_end(float* ms_out_host, int max_out);
/* Bank rows and queries scored by the most recent profiled main-scan launch (HOST pointers):
 * the units behind bench.py's algorithmic FLOP count, 2 * rows * nq * D per launch. */
int aura_profile_last_scan(int64_t* rows_out, int64_t* nq_out);
/* 0 = the profiled launch was the fp32 matrix scan, 1 = the bf16 prefilter scan over the fp32 bank,
 * 2 = the prefilter scan over the bf16 shadow, 3 = the inverted-list prefilter scan (sorted shadow). */
int aura_profile_last_scan_kind(void);

/* Merge S per-shard top-k lists into the global top-k (the step after the RCCL all-gather,
 * SURVEY.md section 8e).  in_scores/in_idx: [S][nq][k]; out: [nq][k]; ties -> lower index. */
int aura_topk_merge(const float* in_scores, const int32_t* in_idx, int S, int64_t nq, int k,
                    float* out_scores, int32_t* out_idx, void* stream);

/* k-means-lite pieces of rebuild_centroids, src/core/hippocampal.py:358-376.
 * assign: assign_out[i] = argmin_c ||bank[i] - centroids[c]|| over the first k centroids
 *   (computed as argmin |c|^2 - 2 x.c on the fp32 matrix cores; ties -> lower c);
 *   cnorm2_ws: k floats of scratch.
 * segment_means: the masked means of :358-363 as a segmented reduction.  The caller groups the rows
 *   by cluster (a stable sort of assign): order [N] int32 row ids, seg_off [k+1] int32 with
 *   order[seg_off[c] .. seg_off[c+1]) = rows of cluster c.  centroids[c] = mean of those rows, summed
 *   in a fixed order (reproducible); empty clusters keep their centroid (:362-363).  With sums_only
 *   the per-cluster SUMS are written instead (zeros for empty clusters): a rank's partial result
 *   of a row-sharded bank, all-reduced with the counts.  Reads the bank once.
 *   D % 4 == 0; workspace: aura_kmeans_means_workspace_bytes(N, D, k) bytes, 16-byte aligned.
 * commit: meta[i][2] = assign[i] for i < N and counts[c] = seg_off[c+1] - seg_off[c] (counts may be
 *   NULL) -- the recount / metadata write of :370-376. */
int aura_kmeans_assign(const float* bank, const float* centroids, float* cnorm2_ws,
                       int32_t* assign_out, int64_t N, int64_t D, int k, void* stream);
int64_t aura_kmeans_means_workspace_bytes(int64_t N, int64_t D, int k);
int aura_kmeans_segment_means(const float* bank, const int32_t* order, const int32_t* seg_off, float* centroids,
                              void* workspace, int64_t workspace_bytes, int64_t N, int64_t D, int k,
                              int sums_only, void* stream);
int aura_kmeans_commit(const int32_t* assign, const int32_t* seg_off, float* meta, float* counts, int64_t N,
                       int k, void* stream);

/* Gather rows: out[i] = bank[idx[i]] (idx outside [0, rows) -> zeros; rows = rows of the bank); used
 * to return [B,k,D] memory features (memory_augmented_layer.py:124-128). */
int aura_bank_gather(const float* bank, int64_t rows, const int32_t* idx, float* out, int64_t n, int64_t D,
                     void* stream);

/* Optional bf16 shadow of the bank for the two-stage recall's prefilter (halves the bytes the
 * prefilter streams; results unchanged: the survivors are still re-scored from the fp32 bank).
 * [build-side] no upstream counterpart; the product keeps it beside memory_features
 * (src/core/hippocampal.py:88) exactly as it keeps 1/||row||.
 * A shadow row is the NORMALISED row rounded to bf16: bank_bf16[r] = bf16(bank[r] * inv_norm[r]);
 * rho[r] (fp32, [rows of the bank]) is an upper bound of its L2 rounding residual against the unit
 * row -- the row's part of the prefilter's error bound, measured instead of assumed (<= 2^-8).
 * aura_bank_shadow_update: rows r in slots[0..n) (device int64) or, with slots == NULL,
 * [row0, row0 + n); inv_norm must be current for them.  D % 8 == 0, both bases 16-byte aligned. */
int aura_bank_shadow_update(const float* bank, const float* inv_norm, uint16_t* bank_bf16, float* rho,
                            const int64_t* slots, int64_t row0, int64_t n, int64_t D, void* stream);

/* aura_knn_search_ex without location term, with the shadow (bank_bf16 == NULL: same as
 * aura_knn_search_ex).  bank_bf16 / rho must be current (aura_bank_shadow_update) for every r < N.
 * centroids / nprobe as in aura_knn_search_ex: with the shadow the centroid-candidate restriction
 * (hippocampal.py:259-270) is applied inside the two-stage scan (probe masks in LDS) instead of the
 * fp32 scan. */
int aura_knn_search_shadow(const float* bank, const uint16_t* bank_bf16, const float* rho, const float* inv_norm,
                           const float* meta, const float* queries, float now, int64_t N, int64_t D,
                           int64_t nq, int k, int32_t idx_base, float* out_scores, int32_t* out_idx,
                           void* workspace, int64_t workspace_bytes, int flags, int32_t* overflow_out,
                           const float* centroids, int nprobe, void* stream);

/* Centroid-index recall (hippocampal.py:259-270) as inverted lists on the two-stage machinery: the
 * caller keeps a LIST-SORTED bf16 shadow -- sorted row i holds the shadow row of bank row
 * sorted_rows[i] (-1: no row, scanned as padding), the rows of one centroid contiguous.
 * pad_off [257]: first sorted row of each list, multiples of 16 (pad_off[256] <= n_sorted);
 * list_len [256]: entries in use, list_len[c] <= pad_off[c+1] - pad_off[c] -- the difference is
 * slack that aura_ivf2_append fills after writes, so the lists are re-packed only when the centroids
 * are rebuilt; every entry beyond list_len[c] must be -1.
 * aura_bank_shadow_sorted converts all n_sorted rows (and refreshes rho[row]; pos_of_row, if not NULL,
 * receives the reverse map bank row -> sorted row for aura_ivf2_append).
 * aura_ivf2_append: after a write of n DISTINCT bank rows `slots` (device int64) whose centroid ids
 * are in meta[.][2]: the row's previous entry becomes a hole (-1), the row is appended to its list
 * (bf16 row converted, rho refreshed, pos_of_row updated).  A row whose list has no free entry left is
 * dropped from the lists and *flag |= 1: pass that flag to aura_knn_search_ivf2 as lists_flag (it is
 * reported as AURA_KNN_FLAG_LISTS_STALE) or re-pack after at most `slack` appended rows.
 * aura_knn_search_ivf2: each probed list is streamed once per 256 of the queries that probe it, in
 * passes of up to 8192 queries; results (rows, score bits) equal aura_knn_search_ivf's.  N = rows of
 * the bank (every sorted_rows entry is < N).  D % 8 == 0, D <= 768, k <= 256, nprobe <= 8;
 * overflow_out as in aura_knn_search_ex (non-zero: re-run aura_knn_search_ivf or the masked scan).
 * aura_centroid_probe: ids_out[q][p] (p < nprobe, [nq][8] int32) = the p-th nearest of the 256 centroid
 * rows to query q (L2 on the unnormalised query, ties to the lower row: hippocampal.py:261-262), exactly
 * what aura_knn_search_ivf2 computes for itself; workspace: aura_centroid_probe_workspace_bytes(nq)
 * bytes, 256-byte aligned.  aura_knn_search_ivf2_probed takes such ids (of the same queries and
 * centroid table) instead of recomputing them: a bank sharded over ranks that share one centroid table
 * probes every query once, on the rank that owns it, not once per rank.
 * row_constants (optional, [n_sorted][4] fp32, 16-byte aligned): the sorted rows' score constants
 * {0.5 strength, temporal term + error, temporal term - error, bank row id bits} -- the per-row half of
 * hippocampal.py:290-303's combined score as the prefilter bounds it.  They depend on the bank's metadata,
 * rho, the list layout and `now` only, so a caller may build them once with aura_ivf2_row_constants(now)
 * and pass them to every search that uses the SAME `now` (fp32: the reference's fp32 timestamps give it a
 * 128-second grain) until metadata or layout change; aura_ivf2_append keeps a table current for the
 * entries it touches (pass the table and its `now`; NULL: no table).  NULL in the search: computed per
 * call into the workspace. */
int aura_ivf2_row_constants(const float* meta, const float* rho, const int32_t* sorted_rows, int64_t n_sorted,
                            int64_t D, float now, float* row_constants, void* stream);
int64_t aura_knn_ivf2_workspace_bytes(int64_t n_sorted, int64_t nq, int k);
int64_t aura_centroid_probe_workspace_bytes(int64_t nq);
int aura_centroid_probe(const float* centroids, const float* queries, int64_t D, int64_t nq, int nprobe,
                        int32_t* ids_out, void* workspace, int64_t workspace_bytes, void* stream);
int aura_bank_shadow_sorted(const float* bank, const float* inv_norm, const int32_t* sorted_rows,
                            uint16_t* sorted_bf16, float* rho, int32_t* pos_of_row, int64_t n_sorted, int64_t D,
                            void* stream);
int aura_ivf2_append(const float* bank, const float* inv_norm, const float* meta, const int64_t* slots,
                     int64_t n, int64_t D, uint16_t* sorted_bf16, int32_t* sorted_rows, const int32_t* pad_off,
                     int32_t* list_len, int32_t* pos_of_row, float* rho, int32_t* flag, float* row_constants,
                     float row_constants_now, void* stream);
int aura_knn_search_ivf2(const float* bank, const float* inv_norm, const float* meta,
                         const uint16_t* sorted_bf16, const float* rho, const int32_t* sorted_rows,
                         const int32_t* pad_off, const int32_t* list_len, const int32_t* lists_flag,
                         const float* row_constants,
                         int64_t n_sorted, int64_t N, const float* queries, float now, int64_t D, int64_t nq, int k,
                         const float* centroids, int nprobe, int32_t idx_base, float* out_scores,
                         int32_t* out_idx, void* workspace, int64_t workspace_bytes, int32_t* overflow_out,
                         void* stream);
int aura_knn_search_ivf2_probed(const float* bank, const float* inv_norm, const float* meta,
                                const uint16_t* sorted_bf16, const float* rho, const int32_t* sorted_rows,
                                const int32_t* pad_off, const int32_t* list_len, const int32_t* lists_flag,
                                const float* row_constants,
                                int64_t n_sorted, int64_t N, const float* queries, float now, int64_t D, int64_t nq,
                                int k, const float* centroids, int nprobe, const int32_t* probe_ids, int32_t idx_base,
                                float* out_scores, int32_t* out_idx, void* workspace, int64_t workspace_bytes,
                                int32_t* overflow_out, void* stream);

/* aura_knn_search_ivf2[_probed] (probe_ids may be NULL) that also tells the HOST when the call is done and what
 * its flag is, without a device-to-host copy: host_word is 64 bytes of host-mapped memory from
 * aura_host_word_alloc; a one-thread launch behind the call's last kernel stores the flag into host_word[0] and then
 * host_seq into host_word[1].  The caller polls host_word[1] == host_seq (choose a new host_seq per call) and reads
 * the flag from host_word[0]; results are then final on the stream as after any launch.  One call in flight per
 * host_word.  (The flag read costs the reference-style caller a stream synchronisation per recall otherwise:
 * hippocampal.py's retrieval is synchronous too, every .item() in :311-317 waits for the GPU.) */
int aura_host_word_alloc(void** host_word_out);
int aura_host_word_free(void* host_word);
/* The completion signal of aura_knn_search_ivf2_signal on its own: a one-thread launch on `stream` stores *flag_dev
 * (0 if NULL) into host_word[0], then host_seq into host_word[1].  For call chains that end in another entry point
 * (the staged recall): poll instead of a stream synchronisation. */
int aura_signal_flag(const int32_t* flag_dev, uint32_t* host_word, uint32_t host_seq, void* stream);
int aura_knn_search_ivf2_signal(const float* bank, const float* inv_norm, const float* meta,
                                const uint16_t* sorted_bf16, const float* rho, const int32_t* sorted_rows,
                                const int32_t* pad_off, const int32_t* list_len, const int32_t* lists_flag,
                                const float* row_constants,
                                int64_t n_sorted, int64_t N, const float* queries, float now, int64_t D, int64_t nq,
                                int k, const float* centroids, int nprobe, const int32_t* probe_ids, int32_t idx_base,
                                float* out_scores, int32_t* out_idx, void* workspace, int64_t workspace_bytes,
                                int32_t* overflow_out, uint32_t* host_word, uint32_t host_seq, void* stream);

/* aura_knn_search_ivf2[_probed] in stages, for a bank that is row-sharded over ranks: the
 * prefilter's threshold of a query is a lower bound of its k-th best score, and bounds found on different
 * shards can be combined before any shard filters -- every shard then keeps about 1/S of the candidates
 * and survivors it would keep against its own bound.
 *   stage 1: everything up to the sampled bounds; bounds[q][0] = the k-th, bounds[q][1] = the k2-th
 *            (1 <= k2 <= k; 0: same as k) largest sampled lower bound of query q on THIS bank: at least k
 *            (k2) distinct rows of this bank score at least that.
 *   stage 2: same arguments, same workspace (untouched in between), bounds[q] = ONE float per query: any valid
 *            lower bound of the query's global k-th best score, e.g. max(max over shards of bounds[.][0],
 *            min over shards of bounds[.][1]) with S k2 >= k; thresholds are raised to it, then filter + refine.
 *   stage 4 + stage 3 (instead of stage 2, a second combination over the shards): stage 4 is stage 2 up to the
 *            filter scan; bounds (room for [nq][2]) carries the combined bound in its first nq floats on entry and
 *            the FILTERED CANDIDATES' bounds on return -- bounds[q][0] = the k-th, [q][1] = the k2-th largest lower
 *            bound among this bank's candidates of query q (-inf where there are fewer).  Combined like stage 1's
 *            they are close to the global k-th best score itself.  Stage 3: bounds[q] = that combination (one
 *            float per query); the refine re-scores only candidates whose upper bound reaches it, so a shard
 *            returns the rows of its top k that can be in the global top k and -1 for the rest.
 * probe_ids may be NULL (probes computed in stage 1).  nq <= 8192 per staged call. */
int aura_knn_search_ivf2_staged(const float* bank, const float* inv_norm, const float* meta,
                                const uint16_t* sorted_bf16, const float* rho, const int32_t* sorted_rows,
                                const int32_t* pad_off, const int32_t* list_len, const int32_t* lists_flag,
                                const float* row_constants,
                                int64_t n_sorted, int64_t N, const float* queries, float now, int64_t D, int64_t nq,
                                int k, const float* centroids, int nprobe, const int32_t* probe_ids, int32_t idx_base,
                                float* out_scores, int32_t* out_idx, void* workspace, int64_t workspace_bytes,
                                int32_t* overflow_out, int stage, int k2, float* bounds, void* stream);

/* ---------------------------------------------------------------------------------------
 * Surrogate-gradient training path and the prosody-modulated GIF (fp32, [rows][T][H] layout)
 * ------------------------------------------------------------------------------------- */

/* GIF loop for training: as aura_gif_run (fp32, no flags) and additionally saves, per step, the
 * pre-clamp membrane potential (save_a) and the threshold the step started from (save_theta),
 * both [rows][T][H].  Forward of GIFNeuron.forward when autograd is recording,
 * src/core/language_zone/gif_neuron.py:54-69. */
int aura_gif_train_forward(const float* h, float* spikes, float* v, float* theta, float* save_a,
                           float* save_theta, float decay, int L, float alpha, float threshold,
                           int64_t rows, int64_t T, int64_t H, void* stream);

/* Backward of that loop (BPTT over T): g_spikes [rows][T][H] in; g_h [rows][T][H] out; g_v,
 * g_theta [rows][H]: in = gradients of the final state, out = gradients of the initial state.
 * Spike gradient = MultiBitSurrogate.backward (triangular window, gif_neuron.py:16-22); the
 * clamp with tensor bounds routes the gradient of clamped values to theta, as autograd does. */
int aura_gif_backward(const float* save_a, const float* save_theta, const float* g_spikes, float* g_h,
                      float* g_v, float* g_theta, float decay, int L, float alpha, float threshold,
                      int64_t rows, int64_t T, int64_t H, void* stream);

/* The same pair for bf16 tensors (bit patterns, uint16_t): forward = aura_gif_run's per-op-rounded bf16
 * loop (bit-identical spikes and state; save_a / save_theta are the bf16 values the reference's graph
 * holds); backward = the fp32 BPTT evaluated from them with the forward's roundings re-applied to
 * b, d, n, s, gradients carried in fp32 over the T steps and rounded to bf16 on the way out (the
 * reference's bf16 autograd rounds every intermediate gradient instead).  L <= 256. */
int aura_gif_train_forward_bf16(const uint16_t* h, uint16_t* spikes, uint16_t* v, uint16_t* theta,
                                uint16_t* save_a, uint16_t* save_theta, float decay, int L, float alpha,
                                float threshold, int64_t rows, int64_t T, int64_t H, void* stream);
int aura_gif_backward_bf16(const uint16_t* save_a, const uint16_t* save_theta, const uint16_t* g_spikes,
                           uint16_t* g_h, uint16_t* g_v, uint16_t* g_theta, float decay, int L, float alpha,
                           float threshold, int64_t rows, int64_t T, int64_t H, void* stream);

/* LIF step for training: spikes, mem_out as aura_lif_run (T = 1) plus pre = beta*mem + x - thr
 * (the surrogate's input), all [B][size]; mem_in is not modified. */
int aura_lif_train_forward(const float* x, const float* mem_in, const float* beta,
                           const float* threshold, float* spikes, float* mem_out, float* pre,
                           int64_t B, int64_t size, void* stream);

/* Backward of the LIF step with LearnableSurrogateGradient.backward (src/base/neuron.py:80-108):
 * g_x, g_mem_prev [B][size] and raw_slope [B][size] (sum it over the batch to get d/d slope). */
int aura_lif_backward(const float* pre, const float* g_spikes, const float* g_mem, const float* beta,
                      const float* threshold, const float* slope, float* g_x, float* g_mem_prev,
                      float* raw_slope, int64_t B, int64_t size, void* stream);

/* ProsodyModulatedGIF.forward loop, src/core/language_zone/prosody_gif.py:69-106: gains [rows][T]
 * (NULL = no modulation) scale the input, the effective threshold (x clamp(1 - strength*(g-1),
 * 0.5, 1.5)) and the adaptation rate; no 1e-6 in the division (unlike GIFNeuron). */
int aura_gif_prosody_run(const float* h, const float* gains, float* spikes, float* v, float* theta,
                         float decay, int L, float alpha, float threshold, float strength,
                         int64_t rows, int64_t T, int64_t H, void* stream);

/* Training forward / backward of that loop: as aura_gif_train_forward / aura_gif_backward with the
 * gains; additionally g_gains [rows][T] receives dL/d gains (ZEROED by the caller; channels are summed
 * with float atomics): through the input gain, the threshold scale (where its clamp is inactive) and the
 * adaptation rate.  h = the currents of the forward pass.  Backward of prosody_gif.py:69-106 with
 * MultiBitSurrogate (gif_neuron.py:16-22). */
int aura_gif_prosody_train_forward(const float* h, const float* gains, float* spikes, float* v, float* theta,
                                   float* save_a, float* save_theta, float decay, int L, float alpha,
                                   float threshold, float strength, int64_t rows, int64_t T, int64_t H,
                                   void* stream);
int aura_gif_prosody_backward(const float* save_a, const float* save_theta, const float* h, const float* gains,
                              const float* g_spikes, float* g_h, float* g_gains, float* g_v, float* g_theta,
                              float decay, int L, float alpha, float threshold, float strength, int64_t rows,
                              int64_t T, int64_t H, void* stream);

/* bf16 forms of the LIF step and of the prosody GIF (bit patterns, uint16_t) for modules moved to bf16 with
 * .bfloat16() / .to(torch.bfloat16): buffers, parameters, input, state AND gains are bf16 and every op of
 * src/base/neuron.py:135-139 / src/core/language_zone/prosody_gif.py:64-101 rounds its fp32 result to bf16
 * (Python scalars enter as fp32), which the forward kernels reproduce bit for bit.  The backward kernels
 * evaluate the fp32 chains of aura_lif_backward / aura_gif_prosody_backward on the saved bf16 values with the
 * forward's roundings re-applied, carry state gradients in fp32 and round once on the way out; raw_slope and
 * g_gains stay fp32 (the caller reduces / rounds them).  L <= 256.  (A bf16 input to an fp32 module, or fp32
 * gains with a bf16 input, promote to fp32 in the reference: the host runs those on the fp32 entry points.) */
int aura_lif_run_bf16(const uint16_t* x, uint16_t* spikes, uint16_t* mem, const uint16_t* beta,
                      const uint16_t* threshold, int64_t B, int64_t T, int64_t size, void* stream);
int aura_lif_train_forward_bf16(const uint16_t* x, const uint16_t* mem_in, const uint16_t* beta,
                                const uint16_t* threshold, uint16_t* spikes, uint16_t* mem_out, uint16_t* pre,
                                int64_t B, int64_t size, void* stream);
int aura_lif_backward_bf16(const uint16_t* pre, const uint16_t* g_spikes, const uint16_t* g_mem, const uint16_t* beta,
                           const uint16_t* threshold, const uint16_t* slope, uint16_t* g_x, uint16_t* g_mem_prev,
                           float* raw_slope, int64_t B, int64_t size, void* stream);
int aura_gif_prosody_run_bf16(const uint16_t* h, const uint16_t* gains, uint16_t* spikes, uint16_t* v, uint16_t* theta,
                              float decay, int L, float alpha, float threshold, float strength, int64_t rows,
                              int64_t T, int64_t H, void* stream);
int aura_gif_prosody_train_forward_bf16(const uint16_t* h, const uint16_t* gains, uint16_t* spikes, uint16_t* v,
                                        uint16_t* theta, uint16_t* save_a, uint16_t* save_theta, float decay, int L,
                                        float alpha, float threshold, float strength, int64_t rows, int64_t T,
                                        int64_t H, void* stream);
int aura_gif_prosody_backward_bf16(const uint16_t* save_a, const uint16_t* save_theta, const uint16_t* h,
                                   const uint16_t* gains, const uint16_t* g_spikes, uint16_t* g_h, float* g_gains,
                                   uint16_t* g_v, uint16_t* g_theta, float decay, int L, float alpha, float threshold,
                                   float strength, int64_t rows, int64_t T, int64_t H, void* stream);

/* ---------------------------------------------------------------------------------------
 * Brain-zone projection
 * ------------------------------------------------------------------------------------- */

/* out[b][o] = -sum_k |x[b][k] - weight_patterns[o][k]| (+ bias[o] if bias != NULL).
 * Replaces AdditionLinear.forward, src/maths/addition_linear.py:42-67, the in/out projection of
 * NeuromorphicBrainZone.forward (src/base/snn_brain_zones.py:139,161). */
int aura_addition_linear(const float* x, const float* weight_patterns, const float* bias,
                         float* out, int64_t B, int64_t in_features, int64_t out_features,
                         void* stream);

/* Backward of that projection, as autograd derives it from src/maths/addition_linear.py:50-64 (abs -> sign, with
 * sign(0) = 0):  g_x[b][k] = -sum_o g_out[b][o] sgn(x[b][k] - w[o][k]),  g_w[o][k] = +sum_b g_out[b][o] sgn(x[b][k] -
 * w[o][k]).  g_x [B][in] and g_w [out][in] are optional (NULL = not wanted); the bias gradient is g_out summed over
 * the batch (the caller's reduction). */
int aura_addition_linear_backward(const float* x, const float* weight_patterns, const float* g_out,
                                  float* g_x, float* g_w, int64_t B, int64_t in_features,
                                  int64_t out_features, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AURA_HIP_H */
