/*
 * pilot_ot.h -- C ABI of libpilot_ot.so, the MI355X (gfx950) pairwise-Wasserstein engine.
 *
 * Drop-in boundary.  The reference (CostaLab/PILOT, pilotpy 2.0.6) has no FFI layer: the hot
 * path sits behind two Python call signatures,
 *     wasserstein_d(Clu_rep, cost, regularized, reg)   pilotpy/tools/Trajectory.py:479-523
 *     cost_matrix(annot, data, metric)                 pilotpy/tools/Trajectory.py:441-475
 * whose inner arithmetic is one POT call per ordered sample pair
 *     ot.sinkhorn2(a, b, M, reg, method="sinkhorn_stabilized")     Trajectory.py:515
 *     ot.emd2(a, b, M)                                             Trajectory.py:511
 * and one scipy call  pdist(centroids, metric) + squareform        Trajectory.py:468-469.
 * The entry points below are what a ctypes binding for that path binds (INTEGRATION.md shows
 * the binding).  Plain pointers and sizes only; no torch / numpy types.
 *
 * Conventions
 *   - every function returns PILOT_OT_OK (0) or a negative PILOT_OT_E* code; the message of the
 *     last failure on the calling thread is pilot_ot_last_error().  No exceptions cross the ABI.
 *   - "host" entry points take caller-owned host buffers, copy in/out internally and retain no
 *     pointer to them after returning (device workspace is cached per thread: pilot_ot_shutdown).  "_dev" entry points take device pointers (hipMalloc'ed by the
 *     caller or by pilot_ot_dev_alloc) and enqueue on the given hipStream_t without synchronising.
 *   - there is NO CPU implementation behind this ABI: without a gfx950 device every compute entry
 *     point fails with PILOT_OT_EHIP.
 *   - matrices are row-major; P is N x K (one proportion vector per sample, rows sum to 1,
 *     Trajectory.py:428-430); M is K x K, already divided by its max (Trajectory.py:101).
 *   - rows of the pair grid are selected as row_begin, row_begin+row_step, ... < row_end; every
 *     selected row is paired with ALL N columns (diagonal included, no symmetry shortcut,
 *     Trajectory.py:508-515).  Outputs hold n_rows x N values, n_rows = ceil((row_end-row_begin)/row_step).
 */
#ifndef PILOT_OT_H
#define PILOT_OT_H

#ifdef __cplusplus
extern "C" {
#endif

#define PILOT_OT_VERSION 100 /* 0.1.0 */

/* return codes */
#define PILOT_OT_OK 0
#define PILOT_OT_EINVAL (-1)  /* bad argument                                              */
#define PILOT_OT_EHIP (-2)    /* HIP runtime error / no gfx950 device                      */
#define PILOT_OT_ENOTSUP (-3) /* shape outside what the kernels support (K > 2048, ...) */
#define PILOT_OT_ERCCL (-4)   /* RCCL error / librccl missing (multi-GPU entry points only) */

/* precision of the Sinkhorn pair-grid kernel */
#define PILOT_OT_MAX_COST_OVER_REG 600.0 /* beyond it exp(-M/reg) leaves the f64 range: every entry point runs PREC_GENERIC */
#define PILOT_OT_PREC_AUTO 0 /* F16X2 while max(M)/reg <= 16, BF16X3 (both f32 values) while exp(-max(M)/reg) stays a normal f32
                              * far from underflow (<= 60), else AUTO_MIXED / f64 */
#define PILOT_OT_PREC_F32 1    /* f32 values, products on the f32-input MFMA (v_mfma_f32_16x16x4_f32): IEEE f32 FMA chains */
#define PILOT_OT_PREC_F64 2
#define PILOT_OT_PREC_AUTO_MIXED 4 /* what AUTO resolves to beyond the f32 range (60 < max(M)/reg <= 140): every pair is iterated in
                                   * f32 (BF16X3 tau-tracking kernel) with the Gibbs kernel held in TWO EXPONENT BANDS (entries
                                   * below 2^-110 are kept times 2^128 in a second operand image, so exp(-M/reg) is faithful down
                                   * to exp(-165)); a pair that still goes NaN / inf is solved again in f64 (PILOT_OT_FLAG_F64
                                   * tells which).  Falls back to F64 where the images do not fit LDS or max(M)/reg > 140. */
#define PILOT_OT_PREC_GENERIC 5 /* reference-semantics fallback: POT's sinkhorn_stabilized loop literally in fp64, one workgroup per
                                * pair, absorbed kernel exp(-(M - alpha - beta)/reg) rebuilt at every tau-absorption.  Taken
                                * automatically when K > 256, when 128 < K <= 256 falls outside the range of the eight-waves-per-tile
                                * kernel (below), or max(M)/reg > 600 (where the fixed Gibbs image of the fast kernels leaves the
                                * f64 range); NaN handling is POT's (revert to the last good iterate). */
#define PILOT_OT_PREC_BF16X3 3 /* f32 values, products on v_mfma_f32_16x16x32_bf16 through exact 3-way bf16 operand splits
                                * (six piece products per term, f32 accumulation): f32-level rounding, not bit-identical
                                * to PREC_F32, ~2x its speed */
#define PILOT_OT_PREC_F16X2 6  /* f32 values, products on v_mfma_f32_16x16x32_f16 through 2-way fp16 operand splits (11 + 11
                                * significant bits, three piece products per term) in a fixed scaled domain (2^15 G, 32 u, 32 v):
                                * valid while max(M)/reg <= 16 and tau <= 2000 (PILOT's defaults: reg 0.1 on cost/max, tau 1e3);
                                * outside that range the call runs BF16X3.  Pairs in which POT would tau-absorb are redone by
                                * the BF16X3 tracking kernel.  Same stopping checks as BF16X3 on 99.9 % of the pairs. */

/* per-pair flag bits (flags output) */
#define PILOT_OT_FLAG_CONVERGED 1      /* stopped on err <= stop_thr                              */
#define PILOT_OT_FLAG_NAN 2            /* a scaling became NaN (POT: "Numerical errors"): the pair was re-solved by the
                                        * POT-literal kernel, emd = cost of the last good iterate like POT returns
                                        * (also: histograms with empty bins that reach a tau-absorption, where POT's
                                        * log(0) leads to 0/0 one update later; and pairs of unequal mass whose total
                                        * scalings leave the exact range of the fast kernels) */
#define PILOT_OT_FLAG_ABSORB_LAST 4    /* POT tau-absorption fell on the final update (plan /K^2)   */
#define PILOT_OT_FLAG_ABSORBED 8       /* at least one POT tau-absorption happened                  */
#define PILOT_OT_FLAG_F64 16           /* pair was solved by the f64 kernel                         */

/* ground metrics of pilot_ot_cost_matrix (scipy.spatial.distance.pdist names, Trajectory.py:468) */
#define PILOT_OT_METRIC_COSINE 0
#define PILOT_OT_METRIC_EUCLIDEAN 1
#define PILOT_OT_METRIC_SQEUCLIDEAN 2
#define PILOT_OT_METRIC_CITYBLOCK 3
#define PILOT_OT_METRIC_CHEBYSHEV 4
#define PILOT_OT_METRIC_CORRELATION 5
#define PILOT_OT_METRIC_MINKOWSKI 6   /* p = 2, scipy's default (the reference forwards only the metric name) */
#define PILOT_OT_METRIC_SEUCLIDEAN 7  /* V = per-dimension variance of the centroids, ddof = 1 (scipy's default) */
#define PILOT_OT_METRIC_BRAYCURTIS 8
#define PILOT_OT_METRIC_CANBERRA 9
#define PILOT_OT_METRIC_HAMMING 10
/* scipy's boolean dissimilarities (pdist converts the rows to bool: non-zero = True) and the rest of scipy 1.15's pdist names */
#define PILOT_OT_METRIC_JACCARD 11
#define PILOT_OT_METRIC_DICE 12           /* (evaluated on the values, like scipy: ntt = sum u v, ...) */
#define PILOT_OT_METRIC_YULE 13
#define PILOT_OT_METRIC_RUSSELLRAO 14
#define PILOT_OT_METRIC_SOKALSNEATH 15
#define PILOT_OT_METRIC_ROGERSTANIMOTO 16
#define PILOT_OT_METRIC_SOKALMICHENER 17
#define PILOT_OT_METRIC_KULCZYNSKI1 18
#define PILOT_OT_METRIC_JENSENSHANNON 19
#define PILOT_OT_METRIC_MAHALANOBIS 20    /* needs aux = VI, the D x D inverse covariance (cost_matrix_ex) */

/* ---- library / device --------------------------------------------------------------------- */
int pilot_ot_version(void);
const char *pilot_ot_last_error(void);
int pilot_ot_device_count(int *count);            /* number of visible HIP devices (0 is not an error) */
int pilot_ot_set_device(int device);              /* device used by the calling thread's later calls   */
int pilot_ot_get_device(int *device);             /* the calling thread's current device (a new host thread starts on device 0) */
int pilot_ot_device_name(char *buf, int buflen);  /* gcnArchName of the current device                 */
int pilot_ot_shutdown(void);                      /* free the calling thread's cached host-API workspace */

/* TEST HOOK, not part of the drop-in surface: forces a kernel variant or an out-of-range configuration for the GPU tests and the
 * A/B tools (names as the tests use them, e.g. "PILOT_OT_NO_SMALL_MEDIANS", "PILOT_OT_RAW_PRECISION"); some settings return results
 * OUTSIDE the stated tolerance -- that is what they are for.  Process-wide; value NULL clears one switch, name NULL clears all.
 * The library reads no environment variable that changes what it computes. */
int pilot_ot_test_switch(const char *name, const char *value);

/* thin device-memory helpers so a host language without a HIP binding can keep data resident */
int pilot_ot_dev_alloc(void **dptr, unsigned long long bytes);
int pilot_ot_dev_free(void *dptr);
int pilot_ot_memcpy_h2d(void *dst, const void *src, unsigned long long bytes);
int pilot_ot_memcpy_d2h(void *dst, const void *src, unsigned long long bytes);
int pilot_ot_stream_sync(void *stream);

/* ---- label columns -> codes (host only, no device work) -------------------------------------- */
/* Numbers the n labels of one column in order of first appearance: what annot.cell_type.unique() /
 * annot.sampleID.unique() plus one boolean mask per label amount to (Trajectory.py:402-425).
 * ids: the labels as fixed-width integers: id_bytes = 1, 2, 4 -- signed, negative = missing (the codes of a pandas
 *      Categorical, what AnnData stores for obs labels); id_bytes = 8 -- opaque 64-bit identities, 0 = missing (the object
 *      pointers of an object column: a cohort's 1.8 M labels are a few hundred distinct Python objects).
 * codes: n int32 out, -1 for missing; first_rows[j]: row where code j first appears (max_uniques entries);
 * *n_uniques: how many codes.  More than max_uniques distinct labels -> PILOT_OT_ENOTSUP.
 * n_threads >= 1 host threads share the pass (slices side by side, their first-appearance lists merged in order). */
int pilot_ot_label_codes(const void *ids, int id_bytes, long long n, int max_uniques, int n_threads, int *codes,
                         long long *first_rows, int *n_uniques);

/* ---- pre-pass: replaces the pandas loops of Cluster_Representations and cost_matrix ---------- */
/* cell_code / sample_code: per cell, index of its cell type / sample in FIRST-APPEARANCE order
 * (what Series.unique() yields, Trajectory.py:402,412); negative codes (missing values) are skipped.
 * n_total = len(df) (the prior's denominator is n_total - 1, Trajectory.py:407).
 * P: N x K fp64, bit-identical to the reference's dict values (every fp64 operation in the same order). */
int pilot_ot_proportions(const int *cell_code, const int *sample_code, long long n_cells, long long n_total,
                         int N, int K, double regulizer, int normalization, double *P);
/* first_row (nullable, N entries): additionally the smallest row number of every sample (-1: no row), i.e. the row whose
 * status return_real_labels reports (Trajectory.py:617-642) -- one more atomic in the same pass over the codes. */
int pilot_ot_proportions_ex(const int *cell_code, const int *sample_code, long long n_cells, long long n_total,
                            int N, int K, double regulizer, int normalization, double *P, long long *first_row);
/* Per-cell-type column-wise median of the n_cells x D embedding X (Trajectory.py:465-466), exact.
 * dtype: 0 = float32, 1 = float64 (the median of an even count is averaged in that dtype, like
 * numpy/pandas, then widened).  centroids: K x D fp64; a cell type without cells yields NaN. */
#define PILOT_OT_F32 0
#define PILOT_OT_F64 1
int pilot_ot_centroid_medians(const void *X, int dtype, long long n_cells, int D, const int *cell_code, int K,
                              double *centroids);
/* The same with the embedding already resident: pilot_ot_embedding_upload copies the C x D array to the current device
 * (synchronously -- call it from a helper thread to overlap the transfer with host work on the label columns, which is
 * what pilot_amd.tl does); _medians_dev then only moves the codes. */
typedef struct pilot_ot_embedding pilot_ot_embedding;
int pilot_ot_embedding_upload(const void *X, int dtype, long long n_cells, int D, pilot_ot_embedding **emb);
int pilot_ot_embedding_destroy(pilot_ot_embedding *emb);
int pilot_ot_centroid_medians_dev(pilot_ot_embedding *emb, const int *cell_code, int K, double *centroids);
/* The whole device pre-pass of wasserstein_distance from ONE upload of the two code columns (emb->C entries each):
 * proportions P (N x K), first rows (nullable) and centroids (K x D) as the three calls above give them, bit for bit --
 * Cluster_Representations (Trajectory.py:377-436), return_real_labels (:617-642), the medians of cost_matrix (:462-466). */
int pilot_ot_prepass_dev(pilot_ot_embedding *emb, const int *cell_code, const int *sample_code, long long n_total, int N, int K,
                         double regulizer, int normalization, double *P, long long *first_row, double *centroids);
/* Device time (HIP events on the launch stream, first kernel to last; transfers excluded) of the calling thread's last
 * pilot_ot_prepass_dev / pilot_ot_centroid_medians(_dev) call -- what bench.py's `prepass.roofline` is computed from. */
int pilot_ot_prepass_device_ms(float *ms);

/* ---- cost matrix: replaces scipy pdist+squareform at Trajectory.py:468-469 ------------------ */
/* centroids: K x D row-major (per-cell-type medians, Trajectory.py:465-466).  cost: K x K,
 * symmetric, zero diagonal, NOT normalised (the reference stores the raw matrix, :98-99). */
int pilot_ot_cost_matrix(const double *centroids, int K, int D, int metric, double *cost);
int pilot_ot_cost_matrix_dev(const double *d_centroids, int K, int D, int metric, double *d_cost,
                             void *stream);
/* aux: metric-specific extra input, NULL otherwise -- mahalanobis: VI = inv(cov(centroids^T))^T, D x D, which the host
 * computes like scipy does (numpy.linalg.inv) */
int pilot_ot_cost_matrix_ex(const double *centroids, int K, int D, int metric, const double *aux, double *cost);
int pilot_ot_cost_matrix_dev_ex(const double *d_centroids, int K, int D, int metric, const double *d_aux, double *d_cost,
                                void *stream);

/* ---- Sinkhorn pair grid: replaces the loop at Trajectory.py:512-515 ------------------------- */
/* Each pair follows POT 0.9.x sinkhorn_stabilized control flow (v-update then u-update; marginal
 * error ||Gamma^T 1 - b||_2 evaluated when ii % check_period == 0; stop on err <= stop_thr or
 * after num_iter_max updates) and returns <Gamma, M> like ot.sinkhorn2.  POT defaults:
 * num_iter_max=1000, stop_thr=1e-9, tau=1e3, check_period=20.  In f32 the stop threshold is
 * floored at f32_floor_ulps * FLT_EPSILON * ||b||_2 (pass 0 for the default of 8).
 * cost_is_symmetric: 1 if M == M^T exactly (always true for pdist output), 0 otherwise.
 * Range: K <= 128 and max(M)/reg <= 600 run on the one-wave-per-tile MFMA kernels (they keep total scalings against the fixed
 * exp(-M/reg), so the ratio must fit the f64 exponent range).  128 < K <= 256 with a symmetric cost, max(M)/reg <= 16 and
 * tau <= 2000 runs the fp16-split products with a tile's cell types spread over the eight waves of a workgroup (any f32-class
 * precision request, AUTO included; f32 tolerance; PILOT_OT_PREC_F64 / _GENERIC keep the POT-literal kernel).  Larger K
 * (<= 2048), the rest of 128 < K <= 256, or a smaller reg run PILOT_OT_PREC_GENERIC, whatever precision was asked for.  The device-resident form cannot see max(M): it judges the range by the plan's max_cost / reg, which is
 * 1/reg (M divided by its max, Trajectory.py:101) until pilot_ot_plan_set_max_cost says otherwise.
 * emd / iters / err / flags: n_rows x N; iters, err, flags may be NULL. */
int pilot_ot_sinkhorn_grid(const double *P, int N, int K, const double *M, double reg,
                           int num_iter_max, double stop_thr, double tau, int check_period,
                           int precision, double f32_floor_ulps, int cost_is_symmetric,
                           int row_begin, int row_end, int row_step,
                           double *emd, int *iters, double *err, int *flags);

/* Device-resident form.  A plan owns the device workspace for one (N, K) shape so the call itself
 * allocates nothing (HIP-graph capturable) -- with two exceptions, each ONE allocation made by the first call that needs it
 * and kept until pilot_ot_plan_destroy (so make that first call outside a stream capture): the scratch of the POT-literal
 * kernel (PREC_GENERIC, 2 K^2 doubles per resident workgroup), and the flow slab of the exact-OT kernels
 * (pilot_ot_emd_grid_dev: K^2 doubles per resident pair, sized once for every kernel variant of the plan's K). */
typedef struct pilot_ot_plan pilot_ot_plan;
int pilot_ot_plan_create(int N, int K, pilot_ot_plan **plan);
int pilot_ot_plan_destroy(pilot_ot_plan *plan);
/* max(M) of the cost matrix the caller keeps on the device (default 1: the cost divided by its maximum, what
 * Trajectory.py:101 hands to the pair loop).  Every range decision of pilot_ot_sinkhorn_grid_dev -- which precision AUTO
 * means, the fp16-split domain (max(M)/reg <= 16), the hand-over and two-band thresholds, the POT-literal fallback -- is
 * taken on max_cost / reg.  A caller whose M is not normalised MUST set it: with max(M) = 3 and reg = 0.1 the Gibbs entries
 * of the fp16-split images would underflow and the call would return finite but wrong distances. */
int pilot_ot_plan_set_max_cost(pilot_ot_plan *plan, double max_cost);
int pilot_ot_sinkhorn_grid_dev(pilot_ot_plan *plan, const double *d_P, const double *d_M, double reg,
                               int num_iter_max, double stop_thr, double tau, int check_period,
                               int precision, double f32_floor_ulps, int cost_is_symmetric,
                               int row_begin, int row_end, int row_step,
                               double *d_emd, int *d_iters, double *d_err, int *d_flags,
                               void *stream);
/* Per-launch kernel timing (HIP events recorded on the call's own stream, around the main pair-grid
 * kernel and around the tau-tracking kernel).  A ring of the 64 most recent sinkhorn_grid_dev calls is
 * kept; read it after synchronising the stream.  main_ms / track_ms receive up to max_n entries, oldest
 * first; *n_out = entries written.  enable = n > 1 records every n-th call only (the four event records of a call cost a
 * 0.7 ms call about 2 %). */
int pilot_ot_plan_enable_timing(pilot_ot_plan *plan, int enable);
int pilot_ot_plan_kernel_times(pilot_ot_plan *plan, int max_n, float *main_ms, float *track_ms, int *n_out);
/* hipGraph replay for a caller that repeats one sinkhorn_grid_dev call (same buffers and arguments; the CONTENTS of
 * P and M may change): the call's launch sequence (control-block memset, prep, order scatter, pair-grid kernel, tracking
 * kernel, NaN hand-over) is captured on the second identical call and replayed as one graph launch from the third on.
 * Any change of arguments falls back to ordinary launches and re-captures.  Off while kernel timing is enabled. */
int pilot_ot_plan_enable_graph(pilot_ot_plan *plan, int enable);

/* ---- exact OT pair grid: replaces the loop at Trajectory.py:507-511 (the reference default) ---- */
/* Each pair returns the exact transportation-LP optimum, the value ot.emd2(a, b, M) returns
 * (after POT's own pre-step b *= sum(a)/sum(b)).  fp64 throughout.
 * mode: PILOT_OT_EMD_ALL    every selected (row, column) pair is solved;
 *       PILOT_OT_EMD_UPPER  only pairs with column >= row are solved, the rest of emd is left
 *                           untouched (valid when M is symmetric AND all histograms carry the same mass -- emd2 rescales
 *                           b to the mass of a, so the value scales with sum(a): the caller mirrors);
 *       PILOT_OT_EMD_MIRROR like UPPER, then the lower triangle is filled from the upper one on the
 *                           device (requires the full square grid: rows 0..N step 1).
 * n_aug (nullable): augmenting paths used per pair (negative: iteration guard tripped, emd = NaN). */
#define PILOT_OT_EMD_ALL 0
#define PILOT_OT_EMD_UPPER 1
#define PILOT_OT_EMD_MIRROR 2
int pilot_ot_emd_grid(const double *P, int N, int K, const double *M, int mode,
                      int row_begin, int row_end, int row_step, double *emd, int *n_aug);
int pilot_ot_emd_grid_dev(pilot_ot_plan *plan, const double *d_P, const double *d_M, int mode,
                          int row_begin, int row_end, int row_step, double *d_emd, int *d_n_aug,
                          void *stream);

/* fill the strictly-lower triangle of the N x N device matrix from the upper one (what PILOT_OT_EMD_MIRROR runs) */
int pilot_ot_mirror_upper_dev(double *d_emd, int N, void *stream);

/* ---- multi-GPU: the pair grid row-sharded over the GPUs of one node ---------------------------------------------------
 * The N^2 pair problems of Trajectory.py:505-515 are independent given the replicated proportions and cost, so the
 * grid is partitioned, never exchanged: shard s of G solves rows s, s+G, s+2G, ... against all N columns; the only
 * exchange step is ONE all-gather of the row blocks (RCCL over xGMI) followed by a device-side row interleave.  A pair
 * is solved by the same kernel with the same arithmetic whichever shard owns it: the assembled matrix is bit-identical
 * to the single-device one.  librccl is loaded (dlopen) by the first call that needs it.
 *
 * (1) one process drives G devices: pilot_ot_multi_*  (ncclCommInitAll, one plan + one stream per device, grouped
 *     ncclAllGather).  devices[s] is the HIP device of shard s.  gather: PILOT_OT_GATHER_RCCL needs G distinct devices
 *     and leaves the full matrix on EVERY device; PILOT_OT_GATHER_COPY assembles it on the device of shard 0 with
 *     peer copies and also accepts repeated device ids (logical shards on one GPU); PILOT_OT_GATHER_AUTO picks RCCL
 *     when the devices are distinct.  All calls are asynchronous on the shards' own streams until _sync / _fetch; every
 *     shard's launches are enqueued by a host thread of its own (PILOT_OT_MULTI_SERIAL=1: by the calling thread).
 *     _sinkhorn resolves the precision once for all shards from max(M) of the inputs given to _set_inputs. */
#define PILOT_OT_GATHER_AUTO 0
#define PILOT_OT_GATHER_RCCL 1
#define PILOT_OT_GATHER_COPY 2
typedef struct pilot_ot_multi pilot_ot_multi;
int pilot_ot_multi_create(int N, int K, const int *devices, int n_shards, int gather, pilot_ot_multi **m);
int pilot_ot_multi_destroy(pilot_ot_multi *m);
int pilot_ot_multi_set_inputs(pilot_ot_multi *m, const double *P, const double *M);   /* host -> every device */
int pilot_ot_multi_sinkhorn(pilot_ot_multi *m, double reg, int num_iter_max, double stop_thr, double tau,
                            int check_period, int precision, double f32_floor_ulps, int cost_is_symmetric);
int pilot_ot_multi_emd(pilot_ot_multi *m, int cost_is_symmetric);  /* symmetric (and equal masses, see PILOT_OT_EMD_UPPER): columns >= row solved, mirrored after the gather */
int pilot_ot_multi_sync(pilot_ot_multi *m);
/* emd: N x N from the device of shard 0; iters / err / flags (nullable; exact mode: iters = n_aug) are fetched shard by
 * shard and interleaved on the host */
int pilot_ot_multi_fetch(pilot_ot_multi *m, double *emd, int *iters, double *err, int *flags);
int pilot_ot_multi_device_matrix(pilot_ot_multi *m, int shard, double **d_full);      /* the assembled matrix in HBM */
/* HIP-event times of the last call: grid_ms[s] = shard s's kernels; gather_ms = all-gather (or peer copies) + interleave: the
 * smallest per-shard (gather start -> matrix assembled) time, since a shard's collective also waits for its slower peers */
int pilot_ot_multi_times(pilot_ot_multi *m, float *grid_ms, float *gather_ms);
/* what RCCL itself reports for every shard's communicator (ncclCommCount / ncclCommUserRank): n_ranks[s], ranks[s], G entries
 * each; 0 / -1 with the peer-copy gather (no communicator).  A record that says "8 GPUs" can prove RCCL saw 8 ranks. */
int pilot_ot_multi_rccl_info(pilot_ot_multi *m, int *n_ranks, int *ranks);
/* host-buffer forms (context cached per calling thread, released by pilot_ot_shutdown) */
int pilot_ot_sinkhorn_grid_multi(const double *P, int N, int K, const double *M, double reg, int num_iter_max,
                                 double stop_thr, double tau, int check_period, int precision, double f32_floor_ulps,
                                 int cost_is_symmetric, const int *devices, int n_devices, int gather,
                                 double *emd, int *iters, double *err, int *flags);
int pilot_ot_emd_grid_multi(const double *P, int N, int K, const double *M, int cost_is_symmetric,
                            const int *devices, int n_devices, int gather, double *emd, int *n_aug);

/* (2) one process PER device (a launcher starts G of them): every rank calls the single-device *_dev entry points on
 *     its own rows (row_begin = rank, row_step = n_ranks) and assembles the matrix with pilot_ot_comm_all_gather_rows.
 *     Rank 0 creates the id and hands its 128 bytes to the others by whatever means the host language has. */
#define PILOT_OT_UNIQUE_ID_BYTES 128
typedef struct pilot_ot_comm pilot_ot_comm;
int pilot_ot_comm_unique_id(char *uid);                                   /* PILOT_OT_UNIQUE_ID_BYTES bytes out */
int pilot_ot_comm_init_rank(const char *uid, int n_ranks, int rank, pilot_ot_comm **comm);   /* on the current device */
int pilot_ot_comm_destroy(pilot_ot_comm *comm);
int pilot_ot_comm_info(pilot_ot_comm *comm, int *n_ranks, int *rank);     /* ncclCommCount / ncclCommUserRank of this communicator */
/* d_local: n_pad x N row block of this rank (n_pad = ceil(N / n_ranks), unused rows zero); d_stage: n_ranks * n_pad x N
 * scratch; d_full: N x N result on every rank.  Enqueued on `stream`. */
int pilot_ot_comm_all_gather_rows(pilot_ot_comm *comm, const double *d_local, int n_pad, int N, double *d_stage,
                                  double *d_full, void *stream);
int pilot_ot_comm_all_reduce_max(pilot_ot_comm *comm, double *d_vals, int n, void *stream);  /* in place; also the barrier */

/* ---- consumers of the finished matrix: what pilotpy does with adata.uns['EMD'] next, kept on the device ----
 * The ROWS of the N x N matrix are the data points of pl.trajectory's diffusion map (pilotpy/plot/ploting.py:95-110, after
 * EMD / EMD.max()) and of the silhouette scores (Sil_computing, pilotpy/tools/Trajectory.py:592-612; ploting.py:324, :425-431).
 * row_distances: D[i][j] = distance between rows i and j (of E / max(E) when normalize_by_max), Euclidean (scipy cdist) or
 *                cosine (sklearn cosine_distances: clipped to [0, 2], zero diagonal).
 * silhouette:    sklearn.metrics.silhouette_score(D, labels, metric="precomputed"); labels in [0, n_clusters);
 *                samples (nullable, N) receives silhouette_samples.
 * knn_kernel:    Kmat[i][j] = exp(-D[i][j]^2 / (4 epsilon)) for the k smallest entries of row i (the point itself included,
 *                like sklearn's kneighbors_graph on the fitted data), 0 elsewhere: the kernel matrix pydiffmap builds. */
#define PILOT_OT_ROWMETRIC_EUCLIDEAN 0
#define PILOT_OT_ROWMETRIC_COSINE 1
int pilot_ot_row_distances(const double *E, int N, int normalize_by_max, int metric, double *D);
int pilot_ot_row_distances_dev(const double *d_E, int N, int normalize_by_max, int metric, double *d_D,
                               double *d_max_scratch /* 8 bytes, needed when normalize_by_max */, void *stream);
int pilot_ot_silhouette(const double *D, const int *labels, int N, int n_clusters, double *score, double *samples);
int pilot_ot_knn_kernel(const double *D, int N, int k, double epsilon, double *Kmat);
/* device-resident forms: device pointers + a stream, nothing allocated, nothing synchronised.  d_sizes_scratch: n_clusters
 * ints; d_samples: N doubles (silhouette_samples; the score is their mean).  knn_kernel keeps EXACTLY k entries per row
 * (rows tied at the k-th distance in index order). */
int pilot_ot_silhouette_dev(const double *d_D, const int *d_labels, int N, int n_clusters, int *d_sizes_scratch,
                            double *d_samples, void *stream);
int pilot_ot_knn_kernel_dev(const double *d_D, int N, int k, double epsilon, double *d_Kmat, void *stream);
/* fused chains: the matrix goes to the device once (E_is_device != 0: it is there already, e.g. the pair grid's output or
 * pilot_ot_multi_device_matrix), the N x N row distances never leave it.
 * silhouette_of_rows       = Sil_computing(E [/ max(E)], labels, metric)                    (Trajectory.py:592-612, ploting.py:324)
 * diffusion_kernel_of_rows = E / max(E) -> Euclidean row distances -> k-nn Gaussian kernel   (ploting.py:95-110); D_out nullable */
int pilot_ot_silhouette_of_rows(const double *E, int E_is_device, int N, int normalize_by_max, int metric, const int *labels,
                                int n_clusters, double *score, double *samples);
int pilot_ot_diffusion_kernel_of_rows(const double *E, int E_is_device, int N, int k, double epsilon, double *D_out, double *Kmat);

/* ---- cell-level W2 pair grid (EXTENSION: not in the reference; BASELINE config 5) ------ */
/* Compares patients by their raw cell clouds instead of cell-type proportions.  X: n_cells x D float32 embedding
 * with the cells of patient i stored contiguously at rows offsets[i] .. offsets[i+1] (offsets: N + 1 entries).
 * Pair (i, j): uniform weights 1/n_i, 1/n_j, cost C = |x - y|^2 / scale, entropic OT solved in the log domain with
 * the control flow of POT's ot.bregman.sinkhorn_log (v-update then u-update; marginal error every check_period
 * updates; stop on err < stop_thr, floored in f32 like the proportion-level kernel, or after num_iter_max updates);
 * the value is <Gamma, C>.  The n_i x n_j cost matrix is never materialised.  D <= 64; patients up to ~13 000 cells.
 * w2 / iters / err: n_rows x N (iters, err nullable). */
int pilot_ot_cell_w2_grid(const float *X, const long long *offsets, int N, int D, double scale, double reg,
                          int num_iter_max, double stop_thr, int check_period, double f32_floor_ulps,
                          int row_begin, int row_end, int row_step, double *w2, int *iters, double *err);
/* Device-resident form: a cohort keeps the cells (as bf16 operand pieces) in HBM across calls, so a call moves only the
 * result rows.  kernel_ms (nullable): HIP-event time of the pair-grid kernel of this call. */
typedef struct pilot_ot_cell_cohort pilot_ot_cell_cohort;
int pilot_ot_cell_cohort_create(const float *X, const long long *offsets, int N, int D, pilot_ot_cell_cohort **cohort);
int pilot_ot_cell_cohort_destroy(pilot_ot_cell_cohort *cohort);
int pilot_ot_cell_cohort_pieces(pilot_ot_cell_cohort *cohort, int *pieces);   /* operand pieces per coordinate of the last call: 2 (fp16) | 3 (bf16) | 0 */
int pilot_ot_cell_w2_grid_cohort(pilot_ot_cell_cohort *cohort, double scale, double reg, int num_iter_max, double stop_thr,
                                 int check_period, double f32_floor_ulps, int row_begin, int row_end, int row_step,
                                 double *w2, int *iters, double *err, float *kernel_ms);
/* Full N x N grid, rows dealt round-robin over the listed devices (every device holds the cohort; the shards run
 * concurrently; the result rows are assembled on the host -- there is no device-side exchange step to make). */
int pilot_ot_cell_w2_grid_multi(const float *X, const long long *offsets, int N, int D, double scale, double reg,
                                int num_iter_max, double stop_thr, int check_period, double f32_floor_ulps,
                                const int *devices, int n_devices, double *w2, int *iters, double *err);

/* precision selected by PILOT_OT_PREC_AUTO for a given max(M)/reg (PILOT_OT_PREC_F16X2, PILOT_OT_PREC_BF16X3 or
 * PILOT_OT_PREC_F64; a shape whose split operand image does not fit LDS runs PILOT_OT_PREC_F32 instead) */
int pilot_ot_auto_precision(double max_cost_over_reg);
int pilot_ot_auto_precision_for(double max_cost_over_reg, int K, int cost_is_symmetric);   /* with the LDS fit of this K */
/* the precision a call with these arguments RUNS (every Sinkhorn entry point goes through it): GENERIC beyond
 * PILOT_OT_MAX_COST_OVER_REG; AUTO by range; an explicit f32-class precision (F32, BF16X3, F16X2) beyond the f32 range
 * (max(M)/reg > 60) runs AUTO_MIXED -- explicit precisions are honoured inside their valid range only; F16X2 outside its
 * scaled domain (max(M)/reg > 16 or tau > 2000) runs BF16X3. */
int pilot_ot_resolve_precision(int precision, double max_cost_over_reg, int K, int cost_is_symmetric, double tau);

#ifdef __cplusplus
}
#endif
#endif /* PILOT_OT_H */
