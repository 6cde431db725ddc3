/*
 * lapwarm_hip.h -- C ABI of liblapwarm_hip.so, the MI355X (gfx950) implementation of the
 * warm-started LAP hot path.  Plain pointers and sizes only; no torch / C++ types.
 *
 * Two families of entry points:
 *   (1) drop-in replacements for the reference's native functions: HOST pointers, same
 *       argument meaning and return codes, one instance per call;
 *   (2) the batched DEVICE-pointer API that the pipeline and bench use: inputs already in
 *       HBM, stream-ordered, no host synchronisation inside (graph-capturable).
 *
 * Citations are relative to the reference repository root.
 */
#ifndef LAPWARM_HIP_H
#define LAPWARM_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * (1) Drop-in, host pointers
 * ---------------------------------------------------------------------------------------- */

/* Replaces `lapjv_seeded` of LAP/lap/lapjv_seeded.h:8-13 (defined in
 * LAP/_lapjv_cpp/lapjv_seeded.cpp:19-173).  Identical signature and return codes:
 * 0 ok, -1 allocation failure, -2 n <= 0, -4 non-square, -3 infeasible after projection.
 * C, u_seed, v_seed are borrowed and never modified; x, y (caller-allocated, n each) are
 * written only when 0 is returned.  Other negative values: -5 n too large for this build
 * (> 16384), <= -100 internal guard tripped, <= -1000 HIP runtime error (-1000 - hipError_t). */
int lapjv_seeded(const double *C, int n_rows, int n_cols, long long *x, long long *y,
                 const double *u_seed, const double *v_seed, double eps);

/* Replaces `lapjv_internal(n, cost[], x, y)` of LAP/_lapjv_cpp/lapjv.cpp:323-346 (declared
 * LAP/_lapjv_cpp/lapjv.h:60-62) for a contiguous row-major matrix (the reference builds the
 * row-pointer array from exactly such a matrix, LAP/_lapjv_cpp/_lapjv.pyx:97-101).
 * Returns 0 or the codes above. */
int lapwarm_lapjv_dense(const double *C, int n, int *x, int *y);

/* gnn/features.py:161-243 `compute_row_features`: C (n*n fp64) -> feat (n*21 float32).
 * `topk16` (n*16 float32, ascending, +inf padded) may be NULL. */
int lapwarm_row_features(const double *C, int n, float *feat, float *topk16);

/* scripts/gnn_benchmark.py:262  v_j = min_i (C_ij - u_i), fp64 (u NULL -> plain column minima,
 * gnn/features.py:218). */
int lapwarm_min_trick(const double *C, int n, const double *u, double *v);

/* Row minima  out_i = min_j (C_ij - v_j)  (v NULL -> plain row minima): the `row_min` /
 * `u_cap` sweeps of solvers/seed_baselines.py:29 and solvers/advanced_dual.py:29. */
int lapwarm_row_min(const double *C, int n, const double *v, double *out);

/* solvers/advanced_dual.py:14-36 `project_feasible`: u, v updated in place. */
int lapwarm_project_feasible(const double *C, int n, double *u, double *v, int max_rounds, double tol);

/* solvers/advanced_dual.py:39-53 `reduce_costs`: out (n*n) = C - u 1^T - 1 v^T, shifted to be
 * non-negative when asked.  `min_out` (may be NULL) receives the unshifted minimum, which is what
 * `check_dual_feasible` (advanced_dual.py:56-63) tests. */
int lapwarm_reduce_costs(const double *C, int n, const double *u, const double *v, int shift_nonneg,
                         double *out, double *min_out);

/* WarmStartLAPSolver.solve (solvers/warmstart_solver.py:31-63): C' = C - u 1^T - 1 v^T
 * (minus min(C') when negative and shift_nonneg), then the cold JV on C'.  One host-to-device
 * copy of C; the reduced matrix is formed and solved on the device; x, y [n] int32 come back. */
int lapwarm_warmstart_lapjv(const double *C, int n, const double *u, const double *v, int shift_nonneg,
                            int *x, int *y);

/* ------------------------------------------------------------------------------------------
 * (2) Batched, device pointers, stream-ordered.  `stream` is a hipStream_t (NULL = default).
 *     Every function returns 0 or <= -1000 (HIP error); per-instance codes go to `ret`.
 * ---------------------------------------------------------------------------------------- */

#define LAPWARM_STATS_PER_INSTANCE 32
/* stats[b][...]: 0 branch (1 ssp, 2 all matched, 3 fallback, 4 cold), 1 tight edges,
 * 2 free rows, 3 micro-ARR firings, 4 paths, 5 minima collections, 6 relax steps,
 * 7 relax elements (sum of n-hi), 8 path-init elements, 9 column-reduction elements,
 * 10 reduction-transfer rows, 11 ARR iterations, 12 internal error bits, 13 kernel time and
 * 14 greedy+micro-ARR time (10 ns ticks); 15 paths completed by the cooperative kernel | stop reason << 32
 * (-1: that kernel was not part of the solve); 16..26 its exchange-round counters; 27 row-reduction
 * iterations answered from candidate lists (cold solves); 16..31 cycle stamps in -DLAPWARM_STAMPS builds.
 * Where a solve consists of several launches (n >= 4428, cold solves with lists) slot 13 adds up the
 * preparation launch and the final one. */

size_t lapwarm_seeded_workspace_bytes(int batch, int n);
/* Workspace of the cold entry points below: the seeded workspace plus, from n = 512, the candidate lists
 * of the augmenting row reduction (1,544 bytes per row) and the hand-over state of the two launches a cold
 * solve then consists of (preparation with the lists, shortest augmenting paths; 44 bytes per row).  A
 * workspace of only lapwarm_seeded_workspace_bytes() is accepted too: one launch, every row-reduction
 * iteration scans its row (LAP/_lapjv_cpp/lapjv.cpp:76-149 as written). */
size_t lapwarm_lapjv_workspace_bytes(int batch, int n);

/* Batched lapjv_seeded over C[batch][n][n]; u_seed, v_seed [batch][n]; x, y [batch][n] int64;
 * ret [batch] int; stats [batch][32] int64 or NULL.  `threads_hint` = workgroup size of the
 * per-instance kernel (0 = auto). */
int lapwarm_seeded_batched(const double *C, int batch, int n, const double *u_seed,
                           const double *v_seed, double eps, long long *x, long long *y, int *ret,
                           long long *stats, void *workspace, size_t workspace_bytes,
                           int threads_hint, void *stream);

/* Batched cold lapjv; x, y [batch][n] int32.  Workspace: lapwarm_lapjv_workspace_bytes(). */
int lapwarm_lapjv_batched(const double *C, int batch, int n, int *x, int *y, int *ret,
                          long long *stats, void *workspace, size_t workspace_bytes,
                          int threads_hint, void *stream);

/* As lapwarm_lapjv_batched, and also returns the optimal dual pair of each instance:
 * v [batch][n] = the solver's final column duals, u [batch][n] with u_i = C[i][x_i] - v[x_i]
 * (complementary slackness on the matched edges).  This is how the K2 configuration gets its
 * "oracle u" without the reference's O(n^3) Bellman-Ford (solvers/dual_computation.py:34-52). */
int lapwarm_lapjv_duals_batched(const double *C, int batch, int n, int *x, int *y, double *u, double *v,
                                int *ret, long long *stats, void *workspace, size_t workspace_bytes,
                                int threads_hint, void *stream);

size_t lapwarm_sweep_workspace_bytes(int batch, int n);

/* out[b][j] = min_i (C[b][i][j] - u[b][i]); u may be NULL. */
int lapwarm_colmin_batched(const double *C, int batch, int n, const double *u, double *out,
                           void *workspace, size_t workspace_bytes, void *stream);

/* out[b][i] = min_j (C[b][i][j] - v[b][j]); v may be NULL. */
int lapwarm_rowmin_batched(const double *C, int batch, int n, const double *v, double *out, void *stream);

/* feat [batch][n][21] float32, topk16 [batch][n][16] float32 or NULL; posenc [n][8] float32 is
 * the table of gnn/features.py:21-31 (built once per n on the host). */
int lapwarm_row_features_batched(const double *C, int batch, int n, const float *posenc, float *feat,
                                 float *topk16, void *workspace, size_t workspace_bytes, void *stream);

/* One round of project_feasible: u = min(u, rowmin(C-v)); v = min(v, colmin(C-u));
 * gmin[b] = min((C-u)-v).  The host loop decides when to stop. */
int lapwarm_project_round_batched(const double *C, int batch, int n, double *u, double *v,
                                  double *gmin, void *workspace, size_t workspace_bytes, void *stream);

/* out [batch][n][n]; gmin [batch] receives the unshifted minima. */
int lapwarm_reduce_costs_batched(const double *C, int batch, int n, const double *u, const double *v,
                                 int shift_nonneg, double *out, double *gmin, void *workspace,
                                 size_t workspace_bytes, void *stream);

/* OneGNN top-k refinement, aggregation part (gnn/one_gnn.py:139-155), float32:
 *   val_k = topk16[row][k] - u_pre[row]   (== topk(cost - u_pre): x -> x - c is monotone)
 *   w     = softmax(-val) over the finite entries (0 elsewhere)
 *   out[row][h] = sum_k w_k * GELU(w1[h] * val_k + b1[h]),  wsum[row] = sum_k w_k
 * The caller applies the second edge-MLP layer once per row: out @ W2^T + b2 * wsum (linearity).
 * topk16 [rows][16], u_pre [rows], w1/b1 [H], out [rows][H], wsum [rows]. */
int lapwarm_refine_aggregate_batched(const float *topk16, const float *u_pre, const float *w1,
                                     const float *b1, float *out, int rows, int H, int reserved,
                                     void *stream);
/* same call, with the [rows] weight sums; `wsum` may be NULL */
int lapwarm_refine_aggregate_wsum(const float *topk16, const float *u_pre, const float *w1,
                                  const float *b1, float *out, float *wsum, int rows, int H, void *stream);

/* Profiling hook for bench.py: when enabled, lapwarm_seeded_batched / lapwarm_lapjv_batched
 * bracket the per-instance solver kernel with HIP events on the caller's stream;
 * lapwarm_profile_last_solver_ms() waits for the last bracket and returns its duration. */
void lapwarm_profile_enable(int on);
double lapwarm_profile_last_solver_ms(void);

/* 1 when lapwarm_seeded_batched launches one helper workgroup per instance for this n (the helper
 * pulls announced head rows towards the L2 its solver shares; LAPWARM_HELPER=0 turns it off):
 * the solver kernel then occupies about 2 * batch CUs.  n = 1024 .. 8192. */
int lapwarm_solver_uses_helpers(int n);

/* Number of single-wave workgroups ("members", one compute unit each) that share ONE instance's
 * shortest-augmenting-path phase for this n, or 0 when that phase runs inside the one-workgroup-per-
 * instance kernel (n < 4428 by default; LAPWARM_COOP=0 / LAPWARM_COOP_MIN_N change it).  New, additive:
 * the reference has no counterpart (its _ca_dense, LAP/_lapjv_cpp/lapjv.cpp:286-319, is serial). */
int lapwarm_coop_members(int n);

/* Misc */
const char *lapwarm_last_error(void);
int lapwarm_device_count(void);
const char *lapwarm_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* LAPWARM_HIP_H */
