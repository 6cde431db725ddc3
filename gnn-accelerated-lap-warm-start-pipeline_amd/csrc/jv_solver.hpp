// jv_solver.hpp -- host-visible interface of the per-instance solver kernel (jv_solver.hip)
// and of the dense sweep kernels (dense_sweeps.hip).  Internal to the shared library; the
// public C ABI is include/lapwarm_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace lapwarm {

constexpr int kModeSeeded = 0;  // continue from the dense prelude (lapjv_seeded)
constexpr int kModeCold = 1;    // plain lapjv

constexpr int kBranchSsp = 1;
constexpr int kBranchAllMatched = 2;
constexpr int kBranchFallback = 3;
constexpr int kBranchCold = 4;

constexpr int kStatsPerInstance = 32;  // 0..15 counters, 16..31 cycle stamps (diagnostic builds)
constexpr size_t kLdsBudgetBytes = 160 * 1024;

struct SolverParams {
    const double *C;  // [batch][n][n] row-major fp64
    int n;
    int batch;
    int mode;
    // seeded mode inputs, produced by the prelude kernel
    const double *u_tight;       // [batch][n]      u after row tightening (P3)
    const double *v_work;        // [batch][n]      v after the (optional) projection
    const int *tight_cnt;        // [batch][n]      tight edges per row
    const uint32_t *tight_bits;  // [batch][n][W]   tight-edge bitmap per row, W = ceil(n/32)
    const int *inst_flags;       // [batch]
    double tight_eps;
    // outputs
    long long *x_out, *y_out;  // [batch][n] int64 (seeded API) or null
    int *x32_out, *y32_out;    // [batch][n] int32 (lapjv API) or null
    double *v_out;             // [batch][n] final column duals or null
    double *u_out;             // [batch][n] u_i = C[i][x_i] - v[x_i] or null
    int *ret;                  // [batch]
    long long *stats;          // [batch][kStatsPerInstance] or null
    // per-instance state in global memory, only used when the state does not fit LDS
    double *g_dist, *g_v;
    int *g_order, *g_pred, *g_y, *g_x, *g_fr, *g_evl, *g_tmpcol;
    // helper workgroups (EXPERIMENT, LAPWARM_HELPER=1): [batch][kRingInts] ring of upcoming head
    // rows, zeroed by the caller; word 0 = done flag, words 2.. = (generation << 16 | row)
    int *pf_ring;
    int helper;
    // cooperative shortest-path phase (coop_ssp.hip): phase 0 = the whole solve in this kernel;
    // 1 = stop before the shortest-path phase and leave x, y, v, the free-row list in the global
    // state arrays + hand[]; 2 = take x, y, v back, run the paths the cooperative kernel left
    // (hand[1] .. hand[0]) and write the outputs
    int phase;
    int *hand;                 // [batch][kHandInts]
    long long *cstats;         // [batch][kCoopStats] path counters of the cooperative kernel
    unsigned long long *mail;  // [batch][coop_mail_granules(n)] zeroed by phase 1
    int mail_granules;
    // candidate lists of the augmenting row reduction (cold solves, lapwarm_lapjv_*_batched):
    // [batch][n][kArrListEntries] raw costs / columns, [batch][n] thresholds; null = plain row scans
    // (last: a member in the middle moved the kernel arguments behind it and cost the seeded kernel 12 SGPR spills)
    double *arr_lval, *arr_ltau;
    int *arr_lcol;
};
constexpr int kArrListEntries = 128;
constexpr int kRingSlots = 64;
constexpr int kRingInts = 2 + kRingSlots;

// hand[]: 0 free rows, 1 paths done by the cooperative kernel (resume index), 2 its error code,
// 3 why it stopped early, 4 error seen by a member other than the leader, 5.. what phase 1 knows
// and phase 2 reports (14: row-reduction iterations answered from candidate lists, 15: duration of the phase-1 launch) (branch, tight edges, free rows after greedy, micro-ARR firings, transfer
// rows, ARR iterations, column-reduction elements lo/hi, phase-1 error)
constexpr int kHandInts = 16;
constexpr int kCoopStats = 16;

struct CoopParams {
    const double *C;
    int n, batch;
    int G;             // members (single-wave workgroups) per instance, filled in by launch_coop
    int first, count;  // instances [first, first + count) of this launch
    int xcd_stores;    // allow workgroup-scope mailbox stores when all members of an instance share an XCD
    double *v;         // [batch][n] column duals (the solver's global state arrays)
    int *x, *y, *pred;
    const int *fr;     // [batch][n] free rows, hand[0] of them
    int *hand;
    long long *cstats;
    unsigned long long *mail;
};
bool coop_enabled(int n);
int coop_members(int n);
size_t coop_mail_granules(int n);
hipError_t launch_coop(const CoopParams &p, hipStream_t stream);

size_t solver_lds_bytes(int n, int ch, int level);
int solver_lds_level(int n, int ch);
bool solver_needs_global_state(int n);
// candidate lists for the augmenting row reduction: from the size where a row is a few times its list
// (LAPWARM_ARR_LISTS=0 turns them off: every iteration then scans its whole row)
bool arr_lists_enabled(int n);
void solver_geometry(int n, int threads_hint, int *threads, int *ch);
hipError_t launch_solver(const SolverParams &p, int threads_hint, hipStream_t stream);
bool solver_uses_helpers(int n);  // seeded mode with a ring: one helper workgroup per instance

// ---- dense sweeps (dense_sweeps.hip) ----------------------------------------------------
struct PreludeParams {
    const double *C;
    int n, batch;
    const double *u;  // [batch][n] duals the verify step uses (seed, or projected)
    const double *v;  // [batch][n]
    double eps, tight_eps;
    int rerun;        // 0: first pass; 1: only instances whose duals were projected
    double *u_tight;
    int *viol_cnt;    // [batch][n] candidates of the projection per row (first pass only)
    int *tight_cnt;
    uint32_t *tight_bits;
    int *inst_flags;
};
hipError_t launch_prelude(const PreludeParams &p, hipStream_t stream);
hipError_t launch_seed_prepare(const double *u_seed, const double *v_seed, double *u_work, double *v_work,
                               size_t count, int *flags, int n_flags, int *ring, int n_ring, hipStream_t stream);

// Gauss-Seidel projection of (u, v) for the instances flagged kFlagHasViolation; in place.
hipError_t launch_projection(const double *C, int n, int batch, double *u, double *v,
                             const int *viol_cnt, int *inst_flags, double eps, hipStream_t stream);

// out[b][j] = min_i (C[b][i][j] - (u ? u[b][i] : 0)); `partial` holds batch*chunks*n doubles.
int colmin_chunks(int n, int batch);
hipError_t launch_colmin(const double *C, int n, int batch, const double *u, double *out,
                         double *partial, hipStream_t stream);

// out[b][i] = min_j (C[b][i][j] - (v ? v[b][j] : 0))
hipError_t launch_rowmin(const double *C, int n, int batch, const double *v, double *out,
                         hipStream_t stream);

// R[b][i][j] = (C - u_i) - v_j - shift[b] ; gmin[b] = min_ij ((C - u_i) - v_j)
hipError_t launch_reduced_min(const double *C, int n, int batch, const double *u, const double *v,
                              double *gmin_partial, double *gmin, hipStream_t stream);
hipError_t launch_reduce_costs(const double *C, int n, int batch, const double *u, const double *v,
                               const double *gmin, int shift_nonneg, double *out, hipStream_t stream);
// one round of project_feasible's u/v caps: u = min(u, rowmin(C - v)) ; v = min(v, colmin(C - u))
hipError_t launch_cap_rows(const double *C, int n, int batch, double *u, const double *v,
                           hipStream_t stream);
hipError_t launch_cap_cols(const double *C, int n, int batch, const double *u, double *v,
                           double *partial, hipStream_t stream);

// 13 row statistics + 8 positional encodings (float32) and the 16 smallest costs per row.
struct FeatureParams {
    const double *C;
    int n, batch;
    const double *colmin;  // [batch][n]
    const float *posenc;   // [n][8] host-computed table
    float *feat;           // [batch][n][21]
    float *topk;           // [batch][n][16] ascending, +inf padded, or null
};
hipError_t launch_row_features(const FeatureParams &p, hipStream_t stream);

// OneGNN refinement aggregation (onegnn_refine.hip)
hipError_t launch_refine_aggregate(const float *topk16, const float *u_pre, const float *w1,
                                   const float *b1, float *out, float *wsum, int rows, int H,
                                   hipStream_t stream);

}  // namespace lapwarm
