/* jv_oracle.h -- TEST INFRASTRUCTURE (CPU checker), see jv_oracle.c. */
#ifndef JV_ORACLE_H
#define JV_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum {
    JVO_BRANCH_NONE = 0,
    JVO_BRANCH_SSP = 1,         /* seeded: greedy + shortest paths for the free rows */
    JVO_BRANCH_ALL_MATCHED = 2, /* seeded: greedy matched every row */
    JVO_BRANCH_FALLBACK = 3,    /* seeded: < 1.2 n tight edges -> cold JV */
    JVO_BRANCH_COLD = 4         /* plain lapjv */
};

/* Element-visit counters of the serial phase.
 * All fields are long long so that ctypes can map the struct trivially. */
typedef struct jvo_stats {
    long long branch;
    long long proj_events;   /* P1 adjustments applied */
    long long tight_edges;   /* P6 count */
    long long free_rows;     /* rows left free by the greedy phase (seeded) or by ARR (cold) */
    long long arr_fired;     /* P8 micro-ARR dual raises */
    long long paths;         /* shortest-path searches */
    long long finds;         /* minima collections */
    long long scan_steps;    /* outer iterations of the relax loop = dependent row reads */
    long long scan_elems;    /* sum of (n - hi) over scan steps */
    long long init_elems;    /* n per path (distance initialisation) */
    long long colred_elems;  /* n*n for the cold column reduction */
    long long transfer_rows; /* rows visited by the reduction transfer */
    long long arr_iters;     /* iterations of the cold augmenting row reduction */
} jvo_stats;

/* Same ABI as the reference's lapjv_seeded (LAP/lap/lapjv_seeded.h:8-13). */
int jvo_lapjv_seeded(const double *C, int n_rows, int n_cols, long long *x, long long *y,
                     const double *u_seed, const double *v_seed, double eps);

/* As above, plus counters and the dual vectors at exit (either may be NULL). */
int jvo_lapjv_seeded_ex(const double *C, int n_rows, int n_cols, long long *x, long long *y,
                        const double *u_seed, const double *v_seed, double eps, jvo_stats *st,
                        double *u_final, double *v_final);

/* Cold dense JV on a row-major n*n matrix (reference: lapjv_internal,
 * LAP/_lapjv_cpp/lapjv.cpp:323-346, which takes row pointers). */
int jvo_lapjv_dense(const double *C, int n, int *x, int *y, jvo_stats *st);

#ifdef __cplusplus
}
#endif
#endif
