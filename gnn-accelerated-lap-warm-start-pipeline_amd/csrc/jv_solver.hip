// jv_solver.hip -- one LAP instance per workgroup: the serial phases of the seeded
// Jonker-Volgenant solve (and the cold JV it falls back to), written for gfx950.
//
// Reference behaviour reproduced bit for bit (paths relative to /root/reference):
//   greedy tight-edge matching     LAP/_lapjv_cpp/lapjv_seeded.cpp:76-102
//   quality gate + fallback        LAP/_lapjv_cpp/lapjv_seeded.cpp:105-125
//   micro-ARR on free rows         LAP/_lapjv_cpp/lapjv_seeded.cpp:136-159
//   shortest augmenting paths      LAP/_lapjv_cpp/lapjv.cpp:153-319
//   cold JV (column reduction, reduction transfer, 2 ARR sweeps)  lapjv.cpp:8-149,323-346
//
// Design (not a translation of the serial code):
//   * all O(n) solver state (dist, v, column order, pred, x, y, free list) lives in LDS
//     (36 n bytes: 72 KiB at n=2048, 144 KiB at n=4096 of the CU's 160 KiB); larger n use a
//     per-instance global workspace that stays L2 resident;
//   * every thread owns CH consecutive POSITIONS of the column order.  One relax step is one
//     pass over the TODO positions: gather C[i][col], v[col], dist[col], update, and keep the
//     new distances in registers;
//   * the serial code's order-dependent swaps ("events") are found in parallel -- an
//     exclusive prefix-min over positions for the minima collection, an equality test for the
//     relax loop -- published as position bitmaps in LDS and replayed in position order by
//     wave 0 only.  Positions above the one being examined are never touched by the serial
//     loops, so event detection on the pre-loop order is exact; random data has ~ln n events.
//   * the minima collection that follows an event-free relax step reuses the distances that
//     are still in registers (no second LDS sweep);
//   * the relax loop is batched: up to SMAX queued SCAN columns (they all sit at the same level)
//     are relaxed in ONE pass -- every TODO position loads its entry of all their rows, then
//     replays the serial sequence in registers.  A column's evolution inside the batch depends
//     on no other column; only the ORDER of the tie events does, and that is rebuilt by wave 0
//     step by step from per-step column bitmaps and the inverse permutation pos[].  Writes made
//     by the pass after an early return are unobservable (dist/order die with the path, pred is
//     only read along READY/SCAN columns), so the pass never has to be rolled back.
#include "device_utils.hpp"
#include "jv_solver.hpp"

namespace lapwarm {

namespace {

struct Ctrl {
    double level;
    long long batch_elems;
    int hi;
    int target;
    int evt_step;
    int first_fire;
    int nfree;
    int err;
    unsigned ev_mask;
    int batch_steps;
};

template <int CH>
struct BatchDepth {
    static constexpr int value = (CH >= 16) ? 1 : (CH == 8 ? 2 : (CH == 4 ? 4 : 8));
};

constexpr int kSentinelIdx = 0x7ffffffe;  // the LARGE sentinel of the ARR scan (index -1 in the reference)
constexpr int kEmptyIdx = 0x7fffffff;

template <int CH, int LDSL>
struct Solver {
    static constexpr bool LDS_STATE = LDSL > 0;
    static constexpr int SMAX = BatchDepth<CH>::value;
    // problem
    const double *C;
    int n, W;
    // state
    double *dist, *v;
    int *order, *pred, *y, *x, *fr, *pos;
    uint32_t *evt, *sbits, *used, *evb, *tmpb;
    int Wpad;
    Ctrl *ctrl;
    BlockCtx bc;
    // uniform counters (identical in every thread)
    long long scan_elems, init_elems, colred_elems;
    int paths, finds, scan_steps, arr_iters, transfer_rows, arr_fired;
    int step_id;
    int err;

    __device__ __forceinline__ int base() const { return bc.tid * CH; }

    __device__ __forceinline__ void fence_if_global()
    {
        if constexpr (!LDS_STATE) __threadfence_block();
    }

    // ------------------------------------------------------------------ event replay (wave 0)
    // Minima collection, lapjv.cpp:153-171, given the event / strict bitmaps.
    __device__ __forceinline__ void replay_find(int lo)
    {
        const int lane = bc.lane;
        int hi = lo + 1;
        for (int wbase = 0; wbase < W; wbase += kWave) {
            const int idx = wbase + lane;
            uint32_t ew = 0, sw = 0;
            if (idx < W) {
                ew = evt[idx];
                sw = sbits[idx];
                if (ew) {
                    evt[idx] = 0;
                    sbits[idx] = 0;
                }
            }
            unsigned long long mask = __ballot(ew != 0);
            while (mask) {
                const int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                uint32_t e = __shfl(ew, l, kWave);
                const uint32_t s = __shfl(sw, l, kWave);
                while (e) {
                    const int bit = __builtin_ctz(e);
                    e &= e - 1;
                    const int k = ((wbase + l) << 5) + bit;
                    const int j = order[k];
                    if ((s >> bit) & 1u) hi = lo;
                    const int a = order[hi];
                    if (lane == 0) {
                        order[k] = a;
                        pos[a] = k;
                        order[hi] = j;
                        pos[j] = hi;
                    }
                    fence_if_global();
                    ++hi;
                }
            }
        }
        // last free column of the SCAN list wins (lapjv.cpp:250-255)
        int best = -1;
        for (int kk = lo + lane; kk < hi; kk += kWave) {
            if (y[order[kk]] < 0) best = kk;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const int o = __shfl_xor(best, m, kWave);
            best = (o > best) ? o : best;
        }
        const int target = (best >= 0) ? order[best] : -1;
        const double level = dist[order[lo]];
        if (lane == 0) {
            ctrl->hi = hi;
            ctrl->target = target;
            ctrl->level = level;
        }
    }

    // Tie events of one batch of relax steps (lapjv.cpp:199-205), replayed step by step.
    // evb[s] holds, per COLUMN, the events the pass found at step s; their order inside a step
    // is the order of the columns' positions at that step's start, read from pos[].
    __device__ __forceinline__ void replay_batch(int hi, int S)
    {
        const int lane = bc.lane;
        const unsigned mask = ctrl->ev_mask;
        int target = -1;
        long long elems = 0;
        int steps = 0;
        for (int st = 0; st < S && target < 0; ++st) {
            elems += (long long)(n - hi);
            ++steps;
            if (!((mask >> st) & 1u)) continue;
            uint32_t *eb = evb + (size_t)st * Wpad;
            for (int wbase = 0; wbase < W; wbase += kWave) {
                const int idx = wbase + lane;
                uint32_t word = 0;
                if (idx < W) {
                    word = eb[idx];
                    if (word) eb[idx] = 0;
                }
                while (word) {
                    const int bit = __builtin_ctz(word);
                    word &= word - 1;
                    const int p = pos[(idx << 5) + bit];
                    atomicOr(&tmpb[p >> 5], 1u << (p & 31));
                }
            }
            fence_if_global();
            for (int wbase = 0; wbase < W; wbase += kWave) {
                const int idx = wbase + lane;
                uint32_t tw = 0;
                if (idx < W) {
                    tw = tmpb[idx];
                    if (tw) tmpb[idx] = 0;
                }
                unsigned long long m = __ballot(tw != 0);
                while (m && target < 0) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    uint32_t e = __shfl(tw, l, kWave);
                    while (e && target < 0) {
                        const int bit = __builtin_ctz(e);
                        e &= e - 1;
                        const int p = ((wbase + l) << 5) + bit;
                        const int j = order[p];
                        if (y[j] < 0) {
                            target = j;
                        } else {
                            const int a = order[hi];
                            if (lane == 0) {
                                order[p] = a;
                                pos[a] = p;
                                order[hi] = j;
                                pos[j] = hi;
                            }
                            fence_if_global();
                            ++hi;
                        }
                    }
                }
            }
        }
        if (target >= 0) {
            // early return: drop whatever is still queued in the bitmaps
            for (int w = lane; w < Wpad; w += kWave) tmpb[w] = 0;
            for (int w = lane; w < SMAX * Wpad; w += kWave) evb[w] = 0;
        }
        if (lane == 0) {
            ctrl->hi = hi;
            ctrl->target = target;
            ctrl->ev_mask = 0;
            ctrl->batch_elems = elems;
            ctrl->batch_steps = steps;
        }
    }

    // ------------------------------------------------------------------ one shortest path
    // lapjv.cpp:221-282.  Returns the free column reached; updates v for the READY columns.
    __device__ __forceinline__ int find_path(int start)
    {
        const int b0 = base();
        const int wordi = b0 >> 5, shift = b0 & 31;
        double dk[CH];
        {
            const double *row = C + (size_t)start * n;
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const int k = b0 + r;
                if (k < n) {
                    const double val = row[k] - v[k];
                    order[k] = k;
                    pos[k] = k;
                    pred[k] = start;
                    dist[k] = val;
                    dk[r] = val;
                } else {
                    dk[r] = pos_inf();
                }
            }
        }
        paths++;
        init_elems += n;
        int lo = 0, hi = 0, ready = 0, target = -1;
        double level = 0.0;
        int guard = 0;
        while (target < 0) {
            if (++guard > 2 * n + 4) {
                err = 1;
                break;
            }
            if (lo == hi) {
                // ---------------- minima collection over positions [lo, n); dk[] is current
                ready = lo;
                double tmin = pos_inf();
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    const int k = b0 + r;
                    if (k >= lo && k < n) tmin = dmin(tmin, dk[r]);
                }
                double wtot;
                double run = wave_excl_prefix_min(tmin, bc.lane, &wtot);
                {
                    const int p = bc.parity;
                    bc.parity ^= 1;
                    if (bc.lane == 0) bc.ex->d[p][bc.wave] = wtot;
                    __syncthreads();
                    for (int w = 0; w < bc.wave; ++w) run = dmin(run, bc.ex->d[p][w]);
                }
                uint32_t eb = 0, sb = 0;
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    const int k = b0 + r;
                    if (k >= lo && k < n) {
                        if (k > lo && dk[r] <= run) {
                            eb |= 1u << r;
                            if (dk[r] < run) sb |= 1u << r;
                        }
                        run = dmin(run, dk[r]);
                    }
                }
                if (eb) {
                    atomicOr(&evt[wordi], eb << shift);
                    if (sb) atomicOr(&sbits[wordi], sb << shift);
                }
                __syncthreads();
                if (bc.wave == 0) replay_find(lo);
                __syncthreads();
                hi = ctrl->hi;
                target = ctrl->target;
                level = ctrl->level;
                finds++;
                if (target >= 0) break;
            }
            // ---------------- relax a batch of queued SCAN columns (lapjv.cpp:178-213)
            const int S = (hi - lo < SMAX) ? hi - lo : SMAX;
            const double *rows[SMAX];
            double hs[SMAX];
            int is[SMAX];
#pragma unroll
            for (int q = 0; q < SMAX; ++q) {
                rows[q] = C;
                hs[q] = 0.0;
                is[q] = 0;
                if (q < S) {
                    const int jc = order[lo + q];
                    const int i = y[jc];
                    is[q] = i;
                    rows[q] = C + (size_t)i * n;
                    hs[q] = (rows[q][jc] - v[jc]) - level;
                }
            }
            unsigned my_mask = 0;
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const int k = b0 + r;
                if (k >= hi && k < n) {
                    const int j = order[k];
                    double c[SMAX];
#pragma unroll
                    for (int q = 0; q < SMAX; ++q) c[q] = (q < S) ? rows[q][j] : 0.0;
                    const double vj = v[j];
                    double dj = dist[j];
                    int pj = -1, ev = -1;
#pragma unroll
                    for (int q = 0; q < SMAX; ++q) {
                        if (q < S && ev < 0) {
                            const double cand = (c[q] - vj) - hs[q];
                            if (cand < dj) {
                                dj = cand;
                                pj = is[q];
                                if (cand == level) ev = q;
                            }
                        }
                    }
                    if (pj >= 0) {
                        dist[j] = dj;
                        pred[j] = pj;
                    }
                    dk[r] = dj;
                    if (ev >= 0) {
                        atomicOr(&evb[(size_t)ev * Wpad + (j >> 5)], 1u << (j & 31));
                        my_mask |= 1u << ev;
                    }
                } else {
                    dk[r] = pos_inf();
                }
            }
            if (my_mask) {
                atomicOr(&ctrl->ev_mask, my_mask);
                ctrl->evt_step = step_id;
            }
            __syncthreads();
            if (ctrl->evt_step == step_id) {
                if (bc.wave == 0) replay_batch(hi, S);
                __syncthreads();
                hi = ctrl->hi;
                target = ctrl->target;
                scan_steps += ctrl->batch_steps;
                scan_elems += ctrl->batch_elems;
            } else {
                scan_steps += S;
                scan_elems += (long long)S * (n - hi);
            }
            step_id++;
            lo += S;
        }
        // dual update for the READY columns (lapjv.cpp:270-276): v[j] += d[j] - level
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int k = b0 + r;
            if (k < ready) {
                const int j = order[k];
                v[j] += dist[j] - level;
            }
        }
        return target;
    }

    // lapjv.cpp:286-319: one path per free row, in list order.
    __device__ __forceinline__ void augment_all(int n_free)
    {
        for (int f = 0; f < n_free && !err; ++f) {
            const int start = fr[f];
            const int target = find_path(start);
            if (err) break;
            if (bc.tid == 0) {
                int j = target, i = -1, hops = 0;
                while (i != start && hops <= n) {
                    i = pred[j];
                    y[j] = i;
                    const int prev = x[i];
                    x[i] = j;
                    j = prev;
                    ++hops;
                }
                if (i != start) ctrl->err = 3;
            }
            __syncthreads();
        }
    }

    // ------------------------------------------------------------------ cold JV
    // lapjv.cpp:8-72.  Returns the number of free rows (list in fr[], ascending).
    __device__ __forceinline__ int cold_column_reduction()
    {
        const int b0 = base();
        double vm[CH];
        int ya[CH];
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            vm[r] = kLarge;
            ya[r] = 0;
        }
        for (int i = 0; i < n; ++i) {
            const double *row = C + (size_t)i * n;
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const int j = b0 + r;
                if (j < n) {
                    const double c = row[j];
                    if (c < vm[r]) {
                        vm[r] = c;
                        ya[r] = i;
                    }
                }
            }
        }
        colred_elems += (long long)n * n;
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int j = b0 + r;
            if (j < n) {
                v[j] = vm[r];
                y[j] = ya[r];
                x[j] = -1;
                pred[j] = 0;  // number of columns whose minimum sits in row j
            }
        }
        __syncthreads();
        // columns are claimed from j = n-1 downwards: the largest j keeps the row
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int j = b0 + r;
            if (j < n) {
                atomicMax(&x[ya[r]], j);
                atomicAdd(&pred[ya[r]], 1);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int j = b0 + r;
            if (j < n && x[ya[r]] != j) y[j] = -1;
        }
        __syncthreads();
        // free rows (ascending) and reduction transfer for rows that own exactly one column;
        // serial over rows because each transfer lowers a v[] that later rows read.
        int nf = 0;
        for (int i = 0; i < n; ++i) {
            const int xi = x[i];
            if (xi < 0) {
                if (bc.tid == 0) fr[nf] = i;
                ++nf;
            } else if (pred[i] == 1) {
                const double *row = C + (size_t)i * n;
                double m = kLarge;
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    const int j2 = b0 + r;
                    if (j2 < n && j2 != xi) {
                        const double c = row[j2] - v[j2];
                        if (c < m) m = c;
                    }
                }
                m = bc.min_f64(m);
                if (xi >= b0 && xi < b0 + CH) v[xi] -= m;  // owner thread only
                transfer_rows++;
            }
        }
        __syncthreads();
        return nf;
    }

    // lapjv.cpp:76-149.  One augmenting-row-reduction sweep over fr[0..n_free).
    __device__ __forceinline__ int cold_arr_sweep(int n_free)
    {
        const int b0 = base();
        unsigned current = 0, rr = 0;
        int new_free = 0;
        int fwd = -1;
        const unsigned un = (unsigned)n;
        while (current < (unsigned)n_free) {
            rr++;
            const int free_i = (fwd >= 0) ? fwd : fr[current];
            fwd = -1;
            current++;
            const double *row = C + (size_t)free_i * n;
            double v1, v2;
            int j1, j2;
            // Normal regime (column 0 not above the sentinel): the two smallest (value, index)
            // pairs among column 0, the columns with c < LARGE and the LARGE sentinel (which
            // only loses a tie to column 0).  c0 is owned by thread 0 and broadcast with the
            // reduction, so no thread reads a v[] entry it does not own before the barrier.
            double cs[CH];
            double c0 = 0.0;
            Top2 t = top2_empty();
            if (bc.tid == 0) top2_push(t, kLarge, kSentinelIdx);
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const int j = b0 + r;
                cs[r] = pos_inf();
                if (j < n) {
                    cs[r] = row[j] - v[j];
                    if (j == 0) c0 = cs[r];
                    if (j == 0 || cs[r] < kLarge) top2_push(t, cs[r], j);
                }
            }
            t = bc.top2_bcast(t, &c0);
            if (c0 <= kLarge) {
                v1 = t.a1;
                j1 = t.i1;
                v2 = t.a2;
                j2 = (t.i2 == kSentinelIdx) ? -1 : t.i2;
            } else {
                // column 0 starts above the sentinel: nothing is accepted before the first
                // column with c < LARGE; from there on it is a plain two-minimum scan that
                // still holds (c0, 0) as a candidate.
                int js = kEmptyIdx;
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    const int j = b0 + r;
                    if (j < n && j >= 1 && cs[r] < kLarge && j < js) js = j;
                }
                js = bc.min_i32(js);
                if (js == kEmptyIdx) {
                    v1 = c0;
                    j1 = 0;
                    v2 = kLarge;
                    j2 = -1;
                } else {
                    Top2 t2 = top2_empty();
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
                        const int j = b0 + r;
                        if (j < n && (j == 0 || j >= js) && cs[r] == cs[r]) top2_push(t2, cs[r], j);
                    }
                    t2 = bc.top2(t2);
                    v1 = t2.a1;
                    j1 = t2.i1;
                    v2 = t2.a2;
                    j2 = t2.i2;
                }
            }
            arr_iters++;
            // uniform reads of entries owned by other threads, then a barrier, then the
            // owners' writes: nobody may see this iteration's update while still reading.
            int i0 = y[j1];
            const int i0_second = (j2 >= 0) ? y[j2] : -1;
            const double vj1 = v[j1];
            __syncthreads();
            const double v1_new = vj1 - (v2 - v1);
            const bool lowers = v1_new < vj1;
            if (rr < current * un) {
                if (lowers) {
                    if (j1 >= b0 && j1 < b0 + CH) v[j1] = v1_new;
                } else if (i0 >= 0 && j2 >= 0) {
                    j1 = j2;
                    i0 = i0_second;
                }
                if (i0 >= 0) {
                    if (lowers) {
                        --current;
                        fwd = i0;
                        if (bc.tid == 0) fr[current] = i0;
                    } else {
                        if (bc.tid == 0) fr[new_free] = i0;
                        ++new_free;
                    }
                }
            } else if (i0 >= 0) {
                if (bc.tid == 0) fr[new_free] = i0;
                ++new_free;
            }
            if (bc.tid == 0) x[free_i] = j1;
            if (j1 >= b0 && j1 < b0 + CH) y[j1] = free_i;
            if (arr_iters > (1 << 26)) {
                err = 4;
                break;
            }
        }
        __syncthreads();
        return new_free;
    }

    // lapjv.cpp:323-346
    __device__ __forceinline__ int cold_solve()
    {
        int nf = cold_column_reduction();
        for (int sweep = 0; nf > 0 && sweep < 2 && !err; ++sweep) nf = cold_arr_sweep(nf);
        if (nf > 0 && !err) augment_all(nf);
        return nf;
    }

    // ------------------------------------------------------------------ seeded phases
    // lapjv_seeded.cpp:79-102: first tight column not yet used, rows in ascending order.
    // Wave 0 walks the per-row tight bitmaps written by the prelude kernel.
    __device__ __forceinline__ void greedy_wave0(const uint32_t *tight_bits, const int *tight_cnt)
    {
        const int lane = bc.lane;
        int nf = 0;
        for (int i = 0; i < n; ++i) {
            int found = -1;
            if (tight_cnt[i] > 0) {
                const uint32_t *rowbits = tight_bits + (size_t)i * W;
                for (int wbase = 0; wbase < W && found < 0; wbase += kWave) {
                    const int idx = wbase + lane;
                    uint32_t word = 0;
                    if (idx < W) word = rowbits[idx] & ~used[idx];
                    const unsigned long long mask = __ballot(word != 0);
                    if (mask) {
                        const int l = __builtin_ctzll(mask);
                        const uint32_t wv = __shfl(word, l, kWave);
                        found = ((wbase + l) << 5) + __builtin_ctz(wv);
                    }
                }
            }
            if (lane == 0) {
                if (found >= 0) {
                    x[i] = found;
                    y[found] = i;
                    used[found >> 5] |= 1u << (found & 31);
                } else {
                    fr[nf] = i;
                }
            }
            if (found < 0) ++nf;
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) ctrl->nfree = nf;
    }

    // lapjv_seeded.cpp:136-159.  Rows are evaluated wave-parallel against the current v; the
    // first row (in list order) whose test fires is applied, later rows are re-evaluated.
    __device__ __forceinline__ void micro_arr(int n_free, const double *u_tight, double tight_eps)
    {
        int start = 0;
        int rounds = 0;
        while (start < n_free) {
            if (bc.tid == 0) ctrl->first_fire = kEmptyIdx;
            __syncthreads();
            for (int f = start + bc.wave; f < n_free; f += bc.nwaves) {
                const int i = fr[f];
                const double ui = u_tight[i];
                const double *row = C + (size_t)i * n;
                Top2 t = top2_empty();
                for (int j = bc.lane; j < n; j += kWave) {
                    const double r = (row[j] - ui) - v[j];
                    if (r == r) top2_push(t, r, j);
                }
                t = wave_top2(t);
                const int j1 = (t.a1 < pos_inf()) ? t.i1 : -1;
                if (j1 >= 0 && (t.a2 - t.a1) > tight_eps && y[j1] < 0) {
                    if (bc.lane == 0) atomicMin(&ctrl->first_fire, f);
                }
            }
            __syncthreads();
            const int ff = ctrl->first_fire;
            if (ff == kEmptyIdx) break;
            // re-evaluate row ff with the whole workgroup and apply it
            {
                const int i = fr[ff];
                const double ui = u_tight[i];
                const double *row = C + (size_t)i * n;
                Top2 t = top2_empty();
                for (int j = bc.tid; j < n; j += blockDim.x) {
                    const double r = (row[j] - ui) - v[j];
                    if (r == r) top2_push(t, r, j);
                }
                t = bc.top2(t);
                if (bc.tid == 0) v[t.i1] += t.a2 - t.a1;
                arr_fired++;
            }
            start = ff + 1;
            if (++rounds > n) {
                err = 5;
                break;
            }
            __syncthreads();
        }
        __syncthreads();
    }
};

template <int CH, int LDSL>
__global__ void __launch_bounds__(1024) jv_instance_kernel(SolverParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int b = blockIdx.x;
    const int n = p.n;
    const int W = (n + 31) >> 5;
    const int Wpad = (W + 1) & ~1;

    Solver<CH, LDSL> s;
    constexpr int SMAX = BatchDepth<CH>::value;
    s.Wpad = Wpad;
    unsigned char *cur = smem;
    BlockExchange *ex = reinterpret_cast<BlockExchange *>(cur);
    cur += sizeof(BlockExchange);
    s.ctrl = reinterpret_cast<Ctrl *>(cur);
    cur += sizeof(Ctrl);
    s.evt = reinterpret_cast<uint32_t *>(cur);
    cur += sizeof(uint32_t) * Wpad;
    s.sbits = reinterpret_cast<uint32_t *>(cur);
    cur += sizeof(uint32_t) * Wpad;
    s.used = reinterpret_cast<uint32_t *>(cur);
    cur += sizeof(uint32_t) * Wpad;
    s.tmpb = reinterpret_cast<uint32_t *>(cur);
    cur += sizeof(uint32_t) * Wpad;
    s.evb = reinterpret_cast<uint32_t *>(cur);
    cur += sizeof(uint32_t) * Wpad * SMAX;
    if constexpr (LDSL > 0) {
        s.dist = reinterpret_cast<double *>(cur);
        cur += sizeof(double) * n;
        s.v = reinterpret_cast<double *>(cur);
        cur += sizeof(double) * n;
        s.order = reinterpret_cast<int *>(cur);
        cur += sizeof(int) * n;
        s.pred = reinterpret_cast<int *>(cur);
        cur += sizeof(int) * n;
        s.y = reinterpret_cast<int *>(cur);
        cur += sizeof(int) * n;
        s.pos = reinterpret_cast<int *>(cur);
        cur += sizeof(int) * n;
        if constexpr (LDSL > 1) {
            s.x = reinterpret_cast<int *>(cur);
            cur += sizeof(int) * n;
            s.fr = reinterpret_cast<int *>(cur);
        } else {
            const size_t o = (size_t)b * n;
            s.x = p.g_x + o;
            s.fr = p.g_fr + o;
        }
    } else {
        const size_t o = (size_t)b * n;
        s.dist = p.g_dist + o;
        s.v = p.g_v + o;
        s.order = p.g_order + o;
        s.pred = p.g_pred + o;
        s.y = p.g_y + o;
        s.x = p.g_x + o;
        s.fr = p.g_fr + o;
        s.pos = p.g_pos + o;
    }
    s.bc.init(ex);
    s.C = p.C + (size_t)b * n * n;
    s.n = n;
    s.W = W;
    s.scan_elems = s.init_elems = s.colred_elems = 0;
    s.paths = s.finds = s.scan_steps = s.arr_iters = s.transfer_rows = s.arr_fired = 0;
    s.step_id = 1;
    s.err = 0;

    const int tid = s.bc.tid;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    unsigned long long t_serial = t_start;
    const int flags = p.inst_flags ? p.inst_flags[b] : 0;
    if (p.mode == kModeSeeded && (flags & kFlagInfeasible)) {
        if (tid == 0) {
            p.ret[b] = -3;
            if (p.stats) {
                for (int q = 0; q < kStatsPerInstance; ++q) p.stats[(size_t)b * kStatsPerInstance + q] = 0;
            }
        }
        return;
    }

    if (tid == 0) {
        s.ctrl->evt_step = 0;
        s.ctrl->err = 0;
        s.ctrl->nfree = 0;
        s.ctrl->hi = 0;
        s.ctrl->target = -1;
        s.ctrl->ev_mask = 0;
        s.ctrl->batch_steps = 0;
        s.ctrl->batch_elems = 0;
    }
    for (int w = tid; w < Wpad; w += blockDim.x) {
        s.evt[w] = 0;
        s.sbits[w] = 0;
        s.used[w] = 0;
        s.tmpb[w] = 0;
    }
    for (int w = tid; w < Wpad * SMAX; w += blockDim.x) s.evb[w] = 0;
    int tight_local = 0;
    for (int j = tid; j < n; j += blockDim.x) {
        s.x[j] = -1;
        s.y[j] = -1;
        if (p.mode == kModeSeeded) {
            s.v[j] = p.v_work[(size_t)b * n + j];
            tight_local += p.tight_cnt[(size_t)b * n + j];
        }
    }
    long long branch = kBranchCold;
    long long tight_total = 0;
    long long free_after_greedy = 0;
    int nf = 0;
    if (p.mode == kModeSeeded) {
        const int tt = s.bc.sum_i32(tight_local);  // includes the barrier that publishes the init
        tight_total = tt;
        const bool fallback = (double)tt < 1.2 * n;  // lapjv_seeded.cpp:116
        if (fallback) {
            branch = kBranchFallback;
            nf = s.cold_solve();
        } else {
            if (s.bc.wave == 0)
                s.greedy_wave0(p.tight_bits + (size_t)b * n * W, p.tight_cnt + (size_t)b * n);
            __syncthreads();
            nf = s.ctrl->nfree;
            free_after_greedy = nf;
            if (nf == 0) {
                branch = kBranchAllMatched;
            } else {
                branch = kBranchSsp;
                s.micro_arr(nf, p.u_tight + (size_t)b * n, p.tight_eps);
                t_serial = __builtin_amdgcn_s_memrealtime();
                if (!s.err) s.augment_all(nf);
            }
        }
    } else {
        __syncthreads();
        nf = s.cold_solve();
        free_after_greedy = nf;
    }
    __syncthreads();
    const int err = s.err | s.ctrl->err;
    for (int j = tid; j < n; j += blockDim.x) {
        if (p.x_out) {
            p.x_out[(size_t)b * n + j] = s.x[j];
            p.y_out[(size_t)b * n + j] = s.y[j];
        }
        if (p.x32_out) {
            p.x32_out[(size_t)b * n + j] = s.x[j];
            p.y32_out[(size_t)b * n + j] = s.y[j];
        }
        if (p.v_out) p.v_out[(size_t)b * n + j] = s.v[j];
    }
    if (tid == 0) {
        p.ret[b] = err ? (-100 - err) : 0;
        if (p.stats) {
            long long *st = p.stats + (size_t)b * kStatsPerInstance;
            st[0] = branch;
            st[1] = tight_total;
            st[2] = (branch == kBranchFallback || branch == kBranchCold) ? nf : free_after_greedy;
            st[3] = s.arr_fired;
            st[4] = s.paths;
            st[5] = s.finds;
            st[6] = s.scan_steps;
            st[7] = s.scan_elems;
            st[8] = s.init_elems;
            st[9] = s.colred_elems;
            st[10] = s.transfer_rows;
            st[11] = s.arr_iters;
            st[12] = err;
            const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
            st[13] = (long long)(t_end - t_start);     // whole kernel, 10 ns ticks
            st[14] = (long long)(t_serial - t_start);  // greedy + micro-ARR part (SSP branch)
            st[15] = 0;
        }
    }
}

template <int CH, int LDSL>
hipError_t launch_one(const SolverParams &p, int threads, size_t lds_bytes, hipStream_t stream)
{
    auto kern = jv_instance_kernel<CH, LDSL>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(p.batch), dim3(threads), lds_bytes, stream, p);
    return hipGetLastError();
}

int batch_depth(int ch) { return ch >= 16 ? 1 : (ch == 8 ? 2 : (ch == 4 ? 4 : 8)); }

}  // namespace

// level 2: every array in LDS; 1: x and the free-row list in global memory; 0: all global
size_t solver_lds_bytes(int n, int ch, int level)
{
    const int W = (n + 31) >> 5;
    const int Wpad = (W + 1) & ~1;
    size_t bytes = sizeof(BlockExchange) + sizeof(Ctrl) + sizeof(uint32_t) * (size_t)Wpad * (4 + batch_depth(ch));
    if (level >= 1) bytes += (size_t)n * (2 * sizeof(double) + 4 * sizeof(int));
    if (level >= 2) bytes += (size_t)n * 2 * sizeof(int);
    return bytes;
}

int solver_lds_level(int n, int ch)
{
    if (solver_lds_bytes(n, ch, 2) <= kLdsBudgetBytes) return 2;
    if (solver_lds_bytes(n, ch, 1) <= kLdsBudgetBytes) return 1;
    return 0;
}

// Picks (threads, CH) with threads*CH >= n.  `threads_hint` (0 = auto) lets the bench sweep
// the geometry; it is rounded to a supported value.
void solver_geometry(int n, int threads_hint, int *threads, int *ch)
{
    int t = threads_hint;
    if (t <= 0) {
        // measured on MI355X (K3, n=2048): 1024 threads 107 ms, 512: 128 ms, 256: 180 ms --
        // the per-step fixed latency dominates, so use as many lanes as there are columns
        if (n <= 64) t = 64;
        else if (n <= 128) t = 128;
        else if (n <= 256) t = 256;
        else if (n <= 512) t = 512;
        else t = 1024;
    }
    t = ((t + 63) / 64) * 64;
    if (t > 1024) t = 1024;
    if (t < 64) t = 64;
    int c = 1;
    while ((long long)t * c < n && c < 16) c <<= 1;
    while ((long long)t * c < n && t < 1024) t += 64;
    *threads = t;
    *ch = c;
}

hipError_t launch_solver(const SolverParams &p, int threads_hint, hipStream_t stream)
{
    int threads, ch;
    solver_geometry(p.n, threads_hint, &threads, &ch);
    if ((long long)threads * ch < p.n) return hipErrorInvalidValue;  // n > 16384
    const int level = solver_lds_level(p.n, ch);
    if (level < 2 && !p.g_dist) return hipErrorInvalidValue;
    const size_t lds = solver_lds_bytes(p.n, ch, level);
#define LAPWARM_CASE(CHV)                                                         \
    case CHV:                                                                     \
        if (level == 2) return launch_one<CHV, 2>(p, threads, lds, stream);       \
        if (level == 1) return launch_one<CHV, 1>(p, threads, lds, stream);       \
        return launch_one<CHV, 0>(p, threads, lds, stream);
    switch (ch) {
        LAPWARM_CASE(1)
        LAPWARM_CASE(2)
        LAPWARM_CASE(4)
        LAPWARM_CASE(8)
        LAPWARM_CASE(16)
    }
#undef LAPWARM_CASE
    return hipErrorInvalidValue;
}

bool solver_needs_global_state(int n)
{
    int threads, ch;
    solver_geometry(n, 0, &threads, &ch);
    int worst = solver_lds_level(n, ch);
    // a threads_hint may pick another CH: be conservative for every supported geometry
    for (int c = 1; c <= 16; c <<= 1) {
        const int l = solver_lds_level(n, c);
        if (l < worst) worst = l;
    }
    return worst < 2;
}

}  // namespace lapwarm
