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
//     (44 n bytes + bitmaps: 91 KiB at n=2048; x and the free list move to global memory above
//     n = 3,634, everything above n = 4,427 -- solver_lds_bytes() is the authority);
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
//   * measured on MI355X: the search tree grows as a CHAIN (a relax step typically uncovers
//     exactly one new tight column), so steps cannot be batched and per-step latency is the
//     whole game.  Each thread therefore keeps the column, its dual and its distance for the
//     positions it owns in registers (a relax step is: one global gather, two subtractions, one
//     compare), and a step with exactly one tie event -- the common case -- is resolved with a
//     single barrier: the finder publishes (column, position, matched row, displaced column) in
//     a double-buffered LDS slot, every thread advances the uniform state from it, and only the
//     owner of the affected position touches order[].  Steps with 2-4 events are resolved the same
//     way from arrival-slot records (round 2); more events, or events inside the swap window,
//     fall back to the ordered replay by wave 0.
//   * round 2: one HELPER workgroup per instance (blocks beyond the batch, same XCD) pulls the rows
//     the solver announces through a ring in global memory -- columns joining the SCAN list, the
//     winner of a minima collection, the next path's start row -- into the L2 both share.  It writes
//     nothing the solver reads.  DESIGN.md section 4 has the measurements and the happens-before table.
#include <stdlib.h>
#include <string.h>

#include "device_utils.hpp"
#include "jv_solver.hpp"

namespace lapwarm {

__host__ __device__ int solver_row_slots(int n, int ch);  // LDS level 8, defined with solver_lds_bytes
bool large_row_geometry(int n, int *threads, int *ch);

namespace {

struct EventSlot {
    int j, p, i, a;  // event column, its position, its matched row (-1: free), column at order[hi]
};

constexpr int kRecEvents = 4;  // tie events of one relax step that are resolved without a second barrier

struct alignas(16) Ctrl {
    double level;
    int hi;
    int target;
    // records of the first kRecEvents tie events of a step, per step parity, in ARRIVAL order
    // (slot = returning atomicAdd on ev_total); entry 0 alone describes a single-event step
    alignas(16) EventSlot rec[2][kRecEvents];
    // columns at order[hi .. hi+3], published each step by the owners of those positions: what
    // the events of the step will displace
    alignas(16) int a_pub[2][kRecEvents];
    int tie_find;     // sequence number of the last minima collection that saw a tie
    int ev_total[2];  // monotonic tie-event counters, one per step parity (readers diff them;
                      // a reader of step t can never see an increment of step t+1)
    int first_fire;
    int nfree;
    int err;
    int head_j, head_i;
    int free_pos[2];  // smallest position of a tie event whose column is FREE, per step parity
                      // (INT_MAX: none); lets the ordered replay skip the y[] lookup per event
};
// Augmenting row reduction with candidate lists: the loop state wave 0 hands to the workgroup when a row
// needs the full scan.  It lives in Ctrl::rec, which only the shortest-path phase uses (a Ctrl that grew by
// these 28 bytes shifted every LDS array of the seeded kernel and cost K3 2 %: 76.6 -> 78.4 ms).
struct ArrState {
    unsigned arr_current, arr_rr;
    int arr_new_free, arr_fwd, arr_iters, arr_fast, arr_err;
};
static_assert(sizeof(ArrState) <= sizeof(EventSlot) * 2 * kRecEvents, "ArrState is overlaid on Ctrl::rec");

constexpr int kSentinelIdx = 0x7ffffffe;  // the LARGE sentinel of the ARR scan (index -1 in the reference)
constexpr int kEmptyIdx = 0x7fffffff;

#ifdef LAPWARM_STAMPS
// s_memtime returns through the LGKM counter and out of order with LDS reads: the wait must be
// part of the same asm statement, or the compiler's counted lgkmcnt waits pair up with the wrong
// LDS read and the kernel computes on stale registers (cdna_hip_programming.md, in-kernel stamps).
__device__ __forceinline__ unsigned long long stamp_now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#ifndef LAPWARM_STAMP_GROUPS
#define LAPWARM_STAMP_GROUPS 7  // bit 0: minima collection, bit 1: relax step, bit 2: path
#endif
#define STAMP_G(g, var) const unsigned long long var = ((LAPWARM_STAMP_GROUPS) & (g)) ? stamp_now() : 0ull
#ifndef LAPWARM_FIND_MASK
#define LAPWARM_FIND_MASK 0x1ff
#endif
#define STAMP_FI(bit, var) const unsigned long long var = (((LAPWARM_STAMP_GROUPS) & 1) && ((LAPWARM_FIND_MASK) >> (bit) & 1)) ? stamp_now() : 0ull
#define STAMP(var) STAMP_G(1, var)
#define STAMPR(var) STAMP_G(2, var)
#define STAMPP(var) STAMP_G(4, var)
#define STAMP_ADD(slot, t1, t0) stamps[slot] += (long long)((t1) - (t0))
#define STAMP_INC(slot) stamps[slot] += 1
#else
#define STAMP(var)
#define STAMPR(var)
#define STAMPP(var)
#define STAMP_FI(bit, var)
#define STAMP_ADD(slot, t1, t0)
#define STAMP_INC(slot)
#endif

template <int CH, int LDSL, int TB>
struct Solver {
    // LDS levels: 2 = all solver state in LDS, 1 = x and the free-row list in global memory,
    // 0 = all state in global memory, 8 = as 0 with the head rows staged in LDS (below)
    static constexpr bool LDS_STATE = LDSL > 0 && LDSL != 8;
    // level 8 (rows longer than the CU's L1 can hold, n > 4,427): solver state in global memory as
    // level 0, but EVERY head row is brought into an LDS slot by coalesced LDS-DMA requests and
    // gathered from there -- the position-owned gather C[i][order[k]] touches a random 128-byte line
    // per lane, and a 64-128 KiB row thrashes the 32-KiB L1 (each line re-fetched up to CH times)
    static constexpr bool ROWLDS = LDSL == 8;
    static constexpr bool PF = ROWLDS;  // ... and the next queued head's row requested one step ahead
    static constexpr int kCacheLimit = (TB <= 256) ? 16 : ((TB <= 512) ? 32 : 4);  // positions per thread whose duals fit in registers
    static constexpr int kCacheLimitY = (TB <= 256) ? 16 : 2;
    // problem
    const double *C;
    int n, W;
    // state
    double *dist, *v;
    int *order, *pred, *y, *x, *fr;
    int *ring;     // helper experiment: ring of upcoming head rows (null: off)
    int ring_count;
    unsigned char *slots;  // level 8: row slots of slot_bytes each
    int slot_bytes;
    int nslots;
    uint32_t *evt, *sbits, *used;
    uint32_t *evb;  // tie-event bitmap of a relax step, TWO copies selected by the step parity: the
                    // post phase of step t clears bits while pass t+1 may already be setting its own
    int Wpad;
    int *evl;     // n entries: events of one minima collection, in position order
    int *tmpcol;  // n+1 entries: tie columns while they are re-packed
    Ctrl *ctrl;
    BlockCtx bc;
    // uniform counters (identical in every thread)
    long long scan_elems, init_elems, colred_elems;
    int paths, finds, scan_steps, arr_iters, transfer_rows, arr_fired;
    int step_id;
    int ctrl_seen0, ctrl_seen1, ctrl_find_seq;
    int err;
#ifdef LAPWARM_STAMPS
    long long stamps[16];
#endif

    __device__ __forceinline__ int base() const { return bc.tid * CH; }

    // Announce the row of a column that just joined the SCAN list (or the start row of the next
    // path) to the helper workgroup.  Called by every thread with the same argument: ring_count is
    // uniform, thread 0 stores.  One word per slot carries its own generation, so no ordering
    // between two stores is needed.
    __device__ __forceinline__ void ring_push(int row)
    {
        if (ring && (unsigned)row < (unsigned)n) {
            if (bc.tid == 0) {
                const int gen = (ring_count >> 6) + 1;
                __hip_atomic_store(&ring[2 + (ring_count & (kRingSlots - 1))], (gen << 16) | row, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            ++ring_count;
        }
    }
    // ... and the rows of the SCAN entries lo+1 .. hi-1 after a minima collection with ties (the
    // head at lo is needed at once; the others have a lead).  Wave 0 stores, one entry per lane.
    __device__ __forceinline__ void ring_push_scan(int lo, int hi)
    {
        if (!ring) return;
        int cnt = hi - lo - 1;
        if (cnt <= 0) return;
        cnt = (cnt < kRingSlots - 2) ? cnt : kRingSlots - 2;
        if (bc.wave == 0 && bc.lane < cnt) {
            const int j = order[lo + 1 + bc.lane];
            const int i = ((unsigned)j < (unsigned)n) ? y[j] : -1;
            const int c = ring_count + bc.lane;
            const int gen = (c >> 6) + 1;
            // (an unmatched column ends the path when it is reached: nothing to fetch)
            const int word = (gen << 16) | (((unsigned)i < (unsigned)n) ? i : 0xffff);
            __hip_atomic_store(&ring[2 + (c & (kRingSlots - 1))], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ring_count += cnt;
    }

    __device__ __forceinline__ void fence_if_global()
    {
        if constexpr (!LDS_STATE) __threadfence_block();
    }

    // ------------------------------------------------------------------ event replay (wave 0)
    // Minima collection with ties, lapjv.cpp:153-171, from the event / strict bitmaps.
    //
    // Events in position order: e_0 < e_1 < ...  Let L be the index of the last STRICT event.
    // Usual shape (measured: clustered family, ~3 strict events then ~150 ties with the global
    // minimum): every event up to L is strict, everything after it is a tie.  Then
    //   * events 0..L shift: position e_i receives the column that sat at e_(i-1) (at lo for
    //     i = 0) and slot lo receives the column of e_L;
    //   * the T ties after L fill slots lo+1 .. lo+T in order; slot lo+s held some column B
    //     before: if lo+s is not itself a tie position, B ends at the first tie position OUTSIDE
    //     the window reached by hopping s -> (t_s - lo) -> ...  (each hop is one serial swap that
    //     moved B on); hops only go up, so every slot is resolved independently.
    // Both parts are data-parallel over the lanes of wave 0.  A tie BEFORE the last strict event
    // (rare) takes the serial loop.
    __device__ __forceinline__ void replay_find(int lo)
    {
        const int lane = bc.lane;
        // ---- 1. ordered event list: evl[i] = position | strict << 31
        const int wpl = (W + kWave - 1) / kWave;  // bitmap words per lane, contiguous per lane
        int mine = 0;
        for (int q = 0; q < wpl; ++q) {
            const int idx = lane * wpl + q;
            if (idx < W) mine += __popc(evt[idx]);
        }
        int incl = mine;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const int o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        const int E = __shfl(incl, kWave - 1, kWave);
        int slot = incl - mine;
        int last_strict_local = -1;
        for (int q = 0; q < wpl; ++q) {
            const int idx = lane * wpl + q;
            if (idx < W) {
                uint32_t ew = evt[idx];
                const uint32_t sw = sbits[idx];
                if (ew) {
                    evt[idx] = 0;
                    sbits[idx] = 0;
                }
                while (ew) {
                    const int bit = __builtin_ctz(ew);
                    ew &= ew - 1;
                    const int st = (sw >> bit) & 1u;
                    if (st) last_strict_local = slot;
                    evl[slot++] = ((idx << 5) + bit) | (st << 31);
                }
            }
        }
        int L = last_strict_local;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const int o = __shfl_xor(L, m, kWave);
            L = (o > L) ? o : L;
        }
        // a tie before the last strict event?
        bool early_tie = false;
        for (int i = lane; i < L; i += kWave) early_tie |= (evl[i] >= 0);
        int hi;
        if (__ballot(early_tie) != 0ull || L + 1 > kWave) {
            // ---- serial replay (exact for any pattern)
            hi = lo + 1;
            for (int i = 0; i < E; ++i) {
                const int ev = evl[i];
                const int k = ev & 0x7fffffff;
                const int j = order[k];
                if (ev < 0) hi = lo;
                const int a = order[hi];
                if (lane == 0) {
                    order[k] = a;
                    order[hi] = j;
                }
                fence_if_global();
                ++hi;
            }
        } else {
            // ---- 2. strict prefix 0..L: shift
            const int T = E - (L + 1);
            for (int base_i = 0; base_i <= L; base_i += kWave) {
                const int i = base_i + lane;
                int newcol = 0, e = 0;
                if (i <= L) {
                    e = evl[i] & 0x7fffffff;
                    newcol = order[(i == 0) ? lo : (evl[i - 1] & 0x7fffffff)];
                }
                const int lastcol = (L >= 0) ? order[evl[L] & 0x7fffffff] : 0;
                // all reads of this chunk precede its writes (one wave, in-order LDS); chunks
                // beyond the first read order[evl[i-1]] of the previous chunk's last event, which
                // that chunk has already overwritten -- keep L+1 <= 64 on this path
                if (i <= L) order[e] = newcol;
                if (base_i == 0 && L >= 0 && lane == 0) order[lo] = lastcol;
            }
            fence_if_global();
            // ---- 3. tie tail
            const int tb = L + 1;  // evl[tb + s - 1] = position of tie number s (1-based)
            for (int s0 = 1; s0 <= T; s0 += kWave) {  // save the tie columns
                const int sidx = s0 + lane;
                if (sidx <= T) tmpcol[sidx] = order[evl[tb + sidx - 1] & 0x7fffffff];
            }
            fence_if_global();
            for (int s0 = 1; s0 <= T; s0 += kWave) {  // move the displaced columns out of the window
                const int sidx = s0 + lane;
                if (sidx <= T) {
                    const int p0 = lo + sidx;
                    // is p0 itself one of the tail ties?  t_s >= lo+s, so look at ties <= s
                    int tpos = evl[tb + sidx - 1] & 0x7fffffff;
                    bool is_tie;
                    {
                        // binary search p0 among the (sorted) tie positions t_1..t_sidx
                        int lo_i = 1, hi_i = sidx;
                        while (lo_i < hi_i) {
                            const int mid = (lo_i + hi_i) >> 1;
                            if ((evl[tb + mid - 1] & 0x7fffffff) < p0)
                                lo_i = mid + 1;
                            else
                                hi_i = mid;
                        }
                        is_tie = ((evl[tb + lo_i - 1] & 0x7fffffff) == p0);
                    }
                    if (!is_tie) {
                        const int col = order[p0];
                        int p = tpos;
                        int guard = 0;
                        while (p <= lo + T && guard++ < n) p = evl[tb + (p - lo) - 1] & 0x7fffffff;
                        order[p] = col;  // a tie position outside the window: nobody reads it again
                    }
                }
            }
            fence_if_global();
            for (int s0 = 1; s0 <= T; s0 += kWave) {  // pack the ties behind slot lo
                const int sidx = s0 + lane;
                if (sidx <= T) order[lo + sidx] = tmpcol[sidx];
            }
            fence_if_global();
            hi = lo + 1 + T;
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
        const int head_j = order[lo];
        const double level = dist[head_j];
        const int head_i = y[head_j];
        if (lane == 0) {
            ctrl->hi = hi;
            ctrl->target = target;
            ctrl->level = level;
            ctrl->head_j = head_j;
            ctrl->head_i = head_i;
        }
    }

    // Several tie events in one relax step (lapjv.cpp:199-205): replay them in position order.
    __device__ __forceinline__ void replay_scan(int hi, int par)
    {
        const int lane = bc.lane;
        uint32_t *evb = this->evb + (size_t)par * Wpad;
        int target = -1;
        // the first event (in position order) whose column is free ends the scan (lapjv.cpp:200-201);
        // its finder recorded the position, so no event needs its y[] looked up here
        const int fp = uni(ctrl->free_pos[par]);
        if (fp != 0x7fffffff && lane == 0) ctrl->free_pos[par] = 0x7fffffff;
        for (int wbase = 0; wbase < W; wbase += kWave) {
            const int idx = wbase + lane;
            uint32_t ew = 0;
            if (idx < W) {
                ew = evb[idx];
                if (ew) evb[idx] = 0;
            }
            unsigned long long mask = __ballot(ew != 0);
            while (mask && target < 0) {
                const int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                uint32_t e = __shfl(ew, l, kWave);
                while (e && target < 0) {
                    const int bit = __builtin_ctz(e);
                    e &= e - 1;
                    const int k = ((wbase + l) << 5) + bit;
                    const int j = order[k];
                    if (k == fp) {
                        target = j;
                    } else {
                        const int a = order[hi];
                        if (lane == 0) {
                            order[k] = a;
                            order[hi] = j;
                        }
                        fence_if_global();
                        ++hi;
                    }
                }
            }
        }
        if (lane == 0) {
            ctrl->hi = hi;
            ctrl->target = target;
        }
    }

    // ------------------------------------------------------------------ one shortest path
    // lapjv.cpp:221-282.  Returns the free column reached; updates v for the READY columns.
    __device__ __forceinline__ int find_path(int start)
    {
        const int b0 = base();
        const int wordi = b0 >> 5, shift = b0 & 31;
        STAMPP(tpath);
        // the duals of the owned columns are cached in registers while the budget allows
        // (1024-thread workgroups cap a thread at 128 VGPRs)
        constexpr bool CACHE_V = CH <= kCacheLimit;
        double dk[CH], vr[CACHE_V ? CH : 1];
        constexpr bool CACHE_Y = CH <= kCacheLimitY;
        int jr[CH], yr[CACHE_Y ? CH : 1];  // yr: matched row of the owned column (y is constant during a path)
        {
            // every load is issued before the first use: indices are clamped instead of
            // branching, so the CH gathers of a thread are all in flight together
            const double *row = C + (size_t)start * n;
            double c0[CH];
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const int k = b0 + r;
                jr[r] = (k < n) ? k : n - 1;
                c0[r] = row[jr[r]];
            }
#pragma unroll
            for (int r = 0; r < CH; ++r) pin(c0[r]);
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const int k = b0 + r;
                const double vk = v[jr[r]];
                if constexpr (CACHE_V) {
                    vr[r] = vk;
                    if constexpr (CACHE_Y) yr[r] = y[jr[r]];
                }
                const double val = c0[r] - vk;
                dk[r] = (k < n) ? val : pos_inf();
                if (k < n) {
                    order[k] = k;
                    pred[k] = start;
                    dist[k] = val;
                }
            }
        }
        paths++;
        init_elems += n;
        int lo = 0, hi = 0, ready = 0, target = -1;
        int head_j = 0, head_i = 0;
        STAMPP(tp0);
        STAMP_ADD(8, tp0, tpath);
        int seen0 = ctrl_seen0, seen1 = ctrl_seen1;
        int find_seq = ctrl_find_seq;
        // columns appended to the SCAN list by the previous step's single-barrier paths (their
        // owners may still be writing order[]): position app_pos and, after a multi-event step,
        // app_pos + 1 -- the only two fresh entries the directly following step can look at
        int app_pos = -1, app_j = 0, app_i = 0, app_j1 = 0, app_i1 = 0;
        bool app_two = false;
        // 2-4 tie events behind the single barrier: only with <= 4 positions per thread (with 8+
        // the extra live values of that path spill: n = 16384 went from 25 s to 55 s with it)
        constexpr bool kFastMulti = CH <= 4;
#ifdef LAPWARM_DIAG_BARRIER
        int diag_prev_cnt = -1;
#endif
        bool pf_have = false;  // level 8: the current head's row sits in slot pf_slot
        int pf_slot = 0;
        double level = 0.0;
        int guard = 0;
        while (target < 0) {
            if (lo == hi) {
                // ---------------- minima collection over positions [lo, n); dk[] is current.
                // Scan of the lexicographic (distance, position) minimum = "the first position that
                // holds the running minimum".  A position is a STRICT event when it undercuts the
                // minimum of everything before it, a TIE event when it equals it.  Without tie
                // events the serial swap sequence (lapjv.cpp:158-168) collapses to a shift: every
                // event position receives the column of the previous record holder and slot lo
                // receives the final minimum -- applied here by the event owners themselves, no
                // ordered replay.  Any tie anywhere sends the whole collection to the exact replay.
                STAMP_FI(0, tf0);
                ready = lo;
                ++find_seq;
                double tv = pos_inf();
                int tp = 0x7fffffff;
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    const int k = b0 + r;
                    // position lo always starts as the holder, whatever its value (even +inf)
                    if (k >= lo && k < n && (k == lo || dk[r] < tv)) {
                        tv = dk[r];
                        tp = k;
                    }
                }
                double runv = tv, wtv;
                int runp = tp, wtp;
                wave_excl_prefix_min_pair(runv, runp, bc.lane, &wtv, &wtp);
                const int xp = bc.parity;
                bc.parity ^= 1;
                if (bc.lane == 0) {
                    bc.ex->d[xp][bc.wave] = wtv;
                    bc.ex->i[xp][bc.wave] = wtp;
                }
                STAMP_FI(1, tfa);
                STAMP_ADD(9, tfa, tf0);
                __syncthreads();
                STAMP_FI(2, tfb);
                STAMP_ADD(10, tfb, tfa);
                double totv;
                int totp;
                {
                    // one LDS round trip: lane l reads the slot of wave (l & 15); ONE inclusive
                    // prefix-min scan along each row of 16 lanes (four DPP row_shr steps) holds both
                    // answers: lane 15 = block total, lane wave-1 = minimum over the earlier waves
                    const int w = bc.lane & (kMaxWaves - 1);
                    double av = (w < bc.nwaves) ? bc.ex->d[xp][w] : pos_inf();
                    int ap = (w < bc.nwaves) ? bc.ex->i[xp][w] : 0x7fffffff;
                    scan_step_min_pair<kDppRowShr1, 0xf>(av, ap);
                    scan_step_min_pair<kDppRowShr2, 0xf>(av, ap);
                    scan_step_min_pair<kDppRowShr4, 0xf>(av, ap);
                    scan_step_min_pair<kDppRowShr8, 0xf>(av, ap);
                    totv = readlane_f64(av, kMaxWaves - 1);
                    totp = __builtin_amdgcn_readlane(ap, kMaxWaves - 1);
                    if (bc.wave > 0) {
                        const int wl = uni(bc.wave) - 1;
                        const double pv = readlane_f64(av, wl);
                        const int pp = __builtin_amdgcn_readlane(ap, wl);
                        if (pair_less(pv, pp, runv, runp)) {
                            runv = pv;
                            runp = pp;
                        }
                    }
                }
                STAMP_FI(3, tfc);
                // classify the owned positions; remember what each strict event receives
                uint32_t eb = 0, sb = 0;
                bool tie = false;
                int prevpos[CH];
                double prevval[CH];
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    const int k = b0 + r;
                    prevpos[r] = -1;
                    prevval[r] = 0.0;
                    if (k >= lo && k < n) {
                        if (k > lo && dk[r] <= runv) {
                            eb |= 1u << r;
                            if (dk[r] < runv) {
                                sb |= 1u << r;
                                prevpos[r] = runp;
                                prevval[r] = runv;
                            } else {
                                tie = true;
                            }
                        }
                        if (k == lo || dk[r] < runv) {
                            runv = dk[r];
                            runp = k;
                        }
                    }
                }
                int prevcol[CH];
#pragma unroll
                for (int r = 0; r < CH; ++r) prevcol[r] = (prevpos[r] >= 0) ? order[prevpos[r]] : 0;
                const int min_col = uni(order[totp]);  // column of the global minimum (uniform)
                if (tie) ctrl->tie_find = find_seq;
                // The matched row of the winning column decides whether the path ends here.  It MUST
                // be read before the barrier: once the waves are released, a fast wave can finish
                // the path and thread 0's backtrack (augment_all) rewrites y[] -- a slow wave that
                // looked y[min_col] up after the barrier could then see the column as matched, carry
                // on alone and corrupt the search (found with tools/stress_determinism.py: ~1 run
                // in 100-300 ended in a consistency guard; 0 in 900 with this order).
                const int min_row_raw = y[min_col];
                // the likely next head (certain without ties), a whole collection tail ahead of its use
                ring_push(uni(min_row_raw));
                STAMP_FI(4, tfd);
                STAMP_ADD(11, tfd, tfc);
                __syncthreads();
                STAMP_FI(5, tfe);
                STAMP_ADD(12, tfe, tfd);
                finds++;
                if (uni(ctrl->tie_find) != find_seq) {
                    // ---- tie-free.  Does the path end here?  Decided first, from values read
                    // BEFORE the barrier: on a path-ending collection nothing below may touch
                    // y[] / v[] / order[] -- another wave may already be past the loop.
                    hi = lo + 1;
                    level = totv;
                    head_j = min_col;
                    head_i = uni(min_row_raw);
                    target = (head_i < 0) ? head_j : -1;
                    if (target >= 0) break;
                    // apply the shift locally
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
                        if (sb & (1u << r)) {
                            const int k = b0 + r;
                            order[k] = prevcol[r];
                            jr[r] = prevcol[r];
                            dk[r] = prevval[r];
                            if constexpr (CACHE_V) {
                                vr[r] = v[prevcol[r]];
                                if constexpr (CACHE_Y) yr[r] = y[prevcol[r]];
                            }
                        }
                    }
                    if (totp != lo && lo >= b0 && lo < b0 + CH) order[lo] = min_col;
                    STAMP_FI(6, tff);
                    STAMP_ADD(13, tff, tfe);
                } else {
                    // ---- ties: exact ordered replay by wave 0 (bitmaps are only built here)
                    if (eb) {
                        atomicOr(&evt[wordi], eb << shift);
                        if (sb) atomicOr(&sbits[wordi], sb << shift);
                    }
                    __syncthreads();
                    if (bc.wave == 0) replay_find(lo);
                    __syncthreads();
                    hi = uni(ctrl->hi);
                    target = uni(ctrl->target);
                    level = uni(ctrl->level);
                    head_j = uni(ctrl->head_j);
                    head_i = uni(ctrl->head_i);
                    STAMP_FI(7, tfg);
                    STAMP_ADD(14, tfg, tfe);
                    STAMP_INC(15);
                    if (target >= 0) break;
                    ring_push_scan(lo, hi);
                    // the collection permuted order[]: rebind the registers of the positions we own
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
                        const int k = b0 + r;
                        if (k >= hi && k < n) {
                            const int j = order[k];
                            if (j != jr[r]) {  // most positions keep their column
                                jr[r] = j;
                                if constexpr (CACHE_V) {
                                    vr[r] = v[j];
                                    if constexpr (CACHE_Y) yr[r] = y[j];
                                }
                                dk[r] = dist[j];
                            }
                        }
                    }
                }
                STAMP_FI(8, tf1);
                STAMP_ADD(0, tf1, tf0);
            }
            // ---------------- relax the head of the SCAN list (lapjv.cpp:185-207)
            STAMPR(tr0);
            // a row index must be a matched row; anything else means corrupted state -- never
            // turn it into a global address
            if ((unsigned)head_i >= (unsigned)n || (unsigned)head_j >= (unsigned)n) {
                err = 6;
                break;
            }
            const double *row = C + (size_t)head_i * n;
            double c[CH];
            double c_head;
            bool from_slot = false;
            if constexpr (PF) from_slot = pf_have;
            if constexpr (ROWLDS) {
                if (!from_slot) {
                    // not requested ahead: request it now, wait, and let one barrier publish it
                    pf_slot = 0;
                    const unsigned sbase = lds_address(slots) + (unsigned)bc.wave * 1024u;
                    const int nt = (int)blockDim.x;
#pragma unroll
                    for (int q = 0; q < CH / 2; ++q) {
                        int col = q * 2 * nt + 2 * bc.tid;
                        col = (col < n - 2) ? col : n - 2;
                        dma_request16(row + col, sbase + (unsigned)q * (unsigned)nt * 16u);
                    }
                    dma_wait<0>();
                    __syncthreads();
                    from_slot = true;
                }
            }
            const double *srow = reinterpret_cast<const double *>(slots + (size_t)pf_slot * slot_bytes);
            if constexpr (ROWLDS) {
                // the row is in LDS: the compare loop below reads it element by element (no
                // CH-sized staging array: CH = 8..16 positions per thread would spill it)
                c_head = srow[head_j];
            } else if (from_slot) {
                // the row was requested during the previous step and every wave waited for its
                // pieces before that step's barrier: gather from LDS
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    const unsigned jc = umin_u32((unsigned)jr[r], (unsigned)(n - 1));
                    c[r] = srow[jc];
                }
                c_head = srow[head_j];
            } else {
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    // never form a global address from an out-of-range column: a clamp (one
                    // instruction; the step is instruction-issue bound) rather than a test and branch.
                    // jr[] only ever holds entries of order[], so the clamp is a no-op by construction.
                    const unsigned jc = umin_u32((unsigned)jr[r], (unsigned)(n - 1));
                    c[r] = row[jc];  // unconditional: all gathers in flight
                }
                c_head = row[head_j];
            }
            const int par = step_id & 1;
            // per-step bookkeeping goes here, in the shadow of the gathers, not behind the barrier
            step_id++;
            scan_steps++;
            scan_elems += (long long)(n - hi);
            if (++guard > 2 * n + 4) {
                err = 1;
                break;
            }
            // While the gathers are in flight: (a) the next queued SCAN column, if there is one,
            // is already known -- fetch it and its row now; (b) the owner of position hi publishes
            // the column sitting there (what a single tie event will displace).
            const bool queued = (lo + 1 < hi);
            int nq_j = 0, nq_i = 0;
            const int fwd_pos = app_pos;  // only meaningful for the pass that directly follows
            const bool fwd_two = app_two;
            app_pos = -1;
            app_two = false;
            if (queued) {
                if (fwd_two && lo == fwd_pos) {
                    nq_j = app_j1;
                    nq_i = app_i1;
                } else if (lo + 1 == fwd_pos) {
                    // appended by the previous step's single-event path: its owner may still be
                    // writing order[app_pos] (no barrier since), so take it from registers
                    nq_j = app_j;
                    nq_i = app_i;
                } else {
                    nq_j = order[lo + 1];
                    nq_i = y[nq_j];
                }
            }

            bool pf_issued = false;
            int pf_next_slot = 0;
            if constexpr (PF) {
                // Request the row of the next queued head into the slot this step does not read.
                // The other slot's last readers finished before the previous step's barrier.
                const int ri = uni(nq_i);
                if (queued && (unsigned)ri < (unsigned)n && nslots > 1) {
                    pf_next_slot = from_slot ? (pf_slot ^ 1) : 0;
                    const double *nrow = C + (size_t)ri * n;
                    const unsigned sbase = lds_address(slots) + (unsigned)pf_next_slot * (unsigned)slot_bytes +
                                           (unsigned)bc.wave * 1024u;
                    const int nt = (int)blockDim.x;
#pragma unroll
                    for (int q = 0; q < CH / 2; ++q) {
                        // lane's 16 bytes: columns q*2*nt + 2*tid, +1 (clamped into the row; the
                        // padding beyond n is never read)
                        int col = q * 2 * nt + 2 * bc.tid;
                        col = (col < n - 2) ? col : n - 2;
                        dma_request16(nrow + col, sbase + (unsigned)q * (unsigned)nt * 16u);
                    }
                    pf_issued = true;
                }
            }
            if constexpr (kFastMulti) {
                // owners of positions hi .. hi+3 publish the columns sitting there
                if (b0 <= hi + kRecEvents - 1 && b0 + CH - 1 >= hi) {  // our positions overlap the window
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
                        const unsigned w = (unsigned)(b0 + r - hi);
                        if (w < (unsigned)kRecEvents && b0 + r < n) ctrl->a_pub[par][w] = jr[r];
                    }
                }
            } else {
                if (hi >= b0 && hi < b0 + CH && hi < n) {
#pragma unroll
                    for (int r = 0; r < CH; ++r)
                        if (b0 + r == hi) ctrl->a_pub[par][0] = jr[r];
                }
            }
            const double v_head = v[head_j];
            if constexpr (!ROWLDS) {
#pragma unroll
                for (int r = 0; r < CH; ++r) pin(c[r]);
            }
            pin(c_head);
            STAMPR(tr1);
            STAMP_ADD(1, tr1, tr0);
            const double h = (c_head - v_head) - level;
            // branch-free core (select, not jump, per element); the LDS bookkeeping of improved
            // columns and the rare tie events sit behind one thread-level test each
            int my_events = 0, ev_r = -1;
            unsigned imp_bits = 0, ev_mask = 0;
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const int k = b0 + r;
                const bool act = (k >= hi) & (k < n);
                double vj;
                if constexpr (CACHE_V)
                    vj = vr[r];
                else
                    vj = v[jr[r]];
                double cr;
                if constexpr (ROWLDS)
                    cr = srow[umin_u32((unsigned)jr[r], (unsigned)(n - 1))];
                else
                    cr = c[r];
                const double cand = (cr - vj) - h;
                const bool imp = act & (cand < dk[r]);
                const bool ev = imp & (cand == level);
                dk[r] = imp ? cand : (act ? dk[r] : pos_inf());
                imp_bits |= imp ? (1u << r) : 0u;
                ev_mask |= ev ? (1u << r) : 0u;
            }
            if (imp_bits) {
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    if ((imp_bits >> r) & 1u) {
                        dist[jr[r]] = dk[r];
                        pred[jr[r]] = head_i;
                    }
                }
                if (ev_mask) {
#pragma unroll
                    for (int r = CH - 1; r >= 0; --r) {
                        if ((ev_mask >> r) & 1u) {
                            ev_r = r;
                            ++my_events;
                            const int k = b0 + r;
                            atomicOr(&evb[par * Wpad + (k >> 5)], 1u << (k & 31));
                            int yj;
                            if constexpr (CACHE_Y)
                                yj = yr[r];
                            else
                                yj = y[jr[r]];
                            if (yj < 0) atomicMin(&ctrl->free_pos[par], k);
                        }
                    }
                }
            }
            const int seen = par ? seen1 : seen0;
            if (my_events && !kFastMulti) {
                // no returning atomic and no LDS read on this path: if this turns out to be the
                // only event of the step the record is exactly right, otherwise nobody reads it
                atomicAdd(&ctrl->ev_total[par], my_events);
                EventSlot sl;
                sl.j = 0;
                sl.i = 0;
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    if (r == ev_r) {
                        sl.j = jr[r];
                        if constexpr (CACHE_Y)
                            sl.i = yr[r];
                        else
                            sl.i = y[jr[r]];
                    }
                }
                sl.p = b0 + ev_r;
                sl.a = 0;
                ctrl->rec[par][0] = sl;
            }
            if (my_events && kFastMulti) {
                // arrival slot of our first event (the counter is cumulative: `seen` is what it
                // held when the step began); the first kRecEvents events of a step leave a record
                int slot_idx = atomicAdd(&ctrl->ev_total[par], my_events) - seen;
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    if ((ev_mask >> r) & 1u) {
                        if (slot_idx < kRecEvents) {
                            EventSlot sl;
                            sl.j = jr[r];
                            if constexpr (CACHE_Y)
                                sl.i = yr[r];
                            else
                                sl.i = y[jr[r]];
                            sl.p = b0 + r;
                            sl.a = 0;
                            ctrl->rec[par][slot_idx] = sl;
                        }
                        ++slot_idx;
                    }
                }
            }
            STAMPR(tr2);
            STAMP_ADD(2, tr2, tr1);
            if constexpr (PF) {
                // every wave's pieces of the requested row have landed before the barrier releases
                // the readers of the next step
                if (pf_issued) dma_wait<0>();
                pf_have = pf_issued;
                pf_slot = pf_next_slot;
            }
            __syncthreads();
            STAMPR(tr3);
            STAMP_ADD(3, tr3, tr2);
#ifdef LAPWARM_DIAG_BARRIER  // barrier wait by the event count of the PREVIOUS step (slots 9-12 reused)
            if (diag_prev_cnt == 0) {
                STAMP_ADD(9, tr3, tr2);
                STAMP_INC(10);
            } else if (diag_prev_cnt == 1) {
                STAMP_ADD(11, tr3, tr2);
                STAMP_INC(12);
            }
#endif
            // one LDS round trip for everything the post phase can need
            const int tot_raw = ctrl->ev_total[par];
            EventSlot sl = ctrl->rec[par][0];
            const int a_raw = ctrl->a_pub[par][0];
            const int tot = uni(tot_raw);
            const int cnt = tot - seen;
#ifdef LAPWARM_DIAG_BARRIER
            diag_prev_cnt = cnt;
#endif
            if (par)
                seen1 = tot;
            else
                seen0 = tot;
            if (cnt == 0) {
                STAMP_INC(5);
                ++lo;
                if (queued) {
                    head_j = uni(nq_j);
                    head_i = uni(nq_i);
                }
            } else if (cnt == 1) {
                STAMP_INC(6);
                sl.j = uni(sl.j);
                sl.p = uni(sl.p);
                sl.i = uni(sl.i);
                sl.a = uni(a_raw);
                if ((unsigned)sl.j >= (unsigned)n || (unsigned)sl.a >= (unsigned)n || (unsigned)sl.p >= (unsigned)n ||
                    sl.i >= n) {
                    err = 8;
                    break;
                }
                if (sl.p >= b0 && sl.p < b0 + CH) {
                    // we own the event position: it now holds the column displaced from order[hi]
                    evb[par * Wpad + (sl.p >> 5)] = 0;  // the only bit set in this step's copy
                    if (sl.i >= 0) {
                        const double va = v[sl.a], da = dist[sl.a];
                        const int ya = y[sl.a];
#pragma unroll
                        for (int r = 0; r < CH; ++r) {
                            if (b0 + r == sl.p) {
                                jr[r] = sl.a;
                                if constexpr (CACHE_V) {
                                    vr[r] = va;
                                    if constexpr (CACHE_Y) yr[r] = ya;
                                }
                                dk[r] = da;
                            }
                        }
                        order[sl.p] = sl.a;
                        order[hi] = sl.j;
                    }
                }
                if (sl.i < 0) {
                    if (bc.tid == 0) ctrl->free_pos[par] = 0x7fffffff;  // (its finder recorded it)
                    target = sl.j;
                    break;
                }
                app_pos = hi;
                app_j = sl.j;
                app_i = sl.i;
                ring_push(sl.i);
                ++hi;
                ++lo;
                if (queued) {
                    head_j = uni(nq_j);
                    head_i = uni(nq_i);
                } else {
                    head_j = sl.j;
                    head_i = sl.i;
                }
            } else {
                STAMP_INC(7);
                // ---- 2..kRecEvents events (24% of the steps of a uniform instance): still one
                // barrier.  The serial rule (lapjv.cpp:199-205) takes the events in POSITION order;
                // event s swaps the column at its position P_s with the one at hi+s.  When no
                // event sits inside the window [hi, hi+cnt) the swaps are independent: the owner of
                // P_s moves its column to hi+s and adopts the column published for hi+s.  A free
                // column among the events ends the path at the first one in position order.
                bool resolved = false;
                // (with 8+ positions per thread the extra live values of this path spill: n = 16384
                // went from 25 s to 55 s per instance with it -- those sizes keep the ordered replay)
                if (kFastMulti && cnt <= kRecEvents) {
                    // Every lane reads the same records, so the values below are wave-uniform but
                    // deliberately kept in VECTOR registers: ranking them on the scalar unit is a
                    // long dependent chain (and spills SGPRs); only the few results that steer
                    // control flow are moved to scalars.
                    const EventSlot e1 = ctrl->rec[par][1];
                    const EventSlot e2 = ctrl->rec[par][2];
                    const EventSlot e3 = ctrl->rec[par][3];
                    const int4 aw = *reinterpret_cast<const int4 *>(ctrl->a_pub[par]);
                    const int fp = ctrl->free_pos[par];
                    const int kNone = 0x7fffffff;
                    const int P0 = sl.p, P1 = e1.p, P2 = (cnt > 2) ? e2.p : kNone, P3 = (cnt > 3) ? e3.p : kNone;
                    int pmin = (P0 < P1) ? P0 : P1;
                    pmin = (P2 < pmin) ? P2 : pmin;
                    pmin = (P3 < pmin) ? P3 : pmin;
                    unsigned bad = ((unsigned)P0 >= (unsigned)n) | ((unsigned)sl.j >= (unsigned)n) | (sl.i >= n) |
                                   ((unsigned)P1 >= (unsigned)n) | ((unsigned)e1.j >= (unsigned)n) | (e1.i >= n);
                    if (cnt > 2) bad |= ((unsigned)P2 >= (unsigned)n) | ((unsigned)e2.j >= (unsigned)n) | (e2.i >= n);
                    if (cnt > 3) bad |= ((unsigned)P3 >= (unsigned)n) | ((unsigned)e3.j >= (unsigned)n) | (e3.i >= n);
                    bad |= (hi + cnt > n);
                    if (uni((int)bad)) {
                        err = 8;
                        break;
                    }
                    if (uni(fp) != kNone) {
                        // the first free column in position order is the target (its finder recorded
                        // the position); whatever the earlier swaps would change is never read again
                        int tj = -1;
                        tj = (P0 == fp) ? sl.j : tj;
                        tj = (P1 == fp) ? e1.j : tj;
                        tj = (P2 == fp) ? e2.j : tj;
                        tj = (P3 == fp) ? e3.j : tj;
                        tj = uni(tj);
                        if (tj >= 0) {
                            // (free_pos[par] is read by every thread in this phase: it is reset
                            // behind the barrier at the path end, not here)
                            // our event bits of this step's bitmap copy must not survive the path
#pragma unroll
                            for (int r = 0; r < CH; ++r)
                                if ((ev_mask >> r) & 1u) atomicAnd(&evb[par * Wpad + ((b0 + r) >> 5)], ~(1u << ((b0 + r) & 31)));
                            target = tj;
                            break;
                        }
                    } else if (uni(pmin) >= hi + cnt) {
                        // ranks = position order; entry q goes to SCAN slot hi + rank(q)
                        const int r0 = (P1 < P0) + (P2 < P0) + (P3 < P0);
                        const int r1 = (P0 < P1) + (P2 < P1) + (P3 < P1);
                        const int r2 = (P0 < P2) + (P1 < P2) + (P3 < P2);
                        const int r3 = (P0 < P3) + (P1 < P3) + (P2 < P3);
                        int first_j = sl.j, first_i = sl.i, second_j = sl.j, second_i = sl.i;
                        first_j = (r1 == 0) ? e1.j : first_j, first_i = (r1 == 0) ? e1.i : first_i;
                        first_j = (r2 == 0) ? e2.j : first_j, first_i = (r2 == 0) ? e2.i : first_i;
                        first_j = (r3 == 0) ? e3.j : first_j, first_i = (r3 == 0) ? e3.i : first_i;
                        second_j = (r1 == 1) ? e1.j : second_j, second_i = (r1 == 1) ? e1.i : second_i;
                        second_j = (r2 == 1) ? e2.j : second_j, second_i = (r2 == 1) ? e2.i : second_i;
                        second_j = (r3 == 1) ? e3.j : second_j, second_i = (r3 == 1) ? e3.i : second_i;
                        if (ev_mask) {
                            // we found one (or more) of these events: our column goes to the SCAN
                            // list, the column displaced from hi + rank comes to our position
#pragma unroll
                            for (int r = 0; r < CH; ++r) {
                                if ((ev_mask >> r) & 1u) {
                                    const int pk = b0 + r;
                                    int rank = r0, jq = sl.j;
                                    rank = (pk == P1) ? r1 : rank, jq = (pk == P1) ? e1.j : jq;
                                    rank = (pk == P2) ? r2 : rank, jq = (pk == P2) ? e2.j : jq;
                                    rank = (pk == P3) ? r3 : rank, jq = (pk == P3) ? e3.j : jq;
                                    int a = aw.x;
                                    a = (rank == 1) ? aw.y : a;
                                    a = (rank == 2) ? aw.z : a;
                                    a = (rank == 3) ? aw.w : a;
                                    if ((unsigned)a >= (unsigned)n || (pk != P0 && pk != P1 && pk != P2 && pk != P3)) {
                                        ctrl->err = 8;  // (read by all threads at the end of the kernel)
                                    } else {
                                        jr[r] = a;
                                        if constexpr (CACHE_V) {
                                            vr[r] = v[a];
                                            if constexpr (CACHE_Y) yr[r] = y[a];
                                        }
                                        dk[r] = dist[a];
                                        order[pk] = a;
                                        order[hi + rank] = jq;
                                    }
                                    atomicAnd(&evb[par * Wpad + (pk >> 5)], ~(1u << (pk & 31)));
                                }
                            }
                        }
                        app_pos = hi;
                        app_j = uni(first_j);
                        app_i = uni(first_i);
                        app_j1 = uni(second_j);
                        app_i1 = uni(second_i);
                        app_two = true;
                        ring_push(app_i);
                        ring_push(app_i1);
                        if (cnt > 2) {
                            int third_i = sl.i, fourth_i = sl.i;
                            third_i = (r1 == 2) ? e1.i : third_i, third_i = (r2 == 2) ? e2.i : third_i;
                            third_i = (r3 == 2) ? e3.i : third_i;
                            fourth_i = (r1 == 3) ? e1.i : fourth_i, fourth_i = (r2 == 3) ? e2.i : fourth_i;
                            fourth_i = (r3 == 3) ? e3.i : fourth_i;
                            ring_push(uni(third_i));
                            if (cnt > 3) ring_push(uni(fourth_i));
                        }
                        hi += cnt;
                        ++lo;
                        if (queued) {
                            head_j = uni(nq_j);
                            head_i = uni(nq_i);
                        } else {
                            head_j = app_j;
                            head_i = app_i;
                        }
                        resolved = true;
                    }
                }
                if (resolved) {
                    STAMPR(tr4m);
                    STAMP_ADD(4, tr4m, tr3);
                    continue;
                }
                const int hi_before = hi;
                if (bc.wave == 0) replay_scan(hi, par);
                __syncthreads();
                hi = uni(ctrl->hi);
                target = uni(ctrl->target);
                if (target >= 0) break;
                ring_push_scan(hi_before - 1, hi);
                ++lo;
                if (queued) {
                    head_j = uni(nq_j);
                    head_i = uni(nq_i);
                } else {
                    head_j = uni(order[lo]);
                    head_i = uni(y[head_j]);
                }
                // The replay swapped every event position with a slot of [old hi, new hi): the only
                // TODO positions whose column changed are this step's event positions -- and a
                // thread knows its own (ev_mask).  Rebinding all CH positions here cost four
                // gathers per position in 23% of the steps: harmless with the state in LDS,
                // ruinous with the state in global memory (n > 4,427).
                if (ev_mask) {
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
                        const int k = b0 + r;
                        if (((ev_mask >> r) & 1u) && k >= hi && k < n) {
                            const int j = order[k];
                            jr[r] = j;
                            if constexpr (CACHE_V) {
                                vr[r] = v[j];
                                if constexpr (CACHE_Y) yr[r] = y[j];
                            }
                            dk[r] = dist[j];
                        }
                    }
                }
            }
            STAMPR(tr4);
            STAMP_ADD(4, tr4, tr3);
        }
        ctrl_seen0 = seen0;
        ctrl_seen1 = seen1;
        ctrl_find_seq = find_seq;
        // Path exit made safe by construction: every wave has left the search loop (and finished
        // whatever it still read there) before any wave updates v[] below or thread 0 rewrites
        // y[] / x[] in the backtrack.  ~25 ns per path (happens-before table).
        __syncthreads();
        if (bc.tid == 0) {
            ctrl->free_pos[0] = 0x7fffffff;
            ctrl->free_pos[1] = 0x7fffffff;
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
    __device__ __forceinline__ void augment_all(int f_first, int n_free)
    {
        for (int f = f_first; f < n_free && !err; ++f) {
            const int start = uni(fr[f]);
            if ((unsigned)start >= (unsigned)n) {
                err = 2;
                break;
            }
            if (f + 1 < n_free) ring_push(uni(fr[f + 1]));  // the helper fetches it while this path runs
            const int target = find_path(start);
            if (err) break;
            if (bc.tid == 0) {
                // a corrupted chain must end in a return code, never in an out-of-range access
                int j = target, i = -1, hops = 0;
                bool ok = true;
                while (i != start && hops <= n) {
                    if ((unsigned)j >= (unsigned)n) {
                        ok = false;
                        break;
                    }
                    i = pred[j];
                    if ((unsigned)i >= (unsigned)n) {
                        ok = false;
                        break;
                    }
                    y[j] = i;
                    const int prev = x[i];
                    x[i] = j;
                    j = prev;
                    ++hops;
                }
                if (!ok || i != start) ctrl->err = 3;
            }
            __syncthreads();
            if (uni(ctrl->err)) {  // uniform: read after the barrier
                err = 3;
                break;
            }
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
        int jc[CH];
#pragma unroll
        for (int r = 0; r < CH; ++r) jc[r] = (b0 + r < n) ? b0 + r : n - 1;
        constexpr int RU = (CH <= 2) ? 8 : ((CH <= 4) ? 4 : 2);  // rows in flight per thread
        int i = 0;
        for (; i + RU <= n; i += RU) {
            double c[RU][CH];
#pragma unroll
            for (int q = 0; q < RU; ++q) {
                const double *row = C + (size_t)(i + q) * n;
#pragma unroll
                for (int r = 0; r < CH; ++r) c[q][r] = row[jc[r]];
            }
#pragma unroll
            for (int q = 0; q < RU; ++q) {
#pragma unroll
                for (int r = 0; r < CH; ++r) pin(c[q][r]);
            }
#pragma unroll
            for (int q = 0; q < RU; ++q) {
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    if (c[q][r] < vm[r]) {
                        vm[r] = c[q][r];
                        ya[r] = i + q;
                    }
                }
            }
        }
        for (; i < n; ++i) {
            const double *row = C + (size_t)i * n;
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const double c = row[jc[r]];
                if (c < vm[r]) {
                    vm[r] = c;
                    ya[r] = i;
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
                double ct[CH];
#pragma unroll
                for (int r = 0; r < CH; ++r) ct[r] = row[jc[r]];
#pragma unroll
                for (int r = 0; r < CH; ++r) pin(ct[r]);
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    const int j2 = b0 + r;
                    const double c = ct[r] - v[jc[r]];
                    if (j2 < n && j2 != xi && c < m) m = c;
                }
                m = bc.min_f64(m);
                if (xi >= b0 && xi < b0 + CH) v[xi] -= m;  // owner thread only
                transfer_rows++;
            }
        }
        __syncthreads();
        return nf;
    }

    // ---- candidate lists of the augmenting row reduction
    // 99 % of the ARR iterations of a uniform instance continue with the row the last one displaced
    // (tools/micro/arr_chain_stats.c): the iterations form one dependent chain of ~1.4e5 row scans at
    // n = 2048, each a 16-KiB row from beyond the L2 plus a workgroup reduction (~2 us).  The chain
    // itself cannot be shortened, but an iteration can: v[] only ever DECREASES during the row reduction
    // (lapjv.cpp:117, v2 >= v1), so c[i][j] - v[j] only grows.  For row i keep the kArrListD smallest
    // c - v of each of 64 column classes (j mod 64) as candidates, and tau_i = the smallest value left
    // out (the minimum over the classes of their (kArrListD+1)-th smallest).  Whenever the second
    // smallest candidate is < tau_i at today's v, no column outside the list can be among the two
    // smallest (it was >= tau_i when the list was built and has only grown), so one wave finds
    // (v1, j1, v2, j2) from 128 entries without a workgroup barrier; ties at equal values are
    // inside the list and resolved by column index exactly as the serial scan does.  Otherwise the whole
    // workgroup scans the row as before and wave 0 rebuilds that row's list at the current v.
    // Measured on the CPU (tools/micro/arr_lists_sim.py, n = 1024): uniform 0.01 %, sparse
    // 0.4 %, tie 0.01 %, noisy_linear 1.4 % of the iterations take the full scan.
    static constexpr int kArrListD = 2;
    static constexpr int kArrListLen = kArrListD * kWave;
    static_assert(kArrListLen == kArrListEntries, "workspace layout");

    // One wave builds the list of row i at the current v (entries: raw cost + column, -1 = none).
    __device__ __forceinline__ void arr_build_list(int i, double *lval, int *lcol, double *ltau)
    {
        const int lane = bc.lane;
        const double *row = C + (size_t)i * n;
        double a0 = pos_inf(), a1 = pos_inf(), a2 = pos_inf(), r0 = 0.0, r1 = 0.0;
        int j0 = -1, j1 = -1;
        bool nan0 = false;
        constexpr int U = 4;
        for (int jb = 0; jb < n; jb += U * kWave) {
            double c[U], vv[U];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int j = jb + q * kWave + lane;
                c[q] = row[(j < n) ? j : n - 1];
                vv[q] = v[(j < n) ? j : n - 1];
            }
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int j = jb + q * kWave + lane;
                const double xq = c[q] - vv[q];
                if (j == 0) nan0 = !(xq == xq);
                if (j < n && xq < a2) {  // strict: among equal values of a class the lower column stays ahead
                    if (xq < a1) {
                        a2 = a1;
                        if (xq < a0) {
                            a1 = a0, r1 = r0, j1 = j0;
                            a0 = xq, r0 = c[q], j0 = j;
                        } else {
                            a1 = xq, r1 = c[q], j1 = j;
                        }
                    } else {
                        a2 = xq;
                    }
                }
            }
        }
        double tau = wave_min(a2);
        // a NaN in column 0 poisons the serial scan's running minimum (lapjv.cpp:87-104): such a row
        // always takes the full scan, which reproduces that
        if (__ballot(nan0)) tau = -pos_inf();
        // the two entries of a lane sit side by side: one 16-byte and one 8-byte load per lane and iteration
        const size_t o = (size_t)i * kArrListLen + 2 * lane;
        *reinterpret_cast<double2 *>(lval + o) = make_double2(r0, r1);
        *reinterpret_cast<int2 *>(lcol + o) = make_int2(j0, j1);
        if (lane == 0) ltau[i] = tau;
    }

    // lapjv.cpp:84-147: one iteration of the augmenting row reduction by the whole workgroup (two-minimum
    // scan of the row, dual update, reassignment) for the sweep with candidate lists -- the body of
    // cold_arr_sweep_plain()'s loop.  Returns false on the iteration guard.
    __device__ __forceinline__ bool arr_scan_row(unsigned &current, unsigned &rr, int &new_free, int &fwd,
                                                 int &free_i_out)
    {
        const int b0 = base();
        const unsigned un = (unsigned)n;
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
        Arr2 t = arr2_empty();
        if (bc.tid == 0) arr2_push(t, kLarge, kSentinelIdx, 0.0, -1);
#pragma unroll
        for (int r = 0; r < CH; ++r) cs[r] = row[(b0 + r < n) ? b0 + r : n - 1];
#pragma unroll
        for (int r = 0; r < CH; ++r) pin(cs[r]);
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int j = b0 + r;
            const int jc = (j < n) ? j : n - 1;
            const double vj = v[jc];
            const int yj = y[jc];  // owner-only reads: v[j], y[j] are written by their owner
            const double cv = cs[r] - vj;
            cs[r] = (j < n) ? cv : pos_inf();
            if (j == 0) c0 = cv;
            if (j < n && (j == 0 || cv < kLarge)) arr2_push(t, cv, j, vj, yj);
        }
        t = bc.arr2(t, &c0);
        int i0, i0_second;
        double vj1;
        if (c0 <= kLarge) {
            // everything the serial code reads next came along as payload: no second barrier
            v1 = t.a1;
            j1 = t.i1;
            v2 = t.a2;
            j2 = (t.i2 == kSentinelIdx) ? -1 : t.i2;
            i0 = t.y1;
            i0_second = (j2 >= 0) ? t.y2 : -1;
            vj1 = t.vj1;
        } else {
            // column 0 starts above the sentinel: nothing is accepted before the first
            // column with c < LARGE; from there on it is a plain two-minimum scan that
            // still holds (c0, 0) as a candidate.  Rare: keep the simple two-barrier form.
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
            // uniform reads of entries owned by other threads, then a barrier, then the
            // owners' writes: nobody may see this iteration's update while still reading.
            i0 = y[j1];
            i0_second = (j2 >= 0) ? y[j2] : -1;
            vj1 = v[j1];
            __syncthreads();
        }
        arr_iters++;
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
            return false;
        }
        free_i_out = free_i;
        return true;
    }

    // lapjv.cpp:76-149.  One augmenting-row-reduction sweep over fr[0..n_free): every iteration scans its row.
    // (Kept word for word as in round 2, beside arr_scan_row() below: the seeded kernel's register allocation
    // follows the text of this function -- built from the shared helper it cost K4 3 %.)
    __device__ __forceinline__ int cold_arr_sweep_plain(int n_free)
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
            Arr2 t = arr2_empty();
            if (bc.tid == 0) arr2_push(t, kLarge, kSentinelIdx, 0.0, -1);
#pragma unroll
            for (int r = 0; r < CH; ++r) cs[r] = row[(b0 + r < n) ? b0 + r : n - 1];
#pragma unroll
            for (int r = 0; r < CH; ++r) pin(cs[r]);
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const int j = b0 + r;
                const int jc = (j < n) ? j : n - 1;
                const double vj = v[jc];
                const int yj = y[jc];  // owner-only reads: v[j], y[j] are written by their owner
                const double cv = cs[r] - vj;
                cs[r] = (j < n) ? cv : pos_inf();
                if (j == 0) c0 = cv;
                if (j < n && (j == 0 || cv < kLarge)) arr2_push(t, cv, j, vj, yj);
            }
            t = bc.arr2(t, &c0);
            int i0, i0_second;
            double vj1;
            if (c0 <= kLarge) {
                // everything the serial code reads next came along as payload: no second barrier
                v1 = t.a1;
                j1 = t.i1;
                v2 = t.a2;
                j2 = (t.i2 == kSentinelIdx) ? -1 : t.i2;
                i0 = t.y1;
                i0_second = (j2 >= 0) ? t.y2 : -1;
                vj1 = t.vj1;
            } else {
                // column 0 starts above the sentinel: nothing is accepted before the first
                // column with c < LARGE; from there on it is a plain two-minimum scan that
                // still holds (c0, 0) as a candidate.  Rare: keep the simple two-barrier form.
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
                // uniform reads of entries owned by other threads, then a barrier, then the
                // owners' writes: nobody may see this iteration's update while still reading.
                i0 = y[j1];
                i0_second = (j2 >= 0) ? y[j2] : -1;
                vj1 = v[j1];
                __syncthreads();
            }
            arr_iters++;
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

    // wave minimum of values that are never NaN (the list keys below): one v_min_f64 per step instead of a
    // compare and two selects (as inline asm it would also lose the canonicalising v_max in front, but the
    // hazard recogniser does not see an asm's result feeding the next DPP move)
    __device__ __forceinline__ static double wave_min_nn(double x)
    {
#define LAPWARM_STEP(C, M) x = __builtin_fmin(x, dpp_move<C, M>(pos_inf(), x));
        LAPWARM_DPP_REDUCE(LAPWARM_STEP)
#undef LAPWARM_STEP
        return readlane_f64(x, kWave - 1);
    }

    // The same sweep with the candidate lists in front of the row scans.
    template <bool LISTS>
    __device__ __forceinline__ int cold_arr_sweep(int n_free, double *lval, int *lcol, double *ltau)
    {
        unsigned current = 0, rr = 0;
        int new_free = 0;
        int fwd = -1;
        const unsigned un = (unsigned)n;
        bool lists = LISTS && lval != nullptr;
        int n_fast = 0, n_slow = 0;
        ArrState *as = reinterpret_cast<ArrState *>(&ctrl->rec[0][0]);
        if (LISTS && lists) {
            // every row's list at the v the sweep starts from (the second sweep finds them in place:
            // v has only decreased since)
            if (ltau[0] != ltau[0]) {  // (uniform; NaN marks "not built yet", set by the caller)
                __syncthreads();
                for (int i = bc.wave; i < n; i += bc.nwaves) arr_build_list(i, lval, lcol, ltau);
                fence_if_global();
                __threadfence_block();
                __syncthreads();
            }
        }
        while (true) {
            if (LISTS && lists) {
                if (bc.wave == 0) {
                    const int lane = bc.lane;
                    int bad = 0;
                    // list of the row the previous iteration expected to displace, requested as soon as
                    // that row was known (before the second minimum and the updates)
                    int cand = -1, pca = -1, pcb = -1;
                    double pra = 0.0, prb = 0.0, ptau = 0.0;
                    // The loop state in SCALAR registers (readfirstlane): the compiler cannot know that a value
                    // computed by every lane of this wave alike is uniform, and would steer every branch of
                    // the iteration through the exec mask (~60 of its ~330 instructions).
                    unsigned cur_s = (unsigned)uni((int)current), rr_s = (unsigned)uni((int)rr);
                    int nf_s = uni(new_free), fwd_s = uni(fwd), it_s = uni(arr_iters), fast_s = uni(n_fast);
                    const unsigned nfree_s = (unsigned)uni(n_free), un_s = (unsigned)uni(n);
                    while (cur_s < nfree_s) {
                        const int free_i = (fwd_s >= 0) ? fwd_s : uni(fr[cur_s]);
                        if (free_i != cand) {
                            const size_t o = (size_t)free_i * kArrListLen + 2 * lane;
                            const double2 rv = *reinterpret_cast<const double2 *>(lval + o);
                            const int2 cv = *reinterpret_cast<const int2 *>(lcol + o);
                            pca = cv.x, pcb = cv.y;
                            pra = rv.x, prb = rv.y;
                            ptau = ltau[free_i];
                        }
                        const int ca = pca, cb = pcb;
                        const double ra = pra, rb = prb;
                        const double tau = ptau;
                        cand = -1;
                        const int qa = (ca >= 0) ? ca : 0, qb = (cb >= 0) ? cb : 0;
                        const double va = v[qa], vb = v[qb];
                        const int ya_ = y[qa], yb_ = y[qb];
                        const double xa = (ca >= 0) ? ra - va : pos_inf();
                        const double xb = (cb >= 0) ? rb - vb : pos_inf();
                        // this lane's two candidates in (value, column) order; NaN sorts last
                        const bool swap = pair_less(xb, cb, xa, ca) || (!(xa == xa) && xb == xb);
                        const double l1 = swap ? xb : xa, l2 = swap ? xa : xb;
                        const int i1 = swap ? cb : ca, i2 = swap ? ca : cb;
                        const double w1 = swap ? vb : va;
                        const int y1 = swap ? yb_ : ya_, y2 = swap ? ya_ : yb_;
                        const double k1 = (l1 == l1) ? l1 : pos_inf(), k2 = (l2 == l2) ? l2 : pos_inf();
                        // smallest (value, column) pair: the value by a DPP reduction; the column needs a
                        // second reduction only when several lanes hold that value
                        const double m1 = wave_min_nn(k1);
                        unsigned long long b1 = __ballot(k1 == m1 && i1 >= 0);
                        int g1 = 0x7fffffff;
                        if (__popcll(b1) == 1) {
                            g1 = __builtin_amdgcn_readlane(i1, __builtin_ctzll(b1));
                        } else if (b1) {
                            g1 = wave_min_i32((k1 == m1 && i1 >= 0) ? i1 : 0x7fffffff);
                            b1 = __ballot(k1 == m1 && i1 == g1);
                        }
                        const bool win = (k1 == m1) && (i1 == g1);
                        if (b1) {
                            const int i0s = __builtin_amdgcn_readlane(y1, __builtin_ctzll(b1));
                            if ((unsigned)i0s < un_s) {
                                cand = i0s;
                                const size_t o = (size_t)i0s * kArrListLen + 2 * lane;
                                const double2 rv = *reinterpret_cast<const double2 *>(lval + o);
                                const int2 cv = *reinterpret_cast<const int2 *>(lcol + o);
                                pca = cv.x, pcb = cv.y;
                                pra = rv.x, prb = rv.y;
                                ptau = ltau[i0s];
                            }
                        }
                        const double s2 = win ? k2 : k1;
                        const int si = win ? i2 : i1, sy = win ? y2 : y1;
                        const double m2 = wave_min_nn(s2);
                        unsigned long long b2 = __ballot(s2 == m2 && si >= 0);
                        int g2 = 0x7fffffff;
                        if (__popcll(b2) == 1) {
                            g2 = __builtin_amdgcn_readlane(si, __builtin_ctzll(b2));
                        } else if (b2) {
                            g2 = wave_min_i32((s2 == m2 && si >= 0) ? si : 0x7fffffff);
                            b2 = __ballot(s2 == m2 && si == g2);
                        }
                        const bool ok = (m2 < tau) && (m2 < kLarge) && (m1 > -kLarge) && g1 != 0x7fffffff && g2 != 0x7fffffff;
                        if (__ballot(!ok)) break;  // (uniform) this row takes the full scan; nothing was changed
                        const int ln1 = __builtin_ctzll(b1), ln2 = __builtin_ctzll(b2);
                        const double vj1 = readlane_f64(w1, ln1);
                        int i0 = __builtin_amdgcn_readlane(y1, ln1);
                        const int i0_second = __builtin_amdgcn_readlane(sy, ln2);
                        int j1 = g1;
                        const int j2 = g2;
                        rr_s++;
                        cur_s++;
                        fwd_s = -1;
                        it_s++;
                        fast_s++;
                        const double v1_new = vj1 - (m2 - m1);
                        const bool lowers = __ballot(v1_new < vj1) != 0ull;  // (a scalar condition)
                        // the decisions first (uniform), then ONE masked block with every store of the iteration
                        bool store_v = false;
                        int fr_at = -1;
                        if (rr_s < cur_s * un_s) {
                            if (lowers) {
                                store_v = true;
                            } else if (i0 >= 0) {
                                j1 = j2;
                                i0 = i0_second;
                            }
                            if (i0 >= 0) {
                                if (lowers) {
                                    --cur_s;
                                    fwd_s = i0;
                                    fr_at = (int)cur_s;
                                } else {
                                    fr_at = nf_s;
                                    ++nf_s;
                                }
                            }
                        } else if (i0 >= 0) {
                            fr_at = nf_s;
                            ++nf_s;
                        }
                        if (lane == 0) {
                            if (store_v) v[g1] = v1_new;
                            if (fr_at >= 0) fr[fr_at] = i0;
                            x[free_i] = j1;
                            y[j1] = free_i;
                        }
                        fence_if_global();  // (state in global memory: the next iteration's lanes read these)
                        if (it_s > (1 << 26)) {
                            bad = 4;
                            break;
                        }
                    }
                    current = cur_s;
                    rr = rr_s;
                    new_free = nf_s;
                    fwd = fwd_s;
                    arr_iters = it_s;
                    n_fast = fast_s;
                    if (lane == 0) {
                        as->arr_current = current;
                        as->arr_rr = rr;
                        as->arr_new_free = new_free;
                        as->arr_fwd = fwd;
                        as->arr_iters = arr_iters;
                        as->arr_fast = n_fast;
                        as->arr_err = bad;
                    }
                }
                fence_if_global();
                __syncthreads();
                current = as->arr_current;
                rr = as->arr_rr;
                new_free = as->arr_new_free;
                fwd = as->arr_fwd;
                arr_iters = as->arr_iters;
                n_fast = as->arr_fast;
                if (as->arr_err) {
                    err = as->arr_err;
                    break;
                }
            }
            if (current >= (unsigned)n_free) break;
            int free_i = 0;
            if (!arr_scan_row(current, rr, new_free, fwd, free_i)) break;
            if (LISTS && lists) {
                // the owners' writes of this iteration, then the list of this row at today's v by wave 0
                // (it is the one that reads it next); a family whose lists keep going stale faster than
                // they pay (more full scans than list iterations after the first 256) stops using them
                ++n_slow;
                fence_if_global();
                __syncthreads();
                if (n_slow >= 256 && n_fast < n_slow) {
                    lists = false;
                } else if (bc.wave == 0) {
                    arr_build_list(free_i, lval, lcol, ltau);
                    fence_if_global();
                }
            }
        }
        __syncthreads();
        if (bc.tid == 0) ctrl->first_fire += n_fast;  // (stats; the field is the micro-ARR's, which a cold solve never runs)
        return new_free;
    }

    // lapjv.cpp:323-346
    // Column reduction + the two ARR sweeps; returns the rows still free (the caller runs the
    // shortest-path phase, which is shared with the seeded branch -- one call site, one copy).
    template <bool LISTS>
    __device__ __forceinline__ int cold_prepare(double *lval, int *lcol, double *ltau)
    {
        int nf = cold_column_reduction();
        if (LISTS && bc.tid == 0) ctrl->first_fire = 0;  // (counts the list iterations: stats slot 27)
        if (LISTS && lval && nf > 0) {
            if (bc.tid == 0) ltau[0] = __longlong_as_double(0x7ff8000000000000LL);  // "lists not built yet"
            fence_if_global();
            __threadfence_block();
            __syncthreads();
        }
        for (int sweep = 0; nf > 0 && sweep < 2 && !err; ++sweep) {
            if constexpr (LISTS)
                nf = cold_arr_sweep<true>(nf, lval, lcol, ltau);
            else
                nf = cold_arr_sweep_plain(nf);
        }
        return nf;
    }

    // ------------------------------------------------------------------ seeded phases
    // lapjv_seeded.cpp:79-102: first tight column not yet used, rows in ascending order.
    // Wave 0 walks the per-row tight bitmaps written by the prelude kernel.  The rows are taken in order (the
    // rule is serial) but their bitmaps do not depend on each other: the words of several rows are in flight at
    // a time (lane l holds words l, l + 64, ... of a row: WPL per lane) and the used-column bitmap lives in
    // registers with the same layout.  The row-at-a-time form this replaces paid two dependent global loads
    // (~2 us) per row: most of a K2 solve (solver 1.37 -> 1.20 ms).  The prelude writes every word of every row,
    // rows without a tight edge included.
    template <int WPL>
    __device__ __forceinline__ void greedy_rows(const uint32_t *tight_bits)
    {
        constexpr int RB = (WPL >= 8) ? 2 : 8 / WPL;  // rows in flight
        const int lane = bc.lane;
        int nf = 0;
        uint32_t myused[WPL];
#pragma unroll
        for (int k = 0; k < WPL; ++k) myused[k] = 0;
        for (int i0 = 0; i0 < n; i0 += RB) {
            uint32_t w[RB][WPL];
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int row = (i0 + q < n) ? i0 + q : n - 1;
#pragma unroll
                for (int k = 0; k < WPL; ++k) {
                    const int idx = k * kWave + lane;
                    w[q][k] = (idx < W) ? tight_bits[(size_t)row * W + idx] : 0u;
                }
            }
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int i = i0 + q;
                if (i < n) {
                    int found = -1;
#pragma unroll
                    for (int k = 0; k < WPL; ++k) {
                        if (found < 0) {  // (uniform)
                            const uint32_t word = w[q][k] & ~myused[k];
                            const unsigned long long mask = __ballot(word != 0);
                            if (mask) {
                                const int l = __builtin_ctzll(mask);
                                const uint32_t wv = (uint32_t)__builtin_amdgcn_readlane((int)word, l);
                                found = ((k * kWave + l) << 5) + __builtin_ctz(wv);
                                if (lane == l) myused[k] |= 1u << (found & 31);
                            }
                        }
                    }
                    if (lane == 0) {
                        if (found >= 0) {
                            x[i] = found;
                            y[found] = i;
                        } else {
                            fr[nf] = i;
                        }
                    }
                    if (found < 0) ++nf;
                }
            }
        }
        if (lane == 0) ctrl->nfree = nf;
    }
    __device__ __forceinline__ void greedy_wave0(const uint32_t *tight_bits, const int *)
    {
        constexpr int kMaxWords = (CH * TB + 31) / 32;  // the largest bitmap this instantiation can meet
        if (kMaxWords <= kWave || W <= kWave)
            greedy_rows<1>(tight_bits);
        else if (kMaxWords <= 2 * kWave || W <= 2 * kWave)
            greedy_rows<2>(tight_bits);
        else if (kMaxWords <= 4 * kWave || W <= 4 * kWave)
            greedy_rows<4>(tight_bits);
        else
            greedy_rows<8>(tight_bits);
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

// TB = compile-time bound on the workgroup size: 1024-thread workgroups cap a thread at 128 VGPRs,
// the 256-thread variant (one wave per SIMD) gets the whole register file.
// LISTS = true: the PREPARATION of a cold solve with per-row candidate lists in the augmenting row reduction
// (cold_arr_sweep<true>) and nothing else: it is only ever launched as phase 1 (column reduction + row
// reduction, state handed over through the global arrays) by the cold entry points (lap.lapjv), and the
// shortest-path phase follows in the plain instantiation (phase 2).  Three arrangements were measured first:
//  * list code compiled into the one kernel: its register pressure is paid by the shortest-path loops of
//    EVERY instance (SGPR spills 170 -> 213; K3 76.5 -> 79 ms, K4 288 -> 310 ms);
//  * a seeded-only kernel with the full kernel launched behind it for the quality-gate fallbacks: those then
//    run AFTER the seeded instances of the batch instead of beside them (K3 76.6 -> 81.7 ms);
//  * this instantiation running the shortest-path phase itself: integer-cost instances (int100, n = 1536 /
//    2048, 512 threads x 4 columns) came out of a CORRECT row reduction (same state handed to the cooperative
//    kernel: exact, six runs of six) and then failed in the shortest-path loops, differently from run to run
//    (ret -106 / -103 / wrong x) -- the same source that is exact in the plain instantiation.  Not explained
//    (the instantiation spills 87 VGPRs there against 30); avoided by not compiling those loops in here.
// The seeded launches keep the plain row scans for their fallbacks: the code, and the timing, of round 2.
template <int CH, int LDSL, int TB, bool LISTS>
__global__ void __launch_bounds__(TB) jv_instance_kernel(SolverParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int n = p.n;
    if (blockIdx.x >= (unsigned)p.batch) {
        // ---- helper workgroup of instance blockIdx.x - batch (same XCD when batch % 8 == 0: the
        // dispatcher deals workgroups to the 8 XCDs round robin): wave 0 pulls the announced rows
        // towards the shared L2 with LDS-DMA requests into a dummy area; nothing reads them here.
        // Exits on the solver's done flag, or after 0.5 s whatever happens.
        // grid layout: [0, batch) solvers; then regions of P = batch rounded up to 8 workgroups, one
        // per helper copy, so that helper and solver of an instance have the same index modulo 8
        if (threadIdx.x >= kWave) return;
        const int P = (p.batch + 7) & ~7;
        if ((int)blockIdx.x < P) return;  // padding between the solvers and the first helper region
        const int hsel = ((int)blockIdx.x - P) / P;  // which of the p.helper helpers of the instance
        const int hb = ((int)blockIdx.x - P) % P;
        if (hb >= p.batch) return;
        int *ring = p.pf_ring + (size_t)hb * kRingInts;
        const double *Cb = p.C + (size_t)hb * n * n;
        const int lane = threadIdx.x;
        const unsigned dummy = lds_address(smem);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
        int seen = 0;
        const int pieces = (n * 8 + 1023) / 1024;
        while (true) {
            const int w = __hip_atomic_load(&ring[2 + (seen & (kRingSlots - 1))], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int expect = ((seen >> 6) + 1) & 0xffff;
            const int got = (w >> 16) & 0xffff;
            if (got == expect) {
                const int row = w & 0xffff;
                if (row < n) {
                    const double *r = Cb + (size_t)row * n;
                    for (int k = hsel; k < pieces; k += p.helper) {
                        int col = k * 128 + lane * 2;
                        col = (col < n - 2) ? col : n - 2;
                        dma_request16(r + col, dummy);
                    }
                }
                ++seen;
                continue;
            }
            const int ahead = (got - expect) & 0xffff;
            if (got != 0 && ahead != 0 && ahead < 0x8000) {  // the solver lapped us: skip a ring's worth
                seen += kRingSlots * ahead;
                continue;
            }
            if (__hip_atomic_load(&ring[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > (n <= 4096 ? 50000000ull : 6000000000ull)) break;  // 0.5 s / 60 s
            __builtin_amdgcn_s_sleep(1);
        }
        dma_wait<0>();
        return;
    }
    const int b = blockIdx.x;
    const int W = (n + 31) >> 5;
    const int Wpad = (W + 1) & ~1;

    Solver<CH, LDSL, TB> s;
    unsigned char *cur = smem;
    s.slots = smem;
    s.slot_bytes = 0;
    s.nslots = 0;
    if constexpr (LDSL == 8) {
        s.slot_bytes = (int)blockDim.x * CH * (int)sizeof(double);
        s.nslots = solver_row_slots(n, CH);
        cur += (size_t)s.nslots * (size_t)s.slot_bytes + 16;
    }
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
    s.evb = reinterpret_cast<uint32_t *>(cur);
    cur += sizeof(uint32_t) * Wpad * 2;
    s.Wpad = Wpad;
    if constexpr (LDSL > 0 && LDSL != 8) {
        s.evl = reinterpret_cast<int *>(cur);
        cur += sizeof(int) * n;
        s.tmpcol = reinterpret_cast<int *>(cur);
        cur += sizeof(int) * (n + 2);
    } else {
        s.evl = p.g_evl + (size_t)b * n;
        s.tmpcol = p.g_tmpcol + (size_t)b * (n + 2);
    }
    if constexpr (LDSL > 0 && LDSL != 8) {
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
    }
    s.bc.init(ex);
    s.C = p.C + (size_t)b * n * n;
    s.n = n;
    s.W = W;
    s.ring = (p.helper && p.pf_ring) ? p.pf_ring + (size_t)b * kRingInts : nullptr;
    s.ring_count = 0;
    s.scan_elems = s.init_elems = s.colred_elems = 0;
    s.paths = s.finds = s.scan_steps = s.arr_iters = s.transfer_rows = s.arr_fired = 0;
    s.step_id = 1;
    s.ctrl_seen0 = s.ctrl_seen1 = 0;
    s.ctrl_find_seq = 0;
#ifdef LAPWARM_STAMPS
    for (int q = 0; q < 16; ++q) s.stamps[q] = 0;
#endif
    s.err = 0;

    const int tid = s.bc.tid;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    unsigned long long t_serial = t_start;
    const int flags = p.inst_flags ? p.inst_flags[b] : 0;
    if (p.mode == kModeSeeded && (flags & kFlagInfeasible)) {
        if (tid == 0) {
            if (p.helper && p.pf_ring)
                __hip_atomic_store(&p.pf_ring[(size_t)b * kRingInts], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.phase == 3) return;
            if (p.phase == 1) {  // nothing for the cooperative kernel to do; phase 2 comes through here again
                for (int q = 0; q < kHandInts; ++q) p.hand[(size_t)b * kHandInts + q] = 0;
                return;
            }
            p.ret[b] = -3;
            if (p.stats) {
                for (int q = 0; q < kStatsPerInstance; ++q) p.stats[(size_t)b * kStatsPerInstance + q] = 0;
            }
        }
        return;
    }

    if (tid == 0) {
        s.ctrl->tie_find = 0;
        s.ctrl->ev_total[0] = 0;
        s.ctrl->ev_total[1] = 0;
        s.ctrl->free_pos[0] = 0x7fffffff;
        s.ctrl->free_pos[1] = 0x7fffffff;
        s.ctrl->err = 0;
        s.ctrl->nfree = 0;
        s.ctrl->hi = 0;
        s.ctrl->target = -1;
        s.ctrl->head_j = 0;
        s.ctrl->head_i = 0;
    }
    for (int w = tid; w < Wpad; w += blockDim.x) {
        s.evt[w] = 0;
        s.sbits[w] = 0;
        s.used[w] = 0;
        s.evb[w] = 0;
        s.evb[Wpad + w] = 0;
    }
    long long branch = kBranchCold;
    long long tight_total = 0;
    long long free_after_greedy = 0;
    int nf = 0;
    if (p.phase == 3) {
        // ---- between two launches of the cooperative kernel: its mailbox starts from zero again (its
        // round tags restart with every launch), and if it stopped at a path it does not handle, THAT
        // ONE path is searched here; the cooperative kernel carries on behind it
        const size_t ng = (size_t)p.mail_granules;
        for (size_t q = tid; q < ng; q += blockDim.x) p.mail[(size_t)b * ng + q] = 0ull;
        const int *hand = p.hand + (size_t)b * kHandInts;
        if (hand[3] == 0 || hand[2] != 0 || hand[4] != 0 || hand[1] >= hand[0]) return;  // nothing pending (uniform)
    }
    if (!LISTS && (p.phase == 2 || p.phase == 3)) {
        // ---- resume behind the cooperative kernel: x, y, v and the free rows come back from the
        // global state arrays, the rows hand[1] .. hand[0] are still to be augmented
        const size_t o = (size_t)b * n;
        int *hand = p.hand + (size_t)b * kHandInts;
        nf = hand[0];
        const int f0 = hand[1];
        const int f_end = (p.phase == 3) ? ((f0 + 1 < nf) ? f0 + 1 : nf) : nf;
        for (int j = tid; j < n; j += blockDim.x) {
            if (s.x != p.g_x + o) s.x[j] = p.g_x[o + j];
            if (s.y != p.g_y + o) s.y[j] = p.g_y[o + j];
            if (s.v != p.g_v + o) s.v[j] = p.g_v[o + j];
            if (s.fr != p.g_fr + o && j < nf) s.fr[j] = p.g_fr[o + j];
        }
        branch = hand[5];
        tight_total = hand[6];
        free_after_greedy = hand[7];
        s.arr_fired = hand[8];
        s.transfer_rows = hand[9];
        s.arr_iters = hand[10];
        s.colred_elems = ((long long)hand[12] << 32) | (unsigned)hand[11];
        s.err = hand[13];
        if (hand[2] | hand[4]) s.err = 30 + ((hand[2] | hand[4]) & 31);  // the cooperative kernel failed
        __syncthreads();
        if (!s.err && f0 >= 0 && f0 < f_end) s.augment_all(f0, f_end);
        if (p.phase == 3) {
            // hand the state back: x, y, v to the global arrays (a no-op where they live there anyway),
            // the path counters into the cooperative kernel's totals, the resume index one further
            __syncthreads();
            for (int j = tid; j < n; j += blockDim.x) {
                if (s.x != p.g_x + o) p.g_x[o + j] = s.x[j];
                if (s.y != p.g_y + o) p.g_y[o + j] = s.y[j];
                if (s.v != p.g_v + o) p.g_v[o + j] = s.v[j];
            }
            if (tid == 0) {
                const int e3 = s.err | s.ctrl->err;
                hand[1] = f_end;
                hand[3] = 0;
                if (e3) hand[13] = e3;
                if (p.cstats) {
                    long long *cs = p.cstats + (size_t)b * kCoopStats;
                    cs[0] += s.paths;
                    cs[1] += s.finds;
                    cs[2] += s.scan_steps;
                    cs[3] += s.scan_elems;
                    cs[4] += s.init_elems;
                    cs[15] += 1;  // paths searched outside the cooperative kernel, one launch each
                }
            }
            return;
        }
        if (p.cstats) {
            const long long *cs = p.cstats + (size_t)b * kCoopStats;
            s.paths += (int)cs[0];
            s.finds += (int)cs[1];
            s.scan_steps += (int)cs[2];
            s.scan_elems += cs[3];
            s.init_elems += cs[4];
        }
    }
    int tight_local = 0;
    if (p.phase < 2) {
        for (int j = tid; j < n; j += blockDim.x) {
            s.x[j] = -1;
            s.y[j] = -1;
            if (p.mode == kModeSeeded) {
                s.v[j] = p.v_work[(size_t)b * n + j];
                tight_local += p.tight_cnt[(size_t)b * n + j];
            }
        }
    }
    bool cold = (p.mode != kModeSeeded);
    if (p.phase >= 2) {
        // (everything below up to the outputs belongs to phases 0 and 1)
    } else if (p.mode == kModeSeeded) {
        const int tt = s.bc.sum_i32(tight_local);  // includes the barrier that publishes the init
        tight_total = tt;
        cold = (double)tt < 1.2 * n;  // quality gate, lapjv_seeded.cpp:116
        if (cold) branch = kBranchFallback;
    } else {
        __syncthreads();
    }
    bool run_paths = false;
    if (p.phase >= 2) {
    } else if (cold) {
        if constexpr (LISTS) {
            const size_t lo_ = (size_t)b * n;
            nf = s.template cold_prepare<true>(p.arr_lval ? p.arr_lval + lo_ * kArrListEntries : nullptr,
                                               p.arr_lval ? p.arr_lcol + lo_ * kArrListEntries : nullptr,
                                               p.arr_lval ? p.arr_ltau + lo_ : nullptr);
        } else {
            nf = s.template cold_prepare<false>(nullptr, nullptr, nullptr);
        }
        free_after_greedy = nf;
        run_paths = nf > 0;
    } else if constexpr (!LISTS) {
        if (s.bc.wave == 0) s.greedy_wave0(p.tight_bits + (size_t)b * n * W, p.tight_cnt + (size_t)b * n);
        __syncthreads();
        nf = s.ctrl->nfree;
        free_after_greedy = nf;
        if (nf == 0) {
            branch = kBranchAllMatched;
        } else {
            branch = kBranchSsp;
            s.micro_arr(nf, p.u_tight + (size_t)b * n, p.tight_eps);
            t_serial = __builtin_amdgcn_s_memrealtime();
            run_paths = true;
        }
    }
    if (p.phase == 1) {
        // ---- hand over to the cooperative kernel: state to the global arrays, mailbox zeroed
        __syncthreads();
        const size_t o = (size_t)b * n;
        for (int j = tid; j < n; j += blockDim.x) {
            if (s.x != p.g_x + o) p.g_x[o + j] = s.x[j];
            if (s.y != p.g_y + o) p.g_y[o + j] = s.y[j];
            if (s.v != p.g_v + o) p.g_v[o + j] = s.v[j];
            if (s.fr != p.g_fr + o && j < nf) p.g_fr[o + j] = s.fr[j];
        }
        const size_t ng = (size_t)p.mail_granules;
        for (size_t q = tid; q < ng; q += blockDim.x) p.mail[(size_t)b * ng + q] = 0ull;
        if (tid == 0) {
            int *hand = p.hand + (size_t)b * kHandInts;
            const int e1 = s.err | s.ctrl->err;
            hand[0] = (run_paths && !e1) ? nf : 0;
            hand[1] = 0;
            hand[2] = 0;
            hand[3] = 0;
            hand[4] = 0;
            hand[5] = (int)branch;
            hand[6] = (int)tight_total;
            hand[7] = (int)free_after_greedy;
            hand[8] = s.arr_fired;
            hand[9] = s.transfer_rows;
            hand[10] = s.arr_iters;
            hand[11] = (int)(s.colred_elems & 0xffffffffLL);
            hand[12] = (int)(s.colred_elems >> 32);
            hand[13] = e1;
            hand[14] = LISTS ? s.ctrl->first_fire : 0;  // row-reduction iterations answered from the candidate lists
            {  // this launch's duration, 10 ns ticks (phase 2 adds it to stats slot 13)
                const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t_start;
                hand[15] = (dt < 0x7fffffffull) ? (int)dt : 0x7fffffff;
            }
            if (p.cstats) {
                for (int q = 0; q < kCoopStats; ++q) p.cstats[(size_t)b * kCoopStats + q] = 0;
            }
        }
        return;
    }
    if constexpr (LISTS) return;  // (phase 1 only: everything below belongs to the plain instantiation)
    if (p.phase == 0 && run_paths && !s.err) s.augment_all(0, nf);
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
        if (p.u_out) {
            const int xj = s.x[j];  // row j is matched to column xj
            p.u_out[(size_t)b * n + j] = (xj >= 0) ? s.C[(size_t)j * n + xj] - s.v[xj] : 0.0;
        }
    }
    if (tid == 0) {
        if (p.helper && p.pf_ring)
            __hip_atomic_store(&p.pf_ring[(size_t)b * kRingInts], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
            if (p.phase == 2) st[13] += p.hand[(size_t)b * kHandInts + 15];  // + the preparation launch
            st[14] = (long long)(t_serial - t_start);  // greedy + micro-ARR part (SSP branch)
            // paths the cooperative kernel completed | why it stopped early << 32 (-1: not used)
            st[15] = (p.phase == 2 && p.mail) ? ((long long)p.hand[(size_t)b * kHandInts + 1] |
                                       ((long long)p.hand[(size_t)b * kHandInts + 3] << 32))
                                    : -1;
            for (int q = 16; q < kStatsPerInstance; ++q) st[q] = 0;
            if (p.phase == 2) st[27] = p.hand[(size_t)b * kHandInts + 14];  // row-reduction iterations answered from the candidate lists
            if (p.phase == 2 && p.cstats) {
                for (int q = 0; q < 11; ++q) st[16 + q] = p.cstats[(size_t)b * kCoopStats + 5 + q];  // exchange rounds; stamps
            }
#ifdef LAPWARM_STAMPS
            for (int q = 0; q < 16; ++q) st[16 + q] = s.stamps[q];
#endif
        }
    }
}

template <int CH, int LDSL, int TB, bool LISTS>
hipError_t launch_one(const SolverParams &p, int threads, size_t lds_bytes, hipStream_t stream)
{
    auto kern = jv_instance_kernel<CH, LDSL, TB, LISTS>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    const int padded = (p.batch + 7) & ~7;  // helpers sit at the same index modulo 8 (= same XCD) as their solver
    hipLaunchKernelGGL(kern, dim3(p.helper ? padded * (1 + p.helper) : p.batch), dim3(threads), lds_bytes, stream, p);
    return hipGetLastError();
}

}  // namespace

bool arr_lists_enabled(int n)
{
    static const int on = [] {
        const char *e = getenv("LAPWARM_ARR_LISTS");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    return on && n >= 512;
}

// LDS levels: 2 = position-owned search, every array in LDS; 1 = x and the free-row list in global
// memory; 0 = all global; 8 = all global + head rows staged in LDS row slots.
// row slots of level 8: two (the next queued head's row is requested one step ahead) when they
// fit beside the control blocks, else one, else none
__host__ __device__ int solver_row_slots(int n, int ch)
{
    const size_t padded = ((size_t)n + (size_t)ch * 64 - 1) / ((size_t)ch * 64) * ((size_t)ch * 64);
    const size_t W = ((size_t)n + 31) >> 5;
    const size_t fixed = sizeof(BlockExchange) + sizeof(Ctrl) + sizeof(uint32_t) * ((W + 1) & ~(size_t)1) * 5 + 16;
    if (fixed + 2 * padded * sizeof(double) <= (size_t)kLdsBudgetBytes) return 2;
    if (fixed + padded * sizeof(double) <= (size_t)kLdsBudgetBytes) return 1;
    return 0;
}

// n = 8192 -> 512 threads x 16 positions, two row slots.  Measured (uniform instance, ms per solve):
// n = 8192: 2,278 here against 2,658 with 1024 x 8 and the state in global memory (level 0);
// n = 16384 (512 x 32, one slot): 51 s against 25 s for 1024 x 16 at level 0 -- the step time grows
// with the positions per thread, so K5 stays on the generic path.
bool large_row_geometry(int n, int *threads, int *ch)
{
    if (n != 8192) return false;
    *threads = 512;
    *ch = n / 512;
    return solver_row_slots(n, *ch) >= 1;
}

size_t solver_lds_bytes(int n, int ch, int level)
{
    const int W = (n + 31) >> 5;
    const int Wpad = (W + 1) & ~1;
    (void)ch;
    size_t bytes = sizeof(BlockExchange) + sizeof(Ctrl) + sizeof(uint32_t) * (size_t)Wpad * 5;
    if (level == 8) {  // control blocks + the row slots; all solver state in global memory
        const size_t padded = ((size_t)n + (size_t)ch * 64 - 1) / ((size_t)ch * 64) * ((size_t)ch * 64);
        return bytes + (size_t)solver_row_slots(n, ch) * padded * sizeof(double) + 16;
    }
    if (level >= 1) bytes += (size_t)n * (2 * sizeof(double) + 5 * sizeof(int)) + 2 * sizeof(int);
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

// Helper workgroups: rows of 8-64 KiB (n = 1024 .. 8192, even).  Measured on the same box, solver
// kernel per launch: K3 79.6 -> 73.2 ms, K4 slice 327.8 -> 276.0 ms, n = 8192 2.28 -> 2.09 s,
// K2 (n = 512) no change, n = 16384 worse.
bool solver_uses_helpers(int n)
{
    static const int want = [] {
        const char *e = getenv("LAPWARM_HELPER");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    static const int max_n = [] {
        const char *e = getenv("LAPWARM_HELPER_MAX_N");
        return e ? atoi(e) : 8192;  // n = 16384: 29.3 s with a helper against 25.6 s without
    }();
    return want && n >= 1024 && n <= max_n && n % 2 == 0;
}

static hipError_t launch_phase(const SolverParams &p_in, int threads_hint, hipStream_t stream);

// Does this launch use the instantiation with the candidate-list row reduction?  (Cold solves whose
// caller provided the list workspace; their phase 0 or, with the cooperative shortest-path phase, phase 1.)
static bool phase_uses_lists(const SolverParams &p)
{
    return p.mode == kModeCold && p.phase == 1 && p.arr_lval && p.arr_lcol && p.arr_ltau && p.hand && p.g_x;
}

// The whole solve: one launch of jv_instance_kernel, or -- where the cooperative shortest-path phase
// is enabled for this size (coop_ssp.hip) -- three: phase 1 (greedy / micro-ARR / cold preparation),
// the cooperative kernel, phase 2 (whatever it left + the outputs).
hipError_t launch_solver(const SolverParams &p_in, int threads_hint, hipStream_t stream)
{
    if (!(coop_enabled(p_in.n) && p_in.hand && p_in.mail && p_in.cstats && p_in.g_x)) {
        SolverParams p = p_in;
        p.phase = 1;
        p.mail = nullptr;
        p.mail_granules = 0;
        if (phase_uses_lists(p)) {
            // cold solve with candidate lists: preparation in its own instantiation, then the shortest paths
            hipError_t e = launch_phase(p, threads_hint, stream);
            if (e != hipSuccess) return e;
            p.phase = 2;
            return launch_phase(p, threads_hint, stream);
        }
        p.phase = 0;
        return launch_phase(p, threads_hint, stream);
    }
    SolverParams p = p_in;
    p.phase = 1;
    p.mail_granules = (int)coop_mail_granules(p.n);
    hipError_t e = launch_phase(p, threads_hint, stream);
    if (e != hipSuccess) return e;
    CoopParams c;
    c.C = p.C;
    c.n = p.n;
    c.batch = p.batch;
    c.G = 0;
    c.first = 0;
    c.count = p.batch;
    c.v = p.g_v;
    c.x = p.g_x;
    c.y = p.g_y;
    c.pred = p.g_pred;
    c.fr = p.g_fr;
    c.hand = p.hand;
    c.cstats = p.cstats;
    c.mail = p.mail;
    // The cooperative kernel stops at a path it does not handle (a minima collection with a tie: rare,
    // but seeds that went through float32 produce a few dozen per instance); jv_instance_kernel then
    // searches that ONE path (phase 3) and the cooperative kernel carries on.  The host cannot know how
    // often that happens, so a fixed number of (cooperative, one-path) pairs is enqueued -- a launch with
    // nothing to do returns at once (~2 us) -- and the final phase 2 finishes whatever is left.
    static const int pairs = [] {
        const char *ev = getenv("LAPWARM_COOP_RELAUNCHES");
        const int k = ev ? atoi(ev) : 96;
        return (k >= 0 && k <= 4096) ? k : 96;
    }();
    for (int k = 0; k <= pairs; ++k) {
        e = launch_coop(c, stream);
        if (e != hipSuccess) return e;
        if (k == pairs) break;
        p.phase = 3;
        e = launch_phase(p, threads_hint, stream);
        if (e != hipSuccess) return e;
    }
    p.phase = 2;
    return launch_phase(p, threads_hint, stream);
}

static hipError_t launch_phase(const SolverParams &p_in, int threads_hint, hipStream_t stream)
{
    SolverParams p = p_in;
    static const int n_helpers = [] {
        const char *e = getenv("LAPWARM_HELPERS_PER_INSTANCE");
        const int k = e ? atoi(e) : 1;
        return (k >= 1 && k <= 4) ? k : 1;
    }();
    // (a helper can only help while its solver runs: with more workgroups than CUs the helpers would
    // be dispatched after the solvers they serve and leave at once -- skip them.  Assumptions, stated:
    // workgroups are dispatched in index order, so every solver of THIS launch is resident before its
    // helper; a helper spins until its solver's done flag or 0.5 s (60 s above n = 4096) and holds a
    // CU's LDS meanwhile, so with several launches resident -- bench.py --inflight -- helpers can delay
    // the solvers of a later launch, never deadlock them: every solver exit sets the flag.)
    static const int n_cus = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        }
        return cus;
    }();
    p.helper = (p.phase == 0 && p.mode == kModeSeeded && p.pf_ring && solver_uses_helpers(p.n) &&
                p.batch * (1 + n_helpers) <= n_cus)
                   ? n_helpers
                   : 0;
    int threads, ch;
    // measured (n=2048, ARR-dominated cold solve): 512 threads 2.6 us/iteration, 1024: 3.2, 256: 3.1
    if (threads_hint <= 0 && p.mode == kModeCold && p.n > 1024 && p.n <= 2048) threads_hint = 512;
    solver_geometry(p.n, threads_hint, &threads, &ch);
    if ((long long)threads * ch < p.n) return hipErrorInvalidValue;  // n > 16384
    // Rows that no longer fit the L1 (n > 4,427, where the state leaves LDS as well): 512 threads
    // with n/512 positions each -- duals cached in registers (256 VGPRs per thread at this size),
    // every head row brought into LDS by coalesced LDS-DMA (level 8).  Seeded mode only: the cold
    // ARR loop keeps the generic geometry.
    if (threads_hint <= 0 && p.mode == kModeSeeded && large_row_geometry(p.n, &threads, &ch)) {
        if (!p.g_x) return hipErrorInvalidValue;
        const size_t lds8 = solver_lds_bytes(p.n, ch, 8);
        return launch_one<16, 8, 512, false>(p, threads, lds8, stream);
    }
    const int level = solver_lds_level(p.n, ch);
    if (level < 2 && !p.g_x) return hipErrorInvalidValue;
    const size_t lds = solver_lds_bytes(p.n, ch, level);
#define LAPWARM_CASE(CHV, LISTV)                                                              \
    case CHV:                                                                                  \
        if (threads <= 256) {                                                                  \
            if (level == 2) return launch_one<CHV, 2, 256, LISTV>(p, threads, lds, stream);    \
            if (level == 1) return launch_one<CHV, 1, 256, LISTV>(p, threads, lds, stream);    \
            return launch_one<CHV, 0, 256, LISTV>(p, threads, lds, stream);                    \
        }                                                                                      \
        if (level == 2) return launch_one<CHV, 2, 1024, LISTV>(p, threads, lds, stream);       \
        if (level == 1) return launch_one<CHV, 1, 1024, LISTV>(p, threads, lds, stream);       \
        return launch_one<CHV, 0, 1024, LISTV>(p, threads, lds, stream);
    if (!phase_uses_lists(p)) {
        switch (ch) {
            LAPWARM_CASE(1, false)
            LAPWARM_CASE(2, false)
            LAPWARM_CASE(4, false)
            LAPWARM_CASE(8, false)
            LAPWARM_CASE(16, false)
        }
    } else {
        switch (ch) {
            LAPWARM_CASE(1, true)
            LAPWARM_CASE(2, true)
            LAPWARM_CASE(4, true)
            LAPWARM_CASE(8, true)
            LAPWARM_CASE(16, true)
        }
    }
#undef LAPWARM_CASE
    return hipErrorInvalidValue;
}

bool solver_needs_global_state(int n)
{
    // a threads_hint may pick another CH: be conservative for every supported geometry.
    // Level 2 keeps everything in LDS; 0, 1 and 8 use the global workspace.
    for (int c = 1; c <= 16; c <<= 1) {
        if (solver_lds_level(n, c) < 2) return true;
    }
    int t, c;
    if (large_row_geometry(n, &t, &c)) return true;
    return false;
}

}  // namespace lapwarm
