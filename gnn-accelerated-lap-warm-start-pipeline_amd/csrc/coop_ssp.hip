// coop_ssp.hip -- the shortest-augmenting-path phase of ONE LAP instance spread over G single-wave
// workgroups on G compute units (round 3).  Reference semantics reproduced bit for bit:
// LAP/_lapjv_cpp/lapjv.cpp:153-319 (_find_dense, _scan_dense, find_path_dense, _ca_dense).
//
// Why: with one workgroup per instance a relax step costs O(n) work on ONE CU (1.7 us at n = 2048,
// 28 us at n = 16384 where the state no longer fits LDS).  Here member g of an instance owns the
// POSITIONS [g*64*CH, (g+1)*64*CH) of the column order with the column, its dual, its matched row and
// its distance in registers, reads only its own piece of the head row, and the members agree on the
// (few) order-changing events of a step through one all-to-all exchange of 8-byte {value, tag}
// granules in global memory (cdna_hip_programming.md Guideline 16, recipe R2: every shared word is
// an agent-scope relaxed atomic = sc1 access, the data is the flag, no fences).  A step then costs
// one exchange (~1 us) whatever n is.  Every member replays the same events on the same data, so the
// control state (lo, hi, head, level, SCAN queue) is replicated, never communicated.
//
// What is NOT handled here ends the cooperative phase for that instance at a path boundary
// ("bail"): x, y, v are only written at the end of a path, so the state in global memory is that of
// the path's start and jv_instance_kernel (phase 2) resumes from free row `hand[1]`.  Bails today:
// a minima collection with a tie event (never on continuous random costs),
// a SCAN list longer than the replicated queue.
//
// Happens-before (every cross-member datum; "round" = publish + poll of all members' records):
//   v[], y[] (path-constant)      written in path_end / backtrack, drained, then the ack / go round;
//                                 read with sc1 loads after that round
//   pred[]                        a column's predecessor travels with it in registers; the leader alone
//                                 writes pred[j] when j joins the SCAN list (or ends the path) and
//                                 alone reads it in the backtrack: no cross-member datum
//   x[]                           only the leader touches it
//   mailbox buf[seq & 1]          a member publishes round r+2 (same buffer as r) only after it has
//                                 all records of round r+1, which every member publishes after it
//                                 finished reading round r
#include <stdlib.h>

#include "device_utils.hpp"
#include "jv_solver.hpp"

namespace lapwarm {

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
#define LAPWARM_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int kK = 6;      // granules per member record
constexpr int kEmax = 3;   // tie events a member publishes per round (2 granules each)
constexpr int kWin = 4;    // window: events applied per round = columns at order[hi .. hi+3] published per round
constexpr int kWinGran = 4 * kWin;
constexpr int kQ = 1024;   // replicated SCAN queue (entries: column, matched row)
constexpr unsigned kFlagBail = 1u, kFlagErr = 2u;

__device__ __forceinline__ void st_gran(unsigned long long *g, unsigned tag, unsigned val)
{
    __hip_atomic_store((gu64 *)g, ((unsigned long long)tag << 32) | val, LAPWARM_RLX_AGENT);
}
__device__ __forceinline__ unsigned long long ld_gran(const unsigned long long *g)
{
    return __hip_atomic_load((gu64 *)g, LAPWARM_RLX_AGENT);
}
__device__ __forceinline__ double ld_f64(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load((gu64 *)p, LAPWARM_RLX_AGENT));
}
__device__ __forceinline__ void st_f64(double *p, double x)
{
    __hip_atomic_store((gu64 *)p, (unsigned long long)__double_as_longlong(x), LAPWARM_RLX_AGENT);
}
__device__ __forceinline__ int ld_i32(const int *p) { return (int)__hip_atomic_load((gu32 *)p, LAPWARM_RLX_AGENT); }
__device__ __forceinline__ void st_i32(int *p, int x) { __hip_atomic_store((gu32 *)p, (unsigned)x, LAPWARM_RLX_AGENT); }
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lo32(double x) { return (unsigned)(__double_as_longlong(x) & 0xffffffffLL); }
__device__ __forceinline__ unsigned hi32(double x) { return (unsigned)((unsigned long long)__double_as_longlong(x) >> 32); }
__device__ __forceinline__ double mk_f64(unsigned lo, unsigned hi)
{
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

enum { kRcGo = 0, kRcTarget = 1, kRcBail = 2, kRcErr = 3 };

template <int CH, int NL>
struct Member {
    static constexpr int P = 64 * CH;  // positions per member
    const double *C;
    int n;
    double *v;
    int *x, *y, *pred;
    unsigned long long *mail;  // [2][NGtot]
    int G, g, lane, base, b0, NGm, NGtot;
    int2 *q;       // LDS: replicated SCAN queue
    unsigned *rx;  // LDS: payloads of the round just polled
    int2 *lj;      // LDS: (column, matched row) of this member's positions during a collection
    int jr[CH], yr[CH], pr[CH];  // column at the position, its matched row, its predecessor row
    double vr[CH], dk[CH];       // its dual, its tentative distance
    int lo, hi, ready, head_i, head_j;
    double level;
    unsigned seq;
    long long scan_elems, init_elems;
    int paths, finds, scan_steps;
    int err, bail_reason;

    // ---------------------------------------------------------------- exchange
    __device__ __forceinline__ void publish(unsigned w0, unsigned w1, unsigned w2, unsigned w3, unsigned w4, unsigned w5)
    {
        unsigned w = w0;
        w = (lane == 1) ? w1 : w;
        w = (lane == 2) ? w2 : w;
        w = (lane == 3) ? w3 : w;
        w = (lane == 4) ? w4 : w;
        w = (lane == 5) ? w5 : w;
        if (lane < kK) st_gran(mail + (size_t)(seq & 1u) * NGtot + g * kK + lane, seq, w);
    }
    // Waits until the first `need` granules of this round's buffer carry this round's tag and
    // leaves their payloads in rx[].  Bounded: ~2 s (a member that never arrives, e.g. because it
    // was never dispatched, must end in an error code, not in a hang).
    __device__ __forceinline__ bool poll(int need)
    {
        const unsigned long long *buf = mail + (size_t)(seq & 1u) * NGtot;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            unsigned val[NL];
#pragma unroll
            for (int qd = 0; qd < NL; ++qd) {
                const int idx = qd * 64 + lane;
                val[qd] = 0;
                if (idx < need) {
                    const unsigned long long w = ld_gran(buf + idx);
                    val[qd] = (unsigned)w;
                    ok &= (unsigned)(w >> 32) == seq;
                }
            }
            if (__all(ok)) {
#pragma unroll
                for (int qd = 0; qd < NL; ++qd) {
                    const int idx = qd * 64 + lane;
                    if (idx < need) rx[idx] = val[qd];
                }
                return true;
            }
            if ((++spins & 255u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) return false;
        }
    }
    __device__ __forceinline__ unsigned rxu(int idx) const { return (unsigned)uni((int)rx[idx]); }

    // flags word of every member (granule `slot` of its record, bits `shift`..): any bail / error?
    __device__ __forceinline__ int check_flags(int slot, int shift)
    {
        unsigned f = 0;
        if (lane < G) f = (rx[lane * kK + slot] >> shift) & 3u;
        if (__ballot((f & kFlagErr) != 0)) {
            if (!err) err = 21;  // another member reported an error
            return kRcErr;
        }
        if (__ballot((f & kFlagBail) != 0)) return kRcBail;
        return kRcGo;
    }

    // ---------------------------------------------------------------- path start (lapjv.cpp:233-237)
    __device__ __forceinline__ void path_init(int start)
    {
        const double *row = C + (size_t)start * n;
        double c0[CH];
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int k = b0 + r;
            jr[r] = (k < n) ? k : n - 1;
            c0[r] = row[jr[r]];
        }
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            vr[r] = ld_f64(v + jr[r]);
            yr[r] = ld_i32(y + jr[r]);
        }
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int k = b0 + r;
            dk[r] = (k < n) ? c0[r] - vr[r] : pos_inf();
            pr[r] = start;
        }
        paths++;
        init_elems += n;
        lo = hi = ready = 0;
    }

    // ---------------------------------------------------------------- minima collection (lapjv.cpp:153-171, :243-256)
    __device__ __forceinline__ int collect(int &target)
    {
        ready = lo;
        finds++;
        double tv = pos_inf();
        int tp = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int k = b0 + r;
            // position lo always starts as the holder, whatever its value
            if (k >= lo && k < n && (k == lo || dk[r] < tv)) {
                tv = dk[r];
                tp = k;
            }
        }
        double runv = tv, wtv;
        int runp = tp, wtp;
        wave_excl_prefix_min_pair(runv, runp, lane, &wtv, &wtp);
        // the column (and its matched row) at this member's first minimum
        int selj = 0, sely = -1, selp = 0;
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            if (b0 + r == wtp) {
                selj = jr[r];
                sely = yr[r];
                selp = pr[r];
            }
        }
        const int ol = (wtp != 0x7fffffff) ? uni((wtp - base) / CH) : 0;
        const int mj_own = __builtin_amdgcn_readlane(selj, ol);
        const int my_own = __builtin_amdgcn_readlane(sely, ol);
        const int mp_own = __builtin_amdgcn_readlane(selp, ol);

        // ---- round A: every member's (minimum, first position, column, matched row)
        ++seq;
        publish(lo32(wtv), hi32(wtv), (unsigned)wtp, (unsigned)mj_own, (unsigned)(my_own + 1),
                (err ? kFlagErr : 0u) | ((unsigned)mp_own << 4));
        if (!poll(NGm)) {
            err = 20;
            return kRcErr;
        }
        if (const int rc = check_flags(5, 0)) return rc;
        double mv = pos_inf();
        int mp = 0x7fffffff, mcol = 0, mrow = -1, mpred = 0;
        if (lane < G) {
            const unsigned *rec = rx + lane * kK;
            mv = mk_f64(rec[0], rec[1]);
            mp = (int)rec[2];
            mcol = (int)rec[3];
            mrow = (int)rec[4] - 1;
            mpred = (int)(rec[5] >> 4);
        }
        double pv = mv, totv;
        int pp = mp, totp;
        wave_excl_prefix_min_pair(pv, pp, lane, &totv, &totp);
        if ((unsigned)totp >= (unsigned)n) {
            err = 7;  // no TODO position left: the search should have ended before
            return kRcErr;
        }
        const double pfv = readlane_f64(pv, g);
        const int pfp = __builtin_amdgcn_readlane(pp, g);
        const int tm = uni(totp / P);
        const int min_col = __builtin_amdgcn_readlane(mcol, tm);
        const int min_row = __builtin_amdgcn_readlane(mrow, tm);
        const int min_pred = __builtin_amdgcn_readlane(mpred, tm);
        int pcol = 0, prow = -1, ppred = 0;
        if (pfp != 0x7fffffff) {
            const int pm = uni(pfp / P);
            pcol = __builtin_amdgcn_readlane(mcol, pm);
            prow = __builtin_amdgcn_readlane(mrow, pm);
            ppred = __builtin_amdgcn_readlane(mpred, pm);
        }
        if ((unsigned)min_col >= (unsigned)n || min_row >= n || (unsigned)pcol >= (unsigned)n || prow >= n ||
            (unsigned)min_pred >= (unsigned)n || (unsigned)ppred >= (unsigned)n) {
            err = 8;
            return kRcErr;
        }
        // positions of earlier members precede every position of this one
        if (pair_less(pfv, pfp, runv, runp)) {
            runv = pfv;
            runp = pfp;
        }
        // classify the owned positions: strict event = undercuts everything before it, tie event = equals it
        unsigned sb = 0;
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
        // what each strict event position receives: the column of the previous record holder
#pragma unroll
        for (int r = 0; r < CH; ++r) lj[lane * CH + r] = make_int2(jr[r] | (pr[r] << 16), yr[r]);
        int pc[CH], py[CH], pq[CH];
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            pc[r] = pcol;
            py[r] = prow;
            pq[r] = ppred;
            if (((sb >> r) & 1u) && prevpos[r] >= base) {
                const int2 t = lj[prevpos[r] - base];
                pc[r] = t.x & 0xffff;
                pq[r] = (int)((unsigned)t.x >> 16);
                py[r] = t.y;
            }
        }

        // ---- round B: did any member see a tie event?  (a tie changes the permutation of the TODO
        // positions in a way that needs the whole ordered event list: not handled here)
        const unsigned long long anytie_local = __ballot(tie);
        ++seq;
        publish((anytie_local ? 4u : 0u) | (err ? kFlagErr : 0u), 0, 0, 0, 0, 0);
        if (!poll(NGm)) {
            err = 20;
            return kRcErr;
        }
        if (const int rc = check_flags(0, 0)) return rc;
        {
            unsigned f = 0;
            if (lane < G) f = rx[lane * kK] & 4u;
            if (__ballot(f != 0)) {
                bail_reason = 1;
                return kRcBail;
            }
        }
        hi = lo + 1;
        level = totv;
        head_j = min_col;
        head_i = min_row;
        // the leader keeps the predecessor of every column that joins the SCAN list (final from here
        // on): the backtrack only ever follows those
        if (g == 0 && lane == 0) st_i32(pred + min_col, min_pred);
        if (head_i < 0) {
            target = head_j;  // the only minimum is a free column: the path ends here
            return kRcTarget;
        }
        // tie-free: the serial swap sequence collapses to a shift, applied by the owners themselves
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            if ((sb >> r) & 1u) {
                jr[r] = pc[r];
                yr[r] = py[r];
                pr[r] = pq[r];
                dk[r] = prevval[r];
                vr[r] = ld_f64(v + pc[r]);
            }
        }
        if (totp != lo && (unsigned)(lo - b0) < (unsigned)CH) {
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                if (b0 + r == lo) {
                    jr[r] = min_col;
                    yr[r] = min_row;
                    pr[r] = min_pred;
                    dk[r] = totv;
                }
            }
        }
        if (lane == 0) q[lo & (kQ - 1)] = make_int2(min_col, min_row);
        return kRcGo;
    }

    // ---------------------------------------------------------------- relax the head of the SCAN list (lapjv.cpp:185-207)
    __device__ __forceinline__ int relax(int &target)
    {
        if ((unsigned)head_i >= (unsigned)n || (unsigned)head_j >= (unsigned)n) {
            err = 6;
            return kRcErr;
        }
        const double *row = C + (size_t)head_i * n;
        double c[CH];
#pragma unroll
        for (int r = 0; r < CH; ++r) c[r] = row[umin_u32((unsigned)jr[r], (unsigned)(n - 1))];
        const double c_head = row[head_j];
        const double v_head = ld_f64(v + head_j);
        scan_steps++;
        scan_elems += (long long)(n - hi);
#pragma unroll
        for (int r = 0; r < CH; ++r) pin(c[r]);
        const double h = (c_head - v_head) - level;  // (cost - v) - level : lapjv.cpp:189
        unsigned evm = 0;
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int k = b0 + r;
            const bool act = (k >= hi) & (k < n);
            const double cand = (c[r] - vr[r]) - h;  // (cost - v) - h : lapjv.cpp:195
            const bool imp = act & (cand < dk[r]);
            const bool ev = imp & (cand == level);
            dk[r] = imp ? cand : dk[r];
            pr[r] = imp ? head_i : pr[r];
            evm |= ev ? (1u << r) : 0u;
        }
        for (int round = 0;; ++round) {
            // ---- this member's pending tie events, in position order
            int cnt = 0;
#pragma unroll
            for (int r = 0; r < CH; ++r) cnt += __popcll(__ballot((evm >> r) & 1u));
            unsigned e0a = 0, e0b = 0, e1a = 0, e1b = 0, e2a = 0, e2b = 0;
            if (cnt) {
                unsigned long long any = __ballot(evm != 0);
                int emitted = 0;
                while (any && emitted < kEmax) {
                    const int l = __builtin_ctzll(any);
                    any &= any - 1;
                    unsigned em = (unsigned)__builtin_amdgcn_readlane((int)evm, l);
                    while (em && emitted < kEmax) {
                        const int r = __builtin_ctz(em);
                        em &= em - 1;
                        int sj = jr[0], sy = yr[0];
#pragma unroll
                        for (int qd = 1; qd < CH; ++qd) {
                            if (r == qd) {
                                sj = jr[qd];
                                sy = yr[qd];
                            }
                        }
                        const unsigned j = (unsigned)__builtin_amdgcn_readlane(sj, l);
                        const unsigned yv = (unsigned)(__builtin_amdgcn_readlane(sy, l) + 1);
                        const unsigned pos = (unsigned)(base + l * CH + r);
                        const unsigned wa = pos | (j << 16);
                        if (emitted == 0) {
                            e0a = wa;
                            e0b = yv;
                        } else if (emitted == 1) {
                            e1a = wa;
                            e1b = yv;
                        } else {
                            e2a = wa;
                            e2b = yv;
                        }
                        ++emitted;
                    }
                }
            }
            e0b |= (unsigned)(cnt > 255 ? 255 : cnt) << 20;
            e0b |= (err ? kFlagErr : 0u) << 28;
            ++seq;
            // ---- the columns at order[hi .. hi+3] (what this round's events displace), with their
            // matched rows and distances as they stand after this step's update
            {
                unsigned long long *win = mail + (size_t)(seq & 1u) * NGtot + NGm;
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    const int k = b0 + r;
                    const unsigned t = (unsigned)(k - hi);
                    if (t < (unsigned)kWin && k < n) {
                        st_gran(win + 4 * t + 0, seq, (unsigned)jr[r] | ((unsigned)pr[r] << 16));
                        st_gran(win + 4 * t + 1, seq, (unsigned)(yr[r] + 1));
                        st_gran(win + 4 * t + 2, seq, lo32(dk[r]));
                        st_gran(win + 4 * t + 3, seq, hi32(dk[r]));
                    }
                }
                // slots beyond the last position: nobody owns them, the last member fills them in
                if (g == G - 1 && hi + kWin > n && lane < kWinGran) {
                    const int t = lane >> 2;
                    if (hi + t >= n) st_gran(win + lane, seq, 0u);
                }
            }
            publish(e0a, e0b, e1a, e1b, e2a, e2b);
            if (!poll(NGm + kWinGran)) {
                err = 20;
                return kRcErr;
            }
            if (const int rc = check_flags(1, 28)) return rc;
            int cm = 0;
            if (lane < G) cm = (int)((rx[lane * kK + 1] >> 20) & 0xffu);
            const int total = wave_sum_i32(cm);
            if (total == 0) break;
            // ---- the first events of the round in global position order = member order
            int take = 0;
            int ep0 = 0, ep1 = 0, ep2 = 0, ep3 = 0, ej0 = 0, ej1 = 0, ej2 = 0, ej3 = 0, ey0 = 0, ey1 = 0, ey2 = 0, ey3 = 0;
            {
                unsigned long long mm = __ballot(cm > 0);
                bool blocked = false;
                while (mm && take < kWin && !blocked) {
                    const int m = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    const int c_m = __builtin_amdgcn_readlane(cm, m);
                    const int pub = (c_m < kEmax) ? c_m : kEmax;
                    for (int e = 0; e < pub && take < kWin; ++e) {
                        const unsigned wa = rxu(m * kK + 2 * e);
                        const unsigned wb = rxu(m * kK + 2 * e + 1);
                        const int pp_ = (int)(wa & 0xffffu), jj_ = (int)(wa >> 16), yy_ = (int)(wb & 0xfffffu) - 1;
                        if (take == 0) {
                            ep0 = pp_, ej0 = jj_, ey0 = yy_;
                        } else if (take == 1) {
                            ep1 = pp_, ej1 = jj_, ey1 = yy_;
                        } else if (take == 2) {
                            ep2 = pp_, ej2 = jj_, ey2 = yy_;
                        } else {
                            ep3 = pp_, ej3 = jj_, ey3 = yy_;
                        }
                        ++take;
                    }
                    if (c_m > kEmax) blocked = true;  // it has events it could not publish: later members wait
                }
            }
            if (take <= 0 || hi + take > n) {
                err = 9;
                return kRcErr;
            }
            // ---- a free column among them ends the path at the first one (lapjv.cpp:200-201).  Every
            // event column was improved by this very step: its predecessor is the head's row.
            if (ey0 < 0) {
                target = ej0;
            } else if (take > 1 && ey1 < 0) {
                target = ej1;
            } else if (take > 2 && ey2 < 0) {
                target = ej2;
            } else if (take > 3 && ey3 < 0) {
                target = ej3;
            }
            if (target >= 0) {
                if ((unsigned)target >= (unsigned)n) {
                    err = 8;
                    return kRcErr;
                }
                if (g == 0 && lane == 0) st_i32(pred + target, head_i);
                return kRcTarget;
            }
            // ---- replay the swaps cols[k] = cols[hi]; cols[hi++] = j (lapjv.cpp:203-204) on the window
            int wa_[kWin], wy_[kWin], wq_[kWin];
            double wd_[kWin];
#pragma unroll
            for (int t = 0; t < kWin; ++t) {
                const unsigned w0 = rxu(NGm + 4 * t);
                wa_[t] = (int)(w0 & 0xffffu);
                wq_[t] = (int)(w0 >> 16);
                wy_[t] = (int)rxu(NGm + 4 * t + 1) - 1;
                wd_[t] = mk_f64(rxu(NGm + 4 * t + 2), rxu(NGm + 4 * t + 3));
            }
            const int eps[kWin] = {ep0, ep1, ep2, ep3};
            const int ejs[kWin] = {ej0, ej1, ej2, ej3};
            const int eys[kWin] = {ey0, ey1, ey2, ey3};
#pragma unroll
            for (int s = 0; s < kWin; ++s) {
                if (s < take) {
                    const int a = wa_[s], ya = wy_[s], qa = wq_[s];
                    const double da = wd_[s];
                    const int t = eps[s] - hi;  // >= s: the events are in position order
                    if ((unsigned)eps[s] >= (unsigned)n || (unsigned)ejs[s] >= (unsigned)n || eys[s] >= n ||
                        (unsigned)a >= (unsigned)n || ya >= n || (unsigned)qa >= (unsigned)n || t < s) {
                        err = 8;
                        return kRcErr;
                    }
                    bool moved_inside = false;
#pragma unroll
                    for (int u = s + 1; u < kWin; ++u) {
                        if (t == u && u < take) {  // the event position is a later window slot
                            wa_[u] = a;
                            wq_[u] = qa;
                            wy_[u] = ya;
                            wd_[u] = da;
                            moved_inside = true;
                        }
                    }
                    if (!moved_inside && t != s && (unsigned)(eps[s] - base) < (unsigned)P) {
                        // the event position adopts the column displaced from order[hi + s]
                        const double va = ld_f64(v + a);
#pragma unroll
                        for (int r = 0; r < CH; ++r) {
                            if (b0 + r == eps[s]) {
                                jr[r] = a;
                                yr[r] = ya;
                                pr[r] = qa;
                                dk[r] = da;
                                vr[r] = va;
                            }
                        }
                    }
                    if ((unsigned)(hi + s - base) < (unsigned)P) {
                        // order[hi + s] = the event column: it joins the SCAN list at distance `level`
#pragma unroll
                        for (int r = 0; r < CH; ++r) {
                            if (b0 + r == hi + s) {
                                jr[r] = ejs[s];
                                yr[r] = eys[s];
                                pr[r] = head_i;
                                dk[r] = level;
                            }
                        }
                    }
                    if (lane == 0) {
                        q[(hi + s) & (kQ - 1)] = make_int2(ejs[s], eys[s]);
                        if (g == 0) st_i32(pred + ejs[s], head_i);
                    }
                }
            }
            const int p_last = (take == 1) ? ep0 : ((take == 2) ? ep1 : ((take == 3) ? ep2 : ep3));
            hi += take;
            if (hi - lo >= kQ) {
                bail_reason = 2;
                return kRcBail;  // uniform: every member computes the same hi, lo
            }
            if (total == take) break;
            // more events than one round could carry: drop the ones just applied and go again
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                if (b0 + r <= p_last) evm &= ~(1u << r);
            }
            if (round > n) {
                err = 1;
                return kRcErr;
            }
        }
        ++lo;
        if (lo < hi) {
            const int2 e = q[lo & (kQ - 1)];
            head_j = uni(e.x);
            head_i = uni(e.y);
        }
        return kRcGo;
    }

    // ---------------------------------------------------------------- path end (lapjv.cpp:270-276, :302-314)
    __device__ __forceinline__ int path_end(int target, int start)
    {
        // dual update of the READY columns: v[j] += d[j] - level.  Positions below `ready` hold the
        // column that joined the SCAN list there and the level it joined at.
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int k = b0 + r;
            if (k < ready) {
                const int j = jr[r];
                const double vj = ld_f64(v + j);
                st_f64(v + j, vj + (dk[r] - level));
            }
        }
        drain();  // v[] and every pred[] store of this path have left before the ack record does
        ++seq;
        publish(err ? kFlagErr : 0u, 0, 0, 0, 0, 0);
        if (!poll(NGm)) {
            err = 20;
            return kRcErr;
        }
        if (check_flags(0, 0)) return kRcErr;
        if (g == 0) {
            // the leader walks the predecessor chain (every lane the same loads, lane 0 stores)
            int j = target, i = -1, hops = 0;
            bool ok = true;
            while (i != start && hops <= n) {
                if ((unsigned)j >= (unsigned)n) {
                    ok = false;
                    break;
                }
                i = uni(ld_i32(pred + j));
                if ((unsigned)i >= (unsigned)n) {
                    ok = false;
                    break;
                }
                if (lane == 0) st_i32(y + j, i);
                const int prev = uni(ld_i32(x + i));
                if (lane == 0) st_i32(x + i, j);
                j = prev;
                ++hops;
            }
            if (!ok || i != start) err = 3;
            drain();
        }
        ++seq;
        publish(err ? kFlagErr : 0u, 0, 0, 0, 0, 0);
        if (!poll(NGm)) {
            err = 20;
            return kRcErr;
        }
        if (check_flags(0, 0)) return kRcErr;
        return kRcGo;
    }
};

template <int CH, int NL>
__global__ void __launch_bounds__(64) coop_ssp_kernel(CoopParams p)
{
    __shared__ int2 q_s[kQ];
    __shared__ unsigned rx_s[NL * 64];
    __shared__ int2 lj_s[64 * CH];
    // members of instance b sit at block indices with the same value modulo 8: workgroups are dealt
    // to the 8 XCDs round robin, so they share an XCD (speed only -- nothing depends on it)
    const int G = p.G;
    const int grp = (int)blockIdx.x / (8 * G), rem = (int)blockIdx.x % (8 * G);
    const int g = rem / 8;
    const int b = p.first + grp * 8 + rem % 8;
    if (b >= p.first + p.count || b >= p.batch) return;
    int *hand = p.hand + (size_t)b * kHandInts;
    const int nf = hand[0];
    if (nf <= 0 || hand[2] != 0) return;
    const int n = p.n;
    const size_t o = (size_t)b * n;

    Member<CH, NL> m;
    m.C = p.C + o * n;
    m.n = n;
    m.v = p.v + o;
    m.x = p.x + o;
    m.y = p.y + o;
    m.pred = p.pred + o;
    m.G = G;
    m.g = g;
    m.lane = threadIdx.x;
    m.base = g * Member<CH, NL>::P;
    m.b0 = m.base + m.lane * CH;
    m.NGm = G * kK;
    m.NGtot = G * kK + kWinGran;
    m.mail = p.mail + (size_t)b * 2 * m.NGtot;
    m.q = q_s;
    m.rx = rx_s;
    m.lj = lj_s;
    m.seq = 0;
    m.scan_elems = m.init_elems = 0;
    m.paths = m.finds = m.scan_steps = 0;
    m.err = 0;
    m.bail_reason = 0;
    m.level = 0.0;
    m.lo = m.hi = m.ready = m.head_i = m.head_j = 0;
    const int *fr = p.fr + o;

    int done = 0;
    for (int f = 0; f < nf; ++f) {
        const long long s_scan = m.scan_elems, s_init = m.init_elems;
        const int s_paths = m.paths, s_finds = m.finds, s_steps = m.scan_steps;
        const int start = uni(fr[f]);
        if ((unsigned)start >= (unsigned)n) {
            m.err = 2;
            break;
        }
        m.path_init(start);
        int target = -1, rc = kRcGo;
        for (int guard = 0;; ++guard) {
            if (m.lo == m.hi) {
                rc = m.collect(target);
                if (rc) break;
            }
            rc = m.relax(target);
            if (rc) break;
            if (guard > 2 * n + 4) {
                m.err = 1;
                rc = kRcErr;
                break;
            }
        }
        if (rc == kRcBail) {
            // nothing of this path has touched x, y or v: hand the rest to jv_instance_kernel
            m.scan_elems = s_scan;
            m.init_elems = s_init;
            m.paths = s_paths;
            m.finds = s_finds;
            m.scan_steps = s_steps;
            break;
        }
        if (rc == kRcErr) break;
        if (m.path_end(target, start)) break;
        done = f + 1;
    }
    if (g == 0 && m.lane == 0) {
        hand[1] = done;
        hand[2] = m.err;
        hand[3] = m.bail_reason;
        long long *cs = p.cstats + (size_t)b * kCoopStats;
        cs[0] = m.paths;
        cs[1] = m.finds;
        cs[2] = m.scan_steps;
        cs[3] = m.scan_elems;
        cs[4] = m.init_elems;
        cs[5] = (long long)m.seq;
    } else if (m.err && m.lane == 0) {
        // a member other than the leader saw the error first: make sure it is not lost
        atomicMax(&hand[4], m.err);
    }
}

template <int CH, int NL>
hipError_t launch_cfg(const CoopParams &p, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL((coop_ssp_kernel<CH, NL>), dim3(grid), dim3(64), 0, stream, p);
    return hipGetLastError();
}

}  // namespace

// Positions per lane for a problem size: enough members to spread the row over many CUs, few
// enough that one exchange stays within four granule loads per lane (G <= 32).
int coop_ch(int n)
{
    static const int forced = [] {
        const char *e = getenv("LAPWARM_COOP_CH");
        return e ? atoi(e) : 0;
    }();
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16) {
        if ((n + 64 * forced - 1) / (64 * forced) <= 18) return forced;
    }
    // Measured (tools/micro/hop_bench.hip, profiles/r03_hop_bench.txt): an exchange among 8 members costs
    // 1.0-1.3 us, among 16 1.4-2.2 us (32 instances in flight), among 32 1.9-2.6 us -- the fewer members the
    // better, as long as a lane's positions fit the register file (16 positions = ~250 VGPRs).
    if (n <= 512) return 1;
    if (n <= 1024) return 2;
    if (n <= 2048) return 4;
    if (n <= 4096) return 8;
    return 16;  // 8192 -> 8 members, 16384 -> 16
}

int coop_members(int n)
{
    const int ch = coop_ch(n);
    return (n + 64 * ch - 1) / (64 * ch);
}

size_t coop_mail_granules(int n) { return 2 * ((size_t)coop_members(n) * kK + kWinGran); }

bool coop_enabled(int n)
{
    static const int min_n = [] {
        const char *e = getenv("LAPWARM_COOP_MIN_N");
        return e ? atoi(e) : 4096;
    }();
    static const int on = [] {
        const char *e = getenv("LAPWARM_COOP");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    const int ch = coop_ch(n);
    return on && n >= min_n && n <= 16384 && (ch == 1 || ch == 2 || ch == 4 || ch == 8 || ch == 16) &&
           coop_members(n) <= 18;
}

hipError_t launch_coop(const CoopParams &p_in, hipStream_t stream)
{
    CoopParams p = p_in;
    const int ch = coop_ch(p.n);
    p.G = coop_members(p.n);
    const int ng = p.G * kK + kWinGran;
    const int nl = (ng + 63) / 64;
    // every member of an instance must be resident while the instance runs: at most 1024 single-wave
    // workgroups per launch (a quarter of what the chip holds), instances in groups of 8
    int per_launch = (1024 / p.G) & ~7;
    if (per_launch < 8) per_launch = 8;
    for (int first = 0; first < p.batch; first += per_launch) {
        p.first = first;
        p.count = (p.batch - first < per_launch) ? p.batch - first : per_launch;
        const int grid = ((p.count + 7) / 8) * 8 * p.G;
        hipError_t e = hipErrorInvalidValue;
#define LAPWARM_COOP_CASE(CHV)                                              \
    if (ch == CHV) {                                                        \
        if (nl == 1) e = launch_cfg<CHV, 1>(p, grid, stream);               \
        else if (nl == 2) e = launch_cfg<CHV, 2>(p, grid, stream);          \
    }
        LAPWARM_COOP_CASE(1)
        LAPWARM_COOP_CASE(2)
        LAPWARM_COOP_CASE(4)
        LAPWARM_COOP_CASE(8)
        LAPWARM_COOP_CASE(16)
#undef LAPWARM_COOP_CASE
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace lapwarm
