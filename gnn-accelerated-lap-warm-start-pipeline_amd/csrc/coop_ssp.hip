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
// the path's start; jv_instance_kernel searches that ONE path (phase 3) and this kernel is launched
// again behind it (launch_solver enqueues a fixed number of such pairs; phase 2 finishes whatever is
// left).  Bails today: a minima collection with several ties of which one has an intermediate
// minimum, several final ties in one member or more than four in all, a tie inside the window, a
// SCAN list longer than the replicated queue.
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

constexpr int kK = 4;      // granules per member record (collection / path rounds)
constexpr int kKr = 2;     // ... of a relax round: ONE tie event per member and round.  Every granule load of a
                           // poll costs ~0.3 us per round (hop_bench: 32 members, 64 / 128 / 192 granules: 1.27 /
                           // 1.57 / 1.88 us), so the round that runs once per relax step polls as little as it can
constexpr int kEmax = 1;   // tie events a member publishes per relax round (2 granules each)
constexpr int kWin = 4;    // window: events applied per round = columns at order[hi .. hi+3] published per round
constexpr int kSlot = 6;   // granules per window slot: column | predecessor << 16, matched row + 1, distance, dual
constexpr int kWinGran = kSlot * kWin;
constexpr int kQ = 1024;   // replicated SCAN queue (entries: column, matched row)
constexpr int kFL = 256;   // foreign-column list of a member (prefetch hint only: overflow just loses hints)
constexpr unsigned kFlagBail = 1u, kFlagErr = 2u;

__device__ __forceinline__ void st_gran(unsigned long long *g, unsigned tag, unsigned val)
{
    __hip_atomic_store((gu64 *)g, ((unsigned long long)tag << 32) | val, LAPWARM_RLX_AGENT);
}
// Same granule, stored with WORKGROUP scope (sc0): the line stays in the XCD's L2 instead of being
// written through to memory, and the agent-scope (sc1) polls of members ON THE SAME XCD are served
// from that L2 (tools/micro/hop_bench.hip: 0.99 vs 1.22 us per 8-member round).  Never visible to
// another XCD -- only used after every member of the instance has reported the same XCC id.
__device__ __forceinline__ void st_gran_xcd(unsigned long long *g, unsigned tag, unsigned val)
{
    __hip_atomic_store((gu64 *)g, ((unsigned long long)tag << 32) | val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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

// diagnostic build (make coop-stamps): where a relax step's cycles go.  Never shipped or benchmarked.
#ifdef LAPWARM_COOP_STAMPS
__device__ __forceinline__ unsigned long long cstamp()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define CSTAMP(var) const unsigned long long var = cstamp()
#define CSTAMP_ADD(slot, t1, t0) stamps[slot] += (long long)((t1) - (t0))
#define CSTAMP_INC(slot) stamps[slot] += 1
#define CSTAMP_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define CSTAMP(var)
#define CSTAMP_ADD(slot, t1, t0)
#define CSTAMP_INC(slot)
#define CSTAMP_WAIT()
#endif

template <int CH, int NL>
struct Member {
    static constexpr int P = 64 * CH;  // positions per member
    const double *C;
    int n;
    double *v;
    int *x, *y, *pred;
    unsigned long long *mail;  // [2][NGtot]
    int G, g, lane, base, b0, NGm, NGtot;
    // geometry of the current round inside a buffer laid out [relax records: 2G][window][other records: 4G]:
    // granules per record, first granule of the record region, of the window.  A poll's index space is
    // "records, then window"; relax rounds are physically contiguous, the others take the window from the front
    int rk, rbase, wphys;
    __device__ __forceinline__ void round_kind(bool relax_round)
    {
        rk = relax_round ? kKr : kK;
        rbase = relax_round ? 0 : kKr * G + kWinGran;
        wphys = kKr * G;
        NGm = rk * G;
    }
    int2 *q;       // LDS: replicated SCAN queue
    unsigned gv[NL];  // payloads of the round just polled: granule i sits in lane i % 64 of gv[i / 64]
    int2 *lj;      // LDS: (column, matched row) of this member's positions during a collection
    unsigned *wb;  // LDS: [4][CH][kSlot] state of the (at most four) lanes that own positions hi .. hi+3
    int *fl;       // LDS: columns from other members' ranges that sit at positions of this member
    int nfl;       // (their cost entries are not covered by the range prefetch)
    unsigned dummy_lds;  // LDS byte address of a 1-KiB area the prefetch requests land in (never read)
    int jr[CH], yr[CH], pr[CH];  // column at the position, its matched row, its predecessor row
    double vr[CH], dk[CH];       // its dual, its tentative distance
    int lo, hi, ready, head_i, head_j;
    double level;
    unsigned seq;
    long long scan_elems, init_elems;
    int paths, finds, scan_steps;
    int err, bail_reason;
    bool same_xcd, allow_xcd_stores;
    int win_t, win_e;  // this lane's window slot / word when it stores a window granule (lanes 8 .. 31)
#ifdef LAPWARM_COOP_STAMPS
    long long stamps[10];
#endif

    // ---------------------------------------------------------------- exchange
    // ONE store instruction per round: lanes 0..3 carry this member's record, lanes 8..31 the window
    // granules this member is the writer of (the owner of position hi + t; the last member for slots
    // beyond position n-1).  Window values are staged in wb[] by the lanes that own the positions.
    // (Separate store blocks per position made hipcc drain vmcnt before each of them: one write-through
    // store latency per block, ~2,800 cycles per step -- profiles/r03_coop_stamps.txt.)
    __device__ __forceinline__ void publish(unsigned w0, unsigned w1, unsigned w2, unsigned w3, bool with_window)
    {
        unsigned w = w0;
        w = (lane == 1) ? w1 : w;
        w = (lane == 2) ? w2 : w;
        w = (lane == 3) ? w3 : w;
        bool mine = lane < rk;
        int idx = rbase + g * rk + lane;
        // (uniform: does this member write any window slot this round?  most members do not)
        const bool win_here = with_window && ((unsigned)(hi + kWin - 1 - base) < (unsigned)(P + kWin - 1) ||
                                              (g == G - 1 && hi + kWin > n));
        if (win_here) {
            const int wl = lane - 8;
            if ((unsigned)wl < (unsigned)kWinGran) {
                const int t = win_t, e = win_e;  // (lane - 8) / kSlot, (lane - 8) % kSlot: fixed per lane
                const int pos = hi + t;
                mine = (pos < n) ? ((unsigned)(pos - base) < (unsigned)P) : (g == G - 1);
                // staged by lane (pos - base) / CH as its position (pos - base) % CH; stage row 0 is lane L0
                const int lp = pos - base, L0 = (hi >= base) ? (hi - base) / CH : 0;
                const bool owned = (pos < n) && ((unsigned)lp < (unsigned)P);
                w = owned ? wb[((lp / CH - L0) * CH + lp % CH) * kSlot + e] : 0u;
                idx = wphys + wl;
            }
        }
        if (mine) {
            if (same_xcd)
                st_gran_xcd(mail + (size_t)(seq & 1u) * NGtot + idx, seq, w);
            else
                st_gran(mail + (size_t)(seq & 1u) * NGtot + idx, seq, w);
        }
    }
    // First round of the launch: where does every member run?  (HW_REG_XCC_ID, bits 3:0.)  Dispatch
    // deals workgroups to the XCDs round robin and the grid is laid out for it, but nothing promises
    // that: the cheaper same-XCD stores are used only when the hardware says so.
    __device__ __forceinline__ bool setup_round()
    {
        same_xcd = false;
        round_kind(false);
        const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;
        ++seq;
        publish(xcc | 0x100u, 0, 0, 0, false);
        if (!poll(NGm)) {
            err = 20;
            return false;
        }
        const unsigned w0 = member_word(0);  // (cross-lane read: every lane takes part)
        unsigned other = xcc | 0x100u;
        if (lane < G) other = w0;
        same_xcd = (__ballot(other != (xcc | 0x100u)) == 0ull) && allow_xcd_stores;
        return true;
    }
    // Waits until the first `need` granules of this round's buffer carry this round's tag and
    // leaves their payloads in gv[].  Bounded: ~2 s (a member that never arrives, e.g. because it
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
                if (qd * 64 < need) {  // (uniform: a relax round needs fewer loads than a collection round)
                    if (idx < need) {
                        const int phys = (idx < NGm) ? rbase + idx : wphys + (idx - NGm);
                        const unsigned long long w = ld_gran(buf + phys);
                        val[qd] = (unsigned)w;
                        ok &= (unsigned)(w >> 32) == seq;
                    }
                }
            }
            if (__all(ok)) {
#pragma unroll
                for (int qd = 0; qd < NL; ++qd) gv[qd] = val[qd];
                return true;
            }
            if ((++spins & 255u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) return false;
        }
    }
    // payload of granule idx (idx wave-uniform): a register read, no LDS round trip
    __device__ __forceinline__ unsigned rxu(int idx) const
    {
        const int i = uni(idx);
        if constexpr (NL == 1) {
            return (unsigned)__builtin_amdgcn_readlane((int)gv[0], i);
        } else if constexpr (NL == 2) {
            return (i < 64) ? (unsigned)__builtin_amdgcn_readlane((int)gv[0], i)
                            : (unsigned)__builtin_amdgcn_readlane((int)gv[1], i - 64);
        } else {
            if (i < 64) return (unsigned)__builtin_amdgcn_readlane((int)gv[0], i);
            if (i < 128) return (unsigned)__builtin_amdgcn_readlane((int)gv[1], i - 64);
            return (unsigned)__builtin_amdgcn_readlane((int)gv[NL - 1], i - 128);
        }
    }
    // the records fill granules 0 .. G*rk-1: one register while that is at most 64 granules (16 members of
    // a collection round), two beyond.  Wave-uniform, and decided from the member count, not from NL:
    // 17 .. 26 members poll two registers (NL = 2) as well, and their records do cross into the second.
    __device__ __forceinline__ bool wide_rec() const
    {
        if constexpr (NL < 2) return false;
        return NGm > kWave;
    }
    // lane m < G: word `slot` of member m's record
    __device__ __forceinline__ unsigned member_word(int slot) const
    {
        const int idx = lane * rk + slot;
        const unsigned a = (unsigned)__shfl((int)gv[0], idx & 63, kWave);
        if constexpr (NL >= 2) {
            const unsigned b = (unsigned)__shfl((int)gv[1], idx & 63, kWave);
            return (idx < 64) ? a : b;
        }
        return a;
    }

    // flags word of every member (granule `slot` of its record, bits `shift`..): any bail / error?
    __device__ __forceinline__ int check_flags(int slot, int shift)
    {
        // (looked at on the lanes that hold that granule: no cross-lane traffic)
        unsigned f = 0;
        if (wide_rec()) {
            const unsigned w = member_word(slot);
            if (lane < G) f = (w >> shift) & 3u;
        } else {
            if (lane < NGm && (lane & (rk - 1)) == slot) f = (gv[0] >> shift) & 3u;
        }
        if (__ballot((f & kFlagErr) != 0)) {
            if (!err) err = 21;  // another member reported an error
            return kRcErr;
        }
        if (__ballot((f & kFlagBail) != 0)) return kRcBail;
        return kRcGo;
    }

    // ---------------------------------------------------------------- row prefetch towards the XCD's L2
    // The next queued head's row is known one step ahead in most steps: request this member's piece
    // of it (its own column range + the foreign columns it has adopted) by LDS-DMA into a dummy
    // area -- no register is a destination, nothing ever waits for it on purpose, and the row gather
    // of the next step then hits the L2 instead of HBM.  Issued after this step's row has arrived:
    // vmcnt retires in order, an earlier request would sit in front of the loads the step waits for.
    __device__ __forceinline__ void prefetch_row(int row_i)
    {
        if ((unsigned)row_i >= (unsigned)n || (n & 1)) return;  // (16-byte requests: rows of an odd n are only 8-byte aligned)
        const double *row = C + (size_t)row_i * n;
#pragma unroll
        for (int qd = 0; qd < (CH + 1) / 2; ++qd) {
            int col = base + qd * 128 + lane * 2;
            if (CH == 1 && lane >= 32) col = base;  // (64 columns = 32 lanes' worth)
            col = (col < n - 2) ? col : n - 2;
            if (col < 0) col = 0;
            dma_request16(row + col, dummy_lds);
        }
        for (int e0 = 0; e0 < nfl; e0 += 64) {
            const int e = e0 + lane;
            int col = (e < nfl) ? fl[e] : base;
            col = (col < n - 2) ? col : n - 2;
            if (col < 0) col = 0;
            dma_request16(row + col, dummy_lds);
        }
    }
    // a column from another member's range now sits at one of this member's positions
    __device__ __forceinline__ void foreign_push(int col)
    {
        if ((unsigned)(col - base) >= (unsigned)P && nfl < kFL) {
            if (lane == 0) fl[nfl] = col;
            ++nfl;
        }
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
        nfl = 0;
    }

    // ---------------------------------------------------------------- minima collection (lapjv.cpp:153-171, :243-256)
    __device__ __forceinline__ int collect(int &target)
    {
        round_kind(false);
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
        publish(lo32(wtv), hi32(wtv), (unsigned)((wtp != 0x7fffffff) ? wtp : 0xffff) | ((unsigned)(my_own + 1) << 16),
                (unsigned)mj_own | ((unsigned)mp_own << 14) | ((err ? kFlagErr : 0u) << 28), false);
        if (!poll(NGm)) {
            err = 20;
            return kRcErr;
        }
        if (const int rc = check_flags(3, 28)) return rc;
        double mv = pos_inf();
        int mp = 0x7fffffff, mcol = 0, mrow = -1, mpred = 0;
        {
            const unsigned r0 = member_word(0), r1 = member_word(1), r2 = member_word(2), r3 = member_word(3);
            const unsigned rec[4] = {r0, r1, r2, r3};
            if (lane < G) {
            mv = mk_f64(rec[0], rec[1]);
            mp = (int)(rec[2] & 0xffffu);
            if (mp == 0xffff) mp = 0x7fffffff;
            mrow = (int)(rec[2] >> 16) - 1;
            mcol = (int)(rec[3] & 0x3fffu);
            mpred = (int)((rec[3] >> 14) & 0x3fffu);
            }
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
        unsigned sb = 0, tb = 0;
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
                        tb |= 1u << r;
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

        // ---- round B: tie events.  A tie with the FINAL minimum (d == totv: nothing later undercuts it)
        // is one more column for the SCAN list; the serial swaps of such ties (lapjv.cpp:165-167 behind
        // the last strict event) are exactly the swaps of a relax step's events with hi = lo + 1, so they
        // are exchanged and applied the same way: one event per member, the columns at order[lo+1 ..
        // lo+4] as they stand AFTER the shift in the window.  (Seeds that went through float32 produce
        // such a tie in ~0.3 % of the collections -- almost always exactly one.)  A tie with an
        // intermediate minimum, several ties in one member, more than kWin ties or a tie inside the
        // window end the cooperative search of this path (bail).
        unsigned tf = 0;
#pragma unroll
        for (int r = 0; r < CH; ++r) tf |= (((tb >> r) & 1u) && dk[r] == totv) ? (1u << r) : 0u;
        // (a single tie with an INTERMEDIATE minimum is handled too: it swaps order[t] with order[lo+1]
        // like a final tie would, but its column stays a TODO column at its own distance)
        int cntf = 0, cnte = 0;
        bool first_early = false;
        unsigned t0a = 0, t0b = 0, t0c = 0;
        {
            const unsigned long long any = __ballot(tb != 0);
            if (any) {
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    cntf += __popcll(__ballot((tf >> r) & 1u));
                    cnte += __popcll(__ballot(((tb & ~tf) >> r) & 1u));
                }
                const int l = __builtin_ctzll(any);
                const unsigned em = (unsigned)__builtin_amdgcn_readlane((int)tb, l);
                const int r0 = __builtin_ctz(em);
                first_early = !(((unsigned)__builtin_amdgcn_readlane((int)tf, l) >> r0) & 1u);
                int sj = jr[0], sy = yr[0], sp = pr[0];
#pragma unroll
                for (int qd = 1; qd < CH; ++qd) {
                    if (r0 == qd) {
                        sj = jr[qd];
                        sy = yr[qd];
                        sp = pr[qd];
                    }
                }
                t0a = (unsigned)(base + l * CH + r0) | ((unsigned)__builtin_amdgcn_readlane(sj, l) << 16);
                t0b = (unsigned)(__builtin_amdgcn_readlane(sy, l) + 1);
                t0c = (unsigned)__builtin_amdgcn_readlane(sp, l);
            }
        }
        hi = lo + 1;  // (the window below is order[hi .. hi+3])
        if ((unsigned)(hi + kWin - 1 - base) < (unsigned)(P + kWin - 1)) {
            const int L0 = (hi >= base) ? (hi - base) / CH : 0;
            const int row = lane - L0;
            if ((unsigned)row < 4u) {
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    unsigned *d = wb + (row * CH + r) * kSlot;
                    const bool sh = (sb >> r) & 1u;  // a strict event position: what the shift will leave there
                    const double dv = sh ? prevval[r] : dk[r];
                    d[0] = (unsigned)(sh ? pc[r] : jr[r]) | ((unsigned)(sh ? pq[r] : pr[r]) << 16);
                    d[1] = (unsigned)((sh ? py[r] : yr[r]) + 1);
                    d[2] = lo32(dv);
                    d[3] = hi32(dv);
                    d[4] = 0u;  // (the dual is not known for a shifted column yet: the adopter loads it)
                    d[5] = 0u;
                }
            }
        }
        ++seq;
        publish((first_early ? 4u : 0u) | (err ? kFlagErr : 0u) | ((unsigned)(cntf > 255 ? 255 : cntf) << 8) |
                    ((unsigned)(cnte > 255 ? 255 : cnte) << 16),
                t0a, t0b, t0c, true);
        if (!poll(NGm + kWinGran)) {
            err = 20;
            return kRcErr;
        }
        if (const int rc = check_flags(0, 0)) return rc;
        int ntie = 0;
        bool early1 = false;
        int tp_[kWin] = {0, 0, 0, 0}, tj_[kWin] = {0, 0, 0, 0}, ty_[kWin] = {0, 0, 0, 0}, tq_[kWin] = {0, 0, 0, 0};
        {
            unsigned w0 = 0;
            const bool wide = wide_rec();
            if (wide) {
                const unsigned w = member_word(0);
                if (lane < G) w0 = w;
            } else {
                if (lane < NGm && (lane & (kK - 1)) == 0) w0 = gv[0];
            }
            const int cmf = (int)((w0 >> 8) & 0xffu), cme = (int)((w0 >> 16) & 0xffu);
            const int cm = cmf + cme;
            const int early_total = wave_sum_i32(cme);
            if (early_total > 0) {
                // exactly one tie in the whole collection, and it is the early one: handled below
                if (early_total != 1 || __ballot(cmf > 0)) {
                    bail_reason = 1;  // ties with an intermediate minimum beyond the single-tie case
                    return kRcBail;
                }
                early1 = true;
            }
            unsigned long long mm = __ballot(cm > 0);
            if (__ballot(cm > 1) || __popcll(mm) > kWin) {
                bail_reason = 3;  // more final ties than one round carries
                return kRcBail;
            }
            while (mm) {
                const int ml = __builtin_ctzll(mm);
                mm &= mm - 1;
                const int gb = wide ? ml * kK : ml;  // granule 0 of that member's record
                const unsigned wa = rxu(gb + 1), wy = rxu(gb + 2), wq = rxu(gb + 3);
                const int pp_ = (int)(wa & 0xffffu), jj_ = (int)(wa >> 16), yy_ = (int)wy - 1, qq_ = (int)wq;
                if (ntie == 0) {
                    tp_[0] = pp_, tj_[0] = jj_, ty_[0] = yy_, tq_[0] = qq_;
                } else if (ntie == 1) {
                    tp_[1] = pp_, tj_[1] = jj_, ty_[1] = yy_, tq_[1] = qq_;
                } else if (ntie == 2) {
                    tp_[2] = pp_, tj_[2] = jj_, ty_[2] = yy_, tq_[2] = qq_;
                } else {
                    tp_[3] = pp_, tj_[3] = jj_, ty_[3] = yy_, tq_[3] = qq_;
                }
                ++ntie;
            }
            if (early1 && tp_[0] == lo + 1) {
                early1 = false;  // order[lo+1] swapped with itself
                ntie = 0;
            }
            if (ntie > 0) {
                bool bad = tp_[0] <= lo + ntie;  // a tie inside the window [lo+1, lo+ntie]: the swaps are not independent
#pragma unroll
                for (int t = 0; t < kWin; ++t) {
                    if (t < ntie)
                        bad = bad || (unsigned)tp_[t] >= (unsigned)n || (unsigned)tj_[t] >= (unsigned)n || ty_[t] >= n ||
                              (unsigned)tq_[t] >= (unsigned)n;
                }
                if (bad) {
                    bail_reason = 3;
                    return kRcBail;
                }
            }
        }
        level = totv;
        head_j = min_col;
        head_i = min_row;
        // the leader keeps the predecessor of every column that joins the SCAN list (final from here
        // on): the backtrack only ever follows those
        if (g == 0 && lane == 0) st_i32(pred + min_col, min_pred);
        // the LAST free column of the SCAN list ends the path (lapjv.cpp:250-255)
        {
            int tg = (head_i < 0) ? head_j : -1, tgq = min_pred;
#pragma unroll
            for (int t = 0; t < kWin; ++t) {
                if (!early1 && t < ntie && ty_[t] < 0) {
                    tg = tj_[t];
                    tgq = tq_[t];
                }
            }
            if (tg >= 0) {
                if (g == 0 && lane == 0) st_i32(pred + tg, tgq);
                target = tg;
                return kRcTarget;
            }
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
        // (only the first strict event of a member can receive a column from an earlier member: pcol)
        {
            bool got_foreign = false;
#pragma unroll
            for (int r = 0; r < CH; ++r) got_foreign |= ((sb >> r) & 1u) && prevpos[r] < base;
            if (__ballot(got_foreign)) foreign_push(pcol);
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
        if (early1) {
            // the single early tie: order[tp] <-> order[lo+1]; the tie column stays a TODO column with its own
            // distance, which only its owner knows: one more round carries it
            const unsigned w0 = rxu(NGm);
            const int a = (int)(w0 & 0xffffu), qa = (int)(w0 >> 16), ya = (int)rxu(NGm + 1) - 1;
            const double da = mk_f64(rxu(NGm + 2), rxu(NGm + 3));
            if ((unsigned)a >= (unsigned)n || ya >= n || (unsigned)qa >= (unsigned)n) {
                err = 8;
                return kRcErr;
            }
            double dt_own = 0.0;
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                if (b0 + r == tp_[0]) dt_own = dk[r];
            }
            const int own_lane = uni((tp_[0] - base) / CH);
            const bool own_here = (unsigned)(tp_[0] - base) < (unsigned)P;
            const double dt_pub = own_here ? readlane_f64(dt_own, own_here ? own_lane : 0) : 0.0;
            ++seq;
            publish(lo32(dt_pub), hi32(dt_pub), 0, (err ? kFlagErr : 0u) << 28, false);
            if (!poll(NGm)) {
                err = 20;
                return kRcErr;
            }
            if (const int rc = check_flags(3, 28)) return rc;
            const int om = tp_[0] / P;  // the member that owns the tie's position
            const double d_tie = mk_f64(rxu(om * kK), rxu(om * kK + 1));
            if ((unsigned)(tp_[0] - base) < (unsigned)P) {
                foreign_push(a);
                const double va = ld_f64(v + a);
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    if (b0 + r == tp_[0]) {
                        jr[r] = a;
                        yr[r] = ya;
                        pr[r] = qa;
                        dk[r] = da;
                        vr[r] = va;
                    }
                }
            }
            if ((unsigned)(lo + 1 - base) < (unsigned)P) {
                foreign_push(tj_[0]);
                const double vt = ld_f64(v + tj_[0]);
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    if (b0 + r == lo + 1) {
                        jr[r] = tj_[0];
                        yr[r] = ty_[0];
                        pr[r] = tq_[0];
                        dk[r] = d_tie;
                        vr[r] = vt;
                    }
                }
            }
            hi = lo + 1;
            return kRcGo;
        }
        // final ties: tie s swaps order[its position] with order[lo + 1 + s] (independent: no tie sits
        // inside [lo+1, lo+ntie]); its column joins the SCAN list at the level, with the predecessor it had
#pragma unroll
        for (int t = 0; t < kWin; ++t) {
            if (t < ntie) {
                const unsigned w0 = rxu(NGm + kSlot * t);
                const int a = (int)(w0 & 0xffffu), qa = (int)(w0 >> 16), ya = (int)rxu(NGm + kSlot * t + 1) - 1;
                const double da = mk_f64(rxu(NGm + kSlot * t + 2), rxu(NGm + kSlot * t + 3));
                if ((unsigned)a >= (unsigned)n || ya >= n || (unsigned)qa >= (unsigned)n) {
                    err = 8;
                    return kRcErr;
                }
                if ((unsigned)(tp_[t] - base) < (unsigned)P) {
                    foreign_push(a);
                    const double va = ld_f64(v + a);
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
                        if (b0 + r == tp_[t]) {
                            jr[r] = a;
                            yr[r] = ya;
                            pr[r] = qa;
                            dk[r] = da;
                            vr[r] = va;
                        }
                    }
                }
                if ((unsigned)(lo + 1 + t - base) < (unsigned)P) {
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
                        if (b0 + r == lo + 1 + t) {
                            jr[r] = tj_[t];
                            yr[r] = ty_[t];
                            pr[r] = tq_[t];
                            dk[r] = totv;
                        }
                    }
                }
                if (lane == 0) {
                    q[(lo + 1 + t) & (kQ - 1)] = make_int2(tj_[t], ty_[t]);
                    if (g == 0) st_i32(pred + tj_[t], tq_[t]);
                }
            }
        }
        hi = lo + 1 + ntie;
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
        CSTAMP(ts0);
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            // SCAN / READY positions are not relaxed: point them at the head's own entry (one line every
            // lane reads anyway) instead of a column whose line nobody prefetched
            const int k = b0 + r;
            // (jr[] only ever holds columns that were range-checked when they arrived: path start, shift,
            // adoption from a validated record)
            const int jc = ((k >= hi) & (k < n)) ? jr[r] : head_j;
            c[r] = row[jc];
        }
        const double c_head = row[head_j];
        const double v_head = ld_f64(v + head_j);
        scan_steps++;
        scan_elems += (long long)(n - hi);
#pragma unroll
        for (int r = 0; r < CH; ++r) pin(c[r]);
        CSTAMP_WAIT();
        CSTAMP(ts1);
        CSTAMP_ADD(0, ts1, ts0);
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
        if (lo + 1 < hi) prefetch_row(uni(q[(lo + 1) & (kQ - 1)].y));
        for (int round = 0;; ++round) {
            // ---- this member's pending tie events, in position order
            CSTAMP(tr0);
            int cnt = 0;
            unsigned e0a = 0, e0b = 0;
            unsigned long long any = __ballot(evm != 0);  // (94% of the member-rounds: none)
            if (any) {
#pragma unroll
                for (int r = 0; r < CH; ++r) cnt += __popcll(__ballot((evm >> r) & 1u));
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
                        e0a = wa;
                        e0b = yv;
                        ++emitted;
                    }
                }
            }
            e0b |= (unsigned)(cnt > 255 ? 255 : cnt) << 20;
            e0b |= (err ? kFlagErr : 0u) << 28;
            ++seq;
            round_kind(true);
            CSTAMP(tr1);
            CSTAMP_ADD(1, tr1, tr0);
            // ---- the columns at order[hi .. hi+3] (what this round's events displace), with their
            // predecessors, matched rows, distances and duals as they stand after this step's update:
            // the one or two lanes that own those positions stage ALL their positions (one exec-masked
            // block, no per-position branches); the storing lanes pick the four slots out of that
            if ((unsigned)(hi + kWin - 1 - base) < (unsigned)(P + kWin - 1)) {
                const int L0 = (hi >= base) ? (hi - base) / CH : 0;  // first lane of this member inside the window
                const int row = lane - L0;
                if ((unsigned)row < 4u) {
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
                        unsigned *d = wb + (row * CH + r) * kSlot;
                        d[0] = (unsigned)jr[r] | ((unsigned)pr[r] << 16);
                        d[1] = (unsigned)(yr[r] + 1);
                        d[2] = lo32(dk[r]);
                        d[3] = hi32(dk[r]);
                        d[4] = lo32(vr[r]);
                        d[5] = hi32(vr[r]);
                    }
                }
            }
            publish(e0a, e0b, 0, 0, true);
            CSTAMP(tr2);
            CSTAMP_ADD(2, tr2, tr1);
            if (!poll(NGm + kWinGran)) {
                err = 20;
                return kRcErr;
            }
            CSTAMP(tr3);
            CSTAMP_ADD(3, tr3, tr2);
            CSTAMP_INC(8);
            // event counts and flags, on the lanes that hold word 1 of a member record
            int cm = 0;
            {
                unsigned w1 = 0;  // (2 granules per member: up to 32 members sit in gv[0])
                if (lane < NGm && (lane & 1) == 1) w1 = gv[0];
                if (__ballot((w1 >> 28) & 3u)) {  // rare: somebody reports an error or asks to stop
                    if (const int rc = check_flags(1, 28)) return rc;
                }
                cm = (int)((w1 >> 20) & 0xffu);
            }
            const unsigned long long evl = __ballot(cm > 0);
            if (evl == 0ull) {
                CSTAMP(tr4);
                CSTAMP_ADD(4, tr4, tr3);
                break;
            }
            const int first_l = __builtin_ctzll(evl);
            if ((evl & (evl - 1)) == 0ull && __builtin_amdgcn_readlane(cm, first_l) == 1) {
                // ---- exactly one tie event in the whole step (60% of the steps that have any)
                const int gb = first_l - 1;  // first granule of that member's record
                const unsigned wa = rxu(gb), wbv = rxu(gb + 1);
                const int ep = (int)(wa & 0xffffu), ej = (int)(wa >> 16), ey = (int)(wbv & 0xfffffu) - 1;
                const unsigned w0 = rxu(NGm);
                const int a = (int)(w0 & 0xffffu), qa = (int)(w0 >> 16), ya = (int)rxu(NGm + 1) - 1;
                const int t = ep - hi;
                if ((unsigned)ep >= (unsigned)n || (unsigned)ej >= (unsigned)n || ey >= n || t < 0 ||
                    (unsigned)a >= (unsigned)n || ya >= n || (unsigned)qa >= (unsigned)n) {
                    err = 8;
                    return kRcErr;
                }
                if (ey < 0) {
                    target = ej;
                    if (g == 0 && lane == 0) st_i32(pred + ej, head_i);
                    return kRcTarget;
                }
                if (t != 0 && (unsigned)(ep - base) < (unsigned)P) {
                    foreign_push(a);
                    const double da = mk_f64(rxu(NGm + 2), rxu(NGm + 3)), va = mk_f64(rxu(NGm + 4), rxu(NGm + 5));
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
                        if (b0 + r == ep) {
                            jr[r] = a;
                            yr[r] = ya;
                            pr[r] = qa;
                            dk[r] = da;
                            vr[r] = va;
                        }
                    }
                }
                if ((unsigned)(hi - base) < (unsigned)P) {
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
                        if (b0 + r == hi) {
                            jr[r] = ej;
                            yr[r] = ey;
                            pr[r] = head_i;
                            dk[r] = level;
                        }
                    }
                }
                if (lane == 0) {
                    q[hi & (kQ - 1)] = make_int2(ej, ey);
                    if (g == 0) st_i32(pred + ej, head_i);
                }
                ++hi;
                if (hi - lo >= kQ) {
                    bail_reason = 2;
                    return kRcBail;
                }
                CSTAMP(tr6);
                CSTAMP_ADD(6, tr6, tr3);
                break;
            }
            const int total = wave_sum_i32(cm);
            // ---- the first events of the round in global position order = member order
            int take = 0;
            int ep0 = 0, ep1 = 0, ep2 = 0, ep3 = 0, ej0 = 0, ej1 = 0, ej2 = 0, ej3 = 0, ey0 = 0, ey1 = 0, ey2 = 0, ey3 = 0;
            {
                unsigned long long mm = evl;
                bool blocked = false;
                while (mm && take < kWin && !blocked) {
                    const int ml = __builtin_ctzll(mm);  // lane holding word 1 of that member's record
                    mm &= mm - 1;
                    const int m = ml >> 1;
                    const int c_m = __builtin_amdgcn_readlane(cm, ml);
                    const int pub = (c_m < kEmax) ? c_m : kEmax;
                    for (int e = 0; e < pub && take < kWin; ++e) {
                        const unsigned wa = rxu(m * kKr + 2 * e);
                        const unsigned wbv = rxu(m * kKr + 2 * e + 1);
                        const int pp_ = (int)(wa & 0xffffu), jj_ = (int)(wa >> 16), yy_ = (int)(wbv & 0xfffffu) - 1;
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
            double wd_[kWin], wv_[kWin];
#pragma unroll
            for (int t = 0; t < kWin; ++t) {
                const unsigned w0 = rxu(NGm + kSlot * t);
                wa_[t] = (int)(w0 & 0xffffu);
                wq_[t] = (int)(w0 >> 16);
                wy_[t] = (int)rxu(NGm + kSlot * t + 1) - 1;
                wd_[t] = mk_f64(rxu(NGm + kSlot * t + 2), rxu(NGm + kSlot * t + 3));
                wv_[t] = mk_f64(rxu(NGm + kSlot * t + 4), rxu(NGm + kSlot * t + 5));
            }
            const int eps[kWin] = {ep0, ep1, ep2, ep3};
            const int ejs[kWin] = {ej0, ej1, ej2, ej3};
            const int eys[kWin] = {ey0, ey1, ey2, ey3};
#pragma unroll
            for (int s = 0; s < kWin; ++s) {
                if (s < take) {
                    const int a = wa_[s], ya = wy_[s], qa = wq_[s];
                    const double da = wd_[s], va = wv_[s];
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
                            wv_[u] = va;
                            moved_inside = true;
                        }
                    }
                    if (!moved_inside && t != s && (unsigned)(eps[s] - base) < (unsigned)P) {
                        // the event position adopts the column displaced from order[hi + s]
                        foreign_push(a);
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
            CSTAMP(tr5);
            CSTAMP_ADD(5, tr5, tr3);
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
        round_kind(false);
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
        publish(err ? kFlagErr : 0u, 0, 0, 0, false);
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
        publish(err ? kFlagErr : 0u, 0, 0, 0, false);
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
    __shared__ int2 lj_s[64 * CH];
    __shared__ unsigned wb_s[4 * CH * kSlot];
    __shared__ int fl_s[kFL];
    __shared__ __attribute__((aligned(16))) unsigned char dummy_s[1024];
    // members of instance b sit at block indices with the same value modulo 8: workgroups are dealt
    // to the 8 XCDs round robin, so they share an XCD (speed only -- nothing depends on it)
    const int G = p.G;
    const int grp = (int)blockIdx.x / (8 * G), rem = (int)blockIdx.x % (8 * G);
    const int g = rem / 8;
    const int b = p.first + grp * 8 + rem % 8;
    if (b >= p.first + p.count || b >= p.batch) return;
    int *hand = p.hand + (size_t)b * kHandInts;
    const int nf = hand[0];
    const int f_first = hand[1];  // > 0: relaunched behind a path that jv_instance_kernel searched
    if (nf <= 0 || hand[2] != 0 || hand[4] != 0 || hand[3] != 0 || f_first >= nf) return;
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
    m.NGtot = G * (kK + kKr) + kWinGran;
    m.round_kind(false);
    m.mail = p.mail + (size_t)b * 2 * m.NGtot;
    m.q = q_s;
    m.lj = lj_s;
    m.wb = wb_s;
    m.fl = fl_s;
    m.nfl = 0;
    m.dummy_lds = lds_address(dummy_s);
    m.seq = 0;
    m.scan_elems = m.init_elems = 0;
    m.paths = m.finds = m.scan_steps = 0;
    m.err = 0;
    m.bail_reason = 0;
#ifdef LAPWARM_COOP_STAMPS
    for (int qd = 0; qd < 10; ++qd) m.stamps[qd] = 0;
#endif
    m.level = 0.0;
    m.lo = m.hi = m.ready = m.head_i = m.head_j = 0;
    const int *fr = p.fr + o;

    m.allow_xcd_stores = p.xcd_stores != 0;
    m.win_t = ((int)threadIdx.x - 8) / kSlot;
    m.win_e = ((int)threadIdx.x - 8) % kSlot;
    int done = f_first;
    m.setup_round();
    for (int f = f_first; f < nf && !m.err; ++f) {
        const long long s_scan = m.scan_elems, s_init = m.init_elems;
        const int s_paths = m.paths, s_finds = m.finds, s_steps = m.scan_steps;
        const int start = uni(fr[f]);
        if ((unsigned)start >= (unsigned)n) {
            m.err = 2;
            break;
        }
        {
            CSTAMP(tpa);
            m.path_init(start);
            CSTAMP_WAIT();
            CSTAMP(tpb);
#ifdef LAPWARM_COOP_STAMPS
            m.stamps[9] += (long long)(tpb - tpa);
#endif
        }
        int target = -1, rc = kRcGo;
        for (int guard = 0;; ++guard) {
            if (m.lo == m.hi) {
                CSTAMP(tca);
                rc = m.collect(target);
                CSTAMP(tcb);
#ifdef LAPWARM_COOP_STAMPS
                m.stamps[7] += (long long)(tcb - tca);
#endif
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
        {
            CSTAMP(tpa);
            const int prc = m.path_end(target, start);
            CSTAMP(tpb);
#ifdef LAPWARM_COOP_STAMPS
            m.stamps[9] += (long long)(tpb - tpa);
#endif
            if (prc) break;
        }
        done = f + 1;
    }
    if (g == 0 && m.lane == 0) {
        hand[1] = done;
        hand[2] = m.err;
        hand[3] = m.bail_reason;
        long long *cs = p.cstats + (size_t)b * kCoopStats;
        cs[0] += m.paths;
        cs[1] += m.finds;
        cs[2] += m.scan_steps;
        cs[3] += m.scan_elems;
        cs[4] += m.init_elems;
        cs[5] = ((cs[5] & 0xffffffffffll) + (long long)m.seq) | (m.same_xcd ? (1ll << 40) : 0);
#ifdef LAPWARM_COOP_STAMPS
        for (int qd = 0; qd < 6; ++qd) cs[6 + qd] += m.stamps[qd];
#endif
    }
#ifdef LAPWARM_COOP_STAMPS
    // imbalance between the members of an instance: smallest / largest total of the poll segment and of
    // everything else in a relax round (slots 12 .. 15; reset by phase 1)
    if (m.lane == 0) {
        unsigned long long *cs = reinterpret_cast<unsigned long long *>(p.cstats + (size_t)b * kCoopStats);
        const unsigned long long poll_c = (unsigned long long)m.stamps[3];
        const unsigned long long work_c = (unsigned long long)(m.stamps[0] + m.stamps[1] + m.stamps[2] + m.stamps[4] +
                                                                m.stamps[5] + m.stamps[6]);
        atomicMax(&cs[12], poll_c);
        atomicMax(&cs[13], work_c);
        atomicMax(&cs[14], ~poll_c);  // (max of the complement = min)
        atomicMax(&cs[15], ~work_c);
    }
#endif
    if (g != 0 && m.err && m.lane == 0) {
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
        if ((n + 64 * forced - 1) / (64 * forced) <= 32) return forced;
    }
    // Measured (tools/micro/hop_bench.hip, profiles/r03_hop_bench.txt): an exchange among 8 members costs
    // 1.0-1.3 us, among 16 1.4-2.2 us (32 instances in flight), among 32 1.9-2.6 us -- the fewer members the
    // better, as long as a lane's positions fit the register file (16 positions = ~250 VGPRs).
    if (n <= 512) return 1;
    if (n <= 1024) return 2;
    // Measured with one event per member in a 2-granule relax record (a relax round polls 2G + 24 granules:
    // one load per lane up to 20 members, two up to 32): n = 4608 x 8 467 ms with 4 positions per lane
    // (18 members) against 508 ms with 8; n = 8192 1.13 s with 4 (32 members), 1.21 s with 8, 1.59 s with 16;
    // n = 16384 3.88 s with 8 (32 members), 4.75 s with 16.
    if (n <= 8192) return 4;
    return 8;
}

int coop_members(int n)
{
    const int ch = coop_ch(n);
    return (n + 64 * ch - 1) / (64 * ch);
}

size_t coop_mail_granules(int n) { return 2 * ((size_t)coop_members(n) * (kK + kKr) + kWinGran); }

bool coop_enabled(int n)
{
    static const int min_n = [] {
        const char *e = getenv("LAPWARM_COOP_MIN_N");
        // default: sizes whose solver state no longer fits one CU's LDS (solver_lds_level 0); below that the
        // single-workgroup kernel is faster (n = 4096: 274 ms against 442 ms per 32 instances)
        return e ? atoi(e) : 4428;
    }();
    static const int on = [] {
        const char *e = getenv("LAPWARM_COOP");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    const int ch = coop_ch(n);
    return on && n >= min_n && n <= 16384 && (ch == 1 || ch == 2 || ch == 4 || ch == 8 || ch == 16) &&
           coop_members(n) <= 32;
}

hipError_t launch_coop(const CoopParams &p_in, hipStream_t stream)
{
    CoopParams p = p_in;
    const int ch = coop_ch(p.n);
    p.G = coop_members(p.n);
    const int ng = p.G * kK + kWinGran;  // the largest poll: a collection's round B
    static const int xcd_stores = [] {
        const char *e = getenv("LAPWARM_COOP_XCD_STORES");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    p.xcd_stores = xcd_stores;
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
        // 17 .. 32 members (three granule loads per lane): n = 8192 with 4, n = 16384 with 8 positions per lane
        if (nl == 3 && ch == 4) e = launch_cfg<4, 3>(p, grid, stream);
        if (nl == 3 && ch == 8) e = launch_cfg<8, 3>(p, grid, stream);
#undef LAPWARM_COOP_CASE
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace lapwarm
