// cols_search.hpp -- the shortest-augmenting-path search of the per-instance solver in its
// COLUMN-OWNED form (solver LDS levels 3 and 4, n <= ~2400).
//
// Reference semantics reproduced bit for bit (paths relative to /root/reference):
//   find_path_dense / _find_dense / _scan_dense      LAP/_lapjv_cpp/lapjv.cpp:153-282
//
// Ownership.  Thread t owns COLUMNS for the whole path and keeps their labels in registers:
// distance d, dual v, predecessor, matched row, TODO flag and the column's POSITION in the
// reference's column order (cols[] of lapjv.cpp).  Labels never move between threads.  The
// reference's order-dependent swaps only ever compare / exchange positions, so the order is
// carried by the position labels alone; LDS holds no permutation while relax steps run:
//   * relax step (lapjv.cpp:185-207): coalesced row load (requested one step ahead when the SCAN
//     list already holds the next head), two subtractions and two compares per column.  The
//     columns sitting at positions hi..hi+31 publish themselves before the step's barrier: they
//     are what the step's tie events displace.  A tie event appends a record (column, position,
//     matched row, dual) and -- speculatively, right if it is the step's only event -- the
//     SCAN-list entry at hi.
//       - no event / one event (77% of the steps): ONE barrier.  The column at position hi
//         takes the event's position (a register compare), everybody reads the next head and
//         the one after it from the SCAN list.
//       - 2..32 events: wave 0 ranks the records by position and applies the swaps on scalars
//         (lapjv.cpp:199-205 order), publishes the label moves; second barrier.
//       - more: position bitmap + ordered replay on an order[] rebuilt from the labels.
//   * minima collection (lapjv.cpp:153-171): the TODO columns scatter (distance, column) into
//     position order, the positions are scanned with an exclusive (value, position) prefix
//     minimum, and -- without ties -- the swap sequence collapses to a cyclic shift along the
//     strict events, published as label moves (no permutation is written back).  Ties take the
//     exact ordered replay (replay_find), after which the labels are re-read.
// Synchronisation argument: DESIGN.md section 4, "happens-before table".
#pragma once

#include "device_utils.hpp"

namespace lapwarm {
namespace cols {

#ifdef LAPWARM_STAMPS
// Diagnostic builds only: cycle stamps of one thread (LAPWARM_STAMP_TID, default 0), accumulated
// in registers with STATIC slot numbers (a dynamic index would push the array to scratch) and
// added to Ctl::stamps at the end of each path.  s_memtime returns through the LGKM counter and
// out of order with LDS reads: the wait must be part of the same asm statement.
__device__ __forceinline__ unsigned long long cstamp_now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define CSTAMP(var) const unsigned long long var = cstamp_now()
#define CSTAMP_ADD(slot, t1, t0) cst[slot] += (long long)((t1) - (t0))
#define CSTAMP_INC(slot) cst[slot] += 1
#else
#define CSTAMP(var)
#define CSTAMP_ADD(slot, t1, t0)
#define CSTAMP_INC(slot)
#endif
#ifndef LAPWARM_STAMP_TID
#define LAPWARM_STAMP_TID 0
#endif

constexpr int kRecCap = 32;  // tie events of one relax step that are replayed from records

// One entry of the SCAN list: the column, its matched row and its dual -- everything the relax
// step needs about its head, in one 16-byte LDS read.
struct alignas(16) QDesc {
    int j, i;
    double v;
};

// Record of one tie event of a relax step.
struct alignas(16) Rec {
    int j, p, i, pad;  // event column, its position, its matched row (-1: free)
    double v;          // its dual
    double pad2;
};

struct alignas(16) Ctl {
    Rec rec[2][kRecCap];    // per step parity, slot = arrival order
    alignas(16) int res[4];  // outcome of a multi-event replay: hi, target, label moves, error
    QDesc resq[3];           // ... and the SCAN-list entries at lo+1 .. lo+3 after it
    int apub[kRecCap];       // multi-event steps: the columns at positions hi..hi+31
    alignas(16) int mv_a[kRecCap];  // label moves (column, new position) of a multi-event replay
    alignas(16) int mv_p[kRecCap];
    int ev_total[2];        // monotonic, per step parity: +1 per event, +0x10000 per FREE event
    int free_pos[2];        // bitmap path only: smallest position of a free event column
    int tie_find;           // sequence number of the last minima collection that saw a tie
    int hi, target, head_j, head_i;  // outputs of the replays
    int nmoves;             // label moves published by the multi-event replay
    int err;
    // search state that survives from one path to the next (thread-uniform registers in between)
    int step_id, seen0, seen1, find_seq;
    int paths, finds, scan_steps;
    long long scan_elems, init_elems;
    long long stamps[16];   // diagnostic builds (-DLAPWARM_STAMPS)
};

// Everything the search needs about its LDS arrays, as typed pointers (see make_ctx).
struct Ctx {
    const double *C;
    int n, W, Wpad;
    // LDS arrays
    double *dist;   // minima collection: distances in POSITION order
    double *v;
    int *order;     // minima collection: columns in position order (rebuilt from the labels)
    int *pos;       // column -> position, valid while labels are being re-read
    int *pred, *y;
    QDesc *qdesc;
    uint32_t *evt, *sbits, *evb;
    int *evl, *tmpcol;
    Ctl *ctl;
    BlockExchange *ex;
    unsigned char *slots;  // DMA variant
    int slot_bytes;
};

// What crosses the call boundary of search_path: the matrix pointer, the sizes and the BYTE
// OFFSETS of the arrays inside the workgroup's dynamic LDS block.  Passing LDS pointers through a
// non-inlined call would turn them into generic pointers (flat_* instead of ds_* instructions);
// the callee rebuilds typed LDS pointers from the offsets and its own view of the LDS block.
struct Layout {
    const double *C;
    int n, W, Wpad;
    int dist, v, order, pos, pred, y, qdesc, evt, sbits, evb, evl, tmpcol, ctl, ex;
    int slots, slot_bytes;  // DMA variant: two row slots (direct-to-LDS row requests)
};

__device__ __forceinline__ Ctx make_ctx(const Layout &l_in)
{
    extern __shared__ __align__(16) unsigned char cols_smem[];
    unsigned char *base = cols_smem;
    Ctx c;
    // every field is workgroup-uniform: say so (arguments of a real call arrive in vector registers)
    {
        const unsigned long long a = reinterpret_cast<unsigned long long>(l_in.C);
        const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(a & 0xffffffffull));
        const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32));
        // rebuilt as a GLOBAL pointer (address space 1): a plain integer-to-pointer cast would
        // make it generic and turn every row load into a flat_load
        typedef const double __attribute__((address_space(1))) *global_cptr;
        c.C = (const double *)(global_cptr)(((unsigned long long)hi32 << 32) | lo32);
    }
    c.n = uni(l_in.n);
    c.W = uni(l_in.W);
    c.Wpad = uni(l_in.Wpad);
    c.dist = reinterpret_cast<double *>(base + uni(l_in.dist));
    c.v = reinterpret_cast<double *>(base + uni(l_in.v));
    c.order = reinterpret_cast<int *>(base + uni(l_in.order));
    c.pos = reinterpret_cast<int *>(base + uni(l_in.pos));
    c.pred = reinterpret_cast<int *>(base + uni(l_in.pred));
    c.y = reinterpret_cast<int *>(base + uni(l_in.y));
    c.qdesc = reinterpret_cast<QDesc *>(base + uni(l_in.qdesc));
    c.evt = reinterpret_cast<uint32_t *>(base + uni(l_in.evt));
    c.sbits = reinterpret_cast<uint32_t *>(base + uni(l_in.sbits));
    c.evb = reinterpret_cast<uint32_t *>(base + uni(l_in.evb));
    c.evl = reinterpret_cast<int *>(base + uni(l_in.evl));
    c.tmpcol = reinterpret_cast<int *>(base + uni(l_in.tmpcol));
    c.ctl = reinterpret_cast<Ctl *>(base + uni(l_in.ctl));
    c.ex = reinterpret_cast<BlockExchange *>(base + uni(l_in.ex));
    c.slots = base + uni(l_in.slots);
    c.slot_bytes = uni(l_in.slot_bytes);
    return c;
}

__device__ __forceinline__ void ctl_init(Ctl *c)
{
    // thread 0 only, before the first barrier of the kernel
    for (int q = 0; q < 2 * kRecCap; ++q) {
        Rec &r = c->rec[q / kRecCap][q % kRecCap];
        r.j = r.p = r.i = r.pad = 0;
        r.v = r.pad2 = 0.0;
    }
    for (int q = 0; q < kRecCap; ++q) c->apub[q] = 0;
    for (int q = 0; q < 4; ++q) c->res[q] = 0;
    for (int q = 0; q < 3; ++q) {
        c->resq[q].j = c->resq[q].i = 0;
        c->resq[q].v = 0.0;
    }
    for (int q = 0; q < kRecCap; ++q) c->mv_a[q] = c->mv_p[q] = 0;
    c->ev_total[0] = c->ev_total[1] = 0;
    c->free_pos[0] = c->free_pos[1] = 0x7fffffff;
    c->tie_find = 0;
    c->hi = 0;
    c->target = -1;
    c->head_j = c->head_i = 0;
    c->nmoves = 0;
    c->err = 0;
    c->step_id = 1;
    c->seen0 = c->seen1 = 0;
    c->find_seq = 0;
    c->paths = c->finds = c->scan_steps = 0;
    c->scan_elems = c->init_elems = 0;
    for (int q = 0; q < 16; ++q) c->stamps[q] = 0;
}

// ---------------------------------------------------------------------------------------------
// Minima collection with ties, lapjv.cpp:153-171, from the event / strict bitmaps; wave 0 only.
// order[] holds the columns in position order (just rebuilt from the labels).
//
// Events in position order: e_0 < e_1 < ...  Let L be the index of the last STRICT event.
// Usual shape (clustered family: ~3 strict events then ~150 ties with the global minimum): every
// event up to L is strict, everything after it is a tie.  Then
//   * events 0..L shift: position e_i receives the column that sat at e_(i-1) (at lo for i = 0)
//     and slot lo receives the column of e_L;
//   * the T ties after L fill slots lo+1 .. lo+T in order; slot lo+s held some column B before:
//     if lo+s is not itself a tie position, B ends at the first tie position OUTSIDE the window
//     reached by hopping s -> (t_s - lo) -> ...  (each hop is one serial swap that moved B on);
//     hops only go up, so every slot is resolved independently.
// Both parts are data-parallel over the lanes of wave 0.  A tie BEFORE the last strict event
// (rare) takes the serial loop.
__device__ __forceinline__ void replay_find(const Ctx &cx, int lo, int lane)
{
    int *order = cx.order, *evl = cx.evl, *tmpcol = cx.tmpcol;
    uint32_t *evt = cx.evt, *sbits = cx.sbits;
    const int W = cx.W, n = cx.n;
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
            // all reads precede the writes (one wave, in-order LDS); L + 1 <= 64 on this path
            if (i <= L) order[e] = newcol;
            if (base_i == 0 && L >= 0 && lane == 0) order[lo] = lastcol;
        }
        // ---- 3. tie tail
        const int tb = L + 1;  // evl[tb + s - 1] = position of tie number s (1-based)
        for (int s0 = 1; s0 <= T; s0 += kWave) {  // save the tie columns
            const int sidx = s0 + lane;
            if (sidx <= T) tmpcol[sidx] = order[evl[tb + sidx - 1] & 0x7fffffff];
        }
        for (int s0 = 1; s0 <= T; s0 += kWave) {  // move the displaced columns out of the window
            const int sidx = s0 + lane;
            if (sidx <= T) {
                const int p0 = lo + sidx;
                const int tpos = evl[tb + sidx - 1] & 0x7fffffff;
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
        for (int s0 = 1; s0 <= T; s0 += kWave) {  // pack the ties behind slot lo
            const int sidx = s0 + lane;
            if (sidx <= T) order[lo + sidx] = tmpcol[sidx];
        }
        hi = lo + 1 + T;
    }
    // last free column of the SCAN list wins (lapjv.cpp:250-255)
    int best = -1;
    for (int kk = lo + lane; kk < hi; kk += kWave) {
        if (cx.y[order[kk]] < 0) best = kk;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const int o = __shfl_xor(best, m, kWave);
        best = (o > best) ? o : best;
    }
    const int target = (best >= 0) ? order[best] : -1;
    const int head_j = order[lo];
    const int head_i = cx.y[head_j];
    if (lane == 0) {
        cx.ctl->hi = hi;
        cx.ctl->target = target;
        cx.ctl->head_j = head_j;
        cx.ctl->head_i = head_i;
    }
}

// More than kRecCap tie events in one relax step (lapjv.cpp:199-205): ordered replay from the
// position bitmap on order[] / pos[] (both just rebuilt from the labels).  Wave 0 only.
__device__ __forceinline__ void replay_scan_bitmap(const Ctx &cx, int hi, int lane)
{
    int *order = cx.order, *pos = cx.pos;
    const int n = cx.n, W = cx.W;
    uint32_t *evb = cx.evb;
    Ctl *ctl = cx.ctl;
    int target = -1;
    // the first event (in position order) whose column is free ends the scan (lapjv.cpp:200-201)
    const int fp = uni(ctl->free_pos[0]);
    if (lane == 0) ctl->free_pos[0] = 0x7fffffff;
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
                const int j = uni(order[k]);
                if (k == fp) {
                    target = j;
                } else if ((unsigned)j < (unsigned)n && hi < n) {
                    const int a = uni(order[hi]);
                    const int yj = cx.y[j];
                    const double vj = cx.v[j];
                    if (lane == 0 && (unsigned)a < (unsigned)n) {
                        order[k] = a;
                        pos[a] = k;
                        order[hi] = j;
                        pos[j] = hi;
                        QDesc q;
                        q.j = j;
                        q.i = yj;
                        q.v = vj;
                        cx.qdesc[hi] = q;
                    }
                    ++hi;
                }
            }
        }
    }
    if (lane == 0) {
        ctl->hi = hi;
        ctl->target = target;
        ctl->nmoves = 0;
    }
}

// 2..kRecCap tie events in one relax step (23% of the steps of a uniform instance have 2-4),
// replayed by wave 0 from the finders' RECORDS: lane e holds event e, lane s the column that sits
// at position hi+s (published by its owner).  Events are ranked by position and the swaps of
// lapjv.cpp:199-205 applied in that order on scalars; the outcome is a list of label moves
// (displaced column -> position of the event that displaced it) and the new SCAN-list entries.
__device__ __forceinline__ void replay_records(const Ctx &cx, int hi0, int par, int E, int lane)
{
    Ctl *ctl = cx.ctl;
    const int n = cx.n;
    const Rec rc = ctl->rec[par][(lane < E) ? lane : 0];
    const int ej = rc.j, ei = rc.i;
    const int ep = (lane < E) ? rc.p : 0x7fffffff;
    const double evv = rc.v;
    int A = ctl->apub[lane & (kRecCap - 1)];
    int rank = 0;
    for (int q = 0; q < E; ++q) {
        const int pq = __builtin_amdgcn_readlane(ep, q);
        rank += (pq < ep) ? 1 : 0;
    }
    int target = -1, hi = hi0, bad = 0, moves = 0;
    for (int s = 0; s < E; ++s) {
        const unsigned long long m = __ballot(lane < E && rank == s);
        if (m == 0ull) {  // two records with one position: corrupted state
            bad = 1;
            break;
        }
        const int l = __builtin_ctzll(m);
        const int js = __builtin_amdgcn_readlane(ej, l);
        const int pk = __builtin_amdgcn_readlane(ep, l);
        const int is = __builtin_amdgcn_readlane(ei, l);
        const double vs = readlane_f64(evv, l);
        if (is < 0) {  // first free column in position order ends the scan (lapjv.cpp:200-201)
            target = js;
            break;
        }
        const int a = __builtin_amdgcn_readlane(A, s);
        if ((unsigned)js >= (unsigned)n || (unsigned)a >= (unsigned)n || (unsigned)pk >= (unsigned)n || hi >= n) {
            bad = 1;
            break;
        }
        // position pk now holds a: a later swap of this step that takes its displaced column from
        // exactly that position must see a, not the event column that used to sit there
        if (lane > s && lane < kRecCap && hi0 + lane == pk) A = a;
        if (lane == 0) {
            ctl->mv_a[s] = a;
            ctl->mv_p[s] = pk;
            QDesc q;
            q.j = js;
            q.i = is;
            q.v = vs;
            cx.qdesc[hi] = q;
        }
        ++hi;
        ++moves;
    }
    if (lane == 0) {
        ctl->hi = hi;
        ctl->target = target;
        ctl->nmoves = moves;
        if (bad) ctl->err = 9;
    }
}

// ---------------------------------------------------------------------------------------------
// Direct-to-LDS row request (LDS-DMA, global_load_lds_dwordx4): every lane names its own 16 source
// bytes, the 64 x 16 bytes of a wave land contiguously at a wave-uniform LDS address (M0) + lane*16
// -- no vector register is a destination, so nothing the compiler does with registers can meet a
// load in flight.  The compiler does not count these loads: completion is waited for by hand with
// s_waitcnt vmcnt(N), N = the requests issued AFTER the one that must have landed (vmcnt retires in
// order; extra compiler loads or spills issued in between only make the wait longer, never shorter).
// M0 is written in the same statement that reads it (cdna_hip_programming.md, section 5.7).
__device__ __forceinline__ void dma_request16(const double *gsrc_lane, unsigned lds_dst_uniform)
{
    unsigned keep;
    // (readfirstlane: the "s" operand must be in a scalar register whatever the compiler can prove)
    lds_dst_uniform = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst_uniform);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(lds_dst_uniform)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void dma_wait()
{
    static_assert(N >= 0 && N <= 8, "add the count");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
// LDS byte address of a pointer into the workgroup's LDS block
__device__ __forceinline__ unsigned lds_address(const void *p)
{
    typedef __attribute__((address_space(3))) const unsigned char *lds_cptr;
    return (unsigned)(unsigned long long)(lds_cptr)p;
}

// ---------------------------------------------------------------------------------------------
// One shortest augmenting path from row `start` (lapjv.cpp:221-282).  Called by every thread of
// the workgroup.  Returns the free column reached (and leaves pred[] in LDS for the backtrack,
// v[] updated for the READY columns), or -1 with ctl->err set.  Ends with a barrier.
// TB = upper bound of the workgroup size of the calling kernel: the register budget of this
// function follows from it (1024 threads: 128 VGPRs; 256 threads, one wave per SIMD: 512).
// DMA: the rows of the next two SCAN-list entries are requested TWO relax steps ahead, straight
// into two LDS slots (a CU pulls a 16-KB row from HBM in ~2,500 cycles, more than a step); without
// it the next row is requested one step ahead into registers.  DMA needs VEC2.
template <int CH, bool VEC2, int TB, bool DMA>
__device__ __noinline__ int search_path(Layout layout, int start_in)
{
    static_assert(!DMA || VEC2, "the direct-to-LDS row requests move 16 bytes per lane");
    const Ctx cx = make_ctx(layout);
    const int start = uni(start_in);
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    const int lane = tid & (kWave - 1);
    const int wave = uni(tid >> 6);
    const int nwaves = (nt + kWave - 1) >> 6;
    const int n = cx.n;
    const double *C = cx.C;
    Ctl *ctl = cx.ctl;
    double *dist = cx.dist, *v = cx.v;
    int *order = cx.order, *pos = cx.pos, *y = cx.y;
    QDesc *qdesc = cx.qdesc;
    const int b0 = tid * CH;  // POSITIONS this thread looks at in a minima collection
    const int wordi = b0 >> 5, shift = b0 & 31;
    // owned columns.  VEC2 (n even): pairs (2t, 2t+1) + 2T per chunk, one 16-byte load per pair;
    // otherwise column r*T + t, 8-byte loads.  jl = clamped copy used for addressing only.
    int jc[CH], jl[CH];
    bool inb[CH];
#pragma unroll
    for (int r = 0; r < CH; ++r) {
        if constexpr (VEC2)
            jc[r] = (r >> 1) * 2 * nt + 2 * tid + (r & 1);
        else
            jc[r] = r * nt + tid;
        inb[r] = jc[r] < n;
        if constexpr (VEC2)
            jl[r] = inb[r] ? jc[r] : (n - 2 + (r & 1));  // n is even: the pair stays a pair
        else
            jl[r] = inb[r] ? jc[r] : n - 1;
    }
    auto load_row = [&](const double *row, double (&dst)[CH]) {
        if constexpr (VEC2) {
#pragma unroll
            for (int r = 0; r < CH; r += 2) {
                const double2 t = *reinterpret_cast<const double2 *>(row + jl[r]);
                dst[r] = t.x;
                dst[r + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int r = 0; r < CH; ++r) dst[r] = row[jl[r]];
        }
    };
    double d[CH], vv[CH];
    int prd[CH], yr[CH], ps[CH];
    bool td[CH], rd[CH];  // TODO flag; READY flag (left TODO before the last minima collection)
    // label moves (column -> new position) published in ctl->mv_a / mv_p, applied in list order;
    // four per LDS round trip
    auto apply_moves = [&](int count) {
        for (int q0 = 0; q0 < count; q0 += 4) {
            const int4 a4 = *reinterpret_cast<const int4 *>(&ctl->mv_a[q0]);
            const int4 p4 = *reinterpret_cast<const int4 *>(&ctl->mv_p[q0]);
            const int ma[4] = {a4.x, a4.y, a4.z, a4.w};
            const int mp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (q0 + i < count) {
#pragma unroll
                    for (int r = 0; r < CH; ++r)
                        if (jc[r] == ma[i]) ps[r] = mp[i];
                }
            }
        }
    };
    {
        const double *row = C + (size_t)start * n;
        double c0[CH];
        load_row(row, c0);
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            vv[r] = v[jl[r]];
            yr[r] = y[jl[r]];
        }
#pragma unroll
        for (int r = 0; r < CH; ++r) pin(c0[r]);
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const double val = c0[r] - vv[r];
            d[r] = inb[r] ? val : pos_inf();
            prd[r] = start;
            td[r] = inb[r];
            rd[r] = false;
            ps[r] = jc[r];
        }
    }
    // search state carried from path to path
    int step_id = uni(ctl->step_id);
    int seen0 = uni(ctl->seen0), seen1 = uni(ctl->seen1);
    int find_seq = uni(ctl->find_seq);
    int n_finds = 0, n_steps = 0, sum_hi = 0;  // sum_hi <= (2n+4) * n fits 32 bits for n <= 16384
    int err = 0;

    int lo = 0, hi = 0, target = -1;
    int head_j = 0, head_i = 0;
    double head_v = 0.0, level = 0.0;
    // the entry behind the head, when the SCAN list already holds one (its row is requested one
    // step ahead); pf_valid: cn[] / cn_head hold (or will hold) the row of the CURRENT head
    bool nx_valid = false, pf_valid = false;
    int nx_j = 0, nx_i = 0;
    double nx_v = 0.0;
    // DMA variant: nx2 = the entry at lo+2.  THREE row slots in rotation: slot sx holds / receives
    // the row of nx, slot sx+1 receives the row of nx2, and slot sx+2 is the one the waves took the
    // current row from after the previous barrier -- a request never targets a slot that another
    // wave may still be reading (the head's own entry C[i][j] lies in ANOTHER wave's piece).
    // x_valid: the row of nx has been requested.  The third slot shares its memory with the
    // scratch arrays of the tie replay of a minima collection, which drains the requests first.
    bool nx2_valid = false, x_valid = false;
    int nx2_j = 0, nx2_i = 0, sx = 0;
    (void)nx2_j;
    double cn[CH], cn_head = 0.0;
    double c[CH], c_head = 0.0;
#pragma unroll
    for (int r = 0; r < CH; ++r) cn[r] = c[r] = 0.0;
    // end of a relax step: the row requested for the next head becomes the current row (this is
    // where its loads are waited for)
    auto take_next_row = [&]() {
#pragma unroll
        for (int r = 0; r < CH; ++r) pin(cn[r]);
        pin(cn_head);
#pragma unroll
        for (int r = 0; r < CH; ++r) c[r] = cn[r];
        c_head = cn_head;
    };
    // DMA variant: request row ri (every lane its 16-byte pieces, or everybody the first 16
    // bytes of the matrix when !wide) into slot s.  A wave's 64 pieces land contiguously, so the
    // slot is the row in column order.
    constexpr int kDmaGroup = CH / 2;  // requests per row and wave
    auto dma_row = [&](int ri, bool wide, int s) {
        const double *row = C + (size_t)ri * n;
        const unsigned sbase = lds_address(cx.slots) + (unsigned)s * (unsigned)cx.slot_bytes + (unsigned)wave * 1024u;
#pragma unroll
        for (int r = 0; r < CH; r += 2)
            dma_request16(row + (wide ? jl[r] : 0), sbase + (unsigned)(r >> 1) * (unsigned)nt * 16u);
    };
    // ... and take the row of the new head out of slot s (after the step's barrier: every wave has
    // waited for its own pieces before it)
    auto take_slot = [&](int s, int hj) {
        const unsigned char *sb = cx.slots + (size_t)s * cx.slot_bytes;
#pragma unroll
        for (int r = 0; r < CH; r += 2) {
            const double2 t = *reinterpret_cast<const double2 *>(sb + ((size_t)(r >> 1) * nt + tid) * 16);
            c[r] = t.x;
            c[r + 1] = t.y;
        }
        c_head = *reinterpret_cast<const double *>(sb + (size_t)hj * 8);
    };
#ifdef LAPWARM_STAMPS
    long long cst[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) cst[q] = 0;
#endif
    CSTAMP(tp0);
    while (true) {
        if (lo == hi) {
            // ---------------- minima collection (lapjv.cpp:153-171) over positions [lo, n)
            CSTAMP(tf0);
            // distances and columns of the TODO set in position order; pos[] for the re-read
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                if (td[r]) {
                    dist[ps[r]] = d[r];
                    order[ps[r]] = jc[r];
                    pos[jc[r]] = ps[r];
                }
            }
            __syncthreads();
            CSTAMP(tfa);
            CSTAMP_ADD(9, tfa, tf0);
            ++find_seq;
            ++n_finds;
            const int xp = find_seq & 1;
#pragma unroll
            for (int r = 0; r < CH; ++r) rd[r] = inb[r] && !td[r];
            double dk[CH];
#pragma unroll
            for (int r = 0; r < CH; ++r) dk[r] = dist[(b0 + r < n) ? b0 + r : n - 1];
            double tv = pos_inf();
            int tp = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const int k = b0 + r;
                // position lo always starts as the holder, whatever its value (even +inf / NaN)
                if (k >= lo && k < n && (k == lo || dk[r] < tv)) {
                    tv = dk[r];
                    tp = k;
                }
            }
            double runv = tv, wtv;
            int runp = tp, wtp;
            wave_excl_prefix_min_pair(runv, runp, lane, &wtv, &wtp);
            if (lane == 0) {
                cx.ex->d[xp][wave] = wtv;
                cx.ex->i[xp][wave] = wtp;
            }
            __syncthreads();
            CSTAMP(tfb);
            CSTAMP_ADD(10, tfb, tfa);
            double totv;
            int totp;
            {
                const int w = lane & (kMaxWaves - 1);
                double av = (w < nwaves) ? cx.ex->d[xp][w] : pos_inf();
                int ap = (w < nwaves) ? cx.ex->i[xp][w] : 0x7fffffff;
                scan_step_min_pair<kDppRowShr1, 0xf>(av, ap);
                scan_step_min_pair<kDppRowShr2, 0xf>(av, ap);
                scan_step_min_pair<kDppRowShr4, 0xf>(av, ap);
                scan_step_min_pair<kDppRowShr8, 0xf>(av, ap);
                totv = readlane_f64(av, kMaxWaves - 1);
                totp = __builtin_amdgcn_readlane(ap, kMaxWaves - 1);
                if (wave > 0) {
                    const int wl = wave - 1;
                    const double pv = readlane_f64(av, wl);
                    const int pp = __builtin_amdgcn_readlane(ap, wl);
                    if (pair_less(pv, pp, runv, runp)) {
                        runv = pv;
                        runp = pp;
                    }
                }
            }
            if ((unsigned)totp >= (unsigned)n) {
                err = 7;
                break;
            }
            uint32_t eb = 0, sb = 0;
            bool tie = false;
            int prevpos[CH];
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const int k = b0 + r;
                prevpos[r] = -1;
                if (k >= lo && k < n) {
                    if (k > lo && dk[r] <= runv) {
                        eb |= 1u << r;
                        if (dk[r] < runv) {
                            sb |= 1u << r;
                            prevpos[r] = runp;
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
            // Without ties the swap sequence is a cyclic shift along the strict events: the column
            // of the previous record holder moves to the event's position, the global minimum to
            // lo.  (A uniform instance has ~23 strict events per collection, up to ~160: the new
            // positions go through pos[] and the TODO columns re-read their label.)
            const int min_col = uni(order[totp]);
            if ((unsigned)min_col >= (unsigned)n) {
                err = 7;
                break;
            }
            if (tie) ctl->tie_find = find_seq;
            int prevcol[CH];
#pragma unroll
            for (int r = 0; r < CH; ++r) prevcol[r] = (prevpos[r] >= 0) ? order[prevpos[r]] : 0;
            // head descriptor of the tie-free outcome
            const int min_row_raw = y[min_col];
            const double min_v_raw = v[min_col];
            __syncthreads();
            CSTAMP(tfc);
            CSTAMP_ADD(11, tfc, tfb);
            if (uni(ctl->tie_find) != find_seq) {
                // ---- tie-free
                hi = lo + 1;
                level = totv;
                head_j = min_col;
                head_i = uni(min_row_raw);
                head_v = uni(min_v_raw);
                target = (head_i < 0) ? head_j : -1;
                if (target >= 0) break;
#pragma unroll
                for (int r = 0; r < CH; ++r)
                    if (sb & (1u << r)) pos[prevcol[r]] = b0 + r;
                if (totp != lo && lo >= b0 && lo < b0 + CH) pos[min_col] = lo;
                __syncthreads();
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    if (td[r]) {
                        ps[r] = pos[jc[r]];
                        td[r] = ps[r] >= hi;
                    }
                }
                nx_valid = nx2_valid = false;
            } else {
                // ---- ties: exact ordered replay by wave 0
                CSTAMP_INC(15);
                // (the replay's scratch arrays share the third row slot: no request may be in flight)
                if constexpr (DMA) dma_wait<0>();
                if (eb) {
                    atomicOr(&cx.evt[wordi], eb << shift);
                    if (sb) atomicOr(&cx.sbits[wordi], sb << shift);
                }
                __syncthreads();
                if (wave == 0) replay_find(cx, lo, lane);
                __syncthreads();
                hi = uni(ctl->hi);
                target = uni(ctl->target);
                level = totv;
                head_j = uni(ctl->head_j);
                head_i = uni(ctl->head_i);
                if (target >= 0) break;
                if ((unsigned)head_j >= (unsigned)n || hi <= lo || hi > n) {
                    err = 7;
                    break;
                }
                head_v = uni(v[head_j]);
                // the replay permuted order[]: labels come back through pos[]; queue entries
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    const int k = b0 + r;
                    if (k >= lo && k < n) {
                        const int cj = order[k];
                        pos[cj] = k;
                        if (k > lo && k < hi) {
                            QDesc q;
                            q.j = cj;
                            q.i = y[cj];
                            q.v = v[cj];
                            qdesc[k] = q;
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    if (td[r]) {
                        ps[r] = pos[jc[r]];
                        td[r] = ps[r] >= hi;
                    }
                }
                nx_valid = lo + 1 < hi;
                nx2_valid = lo + 2 < hi;
                {
                    const QDesc q1 = qdesc[(lo + 1 < n) ? lo + 1 : n - 1];
                    const QDesc q2 = qdesc[(lo + 2 < n) ? lo + 2 : n - 1];
                    nx_j = uni(q1.j);
                    nx_i = uni(q1.i);
                    nx_v = uni(q1.v);
                    nx2_j = uni(q2.j);
                    nx2_i = uni(q2.i);
                }
            }
            pf_valid = false;
            x_valid = false;
            CSTAMP(tf1);
            CSTAMP_ADD(12, tf1, tfc);
            CSTAMP_ADD(0, tf1, tf0);
        }
        // ---------------- relax the head of the SCAN list (lapjv.cpp:185-207)
        CSTAMP(tr0);
        // c[] / c_head: the current head's row.  Either it was requested during the previous step
        // (pf_valid: it is already in these registers) or it is requested now.  Indices are clamped
        // for addressing (heads come from SCAN-list entries whose writers validated them).
        if constexpr (!DMA) {
            if (!pf_valid) {
                const int ci = (int)umin_u32((unsigned)head_i, (unsigned)(n - 1));
                const int cj = (int)umin_u32((unsigned)head_j, (unsigned)(n - 1));
                const double *row = C + (size_t)ci * n;
                load_row(row, c);
                c_head = row[cj];
            }
        }
        const bool pf_next = nx_valid;
        if constexpr (DMA) {
            // rows of nx (unless already requested) and of nx2, straight into the LDS slots; the
            // second request is issued whatever the path (a single cache line when there is no
            // nx2), so that the wait before the barrier can always leave kDmaGroup requests in
            // flight
#ifndef LAPWARM_DMA_NOISSUE
            if (nx_valid && !x_valid) dma_row((int)umin_u32((unsigned)nx_i, (unsigned)(n - 1)), true, sx);
            dma_row(nx2_valid ? (int)umin_u32((unsigned)nx2_i, (unsigned)(n - 1)) : 0, nx2_valid, (sx + 1 == 3) ? 0 : sx + 1);
#endif
            x_valid = nx_valid;
            if (!pf_valid) {
                // The current head's row was not requested ahead (first step after a minima
                // collection, or a head that entered the list one step ago): ordinary loads, waited
                // for INSIDE this branch.  The compiler does not know about the requests above and
                // waits for all outstanding loads wherever one of its own is pending; on the
                // prefetched path nothing of its own may be pending, or every step would drain the
                // direct-to-LDS requests.
                const int ci = (int)umin_u32((unsigned)head_i, (unsigned)(n - 1));
                const int cj = (int)umin_u32((unsigned)head_j, (unsigned)(n - 1));
                const double *row = C + (size_t)ci * n;
                load_row(row, c);
                c_head = row[cj];
#pragma unroll
                for (int r = 0; r < CH; ++r) pin(c[r]);
                pin(c_head);
            }
        } else {
            // The request for the NEXT head's row is issued unconditionally (the current row again
            // when the SCAN list holds nothing behind the head: cache hits, result unused), with no
            // branch between the two groups of loads: the number of loads issued after the current
            // row's must not depend on the path, or the compiler waits for ALL outstanding loads
            // before the current row can be used.  The request overlaps this step's arithmetic,
            // barrier and bookkeeping; it is waited for at the end of the step.
            const int pi = (int)umin_u32((unsigned)(nx_valid ? nx_i : head_i), (unsigned)(n - 1));
            const int pj = (int)umin_u32((unsigned)(nx_valid ? nx_j : head_j), (unsigned)(n - 1));
            const double *rown = C + (size_t)pi * n;
            load_row(rown, cn);
            cn_head = rown[pj];
        }
#ifdef LAPWARM_DMA_CHECK
        if constexpr (DMA) {  // diagnostic build: compare the prefetched row with a direct load
            if (pf_valid) {
                double t[CH];
                const double *row = C + (size_t)head_i * n;
                load_row(row, t);
                const double th = row[head_j];
                int bad = 0;
#pragma unroll
                for (int r = 0; r < CH; ++r) bad += (inb[r] && t[r] != c[r]) ? 1 : 0;
                if (bad) atomicAdd((unsigned long long *)&ctl->stamps[0], (unsigned long long)bad);
                if (th != c_head && tid == 0) ctl->stamps[1] += 1;
                if (tid == 0) ctl->stamps[2] += 1;
            } else if (tid == 0) {
                ctl->stamps[3] += 1;
            }
        }
#endif
        const bool y_valid_dma = nx2_valid;  // the other slot will hold the row of nx2
        const int par = step_id & 1;
        step_id++;
        n_steps++;
        sum_hi += hi;
        if (n_steps > 2 * n + 4) {
            err = 1;
            break;
        }
        const int seen = par ? seen1 : seen0;
        if constexpr (!DMA) {
#pragma unroll
            for (int r = 0; r < CH; ++r) pin(c[r]);
            pin(c_head);
        }
        CSTAMP(tr1);
        CSTAMP_ADD(1, tr1, tr0);
        const double h = (c_head - head_v) - level;
        unsigned evm = 0;
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const double cand = (c[r] - vv[r]) - h;
            const bool imp = td[r] & (cand < d[r]);
            const bool ev = imp & (cand == level);
            d[r] = imp ? cand : d[r];
            prd[r] = imp ? head_i : prd[r];
            evm |= ev ? (1u << r) : 0u;
        }
        if (evm) {
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                if ((evm >> r) & 1u) {
                    td[r] = false;
                    const bool fr = yr[r] < 0;
                    // record slot = arrival order (any order: the replay ranks by position)
                    const int old = atomicAdd(&ctl->ev_total[par], fr ? 0x10001 : 1);
                    const int idx = (old - seen) & 0xffff;
                    Rec sl;
                    sl.j = jc[r];
                    sl.p = ps[r];
                    sl.i = yr[r];
                    sl.pad = 0;
                    sl.v = vv[r];
                    sl.pad2 = 0.0;
                    if (idx < kRecCap) ctl->rec[par][idx] = sl;
                    // SCAN-list entry, right if this is the step's only event (else rewritten)
                    if (hi < n) {
                        QDesc q;
                        q.j = jc[r];
                        q.i = yr[r];
                        q.v = vv[r];
                        qdesc[hi] = q;
                    }
                }
            }
        }
        CSTAMP(tr2);
        CSTAMP_ADD(2, tr2, tr1);
        // DMA: this wave's pieces of the row of nx have landed (only the nx2 request stays in flight)
        if constexpr (DMA) dma_wait<kDmaGroup>();
        __syncthreads();
        CSTAMP(tr3);
        CSTAMP_ADD(3, tr3, tr2);
        // one LDS round trip for everything the post phase can need
        const int tot_raw = ctl->ev_total[par];
        const int r0p = ctl->rec[par][0].p;
        const QDesc qd1 = qdesc[(lo + 1 < n) ? lo + 1 : n - 1];
        const QDesc qd2 = qdesc[(lo + 2 < n) ? lo + 2 : n - 1];
        const QDesc qd3 = qdesc[(lo + 3 < n) ? lo + 3 : n - 1];
        const int tot = uni(tot_raw);
        const int dcnt = tot - seen;  // events + 0x10000 * free events of this step
        if (par)
            seen1 = tot;
        else
            seen0 = tot;
        if (dcnt <= 1) {
            // ---- no event, or one event whose column is matched: one barrier
            if (dcnt == 1) {
                // the column at position hi now sits where the event column was
#pragma unroll
                for (int r = 0; r < CH; ++r)
                    if (td[r] && ps[r] == hi) ps[r] = r0p;
                ++hi;
            }
            ++lo;
            if (lo < hi) {
                if (nx_valid) {  // == qd1, already in registers
                    head_j = nx_j;
                    head_i = nx_i;
                    head_v = nx_v;
                } else {
                    head_j = uni(qd1.j);
                    head_i = uni(qd1.i);
                    head_v = uni(qd1.v);
                }
                nx_valid = lo + 1 < hi;
                if (nx_valid) {
                    nx_j = uni(qd2.j);
                    nx_i = uni(qd2.i);
                    nx_v = uni(qd2.v);
                }
            } else {
                nx_valid = false;
            }
            if constexpr (DMA) {
                nx2_valid = lo + 2 < hi;
                nx2_j = uni(qd3.j);
                nx2_i = uni(qd3.i);
                // slot sx holds the row of what is now the head (if it had been requested)
                pf_valid = x_valid && lo < hi;
#ifdef LAPWARM_DMA_NOUSE
                pf_valid = false;
#endif
                if (pf_valid) take_slot(sx, head_j);
                sx = (sx + 1 == 3) ? 0 : sx + 1;
                x_valid = y_valid_dma;
            } else {
                pf_valid = pf_next;
                take_next_row();
            }
#ifdef LAPWARM_STAMPS
            CSTAMP(tr4);
            CSTAMP_ADD(4, tr4, tr3);
            if (dcnt == 0) {
                CSTAMP_INC(5);
                CSTAMP_ADD(13, tr4, tr3);
            } else {
                CSTAMP_INC(6);
                CSTAMP_ADD(14, tr4, tr3);
            }
#endif
        } else {
            const int cnt = dcnt & 0xffff;
            if (cnt == 1) {
                // ---- one event and its column is free: the path ends (lapjv.cpp:200-201)
                target = uni(ctl->rec[par][0].j);
                if ((unsigned)target >= (unsigned)n) err = 8;
                break;
            }
            // ---- several events: ordered replay by wave 0 (two more barriers)
            if (cnt <= kRecCap) {
                // the columns that sit at positions hi..hi+31 publish themselves: they are what the
                // swaps displace (labels are exact for TODO columns and for this step's events)
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    const unsigned off = (unsigned)(ps[r] - hi);
                    if ((td[r] || ((evm >> r) & 1u)) && off < (unsigned)kRecCap) ctl->apub[off] = jc[r];
                }
            } else {
                // more events than record slots: position bitmap + the permutation rebuilt from labels
#pragma unroll
                for (int r = 0; r < CH; ++r) {
                    if ((evm >> r) & 1u) {
                        atomicOr(&cx.evb[ps[r] >> 5], 1u << (ps[r] & 31));
                        if (yr[r] < 0) atomicMin(&ctl->free_pos[0], ps[r]);
                    }
                    if (inb[r] && (td[r] || ((evm >> r) & 1u))) {
                        order[ps[r]] = jc[r];
                        pos[jc[r]] = ps[r];
                    }
                }
            }
            __syncthreads();
            if (wave == 0) {
                if (cnt <= kRecCap)
                    replay_records(cx, hi, par, cnt, lane);
                else
                    replay_scan_bitmap(cx, hi, lane);
                // outcome block: one LDS round trip for the other waves
                const int k1 = (lo + 1 < n) ? lo + 1 : n - 1, k2 = (lo + 2 < n) ? lo + 2 : n - 1;
                const int k3 = (lo + 3 < n) ? lo + 3 : n - 1;
                const QDesc e1 = qdesc[k1], e2 = qdesc[k2], e3 = qdesc[k3];
                const int rh = ctl->hi, rt = ctl->target, rm = ctl->nmoves, re = ctl->err;
                if (lane == 0) {
                    ctl->res[0] = rh;
                    ctl->res[1] = rt;
                    ctl->res[2] = rm;
                    ctl->res[3] = re;
                    ctl->resq[0] = e1;
                    ctl->resq[1] = e2;
                    ctl->resq[2] = e3;
                }
            }
            __syncthreads();
            const int4 res = *reinterpret_cast<const int4 *>(ctl->res);
            const QDesc q1 = ctl->resq[0], q2 = ctl->resq[1], q3 = ctl->resq[2];
            const int hi_new = uni(res.x);
            target = uni(res.y);
            if (target >= 0) break;
            if (uni(res.w)) {
                err = 9;
                break;
            }
            if (cnt <= kRecCap) {
                apply_moves(uni(res.z));
            } else {
#pragma unroll
                for (int r = 0; r < CH; ++r)
                    if (td[r]) ps[r] = pos[jc[r]];
            }
            hi = hi_new;
            ++lo;
            if (lo >= hi || hi > n) {
                err = 9;
                break;
            }
            head_j = uni(q1.j);
            head_i = uni(q1.i);
            head_v = uni(q1.v);
            nx_valid = lo + 1 < hi;
            nx_j = uni(q2.j);
            nx_i = uni(q2.i);
            nx_v = uni(q2.v);
            if constexpr (DMA) {
                nx2_valid = lo + 2 < hi;
                nx2_j = uni(q3.j);
                nx2_i = uni(q3.i);
                pf_valid = x_valid;
#ifdef LAPWARM_DMA_NOUSE
                pf_valid = false;
#endif
                if (pf_valid) take_slot(sx, head_j);
                sx = (sx + 1 == 3) ? 0 : sx + 1;
                x_valid = y_valid_dma;
            } else {
                pf_valid = pf_next;
                take_next_row();
            }
            CSTAMP(tr4);
            CSTAMP_ADD(4, tr4, tr3);
            CSTAMP_INC(7);
        }
    }
    CSTAMP(tpe);
    if (!err) {
        // dual update for the READY columns (lapjv.cpp:270-276): v[j] += d[j] - level
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            if (rd[r]) {
                vv[r] += d[r] - level;
                v[jc[r]] = vv[r];
            }
            if (inb[r]) cx.pred[jc[r]] = prd[r];  // for the backtrack
        }
    }
    if (tid == 0) {
        ctl->step_id = step_id;
        ctl->seen0 = seen0;
        ctl->seen1 = seen1;
        ctl->find_seq = find_seq;
        ctl->paths += 1;
        ctl->finds += n_finds;
        ctl->scan_steps += n_steps;
        ctl->scan_elems += (long long)n_steps * n - (long long)sum_hi;  // sum of (n - hi) over the steps
        ctl->init_elems += n;
        if (err) ctl->err = err;
    }
    __syncthreads();
#ifdef LAPWARM_STAMPS
    {
        CSTAMP(tpf);
        CSTAMP_ADD(8, tpf, tpe);  // path end: dual update, pred dump, barrier
        if (tid == (LAPWARM_STAMP_TID)) {
#pragma unroll
            for (int q = 0; q < 16; ++q) ctl->stamps[q] += cst[q];
        }
        (void)tp0;
    }
#endif
    return err ? -1 : target;
}

}  // namespace cols
}  // namespace lapwarm
