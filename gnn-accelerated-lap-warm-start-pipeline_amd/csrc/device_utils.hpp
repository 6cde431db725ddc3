// device_utils.hpp -- wave64 / workgroup primitives shared by the gfx950 kernels.
//
// Everything here is written for CDNA4 directly: 64-lane wavefronts, LDS as the
// cross-wave exchange, no 32-lane assumptions anywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lapwarm {

constexpr int kWave = 64;
constexpr int kMaxWaves = 16;           // 1024 threads / 64
constexpr double kLarge = 1000000.0;    // the reference's LARGE sentinel (LAP/_lapjv_cpp/lapjv.h:4)

// per-instance flag bits written by the dense prelude / projection kernels
constexpr int kFlagHasViolation = 1;    // some (i,j) has (u_i+v_j)-C_ij > eps under the seed duals
constexpr int kFlagInfeasible = 2;      // some (i,j) has (C_ij-u_i)-v_j < -eps (verify step)
constexpr int kFlagProjected = 4;       // the projection kernel changed (u,v): prelude must be redone

// Keeps a loaded value (and therefore its load) where it is: LLVM otherwise sinks a load into the
// divergent branch that consumes it, which serialises the memory latencies of a thread's gathers.
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }

// Values that are identical in every lane (read from a uniform LDS address, or the result of a
// full butterfly) are moved to scalar registers so that the control state machine of the solver
// runs on the scalar unit instead of being replicated on the vector ALUs of 16 waves.
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ double uni(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffLL));
    const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ unsigned umin_u32(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ double pos_inf() { return __longlong_as_double(0x7ff0000000000000LL); }

__device__ __forceinline__ double dmin(double a, double b) { return (b < a) ? b : a; }

__device__ __forceinline__ bool pair_less(double a, int ia, double b, int ib)
{
    return (a < b) || (a == b && ia < ib);
}

// ---- DPP data movement (gfx9 encodings): row_shr:n moves lane i-n -> i inside a row of 16 lanes,
// row_bcast:15 / :31 broadcast the last lane of a row / of the lower half to the following rows.
// Lanes without a source keep `old` (bound_ctrl = false), which callers set to the identity.
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_move(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double old, double src)
{
    const long long o = __double_as_longlong(old), x = __double_as_longlong(src);
    const int lo = dpp_move<CTRL, ROW_MASK>((int)(o & 0xffffffffLL), (int)(x & 0xffffffffLL));
    const int hi = dpp_move<CTRL, ROW_MASK>((int)(o >> 32), (int)(x >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double readlane_f64(double x, int l)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Wave64 reduction skeleton: after the six steps lane 63 holds the reduction of all lanes.
#define LAPWARM_DPP_REDUCE(STEP)      \
    STEP(kDppRowShr1, 0xf)            \
    STEP(kDppRowShr2, 0xf)            \
    STEP(kDppRowShr4, 0xf)            \
    STEP(kDppRowShr8, 0xf)            \
    STEP(kDppRowBcast15, 0xa)         \
    STEP(kDppRowBcast31, 0xc)

// ---- wave-level reductions (all 64 lanes must be active) --------------------------------
__device__ __forceinline__ double wave_min(double v)
{
#define LAPWARM_STEP(C, M) v = dmin(v, dpp_move<C, M>(pos_inf(), v));
    LAPWARM_DPP_REDUCE(LAPWARM_STEP)
#undef LAPWARM_STEP
    return readlane_f64(v, kWave - 1);
}

__device__ __forceinline__ int wave_min_i32(int v)
{
#define LAPWARM_STEP(C, M)                                  \
    {                                                       \
        const int o = dpp_move<C, M>(0x7fffffff, v);        \
        v = (o < v) ? o : v;                                \
    }
    LAPWARM_DPP_REDUCE(LAPWARM_STEP)
#undef LAPWARM_STEP
    return __builtin_amdgcn_readlane(v, kWave - 1);
}

__device__ __forceinline__ int wave_sum_i32(int v)
{
#define LAPWARM_STEP(C, M) v += dpp_move<C, M>(0, v);
    LAPWARM_DPP_REDUCE(LAPWARM_STEP)
#undef LAPWARM_STEP
    return __builtin_amdgcn_readlane(v, kWave - 1);
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
}

__device__ __forceinline__ double wave_max(double v)
{
#define LAPWARM_STEP(C, M)                                  \
    {                                                       \
        const double o = dpp_move<C, M>(-pos_inf(), v);     \
        v = (o > v) ? o : v;                                \
    }
    LAPWARM_DPP_REDUCE(LAPWARM_STEP)
#undef LAPWARM_STEP
    return readlane_f64(v, kWave - 1);
}

// Two lexicographically smallest (value, index) pairs; "empty" = (+inf, INT_MAX).
struct Top2 {
    double a1;
    int i1;
    double a2;
    int i2;
};

__device__ __forceinline__ Top2 top2_empty()
{
    Top2 t;
    t.a1 = pos_inf();
    t.i1 = 0x7fffffff;
    t.a2 = pos_inf();
    t.i2 = 0x7fffffff;
    return t;
}

// Branch-free selects on values only: passing the structs by reference made the compiler keep
// them in scratch memory (a select between addresses), which cost ~25 us per ARR iteration.
__device__ __forceinline__ void top2_push(Top2 &t, double a, int i)
{
    const bool first = pair_less(a, i, t.a1, t.i1);
    const bool second = pair_less(a, i, t.a2, t.i2);
    const double n2a = first ? t.a1 : (second ? a : t.a2);
    const int n2i = first ? t.i1 : (second ? i : t.i2);
    t.a1 = first ? a : t.a1;
    t.i1 = first ? i : t.i1;
    t.a2 = n2a;
    t.i2 = n2i;
}

__device__ __forceinline__ Top2 top2_merge(Top2 p, Top2 q)
{
    const bool qf = pair_less(q.a1, q.i1, p.a1, p.i1);
    const double fa = qf ? q.a1 : p.a1;
    const int fi = qf ? q.i1 : p.i1;
    const double la = qf ? p.a1 : q.a1;  // the losing head
    const int li = qf ? p.i1 : q.i1;
    const double sa = qf ? q.a2 : p.a2;  // the winner's own runner-up
    const int si = qf ? q.i2 : p.i2;
    const bool ls = pair_less(la, li, sa, si);
    Top2 r;
    r.a1 = fa;
    r.i1 = fi;
    r.a2 = ls ? la : sa;
    r.i2 = ls ? li : si;
    return r;
}

__device__ __forceinline__ Top2 wave_top2(Top2 t)
{
    // the element sets merged at every step are disjoint (prefix of a row / of the wave), so the
    // merge of the two runner-up lists is exact
#define LAPWARM_STEP(C, M)                                          \
    {                                                               \
        Top2 o;                                                     \
        o.a1 = dpp_move<C, M>(pos_inf(), t.a1);                     \
        o.i1 = dpp_move<C, M>(0x7fffffff, t.i1);                    \
        o.a2 = dpp_move<C, M>(pos_inf(), t.a2);                     \
        o.i2 = dpp_move<C, M>(0x7fffffff, t.i2);                    \
        t = top2_merge(t, o);                                       \
    }
    LAPWARM_DPP_REDUCE(LAPWARM_STEP)
#undef LAPWARM_STEP
    Top2 r;
    r.a1 = readlane_f64(t.a1, kWave - 1);
    r.i1 = __builtin_amdgcn_readlane(t.i1, kWave - 1);
    r.a2 = readlane_f64(t.a2, kWave - 1);
    r.i2 = __builtin_amdgcn_readlane(t.i2, kWave - 1);
    return r;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void scan_step_min_pair(double &v, int &i)
{
    const double ov = dpp_move<CTRL, ROW_MASK>(pos_inf(), v);
    const int oi = dpp_move<CTRL, ROW_MASK>(0x7fffffff, i);
    if (pair_less(ov, oi, v, i)) {
        v = ov;
        i = oi;
    }
}

// Exclusive prefix of the lexicographic (value, index) minimum over the lanes of a wave; lane 0
// gets (+inf, INT_MAX).  (*tv, *ti) = the wave's total.  Six DPP steps (register-to-register,
// no LDS crossbar round trips) + one shift.
__device__ __forceinline__ void wave_excl_prefix_min_pair(double &v, int &idx, int lane, double *tv, int *ti)
{
    double iv = v;
    int ii = idx;
#ifdef LAPWARM_NO_DPP_SCAN
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const double ov = __shfl_up(iv, off, kWave);
        const int oi = __shfl_up(ii, off, kWave);
        if (lane >= off && pair_less(ov, oi, iv, ii)) {
            iv = ov;
            ii = oi;
        }
    }
    *tv = __shfl(iv, kWave - 1, kWave);
    *ti = __shfl(ii, kWave - 1, kWave);
    {
        const double pv0 = __shfl_up(iv, 1, kWave);
        const int pi0 = __shfl_up(ii, 1, kWave);
        v = (lane == 0) ? pos_inf() : pv0;
        idx = (lane == 0) ? 0x7fffffff : pi0;
        return;
    }
#endif
    scan_step_min_pair<kDppRowShr1, 0xf>(iv, ii);
    scan_step_min_pair<kDppRowShr2, 0xf>(iv, ii);
    scan_step_min_pair<kDppRowShr4, 0xf>(iv, ii);
    scan_step_min_pair<kDppRowShr8, 0xf>(iv, ii);
    scan_step_min_pair<kDppRowBcast15, 0xa>(iv, ii);
    scan_step_min_pair<kDppRowBcast31, 0xc>(iv, ii);
    {
        const long long b = __double_as_longlong(iv);
        const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), kWave - 1);
        const int hi = __builtin_amdgcn_readlane((int)(b >> 32), kWave - 1);
        *tv = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        *ti = __builtin_amdgcn_readlane(ii, kWave - 1);
    }
    const double pv = __shfl_up(iv, 1, kWave);
    const int pi = __shfl_up(ii, 1, kWave);
    v = (lane == 0) ? pos_inf() : pv;
    idx = (lane == 0) ? 0x7fffffff : pi;
}

// 16-lane row reduction (every row of the wave holds the same 16 values): lane 15 of each row ends
// with the result; read it from lane 15.
#define LAPWARM_DPP_ROW_REDUCE(STEP) \
    STEP(kDppRowShr1, 0xf)           \
    STEP(kDppRowShr2, 0xf)           \
    STEP(kDppRowShr4, 0xf)           \
    STEP(kDppRowShr8, 0xf)

// ---- workgroup-level exchange through LDS ------------------------------------------------
// Two alternating slot sets: call k writes set (k&1), one barrier, then reads it.  A thread
// can only reach the write of call k+2 after the barrier of call k+1, which every reader of
// call k has passed, so one barrier per call is enough.
struct BlockExchange {
    double d[2][kMaxWaves * 4];
    double bcast[2];
    int i[2][kMaxWaves * 4];
};

// Two smallest (value, index) candidates of an ARR row scan with what the serial code reads next
// as payload: v[j1], y[j1], y[j2].  Lanes (and waves) own ascending column ranges, so "lowest
// lane / wave holding the minimum value" IS the lexicographic (value, index) tie-break; the
// reduction therefore needs only value minima (DPP) + a ballot, not tuple merges.
struct Arr2 {
    double a1, vj1, a2;
    int i1, y1, i2, y2;
};

__device__ __forceinline__ Arr2 arr2_empty()
{
    Arr2 t;
    t.a1 = t.a2 = pos_inf();
    t.vj1 = 0.0;
    t.i1 = t.i2 = 0x7fffffff;
    t.y1 = t.y2 = -1;
    return t;
}

__device__ __forceinline__ void arr2_push(Arr2 &t, double a, int i, double vj, int yj)
{
    const bool first = pair_less(a, i, t.a1, t.i1);
    const bool second = pair_less(a, i, t.a2, t.i2);
    t.a2 = first ? t.a1 : (second ? a : t.a2);
    t.i2 = first ? t.i1 : (second ? i : t.i2);
    t.y2 = first ? t.y1 : (second ? yj : t.y2);
    t.a1 = first ? a : t.a1;
    t.i1 = first ? i : t.i1;
    t.vj1 = first ? vj : t.vj1;
    t.y1 = first ? yj : t.y1;
}

// Reduce over `width` consecutive lanes holding candidates in ascending index order (width = 64:
// whole wave via the six-step DPP reduction; width = 16: one row).  Result is wave-uniform.
template <int WIDTH>
__device__ __forceinline__ Arr2 arr2_reduce_lanes(const Arr2 &t)
{
    constexpr int last = WIDTH - 1;
    const unsigned long long lanes = (WIDTH == 64) ? ~0ull : 0xffffull;
    double m1 = t.a1;
    if constexpr (WIDTH == 64) {
#define LAPWARM_STEP(C, M) m1 = dmin(m1, dpp_move<C, M>(pos_inf(), m1));
        LAPWARM_DPP_REDUCE(LAPWARM_STEP)
#undef LAPWARM_STEP
    } else {
#define LAPWARM_STEP(C, M) m1 = dmin(m1, dpp_move<C, M>(pos_inf(), m1));
        LAPWARM_DPP_ROW_REDUCE(LAPWARM_STEP)
#undef LAPWARM_STEP
    }
    m1 = readlane_f64(m1, last);
    const unsigned long long k1 = __ballot(t.a1 == m1) & lanes;
    const int l1 = k1 ? __builtin_ctzll(k1) : 0;  // all +inf / NaN: any lane, values are empty
    Arr2 r;
    r.a1 = readlane_f64(t.a1, l1);
    r.vj1 = readlane_f64(t.vj1, l1);
    r.i1 = __builtin_amdgcn_readlane(t.i1, l1);
    r.y1 = __builtin_amdgcn_readlane(t.y1, l1);
    const int lane = threadIdx.x & (kWave - 1);
    const double c2 = ((lane & last) == l1) ? t.a2 : t.a1;
    const int c2i = ((lane & last) == l1) ? t.i2 : t.i1;
    const int c2y = ((lane & last) == l1) ? t.y2 : t.y1;
    double m2 = c2;
    if constexpr (WIDTH == 64) {
#define LAPWARM_STEP(C, M) m2 = dmin(m2, dpp_move<C, M>(pos_inf(), m2));
        LAPWARM_DPP_REDUCE(LAPWARM_STEP)
#undef LAPWARM_STEP
    } else {
#define LAPWARM_STEP(C, M) m2 = dmin(m2, dpp_move<C, M>(pos_inf(), m2));
        LAPWARM_DPP_ROW_REDUCE(LAPWARM_STEP)
#undef LAPWARM_STEP
    }
    m2 = readlane_f64(m2, last);
    const unsigned long long k2 = __ballot(c2 == m2) & lanes;
    const int l2 = k2 ? __builtin_ctzll(k2) : 0;
    r.a2 = readlane_f64(c2, l2);
    r.i2 = __builtin_amdgcn_readlane(c2i, l2);
    r.y2 = __builtin_amdgcn_readlane(c2y, l2);
    return r;
}

struct BlockCtx {
    int tid, lane, wave, nwaves;
    int parity;
    BlockExchange *ex;

    __device__ __forceinline__ void init(BlockExchange *e)
    {
        tid = threadIdx.x;
        lane = tid & (kWave - 1);
        wave = tid >> 6;
        nwaves = (blockDim.x + kWave - 1) >> 6;
        parity = 0;
        ex = e;
    }

    // Cross-wave combine without a serial LDS loop: lane l reads the slot of wave (l & 15) -- one
    // LDS round trip -- and a 4-step butterfly over 16 lanes finishes the reduction.
    __device__ __forceinline__ double min_f64(double v)
    {
        v = wave_min(v);
        const int p = parity;
        parity ^= 1;
        if (lane == 0) ex->d[p][wave] = v;
        __syncthreads();
        const int w = lane & (kMaxWaves - 1);
        double r = (w < nwaves) ? ex->d[p][w] : pos_inf();
#define LAPWARM_STEP(C, M) r = dmin(r, dpp_move<C, M>(pos_inf(), r));
        LAPWARM_DPP_ROW_REDUCE(LAPWARM_STEP)
#undef LAPWARM_STEP
        return readlane_f64(r, 15);
    }

    __device__ __forceinline__ double max_f64(double v)
    {
        v = wave_max(v);
        const int p = parity;
        parity ^= 1;
        if (lane == 0) ex->d[p][wave] = v;
        __syncthreads();
        double r = ex->d[p][0];
        for (int w = 1; w < nwaves; ++w) {
            const double o = ex->d[p][w];
            r = (o > r) ? o : r;
        }
        return r;
    }

    // deterministic: butterfly inside each wave, then waves in index order
    __device__ __forceinline__ double sum_f64(double v)
    {
        v = wave_sum_f64(v);
        const int p = parity;
        parity ^= 1;
        if (lane == 0) ex->d[p][wave] = v;
        __syncthreads();
        double r = ex->d[p][0];
        for (int w = 1; w < nwaves; ++w) r += ex->d[p][w];
        return r;
    }

    __device__ __forceinline__ int min_i32(int v)
    {
        v = wave_min_i32(v);
        const int p = parity;
        parity ^= 1;
        if (lane == 0) ex->i[p][wave] = v;
        __syncthreads();
        const int w = lane & (kMaxWaves - 1);
        int r = (w < nwaves) ? ex->i[p][w] : 0x7fffffff;
#define LAPWARM_STEP(C, M)                                  \
    {                                                       \
        const int o = dpp_move<C, M>(0x7fffffff, r);        \
        r = (o < r) ? o : r;                                \
    }
        LAPWARM_DPP_ROW_REDUCE(LAPWARM_STEP)
#undef LAPWARM_STEP
        return __builtin_amdgcn_readlane(r, 15);
    }

    __device__ __forceinline__ int sum_i32(int v)
    {
        v = wave_sum_i32(v);
        const int p = parity;
        parity ^= 1;
        if (lane == 0) ex->i[p][wave] = v;
        __syncthreads();
        const int w = lane & (kMaxWaves - 1);
        int r = (w < nwaves) ? ex->i[p][w] : 0;
#define LAPWARM_STEP(C, M) r += dpp_move<C, M>(0, r);
        LAPWARM_DPP_ROW_REDUCE(LAPWARM_STEP)
#undef LAPWARM_STEP
        return __builtin_amdgcn_readlane(r, 15);
    }

    // Workgroup-wide Arr2 reduction; *c0 (owned by thread 0) is broadcast alongside.  One barrier.
    __device__ __forceinline__ Arr2 arr2(const Arr2 &t, double *c0)
    {
        const Arr2 w = arr2_reduce_lanes<64>(t);
        const int p = parity;
        parity ^= 1;
        if (lane == 0) {
            ex->d[p][4 * wave + 0] = w.a1;
            ex->d[p][4 * wave + 1] = w.vj1;
            ex->d[p][4 * wave + 2] = w.a2;
            ex->i[p][4 * wave + 0] = w.i1;
            ex->i[p][4 * wave + 1] = w.y1;
            ex->i[p][4 * wave + 2] = w.i2;
            ex->i[p][4 * wave + 3] = w.y2;
        }
        if (tid == 0) ex->bcast[p] = *c0;
        __syncthreads();
        const int q = lane & (kMaxWaves - 1);
        Arr2 s = arr2_empty();
        if (q < nwaves) {
            s.a1 = ex->d[p][4 * q + 0];
            s.vj1 = ex->d[p][4 * q + 1];
            s.a2 = ex->d[p][4 * q + 2];
            s.i1 = ex->i[p][4 * q + 0];
            s.y1 = ex->i[p][4 * q + 1];
            s.i2 = ex->i[p][4 * q + 2];
            s.y2 = ex->i[p][4 * q + 3];
        }
        *c0 = ex->bcast[p];
        return arr2_reduce_lanes<16>(s);
    }

    __device__ __forceinline__ Top2 top2(Top2 t)
    {
        t = wave_top2(t);
        const int p = parity;
        parity ^= 1;
        if (lane == 0) {
            ex->d[p][2 * wave] = t.a1;
            ex->d[p][2 * wave + 1] = t.a2;
            ex->i[p][2 * wave] = t.i1;
            ex->i[p][2 * wave + 1] = t.i2;
        }
        __syncthreads();
        const int w = lane & (kMaxWaves - 1);
        Top2 r = top2_empty();
        if (w < nwaves) {
            r.a1 = ex->d[p][2 * w];
            r.a2 = ex->d[p][2 * w + 1];
            r.i1 = ex->i[p][2 * w];
            r.i2 = ex->i[p][2 * w + 1];
        }
#define LAPWARM_STEP(C, M)                                          \
    {                                                               \
        Top2 o;                                                     \
        o.a1 = dpp_move<C, M>(pos_inf(), r.a1);                     \
        o.i1 = dpp_move<C, M>(0x7fffffff, r.i1);                    \
        o.a2 = dpp_move<C, M>(pos_inf(), r.a2);                     \
        o.i2 = dpp_move<C, M>(0x7fffffff, r.i2);                    \
        r = top2_merge(r, o);                                       \
    }
        LAPWARM_DPP_ROW_REDUCE(LAPWARM_STEP)
#undef LAPWARM_STEP
        Top2 out;
        out.a1 = readlane_f64(r.a1, 15);
        out.i1 = __builtin_amdgcn_readlane(r.i1, 15);
        out.a2 = readlane_f64(r.a2, 15);
        out.i2 = __builtin_amdgcn_readlane(r.i2, 15);
        return out;
    }
};

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
    static_assert(N == 0, "only the full drain is used");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// LDS byte address of a pointer into the workgroup's LDS block
__device__ __forceinline__ unsigned lds_address(const void *p)
{
    typedef __attribute__((address_space(3))) const unsigned char *lds_cptr;
    return (unsigned)(unsigned long long)(lds_cptr)p;
}

}  // namespace lapwarm
