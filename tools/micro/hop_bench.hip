// hop_bench.hip -- price of one all-to-all exchange round between G single-wave workgroups
// (the cooperative solver's step): every member stores K 8-byte {value, tag} granules (sc1,
// relaxed agent-scope atomics: cdna_hip_programming.md Guideline 16, recipe R2) and polls the
// G*K granules of the round until every tag matches.  Optionally every member also does one
// dependent 8-byte-per-lane x CH gather from a large buffer per round (the head-row piece).
//
//   hipcc -O3 --offload-arch=gfx950 -o hop_bench hop_bench.hip && ./hop_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Params {
    unsigned long long *mail;  // [inst][2][G*K]
    const double *rows;        // row pool (rows_n rows of rowlen doubles) or null
    long long *out;            // [inst][G][2]: ticks (100 MHz), failures
    int G, K, rounds, same_xcd, rowlen, rows_n, ch, inst;
    int store_mode;  // 0: agent-scope atomic store (sc1, write-through); 1: workgroup-scope atomic store (sc0: stays in
                     // the XCD's L2 -- only meaningful when every member sits on one XCD); 2: plain volatile store
    int sleep;       // s_sleep between polls
    int *xcc;        // [inst][G] XCC id of each member
};

template <int CH>
__global__ void __launch_bounds__(64) hop_kernel(Params p)
{
    // placement: same_xcd -> members of instance b sit at blockIdx = (b/8)*8G + g*8 + b%8
    int b, g;
    if (p.same_xcd) {
        const int grp = blockIdx.x / (8 * p.G), rem = blockIdx.x % (8 * p.G);
        g = rem / 8;
        b = grp * 8 + rem % 8;
    } else {
        b = blockIdx.x / p.G;
        g = blockIdx.x % p.G;
    }
    if (b >= p.inst) return;
    const int lane = threadIdx.x;
    const int GK = p.G * p.K;
    unsigned long long *mail = p.mail + (size_t)b * 2 * GK;
    const int nl = (GK + 63) / 64;
    unsigned long long acc = 0;
    double facc = 0.0;
    int fails = 0;
    unsigned row = (unsigned)(b * 131 + g * 7) % (unsigned)p.rows_n;
    if (lane == 0) p.xcc[b * p.G + g] = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= p.rounds; ++r) {
        if (p.rows) {
            // dependent gather: this member's piece of row `row` (row depends on last round's data)
            const double *src = p.rows + (size_t)row * p.rowlen + (size_t)g * 64 * CH + lane * CH;
            double c[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) c[q] = src[q];
#pragma unroll
            for (int q = 0; q < CH; ++q) facc += c[q];
        }
        unsigned long long *buf = mail + (size_t)(r & 1) * GK;
        if (lane < p.K) {
            const unsigned val = (unsigned)(g * 1000 + lane) + (unsigned)r * 3u + (unsigned)(facc != 12345.0);
            const unsigned long long w = ((unsigned long long)(unsigned)r << 32) | val;
            if (p.store_mode == 0)
                __hip_atomic_store((gu64 *)(buf + g * p.K + lane), w, RLX_AGENT);
            else if (p.store_mode == 1)
                __hip_atomic_store((gu64 *)(buf + g * p.K + lane), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
                *(volatile unsigned long long *)(buf + g * p.K + lane) = w;
        }
        unsigned long long got[4] = {0, 0, 0, 0};
        unsigned spins = 0;
        while (true) {
            bool ok = true;
            for (int q = 0; q < nl && q < 4; ++q) {
                const int idx = q * 64 + lane;
                if (idx < GK) {
                    const unsigned long long x = __hip_atomic_load((gu64 *)(buf + idx), RLX_AGENT);
                    got[q] = x;
                    ok &= (unsigned)(x >> 32) == (unsigned)r;
                }
            }
            if (__all(ok)) break;
            if (p.sleep) __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 21)) {
                fails++;
                break;
            }
        }
        if (fails) break;
        unsigned long long s = 0;
        for (int q = 0; q < 4; ++q) s += (unsigned)got[q];
        // wave sum (so that the next row really depends on everything received)
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        acc += s;
        row = (unsigned)(acc % (unsigned long long)p.rows_n);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        p.out[((size_t)b * p.G + g) * 2 + 0] = (long long)(t1 - t0);
        p.out[((size_t)b * p.G + g) * 2 + 1] = fails + (long long)((acc == 0x123456789ull) + (facc == 0.5));
    }
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("%s: %s\n", #x, hipGetErrorString(e));                      \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

int main()
{
    const int rounds = 20000;
    const int rowlen = 16384, rows_n = 8192;  // 1 GiB pool: beyond the Infinity Cache
    double *rows;
    CK(hipMalloc(&rows, (size_t)rowlen * rows_n * 8));
    CK(hipMemset(rows, 0, (size_t)rowlen * rows_n * 8));
    printf("inst G K same_xcd gather store sleep   us/round (max over members)  fails  xcds\n");
    const int cfgs[][7] = {
        // inst, G, K, same_xcd, gather(CH or 0), store_mode, sleep
        {1, 2, 1, 1, 0, 0, 0},  {1, 2, 1, 1, 0, 1, 0},  {1, 2, 1, 1, 0, 2, 0},  {1, 2, 1, 0, 0, 1, 0},
        {1, 8, 6, 1, 0, 0, 0},  {1, 8, 6, 1, 0, 1, 0},  {1, 8, 6, 1, 0, 2, 0},  {1, 8, 6, 1, 0, 0, 1},
        {1, 16, 6, 1, 0, 1, 0}, {1, 32, 6, 1, 0, 1, 0}, {1, 16, 6, 1, 4, 1, 0},
        {32, 8, 6, 1, 0, 0, 0}, {32, 8, 6, 1, 0, 1, 0}, {32, 8, 6, 1, 4, 1, 0}, {32, 8, 6, 1, 4, 0, 1},
        {32, 16, 6, 1, 0, 1, 0}, {32, 16, 6, 1, 4, 1, 0}, {32, 16, 6, 1, 4, 0, 1}, {8, 32, 6, 1, 8, 1, 0},
        {32, 8, 6, 0, 4, 1, 0},
    };
    for (auto &c : cfgs) {
        Params p;
        p.inst = c[0];
        p.G = c[1];
        p.K = c[2];
        p.same_xcd = c[3];
        p.ch = c[4];
        p.rounds = rounds;
        p.rowlen = rowlen;
        p.rows_n = rows_n;
        p.rows = c[4] ? rows : nullptr;
        p.store_mode = c[5];
        p.sleep = c[6];
        CK(hipMalloc(&p.xcc, (size_t)p.inst * p.G * 4));
        const size_t mail_bytes = (size_t)p.inst * 2 * p.G * p.K * 8;
        CK(hipMalloc(&p.mail, mail_bytes));
        CK(hipMemset(p.mail, 0, mail_bytes));
        CK(hipMalloc(&p.out, (size_t)p.inst * p.G * 16));
        const int grid = p.same_xcd ? ((p.inst + 7) / 8) * 8 * p.G : p.inst * p.G;
        if (c[4] == 8)
            hipLaunchKernelGGL(hop_kernel<8>, dim3(grid), dim3(64), 0, 0, p);
        else
            hipLaunchKernelGGL(hop_kernel<4>, dim3(grid), dim3(64), 0, 0, p);
        CK(hipDeviceSynchronize());
        std::vector<long long> out((size_t)p.inst * p.G * 2);
        CK(hipMemcpy(out.data(), p.out, out.size() * 8, hipMemcpyDeviceToHost));
        long long mx = 0, fails = 0;
        for (size_t i = 0; i < out.size(); i += 2) {
            if (out[i] > mx) mx = out[i];
            fails += out[i + 1];
        }
        std::vector<int> xcc((size_t)p.inst * p.G);
        CK(hipMemcpy(xcc.data(), p.xcc, xcc.size() * 4, hipMemcpyDeviceToHost));
        int mixed = 0;  // instances whose members do not share an XCD
        for (int b = 0; b < p.inst; ++b)
            for (int g = 1; g < p.G; ++g)
                if (xcc[(size_t)b * p.G + g] != xcc[(size_t)b * p.G]) {
                    ++mixed;
                    break;
                }
        printf("%4d %2d %d %d %d %d %d   %.3f   %lld   mixed-XCD instances %d (inst0: %d %d)\n", p.inst, p.G, p.K, p.same_xcd, c[4],
               c[5], c[6], mx * 0.01 / rounds, fails, mixed, xcc[0], xcc[p.G - 1]);
        CK(hipFree(p.xcc));
        CK(hipFree(p.mail));
        CK(hipFree(p.out));
    }
    return 0;
}
