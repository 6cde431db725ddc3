// chain_bench.hip -- latency floor of the relax chain: one 1024-thread workgroup per matrix does
// `steps` DEPENDENT row reads (next row index comes from the data just read, through LDS + one
// barrier), each thread loading 2 doubles of the row.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void __launch_bounds__(1024) chain(const double *C, int n, int steps, int row_limit, int barriers,
                                              int prefetch, unsigned long long *ticks, double *sink)
{
    __shared__ int next_row;
    __shared__ int next2;
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *base = C + (size_t)b * n * n;
    int row = b % row_limit;
    double acc = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; ++s) {
        const double *rp = base + (size_t)row * n;
        const double a = rp[2 * tid], c = rp[2 * tid + 1];
        acc += a - c;
        if (tid == 0) {
            unsigned long long bits = __double_as_longlong(a) ^ (unsigned long long)(s * 2654435761u);
            bits ^= bits >> 29;
            bits *= 0x9E3779B97F4A7C15ULL;
            bits ^= bits >> 32;
            next_row = (int)(bits % (unsigned)row_limit);
            next2 = (int)((bits >> 20) % (unsigned)row_limit);
        }
        __syncthreads();
        row = next_row;
        if (prefetch) {  // touch the row after next (independent of this step's data)
            const double *pp = base + (size_t)next2 * n;
            int junk;
            asm volatile("global_load_dword %0, %1, off" : "=v"(junk) : "v"(pp + 2 * tid) : "memory");
            asm volatile("" ::"v"(junk));
        }
        for (int q = 1; q < barriers; ++q) __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) ticks[b] = t1 - t0;
    sink[(size_t)b * 1024 + tid] = acc;
}

int main(int argc, char **argv)
{
    const int nmat = argc > 1 ? atoi(argv[1]) : 32;
    const int n = 2048, steps = 20000;
    double *C, *sink;
    unsigned long long *ticks;
    hipMalloc(&C, sizeof(double) * (size_t)nmat * n * n);
    hipMalloc(&sink, sizeof(double) * (size_t)nmat * 1024);
    hipMalloc(&ticks, sizeof(unsigned long long) * nmat);
    std::vector<double> h((size_t)n * n);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)rand() / RAND_MAX;
    for (int b = 0; b < nmat; ++b) hipMemcpy(C + (size_t)b * n * n, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    const int limits[] = {2048, 256, 16};
    for (int limit : limits)
        for (int barriers : {1, 3, 5}) {  // 5 = one barrier + prefetch of an independent row
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(chain, dim3(nmat), dim3(1024), 0, 0, C, n, steps, limit, barriers & 3, barriers >> 2, ticks, sink);
                hipDeviceSynchronize();
            }
            std::vector<unsigned long long> t(nmat);
            hipMemcpy(t.data(), ticks, sizeof(unsigned long long) * nmat, hipMemcpyDeviceToHost);
            double mx = 0, sum = 0;
            for (auto v : t) {
                sum += v;
                if (v > mx) mx = v;
            }
            printf("nmat=%d rows_touched=%4d barriers=%d : %.3f us/step mean, %.3f max\n", nmat, limit, barriers,
                   sum / nmat * 0.01 / steps, mx * 0.01 / steps);
        }
    return 0;
}
