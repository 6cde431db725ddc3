// mlp_bench.hip -- how fast can ONE CU pull 16-KB rows from HBM as a function of the rows it
// keeps in flight?  One workgroup per matrix (32 matrices x 32 MiB = 1 GiB, nothing cache
// resident), each step loads K random rows completely (every thread its slice of each row), then
// one barrier.  Reports microseconds per ROW.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int K, int VEC>
__global__ void __launch_bounds__(1024) rows_kernel(const double *C, int n, int steps, const int *rowseq,
                                                    unsigned long long *ticks, double *sink)
{
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const double *base = C + (size_t)b * n * n;
    const int per = n / nt;  // doubles per thread per row
    double acc = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; ++s) {
        double v[K][8];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int row = rowseq[(s * K + k) & 65535];
            const double *rp = base + (size_t)row * n;
            if (VEC == 2) {
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    if (q < per) {
                        const double2 t = *reinterpret_cast<const double2 *>(rp + (q / 2) * 2 * nt + 2 * tid);
                        v[k][q] = t.x;
                        v[k][q + 1] = t.y;
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < per) v[k][q] = rp[q * nt + tid];
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < per) acc += v[k][q];
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) ticks[b] = t1 - t0;
    sink[(size_t)b * 1024 + tid] = acc;
}

template <int K, int VEC>
void run(const double *C, int n, int nmat, int threads, const int *rowseq, unsigned long long *ticks, double *sink)
{
    const int steps = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((rows_kernel<K, VEC>), dim3(nmat), dim3(threads), 0, 0, C, n, steps, rowseq, ticks, sink);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> t(nmat);
    hipMemcpy(t.data(), ticks, sizeof(unsigned long long) * nmat, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : t) sum += v;
    const double us_row = sum / nmat * 0.01 / steps / K;
    printf("wgs=%3d threads=%4d vec=%d rows_in_flight=%d : %.3f us/row  (%.1f GB/s per CU)\n", nmat, threads, VEC, K,
           us_row, n * 8.0 / us_row * 1e-3);
}

int main(int argc, char **argv)
{
    const int nmat = argc > 1 ? atoi(argv[1]) : 32;
    const int n = 2048;
    double *C, *sink;
    unsigned long long *ticks;
    int *rowseq;
    hipMalloc(&C, sizeof(double) * (size_t)nmat * n * n);
    hipMalloc(&sink, sizeof(double) * (size_t)nmat * 1024);
    hipMalloc(&ticks, sizeof(unsigned long long) * nmat);
    hipMalloc(&rowseq, sizeof(int) * 65536);
    std::vector<double> h((size_t)n * n);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)rand() / RAND_MAX;
    for (int b = 0; b < nmat; ++b)
        hipMemcpy(C + (size_t)b * n * n, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    std::vector<int> rs(65536);
    for (auto &r : rs) r = rand() % n;
    hipMemcpy(rowseq, rs.data(), sizeof(int) * 65536, hipMemcpyHostToDevice);
    for (int threads : {1024, 512, 256}) {
        run<1, 1>(C, n, nmat, threads, rowseq, ticks, sink);
        run<2, 1>(C, n, nmat, threads, rowseq, ticks, sink);
        run<4, 1>(C, n, nmat, threads, rowseq, ticks, sink);
        if (threads == 1024) run<8, 1>(C, n, nmat, threads, rowseq, ticks, sink);
        run<1, 2>(C, n, nmat, threads, rowseq, ticks, sink);
        run<2, 2>(C, n, nmat, threads, rowseq, ticks, sink);
        run<4, 2>(C, n, nmat, threads, rowseq, ticks, sink);
    }
    return 0;
}
