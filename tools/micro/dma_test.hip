// dma_test.hip -- checks the direct-to-LDS row request used by cols_search.hpp in isolation:
// 1024 threads request a 2048-double row into an LDS slot at a given offset, wait, barrier, read
// it back and compare with the source.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../gnn-accelerated-lap-warm-start-pipeline_amd/csrc/cols_search.hpp"
using namespace lapwarm;
using namespace lapwarm::cols;

__global__ void __launch_bounds__(1024) k(const double *C, int n, int row, int slot_off, double *out, unsigned *dbg)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (slot_off + n * 8) / 8 + 64; i += nt) reinterpret_cast<double *>(smem)[i] = -1.0;
    __syncthreads();
    const double *r = C + (size_t)row * n;
    const unsigned sbase = lds_address(smem + slot_off) + (unsigned)wave * 1024u;
    if (tid == 0) dbg[0] = lds_address(smem), dbg[1] = sbase;
    dma_request16(r + 2 * tid, sbase);
    dma_request16(C, lds_address(smem + slot_off + n * 8 + 0));  // a younger dummy (to one spot)
    dma_wait<1>();
    __syncthreads();
    const double2 t = *reinterpret_cast<const double2 *>(smem + slot_off + (size_t)tid * 16);
    out[2 * tid] = t.x;
    out[2 * tid + 1] = t.y;
}

int main()
{
    const int n = 2048;
    std::vector<double> h((size_t)n * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)i + 0.5;
    double *C, *out;
    unsigned *dbg;
    hipMalloc(&C, h.size() * 8);
    hipMalloc(&out, n * 8);
    hipMalloc(&dbg, 64);
    hipMemcpy(C, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int slot_off : {0, 16384, 49152, 98304}) {
        hipMemset(out, 0, n * 8);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
        hipLaunchKernelGGL(k, dim3(1), dim3(1024), 150000, 0, C, n, 2, slot_off, out, dbg);
        hipDeviceSynchronize();
        std::vector<double> o(n);
        unsigned d[2];
        hipMemcpy(o.data(), out, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(d, dbg, 8, hipMemcpyDeviceToHost);
        int bad = 0, first = -1;
        for (int j = 0; j < n; ++j)
            if (o[j] != h[(size_t)2 * n + j]) {
                if (first < 0) first = j;
                ++bad;
            }
        printf("slot_off=%6d lds_base=%u sbase0=%u bad=%d first_bad=%d got=%.1f want=%.1f\n", slot_off, d[0], d[1], bad, first,
               first >= 0 ? o[first] : 0.0, first >= 0 ? h[(size_t)2 * n + first] : 0.0);
    }
    return 0;
}
