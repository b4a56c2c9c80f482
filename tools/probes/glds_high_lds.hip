// Probe: does global_load_lds_dwordx4 (LDS-DMA, base in M0) reach LDS addresses above 64 KB on gfx950?  One workgroup with 128 KB of dynamic LDS copies
// 1 KB pieces from a global pattern to LDS offsets 0 KB ... 127 KB by LDS-DMA and reads them back with ds_read.  Prints the first offset that differs.
// build + run (GPU box): hipcc --offload-arch=gfx950 -O2 -o /tmp/glds_probe tools/probes/glds_high_lds.hip && /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
__global__ void __launch_bounds__(64) probe(const int4 *src, int *bad, int n_kb) {
    const int lane = threadIdx.x;
    for (int k = 0; k < n_kb; ++k)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + k * 64 + lane), (__attribute__((address_space(3))) void *)(lds + k * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; k < n_kb; ++k) {
        const int4 v = ((const int4 *)(lds + k * 1024))[lane], w = src[k * 64 + lane];
        if (v.x != w.x || v.y != w.y || v.z != w.z || v.w != w.w) atomicMin(bad, k);
    }
}
int main() {
    const int n_kb = 128;
    std::vector<int> h(n_kb * 256); for (size_t i = 0; i < h.size(); ++i) h[i] = (int)(i * 2654435761u);
    int4 *d; int *bad, hb = 1 << 30;
    hipMalloc(&d, h.size() * 4); hipMalloc(&bad, 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(bad, &hb, 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, n_kb * 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), n_kb * 1024, 0, d, bad, n_kb);
    hipError_t e = hipDeviceSynchronize(); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("launch: %s; first 1 KB piece that differs: %s (%d)\n", hipGetErrorString(e), hb == (1 << 30) ? "none -- LDS-DMA reaches all 128 KB" : "KB offset", hb);
    return 0;
}
