// How many single-wavefront workgroups of a given dynamic-LDS size does a gfx950 CU hold?  Each workgroup spins ~2 ms; a launch of N x 256 workgroups
// that takes ~2 ms holds N per CU, ~4 ms means the last ones waited for a slot.  (hipOccupancyMaxActiveBlocksPerMultiprocessor is printed beside it.)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_granule tools/probes/lds_granule.hip && /tmp/lds_granule
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ unsigned char lds_raw[];
__global__ void __launch_bounds__(64) spin(long long ticks, int *sink) {
    lds_raw[threadIdx.x] = (unsigned char)threadIdx.x;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
    if (lds_raw[threadIdx.x] == 255 && threadIdx.x == 77) *sink = 1;
}
int main() {
    int *sink; hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int sizes[] = {40960, 41216, 53248, 53760, 54080, 54272, 54528, 54613, 65536, 81920};
    for (int s : sizes) for (int per_cu = 2; per_cu <= 4; ++per_cu) {
        if ((long long)s * per_cu > 163840) continue;
        int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, 64, (size_t)s);
        hipLaunchKernelGGL(spin, dim3(256), dim3(64), s, 0, 1000LL, sink); hipDeviceSynchronize();
        hipEventRecord(a); hipLaunchKernelGGL(spin, dim3(256 * per_cu), dim3(64), s, 0, 200000LL, sink); hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        printf("lds %6d B x %d per CU (%4d workgroups): %.2f ms   (occupancy API: %d per CU)\n", s, per_cu, 256 * per_cu, ms, occ);
    }
    return 0;
}
