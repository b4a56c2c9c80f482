// micro-latency probes for gfx950: dependent LDS reads, readlane chains, scalar ALU chains (one wave per workgroup)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(64) probe(long long *out, int n, int seed) {
    __shared__ int lds[4096];
    __shared__ int4 lds4[1024];
    int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = (i * 7 + seed) & 4095;
    for (int i = lane; i < 1024; i += 64) lds4[i] = make_int4((i * 5 + seed) & 1023, i, i, i);
    __syncthreads();
    long long t0, t1; int idx = seed & 4095;
    // 1. dependent ds_read_b32 chain (uniform address)
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) idx = lds[idx];
    asm volatile("" :: "v"(idx));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[0] = (t1 - t0) / n;
    // 2. dependent ds_read_b128 chain
    int j = seed & 1023;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) j = lds4[j].x;
    asm volatile("" :: "v"(j));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[1] = (t1 - t0) / n;
    // 3. dependent ds_read through readfirstlane (LDS value -> SGPR -> address)
    int s = seed & 4095;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) s = __builtin_amdgcn_readfirstlane(lds[s]);
    asm volatile("" :: "s"(s));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[2] = (t1 - t0) / n;
    // 4. VALU dependent chain
    int v = lane + seed;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) { v = v * 3 + 1; asm volatile("" : "+v"(v)); }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[3] = (t1 - t0) / n;
    // 5. readlane with SGPR index chain
    int r = seed & 63; int vv = (lane * 13 + 5) & 63;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) r = __builtin_amdgcn_readlane(vv, r);
    asm volatile("" :: "s"(r));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[4] = (t1 - t0) / n;
    // 6. back-to-back s_memtime
    t0 = __builtin_amdgcn_s_memtime(); t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[5] = t1 - t0;
    // 7. independent VALU stream (4 chains)
    int a = lane, b2 = lane + 1, c = lane + 2, d = lane + 3;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) { a = a * 3 + 1; b2 = b2 * 5 + 1; c = c * 7 + 1; d = d * 9 + 1; asm volatile("" : "+v"(a), "+v"(b2), "+v"(c), "+v"(d)); }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[6] = (t1 - t0) / n;
    // 8. SALU dependent chain
    int sa = seed;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) { sa = sa * 3 + 1; asm volatile("" : "+s"(sa)); }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[7] = (t1 - t0) / n;
    // 9. LDS write then read other lane's value (wave barrier-free)
    t0 = __builtin_amdgcn_s_memtime();
    int w = lane;
    for (int i = 0; i < n; ++i) { lds[lane] = w; w = lds[(lane + 1) & 63] + 1; }
    asm volatile("" :: "v"(w));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[8] = (t1 - t0) / n;
    if (lane == 0) out[9] = idx + j + s + v + r + a + b2 + c + d + sa + w;
}
int main() {
    long long *d, h[10];
    hipMalloc(&d, 80);
    for (int grid : {1, 256, 1024, 4096}) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(64), 0, 0, d, 2000, 17);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 80, hipMemcpyDeviceToHost);
        printf("grid %4d: ds_b32 %lld  ds_b128 %lld  ds->readfirstlane %lld  valu_dep %lld  readlane_sgpr %lld  memtime_pair %lld  valu_4chain %lld  salu_dep %lld  lds_wr_rd %lld (ticks per iteration)\n",
               grid, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
    }
    return 0;
}
