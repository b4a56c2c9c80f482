// branch-cost probes for gfx950 (one wave per workgroup)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(64) probe(long long *out, int n, int seed, int zero) {
    int lane = threadIdx.x;
    long long t0, t1;
    // A: 16 dependent SALU ops, straight line, per iteration
    int sa = seed;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { sa = sa * 3 + 1; asm volatile("" : "+s"(sa)); }
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[0] = (t1 - t0) / n;
    // B: same with a not-taken uniform branch after every op
    int sb = seed;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { sb = sb * 3 + 1; asm volatile("" : "+s"(sb)); if (__builtin_expect(sb == zero + 123456789, 0)) { out[9] = sb; asm volatile("s_nop 0"); } }
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[1] = (t1 - t0) / n;
    // C: same with a TAKEN forward branch (skips a block) after every op
    int sc = seed;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { sc = sc * 3 + 1; asm volatile("" : "+s"(sc)); if (__builtin_expect(sc != zero + 123456789, 0)) {} else { out[9] = sc; asm volatile("s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0"); } }
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[2] = (t1 - t0) / n;
    // D: 16 dependent VALU ops straight line
    int v = lane + seed;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { v = v + (v >> 3); asm volatile("" : "+v"(v)); }
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[3] = (t1 - t0) / n;
    // E: 16 independent VALU ops (4 chains x 4)
    int a = lane, b = lane + 1, c = lane + 2, d = lane + 3;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { a += a >> 3; b += b >> 3; c += c >> 3; d += d >> 3; asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[4] = (t1 - t0) / n;
    // F: empty loop iteration cost
    int cnt = 0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) { asm volatile("" : "+s"(cnt)); }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[5] = (t1 - t0) * 100 / n;
    // G: v_readlane const x16 dependent-free
    int rl = 0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { rl += __builtin_amdgcn_readlane(v, u); }
        asm volatile("" : "+s"(rl));
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[6] = (t1 - t0) / n;
    // H: DPP max chain x16
    int dp = lane;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { int t = __builtin_amdgcn_update_dpp(dp, dp, 0x111, 0xF, 0xF, false); dp = dp > t ? dp : t + 1; }
        asm volatile("" : "+v"(dp));
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[7] = (t1 - t0) / n;
    if (lane == 0) out[8] = sa + sb + sc + v + a + b + c + d + cnt + rl + dp;
}
int main() {
    long long *d, h[10];
    hipMalloc(&d, 80);
    hipLaunchKernelGGL(probe, dim3(256), dim3(64), 0, 0, d, 1000, 17, 0);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 80, hipMemcpyDeviceToHost);
    printf("per 16 ops: salu_dep %lld | +not-taken branches %lld | +taken branches %lld | valu_dep %lld | valu_indep %lld | empty loop iter x100 %lld | 16 readlane %lld | 16 dpp+max %lld\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    return 0;
}
