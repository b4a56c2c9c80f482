// Measured roofs of one MI355X for this engine's kernels (VERDICT round 4, item 2 / "missing" 6; SURVEY.md section 8d: "calibrate the
// denominator with a device copy kernel"):
//   * integer VALU issue: independent streams of v_add_u32 / v_max_i32 / v_pk_max_i16 / v_max_i32 with a DPP row_shr operand / s_add_u32, and a
//     1:1 VALU + SALU mix, at 1, 2, 4 and 8 wavefronts per SIMD on every CU -> wave-instructions per cycle per SIMD and 32-bit lane-ops per second
//     (a wave64 instruction = 64 lane-ops; v_pk_* = 128 16-bit lane-ops, reported as 64 "lanes" x 2);
//   * HBM: device copy (read + write), read-only sum, write-only fill over buffers far larger than the 256 MB of MALL.
// Prints ONE JSON object; tools/ubench/run_roofs.sh stores it as profiles/r5_roofs.json, which bench.py reads for its peaks.
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/roofs tools/ubench/roofs.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int UNROLL = 32;      // independent instructions per loop trip (8 registers x 4 passes: no instruction depends on the one before it)

// KIND: 0 v_add_u32, 1 v_max_i32, 2 v_pk_max_i16, 3 v_max_i32 DPP row_shr:1, 4 s_add_u32, 5 v_add_u32 + s_add_u32 interleaved 1:1, 6 v_add3_u32 (VOP3, three sources),
//       7-14 further single ops, 15 the row loops' instruction mix, 16 v_readlane_b32
template <int KIND>
__global__ void __launch_bounds__(256) issue_probe(long long *cycles, int *sink, int trips) {
    int v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7, k = 3;
    int s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7;
    asm volatile("" : "+v"(k), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7));
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int i = 0; i < trips; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; ++u) {
            if (KIND == 0) asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                                        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k));
            if (KIND == 1) asm volatile("v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8"
                                        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k));
            if (KIND == 2) asm volatile("v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8"
                                        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k));
            if (KIND == 3) asm volatile("v_max_i32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                                        "v_max_i32_dpp %2, %8, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %3, %8, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                                        "v_max_i32_dpp %4, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %5, %8, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                                        "v_max_i32_dpp %6, %8, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %7, %8, %7 row_shr:1 row_mask:0xf bank_mask:0xf"
                                        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k));
            if (KIND == 4) asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 3\n s_add_u32 %3, %3, 3\n s_add_u32 %4, %4, 3\n s_add_u32 %5, %5, 3\n s_add_u32 %6, %6, 3\n s_add_u32 %7, %7, 3"
                                        : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) :: "scc");
            if (KIND == 5) asm volatile("v_add_u32 %0, %0, %8\n s_add_u32 %9, %9, 3\n v_add_u32 %1, %1, %8\n s_add_u32 %10, %10, 3\n v_add_u32 %2, %2, %8\n s_add_u32 %11, %11, 3\n v_add_u32 %3, %3, %8\n s_add_u32 %12, %12, 3"
                                        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(k), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
#define OP8(OPSTR) asm volatile(OPSTR " %0, %0, %8\n " OPSTR " %1, %1, %8\n " OPSTR " %2, %2, %8\n " OPSTR " %3, %3, %8\n " OPSTR " %4, %4, %8\n " OPSTR " %5, %5, %8\n " OPSTR " %6, %6, %8\n " OPSTR " %7, %7, %8" \
                                        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k))
            if (KIND == 7) OP8("v_sub_u32");
            if (KIND == 8) OP8("v_min_i32");
            if (KIND == 9) OP8("v_and_b32");
            if (KIND == 10) OP8("v_lshlrev_b32");
            if (KIND == 11) OP8("v_pk_add_i16");
            if (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n"
                                         "v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k) : "vcc");
            if (KIND == 13) asm volatile("v_cmp_gt_i32 vcc, %0, %8\n v_cmp_gt_i32 vcc, %1, %8\n v_cmp_gt_i32 vcc, %2, %8\n v_cmp_gt_i32 vcc, %3, %8\n v_cmp_gt_i32 vcc, %4, %8\n v_cmp_gt_i32 vcc, %5, %8\n v_cmp_gt_i32 vcc, %6, %8\n"
                                         "v_cmp_gt_i32 vcc, %7, %8" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k) : "vcc");
            if (KIND == 14) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                                         "v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                                         "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf"
                                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k));
            // the row loops' mix: add, max, compare + select, shift-or, DPP max, subtract, min (one of each + one more max)
            if (KIND == 15) asm volatile("v_add_u32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_cmp_gt_i32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_lshl_or_b32 %4, %4, 3, %8\n"
                                         "v_max_i32_dpp %5, %8, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_sub_u32 %6, %6, %8\n v_min_i32 %7, %7, %8"
                                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k) : "vcc");
            if (KIND == 16) asm volatile("v_readlane_b32 %8, %0, 3\n v_readlane_b32 %9, %1, 3\n v_readlane_b32 %10, %2, 3\n v_readlane_b32 %11, %3, 3\n v_readlane_b32 %8, %4, 3\n v_readlane_b32 %9, %5, 3\n v_readlane_b32 %10, %6, 3\n"
                                         "v_readlane_b32 %11, %7, 3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
            if (KIND == 6) asm volatile("v_add3_u32 %0, %0, %8, %8\n v_add3_u32 %1, %1, %8, %8\n v_add3_u32 %2, %2, %8, %8\n v_add3_u32 %3, %3, %8, %8\n v_add3_u32 %4, %4, %8, %8\n v_add3_u32 %5, %5, %8, %8\n v_add3_u32 %6, %6, %8, %8\n v_add3_u32 %7, %7, %8, %8"
                                        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k));
        }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 == 0x7fffffff) sink[0] = 1;
}

__global__ void __launch_bounds__(256) copy_kernel(const int4 *__restrict__ src, int4 *__restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n; i += 4 * stride) {      // four loads in flight per lane
        const int4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) read_kernel(const int4 *__restrict__ src, int *sink, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; const size_t stride = (size_t)gridDim.x * 256; int acc = 0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const int4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    for (; i < n; i += stride) acc += src[i].x;
    if (acc == 0x12345678) sink[0] = acc;
}
__global__ void __launch_bounds__(256) fill_kernel(int4 *__restrict__ dst, size_t n, int v) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; const size_t stride = (size_t)gridDim.x * 256; const int4 x = make_int4(v, v + 1, v + 2, v + 3);
    for (; i < n; i += stride) dst[i] = x;
}

template <int KIND>
static std::string run_issue(const char *name, int n_cu, long long *d_cyc, int *d_sink, int lanes_factor) {
    std::string out = std::string("\"") + name + "\": {";
    const int trips = 20000;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = n_cu * wps;      // 256 threads = one wavefront on each of the CU's four SIMDs; `wps` workgroups per CU
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(issue_probe<KIND>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, 200); CK(hipDeviceSynchronize());      // warm-up
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(issue_probe<KIND>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, trips);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> cyc((size_t)blocks * 4); CK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
        double mean = 0; for (long long c : cyc) mean += (double)c; mean /= (double)cyc.size();
        const double insts_per_wave = (double)trips * UNROLL;
        const double cyc_per_inst_wave = mean / insts_per_wave;                       // what ONE wavefront sees
        const double inst_per_cyc_simd = wps / cyc_per_inst_wave;                     // what the SIMD issues (all its wavefronts)
        const double total_insts = insts_per_wave * (double)blocks * 4.0;
        const double tlaneops = total_insts * 64.0 * lanes_factor / (ms * 1e-3) / 1e12;      // wall-clock rate over the whole chip
        char buf[256];
        snprintf(buf, sizeof(buf), "%s\"w%d\": {\"cycles_per_inst_per_wave\": %.3f, \"inst_per_cycle_per_simd\": %.3f, \"T_lane_ops_per_s\": %.2f, \"ms\": %.3f}", wps == 1 ? "" : ", ", wps,
                 cyc_per_inst_wave, inst_per_cyc_simd, tlaneops, ms);
        out += buf;
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    return out + "}";
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    long long *d_cyc; int *d_sink; CK(hipMalloc(&d_cyc, (size_t)n_cu * 8 * 4 * 8)); CK(hipMalloc(&d_sink, 64)); CK(hipMemset(d_sink, 0, 64));
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"n_cu\": %d, \"clock_mhz\": %d, \"memtime_note\": \"s_memtime ticks = shader cycles (MI355X_MICROARCH.md)\",\n \"issue\": {\n  ", prop.name,
           prop.gcnArchName, n_cu, prop.clockRate / 1000);
    printf("%s,\n  ", run_issue<0>("v_add_u32", n_cu, d_cyc, d_sink, 1).c_str());
    printf("%s,\n  ", run_issue<1>("v_max_i32", n_cu, d_cyc, d_sink, 1).c_str());
    printf("%s,\n  ", run_issue<2>("v_pk_max_i16", n_cu, d_cyc, d_sink, 2).c_str());
    printf("%s,\n  ", run_issue<3>("v_max_i32_dpp_row_shr", n_cu, d_cyc, d_sink, 1).c_str());
    printf("%s,\n  ", run_issue<6>("v_add3_u32", n_cu, d_cyc, d_sink, 1).c_str());
    printf("%s,\n  ", run_issue<7>("v_sub_u32", n_cu, d_cyc, d_sink, 1).c_str());
    printf("%s,\n  ", run_issue<8>("v_min_i32", n_cu, d_cyc, d_sink, 1).c_str());
    printf("%s,\n  ", run_issue<9>("v_and_b32", n_cu, d_cyc, d_sink, 1).c_str());
    printf("%s,\n  ", run_issue<10>("v_lshlrev_b32", n_cu, d_cyc, d_sink, 1).c_str());
    printf("%s,\n  ", run_issue<11>("v_pk_add_i16", n_cu, d_cyc, d_sink, 2).c_str());
    printf("%s,\n  ", run_issue<12>("v_cndmask_b32", n_cu, d_cyc, d_sink, 1).c_str());
    printf("%s,\n  ", run_issue<13>("v_cmp_gt_i32", n_cu, d_cyc, d_sink, 1).c_str());
    printf("%s,\n  ", run_issue<14>("v_mov_b32_dpp_row_shr", n_cu, d_cyc, d_sink, 1).c_str());
    printf("%s,\n  ", run_issue<16>("v_readlane_b32", n_cu, d_cyc, d_sink, 0).c_str());
    printf("%s,\n  ", run_issue<15>("row_loop_mix(add,max,cmp,cndmask,lshl_or,max_dpp,sub,min)", n_cu, d_cyc, d_sink, 1).c_str());
    printf("%s,\n  ", run_issue<4>("s_add_u32", n_cu, d_cyc, d_sink, 0).c_str());
    printf("%s\n },\n", run_issue<5>("v_add_u32+s_add_u32", n_cu, d_cyc, d_sink, 1).c_str());
    // ---- HBM
    const size_t bytes = (size_t)4 << 30, n16 = bytes / 16;
    int4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    hipLaunchKernelGGL(fill_kernel, dim3(n_cu * 8), dim3(256), 0, 0, a, n16, 1); hipLaunchKernelGGL(fill_kernel, dim3(n_cu * 8), dim3(256), 0, 0, b, n16, 2); CK(hipDeviceSynchronize());
    auto timed = [&](auto launch, int reps) { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); launch(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); return (double)ms / reps; };
    double best_copy = 0, best_read = 0, best_fill = 0; int g_copy = 0, g_read = 0, g_fill = 0;
    for (int per_cu : {4, 8, 16, 32}) {
        const int grid = n_cu * per_cu;
        const double ms_c = timed([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n16); }, 5);
        const double ms_r = timed([&] { hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, d_sink, n16); }, 5);
        const double ms_f = timed([&] { hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(256), 0, 0, b, n16, 3); }, 5);
        const double gc = 2.0 * bytes / (ms_c * 1e-3) / 1e9, gr = (double)bytes / (ms_r * 1e-3) / 1e9, gf = (double)bytes / (ms_f * 1e-3) / 1e9;
        if (gc > best_copy) { best_copy = gc; g_copy = grid; } if (gr > best_read) { best_read = gr; g_read = grid; } if (gf > best_fill) { best_fill = gf; g_fill = grid; }
    }
    printf(" \"hbm\": {\"buffer_bytes\": %zu, \"copy_read_plus_write_GBps\": %.1f, \"copy_grid\": %d, \"read_GBps\": %.1f, \"read_grid\": %d, \"fill_GBps\": %.1f, \"fill_grid\": %d, \"spec_GBps\": 8000}\n}\n", bytes,
           best_copy, g_copy, best_read, g_read, best_fill, g_fill);
    return 0;
}
