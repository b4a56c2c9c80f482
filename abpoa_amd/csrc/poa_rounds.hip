// All rounds of a read-set in ONE kernel (device-resident driver, narrow bands): the workgroup that owns a read-set takes it through
//     graph -> DP rows (poa_prepare_body), row loop, global best + backtrack, cigar -> graph (poa_fuse_body)
// for read k_lo, k_lo + 1, ... without returning to the host.  With one launch per phase and round (poa_device.hip + the dp_fast kernels) every
// round lasts as long as its SLOWEST alignment; the alignments of a round differ by +-11 % (1000 x 1 kb sets: slowest 4.7 M cycles, mean 3.5 M)
// while whole sets differ by only ~4 %, so a set that advances on its own finishes in the time of the slowest SET, not in the sum of the per-round
// maxima (measured: 201 M vs 163 M cycles of row loop + backtrack per set).
// Shape: GT = 256 threads per read-set as in the graph kernels; the row loop and the backtrack are one-wavefront code (rows_fast.h, backtrack.h) and
// run on wavefront 0 while the other three wait at the workgroup barrier (ABPOA_HIP_ONE_WAVE_PHASE turns the barriers inside those phases into plain
// waits).  Hand-over between phases is through HBM exactly as between the separate kernels; all of it is written and read by this one workgroup
// (one CU, one L1), so a workgroup barrier with vmcnt(0) is all the ordering it needs.
#define ABPOA_HIP_ONE_WAVE_PHASE 1
#include "fast_tail.h"
#include "poa_bodies.h"
#include "msa_device.h"      // MSA_DEVICE_SLOTS

namespace abpoa_hip {

// Kernel arguments live in constant memory (one record per device queue of the host, msa_device.h MSA_DEVICE_SLOTS): the phases are separate,
// NOT inlined functions -- each gets the register allocation it has as a kernel of its own (the row loop alone fills the scalar registers) --
// and constant memory is where a called function can read launch-wide values with scalar loads.
struct RoundsArgs { PoaDev p; DevBatch b; int32_t *cu_ticket; };
__constant__ RoundsArgs g_rounds[MSA_DEVICE_SLOTS];

// (arguments of a called function arrive in vector registers: v_readfirstlane tells the compiler they are wave-uniform, so that everything derived
//  from them -- the argument record, the set's slices of the pools -- stays in scalar registers as in a kernel of its own)
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
__device__ __noinline__ void rounds_prepare(const int slot, const int s, const int k) { poa_prepare_body(g_rounds[UNI(slot)].p, UNI(s), UNI(k)); }
__device__ __noinline__ void rounds_fuse(const int slot, const int s, const int k) { poa_fuse_body(g_rounds[UNI(slot)].p, UNI(s), UNI(k)); }
// the alignment descriptor of the round was written by this workgroup a moment ago, so its load is a vector load; its values are the same in
// every lane and belong in scalar registers (loop bounds, base addresses, branch conditions of the one-wave phases)
__device__ __forceinline__ AlnDesc uniform_desc(const AlnDesc *g) {
    static_assert(sizeof(AlnDesc) % 4 == 0, "AlnDesc is read as 32-bit words");
    AlnDesc d; const int *src = (const int *)g; int *dst = (int *)&d;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(AlnDesc) / 4); ++i) dst[i] = __builtin_amdgcn_readfirstlane(src[i]);
    return d;
}
// row loop, then global best + backtrack, on the calling wavefront; returns the clock between the two
// (DIR: direction-plane arenas, rows_fast.h DirFmt / backtrack_dir.h -- what the device-resident driver uses whenever the penalties allow it)
template <typename T, int GAP, bool DIR>
__device__ __noinline__ long long rounds_rows(const int slot, const int s_) {
    const DevBatch &b = g_rounds[UNI(slot)].b; const int s = UNI(s_);
    const AlnDesc d = uniform_desc(b.aln + s);
    align_fast_rows<T, GAP, 1, false, DIR>(b, d, b.out + s);
    return (long long)__builtin_amdgcn_s_memtime();
}
// role 0: the wavefront that ran the row loop; role 1: a second wavefront of the workgroup that walks the lower half of the graph at the same time
// (backtrack_dir.h); ctl: eight ints of static LDS; gen: the round
template <typename T, int GAP, bool DIR>
__device__ __noinline__ void rounds_tail(const int slot, const int s_, const int role_, int *ctl, const int gen_) {
    const DevBatch &b = g_rounds[UNI(slot)].b; const int s = UNI(s_);
    const AlnDesc d = uniform_desc(b.aln + s);
    align_fast_tail<T, GAP, DIR>(b, d, b.out + s, UNI(role_), ctl, UNI(gen_));
}

template <int GAP>
__global__ void __launch_bounds__(GT, 4) poa_rounds_kernel(const int slot, const int k_lo) {
    const int s = blockIdx.x, tid = threadIdx.x;
    const PoaDev &p = g_rounds[slot].p; const DevBatch &b = g_rounds[slot].b;
    if (s >= p.n_sets) return;
    const int n_reads = p.sets[s].n_reads;
    PoaState *st = p.state + s;
    long long t_prep = 0, t_rows = 0, t_tail = 0, t_fuse = 0;
    // Which wavefront runs the one-wave phases.  A workgroup's four wavefronts sit on the four SIMDs of its CU; if it were always wavefront 0, the
    // four workgroups of a CU would run their row loops on the same SIMD while the other three idle.  Each workgroup draws a ticket from a per-CU
    // counter and the wavefront on SIMD (ticket mod 4) does the work.
    __shared__ int sh_simd[GW], sh_target, sh_walk[8 * SPEC_WK];      // sh_walk: hand-over between the wavefronts of a backtrack
    if (tid < 8 * SPEC_WK) sh_walk[tid] = 0;
    {
        const unsigned hwid = __builtin_amdgcn_s_getreg(63492);      // HW_REG_HW_ID: simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]
        if ((tid & 63) == 0) sh_simd[tid >> 6] = (int)((hwid >> 4) & 3);
        // HW_REG_XCC_ID [3:0]
        if (tid == 0) { const unsigned xcc = __builtin_amdgcn_s_getreg(6164) & 15; sh_target = atomicAdd(g_rounds[slot].cu_ticket + ((xcc << 8) | ((hwid >> 8) & 0xff)), 1) & 3; }
        __syncthreads();
    }
    if (tid == 0) st->algo_bytes_before = st->algo_bytes;
    int worker = 0;
#pragma unroll
    for (int w_ = GW - 1; w_ >= 0; --w_) if (sh_simd[w_] == sh_target) worker = w_;
    static_assert(GW == SPEC_WK, "one wavefront of the workgroup per walk of a backtrack");
    const bool pair = !(b.dbg & 1024);               // (ABPOA_HIP_DBG bit 10: one wavefront per backtrack, for comparison)
    for (int k = k_lo; k < n_reads; ++k) {
        const long long c0 = (long long)__builtin_amdgcn_s_memtime();
        rounds_prepare(slot, s, k);
        __syncthreads();                                    // descriptor, row tables and predecessor lists of this round are complete
        if (ld_fresh(&st->status) != POA_ST_OK) break;      // capacity exceeded (or about to be): the set goes to the second pass / the host driver
        const long long c1 = (long long)__builtin_amdgcn_s_memtime();
        long long c2 = c1;
        if ((tid >> 6) == worker) {
            const int bits = b.aln[s].bits, w = b.aln[s].w, flags = b.aln[s].flags;
            // (the host launches this kernel only for jobs whose reads all take the narrow loop)
            if ((flags & ALN_FAST_OK) && !(b.lds.wide_nw >= 1 && w >= b.lds.wide_w_lo && w <= b.lds.wide_w_hi)) {
                bool done_dir = false;
                if constexpr (GAP != 0) if (b.dir_mode) {      // (linear gaps keep H records: no direction words)
                    if (bits == 16) { c2 = rounds_rows<int16_t, GAP, true>(slot, s); rounds_tail<int16_t, GAP, true>(slot, s, pair ? 0 : -1, sh_walk, k); }
                    else { c2 = rounds_rows<int32_t, GAP, true>(slot, s); rounds_tail<int32_t, GAP, true>(slot, s, pair ? 0 : -1, sh_walk, k); }
                    done_dir = true;
                }
                if (done_dir) {}
                else if (bits == 16) { c2 = rounds_rows<int16_t, GAP, false>(slot, s); rounds_tail<int16_t, GAP, false>(slot, s, -1, sh_walk, k); }
                else { c2 = rounds_rows<int32_t, GAP, false>(slot, s); rounds_tail<int32_t, GAP, false>(slot, s, -1, sh_walk, k); }
            } else if ((tid & 63) == 0) b.out[s].status = ABPOA_HIP_EINVAL;       // -> poa_fuse_body marks the set for the fall-back
        } else if (GAP != 0 && b.dir_mode && pair) {      // the other three wavefronts: helpers of the backtrack (backtrack_dir.h)
            if constexpr (GAP != 0) {
            const int bits = b.aln[s].bits, w = b.aln[s].w, flags = b.aln[s].flags, hr = (((int)(tid >> 6) - worker) & (GW - 1));
            if ((flags & ALN_FAST_OK) && !(b.lds.wide_nw >= 1 && w >= b.lds.wide_w_lo && w <= b.lds.wide_w_hi)) {
                if (bits == 16) rounds_tail<int16_t, GAP, true>(slot, s, hr, sh_walk, k); else rounds_tail<int32_t, GAP, true>(slot, s, hr, sh_walk, k);
            }
            }
        }
        __syncthreads();                                    // graph cigar and result record are complete
        const long long c3 = (long long)__builtin_amdgcn_s_memtime();
        rounds_fuse(slot, s, k);
        __syncthreads();                                    // graph, row order and state of the next round are complete
        t_prep += c1 - c0; t_rows += c2 - c1; t_tail += c3 - c2; t_fuse += (long long)__builtin_amdgcn_s_memtime() - c3;
    }
    if (tid == worker * 64) { st->t_phase[0] += t_prep; st->t_phase[1] += t_rows; st->t_phase[2] += t_tail; st->t_phase[3] += t_fuse; }
}

// dynamic LDS: the largest phase (row loop, backtrack window, the prepare body's per-row records)
size_t poa_rounds_args_bytes() { return sizeof(RoundsArgs); }
hipError_t launch_poa_rounds(const PoaDev &p, const DevBatch &b, int32_t *cu_ticket, void *host_args, int slot, int k_lo, size_t lds_bytes, hipStream_t s) {
    if (p.n_sets <= 0) return hipSuccess;
    if (slot < 0 || slot >= MSA_DEVICE_SLOTS) return hipErrorInvalidValue;
    RoundsArgs &a = *(RoundsArgs *)host_args;      // (the caller's pinned staging memory: it outlives the asynchronous copy)
    a.p = p; a.b = b; a.cu_ticket = cu_ticket;
    hipError_t e = hipMemsetAsync(cu_ticket, 0, 4 * POA_CU_TICKETS, s);
    if (e != hipSuccess) return e;
    e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_rounds), &a, sizeof(a), sizeof(RoundsArgs) * (size_t)slot, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    if (lds_bytes > 65536) {      // (above 64 KB the kernel's dynamic-LDS limit has to be raised; without it the launch gets 64 KB and the phases read and write past it)
        e = hipFuncSetAttribute(b.gap_mode == ABPOA_HIP_LINEAR_GAP ? (const void *)poa_rounds_kernel<0> : (b.gap_mode == ABPOA_HIP_AFFINE_GAP ? (const void *)poa_rounds_kernel<1> : (const void *)poa_rounds_kernel<2>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (b.gap_mode == ABPOA_HIP_LINEAR_GAP) hipLaunchKernelGGL(poa_rounds_kernel<0>, dim3(p.n_sets), dim3(GT), lds_bytes, s, slot, k_lo);
    else if (b.gap_mode == ABPOA_HIP_AFFINE_GAP) hipLaunchKernelGGL(poa_rounds_kernel<1>, dim3(p.n_sets), dim3(GT), lds_bytes, s, slot, k_lo);
    else hipLaunchKernelGGL(poa_rounds_kernel<2>, dim3(p.n_sets), dim3(GT), lds_bytes, s, slot, k_lo);
    return hipGetLastError();
}
// workgroups of the all-rounds kernel one CU holds with `lds_bytes` of dynamic LDS (registers and LDS), and its static LDS
int poa_rounds_residency(int gap_mode, size_t lds_bytes, int *static_lds) {
    int nb = 0; hipFuncAttributes fa; memset(&fa, 0, sizeof(fa));
    if (gap_mode == ABPOA_HIP_LINEAR_GAP) { (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, poa_rounds_kernel<0>, GT, lds_bytes);
            (void)hipFuncGetAttributes(&fa, (const void *)poa_rounds_kernel<0>); }
    else if (gap_mode == ABPOA_HIP_AFFINE_GAP) { (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, poa_rounds_kernel<1>, GT, lds_bytes);
            (void)hipFuncGetAttributes(&fa, (const void *)poa_rounds_kernel<1>); }
    else { (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, poa_rounds_kernel<2>, GT, lds_bytes); (void)hipFuncGetAttributes(&fa, (const void *)poa_rounds_kernel<2>); }
    if (static_lds) *static_lds = (int)fa.sharedSizeBytes;
    return nb;
}

}  // namespace abpoa_hip
