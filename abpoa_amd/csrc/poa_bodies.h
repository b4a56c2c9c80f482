// Bodies of the per-round graph kernels of the device-resident driver (poa_device.h): graph -> DP rows before the alignment of read k
// (poa_prepare_body) and graph cigar -> graph after it (poa_fuse_body).  One workgroup of GT threads owns one read-set.  Two users:
// poa_device.hip wraps each body in a kernel of its own (lock-step rounds: one launch per phase and round for all sets), poa_rounds.hip
// calls them from the kernel that takes a read-set through ALL its rounds.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <string.h>
#include "poa_device.h"
#include "../../include/abpoa_hip.h"

namespace abpoa_hip {

namespace {      // (internal linkage: every translation unit that includes this header has its own copy)

__device__ __forceinline__ int imin_(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax_(int a, int b) { return a > b ? a : b; }
// loads of data that OTHER lanes of this wave stored earlier in the same kernel: agent scope = not served from this CU's L1
__device__ __forceinline__ int ld_fresh(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// slot t of a node's edge lists (X = node0 + node id): the first POA_HOT slots are in the hot arrays, the next ones in the cold arrays; slots from the
// capacity on exist for the source's out-edges and the sink's in-edges only (PoaSet.term0: the set's slice of the terminal pools)
constexpr int POA_TERM_MAX = 250;      // edges of the source / into the sink at most (the counts are bytes)
__device__ __forceinline__ int32_t &in_slot(const PoaDev &p, const PoaSet &S, int64_t X, int t) {
    return t < POA_HOT ? p.nd_in[X * POA_HOT + t] : (t < p.in_cap ? p.nd_inx[X * (p.in_cap - POA_HOT) + t - POA_HOT] : p.t_in[S.term0 + t - p.in_cap]);
}
__device__ __forceinline__ int32_t &out_slot(const PoaDev &p, const PoaSet &S, int64_t X, int t) {
    return t < POA_HOT ? p.nd_out[X * POA_HOT + t] : (t < p.out_cap ? p.nd_outx[X * (p.out_cap - POA_HOT) + t - POA_HOT] : p.t_out[S.term0 + t - p.out_cap]);
}
__device__ __forceinline__ int32_t &outw_slot(const PoaDev &p, const PoaSet &S, int64_t X, int t) {
    return t < POA_HOT ? p.nd_outw[X * POA_HOT + t] : (t < p.out_cap ? p.nd_outwx[X * (p.out_cap - POA_HOT) + t - POA_HOT] : p.t_outw[S.term0 + t - p.out_cap]);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }      // a value that is the same in every lane -> scalar register
__device__ __forceinline__ int shfl(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }

template <int CTRL> __device__ __forceinline__ int dpp_(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, 0xF, false); }
// inclusive prefix sum / max over the 64 lanes
__device__ __forceinline__ int wave_scan_add(int x) {
    x += dpp_<0x111>(0, x); x += dpp_<0x112>(0, x); x += dpp_<0x114>(0, x); x += dpp_<0x118>(0, x);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
    return x;
}
__device__ __forceinline__ int wave_scan_max(int x) {
    x = imax_(x, dpp_<0x111>(x, x)); x = imax_(x, dpp_<0x112>(x, x)); x = imax_(x, dpp_<0x114>(x, x)); x = imax_(x, dpp_<0x118>(x, x));
    x = imax_(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xA, 0xF, false));
    x = imax_(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xC, 0xF, false));
    return x;
}
// value of lane-1, lane 0 receives `lane0`

// reference src/simd_abpoa_align.c:1672-1683 (same arithmetic as abpoa_hip_score_bits in engine.cpp)
__device__ __forceinline__ int score_bits(const PoaDev &p, int n_rows, int qlen, int *inf_min) {
    const int oe1 = p.o1 + p.e1, oe2 = p.o2 + p.e2;
    const int len = qlen > n_rows ? qlen : n_rows;
    const int max_score = imax_(qlen * p.max_mat, len * p.e1 + p.o1);
    int bits, lo;
    if (max_score <= INT16_MAX - p.min_mis - oe1 - oe2) { bits = 16; lo = INT16_MIN; } else { bits = 32; lo = INT32_MIN; }
    *inf_min = imax_(imax_(lo + p.min_mis, lo + oe1), lo + oe2) + 31 * imax_(p.e1, p.e2);
    return bits;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// before the DP of round k: remaining length, rows in order with their predecessor CSR, alignment descriptor
// Four wavefronts per read-set: the row-parallel parts (heaviest edge, CSR) are latency-bound gathers, and four waves per SIMD hide
// most of that latency; the reverse sweep of the remaining length is sequential over 64-row blocks and runs on wavefront 0.
// workgroup size of the row-parallel graph kernels: 4 wavefronts per read-set (8 measured slower: the serial parts on wavefront 0 and the barriers dominate)
constexpr int GT = 256, GW = GT / 64;

__device__ __forceinline__ void poa_prepare_body(const PoaDev &p, const int s, const int k) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const PoaSet S = p.sets[s];
    PoaState *st = p.state + s;
    const int status = uni(st->status), n = uni(st->n_nodes);      // (wave-uniform; after a vector load when the state was written in this very kernel)
    // Early exit for sets that will outgrow their node slots: new nodes per read stay close to constant over the first reads
    // (most errors are novel), so ten reads predict the final size well; failing at once saves the rest of a doomed pass.
    // Read 5: a clear miss only (linear extrapolation 25 % over).  Read 10: a saturating growth model -- the nodes a read adds fall off as a / (1 + b k),
    // because its errors are more and more often ones the graph already holds (15 %-error 10 kb reads: 981 new nodes at read 5, 828 at read 10, 272 at
    // read 49; 5 %: 338, 331, 230) -- fitted to the mean gain of reads 3-6 and 7-10, summed over the remaining reads, +10 %.  (The linear test of round 2
    // projected 50 k nodes for graphs that end at 38.7 k and so sent every 15 % job to 6x node slots, twice the arena memory it needs.)
    bool doomed = false;
    // (not for read-sets with ragged ends, band_extra > 0: their first reads add nodes in bursts -- ends that reach beyond the graph -- and the projection sent a
    //  quarter of them through two more passes although they fit the first)
    if (status == POA_ST_OK && S.n_reads > 20 && !p.last_pass && S.band_extra == 0) {
        if (k == 2 && tid == 0) st->grow_n2 = n;
        if (k == 6 && tid == 0) st->grow_n6 = n;
        if (k == 5) {
            const int n0 = p.read_len[S.read0] + 2;
            const long long projected = (long long)n + (long long)(n - n0) * (S.n_reads - k) * 8 / (10 * k);
            doomed = projected * 4 > (long long)S.node_cap * 5;
        } else if (k == 10) {
            const int n2 = uni(st->grow_n2), n6 = uni(st->grow_n6);
            const float A = (float)(n6 - n2) * 0.25f, B = (float)(n - n6) * 0.25f;      // mean gain per read around read 4.5 / 8.5
            const float den = 8.5f * B - 4.5f * A;
            float bq = (A > B && den > 0.f) ? (A - B) / den : 0.f; bq = bq > 0.2f ? 0.2f : bq;
            const float a = A * (1.f + 4.5f * bq), N = (float)S.n_reads;
            const float rem = bq > 1e-4f ? (a / bq) * __logf((1.f + (N + 0.5f) * bq) / (1.f + 10.5f * bq)) : (A > B ? A : B) * (N - 10.f);
            doomed = (float)n + 1.1f * rem > (float)S.node_cap;
        }
        if (doomed && tid == 0) { st->status = POA_ST_FALLBACK; st->pad = 6; }
    }
    if (status != POA_ST_OK || doomed || k >= S.n_reads) {          // nothing to align for this set in this round: both DP kernels skip it
        if (tid == 0) { AlnDesc d; memset(&d, 0, sizeof(d)); d.n_rows = 3; d.bits = 16; d.flags = ALN_SKIP; p.aln[s] = d; p.out[s].status = 0; p.out[s].n_cigar = 0; p.out[s].n_cells = 0; }
        return;
    }
    const int64_t N0 = S.node0;
    const int32_t *order = p.row_node[uni(st->order_buf)] + N0;
    int32_t *nxt = p.scratch + S.scratch0;                 // [n] row of the heaviest successor
    int32_t *remain = p.row_remain + N0;
    // (1) heaviest out-edge per row (first maximum wins, reference :262-268)
    extern __shared__ unsigned jump_lds[];                 // [n] remaining-length jump records (see (2)), when the launch provides them
    const bool in_lds = p.pad > 0 && n <= p.pad;
    uint8_t *np_lds = (uint8_t *)(jump_lds + p.pad);       // [n] in-degree per row (row 0: none), for (3)
    {
        // four rows per thread and pass, every load level issued for all four before the next one: the chain order -> node -> edge
        // slots -> row of the successor is four dependent HBM/L2 round trips, and a thread owns ~10 rows (150 on a 10 kb graph)
        for (int r0 = tid; r0 < n; r0 += 4 * GT) {
            int u[4], no[4], ni[4], bs[4], best[4], nx[4]; int4 w4[4], o4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int r = r0 + j * GT; u[j] = order[r < n ? r : 0]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t X = N0 + u[j];
                no[j] = p.nd_nout[X]; ni[j] = p.nd_nin[X]; bs[j] = p.nd_base[X];
                w4[j] = *(const int4 *)(p.nd_outw + X * POA_HOT); o4[j] = *(const int4 *)(p.nd_out + X * POA_HOT);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int bw = -1, bb = -1;
                if (no[j] > 0 && w4[j].x > bw) { bw = w4[j].x; bb = o4[j].x; }
                if (no[j] > 1 && w4[j].y > bw) { bw = w4[j].y; bb = o4[j].y; }
                if (no[j] > 2 && w4[j].z > bw) { bw = w4[j].z; bb = o4[j].z; }
                if (no[j] > 3 && w4[j].w > bw) { bw = w4[j].w; bb = o4[j].w; }
                for (int t = POA_HOT; t < no[j]; ++t) { const int w = outw_slot(p, S, N0 + u[j], t); if (w > bw) { bw = w; bb = out_slot(p, S, N0 + u[j], t); } }
                best[j] = bb;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) nx[j] = best[j] >= 0 ? p.nd_row[N0 + best[j]] : -1;
            // how far down the row order the row's scores are still needed (DevBatch.row_sdist): the row of every out-edge (one more load per edge, same level as nx)
            int sd[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = r0 + j * GT; int far = 0;
                if (no[j] == 1) far = nx[j];
                else {
                    int rr[4] = {0, 0, 0, 0};
                    if (no[j] > 0) rr[0] = p.nd_row[N0 + o4[j].x]; if (no[j] > 1) rr[1] = p.nd_row[N0 + o4[j].y]; if (no[j] > 2) rr[2] = p.nd_row[N0 + o4[j].z];
                    if (no[j] > 3) rr[3] = p.nd_row[N0 + o4[j].w];
                    far = imax_(imax_(rr[0], rr[1]), imax_(rr[2], rr[3]));
                    for (int t = POA_HOT; t < no[j]; ++t) far = imax_(far, p.nd_row[N0 + out_slot(p, S, N0 + u[j], t)]);
                }
                sd[j] = far >= n - 1 ? 255 : imin_(imax_(far - r, 0), 255);      // (the sink is the last row)
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = r0 + j * GT;
                if (r < n) {
                    p.row_sdist[N0 + r] = (uint8_t)sd[j];
                    if (in_lds) { jump_lds[r] = nx[j] >= 0 ? ((unsigned)nx[j] << 16) | 1u : ((unsigned)r << 16); np_lds[r] = (uint8_t)(r > 0 ? ni[j] : 0); }
                    else nxt[r] = nx[j];
                    p.row_base[N0 + r] = (uint8_t)bs[j]; p.row_node_id[N0 + r] = u[j];
                }
            }
        }
    }
    __syncthreads();
    // (2) remaining length = (edges to the sink along heaviest successors) - 1, reference :233-274 (only the adaptive band reads it: abpoa_graph.c:303-311)
    if (!p.banded) { for (int r = tid; r < n; r += GT) remain[r] = 0; }
    else if (in_lds) {
        // Pointer jumping over all rows at once in LDS.  A record "row of a later node on the chain << 16 | edges up to it" is one
        // 32-bit word, so a row may read a neighbour's record while that neighbour is being advanced: either version is a valid jump.
        for (int span = 1; span < n; span <<= 1) {
            for (int r = tid; r < n; r += GT) {
                const unsigned w = jump_lds[r]; const int t = (int)(w >> 16);
                if (t != r) { const unsigned wt = jump_lds[t]; jump_lds[r] = (wt & 0xffff0000u) | ((w & 0xffffu) + (wt & 0xffffu)); }
            }
            __syncthreads();
        }
        for (int r = tid; r < n; r += GT) remain[r] = (int)(jump_lds[r] & 0xffffu) - 1;
    } else
    // (graphs too large for the LDS records) reverse sweep over 64-row blocks; inside a block the chains are resolved by pointer jumping.  The sweep is a
    // chain over the blocks, so what a block costs is what counts (600 blocks for a 10 kb graph): the successor rows of the NEXT block are loaded while
    // this one is resolved, and the values of the block above stay in a register -- a heaviest successor is almost always within the next few rows -- so
    // that only a target further than 64 rows beyond the block goes to memory (and waits for this wave's stores).
    if (wave == 0) {
        int t0 = ((n - 1) >> 6) << 6;
        int tgt_next = t0 + lane < n ? ld_fresh(nxt + t0 + lane) : -1, above = 0;      // above: values of block t0 + 64 (lane = row - t0 - 64)
        for (; t0 >= 0; t0 -= 64) {
            const int r = t0 + lane;
            int tgt = tgt_next, dist = 1, val = 0; bool done = r >= n;
            if (t0 >= 64) tgt_next = ld_fresh(nxt + t0 - 64 + lane);
            if (!done && tgt < 0) { val = -1; done = true; }                         // the sink (reference :247)
            const bool far = !done && tgt >= t0 + 128;
            const int from_above = shfl(above, (tgt - t0 - 64) & 63);
            if (!done && tgt >= t0 + 64 && !far) { val = from_above + 1; done = true; }
            if (__any(far)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // stores are write-through: once acknowledged, later blocks read them from L2 (ld_fresh)
                if (far) { val = ld_fresh(remain + tgt) + 1; done = true; }
            }
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int src = (!done) ? tgt - t0 : lane;
                const int t_tgt = shfl(tgt, src), t_dist = shfl(dist, src), t_val = shfl(val, src), t_done = shfl((int)done, src);
                if (!done) {
                    if (t_done) { val = t_val + dist; done = true; }
                    else { tgt = t_tgt; dist += t_dist; }
                }
            }
            if (r < n) remain[r] = val;
            above = val;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // (3) predecessor CSR in row order (in_id order kept, reference pre_index[][] :519-530)
    __shared__ int wtot[GW];
    int carry = 0; bool overflow = false;
    for (int t0 = 0; t0 < n; t0 += GT) {
        const int r = t0 + tid;
        const int u = (r < n && !in_lds) ? order[r] : 0;
        const int np = r < n ? (in_lds ? (int)np_lds[r] : (r > 0 ? (int)p.nd_nin[N0 + u] : 0)) : 0;
        const int incl = wave_scan_add(np);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w_ = 0; w_ < GW; ++w_) { const int x_ = wtot[w_]; all += x_; before += w_ < wave ? x_ : 0; }
        const int off = carry + before + incl - np;
        if (r < n) p.pred_off[N0 + r] = off;
        if (off + np > S.pred_cap) overflow = true;
        carry += all;
        __syncthreads();
    }
    {      // the lists themselves, four rows per thread and pass (see (1)); pred_off is read back by the thread that wrote it
        for (int r0 = tid; r0 < n; r0 += 4 * GT) {
            int u[4], np[4], off[4]; int4 i4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int r = r0 + j * GT; const bool ok = r < n; u[j] = order[ok ? r : 0]; off[j] = p.pred_off[N0 + (ok ? r : 0)];
                                          np[j] = !ok ? 0 : (in_lds ? (int)np_lds[r] : (r > 0 ? (int)p.nd_nin[N0 + u[j]] : 0)); }
#pragma unroll
            for (int j = 0; j < 4; ++j) { if (off[j] + np[j] > S.pred_cap) np[j] = 0; i4[j] = *(const int4 *)(p.nd_in + (N0 + u[j]) * POA_HOT); }
            int pr[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pr[j][0] = np[j] > 0 ? p.nd_row[N0 + i4[j].x] : 0; pr[j][1] = np[j] > 1 ? p.nd_row[N0 + i4[j].y] : 0;
                pr[j][2] = np[j] > 2 ? p.nd_row[N0 + i4[j].z] : 0; pr[j][3] = np[j] > 3 ? p.nd_row[N0 + i4[j].w] : 0;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int32_t *dst = p.pred_row + S.pred0 + off[j];
                const int r = r0 + j * GT;      // (np[j] is 0 for a row whose list does not fit: the set falls back anyway)
                unsigned long long pdv = ~0ull;      // a byte per predecessor (the first eight), 255 = none / further than 254 rows
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t < np[j]) { dst[t] = pr[j][t]; if (r - pr[j][t] <= 254) pdv = (pdv & ~(0xffull << (8 * t))) | ((unsigned long long)(r - pr[j][t]) << (8 * t)); }
                }
                for (int t = POA_HOT; t < np[j]; ++t) { const int pr_ = p.nd_row[N0 + in_slot(p, S, N0 + u[j], t)]; dst[t] = pr_;
                        if (t < 8 && r - pr_ <= 254) pdv = (pdv & ~(0xffull << (8 * t))) | ((unsigned long long)(r - pr_) << (8 * t)); }
                if (r < n) { p.row_pd[2 * (N0 + r)] = (unsigned)pdv; p.row_pd[2 * (N0 + r) + 1] = (unsigned)(pdv >> 32); }
            }
        }
    }
    if (tid == 0) p.pred_off[N0 + n] = carry;
    if (p.out_off) {      // (3b) successor rows per row, out_id order: the general kernel hands a row's arg-max to its successors' band state (reference :1078-1091)
        int carry_o = 0;
        for (int t0 = 0; t0 < n; t0 += GT) {
            const int r = t0 + tid;
            const int u = r < n ? order[r] : 0;
            const int no = r < n ? (int)p.nd_nout[N0 + u] : 0;
            const int incl = wave_scan_add(no);
            __syncthreads();
            if (lane == 63) wtot[wave] = incl;
            __syncthreads();
            int before = 0, all = 0;
#pragma unroll
            for (int w_ = 0; w_ < GW; ++w_) { const int x_ = wtot[w_]; all += x_; before += w_ < wave ? x_ : 0; }
            const int off = carry_o + before + incl - no;
            if (r < n) {
                p.out_off[N0 + r] = off;
                if (off + no > S.pred_cap) overflow = true;
                else for (int t = 0; t < no; ++t) p.out_row[S.pred0 + off + t] = p.nd_row[N0 + out_slot(p, S, N0 + u, t)];
            }
            carry_o += all;
        }
        if (tid == 0) p.out_off[N0 + n] = carry_o;
    }
    overflow = __syncthreads_or(overflow);
    // (4) alignment descriptor of this round
    if (tid == 0) {
        AlnDesc d; memset(&d, 0, sizeof(d));
        const int qlen = p.read_len[S.read0 + k];
        d.n_rows = n; d.qlen = qlen;
        d.bits = score_bits(p, n, qlen, &d.inf_min);
        d.w = p.wb < 0 ? qlen : p.wb + (int)(p.wf * (float)qlen);            // reference :445 (float32 product)
        d.cigar_cap = S.cigar_cap; d.flags = (p.general & 1) ? 0 : ALN_FAST_OK; d.pad0 = S.band_extra;
        d.query_off = p.read_off[S.read0 + k]; d.row0 = N0; d.poff0 = N0; d.pred0 = S.pred0; d.out0 = S.pred0;
        d.plane_off = S.plane_off; d.plane_cap = S.plane_cap / (d.bits / 8); d.cigar_off = S.cigar_off;
        if (overflow || n + qlen + 8 > S.cigar_cap) { st->status = POA_ST_FALLBACK; st->pad = overflow ? 2 : 3; d.flags = ALN_SKIP; d.n_rows = 3; }
        p.aln[s] = d;
        p.out[s].status = 0; p.out[s].n_cigar = 0; p.out[s].n_cells = 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// after the backtrack of round k: fuse the graph cigar of read k into the graph and extend the row order
// Four wavefronts per read-set; the walk over the query (F2) takes GT positions per pass.
__device__ __forceinline__ void poa_fuse_body(const PoaDev &p, const int s, const int k) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int sh_fail, sh_nodes, wtot[GW];
    const PoaSet S = p.sets[s];
    PoaState *st = p.state + s;
    if (uni(st->status) != POA_ST_OK || k >= S.n_reads) return;
    AlnOut res = p.out[s]; res.status = uni(res.status); res.n_cigar = uni(res.n_cigar);
    if (res.status != 0) { if (tid == 0) { st->status = POA_ST_FALLBACK; st->pad = 1000 + res.status; } return; }
    const int64_t N0 = S.node0;
    const int n_old = uni(st->n_nodes), qlen = p.read_len[S.read0 + k], n_cigar = res.n_cigar;
    const bool rc_read = p.is_rc && uni((int)p.is_rc[S.read0 + k]) != 0;      // (-s: the reverse complement of this read won, reference :331-334)
    const uint8_t *seq = (rc_read ? p.reads_rc : p.reads) + p.read_off[S.read0 + k];
    const uint64_t *cg = p.cigar + S.cigar_off;
    const int cur = uni(st->order_buf);
    const int32_t *order_old = p.row_node[cur] + N0; int32_t *order_new = p.row_node[cur ^ 1] + N0;
    int32_t *cand = p.scratch + S.scratch0, *n_anchor = cand + p.max_qlen, *n_j = n_anchor + p.max_qlen, *addcnt = n_j + p.max_qlen;
    if (p.dig_on) {      // (tests) this alignment's cigar into the set's digest -- before the early return: an empty cigar is a cigar too
        __shared__ unsigned long long sh_dig;
        if (tid == 0) sh_dig = 0;
        __syncthreads();
        unsigned long long part = 0;
        for (int i = tid; i < n_cigar; i += GT) part += poa_cigar_word_mix(cg[i], i);
        if (part) atomicAdd(&sh_dig, part);
        __syncthreads();
        if (tid == 0) st->cigar_dig = poa_cigar_digest_round(st->cigar_dig, k, n_cigar, sh_dig);
    }
    if (n_cigar == 0) return;                                                   // reference :614-616
    // F0/F1: node every query base is aligned to (-1: inserted base)
    for (int q = tid; q < qlen; q += GT) cand[q] = -1;
    for (int r = tid; r < n_old; r += GT) addcnt[r] = 0;
    __syncthreads();
    for (int i = tid; i < n_cigar; i += GT) {
        const uint64_t w = cg[i];
        if ((int)(w & 0xf) == ABPOA_HIP_CMATCH) cand[(int)((w >> 4) & 0x3fffffff)] = (int)((w >> 34) & 0x3fffffff);
    }
    __syncthreads();
    // F2: walk the query in chunks of 64 positions
    int n_nodes = n_old, prev_c = 0 /* source */, prev_new_c = 0, carry_ar = 0 /* row of the source */, carry_sq = -1;
    bool fail = false, fail_slots = false;      // (fail_slots: an edge / aligned list is full -- more node slots would not help, PoaState.pad 5)
    // adds edge from -> to (both lane-private; `from_new` / `to_new`: the node was created by this read, its lists are still empty
    // apart from what this very walk put there, which is known without a load)
    // per-base weights of this read (-Q), reference :634-667: an edge takes the weight of the base it leads to
    const int32_t *wq = p.wts ? (rc_read ? p.wts_rc : p.wts) + p.read_off[S.read0 + k] : nullptr;
    auto add_edge = [&](bool act, int from, bool from_new, int to, bool to_new, int w) {
        if (!act) return;
        const int64_t F = N0 + from, T = N0 + to;
        int no = from_new ? 0 : (int)p.nd_nout[F];
        int hit = -1;
        if (!from_new && !to_new) for (int t = 0; t < no; ++t) if (out_slot(p, S, F, t) == to) { hit = t; break; }
        if (hit >= 0) outw_slot(p, S, F, hit) += w;
        else {
            const int ni = to_new ? 0 : (int)p.nd_nin[T];
            // (the source may have any number of out-edges, the sink of in-edges: reads that start / end on different nodes; PoaSet.term0)
            if (no >= (from == 0 ? imin_(POA_TERM_MAX, p.out_cap + S.n_reads + 2) : p.out_cap) || ni >= (to == 1 ? imin_(POA_TERM_MAX, p.in_cap + S.n_reads + 2) : p.in_cap)) { fail = true;
                    fail_slots = true; return; }
            out_slot(p, S, F, no) = to; outw_slot(p, S, F, no) = w; p.nd_nout[F] = (uint8_t)(no + 1);
            in_slot(p, S, T, ni) = from; p.nd_nin[T] = (uint8_t)(ni + 1);
            hit = no;
            // (a new edge: no read went through it yet.  The source's edges beyond the capacity keep no read ids: nothing reads the source's -- the MSA rows
            //  come from the out-edges of nodes 2.., reference abpoa_output.c:142-150)
            if (hit < p.out_cap) for (int w_ = 0; w_ < p.rid_words; ++w_) p.nd_rid[(F * p.out_cap + hit) * p.rid_words + w_] = 0;
        }
        // read k went through this edge (reference :453-472; a node is the tail of at most one edge per read, so no other lane touches these words)
        if (p.rid_words && hit < p.out_cap) p.nd_rid[(F * p.out_cap + hit) * p.rid_words + (k >> 6)] |= 1ull << (k & 63);
        p.nd_nread[F] = (from_new ? 0 : p.nd_nread[F]) + 1;
    };
    // all wavefronts walk the query together, GT positions per pass: per-position work (node lookup, new node, edge) is private to
    // its thread; what flows along the path (new-node ids, previous node, the running maxima AR / sq) crosses wavefronts through LDS
    __shared__ int sCnt[GW], sMax[GW], sSqm[GW];
    extern __shared__ int fuse_dyn[];                       // dynamic LDS of the launch (4 * GT ints; in the all-rounds kernel the other phases' space)
    int *const sNode = fuse_dyn, *const sIsNew = fuse_dyn + GT, *const sAR = fuse_dyn + 2 * GT, *const sSq = fuse_dyn + 3 * GT;
    for (int q0 = 0; q0 < qlen; q0 += GT) {
        const int q = q0 + tid; const bool act = q < qlen;
        const int c = act ? ld_fresh(cand + q) : -1;
        const int b = act ? (int)seq[q] : 0;
        int node = -1; bool isnew = act;
        if (act && c >= 0) {
            if ((int)p.nd_base[N0 + c] == b) { node = c; isnew = false; }
            else {                                                              // reference abpoa_get_aligned_id :377-386
                const int na = p.nd_naln[N0 + c];
                for (int t = 0; t < na; ++t) { const int a = p.nd_aln[(N0 + c) * p.aln_cap + t]; if ((int)p.nd_base[N0 + a] == b) { node = a; isnew = false; break; } }
            }
        }
        // group-end row of the aligned group that places this position in the row order (see the bookkeeping below); read before
        // the group is extended
        int gv = -1;
        if (act && c >= 0) {
            const int ref = isnew ? c : node;
            gv = p.nd_row[N0 + ref];
            const int na = p.nd_naln[N0 + ref];
            for (int t = 0; t < na; ++t) gv = imax_(gv, p.nd_row[N0 + p.nd_aln[(N0 + ref) * p.aln_cap + t]]);
        }
        // exchange 1: ids of the new nodes (path order) and the running maximum AR
        const unsigned long long newmask = __ballot(isnew);
        const int rank = __builtin_popcountll(newmask & ((1ull << lane) - 1));
        const int gmax_w = wave_scan_max(gv);
        if (lane == 63) { sCnt[wave] = __builtin_popcountll(newmask); sMax[wave] = gmax_w; }
        __syncthreads();
        int before_cnt = 0, n_new = 0, before_max = carry_ar;
#pragma unroll
        for (int w_ = 0; w_ < GW; ++w_) { const int x_ = sCnt[w_]; n_new += x_; if (w_ < wave) { before_cnt += x_; before_max = imax_(before_max, sMax[w_]); } }
        if (n_nodes + n_new > S.node_cap) { fail = true; break; }                // (the same in every thread)
        if (isnew) node = n_nodes + before_cnt + rank;
        if (isnew) {
            const int64_t Y = N0 + node;
            p.nd_base[Y] = (uint8_t)b; p.nd_nin[Y] = 0; p.nd_nout[Y] = 0; p.nd_naln[Y] = 0; p.nd_nread[Y] = 0;
            if (c >= 0) {                                                       // mismatch: new node joins c's aligned group, reference :393-401
                const int na = p.nd_naln[N0 + c];
                if (na + 1 > p.aln_cap) { fail = true; fail_slots = true; }
                else {
                    for (int t = 0; t < na; ++t) {
                        const int other = p.nd_aln[(N0 + c) * p.aln_cap + t];
                        const int no_ = p.nd_naln[N0 + other];                   // == na for every member of the group
                        p.nd_aln[(N0 + other) * p.aln_cap + no_] = node; p.nd_naln[N0 + other] = (uint8_t)(no_ + 1);
                        p.nd_aln[Y * p.aln_cap + t] = other;
                    }
                    p.nd_aln[(N0 + c) * p.aln_cap + na] = node; p.nd_naln[N0 + c] = (uint8_t)(na + 1);
                    p.nd_aln[Y * p.aln_cap + na] = c; p.nd_naln[Y] = (uint8_t)(na + 1);
                }
            }
        }
        // row-order bookkeeping.  Invariant (the reference's Kahn walk keeps it too, abpoa_graph.c:213-224): the members of an
        // aligned group are contiguous in the row order.  A new node is therefore spliced in right after the END of a group:
        // the group of the node it mismatches (which it joins), or the group of the nearest old node before it on the path,
        // whichever comes later -- i.e. after row AR = running maximum along the path of "group-end row" (old nodes: their own
        // group; mismatch nodes: the group they join; inserted bases: none).  New nodes with the same AR form a run in path order.
        const int AR = imax_(gmax_w, before_max);
        // exchange 2: the previous position on the path (thread - 1; thread 0: last position of the previous pass).  The barrier also
        // orders the new nodes' initialisation above before the edge updates below.
        sNode[tid] = node; sIsNew[tid] = (int)isnew; sAR[tid] = AR;
        __syncthreads();
        const int prev = tid > 0 ? sNode[tid - 1] : prev_c, prev_new = tid > 0 ? sIsNew[tid - 1] : prev_new_c, prevAR = tid > 0 ? sAR[tid - 1] : carry_ar;
        add_edge(act, prev, prev_new != 0, node, isnew, (wq && act) ? wq[q] : 1);
        const bool start = isnew && (prev_new == 0 || prevAR != AR);
        const int sq_w = wave_scan_max(start ? q : -1);
        if (lane == 63) sSqm[wave] = sq_w;
        __syncthreads();
        int sq = imax_(sq_w, carry_sq);
#pragma unroll
        for (int w_ = 0; w_ < GW; ++w_) if (w_ < wave) sq = imax_(sq, sSqm[w_]);
        if (isnew) { n_anchor[node - n_old] = AR; n_j[node - n_old] = q - sq + 1; atomicAdd(addcnt + AR, 1); }
        sSq[tid] = sq;
        __syncthreads();
        // carries into the next pass (from the last active position)
        const int lt = imin_(GT - 1, qlen - 1 - q0);
        prev_c = sNode[lt]; prev_new_c = sIsNew[lt]; carry_ar = sAR[lt]; carry_sq = sSq[lt];
        n_nodes += n_new;
    }
    // F3: last node -> sink (reference :667)
    bool any_fail = __syncthreads_or(fail);
    if (!any_fail) { add_edge(tid == 0, prev_c, prev_new_c != 0, 1, false, wq ? wq[qlen - 1] : 1); any_fail = __syncthreads_or(fail); }      // reference :667: the last base's weight
    const bool any_slots = __syncthreads_or(fail_slots);
    if (tid == 0) { sh_fail = any_fail ? (any_slots ? 5 : 4) : 0; sh_nodes = n_nodes; }
    __syncthreads();
    if (sh_fail) { if (tid == 0) { st->status = POA_ST_FALLBACK; st->pad = sh_fail; } return; }
    n_nodes = sh_nodes;
    // F4: new row order = old order with every run of new nodes spliced in after its anchor
    int carry = 0;
    for (int t0 = 0; t0 < n_old; t0 += GT) {
        const int r = t0 + tid;
        const int cnt = r < n_old ? ld_fresh(addcnt + r) : 0;
        const int incl = wave_scan_add(cnt);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w_ = 0; w_ < GW; ++w_) { const int x_ = wtot[w_]; all += x_; before += w_ < wave ? x_ : 0; }
        const int shift = carry + before + incl - cnt;      // new nodes anchored at earlier rows
        if (r < n_old) { const int u = order_old[r]; order_new[r + shift] = u; p.nd_row[N0 + u] = r + shift; addcnt[r] = shift; }
        carry += all;
        __syncthreads();
    }
    __syncthreads();
    for (int i = tid; i < n_nodes - n_old; i += GT) {
        const int ar = ld_fresh(n_anchor + i), j = ld_fresh(n_j + i);
        const int nr = ar + ld_fresh(addcnt + ar) + j;
        order_new[nr] = n_old + i; p.nd_row[N0 + n_old + i] = nr;
    }
    if (tid == 0) {
        st->n_nodes = n_nodes; st->order_buf = cur ^ 1; st->n_cells += res.n_cells;
        st->algo_bytes += res.n_cells * (p.aln[s].bits / 8) * (p.gap_mode == ABPOA_HIP_AFFINE_GAP ? 5 : 8);
    }
}

}  // namespace abpoa_hip
