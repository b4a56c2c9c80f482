// Host side of the device-resident read-set driver (poa_device.h): pool allocation, one upload of the reads, the per-round
// kernel sequence (prepare -> DP rows -> backtrack -> fuse) queued back to back with NO host work or synchronisation in
// between, consensus / MSA kernels, one download of the results.  Sets that exceed a device capacity (nodes, edge slots of an
// inner node, arena) are reported back: redone in a pass with more node slots or by the host driver (msa_batch.cpp).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <limits.h>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include <hip/hip_runtime.h>
#include "engine_options.h"
#include "batch_stream.h"
#include "msa_device.h"
#include "poa_device.h"
#include "poa_graph.h"
#include "dir_plane.h"
#include "msa_batch.h"
#include "msa_device_debug.h"

namespace abpoa_hip {

namespace {
#define HIP_OK(expr, code) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
        set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return code; } } while (0)

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// [0, n) split into contiguous ranges, one per worker (the calling thread takes the last range)
template <typename F>
void parallel_ranges(int n_threads, int n, F fn) {
    const int T = std::max(1, std::min(n_threads, n / 64));
    if (T <= 1) { fn(0, n); return; }
    std::vector<std::thread> th; th.reserve(T - 1);
    for (int t = 0; t < T - 1; ++t) th.emplace_back(fn, (int)((int64_t)n * t / T), (int)((int64_t)n * (t + 1) / T));
    fn((int)((int64_t)n * (T - 1) / T), n);
    for (auto &x : th) x.join();
}
size_t up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ABPOA_HIP_DEVICE_AUDIT=1 (multi-GPU runs: SURVEY.md section 8e): every pool allocation of a device queue is looked up with hipPointerGetAttributes and must
//  sit
// on the device the queue serves -- and the calling thread's current device must be that one -- or the call fails with ABPOA_HIP_ENODEV; ABPOA_HIP_VERBOSE=1
// prints where each allocation landed.  (The HIP device is per host thread; a queue thread that forgot hipSetDevice would put its pools on device 0.)
thread_local int t_queue_device = -1;
int audit_allocation(const void *ptr, size_t bytes, const char *what) {
    static const bool on = opt_env("ABPOA_HIP_DEVICE_AUDIT") && atoi(opt_env("ABPOA_HIP_DEVICE_AUDIT"));
    if (!on || t_queue_device < 0) return 0;
    int cur = -1; (void)hipGetDevice(&cur);
    hipPointerAttribute_t at; memset(&at, 0, sizeof(at));
    const hipError_t e = hipPointerGetAttributes(&at, ptr);
    if (opt_env("ABPOA_HIP_VERBOSE")) fprintf(stderr, "[abpoa-hip] audit: %s, %zu bytes: device %d (queue device %d, thread's current device %d)\n", what,
            bytes, e == hipSuccess ? at.device : -1, t_queue_device, cur);
    if (e != hipSuccess || cur != t_queue_device || (at.type == hipMemoryTypeDevice && at.device != t_queue_device)) {
        set_err("device audit: %s of %zu bytes on device %d, queue serves device %d, thread's current device %d", what, bytes,
                e == hipSuccess ? at.device : -1, t_queue_device, cur);
        return ABPOA_HIP_ENODEV;
    }
    return 0;
}

struct Arena {                 // grow-only device / pinned-host buffers kept across calls (one job at a time, see g_mu)
    uint8_t *dev = nullptr, *host = nullptr; size_t dev_cap = 0, host_cap = 0;
    int need_dev(size_t n) {
        if (n <= dev_cap) return 0;
        if (dev) (void)hipFree(dev);
        dev = nullptr; dev_cap = 0;
        HIP_OK(hipMalloc((void **)&dev, n), ABPOA_HIP_ENOMEM); dev_cap = n; return audit_allocation(dev, n, "device pool");
    }
    int need_host(size_t n) {
        if (n <= host_cap) return 0;
        if (host) (void)hipHostFree(host);
        host = nullptr; host_cap = 0;
        HIP_OK(hipHostMalloc((void **)&host, n, hipHostMallocDefault), ABPOA_HIP_ENOMEM); host_cap = n;
        return audit_allocation(host, n, "pinned staging buffer");
    }
};
struct Cache { Arena in, graph, rows, planes, out, msa; hipStream_t stream = nullptr, copy_stream = nullptr; hipEvent_t ev_copy = nullptr;
        std::vector<hipEvent_t> ev; int device = -1; };
Cache g_cache[MSA_DEVICE_SLOTS]; std::mutex g_cache_mu[MSA_DEVICE_SLOTS];      // one per worker of the multi-device batch call

struct Layout {                // byte offsets inside the three device blobs
    // in blob (uploaded): sets, read tables, reads, score matrix
    // (o_rc, o_wrc: device only, behind the uploaded part)
    size_t o_sets, o_roff, o_rlen, o_reads, o_mat, o_rargs, o_msaoff_h, o_wts, in_bytes, o_rc, o_wrc, in_dev_bytes;
    // graph blob (device only, the tail of it downloaded at the end): per-node pools
    size_t o_cnode, o_ccov, o_cbase, o_isrc;
    size_t o_state, o_base, o_nin, o_nout, o_naln, o_in, o_out, o_outw, o_inx, o_outx, o_outwx, o_aln, o_nread, o_row, o_order0, o_order1, o_rid, o_mrank,
            o_msaoff, o_tout, o_toutw, o_tin, graph_bytes;
    // rows blob: DP inputs / outputs per row, descriptors, cigars, scratch
    size_t o_ticket, o_aln_desc, o_out_rec, o_rbase, o_rsd, o_rpd, o_rnid, o_rrem, o_poff, o_pred, o_bsn, o_esn, o_coff, o_rmi, o_cigar, o_scratch, o_ooff,
            o_orow, o_left, o_right, o_act, o_outfwd, o_cigfwd, o_retry, rows_bytes;
};

}  // namespace

void release_msa_device_caches() {
    int cur = -1; (void)hipGetDevice(&cur);
    for (int s = 0; s < MSA_DEVICE_SLOTS; ++s) {
        std::lock_guard<std::mutex> lk(g_cache_mu[s]);
        Cache &C = g_cache[s];
        if (C.device < 0) continue;
        (void)hipSetDevice(C.device);
        if (C.stream) (void)hipStreamSynchronize(C.stream);
        for (Arena *a : {&C.in, &C.graph, &C.rows, &C.planes, &C.out, &C.msa}) { if (a->dev) (void)hipFree(a->dev); if (a->host) (void)hipHostFree(a->host);
                *a = Arena(); }
    }
    if (cur >= 0) (void)hipSetDevice(cur);
}

// What the device-resident driver takes: every gap model and alignment mode (run_msa_device picks the fast row loops or the general kernel per job), any
// alphabet of up to 27 codes, consensus and / or MSA output, per-base weights, the strand retry.
bool msa_device_eligible(const abpoa_hip_scoring_t *sc, unsigned flags) {
    const char *e = opt_env("ABPOA_HIP_HOSTGRAPH");
    if (e && atoi(e)) return false;
    if (sc->m - 1 > POA_ALN_MAX || sc->m < 2) return false;
    // (-s on the host driver, as before round 4)
    if ((flags & ABPOA_HIP_AMB_STRAND) && opt_env("ABPOA_HIP_NO_DEVICE_STRAND") && atoi(opt_env("ABPOA_HIP_NO_DEVICE_STRAND"))) return false;
    if (sc->align_mode == ABPOA_HIP_LOCAL_MODE && opt_env("ABPOA_HIP_NO_DEVICE_LOCAL") && atoi(opt_env("ABPOA_HIP_NO_DEVICE_LOCAL"))) return false;
    // every gap model and alignment mode: the fast row loops where they apply (banded global, short local), the general kernel otherwise (linear gaps,
    // extension mode with or without z-drop, global mode without a band, long local reads).  ABPOA_HIP_NO_DEVICE_GENERAL=1 sends those back to the host driver.
    const bool fast = fast_global_job(sc->gap_mode, sc->align_mode, sc->wb, sc->gap_ext1) || (sc->gap_mode != ABPOA_HIP_LINEAR_GAP && sc->align_mode == ABPOA_HIP_LOCAL_MODE);
    if (!fast && opt_env("ABPOA_HIP_NO_DEVICE_GENERAL") && atoi(opt_env("ABPOA_HIP_NO_DEVICE_GENERAL"))) return false;
    return true;
}

int msa_device_resident_sets(const abpoa_hip_scoring_t *sc, int n_sets, const abpoa_hip_readset_t *sets) {
    int max_qlen = 0; for (int s = 0; s < n_sets; ++s) for (int r = 0; r < sets[s].n_reads; ++r) max_qlen = std::max(max_qlen, sets[s].lens[r]);
    if (max_qlen <= 0) return 0;
    const int w_max = sc->wb + (int)(sc->wf * (float)max_qlen);
    LdsPlan pl; int32_t inf_d; const int mb = abpoa_hip_score_bits(sc, 3 * max_qlen + 1024, max_qlen, &inf_d); const int pn_ = mb == 16 ? 16 : 8;
    make_lds_plan(sc, max_qlen, mb, std::min<int64_t>((int64_t)((max_qlen + pn_) / pn_) * pn_, 2LL * w_max + 3 * pn_ + 32), n_sets, &pl);
    if (pl.wide_nw != 1 || !(w_max >= pl.wide_w_lo && w_max <= pl.wide_w_hi) || pl.total_wide <= 0) return 0;
    int dev = 0, cus = 256; if (hipGetDevice(&dev) == hipSuccess) { hipDeviceProp_t pr;
            if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
    // (LDS is handed out in pieces of 1280 B, 128 per CU: tools/probes/lds_granule.hip; 166-192 VGPRs: two wavefronts per SIMD at most)
#ifdef ABPOA_HIP_WIDE_W3
    const int per_cu = std::max(1, std::min(12, 128 / ((pl.total_wide + 1279) / 1280)));
#else
    const int per_cu = std::max(1, std::min(8, 128 / ((pl.total_wide + 1279) / 1280)));
#endif
    return per_cu * cus;
}

// force_general: every alignment through the general kernel; *want_general: the final LDS plan has no fast row loop for this job although the first estimate had
// one (ragged sets: one node factor more, wider rows -- the score width can flip to 32 bits): the caller runs the job again with force_general
static int run_msa_device_body(const abpoa_hip_scoring_t *sc_in, int n_sets, const abpoa_hip_readset_t *sets, abpoa_hip_msa_t *out, int n_threads,
                               std::vector<int> *fallback, DeviceRunStats *stats, double node_factor, unsigned flags, int device, int slot, bool force_general, bool *want_general,
                               std::vector<int> *fallback_reason) {
    abpoa_hip_scoring_t sc_norm = *sc_in; const bool local = sc_in->align_mode == ABPOA_HIP_LOCAL_MODE, extend = sc_in->align_mode == ABPOA_HIP_EXTEND_MODE;
    if (local) sc_norm.wb = -1;                                  // reference abpoa_post_set_para, src/abpoa_align.c:150
    const abpoa_hip_scoring_t *sc = &sc_norm;
    // -s: low-scoring reads are aligned again as their reverse complement (poa_device.hip poa_strand_check_kernel)
    const bool amb = flags & ABPOA_HIP_AMB_STRAND;
    const bool want_msa = flags & ABPOA_HIP_OUT_MSA, want_cons = (flags & ABPOA_HIP_OUT_CONS) || !want_msa;
    if (slot < 0 || slot >= MSA_DEVICE_SLOTS) { set_err("bad device slot %d", slot); return ABPOA_HIP_EINVAL; }
    std::lock_guard<std::mutex> lk(g_cache_mu[slot]);
    Cache &C = g_cache[slot];
    if (device < 0) device = engine_device();
    if (device < 0) { set_err("engine not initialised"); return ABPOA_HIP_ENODEV; }
    HIP_OK(hipSetDevice(device), ABPOA_HIP_ENODEV);          // (the HIP device is per host thread)
    t_queue_device = device;
    if (C.device != device) {
        if (C.device >= 0) {      // the slot served another device before: its pools and stream live there
            (void)hipSetDevice(C.device);
            for (Arena *a : {&C.in, &C.graph, &C.rows, &C.planes, &C.out, &C.msa}) { if (a->dev) (void)hipFree(a->dev);
                    if (a->host) (void)hipHostFree(a->host); *a = Arena(); }
            if (C.stream) (void)hipStreamDestroy(C.stream); if (C.copy_stream) (void)hipStreamDestroy(C.copy_stream);
            if (C.ev_copy) (void)hipEventDestroy(C.ev_copy);
            for (hipEvent_t e : C.ev) (void)hipEventDestroy(e);
            C = Cache();
            HIP_OK(hipSetDevice(device), ABPOA_HIP_ENODEV);
        }
        if (!C.stream) HIP_OK(hipStreamCreateWithFlags(&C.stream, hipStreamNonBlocking), ABPOA_HIP_ENODEV);
        C.device = device;
    }
    fallback->clear();
    if (stats) memset(stats, 0, sizeof(*stats));
    const double t_begin = now_s();
    // values per DP column in an arena of score records: one padded cell record of the fast loops (4 / 8 values) = the planes of the general kernel (engine.cpp
    //  pv)
    const int CW = sc->gap_mode == ABPOA_HIP_LINEAR_GAP ? 2 : (sc->gap_mode == ABPOA_HIP_AFFINE_GAP ? 4 : 8);      // (linear gaps: {H, match flag} in the fast loops, H alone in the general kernel)
    const bool unbanded = sc->wb < 0;
    // ---- sizes
    int max_reads = 0, max_qlen = 0; int64_t tot_reads = 0, tot_bases = 0, max_cap0 = 0;
    for (int s = 0; s < n_sets; ++s) {
        max_reads = std::max(max_reads, sets[s].n_reads); tot_reads += sets[s].n_reads;
        int64_t sum = 0; int mx = 0;
        for (int r = 0; r < sets[s].n_reads; ++r) { max_qlen = std::max(max_qlen, sets[s].lens[r]); mx = std::max(mx, sets[s].lens[r]); sum += sets[s].lens[r];
                }
        tot_bases += sum; max_cap0 = std::max(max_cap0, std::min<int64_t>(2 + sum, 2 + (int64_t)(node_factor * mx) + 1024));
    }
    // Which kernels: the fast row loops (rows_fast.h: banded global, affine / convex; rows_local.h: local, int16, up to 575 columns) or -- `general` -- the
    // general kernel (rows_general.h: linear gaps, extension mode, global without a band, longer local reads), one launch per round like the wide-band jobs.
    bool general;
    {   LdsPlan pl; int32_t inf_d; const int mb = abpoa_hip_score_bits(sc, (int)max_cap0, max_qlen, &inf_d); const int pn_ = mb == 16 ? 16 : 8;
        const int64_t width_ = (int64_t)((max_qlen + pn_) / pn_) * pn_, w_ = sc->wb + (int)(sc->wf * (float)max_qlen);
        make_lds_plan(sc, max_qlen, mb, (local || unbanded) ? width_ : std::min<int64_t>(width_, 2LL * w_ + 3 * pn_ + 32), n_sets, &pl);
        // (extension mode, round 5: the same banded rows plus the running best cell / z-drop of reference :1018-1026 -- rows_fast.h commit_row)
        bool fast_global = fast_global_job(sc->gap_mode, sc->align_mode, sc->wb, sc->gap_ext1) && pl.fr_cols > 0 && max_qlen <= pl.q_cap;
        // (linear gaps, round 5: the narrow row loop only -- every alignment of the job must take it, dp_common.h takes_fast: band half-widths below the wide
        //  loop's, no read-set with ragged ends; anything else is the general kernel's as before)
        if (fast_global && sc->gap_mode == ABPOA_HIP_LINEAR_GAP) {
            if (w_ >= LINEAR_FAST_W) fast_global = false;
            for (int s = 0; s < n_sets && fast_global; ++s) if (msa_device_set_is_ragged(sets[s])) fast_global = false;      // (the `extra` rule below)
        }
        const bool fast_local = local && sc->gap_mode != ABPOA_HIP_LINEAR_GAP && mb == 16 && pl.loc_cols > 0 && (max_qlen / 16 + 1) * 16 <= pl.loc_cols
                && max_qlen <= pl.q_cap;
        general = !(fast_global || fast_local);
        if (opt_env("ABPOA_HIP_DEVICE_GENERAL") && atoi(opt_env("ABPOA_HIP_DEVICE_GENERAL"))) general = true;      // (tests: the general kernel for every job)
        if (force_general) general = true;
    }
    // direction-plane arenas (dir_plane.h) whenever the penalties allow it: 2 / 4 bytes per cell instead of 8 - 32; ABPOA_HIP_NODIR=1 keeps the score records
    // the last pass of the ladder (msa_hip.cpp device_passes): edge slots for one edge per read at every node -- a node takes at most one new in-edge and one new
    // out-edge per read, so a set can no longer run out of them (the terminals keep their pools: reads that start / end on different nodes); score records
    // instead of direction words there (dir_plane.h names a predecessor by its list index in four bits)
    const bool roomy = node_factor >= 4096.0;
    const int in_cap = roomy ? std::max((int)POA_IN_CAP, std::min(250, max_reads + 1)) : POA_IN_CAP, out_cap = roomy ? std::max((int)POA_OUT_CAP, std::min(250, max_reads + 1)) : POA_OUT_CAP;
    const bool dir = !local && !extend && !general && !amb && in_cap <= POA_IN_CAP && dir_plane_usable(sc->gap_mode, sc->gap_open1, sc->gap_ext1, sc->gap_open2,
            sc->gap_ext2) && !(opt_env("ABPOA_HIP_NODIR") && atoi(opt_env("ABPOA_HIP_NODIR"))) &&
                     !(opt_env("ABPOA_HIP_TEAM") && atoi(opt_env("ABPOA_HIP_TEAM")) > 1);
    const int DB = sc->gap_mode == ABPOA_HIP_AFFINE_GAP ? 2 : 4;

    std::vector<PoaSet> ps(n_sets);
    int64_t node_tot = 0, pred_tot = 0, cig_tot = 0, scr_tot = 0, plane_tot = 0, read_i = 0, cons_tot = 0, term_tot = 0; int max_node_cap = 0;
    const int w_max = sc->wb + (int)(sc->wf * (float)max_qlen);
    const int aln_cap = std::max(1, sc->m - 1), rid_words = want_msa ? std::max(1, (max_reads + 63) / 64) : 0;
    // columns per row: the whole query without a band
    auto est_cols = [&](int64_t width, int w, int pn) { return (local || unbanded) ? width : std::min<int64_t>(width, 2LL * w + 3 * pn + 32); };
    // band half-widths that take the wide row loop (LdsPlan.wide_w_lo / hi; none when the wide kernels are off), depth of its score ring
    int wide_lo = 1, wide_hi = 0, wide_ring_rows = 16;
    { LdsPlan pl; int32_t inf_d; const int mb = abpoa_hip_score_bits(sc, 3 * max_qlen + 1024, max_qlen, &inf_d); const int pn_ = mb == 16 ? 16 : 8;
      make_lds_plan(sc, max_qlen, mb, est_cols((int64_t)((max_qlen + pn_) / pn_) * pn_, w_max, pn_), n_sets, &pl);
      if (pl.wide_nw >= 1 && !local && !general) { wide_lo = pl.wide_w_lo; wide_hi = pl.wide_w_hi; wide_ring_rows = pl.wfr_rows; } }
    // cigar slots: four times the words of a backtrack where the all-rounds kernel's helper wavefronts write their parts (backtrack_dir.h SPEC_WK,
    //  dir_walk_pair)
    // Reads of very different lengths (ends cut at different places; a short read against a long graph): the band is anchored at `qlen - remaining length`
    // (reference abpoa_align.h:34-35), which then sits as far from the alignment's path as the lengths differ, and every row is that much wider than 2 w.
    // Such a set gets `extra` columns in its arena and ring estimates, and its alignments take the wide row loop as if half of them were band half-width
    // (AlnDesc.pad0, dp_common.h takes_wide). Lengths within an eighth of the longest read (at least 64 bases; indel noise: a 25 %-error 400-base set spreads 8
    //  %) count as equal: the estimates' own slack
    // (3 vectors + 32 columns) covers those.
    // (route: the part of `extra` that counts for the choice of the row loop)
    std::vector<int> extra(n_sets, 0), route(n_sets, 0); int max_extra = 0, weff_lo = INT_MAX, weff_hi = 0;
    // (experiments: sets with less extra keep the narrow loop)
    const int route_min = opt_env("ABPOA_HIP_EXTRA_ROUTE_MIN") ? atoi(opt_env("ABPOA_HIP_EXTRA_ROUTE_MIN")) : 0;
    if (!local && !general && sc->wb >= 0) for (int s = 0; s < n_sets; ++s) {
        int mx = 0, mn = INT_MAX; for (int r = 0; r < sets[s].n_reads; ++r) { mx = std::max(mx, sets[s].lens[r]); mn = std::min(mn, sets[s].lens[r]); }
        if (sets[s].n_reads < 2) continue;
        const int spread = mx - mn, tol = std::max(64, mx / 8);
        extra[s] = spread > tol ? std::min((spread + 15) & ~15, 2048) : 0;
        max_extra = std::max(max_extra, extra[s]);
        route[s] = extra[s] >= route_min ? extra[s] : 0;
        weff_lo = std::min(weff_lo, sc->wb + (int)(sc->wf * (float)mn) + route[s] / 2);
        weff_hi = std::max(weff_hi, sc->wb + (int)(sc->wf * (float)mx) + route[s] / 2);
    }
    if (weff_hi == 0) { weff_lo = 0; }
    // (linear gaps on the fast loops keep H records -- no direction words -- and take the all-rounds kernel with them)
    const bool lin_fast = !general && !local && !extend && sc->gap_mode == ABPOA_HIP_LINEAR_GAP && !amb;      // (extension mode: the row order is rebuilt before every read)
    const bool rounds_possible = (dir || lin_fast) && max_reads > 2 && !(w_max >= wide_lo && wide_hi >= wide_lo) && max_extra == 0;
    // Wide-band sets (10 kb reads) keep score records while the record arenas of the whole job fit the device -- their all-chunks row loop is 18-21 % slower
    // with the words, more than the backtrack gains -- and switch to direction words when they do not: an eighth of the bytes per cell, so twice the
    // read-sets are in flight instead of two passes with half the SIMDs idle.  ABPOA_HIP_DIR_WIDE=1 / 0: always / never.
    bool dir_wide = false, any_wide_set = false;
    { const char *e_ = opt_env("ABPOA_HIP_DIR_WIDE"); if (dir && e_ && atoi(e_) > 0) dir_wide = true; }
    const bool dir_wide_auto = dir && !opt_env("ABPOA_HIP_DIR_WIDE");
    // ... and whenever the pass is large enough for two wavefronts per SIMD (the LDS plan then takes a 4-row ring: eight workgroups per CU): the
    // backtrack over words is 2.5x faster there than over records (configs[3] x 2048: 243 vs 609 ms per step), more than the row loop loses (1910 vs 1670 ms)
    if (dir_wide_auto && wide_ring_rows <= 4 && wide_hi >= wide_lo) dir_wide = true;
    for (int s = 0; s < n_sets; ++s) {
        PoaSet &S = ps[s]; memset(&S, 0, sizeof(S));
        int64_t sum = 0; int mx = 0;
        for (int r = 0; r < sets[s].n_reads; ++r) { sum += sets[s].lens[r]; mx = std::max(mx, sets[s].lens[r]); }
        // graph nodes this set may grow to on the device (a set with ragged ends gets one read length more: its reads reach beyond each other's ends, and a
        // straggler that needs a second pass costs the whole job that pass's latency -- 3 of 1024 such sets were 149 ms on top of 216)
        const int64_t cap = std::min<int64_t>(2 + sum, 2 + (int64_t)((node_factor + (extra[s] > 0 ? 1.0 : 0.0)) * mx) + 1024);
        S.n_reads = sets[s].n_reads; S.node_cap = (int)cap; S.pred_cap = (int)(4 * cap);
        S.read0 = read_i; read_i += sets[s].n_reads;
        S.term0 = term_tot; term_tot += sets[s].n_reads + 2;      // (source out-edges / sink in-edges beyond the per-node slots: at most one of each per read)
        S.node0 = node_tot; node_tot += cap + 1;
        S.pred0 = pred_tot; pred_tot += S.pred_cap;
        // (four times: parts 1-3 take the words of the helper wavefronts)
        S.cigar_cap = (int)(cap + mx + 8); S.cigar_off = cig_tot; cig_tot += ((rounds_possible && S.cigar_cap < 65536) ? 4 : 1) * (int64_t)S.cigar_cap;
        // (fuse: 3 x qlen + nodes; order / rank passes: up to four tables of one int per node)
        S.scratch0 = scr_tot; scr_tot += 3LL * max_qlen + 4 * cap + 8;
        S.cons_cap = (int)std::min<int64_t>(cap, 2LL * mx + 64); S.cons0 = cons_tot; cons_tot += S.cons_cap;
        const int w = sc->wb + (int)(sc->wf * (float)mx) + route[s] / 2;      // (for the choice of the row loop: dp_common.h takes_wide)
        S.band_extra = route[s];
        max_node_cap = std::max(max_node_cap, (int)cap);
        any_wide_set |= (w >= wide_lo && w <= wide_hi);
    }
    // arenas: the widest score type a set can reach decides the cell size; columns per row as the band estimate of engine.cpp
    auto size_arenas = [&](bool dw) {
        plane_tot = 0;
        for (int s = 0; s < n_sets; ++s) {
            PoaSet &S = ps[s]; int mx = 0; for (int r = 0; r < sets[s].n_reads; ++r) mx = std::max(mx, sets[s].lens[r]);
            const int64_t cap = S.node_cap;
            int32_t inf_dummy; const int bits = abpoa_hip_score_bits(sc, (int)cap, mx, &inf_dummy); const int pn = bits == 16 ? 16 : 8;
            const int64_t width = (int64_t)((mx + pn) / pn) * pn;
            const int w = sc->wb + (int)(sc->wf * (float)mx);
            int64_t est = std::min<int64_t>(width, est_cols(width, w, pn) + extra[s]);
            // (a set comes back to a later pass also because its ROWS were wider than the estimate -- extension mode on reads that end early, the band pushed off
            //  its anchor -- and for a set of a few reads the node slots of every pass are the same number, the sum of its reads: the later passes grow the columns
            //  with the slots, the last one takes whole rows while that stays under 1 GB per set; found by tools/fuzz_device_vs_oracle.py seed 770500103)
            int64_t sum_len = 0; for (int r = 0; r < sets[s].n_reads; ++r) sum_len += sets[s].lens[r];
            const bool slots_fixed = 2 + sum_len <= 2 + (int64_t)(3.0 * mx) + 1024;      // (a few reads: the 3x estimate already is the bound, no pass has more node slots or rows)
            // (sets whose slots DO grow keep the plain estimate in passes 2 and 3: 10 kb reads at 15 % error start at 4.5x / 6x, and wider arenas would halve the
            //  read-sets a pass holds -- configs[2] 405 -> 281 read-sets/s when this first went in for every set)
            if (node_factor > 3.0 && (roomy || slots_fixed)) {
                const double cellb = dir ? (double)(DB + 8) : (double)CW * (bits / 8);
                int64_t e2 = roomy ? width : std::min<int64_t>(width, (int64_t)((double)est * node_factor / 3.0));
                if (roomy && (double)e2 * (double)cap * cellb > 1e9) e2 = std::min<int64_t>(width, est * 4);
                est = std::max(est, e2);
            }
            // (direction words for every row, score records for the first row and for about one row in four -- rows a successor beyond the score ring or the
            //  global best will read from HBM; half of the rows where the wide loop's ring is only four rows deep; a set that needs more is flagged and
            //  redone like any other capacity miss)
            const bool wide_s = !local && w + route[s] / 2 >= wide_lo && w + route[s] / 2 <= wide_hi;
            const bool dir_s = dir && (dw || !wide_s);      // (dp_common.h takes_dir)
            const int64_t rec_div = (wide_s && wide_ring_rows <= 4) ? 2 : 4;
            // (bytes per cell record of a row that keeps its scores: CW values -- the wide kernel's compact records: 4 B int16 affine, else 8 B; rows_fast.h
            //  CWR)
            const int64_t recb = wide_s ? ((bits == 16 && CW == 4) ? 4 : 8) : CW * (bits / 8);
            // (local row loop, rows_local.h: every row the whole query wide, cell records, 64 records of slack behind the last row)
            const int64_t bytes = local ? (int64_t)up((size_t)((width * cap + 64) * CW * (bits / 8) + 64 * 8 * 4))
                                : dir_s ? (int64_t)up((size_t)(width * (DB + recb) + (est * DB + est * recb / rec_div + 32) * (cap - 1) + 64 * 8 * 4))
                                      : (int64_t)up((size_t)((width + est * (cap - 1)) * CW * (bits / 8) + 64 * 8 * 4));
            S.plane_off = plane_tot; S.plane_cap = bytes - 64 * 8 * 4; plane_tot += bytes;
        }
    };
    size_arenas(dir_wide);
    Layout L; size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = up(o + bytes); return at; };
    // (o_rargs, o_msaoff_h: host-side staging only)
    L.o_sets = take(sizeof(PoaSet) * n_sets); L.o_roff = take(8 * (tot_reads + 1)); L.o_rlen = take(4 * (tot_reads + 1)); L.o_mat = take(4 * sc->m * sc->m);
    L.o_rargs = take(poa_rounds_args_bytes()); L.o_msaoff_h = take(want_msa ? 8 * (size_t)n_sets : 0);
    bool any_w = false; for (int s = 0; s < n_sets && !any_w; ++s) any_w = sets[s].weights != nullptr;
    L.o_wts = take(any_w ? 4 * (size_t)(tot_bases + 64) : 0);      // (per-base weights, -Q: in front of the reads, so that they go up with the first part)
    L.o_reads = take(tot_bases + 64); L.in_bytes = o;      // reads last: they go up in two parts
    // -s: reverse complements / reversed weights of the reads under retry
    L.o_rc = take(amb ? tot_bases + 64 : 0); L.o_wrc = take(amb && any_w ? 4 * (size_t)(tot_bases + 64) : 0); L.in_dev_bytes = o;
    o = 0;
    L.o_state = take(sizeof(PoaState) * n_sets);
    // downloaded part first, contiguous: per-set state and the consensus results
    L.o_cnode = take(4 * cons_tot); L.o_ccov = take(4 * cons_tot); L.o_cbase = take(cons_tot); L.o_isrc = take(amb ? (size_t)tot_reads : 0);
    const size_t dl_bytes = o;
    L.o_order0 = take(4 * node_tot); L.o_order1 = take(4 * node_tot); L.o_base = take(node_tot); L.o_nout = take(node_tot);
    L.o_out = take(4 * node_tot * POA_HOT); L.o_outw = take(4 * node_tot * POA_HOT); L.o_nread = take(4 * node_tot);
    L.o_outx = take(4 * node_tot * (size_t)(out_cap - POA_HOT)); L.o_outwx = take(4 * node_tot * (size_t)(out_cap - POA_HOT));
    L.o_inx = take(4 * node_tot * (size_t)(in_cap - POA_HOT));
    L.o_nin = take(node_tot); L.o_naln = take(node_tot); L.o_in = take(4 * node_tot * POA_HOT); L.o_aln = take(4 * node_tot * (size_t)aln_cap);
    L.o_row = take(4 * node_tot);
    L.o_rid = take(8 * node_tot * (size_t)out_cap * (size_t)rid_words); L.o_mrank = take(want_msa ? 4 * node_tot : 0);
    L.o_msaoff = take(want_msa ? 8 * (size_t)n_sets : 0);
    L.o_tout = take(4 * term_tot); L.o_toutw = take(4 * term_tot); L.o_tin = take(4 * term_tot);
    L.graph_bytes = o;
    o = 0;
    L.o_ticket = take(4 * POA_CU_TICKETS);
    L.o_aln_desc = take(sizeof(AlnDesc) * n_sets); L.o_out_rec = take(sizeof(AlnOut) * n_sets);
    L.o_rbase = take(node_tot); L.o_rsd = take(node_tot); L.o_rpd = take(8 * node_tot); L.o_rnid = take(4 * node_tot); L.o_rrem = take(4 * node_tot);
    L.o_poff = take(4 * node_tot); L.o_pred = take(4 * (pred_tot + 1));
    L.o_bsn = take(4 * node_tot); L.o_esn = take(4 * node_tot); L.o_coff = take(8 * node_tot); L.o_rmi = take(4 * node_tot);
    L.o_cigar = take(8 * cig_tot); L.o_scratch = take(4 * scr_tot);
    // general kernel (rows_general.h): successor CSR, band state per row, the "row is part of the alignment" bytes (all ones: no sub-graph alignments here)
    const bool gen_io = general || amb;      // (-s: the retry runs in the general kernel)
    L.o_ooff = take(gen_io ? 4 * node_tot : 0); L.o_orow = take(gen_io ? 4 * (pred_tot + 1) : 0); L.o_left = take(gen_io ? 4 * node_tot : 0);
    L.o_right = take(gen_io ? 4 * node_tot : 0); L.o_act = take(gen_io ? node_tot : 0);
    L.o_outfwd = take(amb ? sizeof(AlnOut) * n_sets : 0); L.o_cigfwd = take(amb ? 8 * cig_tot : 0); L.o_retry = take(amb ? (size_t)n_sets : 0);
    L.rows_bytes = o;

    {   // the whole job must fit (the caller splits very large jobs): checked on the computed layout, before any cached buffer is given up
        size_t free_b = 0, total_b = 0; (void)hipMemGetInfo(&free_b, &total_b);
        const size_t want[4] = {L.in_dev_bytes, L.graph_bytes, L.rows_bytes, (size_t)plane_tot}, have[4] = {C.in.dev_cap, C.graph.dev_cap, C.rows.dev_cap,
                C.planes.dev_cap};
        size_t need = 0, given_back = 0;
        for (int i = 0; i < 4; ++i) if (want[i] > have[i]) { need += want[i]; given_back += have[i]; }      // a buffer that must grow is freed first
        // the record arenas of the wide-band sets do not fit: direction words for them too
        if (need > free_b + given_back && dir_wide_auto && any_wide_set && !dir_wide) {
            dir_wide = true; size_arenas(true);
            need = 0; given_back = 0;
            const size_t want2[4] = {L.in_dev_bytes, L.graph_bytes, L.rows_bytes, (size_t)plane_tot};
            for (int i = 0; i < 4; ++i) if (want2[i] > have[i]) { need += want2[i]; given_back += have[i]; }
            if (opt_env("ABPOA_HIP_VERBOSE")) fprintf(stderr,
                    "[abpoa-hip] device %d: %d sets: score-record arenas do not fit, direction words for the wide-band sets " "too (arenas %.1f GB)\n", device,
                    n_sets, plane_tot / 1e9);
        }
        if (need > free_b + given_back) {
            if (opt_env("ABPOA_HIP_VERBOSE")) fprintf(stderr,
                    "[abpoa-hip] device %d: %d sets need %.1f GB in growing buffers (arenas %.1f GB), %.1f GB free + %.1f " "GB given back: splitting\n",
                    device, n_sets, need / 1e9, plane_tot / 1e9, free_b / 1e9, given_back / 1e9);
            set_err("device-resident job needs %zu more bytes, %zu free", need, free_b + given_back); return ABPOA_HIP_ENOMEM;
        }
    }
    int rc;
    if ((rc = C.in.need_dev(L.in_dev_bytes)) || (rc = C.in.need_host(L.in_bytes)) || (rc = C.graph.need_dev(L.graph_bytes)) || (rc =
            C.graph.need_host(dl_bytes)) ||
        (rc = C.rows.need_dev(L.rows_bytes)) || (rc = C.planes.need_dev((size_t)plane_tot))) return rc;
    const int n_ev = 4 * max_reads + 8;
    while ((int)C.ev.size() < n_ev) { hipEvent_t e; HIP_OK(hipEventCreate(&e), ABPOA_HIP_ENODEV); C.ev.push_back(e); }

    // ---- upload: set table, reads (already residue codes), score matrix
    uint8_t *hi = C.in.host;
    memcpy(hi + L.o_sets, ps.data(), sizeof(PoaSet) * n_sets);
    // Reads are laid out round by round (read k of every set, then read k + 1 ...): the first two rounds go up at once, the rest is staged
    // and copied while the GPU already works on round 1 (config 2: 50 MB of residue codes, ~1 ms of staging + ~1 ms of PCIe)
    int64_t *roff = (int64_t *)(hi + L.o_roff); int32_t *rlen = (int32_t *)(hi + L.o_rlen); uint8_t *rd = hi + L.o_reads;
    int64_t split_at = 0;
    {
        int64_t at = 0;
        for (int k = 0; k < max_reads; ++k) {
            if (k == 2) split_at = at;
            for (int s = 0; s < n_sets; ++s) if (k < sets[s].n_reads) { const int64_t ri = ps[s].read0 + k; roff[ri] = at; rlen[ri] = sets[s].lens[k];
                    at += sets[s].lens[k]; }
        }
        if (max_reads <= 2) split_at = at;
        roff[tot_reads] = at;
    }
    auto stage_reads = [&](int k_lo, int k_hi) {
        parallel_ranges(std::min(n_threads, 16), n_sets, [&](int lo, int hi_) {
            for (int s = lo; s < hi_; ++s) { const int64_t r0 = ps[s].read0; const int ke = std::min(k_hi, sets[s].n_reads);
                    for (int r = k_lo; r < ke; ++r) memcpy(rd + roff[r0 + r], sets[s].seqs[r], sets[s].lens[r]); }
        });
    };
    stage_reads(0, 2);
    if (any_w) {      // weights of every read, same offsets as the bases; a read (or a set) without weights counts 1 per base
        int32_t *wd = (int32_t *)(hi + L.o_wts);
        parallel_ranges(std::min(n_threads, 16), n_sets, [&](int lo, int hi_) {
            for (int s = lo; s < hi_; ++s) for (int r = 0; r < sets[s].n_reads; ++r) {
                int32_t *dst = wd + roff[ps[s].read0 + r]; const int32_t *src = sets[s].weights ? sets[s].weights[r] : nullptr;
                if (src) memcpy(dst, src, 4 * (size_t)sets[s].lens[r]); else for (int j = 0; j < sets[s].lens[r]; ++j) dst[j] = 1;
            }
        });
    }
    memcpy(hi + L.o_mat, sc->mat, 4 * sc->m * sc->m);
    hipStream_t st = C.stream;
    if (!C.copy_stream) { HIP_OK(hipStreamCreateWithFlags(&C.copy_stream, hipStreamNonBlocking), ABPOA_HIP_ENODEV);
            HIP_OK(hipEventCreateWithFlags(&C.ev_copy, hipEventDisableTiming), ABPOA_HIP_ENODEV); }
    HIP_OK(hipMemcpyAsync(C.in.dev, hi, L.o_reads + (size_t)split_at, hipMemcpyHostToDevice, st), ABPOA_HIP_ELAUNCH);
    bool rest_up = split_at >= roff[tot_reads];      // nothing left for the second part

    // ---- kernel arguments
    uint8_t *di = C.in.dev, *dg = C.graph.dev, *dr = C.rows.dev;
    PoaDev p; memset(&p, 0, sizeof(p));
    p.n_sets = n_sets; p.m = sc->m; p.max_mat = sc->max_mat; p.min_mis = sc->min_mis; p.o1 = sc->gap_open1; p.e1 = sc->gap_ext1; p.o2 = sc->gap_open2;
    p.e2 = sc->gap_ext2;
    p.wb = sc->wb; p.wf = sc->wf; p.gap_mode = sc->gap_mode; p.max_qlen = max_qlen;
    p.last_pass = node_factor >= 6.0 ? 1 : 0;      // (msa_hip.cpp device_passes: 3x, 4.5x, 6x)
    p.in_cap = in_cap; p.out_cap = out_cap;
    p.dig_on = cigar_digest_on() ? 1 : 0;      // (tests: the fuse phase folds every graph cigar into PoaState.cigar_dig)
    // (the reference's own row order where the best cell is the FIRST row that reaches the maximum: local and extension mode, ref :1012-1026; the remaining
    //  length where something reads it: the adaptive band and the z-drop test)
    p.aln_cap = aln_cap; p.rid_words = rid_words; p.order_mode = (local || extend) ? 1 : 0; p.banded = (sc->wb >= 0 || sc->zdrop > 0) ? 1 : 0;
    p.general = general ? 1 : 0; p.msa_rows = 0; p.msa_cons = (want_msa && want_cons) ? 1 : 0;
    // LDS tables of the order / rank kernels (two ints per node; the rank pass packs four tables into the same space): up to 6000 nodes = 52 KB, three
    //  workgroups per CU
    p.order_lds = (p.order_mode || want_msa) ? std::min(((max_node_cap + 3) & ~3), 6000) : 0;
    // (the all-in-LDS order walk: what is left of 40 KB -- four workgroups per CU -- after 13 bytes a node goes to aligned-list entries, 2 bytes each)
    p.order_ecap = p.order_mode ? std::max(1024, std::min(65535, (40 * 1024 - 128 - 13 * p.order_lds) / 2)) : 0;
    { const char *e_ = opt_env("ABPOA_HIP_ORDER_LDS"); if (e_ && !atoi(e_)) p.order_ecap = 0; }      // (ABPOA_HIP_ORDER_LDS=0: the general walk everywhere)
    // (tests: graphs above this many nodes take the walks with tables in memory)
    { const char *e_ = opt_env("ABPOA_HIP_ORDER_CAP"); if (e_ && atoi(e_) >= 0) p.order_lds = std::min(p.order_lds, atoi(e_) & ~3); }
    // per-row records of the prepare kernel in LDS (5 bytes a row, 40 KB at most: four workgroups per CU still fit)
    p.pad = max_node_cap <= 8000 ? ((max_node_cap + 3) & ~3) : 0;
    p.sets = (const PoaSet *)(di + L.o_sets); p.state = (PoaState *)(dg + L.o_state);
    p.read_off = (const int64_t *)(di + L.o_roff); p.read_len = (const int32_t *)(di + L.o_rlen); p.reads = di + L.o_reads;
    p.wts = any_w ? (const int32_t *)(di + L.o_wts) : nullptr;
    p.nd_base = dg + L.o_base; p.nd_nin = dg + L.o_nin; p.nd_nout = dg + L.o_nout; p.nd_naln = dg + L.o_naln;
    p.nd_in = (int32_t *)(dg + L.o_in); p.nd_out = (int32_t *)(dg + L.o_out); p.nd_outw = (int32_t *)(dg + L.o_outw); p.nd_aln = (int32_t *)(dg + L.o_aln);
    p.nd_inx = (int32_t *)(dg + L.o_inx); p.nd_outx = (int32_t *)(dg + L.o_outx); p.nd_outwx = (int32_t *)(dg + L.o_outwx);
    p.nd_nread = (int32_t *)(dg + L.o_nread); p.nd_row = (int32_t *)(dg + L.o_row);
    p.t_out = (int32_t *)(dg + L.o_tout); p.t_outw = (int32_t *)(dg + L.o_toutw); p.t_in = (int32_t *)(dg + L.o_tin);
    p.nd_rid = (uint64_t *)(dg + L.o_rid); p.msa_rank = (int32_t *)(dg + L.o_mrank); p.msa_off = (const int64_t *)(dg + L.o_msaoff); p.msa_out = nullptr;
    p.row_node[0] = (int32_t *)(dg + L.o_order0); p.row_node[1] = (int32_t *)(dg + L.o_order1);
    p.scratch = (int32_t *)(dr + L.o_scratch);
    p.aln = (AlnDesc *)(dr + L.o_aln_desc); p.out = (AlnOut *)(dr + L.o_out_rec);
    p.row_base = dr + L.o_rbase; p.row_sdist = dr + L.o_rsd; p.row_pd = (uint32_t *)(dr + L.o_rpd); p.row_node_id = (int32_t *)(dr + L.o_rnid);
    p.row_remain = (int32_t *)(dr + L.o_rrem);
    p.pred_off = (int32_t *)(dr + L.o_poff); p.pred_row = (int32_t *)(dr + L.o_pred); p.cigar = (uint64_t *)(dr + L.o_cigar);
    p.out_off = gen_io ? (int32_t *)(dr + L.o_ooff) : nullptr; p.out_row = gen_io ? (int32_t *)(dr + L.o_orow) : nullptr;
    if (amb) {
        p.reads_rc = di + L.o_rc; p.wts_rc = any_w ? (int32_t *)(di + L.o_wrc) : nullptr; p.is_rc = dg + L.o_isrc; p.retry = dr + L.o_retry;
        p.out_fwd = (AlnOut *)(dr + L.o_outfwd); p.cigar_fwd = (uint64_t *)(dr + L.o_cigfwd);
    }
    p.cons_node = (int32_t *)(dg + L.o_cnode); p.cons_cov = (int32_t *)(dg + L.o_ccov); p.cons_base = dg + L.o_cbase;

    DevBatch b; memset(&b, 0, sizeof(b));
    b.n = n_sets; b.m = sc->m;
    {
        int32_t inf_dummy; const int max_bits = abpoa_hip_score_bits(sc, max_node_cap, max_qlen, &inf_dummy); const int pn = max_bits == 16 ? 16 : 8;
        const int64_t width = (int64_t)((max_qlen + pn) / pn) * pn;
        // (the rings hold the widest rows expected -- up to 1024 columns: beyond that the plan has no fast row loop at all, and a few ragged sets must not send
        //  the whole job to the host driver; theirs overflow on their own)
        const int64_t est_plain = est_cols(width, w_max, pn), est_ragged = std::min<int64_t>(std::max<int64_t>(est_plain, std::min<int64_t>(1024, width)),
                est_plain + max_extra);
        make_lds_plan(sc, max_qlen, max_bits, est_ragged, n_sets, &b.lds);
        // (a ring that wide does not fit -- int32 scores, convex gaps: the plain estimate then, and the ragged sets' rows that outgrow it are theirs alone)
        if (max_extra > 0 && b.lds.fr_cols == 0) make_lds_plan(sc, max_qlen, max_bits, est_plain, n_sets, &b.lds);
        // (no fast row loop takes anything: dp_common.h takes_fast / rows_local.h takes_local)
        if (general) { b.lds.wide_nw = 0; b.lds.fr_cols = 0; b.lds.loc_cols = 0; }
        // the local row loop (rows_local.h takes_local): int16 scores, at most loc_cols columns, query codes in LDS; anything else is the general kernel's
        else if (local) {
            b.lds.wide_nw = 0;
            if (max_bits != 16 || b.lds.loc_cols <= 0 || (max_qlen / 16 + 1) * 16 > b.lds.loc_cols || max_qlen > b.lds.q_cap) {
                    set_err("local alignment outside the device row loop's range"); return ABPOA_HIP_EINVAL; }
        }
        // score widths the rounds can meet (the width grows with graph and read size): launch only the kernels that can have work
        int min_qlen = max_qlen;
        for (int s = 0; s < n_sets; ++s) for (int r = 1; r < sets[s].n_reads; ++r) min_qlen = std::min(min_qlen, sets[s].lens[r]);
        const int min_bits = abpoa_hip_score_bits(sc, 3, min_qlen, &inf_dummy);
        b.bits_mask = (min_bits == 16 ? 1 : 0) | (max_bits == 32 ? 2 : 0);
        // which row-loop kernels can have work at all: band half-widths of the reads that get aligned (w = b + f * length)
        // (band half-widths as dp_common.h takes_wide counts them: with half of a ragged set's extra columns)
        const int w_min = std::min(sc->wb + (int)(sc->wf * (float)min_qlen), max_extra ? weff_lo : INT_MAX), w_top = std::max(w_max, weff_hi);
        const bool mixed = max_extra > 0;      // (sets with and without extra columns: both loops may have work whatever the extremes say)
        if (!mixed && (w_top < b.lds.wide_w_lo || w_min > b.lds.wide_w_hi)) b.lds.wide_nw = 0;                       // no read takes the wide loop
        b.lds.narrow_off = (!mixed && b.lds.wide_nw >= 1 && w_min >= b.lds.wide_w_lo && w_top <= b.lds.wide_w_hi) ? 1 : 0;      // every read does
    }
    // caller falls back to the host driver
    if (!local && !general && (b.lds.fr_cols == 0 || max_qlen > b.lds.q_cap)) { *want_general = true; set_err("band too wide for the fast row loop"); return ABPOA_HIP_EINVAL; }
    b.o1 = sc->gap_open1; b.e1 = sc->gap_ext1; b.o2 = sc->gap_open2; b.e2 = sc->gap_ext2;
    b.align_mode = sc->align_mode; b.gap_mode = sc->gap_mode; b.wb = sc->wb; b.zdrop = sc->zdrop; b.ret_cigar = 1; b.rev_cigar = 0;
    b.want_trace = 0; b.fresh_band = 1; b.want_lr = 0; b.dbg = 0;
    { const char *dbg_ = opt_env("ABPOA_HIP_DBG"); if (dbg_) b.dbg = atoi(dbg_); }      // (diagnostics: bit 7 keeps the row loop's counters in AlnOut.seg)
    b.mat = (const int32_t *)(di + L.o_mat); b.aln = p.aln; b.out = p.out;
    b.dir_mode = (dir && b.lds.wide_nw <= 1) ? (dir_wide ? 2 : 1) : 0; b.row_sdist = p.row_sdist; b.row_pd = p.row_pd;
    b.query = p.reads; b.row_base = p.row_base; b.row_node_id = p.row_node_id; b.row_remain = p.row_remain; b.row_active = p.row_base;
    b.pred_off = p.pred_off; b.pred_row = p.pred_row; b.out_off = p.pred_off; b.out_row = p.pred_row;
    b.left = p.scratch; b.right = p.scratch;
    if (gen_io) {
        b.out_off = p.out_off; b.out_row = p.out_row; b.left = (int32_t *)(dr + L.o_left); b.right = (int32_t *)(dr + L.o_right); b.row_active = dr + L.o_act;
        HIP_OK(hipMemsetAsync(dr + L.o_act, 1, (size_t)node_tot, st), ABPOA_HIP_ELAUNCH);
    }
    b.dp_beg_sn = (int32_t *)(dr + L.o_bsn); b.dp_end_sn = (int32_t *)(dr + L.o_esn); b.row_cell_off = (int64_t *)(dr + L.o_coff);
    b.row_max_i = (int32_t *)(dr + L.o_rmi);
    b.planes = C.planes.dev; b.cigar = p.cigar;
    // -s: the forward run leaves max_pos_left/right behind (fast row loops: a post-pass, rows_fast.h; general kernel: its own arrays) and the retry starts
    // from them -- the reference sorts, and so resets them, only before the forward alignment (src/abpoa_align.c:329 calls the DP directly)
    DevBatch b_rc = b;
    if (amb) {
        HIP_OK(hipMemsetAsync(dg + L.o_isrc, 0, (size_t)tot_reads, st), ABPOA_HIP_ELAUNCH);
        b.want_lr = (sc->wb >= 0 && !general) ? 1 : 0;
        b_rc = b; b_rc.want_lr = 0; b_rc.fresh_band = sc->wb >= 0 ? 0 : 1;
    }

    // ---- all-rounds kernel (poa_rounds.hip) for jobs whose reads all take the narrow row loop: round 1 runs as separate launches (the upload of the
    //      later reads hides behind it), rounds 2 .. n in ONE launch in which every read-set advances on its own.  ABPOA_HIP_LOCKSTEP=1: one launch
    //      per phase and round throughout (what the wide-band jobs use, and the per-round diagnostics below).
    const bool dbg_sync = opt_env("ABPOA_HIP_DEVSYNC") && atoi(opt_env("ABPOA_HIP_DEVSYNC"));
    bool use_rounds = !local && rounds_possible && !dbg_sync && b.lds.wide_nw == 0 && !(b.dbg & 64) && max_reads > 2 && !(opt_env("ABPOA_HIP_LOCKSTEP")
            && atoi(opt_env("ABPOA_HIP_LOCKSTEP")));
    DevBatch b_r = b; size_t rounds_lds = 0;
    if (use_rounds) {
        // (prepare: 5 bytes per row; fuse: 16 bytes per thread)
        auto dyn_of = [&](const DevBatch &x) { return std::max<size_t>(std::max<size_t>((size_t)x.lds.total_rows, (size_t)x.lds.total_tail),
                std::max<size_t>((size_t)5 * (size_t)(p.pad > 0 ? p.pad : 0), (size_t)16 * 256)); };
        rounds_lds = dyn_of(b_r);
        int st_lds = 0; int nb = poa_rounds_residency(sc->gap_mode, rounds_lds, &st_lds);
        // four workgroups per CU when the job has that many sets: the kernel's static LDS (graph phases) comes out of the backtrack window
        if (nb < 4 && n_sets > nb * 256) {
            const int budget = 160 * 1024 / 4 - st_lds - 512, excess = (int)rounds_lds - budget;
            if (excess > 0 && (size_t)b_r.lds.total_tail == rounds_lds && b_r.lds.bt_bytes_tail - ((excess + 15) & ~15) >= 8 * 1024) {
                b_r.lds.bt_bytes_tail -= (excess + 15) & ~15; b_r.lds.total_tail -= (excess + 15) & ~15;
                rounds_lds = dyn_of(b_r); nb = poa_rounds_residency(sc->gap_mode, rounds_lds, &st_lds);
            }
        }
        // The kernel pays when every read-set of the job is resident at once (one workgroup each; 4 per CU): a larger job runs faster with one
        // launch per phase and round, whose single-wavefront row-loop kernel then has several alignments per SIMD to hide latency behind
        // (measured, 1 kb reads: 1000 sets 13.0 k vs 10.4 k read-sets/s; 2000 sets 12.9 k vs 13.6 k; 4000 sets 13.2 k vs 17.4 k)
        int n_cu = 256; { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) n_cu =
                pr.multiProcessorCount; }
        if (nb < 1 || n_sets > nb * n_cu) use_rounds = false;
        if (opt_env("ABPOA_HIP_VERBOSE")) fprintf(stderr,
                "[abpoa-hip] all-rounds kernel: %s, %zu B dynamic + %d B static LDS per workgroup, %d workgroups per CU, " "backtrack window %d B\n",
                use_rounds ? "on" : "off", rounds_lds, st_lds, nb, b_r.lds.bt_bytes_tail);
    }

    // ---- the whole progressive alignment, queued back to back (ABPOA_HIP_DEVSYNC=1: synchronise and report after every kernel)
    DeviceDebug dbg(&p, &ps, sets, n_sets, sc->m, aln_cap, max_reads, want_msa, amb, st);      // (ABPOA_HIP_DEVSYNC=1; msa_device_debug.h)
    auto stage = [&](const char *what, int k) { return dbg.stage(what, k); };
    const double t_queue = now_s();
    HIP_OK(hipEventRecord(C.ev[0], st), ABPOA_HIP_ELAUNCH);
    if (stage("upload", 0)) return ABPOA_HIP_ELAUNCH;
    HIP_OK(launch_poa_init(p, st), ABPOA_HIP_ELAUNCH);
    if (stage("init", 0)) return ABPOA_HIP_ELAUNCH;
    dbg.graph_check(0);
    HIP_OK(hipEventRecord(C.ev[1], st), ABPOA_HIP_ELAUNCH);
    for (int k = 1; k < max_reads; ++k) {
        if (k == 2 && !rest_up) {      // round 1 is queued: stage and send the reads of the later rounds behind it
            stage_reads(2, max_reads);
            HIP_OK(hipMemcpyAsync(C.in.dev + L.o_reads + (size_t)split_at, hi + L.o_reads + (size_t)split_at, (size_t)(roff[tot_reads] - split_at),
                    hipMemcpyHostToDevice, C.copy_stream), ABPOA_HIP_ELAUNCH);
            HIP_OK(hipEventRecord(C.ev_copy, C.copy_stream), ABPOA_HIP_ELAUNCH);
            HIP_OK(hipStreamWaitEvent(st, C.ev_copy, 0), ABPOA_HIP_ELAUNCH);
            rest_up = true;
        }
        if (use_rounds && k == 2) {      // rounds 2 .. n - 1 of every set in one launch
            HIP_OK(hipEventRecord(C.ev[2], st), ABPOA_HIP_ELAUNCH);
            HIP_OK(launch_poa_rounds(p, b_r, (int32_t *)(dr + L.o_ticket), C.in.host + L.o_rargs, slot, 2, rounds_lds, st), ABPOA_HIP_ELAUNCH);
            HIP_OK(hipEventRecord(C.ev[3], st), ABPOA_HIP_ELAUNCH);
            break;
        }
        p.round = k;
        hipEvent_t *e = C.ev.data() + 4 * k;
        if (p.order_mode) { HIP_OK(launch_poa_order(p, st), ABPOA_HIP_ELAUNCH); if (stage("row order", k)) return ABPOA_HIP_ELAUNCH; dbg.order_check(k); }
        HIP_OK(launch_poa_prepare(p, st), ABPOA_HIP_ELAUNCH);
        if (stage("prepare", k)) return ABPOA_HIP_ELAUNCH;
        HIP_OK(hipEventRecord(e[0], st), ABPOA_HIP_ELAUNCH);
        if (general) { HIP_OK(launch_dp_general(b, st), ABPOA_HIP_ELAUNCH); HIP_OK(hipEventRecord(e[1], st), ABPOA_HIP_ELAUNCH); }
        else HIP_OK(launch_dp_fast(b, st, e[1]), ABPOA_HIP_ELAUNCH);
        if (stage("dp rows + tail", k)) return ABPOA_HIP_ELAUNCH;
        if (amb) {
            HIP_OK(launch_poa_strand_check(p, st), ABPOA_HIP_ELAUNCH);
            HIP_OK(launch_dp_general(b_rc, st), ABPOA_HIP_ELAUNCH);
            HIP_OK(launch_poa_strand_pick(p, st), ABPOA_HIP_ELAUNCH);
            if (stage("strand retry", k)) return ABPOA_HIP_ELAUNCH;
        }
        dbg.balance_report(k, b, e[0], e[1]);
        HIP_OK(hipEventRecord(e[2], st), ABPOA_HIP_ELAUNCH);
        HIP_OK(launch_poa_fuse(p, st), ABPOA_HIP_ELAUNCH);
        if (stage("fuse", k)) return ABPOA_HIP_ELAUNCH;
        dbg.graph_check(k);
        HIP_OK(hipEventRecord(e[3], st), ABPOA_HIP_ELAUNCH);
    }
    // ---- consensus on the device, then one small download: per-set state + consensus (node ids, bases, coverage)
    if (want_cons) { HIP_OK(launch_poa_consensus(p, st), ABPOA_HIP_ELAUNCH); if (stage("consensus", max_reads)) return ABPOA_HIP_ELAUNCH; }
    if (want_msa) { HIP_OK(launch_poa_msa_rank(p, st), ABPOA_HIP_ELAUNCH); if (stage("msa rank", max_reads)) return ABPOA_HIP_ELAUNCH; }
    uint8_t *hg = C.graph.host;
    HIP_OK(hipMemcpyAsync(hg, dg, dl_bytes, hipMemcpyDeviceToHost, st), ABPOA_HIP_ELAUNCH);
    HIP_OK(hipStreamSynchronize(st), ABPOA_HIP_ELAUNCH);
    // ---- MSA rows (reference abpoa_generate_rc_msa, src/abpoa_output.c:123-166): the rank pass left the column count of every set in its state; the results
    //  are
    //      laid out back to back (rows x columns bytes per set), filled on the device and downloaded in one piece
    std::vector<int64_t> msa_off; int64_t msa_total = 0;
    if (want_msa) {
        const PoaState *hs_ = (const PoaState *)(hg + L.o_state);
        msa_off.resize(n_sets);
        for (int s = 0; s < n_sets; ++s) { msa_off[s] = msa_total; if (hs_[s].status == POA_ST_OK && hs_[s].n_nodes > 2) msa_total += (int64_t)(sets[s].n_reads
                + (want_cons ? 1 : 0)) * std::max(0, hs_[s].msa_len); }
        if (msa_total > 0) {
            if ((rc = C.msa.need_dev((size_t)msa_total)) || (rc = C.msa.need_host((size_t)msa_total))) return rc;
            memcpy(hi + L.o_msaoff_h, msa_off.data(), 8 * (size_t)n_sets);
            HIP_OK(hipMemcpyAsync(dg + L.o_msaoff, hi + L.o_msaoff_h, 8 * (size_t)n_sets, hipMemcpyHostToDevice, st), ABPOA_HIP_ELAUNCH);
            p.msa_out = C.msa.dev;
            HIP_OK(launch_poa_msa_fill(p, st), ABPOA_HIP_ELAUNCH);
            if (stage("msa fill", max_reads)) return ABPOA_HIP_ELAUNCH;
            HIP_OK(hipMemcpyAsync(C.msa.host, C.msa.dev, (size_t)msa_total, hipMemcpyDeviceToHost, st), ABPOA_HIP_ELAUNCH);
            HIP_OK(hipStreamSynchronize(st), ABPOA_HIP_ELAUNCH);
        }
    }
    const double t_done = now_s();
    // (library built with -DABPOA_HIP_ORDER_PROF: the order walk's passes and ticks per kind, mean per set)
    if (opt_env("ABPOA_HIP_ORDER_PROF") && p.order_mode) {
        const PoaState *hs_ = (const PoaState *)(C.graph.host + L.o_state); double a_[4] = {0, 0, 0, 0};
        for (int s = 0; s < n_sets; ++s) for (int i = 0; i < 4; ++i) a_[i] += (double)hs_[s].t_phase[i];
        fprintf(stderr, "[abpoa-hip] order walk per set: %.0f single-node passes x %.0f ticks, %.0f parallel passes x %.0f ticks\n", a_[2] / n_sets,
                a_[2] > 0 ? a_[0] / a_[2] : 0.0, a_[3] / n_sets, a_[3] > 0 ? a_[1] / a_[3] : 0.0);
    }
    if (stats) {
        float ms = 0;
        hipEvent_t prev = C.ev[1];
        for (int k = 1; k < (use_rounds ? 2 : max_reads); ++k) {
            hipEvent_t *e = C.ev.data() + 4 * k;
            (void)hipEventElapsedTime(&ms, prev, e[0]); stats->prepare_ms += ms;
            (void)hipEventElapsedTime(&ms, e[0], e[1]); stats->rows_ms += ms;
            (void)hipEventElapsedTime(&ms, e[1], e[2]); stats->tail_ms += ms;
            (void)hipEventElapsedTime(&ms, e[2], e[3]); stats->fuse_ms += ms;
            prev = e[3];
        }
        if (use_rounds) {      // the all-rounds kernel: its duration, split by the shader-clock ticks the sets spent in each phase (PoaState.t_phase)
            (void)hipEventElapsedTime(&ms, C.ev[2], C.ev[3]); stats->rounds_ms = ms; stats->rounds_launches = 1;
            const PoaState *hs_ = (const PoaState *)(C.graph.host + L.o_state);
            double tp[4] = {0, 0, 0, 0}, tmax = 0; for (int s = 0; s < n_sets; ++s) { double t_ = 0;
                    if (hs_[s].status == POA_ST_OK) stats->rounds_algo_bytes += hs_[s].algo_bytes - hs_[s].algo_bytes_before;
            for (int i = 0; i < 4; ++i) { tp[i] += (double)hs_[s].t_phase[i]; t_ += (double)hs_[s].t_phase[i]; } tmax = std::max(tmax, t_); }
            const double tall = tp[0] + tp[1] + tp[2] + tp[3];
            if (tall > 0) { stats->prepare_ms += ms * tp[0] / tall; stats->rows_ms += ms * tp[1] / tall; stats->tail_ms += ms * tp[2] / tall;
                    stats->fuse_ms += ms * tp[3] / tall;
                            for (int i = 0; i < 4; ++i) stats->rounds_mticks[i] = tp[i] / n_sets * 1e-6; stats->rounds_rows_share = tp[1] / tall;
                            stats->rounds_mean_over_max = n_sets > 0 && tmax > 0 ? tall / n_sets / tmax : 0; }
        }
        stats->n_rounds = max_reads > 0 ? max_reads - 1 : 0; stats->device_s = t_done - t_queue;
    }

    // ---- results
    const PoaState *hs = (const PoaState *)(hg + L.o_state);
    const int32_t *h_cnode = (const int32_t *)(hg + L.o_cnode), *h_ccov = (const int32_t *)(hg + L.o_ccov); const uint8_t *h_cbase = hg + L.o_cbase;
    std::vector<char> need_fb(n_sets, 0);
    parallel_ranges(std::min(n_threads, 8), n_sets, [&](int lo, int hi_) {
        for (int s = lo; s < hi_; ++s) {
            abpoa_hip_msa_t &o_ = out[s];
            memset(&o_, 0, sizeof(o_)); o_.n_reads = sets[s].n_reads;
            // (only for a set that finished here: the record of a set that goes to another pass or to the host driver is overwritten there)
            if (amb && hs[s].status == POA_ST_OK) { o_.is_rc = (uint8_t *)calloc((size_t)std::max(1, sets[s].n_reads), 1);
                    if (o_.is_rc) memcpy(o_.is_rc, hg + L.o_isrc + ps[s].read0, (size_t)sets[s].n_reads); }
            if (hs[s].status != POA_ST_OK) { need_fb[s] = hs[s].pad == 5 ? 2 : 1;
                    if (dbg_sync) fprintf(stderr, "[poa-device] set %d falls back to the host driver: reason %d, %d nodes of %d\n", s, hs[s].pad,
                    hs[s].n_nodes, ps[s].node_cap); continue; }
            o_.n_cells = hs[s].n_cells;
            if (p.dig_on && sets[s].n_reads > 0) cigar_digest_set(sets[s].seqs[0], sets[s].lens[0], hs[s].cigar_dig);
            if (want_cons && hs[s].n_nodes > 2) {
                const int len = hs[s].cons_len; const int64_t c0 = ps[s].cons0;
                o_.cons_len = len;
                o_.cons_base = (uint8_t *)malloc(len + 1); o_.cons_cov = (int32_t *)malloc(4 * (len + 1)); o_.cons_node_id = (int32_t *)malloc(4 * (len + 1));
                memcpy(o_.cons_base, h_cbase + c0, len); memcpy(o_.cons_cov, h_ccov + c0, 4 * (size_t)len);
                memcpy(o_.cons_node_id, h_cnode + c0, 4 * (size_t)len);
            }
            if (want_msa && hs[s].n_nodes > 2) {      // (an empty graph has no MSA: the host driver leaves the record zeroed too)
                const int len = std::max(0, hs[s].msa_len);
                o_.msa_len = len; o_.msa_rows = sets[s].n_reads + (want_cons ? 1 : 0);
                o_.msa_base = (uint8_t *)malloc((size_t)o_.msa_rows * (len > 0 ? len : 1));
                if (len > 0) memcpy(o_.msa_base, C.msa.host + msa_off[s], (size_t)o_.msa_rows * len);
            }
        }
    });
    dbg.msa_check(hs, out);
    dbg.consensus_check(hs, out);
    // (a set with a node out of edge slots gains nothing from a pass with more node slots: -(s + 1) tells the caller to take it to the last pass -- `roomy`
    //  above -- at once, and to the host driver if that was this one: a terminal with more than POA_TERM_MAX edges)
    for (int s = 0; s < n_sets; ++s) if (need_fb[s]) fallback->push_back(need_fb[s] == 2 ? -(s + 1) : s);
    if (fallback_reason) {
        fallback_reason->clear();
        for (int s = 0; s < n_sets; ++s) if (need_fb[s]) { const int r = hs[s].pad;
                fallback_reason->push_back(r >= 1000 ? (r - 1000 == ABPOA_HIP_STATUS_OVERFLOW ? 9 : 10) : (r >= 1 && r <= 8 ? r : 0)); }
    }
    if (opt_env("ABPOA_HIP_VERBOSE") && !fallback->empty()) {
        int hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int n_slots = 0;
        int dp_arena = 0, dp_scores = 0, dp_other = 0;      // DP status: arena too small for the rows' bands / direction words could not decide / anything else
        for (int f : *fallback) {
            const int s = f < 0 ? -f - 1 : f; const int r = hs[s].pad;
            if (r == 5) { n_slots++; continue; }
            if (r >= 1000) { if (r - 1000 == ABPOA_HIP_STATUS_OVERFLOW) dp_arena++; else if (r - 1000 == ABPOA_HIP_STATUS_NEED_SCORES) dp_scores++;
                    else dp_other++; }
            hist[r >= 1000 ? 5 : (r >= 0 && r < 5 ? r : (r == 6 ? 7 : 6))]++;
            if (!(r >= 1000 || (r >= 0 && r < 7))) fprintf(stderr,
                    "[abpoa-hip]   set %d: reason code %d (7: row order walk, 8: MSA rank walk, 1000 + a negative DP status: " "%d)\n", s, r, r - 1000);
        }
        if (hist[5]) fprintf(stderr, "[abpoa-hip]   DP status: arena too small for the bands %d, direction words undecided %d, other %d\n", dp_arena,
                dp_scores, dp_other);
        fprintf(stderr, "[abpoa-hip] fallback reasons: node cap at init %d, pred CSR cap %d, cigar cap %d, node slots in the fuse "
                "phase %d, edge slots of a node full (redone in the last pass) %d, DP status %d, projected node growth "
                "(early exit at read 10) %d, other %d\n", hist[1], hist[2], hist[3], hist[4], n_slots, hist[5], hist[7], hist[6] + hist[0]);
    }
    if (stats) {
        stats->cons_s = now_s() - t_done; stats->total_s = now_s() - t_begin;
        for (int s = 0; s < n_sets; ++s) if (!need_fb[s]) {
            stats->n_cells += hs[s].n_cells; stats->algo_bytes += hs[s].algo_bytes; stats->n_alignments += std::max(0, sets[s].n_reads - 1);
            int mx = 0; for (int r = 0; r < sets[s].n_reads; ++r) mx = std::max(mx, sets[s].lens[r]);
            if ((int64_t)hs[s].n_nodes <= 2 + (int64_t)(3.0 * mx) + 1024) stats->n_fit_3x += 1;
        }
    }
    return ABPOA_HIP_OK;
}


int run_msa_device(const abpoa_hip_scoring_t *sc, int n_sets, const abpoa_hip_readset_t *sets, abpoa_hip_msa_t *out, int n_threads,
                   std::vector<int> *fallback, DeviceRunStats *stats, double node_factor, unsigned flags, int device, int slot, std::vector<int> *fallback_reason) {
    bool want_general = false;
    const int rc = run_msa_device_body(sc, n_sets, sets, out, n_threads, fallback, stats, node_factor, flags, device, slot, false, &want_general, fallback_reason);
    if (rc != ABPOA_HIP_EINVAL || !want_general) return rc;
    // the job stays on the device: the general kernel takes what the fast row loops' final plan could not (it used to leave for the host driver)
    want_general = false;
    return run_msa_device_body(sc, n_sets, sets, out, n_threads, fallback, stats, node_factor, flags, device, slot, true, &want_general, fallback_reason);
}

}  // namespace abpoa_hip
