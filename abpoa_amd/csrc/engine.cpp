// Host side of the flat batch API (include/abpoa_hip.h): validation, packing of N problems into one
// pinned staging blob, one H2D copy, one kernel launch, result unpacking.  Compiled with hipcc.
//
// Replaces the per-alignment driver simd_abpoa_align_sequence_to_subgraph
// (reference src/simd_abpoa_align.c:1645-1712) and the scratch owner simd_abpoa_realloc (:1178-1208):
// instead of one strided rows x (qlen+1) matrix per abpoa_t, a batch shares three grow-only HBM pools
// (inputs, per-row outputs, band-compacted score-plane arenas).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <limits.h>
#include <mutex>
#include <vector>
#include <algorithm>
#include "engine.h"
#include "../../include/abpoa_hip.h"

namespace abpoa_hip {

static thread_local char g_err[512] = "";
static void set_err(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
#define HIP_TRY(expr, code)                                                                          \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) {                                             \
        set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return code; } } while (0)

// A device buffer with a pinned host mirror; grow-only.
struct Blob {
    uint8_t *dev = nullptr, *host = nullptr; size_t cap = 0; bool mirrored;
    explicit Blob(bool m) : mirrored(m) {}
    int reserve(size_t n) {
        if (n <= cap) return 0;
        size_t want = std::max(n, cap + cap / 2);
        want = (want + 0xFFFFF) & ~(size_t)0xFFFFF;
        release();
        if (hipMalloc((void **)&dev, want) != hipSuccess) { dev = nullptr; set_err("hipMalloc(%zu) failed", want); return ABPOA_HIP_ENOMEM; }
        if (mirrored && hipHostMalloc((void **)&host, want, hipHostMallocDefault) != hipSuccess) {
            host = nullptr; set_err("hipHostMalloc(%zu) failed", want); return ABPOA_HIP_ENOMEM; }
        cap = want; return 0;
    }
    void release() { if (dev) (void)hipFree(dev); if (host) (void)hipHostFree(host); dev = host = nullptr; cap = 0; }
};

struct Engine {
    bool ready = false; int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {};
    Blob in{true}, outb{true}, planes{false};
    abpoa_hip_stats_t stats{};
    std::mutex mu;
};
static Engine g;

long long g_dbg[10] = {0};
static size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

}  // namespace abpoa_hip

using namespace abpoa_hip;

extern "C" {

int abpoa_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int abpoa_hip_init(int device) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.ready && g.device == device) return ABPOA_HIP_OK;
    if (g.ready) { set_err("engine already bound to device %d", g.device); return ABPOA_HIP_EINVAL; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_err("no HIP device available (the DP has no CPU fallback)"); return ABPOA_HIP_ENODEV; }
    if (device < 0 || device >= n) { set_err("device %d out of range (0..%d)", device, n - 1); return ABPOA_HIP_ENODEV; }
    HIP_TRY(hipSetDevice(device), ABPOA_HIP_ENODEV);
    HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking), ABPOA_HIP_ENODEV);
    for (auto &e : g.ev) HIP_TRY(hipEventCreate(&e), ABPOA_HIP_ENODEV);
    g.device = device; g.ready = true; memset(&g.stats, 0, sizeof(g.stats));
    return ABPOA_HIP_OK;
}

void abpoa_hip_shutdown(void) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.ready) return;
    (void)hipStreamSynchronize(g.stream);
    g.in.release(); g.outb.release(); g.planes.release();
    for (auto &e : g.ev) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(g.stream);
    g.ready = false; g.device = -1;
}

const char *abpoa_hip_last_error(void) { return g_err; }
void abpoa_hip_get_stats(abpoa_hip_stats_t *out) { std::lock_guard<std::mutex> lk(g.mu); *out = g.stats; }
void abpoa_hip__debug_clocks(long long *out) { for (int i = 0; i < 10; ++i) { out[i] = g_dbg[i]; g_dbg[i] = 0; } }
void abpoa_hip_reset_stats(void) { std::lock_guard<std::mutex> lk(g.mu); memset(&g.stats, 0, sizeof(g.stats)); }

// reference src/simd_abpoa_align.c:1672-1683
int abpoa_hip_score_bits(const abpoa_hip_scoring_t *sc, int n_rows, int qlen, int32_t *inf_min) {
    int oe1 = sc->gap_open1 + sc->gap_ext1, oe2 = sc->gap_open2 + sc->gap_ext2;
    int len = qlen > n_rows ? qlen : n_rows;
    int max_score = std::max(qlen * sc->max_mat, len * sc->gap_ext1 + sc->gap_open1);
    int bits, lo;
    if (max_score <= INT16_MAX - sc->min_mis - oe1 - oe2) { bits = 16; lo = INT16_MIN; } else { bits = 32; lo = INT32_MIN; }
    if (inf_min) *inf_min = std::max(std::max(lo + sc->min_mis, lo + oe1), lo + oe2) + 31 * std::max(sc->gap_ext1, sc->gap_ext2);
    return bits;
}

void abpoa_hip_free_result(abpoa_hip_result_t *r) {
    if (!r) return;
    free(r->cigar);
    if (r->trace) {
        abpoa_hip_trace_t *t = r->trace;
        free(t->dp_beg); free(t->dp_end); free(t->dp_beg_sn); free(t->dp_end_sn); free(t->row_off); free(t->planes); free(t->row_max_i); free(t);
    }
    memset(r, 0, sizeof(*r));
}

static int validate(const abpoa_hip_scoring_t *sc, const abpoa_hip_problem_t *p, int idx) {
    if (p->n_rows < 3 || p->qlen < 0) { set_err("problem %d: n_rows=%d qlen=%d", idx, p->n_rows, p->qlen); return ABPOA_HIP_EINVAL; }
    if (!p->row_base || !p->row_node_id || !p->pred_off || !p->pred_row || !p->out_off || !p->out_row || (p->qlen > 0 && !p->query)) {
        set_err("problem %d: NULL array", idx); return ABPOA_HIP_EINVAL; }
    if ((sc->wb >= 0 || sc->zdrop > 0) && !p->row_remain) { set_err("problem %d: row_remain required", idx); return ABPOA_HIP_EINVAL; }
    if (sc->wb >= 0 && (!p->max_pos_left || !p->max_pos_right)) { set_err("problem %d: max_pos_left/right required when banded", idx); return ABPOA_HIP_EINVAL; }
    const int gn = p->n_rows;
    if (p->pred_off[0] < 0 || p->out_off[0] < 0) { set_err("problem %d: negative CSR offset", idx); return ABPOA_HIP_EINVAL; }
    for (int r = 0; r < gn; ++r) {
        if (p->pred_off[r + 1] < p->pred_off[r] || p->out_off[r + 1] < p->out_off[r]) { set_err("problem %d: CSR offsets not monotone at row %d", idx, r); return ABPOA_HIP_EINVAL; }
        if (p->row_base[r] >= sc->m) { set_err("problem %d: base code %d >= m at row %d", idx, p->row_base[r], r); return ABPOA_HIP_EINVAL; }
        const bool act = !p->row_active || p->row_active[r];
        for (int k = p->pred_off[r]; k < p->pred_off[r + 1]; ++k) {
            int q = p->pred_row[k];
            if (q < 0 || q >= r || (p->row_active && !p->row_active[q])) { set_err("problem %d: row %d has predecessor %d (must be an active earlier row)", idx, r, q); return ABPOA_HIP_EINVAL; }
        }
        if (act && r > 0 && r < gn - 1 && p->pred_off[r + 1] == p->pred_off[r]) { set_err("problem %d: active row %d has no predecessor", idx, r); return ABPOA_HIP_EINVAL; }
        for (int k = p->out_off[r]; k < p->out_off[r + 1]; ++k) {
            int o = p->out_row[k];
            if (o != -1 && (o <= r || o >= gn)) { set_err("problem %d: row %d has successor %d", idx, r, o); return ABPOA_HIP_EINVAL; }
        }
    }
    for (int j = 0; j < p->qlen; ++j) if (p->query[j] >= sc->m) { set_err("problem %d: query code %d >= m", idx, p->query[j]); return ABPOA_HIP_EINVAL; }
    return 0;
}

int abpoa_hip_align_batch(const abpoa_hip_scoring_t *sc, int n, const abpoa_hip_problem_t *pb,
                          abpoa_hip_result_t *res, unsigned flags) {
    if (n < 0 || !sc || (n > 0 && (!pb || !res))) { set_err("bad arguments"); return ABPOA_HIP_EINVAL; }
    if (n == 0) return ABPOA_HIP_OK;
    for (int i = 0; i < n; ++i) memset(&res[i], 0, sizeof(res[i]));
    if (!g.ready) { int rc = abpoa_hip_init(0); if (rc) return rc; }
    if (sc->m <= 0 || !sc->mat || sc->gap_mode < 0 || sc->gap_mode > 2 || sc->align_mode < 0 || sc->align_mode > 2) { set_err("bad scoring"); return ABPOA_HIP_EINVAL; }
    for (int i = 0; i < n; ++i) { int rc = validate(sc, &pb[i], i); if (rc) return rc; }
    std::lock_guard<std::mutex> lk(g.mu);
    HIP_TRY(hipSetDevice(g.device), ABPOA_HIP_ENODEV);

    const int P = sc->gap_mode == ABPOA_HIP_LINEAR_GAP ? 1 : (sc->gap_mode == ABPOA_HIP_AFFINE_GAP ? 3 : 5);
    const bool banded = sc->wb >= 0;
    const bool trace = flags & ABPOA_HIP_FLAG_TRACE;

    // ---- sizes and offsets ----
    std::vector<AlnDesc> desc(n);
    int64_t rows_tot = 0, preds_tot = 0, outs_tot = 0, q_tot = 0, cig_tot = 0;
    std::vector<int64_t> full_cells(n);
    for (int i = 0; i < n; ++i) {
        const abpoa_hip_problem_t &p = pb[i]; AlnDesc &d = desc[i];
        d.n_rows = p.n_rows; d.qlen = p.qlen;
        d.bits = abpoa_hip_score_bits(sc, p.n_rows, p.qlen, &d.inf_min);
        d.w = sc->wb < 0 ? p.qlen : sc->wb + (int)(sc->wf * p.qlen);     // reference :445 (float32 product)
        d.cigar_cap = p.n_rows + p.qlen + 8;
        d.query_off = q_tot; d.row0 = rows_tot; d.poff0 = rows_tot + i; d.pred0 = preds_tot; d.out0 = outs_tot; d.cigar_off = cig_tot;
        q_tot += p.qlen; rows_tot += p.n_rows; preds_tot += p.pred_off[p.n_rows]; outs_tot += p.out_off[p.n_rows]; cig_tot += d.cigar_cap;
        const int pn = d.bits == 16 ? 16 : 8;
        const int64_t width = (int64_t)((p.qlen + pn) / pn) * pn;
        full_cells[i] = width * P * p.n_rows;
        int64_t est = banded ? std::min<int64_t>(width, 2LL * d.w + 3 * pn + 32) : width;
        d.plane_cap = std::min<int64_t>(full_cells[i], width * P + est * P * (p.n_rows - 1));
    }
    // input blob layout
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes); return at; };
    const size_t o_desc = take(sizeof(AlnDesc) * n), o_mat = take(sizeof(int32_t) * sc->m * sc->m), o_query = take(q_tot + 1),
                 o_base = take(rows_tot), o_nid = take(4 * rows_tot), o_rem = take(4 * rows_tot), o_act = take(rows_tot),
                 o_poff = take(4 * (rows_tot + n)), o_pred = take(4 * (preds_tot + 1)), o_ooff = take(4 * (rows_tot + n)), o_out = take(4 * (outs_tot + 1));
    const size_t in_bytes = o;
    // output blob layout
    o = 0;
    const size_t o_rec = take(sizeof(AlnOut) * n), o_left = take(4 * rows_tot), o_right = take(4 * rows_tot), o_bsn = take(4 * rows_tot),
                 o_esn = take(4 * rows_tot), o_coff = take(8 * rows_tot), o_rmi = take(4 * rows_tot), o_cig = take(8 * cig_tot);
    const size_t out_bytes = o;
    int rc;
    if ((rc = g.in.reserve(in_bytes)) || (rc = g.outb.reserve(out_bytes))) return rc;

    // ---- pack ----
    uint8_t *hi = g.in.host, *ho = g.outb.host;
    memcpy(hi + o_mat, sc->mat, sizeof(int32_t) * sc->m * sc->m);
    for (int i = 0; i < n; ++i) {
        const abpoa_hip_problem_t &p = pb[i]; const AlnDesc &d = desc[i]; const int gn = p.n_rows;
        if (p.qlen) memcpy(hi + o_query + d.query_off, p.query, p.qlen);
        memcpy(hi + o_base + d.row0, p.row_base, gn);
        memcpy(hi + o_nid + 4 * d.row0, p.row_node_id, 4 * gn);
        if (p.row_remain) memcpy(hi + o_rem + 4 * d.row0, p.row_remain, 4 * gn); else memset(hi + o_rem + 4 * d.row0, 0, 4 * gn);
        if (p.row_active) memcpy(hi + o_act + d.row0, p.row_active, gn); else memset(hi + o_act + d.row0, 1, gn);
        memcpy(hi + o_poff + 4 * d.poff0, p.pred_off, 4 * (gn + 1));
        memcpy(hi + o_pred + 4 * d.pred0, p.pred_row + 0, 4 * (size_t)(p.pred_off[gn]));
        memcpy(hi + o_ooff + 4 * d.poff0, p.out_off, 4 * (gn + 1));
        memcpy(hi + o_out + 4 * d.out0, p.out_row + 0, 4 * (size_t)(p.out_off[gn]));
        if (p.pred_off[0] != 0 || p.out_off[0] != 0) { set_err("problem %d: CSR offsets must start at 0", i); return ABPOA_HIP_EINVAL; }
    }

    std::vector<int> todo(n); for (int i = 0; i < n; ++i) todo[i] = i;
    bool first_pass = true;
    while (!todo.empty()) {
        // arena offsets for the alignments of this pass
        int64_t plane_bytes = 0;
        std::vector<AlnDesc> pass(todo.size());
        for (size_t t = 0; t < todo.size(); ++t) {
            AlnDesc &d = desc[todo[t]];
            if (!first_pass) d.plane_cap = full_cells[todo[t]];
            d.plane_off = plane_bytes; plane_bytes += (int64_t)align_up((size_t)d.plane_cap * (d.bits / 8));
            pass[t] = d;
        }
        if ((rc = g.planes.reserve((size_t)plane_bytes))) return rc;
        memcpy(hi + o_desc, pass.data(), sizeof(AlnDesc) * pass.size());
        if (banded)   // max_pos_left/right are in/out: stage the caller's values (a retried alignment restarts from them)
            for (int i : todo) {
                memcpy(ho + o_left + 4 * desc[i].row0, pb[i].max_pos_left, 4 * pb[i].n_rows);
                memcpy(ho + o_right + 4 * desc[i].row0, pb[i].max_pos_right, 4 * pb[i].n_rows);
            }

        DevBatch b; memset(&b, 0, sizeof(b));
        b.n = (int)pass.size(); b.m = sc->m;
        {   // ---- LDS plan (engine.h LdsPlan): sized for the widest expected band / largest query of this pass
            int max_qlen = 0, max_bits = 16; int64_t est_cols = 0;
            for (const AlnDesc &d : pass) {
                max_qlen = std::max(max_qlen, d.qlen); max_bits = std::max(max_bits, d.bits);
                const int pn = d.bits == 16 ? 16 : 8; const int64_t width = (int64_t)((d.qlen + pn) / pn) * pn;
                est_cols = std::max<int64_t>(est_cols, banded ? std::min<int64_t>(width, 2LL * d.w + 3 * pn + 32) : width);
            }
            LdsPlan &L = b.lds; const int cell = max_bits / 8, npr = P == 1 ? 1 : (P == 3 ? 2 : 3);
            L.q_off = 0; L.q_cap = max_qlen + 1 <= 16384 ? (int)align_up(max_qlen + 1, 16) : 0;
            L.mat_off = L.q_cap; L.phase_off = L.mat_off + (int)align_up(4 * sc->m * sc->m, 16);
            L.ring_off = lds_fixed_bytes_dp(); L.ring_rows = 16; L.ring_cols = (int)align_up((size_t)est_cols, 64);
            while ((int64_t)L.ring_rows * npr * L.ring_cols * cell > 40 * 1024 && L.ring_rows > 4) L.ring_rows /= 2;
            if ((int64_t)L.ring_rows * npr * L.ring_cols * cell > 40 * 1024) L.ring_cols = 0;     // rows too wide: HBM path only
            const int ring_bytes = L.ring_rows * npr * L.ring_cols * cell;
            L.bt_off = lds_fixed_bytes_bt();
            L.bt_bytes = std::max(16 * 1024, L.ring_off + ring_bytes - L.bt_off) & ~15;
            L.total = L.phase_off + std::max(L.ring_off + ring_bytes, L.bt_off + L.bt_bytes);
        }
        b.o1 = sc->gap_open1; b.e1 = sc->gap_ext1; b.o2 = sc->gap_open2; b.e2 = sc->gap_ext2;
        b.align_mode = sc->align_mode; b.gap_mode = sc->gap_mode; b.wb = sc->wb; b.zdrop = sc->zdrop; b.ret_cigar = sc->ret_cigar; b.rev_cigar = sc->rev_cigar; b.want_trace = trace ? 1 : 0;
        uint8_t *di = g.in.dev, *dout = g.outb.dev;
        b.mat = (const int32_t *)(di + o_mat); b.aln = (const AlnDesc *)(di + o_desc); b.out = (AlnOut *)(dout + o_rec);
        b.query = di + o_query; b.row_base = di + o_base; b.row_node_id = (const int32_t *)(di + o_nid); b.row_remain = (const int32_t *)(di + o_rem);
        b.row_active = di + o_act; b.pred_off = (const int32_t *)(di + o_poff); b.pred_row = (const int32_t *)(di + o_pred);
        b.out_off = (const int32_t *)(di + o_ooff); b.out_row = (const int32_t *)(di + o_out);
        b.left = (int32_t *)(dout + o_left); b.right = (int32_t *)(dout + o_right);
        b.dp_beg_sn = (int32_t *)(dout + o_bsn); b.dp_end_sn = (int32_t *)(dout + o_esn); b.row_cell_off = (int64_t *)(dout + o_coff);
        b.row_max_i = (int32_t *)(dout + o_rmi); b.planes = g.planes.dev; b.cigar = (uint64_t *)(dout + o_cig);

        HIP_TRY(hipEventRecord(g.ev[0], g.stream), ABPOA_HIP_ELAUNCH);
        HIP_TRY(hipMemcpyAsync(di, hi, first_pass ? in_bytes : o_mat, hipMemcpyHostToDevice, g.stream), ABPOA_HIP_ELAUNCH);
        if (banded)
            HIP_TRY(hipMemcpyAsync(dout + o_left, ho + o_left, (o_right - o_left) + 4 * rows_tot, hipMemcpyHostToDevice, g.stream), ABPOA_HIP_ELAUNCH);
        // -1 = "row never computed" (inactive rows, rows behind a z-drop break)
        HIP_TRY(hipMemsetAsync(dout + o_bsn, 0xFF, (o_esn - o_bsn) + 4 * rows_tot, g.stream), ABPOA_HIP_ELAUNCH);
        if (trace) HIP_TRY(hipMemsetAsync(dout + o_rmi, 0xFE, 4 * rows_tot, g.stream), ABPOA_HIP_ELAUNCH);
        HIP_TRY(hipEventRecord(g.ev[1], g.stream), ABPOA_HIP_ELAUNCH);
        HIP_TRY(launch_dp(b, g.stream), ABPOA_HIP_ELAUNCH);
        HIP_TRY(hipEventRecord(g.ev[2], g.stream), ABPOA_HIP_ELAUNCH);
        std::vector<AlnOut> recs(pass.size());
        HIP_TRY(hipMemcpyAsync(ho + o_rec, dout + o_rec, sizeof(AlnOut) * pass.size(), hipMemcpyDeviceToHost, g.stream), ABPOA_HIP_ELAUNCH);
        HIP_TRY(hipStreamSynchronize(g.stream), ABPOA_HIP_ELAUNCH);
        memcpy(recs.data(), ho + o_rec, sizeof(AlnOut) * pass.size());
        // everything else the host needs: band state, per-row outputs, cigars
        HIP_TRY(hipMemcpyAsync(ho + o_left, dout + o_left, out_bytes - o_left, hipMemcpyDeviceToHost, g.stream), ABPOA_HIP_ELAUNCH);
        HIP_TRY(hipEventRecord(g.ev[3], g.stream), ABPOA_HIP_ELAUNCH);
        HIP_TRY(hipStreamSynchronize(g.stream), ABPOA_HIP_ELAUNCH);
        float ms_h2d = 0, ms_k = 0, ms_d2h = 0;
        (void)hipEventElapsedTime(&ms_h2d, g.ev[0], g.ev[1]); (void)hipEventElapsedTime(&ms_k, g.ev[1], g.ev[2]); (void)hipEventElapsedTime(&ms_d2h, g.ev[2], g.ev[3]);
        g.stats.n_launches += 1; g.stats.kernel_ms += ms_k; g.stats.h2d_ms += ms_h2d; g.stats.d2h_ms += ms_d2h;

        std::vector<int> again;
        for (size_t t = 0; t < todo.size(); ++t) {
            const int i = todo[t]; const AlnOut &r = recs[t]; const AlnDesc &d = desc[i]; const abpoa_hip_problem_t &p = pb[i];
            if (r.status == ABPOA_HIP_STATUS_OVERFLOW) {
                if (d.plane_cap >= full_cells[i]) { set_err("problem %d: arena overflow at full width (internal error)", i); return ABPOA_HIP_ELAUNCH; }
                again.push_back(i); continue;
            }
            abpoa_hip_result_t &R = res[i];
            R.status = r.status; R.bits = d.bits; R.best_score = r.best_score; R.best_row = r.best_row; R.best_col = r.best_col;
            R.node_s = r.node_s; R.node_e = r.node_e; R.query_s = r.query_s; R.query_e = r.query_e;
            R.n_aln_bases = r.n_aln_bases; R.n_matched_bases = r.n_matched_bases; R.n_cells = r.n_cells;
            R.n_cigar = r.status == 0 ? r.n_cigar : 0;
            if (R.n_cigar > 0) {
                R.cigar = (uint64_t *)malloc(sizeof(uint64_t) * R.n_cigar);
                if (!R.cigar) { set_err("malloc failed"); return ABPOA_HIP_ENOMEM; }
                memcpy(R.cigar, ho + o_cig + 8 * d.cigar_off, sizeof(uint64_t) * R.n_cigar);
            }
            if (banded) { memcpy(p.max_pos_left, ho + o_left + 4 * d.row0, 4 * p.n_rows); memcpy(p.max_pos_right, ho + o_right + 4 * d.row0, 4 * p.n_rows); }
            const int pn = d.bits == 16 ? 16 : 8;
            g.stats.n_alignments += 1; g.stats.n_cells += r.n_cells;
            g_dbg[0] += r.clk_dp; g_dbg[1] += r.clk_bt; g_dbg[2] += r.n_rows_done; g_dbg[3] += r.n_bt_steps; for (int q_ = 0; q_ < 6; ++q_) g_dbg[4 + q_] += r.seg[q_];
            g.stats.algo_bytes += r.n_cells * (d.bits / 8) * (P == 1 ? 2 : (P == 3 ? 5 : 8));
            if (trace) {
                abpoa_hip_trace_t *T = (abpoa_hip_trace_t *)calloc(1, sizeof(abpoa_hip_trace_t));
                const int gn = p.n_rows;
                T->bits = d.bits; T->n_planes = P;
                T->dp_beg = (int32_t *)malloc(4 * gn); T->dp_end = (int32_t *)malloc(4 * gn); T->dp_beg_sn = (int32_t *)malloc(4 * gn); T->dp_end_sn = (int32_t *)malloc(4 * gn);
                T->row_off = (int64_t *)malloc(8 * (gn + 1)); T->row_max_i = (int32_t *)malloc(4 * gn);
                const int32_t *bsn = (const int32_t *)(ho + o_bsn) + d.row0, *esn = (const int32_t *)(ho + o_esn) + d.row0;
                const int64_t *coff = (const int64_t *)(ho + o_coff) + d.row0;
                memcpy(T->row_max_i, (const int32_t *)(ho + o_rmi) + d.row0, 4 * gn);
                // host copy of the arena
                std::vector<uint8_t> arena((size_t)r.cells_used * (d.bits / 8));
                if (!arena.empty()) HIP_TRY(hipMemcpy(arena.data(), g.planes.dev + d.plane_off, arena.size(), hipMemcpyDeviceToHost), ABPOA_HIP_ELAUNCH);
                int64_t tot = 0;
                for (int rr = 0; rr < gn; ++rr) {
                    T->row_off[rr] = tot;
                    const bool computed = rr < gn - 1 && bsn[rr] >= 0;
                    if (!computed) { T->dp_beg[rr] = T->dp_end[rr] = T->dp_beg_sn[rr] = T->dp_end_sn[rr] = -1; continue; }
                    T->dp_beg_sn[rr] = bsn[rr]; T->dp_end_sn[rr] = esn[rr]; T->dp_beg[rr] = bsn[rr] * pn;
                    T->dp_end[rr] = (banded || rr == 0) ? (esn[rr] + 1) * pn - 1 : p.qlen;
                    tot += (int64_t)(esn[rr] - bsn[rr] + 1) * pn * P;
                }
                T->row_off[gn] = tot;
                T->planes = malloc((size_t)std::max<int64_t>(tot, 1) * (d.bits / 8));
                for (int rr = 0; rr < gn; ++rr) {
                    if (T->dp_beg_sn[rr] < 0) continue;
                    size_t nb = (size_t)(T->row_off[rr + 1] - T->row_off[rr]) * (d.bits / 8);
                    memcpy((uint8_t *)T->planes + T->row_off[rr] * (d.bits / 8), arena.data() + coff[rr] * (d.bits / 8), nb);
                }
                R.trace = T;
            }
        }
        todo.swap(again); first_pass = false;   // retried alignments keep their slots in the row pools
    }
    return ABPOA_HIP_OK;
}

}  // extern "C"
