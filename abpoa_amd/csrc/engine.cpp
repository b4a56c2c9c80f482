// Host side of the engine: device binding, BatchStream (pinned staging + HBM pools + launch) and the flat
// batch API of include/abpoa_hip.h on top of it.  Compiled with hipcc.
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
#include "engine_options.h"
#include "batch_stream.h"
#include "dir_plane.h"

namespace abpoa_hip {

// Last error of the PROCESS (the batch calls run on worker threads; the caller reads the message from its own thread).  Readers get a
// per-thread copy taken under the lock.
static char g_err[512] = ""; static std::mutex g_err_mu;
static thread_local char g_err_copy[512] = "";
static thread_local char tl_err[512] = "";      // last error raised on THIS thread (a context's call copies it: abpoa_hip_ctx_last_error)
void set_err(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(tl_err, sizeof(tl_err), fmt, ap); va_end(ap);
    std::lock_guard<std::mutex> lk(g_err_mu);
    memcpy(g_err, tl_err, sizeof(g_err));
}
const char *thread_last_error() { return tl_err; }
void clear_thread_error() { tl_err[0] = 0; }
#define HIP_TRY(expr, code)                                                                          \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) {                                             \
        set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return code; } } while (0)

int Blob::reserve(size_t n) {
    if (n <= cap) return 0;
    size_t want = std::max(n, cap + cap / 2);
    want = (want + 0xFFFFF) & ~(size_t)0xFFFFF;
    release();
    if (hipMalloc((void **)&dev, want) != hipSuccess) { dev = nullptr; set_err("hipMalloc(%zu) failed", want); return ABPOA_HIP_ENOMEM; }
    if (mirrored && hipHostMalloc((void **)&host, want, hipHostMallocDefault) != hipSuccess) {
        host = nullptr; set_err("hipHostMalloc(%zu) failed", want); return ABPOA_HIP_ENOMEM; }
    cap = want; return 0;
}
void Blob::release() { if (dev) (void)hipFree(dev); if (host) (void)hipHostFree(host); dev = host = nullptr; cap = 0; }

struct Engine {
    bool ready = false; int device = -1;
    BatchStream flat;                  // default stream of the flat C API
    abpoa_hip_stats_t stats{};
    std::mutex mu, stats_mu;
};
static Engine g;
long long g_dbg[10] = {0};
long long g_dir_counts[2] = {0, 0};      // alignments whose backtrack walked a direction plane; alignments redone with score records (NEED_SCORES)

int engine_device() { return g.ready ? g.device : -1; }
void add_global_stats(const StreamStats &s) {
    std::lock_guard<std::mutex> lk(g.stats_mu);
    g.stats.n_launches += s.n_launches; g.stats.n_alignments += s.n_alignments; g.stats.n_cells += s.n_cells;
    g.stats.algo_bytes += s.algo_bytes; g.stats.kernel_ms += s.kernel_ms; g.stats.h2d_ms += s.h2d_ms; g.stats.d2h_ms += s.d2h_ms; g.stats.tail_ms += s.tail_ms; g.stats.rounds_ms += s.rounds_ms;
    g.stats.rounds_launches += s.rounds_launches; g.stats.rounds_algo_bytes += s.rounds_algo_bytes;
}

static size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ------------------------------------------------------------------------------------------------ BatchStream
int BatchStream::open(int device) {
    if (open_) return 0;
    HIP_TRY(hipSetDevice(device), ABPOA_HIP_ENODEV);
    HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), ABPOA_HIP_ENODEV);
    for (auto &e : ev_) HIP_TRY(hipEventCreate(&e), ABPOA_HIP_ENODEV);
    device_ = device; open_ = true; return 0;
}
void BatchStream::close() {
    if (!open_) return;
    (void)hipSetDevice(device_);
    (void)hipStreamSynchronize(stream_);
    in_.release(); out_.release(); planes_.release();
    for (auto &e : ev_) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(stream_);
    open_ = false;
}

// LDS carve-up of one wavefront (engine.h LdsPlan) for a launch whose largest query is max_qlen, widest score type max_bits
// and widest expected band est_cols columns.
// (-DABPOA_HIP_WIDE_W3, experiment: three wavefronts per SIMD in the wide loop -- twelve workgroups per CU, ring depth down to 2)
#ifdef ABPOA_HIP_WIDE_W3
constexpr int WIDE_PER_CU_MAX = 12, WIDE_RING_MIN = 2;
#else
constexpr int WIDE_PER_CU_MAX = 8, WIDE_RING_MIN = 4;
#endif
void make_lds_plan(const abpoa_hip_scoring_t *sc, int max_qlen, int max_bits, int64_t est_cols, int n_aln, LdsPlan *Lp) {
    LdsPlan &L = *Lp;
    const int P = sc->gap_mode == ABPOA_HIP_LINEAR_GAP ? 1 : (sc->gap_mode == ABPOA_HIP_AFFINE_GAP ? 3 : 5);
    const int cell = max_bits / 8, npr = P == 1 ? 1 : (P == 3 ? 2 : 3);
    // query codes in LDS: up to 32000 bases keep the fast row loops (their per-row registers hold band vectors in 12 bits: 4095 x 8 columns);
    // longer reads take the general kernel.  Above 16 K bases the kernels may use up to 62 KB of LDS per wavefront instead of 36 - 38 KB.
    const bool longq = max_qlen + 1 > 16384;
    L.q_off = 0; L.q_cap = max_qlen + 1 <= 32000 ? (int)align_up(max_qlen + 1, 16) : 0;
    L.mat_off = L.q_cap; L.mx_off = L.mat_off + (int)align_up(4 * sc->m * sc->m, 16);
    L.phase_off = L.mx_off + (int)align_up(4 * sc->m * (sc->m + 1), 16);
    L.ring_off = lds_fixed_bytes_dp(); L.ring_rows = 16; L.ring_cols = (int)align_up((size_t)est_cols, 64);
    while ((int64_t)L.ring_rows * npr * L.ring_cols * cell > 40 * 1024 && L.ring_rows > 4) L.ring_rows /= 2;
    if ((int64_t)L.ring_rows * npr * L.ring_cols * cell > 40 * 1024) L.ring_cols = 0;     // rows too wide: HBM path only
    const int ring_bytes = L.ring_rows * npr * L.ring_cols * cell;
    L.bt_off = lds_fixed_bytes_bt();
    // staged arena window of the backtrack: 24 KB, less when a long query already takes much of the 40 KB a workgroup may use
    L.bt_bytes = std::max(std::max(8 * 1024, std::min(24 * 1024, (longq ? 62 : 38) * 1024 - L.phase_off - L.bt_off)), L.ring_off + ring_bytes - L.bt_off) & ~15;
    // fast row loop: packed score ring (words per cell: linear 1, int16 affine 1, int16 convex 2, int32 affine 2, int32 convex 3)
    const int fw = P == 1 ? 1 : (max_bits == 16 ? (P == 3 ? 1 : 2) : (P == 3 ? 2 : 3));      // (linear gaps: H alone)
    L.fr_off = 0; L.fr_rows = 16; L.fr_cols = fw ? std::max(128, (int)align_up((size_t)est_cols, 64)) : 0;      // >= 128: the turbo row pads one chunk unconditionally
    const int fr_budget = (longq ? 62 : 36) * 1024 - L.phase_off;
    while (L.fr_cols && (int64_t)L.fr_rows * fw * (L.fr_cols + 4) * 4 + 64 > fr_budget && L.fr_rows > 4) L.fr_rows /= 2;
    if (L.fr_cols && (int64_t)L.fr_rows * fw * (L.fr_cols + 4) * 4 + 64 > fr_budget) L.fr_cols = 0;
    if (L.q_cap == 0 || est_cols > 1024) L.fr_cols = 0;
    { const char *nf_ = opt_env("ABPOA_HIP_NOFAST"); if (nf_ && atoi(nf_)) L.fr_cols = 0; }
    const int fr_bytes = L.fr_cols ? L.fr_rows * fw * (L.fr_cols + 4) * 4 + 64 : 0;
    L.total = L.phase_off + std::max(std::max(L.ring_off + ring_bytes, L.bt_off + L.bt_bytes), L.fr_off + fr_bytes);
    // the fast path's tail kernel: its own window size -- 28 KB, less when a long query already takes much of the 38 (62) KB that let four (two) of
    // its workgroups share a CU; the general kernel's bt_bytes above also covers its score ring and would halve that residency
    L.bt_bytes_tail = std::max(8 * 1024, std::min(28 * 1024, (longq ? 62 : 38) * 1024 - L.phase_off - L.bt_off)) & ~15;
    // a launch with fewer alignments than 4 per CU can afford a larger window per workgroup (fewer window reloads on wide bands): 160 KB / CU
    // divided by the workgroups a CU has to hold, capped at 56 KB
    { const int per_cu = std::max(1, (n_aln + 255) / 256);
      if (per_cu < 4 && L.fr_cols > 128) L.bt_bytes_tail = std::max(L.bt_bytes_tail, std::min(56 * 1024, 160 * 1024 / per_cu - 2048 - L.phase_off - L.bt_off) & ~15); }
    // more alignments than a GPU holds tail workgroups at the 24 KB window (4 per CU): a 12 KB window doubles the residency, and the tail kernel of such
    // a launch runs in as many turns as it has workgroups per resident set (8000 x 1 kb alignments: tail 153 -> 127 ms per step)
    if (n_aln > 4 * 256 && L.fr_cols && L.fr_cols <= 128) L.bt_bytes_tail = std::min(L.bt_bytes_tail, 12 * 1024);
    { const char *tb_ = opt_env("ABPOA_HIP_BT_BYTES"); if (tb_ && atoi(tb_) >= 4096 && atoi(tb_) <= 65536) L.bt_bytes_tail = atoi(tb_) & ~15; }
    L.total_rows = L.phase_off + L.fr_off + fr_bytes; L.total_tail = L.phase_off + L.bt_off + L.bt_bytes_tail;
    L.bt_wc = 0; { const char *wc_ = opt_env("ABPOA_HIP_BT_WC"); if (wc_ && atoi(wc_) >= 8 && atoi(wc_) <= 64) L.bt_wc = atoi(wc_) & ~7; }
    // local row loop (rows_local.h): unbanded local alignments of at most 9 x 64 columns, int16; ring depth by what 60 KB hold
    L.loc_rows = L.loc_cols = L.total_local = 0;
    if (sc->align_mode == ABPOA_HIP_LOCAL_MODE && sc->wb < 0 && P != 1 && L.q_cap && !(opt_env("ABPOA_HIP_NOFAST") && atoi(opt_env("ABPOA_HIP_NOFAST")))) {
        const int lw = P == 3 ? 1 : 2;                 // ring words per column (int16: H | E1 packed, E2)
        L.loc_cols = 9 * 64; L.loc_rows = 16;
        while ((int64_t)L.loc_rows * lw * (L.loc_cols + 4) * 4 > 60 * 1024 - L.phase_off && L.loc_rows > 4) L.loc_rows /= 2;
        L.total_local = L.phase_off + L.fr_off + (int)align_up((size_t)L.loc_rows * lw * (L.loc_cols + 4) * 4, 16) + 256;      // (+ the team kernel's exchange slots: 2 x 4 x 16 bytes)
    }
    // wide row loop (dp_wide_rows.hip): alignments whose band half-width w is in [wide_w_lo, wide_w_hi] -- rows of 2..7 chunks of 64 columns --
    // go to the kernel that keeps every chunk of a row in registers; it has its own score ring (448 columns; depth by what fits:
    // predecessors up to 15 rows back are common in a graph of noisy reads).  ABPOA_HIP_NOWIDE=1 turns it off, ABPOA_HIP_RING_ROWS sets
    // the depth, ABPOA_HIP_TEAM=1|2|4 sets the wavefronts per alignment.
    L.wide_nw = 0; L.wfr_rows = L.wfr_cols = L.wx_off = L.total_wide = 0; L.wide_w_lo = 1; L.wide_w_hi = 0; L.narrow_off = 0; L.w_mx_off = L.w_phase_off = 0;
    { const char *nw_ = opt_env("ABPOA_HIP_NOWIDE"), *mw_ = opt_env("ABPOA_HIP_TEAM");
      if (L.fr_cols && L.q_cap && !(nw_ && atoi(nw_))) {
          // wavefronts per alignment: 1.  Teams of 2 / 4 (ABPOA_HIP_TEAM=2|4, dp_team_rows.hip) give identical results but are slower on gfx950
          // as measured (3.9 k vs 3.1 k cycles per 5-chunk row): a row's ~370 instructions of scalar bookkeeping are repeated by every wavefront
          // of the team and outweigh the ~28 instructions per chunk that the split saves (profiles/r2_team_vs_single.txt).
          L.wide_nw = 1;
          if (mw_ && (atoi(mw_) == 1 || atoi(mw_) == 2 || atoi(mw_) == 4)) L.wide_nw = atoi(mw_);
          // (rows wider than the 448-column ring -- reads of 20 kb and more: w = 10 + 0.01 L -- take the kernel's long-read form: 704 columns, 8 - 11 chunks a row)
          L.wfr_cols = (est_cols > WIDE_RING_COLS && !(opt_env("ABPOA_HIP_NOXL") && atoi(opt_env("ABPOA_HIP_NOXL")))) ? WIDE_RING_COLS_XL : WIDE_RING_COLS; L.wfr_rows = 16;
          if (sc->m > 16) L.wide_nw = 0;      // (4-bit query codes)
          L.w_mx_off = (int)align_up((size_t)(max_qlen + 2) / 2, 16); L.w_phase_off = L.w_mx_off + (int)align_up(4 * sc->m * (sc->m + 1), 16);
          // ring words per column of the wide kernels: as the narrow loop's, but two instead of three for convex int32 (rows_fast.h EPACK: E as 16-bit
          // differences to H, which needs gap-open + extend <= 65535)
          const int fww = (P == 5 && max_bits == 32) ? 2 : fw;
          if (P == 5 && (sc->gap_open1 + sc->gap_ext1 >= 65535 || sc->gap_open2 + sc->gap_ext2 >= 65535)) L.wide_nw = 0;      // (0xffff: "E is inf" in the compact spill records)
          L.wide_w_lo = 40; L.wide_w_hi = (L.wfr_cols - 2 * 8 - 1) / 2;
          { const char *lo_ = opt_env("ABPOA_HIP_WIDE_LO"); if (lo_ && atoi(lo_) > 0) L.wide_w_lo = atoi(lo_); }
          const char *rr_env_ = opt_env("ABPOA_HIP_RING_ROWS");
          if (rr_env_ && atoi(rr_env_) >= 4) L.wfr_rows = atoi(rr_env_) >= 16 ? 16 : (atoi(rr_env_) >= 8 ? 8 : 4); else rr_env_ = nullptr;
          // (up to 120 KB per wavefront: a convex int32 ring of 16 rows is 58 KB; above 64 KB the launch raises the kernel's dynamic-LDS limit)
          const int budget = 120 * 1024 - L.w_phase_off - 512;
          while ((int64_t)L.wfr_rows * fww * (L.wfr_cols + 4) * 4 > budget && L.wfr_rows > 4) L.wfr_rows /= 2;
          // One wavefront per alignment: LDS is what limits how many alignments a CU holds.  It is handed out in pieces of 1280 B, 128 per CU
          // (tools/probes/lds_granule.hip: 3 x 53760 B fit a CU, 3 x 54080 B do not, whatever the occupancy query says).  The deepest ring with which
          // the whole launch is resident, counting at most eight workgroups per CU -- two wavefronts per SIMD, which is what the registers allow and
          // what pays: a SIMD with two alignments to issue from does 1.6x the rows of one with a single wavefront (tools/two_waves_probe.py).  A
          // shallower ring sends more rows to the HBM gather (predecessor older than the ring: 0.5 % / 14 % / ~45 % of the rows of a 15 %-error
          // graph at depth 16 / 8 / 4; rows +1.6 % / +6.5 %).
          const int extra_ = L.wide_nw > 1 ? 16 * 16 + 64 : 0;      // (exchange slots: teams only)
          auto per_cu_ = [&](int rows_) { return std::min<int64_t>(WIDE_PER_CU_MAX, 128 / ((L.w_phase_off + (int64_t)rows_ * fww * (L.wfr_cols + 4) * 4 + extra_ + 1279) / 1280)); };
          if (!(rr_env_)) {
              const int top_ = L.wfr_rows; int best_ = top_;
              for (int r_ = top_; r_ >= WIDE_RING_MIN; r_ /= 2) {
                  if (per_cu_(r_) > per_cu_(best_)) best_ = r_;
                  if (per_cu_(r_) * 256 >= std::min(n_aln, WIDE_PER_CU_MAX * 256)) { best_ = r_; break; }
              }
              L.wfr_rows = best_;
          }
          L.wx_off = L.fr_off + (int)align_up((size_t)L.wfr_rows * fww * (L.wfr_cols + 4) * 4, 16);
          L.total_wide = L.w_phase_off + L.wx_off + extra_;
          if (P == 1) L.wide_nw = 0;      // (linear gaps: the narrow loop only -- dp_common.h takes_fast)
      } }
}

int BatchStream::prepare(const abpoa_hip_scoring_t *sc, int n, const BatchShape *sh, unsigned flags) {
    if (!open_) { set_err("stream not open"); return ABPOA_HIP_ENODEV; }
    HIP_TRY(hipSetDevice(device_), ABPOA_HIP_ENODEV);
    sc_ = *sc; mat_.assign(sc->mat, sc->mat + sc->m * sc->m); sc_.mat = mat_.data();
    flags_ = flags; n_ = n;
    P_ = sc->gap_mode == ABPOA_HIP_LINEAR_GAP ? 1 : (sc->gap_mode == ABPOA_HIP_AFFINE_GAP ? 3 : 5);
    const bool banded = sc->wb >= 0;
    desc_.resize(n); recs_.resize(n); full_cells_.resize(n); est_cells_.resize(n); dir_full_cells_.resize(n); dir_est_cells_.resize(n); trace_arena_.clear();
    rows_tot_ = preds_tot_ = outs_tot_ = q_tot_ = cig_tot_ = 0;
    for (int i = 0; i < n; ++i) {
        AlnDesc &d = desc_[i];
        d.n_rows = sh[i].n_rows; d.qlen = sh[i].qlen;
        d.bits = abpoa_hip_score_bits(sc, d.n_rows, d.qlen, &d.inf_min);
        d.w = sc->wb < 0 ? d.qlen : sc->wb + (int)(sc->wf * d.qlen);     // reference :445 (float32 product)
        d.cigar_cap = d.n_rows + d.qlen + 8;
        d.query_off = q_tot_; d.row0 = rows_tot_; d.poff0 = rows_tot_ + i; d.pred0 = preds_tot_; d.out0 = outs_tot_; d.cigar_off = cig_tot_;
        q_tot_ += d.qlen; rows_tot_ += d.n_rows; preds_tot_ += sh[i].n_pred; outs_tot_ += sh[i].n_out; cig_tot_ += d.cigar_cap;
        const int pn = d.bits == 16 ? 16 : 8;
        const int64_t width = (int64_t)((d.qlen + pn) / pn) * pn;
        // values per DP column in the arena: P planes (general kernel) or one padded cell record (fast loop: 4 / 8 values)
        // (linear gaps: one plane in the general kernel, {H, match flag} in the fast loops)
        const int pv = sc->gap_mode == ABPOA_HIP_LINEAR_GAP ? 2 : (sc->gap_mode == ABPOA_HIP_AFFINE_GAP ? 4 : 8);
        full_cells_[i] = width * pv * d.n_rows;
        int64_t est = banded ? std::min<int64_t>(width, 2LL * d.w + 3 * pn + 32) : width;
        // (tests: force the overflow -> full-width retry path)
        { const char *pct_ = opt_env("ABPOA_HIP_ARENA_PCT"); if (pct_ && atoi(pct_) > 0 && atoi(pct_) < 100) est = std::max<int64_t>(pn, est * atoi(pct_) / 100); }
        d.plane_cap = std::min<int64_t>(full_cells_[i], width * pv + est * pv * (d.n_rows - 1));
        est_cells_[i] = d.plane_cap;
        // direction-plane arenas (dir_plane.h; run() decides whether a pass uses them): words of DB bytes per column for every row, score records for the
        // first row and about one row in four (cells of the score width: DB bytes = DB * 8 / bits cells); full = every row at full width with its records
        const int64_t dbc = (sc->gap_mode == ABPOA_HIP_AFFINE_GAP ? 2 : 4) * 8 / d.bits + 1;
        dir_full_cells_[i] = width * (pv + dbc) * d.n_rows + 64;
        dir_est_cells_[i] = std::min<int64_t>(dir_full_cells_[i], width * (pv + dbc) + (est * dbc + est * pv / 4 + 2 * pn) * (d.n_rows - 1));
    }
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes); return at; };
    o_desc_ = take(sizeof(AlnDesc) * n); o_mat_ = take(sizeof(int32_t) * sc->m * sc->m); o_query_ = take(q_tot_ + 1);
    o_base_ = take(rows_tot_); o_sdist_ = take(rows_tot_); o_pd_ = take(8 * rows_tot_); o_nid_ = take(4 * rows_tot_); o_rem_ = take(4 * rows_tot_); o_act_ = take(rows_tot_);
    // slack: tile prefetch over-reads up to TP entries
    o_poff_ = take(4 * (rows_tot_ + n)); o_pred_ = take(4 * (preds_tot_ + 1)); o_ooff_ = take(4 * (rows_tot_ + n)); o_out_ = take(4 * (outs_tot_ + 1) + 4 * 512);
    in_bytes_ = o;
    o = 0;
    o_rec_ = take(sizeof(AlnOut) * n); o_cig_ = take(8 * cig_tot_); o_left_ = take(4 * rows_tot_); o_right_ = take(4 * rows_tot_);
    o_bsn_ = take(4 * rows_tot_); o_esn_ = take(4 * rows_tot_); o_coff_ = take(8 * rows_tot_); o_rmi_ = take(4 * rows_tot_);
    out_bytes_ = o;
    int rc;
    if ((rc = in_.reserve(in_bytes_)) || (rc = out_.reserve(out_bytes_))) return rc;
    memcpy(in_.host + o_mat_, sc->mat, sizeof(int32_t) * sc->m * sc->m);
    memset(in_.host + o_act_, 1, rows_tot_);
    return 0;
}

ProblemSlots BatchStream::slots(int i) const {
    const AlnDesc &d = desc_[i]; uint8_t *hi = in_.host, *ho = out_.host; ProblemSlots s;
    s.query = hi + o_query_ + d.query_off; s.row_base = hi + o_base_ + d.row0;
    s.row_node_id = (int32_t *)(hi + o_nid_) + d.row0; s.row_remain = (int32_t *)(hi + o_rem_) + d.row0; s.row_active = hi + o_act_ + d.row0;
    s.pred_off = (int32_t *)(hi + o_poff_) + d.poff0; s.pred_row = (int32_t *)(hi + o_pred_) + d.pred0;
    s.out_off = (int32_t *)(hi + o_ooff_) + d.poff0; s.out_row = (int32_t *)(hi + o_out_) + d.out0;
    s.left = (int32_t *)(ho + o_left_) + d.row0; s.right = (int32_t *)(ho + o_right_) + d.row0;
    return s;
}
const uint64_t *BatchStream::cigar(int i) const { return (const uint64_t *)(out_.host + o_cig_) + desc_[i].cigar_off; }
const int32_t *BatchStream::left(int i) const { return (const int32_t *)(out_.host + o_left_) + desc_[i].row0; }
const int32_t *BatchStream::right(int i) const { return (const int32_t *)(out_.host + o_right_) + desc_[i].row0; }

int BatchStream::run() {
    HIP_TRY(hipSetDevice(device_), ABPOA_HIP_ENODEV);
    const abpoa_hip_scoring_t *sc = &sc_;
    const bool banded = sc->wb >= 0, trace = flags_ & BS_TRACE, fresh = flags_ & BS_FRESH_BAND;
    const int P = P_, n = n_;
    uint8_t *hi = in_.host, *ho = out_.host, *di = in_.dev, *dout = out_.dev;
    std::vector<int> todo(n); for (int i = 0; i < n; ++i) todo[i] = i;
    const size_t lr_words = (o_right_ - o_left_) / 4 + rows_tot_;
    std::vector<int32_t> saved_lr;              // caller's band state, needed again if an alignment is retried
    bool first_pass = true; int rc;
    std::vector<AlnDesc> pass;
    // direction-plane arenas for the fast row loops (dir_plane.h) unless the caller wants the score planes back (trace), the penalties do not fit the
    // words, or ABPOA_HIP_NODIR=1; an alignment whose backtrack meets the one case the words cannot decide is redone with score records
    // (tests: ABPOA_HIP_DIRTRACE=1 keeps the direction plane in trace mode; the trace then carries the WORDS of every cell in plane 0)
    const bool dirtrace = trace && opt_env("ABPOA_HIP_DIRTRACE") && atoi(opt_env("ABPOA_HIP_DIRTRACE"));
    bool dir = (!trace || dirtrace) && banded && sc->align_mode == ABPOA_HIP_GLOBAL_MODE && dir_plane_usable(sc->gap_mode, sc->gap_open1, sc->gap_ext1, sc->gap_open2, sc->gap_ext2) &&
               !(opt_env("ABPOA_HIP_NODIR") && atoi(opt_env("ABPOA_HIP_NODIR"))) && !(opt_env("ABPOA_HIP_TEAM") && atoi(opt_env("ABPOA_HIP_TEAM")) > 1);
    // alignment-level eligibility for the register-resident row loop: every row active, band state at its reset value
    for (int i = 0; i < n; ++i) {
        AlnDesc &d = desc_[i]; bool ok = banded || (sc->align_mode == ABPOA_HIP_LOCAL_MODE && sc->wb < 0);      // (unbanded local: the local row loop; band state is not used)
        if (ok && memchr(hi + o_act_ + d.row0, 0, (size_t)d.n_rows)) ok = false;
        if (ok && banded && !fresh) {
            const int32_t *l = (const int32_t *)(ho + o_left_) + d.row0, *r = (const int32_t *)(ho + o_right_) + d.row0;
            for (int k = 0; k < d.n_rows && ok; ++k) ok = l[k] == d.n_rows && r[k] == 0;
        }
        d.flags = ok ? ALN_FAST_OK : 0; d.pad0 = 0;
    }
    if (dir) {      // how far down the row order every row's scores are still needed (DevBatch.row_sdist); a row with more predecessors than a word can name keeps the alignment off the fast loops
        for (int i = 0; i < n; ++i) {
            AlnDesc &d = desc_[i];
            if (!(d.flags & ALN_FAST_OK)) continue;
            const int32_t *po = (const int32_t *)(hi + o_poff_) + d.poff0, *pr = (const int32_t *)(hi + o_pred_) + d.pred0; uint8_t *sd = hi + o_sdist_ + d.row0;
            uint32_t *pdw = (uint32_t *)(hi + o_pd_) + 2 * d.row0;      // row distances to the first eight predecessors, two dwords per row (DevBatch.row_pd)
            memset(sd, 0, (size_t)d.n_rows);
            bool ok = true;
            for (int r = 0; r < d.n_rows && ok; ++r) {
                const int np_ = po[r + 1] - po[r];
                if (np_ > DIR_K_MAX && r < d.n_rows - 1) ok = false;
                uint64_t pdv = ~0ull;       // a byte per predecessor, 255 = none / too far
                for (int k = po[r]; k < po[r + 1]; ++k) {
                    const int p_ = pr[k], dist = r == d.n_rows - 1 ? 255 : std::min(255, r - p_); if (dist > sd[p_]) sd[p_] = (uint8_t)dist;
                    const int t = k - po[r]; if (t < 8 && r - p_ <= 254) pdv = (pdv & ~(0xffull << (8 * t))) | ((uint64_t)(r - p_) << (8 * t));
                }
                (void)np_;
                pdw[2 * r] = (uint32_t)pdv; pdw[2 * r + 1] = (uint32_t)(pdv >> 32);
            }
            if (!ok) d.flags = 0;
        }
    }
    while (!todo.empty()) {
        int64_t plane_bytes = 0;
        pass.resize(todo.size());
        for (size_t t = 0; t < todo.size(); ++t) pass[t] = desc_[todo[t]];
        DevBatch b; memset(&b, 0, sizeof(b));
        b.n = (int)pass.size(); b.m = sc->m;
        {   // ---- LDS plan (engine.h LdsPlan): sized for the widest expected band / largest query of this pass
            int max_qlen = 0, max_bits = 16; int64_t est_cols = 0;
            for (const AlnDesc &d : pass) {
                max_qlen = std::max(max_qlen, d.qlen); max_bits = std::max(max_bits, d.bits);
                const int pn = d.bits == 16 ? 16 : 8; const int64_t width = (int64_t)((d.qlen + pn) / pn) * pn;
                est_cols = std::max<int64_t>(est_cols, banded ? std::min<int64_t>(width, 2LL * d.w + 3 * pn + 32) : width);
            }
            make_lds_plan(sc, max_qlen, max_bits, est_cols, (int)pass.size(), &b.lds);
            for (const AlnDesc &d : pass) b.bits_mask |= d.bits == 16 ? 1 : 2;
            bool any_wide = false, all_wide = true;
            for (const AlnDesc &d : pass) { const bool wd = d.w >= b.lds.wide_w_lo && d.w <= b.lds.wide_w_hi; any_wide |= wd; all_wide &= wd; }
            if (!any_wide) b.lds.wide_nw = 0;
            b.lds.narrow_off = (b.lds.wide_nw >= 1 && all_wide) ? 1 : 0;
        }
        if (b.lds.wide_nw > 1) dir = false;
        // (ABPOA_HIP_DIR_WIDE=1: words for the wide-band alignments too -- the device-resident driver does that by itself when the record arenas of a
        //  job do not fit the device; here it is a test switch)
        const bool dir_wide = dir && opt_env("ABPOA_HIP_DIR_WIDE") && atoi(opt_env("ABPOA_HIP_DIR_WIDE")) > 0;
        for (size_t t = 0; t < todo.size(); ++t) {
            AlnDesc &d = desc_[todo[t]];
            // direction-plane arenas for the narrow-band alignments of a dir pass; the wide-band ones keep score records (dp_common.h takes_dir / takes_wide)
            // (an alignment the general kernel will run -- seeded band of the -s retry, a row with more predecessors than a word names, no fast row loop in the
            //  plan -- stores its planes: the records' estimate, not the words'; host mirror of dp_common.h takes_fast)
            const int dbg_ = opt_env("ABPOA_HIP_DBG") ? atoi(opt_env("ABPOA_HIP_DBG")) : 0;
            const bool fast_a = (d.flags & ALN_FAST_OK) && fast_global_job(sc->gap_mode, sc->align_mode, sc->wb, sc->gap_ext1) && fast_global_aln(sc->gap_mode, d.w, d.pad0) &&
                                b.lds.fr_cols > 0 &&
                                d.qlen <= b.lds.q_cap && !(dbg_ & 64);
            const bool dir_a = dir && fast_a && (dir_wide || !(b.lds.wide_nw >= 1 && d.w >= b.lds.wide_w_lo && d.w <= b.lds.wide_w_hi));
            d.plane_cap = dir_a ? (first_pass ? dir_est_cells_[todo[t]] : dir_full_cells_[todo[t]]) : (first_pass ? est_cells_[todo[t]] : full_cells_[todo[t]]);
            d.plane_off = plane_bytes; plane_bytes += (int64_t)align_up((size_t)d.plane_cap * (d.bits / 8) + 64 * 8 * 4);   // + 64 records of slack (fast loop stores whole 64-lane chunks)
            pass[t] = d;
        }
        if ((rc = planes_.reserve((size_t)plane_bytes))) return rc;
        memcpy(hi + o_desc_, pass.data(), sizeof(AlnDesc) * pass.size());
        b.o1 = sc->gap_open1; b.e1 = sc->gap_ext1; b.o2 = sc->gap_open2; b.e2 = sc->gap_ext2;
        b.align_mode = sc->align_mode; b.gap_mode = sc->gap_mode; b.wb = sc->wb; b.zdrop = sc->zdrop; b.ret_cigar = sc->ret_cigar; b.rev_cigar = sc->rev_cigar;
        b.want_trace = trace ? 1 : 0; b.fresh_band = fresh ? 1 : 0;
        b.dir_mode = dir ? (dir_wide ? 2 : 1) : 0; b.row_sdist = di + o_sdist_; b.row_pd = (const uint32_t *)(di + o_pd_);
        b.want_lr = (trace || (flags_ & BS_WANT_BAND_STATE)) ? 1 : 0;
        { const char *dbg_ = opt_env("ABPOA_HIP_DBG"); b.dbg = dbg_ ? atoi(dbg_) : 0; }
        b.mat = (const int32_t *)(di + o_mat_); b.aln = (const AlnDesc *)(di + o_desc_); b.out = (AlnOut *)(dout + o_rec_);
        b.query = di + o_query_; b.row_base = di + o_base_; b.row_node_id = (const int32_t *)(di + o_nid_); b.row_remain = (const int32_t *)(di + o_rem_);
        b.row_active = di + o_act_; b.pred_off = (const int32_t *)(di + o_poff_); b.pred_row = (const int32_t *)(di + o_pred_);
        b.out_off = (const int32_t *)(di + o_ooff_); b.out_row = (const int32_t *)(di + o_out_);
        b.left = (int32_t *)(dout + o_left_); b.right = (int32_t *)(dout + o_right_);
        b.dp_beg_sn = (int32_t *)(dout + o_bsn_); b.dp_end_sn = (int32_t *)(dout + o_esn_); b.row_cell_off = (int64_t *)(dout + o_coff_);
        b.row_max_i = (int32_t *)(dout + o_rmi_); b.planes = planes_.dev; b.cigar = (uint64_t *)(dout + o_cig_);

        HIP_TRY(hipEventRecord(ev_[0], stream_), ABPOA_HIP_ELAUNCH);
        HIP_TRY(hipMemcpyAsync(di, hi, first_pass ? in_bytes_ : o_mat_, hipMemcpyHostToDevice, stream_), ABPOA_HIP_ELAUNCH);
        if (banded && !fresh) {
            if (first_pass) {
                saved_lr.assign((const int32_t *)(ho + o_left_), (const int32_t *)(ho + o_left_) + lr_words);
                HIP_TRY(hipMemcpyAsync(dout + o_left_, ho + o_left_, lr_words * 4, hipMemcpyHostToDevice, stream_), ABPOA_HIP_ELAUNCH);
            } else for (int i : todo) {      // only the retried alignments restart from the caller's band state
                const AlnDesc &d = desc_[i]; const size_t ro = (o_right_ - o_left_) / 4;
                HIP_TRY(hipMemcpyAsync(dout + o_left_ + 4 * d.row0, saved_lr.data() + d.row0, 4 * (size_t)d.n_rows, hipMemcpyHostToDevice, stream_), ABPOA_HIP_ELAUNCH);
                HIP_TRY(hipMemcpyAsync(dout + o_right_ + 4 * d.row0, saved_lr.data() + ro + d.row0, 4 * (size_t)d.n_rows, hipMemcpyHostToDevice, stream_), ABPOA_HIP_ELAUNCH);
            }
        }
        // -1 = "row never computed" (inactive rows, rows behind a z-drop break)
        if (first_pass) HIP_TRY(hipMemsetAsync(dout + o_bsn_, 0xFF, (o_esn_ - o_bsn_) + 4 * rows_tot_, stream_), ABPOA_HIP_ELAUNCH);
        else for (int i : todo) {
            const AlnDesc &d = desc_[i];
            HIP_TRY(hipMemsetAsync(dout + o_bsn_ + 4 * d.row0, 0xFF, 4 * (size_t)d.n_rows, stream_), ABPOA_HIP_ELAUNCH);
            HIP_TRY(hipMemsetAsync(dout + o_esn_ + 4 * d.row0, 0xFF, 4 * (size_t)d.n_rows, stream_), ABPOA_HIP_ELAUNCH);
        }
        HIP_TRY(hipEventRecord(ev_[1], stream_), ABPOA_HIP_ELAUNCH);
        int n_fast = 0;       // mirrors takes_fast() in dp_kernel.hip
        if (fast_global_job(sc->gap_mode, sc->align_mode, sc->wb, sc->gap_ext1) && b.lds.fr_cols > 0 && !(b.dbg & 64))
            for (const AlnDesc &d : pass) n_fast += ((d.flags & ALN_FAST_OK) && d.qlen <= b.lds.q_cap && fast_global_aln(sc->gap_mode, d.w, d.pad0)) ? 1 : 0;
        if (sc->gap_mode != ABPOA_HIP_LINEAR_GAP && sc->align_mode == ABPOA_HIP_LOCAL_MODE && sc->wb < 0 && b.lds.loc_cols > 0 && !(b.dbg & 64))      // mirrors takes_local() in rows_local.h
            for (const AlnDesc &d : pass) n_fast += ((d.flags & ALN_FAST_OK) && d.bits == 16 && (d.qlen / 16 + 1) * 16 <= b.lds.loc_cols && d.qlen <= b.lds.q_cap) ? 1 : 0;
        HIP_TRY(launch_dp(b, n_fast, stream_, ev_[4]), ABPOA_HIP_ELAUNCH);
        HIP_TRY(hipEventRecord(ev_[2], stream_), ABPOA_HIP_ELAUNCH);
        // results: records + cigars are adjacent at the start of the output blob; band state / trace arrays on demand
        size_t d2h = o_left_;
        if (banded && (flags_ & BS_WANT_BAND_STATE)) d2h = o_bsn_;
        if (trace) d2h = out_bytes_;
        HIP_TRY(hipMemcpyAsync(ho, dout, d2h, hipMemcpyDeviceToHost, stream_), ABPOA_HIP_ELAUNCH);
        HIP_TRY(hipEventRecord(ev_[3], stream_), ABPOA_HIP_ELAUNCH);
        HIP_TRY(hipStreamSynchronize(stream_), ABPOA_HIP_ELAUNCH);
        float ms_h2d = 0, ms_k = 0, ms_d2h = 0;
        (void)hipEventElapsedTime(&ms_h2d, ev_[0], ev_[1]); (void)hipEventElapsedTime(&ms_k, ev_[1], ev_[2]); (void)hipEventElapsedTime(&ms_d2h, ev_[2], ev_[3]);
        float ms_tail = 0;                    // pure fast-path batches: ev_[4] sits between the row-loop kernel and the tail kernel
        if (n_fast == b.n) { (void)hipEventElapsedTime(&ms_tail, ev_[4], ev_[2]); ms_k -= ms_tail; }
        stats_.n_launches += 1; stats_.kernel_ms += ms_k; stats_.tail_ms += ms_tail; stats_.h2d_ms += ms_h2d; stats_.d2h_ms += ms_d2h;

        // records are indexed by position in this pass; cigar / per-row slots by alignment, so a retry pass cannot
        // clobber the results of alignments that finished earlier
        const AlnOut *got = (const AlnOut *)(ho + o_rec_);
        std::vector<int> again; bool need_scores = false;
        for (size_t t = 0; t < todo.size(); ++t) {
            const int i = todo[t]; const AlnOut &r = got[t]; const AlnDesc &d = desc_[i];
            if (r.status == ABPOA_HIP_STATUS_NEED_SCORES && dir) { need_scores = true; again.push_back(i); stats_.n_need_scores += 1; __atomic_fetch_add(&g_dir_counts[1], 1, __ATOMIC_RELAXED);
                    continue; }
            if (r.status == ABPOA_HIP_STATUS_OVERFLOW) {
                if (!first_pass) { set_err("problem %d: arena overflow at full width (internal error)", i); return ABPOA_HIP_ELAUNCH; }
                again.push_back(i); continue;
            }
            recs_[i] = r;
            if (r.pad < 0) __atomic_fetch_add(&g_dir_counts[0], 1, __ATOMIC_RELAXED);
            // trace mode: the arena of a finished alignment is kept on the host now -- a retry pass of OTHER alignments re-uses the device arenas
            if (trace) {
                trace_arena_.resize(desc_.size());
                trace_arena_[i].resize((size_t)r.cells_used * (d.bits / 8));
                if (!trace_arena_[i].empty()) HIP_TRY(hipMemcpy(trace_arena_[i].data(), planes_.dev + d.plane_off, trace_arena_[i].size(), hipMemcpyDeviceToHost), ABPOA_HIP_ELAUNCH);
            }
            stats_.n_alignments += 1; stats_.n_cells += r.n_cells;
            stats_.algo_bytes += r.n_cells * (d.bits / 8) * (P == 1 ? 2 : (P == 3 ? 5 : 8));
            g_dbg[0] += r.clk_dp; g_dbg[1] += r.clk_bt; g_dbg[2] += r.n_rows_done; g_dbg[3] += r.n_bt_steps; for (int q_ = 0; q_ < 6; ++q_) g_dbg[4 + q_] += r.seg[q_];
        }
        todo.swap(again); first_pass = false;
        if (need_scores) dir = false;                 // the retry pass runs with score records (full width: simplest, and rare)
    }
    return ABPOA_HIP_OK;
}

int BatchStream::fetch_trace(int i, const uint8_t *row_active, abpoa_hip_trace_t *T) {
    (void)row_active;
    const AlnDesc &d = desc_[i]; const AlnOut &r = recs_[i]; const int gn = d.n_rows, P = P_;
    const bool banded = sc_.wb >= 0; const int pn = d.bits == 16 ? 16 : 8;
    uint8_t *ho = out_.host;
    T->bits = d.bits; T->n_planes = P;
    T->dp_beg = (int32_t *)malloc(4 * gn); T->dp_end = (int32_t *)malloc(4 * gn); T->dp_beg_sn = (int32_t *)malloc(4 * gn); T->dp_end_sn = (int32_t *)malloc(4 * gn);
    T->row_off = (int64_t *)malloc(8 * (gn + 1)); T->row_max_i = (int32_t *)malloc(4 * gn);
    const int32_t *bsn = (const int32_t *)(ho + o_bsn_) + d.row0, *esn = (const int32_t *)(ho + o_esn_) + d.row0;
    const int64_t *coff = (const int64_t *)(ho + o_coff_) + d.row0;
    memcpy(T->row_max_i, (const int32_t *)(ho + o_rmi_) + d.row0, 4 * gn);
    if ((size_t)i >= trace_arena_.size()) { set_err("no trace kept for problem %d", i); return ABPOA_HIP_EINVAL; }
    const std::vector<uint8_t> &arena = trace_arena_[i];          // saved when the alignment finished (BatchStream::run)
    int64_t tot = 0;
    for (int rr = 0; rr < gn; ++rr) {
        T->row_off[rr] = tot;
        const bool computed = rr < gn - 1 && bsn[rr] >= 0;
        if (!computed) { T->dp_beg[rr] = T->dp_end[rr] = T->dp_beg_sn[rr] = T->dp_end_sn[rr] = -1; T->row_max_i[rr] = -2; continue; }
        T->dp_beg_sn[rr] = bsn[rr]; T->dp_end_sn[rr] = esn[rr]; T->dp_beg[rr] = bsn[rr] * pn;
        T->dp_end[rr] = (banded || rr == 0) ? (esn[rr] + 1) * pn - 1 : d.qlen;
        if (rr == 0) T->row_max_i[rr] = -2;
        tot += (int64_t)(esn[rr] - bsn[rr] + 1) * pn * P;
    }
    T->row_off[gn] = tot;
    T->planes = malloc((size_t)std::max<int64_t>(tot, 1) * (d.bits / 8));
    const int cw = r.pad;          // arena cell stride: 0 = plane-major rows (general kernel), > 0 cell records (fast loop), < 0 direction words of -cw bytes
    for (int rr = 0; rr < gn; ++rr) {
        if (T->dp_beg_sn[rr] < 0) continue;
        const int64_t nv = T->row_off[rr + 1] - T->row_off[rr];
        if (cw < 0) {      // (ABPOA_HIP_DIRTRACE) plane 0 = the cell's direction word, the other planes 0; row 0 has no words
            const int64_t W = nv / P;
            for (int64_t x = 0; x < nv; ++x) { if (d.bits == 16) ((int16_t *)T->planes)[T->row_off[rr] + x] = 0; else ((int32_t *)T->planes)[T->row_off[rr] + x] = 0; }
            if (rr == 0) continue;
            const uint8_t *src = arena.data() + coff[rr] * (d.bits / 8);
            for (int64_t x = 0; x < W; ++x) {
                const uint32_t wv = cw == -2 ? (uint32_t)((const uint16_t *)src)[x] : ((const uint32_t *)src)[x];
                // (32-bit words in 16-bit planes: low half in plane 0, high half in plane 1)
                if (d.bits == 16) { ((int16_t *)T->planes)[T->row_off[rr] + x] = (int16_t)wv; if (cw == -4) ((int16_t *)T->planes)[T->row_off[rr] + W + x] = (int16_t)(wv >> 16); }
                else ((int32_t *)T->planes)[T->row_off[rr] + x] = (int32_t)wv;
            }
            continue;
        }
        if (cw == 0) { memcpy((uint8_t *)T->planes + T->row_off[rr] * (d.bits / 8), arena.data() + coff[rr] * (d.bits / 8), (size_t)nv * (d.bits / 8)); continue; }
        const int64_t W = nv / P;
        for (int pl = 0; pl < P; ++pl) for (int64_t x = 0; x < W; ++x) {
            const int64_t src = coff[rr] + x * cw + pl, dst = T->row_off[rr] + pl * W + x;
            if (d.bits == 16) ((int16_t *)T->planes)[dst] = ((const int16_t *)arena.data())[src];
            else ((int32_t *)T->planes)[dst] = ((const int32_t *)arena.data())[src];
        }
    }
    return 0;
}

}  // namespace abpoa_hip

using namespace abpoa_hip;

extern "C" {
int abpoa_hip_set_option(const char *name, const char *value) { return abpoa_hip::set_option(name, value) == 0 ? ABPOA_HIP_OK : ABPOA_HIP_EINVAL; }
int abpoa_hip_list_options(const char **names, const char **help, int cap) { return abpoa_hip::list_options(names, help, cap); }

int abpoa_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int abpoa_hip_init(int device) {
    abpoa_hip::refresh_options();
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.ready && g.device == device) return ABPOA_HIP_OK;
    if (g.ready) { set_err("engine already bound to device %d", g.device); return ABPOA_HIP_EINVAL; }
    // the read-set driver runs one stream per group; give the runtime enough hardware queues for them to overlap
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_err("no HIP device available (the DP has no CPU fallback)"); return ABPOA_HIP_ENODEV; }
    if (device < 0 || device >= n) { set_err("device %d out of range (0..%d)", device, n - 1); return ABPOA_HIP_ENODEV; }
    int rc = g.flat.open(device);
    if (rc) return rc;
    g.device = device; g.ready = true; memset(&g.stats, 0, sizeof(g.stats));
    return ABPOA_HIP_OK;
}

void abpoa_hip_shutdown(void) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.ready) return;
    g.flat.close();
    g.ready = false; g.device = -1;
}

const char *abpoa_hip_last_error(void) { std::lock_guard<std::mutex> lk(g_err_mu); memcpy(g_err_copy, g_err, sizeof(g_err)); return g_err_copy; }
void abpoa_hip_get_stats(abpoa_hip_stats_t *out) { std::lock_guard<std::mutex> lk(g.stats_mu); *out = g.stats; }
void abpoa_hip__dir_counts(long long *out) { out[0] = __atomic_exchange_n(&g_dir_counts[0], 0, __ATOMIC_RELAXED); out[1] = __atomic_exchange_n(&g_dir_counts[1], 0, __ATOMIC_RELAXED); }
// (test hook, host only: the LDS carve-up of the wide row loop for a launch of n_aln alignments -- out: wide_nw, ring rows, LDS bytes per workgroup,
//  workgroups per CU by the 1280-byte allocation granule of gfx950, the wide kernels' phase offset, the band half-widths [lo, hi] that take it)
void abpoa_hip__wide_plan(const abpoa_hip_scoring_t *sc, int max_qlen, int max_bits, int n_aln, int *out) {
    abpoa_hip::LdsPlan L; const int pn = max_bits == 16 ? 16 : 8; const int w = sc->wb + (int)(sc->wf * (float)max_qlen);
    abpoa_hip::make_lds_plan(sc, max_qlen, max_bits, std::min<int64_t>((int64_t)((max_qlen + pn) / pn) * pn, 2LL * w + 3 * pn + 32), n_aln, &L);
    out[0] = L.wide_nw; out[1] = L.wfr_rows; out[2] = L.total_wide; out[3] = L.total_wide > 0 ? std::min(WIDE_PER_CU_MAX, 128 / ((L.total_wide + 1279) / 1280)) : 0;
    out[4] = L.w_phase_off; out[5] = L.wide_w_lo; out[6] = L.wide_w_hi;
}
void abpoa_hip__debug_clocks(long long *out) { for (int i = 0; i < 10; ++i) { out[i] = g_dbg[i]; g_dbg[i] = 0; } }
void abpoa_hip_reset_stats(void) { std::lock_guard<std::mutex> lk(g.stats_mu); memset(&g.stats, 0, sizeof(g.stats)); }

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
    if (p->pred_off[0] != 0 || p->out_off[0] != 0) { set_err("problem %d: CSR offsets must start at 0", idx); return ABPOA_HIP_EINVAL; }
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
    abpoa_hip::refresh_options();
    if (n < 0 || !sc || (n > 0 && (!pb || !res))) { set_err("bad arguments"); return ABPOA_HIP_EINVAL; }
    if (n == 0) return ABPOA_HIP_OK;
    for (int i = 0; i < n; ++i) memset(&res[i], 0, sizeof(res[i]));
    if (!g.ready) { int rc = abpoa_hip_init(0); if (rc) return rc; }
    if (sc->m <= 0 || !sc->mat || sc->gap_mode < 0 || sc->gap_mode > 2 || sc->align_mode < 0 || sc->align_mode > 2) { set_err("bad scoring"); return ABPOA_HIP_EINVAL; }
    for (int i = 0; i < n; ++i) { int rc = validate(sc, &pb[i], i); if (rc) return rc; }
    std::lock_guard<std::mutex> lk(g.mu);
    const bool banded = sc->wb >= 0, trace = flags & ABPOA_HIP_FLAG_TRACE;
    std::vector<BatchShape> sh(n);
    for (int i = 0; i < n; ++i) sh[i] = BatchShape{pb[i].n_rows, pb[i].qlen, pb[i].pred_off[pb[i].n_rows], pb[i].out_off[pb[i].n_rows]};
    BatchStream &S = g.flat;
    int rc = S.prepare(sc, n, sh.data(), (trace ? BS_TRACE : 0) | BS_WANT_BAND_STATE);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        const abpoa_hip_problem_t &p = pb[i]; const int gn = p.n_rows; ProblemSlots s = S.slots(i);
        if (p.qlen) memcpy(s.query, p.query, p.qlen);
        memcpy(s.row_base, p.row_base, gn); memcpy(s.row_node_id, p.row_node_id, 4 * gn);
        if (p.row_remain) memcpy(s.row_remain, p.row_remain, 4 * gn); else memset(s.row_remain, 0, 4 * gn);
        if (p.row_active) memcpy(s.row_active, p.row_active, gn);
        memcpy(s.pred_off, p.pred_off, 4 * (gn + 1)); memcpy(s.pred_row, p.pred_row, 4 * (size_t)p.pred_off[gn]);
        memcpy(s.out_off, p.out_off, 4 * (gn + 1)); memcpy(s.out_row, p.out_row, 4 * (size_t)p.out_off[gn]);
        if (banded) { memcpy(s.left, p.max_pos_left, 4 * gn); memcpy(s.right, p.max_pos_right, 4 * gn); }
    }
    rc = S.run();
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        const AlnOut &r = S.rec(i); abpoa_hip_result_t &R = res[i]; const abpoa_hip_problem_t &p = pb[i];
        R.status = r.status; R.bits = S.desc(i).bits; R.best_score = r.best_score; R.best_row = r.best_row; R.best_col = r.best_col;
        R.node_s = r.node_s; R.node_e = r.node_e; R.query_s = r.query_s; R.query_e = r.query_e;
        R.n_aln_bases = r.n_aln_bases; R.n_matched_bases = r.n_matched_bases; R.n_cells = r.n_cells;
        R.n_cigar = r.status == 0 ? r.n_cigar : 0;
        if (R.n_cigar > 0) {
            R.cigar = (uint64_t *)malloc(sizeof(uint64_t) * R.n_cigar);
            if (!R.cigar) { set_err("malloc failed"); return ABPOA_HIP_ENOMEM; }
            memcpy(R.cigar, S.cigar(i), sizeof(uint64_t) * R.n_cigar);
        }
        if (banded) { memcpy(p.max_pos_left, S.left(i), 4 * p.n_rows); memcpy(p.max_pos_right, S.right(i), 4 * p.n_rows); }
        if (trace) {
            R.trace = (abpoa_hip_trace_t *)calloc(1, sizeof(abpoa_hip_trace_t));
            if ((rc = S.fetch_trace(i, p.row_active, R.trace))) return rc;
        }
    }
    add_global_stats(S.take_stats());
    return ABPOA_HIP_OK;
}

}  // extern "C"
