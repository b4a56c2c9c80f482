// Diagnostics of the device-resident driver: see msa_device_debug.h.
#include <algorithm>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "engine_options.h"
#include "msa_device_debug.h"

namespace abpoa_hip {

// Heaviest-bundling consensus (reference src/abpoa_output.c:361-415, :343-356) straight from the flat device arrays, walking
// the rows in reverse topological order instead of the reference's reverse Kahn queue: every quantity is a function of the
// successors' values only, so the visiting order does not matter as long as successors come first.
void consensus_flat(int n, const int32_t *order, const uint8_t *base, const uint8_t *nout, const int32_t *out_id, const int32_t *out_w,
                    const int32_t *n_read, std::vector<int> *ids, std::vector<uint8_t> *bases, std::vector<int> *cov, std::vector<int> &score,
                            std::vector<int> &max_out, int out_cap) {
    ids->clear(); bases->clear(); cov->clear();
    if (n <= 2) return;
    score.assign(n, 0); max_out.assign(n, -1);
    for (int r = n - 1; r >= 0; --r) {
        const int cur = order[r];
        const int32_t *oi = out_id + (size_t)cur * out_cap, *ow = out_w + (size_t)cur * out_cap; const int no = nout[cur];
        if (cur == 1) { max_out[cur] = -1; score[cur] = 0; }
        else if (cur == 0) {
            int path_score = -1, path_max_w = -1, max_id = -1;
            for (int i = 0; i < no; ++i) if (ow[i] > path_max_w || (ow[i] == path_max_w && score[oi[i]] > path_score)) { max_id = oi[i]; path_score =
                    score[oi[i]]; path_max_w = ow[i]; }
            max_out[cur] = max_id;
        } else {
            int max_w = INT_MIN, max_id = -1;
            for (int i = 0; i < no; ++i) {
                if (max_w < ow[i]) { max_w = ow[i]; max_id = oi[i]; }
                else if (max_w == ow[i] && score[max_id] <= score[oi[i]]) max_id = oi[i];
            }
            score[cur] = max_w + score[max_id]; max_out[cur] = max_id;
        }
    }
    for (int cur = max_out[0]; cur != 1 && cur >= 0; cur = max_out[cur]) { ids->push_back(cur); bases->push_back(base[cur]); cov->push_back(n_read[cur]); }
}

DeviceDebug::DeviceDebug(const PoaDev *p, const std::vector<PoaSet> *ps, const abpoa_hip_readset_t *sets, int n_sets, int m, int aln_cap, int max_reads,
        bool want_msa,
                         bool amb, hipStream_t stream)
    : p_(p), ps_(ps), sets_(sets), n_sets_(n_sets), m_(m), aln_cap_(aln_cap), max_reads_(max_reads), want_msa_(want_msa), amb_(amb), st_(stream) {
    on_ = opt_env("ABPOA_HIP_DEVSYNC") && atoi(opt_env("ABPOA_HIP_DEVSYNC"));
    if (on_) { graphs_.resize(std::min(n_sets, 4)); for (size_t i = 0; i < graphs_.size(); ++i) graphs_[i].reset(sets[i].n_reads, want_msa); }
}

int DeviceDebug::stage(const char *what, int k) {
    if (!on_) return 0;
    fprintf(stderr, "[poa-device] round %d: %s queued\n", k, what); fflush(stderr);
    hipError_t e_ = hipStreamSynchronize(st_);
    fprintf(stderr, "[poa-device] round %d: %s -> %s\n", k, what, hipGetErrorString(e_)); fflush(stderr);
    return e_ == hipSuccess ? 0 : 1;
}

// the row order the order kernel left against the host graph's Kahn walk (poa_graph.cpp topological_sort)
void DeviceDebug::order_check(int k) {
    if (!on_) return;
    const PoaDev &p = *p_; const std::vector<PoaSet> &ps = *ps_; const abpoa_hip_readset_t *sets = sets_; const int n_sets = n_sets_, aln_cap = aln_cap_,
            max_reads = max_reads_; const bool amb = amb_;
    std::vector<PoaGraph> &dbg_graphs = graphs_; (void)n_sets; (void)aln_cap; (void)max_reads; (void)amb; (void)ps; (void)sets;
    for (int s = 0; s < (int)dbg_graphs.size(); ++s) {
        if (k >= sets[s].n_reads) continue;
        PoaState hst; (void)hipMemcpy(&hst, (uint8_t *)p.state + sizeof(PoaState) * s, sizeof(hst), hipMemcpyDeviceToHost);
        if (hst.status != POA_ST_OK) { fprintf(stderr, "[poa-device]   set %d round %d: row order: set not ok (status %d reason %d)\n", s, k, hst.status,
                hst.pad); continue; }
        const int n = hst.n_nodes; std::vector<int32_t> order(n), row(n);
        (void)hipMemcpy(order.data(), p.row_node[hst.order_buf] + ps[s].node0, 4 * (size_t)n, hipMemcpyDeviceToHost);
        (void)hipMemcpy(row.data(), p.nd_row + ps[s].node0, 4 * (size_t)n, hipMemcpyDeviceToHost);
        PoaGraph &G = dbg_graphs[s]; int bad = 0;
        try { G.topological_sort(false); } catch (...) { fprintf(stderr, "[poa-device]   set %d round %d: host sort failed\n", s, k); continue; }
        if (G.n_nodes() != n) { fprintf(stderr, "[poa-device]   set %d round %d: row order check FAILED: node count %d vs host %d\n", s, k, n, G.n_nodes());
                continue; }
        for (int r = 0; r < n; ++r) { if (order[r] != G.index_to_node()[r] && bad++ < 6) fprintf(stderr,
                "[poa-device]   set %d round %d: row %d: device node %d, host node %d\n", s, k, r, order[r], G.index_to_node()[r]);
                                      if (order[r] >= 0 && order[r] < n && row[order[r]] != r && bad++ < 6) fprintf(stderr,
                                              "[poa-device]   set %d round %d: nd_row / order mismatch at row %d\n", s, k, r); }
        fprintf(stderr, "[poa-device]   set %d round %d: row order check %s (%d rows)\n", s, k, bad ? "FAILED" : "ok", n);
    }
}

// after every fuse: sets 0..3 structurally and against the host graph fed with the same cigars
void DeviceDebug::graph_check(int k) {
    if (!on_) return;
    const PoaDev &p = *p_; const std::vector<PoaSet> &ps = *ps_; const abpoa_hip_readset_t *sets = sets_; const int n_sets = n_sets_, aln_cap = aln_cap_,
            max_reads = max_reads_; const bool amb = amb_;
    std::vector<PoaGraph> &dbg_graphs = graphs_; (void)n_sets; (void)aln_cap; (void)max_reads; (void)amb; (void)ps; (void)sets;
    for (int s = 0; s < (int)dbg_graphs.size(); ++s) {
        if (k >= sets[s].n_reads) continue;
        PoaState hst; (void)hipMemcpy(&hst, (uint8_t *)p.state + sizeof(PoaState) * s, sizeof(hst), hipMemcpyDeviceToHost);
        const PoaSet &S = ps[s]; const int n = hst.n_nodes;
        // host graph: same cigar
        if (k == 0) dbg_graphs[s].add_alignment(sets[s].seqs[0], sets[s].lens[0], nullptr, 0, 0, sets[s].weights ? sets[s].weights[0] : nullptr);
        else {
            AlnOut ao; (void)hipMemcpy(&ao, (uint8_t *)p.out + sizeof(AlnOut) * s, sizeof(ao), hipMemcpyDeviceToHost);
            std::vector<uint64_t> cg(std::max(1, ao.n_cigar)); (void)hipMemcpy(cg.data(), (uint8_t *)p.cigar + 8 * S.cigar_off, 8 * (size_t)ao.n_cigar,
                    hipMemcpyDeviceToHost);
            fprintf(stderr, "[poa-device]   set %d round %d: dp status %d score %d n_cigar %d rows %d; device state status %d reason %d nodes %d\n", s, k,
                    ao.status, ao.best_score, ao.n_cigar, ao.n_rows_done, hst.status, hst.pad, n);
            if (ao.status != 0) continue;
            uint8_t rcf = 0; if (amb) (void)hipMemcpy(&rcf, p.is_rc + S.read0 + k, 1, hipMemcpyDeviceToHost);
            const int ql_ = sets[s].lens[k]; std::vector<uint8_t> rq_; std::vector<int32_t> rw_;
            if (rcf) { rq_.resize(ql_); for (int j = 0; j < ql_; ++j) { const uint8_t c_ = sets[s].seqs[k][ql_ - 1 - j]; rq_[j] = c_ < 4 ? (uint8_t)(3 - c_)
                    : (uint8_t)4; }
                       if (sets[s].weights && sets[s].weights[k]) { rw_.resize(ql_); for (int j = 0; j < ql_; ++j) rw_[j] = sets[s].weights[k][ql_ - 1 - j]; } }
            dbg_graphs[s].add_alignment(rcf ? rq_.data() : sets[s].seqs[k], ql_, cg.data(), ao.n_cigar, k, rcf && !rw_.empty() ? rw_.data() : (sets[s].weights
                    ? sets[s].weights[k] : nullptr));
        }
        if (hst.status != POA_ST_OK) continue;
        std::vector<uint8_t> base(n), nin(n), nout(n), naln(n); std::vector<int32_t> in(n * (int)p.in_cap), outv(n * (int)p.out_cap), outw(n * (int)p.out_cap),
                aln((size_t)n * aln_cap), nread(n), row(n), order(n);
        auto dl = [&](void *dst, const void *pool, size_t elem, size_t per) { (void)hipMemcpy(dst, (const uint8_t *)pool + (size_t)S.node0 * elem * per,
                (size_t)n * elem * per, hipMemcpyDeviceToHost); };
        dl(base.data(), p.nd_base, 1, 1); dl(nin.data(), p.nd_nin, 1, 1); dl(nout.data(), p.nd_nout, 1, 1); dl(naln.data(), p.nd_naln, 1, 1);
        // edge lists come back as hot + cold halves and are merged into [node][CAP] arrays
        auto dl_list = [&](int32_t *dst, const int32_t *hot, const int32_t *cold, int cap_) {
            std::vector<int32_t> h_((size_t)n * POA_HOT), c_((size_t)n * (cap_ - POA_HOT));
            (void)hipMemcpy(h_.data(), hot + (size_t)S.node0 * POA_HOT, h_.size() * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(c_.data(), cold + (size_t)S.node0 * (cap_ - POA_HOT), c_.size() * 4, hipMemcpyDeviceToHost);
            for (int u_ = 0; u_ < n; ++u_) for (int t_ = 0; t_ < cap_; ++t_) dst[(size_t)u_ * cap_ + t_] = t_ < POA_HOT ? h_[(size_t)u_ * POA_HOT + t_]
                    : c_[(size_t)u_ * (cap_ - POA_HOT) + t_ - POA_HOT];
        };
        dl_list(in.data(), p.nd_in, p.nd_inx, (int)p.in_cap); dl_list(outv.data(), p.nd_out, p.nd_outx, (int)p.out_cap); dl_list(outw.data(), p.nd_outw, p.nd_outwx,
                (int)p.out_cap); dl(aln.data(), p.nd_aln, 4, aln_cap);
        dl(nread.data(), p.nd_nread, 4, 1); dl(row.data(), p.nd_row, 4, 1); dl(order.data(), p.row_node[hst.order_buf], 4, 1);
        int bad = 0;
        auto complain = [&](const char *what, int a, int b_) { if (bad++ < 8) fprintf(stderr, "[poa-device]   set %d round %d: %s (%d, %d)\n", s, k, what, a,
                b_); };
        const PoaGraph &G = dbg_graphs[s];
        if (G.n_nodes() != n) complain("node count differs from host graph", n, G.n_nodes());
        for (int r = 0; r < n; ++r) { if (order[r] < 0 || order[r] >= n) { complain("order entry out of range", r, order[r]); continue;
                } if (row[order[r]] != r) complain("nd_row / order mismatch", r, order[r]); }
        for (int u = 0; u < n && u < G.n_nodes(); ++u) {
            const PoaNode &h = G.node(u);
            if (h.base != base[u]) complain("base differs", u, base[u]);
            if (h.in_id.size() != nin[u]) complain("in-degree differs", u, nin[u]);
            else for (int t = 0; t < std::min<int>(nin[u], (int)p.in_cap); ++t) { if (h.in_id[t] != in[u * (int)p.in_cap + t]) complain("in edge differs", u, t);
                    if (row[in[u * (int)p.in_cap
                    + t]] >= row[u]) complain("order violated (pred row >= row)", in[u * (int)p.in_cap + t], u); }
            if (h.out_id.size() != nout[u]) complain("out-degree differs", u, nout[u]);
            else for (int t = 0; t < std::min<int>(nout[u], (int)p.out_cap); ++t) { if (h.out_id[t] != outv[u * (int)p.out_cap + t]) complain("out edge differs", u,
                    t);
                    if (h.out_w[t] != outw[u * (int)p.out_cap + t]) complain("out weight differs", u, t); }
            if (h.aligned.size() != naln[u]) complain("aligned count differs", u, naln[u]);
            else for (int t = 0; t < naln[u]; ++t) if (h.aligned[t] != aln[(size_t)u * aln_cap + t]) complain("aligned node differs", u, t);
            if (h.n_read != nread[u]) complain("n_read differs", u, nread[u]);
        }
        fprintf(stderr, "[poa-device]   set %d round %d: graph check %s (%d nodes)\n", s, k, bad ? "FAILED" : "ok", n);
    }
}

// load balance of the round: ticks of the mean and of the slowest alignment (ABPOA_HIP_IMBAL; the censuses need the diagnostic builds)
void DeviceDebug::balance_report(int k, const DevBatch &b, hipEvent_t rows_begin, hipEvent_t rows_end) {
    if (!on_ || !opt_env("ABPOA_HIP_IMBAL")) return;
    const PoaDev &p = *p_; const std::vector<PoaSet> &ps = *ps_; const abpoa_hip_readset_t *sets = sets_; const int n_sets = n_sets_, aln_cap = aln_cap_,
            max_reads = max_reads_; const bool amb = amb_;
    std::vector<PoaGraph> &dbg_graphs = graphs_; (void)n_sets; (void)aln_cap; (void)max_reads; (void)amb; (void)ps; (void)sets;
    hipEvent_t e[2] = {rows_begin, rows_end};
    std::vector<double> &tot_set = tot_set_; double &sum_max = sum_max_;
    std::vector<AlnOut> ho(n_sets); (void)hipMemcpy(ho.data(), p.out, sizeof(AlnOut) * n_sets, hipMemcpyDeviceToHost);
    double sd = 0, sb = 0; long long md = 0, mb = 0, ms_ = 0; for (const AlnOut &o_ : ho) { sd += o_.clk_dp; sb += o_.clk_bt; md = std::max<long long>(md,
            o_.clk_dp); mb = std::max<long long>(mb, o_.clk_bt); ms_ = std::max<long long>(ms_, o_.clk_dp + o_.clk_bt); }
    // wall time of the row-loop launches: max ticks / this = tick rate the slowest wave saw
    float rows_ms_ = 0; (void)hipEventElapsedTime(&rows_ms_, e[0], e[1]);
    fprintf(stderr,
            "[poa-device] round %d balance: rows mean %.0f max %lld (%.2f ms on the stream, %.2f Gticks/s) | tail "
                    "mean %.0f max %lld | rows+tail mean %.0f max %lld\n", k, sd / n_sets, md, rows_ms_, rows_ms_ > 0 ? md / rows_ms_ * 1e-6 : 0.0,
                    sb / n_sets, mb, (sd + sb) / n_sets, ms_);
    if (k == 1) { tot_set.assign(n_sets, 0.0); sum_max = 0; }
    for (int s_ = 0; s_ < n_sets; ++s_) tot_set[s_] += (double)ho[s_].clk_dp + (double)ho[s_].clk_bt;
    sum_max += (double)ms_;
    if (k == max_reads - 1) { double mx_ = 0, mean_ = 0; for (double v_ : tot_set) { mx_ = std::max(mx_, v_); mean_ += v_; } fprintf(stderr,
            "[poa-device] rows+tail ticks over all rounds: sum of per-round maxima %.0f | slowest set alone %.0f | mean set %.0f\n", sum_max, mx_,
            mean_ / n_sets); }
    double sg[6] = {0, 0, 0, 0, 0, 0}, st_ = 0; for (const AlnOut &o_ : ho) { for (int q_ = 0; q_ < 6; ++q_) sg[q_] += o_.seg[q_]; st_ += o_.n_bt_steps; }
    // (direction-plane arenas: how much of them is score records of rows kept for later readers)
    if (b.dir_mode) { double cu = 0, nc = 0, rd = 0; for (const AlnOut &o_ : ho) { cu += (double)o_.cells_used; nc += (double)o_.n_cells; rd += o_.n_rows_done;
            }
                      fprintf(stderr,
                              "[poa-device] round %d arenas: %.0f rows, %.0f columns, %.0f units of 32 B per alignment; words alone "
                                      "would take %.0f (int16 affine) -> rows keeping their records: ~%.1f %%\n", k, rd / n_sets, nc / n_sets,
                                      cu / n_sets / 16, nc / n_sets / 16, 100.0 * (cu - nc) / (4.0 * nc + 1)); }
    // placement report (row-loop seg[5] = HW_ID | XCC_ID << 32 survives the tail under dbg bit 7): how many alignments shared a SIMD, and how the sharers fared
    if (b.dbg & 128) {
        // xcc | se, sh, cu | simd
        std::vector<std::pair<unsigned long long, int>> pl; for (int s_ = 0; s_ < n_sets; ++s_) { const unsigned long long h_ =
                (unsigned long long)ho[s_].seg[5]; pl.push_back({((h_ >> 32) & 15) << 16 | (h_ & 0xff30) , s_}); }
        std::sort(pl.begin(), pl.end()); double t_sh = 0, t_al = 0; int n_sh = 0, n_al = 0;
        for (size_t i_ = 0; i_ < pl.size(); ++i_) { const bool sh_ = (i_ > 0 && pl[i_ - 1].first == pl[i_].first) || (i_ + 1 < pl.size() && pl[i_
                + 1].first == pl[i_].first); (sh_ ? t_sh : t_al) += (double)ho[pl[i_].second].clk_dp; (sh_ ? n_sh : n_al)++; }
        fprintf(stderr, "[poa-device] round %d placement: %d alignments alone on their SIMD (mean ticks %.0f), %d sharing one (mean ticks %.0f)\n", k, n_al,
                n_al ? t_al / n_al : 0.0, n_sh, n_sh ? t_sh / n_sh : 0.0);
    }
    // (library built with -DABPOA_HIP_ROW_CENSUS) rows and ticks per body of the narrow row loop, mean per alignment
    if ((b.dbg & 128) && opt_env("ABPOA_HIP_ROW_CENSUS")) {
        double rw[6] = {0, 0, 0, 0, 0, 0}, tk[6] = {0, 0, 0, 0, 0, 0}; for (const AlnOut &o_ : ho) for (int q_ = 0; q_ < 6; ++q_) {
                rw[q_] += (double)(o_.seg[q_] >> 40); tk[q_] += (double)(o_.seg[q_] & ((1ll << 40) - 1)); }
        // (a build with -DABPOA_HIP_ASM_CENSUS as well: slot 0 = the rows of the assembly loop, slot 1 = the C++ copies of the one- / two-predecessor body)
        const bool asmc_ = opt_env("ABPOA_HIP_ROW_CENSUS") && atoi(opt_env("ABPOA_HIP_ROW_CENSUS")) == 2;
        const char *nm_[5] = {asmc_ ? "assembly loop" : "1 predecessor", asmc_ ? "C++ 1-2 predecessors" : "2 predecessors", "3-8 predecessors (C++)", "all-chunks / exact bodies", "tile switches"};
        fprintf(stderr, "[poa-device] round %d narrow-loop census per alignment:", k);
        for (int q_ = 0; q_ < 5; ++q_) fprintf(stderr, " %s %.0f x %.0f ticks |", nm_[q_], rw[q_] / n_sets, rw[q_] > 0 ? tk[q_] / rw[q_] : 0.0);
        { int w_ = 0; for (int s_ = 0; s_ < n_sets; ++s_) if (ho[s_].clk_dp > ho[w_].clk_dp) w_ = s_; const AlnOut &o_ = ho[w_];
          fprintf(stderr, "\n[poa-device] round %d slowest row loop (set %d, %lld ticks):", k, w_, (long long)o_.clk_dp);
          for (int q_ = 0; q_ < 5; ++q_) fprintf(stderr, " %s %lld x %.0f |", nm_[q_], (long long)(o_.seg[q_] >> 40), (o_.seg[q_] >> 40)
                  ? (double)(o_.seg[q_] & ((1ll << 40) - 1)) / (double)(o_.seg[q_] >> 40) : 0.0);
          fprintf(stderr, " exact-body rows: > 4 predecessors %lld, straight-line declined %lld, beyond the ring %lld;", (long long)(o_.seg[5] >> 40),
                  (long long)((o_.seg[5] >> 20) & 0xfffff), (long long)(o_.seg[5] & 0xfffff)); }
        double why[3] = {0, 0, 0}; for (const AlnOut &o_ : ho) { why[0] += (double)(o_.seg[5] >> 40); why[1] += (double)((o_.seg[5] >> 20) & 0xfffff);
                why[2] += (double)(o_.seg[5] & 0xfffff); }
        fprintf(stderr, " exact-body rows: > 4 predecessors %.0f, straight-line body declined %.0f, predecessor beyond the ring %.0f\n", why[0] / n_sets,
                why[1] / n_sets, why[2] / n_sets);
    }
    if ((b.dbg & 128) && opt_env("ABPOA_HIP_WIDE_COUNTERS")) { int w_ = 0; for (int s_ = 0; s_ < n_sets; ++s_) if (ho[s_].clk_dp > ho[w_].clk_dp) w_ = s_;
            const AlnOut &o_ = ho[w_];
        fprintf(stderr,
                "[poa-device] round %d slowest row loop: set %d ticks %lld rows %d | all-chunk body %lld | not eligible "
                        "%lld | ring-geometry %lld | > 5 chunks %lld | slow vectors straddle %lld | key window / wrap %lld\n", k, w_, (long long)o_.clk_dp,
                        o_.n_rows_done, (long long)o_.seg[0], (long long)o_.seg[1], (long long)o_.seg[2], (long long)o_.seg[3], (long long)o_.seg[4],
                        (long long)o_.seg[5]); }
    if (opt_env("ABPOA_HIP_WIDE_COUNTERS")) fprintf(stderr,
            "[poa-device] round %d wide-loop rows per alignment (diagnostic build): all-chunk body %.0f | not eligible "
                    "(preds > 8 / distance) %.0f | ring-geometry %.0f | > 5 chunks %.0f | slow vectors straddle %.0f | key " "window / wrap %.0f\n", k,
                    sg[0] / n_sets, sg[1] / n_sets, sg[2] / n_sets, sg[3] / n_sets, sg[4] / n_sets, sg[5] / n_sets);
    fprintf(stderr,
            "[poa-device] round %d tail means: steps %.0f  flag steps %.0f  slow steps %.0f  windows %.1f  window ticks %.0f (setup %.0f)  walk ticks %.0f\n",
            k, st_ / n_sets, sg[2] / n_sets / 1000, sg[3] / n_sets / 1000, sg[4] / n_sets / 1000, sg[5] / n_sets, sg[0] / n_sets, sg[1] / n_sets);
}

// the device MSA of the first sets against the host routine on the host graph that was fed the same cigars
void DeviceDebug::msa_check(const PoaState *hs, const abpoa_hip_msa_t *out) {
    if (!on_ || !want_msa_) return;
    const PoaDev &p = *p_; const std::vector<PoaSet> &ps = *ps_; const abpoa_hip_readset_t *sets = sets_; const int n_sets = n_sets_, aln_cap = aln_cap_,
            max_reads = max_reads_; const bool amb = amb_;
    std::vector<PoaGraph> &dbg_graphs = graphs_; (void)n_sets; (void)aln_cap; (void)max_reads; (void)amb; (void)ps; (void)sets;
    for (int s = 0; s < (int)dbg_graphs.size(); ++s) {
        if (hs[s].status != POA_ST_OK || hs[s].n_nodes <= 2) continue;
        int ml = 0; std::vector<std::vector<uint8_t>> rows; std::vector<int> col;
        try { dbg_graphs[s].rc_msa(m_, &ml, &rows, &col); } catch (...) { fprintf(stderr, "[poa-device]   set %d: host rc_msa failed\n", s); continue; }
        bool same = ml == out[s].msa_len;
        for (int r = 0; same && r < sets[s].n_reads; ++r) same = memcmp(rows[r].data(), out[s].msa_base + (size_t)r * ml, (size_t)ml) == 0;
        fprintf(stderr, "[poa-device]   set %d: msa check %s (device %d columns, host %d)\n", s, same ? "ok" : "FAILED", out[s].msa_len, ml);
    }
}

// the device consensus of the first sets against the host routine on the downloaded graph
void DeviceDebug::consensus_check(const PoaState *hs, const abpoa_hip_msa_t *out) {
    if (!on_) return;
    const PoaDev &p = *p_; const std::vector<PoaSet> &ps = *ps_; const abpoa_hip_readset_t *sets = sets_; const int n_sets = n_sets_, aln_cap = aln_cap_,
            max_reads = max_reads_; const bool amb = amb_;
    std::vector<PoaGraph> &dbg_graphs = graphs_; (void)n_sets; (void)aln_cap; (void)max_reads; (void)amb; (void)ps; (void)sets;
    for (int s = 0; s < std::min(n_sets, 4); ++s) {
        if (hs[s].status != POA_ST_OK) continue;
        const PoaSet &S = ps[s]; const int n = hs[s].n_nodes;
        std::vector<uint8_t> base(n), nout(n); std::vector<int32_t> outv((size_t)n * (int)p.out_cap), outw((size_t)n * (int)p.out_cap), nread(n), order(n);
        auto dl = [&](void *dst, const void *pool, size_t elem, size_t per) { (void)hipMemcpy(dst, (const uint8_t *)pool + (size_t)S.node0 * elem * per,
                (size_t)n * elem * per, hipMemcpyDeviceToHost); };
        // edge lists come back as hot + cold halves and are merged into [node][CAP] arrays
        auto dl_list = [&](int32_t *dst, const int32_t *hot, const int32_t *cold, int cap_) {
            std::vector<int32_t> h_((size_t)n * POA_HOT), c_((size_t)n * (cap_ - POA_HOT));
            (void)hipMemcpy(h_.data(), hot + (size_t)S.node0 * POA_HOT, h_.size() * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(c_.data(), cold + (size_t)S.node0 * (cap_ - POA_HOT), c_.size() * 4, hipMemcpyDeviceToHost);
            for (int u_ = 0; u_ < n; ++u_) for (int t_ = 0; t_ < cap_; ++t_) dst[(size_t)u_ * cap_ + t_] = t_ < POA_HOT ? h_[(size_t)u_ * POA_HOT + t_]
                    : c_[(size_t)u_ * (cap_ - POA_HOT) + t_ - POA_HOT];
        };
        dl(base.data(), p.nd_base, 1, 1); dl(nout.data(), p.nd_nout, 1, 1);
        if (nout[0] > (int)p.out_cap) { fprintf(stderr, "[poa-device]   set %d: consensus check skipped (the source has %d out-edges: the check's arrays hold "
                "%d per node)\n", s, (int)nout[0], (int)p.out_cap); continue; }
        dl_list(outv.data(), p.nd_out, p.nd_outx, (int)p.out_cap); dl_list(outw.data(),
                p.nd_outw, p.nd_outwx, (int)p.out_cap);
        dl(nread.data(), p.nd_nread, 4, 1); dl(order.data(), p.row_node[hs[s].order_buf], 4, 1);
        std::vector<int> ids, cov, sc_, mo; std::vector<uint8_t> bases;
        consensus_flat(n, order.data(), base.data(), nout.data(), outv.data(), outw.data(), nread.data(), &ids, &bases, &cov, sc_, mo, (int)p.out_cap);
        bool same = (int)ids.size() == out[s].cons_len;
        for (size_t i = 0; same && i < ids.size(); ++i) same = ids[i] == out[s].cons_node_id[i] && bases[i] == out[s].cons_base[i]
                && cov[i] == out[s].cons_cov[i];
        fprintf(stderr, "[poa-device]   set %d: consensus check %s (device %d, host %zu bases)\n", s, same ? "ok" : "FAILED", out[s].cons_len, ids.size());
    }
}

}  // namespace abpoa_hip
