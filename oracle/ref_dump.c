/*
 * TEST INFRASTRUCTURE -- golden-vector generator.  Links against the UNMODIFIED reference
 * (oracle/_ref/libabpoa_ref.so, built by oracle/Makefile from /root/reference) and replays the
 * reference's own per-read loop (ref: abpoa_poa, src/abpoa_align.c:302-344).  For selected reads it
 * dumps, per alignment, (a) the flat problem the DP consumed (graph snapshot in topological order,
 * query, scoring) and (b) everything the reference DP produced: dp_beg/dp_end per row, all score
 * planes inside the vector-rounded band, best score, cigar and abpoa_res_t fields.  The planes are
 * read straight out of ab->abm->s_mem after the call (layout: src/simd_abpoa_align.c:469,480,494).
 *
 * Output: one ".abpg" container per dumped alignment + the reference's final consensus/MSA text.
 * Container = magic "ABPG0001", then records {char name[24]; int32 dtype; int32 pad; int64 count; data}
 * dtype: 0 u8, 1 i32, 2 i64, 3 f32, 4 i16, 5 u64.
 *
 * usage: ref_dump [abpoa-like options] -D outdir -R reads(comma list|all) [-P 0|1 planes] [-G beg,end sub-graph node ids] in.fa
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <getopt.h>
#include <ctype.h>
#include "abpoa.h"
#include "abpoa_graph.h"
#include "abpoa_align.h"
#include "simd_abpoa_align.h"

extern char ab_char26_table[256];

static void put(FILE *fp, const char *name, int dtype, int64_t count, const void *data) {
    static const int sz[] = {1, 4, 8, 4, 2, 8};
    char nm[24]; memset(nm, 0, sizeof(nm)); strncpy(nm, name, 23);
    int32_t dt = dtype, pad = 0;
    fwrite(nm, 1, 24, fp); fwrite(&dt, 4, 1, fp); fwrite(&pad, 4, 1, fp); fwrite(&count, 8, 1, fp);
    if (count > 0) fwrite(data, sz[dtype], (size_t)count, fp);
}
static void put_i32(FILE *fp, const char *name, int32_t v) { put(fp, name, 1, 1, &v); }

static uint64_t mix64(uint64_t x) { /* splitmix64 finaliser: weight of a (col,plane) slot in the row checksum */
    x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31);
}

typedef struct { char **name; uint8_t **seq; int *len; int n; } reads_t;

static reads_t read_fasta(const char *fn) {
    reads_t r = {0, 0, 0, 0}; int m = 0; FILE *fp = fopen(fn, "r");
    if (!fp) { fprintf(stderr, "cannot open %s\n", fn); exit(1); }
    char *line = NULL; size_t cap = 0; ssize_t l; int cur = -1, curm = 0;
    while ((l = getline(&line, &cap, fp)) > 0) {
        while (l > 0 && isspace((unsigned char)line[l - 1])) line[--l] = 0;
        if (l == 0) continue;
        if (line[0] == '>') {
            if (r.n == m) { m = m ? m * 2 : 64; r.name = realloc(r.name, m * sizeof(char *)); r.seq = realloc(r.seq, m * sizeof(uint8_t *)); r.len = realloc(r.len, m * sizeof(int)); }
            cur = r.n++; char *sp = line + 1; while (*sp && !isspace((unsigned char)*sp)) ++sp; *sp = 0;
            r.name[cur] = strdup(line + 1); r.seq[cur] = NULL; r.len[cur] = 0; curm = 0;
        } else if (cur >= 0) {
            if (r.len[cur] + l > curm) { curm = (r.len[cur] + (int)l) * 2; r.seq[cur] = realloc(r.seq[cur], curm); }
            for (int i = 0; i < l; ++i) r.seq[cur][r.len[cur]++] = (uint8_t)ab_char26_table[(unsigned char)line[i]];
        }
    }
    free(line); fclose(fp); return r;
}

int main(int argc, char **argv) {
    abpoa_para_t *abpt = abpoa_init_para();
    int c; char *s; const char *outdir = "."; const char *which = "all"; int dump_planes = 1; int sub_beg = -1, sub_end = -1;
    while ((c = getopt(argc, argv, "m:M:X:t:O:E:b:f:z:cr:sD:R:P:G:")) >= 0) {
        switch (c) {
            case 'm': abpt->align_mode = atoi(optarg); break;
            case 'M': abpt->match = atoi(optarg); break;
            case 'X': abpt->mismatch = atoi(optarg); break;
            case 't': abpt->use_score_matrix = 1; abpt->mat_fn = strdup(optarg); break;
            case 'O': abpt->gap_open1 = strtol(optarg, &s, 10); if (*s == ',') abpt->gap_open2 = strtol(s + 1, &s, 10); break;
            case 'E': abpt->gap_ext1 = strtol(optarg, &s, 10); if (*s == ',') abpt->gap_ext2 = strtol(s + 1, &s, 10); break;
            case 'b': abpt->wb = atoi(optarg); break;
            case 'f': abpt->wf = atof(optarg); break;
            case 'z': abpt->zdrop = atoi(optarg); break;
            case 'c': abpt->m = 27; abpt->mat = (int *)realloc(abpt->mat, abpt->m * abpt->m * sizeof(int)); break;
            case 's': abpt->amb_strand = 1; break;
            case 'r': { int r = atoi(optarg);
                      if (r == 0) abpt->out_cons = 1, abpt->out_msa = 0; else if (r == 1) abpt->out_cons = 0, abpt->out_msa = 1;
                      else if (r == 2) abpt->out_cons = abpt->out_msa = 1; } break;
            case 'D': outdir = optarg; break;
            case 'R': which = optarg; break;
            case 'P': dump_planes = atoi(optarg); break;
            case 'G': sub_beg = strtol(optarg, &s, 10); if (*s == ',') sub_end = strtol(s + 1, &s, 10); break;
            default: fprintf(stderr, "bad option\n"); return 1;
        }
    }
    if (argc - optind != 1) { fprintf(stderr, "usage: ref_dump [opts] in.fa\n"); return 1; }
    abpoa_post_set_para(abpt);
    reads_t rd = read_fasta(argv[optind]);
    uint8_t *sel = calloc(rd.n + 1, 1);
    if (strcmp(which, "all") == 0) memset(sel, 1, rd.n);
    else if (strcmp(which, "none") != 0) { char *w = strdup(which), *tok = strtok(w, ","); while (tok) { int i = atoi(tok); if (i >= 0 && i < rd.n) sel[i] = 1; tok = strtok(NULL, ","); } free(w); }

    abpoa_t *ab = abpoa_init();
    abpoa_reset(ab, abpt, 1024);
    abpoa_seq_t *abs = ab->abs; abs->n_seq = rd.n;
    /* names: grow like abpoa_realloc_seq would; we only need n_seq and is_rc for the output stage */
    if (rd.n > abs->m_seq) {
        int old = abs->m_seq; abs->m_seq = rd.n;
        abs->seq = realloc(abs->seq, abs->m_seq * sizeof(abpoa_str_t)); abs->name = realloc(abs->name, abs->m_seq * sizeof(abpoa_str_t));
        abs->comment = realloc(abs->comment, abs->m_seq * sizeof(abpoa_str_t)); abs->qual = realloc(abs->qual, abs->m_seq * sizeof(abpoa_str_t));
        abs->is_rc = realloc(abs->is_rc, abs->m_seq);
        for (int i = old; i < abs->m_seq; ++i) { memset(&abs->seq[i], 0, sizeof(abpoa_str_t)); memset(&abs->name[i], 0, sizeof(abpoa_str_t)); memset(&abs->comment[i], 0, sizeof(abpoa_str_t)); memset(&abs->qual[i], 0, sizeof(abpoa_str_t)); abs->is_rc[i] = 0; }
    }
    for (int i = 0; i < rd.n; ++i) { abs->name[i].l = (int)strlen(rd.name[i]); abs->name[i].m = abs->name[i].l + 1; abs->name[i].s = strdup(rd.name[i]); abs->is_rc[i] = 0; }

    for (int ri = 0; ri < rd.n; ++ri) {
        int qlen = rd.len[ri]; uint8_t *query = rd.seq[ri];
        abpoa_res_t res; memset(&res, 0, sizeof(res)); res.graph_cigar = 0; res.n_cigar = 0;
        abpoa_graph_t *g = ab->abg;
        if (g->node_n > 2) {
            if (g->is_topological_sorted == 0) abpoa_topological_sort(g, abpt);
            int beg_id = ABPOA_SRC_NODE_ID, end_id = ABPOA_SINK_NODE_ID;
            if (sub_beg >= 0 && sub_end >= 0 && sub_beg < g->node_n && sub_end < g->node_n && sel[ri]) { beg_id = sub_beg; end_id = sub_end; }
            int beg_index = g->node_id_to_index[beg_id], end_index = g->node_id_to_index[end_id];
            int gn = end_index - beg_index + 1;
            FILE *fp = NULL;
            if (sel[ri]) {
                char fn[1024]; snprintf(fn, sizeof(fn), "%s/aln_%03d.abpg", outdir, ri);
                fp = fopen(fn, "wb"); if (!fp) { fprintf(stderr, "cannot write %s\n", fn); return 1; }
                fwrite("ABPG0001", 1, 8, fp);
                /* ---- inputs (captured BEFORE the call: the DP mutates max_pos_left/right) ---- */
                put_i32(fp, "m", abpt->m); put(fp, "mat", 1, abpt->m * abpt->m, abpt->mat);
                put_i32(fp, "max_mat", abpt->max_mat); put_i32(fp, "min_mis", abpt->min_mis);
                put_i32(fp, "gap_open1", abpt->gap_open1); put_i32(fp, "gap_ext1", abpt->gap_ext1);
                put_i32(fp, "gap_open2", abpt->gap_open2); put_i32(fp, "gap_ext2", abpt->gap_ext2);
                put_i32(fp, "align_mode", abpt->align_mode); put_i32(fp, "gap_mode", abpt->gap_mode);
                put_i32(fp, "wb", abpt->wb); put(fp, "wf", 3, 1, &abpt->wf); put_i32(fp, "zdrop", abpt->zdrop);
                put_i32(fp, "ret_cigar", abpt->ret_cigar); put_i32(fp, "rev_cigar", abpt->rev_cigar);
                put_i32(fp, "read_index", ri); put_i32(fp, "n_rows", gn); put_i32(fp, "qlen", qlen);
                put_i32(fp, "beg_index", beg_index); put_i32(fp, "end_index", end_index); put_i32(fp, "node_n", g->node_n);
                put(fp, "query", 0, qlen, query);
                /* index_map, ref: simd_abpoa_align.c:1650-1660 */
                uint8_t *index_map = calloc(g->node_n, 1); index_map[beg_index] = index_map[end_index] = 1;
                for (int i = beg_index; i < end_index - 1; ++i) {
                    if (!index_map[i]) continue;
                    int id = g->index_to_node_id[i];
                    for (int j = 0; j < g->node[id].out_edge_n; ++j) index_map[g->node_id_to_index[g->node[id].out_id[j]]] = 1;
                }
                uint8_t *base = malloc(gn); int32_t *nid = malloc(gn * 4), *remain = malloc(gn * 4), *left = malloc(gn * 4), *right = malloc(gn * 4);
                int32_t *poff = malloc((gn + 1) * 4), *ooff = malloc((gn + 1) * 4); int np = 0, no = 0;
                for (int r = 0; r < gn; ++r) { int id = g->index_to_node_id[beg_index + r]; np += g->node[id].in_edge_n; no += g->node[id].out_edge_n; }
                int32_t *prow = malloc((np + 1) * 4), *orow = malloc((no + 1) * 4); np = no = 0;
                for (int r = 0; r < gn; ++r) {
                    int id = g->index_to_node_id[beg_index + r];
                    base[r] = g->node[id].base; nid[r] = id;
                    remain[r] = g->node_id_to_max_remain ? g->node_id_to_max_remain[id] : 0;
                    left[r] = g->node_id_to_max_pos_left ? g->node_id_to_max_pos_left[id] : 0;
                    right[r] = g->node_id_to_max_pos_right ? g->node_id_to_max_pos_right[id] : 0;
                    poff[r] = np; ooff[r] = no;
                    if (r > 0) for (int j = 0; j < g->node[id].in_edge_n; ++j) {   /* ref :519-530 */
                        int pi = g->node_id_to_index[g->node[id].in_id[j]];
                        if (index_map[pi]) prow[np++] = pi - beg_index;
                    }
                    for (int j = 0; j < g->node[id].out_edge_n; ++j) {
                        int oi = g->node_id_to_index[g->node[id].out_id[j]];
                        orow[no++] = (oi >= beg_index && oi <= end_index) ? oi - beg_index : -1;
                    }
                }
                poff[gn] = np; ooff[gn] = no;
                put(fp, "row_base", 0, gn, base); put(fp, "row_node_id", 1, gn, nid); put(fp, "row_remain", 1, gn, remain);
                put(fp, "row_active", 0, gn, index_map + beg_index);
                put(fp, "pred_off", 1, gn + 1, poff); put(fp, "pred_row", 1, np, prow);
                put(fp, "out_off", 1, gn + 1, ooff); put(fp, "out_row", 1, no, orow);
                put(fp, "left_in", 1, gn, left); put(fp, "right_in", 1, gn, right);
                free(index_map); free(base); free(nid); free(remain); free(left); free(right); free(poff); free(ooff); free(prow); free(orow);
            }
            res.n_aln_bases = 0; res.n_matched_bases = 0;
            if (fp) {   /* sentinel-fill the band arrays so rows the DP never reaches (z-drop break) are recognisable;
                           grow them first exactly like simd_abpoa_realloc would (ref :1200-1206) */
                abpoa_simd_matrix_t *abm = ab->abm;
                if (gn > abm->rang_m) {
                    abm->rang_m = gn; kroundup32(abm->rang_m);
                    abm->dp_beg = realloc(abm->dp_beg, abm->rang_m * sizeof(int)); abm->dp_end = realloc(abm->dp_end, abm->rang_m * sizeof(int));
                    abm->dp_beg_sn = realloc(abm->dp_beg_sn, abm->rang_m * sizeof(int)); abm->dp_end_sn = realloc(abm->dp_end_sn, abm->rang_m * sizeof(int));
                }
                for (int r = 0; r < gn; ++r) abm->dp_beg[r] = abm->dp_end[r] = abm->dp_beg_sn[r] = abm->dp_end_sn[r] = -1;
            }
            simd_abpoa_align_sequence_to_subgraph(ab, abpt, beg_id, end_id, query, qlen, &res);
            if (fp) {
                /* ---- expected outputs ---- */
                int32_t max_score, bits; int len = qlen > gn ? qlen : gn;     /* ref :1672-1683 */
                int oe1 = abpt->gap_open1 + abpt->gap_ext1, oe2 = abpt->gap_open2 + abpt->gap_ext2;
                max_score = qlen * abpt->max_mat > len * abpt->gap_ext1 + abpt->gap_open1 ? qlen * abpt->max_mat : len * abpt->gap_ext1 + abpt->gap_open1;
                bits = (max_score <= INT16_MAX - abpt->min_mis - oe1 - oe2) ? 16 : 32;
                int pn = bits == 16 ? 16 : 8, P = abpt->gap_mode == ABPOA_LINEAR_GAP ? 1 : (abpt->gap_mode == ABPOA_AFFINE_GAP ? 3 : 5);
                int64_t dp_sn = (qlen + pn) / pn;
                put_i32(fp, "bits", bits); put_i32(fp, "n_planes", P);
                put_i32(fp, "best_score", res.best_score);
                put_i32(fp, "node_s", res.node_s); put_i32(fp, "node_e", res.node_e);
                put_i32(fp, "query_s", res.query_s); put_i32(fp, "query_e", res.query_e);
                put_i32(fp, "n_aln_bases", res.n_aln_bases); put_i32(fp, "n_matched_bases", res.n_matched_bases);
                put(fp, "cigar", 5, res.n_cigar, res.graph_cigar);
                abpoa_simd_matrix_t *abm = ab->abm;
                /* rows never computed (inactive) keep stale dp_beg/dp_end: mask them out */
                int32_t *dbeg = malloc(gn * 4), *dend = malloc(gn * 4), *dbsn = malloc(gn * 4), *desn = malloc(gn * 4);
                int64_t *roff = malloc((gn + 1) * 8); uint64_t *rsum = calloc(gn, 8); int64_t tot = 0, cells = 0;
                uint8_t *act = calloc(gn, 1);
                { uint8_t *index_map = calloc(g->node_n, 1); index_map[beg_index] = index_map[end_index] = 1;
                  for (int i = beg_index; i < end_index - 1; ++i) { if (!index_map[i]) continue; int id = g->index_to_node_id[i];
                      for (int j = 0; j < g->node[id].out_edge_n; ++j) index_map[g->node_id_to_index[g->node[id].out_id[j]]] = 1; }
                  for (int r = 0; r < gn - 1; ++r) act[r] = index_map[beg_index + r]; free(index_map); }
                for (int r = 0; r < gn; ++r) {
                    roff[r] = tot;
                    if (r == gn - 1 || !act[r] || abm->dp_beg_sn[r] < 0) { dbeg[r] = dend[r] = dbsn[r] = desn[r] = -1; continue; }
                    dbeg[r] = abm->dp_beg[r]; dend[r] = abm->dp_end[r]; dbsn[r] = abm->dp_beg_sn[r]; desn[r] = abm->dp_end_sn[r];
                    int64_t wv = (int64_t)(desn[r] - dbsn[r] + 1) * pn; tot += wv * P; if (r > 0) cells += wv;
                }
                roff[gn] = tot;
                put(fp, "dp_beg", 1, gn, dbeg); put(fp, "dp_end", 1, gn, dend); put(fp, "dp_beg_sn", 1, gn, dbsn); put(fp, "dp_end_sn", 1, gn, desn);
                put(fp, "row_off", 2, gn + 1, roff); put(fp, "n_cells", 2, 1, &cells);
                const char *dp0 = (const char *)abm->s_mem + dp_sn * abpt->m * 32;       /* skip the query profile */
                void *pl = dump_planes ? malloc((size_t)(tot > 0 ? tot : 1) * (bits / 8)) : NULL;
                for (int r = 0; r < gn; ++r) {
                    if (dbsn[r] < 0) continue;
                    int64_t wv = (int64_t)(desn[r] - dbsn[r] + 1) * pn; uint64_t acc = 0;
                    for (int p = 0; p < P; ++p) {
                        const char *src = dp0 + (((int64_t)r * P + p) * dp_sn + dbsn[r]) * 32;
                        for (int64_t x = 0; x < wv; ++x) {
                            int32_t v = bits == 16 ? ((const int16_t *)src)[x] : ((const int32_t *)src)[x];
                            acc += (uint64_t)(uint32_t)v * mix64((uint64_t)(dbsn[r] * pn + x) * 8 + p);
                            if (pl) { if (bits == 16) ((int16_t *)pl)[roff[r] + p * wv + x] = (int16_t)v; else ((int32_t *)pl)[roff[r] + p * wv + x] = v; }
                        }
                    }
                    rsum[r] = acc;
                }
                put(fp, "row_checksum", 5, gn, rsum);
                if (pl) put(fp, "planes", bits == 16 ? 4 : 1, tot, pl);
                /* post-call band state */
                int32_t *left = malloc(gn * 4), *right = malloc(gn * 4);
                for (int r = 0; r < gn; ++r) { int id = g->index_to_node_id[beg_index + r];
                    left[r] = g->node_id_to_max_pos_left ? g->node_id_to_max_pos_left[id] : 0; right[r] = g->node_id_to_max_pos_right ? g->node_id_to_max_pos_right[id] : 0; }
                put(fp, "left_out", 1, gn, left); put(fp, "right_out", 1, gn, right);
                free(left); free(right); free(dbeg); free(dend); free(dbsn); free(desn); free(roff); free(rsum); free(act); free(pl);
                fclose(fp);
            }
        }
        if (sub_beg >= 0 && sel[ri] && ab->abg->node_n > 2) {   /* the dump was a sub-graph alignment: redo the full one */
            if (res.n_cigar) free(res.graph_cigar);
            memset(&res, 0, sizeof(res));
            abpoa_topological_sort(ab->abg, abpt);
            simd_abpoa_align_sequence_to_graph(ab, abpt, query, qlen, &res);
        }
        abpoa_add_graph_alignment(ab, abpt, query, NULL, qlen, NULL, res, ri, rd.n, 1);
        if (res.n_cigar) free(res.graph_cigar);
    }
    { char fn[1024]; snprintf(fn, sizeof(fn), "%s/output.txt", outdir); FILE *fo = fopen(fn, "w"); abpoa_output(ab, abpt, fo); fclose(fo); }
    abpoa_free(ab); abpoa_free_para(abpt);
    return 0;
}
