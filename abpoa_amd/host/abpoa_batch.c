/*
 * abpoa_batch -- C front end of the MI355X engine's read-set batch entry (include/abpoa_hip.h, abpoa_hip_msa_batch).
 *
 * Same command line as the reference's `abpoa` for what the engine covers (option letters and defaults of
 * src/abpoa.c:22-59, :148-214 and abpoa_init_para, src/abpoa_align.c:93-141), same output text
 * (abpoa_output_fx_consensus, src/abpoa_output.c:495-512; abpoa_output_rc_msa, :70-101).  The one difference is the
 * reason it exists: with -l the reference runs the files of the list one after the other through one abpoa_t
 * (src/abpoa.c:131-141); here every file of the list is one read-set and a PIECE of the list (--piece, default 2048 files) goes to the GPU in
 * one call.  Like the reference's loop the list is STREAMED: reader threads (--readers, default 8) parse and encode the files of piece k + 1
 * while abpoa_hip_msa_batch runs piece k, so memory is two pieces whatever the length of the list (BASELINE.json configs[3]: 100 k files of
 * 50 x 10 kb reads = 50 GB of bases), and the output comes in list order.
 *
 *   abpoa_batch [options] <in.fa|in.fq|list.txt>      (plain or gzip'ed; FASTA / FASTQ)
 *     -m INT   0 global, 1 local, 2 extension          -M INT match [2]      -X INT mismatch [4]      -t FILE score matrix
 *     -O INT[,INT] gap open [4,24]   -E INT[,INT] gap extension [2,1]   -b INT [10] / -f FLOAT [0.01] adaptive band (b < 0: off)
 *     -z INT z-drop of extension mode [-1: off]   -e INT end bonus (accepted; as in the reference, nothing in the DP reads it)
 *     -c amino acids   -l the input is a list of files   -o FILE output [stdout]   -r INT 0 consensus, 1 MSA, 2 both, 5 consensus as FASTQ
 *     -s ambiguous strand   -Q base qualities as edge weights   -T INT host threads [all]   -v version
 *     --piece INT files per GPU call with -l [2048]   --readers INT reader threads [8]      (no reference counterpart)
 * Degenerate inputs as the reference treats them: a file without records prints nothing; a record without bases after the first one is an MSA row
 * of gaps and adds nothing to the graph; a first record without bases ends the run as abpoa_add_graph_sequence does (src/abpoa_graph.c:487).
 * Options of the reference that the engine does not cover (-S -k -w -n -p -i -g -d -q, -r 3/4) are refused, not ignored.
 *
 * Plain C99 + zlib; links against libabpoa_hip.so only.  Own code throughout: no klib / kseq.
 */
#include <getopt.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>
#include "abpoa_hip.h"

#define DIE(...) do { fprintf(stderr, "abpoa_batch: " __VA_ARGS__); fputc('\n', stderr); exit(1); } while (0)

/* ---- alphabets: codes and letters of the reference's tables (src/abpoa_seq.c:15-95) */
static uint8_t nt_code[256], aa_code[256];
static const char nt_letter[] = "ACGTN-", aa_letter[] = "ACGTNBDEFHIJKLMOPQRSUVWXYZ*-";
static void init_tables(void) {
    memset(nt_code, 4, sizeof nt_code); memset(aa_code, 26, sizeof aa_code);
    const char *nt = "ACGT";
    for (int i = 0; i < 4; ++i) { nt_code[(uint8_t)nt[i]] = (uint8_t)i; nt_code[(uint8_t)(nt[i] + 32)] = (uint8_t)i; nt_code[i] = (uint8_t)i; }
    nt_code['U'] = nt_code['u'] = 3;
    for (int i = 0; i < 26; ++i) { aa_code[(uint8_t)aa_letter[i]] = (uint8_t)i; if (aa_letter[i] >= 'A' && aa_letter[i] <= 'Z') aa_code[(uint8_t)(aa_letter[i] + 32)] = (uint8_t)i; }
    for (int i = 0; i < 27; ++i) aa_code[i] = (uint8_t)i;
}

/* ---- growable byte / pointer arrays */
typedef struct { char *s; size_t n, cap; } str_t;
static void str_push(str_t *a, const char *p, size_t k) {
    if (a->n + k + 1 > a->cap) { a->cap = (a->n + k + 1) * 2 + 64; a->s = (char *)realloc(a->s, a->cap); if (!a->s) DIE("out of memory"); }
    memcpy(a->s + a->n, p, k); a->n += k; a->s[a->n] = 0;
}

/* one input file = one read-set */
typedef struct {
    int n, cap;
    char **name; uint8_t **code; int32_t *len; int32_t **weight;      /* weight: NULL unless -Q */
} readset_t;
static void rs_add(readset_t *r, const char *name, const str_t *seq, const str_t *qual, const uint8_t *tbl, int use_qv) {
    if (r->n == r->cap) {
        r->cap = r->cap ? 2 * r->cap : 64;
        r->name = (char **)realloc(r->name, sizeof(char *) * r->cap); r->code = (uint8_t **)realloc(r->code, sizeof(uint8_t *) * r->cap);
        r->len = (int32_t *)realloc(r->len, sizeof(int32_t) * r->cap); r->weight = (int32_t **)realloc(r->weight, sizeof(int32_t *) * r->cap);
        if (!r->name || !r->code || !r->len || !r->weight) DIE("out of memory");
    }
    const int i = r->n++, L = (int)seq->n;
    r->name[i] = strdup(name); r->len[i] = L;
    r->code[i] = (uint8_t *)malloc(L > 0 ? L : 1);
    for (int j = 0; j < L; ++j) r->code[i][j] = tbl[(uint8_t)seq->s[j]];
    r->weight[i] = NULL;
    if (use_qv) {      /* reference src/abpoa_align.c:462-467: quality - 32 where the record has as many qualities as bases, else 1 */
        r->weight[i] = (int32_t *)malloc(sizeof(int32_t) * (L > 0 ? L : 1));
        const int has_q = qual->n == seq->n && L > 0;
        for (int j = 0; j < L; ++j) r->weight[i][j] = has_q ? (int32_t)(uint8_t)qual->s[j] - 32 : 1;
    }
}

/* FASTA / FASTQ records of a (possibly gzip'ed) file: header line '>' or '@' (name = text up to the first blank), sequence lines up to the next
 * header or a '+' line, then -- FASTQ -- quality lines until as many characters as bases */
static void read_file(const char *fn, readset_t *r, const uint8_t *tbl, int use_qv) {
    gzFile fp = gzopen(fn, "r");
    if (!fp) DIE("cannot open %s", fn);
    gzbuffer(fp, 1 << 20);
    enum { LINE_CAP = 1 << 16 };
    char *line = (char *)malloc(LINE_CAP);      /* (own buffer: files are read by several threads) */
    if (!line) DIE("out of memory");
    str_t seq = {0, 0, 0}, qual = {0, 0, 0}, name = {0, 0, 0};
    int have = 0, in_qual = 0, cont = 0;                             /* cont: the previous piece did not end its line */
    str_push(&seq, "", 0); str_push(&qual, "", 0); str_push(&name, "", 0);
    for (;;) {
        char *got = gzgets(fp, line, LINE_CAP);
        size_t k = got ? strlen(line) : 0;
        const int whole = got && k > 0 && line[k - 1] == '\n';      /* (a line longer than the buffer comes in pieces) */
        while (k > 0 && (line[k - 1] == '\n' || line[k - 1] == '\r')) line[--k] = 0;
        const int is_head = got && !cont && !in_qual && (line[0] == '>' || line[0] == '@');
        if (!got || is_head) {
            if (have) rs_add(r, name.s, &seq, &qual, tbl, use_qv);
            if (!got) break;
            size_t e = 1; while (e < k && line[e] != ' ' && line[e] != '\t') ++e;
            name.n = 0; str_push(&name, line + 1, e - 1);
            seq.n = 0; seq.s[0] = 0; qual.n = 0; qual.s[0] = 0; have = 1; in_qual = 0;
        } else if (have && !cont && !in_qual && line[0] == '+') in_qual = 1;
        else if (have && in_qual) { str_push(&qual, line, k); if (qual.n >= seq.n) in_qual = 0; }
        else if (have) {      /* sequence text: whole runs between blanks (a 10 kb read is one line: one copy) */
            size_t j = 0;
            while (j < k) {
                while (j < k && (line[j] == ' ' || line[j] == '\t')) ++j;
                size_t e = j; while (e < k && line[e] != ' ' && line[e] != '\t') ++e;
                if (e > j) str_push(&seq, line + j, e - j);
                j = e;
            }
        }
        cont = !whole;
    }
    gzclose(fp); free(seq.s); free(qual.s); free(name.s); free(line);
}

/* ---- score matrix: abpoa_set_mat_from_file (src/abpoa_align.c:34-85) / gen_simple_mat (:12-25) */
static void simple_matrix(int m, int match, int mismatch, int32_t *mat, int *max_mat, int *min_mis) {
    match = abs(match); mismatch = -abs(mismatch);
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) mat[i * m + j] = (i == m - 1 || j == m - 1) ? 0 : (i == j ? match : mismatch);
    *max_mat = match; *min_mis = -mismatch;
}
static void matrix_from_file(const char *fn, int m, const uint8_t *tbl, int32_t *mat, int *max_mat, int *min_mis) {
    FILE *f = fopen(fn, "r");
    if (!f) DIE("cannot open matrix file %s", fn);
    char line[4096]; int order[64], n_order = -1;
    memset(mat, 0, sizeof(int32_t) * m * m);
    while (fgets(line, sizeof line, f)) {
        if (line[0] == '#') continue;
        if (n_order < 0) { n_order = 0; for (char *p = line; *p; ++p) if (*p != ' ' && *p != '\t' && *p != '\n' && *p != '\r' && n_order < 64) order[n_order++] = tbl[(uint8_t)*p]; continue; }
        char *p = line; while (*p == ' ' || *p == '\t') ++p;
        if (!*p || *p == '\n' || *p == '\r') continue;
        const int row = tbl[(uint8_t)*p];
        if (row >= m) DIE("unknown residue '%c' in %s", *p, fn);
        while (*p && *p != ' ' && *p != '\t') ++p;
        for (int n = 0;; ++n) {
            char *e; const long v = strtol(p, &e, 10);
            if (e == p) break;
            if (n >= m || n >= n_order) DIE("too many scores in a row of %s", fn);
            mat[row * m + order[n]] = (int32_t)v; p = e;
        }
    }
    fclose(f);
    int mx = 0, mn = 0;
    for (int i = 0; i < m * m; ++i) { if (mat[i] > mx) mx = mat[i]; if (-mat[i] > mn) mn = -mat[i]; }
    *max_mat = mx; *min_mis = mn;
}

static const struct option long_opt[] = {
    {"aln-mode", 1, NULL, 'm'}, {"match", 1, NULL, 'M'}, {"mismatch", 1, NULL, 'X'}, {"matrix", 1, NULL, 't'}, {"gap-open", 1, NULL, 'O'}, {"gap-ext", 1, NULL, 'E'},
    {"extra-b", 1, NULL, 'b'}, {"extra-f", 1, NULL, 'f'}, {"zdrop", 1, NULL, 'z'}, {"bonus", 1, NULL, 'e'}, {"seeding", 0, NULL, 'S'}, {"k-mer", 1, NULL, 'k'},
    {"window", 1, NULL, 'w'}, {"min-poa-win", 1, NULL, 'n'}, {"progressive", 0, NULL, 'p'}, {"use-qual-weight", 0, NULL, 'Q'}, {"amino-acid", 0, NULL, 'c'},
    {"in-list", 0, NULL, 'l'}, {"increment", 1, NULL, 'i'}, {"amb-strand", 0, NULL, 's'}, {"output", 1, NULL, 'o'}, {"result", 1, NULL, 'r'}, {"out-pog", 1, NULL, 'g'},
    {"max-num-cons", 1, NULL, 'd'}, {"min-freq", 1, NULL, 'q'}, {"threads", 1, NULL, 'T'}, {"help", 0, NULL, 'h'}, {"version", 0, NULL, 'v'},
    {"piece", 1, NULL, 1001}, {"readers", 1, NULL, 1002}, {0, 0, 0, 0}};

static void rs_free(readset_t *r) {
    for (int i = 0; i < r->n; ++i) { free(r->name[i]); free(r->code[i]); free(r->weight[i]); }
    free(r->name); free(r->code); free(r->len); free(r->weight); memset(r, 0, sizeof *r);
}

/* one output line of `n` residue codes (a 10 kb consensus per file, thousands of files per piece: one fwrite each, not a putchar per base) */
static void put_codes(const uint8_t *row, int n, const char *letter) {
    static char *buf = NULL; static int cap = 0;
    if (n + 1 > cap) { cap = 2 * (n + 1) + 1024; buf = (char *)realloc(buf, (size_t)cap); if (!buf) DIE("out of memory"); }
    for (int j = 0; j < n; ++j) buf[j] = letter[row[j]];
    buf[n] = '\n';
    fwrite(buf, 1, (size_t)n + 1, stdout);
}

/* ---- a piece of the list: files [lo, hi) parsed and encoded by `readers` threads (one file at a time each, off a shared counter) */
typedef struct { char **files; int n; readset_t *rs; const uint8_t *tbl; int use_qv, readers; pthread_mutex_t mu; int next; pthread_t th; int started; } piece_t;
static void *piece_worker(void *arg) {
    piece_t *p = (piece_t *)arg;
    for (;;) {
        pthread_mutex_lock(&p->mu); const int i = p->next++; pthread_mutex_unlock(&p->mu);
        if (i >= p->n) return NULL;
        read_file(p->files[i], &p->rs[i], p->tbl, p->use_qv);
    }
}
static void *piece_main(void *arg) {
    piece_t *p = (piece_t *)arg; pthread_t th[64]; const int nt = p->readers < 1 ? 1 : (p->readers > 64 ? 64 : p->readers);
    int made = 0;
    for (int t = 1; t < nt && t < p->n; ++t) if (pthread_create(&th[made], NULL, piece_worker, p) == 0) ++made;
    piece_worker(p);
    for (int t = 0; t < made; ++t) pthread_join(th[t], NULL);
    return NULL;
}
/* take the next `want` file names of the list (or the single input) and start reading them in the background */
static piece_t *piece_start(FILE *lf, const char *single, int *single_done, int want, const uint8_t *tbl, int use_qv, int readers) {
    piece_t *p = (piece_t *)calloc(1, sizeof *p);
    p->files = (char **)calloc(want > 0 ? want : 1, sizeof(char *)); p->tbl = tbl; p->use_qv = use_qv; p->readers = readers;
    char fn[4096];
    while (p->n < want) {
        if (lf) { if (!fgets(fn, sizeof fn, lf)) break; size_t k = strlen(fn); while (k > 0 && (fn[k - 1] == '\n' || fn[k - 1] == '\r')) fn[--k] = 0; if (k == 0) continue; }
        else { if (*single_done) break; *single_done = 1; snprintf(fn, sizeof fn, "%s", single); }
        p->files[p->n++] = strdup(fn);
    }
    if (p->n == 0) { free(p->files); free(p); return NULL; }
    p->rs = (readset_t *)calloc(p->n, sizeof(readset_t));
    pthread_mutex_init(&p->mu, NULL);
    if (pthread_create(&p->th, NULL, piece_main, p) == 0) p->started = 1; else piece_main(p);
    return p;
}
static void piece_wait(piece_t *p) { if (p->started) { pthread_join(p->th, NULL); p->started = 0; } }
static void piece_free(piece_t *p) {
    for (int i = 0; i < p->n; ++i) { rs_free(&p->rs[i]); free(p->files[i]); }
    free(p->rs); free(p->files); pthread_mutex_destroy(&p->mu); free(p);
}

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

int main(int argc, char **argv) {
    const int timing = getenv("ABPOA_BATCH_TIMING") != NULL;      /* per piece on stderr: wait for the readers, the batch call, output */
    const double t_start = now_s();
    int mode = 0, match = 2, mismatch = 4, o1 = 4, o2 = 24, e1 = 2, e2 = 1, wb = 10, m = 5, in_list = 0, out_cons = 1, out_msa = 0, amb = 0, use_qv = 0, threads = 0, c;
    int piece_sets = 2048, readers = 8, zdrop = -1, out_fq = 0;
    float wf = 0.01f; const char *mat_fn = NULL; char *s;
    while ((c = getopt_long(argc, argv, "m:M:X:t:O:E:b:f:z:e:QSk:w:n:i:clpso:r:g:d:q:T:hvV:", long_opt, NULL)) >= 0) {
        switch (c) {
            case 'm': mode = atoi(optarg); if (mode < 0 || mode > 2) DIE("unknown alignment mode: %d", mode); break;
            case 'M': match = atoi(optarg); break;
            case 'X': mismatch = atoi(optarg); break;
            case 't': mat_fn = optarg; break;
            case 'O': o1 = (int)strtol(optarg, &s, 10); if (*s == ',') o2 = (int)strtol(s + 1, &s, 10); break;
            case 'E': e1 = (int)strtol(optarg, &s, 10); if (*s == ',') e2 = (int)strtol(s + 1, &s, 10); break;
            case 'b': wb = atoi(optarg); break;
            case 'f': wf = (float)atof(optarg); break;
            case 'z': zdrop = atoi(optarg); break;      /* extension mode: the row loop ends when the row maximum falls this far below the best one (ref src/simd_abpoa_align.c:1018-1026) */
            case 'e': break;                            /* end bonus: parsed by the reference, read by nothing in its DP (ref src/simd_abpoa_align.c:1069 "TODO") */
            case 'Q': use_qv = 1; break;
            case 'c': m = 27; break;
            case 'l': in_list = 1; break;
            case 's': amb = 1; break;
            case 'T': threads = atoi(optarg); break;
            case 1001: piece_sets = atoi(optarg); if (piece_sets < 1) DIE("--piece must be at least 1"); break;
            case 1002: readers = atoi(optarg); break;
            case 'o': if (strcmp(optarg, "-") != 0 && freopen(optarg, "wb", stdout) == NULL) DIE("failed to open the output file %s", optarg); break;
            case 'r': { const int r = atoi(optarg); if (r == 0) { out_cons = 1; out_msa = 0; } else if (r == 1) { out_cons = 0; out_msa = 1; } else if (r == 2) { out_cons = out_msa = 1; }
                        else if (r == 5) { out_cons = 1; out_msa = 0; out_fq = 1; }      /* consensus as FASTQ: a quality per base from its coverage */
                        else { fprintf(stderr, "abpoa_batch: -r %d (GFA output) is outside this engine\n", r); return 2; } } break;
            case 'v': printf("abpoa_batch (MI355X engine; output of abPOA 1.4.1)\n"); return 0;
            case 'V': break;
            case 'h': fprintf(stderr, "usage: abpoa_batch [-m -M -X -t -O -E -b -f -z -e -c -l -o -r -s -Q -T --piece --readers] <in.fa|in.fq|list.txt>   (see the head of abpoa_batch.c)\n"); return 1;
            default: fprintf(stderr, "abpoa_batch: option -%c is outside this engine (seeding, guide tree, incremental graphs, plots, multiple consensus)\n", c); return 2;
        }
    }
    if (argc - optind != 1) { fprintf(stderr, "usage: abpoa_batch [options] <in.fa|in.fq|list.txt>\n"); return 1; }
    init_tables();
    const uint8_t *tbl = m > 5 ? aa_code : nt_code; const char *letter = m > 5 ? aa_letter : nt_letter;
    const char gap_letter = letter[m];      /* '-' : the code after the alphabet's last residue */

    /* ---- parameters: abpoa_post_set_para (src/abpoa_align.c:143-168) */
    abpoa_hip_scoring_t sc; memset(&sc, 0, sizeof sc);
    int32_t *mat = (int32_t *)malloc(sizeof(int32_t) * m * m); int max_mat, min_mis;
    if (mat_fn) matrix_from_file(mat_fn, m, tbl, mat, &max_mat, &min_mis); else simple_matrix(m, match, mismatch, mat, &max_mat, &min_mis);
    sc.m = m; sc.mat = mat; sc.max_mat = max_mat; sc.min_mis = min_mis;
    sc.gap_open1 = o1; sc.gap_ext1 = e1; sc.gap_open2 = o2; sc.gap_ext2 = e2;
    sc.align_mode = mode; sc.gap_mode = o1 == 0 ? ABPOA_HIP_LINEAR_GAP : ((o1 > 0 && o2 == 0) ? ABPOA_HIP_AFFINE_GAP : ABPOA_HIP_CONVEX_GAP);
    sc.wb = mode == ABPOA_HIP_LOCAL_MODE ? -1 : wb; sc.wf = wf; sc.zdrop = zdrop; sc.ret_cigar = 1; sc.rev_cigar = 0;
    const unsigned flags = (out_cons ? ABPOA_HIP_OUT_CONS : 0u) | (out_msa ? ABPOA_HIP_OUT_MSA : 0u) | (amb ? ABPOA_HIP_AMB_STRAND : 0u);

    FILE *lf = in_list ? fopen(argv[optind], "r") : NULL;
    if (in_list && !lf) DIE("cannot open list %s", argv[optind]);
    int single_done = 0, gpu_up = 0;
    long file_no = 0;
    /* A record without a name: the reference prints ">Seq_<i>" -- unless an earlier file of the list had a named record at the same position: its abpoa_seq_t
     * lives across the files of a list and abpoa_cpy_str leaves a string alone when the new one is empty (src/abpoa_seq.c:123-130), so the old name shows
     * (src/abpoa_output.c:75-81).  Reproduced here: the last non-empty name seen at each record position (own copies: they outlive the pieces). */
    int sticky_n = 0; char **sticky = NULL;

    piece_t *next = piece_start(lf, argv[optind], &single_done, in_list ? piece_sets : 1, tbl, use_qv, readers);
    if (next && in_list) {      /* the GPU comes up while the first piece is being read (a single input keeps the lazy start: an empty file needs no GPU) */
        const int rc0 = abpoa_hip_init(0); if (rc0 != ABPOA_HIP_OK) DIE("no usable GPU (%d): %s", rc0, abpoa_hip_last_error()); gpu_up = 1;
    }
    while (next) {
        piece_t *cur = next;
        const double t0 = now_s();
        piece_wait(cur);
        const double t1 = now_s();
        next = piece_start(lf, argv[optind], &single_done, piece_sets, tbl, use_qv, readers);      /* ... is read while the GPU works on `cur` */

        /* ---- the sets of the call: files with records; within a set the records that have bases (kept[]: index in the file) */
        abpoa_hip_readset_t *sets = (abpoa_hip_readset_t *)calloc(cur->n, sizeof *sets);
        abpoa_hip_msa_t *out = (abpoa_hip_msa_t *)calloc(cur->n, sizeof *out);
        int *set_of = (int *)malloc(sizeof(int) * cur->n), **kept = (int **)calloc(cur->n, sizeof(int *)), *n_kept = (int *)calloc(cur->n, sizeof(int));
        const uint8_t ***seqv = (const uint8_t ***)calloc(cur->n, sizeof(*seqv)); const int32_t ***wgtv = (const int32_t ***)calloc(cur->n, sizeof(*wgtv)); int32_t **lenv = (int32_t **)calloc(cur->n, sizeof(*lenv));
        int n_call = 0, fatal_at = -1;
        for (int i = 0; i < cur->n; ++i) {
            const readset_t *r = &cur->rs[i]; set_of[i] = -1;
            if (r->n == 0) continue;                                     /* no records: the reference prints nothing for the file */
            if (r->len[0] == 0) { fatal_at = i; break; }                 /* the reference dies in abpoa_add_graph_sequence; the files before it were printed */
            kept[i] = (int *)malloc(sizeof(int) * r->n); seqv[i] = (const uint8_t **)malloc(sizeof(uint8_t *) * r->n); lenv[i] = (int32_t *)malloc(sizeof(int32_t) * r->n);
            wgtv[i] = (const int32_t **)malloc(sizeof(int32_t *) * r->n);
            for (int q = 0; q < r->n; ++q) if (r->len[q] > 0) { const int k = n_kept[i]++; kept[i][k] = q; seqv[i][k] = r->code[q]; lenv[i][k] = r->len[q]; wgtv[i][k] = r->weight[q]; }
            set_of[i] = n_call;
            sets[n_call].n_reads = n_kept[i]; sets[n_call].seqs = seqv[i]; sets[n_call].lens = lenv[i]; sets[n_call].weights = use_qv ? wgtv[i] : NULL;
            ++n_call;
        }
        if (n_call > 0) {
            if (!gpu_up) { const int rc0 = abpoa_hip_init(0); if (rc0 != ABPOA_HIP_OK) DIE("no usable GPU (%d): %s", rc0, abpoa_hip_last_error()); gpu_up = 1; }
            const int rc = abpoa_hip_msa_batch(&sc, n_call, sets, out, flags, threads);
            if (rc != ABPOA_HIP_OK) DIE("abpoa_hip_msa_batch failed (%d): %s", rc, abpoa_hip_last_error());
        }
        const double t2 = now_s();

        /* ---- output, file by file as the reference prints it (src/abpoa_align.c:346-371: MSA when asked for -- with the consensus row if both --, else consensus) */
        const int n_print = fatal_at >= 0 ? fatal_at : cur->n;
        for (int i = 0; i < n_print; ++i) {
            readset_t *r = &cur->rs[i]; ++file_no;
            if (r->n > sticky_n) { sticky = (char **)realloc(sticky, sizeof(char *) * r->n); for (int q = sticky_n; q < r->n; ++q) sticky[q] = NULL; sticky_n = r->n; }
            for (int q = 0; q < r->n; ++q) {
                if (r->name[q][0]) { free(sticky[q]); sticky[q] = strdup(r->name[q]); }
                else if (sticky[q]) { free(r->name[q]); r->name[q] = strdup(sticky[q]); }
            }
            if (set_of[i] < 0) continue;
            const abpoa_hip_msa_t *o = &out[set_of[i]];
            if (o->status != ABPOA_HIP_OK) DIE("alignment failed for input %ld (status %d)", file_no, o->status);
            if (out_msa) {
                if (o->msa_len <= 0) continue;
                int k = 0;                                               /* next kept record */
                for (int q = 0; q < r->n; ++q) {
                    const int has = k < n_kept[i] && kept[i][k] == q;
                    if (r->name[q][0]) printf(">%s%s\n", r->name[q], (has && o->is_rc && o->is_rc[k]) ? "_reverse_complement" : ""); else printf(">Seq_%d\n", q + 1);
                    if (has) { put_codes(o->msa_base + (size_t)k * o->msa_len, o->msa_len, letter); ++k; }
                    else { for (int j = 0; j < o->msa_len; ++j) putchar(gap_letter); putchar('\n'); }      /* a record without bases: a row of gaps */
                }
                if (out_cons) {
                    printf(">Consensus_sequence\n");
                    put_codes(o->msa_base + (size_t)n_kept[i] * o->msa_len, o->msa_len, letter);
                }
            } else if (out_cons && o->cons_len > 0) {                    /* (abpoa_output_fx_consensus prints nothing without a consensus) */
                printf(out_fq ? "@Consensus_sequence\n" : ">Consensus_sequence\n");
                put_codes(o->cons_base, o->cons_len, letter);
                if (out_fq) {      /* -r 5 (ref src/abpoa_output.c:270-276, :516-525): phred + 33 of a logistic in the share of the reads that pass through the base */
                    printf("+Consensus_sequence\n");
                    for (int j = 0; j < o->cons_len; ++j) {
                        const double x = 13.8 * (1.25 * o->cons_cov[j] / r->n - 0.25), pe = 1 - 1.0 / (1.0 + pow(2.718281828459045, -1 * x));
                        putchar(33 + (int)(-10 * log10(pe) + 0.499));
                    }
                    putchar('\n');
                }
            }
        }
        if (timing) fprintf(stderr, "[abpoa_batch] piece of %d files at %.2f s: waited %.2f s for the readers, batch call %.2f s, output %.2f s\n", cur->n, t0 - t_start, t1 - t0, t2 - t1,
                            now_s() - t2);
        abpoa_hip_free_msa_array(out, n_call);
        for (int i = 0; i < cur->n; ++i) { free(kept[i]); free((void *)seqv[i]); free((void *)wgtv[i]); free(lenv[i]); }
        free(kept); free(n_kept); free((void *)seqv); free((void *)wgtv); free(lenv); free(set_of); free(sets); free(out);
        if (fatal_at >= 0) {
            fflush(stdout);
            fprintf(stderr, "abpoa_batch: %s: the first record has no bases ([abpoa_add_graph_sequence] seq_l: 0\tstart: 0\tend: 0.)\n", cur->files[fatal_at]);
            if (next) { piece_wait(next); piece_free(next); }
            piece_free(cur);
            return 1;
        }
        piece_free(cur);
    }
    if (lf) fclose(lf);
    for (int q = 0; q < sticky_n; ++q) free(sticky[q]);
    free(sticky); free(mat);
    fflush(stdout);
    /* every result is out: leave without tearing the engine down piece by piece (abpoa_hip_shutdown would hipFree ~150 GB of pools one by one and the HIP
     * runtime's exit handlers would follow; the driver reclaims a dead process's memory in one go: ~0.4 s of a 10 s job) */
    if (gpu_up) _exit(ferror(stdout) ? 1 : 0);
    return 0;
}
