/*
 * abpoa_hip.h -- C-ABI of the MI355X (gfx950) adaptive-banded sequence-to-graph DP engine.
 *
 * This library replaces ONE path of Xinglab/abPOA v1.4.1: the banded partial-order-alignment
 * dynamic programme of src/simd_abpoa_align.c (+ its SIMD dispatch header src/simd_instruction.h
 * and the inner driver in src/abpoa_align.c:178-190).  Everything here is plain C: pointers and
 * sizes, no C++ / torch types.  Three layers of entry points, narrowest first:
 *
 *  (1) FLAT BATCH API  (abpoa_hip_align_batch)          -- what a foreign-function binding targets.
 *      One call aligns N independent (graph snapshot, query) problems on the GPU.  A problem is the
 *      information the reference DP reads through abpoa_graph_t accessors, flattened into arrays
 *      indexed by DP row (row = topological index - beg_index, reference: simd_abpoa_align.c:1648).
 *
 *  (2) SEAM API  (simd_abpoa_align_sequence_to_graph & co, declared in abpoa_seam.h)
 *      Byte-compatible replacements for the 4 symbols of src/simd_abpoa_align.h:11-14, so the stock
 *      reference host code (graph, consensus, CLI, pyabpoa) links against the GPU DP unchanged.
 *
 *  (3) READ-SET BATCH API (abpoa_hip_msa_batch)         -- additive, no reference counterpart:
 *      progressive POA of many independent read-sets in lock-step rounds (SURVEY.md 8b "new entry").
 *
 * Error behaviour: the reference has no error returns on this path (fatal -> exit(1),
 * src/utils.c:91-116).  The flat API returns 0 on success and a negative ABPOA_HIP_E* code
 * otherwise; per-problem failures are reported in abpoa_hip_result_t.status.  There is NO CPU
 * fallback: without a usable GPU every entry point fails with ABPOA_HIP_ENODEV.
 */
#ifndef ABPOA_HIP_H
#define ABPOA_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* align_mode / gap_mode values are the reference's (src/abpoa.h:7-15). */
#define ABPOA_HIP_GLOBAL_MODE 0
#define ABPOA_HIP_LOCAL_MODE  1
#define ABPOA_HIP_EXTEND_MODE 2
#define ABPOA_HIP_LINEAR_GAP  0
#define ABPOA_HIP_AFFINE_GAP  1
#define ABPOA_HIP_CONVEX_GAP  2

/* cigar op codes in the low 4 bits of a cigar word (src/abpoa.h:20-26). */
#define ABPOA_HIP_CMATCH 0
#define ABPOA_HIP_CINS   1
#define ABPOA_HIP_CDEL   2

/* error codes */
#define ABPOA_HIP_OK        0
#define ABPOA_HIP_ENODEV   -1   /* no HIP device / runtime failure at init            */
#define ABPOA_HIP_EINVAL   -2   /* malformed problem (checked on the host before launch) */
#define ABPOA_HIP_ENOMEM   -3   /* device or host allocation failed                     */
#define ABPOA_HIP_ELAUNCH  -4   /* kernel launch / execution error                      */
#define ABPOA_HIP_EBACKTRACK -5 /* dead end in backtrack (reference: err_fatal "Error in *_backtrack") */
#define ABPOA_HIP_ESTRICT  -6   /* ABPOA_HIP_STRICT=1 and some read-set would have left the device-resident driver for the host driver */

/* The scoring / mode fields the DP reads from abpoa_para_t (src/abpoa.h:62-81). */
typedef struct abpoa_hip_scoring_t {
    int32_t m;              /* alphabet size: 5 (nt) or 27 (aa)                                       */
    const int32_t *mat;     /* m*m scores, mat[m*graph_base + query_code] (abpoa_para_t.mat)          */
    int32_t max_mat;        /* largest entry  (feeds the int16/int32 decision, simd_abpoa_align.c:1673) */
    int32_t min_mis;        /* -(smallest entry), >= 0                                                */
    int32_t gap_open1, gap_ext1, gap_open2, gap_ext2;
    int32_t align_mode;     /* ABPOA_HIP_{GLOBAL,LOCAL,EXTEND}_MODE                                   */
    int32_t gap_mode;       /* ABPOA_HIP_{LINEAR,AFFINE,CONVEX}_GAP                                   */
    int32_t wb;             /* extra band width b; < 0 disables banding (abpoa_para_t.wb)             */
    float   wf;             /* extra band width f: w = wb + (int)(wf*qlen), float32 product            */
    int32_t zdrop;          /* <= 0: disabled (extend mode only)                                      */
    int32_t ret_cigar;      /* 0: score only                                                          */
    int32_t rev_cigar;      /* 1: leave the cigar in backtrack order (abpoa_para_t.rev_cigar)         */
} abpoa_hip_scoring_t;

/* One (sub)graph snapshot + query.  Row r is the node with topological index beg_index + r; row 0
 * is the begin node, row n_rows-1 the end node (both excluded from the alignment, reference
 * simd_abpoa_align.c:1643-1649).  All arrays are HOST memory owned by the caller. */
typedef struct abpoa_hip_problem_t {
    int32_t n_rows;              /* gn = end_index - beg_index + 1, >= 3                              */
    int32_t qlen;                /* >= 0                                                              */
    const uint8_t *query;        /* [qlen] residue codes 0..m-1                                       */
    const uint8_t *row_base;     /* [n_rows] node base code                                           */
    const int32_t *row_node_id;  /* [n_rows] graph node id (only echoed into cigar words)             */
    const int32_t *row_remain;   /* [n_rows] node_id_to_max_remain; may be NULL when wb<0 && zdrop<=0 */
    const uint8_t *row_active;   /* [n_rows] the reference's index_map (simd_abpoa_align.c:1650-1660);
                                    NULL = every row active                                           */
    const int32_t *pred_off;     /* [n_rows+1] CSR offsets into pred_row                              */
    const int32_t *pred_row;     /* predecessor rows in in_id order, already filtered by row_active
                                    (reference pre_index[][], simd_abpoa_align.c:519-530)             */
    const int32_t *out_off;      /* [n_rows+1] CSR offsets into out_row                               */
    const int32_t *out_row;      /* successor rows in out_id order; -1 for a successor outside the
                                    [0,n_rows) window (sub-graph alignment only)                      */
    int32_t *max_pos_left;       /* [n_rows] in/out: node_id_to_max_pos_left  (only when wb >= 0)     */
    int32_t *max_pos_right;      /* [n_rows] in/out: node_id_to_max_pos_right (only when wb >= 0)     */
} abpoa_hip_problem_t;

/* Optional per-problem DP trace for parity tests (ABPOA_HIP_FLAG_TRACE).  Band-compacted: row r owns
 * cells [row_off[r], row_off[r+1]) of `planes`; inside, plane p (H,E1,[E2],F1,[F2]) of width
 * W = (dp_end_sn-dp_beg_sn+1)*pn starts at row_off[r] + p*W and covers columns dp_beg_sn*pn ... */
typedef struct abpoa_hip_trace_t {
    int32_t  bits;               /* 16 or 32                                                          */
    int32_t  n_planes;           /* 1 / 3 / 5                                                         */
    int32_t *dp_beg, *dp_end;    /* [n_rows] the reference's abm->dp_beg/dp_end                       */
    int32_t *dp_beg_sn, *dp_end_sn; /* [n_rows]                                                       */
    int64_t *row_off;            /* [n_rows+1] in cells                                               */
    void    *planes;             /* int16_t or int32_t cells                                          */
    int32_t *row_max_i;          /* [n_rows] arg-max column per row (-2 where not computed)           */
} abpoa_hip_trace_t;

/* The fields of abpoa_res_t (src/abpoa.h:53-60) plus bookkeeping. */
typedef struct abpoa_hip_result_t {
    int32_t  status;             /* ABPOA_HIP_OK or a negative code                                   */
    int32_t  bits;               /* score width chosen for this alignment (16/32)                     */
    int32_t  best_score, best_row, best_col;
    int32_t  node_s, node_e, query_s, query_e;
    int32_t  n_aln_bases, n_matched_bases;   /* of THIS alignment (the reference accumulates into res) */
    int32_t  n_cigar;
    uint64_t *cigar;             /* libc-malloc'ed by the library (caller frees, as in the reference) */
    int64_t  n_cells;            /* sum over rows of (dp_end_sn-dp_beg_sn+1)*pn  (the reference's
                                    commented-out tot_dp_sn counter, simd_abpoa_align.c:910)          */
    abpoa_hip_trace_t *trace;    /* NULL unless ABPOA_HIP_FLAG_TRACE                                  */
} abpoa_hip_result_t;

#define ABPOA_HIP_FLAG_TRACE   0x1u   /* return the DP planes/bands (tests only; large)               */

/* Cumulative engine counters (since abpoa_hip_init or the last abpoa_hip_reset_stats). */
typedef struct abpoa_hip_stats_t {
    int64_t n_launches;          /* DP kernel launches                                                */
    int64_t n_alignments;
    int64_t n_cells;             /* DP cells (definition above)                                       */
    int64_t algo_bytes;          /* algorithmic HBM bytes: cells * (linear 2S | affine 5S | convex 8S) */
    double  kernel_ms;           /* sum of DP kernel durations measured with hipEvents on the
                                    engine's own stream                                               */
    double  h2d_ms, d2h_ms;      /* copy durations (same events)                                      */
    double  tail_ms;             /* fast path only: global-best + backtrack kernel (dp_fast_tail_kernel), timed apart
                                    from the row-loop kernel that kernel_ms then covers alone                          */
    double  rounds_ms;           /* read-set driver, narrow-band jobs: duration of the all-rounds kernel launches
                                    (poa_rounds_kernel: graph -> rows, row loop, backtrack, cigar -> graph of every round
                                    after the first).  kernel_ms / tail_ms then hold the row loop's / backtrack's share of
                                    it, split by the shader-clock ticks the read-sets spent in each phase             */
    int64_t rounds_launches;
    int64_t rounds_algo_bytes;   /* algorithmic bytes of the DP cells computed inside those launches                 */
} abpoa_hip_stats_t;

/* ---- engine life cycle ------------------------------------------------------------------------ */
int  abpoa_hip_device_count(void);          /* number of HIP devices, 0 if none (never initialises one) */
int  abpoa_hip_init(int device);            /* bind the calling process to `device`; idempotent       */
void abpoa_hip_shutdown(void);
const char *abpoa_hip_last_error(void);
void abpoa_hip_get_stats(abpoa_hip_stats_t *out);
void abpoa_hip_reset_stats(void);
/* Releases the device / pinned pools the read-set driver keeps between calls (a 10 kb job leaves >100 GB cached); the next call re-allocates. */
void abpoa_hip_trim(void);

/* ---- (1) flat batch API ----------------------------------------------------------------------- */
/* Replaces: simd_abpoa_align_sequence_to_subgraph (src/simd_abpoa_align.c:1645-1712), N at a time.
 * results[i] is fully overwritten; results[i].cigar must be free()d by the caller.               */
int  abpoa_hip_align_batch(const abpoa_hip_scoring_t *sc, int n,
                           const abpoa_hip_problem_t *problems,
                           abpoa_hip_result_t *results, unsigned flags);
void abpoa_hip_free_result(abpoa_hip_result_t *r);   /* frees cigar + trace, zeroes the struct     */

/* Score width the reference would choose (src/simd_abpoa_align.c:1672-1683): returns 16 or 32 and
 * stores inf_min.  Pure host arithmetic, usable without a GPU. */
int  abpoa_hip_score_bits(const abpoa_hip_scoring_t *sc, int n_rows, int qlen, int32_t *inf_min);

/* ---- (3) read-set batch API -------------------------------------------------------------------- */
/* Progressive POA of N independent read-sets in lock-step rounds: round k aligns read k of every
 * set to that set's graph (one launch for all sets) and fuses the cigars into the graphs.
 * Per set this is the reference's abpoa_msa() (src/abpoa_align.c:373-437) with plain abpoa_poa()
 * (:302-344): no seeding / guide tree, unit weights, no reverse-complement retry.  Graph fusion and
 * the consensus / MSA calls follow the reference so that outputs are identical.
 * Two drivers, same results: the DEVICE-RESIDENT driver (global mode, adaptive band, affine / convex
 * gaps, consensus output, nucleotides) keeps every graph in HBM and runs fusion, row order, band
 * inputs, DP, backtrack and the consensus as kernels with no host work in between; everything else --
 * and any set that outgrows a device capacity -- takes the HOST driver (host graph, one H2D / launch /
 * D2H per round).                                                                                   */
typedef struct abpoa_hip_readset_t {
    int32_t n_reads;
    const uint8_t *const *seqs;   /* [n_reads] residue codes 0..m-1                                  */
    const int32_t *lens;          /* [n_reads] each > 0                                               */
    const int32_t *const *weights;/* NULL, or [n_reads] per-base edge weights (NULL entry = all 1): the reference's
                                     qv weights, abpoa_msa1 src/abpoa_align.c:462-467 (quality - 32 with -Q)  */
} abpoa_hip_readset_t;

typedef struct abpoa_hip_msa_t {
    int32_t  status;              /* ABPOA_HIP_OK or the first failing alignment's code              */
    int32_t  n_reads;
    int32_t  cons_len;            /* single heaviest-bundling consensus (abpoa_cons_t, n_cons = 1)   */
    uint8_t *cons_base;           /* [cons_len]                                                       */
    int32_t *cons_cov;            /* [cons_len]                                                       */
    int32_t *cons_node_id;        /* [cons_len]                                                       */
    int32_t  msa_len;             /* 0 unless ABPOA_HIP_OUT_MSA                                       */
    int32_t  msa_rows;            /* n_reads (+1 consensus row when both outputs are requested)       */
    uint8_t *msa_base;            /* [msa_rows*msa_len] row-major, gap = m (abpoa_cons_t.msa_base)    */
    int64_t  n_cells;             /* DP cells over all alignments of the set                          */
    uint8_t *is_rc;               /* NULL unless ABPOA_HIP_AMB_STRAND: [n_reads] 1 where the reverse complement of the read
                                     went into the graph (abpoa_seq_t.is_rc, src/abpoa_align.c:331)   */
} abpoa_hip_msa_t;

#define ABPOA_HIP_OUT_CONS 0x1u   /* abpoa_para_t.out_cons */
#define ABPOA_HIP_OUT_MSA  0x2u   /* abpoa_para_t.out_msa  */
#define ABPOA_HIP_AMB_STRAND 0x4u /* abpoa_para_t.amb_strand (-s): a read whose best score is below min(qlen, nodes - 2) * max_mat / 3 is aligned
                                     again as its reverse complement and the better strand goes into the graph (src/abpoa_align.c:315-336) */

/* n_threads <= 0: one host thread per online core.  Every out[i] must be released with
 * abpoa_hip_free_msa.  Fails with ABPOA_HIP_ENODEV when no GPU is usable. */
int  abpoa_hip_msa_batch(const abpoa_hip_scoring_t *sc, int n_sets, const abpoa_hip_readset_t *sets,
                         abpoa_hip_msa_t *out, unsigned flags, int n_threads);
void abpoa_hip_free_msa(abpoa_hip_msa_t *r);
void abpoa_hip_free_msa_array(abpoa_hip_msa_t *r, int n);   /* abpoa_hip_free_msa on r[0..n): one call for a whole batch (bindings whose calls are dear) */

/* Phase timers of the last abpoa_hip_msa_batch call (seconds): host graph work, engine calls. */
typedef struct abpoa_hip_msa_timing_t {
    double host_sort_s, host_fuse_s, engine_s, cons_s, total_s;   /* per-group averages for the first three */
    int32_t n_rounds, n_threads, n_groups;                        /* n_groups = read-set groups run concurrently (one stream each) */
    int32_t n_host_sets;          /* read-sets of the call that took the HOST driver (host graph, one H2D / launch / D2H per round) instead of the device-resident
                                     one: the whole job when its options are not the device driver's, else the sets that outgrew a device capacity.
                                     With ABPOA_HIP_STRICT=1 in the environment such a call fails with ABPOA_HIP_ESTRICT instead of slowing down silently. */
} abpoa_hip_msa_timing_t;
void abpoa_hip_get_msa_timing(abpoa_hip_msa_timing_t *out);
/* Why the n_host_sets read-sets of the last abpoa_hip_msa_batch call left the device-resident driver: out12[r] = sets with reason r -- 1 node slots at the
 * first read, 2 predecessor-list slots, 3 cigar slots, 4 node slots while a read was fused, 5 edge / aligned slots of one node, 6 projected graph growth,
 * 7 row-order walk, 8 MSA rank walk, 9 DP arena too small for the bands, 10 other DP status, 0 other, 11 the job's options (or a pass that did not fit the
 * device memory even alone).  Process-wide like the timing record. */
void abpoa_hip_get_host_reasons(int32_t *out12);

/* ---- switches (additive) -----------------------------------------------------------------------------------------------------------------
 * The library's behaviour switches (device list, strict mode, verbose output; test hooks that force a code path; diagnostics) form one table
 * (abpoa_amd/csrc/engine_options.cpp; abpoa_hip_list_options returns it).  Their values are read ONCE at every entry point of this header -- from the
 * environment variable of the same name unless abpoa_hip_set_option gave one (value NULL: back to the environment) -- and hold for the whole call on every
 * thread.  Unknown names are refused (ABPOA_HIP_EINVAL); nothing outside the table is ever read from the environment. */
int abpoa_hip_set_option(const char *name, const char *value);
int abpoa_hip_list_options(const char **names, const char **help, int cap);      /* returns the number of switches; fills up to cap entries */

/* ---- contexts (additive; SURVEY.md 8b: "the replacement should be re-entrant per abpoa_t") --------------------------------------------------
 * abpoa_hip_msa_batch keeps its timing record and last error per PROCESS and drives the device queues named by ABPOA_GPU_DEVICES: one caller at a
 * time.  A context is the same entry with the per-caller state in a handle: its own device queue (stream, pool cache, argument record), timing and
 * last error -- one host thread per context may call at the same time as others (the reference's abpoa_t plays this role: one graph, one caller).
 * Up to 8 contexts; a context may name any device.  Results are those of abpoa_hip_msa_batch. */
typedef struct abpoa_hip_ctx abpoa_hip_ctx_t;
abpoa_hip_ctx_t *abpoa_hip_ctx_create(int device);          /* device < 0: the device of abpoa_hip_init; NULL on failure (abpoa_hip_last_error) */
void abpoa_hip_ctx_destroy(abpoa_hip_ctx_t *ctx);
int  abpoa_hip_msa_batch_ctx(abpoa_hip_ctx_t *ctx, const abpoa_hip_scoring_t *sc, int n_sets, const abpoa_hip_readset_t *sets,
                             abpoa_hip_msa_t *out, unsigned flags, int n_threads);
void abpoa_hip_ctx_get_msa_timing(const abpoa_hip_ctx_t *ctx, abpoa_hip_msa_timing_t *out);
const char *abpoa_hip_ctx_last_error(const abpoa_hip_ctx_t *ctx);

#ifdef __cplusplus
}
#endif
#endif /* ABPOA_HIP_H */
