// Device-resident progressive POA: the partial-order graph of every read-set lives in HBM for the whole job and the
// per-read loop of the reference (abpoa_poa, src/abpoa_align.c:302-344) runs as four kernels per round with no host
// work in between:
//     poa_prepare_kernel   graph -> DP rows (remaining length, row-order CSR, alignment descriptor)   [poa_device.hip]
//     dp_fast_kernel       banded DP row loop                                                          [dp_kernel.hip]
//     dp_fast_tail_kernel  global best + backtrack -> graph cigar                                      [dp_kernel.hip]
//     poa_fuse_kernel      cigar -> graph (abpoa_add_graph_alignment) + incremental row order          [poa_device.hip]
// One wavefront (= one workgroup) owns one read-set.  Internal interface between msa_device.cpp and the kernels.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "engine.h"

namespace abpoa_hip {

constexpr int POA_IN_CAP = 15;     // in-edges per node kept on the device (more -> the set falls back to the host driver); = DIR_K_MAX of dir_plane.h: a
                                   // direction word names a predecessor by 1 + its list index in 4 bits
constexpr int POA_OUT_CAP = 16;    // out-edges per node
constexpr int POA_HOT = 4;         // slots of every edge list in the hot arrays (16-byte records: most nodes have 1-2 edges, and the graph
                                   // kernels stream the whole graph every round); the other slots live in the cold arrays
constexpr int POA_ALN_MAX = 26;    // aligned (mismatch-alternative) nodes per node: at most m - 1 (every member of a group has its own base); the stride of
                                   // nd_aln is PoaDev.aln_cap = m - 1 (4 for nucleotides, 26 for the 27-code amino-acid alphabet)

#define POA_ST_OK        0
#define POA_ST_FALLBACK  100       // capacity exceeded somewhere (nodes, edges, arena, cigar): redo this set on the host driver

struct PoaSet {                    // immutable per read-set
    int32_t n_reads, node_cap;     // node_cap: slots in every per-node / per-row pool
    int32_t pred_cap, band_extra;  // entries of this set's pred_row slice; band_extra: columns a row's band may be wider than 2 w (reads of very different lengths:
                                   // the band's anchor `qlen - remaining length` then sits that far from the path) -- AlnDesc.pad0 of the set's alignments
    int64_t read0;                 // first read in the read tables
    int64_t node0;                 // first slot in the node / row pools
    int64_t pred0;                 // first slot in pred_row
    int64_t plane_off, plane_cap;  // arena: byte offset, capacity in bytes
    int64_t cigar_off; int32_t cigar_cap, pad2;
    int64_t scratch0;              // first slot of this set's int32 scratch (3 * max_qlen + node_cap + 1 entries)
    int64_t cons0; int32_t cons_cap, pad3;     // this set's slice of the consensus result pools
    int64_t term0;                 // first slot of this set's terminal edge pools (PoaDev.t_out / t_outw / t_in): n_reads + 2 entries each
};

struct PoaState {                  // mutable per read-set
    int32_t n_nodes, status, order_buf, pad;     // order_buf: which row_node buffer is current; pad: fall-back reason
    int32_t cons_len, msa_len;     // heaviest-bundling consensus length (poa_consensus_kernel); MSA columns (poa_msa_rank_kernel)
    int32_t grow_n2, grow_n6;      // nodes after 2 / 6 reads: the growth model of the doomed-pass test (poa_bodies.h)
    int64_t n_cells;               // DP cells over all alignments so far
    int64_t algo_bytes;            // cells * algorithmic bytes per cell (affine 5S, convex 8S; S = 2 | 4)
    int64_t algo_bytes_before;     // algo_bytes when the all-rounds kernel took the set over (what it computed itself = algo_bytes - this)
    int64_t t_phase[4];            // all-rounds kernel (poa_rounds.hip): shader-clock ticks this set spent in prepare / row loop / backtrack / fuse
    uint64_t cigar_dig;            // (PoaDev.dig_on, tests) digest of every graph cigar fused into this set so far, folded in read order: poa_cigar_digest_round
};
// Test hook (ABPOA_HIP_CIGAR_DIGEST=1): one 64-bit digest per read-set over the graph cigars of all its alignments, computed the same way by the fuse phase on
// the device (all lanes: a sum of per-word mixes) and by the host driver (msa_batch.cpp), so that the cigars of BOTH forms of the device driver -- the all-rounds
// kernel with its four-wavefront backtrack included -- can be compared with the oracle-backed host run word for word without leaving the device.
__host__ __device__ inline uint64_t poa_mix64(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
        }
__host__ __device__ inline uint64_t poa_cigar_word_mix(uint64_t word, int i) { return poa_mix64(word + 0xD6E8FEB86659FD93ull * (uint64_t)(i + 1)); }
__host__ __device__ inline uint64_t poa_cigar_digest_round(uint64_t before, int read_index, int n_cigar, uint64_t sum_of_word_mixes) {
    return (before * 0x9E3779B97F4A7C15ull) ^ (sum_of_word_mixes + poa_mix64(((uint64_t)(uint32_t)read_index << 32) | (uint32_t)n_cigar));
}

struct PoaDev {                    // everything the poa_* kernels need; passed by value
    int32_t n_sets, m;
    int32_t max_mat, min_mis, o1, e1, o2, e2, wb; float wf;
    int32_t gap_mode, round;       // round k: read k of every set is aligned / fused
    int32_t max_qlen, pad;         // pad: node capacity of the prepare kernel's LDS jump records (4 bytes each; 0 = use the global-memory sweep)
    int32_t aln_cap;               // slots per node in nd_aln (m - 1)
    int32_t rid_words;             // read-id bitsets per out-edge (abpoa_para_t.use_read_ids: MSA output): 64-bit words per edge, 0 = not kept
    int32_t order_mode;            // 0: the row order is maintained incrementally by the fuse phase (global mode: any topological order gives the same result);
                                   // 1: the reference's own order, rebuilt before every alignment (poa_order_kernel: local mode breaks score ties by row index)
    int32_t banded;                // 0: no adaptive band (local mode): the remaining length is not computed
    int32_t msa_rows, msa_cons;    // rows of a set's MSA = its reads (+ 1 when msa_cons: the consensus row, abpoa_output.c:151-164)
    int32_t order_ecap, general;   // order_ecap: aligned-list entries (16 bits each) the order kernel's all-in-LDS walk holds; general: the job's alignments run in the general
                                   // kernel (rows_general.h: linear gaps, extension mode, no band, long local reads) -- prepare also writes the successor CSR it reads
    int32_t last_pass, pad_lp;     // last_pass: the largest node capacity there is -- no early exit on projected growth (the projection errs on ragged read-sets,
                                   // whose reads add their nodes in bursts; a set that really outgrows this pass goes to the host driver either way)
    int32_t in_cap, out_cap;       // edge slots per inner node (hot + cold): POA_IN_CAP / POA_OUT_CAP in every pass but the last, which has room for one edge per read
                                   // (msa_device.cpp: a set whose node ran out of edge slots is redone there; no direction words then -- dir_plane.h names a predecessor in four bits)
    int32_t order_lds, dig_on;     // order_lds: node capacity of the order / rank kernels' LDS tables (0: the tables live in the set's scratch slice); dig_on: PoaState.cigar_dig is kept
    const PoaSet *sets; PoaState *state;
    const int64_t *read_off; const int32_t *read_len; const uint8_t *reads;       // resident reads: codes 0..m-1
    const int32_t *wts;            // per-base edge weights, parallel to `reads` (the reference's -Q: abpoa_msa1 src/abpoa_align.c:462-467); NULL: every weight 1
    // graph, indexed node0 + node id
    uint8_t *nd_base, *nd_nin, *nd_nout, *nd_naln;
    int32_t *nd_in, *nd_out, *nd_outw;              // hot slots  [node][POA_HOT]: in ids, out ids, out weights
    int32_t *nd_inx, *nd_outx, *nd_outwx;           // cold slots [node][in_cap / out_cap - POA_HOT]
    // The source's out-edges beyond POA_OUT_CAP and the sink's in-edges beyond POA_IN_CAP (reads that do not all start / end on the same node: every read adds
    // at most one of each, so n_reads + 2 entries per set hold them all); slot t >= CAP of node 0 / node 1 is entry t - CAP of the set's slice (poa_bodies.h out_slot / in_slot)
    int32_t *t_out, *t_outw, *t_in;
    int32_t *nd_aln;                                // [node][aln_cap]
    uint64_t *nd_rid;                               // [node][out_cap][rid_words]: reads that went through the out-edge (reference abpoa_node_t.read_ids)
    int32_t *nd_nread, *nd_row;
    int32_t *row_node[2];          // row order (double buffered), indexed node0 + row
    int32_t *scratch;
    // DP inputs produced by the prepare kernel (same arrays DevBatch points at)
    AlnDesc *aln; AlnOut *out;
    uint8_t *row_base; int32_t *row_node_id, *row_remain, *pred_off, *pred_row;
    int32_t *out_off, *out_row;    // general jobs: successor rows per row (DevBatch.out_off / out_row), offsets indexed like pred_off, entries in the set's pred_row-sized slice
    uint32_t *row_pd;              // per row, two dwords: distances to the first eight predecessors (DevBatch.row_pd)
    uint8_t *row_sdist;            // per row: min(255, largest row distance to a successor), 255 for a predecessor of the sink (DevBatch.row_sdist)
    uint64_t *cigar;
    // consensus results (poa_consensus_kernel), indexed cons0 + position
    int32_t *cons_node, *cons_cov; uint8_t *cons_base;
    // MSA output (poa_msa_rank_kernel / poa_msa_fill_kernel): per node its MSA column + 1 (node0 + node id), the row-major result pool
    // ambiguous strand (-s; reference abpoa_poa, src/abpoa_align.c:315-336): a read that scores below a third of the best possible is aligned again as its
    // reverse complement on the same rows; the strand with the strictly better score goes into the graph.  NULL pointers: not an -s job.
    uint8_t *reads_rc; int32_t *wts_rc;      // reverse complements (and reversed weights) of the reads being retried, same offsets as reads / wts, same allocation as `reads`
    uint8_t *is_rc;                          // [read]: the reverse complement went into the graph (abpoa_seq_t.is_rc)
    uint8_t *retry;                          // [set]: this round's alignment is being repeated
    AlnOut *out_fwd; uint64_t *cigar_fwd;    // forward result of a set under retry (cigar_fwd: same slices as cigar)
    int32_t *msa_rank; uint8_t *msa_out; const int64_t *msa_off;      // msa_off[set]: first byte of the set's rows in msa_out (host prefix sum over rows x msa_len)
};

hipError_t launch_poa_init(const PoaDev &p, hipStream_t s);
hipError_t launch_poa_prepare(const PoaDev &p, hipStream_t s);
hipError_t launch_poa_fuse(const PoaDev &p, hipStream_t s);
// -s: after the forward alignment, pick the reads to repeat as reverse complement (descriptor -> general kernel on the rc read, every other set skipped);
// after the retry, keep the better strand
hipError_t launch_poa_strand_check(const PoaDev &p, hipStream_t s);
hipError_t launch_poa_strand_pick(const PoaDev &p, hipStream_t s);
hipError_t launch_poa_consensus(const PoaDev &p, hipStream_t s);
// the reference's row order (abpoa_BFS_set_node_index, src/abpoa_graph.c:186-231) rebuilt on the device: order_mode 1, before every prepare
hipError_t launch_poa_order(const PoaDev &p, hipStream_t s);
// MSA output: rank pass (abpoa_DFS_set_msa_rank, src/abpoa_graph.c:315-362) -> PoaState.msa_len, msa_rank; fill pass (abpoa_output.c:103-166) -> msa_out
hipError_t launch_poa_msa_rank(const PoaDev &p, hipStream_t s);
hipError_t launch_poa_msa_fill(const PoaDev &p, hipStream_t s);
size_t poa_order_lds_bytes(int node_cap, int ecap);      // dynamic LDS of the order / rank kernels for a table capacity of node_cap nodes
// poa_rounds.hip: rounds k_lo .. n_reads - 1 of every set in one launch (narrow-band jobs), and how many of its workgroups a CU holds
constexpr int POA_CU_TICKETS = 4096;      // per-CU ticket counters of the all-rounds kernel (index: XCC id, SE, SH, CU), zeroed by the launch
// host_args: poa_rounds_args_bytes() of pinned host memory that stays valid until the stream has passed the launch (8-byte aligned)
size_t poa_rounds_args_bytes();
hipError_t launch_poa_rounds(const PoaDev &p, const DevBatch &b, int32_t *cu_ticket, void *host_args, int slot, int k_lo, size_t lds_bytes, hipStream_t s);
int poa_rounds_residency(int gap_mode, size_t lds_bytes, int *static_lds);

}  // namespace abpoa_hip
