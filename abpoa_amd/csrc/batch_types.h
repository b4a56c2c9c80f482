// Plain (HIP-free) types shared by the engine's BatchStream and the read-set driver.
#pragma once
#include <stdint.h>
#include "../../include/abpoa_hip.h"

namespace abpoa_hip {

struct BatchShape { int32_t n_rows, qlen; int64_t n_pred, n_out; };

struct ProblemSlots {           // host pointers for one problem; sizes from its BatchShape
    uint8_t *query;             // [qlen]
    uint8_t *row_base;          // [n_rows]
    int32_t *row_node_id;       // [n_rows]
    int32_t *row_remain;        // [n_rows]
    uint8_t *row_active;        // [n_rows]   (pre-filled with 1)
    int32_t *pred_off;          // [n_rows+1] offsets relative to this problem's pred_row
    int32_t *pred_row;          // [n_pred]
    int32_t *out_off;           // [n_rows+1]
    int32_t *out_row;           // [n_out]
    int32_t *left, *right;      // [n_rows]   only meaningful without BS_FRESH_BAND
};

// What the read-set driver needs from "something that aligns a batch": the product implements it on a BatchStream
// (HIP), the CPU test shim on the oracle.
enum { GA_BAND_FRESH = 0, GA_BAND_KEEP = 1, GA_BAND_SEEDED = 2 };
class GroupAligner {
  public:
    virtual ~GroupAligner() {}
    // band: GA_BAND_FRESH = max_pos_left/right start at their reset value (n_rows, 0) and are not read back; GA_BAND_KEEP = fresh start, and
    // left(i) / right(i) hold the state the DP left behind after run(); GA_BAND_SEEDED = the caller fills slots(i).left / .right with the start state
    virtual int prepare(const abpoa_hip_scoring_t *sc, int n, const BatchShape *shapes, int band = 0) = 0;
    virtual const int32_t *left(int i) = 0;
    virtual const int32_t *right(int i) = 0;
    virtual ProblemSlots slots(int i) = 0;
    virtual int run() = 0;
    virtual int status(int i) = 0;
    virtual int64_t n_cells(int i) = 0;
    virtual int best_score(int i) = 0;
    virtual int n_cigar(int i) = 0;
    virtual const uint64_t *cigar(int i) = 0;
};
typedef GroupAligner *(*AlignerFactory)(void);

// Host cores this process may really use: min(online CPUs, cgroup CPU quota) -- a GPU box shows 256 logical CPUs and grants 16.
int effective_host_cores();

}  // namespace abpoa_hip
