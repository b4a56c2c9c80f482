// Adaptive-banded sequence-to-graph DP for gfx950 (MI355X).
//
// One 64-lane wavefront (= one workgroup) owns one alignment and walks its graph rows in topological
// order; rows of one alignment are strictly sequential because the band of row r depends on the arg-max
// of all its predecessor rows (reference src/simd_abpoa_align.c:1059-1067), so per-row LATENCY is what
// the design minimises:
//   * lanes map to consecutive band columns, 64 columns ("chunk") at a time; the reference's SIMD register
//     (pn = 16 int16 / 8 int32 lanes) is a group of pn adjacent lanes, its whole-register lane shifts are
//     DPP row shifts and its vector-to-vector carry ("first") is a wave-uniform scalar;
//   * everything a row needs from earlier rows lives in LDS: a 64-row tile of static graph metadata
//     (base, predecessor / successor lists, remaining length), a ring with the band geometry of the last
//     256 rows, a look-ahead ring of max_pos_left/right for the next 256 rows and a ring with the H/E
//     score rows of the last `ring_rows` rows (predecessor distance is 1-12 rows in practice).  Older
//     predecessors and over-wide rows fall back to the HBM copy;
//   * score planes are also streamed band-compacted to HBM (row r owns P*(end_sn-beg_sn+1)*pn cells,
//     written once, coalesced) because the backtrack compares H/E/F by value; the backtrack then pulls
//     64-row windows of that arena back into LDS with wide coalesced loads and walks them there.
//
// Bit-exactness contract: every add/sub is done in the score width with
// two's-complement wrap, the masked log-step scan of SIMD_SET_F (:665-699) is reproduced step by step,
// and the band, arg-max tie-break and backtrack priority follow the reference literally.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <type_traits>
#include "engine.h"
#include "../../include/abpoa_hip.h"

namespace abpoa_hip {

#define GLOBAL_AS __attribute__((address_space(1)))
#define OP_M   0x1
#define OP_E1  0x2
#define OP_E2  0x4
#define OP_E   0x6
#define OP_F1  0x8
#define OP_F2  0x10
#define OP_F   0x18
#define OP_ALL 0x1f

constexpr int TS = 64;      // rows per static-metadata tile
constexpr int TP = 256;     // predecessor / successor entries staged per tile
constexpr int RB = 256;     // band-geometry ring depth (rows)
constexpr int RL = 256;     // max_pos_left/right window (two halves of 128 rows)
constexpr int RLH = RL / 2;
constexpr int MAX_RING_ROWS = 32;
constexpr int BTR = 64;     // backtrack tile: rows
constexpr int BTP = 256;    // backtrack tile: predecessor entries

struct __attribute__((aligned(16))) DpLds {   // fixed part of the DP-phase LDS image; 16-byte records = one ds_read_b128 each
    int4 t_rec[TS + 1];     // static tile, per row: {pred_off, out_off, remain, base | active << 8}; entry [TS] = end offsets
    int4 b_rec[RB];         // band ring: {beg_sn, end_sn, cell_off / PN, row id while its H/E rows sit in the score ring else -1}
    int2 l_lr[RL];          // look-ahead window: {max_pos_left, max_pos_right}
    int4 t_fast[TS];        // fast-row record: {flag<<31 | base<<16 | dist(pred1)<<8 | dist(pred0), rterm, out0, out1}
    int32_t t_pred[TP], t_out[TP];
};
struct BtLds {              // fixed part of the backtrack-phase LDS image
    long long coff[BTR + 1];
    int32_t bsn[BTR], esn[BTR], poff[BTR + 1], nid[BTR], pred[BTP];
    uint8_t base[BTR];
    // lane-parallel walk (cell-record arenas): one record per window row and per predecessor edge, so that a step is two LDS
    // round trips.  rinfo = {first column | columns << 16, arena offset (values, relative to the window), edge index | n_pred << 16 |
    // base << 24, node id}; edge = {predecessor row, its rinfo.x, its rinfo.y, inside the window?}; edge2 = {its rinfo.z, its rinfo.w, its rinfo2, -}
    int4 rinfo[BTR]; int4 edge[BTP]; int4 edge2[BTP];
    int32_t rinfo2[BTR];        // staged column range of the row: first staged column | count << 16 (the window holds a column slice, not whole rows)
    long long srcoff[BTR];      // arena offset (values) of the row's first staged record
};
int lds_fixed_bytes_dp() { return (int)((sizeof(DpLds) + 15) & ~15u); }
int lds_fixed_bytes_bt() { return (int)((sizeof(BtLds) + 15) & ~15u); }

template <int CTRL>
__device__ __forceinline__ int dpp_mov(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, 0xF, false);
}
// value of lane-S inside a 16-lane DPP row; lanes whose source falls outside the row keep `old`
template <int S>
__device__ __forceinline__ int row_shr(int old, int src) { return dpp_mov<0x110 + S>(old, src); }

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// wave-wide max; every lane must be active
__device__ __forceinline__ int wave_max_i32(int x) {
    x = imax(x, row_shr<1>(x, x));
    x = imax(x, row_shr<2>(x, x));
    x = imax(x, row_shr<4>(x, x));
    x = imax(x, row_shr<8>(x, x));
    int a = __builtin_amdgcn_readlane(x, 15), b = __builtin_amdgcn_readlane(x, 31);
    int c = __builtin_amdgcn_readlane(x, 47), d = __builtin_amdgcn_readlane(x, 63);
    return imax(imax(a, b), imax(c, d));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned x) {
    auto umax = [](unsigned p, unsigned q) { return p > q ? p : q; };
    x = umax(x, (unsigned)row_shr<1>((int)x, (int)x));
    x = umax(x, (unsigned)row_shr<2>((int)x, (int)x));
    x = umax(x, (unsigned)row_shr<4>((int)x, (int)x));
    x = umax(x, (unsigned)row_shr<8>((int)x, (int)x));
    unsigned a = __builtin_amdgcn_readlane((int)x, 15), b = __builtin_amdgcn_readlane((int)x, 31);
    unsigned c = __builtin_amdgcn_readlane((int)x, 47), d = __builtin_amdgcn_readlane((int)x, 63);
    return umax(umax(a, b), umax(c, d));
}

// Rare-path HBM loads.  They are written as inline asm on purpose: with ordinary loads hipcc merges the LDS load of
// the common path and the global load of the fallback path into ONE flat load of a selected pointer, and a flat load
// waits for vmcnt(0)+lgkmcnt(0), i.e. for every outstanding score-plane store of the wave (gfx9 counts stores in vmcnt).
// Each helper waits for its own data (and, as a side effect, for this wave's earlier stores, which these paths need).
__device__ __forceinline__ int gld_i32(GLOBAL_AS const int32_t *p) { int v; asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ int gld_u8(GLOBAL_AS const uint8_t *p) { int v; asm volatile("global_load_ubyte %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ long long gld_i64(GLOBAL_AS const int64_t *p) { long long v; asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ int gld_cell(GLOBAL_AS const int16_t *p) { int v; asm volatile("global_load_sshort %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ int gld_cell(GLOBAL_AS const int32_t *p) { return gld_i32(p); }

// Loads whose completion the CALLER waits for (s_waitcnt vmcnt(0) via gld_wait): used to keep many loads in flight where hipcc
// would pair every load with its own wait.
__device__ __forceinline__ void gld_async(int4 &v, const int4 *p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"((GLOBAL_AS const int4 *)p) : "memory"); }
__device__ __forceinline__ void gld_async(int2 &v, const int2 *p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"((GLOBAL_AS const int2 *)p) : "memory"); }
__device__ __forceinline__ void gld_async(int &v, const int32_t *p) { asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"((GLOBAL_AS const int32_t *)p) : "memory"); }
__device__ __forceinline__ void gld_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Moves a wave-uniform pointer into a VGPR pair and hides its uniformity from the compiler.  The kernel keeps ~20
// per-alignment base pointers; left in SGPRs they (with the per-row uniforms) overflow the 102-SGPR budget and the hot
// loop drowns in v_readlane/v_writelane spill traffic.  VGPRs are plentiful here (one wave per SIMD).
template <typename Pt> __device__ __forceinline__ GLOBAL_AS Pt *vgpr_ptr(Pt *p) { asm("" : "+v"(p)); return (GLOBAL_AS Pt *)p; }
// (the result stays typed as a GLOBAL pointer: a generic pointer would turn every access into a flat load, and an
//  outstanding flat load also blocks s_waitcnt lgkmcnt(0), i.e. every LDS wait of the row loop)

// wave-wide unsigned max with a single v_readlane: 4 in-row steps, then row_bcast:15 / row_bcast:31 fold the four DPP rows
__device__ __forceinline__ unsigned wave_max_u32_b(unsigned x) {
    auto umax = [](unsigned p, unsigned q) { return p > q ? p : q; };
    x = umax(x, (unsigned)row_shr<1>((int)x, (int)x));
    x = umax(x, (unsigned)row_shr<2>((int)x, (int)x));
    x = umax(x, (unsigned)row_shr<4>((int)x, (int)x));
    x = umax(x, (unsigned)row_shr<8>((int)x, (int)x));
    x = umax(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x142, 0xA, 0xF, false));   // rows 1,3 <- lane 15 of the row before
    x = umax(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x143, 0xC, 0xF, false));   // rows 2,3 <- lane 31
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}


// inclusive prefix max over the 64 lanes (Hillis-Steele inside each 16-lane DPP row, then row_bcast:15 / row_bcast:31).
// Written as asm: with update_dpp(old = x, src = x) hipcc emits mov + mov_dpp + max per step; with old = identity it folds
// to one v_max_*_dpp but schedules the surrounding code worse (measured 2 % slower rows).  s_nop 1 = the two wait states a
// DPP read needs after a VALU write of the same VGPR.
#define DPP_SCAN6(OP)                                                                                                    \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                           \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                                           \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                                           \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                                           \
    "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                                        \
    "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
__device__ __forceinline__ int wave_scan_max_i32(int x) { asm(DPP_SCAN6("v_max_i32_dpp") : "+v"(x)); return x; }
// wave-wide max (the same six steps; complete in lane 63), returned as a wave-uniform value
__device__ __forceinline__ unsigned wave_max_u32_s(unsigned x) { asm(DPP_SCAN6("v_max_u32_dpp") : "+v"(x)); return (unsigned)__builtin_amdgcn_readlane((int)x, 63); }
__device__ __forceinline__ int wave_max_i32_s(int x) { asm(DPP_SCAN6("v_max_i32_dpp") : "+v"(x)); return __builtin_amdgcn_readlane(x, 63); }
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ int wave_scan_add_i32(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false); x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false); x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false); x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
    return x;
}
// keeps a wave-uniform value in an SGPR and hides it from pattern matching (hipcc otherwise turns scalar min/max chains
// into VALU v_min3/v_max3 + v_readfirstlane)
__device__ __forceinline__ int sgpr(int x) { x = __builtin_amdgcn_readfirstlane(x); asm("" : "+s"(x)); return x; }
// value of lane-1 (whole wave, DPP wave_shr:1); lane 0 receives `lane0`
__device__ __forceinline__ int wave_shr1(int lane0, int src) { return __builtin_amdgcn_update_dpp(lane0, src, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ int med3i(int a, int lo, int hi) { return imin(imax(a, lo), hi); }

template <typename T> struct Width;
template <> struct Width<int16_t> { static constexpr int PN = 16, LOGN = 4; };
template <> struct Width<int32_t> { static constexpr int PN = 8, LOGN = 3; };

// wrapping arithmetic in the score width (reference: _mm256_add/sub_epi16|32)
template <typename T> __device__ __forceinline__ T wadd(T a, T b) { return (T)((uint32_t)(int32_t)a + (uint32_t)(int32_t)b); }
template <typename T> __device__ __forceinline__ T wsub(T a, T b) { return (T)((uint32_t)(int32_t)a - (uint32_t)(int32_t)b); }
template <typename T> __device__ __forceinline__ T tmax(T a, T b) { return a > b ? a : b; }

// One step-by-step SIMD_SET_F (reference :665-699) on every pn-lane group of the wave at once.
// l = lane % PN.  set_num == PN selects the plain variant.
template <typename T>
__device__ __forceinline__ T set_f(T f, int l, int set_num, T e, T inf) {
    constexpr int PN = Width<T>::PN, LOGN = Width<T>::LOGN;
    T es = e; int cov = set_num;
#define SETF_STEP(K)                                                                          \
    if (K < LOGN) {                                                                           \
        constexpr int S = 1 << K;                                                             \
        if (K > 0) { es = wadd<T>(es, es); cov += S; }                                        \
        T t = wsub<T>(f, es);                                                                 \
        T sh = (T)row_shr<S>((int)inf, (int)t);                                               \
        if (PN == 8) sh = (l < S) ? inf : sh;      /* two vectors share a 16-lane DPP row */   \
        if (set_num != PN) sh = (l > cov) ? inf : sh;                                         \
        f = tmax<T>(f, sh);                                                                   \
    }
    SETF_STEP(0) SETF_STEP(1) SETF_STEP(2) SETF_STEP(3)
#undef SETF_STEP
    return f;
}

// Distance (in lanes) to the nearest "inf" injection of the reference's log-step scan (zero-filled shift | PRE_MIN):
// after the scan lane l holds max( clean prefix scan , inf - INJ[l]*e ); -1 = no injection reaches the lane.
template <int PN> __device__ __forceinline__ int inj_dist(int l);
template <> __device__ __forceinline__ int inj_dist<16>(int l) { return l < 8 ? 0 : (l < 12 ? 8 : (l < 14 ? 12 : (l == 14 ? 14 : -1))); }
template <> __device__ __forceinline__ int inj_dist<8>(int l) { return l < 4 ? 0 : (l < 6 ? 4 : (l == 6 ? 6 : -1)); }

// Closed form of "F = (H<<1 | first) - oe; SIMD_SET_F(F); first = max(H[pn-1], F[pn-1] + o)" (reference :870-874) for the
// first `nfast` vectors of a chunk at once, valid when no subtraction can wrap (the caller checks hs >= MIN + oe + pn*e):
// then max-plus arithmetic distributes and  F[l] = max( scan of the vector's own H , first - oe - l*e , inf - INJ[l]*e ),
// and the vector-to-vector carry is first' = max(H[pn-1], ownscan[pn-1] + o, first - pn*e).  Plain int arithmetic.
template <typename T>
__device__ __forceinline__ int fast_f_chain(int hs, int &first, int nfast, int l, int vvl, int oe, int e, int o, int cl, int inj, int *dbg_cv = nullptr) {
    constexpr int PN = Width<T>::PN, NV = 64 / PN;
    int f = row_shr<1>(hs, hs) - oe;                 // own sources: F0[l] = H[l-1] - oe for l >= 1 (lane 0 has none)
    if (PN == 16) {
        f = (l == 0) ? -(1 << 30) : f;               // int16 values in 32-bit registers: a plain sentinel survives the scan
        f = imax(f, row_shr<1>(f, f) - e);
        f = imax(f, row_shr<2>(f, f) - 2 * e);
        f = imax(f, row_shr<4>(f, f) - 4 * e);
        f = imax(f, row_shr<8>(f, f) - 8 * e);
    } else {                                         // int32: no room for a sentinel, lane l only takes from lanes l-s >= 1
        int t;
        t = row_shr<1>(f, f) - e;     f = (l > 1) ? imax(f, t) : f;
        t = row_shr<2>(f, f) - 2 * e; f = (l > 2) ? imax(f, t) : f;
        t = row_shr<4>(f, f) - 4 * e; f = (l > 4) ? imax(f, t) : f;
    }
    const int cv = imax(hs, f + o);                  // at lane pn-1 of a vector: max(H[pn-1], ownscan[pn-1] + o)
    if (dbg_cv) *dbg_cv = cv;
    int fc[NV + 1]; fc[0] = first;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c_v = __builtin_amdgcn_readlane(cv, v * PN + PN - 1);
        fc[v + 1] = (v < nfast) ? imax(c_v, fc[v] - PN * e) : fc[v];
    }
    int fv = fc[0];
#pragma unroll
    for (int v = 1; v < NV; ++v) fv = (vvl >= v) ? fc[v] : fv;
    first = fc[NV];
    const int own = (l == 0) ? INT_MIN : f;
    return imax(imax(own, fv - cl), inj);
}

// wave-uniform LDS records: tell the compiler (values land in SGPRs, branches on them become scalar branches)
__device__ __forceinline__ int4 uniform4(int4 v) {
    return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y), __builtin_amdgcn_readfirstlane(v.z), __builtin_amdgcn_readfirstlane(v.w));
}
__device__ __forceinline__ int2 uniform2(int2 v) { return make_int2(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y)); }

extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

// =====================================================================================================================
// Register-resident row loop ("fast loop") for the production case: global alignment, adaptive band, affine / convex
// gaps, every row active, band state at its reset value.  Same cells, bands and arg-max as the general loop in
// align_one() (and therefore as the reference), organised to minimise the INSTRUCTION COUNT of one row, because with
// one wavefront per SIMD a row costs (instructions x ~4 cycles) + exposed latency:
//   * max_pos_left/right are PULLED: left/right of row r = min/max over its predecessors p of (argmax_p + 1) (what the
//     reference's push at :1059-1067 leaves there once every predecessor is done), so the band needs no window at all;
//   * band geometry, arg-max and arena offset of the last 64 rows live in three VGPRs (lane = row & 63) and are read /
//     written with v_readlane / v_writelane -- no LDS round trip between a row and its successor;
//   * static per-row metadata (base, first four predecessors, remaining length) sits in VGPRs per 64-row tile
//     (lane = row & 63), loaded from the CSR arrays two tiles ahead;
//   * the H/E rows of the last fr_rows rows sit in an LDS ring as packed words (int16: H | E1 << 16), fr_cols columns per
//     row with "inf" guard cells on both sides and "inf" padding after the band: the first predecessor needs no range
//     masks at all (reading outside its stored band yields exactly what the reference reads or assigns there), further
//     predecessors need one unsigned compare per plane; one ds_read2_b32 fetches H[col-1], H|E[col];
//   * F is one 64-lane prefix-max scan: with g[c] = hs[c] + c*e,  F[c] = max_{c' < c} g[c'] - (oe - e) - c*e  whenever no
//     subtraction can wrap (same condition as fast_f_chain), then max with the lane-constant "inf injection" term of
//     the reference's zero-filled shifts;  vectors beyond max_pre_end_sn use the literal masked scan (set_f);
//   * rows that do not fit (predecessor further back than the ring, > 4 predecessors, band wider than the ring) take
//     the general gather (exact range masks, HBM copies) but share everything else.
// Arena format of the fast loop: one record of CW values per column -- {H, E1, F1, -} (affine) or {H, E1, E2, F1, F2, -, -, -}
// (convex), plane id = index in the record -- so that a row chunk is ONE wide store per lane instead of 3-5 two-byte ones
// (vector-memory instruction issue, not bytes, is what a lone wave pays for).
template <typename T, int GAP> struct FastFmt { static constexpr int CW = GAP == 1 ? 4 : 8; };
template <typename T> struct FastIO {
    GLOBAL_AS const uint8_t *row_base; GLOBAL_AS const int32_t *row_remain, *pred_off, *pred_row;
    GLOBAL_AS int32_t *g_bsn, *g_esn, *row_max_i, *g_left, *g_right; GLOBAL_AS int64_t *g_coff;
    T *planes;
};

// literal SIMD_SET_F for the vectors [nfast, ...) of one 64-lane chunk (global mode), reference :859-875 / :978-997
template <typename T, int GAP>
__device__ __forceinline__ void slow_f_vectors(int vbase, int end_sn, int max_pre, int nfast, int l, int vvl, T hs, T inf,
                                               T e1, T oe1, T o1, T e2, T oe2, T o2, T &F1, T &F2, T &first, T &first2) {
    constexpr int PN = Width<T>::PN, NV = 64 / PN;
#pragma unroll
    for (int vv = 0; vv < NV; ++vv) {
        const int vg = vbase + vv;
        if (vv >= nfast && vg <= end_sn) {
            int set_num = PN;
            if (vg > max_pre) set_num = (vg == max_pre + 1) ? 2 : 1;
            T prev = (T)row_shr<1>((int)first, (int)hs);
            if (PN == 8) prev = (l == 0) ? first : prev;
            T f = wsub<T>(prev, oe1);
            f = set_f<T>(f, l, set_num, e1, inf);
            const T hlast = (T)__builtin_amdgcn_readlane((int)hs, vv * PN + PN - 1);
            first = tmax<T>(hlast, wadd<T>((T)__builtin_amdgcn_readlane((int)f, vv * PN + PN - 1), o1));
            if (vvl == vv) F1 = f;
            if (GAP == 2) {
                T prev2 = (T)row_shr<1>((int)first2, (int)hs);
                if (PN == 8) prev2 = (l == 0) ? first2 : prev2;
                T g = wsub<T>(prev2, oe2);
                g = set_f<T>(g, l, set_num, e2, inf);
                first2 = tmax<T>(hlast, wadd<T>((T)__builtin_amdgcn_readlane((int)g, vv * PN + PN - 1), o2));
                if (vvl == vv) F2 = g;
            }
        }
    }
}

#ifdef ABPOA_HIP_PROFILE
#define FSTAMP(I) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); long long t_ = (long long)__builtin_amdgcn_s_memtime(); fseg[I] += t_ - fseg_last; fseg_last = t_; }
#else
#define FSTAMP(I)
#endif
// Timing-only ablation switches (tools/kernel_bench.py with ABPOA_HIP_DBG=bits on the "prof" build): results are wrong on purpose.
#ifdef ABPOA_HIP_ABLATE
#define ABL(BIT) (b.dbg & (BIT))
#else
#define ABL(BIT) false
#endif
template <typename T, int GAP>
__device__ __forceinline__ void rows_fast(const DevBatch &b, const AlnDesc &d, const FastIO<T> &io, const uint8_t *s_query,
                                          long long &cursor_out, long long &n_cells_out, int &status, int &rows_done_out, int &last_done, long long *fseg) {
#ifdef ABPOA_HIP_PROFILE
    long long fseg_last = 0;
#endif
    constexpr int PN = Width<T>::PN, NV = 64 / PN;
    constexpr int CW = FastFmt<T, GAP>::CW;          // values per arena cell record
    constexpr bool I16 = sizeof(T) == 2;
    constexpr int NPW = I16 ? (GAP == 2 ? 2 : 1) : (GAP == 2 ? 3 : 2);
    constexpr int PL_E1 = 1, PL_E2 = 2, PL_F1 = GAP == 1 ? 2 : 3, PL_F2 = 4;
    constexpr int GEO_RING = 1 << 24;
    const int lane = threadIdx.x & 63, l = lane % PN, vvl = lane / PN;
    const int gn = d.n_rows, qlen = d.qlen, m = b.m, m1 = b.m + 1, w = d.w;
    const int inf = d.inf_min;
    const int e1 = b.e1, o1 = b.o1, oe1 = b.o1 + b.e1, e2 = b.e2, o2 = b.o2, oe2 = b.o2 + b.e2;
    const int RR = b.lds.fr_rows, RC = b.lds.fr_cols, RCS = RC + 4;
    int *fr = (int *)(lds_raw + b.lds.phase_off + b.lds.fr_off);
    // LDS byte address of ring row (r & (RR - 1)), column 0, held by lane r & 63: RR divides 64, so one lane-constant VGPR serves
    // every row -- a v_readlane replaces the and / mul / shift / add chain per predecessor and for the row's own slot
    typedef __attribute__((address_space(3))) int lds_int_t;
    const int vslot = (int)(unsigned)(size_t)(lds_int_t *)fr + 4 * ((threadIdx.x & 63 & (RR - 1)) * (NPW * RCS) + 2);
    auto ring_at = [&](int slot_addr, int col_idx) __attribute__((always_inline)) { return (const int *)(lds_int_t *)(size_t)(unsigned)(slot_addr + 4 * col_idx); };
    int *s_mx = (int *)(lds_raw + b.lds.mx_off);
    const int infw = I16 ? (int)(((unsigned)inf & 0xffffu) | ((unsigned)inf << 16)) : inf;
    const int qlen_sn = qlen / PN;
    auto wr = [](int x) __attribute__((always_inline)) { return (int)(T)x; };          // wrap to the score width

    // per-lane constants of the F scan and of the arg-max key
    const int idist = inj_dist<PN>(l);
    const int inj1 = idist >= 0 ? inf - idist * e1 : INT_MIN, inj2 = idist >= 0 ? inf - idist * e2 : INT_MIN;
    const int le1 = lane * e1, le2 = lane * e2;
    const int cf1 = oe1 - e1 + le1, cf2 = oe2 - e2 + le2;                               // F[c] = S[c] - cf
    const long long lo_ll = (long long)(I16 ? INT16_MIN : INT32_MIN) + imax(oe1, oe2) + (long long)PN * imax(e1, e2);
    const int fast_lo = (int)lo_ll;
    const int kconst = I16 ? (int)(0x80000000u | ((unsigned)(PN - 1 - l) << 12) | (unsigned)(2047 - vvl)) : 0;

    // ---- LDS: extended score matrix (column m = 0) and the score ring, everything "inf"
    { GLOBAL_AS const int32_t *g_mat = vgpr_ptr(b.mat); for (int i = lane; i < m * m1; i += 64) { const int bb = i / m1, qc = i - bb * m1; s_mx[i] = qc < m ? g_mat[bb * m + qc] : 0; } }
    for (int i = lane; i < RR * NPW * RCS; i += 64) { const int pl = (i / RCS) % NPW; fr[i] = (I16 && pl == 0) ? infw : inf; }
    __syncthreads();
    auto ring_put = [&](int slot, int x, int H, int E1, int E2) __attribute__((always_inline)) {
        int *q = fr + slot * (NPW * RCS) + 2 + x;
        if (I16) { q[0] = (int)(((unsigned)H & 0xffffu) | ((unsigned)E1 << 16)); if (GAP == 2) q[RCS] = E2; }
        else { q[0] = H; q[RCS] = E1; if (GAP == 2) q[2 * RCS] = E2; }
    };

    int cur = 0, n_vec_lane = 0;                    // arena cursor in units of PN cells (one reference SIMD vector); cell count: per-lane sums of the flushed rows' vectors
    const int cap_pn = (int)(d.plane_cap / PN > 0x7fffffffLL ? 0x7fffffffLL : d.plane_cap / PN);
    const int cap_turbo = cap_pn - NV * CW;       // arena room test of the straight-line rows (at most NV vectors)
    const int remain_end = __builtin_amdgcn_readfirstlane(io.row_remain[gn - 1]);
    // ------------------------------------------------------------------ row 0, reference :553-662
    int vg_geo = 0, vg_mi = 0, vg_off = 0;          // lane = row & 63: beg_sn | end_sn << 12 | in-ring << 24, arg-max column, arena offset / PN
    {
        const int r = __builtin_amdgcn_readfirstlane(io.row_remain[0]) - remain_end - 1;
        const int dp_end0 = imin(qlen, imax(0, qlen - r) + w);
        const int end_sn0 = dp_end0 / PN, W0 = (end_sn0 + 1) * PN;
        if ((long long)W0 * CW > d.plane_cap) { status = ABPOA_HIP_STATUS_OVERFLOW; cursor_out = 0; n_cells_out = 0; rows_done_out = 0; return; }
        const bool ring0 = W0 <= RC;
        T *H = io.planes;
        for (int i = lane; i < W0; i += 64) {
            int h, x1 = inf, x2 = inf, f1 = inf, f2 = inf;
            if (GAP == 1) { const int g = wr(-o1 - e1 * i); h = i == 0 ? 0 : g; x1 = i == 0 ? wr(-oe1) : inf; f1 = i == 0 ? inf : g; }
            else {
                const int g1 = wr(-o1 - e1 * i), g2 = wr(-o2 - e2 * i);
                h = i == 0 ? 0 : imax(g1, g2); x1 = i == 0 ? wr(-oe1) : inf; x2 = i == 0 ? wr(-oe2) : inf; f1 = i == 0 ? inf : g1; f2 = i == 0 ? inf : g2;
            }
            T *cellp = H + (long long)i * CW;
            cellp[0] = (T)h; cellp[PL_E1] = (T)x1; cellp[PL_F1] = (T)f1;
            if (GAP == 2) { cellp[PL_E2] = (T)x2; cellp[PL_F2] = (T)f2; }
            if (ring0) ring_put(0, i, h, x1, x2);
        }
        cur = (end_sn0 + 1) * CW;
        if (lane == 0) { vg_geo = (end_sn0 << 12) | (ring0 ? GEO_RING : 0); vg_mi = 0; vg_off = 0; }     // source: successors get left = right = 1 (:556-561)
    }

    // ------------------------------------------------------------------ static metadata, two tiles ahead
    struct MetaA { int ps, pe, base, rem; };
    struct MetaB { int p[4]; };
    auto load_a = [&](int t0) __attribute__((always_inline)) {
        MetaA a; const int r = imin(t0 + lane, gn - 1);
        a.ps = io.pred_off[r]; a.pe = io.pred_off[r + 1]; a.base = io.row_base[r]; a.rem = io.row_remain[r];
        return a;
    };
    auto load_b = [&](const MetaA &a) __attribute__((always_inline)) {
        MetaB q; const int np = a.pe - a.ps;
#pragma unroll
        for (int k = 0; k < 4; ++k) q.p[k] = io.pred_row[a.ps + imin(k, imax(np - 1, 0))];
        return q;
    };
    MetaA a1 = load_a(0); MetaB b1 = load_b(a1); MetaA a2 = load_a(64);
    int tv_meta = 0, tv_rterm = 0, tv_ps = 0, tv_p0 = 0, tv_p1 = 0, tv_p2 = 0, tv_p3 = 0;
    int tv_tb = 0;          // turbo rows: dist(pred 0) | dist(pred 1) << 8 | (base * (m + 1) * 4) << 16
    auto switch_tile = [&](int t0) __attribute__((always_inline)) {
        const int myrow = t0 + lane, np = a1.pe - a1.ps;
        bool fastrow = np >= 1 && np <= 4 && myrow < gn - 1 && myrow >= 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int dk = myrow - b1.p[k]; fastrow = fastrow && dk >= 1 && dk < RR; }
        tv_meta = (a1.base & 0xff) | (imin(np, 255) << 8) | (fastrow ? (1 << 16) : 0) | ((fastrow && np <= 2 && RC <= 128) ? (1 << 17) : 0) | ((fastrow && np >= 3 && RC <= 128) ? (1 << 18) : 0);      // bit 17: straight-line body (pads 128 ring columns)
        tv_tb = ((myrow - b1.p[0]) & 0xff) | (((myrow - b1.p[1]) & 0xff) << 8) | (((a1.base & 0xff) * m1 * 4) << 16);
        tv_rterm = qlen - (a1.rem - remain_end - 1); tv_ps = a1.ps;
        tv_p0 = b1.p[0]; tv_p1 = b1.p[1]; tv_p2 = b1.p[2]; tv_p3 = b1.p[3];
        a1 = a2; b1 = load_b(a1); a2 = load_a(t0 + 128);
    };
    int qc_beg_sn = -1, qoff0 = 0, qoff1 = 0;        // cached query code of this lane's column for chunks 0/1 of band start qc_beg_sn
    auto geo_of = [&](int p, int row, int &geo, int &mi, int &off) __attribute__((always_inline)) {
        if (row - p < 64) { const int sl = p & 63; geo = __builtin_amdgcn_readlane(vg_geo, sl); mi = __builtin_amdgcn_readlane(vg_mi, sl); off = __builtin_amdgcn_readlane(vg_off, sl); }
        else {
            geo = __builtin_amdgcn_readfirstlane(gld_i32(io.g_bsn + p) | (gld_i32(io.g_esn + p) << 12)); mi = __builtin_amdgcn_readfirstlane(gld_i32(io.row_max_i + p));
            off = __builtin_amdgcn_readfirstlane((int)(uint32_t)(gld_i64(io.g_coff + p) / PN));
        }
    };
    // ---- per-row working set shared by the two row bodies and the epilogue
    int beg_sn = 0, end_sn = 0, off_pn = 0, max_pe = 0, rterm = 0, base = 0, np = 0;
    bool to_ring = false;
    unsigned am_key = 0; int am_val = INT_MIN, am_v = 0, am_isend = 0; bool am_any = false;

    // band of the row from (min, max) predecessor arg-max and predecessor geometry, reference :710-720
    auto set_band = [&](auto pin, int mn_mi, int mx_mi, int min_pb) __attribute__((always_inline)) {
        auto S = [](int x) __attribute__((always_inline)) { if constexpr (decltype(pin)::value) return sgpr(x); else return x; };
        const int left = S(imin(gn, mn_mi + 1)), right = S(decltype(pin)::value ? mx_mi + 1 : imax(0, mx_mi + 1));      // (row arg-max >= -1)
        const int lo = S(imin(left, rterm) - w), hi = S(imax(right, rterm) + w);
        const int beg = S(imax(0, lo)), end = S(imin(qlen, hi));
        beg_sn = imax((int)((unsigned)beg / PN), min_pb); end_sn = (int)((unsigned)end / PN);
    };
    auto refresh_qc = [&]() __attribute__((always_inline)) {
        if (beg_sn != qc_beg_sn) {                 // band start moved: refresh this lane's cached query codes
            qc_beg_sn = beg_sn;
            const int c0 = beg_sn * PN + lane, c1 = c0 + 64;
            qoff0 = (c0 >= 1 && c0 <= qlen) ? (int)s_query[c0 - 1] : m; qoff1 = (c1 >= 1 && c1 <= qlen) ? (int)s_query[c1 - 1] : m;
        }
    };
    // one predecessor's contribution from the score ring (k == 0: unmasked, see the header comment)
    // kb: 1 + list index of the first predecessor that supplies the maximum of H[.][col-1] (kidx = this one's 1 + list index): the match flag
    auto from_ring = [&](int k, int p, int g_, int col, int &Mv, int &E1v, int &E2v, int &kb, int kidx) __attribute__((always_inline)) {
        const int pb = g_ & 0xfff, pe = (g_ >> 12) & 0xfff, Wp = (pe - pb + 1) * PN;
        const int x = col - pb * PN;
        const int *src = ring_at(__builtin_amdgcn_readlane(vslot, p), med3i(x - 1, -2, RC));
        int hm1, ev1, ev2 = inf;
        if (I16) { const int w0 = src[0], w1 = src[1]; hm1 = (int)(short)w0; ev1 = w1 >> 16; if (GAP == 2) ev2 = src[RCS + 1]; }
        else { hm1 = src[0]; ev1 = src[RCS + 1]; if (GAP == 2) ev2 = src[2 * RCS + 1]; }
        if (k == 0) { Mv = hm1; E1v = ev1; E2v = ev2; kb = kidx; }
        else {
            const bool inH = (unsigned)x < (unsigned)(Wp + PN), inE = (unsigned)x < (unsigned)Wp;
            kb = (inH && hm1 > Mv) ? kidx : kb;
            Mv = inH ? imax(Mv, hm1) : Mv; E1v = inE ? imax(E1v, ev1) : E1v; if (GAP == 2) E2v = inE ? imax(E2v, ev2) : E2v;
        }
    };
    // everything of a chunk after the predecessor gather: F, H, E, stores, ring, arg-max candidate (reference :854-883 / :972-1008)
    auto chunk_tail = [&](int c, int nch, int Wr, int Mv, int E1v, int E2v, int q, int kb, int &first, int &first2, T *H, int my_slot) __attribute__((always_inline)) {
        const int rel = c * 64 + lane, col = beg_sn * PN + rel, vb = beg_sn + c * NV, v = vb + vvl;
        const bool in_band = rel < Wr;
        const int h = wr(Mv + q);
        int hs = h; if (GAP == 2) hs = imax(imax(h, E1v), E2v);
        if (c == 0) { first = __builtin_amdgcn_readlane(h, 0); first2 = first; }
        const int nvec = imin(NV, end_sn - vb + 1);
        int nfast = imin(nvec, max_pe - vb + 1);
        if (nfast < 0) nfast = 0;
        if (nfast > 0 && __any(vvl < nfast && h < fast_lo)) nfast = 0;
        int F1 = inf, F2 = inf;
        if (nfast > 0 && !ABL(8)) {
            const int g1 = hs + le1;
            const int S1 = wave_scan_max_i32(wave_shr1(first - e1, g1));
            F1 = imax(S1 - cf1, inj1);
            if (GAP == 2) { const int g2 = hs + le2; const int S2 = wave_scan_max_i32(wave_shr1(first2 - e2, g2)); F2 = imax(S2 - cf2, inj2);
                            if (nfast < nvec || c + 1 < nch) { const int lastl = nfast * PN - 1; first2 = __builtin_amdgcn_readlane(imax(S2, g2), lastl) - lastl * e2; } }
            if (nfast < nvec || c + 1 < nch) { const int lastl = nfast * PN - 1; first = __builtin_amdgcn_readlane(imax(S1, g1), lastl) - lastl * e1; }
        }
        if (nfast < nvec && !ABL(32)) {
            T f1t = (T)F1, f2t = (T)F2, fi = (T)first, fi2 = (T)first2;
            slow_f_vectors<T, GAP>(vb, end_sn, max_pe, nfast, l, vvl, (T)hs, (T)inf, (T)e1, (T)oe1, (T)o1, (T)e2, (T)oe2, (T)o2, f1t, f2t, fi, fi2);
            F1 = (int)f1t; F2 = (int)f2t; first = (int)fi; first2 = (int)fi2;
        }
        FSTAMP(2)
        int Hout, E1out, E2out = inf;
        if (GAP == 1) {
            const int tmp = imax(h, E1v);
            Hout = imax(tmp, F1);
            const int en = imax(wr(E1v - e1), wr(Hout - oe1));
            E1out = (Hout == tmp) ? en : inf;
        } else {
            Hout = imax(hs, imax(F1, F2));
            E1out = imax(wr(E1v - e1), wr(Hout - oe1));
            E2out = imax(wr(E2v - e2), wr(Hout - oe2));
        }
        // one record store per lane, all 64 lanes (lanes past the band write into cells the NEXT row overwrites: same wave,
        // program order; the arena carries 64 records of slack at its end)
        const int he = (int)(((unsigned)Hout & 0xffffu) | ((unsigned)E1out << 16));      // int16: also the score-ring word
        // match flag for the backtrack (see turbo_body): compared without wrapping, as the reference's backtrack does (:130-160)
        const int mflag = (Mv + q == Hout && kb <= 64) ? kb : 0;
        if (ABL(1)) {}
        else if (I16 && GAP == 1) { int2 rec; rec.x = he; rec.y = (int)__builtin_amdgcn_perm((unsigned)mflag, (unsigned)F1, 0x05040100u); *(int2 *)(H + (long long)rel * CW) = rec; }
        else if (I16) { int4 rec; rec.x = he; rec.y = (int)(((unsigned)E2out & 0xffffu) | ((unsigned)F1 << 16)); rec.z = F2 & 0xffff; rec.w = mflag; *(int4 *)(H + (long long)rel * CW) = rec; }
        else if (GAP == 1) { int4 rec; rec.x = Hout; rec.y = E1out; rec.z = F1; rec.w = mflag; *(int4 *)(H + (long long)rel * CW) = rec; }
        else { int4 r0, r1; r0.x = Hout; r0.y = E1out; r0.z = E2out; r0.w = F1; r1.x = F2; r1.y = mflag; r1.z = 0; r1.w = 0; int4 *dst = (int4 *)(H + (long long)rel * CW); dst[0] = r0; dst[1] = r1; }
        if (to_ring && !ABL(2)) {
            int *qd = fr + my_slot + 2 + rel;
            if (I16) { qd[0] = in_band ? he : infw; if (GAP == 2) qd[RCS] = in_band ? E2out : inf; }
            else { qd[0] = in_band ? Hout : inf; qd[RCS] = in_band ? E1out : inf; if (GAP == 2) qd[2 * RCS] = in_band ? E2out : inf; }
        }
        if (!ABL(4)) {   // running arg-max candidate of this lane, reference :1043-1057
            const bool is_end = (v == end_sn);
            int cand = Hout;
            if (end_sn == qlen_sn) cand = (is_end && col > qlen) ? inf : cand;
            if (I16) {
                const unsigned key = ((unsigned)cand << 16) + (unsigned)(kconst - vb) + (is_end ? 2048u : 0u);
                am_key = (in_band && key > am_key) ? key : am_key;
            } else if (in_band && (!am_any || (is_end ? cand >= am_val : cand > am_val))) { am_val = cand; am_v = v; am_isend = is_end; am_any = true; }
        }
    };
    auto pad_ring = [&](int nch, int my_slot) __attribute__((always_inline)) {        // "inf" after the band, up to the ring width
        if (!ABL(2)) for (int c = nch; c < (RC >> 6); ++c) {
            int *qd = fr + my_slot + 2 + c * 64 + lane;
            qd[0] = infw; if (NPW > 1) qd[RCS] = inf; if (NPW > 2) qd[2 * RCS] = inf;
        }
    };
    // reserve the row's arena cells; false = overflow
    auto reserve = [&]() __attribute__((always_inline)) {
        const int nvr = end_sn - beg_sn + 1;
        if (cur + nvr * CW > cap_pn) return false;
        off_pn = cur; cur += nvr * CW;
        return true;
    };


    // ---- TURBO body: the dominant row shape as straight-line code -- 1 or 2 predecessors (both in the rings), band of at
    //      most 64 columns (one chunk), every vector takes the closed-form F scan (end_sn <= max_pre_end_sn), not the last
    //      query vector, and no value near the wrap limit.  Exactly two rarely-taken exits, both before any side effect.
    //      Returns 1 = done (mi set), 0 = not applicable.
    int mi = -1;
    const int lane4 = lane * 4;
    // arg-max key constants (normal / end_sn vector): lane residue, vector priority, and -- never decisive, it only saves the decoding -- the lane
    const int kN = (int)(0x80000000u | ((unsigned)(PN - 1 - l) << 12) | ((unsigned)(NV - 1 - vvl) << 8) | (unsigned)lane), kE = (int)(0x80000000u | ((unsigned)(PN - 1 - l) << 12) | (8u << 8) | (unsigned)lane);
    auto turbo_body = [&](auto npc, int row, int ti) __attribute__((always_inline)) -> int {
        constexpr int NPC = decltype(npc)::value;
        const int tb = __builtin_amdgcn_readlane(tv_tb, ti);
        const int p0 = row - (tb & 0xff);
        const int g0 = __builtin_amdgcn_readlane(vg_geo, p0 & 63), m0 = __builtin_amdgcn_readlane(vg_mi, p0 & 63);
        const int pb0 = g0 & 0xfff, pe0 = (g0 >> 12) & 0xfff;
        int mn = m0, mx = m0, min_pb = pb0, ring = g0; max_pe = pe0;
        int p1 = p0, g1 = g0, p2 = p0, g2 = g0, p3 = p0, g3 = g0;
        if (NPC >= 2) {
            p1 = row - ((tb >> 8) & 0xff);
            g1 = __builtin_amdgcn_readlane(vg_geo, p1 & 63); const int m1_ = __builtin_amdgcn_readlane(vg_mi, p1 & 63);
            mn = sgpr(imin(m0, m1_)); mx = sgpr(imax(m0, m1_)); min_pb = imin(pb0, g1 & 0xfff); max_pe = imax(pe0, (g1 >> 12) & 0xfff); ring &= g1;
        }
        if (NPC == 4) {      // three or four predecessors (np at run time); a missing fourth repeats the third
            p2 = __builtin_amdgcn_readlane(tv_p2, ti); p3 = np > 3 ? __builtin_amdgcn_readlane(tv_p3, ti) : p2;
            g2 = __builtin_amdgcn_readlane(vg_geo, p2 & 63); g3 = __builtin_amdgcn_readlane(vg_geo, p3 & 63);
            const int m2_ = __builtin_amdgcn_readlane(vg_mi, p2 & 63), m3_ = __builtin_amdgcn_readlane(vg_mi, p3 & 63);
            mn = sgpr(imin(mn, imin(m2_, m3_))); mx = sgpr(imax(mx, imax(m2_, m3_)));
            min_pb = imin(min_pb, imin(g2 & 0xfff, g3 & 0xfff)); max_pe = imax(max_pe, imax((g2 >> 12) & 0xfff, (g3 >> 12) & 0xfff)); ring &= g2 & g3;
        }
        set_band(std::true_type{}, mn, mx, min_pb);
        const int nvr = end_sn - beg_sn + 1;
        // all conditions as sign bits: (x <= y) <=> (x - y - 1) < 0
        // (arena room: checked for a full-width row, cap_turbo = cap_pn - NV * CW)
        const int okbits = (nvr - NV - 1) & (end_sn - max_pe - 1) & (cur - cap_turbo - 1) & (ring << 7);      // GEO_RING (bit 24) -> bit 31
        if (__builtin_expect(okbits >= 0, 0)) return 0;
        const int Wr = nvr * PN;
        if (__builtin_expect(beg_sn != qc_beg_sn, 0)) {
            qc_beg_sn = beg_sn;
            const int c0 = beg_sn * PN + lane, c1 = c0 + 64;
            qoff0 = (c0 >= 1 && c0 <= qlen) ? (int)s_query[c0 - 1] : m; qoff1 = (c1 >= 1 && c1 <= qlen) ? (int)s_query[c1 - 1] : m;
        }
        const int q = *(const int *)((const char *)s_mx + (tb >> 16) + qoff0 * 4);
        const int colrel = beg_sn * PN + lane;                     // this lane's column
        int Mv, E1v, E2v = inf, raw0, raw1, raw2 = inf;            // the first predecessor's words as loaded (decoded after the block below)
        {
            const int x = colrel - pb0 * PN;
            const int *src = ring_at(__builtin_amdgcn_readlane(vslot, p0), med3i(x - 1, -2, RC));
            if (I16) { raw0 = src[0]; raw1 = src[1]; if (GAP == 2) raw2 = src[RCS + 1]; }
            else { raw0 = src[0]; raw1 = src[RCS + 1]; if (GAP == 2) raw2 = src[2 * RCS + 1]; }
        }
        // the second predecessor's words go out with the first one's: both LDS reads are in flight together
        int rb0 = 0, rb1 = 0, rb2 = inf, x1 = 0, Wp1 = 0;
        int rc0 = 0, rc1 = 0, rc2 = inf, x2 = 0, Wp2 = 0, rd0 = 0, rd1 = 0, rd2 = inf, x3 = 0, Wp3 = 0;
        if (NPC == 4) {
            const int pb2 = g2 & 0xfff, pb3 = g3 & 0xfff; Wp2 = (((g2 >> 12) & 0xfff) - pb2 + 1) * PN; Wp3 = (((g3 >> 12) & 0xfff) - pb3 + 1) * PN;
            x2 = colrel - pb2 * PN; x3 = colrel - pb3 * PN;
            const int *s2 = ring_at(__builtin_amdgcn_readlane(vslot, p2), med3i(x2 - 1, -2, RC)), *s3 = ring_at(__builtin_amdgcn_readlane(vslot, p3), med3i(x3 - 1, -2, RC));
            if (I16) { rc0 = s2[0]; rc1 = s2[1]; rd0 = s3[0]; rd1 = s3[1]; if (GAP == 2) { rc2 = s2[RCS + 1]; rd2 = s3[RCS + 1]; } }
            else { rc0 = s2[0]; rc1 = s2[RCS + 1]; rd0 = s3[0]; rd1 = s3[RCS + 1]; if (GAP == 2) { rc2 = s2[2 * RCS + 1]; rd2 = s3[2 * RCS + 1]; } }
        }
        if (NPC >= 2) {
            const int pb1 = g1 & 0xfff; Wp1 = (((g1 >> 12) & 0xfff) - pb1 + 1) * PN;
            x1 = colrel - pb1 * PN;
            const int *src = ring_at(__builtin_amdgcn_readlane(vslot, p1), med3i(x1 - 1, -2, RC));
            if (I16) { rb0 = src[0]; rb1 = src[1]; if (GAP == 2) rb2 = src[RCS + 1]; }
            else { rb0 = src[0]; rb1 = src[RCS + 1]; if (GAP == 2) rb2 = src[2 * RCS + 1]; }
        }
        // work that does not depend on the loaded scores, placed here so that it runs while the LDS reads are in flight (the scheduling
        // barrier keeps the compiler from sinking it behind the wait): band mask, arg-max key constant, ring and arena addresses
        const bool in_band = lane < Wr;
        const int key_c = (vvl == nvr - 1) ? kE : kN;
        const int qd_addr = __builtin_amdgcn_readlane(vslot, ti) + 4 * lane;                                       // LDS byte address of this lane's ring cell
        const unsigned rec_off = (unsigned)(cur * (int)(PN * sizeof(T)) + lane * (int)(CW * sizeof(T)));            // arena byte offset of this lane's record (cur = the row's offset once committed)
        asm volatile("" :: "v"(key_c), "v"(qd_addr), "v"(rec_off));      // (materialised here, not sunk to their uses)
        __builtin_amdgcn_sched_barrier(0);
        if (I16) { Mv = (int)(short)raw0; E1v = raw1 >> 16; E2v = raw2; } else { Mv = raw0; E1v = raw1; E2v = raw2; }
        const int Mv_first = Mv;                                   // (match flag below: which predecessor supplies the diagonal)
        int kfirst = 1;                                            // 1 + index of the first predecessor that reaches the running maximum of H[.][col-1]
        auto merge_pred = [&](int r0_, int r1_, int r2_, int x_, int Wp_, int kidx) __attribute__((always_inline)) {
            int hm1, ev1, ev2 = inf;
            if (I16) { hm1 = (int)(short)r0_; ev1 = r1_ >> 16; ev2 = r2_; } else { hm1 = r0_; ev1 = r1_; ev2 = r2_; }
            const bool inH = (unsigned)x_ < (unsigned)(Wp_ + PN), inE = (unsigned)x_ < (unsigned)Wp_;
            if (NPC == 4) kfirst = (inH && hm1 > Mv) ? kidx : kfirst;
            Mv = inH ? imax(Mv, hm1) : Mv; E1v = inE ? imax(E1v, ev1) : E1v; if (GAP == 2) E2v = inE ? imax(E2v, ev2) : E2v;
        };
        if (NPC >= 2) {
            asm volatile("" : "+v"(rb0), "+v"(rb1));               // (the loads above stay unconditional)
            if (GAP == 2) asm volatile("" : "+v"(rb2));
            merge_pred(rb0, rb1, rb2, x1, Wp1, 2);
        }
        if (NPC == 4) {
            asm volatile("" : "+v"(rc0), "+v"(rc1), "+v"(rd0), "+v"(rd1));
            if (GAP == 2) asm volatile("" : "+v"(rc2), "+v"(rd2));
            merge_pred(rc0, rc1, rc2, x2, Wp2, 3);
            merge_pred(rd0, rd1, rd2, x3, Wp3, 4);                 // (np == 3: the third predecessor again -- no change, kfirst keeps 3 or less)
        }
        const int h = Mv + q;                                      // no wrap possible once the check below passes
        int lowest = imin(h, E1v); if (GAP == 2) lowest = imin(lowest, E2v);
        const bool near_wrap = __any(in_band && lowest < fast_lo);      // decided here, acted on after the scan below: the compare runs beside it, the
                                                                        // branch is off the row's dependent chain (nothing is stored before it)
        int hs = h; if (GAP == 2) hs = imax(imax(h, E1v), E2v);
        // lane 0's scan input is first - e (first = H of the band's first column before any E / F merge = h of lane 0): the shift leaves lane 0's
        // own h - e in place, no trip through an SGPR
        const int g1s = hs + le1;
        int F1 = imax(wave_scan_max_i32(wave_shr1(h - e1, g1s)) - cf1, inj1), F2 = inf;
        if (GAP == 2) { const int g2s = hs + le2; F2 = imax(wave_scan_max_i32(wave_shr1(h - e2, g2s)) - cf2, inj2); }
        if (__builtin_expect(near_wrap, 0)) return 0;
        // ---- from here on the row is committed
        off_pn = cur; cur += nvr * CW;
        int Hout, E1out, E2out = inf;
        if (GAP == 1) {
            const int tmp = imax(h, E1v);
            Hout = imax(tmp, F1);
            E1out = (Hout == tmp) ? imax(E1v - e1, Hout - oe1) : inf;
        } else {
            Hout = imax(hs, imax(F1, F2));
            E1out = imax(E1v - e1, Hout - oe1); E2out = imax(E2v - e2, Hout - oe2);
        }
        // record address = arena base + a 32-bit byte offset (an arena is far below 4 GB): one VALU add, no 64-bit pointer arithmetic per row
        T *const H = (T *)((char *)io.planes + (size_t)rec_off) - lane * CW;
        const int he = I16 ? (int)__builtin_amdgcn_perm((unsigned)E1out, (unsigned)Hout, 0x05040100u) : 0;      // H | E1 << 16 (int16: also the score-ring word)
        // match flag for the backtrack (spare slot of the record, finish_alignment PL_FLAG): 1 + index of the first predecessor k (list order) with
        // H[k][col-1] + q == H[col], 0 = none.  Only a predecessor that supplies the maximum Mv can satisfy it, and only when H == Mv + q.
        // (A predecessor value read from outside its band is `inf`: the backtrack re-checks the column range before it trusts the flag.)
        const int mflag = (h == Hout) ? (NPC == 4 ? kfirst : ((NPC == 2 && Mv != Mv_first) ? 2 : 1)) : 0;
        if (I16 && GAP == 1) { int2 rec; rec.x = he; rec.y = (int)__builtin_amdgcn_perm((unsigned)mflag, (unsigned)F1, 0x05040100u); *(int2 *)(H + lane * CW) = rec; }
        else if (I16) { int4 rec; rec.x = he; rec.y = (int)(((unsigned)E2out & 0xffffu) | ((unsigned)F1 << 16)); rec.z = F2; rec.w = mflag; *(int4 *)(H + lane * CW) = rec; }
        else if (GAP == 1) { int4 rec; rec.x = Hout; rec.y = E1out; rec.z = F1; rec.w = mflag; *(int4 *)(H + lane * CW) = rec; }
        else { int4 r0, r1; r0.x = Hout; r0.y = E1out; r0.z = E2out; r0.w = F1; r1.x = F2; r1.y = mflag; r1.z = 0; r1.w = 0; int4 *dst = (int4 *)(H + lane * CW); dst[0] = r0; dst[1] = r1; }
        {
            int *qd = (int *)ring_at(qd_addr, 0);
            if (I16) { qd[0] = in_band ? he : infw; if (GAP == 2) qd[RCS] = in_band ? E2out : inf; }
            else { qd[0] = in_band ? Hout : inf; qd[RCS] = in_band ? E1out : inf; if (GAP == 2) qd[2 * RCS] = in_band ? E2out : inf; }
            qd[64] = infw; if (NPW > 1) qd[RCS + 64] = inf; if (NPW > 2) qd[2 * RCS + 64] = inf;      // (RC <= 128 for these rows: tv_meta bit 17)
        }
        // ---- arg-max, reference :1043-1057: value, then lowest lane residue, then the end_sn vector, then the lowest vector
        if (I16) {
            const unsigned key = ((unsigned)Hout << 16) + (unsigned)key_c;
            // columns past the query end exist in the last query vector only: computed and stored like the others, never the row's arg-max (ref :1049-1056)
            const unsigned kb = wave_max_u32_s((in_band && colrel <= qlen) ? key : 0u);
            mi = ((int)(kb >> 16) - 32768 > inf) ? beg_sn * PN + (int)(kb & 63) : -1;      // the winning lane IS the column offset
        } else {
            const bool am_ok = in_band && colrel <= qlen;
            const int vmax = wave_max_i32_s(am_ok ? Hout : INT_MIN);
            const unsigned key = (am_ok && Hout == vmax) ? (((unsigned)(PN - 1 - l) << 12) | (unsigned)((vvl == nvr - 1) ? 8 : NV - 1 - vvl)) : 0u;
            const unsigned kb = wave_max_u32_s(key);
            const int vrel = (kb & 8) ? nvr - 1 : NV - 1 - (int)(kb & 7);
            mi = (vmax > inf) ? (beg_sn + vrel) * PN + (PN - 1 - (int)((kb >> 12) & 0xf)) : -1;
        }
        return 1;
    };

    // ---- FAST body: NP (1, 2, or up to 4 with run-time count) predecessors, all in the 64-row geometry ring and the score ring.
    //      Returns 0 = not applicable (nothing touched), 1 = done, 2 = arena overflow.
    auto fast_body = [&](auto npc, int row, int ti) __attribute__((always_inline)) -> int {
        constexpr int NPC = decltype(npc)::value;
        int pr[4], pgeo[4];
        pr[0] = __builtin_amdgcn_readlane(tv_p0, ti);
        pgeo[0] = __builtin_amdgcn_readlane(vg_geo, pr[0] & 63);
        int mn_mi = __builtin_amdgcn_readlane(vg_mi, pr[0] & 63), mx_mi = mn_mi, min_pb = pgeo[0] & 0xfff, allring = pgeo[0];
        max_pe = (pgeo[0] >> 12) & 0xfff;
        auto more = [&](int k, int tvp) __attribute__((always_inline)) {
            pr[k] = __builtin_amdgcn_readlane(tvp, ti); pgeo[k] = __builtin_amdgcn_readlane(vg_geo, pr[k] & 63);
            const int mi_ = __builtin_amdgcn_readlane(vg_mi, pr[k] & 63);
            mn_mi = imin(mn_mi, mi_); mx_mi = imax(mx_mi, mi_); min_pb = imin(min_pb, pgeo[k] & 0xfff); max_pe = imax(max_pe, (pgeo[k] >> 12) & 0xfff); allring &= pgeo[k];
        };
        if (NPC >= 2) more(1, tv_p1);
        if (NPC >= 4) { pr[2] = pr[1]; pgeo[2] = pgeo[1]; pr[3] = pr[1]; pgeo[3] = pgeo[1]; if (np > 2) more(2, tv_p2); if (np > 3) more(3, tv_p3); }
        set_band(std::true_type{}, mn_mi, mx_mi, min_pb);
        const int Wr = (end_sn - beg_sn + 1) * PN;
        if (!(allring & GEO_RING) || Wr > RC) return 0;
        FSTAMP(0)
        if (!reserve()) return 2;
        to_ring = true;
        T *H = io.planes + (long long)off_pn * PN;
        const int my_slot = (row & (RR - 1)) * (NPW * RCS);
        const int nch = (Wr + 63) >> 6;
        refresh_qc();
        const int *mrow = s_mx + base * m1;
        int first = 0, first2 = 0;
        for (int c = 0; c < nch; ++c) {
            const int col = beg_sn * PN + c * 64 + lane;
            int qc = c == 0 ? qoff0 : qoff1;
            if (c >= 2) qc = (col >= 1 && col <= qlen) ? (int)s_query[col - 1] : m;
            const int q = mrow[qc];
            int Mv = lane, E1v = inf, E2v = inf, kb = 0;
            if (!ABL(16)) from_ring(0, pr[0], pgeo[0], col, Mv, E1v, E2v, kb, 1);
            if (NPC >= 2 && !ABL(16)) from_ring(1, pr[1], pgeo[1], col, Mv, E1v, E2v, kb, 2);
            if (NPC >= 4) { if (np > 2) from_ring(2, pr[2], pgeo[2], col, Mv, E1v, E2v, kb, 3); if (np > 3) from_ring(3, pr[3], pgeo[3], col, Mv, E1v, E2v, kb, 4); }
            FSTAMP(1)
            chunk_tail(c, nch, Wr, Mv, E1v, E2v, q, kb, first, first2, H, my_slot);
        }
        pad_ring(nch, my_slot);
        return 1;
    };

    // ---- GENERAL body: any number of predecessors, any distance (HBM copies of geometry and score rows), any band width.
    //      Returns 1 = done, 2 = arena overflow.
    auto general_body = [&](int row, int ti) __attribute__((always_inline)) -> int {
        const int ps = __builtin_amdgcn_readlane(tv_ps, ti);
        int mn_mi = gn, mx_mi = -1, min_pb = 4095; max_pe = -1;
        for (int k = 0; k < np; ++k) {
            int g_, mi_, off_; geo_of(__builtin_amdgcn_readfirstlane(gld_i32(io.pred_row + ps + k)), row, g_, mi_, off_);
            mn_mi = imin(mn_mi, mi_); mx_mi = imax(mx_mi, mi_); min_pb = imin(min_pb, g_ & 0xfff); max_pe = imax(max_pe, (g_ >> 12) & 0xfff);
        }
        if (np == 0) min_pb = 0;
        set_band(std::false_type{}, mn_mi, mx_mi, min_pb);
        const int Wr = (end_sn - beg_sn + 1) * PN;
        if (!reserve()) return 2;
        to_ring = Wr <= RC;
        T *H = io.planes + (long long)off_pn * PN;
        const int my_slot = (row & (RR - 1)) * (NPW * RCS);
        const int nch = (Wr + 63) >> 6;
        refresh_qc();
        const int *mrow = s_mx + base * m1;
        int first = 0, first2 = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // HBM gathers below read cells this wave stored earlier
        for (int c = 0; c < nch; ++c) {
            const int rel = c * 64 + lane, col = beg_sn * PN + rel;
            const bool in_band = rel < Wr;
            int qc = c == 0 ? qoff0 : qoff1;
            if (c >= 2) qc = (col >= 1 && col <= qlen) ? (int)s_query[col - 1] : m;
            const int q = mrow[qc];
            int Mv = inf, E1v = inf, E2v = inf, kb = 0;
            for (int k = 0; k < np; ++k) {
                int g_, mi_, off_; const int p = __builtin_amdgcn_readfirstlane(gld_i32(io.pred_row + ps + k)); geo_of(p, row, g_, mi_, off_);
                if ((g_ & GEO_RING) && row - p < RR) {
                    if (k == 0) from_ring(0, p, g_, col, Mv, E1v, E2v, kb, 1); else from_ring(1, p, g_, col, Mv, E1v, E2v, kb, k + 1);
                } else {
                    const int pb = g_ & 0xfff, pe = (g_ >> 12) & 0xfff, Wp = (pe - pb + 1) * PN;
                    const int x = col - pb * PN;
                    const bool inH = in_band && (unsigned)x < (unsigned)(Wp + PN), inE = in_band && (unsigned)x < (unsigned)Wp;
                    const T *Hp = io.planes + (long long)(uint32_t)off_ * PN;
                    int hval = inf, ev1 = inf, ev2 = inf;
                    if (inH && (unsigned)(x - 1) < (unsigned)Wp) hval = gld_cell((GLOBAL_AS const T *)(Hp + (long long)(x - 1) * CW));
                    if (inE) { ev1 = gld_cell((GLOBAL_AS const T *)(Hp + (long long)x * CW + PL_E1)); if (GAP == 2) ev2 = gld_cell((GLOBAL_AS const T *)(Hp + (long long)x * CW + PL_E2)); }
                    if (k == 0) { Mv = hval; E1v = ev1; E2v = ev2; kb = 1; }
                    else { kb = (inH && hval > Mv) ? k + 1 : kb; Mv = inH ? imax(Mv, hval) : Mv; E1v = inE ? imax(E1v, ev1) : E1v; if (GAP == 2) E2v = inE ? imax(E2v, ev2) : E2v; }
                }
            }
            chunk_tail(c, nch, Wr, Mv, E1v, E2v, q, kb, first, first2, H, my_slot);
        }
        if (to_ring) pad_ring(nch, my_slot);
        return 1;
    };

#ifdef ABPOA_HIP_PROFILE
    fseg_last = (long long)__builtin_amdgcn_s_memtime();
#endif
    for (int t0 = 0; t0 < gn - 1 && status == 0; t0 += 64) {
        if (t0 > 0) {       // geometry of the finished tile goes to HBM in one coalesced burst (older predecessors, backtrack, trace)
            const int rb = t0 - 64 + lane; io.g_bsn[rb] = vg_geo & 0xfff; io.g_esn[rb] = (vg_geo >> 12) & 0xfff; io.g_coff[rb] = (long long)(uint32_t)vg_off * PN; io.row_max_i[rb] = vg_mi;
            if (rb >= 1) n_vec_lane += ((vg_geo >> 12) & 0xfff) - (vg_geo & 0xfff) + 1;
        }
        switch_tile(t0);
        const int r_hi = imin(t0 + 64, gn - 1);
        auto commit_row = [&](int ti, bool ring) __attribute__((always_inline)) {   // v_writelane x3 (no clang builtin); M0 = lane select (two different SGPRs would break the constant-bus limit)
            const int geo_new = sgpr(beg_sn | (end_sn << 12) | (ring ? GEO_RING : 0)), off_new = sgpr(off_pn); mi = sgpr(mi);
            asm volatile("s_mov_b32 m0, %6\n\ts_nop 3\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0"
                         : "+v"(vg_geo), "+v"(vg_mi), "+v"(vg_off) : "s"(geo_new), "s"(mi), "s"(off_new), "s"(ti) : "m0");
        };
        int row = imax(t0, 1);
        while (row < r_hi) {
            // ---- tight loop over consecutive straight-line rows: only these merge at its back edge (in one loop with the other row
            //      bodies every row paid ~30 register copies for the merge of all paths)
            int ok_ = 0;
            for (;;) {
                const int ti_ = row & 63;
                const int meta_ = __builtin_amdgcn_readlane(tv_meta, ti_);
                if (!__builtin_expect((meta_ >> 17) & 1, 1)) break;
                rterm = __builtin_amdgcn_readlane(tv_rterm, ti_);
                base = meta_ & 0xff; np = (meta_ >> 8) & 0xff;
                ok_ = np == 1 ? turbo_body(std::integral_constant<int, 1>{}, row, ti_) : turbo_body(std::integral_constant<int, 2>{}, row, ti_);
                if (__builtin_expect(ok_ != 1, 0)) break;
                commit_row(ti_, true);
                if (++row >= r_hi) break;
            }
            last_done = row - 1;
            if (row >= r_hi) break;
            const int ti = row & 63;
            last_done = row;
            const int meta = __builtin_amdgcn_readlane(tv_meta, ti);
            rterm = __builtin_amdgcn_readlane(tv_rterm, ti);
            base = meta & 0xff; np = (meta >> 8) & 0xff;
            if ((meta >> 18) & 1) {                                   // three or four predecessors: the straight-line body, outside the tight loop
                if (turbo_body(std::integral_constant<int, 4>{}, row, ti)) { commit_row(ti, true); ++row; continue; }
            }
            am_key = 0; am_val = INT_MIN; am_v = 0; am_isend = 0; am_any = false;
            int rc = 0;
            {
                if ((meta >> 16) & 1) {
                    if (np == 1) rc = fast_body(std::integral_constant<int, 1>{}, row, ti);
                    else if (np == 2) rc = fast_body(std::integral_constant<int, 2>{}, row, ti);
                    else rc = fast_body(std::integral_constant<int, 4>{}, row, ti);
                }
                if (rc == 0) rc = general_body(row, ti);
                if (rc == 2) { status = ABPOA_HIP_STATUS_OVERFLOW; break; }
                FSTAMP(3)
                // ---- row arg-max (tie-break: lowest lane residue, then the end_sn vector, then the lowest vector), reference :1043-1057
                mi = -1;
                if (I16) {
                    const unsigned kb = wave_max_u32_s(am_key);
                    const int vmax = (int)(kb >> 16) - 32768;
                    if (vmax > inf) { mi = (2047 - (int)(kb & 0x7ff)) * PN + (PN - 1 - (int)((kb >> 12) & 0xf)); if (mi > qlen) mi = -1; }
                } else {
                    const int vmax = wave_max_i32_s(am_any ? am_val : INT_MIN);
                    if (vmax > inf) {
                        unsigned key = 0;
                        if (am_any && am_val == vmax) key = ((unsigned)(PN - 1 - l) << 27) | ((unsigned)am_isend << 26) | (0x3FFFFFFu - (unsigned)am_v);
                        const unsigned kb = wave_max_u32_s(key);
                        mi = (int)(0x3FFFFFFu - (kb & 0x3FFFFFFu)) * PN + (PN - 1 - (int)(kb >> 27));
                        if (mi > qlen) mi = -1;
                    }
                }
            }
            commit_row(ti, to_ring);
            FSTAMP(4)
            ++row;
        }
        if (status != 0) break;
    }
    // ---- geometry of the last (partial) tile
    if (status == 0) {
        const int tb = last_done & ~63, rb = tb + lane;
        if (rb <= last_done) { io.g_bsn[rb] = vg_geo & 0xfff; io.g_esn[rb] = (vg_geo >> 12) & 0xfff; io.g_coff[rb] = (long long)(uint32_t)vg_off * PN; io.row_max_i[rb] = vg_mi;
                               if (rb >= 1) n_vec_lane += ((vg_geo >> 12) & 0xfff) - (vg_geo & 0xfff) + 1; }
    }
    __syncthreads();
    // ---- max_pos_left/right as the reference leaves them (only when the caller reads them back)
    if (status == 0 && b.want_lr) {
        for (int r = lane; r < gn; r += 64) {
            int lf = gn, rt = 0;
            if (r == 0) { lf = 0; rt = 0; }
            else for (int k = io.pred_off[r]; k < io.pred_off[r + 1]; ++k) {
                const int p = io.pred_row[k]; const int oi = (p == 0 ? 0 : io.row_max_i[p]) + 1;
                lf = imin(lf, oi); rt = imax(rt, oi);
            }
            io.g_left[r] = lf; io.g_right[r] = rt;
        }
    }
    cursor_out = (long long)cur * PN; n_cells_out = (long long)__builtin_amdgcn_readlane(wave_scan_add_i32(n_vec_lane), 63) * PN; rows_done_out = last_done;
}

// Everything after the row loop: global best (reference :1028-1041), backtrack (:109-429) and the result record.  Shared by the
// general kernel and the fast-loop kernel; reads only what the row loops left in HBM (planes, per-row band geometry).
struct TailState { long long cursor, n_cells, clk0, clk1, seg[6]; int status, rows_done, best_score, best_i, best_j; };

// CW = 0: plane-major arena rows (general kernel); CW > 0: cell records of CW values (fast loop, see rows_fast)
template <typename T, int GAP, int CW = 0>
__device__ __forceinline__ void finish_alignment(const DevBatch &b, const AlnDesc &d, AlnOut *out_rec, const TailState &ts) {
    constexpr int PN = Width<T>::PN;
    constexpr int P = CW > 0 ? CW : (GAP == 0 ? 1 : (GAP == 1 ? 3 : 5));      // values per column in the arena
    constexpr int PL_E1 = 1, PL_E2 = 2, PL_F1 = GAP == 1 ? 2 : 3, PL_F2 = 4;
    constexpr int PL_FLAG = GAP == 1 ? 3 : (sizeof(T) == 2 ? 6 : 5);      // cell records only: the row loop's match flag (0 = not known)
    const int lane = threadIdx.x & 63;
    const int gn = d.n_rows, qlen = d.qlen, m = b.m;
    const bool local = b.align_mode == ABPOA_HIP_LOCAL_MODE;
    const bool banded = b.wb >= 0;
    const T e1 = (T)b.e1, oe1 = (T)(b.o1 + b.e1), e2 = (T)b.e2, oe2 = (T)(b.o2 + b.e2);
    GLOBAL_AS const uint8_t *g_query = vgpr_ptr(b.query + d.query_off);
    GLOBAL_AS const uint8_t *row_base = vgpr_ptr(b.row_base + d.row0);
    GLOBAL_AS const int32_t *row_node_id = vgpr_ptr(b.row_node_id + d.row0);
    GLOBAL_AS const int32_t *pred_off = vgpr_ptr(b.pred_off + d.poff0), *pred_row = vgpr_ptr(b.pred_row + d.pred0);
    GLOBAL_AS int32_t *g_bsn = vgpr_ptr(b.dp_beg_sn + d.row0), *g_esn = vgpr_ptr(b.dp_end_sn + d.row0);
    GLOBAL_AS int64_t *g_coff = vgpr_ptr(b.row_cell_off + d.row0);
    T *planes = (T *)(b.planes + d.plane_off);
    uint8_t *s_query = lds_raw + b.lds.q_off;
    int32_t *s_mat = (int32_t *)(lds_raw + b.lds.mat_off);
    const bool q_in_lds = qlen <= b.lds.q_cap;
    auto dp_end_of = [&](int row, int end_sn_row) __attribute__((always_inline)) { return (banded || row == 0) ? (end_sn_row + 1) * PN - 1 : qlen; };
    int status = ts.status, best_score = ts.best_score, best_i = ts.best_i, best_j = ts.best_j, bt_steps = 0;
    const long long cursor = ts.cursor, n_cells = ts.n_cells, clk0 = ts.clk0, clk1 = ts.clk1; const int rows_done = ts.rows_done;
    const long long *seg = ts.seg;
    __syncthreads();       // all of this wave's plane / band stores have landed before the loads below

    // ------------------------------------------------------------------ global best, reference :1028-1041
    if (status == 0 && b.align_mode == ABPOA_HIP_GLOBAL_MODE) {
        for (int k = pred_off[gn - 1]; k < pred_off[gn]; ++k) {
            int in_row = pred_row[k];
            int pe = g_esn[in_row], pb = g_bsn[in_row];
            int dpe = dp_end_of(in_row, pe);
            int end = qlen > dpe ? dpe : qlen;
            int score = (int)planes[g_coff[in_row] + (long long)(end - pb * PN) * (CW > 0 ? CW : 1)];
            if (score > best_score) { best_score = score; best_i = in_row; best_j = end; }
        }
    }

    // ------------------------------------------------------------------ backtrack, reference :109-429
    // The walk is executed redundantly (uniformly) by all lanes so that the LDS window of the arena can be
    // refilled cooperatively; only lane 0 writes cigar words.
    int n_cigar = 0, node_s = 0, node_e = 0, query_s = 0, query_e = 0, n_aln = 0, n_match = 0;
    long long bt_win_ticks = 0, bt_n_windows = 0, bt_slow_steps = 0, bt_wa = 0, bt_wb = 0, bt_flag_steps = 0;
    if (status == 0 && b.ret_cigar) {
        BtLds &B = *(BtLds *)(lds_raw + b.lds.phase_off);
        T *bt = (T *)(lds_raw + b.lds.phase_off + b.lds.bt_off);
        const long long bt_cells = b.lds.bt_bytes / (int)sizeof(T);
        int bt_lo = 1, bt_hi = 0, bt_pbase = 0, bt_margin = 0;   // window = rows [bt_lo, bt_hi], empty at start
        long long bt_c0 = 0;                                     // arena cell of B.coff[0]
        GLOBAL_AS uint64_t *cg = vgpr_ptr(b.cigar + d.cigar_off);
        const int cap = d.cigar_cap;
        const bool cap_safe = cap >= gn + qlen + 2;               // a walk emits at most one word per row or column it leaves: no per-step capacity check needed
        uint64_t last_word = 0;
        long long win_ticks = 0, win_a = 0; int n_windows = 0;
        auto load_window = [&](int hi) __attribute__((always_inline)) {
            const long long tw0 = (long long)__builtin_amdgcn_s_memtime(); ++n_windows;
            __syncthreads();
            int lo = imax(0, hi - BTR + 1);
            const int r = lo + lane;
            int my_b = -1, my_e = -1; long long my_c = 0;
            if (r <= hi) { my_b = g_bsn[r]; my_e = g_esn[r]; my_c = g_coff[r]; }
            // end of row hi = its offset + P * width (never-computed rows carry zero width)
            const int hb = g_bsn[hi], he = g_esn[hi];
            const long long c_end = g_coff[hi] + (hb >= 0 ? (long long)(he - hb + 1) * PN * P : 0);
            // smallest lo' whose segment [coff[lo'], c_end) fits the LDS tile
            const bool fits = (r <= hi) && (c_end - my_c) <= bt_cells;
            const unsigned long long mk = __ballot(fits);
            const int sh = mk ? __builtin_ctzll(mk) : (hi - lo);     // worst case: a single row (may still not fit -> HBM path)
            lo += sh;
            if (r >= lo && r <= hi) {
                const int i = r - lo;
                B.bsn[i] = my_b; B.esn[i] = my_e; B.coff[i] = my_c;
                B.poff[i] = pred_off[r]; B.nid[i] = row_node_id[r]; B.base[i] = row_base[r];
            }
            if (lane == 0) { B.coff[hi - lo + 1] = c_end; B.poff[hi - lo + 1] = pred_off[hi + 1]; }
            __syncthreads();
            bt_lo = lo; bt_hi = hi; bt_c0 = B.coff[0]; bt_pbase = B.poff[0]; bt_margin = imin(4, (hi - lo) / 2);
            const int pn_t = imin(BTP, B.poff[hi - lo + 1] - bt_pbase);
            for (int i = lane; i < pn_t; i += 64) B.pred[i] = pred_row[bt_pbase + i];
            long long ncell = c_end - bt_c0; if (ncell > bt_cells) ncell = 0;        // does not fit: leave the tile empty
            if (ncell == 0) { bt_hi = bt_lo - 1; }
            // 16-byte coalesced copy (arena offsets are multiples of PN cells = 32 bytes)
            const int4 *src = (const int4 *)(planes + bt_c0); int4 *dst = (int4 *)bt;
            const int n16 = (int)(ncell * (long long)sizeof(T) / 16);
            for (int i0 = 0; i0 < n16; i0 += 64 * 8) {                            // 8 loads in flight per lane, then 8 LDS stores
                int4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int idx = i0 + u * 64 + lane; v[u] = idx < n16 ? src[idx] : make_int4(0, 0, 0, 0); }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int idx = i0 + u * 64 + lane; if (idx < n16) dst[idx] = v[u]; }
            }
            __syncthreads();
            win_ticks += (long long)__builtin_amdgcn_s_memtime() - tw0;
        };
        // ---- window of the lane-parallel walk (cell-record arenas): rows [hi - R + 1, hi] x columns [jtop - WC + 1, jtop].  The walk
        //      moves up-left (each step: a predecessor row and / or one column back), so a column SLICE of every row is enough;
        //      with 10 kb reads a whole row is 4-8 KB and whole-row staging would hold 3-6 rows.  Leaving the slice (a long
        //      insertion run) simply reloads the window at the current cell.
        int win_i = -1, win_j = -1;                               // cell the current window was loaded for
        bool win_narrow = false;                                  // the window holds whole rows (every cell of a window row is staged)
        auto load_window_cols = [&](int hi, int jtop) __attribute__((always_inline)) {
            const long long tw0 = (long long)__builtin_amdgcn_s_memtime(); ++n_windows;
            __syncthreads();
            const int max_rec = (int)(bt_cells / (CW > 0 ? CW : 1));
            const int WC = max_rec >= 2048 ? 64 : 48;
            // candidate rows: the 64 rows ending at hi (lane = row - lo64); how many of them are staged is decided below
            const int lo64 = imax(0, hi - BTR + 1), n64 = hi - lo64 + 1;
            const int r = lo64 + lane; const bool rv64 = lane < n64;
            int b_ = -1, e_ = -1, po = 0, po1 = 0, nid_ = 0, bs_ = 0; long long c_ = 0;
            if (rv64) { b_ = g_bsn[r]; e_ = g_esn[r]; c_ = g_coff[r]; po = pred_off[r]; po1 = pred_off[r + 1]; nid_ = row_node_id[r]; bs_ = row_base[r]; }
            const int pbc = b_ >= 0 ? b_ * PN : 0, W = b_ >= 0 ? (e_ - b_ + 1) * PN : 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const long long tw1 = (long long)__builtin_amdgcn_s_memtime(); win_a += tw1 - tw0;
            // whole rows if at least 16 of them fit (narrow bands; rows are adjacent in the arena -> one contiguous copy), else slices
            const int Wrev = __builtin_amdgcn_ds_bpermute((n64 - 1 - lane) << 2, rv64 ? W : 0x100000);      // lane l <- row hi - l
            const int cum = wave_scan_add_i32(lane < n64 ? Wrev : 0x100000);
            const int r_full = __builtin_popcountll(__ballot(cum <= max_rec));
            const bool narrow = r_full >= imin(16, n64);
            const int R = narrow ? r_full : imin(n64, imax(4, max_rec / WC));
            const int lo = hi - R + 1, nrow = R, li = lane - (lo - lo64);           // li: index of this lane's row inside the window
            const bool rv = rv64 && li >= 0;
            const int sl = narrow ? pbc : imax(pbc, jtop - WC + 1), sh = narrow ? pbc + W : imin(pbc + W, jtop + 1), ns = rv ? imax(0, sh - sl) : 0;
            const int incl = wave_scan_add_i32(ns);
            const int off_rec = incl - ns;
            const int pbase = __builtin_amdgcn_readlane(po, lo - lo64);
            const int pn_t = imin(BTP, __builtin_amdgcn_readlane(po1, n64 - 1) - pbase);
            if (rv) {
                B.rinfo[li] = make_int4(pbc | (W << 16), off_rec * CW, ((po - pbase) & 0xffff) | (((po1 - po > 64 || po1 - pbase > BTP) ? 255 : po1 - po) << 16) | (bs_ << 24), nid_);      // n_pred 255: not for the lane-parallel steps
                B.rinfo2[li] = sl | (ns << 16);
                B.srcoff[li] = c_ + (long long)(sl - pbc) * CW;
            }
            __syncthreads();
            const long long tw1b = (long long)__builtin_amdgcn_s_memtime(); win_a += tw1b - tw1;       // (debug split: scan + LDS tables)
            // the window's predecessor rows travel with the cell copy below (issued here, waited for with the first batch of cells)
            int prv[BTP / 64];
#pragma unroll
            for (int k_ = 0; k_ < BTP / 64; ++k_) { const int e_ = k_ * 64 + lane; gld_async(prv[k_], (const int32_t *)pred_row + pbase + (e_ < pn_t ? e_ : 0)); }
            // staged records: 8 rows per batch, lane = column inside the slice
            typedef typename std::conditional<(CW * sizeof(T) == 8), int2, int4>::type RecT;       // 8-byte or 16-byte pieces (CW * sizeof(T) = 8, 16 or 32)
            constexpr int PIECES = (int)(CW * sizeof(T) / sizeof(RecT));
            // narrow bands: every slice is a whole row, and the rows are adjacent in the arena -> one contiguous 16-byte-wide copy
            if (narrow) {
                const int l0 = lo - lo64;
                const long long c_lo = (long long)(unsigned)__builtin_amdgcn_readlane((int)(c_ & 0xffffffffll), l0) | ((long long)__builtin_amdgcn_readlane((int)(c_ >> 32), l0) << 32);
                const int n16 = (int)((long long)__builtin_amdgcn_readlane(incl, 63) * CW * (int)sizeof(T) / 16);
                const int4 *src = (const int4 *)(planes + c_lo); int4 *dst = (int4 *)bt;
                constexpr int NB = 24;                       // 24 x 64 lanes x 16 bytes = the whole 24 KB window in one HBM round trip
                for (int i0 = 0; i0 < n16; i0 += 64 * NB) {
                    int4 v[NB];
#pragma unroll
                    for (int u = 0; u < NB; ++u) { const int idx = i0 + u * 64 + lane; gld_async(v[u], src + (idx < n16 ? idx : 0)); }
                    gld_wait();
#pragma unroll
                    for (int u = 0; u < NB; ++u) { const int idx = i0 + u * 64 + lane; if (idx < n16) dst[idx] = v[u]; }
                }
            } else {
                // slices: lane = column inside the slice, 16 rows per batch; the per-row constants travel by v_readlane, not through LDS
                const int offv = off_rec * CW; const long long srcv = c_ + (long long)(sl - pbc) * CW;
                const int src_lo = (int)(srcv & 0xffffffffll), src_hi = (int)(srcv >> 32);
                for (int r0 = 0; r0 < nrow; r0 += 16) {
                    RecT v[16][PIECES]; int nn[16], oo[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int rr = imin(r0 + u, nrow - 1) + (lo - lo64);
                        nn[u] = (r0 + u < nrow) ? __builtin_amdgcn_readlane(ns, rr) : 0; oo[u] = __builtin_amdgcn_readlane(offv, rr);
                        const long long so = (long long)(unsigned)__builtin_amdgcn_readlane(src_lo, rr) | ((long long)__builtin_amdgcn_readlane(src_hi, rr) << 32);
                        const RecT *src = (const RecT *)(planes + so) + (long long)(lane < nn[u] ? lane : 0) * PIECES;      // unconditional loads (a
#pragma unroll                                                                                                           // conditional one is waited for at once)
                        for (int q_ = 0; q_ < PIECES; ++q_) gld_async(v[u][q_], src + q_);
                    }
                    gld_wait();                             // all 16 loads in flight, one wait (hipcc pairs load / wait / store otherwise)
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        RecT *dst = (RecT *)(bt + oo[u]) + lane * PIECES;
#pragma unroll
                        for (int q_ = 0; q_ < PIECES; ++q_) if (lane < nn[u]) dst[q_] = v[u][q_];
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            gld_wait();                                     // (a window without cells to copy still has predecessor rows in flight)
#pragma unroll
            for (int k_ = 0; k_ < BTP / 64; ++k_) {
                const int e = k_ * 64 + lane; if (e >= pn_t) continue;
                const int pr_ = prv[k_]; const bool ok = pr_ >= lo && pr_ <= hi;
                const int4 ri_ = B.rinfo[ok ? pr_ - lo : 0];
                B.edge[e] = make_int4(pr_, ok ? ri_.x : 0, ri_.y, ok ? 1 : 0); B.edge2[e] = make_int4(ri_.z, ri_.w, B.rinfo2[ok ? pr_ - lo : 0], 0);      // (not staged: empty band)
            }
            __syncthreads();
            bt_lo = lo; bt_hi = hi; bt_pbase = pbase; win_i = hi; win_j = jtop; win_narrow = narrow;
            win_ticks += (long long)__builtin_amdgcn_s_memtime() - tw0;
        };
        // cigar words are collected 64 at a time in a VGPR pair (lane = word index & 63) and written out as one coalesced store per 64
        // words: a store per step would sit in the memory pipeline when the next step's LDS reads are issued, and the compiler's
        // s_waitcnt vmcnt(0) in front of those reads then costs a full HBM write round trip per step
        int cgw_lo = 0, cgw_hi = 0;
        auto flush_cigar = [&](int base, int n) __attribute__((always_inline)) {
            if (lane < n) cg[base + lane] = ((uint64_t)(unsigned)cgw_hi << 32) | (uint64_t)(unsigned)cgw_lo;
        };
        auto push = [&](int op, int len, int node_id, int query_id) __attribute__((always_inline)) {      // reference abpoa_align.h:54-73
            uint64_t L = (uint64_t)(int64_t)len;
            if (n_cigar == 0 || op != ABPOA_HIP_CINS || op != (int)(last_word & 0xf)) {
                if (n_cigar >= cap) { status = ABPOA_HIP_EBACKTRACK; return; }
                if (n_cigar > 0 && (n_cigar & 63) == 0) flush_cigar(n_cigar - 64, 64);      // the previous 64 words are final now
                uint64_t n_id = (uint64_t)(int64_t)node_id, q_id = (uint64_t)(int64_t)query_id, wv;
                if (op == ABPOA_HIP_CMATCH) wv = n_id << 34 | q_id << 4 | (uint64_t)op;
                else if (op == ABPOA_HIP_CINS) wv = q_id << 34 | L << 4 | (uint64_t)op;
                else wv = n_id << 34 | L << 4 | (uint64_t)op;
                last_word = wv; ++n_cigar;
            } else last_word += L << 4;
            const int w_lo = sgpr((int)(last_word & 0xffffffffull)), w_hi = sgpr((int)(last_word >> 32)), w_idx = sgpr((n_cigar - 1) & 63);
            asm volatile("s_mov_b32 m0, %4\n\ts_nop 3\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                         : "+v"(cgw_lo), "+v"(cgw_hi) : "s"(w_lo), "s"(w_hi), "s"(w_idx) : "m0");
        };
        struct Geo { int pb, pe; long long off; bool in_tile; };
        auto geo_of = [&](int row_) __attribute__((always_inline)) {
            Geo g;
            g.in_tile = CW == 0 && row_ >= bt_lo && row_ <= bt_hi;      // (cell-record arenas: the LDS window holds column slices for the lane-parallel walk only)
            const int i = g.in_tile ? row_ - bt_lo : 0;
            g.pb = B.bsn[i]; g.pe = B.esn[i]; g.off = B.coff[i] - bt_c0;
            if (!g.in_tile) { g.pb = gld_i32(g_bsn + row_); g.pe = gld_i32(g_esn + row_); g.off = gld_i64(g_coff + row_); }
            return g;
        };
        auto cell = [&](const Geo &g, int plane, int col_) __attribute__((always_inline)) -> int {
            const long long Wp = (long long)(g.pe - g.pb + 1) * PN;
            const long long idx = CW > 0 ? g.off + (long long)(col_ - g.pb * PN) * CW + plane : g.off + plane * Wp + (col_ - g.pb * PN);
            int v = (int)bt[g.in_tile ? idx : 0];
            if (!g.in_tile) v = gld_cell((GLOBAL_AS const T *)(planes + idx));
            return v;
        };
        auto in_range = [&](const Geo &g, int row_, int col_) __attribute__((always_inline)) { return col_ >= g.pb * PN && col_ <= dp_end_of(row_, g.pe); };
        auto stored = [&](const Geo &g, int col_) __attribute__((always_inline)) { return col_ >= g.pb * PN && col_ <= (g.pe + 1) * PN - 1; };
        auto qcode = [&](int j_) __attribute__((always_inline)) { int v = (int)s_query[q_in_lds ? j_ : 0]; if (!q_in_lds) v = gld_u8(g_query + j_); return v; };

        int i = best_i, j = best_j, start_i = best_i, start_j = best_j, cur_op = OP_ALL, indel_first = 1;
        if (best_j < qlen) push(ABPOA_HIP_CINS, qlen - j, -1, qlen - 1);
        // ---- lane-parallel step (cell-record arenas, i.e. the fast path): one LDS round trip each for (a) the row's own record,
        //      (b) its predecessor list, (c) the predecessors' geometry, (d) every score the decision can need -- lane k holds
        //      predecessor k -- then the reference's priority order (:109-429) is evaluated on ballot masks.  Falls through to
        //      the one-read-at-a-time walk below whenever a predecessor is outside the staged window or the row has > 64 of them.
        bool bt_walk_narrow = true;                                // false once a window had to be staged as column slices
        // the walk's state is the same in every lane; values that came out of vector loads (best cell, the one-read-at-a-time step) are
        // moved to SGPRs so that the step loops below run on scalar branches
        auto uniformize = [&]() __attribute__((always_inline)) {
            i = sgpr(i); j = sgpr(j); cur_op = sgpr(cur_op); indel_first = sgpr(indel_first); status = sgpr(status); n_cigar = sgpr(n_cigar);
            n_aln = sgpr(n_aln); n_match = sgpr(n_match); start_i = sgpr(start_i); start_j = sgpr(start_j); bt_steps = sgpr(bt_steps);
            bt_lo = sgpr(bt_lo); bt_hi = sgpr(bt_hi); bt_pbase = sgpr(bt_pbase); win_i = sgpr(win_i); win_j = sgpr(win_j);
            last_word = (uint64_t)(unsigned)sgpr((int)(last_word & 0xffffffffull)) | ((uint64_t)(unsigned)sgpr((int)(last_word >> 32)) << 32);
        };
        uniformize();
        const long long t_walk0 = (long long)__builtin_amdgcn_s_memtime();
        do {      // fast steps; one slow step whenever a fast one cannot be taken; back to fast steps
        // ---- lane-parallel step (cell-record arenas, i.e. the fast path).  The current row's record is carried in SGPRs; round
        //      trip 1 fetches its predecessor edge records (lane k = predecessor k), its own cells and the query code, round trip 2
        //      the predecessors' cells and the substitution score; the reference's priority order (:109-429) is then evaluated on
        //      ballot masks and the chosen predecessor's record becomes the current one.  Falls through to ONE step of the
        //      one-read-at-a-time walk below whenever a predecessor is outside the staged window or the row has > 64 of them.
        int4 cr = make_int4(0, 0, 0, 0); int cr2 = 0, cr_row = -1;               // rinfo / rinfo2 of row cr_row
        // two copies of the step loop: whole-row windows (narrow bands: no slice bookkeeping at all) and column-slice windows
        while (CW > 0 && i > 0 && j > 0 && status == 0 && bt_walk_narrow) {
            if (i > bt_hi || i < bt_lo) { load_window_cols(i, j); cr_row = -1; if (!win_narrow) { bt_walk_narrow = false; break; } }
            if (cr_row != i) { cr = uniform4(B.rinfo[i - bt_lo]); cr_row = i; }
            // ---- match run.  The row loop left "1 + index of the first predecessor whose diagonal cell gives H" in every cell record it
            //      wrote on its straight-line path (0 = not known).  While a match is what the reference tries first (:130-160 with M allowed
            //      and indel_first == 0) and the flag is set, a step is ONE LDS round trip (flag, query code, the row's edge records) and a
            //      handful of scalar instructions; anything else leaves the loop for the full step below.
            if ((cur_op & OP_M) && indel_first == 0 && q_in_lds && cap_safe) {
                int mi_ = i, mj = j, pi_ = i, nm_v = 0, w_lo = 0, w_hi = 0; int4 mc = cr;
                const int nc0 = n_cigar;
                int slots = n_cigar == 0 ? 64 : ((64 - (n_cigar & 63)) & 63);           // words that still fit before the VGPR pair has to be written out
                for (;;) {
                    const int si_ = mj - (mc.x & 0xffff), np_ = (mc.z >> 16) & 0xff;
                    if ((unsigned)si_ >= ((unsigned)mc.x >> 16)) break;                  // (cannot happen on a sane path: the cell lies in its row's band)
                    int fl_v = (int)bt[mc.y + si_ * CW + PL_FLAG];
                    int qc_v = (int)s_query[mj - 1];
                    const int e_idx = (mc.z & 0xffff) + (lane < np_ ? lane : 0);        // (n_pred 255 = row not eligible: the reads stay inside the LDS image, the result is not used)
                    int4 er = B.edge[e_idx & (BTP - 1)]; int4 er2 = B.edge2[e_idx & (BTP - 1)];
                    // this step's cigar word, lane slot and flush test need nothing from the loads: computed while they are in flight
                    const int w_lo_n = sgpr(((mj - 1) << 4) | ABPOA_HIP_CMATCH), w_hi_n = sgpr(mc.w << 2), w_idx_n = sgpr(n_cigar & 63), bs_n = sgpr((int)((unsigned)mc.z >> 24));
                    asm volatile("" :: "s"(w_lo_n), "s"(w_hi_n), "s"(w_idx_n), "s"(bs_n));
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("" : "+v"(fl_v), "+v"(qc_v), "+v"(er.x), "+v"(er.y), "+v"(er.z), "+v"(er2.x), "+v"(er2.y));      // every load issued before the one wait
                    const int fl = __builtin_amdgcn_readfirstlane(fl_v);
                    const int ks = (fl - 1) & 63, ery = __builtin_amdgcn_readlane(er.y, ks);
                    // flag known and in range, row eligible, column j-1 inside that predecessor's band (empty when it is not staged)
                    if (!((unsigned)(fl - 1) < (unsigned)np_ && np_ != 255 && (unsigned)(mj - 1 - (ery & 0xffff)) < ((unsigned)ery >> 16))) break;
                    if (slots == 0) { flush_cigar(n_cigar - 64, 64); slots = 64; }
                    --slots;
                    w_lo = w_lo_n; w_hi = w_hi_n;                                                                 // node id << 34 | query index << 4 | op
                    asm volatile("s_mov_b32 m0, %4\n\ts_nop 3\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0" : "+v"(cgw_lo), "+v"(cgw_hi) : "s"(w_lo), "s"(w_hi), "s"(w_idx_n) : "m0");
                    ++n_cigar; nm_v += (qc_v == bs_n) ? 1 : 0;
                    pi_ = mi_; --mj;
                    mi_ = __builtin_amdgcn_readlane(er.x, ks);
                    mc = make_int4(ery, __builtin_amdgcn_readlane(er.z, ks), __builtin_amdgcn_readlane(er2.x, ks), __builtin_amdgcn_readlane(er2.y, ks));
                    if (imin(mi_, mj) <= 0) break;
                }
                const int moved = n_cigar - nc0;
                if (moved) {
                    start_i = pi_; start_j = mj + 1; bt_steps += moved; bt_flag_steps += moved; n_aln += moved; n_match += __builtin_amdgcn_readfirstlane(nm_v);
                    last_word = ((uint64_t)(unsigned)w_hi << 32) | (uint64_t)(unsigned)w_lo; cur_op = OP_ALL;
                    i = mi_; j = mj; cr = mc; cr_row = i;
                    if (i <= 0 || j <= 0) continue;
                }
            }
            const int pbi = cr.x & 0xffff, Wi = (int)((unsigned)cr.x >> 16), offi = cr.y;
            const int eb = cr.z & 0xffff, np = (cr.z >> 16) & 0xff, bs_ = (int)((unsigned)cr.z >> 24), id = cr.w;
            const int sli = pbi, nsi = Wi;                                      // whole rows are staged
            if (np > 64 || eb + np > BTP) break;
            // round trip 1: predecessor edge records, own cells, query code
            const int4 er = B.edge[eb + (lane < np ? lane : 0)]; const int4 er2 = B.edge2[eb + (lane < np ? lane : 0)];
            const int xi = j - pbi, si = j - sli;
            const bool st_jm1 = xi - 1 >= 0 && xi - 1 < Wi;                      // stored(gi, j-1)
            bool need = false;
            const T *ri = bt + offi + ((unsigned)si < (unsigned)nsi ? si : 0) * CW;
            const int Hij = __builtin_amdgcn_readfirstlane((int)ri[0]), E1ij = __builtin_amdgcn_readfirstlane((int)ri[PL_E1]), E2ij = GAP == 2 ? __builtin_amdgcn_readfirstlane((int)ri[PL_E2]) : 0, F1ij = __builtin_amdgcn_readfirstlane((int)ri[PL_F1]), F2ij = GAP == 2 ? __builtin_amdgcn_readfirstlane((int)ri[PL_F2]) : 0;      // (every lane reads the same cell: keep the walk's state in SGPRs)
            const T *rim1 = bt + offi + ((st_jm1 && si - 1 >= 0) ? si - 1 : 0) * CW;
            const int Hijm1 = __builtin_amdgcn_readfirstlane((int)rim1[0]), F1ijm1 = __builtin_amdgcn_readfirstlane((int)rim1[PL_F1]), F2ijm1 = GAP == 2 ? __builtin_amdgcn_readfirstlane((int)rim1[PL_F2]) : 0;
            const int qc = __builtin_amdgcn_readfirstlane(qcode(j - 1));
            const bool act = lane < np;
            // round trip 2: the predecessors' cells (lane k = predecessor k) and the substitution score
            const int pbk = er.y & 0xffff, Wk = (int)((unsigned)er.y >> 16), xk = j - pbk;
            const int sk = xk, nsk = Wk;
            const bool in_j = act && (unsigned)xk < (unsigned)Wk, in_jm1 = act && (unsigned)(xk - 1) < (unsigned)Wk;
            need = __any(act && er.w == 0);
            if (need) {                                                          // something this step reads is not staged: re-centre the window on (i, j) once
                if (win_i == i && win_j == j) break;
                load_window_cols(i, j); cr_row = -1; if (!win_narrow) { bt_walk_narrow = false; break; } continue;
            }
            const T *rk = bt + er.z + (in_j ? sk : 0) * CW, *rkm1 = bt + er.z + (in_jm1 ? sk - 1 : 0) * CW;
            const int Hk_j = (int)rk[0], E1k_j = (int)rk[PL_E1], E2k_j = GAP == 2 ? (int)rk[PL_E2] : 0, Hk_jm1 = (int)rkm1[0];
            const int sc_ = s_mat[m * bs_ + qc];
            start_i = i; start_j = j; ++bt_steps;
            const unsigned long long mA = __ballot(in_jm1 && Hk_jm1 + sc_ == Hij);
            int hit = 0, k_sel = -1;
            auto do_match = [&](int set_indel) __attribute__((always_inline)) {
                if (!mA) return;
                k_sel = __builtin_ctzll(mA);
                cur_op = OP_ALL; hit = 1;
                push(ABPOA_HIP_CMATCH, 1, id, j - 1);
                --j; ++n_aln; n_match += (bs_ == qc);
                if (set_indel) indel_first = 0;
            };
            if ((cur_op & OP_M) && indel_first == 0) do_match(0);
            if (!hit && (cur_op & OP_E)) {
                const bool viaM = cur_op & OP_M;
                unsigned long long m1 = 0, m2 = 0;
                if (cur_op & OP_E1) m1 = __ballot(in_j && (viaM ? Hij == E1k_j : E1ij == E1k_j - (int)e1));
                if (GAP == 2 && (cur_op & OP_E2)) m2 = __ballot(in_j && (viaM ? Hij == E2k_j : E2ij == E2k_j - (int)e2));
                if (m1 | m2) {                                                   // first predecessor in list order, E1 before E2 for the same one
                    const int k1 = m1 ? __builtin_ctzll(m1) : 64, k2 = m2 ? __builtin_ctzll(m2) : 64;
                    const bool use1 = k1 <= k2; k_sel = use1 ? k1 : k2;
                    const unsigned long long mD = __ballot(in_j && (use1 ? Hk_j - (int)oe1 == E1k_j : Hk_j - (int)oe2 == E2k_j));
                    cur_op = ((mD >> k_sel) & 1) ? (OP_M | OP_F) : (use1 ? OP_E1 : OP_E2);
                    hit = 1; push(ABPOA_HIP_CDEL, 1, id, j - 1);
                }
            }
            if (!hit && (cur_op & OP_F)) {
                for (int x = 1; x <= (GAP == 2 ? 2 : 1) && !hit; ++x) {
                    const int bit = x == 1 ? OP_F1 : OP_F2;
                    const int ex = x == 1 ? (int)e1 : (int)e2, oex = x == 1 ? (int)oe1 : (int)oe2;
                    if (!(cur_op & bit)) continue;
                    const int Fij = x == 1 ? F1ij : F2ij;
                    if (!(cur_op & OP_M) || Hij == Fij) {
                        if (st_jm1) {
                            if (Hijm1 - oex == Fij) { cur_op = OP_M | OP_E; hit = 1; }
                            else if ((x == 1 ? F1ijm1 : F2ijm1) - ex == Fij) { cur_op = bit; hit = 1; }
                        }
                    }
                }
                if (hit) { push(ABPOA_HIP_CINS, 1, id, j - 1); --j; ++n_aln; }
            }
            if (!hit && (cur_op & OP_M) && indel_first == 1) do_match(1);
            if (!hit && status == 0) status = ABPOA_HIP_EBACKTRACK;
            if (k_sel >= 0) {                                                    // move to the chosen predecessor: its record comes along
                i = __builtin_amdgcn_readlane(er.x, k_sel);
                cr = make_int4(__builtin_amdgcn_readlane(er.y, k_sel), __builtin_amdgcn_readlane(er.z, k_sel), __builtin_amdgcn_readlane(er2.x, k_sel), __builtin_amdgcn_readlane(er2.y, k_sel));
                cr_row = i;
            }
        }
        while (CW > 0 && i > 0 && j > 0 && status == 0 && true) {
            if (i > bt_hi || i < bt_lo) { load_window_cols(i, j); cr_row = -1; }
            if (cr_row != i) { cr = uniform4(B.rinfo[i - bt_lo]); cr2 = __builtin_amdgcn_readfirstlane(B.rinfo2[i - bt_lo]); cr_row = i; }
            // ---- match run, as in the whole-row loop above; here a row's staged cells are the column slice cr2 = first column | count << 16
            if ((cur_op & OP_M) && indel_first == 0 && q_in_lds && cap_safe) {
                int mi_ = i, mj = j, pi_ = i, nm_v = 0, w_lo = 0, w_hi = 0; int4 mc = cr; int mc2 = cr2;
                const int nc0 = n_cigar;
                int slots = n_cigar == 0 ? 64 : ((64 - (n_cigar & 63)) & 63);           // words that still fit before the VGPR pair has to be written out
                for (;;) {
                    const int si_ = mj - (mc2 & 0xffff), np_ = (mc.z >> 16) & 0xff;
                    if ((unsigned)si_ >= ((unsigned)mc2 >> 16)) break;                   // the cell is outside the staged slice of its row: the full step re-centres the window
                    int fl_v = (int)bt[mc.y + si_ * CW + PL_FLAG];
                    int qc_v = (int)s_query[mj - 1];
                    const int e_idx = (mc.z & 0xffff) + (lane < np_ ? lane : 0);        // (n_pred 255 = row not eligible: the reads stay inside the LDS image, the result is not used)
                    int4 er = B.edge[e_idx & (BTP - 1)]; int4 er2 = B.edge2[e_idx & (BTP - 1)];
                    const int w_lo_n = sgpr(((mj - 1) << 4) | ABPOA_HIP_CMATCH), w_hi_n = sgpr(mc.w << 2), w_idx_n = sgpr(n_cigar & 63), bs_n = sgpr((int)((unsigned)mc.z >> 24));      // (as in the whole-row loop)
                    asm volatile("" :: "s"(w_lo_n), "s"(w_hi_n), "s"(w_idx_n), "s"(bs_n));
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("" : "+v"(fl_v), "+v"(qc_v), "+v"(er.x), "+v"(er.y), "+v"(er.z), "+v"(er2.x), "+v"(er2.y), "+v"(er2.z));      // every load issued before the one wait
                    const int fl = __builtin_amdgcn_readfirstlane(fl_v);
                    const int ks = (fl - 1) & 63, ery = __builtin_amdgcn_readlane(er.y, ks);
                    // flag known and in range, row eligible, column j-1 inside that predecessor's band (empty when it is not staged)
                    if (!((unsigned)(fl - 1) < (unsigned)np_ && np_ != 255 && (unsigned)(mj - 1 - (ery & 0xffff)) < ((unsigned)ery >> 16))) break;
                    if (slots == 0) { flush_cigar(n_cigar - 64, 64); slots = 64; }
                    --slots;
                    w_lo = w_lo_n; w_hi = w_hi_n;                                                                 // node id << 34 | query index << 4 | op
                    asm volatile("s_mov_b32 m0, %4\n\ts_nop 3\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0" : "+v"(cgw_lo), "+v"(cgw_hi) : "s"(w_lo), "s"(w_hi), "s"(w_idx_n) : "m0");
                    ++n_cigar; nm_v += (qc_v == bs_n) ? 1 : 0;
                    pi_ = mi_; --mj;
                    mi_ = __builtin_amdgcn_readlane(er.x, ks);
                    mc = make_int4(ery, __builtin_amdgcn_readlane(er.z, ks), __builtin_amdgcn_readlane(er2.x, ks), __builtin_amdgcn_readlane(er2.y, ks)); mc2 = __builtin_amdgcn_readlane(er2.z, ks);
                    if (imin(mi_, mj) <= 0) break;
                }
                const int moved = n_cigar - nc0;
                if (moved) {
                    start_i = pi_; start_j = mj + 1; bt_steps += moved; bt_flag_steps += moved; n_aln += moved; n_match += __builtin_amdgcn_readfirstlane(nm_v);
                    last_word = ((uint64_t)(unsigned)w_hi << 32) | (uint64_t)(unsigned)w_lo; cur_op = OP_ALL;
                    i = mi_; j = mj; cr = mc; cr2 = mc2; cr_row = i;
                    if (i <= 0 || j <= 0) continue;
                }
            }
            const int pbi = cr.x & 0xffff, Wi = (int)((unsigned)cr.x >> 16), offi = cr.y;
            const int eb = cr.z & 0xffff, np = (cr.z >> 16) & 0xff, bs_ = (int)((unsigned)cr.z >> 24), id = cr.w;
            const int sli = cr2 & 0xffff, nsi = (int)((unsigned)cr2 >> 16);
            if (np > 64 || eb + np > BTP) break;
            // round trip 1: predecessor edge records, own cells, query code
            const int4 er = B.edge[eb + (lane < np ? lane : 0)]; const int4 er2 = B.edge2[eb + (lane < np ? lane : 0)];
            const int xi = j - pbi, si = j - sli;
            const bool st_jm1 = xi - 1 >= 0 && xi - 1 < Wi;                      // stored(gi, j-1)
            bool need = !win_narrow && ((unsigned)si >= (unsigned)nsi || (st_jm1 && si - 1 < 0));     // a cell of the own row outside the staged slice
            const T *ri = bt + offi + ((unsigned)si < (unsigned)nsi ? si : 0) * CW;
            const int Hij = __builtin_amdgcn_readfirstlane((int)ri[0]), E1ij = __builtin_amdgcn_readfirstlane((int)ri[PL_E1]), E2ij = GAP == 2 ? __builtin_amdgcn_readfirstlane((int)ri[PL_E2]) : 0, F1ij = __builtin_amdgcn_readfirstlane((int)ri[PL_F1]), F2ij = GAP == 2 ? __builtin_amdgcn_readfirstlane((int)ri[PL_F2]) : 0;      // (every lane reads the same cell: keep the walk's state in SGPRs)
            const T *rim1 = bt + offi + ((st_jm1 && si - 1 >= 0) ? si - 1 : 0) * CW;
            const int Hijm1 = __builtin_amdgcn_readfirstlane((int)rim1[0]), F1ijm1 = __builtin_amdgcn_readfirstlane((int)rim1[PL_F1]), F2ijm1 = GAP == 2 ? __builtin_amdgcn_readfirstlane((int)rim1[PL_F2]) : 0;
            const int qc = __builtin_amdgcn_readfirstlane(qcode(j - 1));
            const bool act = lane < np;
            // round trip 2: the predecessors' cells (lane k = predecessor k) and the substitution score
            const int pbk = er.y & 0xffff, Wk = (int)((unsigned)er.y >> 16), xk = j - pbk;
            const int slk = er2.z & 0xffff, nsk = (int)((unsigned)er2.z >> 16), sk = j - slk;
            const bool in_j = act && (unsigned)xk < (unsigned)Wk, in_jm1 = act && (unsigned)(xk - 1) < (unsigned)Wk;
            const bool stg_j = (unsigned)sk < (unsigned)nsk, stg_jm1 = (unsigned)(sk - 1) < (unsigned)nsk;
            need = need || (win_narrow ? __any(act && er.w == 0) : __any(act && (er.w == 0 || (in_j && !stg_j) || (in_jm1 && !stg_jm1))));
            if (need) {                                                          // something this step reads is not staged: re-centre the window on (i, j) once
                if (win_i == i && win_j == j) break;
                load_window_cols(i, j); cr_row = -1; continue;
            }
            const T *rk = bt + er.z + (in_j ? sk : 0) * CW, *rkm1 = bt + er.z + (in_jm1 ? sk - 1 : 0) * CW;
            const int Hk_j = (int)rk[0], E1k_j = (int)rk[PL_E1], E2k_j = GAP == 2 ? (int)rk[PL_E2] : 0, Hk_jm1 = (int)rkm1[0];
            const int sc_ = s_mat[m * bs_ + qc];
            start_i = i; start_j = j; ++bt_steps;
            const unsigned long long mA = __ballot(in_jm1 && Hk_jm1 + sc_ == Hij);
            int hit = 0, k_sel = -1;
            auto do_match = [&](int set_indel) __attribute__((always_inline)) {
                if (!mA) return;
                k_sel = __builtin_ctzll(mA);
                cur_op = OP_ALL; hit = 1;
                push(ABPOA_HIP_CMATCH, 1, id, j - 1);
                --j; ++n_aln; n_match += (bs_ == qc);
                if (set_indel) indel_first = 0;
            };
            if ((cur_op & OP_M) && indel_first == 0) do_match(0);
            if (!hit && (cur_op & OP_E)) {
                const bool viaM = cur_op & OP_M;
                unsigned long long m1 = 0, m2 = 0;
                if (cur_op & OP_E1) m1 = __ballot(in_j && (viaM ? Hij == E1k_j : E1ij == E1k_j - (int)e1));
                if (GAP == 2 && (cur_op & OP_E2)) m2 = __ballot(in_j && (viaM ? Hij == E2k_j : E2ij == E2k_j - (int)e2));
                if (m1 | m2) {                                                   // first predecessor in list order, E1 before E2 for the same one
                    const int k1 = m1 ? __builtin_ctzll(m1) : 64, k2 = m2 ? __builtin_ctzll(m2) : 64;
                    const bool use1 = k1 <= k2; k_sel = use1 ? k1 : k2;
                    const unsigned long long mD = __ballot(in_j && (use1 ? Hk_j - (int)oe1 == E1k_j : Hk_j - (int)oe2 == E2k_j));
                    cur_op = ((mD >> k_sel) & 1) ? (OP_M | OP_F) : (use1 ? OP_E1 : OP_E2);
                    hit = 1; push(ABPOA_HIP_CDEL, 1, id, j - 1);
                }
            }
            if (!hit && (cur_op & OP_F)) {
                for (int x = 1; x <= (GAP == 2 ? 2 : 1) && !hit; ++x) {
                    const int bit = x == 1 ? OP_F1 : OP_F2;
                    const int ex = x == 1 ? (int)e1 : (int)e2, oex = x == 1 ? (int)oe1 : (int)oe2;
                    if (!(cur_op & bit)) continue;
                    const int Fij = x == 1 ? F1ij : F2ij;
                    if (!(cur_op & OP_M) || Hij == Fij) {
                        if (st_jm1) {
                            if (Hijm1 - oex == Fij) { cur_op = OP_M | OP_E; hit = 1; }
                            else if ((x == 1 ? F1ijm1 : F2ijm1) - ex == Fij) { cur_op = bit; hit = 1; }
                        }
                    }
                }
                if (hit) { push(ABPOA_HIP_CINS, 1, id, j - 1); --j; ++n_aln; }
            }
            if (!hit && (cur_op & OP_M) && indel_first == 1) do_match(1);
            if (!hit && status == 0) status = ABPOA_HIP_EBACKTRACK;
            if (k_sel >= 0) {                                                    // move to the chosen predecessor: its record comes along
                i = __builtin_amdgcn_readlane(er.x, k_sel);
                cr = make_int4(__builtin_amdgcn_readlane(er.y, k_sel), __builtin_amdgcn_readlane(er.z, k_sel), __builtin_amdgcn_readlane(er2.x, k_sel), __builtin_amdgcn_readlane(er2.y, k_sel));
                cr2 = __builtin_amdgcn_readlane(er2.z, k_sel); cr_row = i;
            }
        }
        int slow_budget = CW > 0 ? 1 : INT_MAX;
        while (i > 0 && j > 0 && status == 0 && slow_budget-- > 0) {
            ++bt_slow_steps;
            if (CW == 0 && ((i < bt_lo + bt_margin && bt_lo > 0) || i > bt_hi || i < bt_lo)) load_window(i);
            const Geo gi = geo_of(i);
            const int Hij = cell(gi, 0, j);
            if (local && Hij == 0) break;
            start_i = i; start_j = j; ++bt_steps;
            int ps, np, id, bs_;
            { const int t = gi.in_tile ? i - bt_lo : 0; ps = B.poff[t]; np = B.poff[t + 1] - ps; id = B.nid[t]; bs_ = B.base[t]; }
            if (!gi.in_tile) { ps = gld_i32(pred_off + i); np = gld_i32(pred_off + i + 1) - ps; id = gld_i32(row_node_id + i); bs_ = gld_u8(row_base + i); }
            auto pred_bt = [&](int idx) __attribute__((always_inline)) { const int t = idx - bt_pbase; const bool ok = gi.in_tile && t >= 0 && t < BTP; int v = B.pred[ok ? t : 0]; if (!ok) v = gld_i32(pred_row + idx); return v; };
            const int qc = qcode(j - 1);
            const int s = s_mat[m * bs_ + qc];
            const int is_match = bs_ == qc;
            int hit = 0;
            auto try_match = [&](int set_indel) __attribute__((always_inline)) {
                for (int k = 0; k < np; ++k) {
                    const int pr = pred_bt(ps + k);
                    const Geo gp = geo_of(pr);
                    if (!in_range(gp, pr, j - 1)) continue;
                    if (cell(gp, 0, j - 1) + s == Hij) {
                        cur_op = OP_ALL; hit = 1;
                        push(ABPOA_HIP_CMATCH, 1, id, j - 1);
                        i = pr; --j; ++n_aln; n_match += is_match;
                        if (set_indel) indel_first = 0;
                        break;
                    }
                }
            };
            if (GAP == 0) {
                if (indel_first == 0) try_match(0);
                if (!hit) {
                    for (int k = 0; k < np; ++k) {
                        const int pr = pred_bt(ps + k);
                        const Geo gp = geo_of(pr);
                        if (!in_range(gp, pr, j)) continue;
                        if (cell(gp, 0, j) - (int)e1 == Hij) { push(ABPOA_HIP_CDEL, 1, id, j - 1); i = pr; hit = 1; break; }
                    }
                }
                if (!hit && stored(gi, j - 1) && cell(gi, 0, j - 1) - (int)e1 == Hij) { push(ABPOA_HIP_CINS, 1, id, j - 1); --j; hit = 1; ++n_aln; }
                if (!hit && indel_first == 1) try_match(1);
            } else {
                if ((cur_op & OP_M) && indel_first == 0) try_match(0);
                if (!hit && (cur_op & OP_E)) {
                    for (int k = 0; k < np && !hit; ++k) {
                        const int pr = pred_bt(ps + k);
                        const Geo gp = geo_of(pr);
                        if (!in_range(gp, pr, j)) continue;
                        for (int x = 1; x <= (GAP == 2 ? 2 : 1); ++x) {
                            const int bit = x == 1 ? OP_E1 : OP_E2, pl = x == 1 ? PL_E1 : PL_E2;
                            const int ex = x == 1 ? (int)e1 : (int)e2, oex = x == 1 ? (int)oe1 : (int)oe2;
                            if (!(cur_op & bit)) continue;
                            const int preE = cell(gp, pl, j);
                            const bool ok = (cur_op & OP_M) ? (Hij == preE) : (cell(gi, pl, j) == preE - ex);
                            if (ok) {
                                cur_op = (cell(gp, 0, j) - oex == preE) ? (OP_M | OP_F) : bit;
                                hit = 1; push(ABPOA_HIP_CDEL, 1, id, j - 1); i = pr; break;
                            }
                        }
                    }
                }
                if (!hit && (cur_op & OP_F)) {
                    for (int x = 1; x <= (GAP == 2 ? 2 : 1) && !hit; ++x) {
                        const int bit = x == 1 ? OP_F1 : OP_F2, pl = x == 1 ? PL_F1 : PL_F2;
                        const int ex = x == 1 ? (int)e1 : (int)e2, oex = x == 1 ? (int)oe1 : (int)oe2;
                        if (!(cur_op & bit)) continue;
                        const int Fij = cell(gi, pl, j);
                        if (!(cur_op & OP_M) || Hij == Fij) {
                            if (stored(gi, j - 1)) {
                                if (cell(gi, 0, j - 1) - oex == Fij) { cur_op = OP_M | OP_E; hit = 1; }
                                else if (cell(gi, pl, j - 1) - ex == Fij) { cur_op = bit; hit = 1; }
                            }
                        }
                    }
                    if (hit) { push(ABPOA_HIP_CINS, 1, id, j - 1); --j; ++n_aln; }
                }
                if (!hit && (cur_op & OP_M) && indel_first == 1) try_match(1);
            }
            if (!hit && status == 0) status = ABPOA_HIP_EBACKTRACK;
        }
        if (CW > 0) uniformize();
        } while (CW > 0 && i > 0 && j > 0 && status == 0);
        bt_win_ticks = win_ticks; bt_n_windows = n_windows; bt_wa = win_a; bt_wb = (long long)__builtin_amdgcn_s_memtime() - t_walk0;
        if (status == 0) {
            if (j > 0) push(ABPOA_HIP_CINS, j, -1, j - 1);
            if (n_cigar > 0) { const int base_ = ((n_cigar - 1) >> 6) << 6; flush_cigar(base_, n_cigar - base_); }
            __syncthreads();
            if (!b.rev_cigar) for (int k = lane; k < n_cigar >> 1; k += 64) { uint64_t t = cg[k]; cg[k] = cg[n_cigar - 1 - k]; cg[n_cigar - 1 - k] = t; }
            node_e = row_node_id[best_i]; query_e = best_j - 1;
            node_s = row_node_id[start_i]; query_s = start_j - 1;
        }
    }
    if (lane == 0) {
        AlnOut o; for (int i_ = 0; i_ < 6; ++i_) o.seg[i_] = 0;
        o.status = status; o.best_score = best_score; o.best_row = best_i; o.best_col = best_j;
        o.node_s = node_s; o.node_e = node_e; o.query_s = query_s; o.query_e = query_e;
        o.n_aln_bases = n_aln; o.n_matched_bases = n_match; o.n_cigar = n_cigar; o.pad = CW;      // pad = arena cell stride (0: plane-major)
        o.n_cells = n_cells; o.cells_used = cursor;
        for (int i_ = 0; i_ < 6; ++i_) o.seg[i_] = seg[i_];
        o.seg[5] = bt_win_ticks; o.seg[4] = bt_n_windows * 1000; o.seg[3] = bt_slow_steps * 1000; o.seg[0] = bt_wa; o.seg[1] = bt_wb; o.seg[2] = bt_flag_steps * 1000;      // backtrack: ticks spent staging arena windows, number of windows
        o.clk_dp = clk1 - clk0; o.clk_bt = (long long)__builtin_amdgcn_s_memtime() - clk1; o.n_rows_done = rows_done; o.n_bt_steps = bt_steps;
        *out_rec = o;
    }
}

// GAP: 0 linear, 1 affine, 2 convex (reference gap_mode)
template <typename T, int GAP>
__device__ __forceinline__ void align_one(const DevBatch &b, const AlnDesc &d, AlnOut *out_rec) {
    constexpr int PN = Width<T>::PN, NV = 64 / PN;
    constexpr int P = GAP == 0 ? 1 : (GAP == 1 ? 3 : 5);
    constexpr int NPR = GAP == 0 ? 1 : (GAP == 1 ? 2 : 3);        // planes kept in the LDS score ring: H, E1, E2
    constexpr int PL_E1 = 1, PL_E2 = 2, PL_F1 = GAP == 1 ? 2 : 3, PL_F2 = 4;
    const int lane = threadIdx.x & 63, l = lane % PN, vvl = lane / PN;
    const int gn = d.n_rows, qlen = d.qlen, m = b.m, w = d.w;
    const bool local = b.align_mode == ABPOA_HIP_LOCAL_MODE, extend = b.align_mode == ABPOA_HIP_EXTEND_MODE;
    const bool banded = b.wb >= 0;
    const T inf = (T)d.inf_min;
    const T e1 = (T)b.e1, o1 = (T)b.o1, oe1 = (T)(b.o1 + b.e1), e2 = (T)b.e2, o2 = (T)b.o2, oe2 = (T)(b.o2 + b.e2);
    const int dp_sn = (qlen + PN) / PN;
    // fast F path (fast_f_chain): per-lane constants and the no-wrap threshold
    const int idist = inj_dist<PN>(l);
    const int cl1 = (int)oe1 + l * (int)e1, cl2 = (int)oe2 + l * (int)e2;
    const int inj1 = idist >= 0 ? (int)inf - idist * (int)e1 : INT_MIN, inj2 = idist >= 0 ? (int)inf - idist * (int)e2 : INT_MIN;
    const long long lo_ll = (long long)(sizeof(T) == 2 ? INT16_MIN : INT32_MIN) + imax((int)oe1, (int)oe2) + (long long)PN * imax((int)e1, (int)e2);
    const int fast_lo = (int)lo_ll;

    GLOBAL_AS const uint8_t *g_query = vgpr_ptr(b.query + d.query_off);
    GLOBAL_AS const uint8_t *row_base = vgpr_ptr(b.row_base + d.row0);
    GLOBAL_AS const int32_t *row_node_id = vgpr_ptr(b.row_node_id + d.row0);
    GLOBAL_AS const int32_t *row_remain = vgpr_ptr(b.row_remain + d.row0);
    GLOBAL_AS const uint8_t *row_active = vgpr_ptr(b.row_active + d.row0);
    GLOBAL_AS const int32_t *pred_off = vgpr_ptr(b.pred_off + d.poff0), *pred_row = vgpr_ptr(b.pred_row + d.pred0);
    GLOBAL_AS const int32_t *out_off = vgpr_ptr(b.out_off + d.poff0), *out_row = vgpr_ptr(b.out_row + d.out0);
    GLOBAL_AS int32_t *g_left = vgpr_ptr(b.left + d.row0), *g_right = vgpr_ptr(b.right + d.row0);
    GLOBAL_AS int32_t *g_bsn = vgpr_ptr(b.dp_beg_sn + d.row0), *g_esn = vgpr_ptr(b.dp_end_sn + d.row0), *row_max_i = vgpr_ptr(b.row_max_i + d.row0);
    GLOBAL_AS int64_t *g_coff = vgpr_ptr(b.row_cell_off + d.row0);
    T *planes = (T *)(b.planes + d.plane_off);

    // ---- LDS carve-up (engine.h LdsPlan)
    uint8_t *s_query = lds_raw + b.lds.q_off;
    int32_t *s_mat = (int32_t *)(lds_raw + b.lds.mat_off);
    DpLds &S = *(DpLds *)(lds_raw + b.lds.phase_off);
    T *s_ring = (T *)(lds_raw + b.lds.phase_off + b.lds.ring_off);
    const int ring_rows = b.lds.ring_rows, ring_cols = b.lds.ring_cols;
    const bool q_in_lds = qlen <= b.lds.q_cap;

    { GLOBAL_AS const int32_t *g_mat = vgpr_ptr(b.mat); for (int i = lane; i < m * m; i += 64) s_mat[i] = g_mat[i]; }
    if (q_in_lds) for (int i = lane; i < qlen; i += 64) s_query[i] = g_query[i];

    // dp_end as the reference stores it: vector-rounded when banded and for row 0, qlen otherwise
    auto dp_end_of = [&](int row, int end_sn_row) __attribute__((always_inline)) { return (banded || row == 0) ? (end_sn_row + 1) * PN - 1 : qlen; };

    // literal (wrap-exact) F recurrence for the vectors [nfast_, ...) of chunk c: reference :859-875 / :978-997
    auto slow_f_tail = [&](int c, int beg_sn_, int end_sn_, int max_pre_, int nfast_, T hs, T &F1, T &F2, T &first, T &first2) __attribute__((always_inline)) {
#pragma unroll
        for (int vv = 0; vv < NV; ++vv) {
            const int vg = beg_sn_ + c * NV + vv;
            if (vv >= nfast_ && vg <= end_sn_) {
                int set_num = PN;
                if (!local && vg > max_pre_) set_num = (vg == max_pre_ + 1) ? 2 : 1;
                T prev = (T)row_shr<1>((int)first, (int)hs);
                if (PN == 8) prev = (l == 0) ? first : prev;
                T f = wsub<T>(prev, oe1);                        // reference :870 / :990
                f = set_f<T>(f, l, set_num, e1, inf);
                const T hlast = (T)__builtin_amdgcn_readlane((int)hs, vv * PN + PN - 1);
                first = tmax<T>(hlast, wadd<T>((T)__builtin_amdgcn_readlane((int)f, vv * PN + PN - 1), o1));  // :874 / :996
                if (vvl == vv) F1 = f;
                if (GAP == 2) {
                    T prev2 = (T)row_shr<1>((int)first2, (int)hs);
                    if (PN == 8) prev2 = (l == 0) ? first2 : prev2;
                    T g = wsub<T>(prev2, oe2);                   // reference :991
                    g = set_f<T>(g, l, set_num, e2, inf);
                    first2 = tmax<T>(hlast, wadd<T>((T)__builtin_amdgcn_readlane((int)g, vv * PN + PN - 1), o2));  // :997
                    if (vvl == vv) F2 = g;
                }
            }
        }
    };

    long long cursor = 0;          // next free arena cell
    long long n_cells = 0;
    int status = 0;
    int rows_done = 0, bt_steps = 0;
    int best_score = d.inf_min, best_i = 0, best_j = 0, best_row_zd = 0;
    int last_done = 0;                                        // last row the loop reached (z-drop may stop early)
    long long clk0 = 0, clk1 = 0;
#ifdef ABPOA_HIP_PROFILE
    long long seg_keep[6] = {0, 0, 0, 0, 0, 0};
#endif
    {
    // ------------------------------------------------------------------ row 0, reference :553-662
    int end_sn0 = 0;
    {
        int dp_end0;
        if (banded) {
            int r = row_remain[0] - row_remain[gn - 1] - 1;
            dp_end0 = imin(qlen, imax(0, qlen - r) + w);          // max_pos_right[begin] == 0
        } else dp_end0 = qlen;
        end_sn0 = dp_end0 / PN;
        const int W0 = (end_sn0 + 1) * PN;
        if ((long long)W0 * P > d.plane_cap) { status = ABPOA_HIP_STATUS_OVERFLOW; }
        else {
            const bool ring0 = W0 <= ring_cols;
            if (lane == 0) { g_bsn[0] = 0; g_esn[0] = end_sn0; g_coff[0] = 0; S.b_rec[0] = make_int4(0, end_sn0, 0, ring0 ? 0 : -1); }
            for (int i = lane; i < W0; i += 64) {
                T h, x1 = inf, x2 = inf, f1 = inf, f2 = inf;
                if (local) { h = 0; x1 = 0; x2 = 0; f1 = 0; f2 = 0; }
                else if (GAP == 0) h = (T)(-(int)e1 * i);
                else if (GAP == 1) {
                    T g = (T)(-(int)o1 - (int)e1 * i);
                    h = i == 0 ? (T)0 : g; x1 = i == 0 ? (T)(-(int)oe1) : inf; f1 = i == 0 ? inf : g;
                } else {
                    T g1 = (T)(-(int)o1 - (int)e1 * i), g2 = (T)(-(int)o2 - (int)e2 * i);
                    h = i == 0 ? (T)0 : tmax<T>(g1, g2);
                    x1 = i == 0 ? (T)(-(int)oe1) : inf; x2 = i == 0 ? (T)(-(int)oe2) : inf;
                    f1 = i == 0 ? inf : g1; f2 = i == 0 ? inf : g2;
                }
                planes[i] = h;
                if (GAP != 0) { planes[(long long)PL_E1 * W0 + i] = x1; planes[(long long)PL_F1 * W0 + i] = f1; }
                if (GAP == 2) { planes[(long long)PL_E2 * W0 + i] = x2; planes[(long long)PL_F2 * W0 + i] = f2; }
                if (ring0) {
                    s_ring[i] = h;
                    if (GAP != 0) s_ring[ring_cols + i] = x1;
                    if (GAP == 2) s_ring[2 * ring_cols + i] = x2;
                }
            }
            cursor = (long long)W0 * P;
        }
    }
    // ---- max_pos_left/right look-ahead window: LDS holds rows [lr_blk, lr_blk + RL)
    int lr_blk = 0;
    if (banded && status == 0) {
        if (b.fresh_band) {       // reference abpoa_topological_sort resets them before every alignment (abpoa_graph.c:303-308)
            for (int i = lane; i < gn; i += 64) { g_left[i] = gn; g_right[i] = 0; }
            for (int i = lane; i < RL; i += 64) S.l_lr[i] = make_int2(gn, 0);
        } else
            for (int i = lane; i < RL; i += 64) { const int r = i; if (r < gn) S.l_lr[i] = make_int2(g_left[r], g_right[r]); }
        __syncthreads();
        if (lane == 0) S.l_lr[0] = make_int2(0, 0);                            // reference :556
        for (int t = out_off[0] + lane; t < out_off[1]; t += 64) {            // reference :557-561
            const int o = out_row[t];
            if (o >= 0 && row_active[o]) {
                if (o < RL) S.l_lr[o] = make_int2(1, 1); else { g_left[o] = 1; g_right[o] = 1; }
            }
        }
    }
    __syncthreads();

#ifdef ABPOA_HIP_PROFILE
    long long seg[6] = {0, 0, 0, 0, 0, 0}, seg_last = 0;
#define STAMP(I) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); long long t_ = (long long)__builtin_amdgcn_s_memtime(); seg[I] += t_ - seg_last; seg_last = t_; }
#else
#define STAMP(I)
#endif
    clk0 = (long long)__builtin_amdgcn_s_memtime();
#ifdef ABPOA_HIP_PROFILE
    seg_last = clk0;
#endif
    const int remain_end = (banded || b.zdrop > 0) ? row_remain[gn - 1] : 0;
    const bool need_max = local || extend || banded;
    int tile_beg = 0, tile_end = 0, pbase = 0, obase = 0;     // static-metadata tile covers rows [tile_beg, tile_end)
    int last_row = 0;                                         // last row whose left/right entry was consumed

    // ------------------------------------------------------------------ rows 1 .. gn-2, reference :1105
    // next-tile prefetch registers (static graph metadata of rows [nt_t0, nt_t0 + TS))
    int4 nt_rec0 = make_int4(0, 0, 0, 0), nt_rec1 = make_int4(0, 0, 0, 0); int nt_pred[TP / 64], nt_out[TP / 64];
    int nt_t0 = 1, nt_pb0 = 0, nt_ob0 = 0; bool far_seen = false;
    // per-lane copy of the CURRENT tile's metadata (lane i <-> row tile_beg + i): the row loop fetches a field with one
    // v_readlane instead of an LDS round trip.  tv_meta = base | active<<8 | fast<<9 | np<<16 | n_out<<24
    int tv_meta = 0, tv_rterm = 0, tv_ps = 0, tv_os = 0, tv_pid[4] = {0, 0, 0, 0}, tv_o[2] = {-1, -1};
    if (gn > 2) {
        nt_pb0 = gld_i32(pred_off + 1); nt_ob0 = gld_i32(out_off + 1);
        const int tend = imin(nt_t0 + TS, gn);
        const int rr = imin(nt_t0 + lane, gn), rc = imin(rr, gn - 1), r2 = imin(nt_t0 + TS, gn);
        nt_rec0.x = pred_off[rr]; nt_rec0.y = out_off[rr];
        const int rem_ = (banded || b.zdrop > 0) ? row_remain[rc] : 0; const int ba_ = (int)row_base[rc] | ((int)row_active[rc] << 8);
        nt_rec0.z = rr < tend ? rem_ : 0; nt_rec0.w = rr < tend ? ba_ : 0;
        nt_rec1.x = pred_off[r2]; nt_rec1.y = out_off[r2];
#pragma unroll
        for (int j = 0; j < TP / 64; ++j) { nt_pred[j] = pred_row[nt_pb0 + j * 64 + lane]; nt_out[j] = out_row[nt_ob0 + j * 64 + lane]; }
    }
    int qc_beg_sn = -1; int qc_cache[2] = {0, 0};            // query codes of this lane's columns for chunks 0/1 of band start qc_beg_sn
    for (int row = 1; row < gn - 1 && status == 0; ++row) {
        if (row >= tile_end) {                                // ---- switch to the next 64-row metadata tile (prefetched in registers)
            // band geometry of the rows of the finished tile goes to HBM in one coalesced burst (backtrack + trace read it)
            if (tile_end > tile_beg && tile_beg + lane < tile_end) {
                const int r = tile_beg + lane; const int4 br = S.b_rec[r % RB];
                g_bsn[r] = br.x; g_esn[r] = br.y; g_coff[r] = (long long)(uint32_t)br.z * PN;
            }
            S.t_rec[lane] = nt_rec0; if (lane == 0) S.t_rec[TS] = nt_rec1;
#pragma unroll
            for (int j = 0; j < TP / 64; ++j) { S.t_pred[j * 64 + lane] = nt_pred[j]; S.t_out[j * 64 + lane] = nt_out[j]; }
            tile_beg = nt_t0; tile_end = imin(nt_t0 + TS, gn); pbase = nt_pb0; obase = nt_ob0;
            {
                const int my_ps = nt_rec0.x, my_os = nt_rec0.y;
                const int4 nx = S.t_rec[lane + 1];
                const int np_ = nx.x - my_ps, on_ = nx.y - my_os;
                bool ok = np_ >= 1 && np_ <= 4 && on_ >= 0 && on_ <= 2;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int idx = my_ps - pbase + imin(kk, imax(np_ - 1, 0));
                    ok = ok && idx >= 0 && idx < TP; tv_pid[kk] = S.t_pred[(idx >= 0 && idx < TP) ? idx : 0];
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int idx = my_os - obase + kk; const bool has = kk < on_;
                    ok = ok && (!has || (idx >= 0 && idx < TP)); tv_o[kk] = has ? S.t_out[(idx >= 0 && idx < TP) ? idx : 0] : -1;
                }
                tv_meta = (nt_rec0.w & 0x1ff) | (ok ? (1 << 9) : 0) | (imin(imax(np_, 0), 255) << 16) | (imin(imax(on_, 0), 255) << 24);
                tv_rterm = qlen - (nt_rec0.z - remain_end - 1); tv_ps = my_ps; tv_os = my_os;
                // fast-row record (static eligibility: active, 1-2 predecessors within score-ring distance, 1-2 successors)
                const int myrow = tile_beg + lane;
                const int d0 = myrow - tv_pid[0], d1 = myrow - tv_pid[np_ >= 2 ? 1 : 0];
                const bool fok = ok && ((nt_rec0.w >> 8) & 1) && np_ <= 2 && on_ >= 1 && d0 >= 1 && d1 >= 1 && d0 < ring_rows && d1 < ring_rows && d0 < 256 && d1 < 256;
                S.t_fast[lane] = make_int4((fok ? (int)0x80000000u : 0) | ((nt_rec0.w & 0xff) << 16) | ((d1 & 0xff) << 8) | (d0 & 0xff), tv_rterm, tv_o[0], tv_o[1]);
            }
            // issue the loads of the tile after this one right away; they complete while this tile is being processed
            nt_t0 = tile_end; nt_pb0 = __builtin_amdgcn_readfirstlane(nt_rec1.x); nt_ob0 = __builtin_amdgcn_readfirstlane(nt_rec1.y);
            if (nt_t0 < gn - 1) {
                const int tend = imin(nt_t0 + TS, gn);
                const int rr = imin(nt_t0 + lane, gn), rc = imin(rr, gn - 1), r2 = imin(nt_t0 + TS, gn);
                nt_rec0.x = pred_off[rr]; nt_rec0.y = out_off[rr];
                const int rem_ = (banded || b.zdrop > 0) ? row_remain[rc] : 0; const int ba_ = (int)row_base[rc] | ((int)row_active[rc] << 8);
                nt_rec0.z = rr < tend ? rem_ : 0; nt_rec0.w = rr < tend ? ba_ : 0;
                nt_rec1.x = pred_off[r2]; nt_rec1.y = out_off[r2]; nt_rec1.z = 0; nt_rec1.w = 0;
#pragma unroll
                for (int j = 0; j < TP / 64; ++j) { nt_pred[j] = pred_row[nt_pb0 + j * 64 + lane]; nt_out[j] = out_row[nt_ob0 + j * 64 + lane]; }
            }
        }
        if (banded && row >= lr_blk + RLH) {                  // ---- slide the left/right window by half
            // rows [lr_blk, lr_blk+RLH) are retired: write them back, then bring in rows [lr_blk+RL, lr_blk+RL+RLH)
            for (int i = lane; i < RLH; i += 64) {
                const int r = lr_blk + i;
                if (r < gn) { const int2 v2 = S.l_lr[r % RL]; g_left[r] = v2.x; g_right[r] = v2.y; }
                const int nr = lr_blk + RL + i;
                if (nr < gn) {
                    if (b.fresh_band && !far_seen) S.l_lr[nr % RL] = make_int2(gn, 0);       // untouched so far: known without a load
                    else S.l_lr[nr % RL] = make_int2(gld_i32(g_left + nr), gld_i32(g_right + nr));
                }
            }
            lr_blk += RLH;
        }
        STAMP(5)
        last_done = row;
        const int ti = row - tile_beg;
        // ====================================================================================================
        // FAST ROW (the common case of a POA graph): global + banded + affine/convex, one or two predecessors whose H/E
        // rows are still in the score ring, band <= 128 columns, successors inside the left/right window.  Straight-line:
        // one record read, two geometry reads, one batch of score reads per chunk, one reduction.  Same arithmetic as the
        // general row below (which handles everything else), so the results are identical.
        if (GAP != 0 && banded && !local && !extend && !(b.dbg & 64)) {
            const int4 fr = uniform4(S.t_fast[ti]);
            if (fr.x < 0) {
                const int2 lr = uniform2(S.l_lr[row % RL]);
                const int fp0 = row - (fr.x & 0xff), fp1 = row - ((fr.x >> 8) & 0xff), fbase = (fr.x >> 16) & 0xff;
                const int4 g0 = uniform4(S.b_rec[fp0 % RB]), g1 = uniform4(S.b_rec[fp1 % RB]);
                const int so0 = fr.z, so1 = fr.w;      // successor rows (-1 = none)
                const int fbeg = imax(0, imin(lr.x, fr.y) - w), fend = imin(qlen, imax(lr.y, fr.y) + w);      // reference :711
                const int fmin_pre = imin(g0.x, g1.x), fmax_pre = imax(g0.y, g1.y);
                const int fbeg_sn = imax(fbeg / PN, fmin_pre), fend_sn = fend / PN;
                const int fWr = (fend_sn - fbeg_sn + 1) * PN;
                const bool feasible = g0.w == fp0 && g1.w == fp1 && fWr <= 128 && fWr <= ring_cols && so0 < lr_blk + RL && so1 < lr_blk + RL &&
                                      q_in_lds && cursor + (long long)fWr * P <= d.plane_cap;
                if (feasible) {
                    const long long off = cursor; cursor += (long long)fWr * P; n_cells += fWr; ++rows_done; last_row = row;
                    if (lane == 0) S.b_rec[row % RB] = make_int4(fbeg_sn, fend_sn, (int)(uint32_t)(off / PN), -1);
                    T *H = planes + off;
                    T *my_ring = s_ring + (long long)(row % ring_rows) * NPR * ring_cols;
                    const T *rp0 = s_ring + (long long)(fp0 % ring_rows) * NPR * ring_cols, *rp1 = s_ring + (long long)(fp1 % ring_rows) * NPR * ring_cols;
                    if (fbeg_sn != qc_beg_sn) {
                        qc_beg_sn = fbeg_sn;
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2) { const int cc = fbeg_sn * PN + c2 * 64 + lane; qc_cache[c2] = (cc >= 1 && cc <= qlen) ? (int)s_query[cc - 1] : -1; }
                    }
                    const int pb0 = g0.x * PN, pse0 = (g0.y + 1) * PN - 1, pb1 = g1.x * PN, pse1 = (g1.y + 1) * PN - 1;   // stored column ranges
                    const int bs0 = imax(g0.x, fbeg_sn), bs1 = imax(g1.x, fbeg_sn);
                    const int esh0 = imin(imin(g0.y + 1, fend_sn), dp_sn - 1), esh1 = imin(imin(g1.y + 1, fend_sn), dp_sn - 1);
                    const int ese0 = imin(g0.y, fend_sn), ese1 = imin(g1.y, fend_sn);
                    T first = 0, first2 = 0; int dbgv = 0;
                    int am_val = INT_MIN, am_v = 0, am_isend = 0; bool am_any = false; unsigned am_key = 0;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        if (c == 1 && fWr <= 64) break;
                        const int rel = c * 64 + lane;
                        const bool in_band = rel < fWr;
                        const int col = fbeg_sn * PN + rel, v = fbeg_sn + c * NV + vvl;
                        const int qc = c == 0 ? qc_cache[0] : qc_cache[1];
                        const int qv = s_mat[fbase * m + (qc >= 0 ? qc : 0)];
                        const int x0 = col - 1 - pb0, x1 = col - 1 - pb1;
                        const bool ok0 = x0 >= 0 && col - 1 <= pse0, ok1 = x1 >= 0 && col - 1 <= pse1;
                        const bool inH0 = in_band && v >= bs0 && v <= esh0, inH1 = in_band && v >= bs1 && v <= esh1;
                        const bool inE0 = in_band && v >= bs0 && v <= ese0, inE1 = in_band && v >= bs1 && v <= ese1;
                        const T h0 = rp0[ok0 ? x0 : 0], h1 = rp1[ok1 ? x1 : 0];
                        const T a0 = rp0[ring_cols + (inE0 ? x0 + 1 : 0)], a1 = rp1[ring_cols + (inE1 ? x1 + 1 : 0)];
                        T c0 = 0, c1 = 0;
                        if (GAP == 2) { c0 = rp0[2 * ring_cols + (inE0 ? x0 + 1 : 0)]; c1 = rp1[2 * ring_cols + (inE1 ? x1 + 1 : 0)]; }
                        const T q = (in_band && qc >= 0) ? (T)qv : (T)0;
                        T Mv = inH0 ? (ok0 ? h0 : inf) : inf;
                        Mv = inH1 ? tmax<T>(Mv, ok1 ? h1 : inf) : Mv;
                        T E1v = inE0 ? a0 : inf; E1v = inE1 ? tmax<T>(E1v, a1) : E1v;
                        T E2v = inf; if (GAP == 2) { E2v = inE0 ? c0 : inf; E2v = inE1 ? tmax<T>(E2v, c1) : E2v; }
                        const T h = wadd<T>(Mv, q);
                        T hs = h; if (GAP == 2) hs = tmax<T>(tmax<T>(h, E1v), E2v);
                        if (c == 0) { first = (T)__builtin_amdgcn_readlane((int)h, 0); first2 = first; }
                        const int nvec = imin(NV, fend_sn - (fbeg_sn + c * NV) + 1);
                        int nfast = imin(nvec, fmax_pre - (fbeg_sn + c * NV) + 1);
                        if (nfast < 0) nfast = 0;
                        if (nfast > 0 && __any(vvl < nfast && (int)h < fast_lo)) nfast = 0;
                        if (b.dbg & 4) nfast = 0;
                        T F1 = inf, F2 = inf;
                        if (nfast > 0) {
                            int fi = (int)first;
                            int dcv = 0;
                            F1 = (T)fast_f_chain<T>((int)hs, fi, nfast, l, vvl, (int)oe1, (int)e1, (int)o1, cl1, inj1, &dcv);
                            if (b.dbg & 512) F1 = (T)dcv;
                            first = (T)fi;
                            if (GAP == 2) { int fi2 = (int)first2; F2 = (T)fast_f_chain<T>((int)hs, fi2, nfast, l, vvl, (int)oe2, (int)e2, (int)o2, cl2, inj2); first2 = (T)fi2; }
                        }
                        if ((b.dbg & 256) && c == 0) dbgv = ((int)first & 0xffff) | (nfast << 16) | (nvec << 20) | ((__builtin_amdgcn_readlane((int)hs, 15) & 0xff) << 24);
                        if (nfast < nvec) slow_f_tail(c, fbeg_sn, fend_sn, fmax_pre, nfast, hs, F1, F2, first, first2);
                        T Hout, E1out, E2out = 0;
                        if (GAP == 1) {                                          // reference :876-883
                            const T tmp = tmax<T>(h, E1v);
                            Hout = tmax<T>(tmp, F1);
                            const T en = tmax<T>(wsub<T>(E1v, e1), wsub<T>(Hout, oe1));
                            E1out = (Hout == tmp) ? en : inf;
                        } else {                                                 // reference :1004-1007
                            Hout = tmax<T>(hs, tmax<T>(F1, F2));
                            E1out = tmax<T>(wsub<T>(E1v, e1), wsub<T>(Hout, oe1));
                            E2out = tmax<T>(wsub<T>(E2v, e2), wsub<T>(Hout, oe2));
                        }
                        if (in_band) {
                            H[rel] = Hout; H[PL_E1 * fWr + rel] = E1out; H[PL_F1 * fWr + rel] = F1;
                            if (GAP == 2) { H[PL_E2 * fWr + rel] = E2out; H[PL_F2 * fWr + rel] = F2; }
                            my_ring[rel] = Hout; my_ring[ring_cols + rel] = E1out;
                            if (GAP == 2) my_ring[2 * ring_cols + rel] = E2out;
                            const bool is_end = (v == fend_sn);
                            int cand = (int)Hout;
                            if (is_end && fend_sn == qlen / PN && col > qlen) cand = (int)inf;
                            if (sizeof(T) == 2) {
                                const unsigned key = ((unsigned)(cand + 32768) << 16) | ((unsigned)(PN - 1 - l) << 12) | ((unsigned)is_end << 11) | (unsigned)(2047 - v);
                                am_key = key > am_key ? key : am_key;
                            } else if (!am_any || (is_end ? cand >= am_val : cand > am_val)) { am_val = cand; am_v = v; am_isend = is_end; am_any = true; }
                        }
                    }
                    if (lane == 0) S.b_rec[row % RB].w = row;
                    int mi = -1;
                    if (sizeof(T) == 2) {
                        const unsigned kb = (b.dbg & 128) ? wave_max_u32(am_key) : wave_max_u32_b(am_key);
                        const int vmax = (int)(kb >> 16) - 32768;
                        if (vmax > d.inf_min) { mi = (2047 - (int)(kb & 0x7ff)) * PN + (PN - 1 - (int)((kb >> 12) & 0xf)); if (mi > qlen) mi = -1; }
                    } else {
                        const int vmax = wave_max_i32(am_any ? am_val : INT_MIN);
                        if (vmax > d.inf_min) {
                            unsigned key = 0;
                            if (am_any && am_val == vmax) key = ((unsigned)(PN - 1 - l) << 27) | ((unsigned)am_isend << 26) | (0x3FFFFFFu - (unsigned)am_v);
                            const unsigned kb = wave_max_u32_b(key);
                            mi = (int)(0x3FFFFFFu - (kb & 0x3FFFFFFu)) * PN + (PN - 1 - (int)(kb >> 27));
                            if (mi > qlen) mi = -1;
                        }
                    }
                    if (b.want_trace && lane == 0) row_max_i[row] = (b.dbg & 256) ? dbgv : mi;
                    {                                                            // reference :1059-1067
                        const int out_i = mi + 1;
                        int2 a2 = S.l_lr[(so0 >= 0 ? so0 : 0) % RL], b2 = S.l_lr[(so1 >= 0 ? so1 : 0) % RL];
                        a2.x = imin(a2.x, out_i); a2.y = imax(a2.y, out_i); b2.x = imin(b2.x, out_i); b2.y = imax(b2.y, out_i);
                        if (so0 >= 0 && lane == 0) S.l_lr[so0 % RL] = a2;
                        if (so1 >= 0 && lane == 0) S.l_lr[so1 % RL] = b2;
                    }
                    continue;
                }
            }
        }
        // ==================================================================================================== general row
        const int meta = __builtin_amdgcn_readlane(tv_meta, ti);
        if (!((meta >> 8) & 1)) { if (lane == 0) S.b_rec[row % RB] = make_int4(-1, -1, (int)(uint32_t)(cursor / PN), -1); continue; }
        const int2 lr = S.l_lr[row % RL];
        const int base = meta & 0xff;
        const bool fastmeta = (meta >> 9) & 1;
        const int rterm = __builtin_amdgcn_readlane(tv_rterm, ti);       // qlen - (remain[row] - remain[end] - 1), reference abpoa_align.h:34-35
        const int ps = __builtin_amdgcn_readlane(tv_ps, ti), os = __builtin_amdgcn_readlane(tv_os, ti);
        auto pred_at = [&](int idx) __attribute__((always_inline)) { const int t = idx - pbase; int v = S.t_pred[t < TP ? t : 0]; if (t >= TP) v = gld_i32(pred_row + idx); return v; };
        // band geometry of an earlier row: LDS ring for the last RB rows, HBM copy otherwise (w = -1: never in the score ring)
        auto geom4 = [&](int p) __attribute__((always_inline)) {
            int4 g4 = S.b_rec[p % RB];
            if (row - p >= RB) { g4.x = gld_i32(g_bsn + p); g4.y = gld_i32(g_esn + p); g4.z = (int)(uint32_t)(gld_i64(g_coff + p) / PN); g4.w = -1; }
            return g4;
        };
        // the first (up to) four predecessors are handled in one batch; np <= 4 covers practically every POA node
        int np, on, pid[4]; int4 pg[4];
        if (fastmeta) {
            np = (meta >> 16) & 0xff; on = (meta >> 24) & 0xff;
#pragma unroll
            for (int k = 0; k < 4; ++k) pid[k] = __builtin_amdgcn_readlane(tv_pid[k], ti);
        } else {
            np = S.t_rec[ti + 1].x - ps; on = S.t_rec[ti + 1].y - os;
#pragma unroll
            for (int k = 0; k < 4; ++k) pid[k] = pred_at(ps + imin(k, np - 1));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) pg[k] = geom4(pid[k]);
        int beg_sn, end_sn, max_pre_end_sn;
        if (!banded) { beg_sn = 0; end_sn = qlen / PN; max_pre_end_sn = end_sn; }        // reference :706-709
        else {                                                                          // reference :710-720
            last_row = row;
            int beg = imax(0, imin(lr.x, rterm) - w), end = imin(qlen, imax(lr.y, rterm) + w);
            beg_sn = beg / PN;
            int min_pre_beg_sn = imin(imin(pg[0].x, pg[1].x), imin(pg[2].x, pg[3].x));          // duplicates of the last one are harmless
            max_pre_end_sn = imax(imax(pg[0].y, pg[1].y), imax(pg[2].y, pg[3].y));
            for (int k = 4; k < np; ++k) { const int4 g4 = geom4(pred_at(ps + k)); min_pre_beg_sn = imin(min_pre_beg_sn, g4.x); max_pre_end_sn = imax(max_pre_end_sn, g4.y); }
            if (beg_sn < min_pre_beg_sn) beg_sn = min_pre_beg_sn;
            end_sn = end / PN;
        }
        const int Wr = (end_sn - beg_sn + 1) * PN;
        const long long off = cursor;
        if (off + (long long)Wr * P > d.plane_cap) { status = ABPOA_HIP_STATUS_OVERFLOW; break; }
        cursor += (long long)Wr * P;
        n_cells += Wr; ++rows_done;
        const bool to_ring = Wr <= ring_cols;
        if (lane == 0) S.b_rec[row % RB] = make_int4(beg_sn, end_sn, (int)(uint32_t)(off / PN), -1);   // score-ring tag set when the row is complete
        T *H = planes + off;
        const int my_slot = row % ring_rows;
        T *my_ring = s_ring + (long long)my_slot * NPR * ring_cols;
        const int nchunk = (b.dbg & 8) ? 0 : (Wr + 63) >> 6;
        T first = 0, first2 = 0;
        // running arg-max state of this lane (reference :1043-1057)
        int am_val = INT_MIN, am_v = 0, am_isend = 0; bool am_any = false;
        unsigned am_key = 0;          // int16: value and tie-break priority packed into one word (value<<16 | 15-lane<<12 | is_end<<11 | 2047-vector)
        // fast gather: every predecessor's H/E row is in the LDS score ring -> straight-line, batched LDS reads
        bool all_ring = np <= 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) all_ring = all_ring && (row - pid[k] < ring_rows) && (pg[k].w == pid[k]);
        if (q_in_lds && beg_sn != qc_beg_sn) {                // band start moved: refresh this lane's cached query codes
            qc_beg_sn = beg_sn;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) { const int cc = beg_sn * PN + c2 * 64 + lane; qc_cache[c2] = (cc >= 1 && cc <= qlen) ? (int)s_query[cc - 1] : -1; }
        }

        STAMP(0)
        for (int c = 0; c < nchunk; ++c) {
            const int rel = c * 64 + lane;
            const bool in_band = rel < Wr;
            const int col = beg_sn * PN + rel;
            const int v = beg_sn + c * NV + vvl;
            T Mv = inf, E1v = inf, E2v = inf;
            // query profile value, reference :504-510
            T q = 0;
            if (!(b.dbg & 32)) {
                int qc;
                if (q_in_lds) qc = c == 0 ? qc_cache[0] : c == 1 ? qc_cache[1] : ((col >= 1 && col <= qlen) ? (int)s_query[col - 1] : -1);
                else qc = (in_band && col >= 1 && col <= qlen) ? gld_u8(g_query + col - 1) : -1;
                const int qv = s_mat[base * m + (qc >= 0 ? qc : 0)];
                q = (in_band && qc >= 0) ? (T)qv : (T)0;
            }
            // ---- predecessors, reference :722-761 / :803-852 / :912-969
            if (b.dbg & 16) { Mv = (T)(col & 15); E1v = inf; }
            else if (all_ring) {
                auto gather_ring = [&](auto npc) __attribute__((always_inline)) {
                    constexpr int N = decltype(npc)::value;
                    int hraw[N], e1raw[N], e2raw[N], vraw[N]; bool inHk[N], inEk[N], srcok[N], vok[N];
#pragma unroll
                    for (int k = 0; k < N; ++k) {                      // issue every LDS read first (clamped addresses), select afterwards
                        const int pb = pg[k].x, pe = pg[k].y;
                        const int p_stored_end = (pe + 1) * PN - 1;
                        int bs, es_h, es_e;
                        if (local) { bs = 0; es_h = end_sn; es_e = end_sn; }
                        else {
                            bs = pb < beg_sn ? beg_sn : pb;
                            es_h = imin(imin((dp_end_of(pid[k], pe) + 1) / PN, end_sn), dp_sn - 1);
                            es_e = imin(pe, end_sn);
                        }
                        inHk[k] = in_band && v >= bs && v <= es_h;
                        inEk[k] = GAP != 0 && in_band && v >= bs && v <= es_e;
                        const T *rp = s_ring + (long long)(pid[k] % ring_rows) * NPR * ring_cols;
                        const int x = col - 1 - pb * PN;                // column col-1 relative to the source row's band start
                        srcok[k] = x >= 0 && col - 1 <= p_stored_end;
                        hraw[k] = (int)rp[srcok[k] ? x : 0];
                        if (GAP == 0) { vok[k] = col <= p_stored_end; vraw[k] = (int)rp[vok[k] ? x + 1 : 0]; }
                        if (GAP != 0) e1raw[k] = (int)rp[ring_cols + (inEk[k] ? x + 1 : 0)];
                        if (GAP == 2) e2raw[k] = (int)rp[2 * ring_cols + (inEk[k] ? x + 1 : 0)];
                    }
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        T hval = srcok[k] ? (T)hraw[k] : (local ? (T)0 : inf);
                        if (GAP == 0) { const T vert = vok[k] ? (T)vraw[k] : inf; hval = tmax<T>(wadd<T>(hval, q), wsub<T>(vert, e1)); }
                        if (k == 0) Mv = inHk[k] ? hval : inf; else Mv = inHk[k] ? tmax<T>(Mv, hval) : Mv;
                        if (GAP != 0) { if (k == 0) E1v = inEk[k] ? (T)e1raw[k] : inf; else E1v = inEk[k] ? tmax<T>(E1v, (T)e1raw[k]) : E1v; }
                        if (GAP == 2) { if (k == 0) E2v = inEk[k] ? (T)e2raw[k] : inf; else E2v = inEk[k] ? tmax<T>(E2v, (T)e2raw[k]) : E2v; }
                    }
                };
                if (np == 1) gather_ring(std::integral_constant<int, 1>{});
                else if (np == 2) gather_ring(std::integral_constant<int, 2>{});
                else if (np == 3) gather_ring(std::integral_constant<int, 3>{});
                else gather_ring(std::integral_constant<int, 4>{});
            } else
            for (int k = 0; k < np; ++k) {
                const int p = pred_at(ps + k);
                const int4 g4 = geom4(p);
                const int pb = g4.x, pe = g4.y; const long long poff = (long long)(uint32_t)g4.z * PN;
                const int Wp = (pe - pb + 1) * PN;
                const int p_stored_end = (pe + 1) * PN - 1;          // last stored column of the predecessor row
                const int pslot = p % ring_rows;
                const bool in_ring = (row - p < ring_rows) && g4.w == p;
                int bs, es_h, es_e;
                if (local) { bs = 0; es_h = end_sn; es_e = end_sn; }
                else {
                    bs = pb < beg_sn ? beg_sn : pb;
                    es_h = imin(imin((dp_end_of(p, pe) + 1) / PN, end_sn), dp_sn - 1);
                    es_e = imin(pe, end_sn);
                }
                const bool inH = in_band && v >= bs && v <= es_h;
                const bool inE = GAP != 0 && in_band && v >= bs && v <= es_e;
                // The source row is read either from the LDS score ring or from its HBM copy; the two paths are
                // instantiated separately so that no generic (flat) pointer is ever formed.
                auto gather = [&](auto from_lds, const T *Hp, const long long pstride) __attribute__((always_inline)) {
                    auto ld = [&](const T *p_) __attribute__((always_inline)) -> T { if constexpr (decltype(from_lds)::value) return *p_; else return (T)gld_cell((GLOBAL_AS const T *)p_); };
                    if (inH) {
                        T hval;
                        if (col == bs * PN) {
                            if (local) hval = 0;
                            else hval = (pb < beg_sn && beg_sn * PN - 1 <= p_stored_end) ? ld(Hp + beg_sn * PN - 1 - pb * PN) : inf;
                        } else hval = (col - 1 <= p_stored_end) ? ld(Hp + col - 1 - pb * PN) : inf;
                        if (GAP == 0) {
                            T vert = (col <= p_stored_end) ? ld(Hp + col - pb * PN) : inf;
                            hval = tmax<T>(wadd<T>(hval, q), wsub<T>(vert, e1));
                        }
                        Mv = (k == 0) ? hval : tmax<T>(Mv, hval);
                    }
                    if (inE) {
                        T ev = ld(Hp + (long long)PL_E1 * pstride + col - pb * PN);
                        E1v = (k == 0) ? ev : tmax<T>(E1v, ev);
                        if (GAP == 2) {
                            T ev2 = ld(Hp + (long long)PL_E2 * pstride + col - pb * PN);
                            E2v = (k == 0) ? ev2 : tmax<T>(E2v, ev2);
                        }
                    }
                };
                if (in_ring) gather(std::true_type{}, s_ring + (long long)pslot * NPR * ring_cols, (long long)ring_cols);
                else gather(std::false_type{}, planes + poff, (long long)Wp);       // HBM copy (older or over-wide row)
            }
            STAMP(1)
            // ---- in-row part
            T Hout, E1out = 0, E2out = 0, F1 = inf, F2 = inf;
            if (GAP == 0) {
                // reference :762-778
                T h = Mv;
                if (c == 0) first = (T)__builtin_amdgcn_readlane((int)h, 0);
#pragma unroll
                for (int vv = 0; vv < NV; ++vv) {
                    const int vg = beg_sn + c * NV + vv;
                    if (vg <= end_sn) {
                        int set_num = PN;
                        if (!local && vg > max_pre_end_sn) set_num = (vg == max_pre_end_sn + 1) ? 1 : 0;
                        T hv = tmax<T>(h, l == 0 ? first : inf);
                        hv = set_f<T>(hv, l, set_num, e1, inf);
                        if (vvl == vv) h = hv;
                        first = wsub<T>((T)__builtin_amdgcn_readlane((int)hv, vv * PN + PN - 1), e1);
                    }
                }
                Hout = local ? tmax<T>((T)0, h) : h;
            } else {
                T h = wadd<T>(Mv, q);                                   // reference :854-856 / :972-974
                T hs = h;                                               // value the F recurrence opens from
                if (GAP == 2) hs = tmax<T>(tmax<T>(h, E1v), E2v);       // reference :988
                if (c == 0) { first = (T)__builtin_amdgcn_readlane((int)h, 0); first2 = first; }   // :858 / :976-977
                // leading vectors that use the plain scan (set_num == pn) go through the closed form when nothing can wrap
                const int nvec = imin(NV, end_sn - (beg_sn + c * NV) + 1);
                int nfast = local ? nvec : imin(nvec, max_pre_end_sn - (beg_sn + c * NV) + 1);
                if (nfast < 0) nfast = 0;
                if (nfast > 0 && __any(vvl < nfast && (int)h < fast_lo)) nfast = 0;
                if (b.dbg & 4) nfast = 0;
                if (nfast > 0) {
                    int fi = (int)first;
                    F1 = (T)fast_f_chain<T>((int)hs, fi, nfast, l, vvl, (int)oe1, (int)e1, (int)o1, cl1, inj1);
                    first = (T)fi;
                    if (GAP == 2) {
                        int fi2 = (int)first2;
                        F2 = (T)fast_f_chain<T>((int)hs, fi2, nfast, l, vvl, (int)oe2, (int)e2, (int)o2, cl2, inj2);
                        first2 = (T)fi2;
                    }
                }
                if (nfast < nvec) slow_f_tail(c, beg_sn, end_sn, max_pre_end_sn, nfast, hs, F1, F2, first, first2);
                if (GAP == 1) {                                          // reference :876-883
                    T tmp = tmax<T>(h, E1v);
                    T hh = tmax<T>(tmp, F1);
                    if (local) hh = tmax<T>((T)0, hh);
                    T en = tmax<T>(wsub<T>(E1v, e1), wsub<T>(hh, oe1));
                    E1out = (hh == tmp) ? en : (local ? (T)0 : inf);
                    Hout = hh;
                } else {                                                 // reference :998-1008
                    T hh = tmax<T>(hs, tmax<T>(F1, F2));
                    if (local) hh = tmax<T>((T)0, hh);
                    E1out = tmax<T>(wsub<T>(E1v, e1), wsub<T>(hh, oe1));
                    E2out = tmax<T>(wsub<T>(E2v, e2), wsub<T>(hh, oe2));
                    if (local) { E1out = tmax<T>((T)0, E1out); E2out = tmax<T>((T)0, E2out); }
                    Hout = hh;
                }
            }
            STAMP(2)
            if (in_band && !(b.dbg & 1)) {
                H[rel] = Hout;
                if (GAP != 0) {
                    H[(long long)PL_E1 * Wr + rel] = E1out;
                    H[(long long)PL_F1 * Wr + rel] = F1;
                    if (GAP == 2) { H[(long long)PL_E2 * Wr + rel] = E2out; H[(long long)PL_F2 * Wr + rel] = F2; }
                }
            }
            if (in_band) {
                if (to_ring) {
                    my_ring[rel] = Hout;
                    if (GAP != 0) my_ring[ring_cols + rel] = E1out;
                    if (GAP == 2) my_ring[2 * ring_cols + rel] = E2out;
                }
                if (need_max) {
                    // per-lane candidate; columns past qlen only exist in vector qlen/PN and are masked there
                    const bool is_end = (v == end_sn);
                    int cand = (int)Hout;
                    if (is_end && end_sn == qlen / PN && col > qlen) cand = (int)inf;
                    if (sizeof(T) == 2) {
                        const unsigned key = ((unsigned)(cand + 32768) << 16) | ((unsigned)(PN - 1 - l) << 12) | ((unsigned)is_end << 11) | (unsigned)(2047 - v);
                        am_key = key > am_key ? key : am_key;
                    } else if (!am_any || (is_end ? cand >= am_val : cand > am_val)) { am_val = cand; am_v = v; am_isend = is_end; am_any = true; }
                }
            }
            STAMP(3)
        }
        if (to_ring && lane == 0) S.b_rec[row % RB].w = row;      // H/E of this row are now readable from the score ring
        // ---- row arg-max, reference simd_abpoa_max_in_row :1043-1057 (tie-break: lowest lane, then the
        //      end_sn vector, then the lowest vector) and band hand-over :1059-1067
        int mx = d.inf_min, mi = -1;
        if (need_max && (b.dbg & 2)) { mx = 0; mi = imin(qlen, row + 1); }
        else if (need_max) {
            if (sizeof(T) == 2) {
                const unsigned kb = wave_max_u32(am_key);
                const int vmax = (int)(kb >> 16) - 32768;
                if (vmax > d.inf_min) { mx = vmax; mi = (2047 - (int)(kb & 0x7ff)) * PN + (PN - 1 - (int)((kb >> 12) & 0xf)); if (mi > qlen) mi = -1; }
            } else {
            int vmax = wave_max_i32(am_any ? am_val : INT_MIN);
            if (vmax > d.inf_min) {
                unsigned key = 0;
                if (am_any && am_val == vmax) key = ((unsigned)(PN - 1 - l) << 27) | ((unsigned)am_isend << 26) | (0x3FFFFFFu - (unsigned)am_v);
                unsigned kb = wave_max_u32(key);
                int wl = PN - 1 - (int)(kb >> 27), wv = (int)(0x3FFFFFFu - (kb & 0x3FFFFFFu));
                mx = vmax; mi = wv * PN + wl;
                if (mi > qlen) mi = -1;          // cannot happen for a value above inf_min, kept for symmetry with qi[]
            }
            }
            if (b.want_trace && lane == 0) row_max_i[row] = mi;
            if (local) { if (mx > best_score) { best_score = mx; best_i = row; best_j = mi; } }
            else if (extend) {
                if (mx > best_score) { best_score = mx; best_i = row; best_j = mi; best_row_zd = row; }
                else if (b.zdrop > 0) {
                    int delta_index = gld_i32(row_remain + best_row_zd) - (qlen - rterm + remain_end + 1);
                    int dd = delta_index - (mi - best_j); if (dd < 0) dd = -dd;
                    if (best_score - mx > b.zdrop + (int)e1 * dd) break;
                }
            }
            if (banded) {
                const int out_i = mi + 1;
                const int o0 = fastmeta ? __builtin_amdgcn_readlane(tv_o[0], ti) : -1, o1 = fastmeta ? __builtin_amdgcn_readlane(tv_o[1], ti) : -1;
                if (fastmeta && o0 < lr_blk + RL && o1 < lr_blk + RL) {      // at most two successors, both inside the LDS window
                    int2 a2 = S.l_lr[(o0 >= 0 ? o0 : 0) % RL], b2 = S.l_lr[(o1 >= 0 ? o1 : 0) % RL];
                    a2.x = imin(a2.x, out_i); a2.y = imax(a2.y, out_i); b2.x = imin(b2.x, out_i); b2.y = imax(b2.y, out_i);
                    if (o0 >= 0 && lane == 0) S.l_lr[o0 % RL] = a2;
                    if (o1 >= 0 && lane == 0) S.l_lr[o1 % RL] = b2;
                } else {
                bool far = false;
                for (int t = lane; t < on; t += 64) {
                    const int tt = os + t - obase;
                    int o = S.t_out[tt < TP ? tt : 0];
                    if (tt >= TP) o = gld_i32(out_row + os + t);
                    if (o >= 0) {
                        if (o < lr_blk + RL) {
                            int2 v2 = S.l_lr[o % RL];
                            v2.x = imin(v2.x, out_i); v2.y = imax(v2.y, out_i);
                            S.l_lr[o % RL] = v2;
                        } else {                                       // beyond the LDS window: update the HBM copy
                            if (out_i > gld_i32(g_right + o)) g_right[o] = out_i;
                            if (out_i < gld_i32(g_left + o)) g_left[o] = out_i;
                            far = true;
                        }
                    }
                }
                if (__any(far)) { far_seen = true; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                }
            }
        } else if (b.want_trace && lane == 0) row_max_i[row] = -2;
        STAMP(4)
    }
    __syncthreads();
    if (tile_end > tile_beg && tile_beg + lane < tile_end && status == 0) {      // band geometry of the last (partial) tile
        const int r = tile_beg + lane;
        if (r <= last_done) { const int4 br = S.b_rec[r % RB]; g_bsn[r] = br.x; g_esn[r] = br.y; g_coff[r] = (long long)(uint32_t)br.z * PN; }
    }
    clk1 = (long long)__builtin_amdgcn_s_memtime();
    // ---- retire the left/right window to HBM (the arrays are in/out for the caller)
    if (banded && status == 0) {
        for (int i = lane; i < RL; i += 64) { const int r = lr_blk + i; if (r < gn) { const int2 v2 = S.l_lr[r % RL]; g_left[r] = v2.x; g_right[r] = v2.y; } }
    }
    (void)last_row;
#ifdef ABPOA_HIP_PROFILE
    for (int i_ = 0; i_ < 6; ++i_) seg_keep[i_] = seg[i_];
#endif
    }   // general row loop
    TailState ts; ts.cursor = cursor; ts.n_cells = n_cells; ts.status = status; ts.rows_done = rows_done; ts.best_score = best_score; ts.best_i = best_i; ts.best_j = best_j;
    ts.clk0 = clk0; ts.clk1 = clk1;
#ifdef ABPOA_HIP_PROFILE
    for (int i_ = 0; i_ < 6; ++i_) ts.seg[i_] = seg_keep[i_];
#else
    for (int i_ = 0; i_ < 6; ++i_) ts.seg[i_] = 0;
#endif
    finish_alignment<T, GAP>(b, d, out_rec, ts);
}

// Which row loop an alignment takes (must agree between the two kernels and with engine.cpp's count).
__device__ __forceinline__ bool takes_fast(const DevBatch &b, const AlnDesc &d) {
    return b.gap_mode != ABPOA_HIP_LINEAR_GAP && b.wb >= 0 && b.align_mode == ABPOA_HIP_GLOBAL_MODE && (d.flags & ALN_FAST_OK) && b.lds.fr_cols > 0 &&
           d.qlen <= b.lds.q_cap && !(b.dbg & 64);
}

template <int GAP>
__global__ void __launch_bounds__(64) dp_kernel(const DevBatch b) {
    const int a = blockIdx.x;
    if (a >= b.n) return;
    const AlnDesc d = b.aln[a];
    if (takes_fast(b, d)) return;            // dp_fast_kernel's
    if (d.bits == 16) align_one<int16_t, GAP>(b, d, b.out + a);
    else align_one<int32_t, GAP>(b, d, b.out + a);
}

// The fast path is two kernels -- row loop, then global best + backtrack -- so that the row loop's register allocation
// (its SGPR budget above all) is not shared with the tail; the hand-over is the AlnOut record in HBM.
template <typename T, int GAP>
__device__ __forceinline__ void align_fast_rows(const DevBatch &b, const AlnDesc &d, AlnOut *out_rec) {
    const int lane = threadIdx.x & 63;
    FastIO<T> io;
    io.row_base = vgpr_ptr(b.row_base + d.row0); io.row_remain = vgpr_ptr(b.row_remain + d.row0);
    io.pred_off = vgpr_ptr(b.pred_off + d.poff0); io.pred_row = vgpr_ptr(b.pred_row + d.pred0);
    io.g_bsn = vgpr_ptr(b.dp_beg_sn + d.row0); io.g_esn = vgpr_ptr(b.dp_end_sn + d.row0); io.row_max_i = vgpr_ptr(b.row_max_i + d.row0);
    io.g_left = vgpr_ptr(b.left + d.row0); io.g_right = vgpr_ptr(b.right + d.row0); io.g_coff = vgpr_ptr(b.row_cell_off + d.row0);
    io.planes = (T *)(b.planes + d.plane_off);
    uint8_t *s_query = lds_raw + b.lds.q_off;
    { GLOBAL_AS const uint8_t *g_query = vgpr_ptr(b.query + d.query_off); for (int i = lane; i < d.qlen; i += 64) s_query[i] = g_query[i]; }
    __syncthreads();
    long long cursor = 0, n_cells = 0; int status = 0, rows_done = 0, last_done = 0;
    const long long clk0 = (long long)__builtin_amdgcn_s_memtime();
    long long fseg[6] = {0, 0, 0, 0, 0, 0};
    rows_fast<T, GAP>(b, d, io, s_query, cursor, n_cells, status, rows_done, last_done, fseg);
    const long long clk1 = (long long)__builtin_amdgcn_s_memtime();
    if (lane == 0) { GLOBAL_AS AlnOut *o = vgpr_ptr(out_rec); o->status = status; o->n_cells = n_cells; o->cells_used = cursor; o->clk_dp = clk1 - clk0; o->n_rows_done = rows_done; for (int i_ = 0; i_ < 6; ++i_) o->seg[i_] = fseg[i_]; }
}

// One kernel per score width: the row loop of one width is ~40 KB of code, and a CU pair's 64 KB instruction cache has to hold
// what its 8 or so resident wavefronts execute; an alignment of the other width is left to the other kernel.
template <int GAP, int BITS>
__global__ void __launch_bounds__(64) dp_fast_kernel(const DevBatch b) {
    const int a = blockIdx.x;
    if (a >= b.n) return;
    const AlnDesc d = b.aln[a];
    if (!takes_fast(b, d) || d.bits != BITS) return;           // dp_kernel's, or the other width's
    align_fast_rows<typename std::conditional<BITS == 16, int16_t, int32_t>::type, GAP>(b, d, b.out + a);
}

template <typename T, int GAP>
__device__ __forceinline__ void align_fast_tail(const DevBatch &b, const AlnDesc &d, AlnOut *out_rec) {
    const int lane = threadIdx.x & 63;
    uint8_t *s_query = lds_raw + b.lds.q_off;
    int32_t *s_mat = (int32_t *)(lds_raw + b.lds.mat_off);
    { GLOBAL_AS const int32_t *g_mat = vgpr_ptr(b.mat); for (int i = lane; i < b.m * b.m; i += 64) s_mat[i] = g_mat[i]; }
    { GLOBAL_AS const uint8_t *g_query = vgpr_ptr(b.query + d.query_off); for (int i = lane; i < d.qlen; i += 64) s_query[i] = g_query[i]; }
    TailState ts;
    ts.status = out_rec->status; ts.n_cells = out_rec->n_cells; ts.cursor = out_rec->cells_used; ts.rows_done = out_rec->n_rows_done;
    ts.best_score = d.inf_min; ts.best_i = 0; ts.best_j = 0;
    for (int i_ = 0; i_ < 6; ++i_) ts.seg[i_] = out_rec->seg[i_];
    ts.clk1 = (long long)__builtin_amdgcn_s_memtime(); ts.clk0 = ts.clk1 - out_rec->clk_dp;
    __syncthreads();
    finish_alignment<T, GAP, FastFmt<T, GAP>::CW>(b, d, out_rec, ts);
}

template <int GAP, int BITS>
__global__ void __launch_bounds__(64) dp_fast_tail_kernel(const DevBatch b) {
    const int a = blockIdx.x;
    if (a >= b.n) return;
    const AlnDesc d = b.aln[a];
    if (!takes_fast(b, d) || d.bits != BITS) return;
    align_fast_tail<typename std::conditional<BITS == 16, int16_t, int32_t>::type, GAP>(b, d, b.out + a);
}

template <typename K>
static hipError_t launch_one(K kern, const DevBatch &b, hipStream_t stream, int lds_bytes = -1) {
    dim3 grid(b.n), block(64);
    const size_t lds = (size_t)(lds_bytes >= 0 ? lds_bytes : b.lds.total);
    if (lds > 65536) { hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, b);
    return hipGetLastError();
}

template <int GAP>
static hipError_t launch_fast_pair(const DevBatch &b, hipStream_t stream, hipEvent_t after_rows) {
    const int mask = b.bits_mask ? b.bits_mask : 3;
    hipError_t e = hipSuccess;
    if (mask & 1) e = launch_one(dp_fast_kernel<GAP, 16>, b, stream, b.lds.total_rows);
    if (e == hipSuccess && (mask & 2)) e = launch_one(dp_fast_kernel<GAP, 32>, b, stream, b.lds.total_rows);
    if (e == hipSuccess) e = hipEventRecord(after_rows, stream);
    if (e == hipSuccess && (mask & 1)) e = launch_one(dp_fast_tail_kernel<GAP, 16>, b, stream, b.lds.total_tail);
    if (e == hipSuccess && (mask & 2)) e = launch_one(dp_fast_tail_kernel<GAP, 32>, b, stream, b.lds.total_tail);
    return e;
}

// n_fast: how many alignments of the batch take the fast row loop (engine.cpp applies takes_fast() on the host); a kernel
// with nothing to do is not launched.
hipError_t launch_dp(const DevBatch &b, int n_fast, hipStream_t stream, hipEvent_t after_rows) {
    if (b.n <= 0) return hipSuccess;
    hipError_t e = hipSuccess;
    if (n_fast > 0) {
        e = b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_fast_pair<1>(b, stream, after_rows) : launch_fast_pair<2>(b, stream, after_rows);
        if (e != hipSuccess) return e;
    }
    if (n_fast < b.n) {
        switch (b.gap_mode) {
            case ABPOA_HIP_LINEAR_GAP: e = launch_one(dp_kernel<0>, b, stream); break;
            case ABPOA_HIP_AFFINE_GAP: e = launch_one(dp_kernel<1>, b, stream); break;
            default: e = launch_one(dp_kernel<2>, b, stream); break;
        }
    }
    return e;
}

// the fast-path kernels alone (device-resident driver: every alignment of the batch is fast-eligible or skipped)
hipError_t launch_dp_fast(const DevBatch &b, hipStream_t stream, hipEvent_t after_rows) {
    if (b.n <= 0) return hipSuccess;
    return b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_fast_pair<1>(b, stream, after_rows) : launch_fast_pair<2>(b, stream, after_rows);
}

}  // namespace abpoa_hip
