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
#pragma once
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
constexpr int BTP = 128;    // backtrack tile: predecessor entries (a window of 30-64 rows has 45-100; more -> the one-read-at-a-time step; 256 cost 4 KB more LDS per tail workgroup)

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
__device__ __forceinline__ void gld_async_cell(int &v, const int32_t *p) { gld_async(v, p); }
__device__ __forceinline__ void gld_async_cell(int &v, const int16_t *p) { asm volatile("global_load_sshort %0, %1, off" : "=v"(v) : "v"((GLOBAL_AS const int16_t *)p) : "memory"); }
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
// Two / three independent 64-lane reductions of the DPP_SCAN6 shape with their steps interleaved: the partners' instructions are
// the wait states a DPP read needs after a VALU write of the same VGPR (two chains: one s_nop 0 per step; three: none).
#define DPP_ROW2(OPA, OPB, CTRL) OPA " %0, %0, %0 " CTRL "\n\t" OPB " %1, %1, %1 " CTRL "\n\t"
#define DPP_ROW3(OPA, OPB, OPC, CTRL) OPA " %0, %0, %0 " CTRL "\n\t" OPB " %1, %1, %1 " CTRL "\n\t" OPC " %2, %2, %2 " CTRL "\n\t"
#define DPP_SCAN6_2(OPA, OPB)                                                           \
    "s_nop 1\n\t" DPP_ROW2(OPA, OPB, "row_shr:1 row_mask:0xf bank_mask:0xf")            \
    "s_nop 0\n\t" DPP_ROW2(OPA, OPB, "row_shr:2 row_mask:0xf bank_mask:0xf")            \
    "s_nop 0\n\t" DPP_ROW2(OPA, OPB, "row_shr:4 row_mask:0xf bank_mask:0xf")            \
    "s_nop 0\n\t" DPP_ROW2(OPA, OPB, "row_shr:8 row_mask:0xf bank_mask:0xf")            \
    "s_nop 0\n\t" DPP_ROW2(OPA, OPB, "row_bcast:15 row_mask:0xa bank_mask:0xf")         \
    "s_nop 0\n\t" DPP_ROW2(OPA, OPB, "row_bcast:31 row_mask:0xc bank_mask:0xf")
#define DPP_SCAN6_3(OPA, OPB, OPC)                                                      \
    "s_nop 1\n\t" DPP_ROW3(OPA, OPB, OPC, "row_shr:1 row_mask:0xf bank_mask:0xf")       \
    DPP_ROW3(OPA, OPB, OPC, "row_shr:2 row_mask:0xf bank_mask:0xf")                     \
    DPP_ROW3(OPA, OPB, OPC, "row_shr:4 row_mask:0xf bank_mask:0xf")                     \
    DPP_ROW3(OPA, OPB, OPC, "row_shr:8 row_mask:0xf bank_mask:0xf")                     \
    DPP_ROW3(OPA, OPB, OPC, "row_bcast:15 row_mask:0xa bank_mask:0xf")                  \
    DPP_ROW3(OPA, OPB, OPC, "row_bcast:31 row_mask:0xc bank_mask:0xf")
// (signed scan, unsigned reduction) / (signed, signed) / (signed, signed, unsigned) / (signed, signed, signed): prefix max in every
// lane, wave maximum in lane 63
__device__ __forceinline__ void wave_scan2_iu(int &a, unsigned &k) { asm(DPP_SCAN6_2("v_max_i32_dpp", "v_max_u32_dpp") : "+v"(a), "+v"(k)); }
__device__ __forceinline__ void wave_scan2_ii(int &a, int &c) { asm(DPP_SCAN6_2("v_max_i32_dpp", "v_max_i32_dpp") : "+v"(a), "+v"(c)); }
__device__ __forceinline__ void wave_scan3_iiu(int &a, int &c, unsigned &k) { asm(DPP_SCAN6_3("v_max_i32_dpp", "v_max_i32_dpp", "v_max_u32_dpp") : "+v"(a), "+v"(c), "+v"(k)); }
__device__ __forceinline__ void wave_scan3_iii(int &a, int &c, int &k) { asm(DPP_SCAN6_3("v_max_i32_dpp", "v_max_i32_dpp", "v_max_i32_dpp") : "+v"(a), "+v"(c), "+v"(k)); }
__device__ __forceinline__ unsigned wave_max_u32_v(unsigned x) { asm(DPP_SCAN6("v_max_u32_dpp") : "+v"(x)); return x; }      // (complete in lane 63, stays in the VGPR)
// maximum over aligned groups of G = 2 / 4 / 8 lanes, left in every lane of the group
template <int G> __device__ __forceinline__ int group_allmax_i32(int x) {
    x = imax(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));                   // quad_perm [1,0,3,2]
    if (G >= 4) x = imax(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));       // quad_perm [2,3,0,1]
    if (G >= 8) x = imax(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false));      // row_half_mirror
    return x;
}
template <int G> __device__ __forceinline__ unsigned group_allmax_u32(unsigned x) {
    auto umax = [](unsigned p, unsigned q) { return p > q ? p : q; };
    x = umax(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xF, 0xF, false));
    if (G >= 4) x = umax(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xF, 0xF, false));
    if (G >= 8) x = umax(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xF, 0xF, false));
    return x;
}
// workgroup barrier for LDS hand-overs only: waits for this wave's LDS operations, NOT for its outstanding global stores
// (__syncthreads() would add s_waitcnt vmcnt(0), i.e. the HBM round trip of the score-plane stores, to every row)
// ABPOA_HIP_ONE_WAVE_PHASE (poa_rounds.hip): the row loop and the tail run on ONE wavefront of a larger workgroup whose other wavefronts wait at
// a workgroup barrier for it -- an s_barrier inside the phase would release them, so a "barrier" is then only the wait for this wave's own operations
#ifdef ABPOA_HIP_ONE_WAVE_PHASE
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#define WG_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define WG_SYNC() __syncthreads()
#endif
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

// Which row loop an alignment takes (must agree between the two kernels and with engine.cpp's count).
__device__ __forceinline__ bool takes_fast(const DevBatch &b, const AlnDesc &d) {
    return fast_global_job(b.gap_mode, b.align_mode, b.wb, b.e1) && fast_global_aln(b.gap_mode, d.w, d.pad0) && (d.flags & ALN_FAST_OK) && b.lds.fr_cols > 0 &&
           d.qlen <= b.lds.q_cap && !(b.dbg & 64);
}

// ... and among those, the ones whose rows are wide enough for NW wavefronts per alignment (dp_wide_rows.hip); the rest keep one wavefront
// (AlnDesc.pad0: columns the rows are expected to be wider than 2 w -- the device-resident driver sets it for read-sets whose reads differ much in length;
//  half of it counts as band half-width for the choice of the row loop, nothing else reads it)
__device__ __forceinline__ bool takes_wide(const DevBatch &b, const AlnDesc &d) {
    const int weff = d.w + (d.pad0 >> 1);
    return b.lds.wide_nw >= 1 && weff >= b.lds.wide_w_lo && weff <= b.lds.wide_w_hi;
}

// ... and which of the fast alignments write direction words instead of score records (dir_plane.h): dir_mode 1 = the narrow-band ones of the launch,
// dir_mode 2 = the wide-band ones too.
// (Measured on MI355X: on 1 kb reads the words cost the row loop 5-8 % and save 36-42 % of the backtrack, a net 8-11 %; on 10 kb reads the all-chunks
//  row loop loses 18-21 % -- more than the backtrack, a fifth of the time there, gains -- so wide-band alignments keep their records while the record
//  arenas of the job fit the device; the host picks mode 2 when they do not: an eighth of the arena bytes, twice the read-sets in flight.)
__device__ __forceinline__ bool takes_dir(const DevBatch &b, const AlnDesc &d) { return b.dir_mode == 2 || (b.dir_mode == 1 && !takes_wide(b, d)); }

// kernel launch helper shared by the translation units (block = NT threads)
template <typename K>
static hipError_t launch_one(K kern, const DevBatch &b, hipStream_t stream, int lds_bytes, int block_threads = 64) {
    dim3 grid(b.n), block(block_threads);
    const size_t lds = (size_t)(lds_bytes >= 0 ? lds_bytes : b.lds.total);
    if (lds > 65536) { hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, b);
    return hipGetLastError();
}

// per-TU launchers (one translation unit per kernel family: parallel builds, one row loop per file)
hipError_t launch_fast_rows(const DevBatch &b, hipStream_t stream);       // dp_fast_rows.hip
hipError_t launch_wide_rows(const DevBatch &b, hipStream_t stream);       // dp_wide_rows.hip
hipError_t launch_team_rows(const DevBatch &b, hipStream_t stream);       // dp_team_rows.hip
hipError_t launch_local_rows(const DevBatch &b, hipStream_t stream);      // dp_local_rows.hip
hipError_t launch_fast_tail(const DevBatch &b, hipStream_t stream);       // dp_fast_tail.hip
hipError_t launch_general(const DevBatch &b, hipStream_t stream);         // dp_general.hip

}  // namespace abpoa_hip
