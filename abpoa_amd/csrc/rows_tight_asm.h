#pragma once
// The narrow row loop's tight loop as hand-placed gfx950 assembly (round 5): int16 scores, affine gaps, direction words.
//
// What it is: the rows that rows_fast.h's tile word marks "straight-line body" (bit 17: one or two predecessors, both inside the geometry ring and the score
// ring, not a row that keeps its score records) -- turbo_body<1 | 2, SLOWV = false> + commit_row, instruction for instruction the same arithmetic
// (reference src/simd_abpoa_align.c:781-885 for the cells, :1043-1067 for the arg-max and the band hand-over, :710-720 for the band), written by hand because
// the compiler's version of this loop is 162 instructions for a one-predecessor row of which ~45 are not the algorithm: copies of every loop-carried value
// on the way to the structuriser's single loop latch, exit flags, a four-instruction loop condition, s_nop wait states between the DPP steps.  One
// wavefront per SIMD issues one instruction per 5 cycles (tools/ubench/roofs.hip: v_max_i32 / s_add_u32 / DPP alike), so a row costs its instruction
// count x 5 cycles.  Here: ~114 instructions for a one-predecessor row, ~147 for two; the DPP wait states carry the ring / arena address arithmetic.
//
// Contract with rows_fast.h (the only includer):
//   * runs rows row, row + 1, ... while each is a bit-17 row that passes the straight-line body's own conditions (band of at most 64 columns inside the
//     predecessors' bands, predecessors in the score ring, no value near the wrap limit); stops BEFORE any side effect of the first row that does not,
//     with code 0 = not a bit-17 row, 1 = declined but the copy that takes vectors beyond the predecessors' bands would accept it (turbo_body's 0),
//     2 = declined for good / a value near the wrap limit (turbo_body's -1), 3 = reached r_hi;
//   * the caller guarantees cur + 4 * (r_hi - row) <= cap_turbo (a row takes at most NV = 4 arena units), so the loop carries no arena-room test;
//   * refreshes the cached query codes (qoff0 / qoff1, qc_beg_sn) itself when the band start moves;
//   * VGPRs v92-v127 and the listed scalar temporaries are its scratch; exec is all ones (one-wavefront phase).
// Hazards observed (CDNA3 ISA 4.5): two wait states between a VALU write and a DPP read of the same VGPR (partner chain + one filler per step);
// lane selects of v_readlane / v_writelane come from SALU-written registers (row, M0) only; no VALU-written SGPR feeds a memory instruction.
#define TA_SDWA0 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
#define TA_SDWA_S1W0 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
#define TA_SDWA_S1W1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
#define TA_SDWA_S0W1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
#define TA_DPP(CTRL) " " CTRL " bank_mask:0xf\n\t"
// scratch VGPRs
#define vQA "v104"
#define vCOL "v105"
#define vX "v106"
#define vQ "v107"
#define vR0 "v108"
#define vR1 "v109"
#define vR01 "v[108:109]"
#define vKC "v110"
#define vH "v111"
#define vHSE "v112"
#define vAK "v113"
#define vS1 "v114"
#define vG "v115"
#define vF "v115"
#define vQD "v116"
#define vRO "v117"
#define vHO "v118"
#define vT2A "v119"
#define vEN "v120"
#define vE "v121"
#define vHE "v121"
#define vU "v122"
#define vD "v123"
#define vKF "v124"
#define vWD "v124"
#define vX1 "v125"
#define vB0 "v126"
#define vB1 "v127"
#define vB01 "v[126:127]"
#define vMV "v100"
#define vE1 "v101"
#define vMV2 "v102"
#define vE12 "v103"
#define vT "v106"
#define vP "v125"
#define vFS "v126"
#define vSH "v127"
#define vKM "v92"
#define vKE "v93"
#define vX2 "v94"
#define vX3 "v95"
#define vC0 "v96"
#define vC1 "v97"
#define vC01 "v[96:97]"
#define vD0 "v98"
#define vD1 "v99"
#define vD01 "v[98:99]"

// band of the row, reference :710-720, in two halves so that the ring reads -- which need the band START only -- go out before the band END and the row's
// conditions are computed (LDS latency under ~17 scalar instructions instead of ~5):  TA_BAND_BEG: mn + 1 - w in sA -> beg_sn before the max with min_pb;
// TA_BAND_END: mx + 1 + w in sB -> end_sn in sESN.  The row's static terms come from two tile registers, R1 = min(rows, qlen - remaining) - w (in sRT) and
// R2 = (qlen - remaining) + w (in sR2): beg = max(0, min(mn + 1 - w, R1)), end = min(qlen, max(mx + 1 + w, R2)) -- min / max commute with the shift by w
#define TA_BAND_BEG                                                                                          \
    "s_min_i32 %[sA], %[sA], %[sRT]\n\t"  "s_max_i32 %[sA], %[sA], 0\n\t"       "s_lshr_b32 %[sA], %[sA], 4\n\t"
#define TA_BAND_END                                                                                          \
    "s_max_i32 %[sB], %[sB], %[sR2]\n\t"  "s_min_i32 %[sB], %[sB], %[qlen]\n\t" "s_lshr_b32 %[sESN], %[sB], 4\n\t"
// the query-code cache (before the reads: the substitution score's address comes from it); LBL = path suffix
#define TA_QCACHE(LBL)                                                                                       \
    "s_cmp_lg_u32 %[sBSN], %[qcb]\n\t"     "s_cbranch_scc1 L_ref" LBL "_%=\n\t"                              \
    "L_refd" LBL "_%=:\n\t"
// conditions of the straight-line body (max_pe in sPE0, ring word in sG0); the reads of the row are in flight: L_dec waits for them.  A row whose band ends
// beyond every predecessor's (one new vector every PN rows as the band moves right) continues at SLOWLBL: the copy of the path with the literal masked scan
#define TA_CHECKS(SLOWLBL)                                                                                   \
    "s_sub_i32 %[sNV1], %[sESN], %[sBSN]\n\t"                                                                \
    "s_cmp_gt_u32 %[sNV1], 3\n\t"          "s_cbranch_scc1 L_dec_%=\n\t"                                     \
    "s_cmp_gt_i32 %[sESN], %[sPE0]\n\t"    "s_cbranch_scc1 " SLOWLBL "_%=\n\t"
#define TA_RINGCHK                                                                                           \
    "s_bitcmp0_b32 %[sG0], 24\n\t"         "s_cbranch_scc1 L_dec_%=\n\t"
// entry of a path's slow copy: exactly ONE vector beyond the predecessors' bands (the row's last), at least one inside (else: the C++ bodies)
#define TA_SLOW_ENTRY(SLOWLBL)                                                                               \
    TA_ALIGN SLOWLBL "_%=:\n\t"                                                                                       \
    "s_add_i32 %[sA], %[sPE0], 1\n\t"      "s_cmp_lg_u32 %[sESN], %[sA]\n\t"      "s_cbranch_scc1 L_dec_%=\n\t" \
    "s_cmp_gt_u32 %[sBSN], %[sPE0]\n\t"    "s_cbranch_scc1 L_dec_%=\n\t"
#define TA_S_YES(x) x
#define TA_S_NO(x) ""
// out of line: this lane's query codes for band start sBSN (chunks 0 and 1), rows_fast.h refresh_qc
#define TA_REFRESH(LBL)                                                                                      \
    "L_ref" LBL "_%=:\n\t"                                                                                   \
    "s_mov_b32 %[qcb], %[sBSN]\n\t"        "s_lshl_b32 %[sA], %[sBSN], 4\n\t"     "s_add_i32 %[sA], %[sA], -1\n\t"      \
    "v_add_u32 " vX ", %[sA], %[lane]\n\t" "v_cmp_gt_u32 vcc, %[qlen], " vX "\n\t" "v_add_u32 " vQ ", %[qb], " vX "\n\t" \
    "ds_read_u8 " vQ ", " vQ "\n\t"                                                                          \
    "v_add_u32 " vX ", 64, " vX "\n\t"     "v_cmp_gt_u32_e64 %[msk], %[qlen], " vX "\n\t" "v_add_u32 " vG ", %[qb], " vX "\n\t" \
    "ds_read_u8 " vG ", " vG "\n\t"        "v_mov_b32 " vX ", %[m]\n\t"                                      \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                               \
    "v_cndmask_b32 %[qoff0], " vX ", " vQ ", vcc\n\t"   "v_cndmask_b32_e64 %[qoff1], " vX ", " vG ", %[msk]\n\t"  \
    "v_lshl_add_u32 " vQA ", %[qoff0], 2, %[mxb]\n\t"                                                        \
    "s_branch L_refd" LBL "_%=\n\t"
// column, substitution score and the first predecessor's ring words (two LDS reads in flight)
#define TA_READS                                                                                             \
    "s_lshl_b32 %[sC0], %[sBSN], 4\n\t"    "v_add_u32 " vCOL ", %[sC0], %[lane]\n\t"                         \
    "s_ashr_i32 %[sA], %[sTB], 16\n\t"     "v_add_u32 " vX ", %[sA], " vQA "\n\t"    "ds_read_b32 " vQ ", " vX "\n\t"   \
    "s_lshl_b32 %[sA], %[sPB0], 4\n\t"     "v_xad_u32 " vX ", %[sA], -1, " vCOL "\n\t"  "v_med3_i32 " vX ", " vX ", -2, %[rc]\n\t" \
    "v_lshl_add_u32 " vX ", " vX ", 2, %[sSL0]\n\t"   "ds_read2_b32 " vR01 ", " vX " offset1:1\n\t"
// band mask, arg-max key constant, "column inside the query" mask
#define TA_MASKS                                                                                             \
    "v_cmp_ge_u32_e64 %[inb], %[sNV1], %[vvl]\n\t"                                                               \
    "v_cmp_eq_u32 vcc, %[sNV1], %[vvl]\n\t"      "v_cndmask_b32 " vKC ", %[kN], %[kE], vcc\n\t"              \
    "v_cmp_ge_i32_e64 %[amok], %[qlen], " vCOL "\n\t"  "s_and_b64 %[amok], %[amok], %[inb]\n\t"
// from h (vH) and max(h, E) (vHSE): the arg-max key, the F scan's input, both 64-lane scans interleaved; the six wait-state slots carry the ring / arena
// addresses of the row's stores and the geometry word
#define TA_SCAN                                                                                              \
    "v_add_u32 " vG ", " vH ", %[le1]\n\t"                                                                  \
    "v_lshl_add_u32 " vAK ", " vHSE ", 16, " vKC "\n\t"   "v_cndmask_b32_e64 " vAK ", 0, " vAK ", %[amok]\n\t" \
    "v_subrev_u32 " vS1 ", %[e1], " vH "\n\t"                                                               \
    "v_mov_b32_dpp " vS1 ", " vG " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                \
    "v_max_u32_dpp " vAK ", " vAK ", " vAK TA_DPP("row_shr:1 row_mask:0xf")                                  \
    "v_readlane_b32 %[sP0], %[vslot], %[row]\n\t"                                                            \
    "v_max_i32_dpp " vS1 ", " vS1 ", " vS1 TA_DPP("row_shr:1 row_mask:0xf")                                  \
    "v_max_u32_dpp " vAK ", " vAK ", " vAK TA_DPP("row_shr:2 row_mask:0xf")                                  \
    "s_lshl_b32 %[sM0], %[cur], 5\n\t"                                                                       \
    "v_max_i32_dpp " vS1 ", " vS1 ", " vS1 TA_DPP("row_shr:2 row_mask:0xf")                                  \
    "v_max_u32_dpp " vAK ", " vAK ", " vAK TA_DPP("row_shr:4 row_mask:0xf")                                  \
    "v_lshl_add_u32 " vQD ", %[lane], 2, %[sP0]\n\t"                                                         \
    "v_max_i32_dpp " vS1 ", " vS1 ", " vS1 TA_DPP("row_shr:4 row_mask:0xf")                                  \
    "v_max_u32_dpp " vAK ", " vAK ", " vAK TA_DPP("row_shr:8 row_mask:0xf")                                  \
    "v_lshl_add_u32 " vRO ", %[lane], 1, %[sM0]\n\t"                                                         \
    "v_max_i32_dpp " vS1 ", " vS1 ", " vS1 TA_DPP("row_shr:8 row_mask:0xf")                                  \
    "v_max_u32_dpp " vAK ", " vAK ", " vAK TA_DPP("row_bcast:15 row_mask:0xa")                               \
    "s_lshl_b32 %[sB], %[sESN], 12\n\t"                                                                      \
    "v_max_i32_dpp " vS1 ", " vS1 ", " vS1 TA_DPP("row_bcast:15 row_mask:0xa")                               \
    "v_max_u32_dpp " vAK ", " vAK ", " vAK TA_DPP("row_bcast:31 row_mask:0xc")                               \
    "s_or_b32 %[sB], %[sB], %[sBSN]\n\t"                                                                     \
    "v_max_i32_dpp " vS1 ", " vS1 ", " vS1 TA_DPP("row_bcast:31 row_mask:0xc")
// SLOW rows, after the scans: F of the row's last vector (vg = max_pre_end_sn + 1: set_num 2) by the reference's literal masked scan, :859-875 / :665-699 --
// rows_fast.h slow_f_vectors + set_f for exactly that vector, in the score width (int16: wrapping subtractions, compared as sign-extended low halves):
//   first = carry out of the closed-form vectors; prev = H shifted by one inside the vector (lane 0: first); f = prev - oe; four log steps
//   f = max(f, shift_S(f - S e)) with the shifted-in lanes and the lanes l > cov set to "inf" (cov = 2, 4, 8, 16).          result: vFS (sign-extended)
#define TA_SLOW_STEP(SH, COV, CTRL)                                                                          \
    "s_lshl_b32 %[sM0], %[e1], " SH "\n\t"      "v_subrev_u32 " vT ", %[sM0], " vFS "\n\t"     "v_mov_b32 " vSH ", %[infv]\n\t" \
    "v_cmp_lt_u32_e64 %[msk], " COV ", %[vl]\n\t"                                                            \
    "v_mov_b32_dpp " vSH ", " vT " " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                  \
    "v_cndmask_b32_e64 " vSH ", " vSH ", %[infv], %[msk]\n\t"                                                \
    "v_max_i32_sdwa " vFS ", sext(" vFS "), sext(" vSH ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0\n\t"
#define TA_SLOW_A                                                                                            \
    "v_max_i32 " vT ", " vS1 ", " vG "\n\t"                                                                  \
    "s_lshl_b32 %[sA], %[sNV1], 4\n\t"          "s_add_i32 %[sA], %[sA], -1\n\t"       "s_mul_i32 %[sM0], %[sA], %[e1]\n\t" \
    "v_readlane_b32 %[sP0], " vT ", %[sA]\n\t"                                                               \
    "s_sub_i32 %[sP0], %[sP0], %[sM0]\n\t"      "s_sext_i32_i16 %[sP0], %[sP0]\n\t"                          \
    "v_mov_b32 " vP ", %[sP0]\n\t"                                                                           \
    "v_mov_b32_dpp " vP ", " vH " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                  \
    "v_subrev_u32 " vFS ", %[oe1], " vP "\n\t"                                                               \
    TA_SLOW_STEP("0", "2", "row_shr:1")                                                                      \
    TA_SLOW_STEP("1", "4", "row_shr:2")                                                                      \
    TA_SLOW_STEP("2", "8", "row_shr:4")                                                                      \
    TA_SLOW_STEP("3", "16", "row_shr:8")
// ... taken by the lanes of that vector
#define TA_SLOW_F                                                                                            \
    "v_cmp_eq_u32 vcc, %[sNV1], %[vvl]\n\t"     "v_cndmask_b32 " vF ", " vF ", " vFS ", vcc\n\t"
// ... whose direction words carry the literal "where F came from" (dir_plane.h; rows_fast.h dir_literal, reference :260-300): 1 opened from H of the left
// neighbour, 2 extended from its F, 3 neither -- in vSH, 0 for the other lanes
#define TA_SLOW_LIT                                                                                          \
    "v_mov_b32_dpp " vP ", " vHO " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                \
    "v_mov_b32_dpp " vSH ", " vF " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                \
    "v_subrev_u32 " vP ", %[oe1], " vP "\n\t"   "v_bfe_i32 " vP ", " vP ", 0, 16\n\t"   "v_cmp_eq_i32_e64 %[msk], " vP ", " vF "\n\t" \
    "v_subrev_u32 " vSH ", %[e1], " vSH "\n\t"  "v_bfe_i32 " vSH ", " vSH ", 0, 16\n\t" "v_cmp_eq_i32 vcc, " vSH ", " vF "\n\t" \
    "v_cndmask_b32_e64 " vSH ", 3, 2, vcc\n\t"  "v_cndmask_b32_e64 " vSH ", " vSH ", 1, %[msk]\n\t"          \
    "v_cmp_le_u32 vcc, %[sNV1], %[vvl]\n\t"     "v_cndmask_b32 " vSH ", 0, " vSH ", vcc\n\t"
// F, H, E; EN_INSTR = the path's instruction that yields (E entering the cell) - e1 in vEN; S: slow-row text
#define TA_HEF(EN_INSTR, S)                                                                                  \
    "v_sub_u32 " vF ", " vS1 ", %[cf1]\n\t"     "v_max_i32 " vF ", " vF ", %[inj1]\n\t"                      \
    S(TA_SLOW_F)                                                                                             \
    "v_max_i32 " vHO ", " vHSE ", " vF "\n\t"   "v_subrev_u32 " vT2A ", %[oe1], " vHO "\n\t"                 \
    EN_INSTR                                                                                                 \
    "v_max_i32 " vEN ", " vEN ", " vT2A "\n\t"                                                               \
    "v_cmp_lt_i32 vcc, " vHSE ", " vF "\n\t"    "v_cndmask_b32 " vE ", " vEN ", %[infv], vcc\n\t"            \
    "v_sub_u32 " vU ", " vEN ", " vT2A "\n\t"   "v_sub_u32 " vD ", " vHO ", " vF "\n\t"     "v_min_u32 " vD ", 7, " vD "\n\t" \
    "v_perm_b32 " vHE ", " vE ", " vHO ", %[perm]\n\t"                                                       \
    S(TA_SLOW_LIT)
// direction word (kf in vKF), stores, arg-max decode, commit, loop
#define TA_TAIL(S)                                                                                           \
    "v_lshl_or_b32 " vU ", " vD ", 3, " vU "\n\t"   "v_lshl_or_b32 " vWD ", " vU ", 8, " vKF "\n\t"           \
    S("v_lshl_or_b32 " vWD ", " vSH ", 14, " vWD "\n\t")                                                     \
    "global_store_short " vRO ", " vWD ", %[planes]\n\t"                                                     \
    "v_cndmask_b32_e64 " vHE ", %[infwv], " vHE ", %[inb]\n\t"                                                   \
    "ds_write2st64_b32 " vQD ", " vHE ", %[infwv] offset1:1\n\t"                                             \
    "v_readlane_b32 %[sA], " vAK ", 63\n\t"                                                                  \
    "s_bitset1_b32 %[sB], 24\n\t"                                                                            \
    "v_writelane_b32 %[geo], %[sB], m0\n\t"     "v_writelane_b32 %[off], %[cur], m0\n\t"                     \
    "s_and_b32 %[sM0], %[sA], 63\n\t"           "s_add_i32 %[sM0], %[sM0], %[sC0]\n\t"                       \
    "s_cmp_gt_u32 %[sA], %[infk]\n\t"           "s_cselect_b32 %[sA], %[sM0], -1\n\t"      /* (row maximum above "inf": the whole key against (inf + 32768) << 16 | 0xffff) */ \
    "v_writelane_b32 %[mi], %[sA], m0\n\t"                                                                   \
    "s_add_i32 %[cur], %[cur], %[sNV1]\n\t"     "s_add_i32 %[cur], %[cur], 1\n\t"                            \
    "s_add_i32 %[row], %[row], 1\n\t"           "s_cmp_lt_i32 %[row], %[rhi]\n\t"     "s_cbranch_scc1 L_row_%=\n\t"

// ---- three / four predecessors (tile bit 18, not a row that keeps its records): the band from all of them first, then every ring read in flight
//      together, then the merges in list order with the running "first predecessor that holds the maximum" of M and E (turbo_body<4>: kfirst, kE1)
#define TA_IF4_YES(x) x
#define TA_IF4_NO(x) ""
// one more predecessor's geometry folded into (sA, sB) = (min, max) arg-max, ring word sG0; its band start / end -> PB / PE.  LANE = SGPR holding its row
#define TA_FOLD(LANE, SL, PB, PE)                                                                            \
    "v_readlane_b32 %[sM1], %[mi], " LANE "\n\t"  "v_readlane_b32 %[sG1], %[geo], " LANE "\n\t"  "v_readlane_b32 " SL ", %[vslot], " LANE "\n\t" \
    "s_min_i32 %[sA], %[sA], %[sM1]\n\t"          "s_max_i32 %[sB], %[sB], %[sM1]\n\t"           "s_and_b32 %[sG0], %[sG0], %[sG1]\n\t" \
    "s_and_b32 " PB ", %[sG1], 0xfff\n\t"         "s_bfe_u32 " PE ", %[sG1], 0xc000c\n\t"
// ring read of a further predecessor: X = its column offset (kept for the range masks), PAIR = destination; then PE becomes its band width in columns
#define TA_READK(X, PAIR, SL, PB, PE)                                                                        \
    "s_lshl_b32 %[sA], " PB ", 4\n\t"             "v_subrev_u32 " X ", %[sA], " vCOL "\n\t"                  \
    "v_add_u32 " vT ", -1, " X "\n\t"             "v_med3_i32 " vT ", " vT ", -2, %[rc]\n\t"                 \
    "v_lshl_add_u32 " vT ", " vT ", 2, " SL "\n\t"  "ds_read2_b32 " PAIR ", " vT " offset1:1\n\t"            \
    "s_sub_i32 " PE ", " PE ", " PB "\n\t"        "s_lshl_b32 " PE ", " PE ", 4\n\t"             "s_add_i32 " PE ", " PE ", 16\n\t"
// merge of predecessor KIDX (1 + list index): words W0 / W1, column offset X, band width PE      turbo_body merge_pred, reference :812-851
#define TA_MERGE(KIDX, W0, W1, X, PE)                                                                        \
    "v_max_i32_sdwa " vT ", " vMV ", sext(" W0 ")" TA_SDWA_S1W0                                              \
    "v_cmp_lt_i32 vcc, " vMV ", " vT "\n\t"                                                                  \
    "s_add_i32 %[sA], " PE ", 16\n\t"             "v_cmp_gt_u32_e64 %[msk], %[sA], " X "\n\t"    "s_and_b64 vcc, vcc, %[msk]\n\t" \
    "v_cndmask_b32_e64 " vKM ", " vKM ", " KIDX ", vcc\n\t"   "v_cndmask_b32_e64 " vMV ", " vMV ", " vT ", %[msk]\n\t" \
    "v_max_i32_sdwa " vT ", " vE1 ", sext(" W1 ")" TA_SDWA_S1W1                                              \
    "v_cmp_lt_i32 vcc, " vE1 ", " vT "\n\t"                                                                  \
    "v_cmp_gt_u32_e64 %[msk], " PE ", " X "\n\t"  "s_and_b64 vcc, vcc, %[msk]\n\t"                           \
    "v_cndmask_b32_e64 " vKE ", " vKE ", " KIDX ", vcc\n\t"   "v_cndmask_b32_e64 " vE1 ", " vE1 ", " vT ", %[msk]\n\t"
// the part of the multi-predecessor path behind its conditions; S: slow-row text
#define TA_MULTI_POST(IF4, S)                                                                                \
    TA_RINGCHK                                                                                               \
    TA_MASKS                                                                                                 \
    "v_mov_b32 " vKM ", 1\n\t"                    "v_mov_b32 " vKE ", 1\n\t"                                 \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                               \
    "v_bfe_i32 " vMV ", " vR0 ", 0, 16\n\t"       "v_ashrrev_i32 " vE1 ", 16, " vR1 "\n\t"                   \
    TA_MERGE("2", vB0, vB1, vX1, "%[sPE1]")                                                                  \
    TA_MERGE("3", vC0, vC1, vX2, "%[sPE2]")                                                                  \
    IF4(TA_MERGE("4", vD0, vD1, vX3, "%[sPE3]"))                                                             \
    "v_add_u32 " vH ", " vMV ", " vQ "\n\t"                                                                  \
    "v_min_i32 " vT ", " vH ", " vE1 "\n\t"                                                                  \
    "v_cmp_gt_i32 vcc, %[fastlo], " vT "\n\t"     "s_and_b64 %[msk], %[inb], vcc\n\t"   "s_cbranch_scc1 L_x2_%=\n\t"  \
    "v_max_i32 " vHSE ", " vH ", " vE1 "\n\t"                                                                \
    TA_SCAN                                                                                                  \
    S(TA_SLOW_A)                                                                                             \
    TA_HEF("v_subrev_u32 " vEN ", %[e1], " vE1 "\n\t", S)                                                    \
    "v_cmp_eq_u32 vcc, " vH ", " vHO "\n\t"       "v_cndmask_b32 " vKF ", 0, " vKM ", vcc\n\t"               \
    "v_lshl_or_b32 " vKF ", " vKE ", 4, " vKF "\n\t"                                                         \
    TA_TAIL(S)                                                                                               \
    "s_branch L_end_%=\n\t"
#define TA_MULTI(IF4, LBL)                                                                                   \
    "s_mov_b32 %[sA], %[sM0]\n\t"                 "s_mov_b32 %[sB], %[sM0]\n\t"                              \
    "s_and_b32 %[sPB0], %[sG0], 0xfff\n\t"        "s_bfe_u32 %[sPE0], %[sG0], 0xc000c\n\t"                   \
    TA_FOLD("%[sP1]", "%[sSL1]", "%[sPB1]", "%[sPE1]")                                                       \
    TA_FOLD("%[sP2]", "%[sSL2]", "%[sPB2]", "%[sPE2]")                                                       \
    IF4(TA_FOLD("%[sP3]", "%[sSL3]", "%[sPB3]", "%[sPE3]"))                                                  \
    "s_add_i32 %[sA], %[sA], %[c1]\n\t"               "s_add_i32 %[sB], %[sB], %[c2]\n\t"                            \
    TA_BAND_BEG                                                                                              \
    "s_min_u32 %[sESN], %[sPB0], %[sPB1]\n\t"     "s_min_u32 %[sESN], %[sESN], %[sPB2]\n\t"   IF4("s_min_u32 %[sESN], %[sESN], %[sPB3]\n\t") \
    "s_max_u32 %[sBSN], %[sA], %[sESN]\n\t"                                                                  \
    TA_QCACHE(LBL)                                                                                           \
    TA_READS                                                                                                 \
    "s_max_u32 %[sPE0], %[sPE0], %[sPE1]\n\t"     "s_max_u32 %[sPE0], %[sPE0], %[sPE2]\n\t" IF4("s_max_u32 %[sPE0], %[sPE0], %[sPE3]\n\t") \
    TA_READK(vX1, vB01, "%[sSL1]", "%[sPB1]", "%[sPE1]")                                                     \
    TA_READK(vX2, vC01, "%[sSL2]", "%[sPB2]", "%[sPE2]")                                                     \
    IF4(TA_READK(vX3, vD01, "%[sSL3]", "%[sPB3]", "%[sPE3]"))                                                \
    TA_BAND_END                                                                                              \
    TA_CHECKS("L_s" LBL)                                                                                     \
    TA_MULTI_POST(IF4, TA_S_NO)                                                                              \
    TA_SLOW_ENTRY("L_s" LBL)                                                                                 \
    TA_MULTI_POST(IF4, TA_S_YES)

// one predecessor, behind its conditions
#define TA_ONE_POST(S)                                                                                       \
    TA_RINGCHK                                                                                               \
    TA_MASKS                                                                                                 \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                               \
    "v_add_u32_sdwa " vH ", sext(" vR0 "), " vQ TA_SDWA0                                                     \
    "v_min_i32_sdwa " vT ", " vH ", sext(" vR1 ")" TA_SDWA_S1W1                                              \
    "v_cmp_gt_i32 vcc, %[fastlo], " vT "\n\t"   "s_and_b64 %[msk], %[inb], vcc\n\t"   "s_cbranch_scc1 L_x2_%=\n\t"  \
    "v_max_i32_sdwa " vHSE ", " vH ", sext(" vR1 ")" TA_SDWA_S1W1                                            \
    TA_SCAN                                                                                                  \
    S(TA_SLOW_A)                                                                                             \
    TA_HEF("v_sub_u32_sdwa " vEN ", sext(" vR1 "), %[e1]" TA_SDWA_S0W1, S)                                   \
    "v_cmp_eq_u32 vcc, " vH ", " vHO "\n\t"     "v_cndmask_b32_e64 " vKF ", 16, 17, vcc\n\t"                 \
    TA_TAIL(S)                                                                                               \
    "s_branch L_end_%=\n\t"
// two predecessors, behind their conditions
#define TA_TWO_POST(S)                                                                                       \
    TA_RINGCHK                                                                                               \
    TA_MASKS                                                                                                 \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                               \
    "v_bfe_i32 " vMV ", " vR0 ", 0, 16\n\t"     "v_ashrrev_i32 " vE1 ", 16, " vR1 "\n\t"                     \
    "v_max_i32_sdwa " vT ", " vMV ", sext(" vB0 ")" TA_SDWA_S1W0                                             \
    "v_cmp_gt_u32 vcc, %[sP1], " vX1 "\n\t"     "v_cndmask_b32 " vMV2 ", " vMV ", " vT ", vcc\n\t"           \
    "v_max_i32_sdwa " vT ", " vE1 ", sext(" vB1 ")" TA_SDWA_S1W1                                             \
    "v_cmp_gt_u32 vcc, %[sM1], " vX1 "\n\t"     "v_cndmask_b32 " vE12 ", " vE1 ", " vT ", vcc\n\t"           \
    "v_add_u32 " vH ", " vMV2 ", " vQ "\n\t"                                                                 \
    "v_min_i32 " vT ", " vH ", " vE12 "\n\t"                                                                 \
    "v_cmp_gt_i32 vcc, %[fastlo], " vT "\n\t"   "s_and_b64 %[msk], %[inb], vcc\n\t"   "s_cbranch_scc1 L_x2_%=\n\t"  \
    "v_max_i32 " vHSE ", " vH ", " vE12 "\n\t"                                                               \
    TA_SCAN                                                                                                  \
    S(TA_SLOW_A)                                                                                             \
    TA_HEF("v_subrev_u32 " vEN ", %[e1], " vE12 "\n\t", S)                                                   \
    "v_cmp_ne_u32 vcc, " vMV2 ", " vMV "\n\t"   "v_cndmask_b32_e64 " vKF ", 1, 2, vcc\n\t"                   \
    "v_cmp_eq_u32 vcc, " vH ", " vHO "\n\t"     "v_cndmask_b32 " vKF ", 0, " vKF ", vcc\n\t"                 \
    "v_cmp_ne_u32 vcc, " vE12 ", " vE1 "\n\t"   "v_cndmask_b32_e64 " vX ", 16, 32, vcc\n\t"                  \
    "v_or_b32 " vKF ", " vKF ", " vX "\n\t"                                                                  \
    TA_TAIL(S)                                                                                               \
    "s_branch L_end_%=\n\t"

// The loop head and the heads of the other paths sit on 64-byte boundaries (whole instruction-cache lines; measured on one box, same run: 49.65 against 50.05 ms
// for the kernel).  The padding in front of the loop head is jumped over; the other heads follow unconditional branches.  -DABPOA_HIP_ASM_ALIGN=n: another power.
#ifndef ABPOA_HIP_ASM_ALIGN
#define ABPOA_HIP_ASM_ALIGN 6
#endif
#define TA_STR2(x) #x
#define TA_STR(x) TA_STR2(x)
#define TA_ALIGN ".p2align " TA_STR(ABPOA_HIP_ASM_ALIGN) "\n\t"
#define TA_LOOP_HEAD "s_branch L_row_%=\n\t" TA_ALIGN
#define TIGHT_ASM_I16_AFFINE_DIR                                                                             \
    "v_lshl_add_u32 " vQA ", %[qoff0], 2, %[mxb]\n\t"                                                        \
    "s_mov_b32 %[code], 3\n\t"                                                                               \
    TA_LOOP_HEAD                                                                                             \
    "L_row_%=:\n\t"                                                                                          \
    "v_readlane_b32 %[sM], %[tvmeta], %[row]\n\t"   "v_readlane_b32 %[sTB], %[tvtb], %[row]\n\t"   "v_readlane_b32 %[sRT], %[tvr1], %[row]\n\t" \
    "v_readlane_b32 %[sR2], %[tvr2], %[row]\n\t"                                                            \
    "s_and_b32 m0, %[row], 63\n\t"                                                                           \
    "s_and_b32 %[sA], %[sTB], 0xff\n\t"         "s_sub_i32 %[sP0], %[row], %[sA]\n\t"                        \
    "v_readlane_b32 %[sM0], %[mi], %[sP0]\n\t"  "v_readlane_b32 %[sG0], %[geo], %[sP0]\n\t"   "v_readlane_b32 %[sSL0], %[vslot], %[sP0]\n\t" \
    "s_bitcmp1_b32 %[sM], 17\n\t"               "s_cbranch_scc0 L_n17_%=\n\t"                                \
    "s_bitcmp1_b32 %[sM], 9\n\t"                "s_cbranch_scc1 L_two_%=\n\t"                                \
    /* ---------------- one predecessor */                                                                   \
    "s_add_i32 %[sA], %[sM0], %[c1]\n\t"        "s_add_i32 %[sB], %[sM0], %[c2]\n\t"                                 \
    TA_BAND_BEG                                                                                              \
    "s_and_b32 %[sPB0], %[sG0], 0xfff\n\t"      "s_max_u32 %[sBSN], %[sA], %[sPB0]\n\t"                      \
    TA_QCACHE("1")                                                                                           \
    TA_READS                                                                                                 \
    TA_BAND_END                                                                                              \
    "s_bfe_u32 %[sPE0], %[sG0], 0xc000c\n\t"                                                                 \
    TA_CHECKS("L_s1")                                                                                        \
    TA_ONE_POST(TA_S_NO)                                                                                     \
    TA_SLOW_ENTRY("L_s1")                                                                                    \
    TA_ONE_POST(TA_S_YES)                                                                                    \
    /* ---------------- two predecessors */                                                                  \
    TA_ALIGN "L_two_%=:\n\t"                                                                                          \
    "s_bfe_u32 %[sA], %[sTB], 0x80008\n\t"      "s_sub_i32 %[sP1], %[row], %[sA]\n\t"                        \
    "v_readlane_b32 %[sM1], %[mi], %[sP1]\n\t"  "v_readlane_b32 %[sG1], %[geo], %[sP1]\n\t"   "v_readlane_b32 %[sSL1], %[vslot], %[sP1]\n\t" \
    "s_min_i32 %[sA], %[sM0], %[sM1]\n\t"       "s_max_i32 %[sB], %[sM0], %[sM1]\n\t"                        \
    "s_add_i32 %[sA], %[sA], %[c1]\n\t"             "s_add_i32 %[sB], %[sB], %[c2]\n\t"                              \
    TA_BAND_BEG                                                                                              \
    "s_and_b32 %[sPB0], %[sG0], 0xfff\n\t"      "s_and_b32 %[sPB1], %[sG1], 0xfff\n\t"                       \
    "s_min_u32 %[sESN], %[sPB0], %[sPB1]\n\t"   "s_max_u32 %[sBSN], %[sA], %[sESN]\n\t"                      \
    TA_QCACHE("2")                                                                                           \
    TA_READS                                                                                                 \
    "s_lshl_b32 %[sA], %[sPB1], 4\n\t"          "v_subrev_u32 " vX1 ", %[sA], " vCOL "\n\t"                  \
    "v_add_u32 " vX ", -1, " vX1 "\n\t"         "v_med3_i32 " vX ", " vX ", -2, %[rc]\n\t"                   \
    "v_lshl_add_u32 " vX ", " vX ", 2, %[sSL1]\n\t"   "ds_read2_b32 " vB01 ", " vX " offset1:1\n\t"           \
    TA_BAND_END                                                                                              \
    "s_bfe_u32 %[sPE0], %[sG0], 0xc000c\n\t"    "s_bfe_u32 %[sPE1], %[sG1], 0xc000c\n\t"                     \
    "s_sub_i32 %[sM1], %[sPE1], %[sPB1]\n\t"    "s_lshl_b32 %[sM1], %[sM1], 4\n\t"                           \
    "s_add_i32 %[sM1], %[sM1], 16\n\t"          "s_add_i32 %[sP1], %[sM1], 16\n\t"                           \
    "s_max_u32 %[sPE0], %[sPE0], %[sPE1]\n\t"   "s_and_b32 %[sG0], %[sG0], %[sG1]\n\t"                       \
    TA_CHECKS("L_s2")                                                                                        \
    TA_TWO_POST(TA_S_NO)                                                                                     \
    TA_SLOW_ENTRY("L_s2")                                                                                    \
    TA_TWO_POST(TA_S_YES)                                                                                    \
    /* ---------------- three / four predecessors */                                                         \
    TA_ALIGN "L_n17_%=:\n\t"                                                                                          \
    "s_bitcmp1_b32 %[sM], 18\n\t"               "s_cbranch_scc0 L_x0_%=\n\t"                                 \
    "s_bitcmp1_b32 %[sM], 21\n\t"               "s_cbranch_scc1 L_x0_%=\n\t"                                 \
    "v_readlane_b32 %[sP2], %[tvp2], %[row]\n\t"  "v_readlane_b32 %[sP3], %[tvp3], %[row]\n\t"               \
    "s_bfe_u32 %[sA], %[sTB], 0x80008\n\t"      "s_sub_i32 %[sP1], %[row], %[sA]\n\t"                        \
    "s_bitcmp1_b32 %[sM], 10\n\t"               "s_cbranch_scc1 L_four_%=\n\t"                               \
    TA_MULTI(TA_IF4_NO, "3")                                                                                 \
    TA_ALIGN "L_four_%=:\n\t"                                                                                         \
    TA_MULTI(TA_IF4_YES, "4")                                                                                \
    /* ---------------- out of line */                                                                       \
    TA_REFRESH("1")                                                                                          \
    TA_REFRESH("2")                                                                                          \
    TA_REFRESH("3")                                                                                          \
    TA_REFRESH("4")                                                                                          \
    "L_dec_%=:\n\t"                                                                                          \
    "s_waitcnt lgkmcnt(0)\n\t"                  /* (the row's ring reads land in scratch registers the compiler may use after the block) */ \
    "s_mov_b32 %[code], 2\n\t"                                                                               \
    "s_cmp_gt_u32 %[sNV1], 3\n\t"               "s_cbranch_scc1 L_end_%=\n\t"                                \
    "s_cmp_gt_i32 %[sBSN], %[sPE0]\n\t"         "s_cbranch_scc1 L_end_%=\n\t"                                \
    "s_bitcmp0_b32 %[sG0], 24\n\t"              "s_cbranch_scc1 L_end_%=\n\t"                                \
    "s_mov_b32 %[code], 1\n\t"                  "s_branch L_end_%=\n\t"                                      \
    "L_x0_%=:\n\t"  "s_mov_b32 %[code], 0\n\t"  "s_branch L_end_%=\n\t"                                      \
    "L_x2_%=:\n\t"  "s_mov_b32 %[code], 2\n\t"                                                               \
    "L_end_%=:\n\t"
