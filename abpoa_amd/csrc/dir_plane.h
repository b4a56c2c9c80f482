/* Direction plane: what the row loops leave in HBM per DP cell instead of its scores, and what the backtrack reads.
 *
 * The reference backtrack (src/simd_abpoa_align.c:109-429) compares stored H / E / F values; every one of those comparisons is a
 * fact about ONE cell that is known when the cell is computed, so the row loop records the facts and the scores never leave the chip:
 *
 *   kM    1 + list index of the first predecessor whose H[.][j-1] + s(base, query) equals H[i][j]; 0 = none        (:130-160)
 *   kE1/2 1 + list index of the first predecessor that holds the maximum E1 / E2 entering the cell (the only one the
 *         reference's in-order scan can stop at, for both forms of its test: H == pre_E and E == pre_E - e)       (:170-230)
 *   uE1/2 max(o - (H - Ein), 0) in [0, o]:  == o  <=>  H == Ein (the deletion is what gives H);  == 0  <=>  E leaving the cell was
 *         opened from H (H - oe == E: the reference then continues with M|F, else with E only)
 *   dF1/2 min(H - F, cap):  == 0  <=>  H == F.  It also decides where F of the NEXT column came from -- the reference tests
 *         "H[j-1] - oe == F[j]" first, then "F[j-1] - e == F[j]" (:260-300), and F[j] = max(hv[j-1] - oe, F[j-1] - e) where hv is H before
 *         the F terms are merged in -- by the rule dir_f_origin() below: with H[j-1] == hv[j-1] (H equals its match or one of its E terms: kM
 *         != 0 or uE == o) "opened" <=> dF >= o, else extended; with H[j-1] > hv[j-1] (H is an F term) "opened" <=> dF == o -- the
 *         reference's comparison can only hold by coincidence then -- else extended.  The one case the word cannot decide, H[j-1] an F term
 *         and dF > o with F[j] strictly opened from hv[j-1] ("neither" for the reference), cannot lie on a backtrack path when e1 >= e2:
 *         an F chain that starts below H[j-1] never overtakes the chain H[j-1] itself continues.
 *   lF1/2 literal override of that rule for the cell's own F, written only where the reference's masked F scan (SIMD_SET_F with
 *         set_num < pn, :665-699) can differ from the recurrence -- vectors beyond every predecessor's band -- and by the exact row
 *         bodies: 0 = use dF of column j-1, 1 = opened from H[j-1], 2 = extended from F[j-1], 3 = neither
 *
 * Affine: one 16-bit word; convex: one 32-bit word.  Usable when 1 <= o1 <= 7 (convex: and 1 <= o2 <= 31, e1 >= e2) and no row has more
 * than 15 predecessors; anything else keeps the score-record arenas (FastFmt).  Plain C: shared with the CPU model in oracle/dir_model.c.
 */
#ifndef ABPOA_DIR_PLANE_H
#define ABPOA_DIR_PLANE_H

#define DIR_K_BITS 4
#define DIR_K_MAX 15

/* affine (16 bits): kM [0,4) kE1 [4,8) uE1 [8,11) dF1 [11,14) lF1 [14,16) */
#define DIRA_KM_SH 0
#define DIRA_KE1_SH 4
#define DIRA_UE1_SH 8
#define DIRA_DF1_SH 11
#define DIRA_LF1_SH 14
#define DIRA_CAP1 7
/* convex (32 bits): kM [0,4) kE1 [4,8) kE2 [8,12) uE1 [12,15) uE2 [15,20) dF1 [20,23) dF2 [23,28) lF1 [28,30) lF2 [30,32) */
#define DIRC_KM_SH 0
#define DIRC_KE1_SH 4
#define DIRC_KE2_SH 8
#define DIRC_UE1_SH 12
#define DIRC_UE2_SH 15
#define DIRC_DF1_SH 20
#define DIRC_DF2_SH 23
#define DIRC_LF1_SH 28
#define DIRC_LF2_SH 30
#define DIRC_CAP1 7
#define DIRC_CAP2 31

#define DIR_LIT_NONE 0
#define DIR_LIT_OPEN 1
#define DIR_LIT_EXT 2
#define DIR_LIT_NEITHER 3

#ifdef __HIPCC__
#define DIR_FN __host__ __device__ static inline
#else
#define DIR_FN static inline
#endif
/* usable for these penalties?  (gap_mode: 1 affine, 2 convex) */
DIR_FN int dir_plane_usable(int gap_mode, int o1, int e1, int o2, int e2) {
    if (gap_mode == 1) return o1 >= 1 && o1 <= DIRA_CAP1 && e1 >= 0;
    if (gap_mode == 2) return o1 >= 1 && o1 <= DIRC_CAP1 && o2 >= 1 && o2 <= DIRC_CAP2 && e1 >= e2 && e2 >= 0;
    return 0;
}
/* Where F[j] of plane x came from, decided from the word of column j-1: h_is_hv = that cell's H equals its match or one of its E terms,
 * dF = its field of plane x, o = the plane's gap-open penalty.  Returns DIR_LIT_OPEN or DIR_LIT_EXT. */
DIR_FN int dir_f_origin(int h_is_hv, int dF, int o) {
    return (h_is_hv ? dF >= o : dF == o) ? 1 /* DIR_LIT_OPEN */ : 2 /* DIR_LIT_EXT */;
}

#endif
