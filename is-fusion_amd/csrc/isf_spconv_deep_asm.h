// isf_spconv_deep_asm.h -- one STEP of spconv_deep_kernel (isf_spconv_deep.hip) as a single hand-scheduled gfx950
// instruction stream: the products of step s AND the issue of everything step s + 1 needs, interleaved.
//
// Register file (by hand): accumulators = the compiler's VGPRs (operands %[cRN]); a[0:31] = two buffers of weight
// fragments (isf_spconv16_mult.h: column tiles in pairs, two pairs in flight); a[32:47] = the row fragments of this
// step, read from the wave's LDS transit at the top (row group 0: hi a[32:35], lo a[36:39]; row group 1: a[40:43],
// a[44:47]).  Once those reads have landed the transit is free, and the stream issues, one vector-memory instruction per
// three (two) MFMAs:
//   * the <= 4 LDS-DMA gathers of step s + 1 (a lane quad = 64 contiguous bytes of one row; M0 = the transit slot, for the
//     lo halves M0 = slot - 64 with offset:64 -- the offset field moves the global and the LDS address alike),
//   * the wave's four 1-KiB pieces of the next weight stage (one M0, offsets 0 .. 3072).
// So the address unit sees one instruction of a wave every ~50 cycles instead of eight in a burst behind the barrier, and
// no wave stands in an issue phase while its SIMD's matrix pipe idles (the phase trace of the tile kernel: 35 % of a
// wave's loop in the issue of eight instructions, 976 cycles per step at p50).
// %[fl] bits: 0 / 1 = row group 0 / 1 gathers for step s + 1, 2 = there is a step s + 1 (weights), 3 / 4 = row group
// 0 / 1 multiplies in step s.  Products per accumulator in the order a_lo b_hi -> a_hi b_lo -> a_hi b_hi.
#pragma once
#include "isf_spconv16_mult.h"

#define ISF_DA_A0H "a[32:35]"
#define ISF_DA_A0L "a[36:39]"
#define ISF_DA_A1H "a[40:43]"
#define ISF_DA_A1L "a[44:47]"

// issue items (C = case letter, unique labels).  Gather of row group R, half H: M0 = transit + MOFF, global offset GOFF
#define ISF_DA_G(C, I, BIT, PTR, MOFF, GOFF)                                                                            \
  "s_bitcmp1_b32 %[fl], " #BIT "\n\t"                                                                                   \
  "s_cbranch_scc0 LS" C #I "_%=\n\t"                                                                                    \
  "s_add_u32 m0, %[tdst], " #MOFF "\n\t"                                                                                \
  "s_nop 0\n\t"                                                                                                         \
  "global_load_lds_dwordx4 " PTR ", off" GOFF "\n\t"                                                                    \
  "LS" C #I "_%=:\n\t"
#define ISF_DA_G0H(C) ISF_DA_G(C, 0, 0, "%[p0]", 0, "")
#define ISF_DA_G0L(C) ISF_DA_G(C, 1, 0, "%[p0]", 960, " offset:64")
#define ISF_DA_G1H(C) ISF_DA_G(C, 2, 1, "%[p1]", 2048, "")
#define ISF_DA_G1L(C) ISF_DA_G(C, 3, 1, "%[p1]", 3008, " offset:64")
// weight piece I (0 .. 3) of the next stage
#define ISF_DA_W(C, I, SETM0, GOFF)                                                                                     \
  "s_bitcmp1_b32 %[fl], 2\n\t"                                                                                          \
  "s_cbranch_scc0 LW" C #I "_%=\n\t"                                                                                    \
  SETM0                                                                                                                 \
  "global_load_lds_dwordx4 %[bsrc], off" GOFF "\n\t"                                                                    \
  "LW" C #I "_%=:\n\t"
#define ISF_DA_W0(C) ISF_DA_W(C, 0, "s_mov_b32 m0, %[bdst]\n\ts_nop 0\n\t", "")
#define ISF_DA_W1(C) ISF_DA_W(C, 1, "s_mov_b32 m0, %[bdst]\n\ts_nop 0\n\t", " offset:1024")
#define ISF_DA_W2(C) ISF_DA_W(C, 2, "s_mov_b32 m0, %[bdst]\n\ts_nop 0\n\t", " offset:2048")
#define ISF_DA_W3(C) ISF_DA_W(C, 3, "s_mov_b32 m0, %[bdst]\n\ts_nop 0\n\t", " offset:3072")

// MFMA triples of a pair (tiles TA, TB; buffer AH AL BH BL), both row groups: 4 x 3
#define ISF_DA_B0_I(TA, TB, AH, AL, BH, BL)                                                                               \
  ISF_TM_MF("%[c0" #TA "]", ISF_DA_A0L, AH) ISF_TM_MF("%[c1" #TA "]", ISF_DA_A1L, AH) ISF_TM_MF("%[c0" #TB "]", ISF_DA_A0L, BH)
#define ISF_DA_B1_I(TA, TB, AH, AL, BH, BL)                                                                               \
  ISF_TM_MF("%[c1" #TB "]", ISF_DA_A1L, BH) ISF_TM_MF("%[c0" #TA "]", ISF_DA_A0H, AL) ISF_TM_MF("%[c1" #TA "]", ISF_DA_A1H, AL)
#define ISF_DA_B2_I(TA, TB, AH, AL, BH, BL)                                                                               \
  ISF_TM_MF("%[c0" #TB "]", ISF_DA_A0H, BL) ISF_TM_MF("%[c1" #TB "]", ISF_DA_A1H, BL) ISF_TM_MF("%[c0" #TA "]", ISF_DA_A0H, AH)
#define ISF_DA_B3_I(TA, TB, AH, AL, BH, BL)                                                                               \
  ISF_TM_MF("%[c1" #TA "]", ISF_DA_A1H, AH) ISF_TM_MF("%[c0" #TB "]", ISF_DA_A0H, BH) ISF_TM_MF("%[c1" #TB "]", ISF_DA_A1H, BH)
// one row group R (fragments RH, RL): 2 x 3
#define ISF_DA_O0_I(R, RH, RL, TA, TB, AH, AL, BH, BL)                                                                    \
  ISF_TM_MF("%[c" #R #TA "]", RL, AH) ISF_TM_MF("%[c" #R #TB "]", RL, BH) ISF_TM_MF("%[c" #R #TA "]", RH, AL)
#define ISF_DA_O1_I(R, RH, RL, TA, TB, AH, AL, BH, BL)                                                                    \
  ISF_TM_MF("%[c" #R #TB "]", RH, BL) ISF_TM_MF("%[c" #R #TA "]", RH, AH) ISF_TM_MF("%[c" #R #TB "]", RH, BH)

// (a buffer is passed as ONE macro argument that expands to its four register names: forwarded through __VA_ARGS__)
#define ISF_DA_B0(TA, TB, ...) ISF_DA_B0_I(TA, TB, __VA_ARGS__)
#define ISF_DA_B1(TA, TB, ...) ISF_DA_B1_I(TA, TB, __VA_ARGS__)
#define ISF_DA_B2(TA, TB, ...) ISF_DA_B2_I(TA, TB, __VA_ARGS__)
#define ISF_DA_B3(TA, TB, ...) ISF_DA_B3_I(TA, TB, __VA_ARGS__)
#define ISF_DA_O0(R, RH, RL, TA, TB, ...) ISF_DA_O0_I(R, RH, RL, TA, TB, __VA_ARGS__)
#define ISF_DA_O1(R, RH, RL, TA, TB, ...) ISF_DA_O1_I(R, RH, RL, TA, TB, __VA_ARGS__)

#define ISF_DA_RDB(P, BUF) ISF_TM_READ(P, BUF)
#define ISF_DA_XB ISF_TM_XAH, ISF_TM_XAL, ISF_TM_XBH, ISF_TM_XBL
#define ISF_DA_YB ISF_TM_YAH, ISF_TM_YAL, ISF_TM_YBH, ISF_TM_YBL

// ---- both row groups multiply
#define ISF_DA_BOTH                                                                                                     \
  "ds_read_b128 " ISF_DA_A0H ", %[va]\n\t"                                                                              \
  "ds_read_b128 " ISF_DA_A0L ", %[va] offset:1024\n\t"                                                                  \
  "ds_read_b128 " ISF_DA_A1H ", %[va] offset:2048\n\t"                                                                  \
  "ds_read_b128 " ISF_DA_A1L ", %[va] offset:3072\n\t"                                                                  \
  ISF_TM_READ(0, ISF_TM_XAH, ISF_TM_XAL, ISF_TM_XBH, ISF_TM_XBL)                                                        \
  ISF_TM_READ(1, ISF_TM_YAH, ISF_TM_YAL, ISF_TM_YBH, ISF_TM_YBL)                                                        \
  "s_waitcnt lgkmcnt(4)\n\t"                                                                                            \
  ISF_DA_B0(0, 1, ISF_DA_XB) ISF_DA_G0H("b") ISF_DA_B1(0, 1, ISF_DA_XB) ISF_DA_G0L("b")                                 \
  ISF_DA_B2(0, 1, ISF_DA_XB) ISF_DA_G1H("b") ISF_DA_B3(0, 1, ISF_DA_XB) ISF_DA_G1L("b")                                 \
  ISF_TM_READ(2, ISF_TM_XAH, ISF_TM_XAL, ISF_TM_XBH, ISF_TM_XBL)                                                        \
  "s_waitcnt lgkmcnt(4)\n\t"                                                                                            \
  ISF_DA_B0(2, 3, ISF_DA_YB) ISF_DA_W0("b") ISF_DA_B1(2, 3, ISF_DA_YB) ISF_DA_W1("b")                                   \
  ISF_DA_B2(2, 3, ISF_DA_YB) ISF_DA_W2("b") ISF_DA_B3(2, 3, ISF_DA_YB) ISF_DA_W3("b")                                   \
  ISF_TM_READ(3, ISF_TM_YAH, ISF_TM_YAL, ISF_TM_YBH, ISF_TM_YBL)                                                        \
  "s_waitcnt lgkmcnt(4)\n\t"                                                                                            \
  ISF_DA_B0(4, 5, ISF_DA_XB) ISF_DA_B1(4, 5, ISF_DA_XB) ISF_DA_B2(4, 5, ISF_DA_XB) ISF_DA_B3(4, 5, ISF_DA_XB)           \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                            \
  ISF_DA_B0(6, 7, ISF_DA_YB) ISF_DA_B1(6, 7, ISF_DA_YB) ISF_DA_B2(6, 7, ISF_DA_YB) ISF_DA_B3(6, 7, ISF_DA_YB)

// ---- one row group R multiplies (C = case letter, RH / RL its fragments, AOFF the transit offset of its hi half)
#define ISF_DA_ONE(C, R, RH, RL, AOFFH, AOFFL)                                                                          \
  "ds_read_b128 " RH ", %[va] offset:" #AOFFH "\n\t"                                                                    \
  "ds_read_b128 " RL ", %[va] offset:" #AOFFL "\n\t"                                                                    \
  ISF_TM_READ(0, ISF_TM_XAH, ISF_TM_XAL, ISF_TM_XBH, ISF_TM_XBL)                                                        \
  ISF_TM_READ(1, ISF_TM_YAH, ISF_TM_YAL, ISF_TM_YBH, ISF_TM_YBL)                                                        \
  "s_waitcnt lgkmcnt(4)\n\t"                                                                                            \
  ISF_DA_O0(R, RH, RL, 0, 1, ISF_DA_XB) ISF_DA_G0H(C) ISF_DA_G0L(C) ISF_DA_O1(R, RH, RL, 0, 1, ISF_DA_XB)               \
  ISF_DA_G1H(C) ISF_DA_G1L(C)                                                                                           \
  ISF_TM_READ(2, ISF_TM_XAH, ISF_TM_XAL, ISF_TM_XBH, ISF_TM_XBL)                                                        \
  "s_waitcnt lgkmcnt(4)\n\t"                                                                                            \
  ISF_DA_O0(R, RH, RL, 2, 3, ISF_DA_YB) ISF_DA_W0(C) ISF_DA_W1(C) ISF_DA_O1(R, RH, RL, 2, 3, ISF_DA_YB)                 \
  ISF_DA_W2(C) ISF_DA_W3(C)                                                                                             \
  ISF_TM_READ(3, ISF_TM_YAH, ISF_TM_YAL, ISF_TM_YBH, ISF_TM_YBL)                                                        \
  "s_waitcnt lgkmcnt(4)\n\t"                                                                                            \
  ISF_DA_O0(R, RH, RL, 4, 5, ISF_DA_XB) ISF_DA_O1(R, RH, RL, 4, 5, ISF_DA_XB)                                           \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                            \
  ISF_DA_O0(R, RH, RL, 6, 7, ISF_DA_YB) ISF_DA_O1(R, RH, RL, 6, 7, ISF_DA_YB)

// ---- no row group of this wave multiplies (another wave of the workgroup has the tap): the issue alone
#define ISF_DA_NONE                                                                                                     \
  ISF_DA_G0H("n") ISF_DA_G0L("n") ISF_DA_G1H("n") ISF_DA_G1L("n") ISF_DA_W0("n") ISF_DA_W1("n") ISF_DA_W2("n")          \
  ISF_DA_W3("n")

#define ISF_DA_TEXT                                                                                                     \
  "s_mov_b32 %[m0s], m0\n\t"                                                                                            \
  "s_bitcmp1_b32 %[fl], 3\n\t"                                                                                          \
  "s_cbranch_scc0 LDA_N0_%=\n\t"                                                                                        \
  "s_bitcmp1_b32 %[fl], 4\n\t"                                                                                          \
  "s_cbranch_scc0 LDA_ONLY0_%=\n\t"                                                                                     \
  ISF_DA_BOTH                                                                                                           \
  "s_branch LDA_END_%=\n\t"                                                                                             \
  "LDA_ONLY0_%=:\n\t"                                                                                                   \
  ISF_DA_ONE("p", 0, ISF_DA_A0H, ISF_DA_A0L, 0, 1024)                                                                   \
  "s_branch LDA_END_%=\n\t"                                                                                             \
  "LDA_N0_%=:\n\t"                                                                                                      \
  "s_bitcmp1_b32 %[fl], 4\n\t"                                                                                          \
  "s_cbranch_scc0 LDA_NONE_%=\n\t"                                                                                      \
  ISF_DA_ONE("q", 1, ISF_DA_A1H, ISF_DA_A1L, 2048, 3072)                                                                \
  "s_branch LDA_END_%=\n\t"                                                                                             \
  "LDA_NONE_%=:\n\t"                                                                                                    \
  ISF_DA_NONE                                                                                                           \
  "LDA_END_%=:\n\t"                                                                                                     \
  "s_mov_b32 m0, %[m0s]\n\t"

#define ISF_DA_CLOBBERS                                                                                                 \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17",  \
      "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33",  \
      "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "scc", "memory"
