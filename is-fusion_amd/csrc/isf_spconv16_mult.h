// isf_spconv16_mult.h -- the hand-scheduled MULTIPLY SECTION of spconv_f16x3_kernel for the wave shape of every deep
// layer (RG = 2 row groups x NT = 8 column tiles, one 32-channel chunk per step, f16x3), gfx950 assembly.
//
// Why (round 6).  hipcc compiles the section -- "for each column tile: for each row group that has
// the tap: three MFMAs on its accumulator" -- into sixteen basic blocks behind vector-condition branches, each holding the
// three MFMAs of ONE accumulator back to back: every MFMA waits for the result of the one in front of it, the B-fragment
// reads sit in the same blocks behind s_waitcnt lgkmcnt, and nothing is interleaved across blocks (the ISA of round 5's
// kernel: v_mfma ... v[78:81] three times in a row, s_cbranch_vccz in between).  The phase trace put the section at
// 1 240 - 1 430 cycles for 384 - 768 cycles of matrix pipe.
// Here a step is ONE straight-line stream per case (both row groups have the tap / only the first / only the second):
//   * column tiles are taken in PAIRS; the pair's four accumulators (two with one row group) are interleaved so that
//     consecutive MFMAs on one accumulator are four (two) instructions apart;
//   * the B fragments of pair p + 2 are read from LDS into the registers of pair p as soon as its last MFMA is issued --
//     two pairs in flight, a pair (12 MFMAs, 192 cycles) of cover for every read; the fragments live in a[0:31] (the
//     accumulation registers are free in this kernel: its own accumulators are allocated as VGPRs by the compiler);
//   * per accumulator the products keep the order a_lo b_hi -> a_hi b_lo -> a_hi b_hi: results are bit-identical.
// Operands: %[cRN] accumulators (row group R, column tile N), %[aRh] / %[aRl] the gathered row fragments, %[vb] this lane's
// byte address in the step's weight buffer (column tile nt: hi at nt * 2048, lo at nt * 2048 + 1024), %[n0] / %[n1] != 0:
// the row group has the tap (at least one of them does).
#pragma once

#define ISF_TM_MF(C, A, B) "v_mfma_f32_16x16x32_f16 " C ", " A ", " B ", " C "\n\t"
// B fragment registers of buffer X (pair's first / second tile: hi, lo) and buffer Y
#define ISF_TM_XAH "a[0:3]"
#define ISF_TM_XAL "a[4:7]"
#define ISF_TM_XBH "a[8:11]"
#define ISF_TM_XBL "a[12:15]"
#define ISF_TM_YAH "a[16:19]"
#define ISF_TM_YAL "a[20:23]"
#define ISF_TM_YBH "a[24:27]"
#define ISF_TM_YBL "a[28:31]"
// the four fragment reads of column-tile pair P (tiles 2P, 2P + 1) into buffer (AH, AL, BH, BL)
#define ISF_TM_READ(P, AH, AL, BH, BL)                                                                                  \
  "ds_read_b128 " AH ", %[vb] offset:4096*" #P "\n\t"                                                                   \
  "ds_read_b128 " BH ", %[vb] offset:4096*" #P "+2048\n\t"                                                              \
  "ds_read_b128 " AL ", %[vb] offset:4096*" #P "+1024\n\t"                                                              \
  "ds_read_b128 " BL ", %[vb] offset:4096*" #P "+3072\n\t"
// both row groups, tiles TA / TB from buffer (AH, AL, BH, BL): 12 MFMAs, four accumulators in rotation
#define ISF_TM_BOTH(TA, TB, AH, AL, BH, BL)                                                                             \
  ISF_TM_MF("%[c0" #TA "]", "%[a0l]", AH) ISF_TM_MF("%[c1" #TA "]", "%[a1l]", AH)                                       \
  ISF_TM_MF("%[c0" #TB "]", "%[a0l]", BH) ISF_TM_MF("%[c1" #TB "]", "%[a1l]", BH)                                       \
  ISF_TM_MF("%[c0" #TA "]", "%[a0h]", AL) ISF_TM_MF("%[c1" #TA "]", "%[a1h]", AL)                                       \
  ISF_TM_MF("%[c0" #TB "]", "%[a0h]", BL) ISF_TM_MF("%[c1" #TB "]", "%[a1h]", BL)                                       \
  ISF_TM_MF("%[c0" #TA "]", "%[a0h]", AH) ISF_TM_MF("%[c1" #TA "]", "%[a1h]", AH)                                       \
  ISF_TM_MF("%[c0" #TB "]", "%[a0h]", BH) ISF_TM_MF("%[c1" #TB "]", "%[a1h]", BH)
// one row group R: 6 MFMAs, two accumulators in rotation
#define ISF_TM_ONE(R, TA, TB, AH, AL, BH, BL)                                                                           \
  ISF_TM_MF("%[c" #R #TA "]", "%[a" #R "l]", AH) ISF_TM_MF("%[c" #R #TB "]", "%[a" #R "l]", BH)                         \
  ISF_TM_MF("%[c" #R #TA "]", "%[a" #R "h]", AL) ISF_TM_MF("%[c" #R #TB "]", "%[a" #R "h]", BL)                         \
  ISF_TM_MF("%[c" #R #TA "]", "%[a" #R "h]", AH) ISF_TM_MF("%[c" #R #TB "]", "%[a" #R "h]", BH)

#define ISF_TM_X ISF_TM_XAH, ISF_TM_XAL, ISF_TM_XBH, ISF_TM_XBL
#define ISF_TM_Y ISF_TM_YAH, ISF_TM_YAL, ISF_TM_YBH, ISF_TM_YBL
// the pipeline over the four pairs; BODY(TA, TB, buffer) = the MFMAs of one pair
#define ISF_TM_PIPE(BODY)                                                                                               \
  ISF_TM_READ(0, ISF_TM_XAH, ISF_TM_XAL, ISF_TM_XBH, ISF_TM_XBL)                                                        \
  ISF_TM_READ(1, ISF_TM_YAH, ISF_TM_YAL, ISF_TM_YBH, ISF_TM_YBL)                                                        \
  "s_waitcnt lgkmcnt(4)\n\t"                                                                                            \
  BODY(0, 1, ISF_TM_X)                                                                                                  \
  ISF_TM_READ(2, ISF_TM_XAH, ISF_TM_XAL, ISF_TM_XBH, ISF_TM_XBL)                                                        \
  "s_waitcnt lgkmcnt(4)\n\t"                                                                                            \
  BODY(2, 3, ISF_TM_Y)                                                                                                  \
  ISF_TM_READ(3, ISF_TM_YAH, ISF_TM_YAL, ISF_TM_YBH, ISF_TM_YBL)                                                        \
  "s_waitcnt lgkmcnt(4)\n\t"                                                                                            \
  BODY(4, 5, ISF_TM_X)                                                                                                  \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                            \
  BODY(6, 7, ISF_TM_Y)

#define ISF_TM_BODY_BOTH(TA, TB, ...) ISF_TM_BOTH(TA, TB, __VA_ARGS__)
#define ISF_TM_BODY_ONE0(TA, TB, ...) ISF_TM_ONE(0, TA, TB, __VA_ARGS__)
#define ISF_TM_BODY_ONE1(TA, TB, ...) ISF_TM_ONE(1, TA, TB, __VA_ARGS__)

#define ISF_TM_TEXT                                                                                                     \
  "s_cmp_eq_u32 %[n0], 0\n\t"                                                                                           \
  "s_cbranch_scc1 LTM1_%=\n\t"                                                                                          \
  "s_cmp_eq_u32 %[n1], 0\n\t"                                                                                           \
  "s_cbranch_scc1 LTM0_%=\n\t"                                                                                          \
  ISF_TM_PIPE(ISF_TM_BODY_BOTH)                                                                                         \
  "s_branch LTME_%=\n\t"                                                                                                \
  "LTM0_%=:\n\t"                                                                                                        \
  ISF_TM_PIPE(ISF_TM_BODY_ONE0)                                                                                         \
  "s_branch LTME_%=\n\t"                                                                                                \
  "LTM1_%=:\n\t"                                                                                                        \
  ISF_TM_PIPE(ISF_TM_BODY_ONE1)                                                                                         \
  "LTME_%=:\n\t"

#define ISF_TM_CLOBBERS                                                                                                 \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17",  \
      "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "scc", "memory"

// one row group per wave (RG = 1: the small-launch shapes): the pair pipeline with the single-group body, no case split
#define ISF_TM_TEXT_RG1 ISF_TM_PIPE(ISF_TM_BODY_ONE0)

// ---------------------------------------------------------------------------------------------------------------------
// The 8-wave shape of the 128-column layers (two workgroups = 16 waves per CU: 128 registers per wave) has room for half
// the fragment buffers: a[0:15], ONE column tile per buffer.  Both row groups: 6 MFMAs per tile, two accumulators in
// rotation, the tile after next read as soon as a tile's MFMAs are issued.  One row group: tiles in pairs across the two
// buffers (two accumulators in rotation), both buffers refilled behind the pair.
#define ISF_TN_XH "a[0:3]"
#define ISF_TN_XL "a[4:7]"
#define ISF_TN_YH "a[8:11]"
#define ISF_TN_YL "a[12:15]"
#define ISF_TN_READ(T, H, L)                                                                                            \
  "ds_read_b128 " H ", %[vb] offset:2048*" #T "\n\t"                                                                    \
  "ds_read_b128 " L ", %[vb] offset:2048*" #T "+1024\n\t"
#define ISF_TN_BOTH(T, H, L)                                                                                            \
  ISF_TM_MF("%[c0" #T "]", "%[a0l]", H) ISF_TM_MF("%[c1" #T "]", "%[a1l]", H)                                           \
  ISF_TM_MF("%[c0" #T "]", "%[a0h]", L) ISF_TM_MF("%[c1" #T "]", "%[a1h]", L)                                           \
  ISF_TM_MF("%[c0" #T "]", "%[a0h]", H) ISF_TM_MF("%[c1" #T "]", "%[a1h]", H)
#define ISF_TN_PAIR1(R, TA, TB)                                                                                         \
  ISF_TM_MF("%[c" #R #TA "]", "%[a" #R "l]", ISF_TN_XH) ISF_TM_MF("%[c" #R #TB "]", "%[a" #R "l]", ISF_TN_YH)           \
  ISF_TM_MF("%[c" #R #TA "]", "%[a" #R "h]", ISF_TN_XL) ISF_TM_MF("%[c" #R #TB "]", "%[a" #R "h]", ISF_TN_YL)           \
  ISF_TM_MF("%[c" #R #TA "]", "%[a" #R "h]", ISF_TN_XH) ISF_TM_MF("%[c" #R #TB "]", "%[a" #R "h]", ISF_TN_YH)
#define ISF_TN_PIPE_BOTH                                                                                                \
  ISF_TN_READ(0, ISF_TN_XH, ISF_TN_XL) ISF_TN_READ(1, ISF_TN_YH, ISF_TN_YL)                                             \
  "s_waitcnt lgkmcnt(2)\n\t" ISF_TN_BOTH(0, ISF_TN_XH, ISF_TN_XL) ISF_TN_READ(2, ISF_TN_XH, ISF_TN_XL)                  \
  "s_waitcnt lgkmcnt(2)\n\t" ISF_TN_BOTH(1, ISF_TN_YH, ISF_TN_YL) ISF_TN_READ(3, ISF_TN_YH, ISF_TN_YL)                  \
  "s_waitcnt lgkmcnt(2)\n\t" ISF_TN_BOTH(2, ISF_TN_XH, ISF_TN_XL) ISF_TN_READ(4, ISF_TN_XH, ISF_TN_XL)                  \
  "s_waitcnt lgkmcnt(2)\n\t" ISF_TN_BOTH(3, ISF_TN_YH, ISF_TN_YL) ISF_TN_READ(5, ISF_TN_YH, ISF_TN_YL)                  \
  "s_waitcnt lgkmcnt(2)\n\t" ISF_TN_BOTH(4, ISF_TN_XH, ISF_TN_XL) ISF_TN_READ(6, ISF_TN_XH, ISF_TN_XL)                  \
  "s_waitcnt lgkmcnt(2)\n\t" ISF_TN_BOTH(5, ISF_TN_YH, ISF_TN_YL) ISF_TN_READ(7, ISF_TN_YH, ISF_TN_YL)                  \
  "s_waitcnt lgkmcnt(2)\n\t" ISF_TN_BOTH(6, ISF_TN_XH, ISF_TN_XL)                                                       \
  "s_waitcnt lgkmcnt(0)\n\t" ISF_TN_BOTH(7, ISF_TN_YH, ISF_TN_YL)
#define ISF_TN_PIPE_ONE(R)                                                                                              \
  ISF_TN_READ(0, ISF_TN_XH, ISF_TN_XL) ISF_TN_READ(1, ISF_TN_YH, ISF_TN_YL)                                             \
  "s_waitcnt lgkmcnt(0)\n\t" ISF_TN_PAIR1(R, 0, 1)                                                                      \
  ISF_TN_READ(2, ISF_TN_XH, ISF_TN_XL) ISF_TN_READ(3, ISF_TN_YH, ISF_TN_YL)                                             \
  "s_waitcnt lgkmcnt(0)\n\t" ISF_TN_PAIR1(R, 2, 3)                                                                      \
  ISF_TN_READ(4, ISF_TN_XH, ISF_TN_XL) ISF_TN_READ(5, ISF_TN_YH, ISF_TN_YL)                                             \
  "s_waitcnt lgkmcnt(0)\n\t" ISF_TN_PAIR1(R, 4, 5)                                                                      \
  ISF_TN_READ(6, ISF_TN_XH, ISF_TN_XL) ISF_TN_READ(7, ISF_TN_YH, ISF_TN_YL)                                             \
  "s_waitcnt lgkmcnt(0)\n\t" ISF_TN_PAIR1(R, 6, 7)
#define ISF_TN_TEXT                                                                                                     \
  "s_cmp_eq_u32 %[n0], 0\n\t"                                                                                           \
  "s_cbranch_scc1 LTN1_%=\n\t"                                                                                          \
  "s_cmp_eq_u32 %[n1], 0\n\t"                                                                                           \
  "s_cbranch_scc1 LTN0_%=\n\t"                                                                                          \
  ISF_TN_PIPE_BOTH                                                                                                      \
  "s_branch LTNE_%=\n\t"                                                                                                \
  "LTN0_%=:\n\t"                                                                                                        \
  ISF_TN_PIPE_ONE(0)                                                                                                    \
  "s_branch LTNE_%=\n\t"                                                                                                \
  "LTN1_%=:\n\t"                                                                                                        \
  ISF_TN_PIPE_ONE(1)                                                                                                    \
  "LTNE_%=:\n\t"
#define ISF_TN_CLOBBERS                                                                                                 \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "scc", "memory"
