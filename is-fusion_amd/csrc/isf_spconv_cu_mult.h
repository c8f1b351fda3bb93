// isf_spconv_cu_mult.h -- the hand-scheduled MULTIPLY PHASE of spconv_cu_kernel (isf_spconv_cu.hip), gfx950 assembly.
//
// Why assembly (round 6).  A step of the one-workgroup-per-CU kernel multiplies, for every 16-row
// group of the unit that has a neighbour through the step's tap (bit j of the scalar mask m), the group's A fragments
// (LDS) with the wave's weight fragments (registers) into the group's accumulators.  Which groups are active is only
// known at run time, the accumulators of a group are fixed registers.  hipcc's answers to that shape were all slow:
//   * sixteen guarded blocks with a "next fragment -> current fragment" rename (round 4): the rename becomes register
//     copies behind an s_waitcnt lgkmcnt(0) two MFMAs after the prefetching ds_read -- every block waits out a full LDS
//     round trip (the multiply phase ran at half the matrix pipe's rate with no global traffic at all);
//   * a loop over the active groups with the accumulators picked by a switch: 241 spilled registers;
//   * fragment sets tied to the parity of the group: the accumulators get renamed per block (copies on the skip path)
//     and 102 registers spill.
// Here the register file is allocated by hand: accumulators of group j, column tile nt = v[128 + 8 j + 4 nt .. + 3]
// (pinned through the asm constraints, so the C++ around the block keeps them there), two fragment sets X / Y used by
// consecutive ACTIVE groups alternately, and every block exists twice (reading X, reading Y).  A block
//   1. waits for its own set (read one block ago),
//   2. reads the NEXT active group's fragments into the other set (address computed one block ago),
//   3. issues its 6 MFMAs with the scalar search for the next-but-one active group and the address VALU in their shadow,
//   4. branches to the next active group's block of the other set (bit test + branch; inactive groups cost two scalar
//      instructions).
// Products and their order per accumulator are those of spconv_f16x3_kernel: a_lo b_hi -> a_hi b_lo -> a_hi b_hi.
// Hazards: ds_read -> MFMA by lgkmcnt; an MFMA's SrcC is the destination of the MFMA two instructions earlier (exact
// overlap, interlocked by the hardware); the caller issues 32 wait states before anything else reads the accumulators.
#pragma once

// accumulator registers of (group J, column tile NT)
#define ISF_CUM_ACC(J, NT) "v[128+8*" #J "+4*" #NT ":131+8*" #J "+4*" #NT "]"
#define ISF_CUM_MFMA(J, NT, A, B) "v_mfma_f32_16x16x32_f16 " ISF_CUM_ACC(J, NT) ", " A ", " B ", " ISF_CUM_ACC(J, NT) "\n\t"

// block of group J (J1 = J + 1, J2 = J + 2) reading set (MH, ML) and prefetching into (OH, OL); ME / OT = label letters
#define ISF_CUM_BLK(J, J1, J2, ME, OT, MH, ML, OH, OL)                                                                  \
  "LB" ME #J "_%=:\n\t"                                                                                                 \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                            \
  "ds_read_b128 " OH ", %[va]\n\t"                                                                                      \
  "ds_read_b128 " OL ", %[va] offset:1024\n\t"                                                                          \
  ISF_CUM_MFMA(J, 0, ML, "%[b0h]")                                                                                      \
  "s_lshr_b32 %[t], %[m], " #J1 "\n\t"                                                                                  \
  "s_add_i32 %[t2], %[t], -1\n\t"                                                                                       \
  ISF_CUM_MFMA(J, 1, ML, "%[b1h]")                                                                                      \
  "s_and_b32 %[t], %[t], %[t2]\n\t"                                                                                     \
  "s_ff1_i32_b32 %[t2], %[t]\n\t"                                                                                       \
  ISF_CUM_MFMA(J, 0, MH, "%[b0l]")                                                                                      \
  "s_max_i32 %[t2], %[t2], 0\n\t"                                                                                       \
  "s_lshl_b32 %[t2], %[t2], 11\n\t"                                                                                     \
  ISF_CUM_MFMA(J, 1, MH, "%[b1l]")                                                                                      \
  "s_add_i32 %[t2], %[t2], 2048*" #J1 "\n\t"                                                                            \
  "v_add_u32 %[va], %[t2], %[vb]\n\t"                                                                                   \
  ISF_CUM_MFMA(J, 0, MH, "%[b0h]")                                                                                      \
  "s_bitcmp1_b32 %[m], " #J1 "\n\t"                                                                                     \
  ISF_CUM_MFMA(J, 1, MH, "%[b1h]")                                                                                      \
  "s_cbranch_scc1 LB" OT #J1 "_%=\n\t"                                                                                  \
  "s_branch LT" OT #J2 "_%=\n\t"

// group 15: nothing can follow
#define ISF_CUM_BLK_LAST(ME, MH, ML)                                                                                    \
  "LB" ME "15_%=:\n\t"                                                                                                  \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                            \
  ISF_CUM_MFMA(15, 0, ML, "%[b0h]") ISF_CUM_MFMA(15, 1, ML, "%[b1h]") ISF_CUM_MFMA(15, 0, MH, "%[b0l]")                 \
  ISF_CUM_MFMA(15, 1, MH, "%[b1l]") ISF_CUM_MFMA(15, 0, MH, "%[b0h]") ISF_CUM_MFMA(15, 1, MH, "%[b1h]")                 \
  "s_branch LEND_%=\n\t"

#define ISF_CUM_TEST(J, ME) "LT" ME #J "_%=:\n\ts_bitcmp1_b32 %[m], " #J "\n\ts_cbranch_scc1 LB" ME #J "_%=\n\t"

#define ISF_CUM_SET(ME, OT, MH, ML, OH, OL)                                                                             \
  ISF_CUM_TEST(0, ME) ISF_CUM_TEST(1, ME) ISF_CUM_TEST(2, ME) ISF_CUM_TEST(3, ME) ISF_CUM_TEST(4, ME)                   \
  ISF_CUM_TEST(5, ME) ISF_CUM_TEST(6, ME) ISF_CUM_TEST(7, ME) ISF_CUM_TEST(8, ME) ISF_CUM_TEST(9, ME)                   \
  ISF_CUM_TEST(10, ME) ISF_CUM_TEST(11, ME) ISF_CUM_TEST(12, ME) ISF_CUM_TEST(13, ME) ISF_CUM_TEST(14, ME)              \
  ISF_CUM_TEST(15, ME)                                                                                                  \
  "LT" ME "16_%=:\n\t"                                                                                                  \
  "s_branch LEND_%=\n\t"                                                                                                \
  ISF_CUM_BLK(0, 1, 2, ME, OT, MH, ML, OH, OL) ISF_CUM_BLK(1, 2, 3, ME, OT, MH, ML, OH, OL)                             \
  ISF_CUM_BLK(2, 3, 4, ME, OT, MH, ML, OH, OL) ISF_CUM_BLK(3, 4, 5, ME, OT, MH, ML, OH, OL)                             \
  ISF_CUM_BLK(4, 5, 6, ME, OT, MH, ML, OH, OL) ISF_CUM_BLK(5, 6, 7, ME, OT, MH, ML, OH, OL)                             \
  ISF_CUM_BLK(6, 7, 8, ME, OT, MH, ML, OH, OL) ISF_CUM_BLK(7, 8, 9, ME, OT, MH, ML, OH, OL)                             \
  ISF_CUM_BLK(8, 9, 10, ME, OT, MH, ML, OH, OL) ISF_CUM_BLK(9, 10, 11, ME, OT, MH, ML, OH, OL)                          \
  ISF_CUM_BLK(10, 11, 12, ME, OT, MH, ML, OH, OL) ISF_CUM_BLK(11, 12, 13, ME, OT, MH, ML, OH, OL)                       \
  ISF_CUM_BLK(12, 13, 14, ME, OT, MH, ML, OH, OL) ISF_CUM_BLK(13, 14, 15, ME, OT, MH, ML, OH, OL)                       \
  ISF_CUM_BLK(14, 15, 16, ME, OT, MH, ML, OH, OL) ISF_CUM_BLK_LAST(ME, MH, ML)

// the whole phase: m != 0.  Entry: first active group -> set X, address of the second active group -> va, then the X chain.
#define ISF_CUM_TEXT                                                                                                    \
  "s_ff1_i32_b32 %[t2], %[m]\n\t"                                                                                       \
  "s_lshl_b32 %[t2], %[t2], 11\n\t"                                                                                     \
  "v_add_u32 %[va], %[t2], %[vb]\n\t"                                                                                   \
  "s_add_i32 %[t], %[m], -1\n\t"                                                                                        \
  "ds_read_b128 %[xh], %[va]\n\t"                                                                                       \
  "ds_read_b128 %[xl], %[va] offset:1024\n\t"                                                                           \
  "s_and_b32 %[t], %[t], %[m]\n\t"                                                                                      \
  "s_ff1_i32_b32 %[t2], %[t]\n\t"                                                                                       \
  "s_max_i32 %[t2], %[t2], 0\n\t"                                                                                       \
  "s_lshl_b32 %[t2], %[t2], 11\n\t"                                                                                     \
  "v_add_u32 %[va], %[t2], %[vb]\n\t"                                                                                   \
  ISF_CUM_SET("X", "Y", "%[xh]", "%[xl]", "%[yh]", "%[yl]")                                                             \
  ISF_CUM_SET("Y", "X", "%[yh]", "%[yl]", "%[xh]", "%[xl]")                                                             \
  "LEND_%=:\n\t"

// ---------------------------------------------------------------------------------------------------------------------
// The same phase for the TWO-WORKGROUPS-PER-CU shape: a unit has <= 8 groups, a wave owns 4 column tiles (64 columns) of
// every group: accumulators of (group j, tile nt) = v[128 + 16 j + 4 nt .. + 3], 12 MFMAs per block (per accumulator still
// a_lo b_hi -> a_hi b_lo -> a_hi b_hi; consecutive MFMAs on one accumulator are four instructions apart).
#define ISF_CUM4_ACC(J, NT) "v[128+16*" #J "+4*" #NT ":131+16*" #J "+4*" #NT "]"
#define ISF_CUM4_MFMA(J, NT, A, B) "v_mfma_f32_16x16x32_f16 " ISF_CUM4_ACC(J, NT) ", " A ", " B ", " ISF_CUM4_ACC(J, NT) "\n\t"

#define ISF_CUM4_BLK(J, J1, J2, ME, OT, MH, ML, OH, OL)                                                                 \
  "LB" ME #J "_%=:\n\t"                                                                                                 \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                            \
  "ds_read_b128 " OH ", %[va]\n\t"                                                                                      \
  "ds_read_b128 " OL ", %[va] offset:1024\n\t"                                                                          \
  ISF_CUM4_MFMA(J, 0, ML, "%[b0h]")                                                                                     \
  "s_lshr_b32 %[t], %[m], " #J1 "\n\t"                                                                                  \
  ISF_CUM4_MFMA(J, 1, ML, "%[b1h]")                                                                                     \
  "s_add_i32 %[t2], %[t], -1\n\t"                                                                                       \
  ISF_CUM4_MFMA(J, 2, ML, "%[b2h]")                                                                                     \
  "s_and_b32 %[t], %[t], %[t2]\n\t"                                                                                     \
  ISF_CUM4_MFMA(J, 3, ML, "%[b3h]")                                                                                     \
  "s_ff1_i32_b32 %[t2], %[t]\n\t"                                                                                       \
  ISF_CUM4_MFMA(J, 0, MH, "%[b0l]")                                                                                     \
  "s_max_i32 %[t2], %[t2], 0\n\t"                                                                                       \
  ISF_CUM4_MFMA(J, 1, MH, "%[b1l]")                                                                                     \
  "s_lshl_b32 %[t2], %[t2], 11\n\t"                                                                                     \
  ISF_CUM4_MFMA(J, 2, MH, "%[b2l]")                                                                                     \
  "s_add_i32 %[t2], %[t2], 2048*" #J1 "\n\t"                                                                            \
  ISF_CUM4_MFMA(J, 3, MH, "%[b3l]")                                                                                     \
  "v_add_u32 %[va], %[t2], %[vb]\n\t"                                                                                   \
  ISF_CUM4_MFMA(J, 0, MH, "%[b0h]")                                                                                     \
  ISF_CUM4_MFMA(J, 1, MH, "%[b1h]")                                                                                     \
  ISF_CUM4_MFMA(J, 2, MH, "%[b2h]")                                                                                     \
  "s_bitcmp1_b32 %[m], " #J1 "\n\t"                                                                                     \
  ISF_CUM4_MFMA(J, 3, MH, "%[b3h]")                                                                                     \
  "s_cbranch_scc1 LB" OT #J1 "_%=\n\t"                                                                                  \
  "s_branch LT" OT #J2 "_%=\n\t"

#define ISF_CUM4_BLK_LAST(ME, MH, ML)                                                                                   \
  "LB" ME "7_%=:\n\t"                                                                                                   \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                            \
  ISF_CUM4_MFMA(7, 0, ML, "%[b0h]") ISF_CUM4_MFMA(7, 1, ML, "%[b1h]") ISF_CUM4_MFMA(7, 2, ML, "%[b2h]")                 \
  ISF_CUM4_MFMA(7, 3, ML, "%[b3h]") ISF_CUM4_MFMA(7, 0, MH, "%[b0l]") ISF_CUM4_MFMA(7, 1, MH, "%[b1l]")                 \
  ISF_CUM4_MFMA(7, 2, MH, "%[b2l]") ISF_CUM4_MFMA(7, 3, MH, "%[b3l]") ISF_CUM4_MFMA(7, 0, MH, "%[b0h]")                 \
  ISF_CUM4_MFMA(7, 1, MH, "%[b1h]") ISF_CUM4_MFMA(7, 2, MH, "%[b2h]") ISF_CUM4_MFMA(7, 3, MH, "%[b3h]")                 \
  "s_branch LEND_%=\n\t"

#define ISF_CUM4_SET(ME, OT, MH, ML, OH, OL)                                                                            \
  ISF_CUM_TEST(0, ME) ISF_CUM_TEST(1, ME) ISF_CUM_TEST(2, ME) ISF_CUM_TEST(3, ME) ISF_CUM_TEST(4, ME)                   \
  ISF_CUM_TEST(5, ME) ISF_CUM_TEST(6, ME) ISF_CUM_TEST(7, ME)                                                           \
  "LT" ME "8_%=:\n\t"                                                                                                   \
  "s_branch LEND_%=\n\t"                                                                                                \
  ISF_CUM4_BLK(0, 1, 2, ME, OT, MH, ML, OH, OL) ISF_CUM4_BLK(1, 2, 3, ME, OT, MH, ML, OH, OL)                           \
  ISF_CUM4_BLK(2, 3, 4, ME, OT, MH, ML, OH, OL) ISF_CUM4_BLK(3, 4, 5, ME, OT, MH, ML, OH, OL)                           \
  ISF_CUM4_BLK(4, 5, 6, ME, OT, MH, ML, OH, OL) ISF_CUM4_BLK(5, 6, 7, ME, OT, MH, ML, OH, OL)                           \
  ISF_CUM4_BLK(6, 7, 8, ME, OT, MH, ML, OH, OL) ISF_CUM4_BLK_LAST(ME, MH, ML)

#define ISF_CUM4_TEXT                                                                                                   \
  "s_ff1_i32_b32 %[t2], %[m]\n\t"                                                                                       \
  "s_lshl_b32 %[t2], %[t2], 11\n\t"                                                                                     \
  "v_add_u32 %[va], %[t2], %[vb]\n\t"                                                                                   \
  "s_add_i32 %[t], %[m], -1\n\t"                                                                                        \
  "ds_read_b128 %[xh], %[va]\n\t"                                                                                       \
  "ds_read_b128 %[xl], %[va] offset:1024\n\t"                                                                           \
  "s_and_b32 %[t], %[t], %[m]\n\t"                                                                                      \
  "s_ff1_i32_b32 %[t2], %[t]\n\t"                                                                                       \
  "s_max_i32 %[t2], %[t2], 0\n\t"                                                                                       \
  "s_lshl_b32 %[t2], %[t2], 11\n\t"                                                                                     \
  "v_add_u32 %[va], %[t2], %[vb]\n\t"                                                                                   \
  ISF_CUM4_SET("X", "Y", "%[xh]", "%[xl]", "%[yh]", "%[yl]")                                                            \
  ISF_CUM4_SET("Y", "X", "%[yh]", "%[yl]", "%[xh]", "%[xl]")                                                            \
  "LEND_%=:\n\t"
