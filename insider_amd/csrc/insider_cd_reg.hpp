// insider_cd_reg.hpp — register-resident elastic-net coordinate descent for K <= 32: four genes per wavefront,
// the Gram matrix in VGPRs, no LDS; and for 32 < K <= 48 with a third coordinate slot whose Gram columns live in LDS
// (REG_DEFINE_SWEEP3 below).
//
// strong_coordinate_descent's sweep loop (src/coordinate_descent.cpp:86-114) is a K-step sequential recurrence per
// gene that runs for hundreds to thousands of sweeps; it is bound by vector-instruction issue.  Layout: gene g owns
// the 16-lane DPP row g of the wave; lane i of the row owns COORDINATES i and 16 + i for the whole solve (h, beta,
// 1/(XtX_kk + l2)) and holds columns i and 16 + i of the gene's Gram matrix in registers (2 x KMAX doubles, zero
// diagonal).  All genes follow the same coordinate order (include/insider_perm.h), so the coordinate k of a step is
// wave-uniform: the step is dispatched through a computed jump (s_setpc_b64) into a table of 96-byte code blocks,
// one per coordinate, whose byte offsets the order table holds as a successor list (34 dwords loaded into SGPRs at
// the start of a sweep; block k reads its successor from its own register).  Inside block k everything is static — the Gram operand is the register holding G[u][k], the lane that
// owns k is k % 16, and its (negated) increment reaches the row as the DPP row_newbcast:k%16 source operand of a
// 64-bit v_fmac_f64 (gfx90a+ "DP ALU DPP").  Per step and wave (4 genes): 6 vector + 4 scalar instructions (round 3: the
// soft threshold is the hardware's output clamp on a scaled gradient, RegState below), no memory or LDS access; the sweep is one inline-asm block (the compiler turns a C++ switch into a compare tree with
// register copies at the merge).  Without LDS the occupancy is set by registers alone (REG_WAVES(KMAX) waves per
// SIMD), where the LDS-resident variant (insider_cd_row16.hpp) held 5 waves per CU at K = 30.
// Measured on MI355X (tools/ubench3.hip): every VALU instruction of the block, including the DPP fmac and v_mov_b64,
// issues at ~2.15 ns per SIMD; a computed jump costs ~11 ns of single-wave latency.
#pragma once

namespace insider {

constexpr int REG_ORDER_OFF = 128;   // byte offset of the block-offset dwords inside an order-table row
constexpr int REG3_ORDER_OFF = 124;  // ... for 32 < K <= 48 (three slots): one dword earlier, so that 1 + 48 dwords fit the row
static_assert(ORDER_ROW == INSIDER_ORDER_ROW && REG_ORDER_OFF + 8 * 33 <= ORDER_ROW, "a row holds 1 + 32 successor pairs behind the order bytes");
constexpr int REG_BLOCK = INSIDER_REG_BLOCK;   // bytes between the code blocks of consecutive coordinates (see REG_BLOCK_HEAD)
#define REG_STR_(x) #x
#define REG_STR(x) REG_STR_(x)

// The sweep state in the SCALED form the step works on (round 3): with la = lambda alpha > 0
//   y   = h / (2 la) + 1/2      h = Xty - offdiag(XtX) beta, the gradient part of src/coordinate_descent.cpp:94
//   tau = 2 la / (XtX_kk + l2)  (0 for a screened-out coordinate or a parked gene)
// and the Gram registers hold G / (2 la).  Then  soft(h, la) = h - clamp(h, -la, la) = 2 la (y - clamp01(y)), and the
// hardware's output clamp to [0, 1] does the soft threshold in ONE instruction where min + max took two:
//   c = clamp01(y);  e = y - c;  dn = beta - e tau  (minus the increment);  beta -= dn;  y_u += bcast(dn) Ghat_u[k]
// — 4 + 2 vector instructions per step instead of 5 + 2, and a dependent chain of 3 before the DPP fmacs instead of 4.
// Exact zeros survive: |h| <= la  <=>  0 <= y <= 1  =>  c = y, e = 0, dn = beta, beta - dn = 0 exactly.  The offset costs
// no accuracy that matters: y is rounded at 1.1e-16 absolute, i.e. h at 2.2e-16 la.
template <int SLOTS>
struct RegState {
    double y[SLOTS], beta[SLOTS], tau[SLOTS];
};

// ---- the sweep as one asm block -----------------------------------------------------------------------------------
// Block for coordinate KK (slot s = KK / 16, owner lane it = KK % 16), src/coordinate_descent.cpp:91-110 in covariance
// form on the scaled state (above); screened-out coordinates and parked genes carry tau = beta = 0, i.e. a zero increment.
// The sweep's order arrives as a successor list of code-block offsets (k_order_table: dword 0 = first block, dword 1 + k
// = the block visited after coordinate k, the exit block after the last), loaded into s[64:97] at the start of the sweep
// (offsets: each block adds the table's base address, s98, itself); the table of blocks (REG_BLOCK bytes apart, placed with .org,
// which also asserts that no block outgrows its slot) starts on a 4 KiB boundary and is shorter than 4 KiB, so it cannot
// straddle a 4 GiB boundary and the high word of every block address is the same (vcc_hi, set once).  Block k forms its
// successor's address from its own table register s[65 + k] in vcc_lo while the vector chain runs, and jumps: no
// position counter, M0 untouched.  Critical chain per step: clamp, sub, fma (dn), DPP fmac.  Hazards respected by
// construction: >= 2 instructions between the write of dn and its DPP read; exec is written by SALU only; nothing in a
// block writes vcc_hi.
// The four scalar-like instructions of a step do useful work on the owner lane of each row only: they run under a four-lane
// exec mask (the DPP fmacs that follow need every lane).  Same instruction count as narrowing the mask around the beta
// update alone, but the part sustains a higher clock (round 3, tools/ab_variants.sh: +4 % sweep rate, bit-identical).
// The address add sits between the clamp and the subtraction that depends on it (round 4: in the shadow of the clamp's
// result latency instead of in front of the chain, sweep kernel 2.08 -> 2.00 ms per launch at c3, tools/exp_probe.sh).
// Measured and NOT adopted in round 4 (same tool): the head on all lanes with the beta update as a bank-masked
// DPP fmac on a lane indicator (no exec writes: slower, 2.17-2.24 ms — four more full-width fp64 instructions per step
// cost more clock than the two exec writes cost issue slots); wave priorities by hardware wave id (slower); the exec
// narrowing after the clamp (slower); other positions of the add (equal).
#define REG_BLOCK_HEAD(KK, HS, BS, IS, IT) REG_ORG(KK) REG_STEP_HEAD(HS, BS, IS, IT)
#define REG_STEP_HEAD(HS, BS, IS, IT)                            \
    "s_lshl_b64 exec, %[lm], " #IT "\n"                          \
    "v_max_f64 %[dn], %[" HS "], %[" HS "] clamp\n"              \
    "v_add_f64 %[dn], %[" HS "], -%[dn]\n"                       \
    "v_fma_f64 %[dn], -%[dn], %[" IS "], %[" BS "]\n"            \
    "v_fmac_f64 %[" BS "], -1.0, %[dn]\n"                        \
    "s_mov_b64 exec, -1\n"
// Round 5: the successor list holds ABSOLUTE code addresses as (lo, hi) dword pairs — entry e (0 = the sweep's first block,
// 1 + k = the block visited after coordinate k) sits in the aligned scalar pair s[REG_PB + 2 e : REG_PB + 2 e + 1], REG_PB =
// 96 - 2 KMAX (KMAX <= 30; KMAX = 32 keeps offsets, see REGO_HEAD), loaded straight from the order-table row — so a block ends in `s_setpc_b64` on its own pair: no address add per
// step (one of the step's ten instructions), no s_getpc / add / addc / mov per sweep, no vcc, and no alignment demand on the
// table of blocks.  The table's address reaches k_order_table from a probe launch of the kernel itself (reg_code_base below).
#define REG_PB REG_STR(REG_PBN)
#define REG_JUMP(KK) "s_setpc_b64 s[" REG_PB "+2+2*" #KK ":" REG_PB "+3+2*" #KK "]\n"
#define REG_ORG(KK) ".org Lc%= + " REG_STR(INSIDER_REG_BLOCK) "*" #KK "\n"
#define REG_FMAC(H, GK, IT) "v_fmac_f64_dpp %[" H "], %[dn], %[" GK "] row_newbcast:" #IT " row_mask:0xf bank_mask:0xf\n"
#define REG_BLOCK2_LO(KK) \
    REG_BLOCK_HEAD(KK, "h0", "b0", "i0", KK) REG_FMAC("h0", "ga" #KK, KK) REG_FMAC("h1", "gb" #KK, KK) REG_JUMP(KK)
#define REG_BLOCK2_HI(KK, IT) \
    REG_BLOCK_HEAD(KK, "h1", "b1", "i1", IT) REG_FMAC("h0", "ga" #KK, IT) REG_FMAC("h1", "gb" #KK, IT) REG_JUMP(KK)
#define REG_BLOCK1(KK) REG_BLOCK_HEAD(KK, "h0", "b0", "i0", KK) REG_FMAC("h0", "ga" #KK, KK) REG_JUMP(KK)
// ---- blocks of TWO coordinate steps (round 5) -----------------------------------------------------------------------------
// A launch whose waves are mostly alone on their SIMDs (few genes, or the tail of the longest genes) is bound by ONE wave's
// step: its instructions at one per ~6 cycles plus the ~28 cycles an instruction buffer takes to refill behind the computed
// jump.  Where two consecutive coordinates of a sweep's order belong to the same coordinate slot, k_order_table can route the
// sweep through a block that steps both — the same instructions in the same order, so the iterates do not change — and a sweep
// takes ~20 jumps instead of 30.  With static operands there is one block per ordered pair of a slot: 16 x 16 + W x W of them
// (W = KMAX - 16; the a == b slots are never visited), 128 bytes apart, pair (a, b) of slot 0 at index 16 a + b, of slot 1 at
// 256 + W (a - 16) + (b - 16).  With absolute successor addresses they need neither a place inside the loop body nor an
// alignment: they sit in a section of their own behind a named label (reg_pair_base), so the sweep loop stays as compact as
// without them (round 4's in-line table of offsets cost 3.7 % in loop overheads and won back 1.9 %), and the
// ROUTING is a property of the order table: option "cd_pairs" decides per call whether sweeps go through them.
// (One LINE per step: the compiler's branch relaxation prices an asm statement at 20 bytes per line, and 452 blocks of 18 lines
// would make it turn the sweep loop's back edge into a four-instruction long branch.  The step is therefore an ASSEMBLER macro,
// defined inside the statement that uses it (every inline-asm statement is assembled by a parser of its own) and purged at its
// end — the same eight instructions as REG_STEP_HEAD + two REG_FMAC; a one-slot step has no second fmac.)
#define REG_STEP_MACROS                                                \
    ".macro INSIDER_CD_STEP2 it, h, b, i, lm, dn, h0, h1, g0, g1\n"    \
    "s_lshl_b64 exec, \\lm, \\it\n"                                    \
    "v_max_f64 \\dn, \\h, \\h clamp\n"                                \
    "v_add_f64 \\dn, \\h, -\\dn\n"                                    \
    "v_fma_f64 \\dn, -\\dn, \\i, \\b\n"                              \
    "v_fmac_f64 \\b, -1.0, \\dn\n"                                    \
    "s_mov_b64 exec, -1\n"                                             \
    "v_fmac_f64_dpp \\h0, \\dn, \\g0 row_newbcast:\\it row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp \\h1, \\dn, \\g1 row_newbcast:\\it row_mask:0xf bank_mask:0xf\n" \
    ".endm\n"                                                          \
    ".macro INSIDER_CD_STEP1 it, h, b, i, lm, dn, g0\n"                \
    "s_lshl_b64 exec, \\lm, \\it\n"                                    \
    "v_max_f64 \\dn, \\h, \\h clamp\n"                                \
    "v_add_f64 \\dn, \\h, -\\dn\n"                                    \
    "v_fma_f64 \\dn, -\\dn, \\i, \\b\n"                              \
    "v_fmac_f64 \\b, -1.0, \\dn\n"                                    \
    "s_mov_b64 exec, -1\n"                                             \
    "v_fmac_f64_dpp \\h, \\dn, \\g0 row_newbcast:\\it row_mask:0xf bank_mask:0xf\n" \
    ".endm\n"
#define REG_STEP_LO(KK) "INSIDER_CD_STEP2 " #KK ", %[h0], %[b0], %[i0], %[lm], %[dn], %[h0], %[h1], %[ga" #KK "], %[gb" #KK "]\n"
#define REG_STEP_HI(KK, IT) "INSIDER_CD_STEP2 " #IT ", %[h1], %[b1], %[i1], %[lm], %[dn], %[h0], %[h1], %[ga" #KK "], %[gb" #KK "]\n"
#define REG_STEP1_LO(KK) "INSIDER_CD_STEP1 " #KK ", %[h0], %[b0], %[i0], %[lm], %[dn], %[ga" #KK "]\n"
#define REGP_LO(A, B) ".p2align 7\n" REG_STEP_LO(A) REG_STEP_LO(B) REG_JUMP(B)
#define REGP_HI(A, IA, B, IB) ".p2align 7\n" REG_STEP_HI(A, IA) REG_STEP_HI(B, IB) REG_JUMP(B)
#define REGP1_LO(A, B) ".p2align 7\n" REG_STEP1_LO(A) REG_STEP1_LO(B) REG_JUMP(B)
#define REG_LIST_LO2(F, A) F(A, 0) F(A, 1) F(A, 2) F(A, 3) F(A, 4) F(A, 5) F(A, 6) F(A, 7) F(A, 8) F(A, 9) F(A, 10) F(A, 11) F(A, 12) F(A, 13) F(A, 14) F(A, 15)
#define REG_HB2_18(F, A, IA) F(A, IA, 16, 0) F(A, IA, 17, 1)
#define REG_HB2_20(F, A, IA) REG_HB2_18(F, A, IA) F(A, IA, 18, 2) F(A, IA, 19, 3)
#define REG_HB2_22(F, A, IA) REG_HB2_20(F, A, IA) F(A, IA, 20, 4) F(A, IA, 21, 5)
#define REG_HB2_24(F, A, IA) REG_HB2_22(F, A, IA) F(A, IA, 22, 6) F(A, IA, 23, 7)
#define REG_HB2_26(F, A, IA) REG_HB2_24(F, A, IA) F(A, IA, 24, 8) F(A, IA, 25, 9)
#define REG_HB2_28(F, A, IA) REG_HB2_26(F, A, IA) F(A, IA, 26, 10) F(A, IA, 27, 11)
#define REG_HB2_30(F, A, IA) REG_HB2_28(F, A, IA) F(A, IA, 28, 12) F(A, IA, 29, 13)
#define REGP_ROW_LO(A) REG_LIST_LO2(REGP_LO, A)
#define REGP1_ROW_LO(A) REG_LIST_LO2(REGP1_LO, A)
#define REGP_ROW_HI(A, IA) REG_CAT(REG_HB2_, REG_KM)(REGP_HI, A, IA)
// the pair blocks of an instantiation, in index order, in their own section; the label in front of them is what reg_pair_base takes
#define REG_PAIRS_OPEN                                                                       \
    REG_STEP_MACROS                                                                          \
    ".pushsection .text.insider_cdpair_" REG_STR(REG_KM) ",\"ax\",@progbits\n"               \
    ".p2align 7\n"                                                                           \
    "insider_cdpair_" REG_STR(REG_KM) ":\n"
#define REG_PAIRS_CLOSE ".p2align 7\n s_endpgm\n .popsection\n .purgem INSIDER_CD_STEP2\n .purgem INSIDER_CD_STEP1\n"
// the row's pairs into s[REG_PB : 97] : 2 (KMAX + 1) dwords from byte REG_ORDER_OFF of the order-table row, in the largest aligned
// pieces (the destination of an x4 / x8 / x16 scalar load is 4-aligned: REG_PB is a multiple of 4 for even KMAX)
#define REG_LOADS_16(AD) "s_load_dwordx16 s[64:79], " AD "0x0\n s_load_dwordx16 s[80:95], " AD "0x40\n s_load_dwordx2 s[96:97], " AD "0x80\n"
#define REG_LOADS_18(AD) "s_load_dwordx16 s[60:75], " AD "0x0\n s_load_dwordx16 s[76:91], " AD "0x40\n s_load_dwordx4 s[92:95], " AD "0x80\n s_load_dwordx2 s[96:97], " AD "0x90\n"
#define REG_LOADS_20(AD) "s_load_dwordx16 s[56:71], " AD "0x0\n s_load_dwordx16 s[72:87], " AD "0x40\n s_load_dwordx8 s[88:95], " AD "0x80\n s_load_dwordx2 s[96:97], " AD "0xa0\n"
#define REG_LOADS_22(AD) "s_load_dwordx16 s[52:67], " AD "0x0\n s_load_dwordx16 s[68:83], " AD "0x40\n s_load_dwordx8 s[84:91], " AD "0x80\n s_load_dwordx4 s[92:95], " AD "0xa0\n s_load_dwordx2 s[96:97], " AD "0xb0\n"
#define REG_LOADS_24(AD) "s_load_dwordx16 s[48:63], " AD "0x0\n s_load_dwordx16 s[64:79], " AD "0x40\n s_load_dwordx16 s[80:95], " AD "0x80\n s_load_dwordx2 s[96:97], " AD "0xc0\n"
#define REG_LOADS_26(AD) "s_load_dwordx16 s[44:59], " AD "0x0\n s_load_dwordx16 s[60:75], " AD "0x40\n s_load_dwordx16 s[76:91], " AD "0x80\n s_load_dwordx4 s[92:95], " AD "0xc0\n s_load_dwordx2 s[96:97], " AD "0xd0\n"
#define REG_LOADS_28(AD) "s_load_dwordx16 s[40:55], " AD "0x0\n s_load_dwordx16 s[56:71], " AD "0x40\n s_load_dwordx16 s[72:87], " AD "0x80\n s_load_dwordx8 s[88:95], " AD "0xc0\n s_load_dwordx2 s[96:97], " AD "0xe0\n"
#define REG_LOADS_30(AD) "s_load_dwordx16 s[36:51], " AD "0x0\n s_load_dwordx16 s[52:67], " AD "0x40\n s_load_dwordx16 s[68:83], " AD "0x80\n s_load_dwordx8 s[84:91], " AD "0xc0\n s_load_dwordx4 s[92:95], " AD "0xe0\n s_load_dwordx2 s[96:97], " AD "0xf0\n"
#define REG_CAT_(a, b) a##b
#define REG_CAT(a, b) REG_CAT_(a, b)
// ---- the sweep LOOP as one asm statement (round 5, K <= 30) -------------------------------------------------------------------
// Not only the sweep but the per-sweep bookkeeping and the loop control are inside the statement: the common path of a sweep
// never leaves it.  Layout:
//     entry:   load the successor list of sweep `sw` (row `off` of the order table) and the stash values; branch to Lgo
//     Lc:      the table of code blocks (one step each; the two-step blocks live in their own section)
//     exit block (index KMAX): the exact loss change of the sweep (the instructions the compiler made of the C++ that
//              stood here until round 4, in its order: same roundings), ++sw, the NEXT sweep's list requested BEFORE the
//              bookkeeping so that its latency hides behind it (the list load at the start of a sweep was exposed: the
//              look-ahead touches alone are worth 5 %), the convergence test on the row sums, and
//              leave (Lout) when a gene may stop or the caller's bound is reached; else fall into
//     Lgo:     wait for the list, jump into the sweep's first block
// so a sweep costs its blocks' jumps plus ONE (no back edge).  The rarely taken part — screening violations, parking a finished
// gene, window sums of a limited pass — stays C++ behind the statement (cd_reg).
// The row reduction moves 64-bit values as two 32-bit DPP moves (64-bit DPP knows row_newbcast only), and inline asm cannot name
// the halves of a 64-bit operand: its two temporaries are therefore pinned to v[2:3] / v[4:5].
#define REG_AD "%[tb0], %[off] offset:"
#define REG_ROWSTR REG_STR(INSIDER_ORDER_ROW)
// (No look-ahead touches of the row after next: with the list requested a whole bookkeeping ahead they are not needed any more —
// A/B in round 5: 0.6 % fewer sweep-kernel ms without them; in the per-sweep form of round 4, where the list load stood in front of
// the sweep, they were worth 5 %.)
#define REG_LOOP_ENTRY                                                \
    REG_LDS REG_CAT(REG_LOADS_, REG_KM)(REG_AD)                       \
    "s_branch Lgo%=\n"                                                \
    ".p2align 8\n"                                                    \
    "insider_cdtab_" REG_STR(REG_KM) "_%c[who]:\n"                     \
    "Lc%=:\n"
#define REG_DPP4(D0, D1, S0, S1, CTRL)                                                          \
    "v_mov_b32_dpp " D0 ", " S0 " " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"           \
    "v_mov_b32_dpp " D1 ", " S1 " " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// sum over each 16-lane row of la * t (t in v[2:3]) -> v[2:3]: row16_sum(la * t) as the compiler contracted it.  (The tails keep
// three of their values in the pinned pairs and in dn, which is free between sweeps: the loop needs no more registers than the C++
// bookkeeping took.)
#define REG_ROWSUM                                                    \
    "v_mul_f64 v[4:5], %[las], v[2:3]\n"                              \
    "s_nop 1\n"                                                       \
    REG_DPP4("v4", "v5", "v4", "v5", "quad_perm:[1,0,3,2]")           \
    "v_fmac_f64 v[4:5], %[las], v[2:3]\n"                             \
    "s_nop 1\n"                                                       \
    REG_DPP4("v2", "v3", "v4", "v5", "quad_perm:[2,3,0,1]")           \
    "v_add_f64 v[2:3], v[4:5], v[2:3]\n"                              \
    "s_nop 1\n"                                                       \
    REG_DPP4("v4", "v5", "v2", "v3", "row_half_mirror")               \
    "v_add_f64 v[2:3], v[2:3], v[4:5]\n"                              \
    "s_nop 1\n"                                                       \
    REG_DPP4("v4", "v5", "v2", "v3", "row_mirror")                    \
    "v_add_f64 v[2:3], v[2:3], v[4:5]\n"
// (stash offsets: D at 0, the sweep-start beta at SA * 8 = 1024, the sweep-start what at 2048; second slot + 512: reg_sa)
#define REG_NEXT_LIST                                                 \
    "s_add_i32 %[sw], %[sw], 1\n"                                     \
    "s_and_b32 %[off], %[sw], (" REG_STR(INSIDER_PERM_PERIOD) "-1)\n"  \
    "s_mulk_i32 %[off], " REG_ROWSTR "\n"                             \
    "s_waitcnt lgkmcnt(0)\n"                                          \
    REG_CAT(REG_LOADS_, REG_KM)(REG_AD)
#define REG_LOOP_END                                                  \
    "v_add_f64 %[aw], %[aw], |v[2:3]|\n"                              \
    "v_cmp_nlt_f64_e64 vcc, %[tol], |v[2:3]|\n"                       \
    "s_nop 0\n"                                                       \
    "s_and_b64 %[cand], vcc, %[run]\n"                                \
    "s_cbranch_scc1 Lout%=\n"                                         \
    "s_cmp_lt_i32 %[sw], %[stop]\n"                                   \
    "s_cbranch_scc0 Lout%=\n"                                         \
    "Lgo%=:\n"                                                        \
    "s_waitcnt lgkmcnt(0)\n"                                          \
    "s_setpc_b64 s[" REG_PB ":" REG_PB "+1]\n"                        \
    "Lout%=:\n"                                                       \
    "s_waitcnt lgkmcnt(0)\n"
// (the stash values a tail needs — D, the sweep-start beta and what — are requested at the END of the previous tail, behind its
// writes, into registers the blocks do not touch: a tail starts computing at once)
#define REG_LDS2                                                      \
    "ds_read_b64 %[t0], %[la]\n"                                      \
    "ds_read_b64 %[t1], %[la] offset:512\n"                           \
    "ds_read_b64 %[t2], %[la] offset:1024\n"                          \
    "ds_read_b64 %[t3], %[la] offset:1536\n"                          \
    "ds_read_b64 %[t4], %[la] offset:2048\n"                          \
    "ds_read_b64 %[t5], %[la] offset:2560\n"
#define REG_LDS1                                                      \
    "ds_read_b64 %[t0], %[la]\n"                                      \
    "ds_read_b64 %[t2], %[la] offset:1024\n"                          \
    "ds_read_b64 %[t4], %[la] offset:2048\n"
#define REG_TAIL2                                                     \
    REG_NEXT_LIST                                                     \
    "v_fma_f64 %[t0], %[b0], %[t0], -%[h0]\n"                         \
    "v_fma_f64 %[t1], %[b1], %[t1], -%[h1]\n"                         \
    "v_add_f64 %[t4], %[t0], %[t4]\n"                                 \
    "v_add_f64 %[dn], %[b0], -%[t2]\n"                                \
    "v_add_f64 %[t4], %[t4], 1.0\n"                                   \
    "v_add_f64 %[t5], %[t1], %[t5]\n"                                 \
    "v_add_f64 %[t2], |%[b0]|, -|%[t2]|\n"                            \
    "v_fma_f64 %[t4], %[dn], %[t4], 0\n"                              \
    "v_add_f64 %[dn], %[b1], -%[t3]\n"                                \
    "v_add_f64 %[t5], %[t5], 1.0\n"                                   \
    "v_add_f64 %[t3], |%[b1]|, -|%[t3]|\n"                            \
    "v_fmac_f64 %[t4], %[dn], %[t5]\n"                                \
    "v_add_f64 %[t2], %[t2], %[t3]\n"                                 \
    "ds_write_b64 %[la], %[b0] offset:1024\n"                         \
    "ds_write_b64 %[la], %[b1] offset:1536\n"                         \
    "ds_write_b64 %[la], %[t0] offset:2048\n"                         \
    "ds_write_b64 %[la], %[t1] offset:2560\n"                         \
    "v_add_f64 v[2:3], %[t2], %[t4]\n"                                \
    REG_LDS2 REG_ROWSUM REG_LOOP_END
#define REG_TAIL1                                                     \
    REG_NEXT_LIST                                                     \
    "v_fma_f64 %[t0], %[b0], %[t0], -%[h0]\n"                         \
    "v_add_f64 %[t4], %[t0], %[t4]\n"                                 \
    "v_add_f64 %[dn], %[b0], -%[t2]\n"                                \
    "ds_write_b64 %[la], %[b0] offset:1024\n"                         \
    "ds_write_b64 %[la], %[t0] offset:2048\n"                         \
    "v_add_f64 %[t4], %[t4], 1.0\n"                                   \
    "v_add_f64 %[t2], |%[b0]|, -|%[t2]|\n"                            \
    "v_fma_f64 %[t4], %[dn], %[t4], 0\n"                              \
    "v_add_f64 v[2:3], %[t2], %[t4]\n"                                \
    REG_LDS1 REG_ROWSUM REG_LOOP_END
#define REG_EPILOGUE(NBLK) REG_ORG(NBLK) " s_waitcnt lgkmcnt(0)\n"   /* exit block */
#define REG_S48_63 "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63",
#define REG_S64_97 "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97"
// what an instantiation clobbers: its pair registers (the compiler keeps the sweep loop's own scalars below them)
#define REG_CLOB_16 "scc", "memory", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97"
#define REG_CLOB_18 "scc", "memory", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97"
#define REG_CLOB_20 "scc", "memory", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97"
#define REG_CLOB_22 "scc", "memory", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97"
#define REG_CLOB_24 "scc", "memory", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97"
#define REG_CLOB_26 "scc", "memory", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97"
#define REG_CLOB_28 "scc", "memory", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97"
#define REG_CLOB_30 "scc", "memory", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97"
#define REG_CLOBBERS REG_CAT(REG_CLOB_, REG_KM)
#define REG_GA(KK) [ga##KK] "v"(G[0][KK]),
#define REG_GB(KK) [gb##KK] "v"(G[1][KK]),
#define REG_HI16(KK, IT) REG_BLOCK2_HI(KK, IT)
#define REG_LIST_LO(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
// coordinates 16 .. KMAX-1, as (coordinate) for operand lists and (coordinate, owner lane) for blocks
#define REG_HI_18(F) F(16) F(17)
#define REG_HI_20(F) REG_HI_18(F) F(18) F(19)
#define REG_HI_22(F) REG_HI_20(F) F(20) F(21)
#define REG_HI_24(F) REG_HI_22(F) F(22) F(23)
#define REG_HI_26(F) REG_HI_24(F) F(24) F(25)
#define REG_HI_28(F) REG_HI_26(F) F(26) F(27)
#define REG_HI_30(F) REG_HI_28(F) F(28) F(29)
#define REG_HI_32(F) REG_HI_30(F) F(30) F(31)
#define REG_HB_18(F) F(16, 0) F(17, 1)
#define REG_HB_20(F) REG_HB_18(F) F(18, 2) F(19, 3)
#define REG_HB_22(F) REG_HB_20(F) F(20, 4) F(21, 5)
#define REG_HB_24(F) REG_HB_22(F) F(22, 6) F(23, 7)
#define REG_HB_26(F) REG_HB_24(F) F(24, 8) F(25, 9)
#define REG_HB_28(F) REG_HB_26(F) F(26, 10) F(27, 11)
#define REG_HB_30(F) REG_HB_28(F) F(28, 12) F(29, 13)
#define REG_HB_32(F) REG_HB_30(F) F(30, 14) F(31, 15)

// The sweep loop (above).  tb0: row 0 of the order table + REG_ORDER_OFF; off: byte offset of sweep `sw`'s row (in / out); sw: the
// sweep counter (in / out: the statement leaves with the number of sweeps done); stop: leave when sw reaches it; run: lane mask
// of the genes still running; lds: LDS byte address of this lane's first stash cell; accw: running sum of |loss change| (the
// windows of a limited pass: the caller resets and files it); dl: the last sweep's loss change (row sum); cand: lanes of running
// genes whose |loss change| <= tol in the last sweep (0: the statement left because sw == stop).
// WHO: which kernel the loop is inlined into (0 = the column update k_cd_cols_reg, 1 = the stand-alone batch k_cd_batch_reg): part
// of the NAMED label of the table of blocks, whose address reg_code_base() takes from another asm statement of the same kernel;
// the column-update kernel also carries the blocks of two steps.
#define REG_LOOP_OUTS2                                                                                                     \
    [h0] "+v"(S.y[0]), [h1] "+v"(S.y[1]), [b0] "+v"(S.beta[0]), [b1] "+v"(S.beta[1]), [aw] "+v"(accw), [dl] "=&{v[2:3]}"(dl), \
        [rb] "=&{v[4:5]}"(rb), [dn] "=&v"(dn), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4),      \
        [t5] "=&v"(t5), [sw] "+s"(sw), [off] "+s"(off), [cand] "=&s"(cand)
#define REG_LOOP_INS_TAIL                                                                                                  \
    [tb0] "s"(tb0), [lm] "s"(lm), [run] "s"(run), [las] "s"(la), [tol] "s"(tol), [stop] "s"(stop), [la] "v"(lds), [who] "i"(WHO)
#define REG_LOOP_CLOBBERS "vcc", REG_CLOBBERS
#if defined(__HIP_DEVICE_COMPILE__)   // gfx950 assembly: hipcc's host pass must not parse it
#define REG_DEFINE_SWEEP2(KMAX)                                                                                          \
    template <int WHO>                                                                                                   \
    __device__ __forceinline__ void reg_sweeps(RegState<2> &S, const double (&G)[2][KMAX], const uint32_t *tb0, int &off, \
                                               int &sw, int stop, uint64_t run, double la, double tol, uint32_t lds,     \
                                               double &accw, double &dl, uint64_t &cand)                                 \
    {                                                                                                                    \
        double dn, rb, t0, t1, t2, t3, t4, t5;                                                                           \
        const uint64_t lm = 0x0001000100010001ull;                                                                       \
        if constexpr (WHO == 0)                                                                                          \
            asm volatile(REG_LOOP_ENTRY REG_LIST_LO(REG_BLOCK2_LO) REG_HB_##KMAX(REG_HI16) REG_ORG(KMAX) REG_TAIL2            \
                             REG_PAIRS_OPEN REG_LIST_LO(REGP_ROW_LO) REG_HB_##KMAX(REGP_ROW_HI) REG_PAIRS_CLOSE               \
                         : REG_LOOP_OUTS2                                                                                \
                         : REG_LIST_LO(REG_GA) REG_HI_##KMAX(REG_GA) REG_LIST_LO(REG_GB) REG_HI_##KMAX(REG_GB)[i0] "v"(  \
                               S.tau[0]),                                                                                \
                           [i1] "v"(S.tau[1]), REG_LOOP_INS_TAIL                                                         \
                         : REG_LOOP_CLOBBERS);                                                                           \
        else                                                                                                             \
            asm volatile(REG_LOOP_ENTRY REG_LIST_LO(REG_BLOCK2_LO) REG_HB_##KMAX(REG_HI16) REG_ORG(KMAX) REG_TAIL2            \
                         : REG_LOOP_OUTS2                                                                                \
                         : REG_LIST_LO(REG_GA) REG_HI_##KMAX(REG_GA) REG_LIST_LO(REG_GB) REG_HI_##KMAX(REG_GB)[i0] "v"(  \
                               S.tau[0]),                                                                                \
                           [i1] "v"(S.tau[1]), REG_LOOP_INS_TAIL                                                         \
                         : REG_LOOP_CLOBBERS);                                                                           \
    }
#else
#define REG_DEFINE_SWEEP2(KMAX)                                                                                          \
    template <int WHO>                                                                                                   \
    __device__ __forceinline__ void reg_sweeps(RegState<2> &, const double (&)[2][KMAX], const uint32_t *, int &, int &, int, \
                                               uint64_t, double, double, uint32_t, double &, double &, uint64_t &) {}
#endif
#define REG_LDS REG_LDS2
#define REG_KM 18
#define REG_PBN 60
REG_DEFINE_SWEEP2(18)
#undef REG_KM
#undef REG_PBN
#define REG_KM 20
#define REG_PBN 56
REG_DEFINE_SWEEP2(20)
#undef REG_KM
#undef REG_PBN
#define REG_KM 22
#define REG_PBN 52
REG_DEFINE_SWEEP2(22)
#undef REG_KM
#undef REG_PBN
#define REG_KM 24
#define REG_PBN 48
REG_DEFINE_SWEEP2(24)
#undef REG_KM
#undef REG_PBN
#define REG_KM 26
#define REG_PBN 44
REG_DEFINE_SWEEP2(26)
#undef REG_KM
#undef REG_PBN
#define REG_KM 28
#define REG_PBN 40
REG_DEFINE_SWEEP2(28)
#undef REG_KM
#undef REG_PBN
#define REG_KM 30
#define REG_PBN 36
REG_DEFINE_SWEEP2(30)
#undef REG_KM
#undef REG_PBN

// KMAX = 32 (K = 31, 32) keeps the successor list as 32-bit block OFFSETS: its 33 pairs would take s34 ... s99, every scalar
// register between the reserved s32 and s100, and the kernel (168 VGPRs for 128 matrix registers) then spills a vector register
// under the step's narrowed exec mask.  The list is loaded into s[64:97] (dword 0 = first block, dword 1 + k = the block after
// coordinate k), each block adds the table's base address (s98; the table starts on a 4 KiB boundary and is shorter than 4 KiB,
// so every block shares the high word, vcc_hi) to its own entry in vcc_lo while the vector chain runs, and jumps through vcc.
#define REGO_HEAD(KK, HS, BS, IS, IT)                            \
    REG_ORG(KK)                                                  \
    "s_lshl_b64 exec, %[lm], " #IT "\n"                          \
    "v_max_f64 %[dn], %[" HS "], %[" HS "] clamp\n"              \
    "s_add_u32 vcc_lo, s[65+" #KK "], s98\n"                     \
    "v_add_f64 %[dn], %[" HS "], -%[dn]\n"                       \
    "v_fma_f64 %[dn], -%[dn], %[" IS "], %[" BS "]\n"            \
    "v_fmac_f64 %[" BS "], -1.0, %[dn]\n"                        \
    "s_mov_b64 exec, -1\n"
#define REGO_LO(KK) REGO_HEAD(KK, "h0", "b0", "i0", KK) REG_FMAC("h0", "ga" #KK, KK) REG_FMAC("h1", "gb" #KK, KK) "s_setpc_b64 vcc\n"
#define REGO_HI(KK, IT) REGO_HEAD(KK, "h1", "b1", "i1", IT) REG_FMAC("h0", "ga" #KK, IT) REG_FMAC("h1", "gb" #KK, IT) "s_setpc_b64 vcc\n"
#define REGO_PROLOGUE                              \
    "s_load_dwordx16 s[64:79], %[tb], 0x0\n"       \
    "s_load_dwordx16 s[80:95], %[tb], 0x40\n"      \
    "s_load_dwordx2 s[96:97], %[tb], 0x80\n"       \
    "s_getpc_b64 s[98:99]\n"                       \
    "Lr%=:\n"                                      \
    "s_add_u32 s98, s98, Lc%=-Lr%=\n"              \
    "s_addc_u32 s99, s99, 0\n"                     \
    "s_mov_b32 vcc_hi, s99\n"                      \
    "s_waitcnt lgkmcnt(0)\n"                       \
    "s_load_dword %[sk], %[tb], " REG_STR(INSIDER_ORDER_ROW) "\n"        \
    "s_load_dword %[p1], %[tb], " REG_STR(INSIDER_ORDER_ROW) "+0x40\n"   \
    "s_load_dword %[p2], %[tb], " REG_STR(INSIDER_ORDER_ROW) "+0x80\n"   \
    "s_add_u32 vcc_lo, s64, s98\n"                 \
    "s_setpc_b64 vcc\n.p2align 12\n"              \
    "Lc%=:\n"
#if defined(__HIP_DEVICE_COMPILE__)
template <int WHO>
__device__ __forceinline__ void reg_sweep(RegState<2> &S, const double (&G)[2][32], const uint32_t *tb)
{
    double dn;
    int sk, p1, p2;
    const uint64_t lm = 0x0001000100010001ull;
    asm volatile(REGO_PROLOGUE REG_LIST_LO(REGO_LO) REG_HB_32(REGO_HI) REG_EPILOGUE(32)
                 : [h0] "+v"(S.y[0]), [h1] "+v"(S.y[1]), [b0] "+v"(S.beta[0]), [b1] "+v"(S.beta[1]), [dn] "=&v"(dn), [sk] "=&s"(sk),
                   [p1] "=&s"(p1), [p2] "=&s"(p2)
                 : REG_LIST_LO(REG_GA) REG_HI_32(REG_GA) REG_LIST_LO(REG_GB) REG_HI_32(REG_GB)[i0] "v"(S.tau[0]), [i1] "v"(S.tau[1]),
                   [tb] "s"(tb), [lm] "s"(lm)
                 : "vcc", "scc", "memory", REG_S64_97, "s98", "s99");
}
#else
template <int WHO>
__device__ __forceinline__ void reg_sweep(RegState<2> &, const double (&)[2][32], const uint32_t *) {}
#endif

#undef REG_LDS
#define REG_LDS REG_LDS1
#define REG_KM 16
#define REG_PBN 64
#define REG_LOOP_OUTS1                                                                                                      \
    [h0] "+v"(S.y[0]), [b0] "+v"(S.beta[0]), [aw] "+v"(accw), [dl] "=&{v[2:3]}"(dl), [rb] "=&{v[4:5]}"(rb), [dn] "=&v"(dn),  \
        [t0] "=&v"(t0), [t2] "=&v"(t2), [t4] "=&v"(t4), [sw] "+s"(sw), [off] "+s"(off), [cand] "=&s"(cand)
template <int WHO>
__device__ __forceinline__ void reg_sweeps(RegState<1> &S, const double (&G)[1][16], const uint32_t *tb0, int &off, int &sw, int stop,
                                           uint64_t run, double la, double tol, uint32_t lds, double &accw, double &dl, uint64_t &cand)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double dn, rb, t0, t2, t4;
    const uint64_t lm = 0x0001000100010001ull;
    if constexpr (WHO == 0)
        asm volatile(REG_LOOP_ENTRY REG_LIST_LO(REG_BLOCK1) REG_ORG(16) REG_TAIL1 REG_PAIRS_OPEN REG_LIST_LO(REGP1_ROW_LO) REG_PAIRS_CLOSE
                     : REG_LOOP_OUTS1
                     : REG_LIST_LO(REG_GA)[i0] "v"(S.tau[0]), REG_LOOP_INS_TAIL
                     : REG_LOOP_CLOBBERS);
    else
        asm volatile(REG_LOOP_ENTRY REG_LIST_LO(REG_BLOCK1) REG_ORG(16) REG_TAIL1
                     : REG_LOOP_OUTS1
                     : REG_LIST_LO(REG_GA)[i0] "v"(S.tau[0]), REG_LOOP_INS_TAIL
                     : REG_LOOP_CLOBBERS);
#endif
}
#undef REG_KM
#undef REG_PBN

// Address of the pair blocks of the KMAX instantiation of the column-update kernel (their own section: a pc-relative relocation)
template <int KMAX>
__device__ __forceinline__ uint64_t reg_pair_base()
{
    uint32_t lo = 0, hi = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_getpc_b64 s[98:99]\n"
                 "s_add_u32 s98, s98, insider_cdpair_%c[km]@rel32@lo+4\n"
                 "s_addc_u32 s99, s99, insider_cdpair_%c[km]@rel32@hi+12\n"
                 "s_mov_b32 %[lo], s98\n"
                 "s_mov_b32 %[hi], s99\n"
                 : [lo] "=s"(lo), [hi] "=s"(hi)
                 : [km] "i"(KMAX)
                 : "s98", "s99", "scc");
#endif
    return ((uint64_t)hi << 32) | lo;
}

// Address of the table of code blocks of THIS kernel's sweep (the named label of REG_PROLOGUE): k_order_table adds the block
// offsets to it, so the kernel that will run the sweeps is launched once with a null gene set to publish it (ColArgs::code_base)
template <int KMAX, int WHO>
__device__ __forceinline__ uint64_t reg_code_base()
{
    uint32_t lo = 0, hi = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_getpc_b64 s[98:99]\n"
                 "Lq%=:\n"
                 "s_add_u32 s98, s98, insider_cdtab_%c[km]_%c[who]-Lq%=\n"
                 "s_addc_u32 s99, s99, 0\n"
                 "s_mov_b32 %[lo], s98\n"
                 "s_mov_b32 %[hi], s99\n"
                 : [lo] "=s"(lo), [hi] "=s"(hi)
                 : [km] "i"(KMAX), [who] "i"(WHO)
                 : "s98", "s99", "scc");
#endif
    return ((uint64_t)hi << 32) | lo;
}

// ---- 32 < K <= 48: three coordinate slots, the third slot's Gram columns in LDS (round 4) ---------------------------------
// Lane i also owns coordinate 32 + i (i < KMAX - 32).  Columns i and 16 + i of the Gram matrix stay in VGPRs (4 KMAX of the 256
// a wave may hold at two waves per SIMD); column 32 + i lives in an LDS panel of this wave — row k at byte k * PS, PS = (4 W + 1) * 8,
// W = KMAX - 32 coordinates per gene, cell g * W + i for lane i < W of gene row g, and ONE zero cell (4 W) per row that every lane
// without a third coordinate reads (so its y stays what it was).  Inside block k the panel row is static like the register
// operands: `ds_read_b64 gc, la offset:k*PS` is the block's first instruction, the third DPP fmac waits for it behind the two
// register ones.  Per step 7 vector + 1 LDS + 4 scalar instructions + the wait.  The successor list takes 49 dwords (dword 0 +
// one per coordinate, K <= 48) in s[48:96], read from byte 124 of the order-table row (REG3_ORDER_OFF, k_order_table).  A block
// is exactly INSIDER_REG3_BLOCK = 80 bytes long and the blocks are packed (.org fails the build if one outgrows its slot): the
// table of 49 blocks stays below 4 KiB, so the shared-high-word argument of the two-slot kernel holds with page alignment alone.
constexpr int reg3_w(int KMAX) { return KMAX - 32; }
constexpr int reg3_ps(int KMAX) { return 4 * reg3_w(KMAX) + 1; }            // panel row pitch in doubles
#define REG3_ORG(KK) ".org Lc%= + " REG_STR(INSIDER_REG3_BLOCK) "*" #KK "\n"
#define REG3_HEAD(KK, HS, BS, IS, IT)                                        \
    REG3_ORG(KK)                                                             \
    "ds_read_b64 %[gc], %[la] offset:(" #KK "*" REG3_PS_STR ")\n"            \
    "s_lshl_b64 exec, %[lm], " #IT "\n"                                      \
    "v_max_f64 %[dn], %[" HS "], %[" HS "] clamp\n"                          \
    "s_add_u32 vcc_lo, s[49+" #KK "], s98\n"                                 \
    "v_add_f64 %[dn], %[" HS "], -%[dn]\n"                                   \
    "v_fma_f64 %[dn], -%[dn], %[" IS "], %[" BS "]\n"                        \
    "v_fmac_f64 %[" BS "], -1.0, %[dn]\n"                                    \
    "s_mov_b64 exec, -1\n"
#define REG3_TAIL(KK, IT)                                                                            \
    REG_FMAC("h0", "ga" #KK, IT)                                                                     \
    REG_FMAC("h1", "gb" #KK, IT) "s_waitcnt lgkmcnt(0)\n"                                            \
    "v_fmac_f64_dpp %[h2], %[dn], %[gc] row_newbcast:" #IT " row_mask:0xf bank_mask:0xf\n" "s_setpc_b64 vcc\n"
#define REG3_BLOCK_LO(KK) REG3_HEAD(KK, "h0", "b0", "i0", KK) REG3_TAIL(KK, KK)
#define REG3_BLOCK_MID(KK, IT) REG3_HEAD(KK, "h1", "b1", "i1", IT) REG3_TAIL(KK, IT)
#define REG3_BLOCK_TOP(KK, IT) REG3_HEAD(KK, "h2", "b2", "i2", IT) REG3_TAIL(KK, IT)
#define REG3_PROLOGUE                              \
    "s_load_dwordx16 s[48:63], %[tb], 0x0\n"       \
    "s_load_dwordx16 s[64:79], %[tb], 0x40\n"      \
    "s_load_dwordx16 s[80:95], %[tb], 0x80\n"      \
    "s_load_dword s96, %[tb], 0xc0\n"              \
    "s_getpc_b64 s[98:99]\n"                       \
    "Lr%=:\n"                                      \
    "s_add_u32 s98, s98, Lc%=-Lr%=\n"              \
    "s_addc_u32 s99, s99, 0\n"                     \
    "s_mov_b32 vcc_hi, s99\n"                      \
    "s_waitcnt lgkmcnt(0)\n"                       \
    "s_load_dword %[sk], %[tb], " REG_STR(INSIDER_ORDER_ROW) "\n"        /* the next row's list: four lines from byte 124 on */ \
    "s_load_dword %[p1], %[tb], " REG_STR(INSIDER_ORDER_ROW) "+0x4\n"    \
    "s_load_dword %[p2], %[tb], " REG_STR(INSIDER_ORDER_ROW) "+0x44\n"   \
    "s_load_dword %[p3], %[tb], " REG_STR(INSIDER_ORDER_ROW) "+0x84\n"   \
    "s_add_u32 vcc_lo, s48, s98\n"                 \
    "s_setpc_b64 vcc\n.p2align 12\n"              \
    "Lc%=:\n"
#define REG3_CLOBBERS "vcc", "scc", "memory", REG_S48_63 REG_S64_97, "s98", "s99"
#define REG_T1_36(F) F(32) F(33) F(34) F(35)
#define REG_T1_40(F) REG_T1_36(F) F(36) F(37) F(38) F(39)
#define REG_T1_44(F) REG_T1_40(F) F(40) F(41) F(42) F(43)
#define REG_T1_48(F) REG_T1_44(F) F(44) F(45) F(46) F(47)
#define REG_TB_36(F) F(32, 0) F(33, 1) F(34, 2) F(35, 3)
#define REG_TB_40(F) REG_TB_36(F) F(36, 4) F(37, 5) F(38, 6) F(39, 7)
#define REG_TB_44(F) REG_TB_40(F) F(40, 8) F(41, 9) F(42, 10) F(43, 11)
#define REG_TB_48(F) REG_TB_44(F) F(44, 12) F(45, 13) F(46, 14) F(47, 15)
// la: LDS byte address of this lane's cell in row 0 of the panel
#if defined(__HIP_DEVICE_COMPILE__)
#define REG_DEFINE_SWEEP3(KMAX)                                                                                          \
    template <int WHO>                                                                                                   \
    __device__ __forceinline__ void reg_sweep(RegState<3> &S, const double (&G)[2][KMAX], const uint32_t *tb, uint32_t la) \
    {                                                                                                                    \
        double dn, gc;                                                                                                   \
        int sk, p1, p2, p3;                                                                                              \
        const uint64_t lm = 0x0001000100010001ull;                                                                       \
        asm volatile(REG3_PROLOGUE REG_LIST_LO(REG3_BLOCK_LO) REG_HB_32(REG3_BLOCK_MID) REG_TB_##KMAX(REG3_BLOCK_TOP)    \
                         REG3_ORG(KMAX) " s_waitcnt lgkmcnt(0)\n"   /* exit block */                                    \
                     : [h0] "+v"(S.y[0]), [h1] "+v"(S.y[1]), [h2] "+v"(S.y[2]), [b0] "+v"(S.beta[0]), [b1] "+v"(S.beta[1]), \
                       [b2] "+v"(S.beta[2]), [dn] "=&v"(dn), [gc] "=&v"(gc), [sk] "=&s"(sk), [p1] "=&s"(p1), [p2] "=&s"(p2),   \
                       [p3] "=&s"(p3)                                                                                    \
                     : REG_LIST_LO(REG_GA) REG_HI_32(REG_GA) REG_T1_##KMAX(REG_GA) REG_LIST_LO(REG_GB) REG_HI_32(REG_GB)   \
                           REG_T1_##KMAX(REG_GB)[i0] "v"(S.tau[0]),                                                      \
                       [i1] "v"(S.tau[1]), [i2] "v"(S.tau[2]), [tb] "s"(tb), [lm] "s"(lm), [la] "v"(la)                  \
                     : REG3_CLOBBERS);                                                                                   \
    }
#else
#define REG_DEFINE_SWEEP3(KMAX) \
    template <int WHO>          \
    __device__ __forceinline__ void reg_sweep(RegState<3> &, const double (&)[2][KMAX], const uint32_t *, uint32_t) {}
#endif
#define REG3_PS_STR "136"
REG_DEFINE_SWEEP3(36)
#undef REG3_PS_STR
#define REG3_PS_STR "264"
REG_DEFINE_SWEEP3(40)
#undef REG3_PS_STR
#define REG3_PS_STR "392"
REG_DEFINE_SWEEP3(44)
#undef REG3_PS_STR
#define REG3_PS_STR "520"
REG_DEFINE_SWEEP3(48)
#undef REG3_PS_STR
static_assert(reg3_ps(36) * 8 == 136 && reg3_ps(40) * 8 == 264 && reg3_ps(44) * 8 == 392 && reg3_ps(48) * 8 == 520, "REG3_PS_STR");

// does the instantiation take its successor list as absolute address pairs (else: 32-bit offsets)?
__host__ __device__ constexpr bool reg_pairs(int KMAX) { return KMAX <= 30; }
// waves per SIMD the register budget of an instantiation is sized for (512 VGPRs per SIMD lane)
#ifndef INSIDER_REG_4WAVE_MAX
#define INSIDER_REG_4WAVE_MAX 20   // largest KMAX built for 4 waves per SIMD (128 VGPRs)
#endif
__host__ __device__ constexpr int reg_waves(int KMAX) { return KMAX <= INSIDER_REG_4WAVE_MAX ? 4 : (KMAX <= 32 ? 3 : 2); }
// smallest instantiated KMAX >= K
__host__ __device__ constexpr int reg_kmax(int K) { return K <= 16 ? 16 : (K <= 32 ? (K + 1) & ~1 : (K + 3) & ~3); }
// register slots of an instantiation (the third slot's Gram columns live in LDS)
__host__ __device__ constexpr int reg_rs(int SLOTS) { return SLOTS < 3 ? SLOTS : 2; }
// the wave's panel of third-slot Gram columns (SLOTS == 3), else nothing
template <int SLOTS, int KMAX>
struct RegPanel {
    static __device__ __forceinline__ double *get() { return nullptr; }
};
template <int KMAX>
struct RegPanel<3, KMAX> {
    static __device__ __forceinline__ double *get()
    {
        __shared__ double panel[KMAX * reg3_ps(KMAX)];
        return panel;
    }
};
// this lane's cell of a panel row: its own third coordinate's, or the row's zero cell
template <int KMAX>
__device__ __forceinline__ int reg3_cell(int lane)
{
    const int i = lane & 15;
    return i < reg3_w(KMAX) ? (lane >> 4) * reg3_w(KMAX) + i : 4 * reg3_w(KMAX);
}

// acc[u] -= sum_{m < K} G[u][m] * v_m  (v in coordinate order); three slots: pc = this lane's cell in row 0 of the panel
template <int KMAX>
__device__ __forceinline__ void reg_gemv3(double (&acc)[3], const double (&v)[3], const double (&G)[2][KMAX], int K,
                                          const double *pc)
{
#define REG_G(M)                                                                                        \
    if constexpr ((M) < KMAX) {                                                                         \
        if ((M) < K) {                                                                                  \
            const double vm = row_bcast<(M) & 15>(v[(M) >> 4]);                                         \
            acc[0] = fma(-vm, G[0][(M)], acc[0]);                                                       \
            acc[1] = fma(-vm, G[1][(M)], acc[1]);                                                       \
            acc[2] = fma(-vm, pc[(M) * reg3_ps(KMAX)], acc[2]);                                         \
        }                                                                                               \
    }
    R16_UNROLL32(REG_G) REG_T1_48(REG_G)
#undef REG_G
}

template <int SLOTS, int KMAX>
__device__ __forceinline__ void reg_gemv(double (&acc)[SLOTS], const double (&v)[SLOTS], const double (&G)[SLOTS][KMAX],
                                         int K)
{
#define REG_G(M)                                                                                        \
    if constexpr ((M) < KMAX) {                                                                         \
        if ((M) < K) {                                                                                  \
            const double vm = row_bcast<(M) & 15>(v[(M) >> 4]);                                         \
            _Pragma("unroll") for (int u = 0; u < SLOTS; ++u) acc[u] = fma(-vm, G[u][(M)], acc[u]);     \
        }                                                                                               \
    }
    R16_UNROLL32(REG_G)
#undef REG_G
}

// LDS doubles per wave: per lane and slot D / (2 la) (D = XtX_kk + l2), the sweep-start beta and what = beta D / (2 la) - y, and the solution
// (values that are not needed inside the sweep live here so that the registers hold only the Gram columns and the
// sweep state)
constexpr int REG_STASH = 6 * 2 * 64;   // D, beta0, w0, solution | two 64-entry windows of |loss change| sums (multi-pass:
                                        // remaining-length estimate) | 1 / D (for the KKT re-admission, :123: no division
                                        // and none of its temporaries inside the sweep loop)
// the same for any slot count: arrays of S = 64 x slots (128 up to two slots) doubles at 0, S, 2S, 3S; the windows at 4S; 1 / D at 4S + 128
__host__ __device__ constexpr int reg_sa(int SLOTS) { return SLOTS < 3 ? 128 : 64 * SLOTS; }
__host__ __device__ constexpr int reg_stash(int SLOTS) { return 5 * reg_sa(SLOTS) + 128; }
static_assert(reg_stash(2) == REG_STASH && reg_stash(1) == REG_STASH, "stash layout");

// The solver, in two parts: cd_reg_begin (start values from q, Gll, the warm start / the saved state; no matrix) and cd_reg
// (the sweeps).  G: columns 16u + i of the row's Gram matrix (zero diagonal).  q, Gll, beta: coordinate 16u + i of
// gene `row`; beta = warm start in (cd_reg_begin), solution out (cd_reg).  gene_ok: the row holds a gene.  stash: this
// wave's REG_STASH doubles of LDS.  cd_reg returns the row's sweep count.
// Loss change of a sweep (:112-114) from per-coordinate start/end values: with g = h - beta XtX_kk (the gradient part
// Xty - XtX beta) the exact change is sum_l [-1/2 db (g0 + g1) + 1/2 l2 (b1^2 - b0^2) + la (|b1| - |b0|)]
// = sum_l [1/2 db (w0 + w1) + la (|b1| - |b0|)],  w = beta (XtX_kk + l2) - h;  the end values of one sweep are the
// start values of the next.  In the solver's scaled units (RegState) w = 2 la (what + 1/2), what = beta D / (2 la) - y, so the
// change is la sum_l [db (what0 + what1 + 1) + |b1| - |b0|].  Lanes without a coordinate carry y = 1/2, beta = 0 and
// contribute nothing.
// Multi-pass (P.sweep_limit / P.start_sweep): with `resume` the row continues a solve that an earlier pass stopped at sweep
// P.start_sweep — hs / is hold its saved h and 1/D-or-0, beta its saved iterate, and nothing is re-derived (the loss
// bookkeeping values beta0 = beta, w0 = beta D - h are the ones the single-pass solve would hold at that sweep), so the
// iterates are bit-identical to an uninterrupted solve.  A row still running when the pass ends at P.sweep_limit returns
// unfinished = true with its (scaled) state in hs / is / beta and `key` = its estimated remaining sweeps (from the geometric decay
// of the loss change over the last two 8-sweep windows; > 0).
// Start values WITHOUT the Gram matrix (so that the kernels can fetch it afterwards: the start values' inputs and the
// 2 x KMAX matrix registers are then never live together): screening, beta, 1/D, h = q (the caller's matrix product
// follows in cd_reg), D and the provisional solution in the stash.
template <int SLOTS>
__device__ __forceinline__ RegState<SLOTS> cd_reg_begin(int K, const double (&q)[SLOTS], const double (&Gll)[SLOTS],
                                                        const double (&beta)[SLOTS], bool gene_ok, const CdParams &P, int lane,
                                                        double *stash, bool resume, const double (&hs)[SLOTS],
                                                        const double (&is)[SLOTS])
{
    const int i = lane & 15;
    const double l2 = P.l2, two_la = P.two_la, gs = P.inv_two_la;   // the scaled state: RegState
    constexpr int SA = reg_sa(SLOTS);
    double *s_d = stash + lane, *s_out = s_d + 3 * SA, *s_ri = s_d + 4 * SA + 128;   // [slot * 64]
    RegState<SLOTS> S;
    if (!resume) {
        // ---- strong rule and start values (:74-80) -----------------------------------------------------------
        double aq = 0.0;
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) aq = fmax(aq, (gene_ok && 16 * u + i < K) ? fabs(q[u]) : 0.0);
        const double thr = P.alpha * (2.0 * P.lambda - row16_max(aq));                    // :74
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            const bool valid = gene_ok && 16 * u + i < K;
            const bool active = valid && !(fabs(q[u]) < thr);
            const double D = (valid ? Gll[u] : 1.0) + l2, rD = cd_rcp(D);
            S.beta[u] = active ? beta[u] : 0.0;                                           // :78
            S.tau[u] = active ? two_la * rD : 0.0;
            S.y[u] = valid ? fma(q[u], gs, 0.5) : 0.5;      // h = q here; cd_reg subtracts offdiag(XtX) beta with the scaled matrix
            s_d[64 * u] = D * gs;
            s_ri[64 * u] = rD;
            s_out[64 * u] = S.beta[u];
        }
    } else {
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {   // the state a limited pass saved: no screening, no re-derivation
            const bool valid = gene_ok && 16 * u + i < K;
            S.beta[u] = valid ? beta[u] : 0.0;
            S.tau[u] = valid ? is[u] : 0.0;
            S.y[u] = valid ? hs[u] : 0.5;
            const double D = (valid ? Gll[u] : 1.0) + l2;
            s_d[64 * u] = D * gs;
            s_ri[64 * u] = cd_rcp(D);
            s_out[64 * u] = S.beta[u];
        }
    }
    return S;
}

// S: the start values of cd_reg_begin; beta / hs / is are outputs here
template <int SLOTS, int KMAX, int WHO>
__device__ __forceinline__ int cd_reg(RegState<SLOTS> S, const double (&G)[reg_rs(SLOTS)][KMAX], int K, double (&beta)[SLOTS],
                                      bool gene_ok, const CdParams &P, int lane, double *stash, bool resume,
                                      double (&hs)[SLOTS], double (&is)[SLOTS], bool &unfinished, int &key, bool &capped,
                                      const double *panel = nullptr)
{
    const int i = lane & 15;
    // the scalars the sweep loop needs, copied out of the kernel-argument tuple: the sweep's assembly clobbers s63-s99,
    // and the compiler otherwise keeps the s_load_dwordx8 result there and restores it (8 v_readlane) twice per sweep
    double la = P.la, tol = P.tol, two_la = P.two_la;   // h = 2 la y - la
    int max_sweeps = P.max_sweeps;
    const uint8_t *order = P.order;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(la), "+s"(tol), "+s"(two_la), "+s"(max_sweeps), "+s"(order));
#endif
    constexpr int SA = reg_sa(SLOTS);
    double *s_d = stash + lane, *s_b = s_d + SA, *s_w = s_d + 2 * SA, *s_out = s_d + 3 * SA;   // [slot * 64]
    double *s_acc = s_d + 4 * SA;                                                        // [window * 64]
    const double *s_ri = s_d + 4 * SA + 128;                                             // [slot * 64]
    s_acc[0] = 0.0;
    s_acc[64] = 0.0;
    unfinished = false;
    capped = false;
    key = 0;
    uint32_t pla = 0;   // three slots: LDS byte address of this lane's cell in row 0 of the panel
    if constexpr (SLOTS == 3) {
        typedef __attribute__((address_space(3))) const double lds_cd;
        const double *pc = panel + reg3_cell<KMAX>(lane);
        pla = (uint32_t)(uintptr_t)(lds_cd *)pc;
        if (!resume) reg_gemv3<KMAX>(S.y, S.beta, G, K, pc);
    } else {
        if (!resume) reg_gemv<SLOTS, KMAX>(S.y, S.beta, G, K);                            // :79 h = q - offdiag(XtX) beta (G scaled)
    }
    // loss bookkeeping in the scaled units too: what = w / (2 la) - 1/2 = beta D / (2 la) - y, w = beta D - h (below)
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        s_b[64 * u] = S.beta[u];
        s_w[64 * u] = fma(S.beta[u], s_d[64 * u], -S.y[u]);
    }

    // Loop control lives in scalar registers: `runm` = lane mask of the genes still running (a wave-uniform integer,
    // changed only in the rarely taken finishing path), the sweep number and the order-table pointer.  Per sweep the
    // control costs one vector compare (the convergence test) and scalar mask arithmetic.
    uint64_t runm = __ballot(gene_ok);
    int sweep = P.start_sweep, my_sweeps = 0;
    int stop = (P.sweep_limit > 0 && P.sweep_limit < max_sweeps) ? P.sweep_limit : max_sweeps;
    // a limited pass sums |loss change| over its last two windows of W sweeps (W = a quarter of the pass, at least 8: the
    // per-sweep change fluctuates with the random coordinate order, a slow decay needs a long window to show)
    int W = (stop - sweep) >> 2;
    W = W < 8 ? 8 : W;
    const int win = stop < max_sweeps ? stop - 2 * W : max_sweeps;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(sweep), "+s"(stop));
#endif
    // the table holds one period of the order sequence: sweep s reads row s mod INSIDER_PERM_PERIOD (include/insider_perm.h)
    const uint32_t *tb0 = reinterpret_cast<const uint32_t *>(order + (SLOTS == 3 ? REG3_ORDER_OFF : REG_ORDER_OFF));
    // ---- genes that may stop (:114-124): wave-uniform, rarely taken ----------------------------------------------------------
    auto candidates = [&](uint64_t cand) {
        // lane masks are formed here, from a laundered lane id, and not kept in registers across the sweeps
        int ln = lane;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(ln));
#endif
        const uint64_t rowmask = 0xffffull << (ln & 48);
        const bool mine = (cand >> ln) & 1ull;
        bool anyv = false;
        if (mine) {
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) {   // :118-119: excluded coordinates have beta = 0, so grad = -h
                const bool viol = gene_ok && 16 * u + i < K && S.tau[u] == 0.0 && fabs(fma(S.y[u], two_la, -la)) > la;
                if (viol) S.tau[u] = two_la * s_ri[64 * u];                             // :123
                anyv = anyv || viol;
            }
        }
        const bool finish = mine && (__ballot(anyv) & rowmask) == 0;                    // :120-121
        if (finish) {   // park the row: zero increments from now on
            my_sweeps = sweep;
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) {
                s_out[64 * u] = S.beta[u];
                S.beta[u] = 0.0;
                S.tau[u] = 0.0;
            }
        }
        runm &= ~__ballot(finish);
    };
    if constexpr (SLOTS < 3 && reg_pairs(KMAX)) {
        // K <= 30: the sweep, the loss change of the sweep (:112-114) and the loop control are ONE asm statement (reg_sweeps); it
        // comes back when a running gene's |loss change| <= tol or at `bound` — the pass limit, or a window edge of a limited
        // pass, whose two sums of |loss change| over W sweeps (the remaining-length estimate below) it accumulates in accw
        typedef __attribute__((address_space(3))) const double lds_cd;
        const uint32_t lds = (uint32_t)(uintptr_t)(lds_cd *)s_d;
        int off = (sweep & (int)(INSIDER_PERM_PERIOD - 1)) * ORDER_ROW;
        double accw = 0.0, dloss = 0.0;
        uint64_t cand = 0;
        while (runm != 0 && sweep < stop) {   // the sweep cap / pass limit is the loop bound: genes still running then are handled below
            int bound = stop;
            if (sweep < win) bound = win < stop ? win : stop;
            else if (sweep < win + W) bound = win + W;
            reg_sweeps<WHO>(S, G, tb0, off, sweep, bound, runm, la, tol, lds, accw, dloss, cand);
            if (sweep == win) accw = 0.0;                          // window 0: sweeps win + 1 ... win + W
            else if (sweep == win + W) { s_acc[0] = accw; accw = 0.0; }   // window 1: the rest of the pass
            if (cand != 0) candidates(cand);
        }
        s_acc[64] = accw;
    } else {
        const uint32_t *tb = tb0 + (size_t)(sweep & (int)(INSIDER_PERM_PERIOD - 1)) * (ORDER_ROW / 4);
        while (runm != 0 && sweep < stop) {   // the sweep cap / pass limit is the loop bound: genes still running then are handled below
            // ---- the sweep (:91-110) -----------------------------------------------------------------------------------
            if constexpr (SLOTS == 3) reg_sweep<WHO>(S, G, tb, pla);
            else if constexpr (!reg_pairs(KMAX)) reg_sweep<WHO>(S, G, tb);
            ++sweep;
            tb = (sweep & (int)(INSIDER_PERM_PERIOD - 1)) ? tb + ORDER_ROW / 4 : tb0;
            // ---- loss change of the sweep (:112-114), per gene ------------------------------------------------------------
            double acc = 0.0, acc1 = 0.0;
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) {
                const double beta0 = s_b[64 * u];
                const double w1 = fma(S.beta[u], s_d[64 * u], -S.y[u]);
                acc = fma(S.beta[u] - beta0, (w1 + s_w[64 * u]) + 1.0, acc);
                acc1 += fabs(S.beta[u]) - fabs(beta0);
                s_b[64 * u] = S.beta[u];
                s_w[64 * u] = w1;
            }
            const double dloss = row16_sum(la * (acc1 + acc));
            if (__builtin_expect(sweep > win, 0)) s_acc[sweep > win + W ? 64 : 0] += fabs(dloss);   // wave-uniform, limited passes only
            const uint64_t cand = __ballot(!(fabs(dloss) > tol)) & runm;                        // :114 genes that may stop now
            if (cand != 0) candidates(cand);
        }
    }
    if ((runm >> lane) & 1ull) {   // stopped by the sweep cap, or by the end of a limited pass
        my_sweeps = sweep;
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) s_out[64 * u] = S.beta[u];
        if (sweep < max_sweeps) {   // to be continued by the next pass
            unfinished = true;
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) { hs[u] = S.y[u]; is[u] = S.tau[u]; }   // the scaled state, as it stands
            // |loss change| decays geometrically: rho^W = b / a over the last two W-sweep windows; sweeps until it reaches tol.
            // Single precision: the estimate only picks a bucket of the next pass's launch order (8 per octave), never a result,
            // and a double-precision log would cost ~20 registers next to the Gram matrix
            const float wa = (float)s_acc[0], wb = (float)s_acc[64], wt = (float)((double)W * tol);
            float est = 1048576.0f;
            if (wa > wb && wb > wt) est = (float)W * __logf(wb / wt) / __logf(wa / wb);
            else if (wb <= wt) est = 1.0f;
            key = (int)fminf(fmaxf(est, 1.0f), 1048576.0f);
        } else {
            capped = true;   // the sweep cap ended the solve, not convergence (the reference loops on, :86-114)
        }
    }
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) beta[u] = (gene_ok && 16 * u + i < K) ? s_out[64 * u] : 0.0;
    return my_sweeps;
}

// ---------------------------------------------------------------------------------------------
// Kernel: column update with the register-resident solver (K <= 32); same contract as k_cd_cols
// ---------------------------------------------------------------------------------------------
// Register discipline (round 3).  The 128-VGPR builds of this kernel computed wrong results twice (KMAX = 22 in round 2,
// KMAX = 20 in round 3) when an unrelated edit moved the register allocation: the allocator had placed the spill STORE of a
// value that is live in all lanes (the gene's row offset j * KP) inside an exec-masked region — the `if (16 + i < K)` body
// that loaded the second coordinate slot — so only the lanes active there saved it; the reload after the sweep loop ran under
// the full mask, the other lanes formed the addresses of their result stores from stale scratch, and the solution (right in
// registers: the loss statistics agreed with the CPU check) went to the wrong place.  Nothing in the sweep assembly is involved.
// The kernel is therefore written so that NOTHING needs spilling and no spill could land in a masked region:
//  * what addresses the gene (row, slot, gene id, record pointer) is recomputed after the loop from a laundered lane id
//    instead of being kept alive across it, and the loss statistics reload q and the diagonal instead of holding them;
//  * every load of the prologue is unconditional (absent genes read gene 0, absent coordinates a valid address; the
//    values are masked), so the prologue has no divergent region at all;
//  * tests/test_lib_cpu.py disassembles the shipped code object and fails on ANY scratch instruction in these kernels
//    (K <= 30) and on any spill store under a narrowed exec mask anywhere in the library (tools/spill_scan.py).
struct RegWho {            // which gene a lane works for
    int i, j;              // coordinate lane inside the row; gene (0 when the row holds none: every pointer stays valid)
    bool gene;             // the row holds a gene
    const double *st;      // its statistics record, or null (tuning == 0; wave-uniform nullness)
};

__device__ __forceinline__ RegWho reg_who(const ColArgs &a, int lane)
{
    RegWho w;
    const int row = lane >> 4, slot = (a.slot_begin ? *a.slot_begin : 0) + blockIdx.x * 4 + row;
    w.i = lane & 15;
    // a resumed pass (multi-pass solve) continues the genes the previous pass left unfinished: the first *pass_count of its
    // order; a split solve's long-gene launch takes the first n_long slots, its majority launch starts at slot n_long
    const int count = a.pass_count ? *a.pass_count : a.p;
    w.gene = slot < a.p && slot < count;
    const int sl = w.gene ? slot : 0;
    const int jj = a.gene_perm ? a.gene_perm[sl] : sl;
    w.j = w.gene ? jj : 0;
    w.st = a.stat ? a.stat + (size_t)w.j * a.stat_len : nullptr;
    return w;
}

// q = X'y over the gene's training entries (src/optimize.cpp:222,235 via level sums, minus the held-out part), XtX_kk
template <int SLOTS>
__device__ __forceinline__ void reg_load_q(const ColArgs &a, const RegWho &w, double (&q)[SLOTS], double (&Gll)[SLOTS])
{
    const int K = a.K, KP = a.KP;
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int c = 16 * u + w.i;                 // < KP: always a valid address
        const bool ok = w.gene && c < K;
        double qv = a.Qfull[(size_t)w.j * KP + c];
        if (w.st) qv -= w.st[stat_index(KP - 1, c)];
        const double d = w.st ? w.st[stat_index(c, c)] : a.RtR[c * KP + c];
        q[u] = ok ? qv : 0.0;
        Gll[u] = ok ? d : 1.0;
    }
}

// SOLVE: the elastic-net solve (a.mode == COL_CD), results and counters only; !SOLVE: the per-gene loss statistics of the
// column as it stands (a.checkpoint; evaluate() / compute_loss(), src/utils.cpp:56-102).  Two kernels instead of one with both
// parts: the statistics need q, the diagonal and two matrix-vector products next to the 2 x KMAX matrix registers, which the
// solve kernel's register budget (set by the sweep loop) does not have; a checkpoint costs one more matrix load per gene
// every tenth outer iteration.
template <int SLOTS, int KMAX, bool SOLVE>
__global__ void __launch_bounds__(64, reg_waves(KMAX)) k_cd_cols_reg(ColArgs a)
{
    const int lane = threadIdx.x;
    if constexpr (SOLVE && reg_pairs(KMAX)) {
        if (a.code_base) {   // probe launch (wave-uniform): where this kernel's table of code blocks lies, for k_order_table
            if (blockIdx.x == 0 && lane == 0) {
                a.code_base[0] = reg_code_base<KMAX, 0>();
                a.code_base[1] = reg_pair_base<KMAX>();
            }
            return;
        }
    }
    const int K = a.K, KP = a.KP;
    const bool resume = a.resume != 0;
    const double gs = a.cd.inv_two_la;
    static_assert(SOLVE || SLOTS < 3, "three slots: the loss statistics come from k_cd_cols_r16<3> (COL_EVAL)");
    __shared__ double stash[reg_stash(SLOTS)];
    double *const panel = RegPanel<SLOTS, KMAX>::get();
    double G[reg_rs(SLOTS)][KMAX], beta[SLOTS], hs[SLOTS], is[SLOTS];
    bool unfinished = false, capped = false;
    int key = 0, sweeps = 0;
    {
        const RegWho w = reg_who(a, lane);
        if (__ballot(w.gene) == 0) return;
        // start values first, the matrix afterwards: their inputs and the 2 x KMAX matrix registers are never live together
        RegState<SLOTS> S;
        {
            double q[SLOTS], Gll[SLOTS];
            reg_load_q<SLOTS>(a, w, q, Gll);
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) {
                const int c = 16 * u + w.i;
                const bool ok = w.gene && c < K;
                const double b = a.C[(size_t)w.j * KP + c];
                beta[u] = ok ? b : 0.0;
                hs[u] = 0.0;
                is[u] = 0.0;
                if (resume) {                                                             // wave-uniform
                    const double hv = a.hsave[(size_t)w.j * KP + c], iv = a.isave[(size_t)w.j * KP + c];
                    hs[u] = ok ? hv : 0.0;
                    is[u] = ok ? iv : 0.0;
                }
            }
            if constexpr (SOLVE) S = cd_reg_begin<SLOTS>(K, q, Gll, beta, w.gene, a.cd, lane, stash, resume, hs, is);
        }
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory");   // the matrix loads stay below the start values
#endif
        // XtX_j = R'R - complement (src/optimize.cpp:218-219: the statistics record holds it ready-made), or the shared R'R
        // (:234); zero diagonal in registers.  One operand stream per element: two would make the allocator spill the matrix.
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            const int c = 16 * u + w.i;
            const bool ok = w.gene && c < K;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                // element (k, c) of the lower-block-stored statistics (stat_index with the block pair known statically)
                const int bk = k >> 4;
                const int si = bk >= u ? (bk * (bk + 1) / 2 + u) * 256 + (k & 15) * 16 + w.i
                                       : (u * (u + 1) / 2 + bk) * 256 + w.i * 16 + (k & 15);
                const double v = w.st ? w.st[si] : a.RtR[k * KP + c];
                const double gv = (ok && k < K && k != c) ? (SOLVE ? v * gs : v) : 0.0;   // the sweeps work on XtX / (2 la): RegState
                if constexpr (SLOTS == 3) {
                    if (u < 2) G[u < 2 ? u : 0][k] = gv;
                    else if (k < K) panel[k * reg3_ps(KMAX) + reg3_cell<KMAX>(lane)] = gv;   // (lanes without a third coordinate: 0 into the row's zero cell)
                } else {
                    G[u][k] = gv;
                }
            }
        }
        if constexpr (SLOTS == 3) wave_sync();
        if constexpr (SOLVE)                                                              // :228,246
            sweeps = cd_reg<SLOTS, KMAX, 0>(S, G, K, beta, w.gene, a.cd, lane, stash, resume, hs, is, unfinished, key, capped, panel);
    }
    // ---- results: everything that addresses the gene is formed again, from a laundered lane id ------------------------
    int lane_c = lane;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(lane_c));
#endif
    const RegWho w = reg_who(a, lane_c);
    if constexpr (SOLVE) {
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
            if (w.gene && 16 * u + w.i < K) {
                a.C[(size_t)w.j * KP + 16 * u + w.i] = beta[u];
                if (unfinished) {
                    a.hsave[(size_t)w.j * KP + 16 * u + w.i] = hs[u];
                    a.isave[(size_t)w.j * KP + 16 * u + w.i] = is[u];
                }
            }
        if (w.gene && w.i == 0) {
            if (unfinished) {
                const int b = cd_bucket(key);
                a.pass_slot[w.j] = ((uint32_t)b << 24) | (uint32_t)atomicAdd(&a.bucket_cnt[b], 1);
            } else {
                if (a.pass_slot) a.pass_slot[w.j] = CD_PASS_DONE;
#ifndef INSIDER_NO_CAP_COUNT   // (diagnostic builds only, tools/k20_variants.sh)
                if (a.cap_hits) {
                    if (capped) atomicAdd(a.cap_hits, 1);
                    if (sweeps > a.cap_hits[1]) atomicMax(a.cap_hits + 1, sweeps);   // longest solve of the call
                }
#endif
                a.sweeps[w.j] = sweeps;
                if (a.sweep_bins) atomicAdd(&a.sweep_bins[blockIdx.x & 255], (unsigned long long)sweeps);
                if (a.sched_key) {   // first half of the next solve's launch order (k_sched_bucket's arithmetic), spread over the solve
                    int k = a.sched_key[w.j];
                    const int s16 = sweeps * 16;
                    k = a.sched_reset ? s16 : (k + s16) / 2;
                    a.sched_key[w.j] = k;
                    const int b = sched_bucket(k, 0);
                    a.sched_bkt[w.j] = (uint16_t)b;
                    a.sched_rank[w.j] = atomicAdd(&a.sched_cnt[b], 1);
                }
            }
        }
        return;
    }
    // ---- loss statistics with the column as it stands: fresh g = q - XtX beta -------------------------------------------
    double q[SLOTS], Gll[SLOTS], g[SLOTS];
    reg_load_q<SLOTS>(a, w, q, Gll);
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) g[u] = (w.gene && 16 * u + w.i < K) ? q[u] - Gll[u] * beta[u] : 0.0;
    if constexpr (SLOTS < 3) reg_gemv<SLOTS, KMAX>(g, beta, G, K);   // (three slots: SOLVE only, this part is dead code there)
    double t_bqg = 0.0, t_b2 = 0.0, t_b1 = 0.0, t_te = 0.0;
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        if (w.gene && 16 * u + w.i < K) {
            t_bqg += beta[u] * (q[u] + g[u]);
            t_b2 += beta[u] * beta[u];
            t_b1 += fabs(beta[u]);
        }
    }
    if (a.test_from_stats && w.st) {
        // sum_test (x - r'b)^2 = sum_held x^2 - 2 b'qc + b'(R'R b) - b'(q - g)      (see k_cd_cols)
        double rb[SLOTS];
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) rb[u] = 0.0;
#define R16_D(M) r16_dense_mv_step<SLOTS, M>(rb, beta, a.RtR, KP, K, w.i, w.gene);
        R16_UNROLL32(R16_D)
#undef R16_D
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            const int c = 16 * u + w.i;
            if (w.gene && c < K) t_te += beta[u] * (rb[u] - (q[u] - g[u]) - 2.0 * w.st[stat_index(KP - 1, c)]);
        }
    }
    const double bqg = row16_sum(t_bqg), sb2 = row16_sum(t_b2), sb1 = row16_sum(t_b1), te = row16_sum(t_te);
    if (w.gene && w.i == 0) {
        a.sse_train[w.j] = a.yy[w.j] - bqg;
        a.b2[w.j] = sb2;
        a.b1[w.j] = sb1;
        if (a.test_from_stats) a.sse_test[w.j] = w.st ? w.st[stat_index(KP - 1, KP - 1)] + te : 0.0;
    }
}

// stand-alone batch form (insider_hip_strong_cd) on dense (XtX, Xty)
// (two waves per SIMD from KMAX = 30 on: the dense problems' per-element addresses need registers the column-update kernel
// does not, and the stand-alone solver is not on the fit's path; no instantiation spills)
template <int SLOTS, int KMAX>
__global__ void __launch_bounds__(64, KMAX >= 30 ? 2 : reg_waves(KMAX))
k_cd_batch_reg(const double *__restrict__ XtX, const double *__restrict__ Xty, const double *__restrict__ wstart, int K,
               int64_t nprob, CdParams cd, double *__restrict__ beta_out, int *__restrict__ sweeps_out,
               unsigned long long *__restrict__ code_base = nullptr)
{
    const int lane = threadIdx.x;
    if constexpr (reg_pairs(KMAX)) {
        if (code_base) {   // probe launch: see k_cd_cols_reg
            if (blockIdx.x == 0 && lane == 0) *code_base = reg_code_base<KMAX, 1>();
            return;
        }
    }
    __shared__ double stash[reg_stash(SLOTS)];
    double *const panel = RegPanel<SLOTS, KMAX>::get();
    double G[reg_rs(SLOTS)][KMAX], beta[SLOTS];
    int sw;
    {
        const int row = lane >> 4, i = lane & 15;
        const int64_t b = (int64_t)blockIdx.x * 4 + row;
        const bool prob = b < nprob;
        const double *Gb = XtX + (size_t)(prob ? b : 0) * K * K;     // absent problems read problem 0: every load unconditional
        const double *qb = Xty + (size_t)(prob ? b : 0) * K, *wb = wstart + (size_t)(prob ? b : 0) * K;
        double hs[SLOTS], is[SLOTS];
        RegState<SLOTS> S;
        {
            double q[SLOTS], Gll[SLOTS];
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) {
                const int c = 16 * u + i;
                const bool ok = prob && c < K;
                const int cc = c < K ? c : 0;
                const double dv = Gb[(size_t)cc * K + cc], qv = qb[cc], wv = wb[cc];
                Gll[u] = ok ? dv : 1.0;
                q[u] = ok ? qv : 0.0;
                beta[u] = ok ? wv : 0.0;
                hs[u] = is[u] = 0.0;
            }
            S = cd_reg_begin<SLOTS>(K, q, Gll, beta, prob, cd, lane, stash, false, hs, is);
        }
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory");   // the matrix loads stay below the start values
#endif
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            const int c = 16 * u + i;
            const bool ok = prob && c < K;
            const int cc = c < K ? c : 0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const double v = Gb[(size_t)(k < K ? k : 0) * K + cc];
                const double gv = (ok && k < K && k != c) ? v * cd.inv_two_la : 0.0;   // the sweeps work on XtX / (2 la): RegState
                if constexpr (SLOTS == 3) {
                    if (u < 2) G[u < 2 ? u : 0][k] = gv;
                    else if (k < K) panel[k * reg3_ps(KMAX) + reg3_cell<KMAX>(lane)] = gv;
                } else {
                    G[u][k] = gv;
                }
            }
        }
        if constexpr (SLOTS == 3) wave_sync();
        bool unfinished, capped;
        int key;
        sw = cd_reg<SLOTS, KMAX, 1>(S, G, K, beta, prob, cd, lane, stash, false, hs, is, unfinished, key, capped, panel);
    }
    int lane_c = lane;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(lane_c));
#endif
    const int row = lane_c >> 4, i = lane_c & 15;
    const int64_t b = (int64_t)blockIdx.x * 4 + row;
    const bool prob = b < nprob;
#pragma unroll
    for (int u = 0; u < SLOTS; ++u)
        if (prob && 16 * u + i < K) beta_out[(size_t)b * K + 16 * u + i] = beta[u];
    if (prob && i == 0 && sweeps_out) sweeps_out[b] = sw;
}

// host-side dispatch over the instantiated (SLOTS, KMAX) pairs: F is a generic lambda taking two integral constants
#define REG_DISPATCH(K, CALL)                                      \
    switch (reg_kmax(K)) {                                         \
        case 16: { constexpr int SL_ = 1, KM_ = 16; CALL; } break; \
        case 18: { constexpr int SL_ = 2, KM_ = 18; CALL; } break; \
        case 20: { constexpr int SL_ = 2, KM_ = 20; CALL; } break; \
        case 22: { constexpr int SL_ = 2, KM_ = 22; CALL; } break; \
        case 24: { constexpr int SL_ = 2, KM_ = 24; CALL; } break; \
        case 26: { constexpr int SL_ = 2, KM_ = 26; CALL; } break; \
        case 28: { constexpr int SL_ = 2, KM_ = 28; CALL; } break; \
        case 30: { constexpr int SL_ = 2, KM_ = 30; CALL; } break; \
        default: { constexpr int SL_ = 2, KM_ = 32; CALL; } break; \
    }
// 32 < K <= 48: three slots
#define REG3_DISPATCH(K, CALL)                                     \
    switch (reg_kmax(K)) {                                         \
        case 36: { constexpr int SL_ = 3, KM_ = 36; CALL; } break; \
        case 40: { constexpr int SL_ = 3, KM_ = 40; CALL; } break; \
        case 44: { constexpr int SL_ = 3, KM_ = 44; CALL; } break; \
        default: { constexpr int SL_ = 3, KM_ = 48; CALL; } break; \
    }
#define REG_ANY_DISPATCH(K, CALL)                                  \
    switch (reg_kmax(K)) {                                         \
        case 16: { constexpr int SL_ = 1, KM_ = 16; CALL; } break; \
        case 18: { constexpr int SL_ = 2, KM_ = 18; CALL; } break; \
        case 20: { constexpr int SL_ = 2, KM_ = 20; CALL; } break; \
        case 22: { constexpr int SL_ = 2, KM_ = 22; CALL; } break; \
        case 24: { constexpr int SL_ = 2, KM_ = 24; CALL; } break; \
        case 26: { constexpr int SL_ = 2, KM_ = 26; CALL; } break; \
        case 28: { constexpr int SL_ = 2, KM_ = 28; CALL; } break; \
        case 30: { constexpr int SL_ = 2, KM_ = 30; CALL; } break; \
        case 32: { constexpr int SL_ = 2, KM_ = 32; CALL; } break; \
        case 36: { constexpr int SL_ = 3, KM_ = 36; CALL; } break; \
        case 40: { constexpr int SL_ = 3, KM_ = 40; CALL; } break; \
        case 44: { constexpr int SL_ = 3, KM_ = 44; CALL; } break; \
        default: { constexpr int SL_ = 3, KM_ = 48; CALL; } break; \
    }

}  // namespace insider
