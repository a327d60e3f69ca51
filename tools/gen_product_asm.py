#!/usr/bin/env python3
"""Generates hmc.jl_amd/csrc/product_asm_k8.inc: the chunk-product loop of the LDS-resident kernel at K = 8 as ONE
inline-assembly statement with fixed physical registers.

What it computes (gibbs_big.hpp, forward filter, local product; src/Hmc.jl:371-440 restated as a matrix product):
    Q = M_0 M_1 ... M_{L-1},   M_l = A diag(f_l),   Q, A 8 x 8, f_l the step's 8 emission values (read back from the
    pdf scratch fscr[l][s][tid] that the pdf pass filled), with the exact power-of-two rescale after steps 7, 15, ...
in exactly the operation order of the C++ loop it replaces (per output: one v_mul, seven v_fma in k order, one v_mul by
f): the results are bit-identical, so every parity / bit-identity test of the C++ form holds for it.

Why assembly: the compiler's version of this loop issues ~930 VALU instructions per step for 576 multiply-adds (AGPR
round trips of the row block, re-materialised constants, lane spills) and runs them at ~5.4 cycles each (the dependent
chains of a dot product are interleaved four deep at best).  Here a step is 576 fp64 instructions + 4 address updates:
  * Q lives in v[0:127] for the whole loop and is updated IN PLACE a block of four rows at a time (row r of Q M needs row
    r of Q only); the block's 32 accumulators v[128:191] are multiplied by f straight back into Q's registers;
  * k-outer order: the 32 multiply-adds of one k are independent of each other (32 chains in flight, the fp64 pipe's full
    rate needs 16);
  * rows 0..3 of A sit in scalar registers s[36:99] for the whole loop (one scalar operand per VALU instruction is what
    gfx9 allows: Q and the accumulator are the vector operands); rows 4..7 come from LDS by four broadcast ds_read_b128
    each into one of two buffers (v[192:207] / v[208:223]), requested three rows ahead.  All eight rows from LDS was
    measured LDS-BOUND: four waves x 64 reads x 8 clocks = the 2 300 clocks the step's multiply-adds take;
  * the step's f comes from the scratch ([L][4][NT][2] doubles) by four global_load_dwordx4 requested one block ahead
    (at the end of the previous step), straight into v[224:239].  (Requested two steps ahead through a ring of AGPRs with
    16 v_accvgpr_read per step: measured 3 % SLOWER -- with one wave per SIMD every instruction of any kind costs its
    ~4.4 ticks of issue, so 16 more instructions outweigh the ~100 ticks of exposed load latency.)
What the loop costs: 576 fp64 + 32 LDS + 4 VMEM + ~24 scalar / wait instructions per step, ~2 900 ticks measured against
~5 000 for the compiler's loop (profiles/r04/).
Registers: v[0:127] Q (outputs), v[128:191] accumulators, v[192:223] A rows 4..7, v[224:239] f, v240-v243 temporaries,
s[36:99] A rows 0..3.
"""
import os
import sys

TEST_NOLDS = "--test-nolds" in sys.argv        # timing experiments only (wrong results): no LDS reads / no scratch loads in the loop
TEST_NOFV = "--test-nofv" in sys.argv
IN_LOOP = False
K = 8
NSR = int(os.environ.get("PRODUCT_NSR", "4"))      # rows of A held in scalar registers (s[SBASE : SBASE + 16 NSR)); 4 measured best
SBASE = 100 - 16 * NSR
QREG = lambda i: "v[%d:%d]" % (2 * i, 2 * i + 1)
ACC = lambda j: "v[%d:%d]" % (128 + 2 * j, 129 + 2 * j)
AB = lambda b, s: "v[%d:%d]" % (192 + 16 * b + 2 * s, 193 + 16 * b + 2 * s)
AS = lambda k, s: "s[%d:%d]" % (SBASE + 16 * k + 2 * s, SBASE + 16 * k + 2 * s + 1)
FV = lambda s: "v[%d:%d]" % (224 + 2 * s, 225 + 2 * s)
VOFF, VM, VE, VT = "v240", "v241", "v242", "v243"

out = []
emit = out.append


def read_row(buf, k):
    """four broadcast ds_read_b128: row k of A (64 bytes) into buffer buf"""
    if TEST_NOLDS and IN_LOOP:
        return
    for h in range(4):
        emit("ds_read_b128 v[%d:%d], %%[lds] offset:%d" % (192 + 16 * buf + 4 * h, 195 + 16 * buf + 4 * h, 64 * k + 16 * h))


def load_f(step, dst4):
    """four global loads (dwordx4): f[0..7] of step `step` (an SGPR name or a number) as its four PAIRS, 4096 bytes apart
    (the scratch is [L][4][NT][2] doubles: 16384 bytes per step), into the register quads dst4(pair).
    VOFF holds the lane's byte offset of the last step loaded and advances only while `step` < L: the loads behind the
    last step re-read it (they are issued unconditionally so that the vmcnt bookkeeping is static; nothing reads them)."""
    if TEST_NOFV and IN_LOOP:
        return
    if step != 0:                                            # (step 0: VOFF is the lane's offset of step 0 already)
        emit("s_cmp_lt_u32 %s, %%[L]" % step)
        emit("s_cselect_b32 %[t], 0x4000, 0")
        emit("v_add_u32 %s, %%[t], %s" % (VOFF, VOFF))
    for pair in range(4):
        base = VOFF
        if pair:
            base = VT
            emit("v_add_u32 %s, 0x%x, %s" % (VT, 0x1000 * pair, VOFF))
        emit("global_load_dwordx4 %s, %s, %%[fb]" % (dst4(pair), base))


FV4 = lambda q: "v[%d:%d]" % (224 + 4 * q, 227 + 4 * q)


def rescale():
    """rescale_pow2<64>(Q) (gibbs_device.hpp): largest biased exponent over the high words, e = 1022 - be when 0 < be < 2040"""
    emit("v_max3_u32 %s, v1, v3, v5" % VM)
    hi = [2 * i + 1 for i in range(3, 64)]
    while len(hi) >= 2:
        emit("v_max3_u32 %s, %s, v%d, v%d" % (VM, VM, hi[0], hi[1]))
        hi = hi[2:]
    if hi:
        emit("v_max_u32 %s, %s, v%d" % (VM, VM, hi[0]))
    emit("v_lshrrev_b32 %s, 20, %s" % (VM, VM))
    emit("v_add_u32 %s, -1, %s" % (VT, VM))
    emit("v_sub_u32 %s, 0x3fe, %s" % (VE, VM))
    emit("v_cmp_gt_u32 vcc, 0x7f7, %s" % VT)
    emit("v_cndmask_b32 %s, 0, %s, vcc" % (VE, VE))
    for i in range(K * K):
        emit("v_ldexp_f64 %s, %s, %s" % (QREG(i), QREG(i), VE))


def fma_row(b, k, src):
    """the 32 multiply-adds of row k of A against the block's four rows of Q; src(s) names A[k][s]"""
    for rr in range(4):
        for s in range(K):
            q = QREG((4 * b + rr) * K + k)
            if k == 0:
                emit("v_mul_f64 %s, %s, %s" % (ACC(rr * K + s), q, src(s)))
            else:
                emit("v_fma_f64 %s, %s, %s, %s" % (ACC(rr * K + s), q, src(s), ACC(rr * K + s)))


# ---- prologue ----
# (the hazard recogniser does not look inside an asm statement: an SGPR operand the compiler has just written with a VALU
#  instruction -- v_readlane of a spilled pair -- needs five wait states before a VMEM instruction reads it)
emit("s_nop 4")
emit("v_mov_b32 %s, %%[voff]" % VOFF)
emit("s_waitcnt vmcnt(0)")                                   # the pdf pass's stores to the scratch have landed
load_f(0, FV4)                                               # f of step 0
# rows 0..NSR-1 of A: through the accumulator registers (and, beyond four rows, the buffers) into scalar registers
STAGE = lambda i: 128 + i if i < 64 else 192 + (i - 64)      # v[128:191], then v[192:207]
for k in range(NSR):
    for h in range(4):
        emit("ds_read_b128 v[%d:%d], %%[lds] offset:%d" % (STAGE(16 * k + 4 * h), STAGE(16 * k + 4 * h) + 3, 64 * k + 16 * h))
emit("s_waitcnt lgkmcnt(0)")
for i in range(16 * NSR):
    emit("v_readfirstlane_b32 s%d, v%d" % (SBASE + i, STAGE(i)))
emit("s_waitcnt vmcnt(0)")                                   # f of step 0
# Q = A diag(f_0): the product of the identity with M_0, exactly
for r in range(NSR):
    for s in range(K):
        emit("v_mul_f64 %s, %s, %s" % (QREG(r * K + s), AS(r, s), FV(s)))
for r in range(NSR, K):
    read_row(r & 1, r)
    emit("s_waitcnt lgkmcnt(0)")
    for s in range(K):
        emit("v_mul_f64 %s, %s, %s" % (QREG(r * K + s), AB(r & 1, s), FV(s)))
emit("s_mov_b32 %[l], 1")
emit("s_cmp_lt_u32 1, %[L]")
emit("s_cbranch_scc0 .Lhmcg_prod_done_%=")
load_f(1, FV4)                                               # f of step 1
read_row(0, NSR)                                             # stream positions 0 and 1
read_row(1, NSR + 1 % (K - NSR))


def step_body(slot):
    """one step (`slot` only numbers the labels).  The rows NSR..7 of A form a cyclic stream through the two LDS buffers:
    stream position p (row NSR + p mod n) sits in buffer p & 1 and is requested two positions ahead, right after the
    multiply-adds of position p - 2 have been issued; a step is 2 n positions (even: the pattern repeats every step)."""
    n = K - NSR
    for b in range(2):
        for k in range(NSR):                                 # scalar operands: nothing to wait for
            fma_row(b, k, lambda s, k=k: AS(k, s))
        for i in range(n):
            pos = b * n + i
            emit("s_waitcnt lgkmcnt(4)")                     # this position's row (the next one may still be on its way)
            fma_row(b, NSR + i, lambda s, pos=pos: AB(pos & 1, s))
            read_row(pos & 1, NSR + (pos + 2) % n)           # two positions ahead (after the last step: drained below)
        if b == 0:
            emit("s_waitcnt vmcnt(0)")                       # this step's f (requested at the end of the previous step)
        for rr in range(4):
            for s in range(K):
                emit("v_mul_f64 %s, %s, %s" % (QREG((4 * b + rr) * K + s), ACC(rr * K + s), FV(s)))
    emit("s_add_u32 %[u], %[l], 1")
    load_f("%[u]", FV4)                                      # the next step's f straight into its registers
    emit("s_add_u32 %[l], %[l], 1")
    # rescale after steps 7, 15, ... (l has been advanced: l & 7 == 0)
    emit("s_and_b32 %[t], %[l], 7")
    emit("s_cmp_eq_u32 %[t], 0")
    emit("s_cbranch_scc0 .Lhmcg_prod_nores%d_%%=" % slot)
    rescale()
    emit(".Lhmcg_prod_nores%d_%%=:" % slot)
    emit("s_cmp_lt_u32 %[l], %[L]")


emit(".Lhmcg_prod_loop_%=:")
IN_LOOP = True
step_body(0)
emit("s_cbranch_scc1 .Lhmcg_prod_loop_%=")
emit(".Lhmcg_prod_done_%=:")
emit("s_waitcnt vmcnt(0) lgkmcnt(0)")                        # the ring's loads and the rows requested ahead of steps that do not come
emit("s_nop 4")                                              # ... nothing of ours is in flight when the compiler's code resumes

path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hmc.jl_amd", "csrc", "product_asm_k8.inc")
if TEST_NOLDS or TEST_NOFV:
    path = "/tmp/t/product_asm_test.inc"
with open(path, "w") as f:
    f.write("// GENERATED by tools/gen_product_asm.py -- do not edit.  The K = 8 chunk-product loop as one asm statement.\n")
    f.write("// Operands: outputs Q[0..63] bound to v[0:127]; %[l], %[t] scalar temporaries; inputs %[L] (steps), %[fb] (scratch\n")
    f.write("// base of the window, SGPR pair), %[voff] (lane byte offset), %[lds] (LDS byte address of A, row-major).\n")
    f.write("asm volatile(\n")
    for ln in out:
        f.write('    "%s\\n"\n' % ln)
    f.write("    : " + ", ".join('"=&{v[%d:%d]}"(Q[%d])' % (2 * i, 2 * i + 1, i) for i in range(K * K)) + ",\n")
    f.write('      [l] "=&s"(asm_l), [t] "=&s"(asm_t), [u] "=&s"(asm_u)\n')
    f.write('    : [L] "s"(L), [fb] "s"(asm_fb), [voff] "v"(asm_voff), [lds] "v"(asm_lds)\n')
    clob = ['"v%d"' % i for i in range(128, 244)] + ['"s%d"' % i for i in range(SBASE, SBASE + 16 * NSR)] + ['"vcc"', '"scc"', '"memory"']
    f.write("    : " + ", ".join(clob) + ");\n")
print("wrote", path, len(out), "instructions")
