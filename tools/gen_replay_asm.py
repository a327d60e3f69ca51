#!/usr/bin/env python3
"""Generates hmc.jl_amd/csrc/replay_asm_k8.inc: the filter replay of the LDS-resident kernel at K = 8 (gibbs_big.hpp:
the normalised forward recursion of forwardupdate_P!, src/Hmc.jl:413-432, fused with the state maps of update_X,
:459-484) as ONE inline-assembly statement with fixed physical registers.  Companion of tools/gen_product_asm.py.

Per step t (each lane walks its own L consecutive steps; av = pif[t-1,:] enters, av = pif[t,:] leaves):
    c[s][r] = sum_{r' <= r} av[r'] A[r'][s]         eight chains of 1 v_mul + 7 v_fma, in r order (the C++ loop's order)
    thr[s]  = u c[s][7],  nv[s] = c[s][7] f[s]      u = the step's uniform, f = its emission values (pdf scratch)
    idx[s]  = #{r < 7 : c[s][r] <= thr[s]}           the categorical draw X[t-1] | X[t] = s for the pre-drawn u
    total   = nv[0] + ... + nv[7] (in s order),  inv = rcp(total) with one Newton step (rcp_fast)
    av[s]   = nv[s] inv;  entries with !(av[s] > eps) take the uniform draw floor(8 u)   (guard, :472-480)
    maps[t-1] = the eight 4-bit entries;  pif[T-1,:] is left in th.pi_end by the lane that owns step T-1
The same operations in the same order as the C++ loop (count_le_sorted_batch's bisection and the linear count used here
are the same integer), so the results are bit-identical; the C++ loop stays for the calls that also want pif per step
(last sweep with extras.pif_final, the smoothing variants) and for the signal / streaming forms.

Why assembly: the compiler's loop is 384 instructions per step (97 v_cndmask for the bisection's 64-bit selects, 24
moves, 32 LDS reads of A, 21 scalar, 18 waits / nops); with one wave per SIMD every instruction of any kind costs ~4.4
ticks.  Here a step is ~275: the count is 7 compares + 7 add-with-carry per state (no selects), A rows 0..3 are scalar
operands and rows 4..7 stay in 64 VGPRs for the whole loop (no LDS reads of A at all), the nibble merge is one v_bfi.

Registers: v[0:15] av (in/out), v[16:79] A rows 4..7, v[80:95] nv, v[96:111] / v[112:127] f of this / the next step
(the loop is unrolled twice, the two sets swap roles), v[128:191] the chains of four states at a time, v[192:199] thr,
v[200:207] idx accumulators, v208.. temporaries; s[36:99] A rows 0..3, s[20:35] compare masks.
"""
import os
import sys
EPS_FAST = os.environ.get("REPLAY_EPS_FAST", "0") == "1"     # min-of-eight fast path for the eps() guard: measured 0.8 % SLOWER (a dependent chain of seven v_min)
K = 8
SBASE = 36
AV = lambda r: "v[%d:%d]" % (2 * r, 2 * r + 1)
AROW = lambda r, s: "v[%d:%d]" % (16 + 16 * (r - 4) + 2 * s, 17 + 16 * (r - 4) + 2 * s)       # r = 4..7
AS = lambda r, s: "s[%d:%d]" % (SBASE + 16 * r + 2 * s, SBASE + 16 * r + 2 * s + 1)           # r = 0..3
NV = lambda s: "v[%d:%d]" % (80 + 2 * s, 81 + 2 * s)
FVB = lambda b, s: "v[%d:%d]" % (96 + 16 * b + 2 * s, 97 + 16 * b + 2 * s)
FV4 = lambda b, q: "v[%d:%d]" % (96 + 16 * b + 4 * q, 99 + 16 * b + 4 * q)
C = lambda j, r: "v[%d:%d]" % (128 + 16 * j + 2 * r, 129 + 16 * j + 2 * r)                    # j = 0..3: state within the group
THR = lambda j: "v[%d:%d]" % (192 + 2 * j, 193 + 2 * j)
IDX = lambda s: "v%d" % (200 + s)
TOT, INV, TMP = "v[208:209]", "v[210:211]", "v[212:213]"
U = "v[214:215]"
MOK, FAIL, IU, T1, ZERO = "v216", "v217", "v218", "v219", "v220"
AU, AM, VOFF, VT, TCUR = "v221", "v222", "v223", "v224", "v225"
_MASKS = ["s[20:21]", "s[22:23]", "s[24:25]", "s[26:27]", "s[28:29]", "s[30:31]", "s[34:35]", "s[16:17]"]     # (s32 / s33 are reserved)
MASK = lambda i: _MASKS[i]                                                                    # i = 0..7
NIB = lambda s: "v%d" % (226 + s)                                                             # 0xF << 4 s

out = []
emit = out.append


def load_f(step, b):
    """the four pairs of f of step `step` (scalar register) into buffer b; the offset advances only while step < L"""
    emit("s_cmp_lt_u32 %s, %%[L]" % step)
    emit("s_cselect_b32 %[t], 0x4000, 0")
    emit("v_add_u32 %s, %%[t], %s" % (VOFF, VOFF))
    for pair in range(4):
        base = VOFF
        if pair:
            base = VT
            emit("v_add_u32 %s, 0x%x, %s" % (VT, 0x1000 * pair, VOFF))
        emit("global_load_dwordx4 %s, %s, %%[fb]" % (FV4(b, pair), base))


def a_op(r, s):
    return AS(r, s) if r < 4 else AROW(r, s)


def step_body(b, tag, incm):
    """one step with its f in buffer b; requests the f of the step after next into the same buffer at its end.
    tag names the labels; incm is what the map address advances by after this step's write"""
    emit("s_waitcnt vmcnt(4) lgkmcnt(0)")                     # this step's f (the next step's four loads may be in flight) and u
    for g in range(2):
        ss = [4 * g + j for j in range(4)]
        for r in range(K):                                   # four chains interleaved
            for j, s in enumerate(ss):
                if r == 0:
                    emit("v_mul_f64 %s, %s, %s" % (C(j, 0), AV(0), a_op(0, s)))
                else:
                    emit("v_fma_f64 %s, %s, %s, %s" % (C(j, r), AV(r), a_op(r, s), C(j, r - 1)))
        for j, s in enumerate(ss):
            emit("v_mul_f64 %s, %s, %s" % (THR(j), U, C(j, 7)))
        for j, s in enumerate(ss):
            emit("v_mul_f64 %s, %s, %s" % (NV(s), C(j, 7), FVB(b, s)))
        # idx[s] = number of r < 7 with c[s][r] <= thr[s]: compare into a scalar mask, add the mask as a carry
        for r in range(K - 1):
            for j, s in enumerate(ss):
                emit("v_cmp_le_f64 %s, %s, %s" % (MASK(j + 4 * (r & 1)), C(j, r), THR(j)))
            for j, s in enumerate(ss):
                if r == 0:
                    emit("v_cndmask_b32 %s, 0, 1, %s" % (IDX(s), MASK(j + 4 * (r & 1))))          # the first term starts the count
                else:
                    emit("v_addc_co_u32 %s, vcc, 0, %s, %s" % (IDX(s), IDX(s), MASK(j + 4 * (r & 1))))
        for j, s in enumerate(ss):
            if s == 0:
                emit("v_mov_b32 %s, %s" % (MOK, IDX(0)))
            else:
                emit("v_lshl_or_b32 %s, %s, %d, %s" % (MOK, IDX(s), 4 * s, MOK))
    # total in s order, its reciprocal (rcp_fast), the uniform-law index floor(8 u)
    emit("v_add_f64 %s, %s, %s" % (TOT, NV(0), NV(1)))
    for s in range(2, K):
        emit("v_add_f64 %s, %s, %s" % (TOT, TOT, NV(s)))
    emit("v_ldexp_f64 %s, %s, 3" % (TMP, U))                  # 8 u, exactly
    emit("v_cvt_i32_f64 %s, %s" % (IU, TMP))
    emit("v_cmp_ngt_f64 vcc, %s, 0" % TOT)
    emit("s_cbranch_vccnz .Lhmcg_rep_rare%s_%%=" % tag)
    emit(".Lhmcg_rep_back%s_%%=:" % tag)
    emit("v_rcp_f64 %s, %s" % (INV, TOT))
    # gfx940 hazard: the result of a transcendental instruction (v_rcp_f64) must not be read by the NEXT vector instruction
    # (the compiler inserts the wait state by itself; nobody does inside an asm statement -- lanes 0 and 2 of every eight
    # read a stale value): three independent instructions stand in between
    emit("v_lshl_or_b32 %s, %s, 4, %s" % (IU, IU, IU))        # floor(8 u) in every nibble
    emit("v_lshl_or_b32 %s, %s, 8, %s" % (IU, IU, IU))
    emit("v_fma_f64 %s, -%s, %s, 1.0" % (TMP, TOT, INV))
    emit("v_lshl_or_b32 %s, %s, 16, %s" % (IU, IU, IU))
    emit("v_fma_f64 %s, %s, %s, %s" % (INV, TMP, INV, INV))
    for s in range(K):
        emit("v_mul_f64 %s, %s, %s" % (AV(s), NV(s), INV))   # pif[t,s]
    # guard (:472-480): an entry whose pif[t,s] is not > eps() takes the uniform draw.  Almost never: the smallest of the
    # eight against eps() first (no NaN can be here: a NaN anywhere makes the total NaN, which took the rare path above), and
    # the per-state masks only in a wave where some lane fails
    if EPS_FAST:
        emit("v_min_f64 %s, %s, %s" % (TMP, AV(0), AV(1)))
        for s in range(2, K):
            emit("v_min_f64 %s, %s, %s" % (TMP, TMP, AV(s)))
        emit("v_cmp_nlt_f64 vcc, s[18:19], %s" % TMP)        # !(eps < min): some entry of this lane fails
        emit("s_cbranch_vccnz .Lhmcg_rep_fail%s_%%=" % tag)
    else:
        emit("v_mov_b32 %s, 0" % FAIL)
        for s in range(K):
            emit("v_cmp_lt_f64 vcc, s[18:19], %s" % AV(s))
            emit("v_cndmask_b32 %s, %s, %s, vcc" % (T1, NIB(s), ZERO))
            emit("v_or_b32 %s, %s, %s" % (FAIL, FAIL, T1))
        emit("v_bfi_b32 %s, %s, %s, %s" % (MOK, FAIL, IU, MOK))
    if EPS_FAST:
        emit(".Lhmcg_rep_merged%s_%%=:" % tag)
    emit("ds_write_b32 %s, %s" % (AM, MOK))                  # g_{t-1}
    emit("v_add_u32 %s, %s, %s" % (AM, incm, AM))
    # pif[T-1,:] -> th.pi_end, by the lane that owns step T-1 (only its wave takes this path)
    emit("s_cmp_eq_u32 %[ownw], 0")
    emit("s_cbranch_scc1 .Lhmcg_rep_noown%s_%%=" % tag)
    emit("v_cmp_eq_u32 vcc, %%[l], %s" % "%[lown]")
    emit("s_and_saveexec_b64 %s, vcc" % MASK(0))
    for s in range(K):
        emit("ds_write_b64 %%[pie], %s offset:%d" % (AV(s), 8 * s))
    emit("s_mov_b64 exec, %s" % MASK(0))
    emit(".Lhmcg_rep_noown%s_%%=:" % tag)
    # the next step's uniform, the f of the step after next
    emit("ds_read_b64 %s, %s" % (U, AU))
    emit("v_add_u32 %s, 8, %s" % (AU, AU))
    emit("s_add_u32 %[u], %[l], 2")
    load_f("%[u]", b)
    emit("s_add_u32 %[l], %[l], 1")
    emit("s_cmp_lt_u32 %[l], %[L]")


# ---- prologue ----
emit("s_nop 4")                                              # (see gen_product_asm.py: no hazard checks inside an asm statement)
emit("v_mov_b32 %s, %%[voff]" % VOFF)
emit("v_mov_b32 %s, %%[au]" % AU)
emit("v_mov_b32 %s, %%[am]" % AM)
emit("v_mov_b32 %s, 0" % ZERO)
for s_ in range(K):
    emit("v_mov_b32 %s, 0x%x" % (NIB(s_), (0xF << (4 * s_)) & 0xFFFFFFFF))
for pair in range(4):                                        # f of step 0
    base = VOFF
    if pair:
        base = VT
        emit("v_add_u32 %s, 0x%x, %s" % (VT, 0x1000 * pair, VOFF))
    emit("global_load_dwordx4 %s, %s, %%[fb]" % (FV4(0, pair), base))
load_f(1, 1)                                                 # f of step 1 (re-reads step 0 when L == 1)
# A: rows 0..3 through v[128:191] into s[36:99], rows 4..7 into v[16:79]
for k in range(4):
    for h in range(4):
        emit("ds_read_b128 v[%d:%d], %%[lds] offset:%d" % (128 + 16 * k + 4 * h, 131 + 16 * k + 4 * h, 64 * k + 16 * h))
for k in range(4, 8):
    for h in range(4):
        emit("ds_read_b128 v[%d:%d], %%[lds] offset:%d" % (16 + 16 * (k - 4) + 4 * h, 19 + 16 * (k - 4) + 4 * h, 64 * k + 16 * h))
emit("ds_read_b64 %s, %s" % (U, AU))                          # u of step 0
emit("v_add_u32 %s, %%[incu], %s" % (AU, AU))
emit("s_waitcnt lgkmcnt(0)")
emit("s_mov_b32 s18, 0")                                     # eps() = 2^-52
emit("s_mov_b32 s19, 0x3cb00000")
for i in range(64):
    emit("v_readfirstlane_b32 s%d, v%d" % (SBASE + i, 128 + i))
emit("s_mov_b32 %[l], 0")
# step 0 by itself: the map address advances by the lane's first-step increment (0 for the thread that owns step 0 of the
# window -- it has no g_{-1} to write: its first write lands on g_0's word and the next step overwrites it -- else 4)
step_body(0, "p", "%[incm]")
emit("s_cbranch_scc0 .Lhmcg_rep_done_%=")
emit(".Lhmcg_rep_loop_%=:")
step_body(1, "1", "4")
emit("s_cbranch_scc0 .Lhmcg_rep_done_%=")
step_body(0, "0", "4")
emit("s_cbranch_scc1 .Lhmcg_rep_loop_%=")
emit("s_branch .Lhmcg_rep_done_%=")
# ---- rare: some lane's total is not > 0 (every pdf of the step underflowed against its prefix): the uniform law, flagged ----
for b in (("p", "1", "0") if EPS_FAST else ()):
    emit(".Lhmcg_rep_fail%s_%%=:" % b)
    emit("v_mov_b32 %s, 0" % FAIL)
    for s in range(K):
        emit("v_cmp_lt_f64 vcc, s[18:19], %s" % AV(s))       # eps() < pif[t,s]
        emit("v_cndmask_b32 %s, %s, %s, vcc" % (T1, NIB(s), ZERO))
        emit("v_or_b32 %s, %s, %s" % (FAIL, FAIL, T1))
    emit("v_bfi_b32 %s, %s, %s, %s" % (MOK, FAIL, IU, MOK))  # (fail & uniform) | (~fail & idx)
    emit("s_branch .Lhmcg_rep_merged%s_%%=" % b)
for b in ("p", "1", "0"):
    emit(".Lhmcg_rep_rare%s_%%=:" % b)
    emit("s_and_saveexec_b64 %s, vcc" % MASK(0))
    for s in range(K):
        emit("v_mov_b32 v%d, 0" % (80 + 2 * s))
        emit("v_mov_b32 v%d, 0x3fc00000" % (81 + 2 * s))      # 1/8
    emit("v_mov_b32 v208, 0")
    emit("v_mov_b32 v209, 0x3ff00000")                       # total = 1
    emit("v_add_u32 %s, %%[l], %%[t0]" % TCUR)
    emit("v_cmp_gt_i32 vcc, %%[T], %s" % TCUR)                # t < T: the window is flagged
    emit("v_cndmask_b32 %s, 0, %s, vcc" % (T1, "%[flagv]"))
    emit("v_or_b32 %[st], %[st], " + T1)
    emit("s_mov_b64 exec, %s" % MASK(0))
    emit("s_branch .Lhmcg_rep_back%s_%%=" % b)
emit(".Lhmcg_rep_done_%=:")
emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
emit("s_nop 4")

path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hmc.jl_amd", "csrc", "replay_asm_k8.inc")
with open(path, "w") as f:
    f.write("// GENERATED by tools/gen_replay_asm.py -- do not edit.  The K = 8 filter replay as one asm statement.\n")
    f.write("asm volatile(\n")
    for ln in out:
        f.write('    "%s\\n"\n' % ln)
    f.write("    : " + ", ".join('"+{v[%d:%d]}"(av[%d])' % (2 * i, 2 * i + 1, i) for i in range(K)) + ",\n")
    f.write('      [st] "+v"(st), [l] "=&s"(asm_l), [t] "=&s"(asm_t), [u] "=&s"(asm_u)\n')
    f.write('    : [L] "s"(L), [fb] "s"(asm_fb), [voff] "v"(asm_voff), [lds] "v"(asm_lds), [au] "v"(asm_au), [am] "v"(asm_am),\n')
    f.write('      [incu] "v"(asm_incu), [incm] "v"(asm_incm), [ownw] "s"(asm_ownw), [lown] "v"(asm_lown), [pie] "v"(asm_pie),\n')
    f.write('      [t0] "v"(t0), [T] "s"(T), [flagv] "v"(asm_flag)\n')
    clob = (['"v%d"' % i for i in range(16, 234)] + ['"s%d"' % i for i in list(range(16, 32)) + list(range(34, 100))] +
            ['"vcc"', '"scc"', '"memory"'])
    f.write("    : " + ", ".join(clob) + ");\n")
print("wrote", path, len(out), "instructions")
