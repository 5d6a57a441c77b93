#!/usr/bin/env python3
"""Generator of the hand-placed instruction streams of the pipelined attention-backward kernels (csrc/attention_pl.hip.h).

Why a generator: hipcc schedules an attention tile as  scores -> softmax -> second product  per wave; an in-order wave then runs its
matrix instructions and its VALU instructions one after the other, and the VALU issue port (one instruction per ~4.4 cycles and SIMD,
whichever wave it comes from; an MFMA takes ~12 cycles of it) is what bounds these loops at head_dim 64.  The streams written here are
software pipelines over 32 x 32 score sub-tiles ("units"): in every slot the wave issues the score MFMAs of unit k+1, the exp2 / dS VALU
work of unit k and the gradient MFMAs of unit k-1, interleaved instruction by instruction, with the minimum number of VALU instructions
per score element -- every instruction is its own `asm volatile` statement, so hipcc allocates registers but the written order IS the
issue order (the technique of nt_run_k_pipe16 in gemm.hip).

The emitted .inc files are plain C++ statement lists that are #included inside the kernels' tile loops; the names they use (S, DP, kf, ...)
are the kernels' local variables.  Regenerate with  `python tools/gen_attn_pl.py`  (the files are committed; the build does not run this).

Variants (schedule knobs) are separate files so that one binary can hold several and the lab harness can time them against each other.
"""
from __future__ import annotations

import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "finetrainers_amd", "csrc")   # shipped streams
EXP_OUT = os.path.join(HERE, "experimental")                 # streams of kernels that exist in lab / experimental builds only


def _out_dir(name):
    """Streams that only a -DFTMI_LAB build compiles (the ablations a_*, results wrong on purpose, and the packed-VALU variant v6) are not product source:
    they live next to the other research code in tools/experimental/."""
    return EXP_OUT if name.startswith("a_") or name == "v6" else OUT

MFMA = "v_mfma_f32_32x32x16_bf16"


class Stream:
    def __init__(self):
        self.lines: list[str] = []

    def emit(self, s: str):
        self.lines.append(s)

    def asm(self, text: str, outs: str = "", ins: str = "", clob: str = ""):
        parts = [f'"{text}"', outs, ins]
        if clob:
            parts.append(clob)
        else:
            while len(parts) > 1 and parts[-1] == "":
                parts.pop()
        self.lines.append("asm volatile(" + " : ".join(parts) + ");")


def spread(n_items: int, n_gaps: int, weights=None) -> list[int]:
    """n_items over n_gaps as evenly as possible (optionally by weight); returns the count per gap."""
    if weights is None:
        weights = [1.0] * n_gaps
    tot = sum(weights)
    out, acc, given = [], 0.0, 0
    for w in weights:
        acc += w / tot * n_items
        k = int(round(acc)) - given
        out.append(k)
        given += k
    assert sum(out) == n_items, (out, n_items)
    return out


# ------------------------------------------------------------------------------------------------------------------------------
# dQ kernel: a wave owns 32 * nq query rows (qt = 0 .. nq-1), loops over 64-key tiles (js = 0, 1: 32 keys each).
# unit u = (t, js, qt), parity p = (unit index in the tile) & 1.  Slot of unit u:
#   C(u-1): 4 MFMAs  dqt[qt'] += K^T . dS[p^1];   A(u+1): 8 MFMAs  S[p^1] = K.Q^T, DP[p^1] = V.dO^T (- delta);
#   B(u): VALU  dS[p] = exp2(S[p] * sl - lse) * DP[p] -> bf16 fragments.
# ------------------------------------------------------------------------------------------------------------------------------
def dq_valu_ops(par: int, qt: int, exact: bool, order: str) -> list[tuple[str, str, str]]:
    """The VALU list of B(u) in issue order; (text, outs, ins) per instruction.
    order "g16" / "g4": passes over groups of 16 / 4 score elements (all fma, all exp2, all mul, the packs);
    order "roll": a rolling pipeline (fma of element s, exp2 of element s-4, mul of element s-8, pack of a finished pair) -- the exp2s
    (transcendental unit, ~7 cycles each) never come back to back.
    hipcc's hazard recognizer counts an asm statement as zero wait states: it puts one s_nop in front of a statement that reads a register
    defined by an earlier asm statement unless a real instruction sits between them (group 4: 12 per slot, group 16: 3, rolling: ~5)."""
    S, DP = f"S[{par}]", f"DP[{par}]"

    def F(r):
        return ("v_fma_f32 %0, %1, %2, %3", f'"=v"(x[{r}])', f'"v"({S}[{r}]), "v"(sl), "v"(nlse[{qt}])')

    def E(r):
        return ("v_exp_f32 %0, %0", f'"+v"(x[{r}])', "")

    def U(r):
        return ("v_sub_f32 %0, %1, %2", f'"=v"(y[{r}])', f'"v"({DP}[{r}]), "v"(del[{qt}])')

    def M(r):
        if exact:
            return ("v_mul_f32 %0, %0, %1", f'"+v"(x[{r}])', f'"v"(y[{r}])')
        return ("v_mul_f32 %0, %0, %1", f'"+v"(x[{r}])', f'"v"({DP}[{r}])')

    def P(r):  # pair (r, r+1), r even
        hh, e = r >> 3, (r & 7) >> 1
        return ("v_cvt_pk_bf16_f32 %0, %1, %2", f'"=v"(dsw[{par}][{hh}][{e}])', f'"v"(x[{r}]), "v"(x[{r + 1}])')

    ops = []
    if order == "pk":
        # packed fp32 VALU on aligned register pairs (v_pk_fma_f32 / v_pk_mul_f32: two scores per instruction): 40 instead of 56 instructions per unit.
        # Same arithmetic per element (a packed op rounds each half like the scalar op), so the results are the bits of the scalar streams.
        assert not exact
        def FP(r): return ("v_pk_fma_f32 %0, %1, %2, %3", f'"=v"(x2[{r >> 1}])', f'"v"(__builtin_shufflevector({S}, {S}, {r}, {r + 1})), "v"(sl2), "v"(nlse2[{qt}])')
        def EL(r): return ("v_exp_f32 %0, %1", f'"=v"(x[{r}])', f'"v"(x2[{r >> 1}][{r & 1}])')
        def MP(r): return ("v_pk_mul_f32 %0, %1, %2", f'"=v"(y2[{r >> 1}])', f'"v"(f32x2{{x[{r}], x[{r + 1}]}}), "v"(__builtin_shufflevector({DP}, {DP}, {r}, {r + 1}))')
        def PK(r):
            hh, e = r >> 3, (r & 7) >> 1
            return ("v_cvt_pk_bf16_f32 %0, %1, %2", f'"=v"(dsw[{par}][{hh}][{e}])', f'"v"(y2[{r >> 1}][0]), "v"(y2[{r >> 1}][1])')
        L = 2
        for s in range(8 + 3 * L + 1):
            if s < 8: ops.append(FP(2 * s))
            if 0 <= s - L < 8: ops += [EL(2 * (s - L)), EL(2 * (s - L) + 1)]
            if 0 <= s - 2 * L < 8: ops.append(MP(2 * (s - 2 * L)))
            if 0 <= s - 3 * L < 8: ops.append(PK(2 * (s - 3 * L)))
        assert len(ops) == 40, len(ops)
        return ops
    if order.startswith("g"):
        group = int(order[1:])
        for m in range(16 // group):
            rs = [group * m + i for i in range(group)]
            ops += [F(r) for r in rs] + [E(r) for r in rs]
            if exact:
                ops += [U(r) for r in rs]
            ops += [M(r) for r in rs] + [P(r) for r in rs[::2]]
    else:
        L = 4
        for s in range(16 + 2 * L + 2):
            if s < 16:
                ops.append(F(s))
            if 0 <= s - L < 16:
                ops.append(E(s - L))
            if exact and 0 <= s - 2 * L + 2 < 16:
                ops.append(U(s - 2 * L + 2))
            if 0 <= s - 2 * L < 16:
                ops.append(M(s - 2 * L))
            r = s - 2 * L - 1
            if 0 <= r < 16 and r % 2 == 1:
                ops.append(P(r - 1))
    assert len(ops) == (72 if exact else 56), len(ops)
    return ops


def gen_dq(name: str, nq: int, exact: bool, order: str = "roll", weights=None, drop=()):
    """drop: ablation builds for the lab (results wrong on purpose): 'valu', 'lds', 'dma', 'mfma'."""
    s = Stream()
    s.emit(f"// GENERATED by tools/gen_attn_pl.py (dq{nq}_{name}: exact={int(exact)} order={order} drop={','.join(drop) or '-'}) -- do not edit")
    units = [(js, qt) for js in range(2) for qt in range(nq)]
    n = len(units)
    # two waves per SIMD share 512 registers: hipcc halves a 256-register budget into 128 VGPRs + 128 AGPRs as soon as a kernel touches an AGPR,
    # and the stream needs ~200 VGPRs -- so the nq = 1 kernel keeps its accumulators in VGPRs too
    acc = "v" if nq == 1 else "a"
    for i, (js, qt) in enumerate(units):
        par, parn = i & 1, (i & 1) ^ 1
        jsn, qtn = units[(i + 1) % n]
        jsp, qtp = units[(i - 1) % n]
        s.emit(f"// ---- slot {i} (js {js}, qt {qt}): C(u-1) on dqt[{qtp}], A(u+1) into S/DP[{parn}] (js {jsn}, qt {qtn}), B(u) on S/DP[{par}]")
        s.emit("{")
        valu = [] if "valu" in drop else dq_valu_ops(par, qt, exact, order)
        mf = []
        for hh in range(2):
            for dt in range(2):
                mf.append((f"{MFMA} %0, %1, %2, %0", f'"+{acc}"(dqt[{qtp}][{dt}])', f'"v"(KTF({hh}, {dt})), "v"(DSF({parn}, {hh}))'))
        for c in range(4):
            if c == 0:
                mf.append((f"{MFMA} %0, %1, %2, 0", f'"=&v"(S[{parn}])', f'"v"(kf[{c}]), QFC(qf[{qtn}][{c}])'))
                if exact:
                    mf.append((f"{MFMA} %0, %1, %2, 0", f'"=&v"(DP[{parn}])', f'"v"(vf[{c}]), QFC(dof[{qtn}][{c}])'))
                else:
                    mf.append((f"{MFMA} %0, %1, %2, %3", f'"=&v"(DP[{parn}])', f'"v"(vf[{c}]), QFC(dof[{qtn}][{c}]), "v"(ND[{qtn}])'))
            else:
                mf.append((f"{MFMA} %0, %1, %2, %0", f'"+v"(S[{parn}])', f'"v"(kf[{c}]), QFC(qf[{qtn}][{c}])'))
                mf.append((f"{MFMA} %0, %1, %2, %0", f'"+v"(DP[{parn}])', f'"v"(vf[{c}]), QFC(dof[{qtn}][{c}])'))
        ngap = len(mf)
        counts = spread(len(valu), ngap, weights)
        early = jsn != js  # the next unit works on another 32-key half: its K / V row fragments are read at the top of this slot
        late = qt == 0     # first unit of this 32-key half: its transposed K fragments (C stage of the next slots) are read in this slot
        hand_over = (js, qt) == (1, 0)
        if (js, qt) == (0, 0):
            s.emit("RING_ADVANCE_TR();  // transposed-fragment addresses -> ring slot of this tile")
        if hand_over:
            s.emit("// tile hand-over: my loads of tile t+1 have landed; everyone's have, and nobody reads tile t-1 any more")
            s.asm("s_waitcnt vmcnt(0)\\n\\ts_barrier", "", "", '"memory"')
        if early and jsn == 0:
            s.emit("RING_ADVANCE_ROW();  // row-fragment addresses -> ring slot of tile t+1")
        s.asm("s_waitcnt lgkmcnt(0)\\n\\ts_nop 1", "", "", '"memory"')
        vi = 0
        for g in range(ngap):
            if g == 4 and early and "lds" not in drop:
                s.emit("// the next unit's K / V row fragments must have landed before its first MFMA")
                s.asm("s_waitcnt lgkmcnt(0)", "", "", '"memory"')
            t, o, ins = mf[g]
            if "mfma16" in drop:
                # time-only ablation: the same FLOPs as two v_mfma_f32_16x16x32_bf16 on dummy 4-register accumulators (same A / B registers): what would the
                # other MFMA shape cost in this stream (energy per FLOP, issue slots)?  Results are wrong on purpose.
                ab = ins.split("), ")
                a_op, b_op = ab[0] + ")", ab[1] if ab[1].endswith(")") else ab[1] + ")"
                b_op = b_op.replace("QFC(", '"v"(')
                for h in range(2):
                    s.asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0", f'"+v"(d16[{(2 * g + h) % 8}])', f"{a_op}, {b_op}")
            elif "mfma" not in drop:
                s.asm(t, o, ins)
            if early and g < 2 and "lds" not in drop:
                which, voff = ("kf", 0) if g == 0 else ("vf", 8192)
                for c in range(4):
                    s.asm(f"ds_read_b128 %0, %1 offset:{jsn * 4096 + voff}", f'"=v"({which}[{c}])', f'"v"(ra[{c}])')
            if late and 5 <= g < 9 and "lds" not in drop:
                k = g - 5
                hh, dt = k >> 1, k & 1
                base = js * 4096 + hh * 2048
                s.asm(f"ds_read_b64_tr_b16 %0, %1 offset:{base}", f'"=v"(ktlo[{hh}][{dt}])', f'"v"(tra[{dt}][0])')
                s.asm(f"ds_read_b64_tr_b16 %0, %1 offset:{base}", f'"=v"(kthi[{hh}][{dt}])', f'"v"(tra[{dt}][1])')
            if hand_over and 4 <= g < 8 and "dma" not in drop:
                s.emit(f"DMA_PIECE({g - 4});")
            for _ in range(counts[g]):
                t2, o2, i2 = valu[vi]
                s.asm(t2, o2, i2)
                vi += 1
        assert vi == len(valu)
        s.emit("}")
    path = os.path.join(_out_dir(name), f"attn_pl_dq{nq}_{name}.inc")
    with open(path, "w") as f:
        f.write("\n".join(s.lines) + "\n")
    return path


# ------------------------------------------------------------------------------------------------------------------------------
# dK / dV kernel: a wave owns 64 keys (kt = 0, 1: 32 each), loops over 64-query tiles (is = 0, 1: 32 query rows each).
# unit u = (t, is, kt), parity p = kt.  Slot of unit u:
#   C(u-1): 8 MFMAs  dV[kt'] += dO^T . P[p^1],  dK[kt'] += Q^T . dSn[p^1];
#   A(u+1): 8 MFMAs  S[p^1] = Q.K^T - lse / sl,  DP[p^1] = dO.V^T - delta  (both row terms as accumulator inputs read from LDS);
#   B(u): VALU  P = exp2(S * sl),  dS = P * DP, both packed to bf16 fragments.
# Row fragments of Q / dO and the two accumulator-input rows are shared by kt = 0, 1 (read once per `is`); so are the transposed fragments.
# ------------------------------------------------------------------------------------------------------------------------------
def dkv_valu_ops(par: int, order: str):
    S, DP = f"S[{par}]", f"DP[{par}]"

    def M1(r):
        return ("v_mul_f32 %0, %1, %2", f'"=v"(x[{r}])', f'"v"({S}[{r}]), "v"(sl)')

    def E(r):
        return ("v_exp_f32 %0, %0", f'"+v"(x[{r}])', "")

    def M2(r):
        return ("v_mul_f32 %0, %1, %2", f'"=v"(y[{r}])', f'"v"(x[{r}]), "v"({DP}[{r}])')

    def PP(r):
        hh, e = r >> 3, (r & 7) >> 1
        return ("v_cvt_pk_bf16_f32 %0, %1, %2", f'"=v"(pw[{par}][{hh}][{e}])', f'"v"(x[{r}]), "v"(x[{r + 1}])')

    def PD(r):
        hh, e = r >> 3, (r & 7) >> 1
        return ("v_cvt_pk_bf16_f32 %0, %1, %2", f'"=v"(dsw[{par}][{hh}][{e}])', f'"v"(y[{r}]), "v"(y[{r + 1}])')

    ops = []
    if order.startswith("g"):
        group = int(order[1:])
        for m in range(16 // group):
            rs = [group * m + i for i in range(group)]
            ops += [M1(r) for r in rs] + [E(r) for r in rs] + [M2(r) for r in rs] + [PP(r) for r in rs[::2]] + [PD(r) for r in rs[::2]]
    else:
        L = 4
        for s_ in range(16 + 2 * L + 2):
            if s_ < 16:
                ops.append(M1(s_))
            if 0 <= s_ - L < 16:
                ops.append(E(s_ - L))
            if 0 <= s_ - 2 * L < 16:
                ops.append(M2(s_ - 2 * L))
            r = s_ - 2 * L - 1
            if 0 <= r < 16 and r % 2 == 1:
                ops.append(PP(r - 1))
                ops.append(PD(r - 1))
    assert len(ops) == 64, len(ops)
    return ops


def gen_dkv(name: str, order: str = "roll", drop=()):
    s = Stream()
    s.emit(f"// GENERATED by tools/gen_attn_pl.py (dkv_{name}: order={order} drop={','.join(drop) or '-'}) -- do not edit")
    units = [(is_, kt) for is_ in range(2) for kt in range(2)]
    n = len(units)
    for i, (is_, kt) in enumerate(units):
        par, parn = kt, kt ^ 1
        isn, ktn = units[(i + 1) % n]
        isp, ktp = units[(i - 1) % n]
        s.emit(f"// ---- slot {i} (is {is_}, kt {kt}): C(u-1) on dV/dK[{ktp}], A(u+1) into S/DP[{parn}] (is {isn}, kt {ktn}), B(u) on S/DP[{par}]")
        s.emit("{")
        valu = [] if "valu" in drop else dkv_valu_ops(par, order)
        mf = []
        for hh in range(2):
            for dt in range(2):
                mf.append((f"{MFMA} %0, %1, %2, %0", f'"+a"(dV[{ktp}][{dt}])', f'"v"(DOT({hh}, {dt})), "v"(PF({parn}, {hh}))'))
                mf.append((f"{MFMA} %0, %1, %2, %0", f'"+a"(dK[{ktp}][{dt}])', f'"v"(QT({hh}, {dt})), "v"(DSF({parn}, {hh}))'))
        for c in range(4):
            if c == 0:
                mf.append((f"{MFMA} %0, %1, %2, %3", f'"=&v"(S[{parn}])', f'"v"(qr[{c}]), "v"(kf[{ktn}][{c}]), "v"(LSI)'))
                mf.append((f"{MFMA} %0, %1, %2, %3", f'"=&v"(DP[{parn}])', f'"v"(dor[{c}]), "v"(nvf[{ktn}][{c}]), "v"(DLI)'))
            else:
                mf.append((f"{MFMA} %0, %1, %2, %0", f'"+v"(S[{parn}])', f'"v"(qr[{c}]), "v"(kf[{ktn}][{c}])'))
                mf.append((f"{MFMA} %0, %1, %2, %0", f'"+v"(DP[{parn}])', f'"v"(dor[{c}]), "v"(nvf[{ktn}][{c}])'))
        ngap = len(mf)  # 16
        counts = spread(len(valu), ngap)
        early = isn != is_   # the next unit works on the other 32-row half: its Q / dO row fragments and accumulator-input rows are read at the top of this slot
        late = kt == 0       # first unit of this 32-row half: its transposed Q / dO fragments (C stage of the next two slots) are read in this slot
        hand_over = (is_, kt) == (1, 0)
        if (is_, kt) == (0, 0):
            s.emit("RING_ADVANCE_TR();  // transposed-fragment addresses -> ring slot of this tile")
        if hand_over:
            s.emit("// tile hand-over: my loads of tile t+1 have landed (wave 0 then turns the lse row into the accumulator input -lse / sl, in place);")
            s.emit("// after the barrier everyone's have, and nobody reads tile t-1 any more")
            s.emit("HAND_OVER();")
        if early and isn == 0:
            s.emit("RING_ADVANCE_ROW();  // row-fragment addresses -> ring slot of tile t+1")
        s.asm("s_waitcnt lgkmcnt(0)\\n\\ts_nop 1", "", "", '"memory"')
        vi = 0
        for g in range(ngap):
            if g == 8 and early and "lds" not in drop:
                s.emit("// the next unit's row fragments and accumulator inputs must have landed before its first MFMA")
                s.asm("s_waitcnt lgkmcnt(0)", "", "", '"memory"')
            t, o, ins = mf[g]
            if "mfma" not in drop:
                s.asm(t, o, ins)
            if early and g < 4 and "lds" not in drop:
                roff = isn * 4096
                if g == 0:
                    for c in range(4):
                        s.asm(f"ds_read_b128 %0, %1 offset:{roff}", f'"=v"(qr[{c}])', f'"v"(ra[{c}])')
                elif g == 1:
                    for c in range(4):
                        s.asm(f"ds_read_b128 %0, %1 offset:{roff + 8192}", f'"=v"(dor[{c}])', f'"v"(ra[{c}])')
                elif g == 2:
                    for rq in range(4):
                        s.asm(f"ds_read_b128 %0, %1 offset:{16384 + isn * 128 + rq * 32}", f'"=v"(lsi[{rq}])', '"v"(la)')
                else:
                    for rq in range(4):
                        s.asm(f"ds_read_b128 %0, %1 offset:{16384 + 256 + isn * 128 + rq * 32}", f'"=v"(dli[{rq}])', '"v"(la)')
            if late and 8 <= g < 16 and "lds" not in drop:
                k = g - 8          # 0..7 -> (image, hh, dt)
                img, hh, dt = k >> 2, (k >> 1) & 1, k & 1
                base = is_ * 4096 + hh * 2048 + img * 8192
                lo, hi = ("qtlo", "qthi") if img == 0 else ("dtlo", "dthi")
                s.asm(f"ds_read_b64_tr_b16 %0, %1 offset:{base}", f'"=v"({lo}[{hh}][{dt}])', f'"v"(tra[{dt}][0])')
                s.asm(f"ds_read_b64_tr_b16 %0, %1 offset:{base}", f'"=v"({hi}[{hh}][{dt}])', f'"v"(tra[{dt}][1])')
            if hand_over and 8 <= g < 13 and "dma" not in drop:
                s.emit(f"DMA_PIECE({g - 8});")
            for _ in range(counts[g]):
                t2, o2, i2 = valu[vi]
                s.asm(t2, o2, i2)
                vi += 1
        assert vi == len(valu)
        s.emit("}")
    path = os.path.join(_out_dir(name), f"attn_pl_dkv_{name}.inc")
    with open(path, "w") as f:
        f.write("\n".join(s.lines) + "\n")
    return path


# ------------------------------------------------------------------------------------------------------------------------------
# forward: a wave owns 64 query rows (qt = 0, 1: 32 each), loops over 64-key tiles; one slot per (tile, qt) = 32 queries x 64 keys.
# Slot of unit u = (t, qt), parity p = qt:
#   C(u-1): 12 MFMAs  O[qt'] += V^T . P[p^1]  (8)  and the row sums of P[p^1] through an all-ones A operand (4);
#   A(u+1):  8 MFMAs  S[p^1] = K . Q[qt']^T  for the two 32-key halves (chains interleaved);
#   B(u):  102 VALU   the row max of S[p] (22; S[p] was finished by the last MFMAs of the slot before, so this starts two MFMAs into the slot), the
#                     lazy-reference decision of attn_fwd_kernel (rare rescale of O[qt]: its last products were issued in the slot before, its next
#                     ones come in the slot after), then P[p] = exp2(S[p] * sl - m) -> bf16 fragments (80).
# MFMA order  C = rs00 pv000 rs01 pv001 rs10 pv010 rs11 pv011 pv100 pv101 pv110 pv111 | A  (accumulators alternate; per accumulator the order of
# attn_fwd_kernel: js, hh ascending).  qt == 0 slots reload each V^T fragment for the new tile right behind the MFMA that used it last (the last
# reload has eight MFMAs to land); qt == 1 slots carry the tile hand-over, the K row fragments of tile t+1 and the DMA of tile t+2.
# ------------------------------------------------------------------------------------------------------------------------------
def fwd_valu_ops(par: int):
    def F(i):
        js, r = i >> 4, i & 15
        return ("v_fma_f32 %0, %1, %2, %3", f'"=v"(x[{i}])', f'"v"(S[{par}][{js}][{r}]), "v"(sl), "v"(nm[{par}])')

    def E(i):
        return ("v_exp_f32 %0, %0", f'"+v"(x[{i}])', "")

    def P(i):  # pair (i, i+1), i even
        js, r = i >> 4, i & 15
        hh, e = r >> 3, (r & 7) >> 1
        return ("v_cvt_pk_bf16_f32 %0, %1, %2", f'"=v"(pw[{par}][{js}][{hh}][{e}])', f'"v"(x[{i}]), "v"(x[{i + 1}])')

    ops, L = [], 4
    for s_ in range(32 + L + 2):
        if s_ < 32:
            ops.append(F(s_))
        if 0 <= s_ - L < 32:
            ops.append(E(s_ - L))
        r = s_ - L - 1
        if 0 <= r < 32 and r % 2 == 1:
            ops.append(P(r - 1))
    assert len(ops) == 80, len(ops)
    return ops


def fwd_max_ops(parn: int):
    """row max of the 32 scores of the next unit -> mx (log2 domain), both half-waves: max16 x 2 (chains interleaved), scale, cross-half."""
    S0, S1 = f"S[{parn}][0]", f"S[{parn}][1]"
    ops = []
    for k in range(8):
        for (S, m) in ((S0, "mxa"), (S1, "mxb")):
            if k == 0:
                ops.append(("v_max3_f32 %0, %1, %2, %3", f'"=&v"({m})', f'"v"({S}[0]), "v"({S}[1]), "v"({S}[2])'))
            elif k < 7:
                ops.append(("v_max3_f32 %0, %0, %1, %2", f'"+v"({m})', f'"v"({S}[{2 * k + 1}]), "v"({S}[{2 * k + 2}])'))
            else:
                ops.append(("v_max_f32 %0, %0, %1", f'"+v"({m})', f'"v"({S}[15])'))
    ops.append(("v_max_f32 %0, %0, %1", '"+v"(mxa)', '"v"(mxb)'))
    ops.append(("v_mul_f32 %0, %1, %2", '"=v"(mx)', '"v"(mxa), "v"(sl)'))
    ops.append(("v_mov_b32 %1, %0\\n\\ts_nop 1\\n\\tv_permlane32_swap_b32 %1, %0\\n\\tv_max_f32 %0, %0, %1", '"+v"(mx), "=&v"(mxb)', ""))
    return ops


def gen_fwd(name: str, drop=()):
    s = Stream()
    s.emit(f"// GENERATED by tools/gen_attn_pl.py (fwd_{name}: drop={','.join(drop) or '-'}) -- do not edit")
    for qt in range(2):
        par, parn = qt, qt ^ 1
        s.emit(f"// ---- slot (t, qt {qt}): max + exp2 of S[{par}]; C(u-1) on O[{parn}] with P[{parn}]; A(u+1) into S[{parn}]")
        s.emit("{")
        valu = [] if "valu" in drop else fwd_valu_ops(par)
        vmax = fwd_max_ops(par)
        def rs(js, hh, first=False):
            if first:
                return (f"{MFMA} %0, %1, %2, 0", '"=&v"(lsum)', f'"v"(ones), "v"(PF({parn}, {js}, {hh}))')
            return (f"{MFMA} %0, %1, %2, %0", '"+v"(lsum)', f'"v"(ones), "v"(PF({parn}, {js}, {hh}))')
        def pv(js, hh, dt):
            return (f"{MFMA} %0, %1, %2, %0", f'"+a"(oacc[{parn}][{dt}])', f'"v"(VTF({js}, {hh}, {dt})), "v"(PF({parn}, {js}, {hh}))', (js, hh, dt))
        def sa(js, c):
            if c == 0:
                return (f"{MFMA} %0, %1, %2, 0", f'"=&v"(S[{parn}][{js}])', f'"v"(kf[{js}][{c}]), "v"(qf[{parn}][{c}])')
            return (f"{MFMA} %0, %1, %2, %0", f'"+v"(S[{parn}][{js}])', f'"v"(kf[{js}][{c}]), "v"(qf[{parn}][{c}])')
        mf = [rs(0, 0, True), pv(0, 0, 0), rs(0, 1), pv(0, 0, 1), rs(1, 0), pv(0, 1, 0), rs(1, 1), pv(0, 1, 1), pv(1, 0, 0), pv(1, 0, 1), pv(1, 1, 0), pv(1, 1, 1)]
        mf += [sa(js, c) for c in range(4) for js in range(2)]
        ngap = len(mf)  # 20
        # S[par] was finished by the last two MFMAs of the slot before: its row max waits two MFMAs (gaps 2 .. 4), then the (rare) rescale decision,
        # then the exp2 work in gaps 5 .. 19; the row sums of C(u-1) (MFMAs 0, 2, 4, 6) are taken at gap 10
        mcounts = [0, 0] + spread(len(vmax), 3) + [0] * 15
        counts = [0] * 5 + spread(len(valu), 15)
        if qt == 0:
            s.emit("RING_ADVANCE_TR();  // transposed-fragment addresses -> ring slot of this tile")
        else:
            s.emit("// tile hand-over: my loads of tile t+1 have landed; after the barrier everyone's have, and nobody reads tile t-1 any more")
            s.asm("s_waitcnt vmcnt(0)\\n\\ts_barrier", "", "", '"memory"')
            s.emit("RING_ADVANCE_ROW();  // row-fragment addresses -> ring slot of tile t+1")
        s.asm("s_waitcnt lgkmcnt(0)\\n\\ts_nop 1", "", "", '"memory"')
        vi = mi = 0
        for g in range(ngap):
            if g == 12 and qt == 1 and "lds" not in drop:
                s.emit("// the K row fragments of tile t+1 must have landed before the first score MFMA")
                s.asm("s_waitcnt lgkmcnt(0)", "", "", '"memory"')
            m = mf[g]
            if "mfma" not in drop:
                s.asm(m[0], m[1], m[2])
            if qt == 0 and len(m) == 4 and "lds" not in drop:
                js, hh, dt = m[3]
                base = 8192 + js * 4096 + hh * 2048
                s.asm(f"ds_read_b64_tr_b16 %0, %1 offset:{base}", f'"=v"(vtlo[{js}][{hh}][{dt}])', f'"v"(tra[{dt}][0])')
                s.asm(f"ds_read_b64_tr_b16 %0, %1 offset:{base}", f'"=v"(vthi[{js}][{hh}][{dt}])', f'"v"(tra[{dt}][1])')
            if qt == 1 and g < 2 and "lds" not in drop:
                for c in range(4):
                    s.asm(f"ds_read_b128 %0, %1 offset:{g * 4096}", f'"=v"(kf[{g}][{c}])', f'"v"(ra[{c}])')
            if qt == 1 and 6 <= g < 10 and "dma" not in drop:
                s.emit(f"DMA_PIECE({g - 6});")
            for _ in range(mcounts[g]):
                t2, o2, i2 = vmax[mi]
                s.asm(t2, o2, i2)
                mi += 1
            if g == 4:
                s.emit(f"DECIDE({par});")
            for _ in range(counts[g]):
                t2, o2, i2 = valu[vi]
                s.asm(t2, o2, i2)
                vi += 1
            if g == 10:
                s.emit(f"L_UPDATE({parn});")
        assert vi == len(valu) and mi == len(vmax)
        s.emit("}")
    path = os.path.join(EXP_OUT, f"attn_pl_fwd_{name}.inc")  # experiment, not shipped (tools/experimental/attention_experimental_6_fwd_pl.hip.h)
    with open(path, "w") as f:
        f.write("\n".join(s.lines) + "\n")
    return path


# ------------------------------------------------------------------------------------------------------------------------------
# dK / dV at head_dim 128 (Wan, HunyuanVideo), ONE pass: a wave owns 32 keys, loops over 64-query tiles (is = 0, 1: 32 query rows each); one slot per
# unit u = (t, is) = 32 queries x 32 keys:
#   C(u-1): 16 MFMAs  dV[dt] += dO^T . P,  dK[dt] += Q^T . dS  (dt = 0..3, hh = 0, 1) -- positions 0 .. 15;
#   A(u+1): 16 MFMAs  S = Q.K^T - lse / sl,  DP = dO.V^T - delta  (chains of 8 over the 128-wide head, interleaved) -- positions 16 .. 31;
#   B(u):   64 VALU   P = exp2(S * sl + bias_j),  dS = P * DP, both packed to bf16 -- gaps 2 .. 15, i.e. finished BEFORE A(u+1) starts, so S and DP are single
#                     buffers (the registers that pays for: K and V fragments of the wave's keys stay resident, 64 VGPRs).
# Every fragment read from LDS is used once: transposed fragments go through an 8-deep rolling buffer (read 8 MFMAs ahead), row fragments through a 4-deep one
# per operand.  LDS reads return in order, so every consumer waits with a COUNTED lgkmcnt (the number of younger reads in flight, capped at the counter's 15).
# ------------------------------------------------------------------------------------------------------------------------------
def gen_dkv128(name: str, drop=()):
    s = Stream()
    s.emit(f"// GENERATED by tools/gen_attn_pl.py (dkv128_{name}: drop={','.join(drop) or '-'}) -- do not edit")
    seq = [0]          # LDS reads issued so far (monotone over the two slots of a tile; relative counts only)
    issued_at = {}     # name -> sequence number of the read that fills it

    def lds_read(text, outs, ins, key):
        if "lds" in drop:
            return
        s.asm(text, outs, ins)
        seq[0] += 1
        issued_at[key] = seq[0]

    landed = [0]       # every read up to this sequence number is known to have landed (by an earlier wait)

    def wait_for(keys):
        """counted wait: everything up to the youngest of `keys` has landed"""
        if "lds" in drop:
            return
        need = max(issued_at.get(k, -10**9) for k in keys)
        if need <= landed[0]:
            return
        n = min(seq[0] - need, 15)
        s.asm(f"s_waitcnt lgkmcnt({n})", "", "", '"memory"')
        landed[0] = max(landed[0], seq[0] - n)

    # reads that the previous tile iteration left in flight (second slot, gaps 8 .. 15: the first eight transposed fragments of unit (t, 0); positions 16 .. 23:
    # the second halves of the row fragments, all consumed there): model them as issued before everything of this iteration
    for m in range(8):
        issued_at[("tr", m)] = -100 + m  # older than anything issued in this iteration: never waited for (the A stage's waits covered them)

    for is_ in range(2):
        par, parn = is_, is_ ^ 1
        s.emit(f"// ---- slot (t, is {is_}): C(u-1) with P / dS[{parn}]; B(u) on S / DP -> P / dS[{par}]; A(u+1) into S / DP")
        s.emit("{")
        def F(r): return ("v_fma_f32 %0, %1, %2, %3", f'"=v"(x[{r}])', f'"v"(S[{r}]), "v"(sl), "v"(bj)')
        def E(r): return ("v_exp_f32 %0, %0", f'"+v"(x[{r}])', "")
        def M2(r): return ("v_mul_f32 %0, %1, %2", f'"=v"(y[{r}])', f'"v"(x[{r}]), "v"(DP[{r}])')
        def PP(r):
            hh, e = r >> 3, (r & 7) >> 1
            return ("v_cvt_pk_bf16_f32 %0, %1, %2", f'"=v"(pw[{par}][{hh}][{e}])', f'"v"(x[{r}]), "v"(x[{r + 1}])')
        def PD(r):
            hh, e = r >> 3, (r & 7) >> 1
            return ("v_cvt_pk_bf16_f32 %0, %1, %2", f'"=v"(dsw[{par}][{hh}][{e}])', f'"v"(y[{r}]), "v"(y[{r + 1}])')
        valu, L = [], 4
        for s_ in range(16 + 2 * L + 2):
            if s_ < 16: valu.append(F(s_))
            if 0 <= s_ - L < 16: valu.append(E(s_ - L))
            if 0 <= s_ - 2 * L < 16: valu.append(M2(s_ - 2 * L))
            r = s_ - 2 * L - 1
            if 0 <= r < 16 and r % 2 == 1:
                valu.append(PP(r - 1)); valu.append(PD(r - 1))
        assert len(valu) == 64
        if "valu" in drop:
            valu = []
        counts = [0, 0] + spread(len(valu), 14) + [0] * 16

        def tr_read(m, unit_is):
            """transposed fragment m (0..15: hh = m >> 3, dt = (m >> 1) & 3, image w = m & 1: 0 = dO^T for dV, 1 = Q^T for dK) of the unit with row half unit_is -> buffer m & 7"""
            hh, dt, w = m >> 3, (m >> 1) & 3, m & 1
            base = (16384 if w == 0 else 0) + (dt >> 1) * 8192 + unit_is * 4096 + hh * 2048
            b = m & 7
            lds_read(f"ds_read_b64_tr_b16 %0, %1 offset:{base}", f'"=v"(trlo[{b}])', f'"v"(tra[{dt & 1}][0])', ("trl", m))
            lds_read(f"ds_read_b64_tr_b16 %0, %1 offset:{base}", f'"=v"(trhi[{b}])', f'"v"(tra[{dt & 1}][1])', ("tr", m))

        if is_ == 1:
            s.emit("// tile hand-over: my loads of tile t+1 have landed (wave 0 then turns its lse / delta rows into the accumulator inputs, in place);")
            s.emit("// after the barrier everyone's have, and nobody reads tile t-1 any more")
            s.emit("HAND_OVER();")
            s.emit("RING_ADVANCE_ROW();  // row-fragment addresses -> ring slot of tile t+1")
        isn = is_ ^ 1  # row half of unit u+1
        for g in range(32):
            # ---- waits in front of the MFMA at position g
            if g < 8:
                wait_for([("tr", g)])
            elif g < 16:
                wait_for([("tr", g)])
            else:
                k = g - 16
                c, which = k >> 1, k & 1
                keys = [("qr" if which == 0 else "dor", c)]
                if c == 0:
                    keys += [("lsi", 3)] if which == 0 else [("dli", 3)]
                wait_for(keys)
            # ---- the MFMA
            if g < 16:
                hh, dt, w = g >> 3, (g >> 1) & 3, g & 1
                frag = f"TRF({g & 7})"
                if w == 0:
                    m_ = (f"{MFMA} %0, %1, %2, %0", f'"+a"(dV[{dt}])', f'"v"({frag}), "v"(PF({parn}, {hh}))')
                else:
                    m_ = (f"{MFMA} %0, %1, %2, %0", f'"+a"(dK[{dt}])', f'"v"({frag}), "v"(DSF({parn}, {hh}))')
            else:
                k = g - 16
                c, which = k >> 1, k & 1
                if which == 0:
                    m_ = (f"{MFMA} %0, %1, %2, %3", '"=&v"(S)', f'"v"(qr[{c & 3}]), "v"(kf[{c}]), "v"(LSI)') if c == 0 else (f"{MFMA} %0, %1, %2, %0", '"+v"(S)', f'"v"(qr[{c & 3}]), "v"(kf[{c}])')
                else:
                    m_ = (f"{MFMA} %0, %1, %2, %3", '"=&v"(DP)', f'"v"(dor[{c & 3}]), "v"(vf[{c}]), "v"(DLI)') if c == 0 else (f"{MFMA} %0, %1, %2, %0", '"+v"(DP)', f'"v"(dor[{c & 3}]), "v"(vf[{c}])')
            if "mfma" not in drop:
                s.asm(m_[0], m_[1], m_[2])
            # ---- LDS reads behind it
            if g < 8:
                tr_read(8 + g, parn)            # second half of C(u-1)'s fragments, into the buffer this MFMA just used
            elif g < 16:
                if g == 8 and is_ == 0:
                    s.emit("RING_ADVANCE_TR();  // transposed-fragment addresses -> ring slot of this tile (unit u is its first)")
                tr_read(g - 8, par)             # first half of C(u)'s fragments (next slot)
                if 8 <= g < 12:                 # the first four row fragments of unit u+1 and its accumulator-input rows
                    c = g - 8
                    lds_read(f"ds_read_b128 %0, %1 offset:{isn * 4096}", f'"=v"(qr[{c}])', f'"v"(ra[{c}])', ("qr", c))
                    lds_read(f"ds_read_b128 %0, %1 offset:{16384 + isn * 4096}", f'"=v"(dor[{c}])', f'"v"(ra[{c}])', ("dor", c))
                    lds_read(f"ds_read_b128 %0, %1 offset:{32768 + isn * 128 + c * 32}", f'"=v"(lsi[{c}])', '"v"(la)', ("lsi", c))
                    lds_read(f"ds_read_b128 %0, %1 offset:{32768 + 256 + isn * 128 + c * 32}", f'"=v"(dli[{c}])', '"v"(la)', ("dli", c))
            elif g < 24:
                k = g - 16
                c, which = k >> 1, k & 1         # this MFMA used row fragment c (< 4) of its operand: reload the buffer with fragment c + 4 (second 64-wide image)
                if which == 0:
                    lds_read(f"ds_read_b128 %0, %1 offset:{8192 + isn * 4096}", f'"=v"(qr[{c}])', f'"v"(ra[{c}])', ("qr", c + 4))
                else:
                    lds_read(f"ds_read_b128 %0, %1 offset:{16384 + 8192 + isn * 4096}", f'"=v"(dor[{c}])', f'"v"(ra[{c}])', ("dor", c + 4))
            if is_ == 1 and 16 <= g < 25 and "dma" not in drop:
                s.emit(f"DMA_PIECE({g - 16});")
            for _ in range(counts[g]):
                t2, o2, i2 = valu.pop(0)
                s.asm(t2, o2, i2)
        assert not valu
        s.emit("}")
    path = os.path.join(OUT, f"attn_pl_dkv128_{name}.inc")
    with open(path, "w") as f:
        f.write("\n".join(s.lines) + "\n")
    return path


# ------------------------------------------------------------------------------------------------------------------------------
# dQ at head_dim 128 (Wan, HunyuanVideo): a wave owns 32 query rows (q and dO fragments resident: 64 VGPRs), loops over 64-key tiles (js = 0, 1: 32 keys each);
# one slot per unit u = (t, js) = 32 queries x 32 keys:
#   C(u-1):  8 MFMAs  dQ[dt] += K^T . dS  (dt = 0..3, hh = 0, 1)                                          -- positions 0 .. 7
#   A(u+1): 16 MFMAs  S = K.Q^T,  DP = V.dO^T  (chains of 8 over the 128-wide head, interleaved)             -- positions 8 .. 23
#   B(u):   88 VALU   bl = bias - lse (gaps 0 .. 1), x = S * sl + bl and y = DP - delta (gaps 2 .. 7: S and DP are free again when A(u+1) starts -- single
#                     buffers), then exp2, the product and the packs out of x / y under A(u+1) (gaps 8 .. 22).
# The arithmetic of attn_bwd_dq_kernel<HAS_KB, 2> operation for operation (a zero bias row where there is no bias: 0 - lse = -lse exactly).
# K / V row fragments through 4-deep rolling buffers, K^T fragments through an 8-deep one (a unit's eight, read a slot ahead), counted lgkmcnt waits.
# ------------------------------------------------------------------------------------------------------------------------------
def gen_dq128(name: str, drop=()):
    s = Stream()
    s.emit(f"// GENERATED by tools/gen_attn_pl.py (dq128_{name}: drop={','.join(drop) or '-'}) -- do not edit")
    seq, issued_at, landed = [0], {}, [0]

    def lds_read(text, outs, ins, key):
        if "lds" in drop:
            return
        s.asm(text, outs, ins)
        seq[0] += 1
        issued_at[key] = seq[0]

    def wait_for(keys):
        if "lds" in drop:
            return
        need = max(issued_at.get(k, -10**9) for k in keys)
        if need <= landed[0]:
            return
        n = min(seq[0] - need, 15)
        s.asm(f"s_waitcnt lgkmcnt({n})", "", "", '"memory"')
        landed[0] = max(landed[0], seq[0] - n)

    for js in range(2):
        par, parn = js, js ^ 1
        jsn = js ^ 1  # key half of unit u+1
        s.emit(f"// ---- slot (t, js {js}): C(u-1) with dS[{parn}]; B(u) on S / DP -> dS[{par}]; A(u+1) into S / DP")
        s.emit("{")
        def BS(r): return ("v_sub_f32 %0, %1, %2", f'"=v"(bl[{r}])', f'"v"(b4[{r >> 2}][{r & 3}]), "v"(lse_i)')
        def F(r): return ("v_fma_f32 %0, %1, %2, %3", f'"=v"(x[{r}])', f'"v"(S[{r}]), "v"(sl), "v"(bl[{r}])')
        def U(r): return ("v_sub_f32 %0, %1, %2", f'"=v"(y[{r}])', f'"v"(DP[{r}]), "v"(del_i)')
        def E(r): return ("v_exp_f32 %0, %0", f'"+v"(x[{r}])', "")
        def M(r): return ("v_mul_f32 %0, %0, %1", f'"+v"(y[{r}])', f'"v"(x[{r}])')
        def P(r):
            hh, e = r >> 3, (r & 7) >> 1
            return ("v_cvt_pk_bf16_f32 %0, %1, %2", f'"=v"(dsw[{par}][{hh}][{e}])', f'"v"(y[{r}]), "v"(y[{r + 1}])')
        v_bias = [BS(r) for r in range(16)]
        v_sd = []
        for r in range(16):
            v_sd += [F(r), U(r)]
        v_rest, L = [], 4
        for s_ in range(16 + L + 2):
            if s_ < 16: v_rest.append(E(s_))
            if 0 <= s_ - L < 16: v_rest.append(M(s_ - L))
            r = s_ - L - 1
            if 0 <= r < 16 and r % 2 == 1: v_rest.append(P(r - 1))
        assert len(v_rest) == 40
        if "valu" in drop:
            v_bias, v_sd, v_rest = [], [], []
        plan = {}  # gap -> list of VALU ops
        for g, k in zip(range(0, 2), spread(len(v_bias), 2)):
            plan.setdefault(g, []); plan[g] += [v_bias.pop(0) for _ in range(k)]
        for g, k in zip(range(2, 8), spread(len(v_sd), 6)):
            plan.setdefault(g, []); plan[g] += [v_sd.pop(0) for _ in range(k)]
        for g, k in zip(range(8, 23), spread(len(v_rest), 15)):
            plan.setdefault(g, []); plan[g] += [v_rest.pop(0) for _ in range(k)]

        def tr_read(m, unit_js):
            """K^T fragment m (hh = m >> 2, dt = m & 3) of the unit with key half unit_js -> buffer m"""
            hh, dt = m >> 2, m & 3
            base = (dt >> 1) * 8192 + unit_js * 4096 + hh * 2048
            lds_read(f"ds_read_b64_tr_b16 %0, %1 offset:{base}", f'"=v"(trlo[{m}])', f'"v"(tra[{dt & 1}][0])', ("trl", m))
            lds_read(f"ds_read_b64_tr_b16 %0, %1 offset:{base}", f'"=v"(trhi[{m}])', f'"v"(tra[{dt & 1}][1])', ("tr", m))

        if js == 1:
            s.emit("// tile hand-over: my loads of tile t+1 have landed (wave 0 then scales its bias row to the log2 domain, in place); after the barrier everyone's have,")
            s.emit("// and nobody reads tile t-1 any more")
            s.emit("HAND_OVER();")
            s.emit("RING_ADVANCE_ROW();  // row-fragment addresses -> ring slot of tile t+1")
        for g in range(24):
            # ---- waits in front of the MFMA at position g
            if g < 8:
                wait_for([("tr", g)])
            else:
                k = g - 8
                c, which = k >> 1, k & 1
                wait_for([("kf" if which == 0 else "vf", c)])
            # ---- the MFMA
            if g < 8:
                hh, dt = g >> 2, g & 3
                m_ = (f"{MFMA} %0, %1, %2, %0", f'"+a"(dqt[{dt}])', f'"v"(TRF({g})), "v"(DSF({parn}, {hh}))')
            else:
                k = g - 8
                c, which = k >> 1, k & 1
                if which == 0:
                    m_ = (f"{MFMA} %0, %1, %2, 0", '"=&v"(S)', f'"v"(kf[{c & 3}]), "v"(qf[{c}])') if c == 0 else (f"{MFMA} %0, %1, %2, %0", '"+v"(S)', f'"v"(kf[{c & 3}]), "v"(qf[{c}])')
                else:
                    m_ = (f"{MFMA} %0, %1, %2, 0", '"=&v"(DP)', f'"v"(vf[{c & 3}]), "v"(dof[{c}])') if c == 0 else (f"{MFMA} %0, %1, %2, %0", '"+v"(DP)', f'"v"(vf[{c & 3}]), "v"(dof[{c}])')
            if "mfma" not in drop:
                s.asm(m_[0], m_[1], m_[2])
            # ---- LDS reads behind it
            if g < 4:   # the first four K / V row fragments of unit u+1
                lds_read(f"ds_read_b128 %0, %1 offset:{jsn * 4096}", f'"=v"(kf[{g}])', f'"v"(ra[{g}])', ("kf", g))
                lds_read(f"ds_read_b128 %0, %1 offset:{16384 + jsn * 4096}", f'"=v"(vf[{g}])', f'"v"(ra[{g}])', ("vf", g))
            if 4 <= g < 8:  # its bias row (the bias subtractions of unit u, gaps 0 .. 1, have read the old one)
                rq = g - 4
                lds_read(f"ds_read_b128 %0, %1 offset:{32768 + jsn * 128 + rq * 32}", f'"=v"(b4[{rq}])', '"v"(la)', ("b4", rq))
            if 8 <= g < 16:
                if g == 8 and js == 0:
                    s.emit("RING_ADVANCE_TR();  // transposed-fragment addresses -> ring slot of this tile (unit u is its first)")
                tr_read(g - 8, par)  # C(u)'s fragments (next slot), into the buffers C(u-1) has released
                k = g - 8
                c, which = k >> 1, k & 1  # this MFMA used row fragment c (< 4): reload the buffer with fragment c + 4 (second 64-wide image)
                if which == 0:
                    lds_read(f"ds_read_b128 %0, %1 offset:{8192 + jsn * 4096}", f'"=v"(kf[{c}])', f'"v"(ra[{c}])', ("kf", c + 4))
                else:
                    lds_read(f"ds_read_b128 %0, %1 offset:{16384 + 8192 + jsn * 4096}", f'"=v"(vf[{c}])', f'"v"(ra[{c}])', ("vf", c + 4))
            if js == 1 and 12 <= g < 21 and "dma" not in drop:
                s.emit(f"DMA_PIECE({g - 12});")
            for (t2, o2, i2) in plan.get(g, []):
                s.asm(t2, o2, i2)
            if g == 22:  # the bias row of unit u+1 must have landed: its subtractions open the next slot
                wait_for([("b4", 3)])
        s.emit("}")
    path = os.path.join(EXP_OUT, f"attn_pl_dq128_{name}.inc")  # experiment, not shipped (tools/experimental/attention_experimental_7_dq128_pl.hip.h)
    with open(path, "w") as f:
        f.write("\n".join(s.lines) + "\n")
    return path


def main():
    made = []
    for nq in (1, 2):
        # x0: exact arithmetic of attn_bwd_dq2_kernel (zero accumulator inputs, explicit dp - delta): bit-identical outputs -> the pipeline's test
        made.append(gen_dq("x0", nq, True))
        # v1: -delta through the accumulator input of the dP chain (16 VALU fewer per unit); rolling VALU order
        made.append(gen_dq("v1", nq, False))
        # v2: as v1, VALU in passes over groups of four elements
        made.append(gen_dq("v2", nq, False, order="g4"))
        # v6: as v1 with packed fp32 fma / mul (two scores per instruction)
        made.append(gen_dq("v6", nq, False, order="pk"))
        # ablations of v1 (lab only; results wrong on purpose)
        made.append(gen_dq("a_novalu", nq, False, drop=("valu",)))
        made.append(gen_dq("a_nolds", nq, False, drop=("lds",)))
        made.append(gen_dq("a_nomfma", nq, False, drop=("mfma",)))
        made.append(gen_dq("a_mfma16", nq, False, drop=("mfma16",)))
    made.append(gen_dkv("v1"))
    made.append(gen_dkv("v2", order="g4"))
    made.append(gen_dkv("a_novalu", drop=("valu",)))
    made.append(gen_dkv("a_nolds", drop=("lds",)))
    made.append(gen_dkv128("v1"))
    made.append(gen_dq128("x0"))
    made.append(gen_fwd("v1"))
    made.append(gen_fwd("a_novalu", drop=("valu",)))
    made.append(gen_fwd("a_nolds", drop=("lds",)))
    for p in made:
        print(os.path.relpath(p, os.path.join(HERE, "..")))


if __name__ == "__main__":
    sys.exit(main())
