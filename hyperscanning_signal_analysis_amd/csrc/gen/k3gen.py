"""Generator of the hand-scheduled 64-channel body of K3 (csrc/tf_inv64_body.inc).

What the stream computes is what `tf_inv_kernel<4, false>` (csrc/tf_inv.hip) computes between its prologue and its
outputs -- A(f) = I - sum_k A_k z_k(f) from the packed coefficients, then the in-place blocked Gauss-Jordan inversion
with partial pivoting (the arithmetic of /root/reference/src/mtmvar.py:155-160) -- operation for operation, in the same
order, so that the two bodies give the same bits (tests/test_gpu_parity.py compares them).  What differs is who
allocates the registers: here every VGPR is placed by hand so that the body needs 96 of them (five workgroups per CU;
the compiler needs 128 and spills ~1 700 at 96), and the panel factorisation -- the workgroup's critical path -- is
a straight line with exec masks set by SALU moves instead of compare / saveexec / branch sequences.

Register map (NT = 4 waves, wave w owns column blocks w, w+4, w+8, w+12; "X layout" of csrc/tf_inv.hip):
  v[0:63]    the wave's 16 register blocks: block (Ig, Jl) = v[4k : 4k+3], k = 4 Ig + Jl = (re.lo, re.hi, im.lo, im.hi)
             of element (row 16 Ig + 4 b + i, column 16 Jl + 4 w + j) on lane (i, b, j) = (l >> 4, (l >> 2) & 3, l & 3)
  v[64:79]   A operands N_t (four (re, im) quads) of an update / the four panel columns x[0..3] of a factorisation
  v[80:87]   B operands (pivot rows by ds_swizzle), two sets / factorisation temporaries
  v[88:93]   scratch / factorisation temporaries
  v94 lane id, v95 LDS address of this lane's A operand
  s[36:87]   see the S_* names below; s[48:79] hold the twiddles while A(f) is built
Inputs of the asm statement (SGPR operands): arx pointer of (item, wave), twiddle pointer of the frequency, p, w, tau,
LDS base.  LDS map (bytes from the base) is exported to the C++ side as K3A_* constants.

Run `python k3gen.py` to regenerate csrc/tf_inv64_body.inc; tests/test_k3_asm_cpu.py executes the stream on the CPU
emulator (csrc/gen/gcnasm.py) against NumPy and checks wait states and wait counts.
"""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gcnasm import EXEC, VCC, Abs, Neg, Program, Reg, S, V  # noqa: E402

NT, MP, NSTEP, NR = 4, 64, 16, 3

# ---- LDS map (bytes).  The first five regions coincide with TfLds<4> of tf_inv.hip (the epilogue reuses them).
PBUF = 0                       # panel, row stride 80 B
NBUF = 5120                    # ring of NR N buffers, row stride 64 B, 4096 B each
SROW = NBUF + NR * 4096        # 17408 (unused by this body)
SWAPB = SROW + 128             # 17536: per wave 512 B for row interchanges
RSUM = SWAPB + NT * 512        # 19584: row-sum partials of the epilogue
SORIG = RSUM + 2048            # 21632: int orig[64]
SSWP = SORIG + 256             # 21888: ring of NR records {any, d0, d1, d2, d3, pad}, 32 B each
SINFO = SSWP + NR * 32         # 21984: int info
LDS_TOTAL = SINFO + 16         # 22000


def ACC(Ig, Jl):
    return V(4 * (4 * Ig + Jl), 4)


def RE(q):
    return q.sub(0, 2)


def IM(q):
    return q.sub(2, 2)


NQ = [V(64 + 4 * k, 4) for k in range(4)]          # A operands / panel columns
U = [V(80, 4), V(84, 4)]                           # (ur, ui) sets
vSWP, vT0, vT1, vT2, vT3, vT4 = V(88), V(89), V(90), V(91), V(92), V(93)
vLANE, vAN = V(94), V(95)
# factorisation temporaries (doubles)
fCAND, fDC, fDD, fY, fE, fIVR, fIVI = V(80, 2), V(82, 2), V(84, 2), V(86, 2), V(88, 2), V(90, 2), V(92, 2)
fMR, fMI = fCAND, fDC

sW, sLDS, sTAU = S(36), S(37), S(38, 2)
sARX, sTW, sP, sP2, sCNT, sTMP = S(40, 2), S(42, 2), S(44), S(45), S(46), S(47)
sIJ = S(48, 2)
sB = [S(50 + 2 * b, 2) for b in range(4)]
sDIAG = S(58, 2)
sPR, sPI = S(60, 2), S(62, 2)
sVALID, sOK = S(64, 2), S(66, 2)
sRS = [S(68 + k) for k in range(4)]
sRSTAR, sKMAX, sM1 = S(72), S(73), S(74, 2)
sIGS, sT0, sT1, sT2 = S(76), S(77), S(78), S(79)
sOKC = [S(80 + 2 * k, 2) for k in range(4)]
# `rlpiv` variant: the pivot row's other three columns as six SGPR pairs (re, im per column); the per-column |pivot|^2 > 0
# masks are not kept then (a branch records the first bad column in sBAD instead)
PVS = [S(80, 2), S(82, 2), S(84, 2), S(86, 2), S(44, 2), S(46, 2)]
sBAD = S(66)
# A(f) phase only (the twiddles occupy s[48:79] then): running pointers and counters
sPTR, sTWP, sCH, sA1, sA2, sSTRIDE = S(80, 2), S(82, 2), S(84), S(85), S(86), S(88)
TW0 = 48                         # twiddles of a chunk of 8 lags: s[48:79]
CLOBBER_S = (36, 89)             # [lo, hi)


class Gen:
    # stream variants (same arithmetic, except `ldsconst`, which may turn a -0.0 into +0.0 on the pivot row's lane):
    #   earlyswz  B operands of the first block requested before the interchange flag is known (redone after a swap)
    #   hoist     address arithmetic of the factorisation placed in LDS-latency shadows instead of the dependent chain
    #   ldsconst  the pivot row's lane gets its zeroed row and its unit multiplier by LDS reads of constants instead of
    #             eight exec-masked VALU instructions per pivot column
    #   rlpiv     the pivot row travels through twelve v_readlane into SGPRs (FMA scalar operands) instead of an LDS
    #             publish / broadcast round trip per pivot column; excludes ldsconst
    #   spec      (with rlpiv) the four pivot columns of a panel run without a single branch: "some row beats the
    #             diagonal" and "pivot not > 0" only set a flag (SALU), tested once per panel; a flagged panel is redone
    #             from its LDS copy by the careful column code (v_cmp -> s_cbranch -> VALU costs ~42 cycles, twice per
    #             column, on the workgroup's critical path)
    #   preaddr   the factorisation's LDS addresses are computed one step ahead, by the wave that will factor next, while
    #             it waits for the current factorisation (instead of right behind its barrier)
    # Same-box A/B (profiles/r03_k3_ab_notes.md): none of the variants moves K3 by more than 1 %; rlpiv is the fastest
    # by that margin and keeps LDS traffic out of the column loop; spec / preaddr / ldsconst measured neutral or slower.
    DEFAULT_OPTS = ("earlyswz", "hoist", "rlpiv")

    def __init__(self, for_text: bool, build_af: bool = True, with_slow: bool = True, opts=None):
        self.opts = set(self.DEFAULT_OPTS if opts is None else opts)
        self.stamps = "stamps" in self.opts        # diagnostic build: s_memtime sums per phase (see stamp())
        if "rlpiv" in self.opts:
            self.opts.discard("ldsconst")
        else:
            self.opts.discard("spec")
        self.p = Program()
        self.for_text = for_text
        self.build_af = build_af
        self.with_slow = with_slow
        self.cold = []           # deferred cold-path emitters (placed after the main stream)
        if for_text:             # %32.. are the inputs of the asm statement (outputs %0..%31 are the accumulators)
            self.in_arx, self.in_tw = Reg("op", 32, 2), Reg("op", 33, 2)
            self.in_p, self.in_w, self.in_tau, self.in_lds = Reg("op", 34), Reg("op", 35), Reg("op", 36, 2), Reg("op", 37)
        else:                    # the emulator places them in s[2:11]
            self.in_arx, self.in_tw = S(2, 2), S(4, 2)
            self.in_p, self.in_w, self.in_tau, self.in_lds = S(6), S(7), S(8, 2), S(10)

    # ------------------------------------------------------------------ diagnostic stamps
    # Eight 32-bit sums of s_memtime differences in s[92:99] (s90 = last stamp, s[100:101] scratch); every stamp drains
    # lgkmcnt (s_memtime returns through it).  Phases: 0 A(f) build | 1 barrier wait | 2 loop, waves that do not factor |
    # 3 chain wave: barrier -> N and flag read | 4 chain wave: B operands + 16 MFMAs | 5 factorisation: panel through LDS
    # | 6 factorisation: pivot search, reciprocal, pivot row through LDS | 7 factorisation: elimination, N write
    def stamp(self, phase):
        if not self.stamps:
            return
        p = self.p
        p.s_memtime(S(100, 2))
        p.s_waitcnt("lgkmcnt(0)")
        p.s_sub_u32(S(101), S(100), S(90))
        p.s_add_u32(S(92 + phase), S(92 + phase), S(101))
        p.s_mov_b32(S(90), S(100))

    def stamp_init(self):
        if not self.stamps:
            return
        p = self.p
        p.s_memtime(S(100, 2))
        for k in range(4):
            p.s_mov_b64(S(92 + 2 * k, 2), 0)
        p.s_waitcnt("lgkmcnt(0)")
        p.s_mov_b32(S(90), S(100))

    def setprio(self, n):
        if "noprio" not in self.opts:
            self.p.s_setprio(n)

    # ------------------------------------------------------------------ prologue: constants
    def prologue(self):
        p = self.p
        p.s_mov_b32(sW, self.in_w)
        p.s_mov_b32(sLDS, self.in_lds)
        p.s_mov_b64(sTAU, self.in_tau)
        p.s_mov_b64(sARX, self.in_arx)
        p.s_mov_b64(sTW, self.in_tw)
        p.s_mov_b32(sP, self.in_p)
        p.v_mbcnt_lo_u32_b32(vLANE, -1, 0)
        p.v_mbcnt_hi_u32_b32(vLANE, -1, vLANE)

    def lane_constants(self):
        """masks and addresses of the main loop (after the twiddle SGPRs are free)"""
        p = self.p
        p.v_lshrrev_b32_e32(vT0, 4, vLANE)                     # i
        p.v_and_b32_e32(vT1, 3, vLANE)                         # j
        p.v_bfe_u32(vT2, vLANE, 2, 2)                          # b
        p.v_cmp_eq_u32_e64(sIJ, vT0, vT1)
        for b in range(4):
            p.v_cmp_eq_u32_e64(sB[b], b, vT2)
        # A-operand address: ((l & 15) * 4 + (l >> 4)) * 16 + base
        p.v_and_b32_e32(vT1, 15, vLANE)
        p.v_lshlrev_b32_e32(vT0, 4, vT0)
        p.v_lshl_add_u32(vAN, vT1, 6, vT0)
        p.v_add_u32_e32(vAN, sLDS, vAN)

    # ------------------------------------------------------------------ A(f)
    def build_a(self):
        """acc = I - sum_k a_k z_k.  Coefficients: packed layout of ar_pack_kernel (block B = 4 Ig + Jl, lag pair h:
        64 lanes x 16 B contiguous at ((B * P2 + h) * 64 + lane) * 16).  Same FMA order as tf_inv.hip:326-336."""
        p = self.p
        vOFF = V(95)
        p.v_lshlrev_b32_e32(vOFF, 4, vLANE)
        p.s_add_i32(sP2, sP, 1)
        p.s_lshr_b32(sP2, sP2, 1)
        p.s_lshl_b32(sSTRIDE, sP2, 10)                          # bytes between blocks
        # identity lanes: b == w and i == j
        p.v_lshrrev_b32_e32(vT0, 4, vLANE)
        p.v_and_b32_e32(vT1, 3, vLANE)
        p.v_bfe_u32(vT2, vLANE, 2, 2)
        p.v_cmp_eq_u32_e64(sDIAG, vT0, vT1)
        p.v_cmp_eq_u32_e64(sM1, sW, vT2)
        p.s_and_b64(sDIAG, sDIAG, sM1)
        p.v_mov_b32_e32(vT3, 0x3FF00000)
        for Ig in range(4):                                     # the four diagonal blocks start from the identity
            q = ACC(Ig, Ig)
            p.v_mov_b32_e32(q.sub(0), 0)
            p.v_cndmask_b32_e64(q.sub(1), 0, vT3, sDIAG)
            p.v_mov_b64_e32(IM(q), 0)
        p.s_lshr_b32(sCNT, sP2, 2)                              # full chunks of four lag pairs
        p.s_mov_b32(sCH, 0)
        first, more, tail, done = (p.newlabel(n) for n in ("AF_first", "AF_more", "AF_tail", "AF_done"))
        p.s_cmp_lg_u32(sCNT, 0)
        p.s_cbranch_scc1(first)
        for b in range(16):                                     # no full chunk: the off-diagonal blocks start from zero
            if b % 5:
                p.v_mov_b64_e32(RE(V(4 * b, 4)), 0)
                p.v_mov_b64_e32(IM(V(4 * b, 4)), 0)
        p.s_branch(tail)
        p.label(first)
        self._af_chunk(4, fresh=True)
        p.s_add_i32(sCH, sCH, 1)
        p.label(more)
        p.s_cmp_ge_u32(sCH, sCNT)
        p.s_cbranch_scc1(tail)
        self._af_chunk(4, fresh=False)
        p.s_add_i32(sCH, sCH, 1)
        p.s_branch(more)
        # ---- remaining single lag pairs
        p.label(tail)
        p.s_lshl_b32(sCH, sCNT, 2)                              # pair index h0 = 4 * chunks
        tl = p.newlabel("AF_tail_loop")
        p.label(tl)
        p.s_cmp_ge_u32(sCH, sP2)
        p.s_cbranch_scc1(done)
        self._af_chunk(1, fresh=False)
        p.s_add_i32(sCH, sCH, 1)
        p.s_branch(tl)
        p.label(done)

    def _af_chunk(self, HC, fresh):
        """One pass over the 16 blocks for HC lag pairs; sCH = chunk index (HC == 4) or pair index (HC == 1).  fresh:
        the off-diagonal blocks have not been touched yet -- their first FMA takes a zero addend, and until then their
        registers serve as load buffers (up to 20 loads of 1 KB per wave in flight instead of 7)."""
        p = self.p
        vOFF = V(95)
        # twiddles of the chunk: 16 B per lag
        p.s_lshl_b32(sA1, sCH, 7 if HC == 4 else 5)
        p.s_add_u32(sTWP.sub(0), sTW.sub(0), sA1)
        p.s_addc_u32(sTWP.sub(1), sTW.sub(1), 0)
        if HC == 4:
            for k in range(4):
                p.s_load_dwordx8(S(TW0 + 8 * k, 8), sTWP, 32 * k)
        else:
            p.s_load_dwordx4(S(TW0, 4), sTWP, 0)
            lbl = p.newlabel("AF_odd")                           # the second lag of the pair exists only if 2 h0 + 1 < p
            p.s_lshl_b32(sA1, sCH, 1)
            p.s_add_i32(sA1, sA1, 1)
            p.s_cmp_ge_u32(sA1, sP)
            p.s_cbranch_scc1(lbl)
            p.s_load_dwordx4(S(TW0 + 4, 4), sTWP, 16)
            p.label(lbl)
        # issue pointer: arx + h0 * 1024 (+ block * stride as the issue advances)
        p.s_lshl_b32(sA1, sCH, 12 if HC == 4 else 10)
        p.s_add_u32(sPTR.sub(0), sARX.sub(0), sA1)
        p.s_addc_u32(sPTR.sub(1), sARX.sub(1), 0)
        n_loads = 16 * HC
        free_t = [V(64 + 4 * k, 4) for k in range(7)]            # v64..v91
        pending = []                                             # (load index, quad, accumulator block lent) in issue order
        acc_buf_free = {b: True for b in range(16)}
        DIAG = {0, 5, 10, 15}
        state = {"issued": 0, "blk": 0}
        MAXFLY = 20

        def blk(n):
            return n // HC

        def alloc(m):
            if fresh:
                for b in range(16):
                    if b > blk(m) and b not in DIAG and acc_buf_free[b]:
                        acc_buf_free[b] = False
                        return V(4 * b, 4), b
            if free_t:
                return free_t.pop(0), None
            return None

        def issue_some():
            while state["issued"] < n_loads and len(pending) < MAXFLY:
                got = alloc(state["issued"])
                if got is None:
                    return
                q, b = got
                m = state["issued"]
                if blk(m) != state["blk"]:
                    p.s_add_u32(sPTR.sub(0), sPTR.sub(0), sSTRIDE)
                    p.s_addc_u32(sPTR.sub(1), sPTR.sub(1), 0)
                    state["blk"] = blk(m)
                h = m % HC
                p.global_load_dwordx4(q, vOFF, sPTR, mods=f"offset:{1024 * h}" if h else "")
                pending.append((m, q, b))
                state["issued"] += 1

        issue_some()
        p.s_waitcnt("lgkmcnt(0)")                                # twiddles
        if HC == 1:                                              # an odd order's padding lag: zero twiddle
            lbl = p.newlabel("AF_even")
            p.s_lshl_b32(sA1, sCH, 1)
            p.s_add_i32(sA1, sA1, 1)
            p.s_cmp_lt_u32(sA1, sP)
            p.s_cbranch_scc1(lbl)
            p.s_mov_b64(S(TW0 + 4, 2), 0)
            p.s_mov_b64(S(TW0 + 6, 2), 0)
            p.label(lbl)
        touched = set(DIAG) if fresh else set(range(16))
        for n in range(n_loads):
            idx, q, b = pending.pop(0)
            assert idx == n
            p.s_waitcnt(f"vmcnt({len(pending)})")
            B, h = blk(n), n % HC
            acc = V(4 * B, 4)
            zr0, zi0 = S(TW0 + 8 * h, 2), S(TW0 + 8 * h + 2, 2)
            zr1, zi1 = S(TW0 + 8 * h + 4, 2), S(TW0 + 8 * h + 6, 2)
            vx, vy = q.sub(0, 2), q.sub(2, 2)
            if B in touched:
                p.v_fma_f64(RE(acc), Neg(vx), zr0, RE(acc))
                p.v_fma_f64(IM(acc), Neg(vx), zi0, IM(acc))
            else:
                p.v_fma_f64(RE(acc), Neg(vx), zr0, 0)
                p.v_fma_f64(IM(acc), Neg(vx), zi0, 0)
                touched.add(B)
            p.v_fma_f64(RE(acc), Neg(vy), zr1, RE(acc))
            p.v_fma_f64(IM(acc), Neg(vy), zi1, IM(acc))
            if b is None:
                free_t.append(q)
            issue_some()
        assert not pending and state["issued"] == n_loads

    # ------------------------------------------------------------------ update pieces
    def load_n(self, t, quads=NQ):
        p = self.p
        for Ig in range(4):
            p.ds_read_b128(quads[Ig], vAN, mods=f"offset:{NBUF + (t % NR) * 4096 + Ig * 1024}")

    def fix_n(self, t, quads=NQ):
        """N_t - E_S on the A operand of row group t >> 2: lanes (b == t & 3, i == j) subtract one."""
        p = self.p
        p.s_and_b64(EXEC, sIJ, sB[t & 3])
        p.v_add_f64(RE(quads[t >> 2]), RE(quads[t >> 2]), -1.0)
        p.s_mov_b64(EXEC, -1)

    def swizzle_b(self, t, J, u):
        """pivot rows 4t .. 4t+3 at this lane's column: lane (i, b, j) <- lane (i, t & 3, j) of block (t >> 2, J)"""
        p = self.p
        src = ACC(t >> 2, J)
        pat = 0x13 | (((t & 3) << 2) << 5)
        for k in range(4):
            p.ds_swizzle_b32(u.sub(k), src.sub(k), mods=f"offset:{pat:#x}")

    def mfma_block(self, J, u, quads=NQ):
        p = self.p
        ur, ui = u.sub(0, 2), u.sub(2, 2)
        for Ig in range(4):
            a = ACC(Ig, J)
            p.v_mfma_f64_4x4x4_4b_f64(RE(a), RE(quads[Ig]), ur, RE(a))
            p.v_mfma_f64_4x4x4_4b_f64(IM(a), RE(quads[Ig]), ui, IM(a))
        for Ig in range(4):
            a = ACC(Ig, J)
            p.v_mfma_f64_4x4x4_4b_f64(RE(a), IM(quads[Ig]), ui, RE(a), mods="neg:[1,0,0]")
            p.v_mfma_f64_4x4x4_4b_f64(IM(a), IM(quads[Ig]), ur, IM(a))

    # ------------------------------------------------------------------ one step
    def step(self, s):
        p = self.p
        own, Js = s % NT, s // NT
        has_next = s + 1 < NSTEP
        nxt, Jn = (s + 1) % NT, (s + 1) // NT
        L_own, L_nxt, L_end = f"S{s}_own", f"S{s}_nxt", f"S{s}_end"
        p.s_cmp_eq_u32(sW, own)
        p.s_cbranch_scc1(L_own)
        if has_next:
            p.s_cmp_eq_u32(sW, nxt)
            p.s_cbranch_scc1(L_nxt)
        # ---- other waves: update s of all four blocks
        self.role_body(s, "oth", [0, 1, 2, 3])
        if "preaddr" in self.opts and s + 2 < NSTEP:
            # the wave that factors panel s + 2 during step s + 1 is one of these: its addresses, while it would wait
            L_pa = p.newlabel(f"S{s}_pa")
            p.s_cmp_lg_u32(sW, (s + 2) % NT)
            p.s_cbranch_scc1(L_pa)
            self.factor_addresses()
            p.label(L_pa)
        p.s_branch(L_end)
        # ---- owner of panel s
        p.label(L_own)
        if s >= 1:                     # update s - 1 of the blocks this wave left behind while it factored panel s
            t = s - 1
            blocks = [J for J in range(4) if J != Js]
            self.load_n(t)
            self.swizzle_b(t, blocks[0], U[0])
            p.s_waitcnt("lgkmcnt(4)")
            self.fix_n(t)
            for k, J in enumerate(blocks):
                if k + 1 < len(blocks):
                    self.swizzle_b(t, blocks[k + 1], U[(k + 1) & 1])
                    p.s_waitcnt("lgkmcnt(4)")
                else:
                    p.s_waitcnt("lgkmcnt(0)")
                self.mfma_block(J, U[k & 1])
        self.role_body(s, "own", [J for J in range(4) if J != Js], take=Js)
        if has_next:
            p.s_branch(L_end)
            # ---- owner of panel s + 1: its panel block only, then the factorisation
            p.label(L_nxt)
            self.setprio(3)
            pre = "preaddr" in self.opts
            self.role_body(s, "nxt", [Jn], hook=self.factor_addresses if ("hoist" in self.opts and not pre) else None)
            self.stamp(4)
            self.factor_panel(s + 1, Jn, have_addresses="hoist" in self.opts or pre)
            self.setprio(0)
        p.label(L_end)
        p.s_waitcnt("lgkmcnt(0)")
        self.stamp(2)
        p.s_barrier()
        self.stamp(1)
        roles = [("own", own, [J for J in range(4) if J != Js][0])] + ([("nxt", nxt, Jn)] if has_next else [])
        if self.with_slow:
            self.cold.append(lambda: self.interchange_cold(s, f"S{s}_swap", roles))
        else:
            self.cold.append(lambda: (p.label(f"S{s}_swap"), p.s_endpgm()))

    def role_body(self, s, role, blocks, take=None, hook=None):
        """load N_s, check the interchange flag, (take the panel), update `blocks`"""
        p = self.p
        early = "earlyswz" in self.opts and bool(blocks)
        L_slow, L_back = f"S{s}_swap", f"S{s}_{role}_swapped"
        self.load_n(s)
        p.v_mov_b32_e32(vSWP, sLDS)
        p.ds_read_b32(vSWP, vSWP, mods=f"offset:{SSWP + (s % NR) * 32}")
        if early:
            self.swizzle_b(s, blocks[0], U[0])
            if hook:
                hook()
            p.s_waitcnt("lgkmcnt(4)")
        else:
            p.s_waitcnt("lgkmcnt(0)")
        if role == "nxt":
            self.stamp(3)
        p.v_cmp_eq_u32_e32(VCC, 0, vSWP)
        p.s_cbranch_vccz(L_slow)
        p.label(L_back)
        if take is not None:
            self.take_addr()
            for Ig in range(4):            # panel block <- N_s (rows 16 Ig + 4 b + i, column j)
                p.ds_read_b128(ACC(Ig, take), vT1, mods=f"offset:{NBUF + (s % NR) * 4096 + Ig * 1024}")
        for k, J in enumerate(blocks):
            if k == 0:
                if not early:
                    self.swizzle_b(s, J, U[0])
                    if hook:
                        hook()
                self.fix_n(s)
            if k + 1 < len(blocks):
                self.swizzle_b(s, blocks[k + 1], U[(k + 1) & 1])
                p.s_waitcnt("lgkmcnt(4)")
            else:
                if len(blocks) == 1:
                    p.s_nop(0)             # VALU write of the A operand (fix_n) -> MFMA: two wait states
                p.s_waitcnt("lgkmcnt(0)")
            self.mfma_block(J, U[k & 1])
        if not blocks:
            p.s_waitcnt("lgkmcnt(0)")

    # ------------------------------------------------------------------ row interchanges of step t (cold)
    def interchange_cold(self, t, L_slow, roles):
        p = self.p
        Igp, bp = t >> 2, t & 3
        vBASE, q = V(93), V(89, 4)       # v89..v92 distances
        p.label(L_slow)
        p.v_mov_b32_e32(vBASE, sLDS)
        # the record is {any, d0, d1, d2, d3}: d0..d3 are not 16-byte aligned -> four dword reads
        for k in range(4):
            p.ds_read_b32(q.sub(k), vBASE, mods=f"offset:{SSWP + (t % NR) * 32 + 4 + 4 * k}")
        p.s_waitcnt("lgkmcnt(0)")
        sD = [sT0, sT1, sT2, sTMP]
        for k in range(4):
            p.v_readfirstlane_b32(sD[k], q.sub(k))
        # this lane's slot in the wave's swap area: base + SWAPB + w * 512 + (l & 3) * 16
        vSW = V(93)
        p.v_and_b32_e32(vSW, 3, vLANE)
        p.v_lshlrev_b32_e32(vSW, 4, vSW)
        p.s_lshl_b32(sIGS, sW, 9)
        p.s_add_i32(sIGS, sIGS, sLDS)
        p.s_nop(0)
        p.v_add_u32_e32(vSW, sIGS, vSW)
        for jj in range(4):
            col = 4 * t + jj
            L_skip = p.newlabel(f"X{t}_{jj}_skip")
            p.s_cmp_eq_u32(sD[jj], 0)
            p.s_cbranch_scc1(L_skip)
            p.s_add_i32(sRSTAR, sD[jj], col)
            # lanes holding row `col`: i == jj, b == bp  (static);  lanes holding row rstar: i == rstar & 3, b == (rstar >> 2) & 3
            p.s_lshl_b64(sM1, 15, 16 * jj + 4 * bp)
            p.s_mov_b64(EXEC, sM1)
            for Jl in range(4):
                p.ds_write_b128(vSW, ACC(Igp, Jl), mods=f"offset:{SWAPB + 64 * Jl}")
            p.s_and_b32(sKMAX, sRSTAR, 15)                  # 4 b + i of rstar -> lane base 16 i + 4 b
            p.s_and_b32(sIGS, sKMAX, 3)
            p.s_lshl_b32(sIGS, sIGS, 4)
            p.s_lshr_b32(sKMAX, sKMAX, 2)
            p.s_lshl_b32(sKMAX, sKMAX, 2)
            p.s_add_i32(sKMAX, sKMAX, sIGS)
            p.s_lshl_b64(sM1, 15, sKMAX)
            p.s_mov_b64(EXEC, sM1)
            p.s_lshr_b32(sIGS, sRSTAR, 4)
            L_done = p.newlabel(f"X{t}_{jj}_done")
            for Ig in range(Igp, 4):
                L_n = p.newlabel(f"X{t}_{jj}_n")
                if Ig < 3:
                    p.s_cmp_lg_u32(sIGS, Ig)
                    p.s_cbranch_scc1(L_n)
                for Jl in range(4):
                    p.ds_write_b128(vSW, ACC(Ig, Jl), mods=f"offset:{SWAPB + 256 + 64 * Jl}")
                for Jl in range(4):
                    p.ds_read_b128(ACC(Ig, Jl), vSW, mods=f"offset:{SWAPB + 64 * Jl}")
                if Ig < 3:
                    p.s_branch(L_done)
                    p.label(L_n)
            p.label(L_done)
            p.s_waitcnt("lgkmcnt(0)")
            p.s_lshl_b64(sM1, 15, 16 * jj + 4 * bp)
            p.s_mov_b64(EXEC, sM1)
            for Jl in range(4):
                p.ds_read_b128(ACC(Igp, Jl), vSW, mods=f"offset:{SWAPB + 256 + 64 * Jl}")
            p.s_mov_b64(EXEC, -1)
            p.s_waitcnt("lgkmcnt(0)")
            p.label(L_skip)
        labels = []                      # back to where this wave came from (B operands requested early are stale now)
        for role, wv, J0 in roles:
            L = p.newlabel(f"X{t}_{role}")
            labels.append((L, role, J0))
            p.s_cmp_eq_u32(sW, wv)
            p.s_cbranch_scc1(L)
        if "earlyswz" in self.opts:
            self.swizzle_b(t, 0, U[0])
            p.s_waitcnt("lgkmcnt(0)")
        p.s_branch(f"S{t}_oth_swapped")
        for L, role, J0 in labels:
            p.label(L)
            if "earlyswz" in self.opts:
                self.swizzle_b(t, J0, U[0])
                p.s_waitcnt("lgkmcnt(0)")
            if role == "nxt" and ("hoist" in self.opts or "preaddr" in self.opts):
                self.factor_addresses()          # v89..v92 held them; the distances went through the same registers
            p.s_branch(f"S{t}_{role}_swapped")

    def take_addr(self):
        """vT1 = base + rowl * 64 + j * 16, rowl = 4 b + i  (panel block rows in the X layout)"""
        p = self.p
        p.v_lshrrev_b32_e32(vT2, 4, vLANE)                  # i
        p.v_bfe_u32(vT3, vLANE, 2, 2)                       # b
        p.v_lshl_add_u32(vT2, vT3, 2, vT2)                  # rowl
        p.v_and_b32_e32(vT3, 3, vLANE)                      # j
        p.v_lshlrev_b32_e32(vT3, 4, vT3)
        p.v_lshl_add_u32(vT1, vT2, 6, vT3)
        p.v_add_u32_e32(vT1, sLDS, vT1)

    # ------------------------------------------------------------------ panel factorisation (one wave)
    def factor_addresses(self):
        """LDS addresses of the factorisation in v89..v92 (free between the flag read and the factorisation's own
        temporaries): panel write (X layout -> row-major, stride 80 B), lane-per-row read, N write, base."""
        p = self.p
        p.v_lshrrev_b32_e32(vT1, 4, vLANE)                  # i
        p.v_bfe_u32(vT2, vLANE, 2, 2)                       # b
        p.v_lshl_add_u32(vT1, vT2, 2, vT1)                  # rowl
        p.v_and_b32_e32(vT2, 3, vLANE)
        p.v_lshlrev_b32_e32(vT2, 4, vT2)                    # j * 16
        p.v_mul_u32_u24_e32(vT0, 80, vT1)
        p.v_add_u32_e32(vT0, vT0, vT2)
        p.v_add_u32_e32(vT0, sLDS, vT0)                     # vT0: Pbuf[(16 Ig + rowl) * 5 + j]
        p.v_mul_u32_u24_e32(vT1, 80, vLANE)
        p.v_add_u32_e32(vT1, sLDS, vT1)                     # vT1: Pbuf[lane * 5]
        p.v_lshl_add_u32(vT2, vLANE, 6, sLDS)               # vT2: Nbuf[lane * 4]
        p.v_mov_b32_e32(vT3, sLDS)                          # vT3: base

    def write_constants(self):
        """16 B of zeros and the pair (-1.0, 0.0) in the (otherwise unused) SROW region, by the wave that factors panel 0"""
        p = self.p
        p.v_mov_b32_e32(vT0, sLDS)
        for k in range(4):
            p.v_mov_b32_e32(V(80 + k), 0)
        p.ds_write_b128(vT0, V(80, 4), mods=f"offset:{SROW}")
        p.v_mov_b32_e32(V(84), 0)
        p.v_mov_b32_e32(V(85), 0xBFF00000)
        p.v_mov_b32_e32(V(86), 0)
        p.v_mov_b32_e32(V(87), 0)
        p.ds_write_b128(vT0, V(84, 4), mods=f"offset:{SROW + 16}")

    def factor_panel(self, t, J, have_addresses=False):
        """Panel t = register block J of this wave -> LDS -> lane per row; four pivot steps; N_t and the interchange
        record -> LDS.  Mirrors factor_panel of tf_inv.hip:368-508 operation for operation."""
        p = self.p
        P = [ACC(Ig, J) for Ig in range(4)]                 # dead until take_panel reloads them next step
        PV = P[:3]                                          # pivot row quads (three columns)
        vPR, vNW, vZ, vZERO = P[3].sub(0), P[3].sub(1), P[3].sub(2), P[3].sub(3)
        x = NQ
        if not have_addresses:
            self.factor_addresses()
        # (the panel block comes straight out of the matrix pipe: nine wait states between an MFMA and an LDS store of
        # its result -- the scalar initialisations and one s_nop fill them)
        p.s_mov_b32(sBAD, 0)
        p.s_mov_b32(S(67), 0)
        for k in range(4):
            p.s_mov_b32(sRS[k], 0)
        for Ig in range(4):
            if Ig == 3 and have_addresses:
                p.s_nop(0)
            p.ds_write_b128(vT0, P[Ig], mods=f"offset:{PBUF + Ig * 1280}")
        for jj in range(4):
            p.ds_read_b128(x[jj], vT1, mods=f"offset:{PBUF + 16 * jj}")
        # the panel block's registers are free now: addresses and constants live there
        p.v_mov_b32_e32(vPR, vT1)
        p.v_mov_b32_e32(vNW, vT2)
        p.v_mov_b32_e32(vZ, vT3)
        p.v_mov_b32_e32(vZERO, 0)
        p.s_waitcnt("lgkmcnt(0)")
        self.stamp(5)
        if "spec" in self.opts:
            sFLAG = S(58, 2)
            L_redo, L_cols = f"F{t}_redo", f"F{t}_cols"
            p.s_mov_b64(sFLAG, 0)
            for jj in range(4):
                self.pivot_column_rl(t, jj, x, vPR, vZ, spec=True)
            p.s_cmp_lg_u64(sFLAG, 0)
            p.s_cbranch_scc1(L_redo)
            p.label(L_cols)

            def redo():
                # some column wanted a row interchange (or met a pivot that is not > 0): the panel again, from the copy
                # the transposition left in LDS (the speculative columns touch registers only), by the careful code
                p.label(L_redo)
                for jj in range(4):
                    p.ds_read_b128(x[jj], vPR, mods=f"offset:{PBUF + 16 * jj}")
                p.s_waitcnt("lgkmcnt(0)")
                for jj in range(4):
                    self.pivot_column_rl(t, jj, x, vPR, vZ, spec=False, tag="r")
                p.s_branch(L_cols)
            self.cold.append(redo)
        else:
            for jj in range(4):
                self.pivot_column(t, jj, x, PV, vPR, vZ)
        # N_t
        for jj in range(4):
            p.ds_write_b128(vNW, x[jj], mods=f"offset:{NBUF + (t % NR) * 4096 + 16 * jj}")
        # interchange record {any, d0..d3}
        L_rec, L_recd = f"F{t}_rec", f"F{t}_recd"
        p.s_or_b32(sT0, sRS[0], sRS[1])
        p.s_or_b32(sT1, sRS[2], sRS[3])
        p.s_or_b32(sT0, sT0, sT1)
        p.s_cmp_lg_u32(sT0, 0)
        p.s_cbranch_scc1(L_rec)
        p.ds_write_b32(vZ, vZERO, mods=f"offset:{SSWP + (t % NR) * 32}")
        p.label(L_recd)
        # zero / NaN pivot: one test per panel
        L_bad, L_badd = f"F{t}_bad", f"F{t}_badd"
        if "rlpiv" in self.opts:
            p.s_cmp_lg_u32(sBAD, 0)
            p.s_cbranch_scc1(L_bad)
        else:
            p.s_and_b64(sM1, sOKC[0], sOKC[1])
            p.s_and_b64(sM1, sM1, sOKC[2])
            p.s_and_b64(sM1, sM1, sOKC[3])
            p.s_cmp_eq_u64(sM1, EXEC)
            p.s_cbranch_scc0(L_bad)
        p.label(L_badd)
        self.stamp(7)

        def cold():
            p.label(L_rec)
            p.v_mov_b32_e32(vT0, 1)
            p.ds_write_b32(vZ, vT0, mods=f"offset:{SSWP + (t % NR) * 32}")
            for k in range(4):
                p.v_mov_b32_e32(vT0, sRS[k])
                p.ds_write_b32(vZ, vT0, mods=f"offset:{SSWP + (t % NR) * 32 + 4 + 4 * k}")
            p.s_branch(L_recd)
            p.label(L_bad)
            # bad = 1 + first column whose |pivot|^2 is not > 0; recorded if info is still 0
            if "rlpiv" in self.opts:
                p.s_mov_b32(sT0, sBAD)
            else:
                p.s_mov_b32(sT0, 4 * t + 4)
                for k in (2, 1, 0):
                    L = p.newlabel(f"F{t}_b")
                    p.s_cmp_eq_u64(sOKC[k], EXEC)
                    p.s_cbranch_scc1(L)
                    p.s_mov_b32(sT0, 4 * t + k + 1)
                    p.label(L)
            p.ds_read_b32(vT0, vZ, mods=f"offset:{SINFO}")
            p.s_waitcnt("lgkmcnt(0)")
            p.v_readfirstlane_b32(sT1, vT0)
            p.s_cmp_lg_u32(sT1, 0)
            p.s_cbranch_scc1(L_badd)
            p.v_mov_b32_e32(vT0, sT0)
            p.ds_write_b32(vZ, vT0, mods=f"offset:{SINFO}")
            p.s_branch(L_badd)
        self.cold.append(cold)

    def reciprocal(self):
        """1 / (pr + i pi) = (pr - i pi) / dd, dd = pr^2 + pi^2: v_rcp_f64 seed and two Newton steps, the second folded
        into the products (tf_inv.hip:411-418)."""
        p = self.p
        p.v_mul_f64(fDD, sPI, sPI)
        p.v_fma_f64(fDD, sPR, sPR, fDD)
        p.v_rcp_f64_e32(fY, fDD)
        p.s_nop(0)
        p.v_fma_f64(fE, Neg(fDD), fY, 1.0)
        p.v_fma_f64(fY, fE, fY, fY)
        p.v_fma_f64(fE, Neg(fDD), fY, 1.0)
        p.v_mul_f64(fIVR, sPR, fY)
        p.v_mul_f64(fIVI, fY, Neg(sPI))
        p.v_fma_f64(fIVR, fIVR, fE, fIVR)
        p.v_fma_f64(fIVI, fIVI, fE, fIVI)

    def constants_to_pivot_lane(self, col, jj, x, vZ):
        """lane `col` <- zeroed row and the multiplier source (-1, 0): its eliminated row is then 1/pivot times the pivot
        row and its column jj is 1/pivot, from the same instructions as every other lane's"""
        p = self.p
        p.s_lshl_b64(EXEC, 1, col)
        for j2 in range(4):
            p.ds_read_b128(x[j2], vZ, mods=f"offset:{SROW + (16 if j2 == jj else 0)}")
        p.s_mov_b64(EXEC, -1)

    def pivot_column_rl(self, t, jj, x, vPR, vZ, spec=False, tag=""):
        """pivot column with the pivot row in SGPRs (v_readlane): no LDS traffic on the common path.  The readlanes are
        independent of the reciprocal's dependent chain and are placed in its gaps (in-order issue)."""
        p = self.p
        col = 4 * t + jj
        others = [j2 for j2 in range(4) if j2 != jj]
        spv = {j2: (PVS[2 * k], PVS[2 * k + 1]) for k, j2 in enumerate(others)}
        xr, xi = RE(x[jj]), IM(x[jj])
        L_search, L_elim, L_badc, L_badcd = (f"F{t}_{jj}{tag}_{n}" for n in ("search", "elim", "badc", "badcd"))
        sFLAG = S(58, 2)
        rl = []                                              # the twelve pivot-row readlanes, issued in the gaps below
        for j2 in others:
            for half, src in ((0, RE(x[j2])), (1, IM(x[j2]))):
                for d in range(2):
                    rl.append((spv[j2][half].sub(d), src.sub(d)))

        def gap(n):
            for _ in range(n):
                if rl:
                    dst, src = rl.pop(0)
                    p.v_readlane_b32(dst, src, col)
        p.v_readlane_b32(sPR.sub(0), xr.sub(0), col)
        p.v_readlane_b32(sPR.sub(1), xr.sub(1), col)
        p.v_readlane_b32(sPI.sub(0), xi.sub(0), col)
        p.v_readlane_b32(sPI.sub(1), xi.sub(1), col)
        p.v_add_f64(fCAND, Abs(xr), Abs(xi))
        gap(1)
        p.v_mov_b64_e32(fDC, sPI)
        p.v_mul_f64(fDD, sPI, sPI)
        gap(1)
        p.v_add_f64(fDC, Abs(sPR), Abs(fDC))
        p.v_fma_f64(fDD, sPR, sPR, fDD)
        gap(1)
        p.v_mul_f64(fCAND, sTAU, fCAND)
        p.v_rcp_f64_e32(fY, fDD)
        gap(2)
        p.s_lshl_b64(sVALID, -1, col)
        p.v_cmp_gt_f64_e32(VCC, fCAND, fDC)
        p.v_fma_f64(fE, Neg(fDD), fY, 1.0)
        gap(2)
        p.s_and_b64(VCC, VCC, sVALID)
        p.v_fma_f64(fY, fE, fY, fY)
        gap(1)
        if spec:
            p.s_or_b64(sFLAG, sFLAG, VCC)
        else:
            p.s_cbranch_vccnz(L_search)
        p.v_fma_f64(fE, Neg(fDD), fY, 1.0)
        gap(2)
        p.v_mul_f64(fIVR, sPR, fY)
        p.v_mul_f64(fIVI, fY, Neg(sPI))
        gap(2)
        p.v_fma_f64(fIVR, fIVR, fE, fIVR)
        p.v_fma_f64(fIVI, fIVI, fE, fIVI)
        gap(12)
        if not spec:
            p.label(L_elim)
        p.v_cmp_lt_f64_e32(VCC, 0, fDD)
        p.v_mul_f64(fMR, xr, Neg(fIVR))
        p.v_mul_f64(fMI, xi, Neg(fIVR))
        if spec:
            p.s_andn2_b64(sM1, EXEC, VCC)
            p.s_or_b64(sFLAG, sFLAG, sM1)
        else:
            p.s_cbranch_vccz(L_badc)
            p.label(L_badcd)
        self.stamp(6)
        p.v_fma_f64(fMR, xi, fIVI, fMR)
        p.v_fma_f64(fMI, Neg(xr), fIVI, fMI)
        p.s_lshl_b64(EXEC, 1, col)
        p.v_mov_b64_e32(fMR, fIVR)
        p.v_mov_b64_e32(fMI, fIVI)
        for j2 in others:
            p.v_mul_f64(RE(x[j2]), RE(x[j2]), 0)
            p.v_mul_f64(IM(x[j2]), IM(x[j2]), 0)
        p.s_mov_b64(EXEC, -1)
        order = sorted(others, key=lambda j2: (j2 != jj + 1, j2))      # the next pivot column first
        for j2 in order:
            a = x[j2]
            pr_, pi_ = spv[j2]
            p.v_fma_f64(RE(a), fMR, pr_, RE(a))
            p.v_fma_f64(IM(a), fMR, pi_, IM(a))
            p.v_fma_f64(RE(a), Neg(fMI), pi_, RE(a))
            p.v_fma_f64(IM(a), fMI, pr_, IM(a))
        p.v_mov_b64_e32(xr, fMR)
        p.v_mov_b64_e32(xi, fMI)
        self.stamp(7)
        if spec:
            return

        def cold():
            # |pivot|^2 not > 0 (zero or NaN): remember the first such column (1-based), carry on
            p.label(L_badc)
            p.s_cmp_lg_u32(sBAD, 0)
            p.s_cbranch_scc1(L_badcd)
            p.s_mov_b32(sBAD, col + 1)
            p.s_branch(L_badcd)
        self.cold.append(cold)
        if self.with_slow:
            self.cold.append(lambda: self.search_cold(t, jj, x, spv, vPR, vZ, L_search, L_elim))
        else:
            self.cold.append(lambda: (p.label(L_search), p.s_endpgm()))

    def pivot_column(self, t, jj, x, PV, vPR, vZ):
        if "rlpiv" in self.opts:
            return self.pivot_column_rl(t, jj, x, vPR, vZ)
        p = self.p
        ldsconst = "ldsconst" in self.opts
        col = 4 * t + jj
        others = [j2 for j2 in range(4) if j2 != jj]
        pv = {j2: PV[k] for k, j2 in enumerate(others)}
        xr, xi = RE(x[jj]), IM(x[jj])
        L_search, L_elim = f"F{t}_{jj}_search", f"F{t}_{jj}_elim"
        # diagonal element (wave-uniform); every row's |re| + |im| (izamax metric) while lane `col` still holds its row
        p.v_readlane_b32(sPR.sub(0), xr.sub(0), col)
        p.v_readlane_b32(sPR.sub(1), xr.sub(1), col)
        p.v_readlane_b32(sPI.sub(0), xi.sub(0), col)
        p.v_readlane_b32(sPI.sub(1), xi.sub(1), col)
        p.v_add_f64(fCAND, Abs(xr), Abs(xi))
        # the pivot row (if the diagonal is kept): lane `col` publishes its other three columns, everybody reads them
        p.s_lshl_b64(EXEC, 1, col)
        for j2 in others:
            p.ds_write_b128(vZ, x[j2], mods=f"offset:{PBUF + col * 80 + 16 * j2}")
        if ldsconst:
            for j2 in range(4):
                p.ds_read_b128(x[j2], vZ, mods=f"offset:{SROW + (16 if j2 == jj else 0)}")
        p.s_mov_b64(EXEC, -1)
        for j2 in others:
            p.ds_read_b128(pv[j2], vZ, mods=f"offset:{PBUF + col * 80 + 16 * j2}")
        # candidates against the diagonal, reciprocal of the diagonal started at once
        p.v_mov_b64_e32(fDC, sPI)
        p.v_mul_f64(fDD, sPI, sPI)
        p.v_add_f64(fDC, Abs(sPR), Abs(fDC))
        p.v_fma_f64(fDD, sPR, sPR, fDD)
        p.v_mul_f64(fCAND, sTAU, fCAND)
        p.v_rcp_f64_e32(fY, fDD)
        p.s_lshl_b64(sVALID, -1, col)
        p.v_cmp_gt_f64_e32(VCC, fCAND, fDC)
        p.v_fma_f64(fE, Neg(fDD), fY, 1.0)
        p.s_and_b64(VCC, VCC, sVALID)
        p.v_fma_f64(fY, fE, fY, fY)
        p.s_cbranch_vccnz(L_search)
        p.v_fma_f64(fE, Neg(fDD), fY, 1.0)
        p.v_mul_f64(fIVR, sPR, fY)
        p.v_mul_f64(fIVI, fY, Neg(sPI))
        p.v_fma_f64(fIVR, fIVR, fE, fIVR)
        p.v_fma_f64(fIVI, fIVI, fE, fIVI)
        p.label(L_elim)
        p.v_cmp_gt_f64_e64(sOKC[jj], fDD, 0)
        if ldsconst:
            p.s_waitcnt("lgkmcnt(0)")
        self.stamp(6)
        # multiplier mu = -x_jj / pivot; the pivot row's lane takes 1 / pivot on a zeroed row
        p.v_mul_f64(fMR, xr, Neg(fIVR))
        p.v_mul_f64(fMI, xi, Neg(fIVR))
        p.v_fma_f64(fMR, xi, fIVI, fMR)
        p.v_fma_f64(fMI, Neg(xr), fIVI, fMI)
        if not ldsconst:
            p.s_lshl_b64(EXEC, 1, col)
            p.v_mov_b64_e32(fMR, fIVR)
            p.v_mov_b64_e32(fMI, fIVI)
            for j2 in others:
                p.v_mul_f64(RE(x[j2]), RE(x[j2]), 0)
                p.v_mul_f64(IM(x[j2]), IM(x[j2]), 0)
            p.s_mov_b64(EXEC, -1)
            p.s_waitcnt("lgkmcnt(0)")
        order = sorted(others, key=lambda j2: (j2 != jj + 1, j2))      # the next pivot column first
        for j2 in order:
            a = x[j2]
            p.v_fma_f64(RE(a), fMR, RE(pv[j2]), RE(a))
            p.v_fma_f64(IM(a), fMR, IM(pv[j2]), IM(a))
            p.v_fma_f64(RE(a), Neg(fMI), IM(pv[j2]), RE(a))
            p.v_fma_f64(IM(a), fMI, RE(pv[j2]), IM(a))
        p.v_mov_b64_e32(xr, fMR)
        p.v_mov_b64_e32(xi, fMI)
        self.stamp(7)
        if self.with_slow:
            self.cold.append(lambda: self.search_cold(t, jj, x, pv, vPR, vZ, L_search, L_elim))
        else:
            self.cold.append(lambda: (p.label(L_search), p.s_endpgm()))

    def search_cold(self, t, jj, x, pv, vPR, vZ, L_search, L_elim):
        """some row beats the diagonal: arg-max of the float-rounded |re|+|im| (lowest lane wins), interchange
        (tf_inv.hip:423-443)."""
        p = self.p
        col = 4 * t + jj
        xr, xi = RE(x[jj]), IM(x[jj])
        key = V(88)
        L_same = p.newlabel(f"F{t}_{jj}_same")
        p.label(L_search)
        if "ldsconst" in self.opts:
            # lane `col` already received the constants: give it its row back (columns j2 != jj from the published slot,
            # column jj = the diagonal element that sits in sPR / sPI)
            p.s_waitcnt("lgkmcnt(0)")
            p.s_lshl_b64(EXEC, 1, col)
            for j2 in pv:
                p.ds_read_b128(x[j2], vZ, mods=f"offset:{PBUF + col * 80 + 16 * j2}")
            p.v_mov_b64_e32(xr, sPR)
            p.v_mov_b64_e32(xi, sPI)
            p.s_mov_b64(EXEC, -1)
            p.s_waitcnt("lgkmcnt(0)")
        p.v_add_f64(fCAND, Abs(xr), Abs(xi))
        p.v_cvt_f32_f64_e32(key, fCAND)
        p.s_nop(1)
        p.v_cndmask_b32_e64(key, 0, key, sVALID)
        p.v_mov_b32_e32(V(89), key)
        p.s_nop(4)
        for ctl in ("quad_perm:[1,0,3,2]", "quad_perm:[2,3,0,1]", "row_half_mirror", "row_mirror"):
            p.v_max_u32_dpp(V(89), V(89), V(89), mods=f"{ctl} row_mask:0xf bank_mask:0xf")
            p.s_nop(1)
        p.v_max_u32_dpp(V(89), V(89), V(89), mods="row_bcast:15 row_mask:0xa bank_mask:0xf")
        p.s_nop(1)
        p.v_max_u32_dpp(V(89), V(89), V(89), mods="row_bcast:31 row_mask:0xc bank_mask:0xf")
        p.s_nop(1)
        p.v_readlane_b32(sKMAX, V(89), 63)
        p.s_nop(1)
        p.v_cmp_eq_u32_e32(VCC, sKMAX, key)
        p.s_and_b64(VCC, VCC, sVALID)
        p.s_ff1_i32_b64(sRSTAR, VCC)
        p.s_cmp_eq_u32(sRSTAR, col)
        p.s_cbranch_scc1(L_same)
        p.s_sub_i32(sRS[jj], sRSTAR, col)
        p.s_nop(3)
        p.v_readlane_b32(sPR.sub(0), xr.sub(0), sRSTAR)
        p.v_readlane_b32(sPR.sub(1), xr.sub(1), sRSTAR)
        p.v_readlane_b32(sPI.sub(0), xi.sub(0), sRSTAR)
        p.v_readlane_b32(sPI.sub(1), xi.sub(1), sRSTAR)
        p.s_nop(1)
        self.reciprocal()
        # every lane parks its row; the pivot row's lane takes the displaced row (slot col); orig[] follows
        for j2 in range(4):
            p.ds_write_b128(vPR, x[j2], mods=f"offset:{PBUF + 16 * j2}")
        p.s_lshl_b64(EXEC, 1, sRSTAR)
        for j2 in range(4):
            p.ds_read_b128(x[j2], vZ, mods=f"offset:{PBUF + col * 80 + 16 * j2}")
        p.s_mov_b64(EXEC, -1)
        p.s_lshl_b32(sT0, sRSTAR, 2)
        p.s_add_i32(sT0, sT0, sLDS)
        p.v_mov_b32_e32(V(88), sT0)
        p.ds_read_b32(V(89), vZ, mods=f"offset:{SORIG + 4 * col}")
        p.ds_read_b32(V(80), V(88), mods=f"offset:{SORIG}")
        p.s_waitcnt("lgkmcnt(0)")
        p.ds_write_b32(vZ, V(80), mods=f"offset:{SORIG + 4 * col}")
        p.ds_write_b32(V(88), V(89), mods=f"offset:{SORIG}")
        # the pivot row as it was parked
        p.s_mul_i32(sT0, sRSTAR, 80)
        p.s_add_i32(sT0, sT0, sLDS)
        p.v_mov_b32_e32(V(88), sT0)
        if "rlpiv" in self.opts:
            # into SGPRs: broadcast reads into the (dead) reciprocal temporaries, then readfirstlane
            for j2, (pr_, pi_) in pv.items():
                p.ds_read_b128(V(80, 4), V(88), mods=f"offset:{PBUF + 16 * j2}")
                p.s_waitcnt("lgkmcnt(0)")
                for d in range(2):
                    p.v_readfirstlane_b32(pr_.sub(d), V(80 + d))
                    p.v_readfirstlane_b32(pi_.sub(d), V(82 + d))
            pv = {}
        for j2 in pv:
            p.ds_read_b128(pv[j2], V(88), mods=f"offset:{PBUF + 16 * j2}")
        if "ldsconst" in self.opts:
            self.constants_to_pivot_lane(col, jj, x, vZ)
        p.s_waitcnt("lgkmcnt(0)")
        p.s_branch(L_elim)
        # the maximum sits on the diagonal after all (ties / threshold): finish the speculative reciprocal
        p.label(L_same)
        if "ldsconst" in self.opts:
            self.constants_to_pivot_lane(col, jj, x, vZ)
        if "rlpiv" in self.opts:          # the branch cut the pivot row's readlanes short: all twelve again
            for j2, (pr_, pi_) in pv.items():
                for d in range(2):
                    p.v_readlane_b32(pr_.sub(d), RE(x[j2]).sub(d), col)
                    p.v_readlane_b32(pi_.sub(d), IM(x[j2]).sub(d), col)
        p.v_fma_f64(fE, Neg(fDD), fY, 1.0)
        p.v_mul_f64(fIVR, sPR, fY)
        p.v_mul_f64(fIVI, fY, Neg(sPI))
        p.v_fma_f64(fIVR, fIVR, fE, fIVR)
        p.v_fma_f64(fIVI, fIVI, fE, fIVI)
        p.s_branch(L_elim)

    # ------------------------------------------------------------------ whole body
    def build(self):
        p = self.p
        self.prologue()
        self.stamp_init()
        if self.build_af:
            self.build_a()
        self.stamp(0)
        self.lane_constants()
        # panel 0 by wave 0
        L0 = "P0_done"
        p.s_cmp_lg_u32(sW, 0)
        p.s_cbranch_scc1(L0)
        self.setprio(3)
        if "ldsconst" in self.opts:
            self.write_constants()
        self.stamp(2)
        self.factor_panel(0, 0)
        self.setprio(0)
        p.label(L0)
        if "preaddr" in self.opts:
            L_p1 = "P1_addr"
            p.s_cmp_lg_u32(sW, 1)
            p.s_cbranch_scc1(L_p1)
            self.factor_addresses()
            p.label(L_p1)
        p.s_waitcnt("lgkmcnt(0)")
        self.stamp(2)
        p.s_barrier()
        self.stamp(1)
        for s in range(NSTEP):
            self.step(s)
        L_exit = "K3A_exit"
        p.s_branch(L_exit)
        for c in self.cold:
            c()
        p.label(L_exit)
        return p


def _emit_macro(f, name, extra_params, lines, extra_outs, s_hi):
    lo, _ = CLOBBER_S
    f.write(f"#define {name}(ACC, ARX, TW, P, W, TAU, LDSBASE{extra_params}) \\\n  asm volatile( \\\n")
    for ln in lines:
        f.write('    "' + ln.replace('"', '\\"') + '\\n" \\\n')
    outs = ", ".join(f'"={{v[{2 * k}:{2 * k + 1}]}}"(ACC[{k}])' for k in range(32))
    f.write(f"    : {outs}{extra_outs} \\\n")
    f.write('    : "s"(ARX), "s"(TW), "s"(P), "s"(W), "s"(TAU), "s"(LDSBASE) \\\n')
    clob = ", ".join(f'"v{k}"' for k in range(64, 96)) + ", " + ", ".join(f'"s{k}"' for k in range(lo, s_hi))
    f.write(f'    : {clob}, "vcc", "scc", "memory")\n')


def emit_inc(path, opts=None):
    g = Gen(for_text=True, opts=opts)
    prog = g.build()
    lines = prog.text_lines(label_prefix="K3A_%=_")
    lo, hi = CLOBBER_S
    with open(path, "w") as f:
        f.write("// GENERATED by csrc/gen/k3gen.py -- do not edit.  The 64-channel body of K3 (A(f) build + blocked\n"
                "// Gauss-Jordan inversion) as one asm statement with hand-allocated registers; see k3gen.py.\n")
        for name, val in (("PBUF", PBUF), ("NBUF", NBUF), ("SWAPB", SWAPB), ("RSUM", RSUM), ("SORIG", SORIG),
                          ("SSWP", SSWP), ("SINFO", SINFO), ("LDS_TOTAL", LDS_TOTAL)):
            f.write(f"#define K3A_{name} {val}\n")
        f.write(f"#define K3A_NUM_INSTRUCTIONS {sum(1 for l in lines if l.startswith('  '))}\n")
        _emit_macro(f, "K3A_BODY", "", lines, "", hi)
        # the diagnostic build's body: the same stream with s_memtime stamps; the five extra outputs are the phase sums
        # (A(f) build, factorisation, barrier wait, rest of the loop) and the last stamp, pinned to s[92:99], s[90:91]
        gs = Gen(for_text=True, opts=sorted(g.opts | {"stamps"}))
        for k in range(5):            # inputs move up by five operand numbers
            pass
        gs.in_arx, gs.in_tw = Reg("op", 37, 2), Reg("op", 38, 2)
        gs.in_p, gs.in_w, gs.in_tau, gs.in_lds = Reg("op", 39), Reg("op", 40), Reg("op", 41, 2), Reg("op", 42)
        slines = gs.build().text_lines(label_prefix="K3A_%=_")
        souts = "".join(f', "={{s[{92 + 2 * k}:{93 + 2 * k}]}}"(T{k})' for k in range(4)) + ', "={s90}"(TLAST)'
        _emit_macro(f, "K3A_BODY_STAMPED", ", T0, T1, T2, T3, TLAST", slines, souts, 90)
    return len(lines)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tf_inv64_body.inc"))
    ap.add_argument("--opts", default=None, help="comma-separated stream options (default: Gen.DEFAULT_OPTS); 'none' = plain")
    a = ap.parse_args()
    opts = None if a.opts is None else [o for o in a.opts.split(",") if o and o != "none"]
    n = emit_inc(os.path.normpath(a.out), opts)
    print(f"wrote {os.path.normpath(a.out)}: {n} lines, options {sorted(Gen(True, opts=opts).opts)}")
