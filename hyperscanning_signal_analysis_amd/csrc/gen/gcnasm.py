"""A small gfx950 assembly builder with a functional wave emulator and a hazard / wait-count checker.

Why this exists: the 64-channel body of K3 (csrc/gen/k3gen.py -> csrc/tf_inv64_body.inc) is written as ONE inline-asm
statement with hand-allocated registers (the compiler cannot fit the body into the 96 VGPRs that five workgroups per CU
allow: DESIGN.md section 5).  hipcc neither counts the memory operations nor pads the hazards inside an asm statement
(/opt/skills/guides/cdna_hip_programming.md section 5.7), and a GPU run costs minutes, so the instruction stream is
built as data, executed here on the CPU (four waves, LDS, barriers) against NumPy, and its executed trace is checked
against the wait-state table that hipcc itself applies (measured with compiler-generated micro-kernels:
profiles/r03_hazard_table.txt).  Only the subset of the ISA that the generator uses is modelled.

Nothing here runs on the GPU box; the product consumes the generated text only.
"""
from __future__ import annotations

import struct

import numpy as np

MASK64 = (1 << 64) - 1


# ----------------------------------------------------------------------------------------------- operands
class Reg:
    __slots__ = ("kind", "idx", "n")

    def __init__(self, kind, idx, n=1):
        self.kind, self.idx, self.n = kind, idx, n

    def __str__(self):
        if self.kind in ("vcc", "exec"):
            return self.kind
        if self.kind == "op":                      # %N operand of the asm statement (SGPR tuple chosen by the compiler)
            return f"%{self.idx}"
        return f"{self.kind}{self.idx}" if self.n == 1 else f"{self.kind}[{self.idx}:{self.idx + self.n - 1}]"

    def sub(self, off, n=1):
        assert off + n <= self.n
        return Reg(self.kind, self.idx + off, n)

    def regs(self):
        return [(self.kind, self.idx + k) for k in range(self.n)]


def V(i, n=1):
    return Reg("v", i, n)


def S(i, n=1):
    return Reg("s", i, n)


VCC = Reg("vcc", 0, 2)
EXEC = Reg("exec", 0, 2)


class Mod:
    """-x, |x|, -|x| source modifiers of f64 VALU operands."""
    __slots__ = ("r", "neg", "abs")

    def __init__(self, r, neg=False, abs_=False):
        self.r, self.neg, self.abs = r, neg, abs_

    def __str__(self):
        t = str(self.r) if not isinstance(self.r, float) else fmt_imm(self.r)
        if self.abs:
            t = f"|{t}|"
        return ("-" + t) if self.neg else t


def Neg(r):
    return Mod(r, neg=True)


def Abs(r):
    return Mod(r, abs_=True)


INLINE_F64 = {0.0: "0", 1.0: "1.0", -1.0: "-1.0", 0.5: "0.5", -0.5: "-0.5", 2.0: "2.0", -2.0: "-2.0", 4.0: "4.0", -4.0: "-4.0"}


def fmt_imm(x):
    if isinstance(x, float):
        return INLINE_F64[x]
    if isinstance(x, int):
        return str(x) if -16 <= x <= 64 else hex(x & 0xFFFFFFFF)
    raise TypeError(x)


def fmt(x):
    if isinstance(x, (Reg, Mod)):
        return str(x)
    if isinstance(x, str):
        return x
    return fmt_imm(x)


class Inst:
    __slots__ = ("op", "args", "mods", "comment")

    def __init__(self, op, args=(), mods="", comment=""):
        self.op, self.args, self.mods, self.comment = op, list(args), mods, comment

    def text(self):
        t = self.op
        if self.args:
            t += " " + ", ".join(fmt(a) for a in self.args)
        if self.mods:
            t += " " + self.mods
        return t


class Label:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name


class Program:
    def __init__(self):
        self.items = []          # Inst | Label
        self._uid = 0

    def emit(self, op, *args, mods="", comment=""):
        self.items.append(Inst(op, args, mods, comment))

    def label(self, name):
        self.items.append(Label(name))

    def newlabel(self, stem):
        self._uid += 1
        return f"{stem}_{self._uid}"

    def __getattr__(self, op):                      # p.v_fma_f64(dst, a, b, c)
        if op.startswith("_"):
            raise AttributeError(op)

        def f(*args, mods="", comment=""):
            self.emit(op, *args, mods=mods, comment=comment)
        return f

    def text_lines(self, label_prefix=""):
        out = []
        for it in self.items:
            if isinstance(it, Label):
                out.append(f"{label_prefix}{it.name}:")
            else:
                t = it.text()
                if label_prefix and it.op.startswith(("s_cbranch", "s_branch")):
                    t = f"{it.op} {label_prefix}{it.args[0]}"
                out.append("  " + t + (f"   ; {it.comment}" if it.comment else ""))
        return out


# ----------------------------------------------------------------------------------------------- emulator
def f64_of(lo, hi):
    return ((hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)).view(np.float64)


def split64(x):
    u = np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)
    return (u & np.uint64(0xFFFFFFFF)).astype(np.uint32), (u >> np.uint64(32)).astype(np.uint32)


class EmuError(Exception):
    pass


class Memory:
    """Flat fake global memory: a dict of (base address -> numpy uint8 buffer)."""

    def __init__(self):
        self.bufs = []

    def add(self, base, arr):
        self.bufs.append((base, np.ascontiguousarray(arr).view(np.uint8).reshape(-1)))

    def read(self, addr, n):
        for base, b in self.bufs:
            if base <= addr and addr + n <= base + len(b):
                return b[addr - base: addr - base + n]
        raise EmuError(f"global read out of bounds: {addr:#x}+{n}")


class Wave:
    NV, NS = 128, 128

    def __init__(self, prog_items, labels, lds, mem, wave_index, check_pending=True):
        self.items, self.labels, self.lds, self.mem = prog_items, labels, lds, mem
        self.v = np.zeros((self.NV, 64), dtype=np.uint32)
        self.s = np.zeros(self.NS, dtype=np.uint32)
        self.vcc, self.exec, self.scc = 0, MASK64, 0
        self.pc = 0
        self.wave_index = wave_index
        self.trace = []
        self.lgkm = []            # outstanding LDS / SMEM operations in issue order: (kind, frozenset(dst regs))
        self.vm = []
        self.check_pending = check_pending
        self.prio = 0
        self.done = False
        self.ninst = 0

    # ---- register access
    def lanes(self):
        return np.array([(self.exec >> l) & 1 for l in range(64)], dtype=bool)

    def _pending_check(self, regs, what):
        if not self.check_pending:
            return
        rs = set(regs)
        for q, name in ((self.lgkm, "lgkmcnt"), (self.vm, "vmcnt")):
            for kind, dst in q:
                if rs & dst:
                    raise EmuError(f"wave {self.wave_index} pc {self.pc} [{self.items[self.pc].text()}]: {what} "
                                   f"{sorted(rs & dst)} before the {kind} result was waited for ({name})")

    def rd32(self, r):
        """32-bit source -> array[64] uint32 (VGPR) or scalar broadcast."""
        if isinstance(r, int):
            return np.full(64, r & 0xFFFFFFFF, dtype=np.uint32)
        if isinstance(r, float):
            return np.full(64, struct.unpack("<I", struct.pack("<f", r))[0], dtype=np.uint32)
        if r.kind == "v":
            self._pending_check(r.regs(), "read of")
            return self.v[r.idx].copy()
        if r.kind == "s":
            self._pending_check(r.regs(), "read of")
            return np.full(64, self.s[r.idx], dtype=np.uint32)
        if r.kind == "vcc":
            return np.full(64, self.vcc & 0xFFFFFFFF, dtype=np.uint32)
        raise EmuError(f"rd32 {r}")

    def rds(self, r):
        """scalar 32-bit source"""
        if isinstance(r, int):
            return r & 0xFFFFFFFF
        if r.kind == "s":
            self._pending_check(r.regs(), "read of")
            return int(self.s[r.idx])
        if r.kind == "vcc":
            return self.vcc & 0xFFFFFFFF
        if r.kind == "exec":
            return self.exec & 0xFFFFFFFF
        raise EmuError(f"rds {r}")

    def rds64(self, r):
        if isinstance(r, int):
            return r & MASK64
        if r.kind == "s":
            self._pending_check(r.regs(), "read of")
            return int(self.s[r.idx]) | (int(self.s[r.idx + 1]) << 32)
        if r.kind == "vcc":
            return self.vcc
        if r.kind == "exec":
            return self.exec
        raise EmuError(f"rds64 {r}")

    def wrs(self, r, val):
        self._pending_check(r.regs() if r.kind == "s" else [], "write of")
        self.s[r.idx] = val & 0xFFFFFFFF

    def wrs64(self, r, val):
        val &= MASK64
        if r.kind == "s":
            self._pending_check(r.regs(), "write of")
            self.s[r.idx] = val & 0xFFFFFFFF
            self.s[r.idx + 1] = val >> 32
        elif r.kind == "vcc":
            self.vcc = val
        elif r.kind == "exec":
            self.exec = val
        else:
            raise EmuError(f"wrs64 {r}")

    def rdf64(self, src):
        """f64 VALU source (with modifiers) -> float64[64]"""
        neg = ab = False
        if isinstance(src, Mod):
            neg, ab, src = src.neg, src.abs, src.r
        if isinstance(src, float):
            x = np.full(64, src)
        elif isinstance(src, int):
            assert src == 0
            x = np.zeros(64)
        elif src.kind == "v":
            assert src.n == 2, src
            self._pending_check(src.regs(), "read of")
            x = f64_of(self.v[src.idx], self.v[src.idx + 1])
        elif src.kind == "s":
            assert src.n == 2
            self._pending_check(src.regs(), "read of")
            u = np.uint64(self.rds64(src))
            x = np.full(64, np.array([u], dtype=np.uint64).view(np.float64)[0])
        else:
            raise EmuError(f"rdf64 {src}")
        if ab:
            x = np.abs(x)
        if neg:
            x = -x
        return x

    def wrv32(self, r, val, mask=None):
        assert r.kind == "v" and r.n == 1
        self._pending_check(r.regs(), "write of")
        m = self.lanes() if mask is None else mask
        self.v[r.idx][m] = np.asarray(val, dtype=np.uint32)[m] if np.ndim(val) else np.uint32(val)

    def wrf64(self, r, val):
        assert r.kind == "v" and r.n == 2
        self._pending_check(r.regs(), "write of")
        lo, hi = split64(val)
        m = self.lanes()
        self.v[r.idx][m] = lo[m]
        self.v[r.idx + 1][m] = hi[m]

    # ---- execution
    def run(self):
        """generator: yields 'barrier' at s_barrier, returns at s_endpgm / end of program"""
        items = self.items
        while self.pc < len(items):
            it = items[self.pc]
            if isinstance(it, Label):
                self.pc += 1
                continue
            self.trace.append(self.pc)
            self.ninst += 1
            if self.ninst > 2_000_000:
                raise EmuError("runaway wave")
            r = self.step(it)
            if r == "barrier":
                self.pc += 1
                yield "barrier"
                continue
            if r == "end":
                break
            if isinstance(r, str):           # branch target
                self.pc = self.labels[r]
            else:
                self.pc += 1
        self.done = True

    def step(self, it):
        op, a = it.op, it.args
        h = getattr(self, "op_" + op, None)
        if h is None:
            raise EmuError(f"emulator: unknown instruction {it.text()}")
        return h(it, *a)

    # ---- SALU
    def op_s_mov_b32(self, it, d, x):
        self.wrs(d, self.rds(x))

    def op_s_mov_b64(self, it, d, x):
        if isinstance(x, int):
            x = x & MASK64 if x >= 0 else (x + (1 << 64))
            self.wrs64(d, x)
        else:
            self.wrs64(d, self.rds64(x))

    def op_s_lshl_b64(self, it, d, x, n):
        v = self.rds64(x) if not isinstance(x, int) else (x & MASK64)
        r = (v << (self.rds(n) & 63)) & MASK64
        self.wrs64(d, r)
        self.scc = int(r != 0)

    def op_s_and_b64(self, it, d, x, y):
        r = self.rds64(x) & self.rds64(y)
        self.wrs64(d, r)
        self.scc = int(r != 0)

    def op_s_or_b64(self, it, d, x, y):
        r = self.rds64(x) | self.rds64(y)
        self.wrs64(d, r)
        self.scc = int(r != 0)

    def op_s_andn2_b64(self, it, d, x, y):
        r = self.rds64(x) & ~self.rds64(y) & MASK64
        self.wrs64(d, r)
        self.scc = int(r != 0)

    def op_s_and_b32(self, it, d, x, y):
        r = self.rds(x) & self.rds(y)
        self.wrs(d, r)
        self.scc = int(r != 0)

    def op_s_or_b32(self, it, d, x, y):
        r = self.rds(x) | self.rds(y)
        self.wrs(d, r)
        self.scc = int(r != 0)

    def op_s_lshr_b32(self, it, d, x, n):
        r = self.rds(x) >> (self.rds(n) & 31)
        self.wrs(d, r)
        self.scc = int(r != 0)

    def op_s_lshl_b32(self, it, d, x, n):
        r = (self.rds(x) << (self.rds(n) & 31)) & 0xFFFFFFFF
        self.wrs(d, r)
        self.scc = int(r != 0)

    def op_s_bfe_u32(self, it, d, x, spec):
        off, width = spec & 31, (spec >> 16) & 0x7F
        r = (self.rds(x) >> off) & ((1 << width) - 1)
        self.wrs(d, r)
        self.scc = int(r != 0)

    def op_s_add_u32(self, it, d, x, y):
        r = self.rds(x) + self.rds(y)
        self.wrs(d, r)
        self.scc = int(r > 0xFFFFFFFF)

    def op_s_addc_u32(self, it, d, x, y):
        r = self.rds(x) + self.rds(y) + self.scc
        self.wrs(d, r)
        self.scc = int(r > 0xFFFFFFFF)

    def op_s_add_i32(self, it, d, x, y):
        self.wrs(d, self.rds(x) + self.rds(y))

    def op_s_sub_i32(self, it, d, x, y):
        self.wrs(d, self.rds(x) - self.rds(y))

    def op_s_mul_i32(self, it, d, x, y):
        self.wrs(d, self.rds(x) * self.rds(y))

    def op_s_ff1_i32_b64(self, it, d, x):
        v = self.rds64(x)
        self.wrs(d, (v & -v).bit_length() - 1 if v else 0xFFFFFFFF)

    def _cmp(self, x, y, f):
        self.scc = int(f(self.rds(x), self.rds(y)))

    def op_s_cmp_eq_u32(self, it, x, y):
        self._cmp(x, y, lambda p, q: p == q)

    def op_s_cmp_lg_u32(self, it, x, y):
        self._cmp(x, y, lambda p, q: p != q)

    def op_s_cmp_lt_u32(self, it, x, y):
        self._cmp(x, y, lambda p, q: p < q)

    def op_s_cmp_ge_u32(self, it, x, y):
        self._cmp(x, y, lambda p, q: p >= q)

    def op_s_cmp_lg_u64(self, it, x, y):
        self.scc = int(self.rds64(x) != self.rds64(y))

    def op_s_cmp_eq_u64(self, it, x, y):
        self.scc = int(self.rds64(x) == self.rds64(y))

    def op_s_cselect_b32(self, it, d, x, y):
        self.wrs(d, self.rds(x) if self.scc else self.rds(y))

    def op_s_cbranch_scc1(self, it, lbl):
        return lbl if self.scc else None

    def op_s_cbranch_scc0(self, it, lbl):
        return lbl if not self.scc else None

    def op_s_cbranch_vccnz(self, it, lbl):
        return lbl if self.vcc != 0 else None

    def op_s_cbranch_vccz(self, it, lbl):
        return lbl if self.vcc == 0 else None

    def op_s_branch(self, it, lbl):
        return lbl

    def op_s_nop(self, it, n):
        pass

    def op_s_setprio(self, it, n):
        self.prio = n

    def op_s_barrier(self, it):
        if self.check_pending and any(k == "lds" for k, _ in self.lgkm):
            raise EmuError(f"wave {self.wave_index}: s_barrier with LDS operations outstanding (no lgkmcnt(0) before it)")
        return "barrier"

    def op_s_endpgm(self, it):
        return "end"

    def op_s_waitcnt(self, it, spec):
        for part in spec.split():
            name, n = part[:-1].split("(")
            n = int(n)
            if name == "lgkmcnt":
                if any(k == "smem" for k, _ in self.lgkm) and n != 0:
                    raise EmuError(f"wave {self.wave_index} pc {self.pc}: lgkmcnt({n}) with scalar loads outstanding (out of order)")
                while len(self.lgkm) > n:
                    self.lgkm.pop(0)
            elif name == "vmcnt":
                while len(self.vm) > n:
                    self.vm.pop(0)
            else:
                raise EmuError(spec)

    def op_s_load_dwordx4(self, it, d, base, off):
        self._sload(d, base, off, 4)

    def op_s_load_dwordx8(self, it, d, base, off):
        self._sload(d, base, off, 8)

    def op_s_load_dwordx2(self, it, d, base, off):
        self._sload(d, base, off, 2)

    def _sload(self, d, base, off, n):
        assert d.n == n
        addr = self.rds64(base) + (off if isinstance(off, int) else self.rds(off))
        data = self.mem.read(addr, 4 * n).view(np.uint32)
        self._pending_check(d.regs(), "write of")
        for k in range(n):
            self.s[d.idx + k] = data[k]
        self.lgkm.append(("smem", frozenset(d.regs())))

    # ---- VALU integer / moves
    def op_v_mbcnt_lo_u32_b32(self, it, d, m, x):
        assert m == -1
        base = self.rd32(x)
        cnt = np.array([min(l, 32) for l in range(64)], dtype=np.uint32)
        self.wrv32(d, base + cnt)

    def op_v_mbcnt_hi_u32_b32(self, it, d, m, x):
        assert m == -1
        base = self.rd32(x)
        cnt = np.array([max(l - 32, 0) for l in range(64)], dtype=np.uint32)
        self.wrv32(d, base + cnt)

    def op_v_mov_b32_e32(self, it, d, x):
        self.wrv32(d, self.rd32(x))

    def op_v_mov_b64_e32(self, it, d, x):
        if isinstance(x, int):
            assert x == 0
            self.wrf64(d, np.zeros(64))
        else:
            self.wrf64(d, self.rdf64(x))

    def op_v_and_b32_e32(self, it, d, x, y):
        self.wrv32(d, self.rd32(x) & self.rd32(y))

    def op_v_or_b32_e32(self, it, d, x, y):
        self.wrv32(d, self.rd32(x) | self.rd32(y))

    def op_v_lshrrev_b32_e32(self, it, d, n, x):
        self.wrv32(d, self.rd32(x) >> (self.rd32(n) & 31))

    def op_v_lshlrev_b32_e32(self, it, d, n, x):
        self.wrv32(d, self.rd32(x) << (self.rd32(n) & 31))

    def op_v_add_u32_e32(self, it, d, x, y):
        self.wrv32(d, self.rd32(x) + self.rd32(y))

    def op_v_sub_u32_e32(self, it, d, x, y):
        self.wrv32(d, self.rd32(x) - self.rd32(y))

    def op_v_lshl_add_u32(self, it, d, x, n, y):
        self.wrv32(d, (self.rd32(x) << (self.rd32(n) & 31)) + self.rd32(y))

    def op_v_mul_u32_u24_e32(self, it, d, x, y):
        self.wrv32(d, (self.rd32(x) & 0xFFFFFF).astype(np.uint64) * (self.rd32(y) & 0xFFFFFF) & 0xFFFFFFFF)

    def op_v_mad_u32_u24(self, it, d, x, y, z):
        self.wrv32(d, ((self.rd32(x) & 0xFFFFFF).astype(np.uint64) * (self.rd32(y) & 0xFFFFFF) + self.rd32(z)) & 0xFFFFFFFF)

    def op_v_bfe_u32(self, it, d, x, off, width):
        self.wrv32(d, (self.rd32(x) >> (self.rd32(off) & 31)) & ((1 << int(self.rd32(width)[0])) - 1))

    def _vcmp_write(self, d, res):
        bits = 0
        ex = self.lanes()
        for l in range(64):
            if ex[l] and res[l]:
                bits |= 1 << l
        self.wrs64(d, bits)

    def op_v_cmp_eq_u32_e32(self, it, d, x, y):
        assert d.kind == "vcc"
        self._vcmp_write(d, self.rd32(x) == self.rd32(y))

    def op_v_cmp_eq_u32_e64(self, it, d, x, y):
        self._vcmp_write(d, self.rd32(x) == self.rd32(y))

    def op_v_cmp_ge_u32_e64(self, it, d, x, y):
        self._vcmp_write(d, self.rd32(x) >= self.rd32(y))

    def op_v_cmp_gt_f64_e32(self, it, d, x, y):
        assert d.kind == "vcc"
        self._vcmp_write(d, self.rdf64(x) > self.rdf64(y))

    def op_v_cmp_lt_f64_e32(self, it, d, x, y):
        assert d.kind == "vcc"
        self._vcmp_write(d, self.rdf64(x) < self.rdf64(y))

    def op_v_cmp_gt_f64_e64(self, it, d, x, y):
        self._vcmp_write(d, self.rdf64(x) > self.rdf64(y))

    def op_v_cmp_lt_f64_e64(self, it, d, x, y):
        self._vcmp_write(d, self.rdf64(x) < self.rdf64(y))

    def op_v_cndmask_b32_e32(self, it, d, x, y, m):
        assert m.kind == "vcc"
        self._cnd(d, x, y, self.vcc)

    def op_v_cndmask_b32_e64(self, it, d, x, y, m):
        self._cnd(d, x, y, self.rds64(m))

    def _cnd(self, d, x, y, mask):
        sel = np.array([(mask >> l) & 1 for l in range(64)], dtype=bool)
        self.wrv32(d, np.where(sel, self.rd32(y), self.rd32(x)))

    def op_v_readlane_b32(self, it, d, x, lane):
        ln = (lane if isinstance(lane, int) else self.rds(lane)) & 63
        self._pending_check(x.regs(), "read of")
        self.wrs(d, int(self.v[x.idx][ln]))

    def op_v_readfirstlane_b32(self, it, d, x):
        ln = (self.exec & -self.exec).bit_length() - 1 if self.exec else 0
        self.wrs(d, int(self.v[x.idx][ln]))

    def op_v_cvt_f32_f64_e32(self, it, d, x):
        f = self.rdf64(x).astype(np.float32)
        self.wrv32(d, f.view(np.uint32))

    def op_v_max_u32_dpp(self, it, d, x, y):
        src = self.rd32(x)
        other = self.rd32(y)
        mods = it.mods
        row_mask = int(mods.split("row_mask:")[1].split()[0], 16)
        idx = np.arange(64)
        valid = np.ones(64, dtype=bool)
        if "quad_perm" in mods:
            perm = [int(t) for t in mods.split("quad_perm:[")[1].split("]")[0].split(",")]
            sl = (idx & ~3) | np.array([perm[l & 3] for l in range(64)])
        elif "row_half_mirror" in mods:
            sl = (idx & ~7) | (7 - (idx & 7))
        elif "row_mirror" in mods:
            sl = (idx & ~15) | (15 - (idx & 15))
        elif "row_bcast:15" in mods:
            sl = ((idx >> 4) - 1) * 16 + 15
            valid = (idx >> 4) >= 1
        elif "row_bcast:31" in mods:
            sl = np.full(64, 31)
            valid = idx >= 32
        else:
            raise EmuError(mods)
        rowen = np.array([(row_mask >> (l >> 4)) & 1 for l in range(64)], dtype=bool)
        m = self.lanes() & rowen & valid
        res = np.maximum(src[np.clip(sl, 0, 63)], other)
        self.wrv32(d, res, mask=m)

    # ---- VALU f64
    def op_v_add_f64(self, it, d, x, y):
        self.wrf64(d, self.rdf64(x) + self.rdf64(y))

    def op_v_mul_f64(self, it, d, x, y):
        self.wrf64(d, self.rdf64(x) * self.rdf64(y))

    def op_v_fma_f64(self, it, d, x, y, z):
        self.wrf64(d, self.rdf64(x) * self.rdf64(y) + self.rdf64(z))

    def op_v_rcp_f64_e32(self, it, d, x):
        with np.errstate(divide="ignore", invalid="ignore"):
            self.wrf64(d, 1.0 / self.rdf64(x))

    # ---- MFMA.  Lane maps measured on gfx950 (csrc/hmv_common.h): A[b][i][k] on lane 16k+4b+i, B[b][k][j] on lane
    # 16k+4b+j, C/D[b][i][j] on lane 16i+4b+j.
    def op_v_mfma_f64_4x4x4_4b_f64(self, it, d, a, b, c):
        A, B, C = self.rdf64(a), self.rdf64(b), self.rdf64(c)
        if "neg:[1,0,0]" in it.mods:
            A = -A
        out = np.empty(64)
        for blk in range(4):
            Am = np.array([[A[16 * k + 4 * blk + i] for k in range(4)] for i in range(4)])
            Bm = np.array([[B[16 * k + 4 * blk + j] for j in range(4)] for k in range(4)])
            Cm = np.array([[C[16 * i + 4 * blk + j] for j in range(4)] for i in range(4)])
            Dm = Cm + Am @ Bm
            for i in range(4):
                for j in range(4):
                    out[16 * i + 4 * blk + j] = Dm[i, j]
        assert self.exec == MASK64, "MFMA under a partial exec mask"
        self.wrf64(d, out)

    # ---- LDS
    def _lds_addr(self, areg, it):
        off = 0
        if "offset:" in it.mods:
            off = int(it.mods.split("offset:")[1].split()[0], 0)
        return self.rd32(areg).astype(np.int64) + off

    def _ds_read(self, it, d, areg, nbytes):
        addr = self._lds_addr(areg, it)
        ex = self.lanes()
        self._pending_check(d.regs(), "write of")
        for l in range(64):
            if ex[l]:
                a = int(addr[l])
                if a < 0 or a + nbytes > len(self.lds) or a % min(nbytes, 8):
                    raise EmuError(f"wave {self.wave_index} pc {self.pc} LDS read at {a} ({it.text()})")
                w = self.lds[a:a + nbytes].view(np.uint32)
                for k in range(nbytes // 4):
                    self.v[d.idx + k][l] = w[k]
        self.lgkm.append(("lds", frozenset(d.regs())))

    def _ds_write(self, it, areg, src, nbytes):
        addr = self._lds_addr(areg, it)
        ex = self.lanes()
        self._pending_check(src.regs(), "read of")
        for l in range(64):
            if ex[l]:
                a = int(addr[l])
                if a < 0 or a + nbytes > len(self.lds) or a % min(nbytes, 8):
                    raise EmuError(f"wave {self.wave_index} pc {self.pc} LDS write at {a} ({it.text()})")
                w = np.array([self.v[src.idx + k][l] for k in range(nbytes // 4)], dtype=np.uint32)
                self.lds[a:a + nbytes] = w.view(np.uint8)
        self.lgkm.append(("lds", frozenset()))

    def op_ds_read_b128(self, it, d, a):
        assert d.n == 4
        self._ds_read(it, d, a, 16)

    def op_ds_read_b64(self, it, d, a):
        assert d.n == 2
        self._ds_read(it, d, a, 8)

    def op_ds_read_b32(self, it, d, a):
        self._ds_read(it, d, a, 4)

    def op_ds_write_b128(self, it, a, s):
        assert s.n == 4
        self._ds_write(it, a, s, 16)

    def op_ds_write_b64(self, it, a, s):
        assert s.n == 2
        self._ds_write(it, a, s, 8)

    def op_ds_write_b32(self, it, a, s):
        self._ds_write(it, a, s, 4)

    def op_ds_swizzle_b32(self, it, d, x):
        pat = int(it.mods.split("offset:")[1].split()[0], 0)
        assert pat < 0x8000, "bitmask mode only"
        andm, orm, xorm = pat & 31, (pat >> 5) & 31, (pat >> 10) & 31
        src = self.rd32(x)
        ex = self.lanes()
        out = np.zeros(64, dtype=np.uint32)
        for l in range(64):
            j = l & 31
            sl = (l & 32) | ((((j & andm) | orm) ^ xorm) & 31)
            out[l] = src[sl] if ex[sl] else 0
        self._pending_check(d.regs(), "write of")
        m = ex
        self.v[d.idx][m] = out[m]
        self.lgkm.append(("lds", frozenset(d.regs())))

    # ---- global
    def op_global_load_dwordx4(self, it, d, voff, sbase):
        assert d.n == 4
        off = 0
        if "offset:" in it.mods:
            off = int(it.mods.split("offset:")[1].split()[0], 0)
        base = self.rds64(sbase)
        vo = self.rd32(voff)
        ex = self.lanes()
        self._pending_check(d.regs(), "write of")
        for l in range(64):
            if ex[l]:
                w = self.mem.read(base + int(vo[l]) + off, 16).view(np.uint32)
                for k in range(4):
                    self.v[d.idx + k][l] = w[k]
        self.vm.append(("vmem", frozenset(d.regs())))


def run_workgroup(prog: Program, n_waves, lds_bytes, mem, init, check_pending=True):
    """Execute the program on n_waves waves sharing an LDS; init(wave) sets up registers.  Returns the waves."""
    items = prog.items
    labels = {it.name: k for k, it in enumerate(items) if isinstance(it, Label)}
    lds = np.zeros(lds_bytes, dtype=np.uint8)
    waves = [Wave(items, labels, lds, mem, w, check_pending) for w in range(n_waves)]
    for w in waves:
        init(w)
    gens = [w.run() for w in waves]
    alive = [True] * n_waves
    while any(alive):
        at_barrier = 0
        for k, g in enumerate(gens):
            if not alive[k]:
                continue
            try:
                next(g)
                at_barrier += 1
            except StopIteration:
                alive[k] = False
        if at_barrier and at_barrier != sum(alive):
            raise EmuError("some waves ended while others wait at a barrier")
    return waves


# ----------------------------------------------------------------------------------------------- hazard checker
# Wait states hipcc applies on gfx950 (measured: profiles/r03_hazard_table.txt).  "Wait states" = issued instructions in
# between; s_nop N counts N + 1; s_waitcnt / s_barrier / s_setprio are counted as zero here (conservative).
WS_MFMA_SRCC, WS_MFMA_SRCAB, WS_MFMA_VALU, WS_MFMA_MEM = 4, 6, 6, 9
WS_VALU_MFMA, WS_VALU_READLANE, WS_SGPR_VALU, WS_SGPR_LANESEL, WS_SGPR_VMEM, WS_TRANS_VALU, WS_VALU_DPP = 2, 1, 2, 4, 5, 1, 2
WS_EXEC_DPP = 5

SALU_PREFIX = ("s_",)
ZERO_WS = ("s_waitcnt", "s_barrier", "s_setprio")


def classify(it: Inst):
    op = it.op
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_"):
        return "vmem"
    if op.startswith("s_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_rcp") or op.startswith("v_rsq") or op.startswith("v_sqrt"):
        return "trans"
    return "valu"


def operands_rw(it: Inst):
    """(vgpr writes, vgpr reads, sgpr writes, sgpr reads); vcc / exec appear as ('vcc', 0), ('exec', 0) in the sgpr sets."""
    op, a = it.op, it.args
    vw, vr, sw, sr = set(), set(), set(), set()

    def add(dst, r):
        if isinstance(r, Mod):
            r = r.r
        if isinstance(r, Reg):
            if r.kind == "v":
                dst[0].update(r.regs())
            elif r.kind in ("s",):
                dst[1].update(r.regs())
            elif r.kind in ("vcc", "exec"):
                dst[1].add((r.kind, 0))

    W, R = (vw, sw), (vr, sr)
    cls = classify(it)
    if op.startswith(("s_cbranch_vcc",)):
        sr.add(("vcc", 0))
    elif op.startswith(("s_cbranch", "s_branch", "s_nop", "s_waitcnt", "s_barrier", "s_setprio", "s_endpgm")):
        pass
    elif op.startswith("s_cmp"):
        for x in a:
            add(R, x)
    elif op.startswith("ds_write"):
        add(R, a[0]); add(R, a[1])
    elif op.startswith("ds_read") or op.startswith("ds_swizzle"):
        add(W, a[0]); add(R, a[1])
    elif op.startswith("global_load"):
        add(W, a[0]); add(R, a[1]); add(R, a[2])
    elif op.startswith("s_load"):
        add(W, a[0]); add(R, a[1])
    else:
        add(W, a[0])
        for x in a[1:]:
            add(R, x)
        if op.startswith("v_cndmask_b32_e32"):
            sr.add(("vcc", 0))
        if op.endswith("_dpp") or op in ("v_fma_f64",) and False:
            pass
    if cls in ("valu", "trans", "mfma", "lds", "vmem") and not op.startswith("v_readfirstlane"):
        sr.add(("exec", 0))
    return vw, vr, sw, sr


def check_hazards(prog: Program, trace, what=""):
    """Walk an executed trace and report every place where the stream gives fewer wait states than hipcc would."""
    items = prog.items
    last_mfma_w, last_valu_w, last_trans_w = {}, {}, {}
    last_valu_sw = {}            # SGPR (incl. vcc) written by a VALU instruction
    last_valu_exec = -10 ** 9
    last_store_data = {}
    t = 0                        # wait-state clock
    problems = []

    def need(pos, kind, have, want, it, reg):
        if have < want:
            problems.append(f"{what} trace#{pos} [{it.text()}]: {kind} on {reg}: {have} wait states, {want} needed")

    for pos, pc in enumerate(trace):
        it = items[pc]
        cls = classify(it)
        vw, vr, sw, sr = operands_rw(it)
        if it.op == "s_nop":
            t += int(it.args[0]) + 1
            continue
        if it.op in ZERO_WS:
            continue
        # ---- consumer checks at time t (number of wait states since producer = t - t_prod - 1)
        def since(tp):
            return t - tp - 1
        if cls == "mfma":
            d, a_, b_, c_ = it.args
            for r in c_.regs():
                if r in last_mfma_w:
                    need(pos, "MFMA -> MFMA srcC", since(last_mfma_w[r]), WS_MFMA_SRCC, it, r)
            for src in (a_, b_):
                for r in src.regs():
                    if r in last_mfma_w:
                        need(pos, "MFMA -> MFMA srcA/B", since(last_mfma_w[r]), WS_MFMA_SRCAB, it, r)
            for r in set(a_.regs()) | set(b_.regs()) | set(c_.regs()):
                if r in last_valu_w:
                    need(pos, "VALU -> MFMA", since(last_valu_w[r]), WS_VALU_MFMA, it, r)
        elif cls in ("valu", "trans"):
            for r in vr | vw:
                if r in last_mfma_w:
                    need(pos, "MFMA -> VALU", since(last_mfma_w[r]), WS_MFMA_VALU, it, r)
            for r in vr:
                if r in last_trans_w:
                    need(pos, "trans -> VALU", since(last_trans_w[r]), WS_TRANS_VALU, it, r)
            if it.op.startswith(("v_readlane", "v_readfirstlane")):
                for r in it.args[1].regs():
                    if r in last_valu_w:
                        need(pos, "VALU -> v_readlane", since(last_valu_w[r]), WS_VALU_READLANE, it, r)
                if it.op.startswith("v_readlane") and isinstance(it.args[2], Reg):
                    for r in it.args[2].regs():
                        if r in last_valu_sw:
                            need(pos, "VALU-written SGPR -> lane select", since(last_valu_sw[r]), WS_SGPR_LANESEL, it, r)
            else:
                for r in sr:
                    if r in last_valu_sw and r != ("exec", 0):
                        need(pos, "VALU-written SGPR -> VALU read", since(last_valu_sw[r]), WS_SGPR_VALU, it, r)
            if it.op.endswith("_dpp"):
                for r in vr:
                    if r in last_valu_w:
                        need(pos, "VALU -> DPP", since(last_valu_w[r]), WS_VALU_DPP, it, r)
                need(pos, "exec write -> DPP", since(last_valu_exec), WS_EXEC_DPP, it, "exec")
            for r in vw:
                if r in last_store_data:
                    need(pos, "wide LDS/VMEM store data -> VALU overwrite", since(last_store_data[r]), 2, it, r)
        elif cls in ("lds", "vmem"):
            for r in vr | vw:
                if r in last_mfma_w:
                    need(pos, "MFMA -> LDS/VMEM", since(last_mfma_w[r]), WS_MFMA_MEM, it, r)
            if cls == "vmem":
                for r in sr:
                    if r in last_valu_sw:
                        need(pos, "VALU-written SGPR -> VMEM", since(last_valu_sw[r]), WS_SGPR_VMEM, it, r)
        # ---- producer bookkeeping
        if cls == "mfma":
            for r in vw:
                last_mfma_w[r] = t
                last_valu_w.pop(r, None)
        elif cls in ("valu", "trans"):
            for r in vw:
                last_valu_w[r] = t
                last_mfma_w.pop(r, None)
                last_trans_w.pop(r, None)
                if cls == "trans":
                    last_trans_w[r] = t
            for r in sw:
                last_valu_sw[r] = t
        elif cls in ("lds", "vmem", "smem"):
            for r in vw:
                last_valu_w.pop(r, None)
                last_mfma_w.pop(r, None)
                last_trans_w.pop(r, None)
            for r in sw:
                last_valu_sw.pop(r, None)
            if it.op in ("ds_write_b128",):
                for r in it.args[1].regs():
                    last_store_data[r] = t
        elif cls == "salu":
            for r in sw:
                if r == ("exec", 0):
                    last_valu_exec = t          # treated like a VALU write of exec for the DPP rule (conservative)
                last_valu_sw.pop(r, None) if r != ("exec", 0) else None
        t += 1
    return problems
