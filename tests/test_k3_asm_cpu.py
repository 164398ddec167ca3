"""The hand-scheduled 64-channel body of K3 (csrc/gen/k3gen.py -> csrc/tf_inv64_body.inc) executed on the CPU emulator
(csrc/gen/gcnasm.py): four waves, LDS and barriers, against NumPy -- A(f) from packed coefficients, the blocked
Gauss-Jordan inverse with and without row interchanges, wait counts (no register is used before its load was waited
for) and the wait states hipcc would have inserted (checked on the executed trace of every wave).  CPU only; the same
stream is compared bit for bit with the compiler-scheduled body on the GPU (tests/test_gpu_parity.py)."""
import os
import sys

import numpy as np
import pytest

GEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hyperscanning_signal_analysis_amd", "csrc", "gen")
sys.path.insert(0, os.path.normpath(GEN))
import gcnasm as G  # noqa: E402
import k3gen as K  # noqa: E402

ARX_BASE, TW_BASE = 0x10000000, 0x20000000


def lane_coords():
    l = np.arange(64)
    return l >> 4, (l >> 2) & 3, l & 3           # i, b, j


def scatter_matrix(M, w):
    """complex 64 x 64 -> the 64 accumulator VGPRs of wave w (X layout)"""
    i, b, j = lane_coords()
    regs = np.zeros((64, 64), dtype=np.uint32)
    for Ig in range(4):
        for Jl in range(4):
            rows, cols = 16 * Ig + 4 * b + i, 16 * Jl + 4 * w + j
            v = M[rows, cols]
            k = 4 * (4 * Ig + Jl)
            regs[k], regs[k + 1] = G.split64(v.real)
            regs[k + 2], regs[k + 3] = G.split64(v.imag)
    return regs


def gather_matrix(waves, ws):
    i, b, j = lane_coords()
    M = np.zeros((64, 64), dtype=np.complex128)
    for wave, w in zip(waves, ws):
        for Ig in range(4):
            for Jl in range(4):
                k = 4 * (4 * Ig + Jl)
                re = G.f64_of(wave.v[k], wave.v[k + 1])
                im = G.f64_of(wave.v[k + 2], wave.v[k + 3])
                M[16 * Ig + 4 * b + i, 16 * Jl + 4 * w + j] = re + 1j * im
    return M


def pack_arx(ar, w):
    """ar (64, 64, p) -> the packed layout of ar_pack_kernel for wave w: [B = 4 Ig + Jl][h][lane][2]"""
    p = ar.shape[2]
    P2 = (p + 1) // 2
    i, b, j = lane_coords()
    out = np.zeros((16, P2, 64, 2))
    for Ig in range(4):
        for Jl in range(4):
            rows, cols = 16 * Ig + 4 * b + i, 16 * Jl + 4 * w + j
            for h in range(P2):
                out[4 * Ig + Jl, h, :, 0] = ar[rows, cols, 2 * h]
                if 2 * h + 1 < p:
                    out[4 * Ig + Jl, h, :, 1] = ar[rows, cols, 2 * h + 1]
    return out


def run(prog, init_regs, rot, p_order=8, tau=1.0, mem=None):
    """rot: hardware wave k plays role w = rot[k].  Returns (waves, lds)."""
    lds_holder = {}

    def init(wave):
        w = rot[wave.wave_index]
        wave.s[2], wave.s[3] = (ARX_BASE + w * (1 << 24)) & 0xFFFFFFFF, (ARX_BASE + w * (1 << 24)) >> 32
        wave.s[4], wave.s[5] = TW_BASE & 0xFFFFFFFF, TW_BASE >> 32
        wave.s[6], wave.s[7] = p_order, w
        lo, hi = G.split64(np.array([tau]))
        wave.s[8], wave.s[9] = lo[0], hi[0]
        wave.s[10] = 0
        if init_regs is not None:
            wave.v[:64] = init_regs[w]
        # what the C++ prologue leaves in LDS: orig[l] = l, info = 0
        wave.lds[K.SORIG:K.SORIG + 256] = np.arange(64, dtype=np.uint32).view(np.uint8)
        wave.lds[K.SINFO:K.SINFO + 4] = 0
        lds_holder["lds"] = wave.lds
    waves = G.run_workgroup(prog, 4, K.LDS_TOTAL, mem or G.Memory(), init)
    return waves, lds_holder["lds"]


def check_inverse(waves, lds, rot, A, tol=1e-9):
    got = gather_matrix(waves, rot)
    orig = lds[K.SORIG:K.SORIG + 256].view(np.uint32).astype(int)
    assert sorted(orig) == list(range(64))
    H = np.zeros_like(got)
    H[:, orig] = got                              # output column of stored column c is orig[c]
    want = np.linalg.inv(A)
    err = np.abs(H - want).max() / np.abs(want).max()
    assert err < tol, err
    assert int(lds[K.SINFO:K.SINFO + 4].view(np.uint32)[0]) == 0
    return orig


VARIANTS = {"plain": (), "default": None, "ldsconst": ("earlyswz", "hoist", "ldsconst"),
            "spec": ("earlyswz", "hoist", "rlpiv", "spec", "preaddr")}


@pytest.fixture(scope="module", params=sorted(VARIANTS))
def inv_prog(request):
    return K.Gen(for_text=False, build_af=False, opts=VARIANTS[request.param]).build()


@pytest.fixture(scope="module", params=sorted(VARIANTS))
def full_prog(request):
    return K.Gen(for_text=False, build_af=True, opts=VARIANTS[request.param]).build()


def hazards(prog, waves, what):
    probs = []
    for wv in waves:
        probs += G.check_hazards(prog, wv.trace, f"{what} wave {wv.wave_index}")
    assert not probs, "\n".join(probs[:40]) + f"\n... {len(probs)} in all"


def test_inverse_without_interchanges(inv_prog):
    rng = np.random.default_rng(1)
    A = np.eye(64) * 4 + 0.3 * (rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64)))
    rot = [2, 3, 0, 1]
    waves, lds = run(inv_prog, {w: scatter_matrix(A, w) for w in range(4)}, rot)
    orig = check_inverse(waves, lds, rot, A)
    assert list(orig) == list(range(64))
    hazards(inv_prog, waves, "diagonally dominant")


@pytest.mark.parametrize("seed", [2, 3])
def test_inverse_with_row_interchanges(inv_prog, seed):
    """a general complex matrix: nearly every pivot column interchanges rows (search path, displaced rows, orig[],
    the other waves' row swaps through LDS)"""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))
    rot = [1, 0, 3, 2]
    waves, lds = run(inv_prog, {w: scatter_matrix(A, w) for w in range(4)}, rot)
    orig = check_inverse(waves, lds, rot, A, 1e-8)
    assert list(orig) != list(range(64))
    hazards(inv_prog, waves, "general matrix")


def test_scaled_permutation_and_threshold(inv_prog):
    rng = np.random.default_rng(5)
    perm = rng.permutation(64)
    A = np.zeros((64, 64), dtype=np.complex128)
    A[np.arange(64), perm] = rng.uniform(0.5, 2.0, 64) * np.exp(2j * np.pi * rng.uniform(size=64))
    rot = [0, 1, 2, 3]
    waves, lds = run(inv_prog, {w: scatter_matrix(A, w) for w in range(4)}, rot)
    check_inverse(waves, lds, rot, A, 1e-12)
    # tau < 1 keeps a diagonal that is within a factor tau of the column maximum
    B = np.eye(64) + 0.6 * rng.standard_normal((64, 64))
    w1, l1 = run(inv_prog, {w: scatter_matrix(B.astype(complex), w) for w in range(4)}, rot, tau=0.01)
    check_inverse(w1, l1, rot, B.astype(complex), 1e-8)


def test_singular_matrix_sets_info(inv_prog):
    A = np.eye(64, dtype=np.complex128)
    A[10, 10] = 0.0
    rot = [0, 1, 2, 3]
    waves, lds = run(inv_prog, {w: scatter_matrix(A, w) for w in range(4)}, rot)
    assert int(lds[K.SINFO:K.SINFO + 4].view(np.uint32)[0]) == 11


@pytest.mark.parametrize("p", [8, 5, 3, 16, 1])
def test_transfer_matrix_build_and_inverse(full_prog, p):
    rng = np.random.default_rng(10 + p)
    ar = 0.08 * rng.standard_normal((64, 64, p))
    ar[np.arange(64), np.arange(64), 0] += 0.5
    fs, f = 500.0, 17.5
    z = np.exp(-(np.arange(p) + 1) * 2 * np.pi * 1j * f / fs)
    A = np.eye(64) - (ar * z).sum(axis=2)
    mem = G.Memory()
    for w in range(4):
        mem.add(ARX_BASE + w * (1 << 24), pack_arx(ar, w))
    mem.add(TW_BASE, np.stack([z.real, z.imag], axis=1).copy())
    rot = [3, 0, 1, 2]
    waves, lds = run(full_prog, None, rot, p_order=p, mem=mem)
    check_inverse(waves, lds, rot, A, 1e-9)
    hazards(full_prog, waves, f"A(f) order {p}")


def test_generated_text_is_current_and_assembles(tmp_path):
    """the committed .inc is what the generator produces, and every line is accepted by the gfx950 assembler"""
    inc = os.path.normpath(os.path.join(GEN, "..", "tf_inv64_body.inc"))
    fresh = tmp_path / "body.inc"
    K.emit_inc(str(fresh))
    assert open(inc).read() == fresh.read_text(), "csrc/tf_inv64_body.inc is stale: run python csrc/gen/k3gen.py"
    mc = "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("llvm-mc not available")
    import subprocess
    for name, opts in VARIANTS.items():
        prog = K.Gen(for_text=False, opts=opts).build()
        src = tmp_path / f"body_{name}.s"
        src.write_text("\n".join(prog.text_lines()) + "\n")
        r = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", str(tmp_path / "body.o"), str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, name + ": " + r.stderr[:3000]
