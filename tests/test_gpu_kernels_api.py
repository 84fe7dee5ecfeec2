"""GPU: the reference's kernel-level API (voxtral_kernels.h:18-159) exported by libvoxtral.so, one test per
function, each against the reference's own CPU function of the same name run live from oracle/_ref
(shape-generic, no weights needed), called through ctypes with the reference's signatures.  The product
library is loaded RTLD_LOCAL and the oracle is linked -Bsymbolic, so neither can resolve into the other
(checked below)."""
import ctypes as C

import numpy as np
import pytest

from conftest import have_ref
from oracle import vox_oracle as vo

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_ref("tiny"), reason="oracle/_ref not shipped")]

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int)
u16p = C.POINTER(C.c_uint16)


def fp(a):
    return a.ctypes.data_as(f32p)


@pytest.fixture(scope="module")
def libs():
    import voxtral_c_amd as v
    from oracle.ref_binding import RefLib
    if v.device_count() < 1:
        pytest.fail("no HIP device: the product has no CPU fallback")
    return v.lib, RefLib("tiny").lib


def both(libs, name, argtypes, make_args, outs):
    """Call `name` in the product and in the reference on identical copies of the arguments; return the
    output arrays of each."""
    res = []
    for L in libs:
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = None
        args, arrays = make_args()
        fn(*args)
        res.append([arrays[i].copy() for i in outs])
    return res


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def test_the_two_libraries_do_not_share_code(libs):
    ours, ref = libs
    a = C.cast(ours.vox_rms_norm, C.c_void_p).value
    b = C.cast(ref.vox_rms_norm, C.c_void_p).value
    assert a != b


@pytest.mark.parametrize("name,op", [("vox_add_inplace", lambda a, b, s: a + b), ("vox_mul_inplace", lambda a, b, s: a * b)])
def test_add_mul_inplace(libs, name, op):
    def mk():
        a, b = rnd(10007, 1), rnd(10007, 2)
        return (fp(a), fp(b), a.size), [a]
    (o,), (r,) = both(libs, name, [f32p, f32p, C.c_int], mk, [0])
    assert np.array_equal(o, r)


def test_axpy_scale_copy(libs):
    def mk():
        a, b = rnd(5001, 3), rnd(5001, 4)
        return (fp(a), C.c_float(0.37), fp(b), a.size), [a]
    (o,), (r,) = both(libs, "vox_axpy", [f32p, C.c_float, f32p, C.c_int], mk, [0])
    assert np.abs(o - r).max() < 1e-6           # fma vs mul+add

    def mk2():
        a = rnd(777, 5)
        return (fp(a), C.c_float(-1.75), a.size), [a]
    (o,), (r,) = both(libs, "vox_scale", [f32p, C.c_float, C.c_int], mk2, [0])
    assert np.array_equal(o, r)

    def mk3():
        a, b = np.zeros(333, np.float32), rnd(333, 6)
        return (fp(a), fp(b), a.size), [a]
    (o,), (r,) = both(libs, "vox_copy", [f32p, f32p, C.c_int], mk3, [0])
    assert np.array_equal(o, r)


@pytest.mark.parametrize("M,K,N", [(37, 129, 65), (1, 384, 200), (130, 64, 7)])
def test_matmul_and_matmul_t(libs, M, K, N):
    def mk(tr):
        def f():
            A, B, Cc = rnd((M, K), 7), rnd((N, K) if tr else (K, N), 8), np.zeros((M, N), np.float32)
            return (fp(Cc), fp(A), fp(B), M, K, N), [Cc]
        return f
    for name, tr in (("vox_matmul", False), ("vox_matmul_t", True)):
        (o,), (r,) = both(libs, name, [f32p, f32p, f32p, C.c_int, C.c_int, C.c_int], mk(tr), [0])
        assert np.abs(o - r).max() < 2e-4 * np.sqrt(K / 64.0), name


def test_linear_and_linear_nobias(libs):
    S, I, O = 19, 257, 130

    def mk():
        y, x, W, b = np.zeros((S, O), np.float32), rnd((S, I), 9), rnd((O, I), 10, 0.1), rnd(O, 11)
        return (fp(y), fp(x), fp(W), fp(b), S, I, O), [y]
    (o,), (r,) = both(libs, "vox_linear", [f32p] * 4 + [C.c_int] * 3, mk, [0])
    assert np.abs(o - r).max() < 1e-4

    def mk2():
        y, x, W = np.zeros((S, O), np.float32), rnd((S, I), 9), rnd((O, I), 10, 0.1)
        return (fp(y), fp(x), fp(W), S, I, O), [y]
    (o,), (r,) = both(libs, "vox_linear_nobias", [f32p] * 3 + [C.c_int] * 3, mk2, [0])
    assert np.abs(o - r).max() < 1e-4


@pytest.mark.parametrize("S,I,O", [(1, 3072, 1024), (38, 1280, 640), (5, 200, 33)])
def test_linear_bf16_variants(libs, S, I, O):
    W = vo.f32_to_bf16(rnd((O, I), 12, 1.0 / np.sqrt(I)))

    def mk():
        y, x, b = np.zeros((S, O), np.float32), rnd((S, I), 13), rnd(O, 14)
        return (fp(y), fp(x), W.ctypes.data_as(u16p), fp(b), S, I, O), [y]
    (o,), (r,) = both(libs, "vox_linear_bf16", [f32p, f32p, u16p, f32p] + [C.c_int] * 3, mk, [0])
    assert np.abs(o - r).max() < 5e-5

    def mk2():
        y, x = np.zeros((S, O), np.float32), rnd((S, I), 13)
        return (fp(y), fp(x), W.ctypes.data_as(u16p), S, I, O), [y]
    for name in ("vox_linear_nobias_bf16", "vox_matmul_t_bf16"):
        (o,), (r,) = both(libs, name, [f32p, f32p, u16p] + [C.c_int] * 3, mk2, [0])
        assert np.abs(o - r).max() < 5e-5, name


@pytest.mark.parametrize("cin,cout,L,ks,stride,pad", [(8, 12, 50, 3, 1, 1), (16, 5, 33, 5, 2, 2), (3, 70, 9, 3, 3, 0)])
def test_conv1d(libs, cin, cout, L, ks, stride, pad):
    ol = (L + 2 * pad - ks) // stride + 1

    def mk():
        out, x, w, b = np.zeros((cout, ol), np.float32), rnd((cin, L), 15), rnd((cout, cin * ks), 16, 0.3), rnd(cout, 17)
        return (fp(out), fp(x), fp(w), fp(b), cin, cout, L, ks, stride, pad), [out]
    (o,), (r,) = both(libs, "vox_conv1d", [f32p] * 4 + [C.c_int] * 6, mk, [0])
    assert np.abs(o - r).max() < 5e-5


@pytest.mark.parametrize("cin,cout,L,stride", [(128, 64, 41, 1), (64, 96, 41, 2), (64, 96, 40, 2), (16, 8, 1, 2)])
def test_causal_conv1d(libs, cin, cout, L, stride):
    """The conv-stem kernel of the batch encoder (voxtral_kernels.c:293-340), odd and even lengths."""
    ol = int(np.ceil((L - 3 + (3 - stride)) / stride + 1.0))

    def mk():
        out, x, w, b = np.zeros((cout, ol), np.float32), rnd((cin, L), 18), rnd((cout, cin * 3), 19, 0.2), rnd(cout, 20)
        return (fp(out), fp(x), fp(w), fp(b), cin, cout, L, 3, stride), [out]
    (o,), (r,) = both(libs, "vox_causal_conv1d", [f32p] * 4 + [C.c_int] * 5, mk, [0])
    assert np.abs(o - r).max() < 5e-5


@pytest.mark.parametrize("S,H", [(7, 1280), (3, 3072), (4, 250), (1, 37)])
def test_rms_norm(libs, S, H):
    def mk():
        out, x, w = np.zeros((S, H), np.float32), rnd((S, H), 21, 3.0), 1.0 + rnd(H, 22, 0.1)
        return (fp(out), fp(x), fp(w), S, H, C.c_float(1e-5)), [out]
    (o,), (r,) = both(libs, "vox_rms_norm", [f32p, f32p, f32p, C.c_int, C.c_int, C.c_float], mk, [0])
    assert np.abs(o - r).max() < 2e-5


def test_silu_gelu_softmax(libs):
    for name in ("vox_silu", "vox_gelu"):
        def mk():
            x = rnd(4099, 23, 3.0)
            return (fp(x), x.size), [x]
        (o,), (r,) = both(libs, name, [f32p, C.c_int], mk, [0])
        assert np.abs(o - r).max() < 2e-6, name

    def mk2():
        x = rnd((9, 1000), 24, 4.0)
        return (fp(x), 9, 1000), [x]
    (o,), (r,) = both(libs, "vox_softmax", [f32p, C.c_int, C.c_int], mk2, [0])
    assert np.abs(o - r).max() < 1e-6 and np.abs(o.sum(1) - 1).max() < 1e-5


@pytest.mark.parametrize("case", [
    # seq_q, seq_k, heads, kv_heads, head_dim, window, q_offset
    (40, 40, 4, 4, 64, 16, 0), (1, 300, 32, 8, 128, 8192, 299), (9, 50, 8, 2, 128, 0, 41),
    (12, 30, 6, 3, 32, 10, 18), (5, 5, 2, 1, 96, -1, 0), (3, 64, 8, 2, 64, 20, 61),
])
def test_causal_attention(libs, case):
    """Production geometries (64 MHA, 128 GQA 4:1) and others (any head_dim <= 256, any GQA ratio, no window)."""
    sq, sk, nh, nkv, hd, win, off = case

    def mk():
        out = np.zeros((sq, nh * hd), np.float32)
        q, k, v = rnd((sq, nh * hd), 25), rnd((sk, nkv * hd), 26), rnd((sk, nkv * hd), 27)
        return (fp(out), fp(q), fp(k), fp(v), sq, sk, nh, nkv, hd, C.c_float(1.0 / np.sqrt(hd)), win, off), [out]
    (o,), (r,) = both(libs, "vox_causal_attention", [f32p] * 4 + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_int], mk, [0])
    assert np.abs(o - r).max() < 2e-5, case


def test_rope_freqs_and_apply(libs):
    pos = np.array([0, 1, 2, 17, 749, 8191, 30000], np.int32)
    for dim in (64, 128):
        def mk():
            f = np.zeros((len(pos), dim // 2, 2), np.float32)
            return (fp(f), pos.ctypes.data_as(i32p), len(pos), dim, C.c_float(1e6)), [f]
        (o,), (r,) = both(libs, "vox_compute_rope_freqs", [f32p, i32p, C.c_int, C.c_int, C.c_float], mk, [0])
        assert np.abs(o - r).max() < 2e-6, dim          # same fp32 angle; cosf/sinf may differ in the last ulp

        def mk2():
            x = rnd((len(pos), 5 * dim), 28)
            return (fp(x), fp(r), len(pos), 5, dim), [x]
        (o2,), (r2,) = both(libs, "vox_apply_rope", [f32p, f32p, C.c_int, C.c_int, C.c_int], mk2, [0])
        assert np.abs(o2 - r2).max() < 1e-6
