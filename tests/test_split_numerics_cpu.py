"""CPU: the arithmetic of the split-bf16 kernels (csrc/mask_decode.hip: skinny_gemm_bf16x6*, csrc/linear_split.hip), restated
in numpy: every fp32 value is split EXACTLY into three bf16 parts by truncation, bf16 x bf16 products are exact in fp32, and
the six retained terms reproduce an fp32 contraction to fp32 rounding (the three dropped terms are <= 3 * 2^-24 per product).
This is the error model behind the tolerances written in tests/test_ops_gpu.py for those kernels."""
import numpy as np
import pytest


def split3(x):
    """x (float32 array) -> h, m, l with h + m + l == x exactly and each part representable in bfloat16."""
    xb = x.view(np.uint32)
    h = (xb & np.uint32(0xFFFF0000)).view(np.float32)
    r = x - h
    m = (r.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    return h, m, r - m


# (|x| below ~2^-100 puts the last part into the subnormal range, where it no longer fits 8 significant bits: what is
# lost there is below 2^-126 in absolute terms)
@pytest.mark.parametrize("scale", [1e-20, 1e-3, 1.0, 777.0, 1e30])
def test_three_way_split_is_exact(scale):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200_000) * scale).astype(np.float32)
    x[:4] = [0.0, -0.0, np.float32(scale), -np.float32(scale)]
    h, m, l = split3(x)
    for part in (h, m, l):
        assert ((part.view(np.uint32) & 0xFFFF) == 0).all()            # a bfloat16 value
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    nz = x != 0
    assert (np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -7).all() and (np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -15).all()
    # power-of-two scaling commutes with the split (why `out(2e) == 2 out(e)` holds bit for bit on the GPU)
    h2, m2, l2 = split3(x * np.float32(2.0))
    assert np.array_equal(h2, 2 * h) and np.array_equal(m2, 2 * m) and np.array_equal(l2, 2 * l)


def test_products_of_parts_are_exact_in_fp32():
    rng = np.random.default_rng(1)
    a = split3(rng.standard_normal(100_000).astype(np.float32))
    b = split3(rng.standard_normal(100_000).astype(np.float32))
    for pa in a:
        for pb in b:
            assert np.array_equal((pa * pb).astype(np.float64), pa.astype(np.float64) * pb.astype(np.float64))


def test_six_terms_reach_fp32_accuracy_three_do_not():
    rng = np.random.default_rng(2)
    Q, K, N = 100, 256, 1024
    A = (rng.standard_normal((Q, K)) * 0.5).astype(np.float32)
    B = (rng.standard_normal((K, N)) * 0.5).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    (ah, am, al), (bh, bm, bl) = split3(A), split3(B)
    acc = np.zeros((Q, N), np.float32)
    for x, y in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):      # the kernels' order: small terms first
        acc = acc + x @ y
    err6 = np.abs(acc - ref).max()
    err_f32 = np.abs(A @ B - ref).max()
    dropped = np.abs(am.astype(np.float64) @ bl + al.astype(np.float64) @ bm + al.astype(np.float64) @ bl).max()
    assert dropped < 3 * 2.0 ** -24 * (np.abs(A).astype(np.float64) @ np.abs(B)).max()
    assert err6 < 2.0 * err_f32 + 2e-6                     # same class as an fp32 GEMM
    err3 = np.abs((am @ bh + ah @ bm + ah @ bh) - ref).max()
    assert err3 > 20 * err6                                 # a 2-way split (3 terms) is NOT enough for the 1e-3 parity bar
