"""CPU: the arithmetic of the three-product fp16 kernels (csrc/f16x3.h, linear_f16x3.hip, gemm_f16x3_stream.hip,
window_attn_f16.hip with TERMS = 3), restated in numpy -- the error model behind the tolerances written in
tests/test_ops_gpu.py for the fp32 Linears, the 3 x 3 convolution and the default Swin window attention.

An fp32 operand row is scaled by a power of two into fp16's range and held as TWO fp16 parts (round to nearest even),
h = fp16(x'), m = fp16(x' - h); a product x * w is accumulated in fp32 from three of the four part products,
m*h' + h*m' + h*h' (the dropped m*m' is <= 2^-22 relative).  W rows are scaled by their maxima into [2^14, 2^15); x rows carry a
RUNNING scale that is set for 2^12 and re-set (accumulators rescaled by the exact ratio) when a later k-step's maximum
would leave three binades of room (linear_f16x3.hip `group`)."""
import numpy as np
import pytest


def split2(x):
    """float32 (already in fp16's range) -> (h, m) as float32 values that are representable in fp16"""
    h = x.astype(np.float16)
    m = (x - h.astype(np.float32)).astype(np.float16)
    return h.astype(np.float32), m.astype(np.float32)


def exponent(a):
    """e with 2^e <= a < 2^(e+1) from the bit pattern, clamped as l3_scale does (zero rows keep a finite scale)"""
    e = ((np.asarray(a, np.float32).view(np.uint32) >> 23) & 255).astype(np.int64) - 127
    return np.clip(e, -100, 128)


def gemm_f16x3(x, w, kstep=32, record=None):
    """y = x @ w.T the way the W-resident kernel computes it; float64 sums of exact part products stand in for the MFMA's
    fp32 accumulation (rounded to fp32 after every k-step)"""
    M, K = x.shape
    ew = exponent(np.abs(w).max(1))
    ws = (w * np.exp2(14 - ew)[:, None].astype(np.float32)).astype(np.float32)          # exact: a power of two
    wh, wm = split2(ws)
    winv = np.exp2(ew - 14).astype(np.float32)
    acc = np.zeros((M, w.shape[0]), np.float32)
    eset = np.full(M, -1000, np.int64)
    resets = np.zeros(M, np.int64)
    for k0 in range(0, K, kstep):
        xs_raw = x[:, k0:k0 + kstep]
        enew = exponent(np.abs(xs_raw).max(1))
        need = enew > eset + 2
        ratio = np.where(need, np.exp2(np.maximum(eset - enew, -126).astype(np.float64)), 1.0).astype(np.float32)
        acc = (acc * ratio[:, None]).astype(np.float32)                                 # exact unless it underflows
        eset = np.where(need, enew, eset)
        resets += need
        sx = np.exp2((12 - eset).astype(np.float64)).astype(np.float32)
        xh, xm = split2((xs_raw * sx[:, None]).astype(np.float32))
        wk_h, wk_m = wh[:, k0:k0 + kstep], wm[:, k0:k0 + kstep]
        part = (xh.astype(np.float64) @ wk_m.T.astype(np.float64) + xm.astype(np.float64) @ wk_h.T.astype(np.float64)
                + xh.astype(np.float64) @ wk_h.T.astype(np.float64))
        acc = (acc.astype(np.float64) + part).astype(np.float32)
    if record is not None:
        record["resets"] = resets
    sx_inv = np.exp2((eset - 12).astype(np.float64)).astype(np.float32)
    return (acc * sx_inv[:, None]) * winv[None, :]


@pytest.mark.parametrize("scale", [1.0e30, 777.0, 1.0, 3.0e-3, 1.0e-30])
def test_two_fp16_parts_hold_22_bits_of_a_scaled_row(scale):
    """Why rows are scaled: after l3_scale a row's maximum sits in [2^12, 2^13) and every element within ten binades of it has
    both parts in fp16's NORMAL range (h + m == x' to 2^-22 relative); smaller elements degrade gracefully -- their parts
    become fp16 subnormals, the absolute error stays below 2^-25, i.e. 2^-37 of the row maximum.  Unscaled, a value near 1e-3
    would keep 11 + ~3 bits."""
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200_000) * scale).astype(np.float32)
    s = np.exp2(12.0 - float(exponent(np.abs(x).max())))
    xs = (x * np.float32(s)).astype(np.float32)                                  # exact: a power of two
    assert 2.0 ** 12 <= np.abs(xs).max() < 2.0 ** 13
    h, m = split2(xs)
    err = np.abs(h.astype(np.float64) + m.astype(np.float64) - xs.astype(np.float64))
    big = np.abs(xs) >= 2.0 ** 2
    assert (err[big] <= np.abs(xs[big]) * 2.0 ** -22).all()
    assert err.max() <= 2.0 ** -10 and (err[~big] <= np.maximum(np.abs(xs[~big]) * 2.0 ** -22, 2.0 ** -25)).all()
    if scale == 3.0e-3:                                                          # the same values without the scaling
        h0, m0 = split2(x)
        e0 = np.abs(h0.astype(np.float64) + m0.astype(np.float64) - x.astype(np.float64))
        assert (e0 / np.abs(x).max()).max() > 20 * (err / np.abs(xs).max()).max()


def test_three_products_reach_fp32_accuracy_two_do_not():
    rng = np.random.default_rng(1)
    M, K, N = 64, 256, 48
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) * K ** -0.5).astype(np.float32)
    ref = x.astype(np.float64) @ w.T.astype(np.float64)
    e32 = np.abs((x @ w.T).astype(np.float64) - ref).max()                      # an fp32 GEMM's own error
    y = gemm_f16x3(x, w)
    e3 = np.abs(y.astype(np.float64) - ref).max()
    assert e3 <= max(2.0 * e32, 2e-6), (e3, e32)                                # the GPU tests allow max(4 e32, 5e-6)
    # the per-product bound: |error| <= 2^-21.7 * sum |x| |w|  (+ the fp32 accumulation)
    budget = (np.abs(x).astype(np.float64) @ np.abs(w).T.astype(np.float64)) * 2.0 ** -21.7
    assert (np.abs(y.astype(np.float64) - ref) <= budget + 8 * e32).all()
    # h * h' alone is a plain fp16 GEMM: three orders of magnitude worse
    xh = x.astype(np.float16).astype(np.float64)
    whh = w.astype(np.float16).astype(np.float64)
    assert np.abs(xh @ whh.T - ref).max() > 100 * e3


def test_running_row_scale_follows_a_growing_row_and_never_overflows():
    """Rows whose magnitude grows by 2^40 along k (the scale is re-set several times, every earlier partial sum rescaled by an
    exact power of two), rows at 1e-30 and 1e+30, a zero row: results stay at fp32 accuracy relative to the row's own scale."""
    rng = np.random.default_rng(2)
    M, K, N = 12, 512, 32
    x = rng.standard_normal((M, K)).astype(np.float32)
    x[0] *= np.exp2(np.linspace(-20, 20, K)).astype(np.float32)
    x[1] *= np.float32(1e-30)
    x[2] *= np.float32(1e30)
    x[3] = 0.0
    x[4, :480] = 0.0                                                             # the first k-steps are all zero
    w = (rng.standard_normal((N, K)) * K ** -0.5).astype(np.float32)
    w[5] *= np.float32(1e-20)
    rec = {}
    y = gemm_f16x3(x, w, record=rec)
    assert np.isfinite(y).all() and (y[3] == 0).all()
    assert rec["resets"][0] >= 5 and rec["resets"][5] == 1                     # the growing row re-set its scale; a stationary one never
    ref = x.astype(np.float64) @ w.T.astype(np.float64)
    scale = np.abs(x).astype(np.float64) @ np.abs(w).T.astype(np.float64)
    ok = scale > 1e-36                                                           # (row 1 x column 5 is 1e-50: below fp32's range)
    assert (np.abs(y.astype(np.float64) - ref)[ok] <= scale[ok] * 2.0 ** -20).all()


def test_splitting_a_product_needs_both_parts_from_the_rounded_product():
    """DESIGN.md toolchain hazard 16: q * s split as h = fp16(RN32(q * s)) but m = fp16(q * s - h) with the UNROUNDED product (what
    fp contraction makes of `m = (half)(q * s - h)`: v_fma_mixlo_f16) disagrees about h at fp16 ties -- h + m is then a whole
    fp16 ulp (2^-11 relative) off.  With both parts from the fp32-rounded product the pair holds 22 bits."""
    rng = np.random.default_rng(3)
    q = rng.standard_normal(2_000_000).astype(np.float32)
    s = np.float32(32 ** -0.5)                                                   # head_dim^-0.5: not a power of two
    p64 = q.astype(np.float64) * np.float64(s)                                   # the unrounded product
    p32 = (q * s).astype(np.float32)                                             # what `float p = q * s` holds
    h = p32.astype(np.float16)
    m_good = (p32 - h.astype(np.float32)).astype(np.float16)
    good = np.abs(h.astype(np.float64) + m_good.astype(np.float64) - p64)
    assert (good <= np.abs(p64) * 2.0 ** -21 + 2.0 ** -25).all()             # (m is an fp16 subnormal below |p| ~ 0.1)
    # the contracted form rounds h from the unrounded product: near an fp16 tie the two roundings pick different neighbours
    h_fused = p64.astype(np.float16)
    differ = h_fused != h
    assert differ.any(), "no tie among 2e6 samples?"
    m_bad = (p64 - h_fused.astype(np.float64)).astype(np.float16)                # m belongs to h_fused ...
    bad = np.abs(h.astype(np.float64) + m_bad.astype(np.float64) - p64)          # ... but is paired with h
    assert (bad[differ] >= np.abs(p64[differ]) * 2.0 ** -12.1).all()
    assert differ.mean() < 1e-3                                                  # rare: one query in a few thousand on the GPU
    # power-of-two multipliers (the GEMMs' row scales) are exact: the hazard cannot occur there
    p2 = (q * np.float32(2.0 ** -3)).astype(np.float64)
    assert np.array_equal(p2, q.astype(np.float64) * 2.0 ** -3)


def test_window_attention_three_product_error_model():
    """softmax(q k^T * scale + bias) v for one 49-token window with every matrix operand as two fp16 parts and three products:
    <= 4e-6 absolute against fp64 (the GPU tolerance), an fp16-operand evaluation of the same window is ~1e-3."""
    rng = np.random.default_rng(4)
    n, hd = 49, 32
    q = rng.standard_normal((n, hd)).astype(np.float32)
    k = rng.standard_normal((n, hd)).astype(np.float32)
    v = rng.standard_normal((n, hd)).astype(np.float32)
    bias = (rng.standard_normal((n, n)) * 0.5).astype(np.float32)
    scale = np.float32(hd ** -0.5)

    def mm3(a, b):
        ah, am = split2(a)
        bh, bm = split2(b)
        return (am.astype(np.float64) @ bh.T.astype(np.float64) + ah.astype(np.float64) @ bm.T.astype(np.float64)
                + ah.astype(np.float64) @ bh.T.astype(np.float64)).astype(np.float32)
    s3 = mm3((q * scale).astype(np.float32), k) + bias
    p3 = np.exp((s3 - s3.max(1, keepdims=True)).astype(np.float32))
    o3 = mm3(p3, v.T.copy()) / p3.sum(1, keepdims=True)
    s64 = (q.astype(np.float64) * np.float64(scale)) @ k.T.astype(np.float64) + bias
    p64 = np.exp(s64 - s64.max(1, keepdims=True))
    o64 = (p64 @ v.astype(np.float64)) / p64.sum(1, keepdims=True)
    assert np.abs(o3 - o64).max() < 4e-6
    s16 = ((q * scale).astype(np.float16).astype(np.float64) @ k.astype(np.float16).astype(np.float64).T) + bias
    p16 = np.exp(s16 - s16.max(1, keepdims=True))
    o16 = (p16.astype(np.float16).astype(np.float64) @ v.astype(np.float16).astype(np.float64)) / p16.sum(1, keepdims=True)
    assert np.abs(o16 - o64).max() > 50 * np.abs(o3 - o64).max()


def test_split_as_two_single_rounding_fmas_gives_the_same_parts():
    """csrc/f16x3.h l3_split8 computes h = fp16(v * s) and m = fp16(v * s - h) as two mixed-precision FMAs with ONE rounding each
    (v_fma_mix{lo,hi}_f16), where the plain expression rounds v * s to fp32 first, subtracts in fp32 and rounds again.  The two agree
    bit for bit because s is a power of two (v * s is exact in fp32) and x - fp16(x) is exact in fp32 whenever fp16(x) is finite:
    checked here in float64 (which holds every intermediate exactly) over values across fp16's whole range, incl. its subnormals."""
    rng = np.random.default_rng(5)
    v = (rng.standard_normal(200000) * np.exp2(rng.integers(-40, 8, 200000))).astype(np.float32)
    v[:8] = [0.0, -0.0, 1.0, -1.0, 2.0 ** -30, 65504.0 / 4096, 3.0e-9, -7.7]
    for e in (12, 4, 0, -6):                                   # the scale places a row's maximum near 2^12; smaller ones for the tail
        s = np.float32(2.0 ** e)
        exact = v.astype(np.float64) * float(s)                # exact product
        keep = np.abs(exact) < 60000.0                         # (the kernels' scale keeps every element below fp16's maximum)
        xs32 = (v * s).astype(np.float32)
        assert np.array_equal(xs32.astype(np.float64)[keep], exact[keep])                  # v * s exact in fp32
        h_chain = xs32.astype(np.float16)
        h_fma = exact.astype(np.float16)                       # one rounding from the exact product
        assert np.array_equal(h_chain[keep].view(np.uint16), h_fma[keep].view(np.uint16))
        d32 = (xs32 - h_chain.astype(np.float32)).astype(np.float32)
        d_exact = exact - h_fma.astype(np.float64)
        assert np.array_equal(d32.astype(np.float64)[keep], d_exact[keep])                 # the difference is exact in fp32
        m_chain = d32.astype(np.float16)
        m_fma = d_exact.astype(np.float16)                     # one rounding from the exact difference
        assert np.array_equal(m_chain[keep].view(np.uint16), m_fma[keep].view(np.uint16))
