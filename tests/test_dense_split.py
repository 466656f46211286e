"""dense_split_kernel (csrc/dense_split.hip, csrc/msplit.h: every fp32 operand split exactly into three bf16 planes, six bf16 MFMAs
per product, fp32 accumulation) against a float64 product: the claim to check is that the result stays in the ERROR CLASS of an
fp32 fma chain — not merely inside the 1e-5 parity bound — so the error is measured against sum|w||x| (the scale a rounding
error of a dot product lives on) next to the fp32-MFMA kernels' error on the same inputs (knob 17 = -1 routes around the split
core).  Shapes: the configs' layers (K known at compile time), and random ones through the run-time-K kernel: every remainder
of the 16-position k-block, one or two segments, both weight layouts, column tiles of 128 and 64, N not a multiple of 32."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# error of one output relative to sum_k |w_k x_k|: an fp32 fma chain of K terms is at 0.75-1.5e-7 (cdna_hip_programming.md §3);
# the split drops terms below 2^-21 of each product and accumulates in fp32 inside the matrix core
SPLIT_BOUND = 6e-7


@pytest.fixture(scope="module")
def gm():
    import torch
    assert torch.cuda.is_available()
    import gnnmp
    gnnmp.load()
    return gnnmp


def run_dense(gm, x, W, b, act, x2=None, W2=None, w_layout=0):
    import torch
    from gnnmp import _lib as L
    N, D1 = x.shape
    Dout = W.shape[0] if w_layout == 0 else W.shape[1]
    out = torch.empty((N, Dout), dtype=torch.float32, device="cuda")
    D2 = 0 if x2 is None else x2.shape[1]
    L.check(L.load().gnnmp_dense_f32(L.ptr(x), L.ptr(W), D1, W.stride(0), L.ptr(x2), L.ptr(W2), D2,
                                     0 if W2 is None else W2.stride(0), w_layout, L.ptr(b), act, L.ptr(out), N, Dout, L.stream_ptr()))
    return out


def case(gm, N, K1, K2, Dout, act, has_bias, w_layout, seed, scale_x=1.0):
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    two = K2 > 0
    x = torch.randn((N, K1), device="cuda", generator=g) * scale_x
    x2 = torch.randn((N, K2), device="cuda", generator=g) * scale_x if two else None
    Wfull = torch.randn((Dout, K1 + K2), device="cuda", generator=g) * 0.3
    b = torch.randn(Dout, device="cuda", generator=g) * 0.2 if has_bias else None
    if w_layout == 0:
        W1, W2 = Wfull[:, :K1], (Wfull[:, K1:] if two else None)
    else:
        Wt = Wfull.t().contiguous()
        W1, W2 = Wt[:K1], (Wt[K1:] if two else None)
    xin = torch.cat([x, x2], 1) if two else x
    pre = xin.double() @ Wfull.double().t()
    mag = xin.double().abs() @ Wfull.double().abs().t()          # sum_k |w_k x_k| per output
    if has_bias:
        pre = pre + b.double()
        mag = mag + b.double().abs()
    ref = torch.relu(pre) if act else pre
    y = run_dense(gm, x, W1, b, act, x2, W2, w_layout)
    before = gm.knob(17)
    gm.tune(17, -1)
    try:
        y32 = run_dense(gm, x, W1, b, act, x2, W2, w_layout)
    finally:
        gm.tune(17, before)
    e_split = float(((y.double() - ref).abs() / mag).max())
    e_f32 = float(((y32.double() - ref).abs() / mag).max())
    assert torch.equal(run_dense(gm, x, W1, b, act, x2, W2, w_layout), y), "not run-to-run identical"
    return e_split, e_f32, float(ref.abs().max()), float((y.double() - ref).abs().max())


CONFIG_SHAPES = [  # (N, K1, K2, Dout): the layer shapes of BASELINE.json's configs, N cut to a few thousand rows
    (5000, 100, 0, 100), (5003, 100, 0, 128), (4097, 128, 0, 128), (3001, 100, 100, 256), (2049, 16, 16, 128),
    (3333, 128, 128, 128), (2708, 64, 0, 64)]


@pytest.mark.parametrize("shape", CONFIG_SHAPES, ids=lambda s: "N%d_K%d+%d_D%d" % s)
def test_config_shapes_error_class(gm, shape):
    N, K1, K2, Dout = shape
    e_split, e_f32, _, _ = case(gm, N, K1, K2, Dout, 1, True, 0, 7)
    print(f"\n{shape}: max err / sum|wx|: split {e_split:.2e}  fp32-MFMA {e_f32:.2e}")
    assert e_split <= SPLIT_BOUND, (shape, e_split, e_f32)
    assert e_split <= 6 * max(e_f32, 5e-8), (shape, e_split, e_f32)


@pytest.mark.parametrize("seed", range(24))
def test_random_shapes_against_float64(gm, seed):
    rng = np.random.default_rng(3000 + seed)
    N = int(rng.choice([32, 33, 63, 65, 100, 1000, 4099, 20011]))
    K1 = int(rng.choice([4, 8, 12, 16, 20, 36, 52, 64, 100, 104, 108, 124, 128, 200, 260]))
    two = bool(rng.integers(0, 2))
    K2 = int(rng.choice([4, 16, 24, 100, 128])) if two else 0
    Dout = int(rng.choice([4, 8, 28, 32, 36, 64, 100, 112, 116, 128, 132, 256, 300]))
    act, has_bias, w_layout = int(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.integers(0, 2))
    tag = f"N={N} K={K1}+{K2} Dout={Dout} act={act} bias={has_bias} layout={w_layout}"
    e_split, e_f32, scale, eabs = case(gm, N, K1, K2, Dout, act, has_bias, w_layout, seed)
    assert eabs <= 1e-5 * scale, tag
    assert e_split <= SPLIT_BOUND, (tag, e_split, e_f32)


@pytest.mark.parametrize("scale", [1e-30, 1e-12, 1e12, 1e30])
def test_extreme_magnitudes(gm, scale):
    """bf16 planes have fp32's exponent range: no scaling is involved, tiny and huge operands behave like fp32"""
    e_split, e_f32, _, _ = case(gm, 1000, 100, 0, 128, 0, False, 0, 11, scale_x=scale)
    assert e_split <= SPLIT_BOUND, (scale, e_split, e_f32)


def test_non_finite_operands_take_the_exact_path(gm):
    """Inf / NaN operands: the split of Inf is (Inf, NaN, ...), so the tile's accumulators go NaN and the kernel recomputes the
    tile with fp32 fma loops: Inf * w = +-Inf like the reference's product, NaN stays NaN, and only the rows that hold them are
    affected (-Inf is what an empty max-aggregation hands to graph_conv / sage_conv)."""
    import torch
    N, K, Dout = 200, 100, 128
    x = torch.randn((N, K), device="cuda")
    x[37, 5] = float("nan")
    x[120, 99] = float("inf")
    x[121] = -float("inf")
    W = torch.randn((Dout, K), device="cuda")
    W[3] = W[3].abs() + 0.1                       # column 3: all weights positive => row 121 gives -Inf there
    W[4] = 0.5                                    # exactly representable in bf16: its lower planes are zero (Inf * 0 hazard)
    y = run_dense(gm, x, W, None, 0)
    bad = ~torch.isfinite(y).all(1)
    assert bad.nonzero().flatten().tolist() == [37, 120, 121]
    ref = (x.double() @ W.double().t())
    good = torch.isfinite(ref).all(1)
    assert float((y[good].double() - ref[good]).abs().max()) <= 1e-5 * float(ref[good].abs().max())
    assert torch.isnan(y[37]).all()
    assert torch.equal(torch.isinf(y[120]), torch.isinf(ref[120])) and torch.equal(torch.sign(y[120]), torch.sign(ref[120]).float())
    assert y[121, 3] == -float("inf") and y[121, 4] == -float("inf")
    gm.tune(17, -1)
    try:
        y32 = run_dense(gm, x, W, None, 0)
    finally:
        gm.tune(17, 0)
    assert torch.equal(torch.isnan(y), torch.isnan(y32)) and torch.equal(torch.isinf(y), torch.isinf(y32))


# ---- dense_wreg_kernel (csrc/dense_wreg.hip): W * vcat(xi, m) => 256 with W in registers, two segments of 64 / 100 / 128 (K <= 200) ------
WREG_SHAPES = [(100, 100), (64, 64), (64, 100), (100, 64), (64, 128), (128, 64), (200, 0)]


@pytest.fixture
def wreg_small(gm):
    """knob 19 bit 9: the register-resident kernel from 4 096 rows on (default 32 768), so that these sizes reach it"""
    before = gm.knob(19)
    gm.tune(19, before | 512)
    yield before | 512
    gm.tune(19, before)


def _wreg_case(gm, N, act, has_bias, w_layout, seed, K1=100, K2=100):
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    x = torch.randn((N, K1), device="cuda", generator=g)
    m = torch.randn((N, K2), device="cuda", generator=g) if K2 else None
    Wfull = torch.randn((256, K1 + K2), device="cuda", generator=g) * 0.3
    b = torch.randn(256, device="cuda", generator=g) * 0.2 if has_bias else None
    if w_layout == 0:
        W1, W2 = Wfull[:, :K1], (Wfull[:, K1:] if K2 else None)      # views of one matrix, row stride K1 + K2 (sage_conv's W * vcat(xi, m))
    else:
        Wt = Wfull.t().contiguous()
        W1, W2 = Wt[:K1], (Wt[K1:] if K2 else None)
    return x, m, W1, W2, b, Wfull


@pytest.mark.parametrize("N", [4096, 4099, 8192 + 31, 70001])      # fewer tiles than CUs (blocks with no tile), ragged ends, several tiles a block
@pytest.mark.parametrize("w_layout", [0, 1])
@pytest.mark.parametrize("K", WREG_SHAPES, ids=lambda k: f"{k[0]}+{k[1]}")
def test_register_resident_kernel_is_bit_identical_to_the_lds_kernel(gm, wreg_small, N, w_layout, K):
    import torch
    K1, K2 = K
    x, m, W1, W2, b, Wfull = _wreg_case(gm, N, 1, True, w_layout, N + w_layout, K1, K2)
    y = run_dense(gm, x, W1, b, 1, m, W2, w_layout)
    gm.tune(19, wreg_small | 64)                                 # bit 6: never dense_wreg_kernel
    try:
        y_lds = run_dense(gm, x, W1, b, 1, m, W2, w_layout)
    finally:
        gm.tune(19, wreg_small)
    assert torch.equal(y, y_lds)
    xc = torch.cat([x, m], 1) if K2 else x
    ref = torch.relu(xc.double() @ Wfull.double().t() + b.double())
    mag = xc.double().abs() @ Wfull.double().abs().t() + b.double().abs()
    assert float(((y.double() - ref).abs() / mag).max()) <= SPLIT_BOUND
    assert torch.equal(run_dense(gm, x, W1, b, 1, m, W2, w_layout), y)


def test_register_resident_kernel_non_finite_operands(gm, wreg_small):
    import torch
    N = 5000
    x, m, W1, W2, _, Wfull = _wreg_case(gm, N, 0, False, 0, 99)
    x[37, 5] = float("nan")
    m[4100, 99] = float("inf")
    m[4999] = -float("inf")                                      # an empty max-aggregation's row, in the last (ragged) tile
    y = run_dense(gm, x, W1, None, 0, m, W2, 0)
    bad = ~torch.isfinite(y).all(1)
    assert bad.nonzero().flatten().tolist() == [37, 4100, 4999]
    ref = torch.cat([x, m], 1).double() @ Wfull.double().t()
    good = torch.isfinite(ref).all(1)
    assert float((y[good].double() - ref[good]).abs().max()) <= 1e-5 * float(ref[good].abs().max())
    assert torch.isnan(y[37]).all()
    assert torch.equal(torch.isinf(y[4100]), torch.isinf(ref[4100])) and torch.equal(torch.sign(y[4100]), torch.sign(ref[4100]).float())
    gm.tune(19, wreg_small | 64)
    try:
        y_lds = run_dense(gm, x, W1, None, 0, m, W2, 0)
    finally:
        gm.tune(19, wreg_small)
    assert torch.equal(torch.isnan(y), torch.isnan(y_lds)) and torch.equal(y[good], y_lds[good])
