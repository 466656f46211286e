"""dense_t16_kernel (csrc/dense_t16.hip: fp32 16x16x4 MFMA, operands straight from HBM) on random shapes inside and at the edges of
its envelope, against a float64 product and against the round-1 kernels on the same inputs (knob 6 = 2 routes around it):
K a multiple of 4 up to 128 per segment (every remainder class of the 16-float k-block: rem = 0, 1, 2, 3), Dout a multiple of
4 from 4 to several 128-column tiles, one or two segments, both weight layouts, N not a multiple of the 16-node tile."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("seed", range(24))
def test_random_shapes_against_float64_and_the_round1_kernels(gm, seed):
    import torch
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([16, 17, 31, 33, 100, 1000, 4099, 20011]))
    K1 = int(rng.choice([4, 8, 12, 16, 20, 36, 52, 64, 100, 104, 108, 124, 128]))
    two = bool(rng.integers(0, 2))
    K2 = int(rng.choice([4, 16, 24, 100, 128])) if two else 0
    Dout = int(rng.choice([4, 8, 28, 32, 36, 64, 100, 112, 116, 128, 132, 256, 300]))
    act = int(rng.integers(0, 2))
    has_bias = bool(rng.integers(0, 2))
    w_layout = int(rng.integers(0, 2))
    x = torch.randn((N, K1), device="cuda")
    x2 = torch.randn((N, K2), device="cuda") if two else None
    Wfull = torch.randn((Dout, K1 + K2), device="cuda") * 0.3           # [Dout][K] row-major; segments are column slices
    b = torch.randn(Dout, device="cuda") * 0.2 if has_bias else None
    if w_layout == 0:
        W1, W2 = Wfull[:, :K1], (Wfull[:, K1:] if two else None)
    else:                                                                  # Julia layout: C row-major [K][Dout]
        Wt = Wfull.t().contiguous()
        W1, W2 = Wt[:K1], (Wt[K1:] if two else None)
    ref = x.double() @ Wfull[:, :K1].double().t()
    if two:
        ref = ref + x2.double() @ Wfull[:, K1:].double().t()
    if has_bias:
        ref = ref + b.double()
    if act:
        ref = torch.relu(ref)
    tag = f"N={N} K={K1}+{K2} Dout={Dout} act={act} bias={has_bias} layout={w_layout}"
    y = run_dense(gm, x, W1, b, act, x2, W2, w_layout)
    scale = float(ref.abs().max()) + 1e-30
    assert float((y.double() - ref).abs().max()) <= 1e-5 * scale, tag
    gm.tune(6, 2)
    try:
        y1 = run_dense(gm, x, W1, b, act, x2, W2, w_layout)
    finally:
        gm.tune(6, 0)
    assert float((y1.double() - ref).abs().max()) <= 1e-5 * scale, tag + " (round-1 kernels)"
    assert float((y - y1).abs().max()) <= 2e-5 * scale, tag + " (the two kernel families disagree)"
    # run-to-run identical
    assert torch.equal(run_dense(gm, x, W1, b, act, x2, W2, w_layout), y)


def test_nan_inf_rows_do_not_leak_into_other_rows(gm):
    """a NaN / Inf input row makes ITS output row non-finite and no other (16 nodes share a tile; the k-block tails of the
    operand image must never multiply garbage)"""
    import torch
    N, K, Dout = 200, 100, 100
    x = torch.randn((N, K), device="cuda")
    x[37, 5] = float("nan")
    x[120, 99] = float("inf")
    W = torch.randn((Dout, K), device="cuda")
    y = run_dense(gm, x, W, None, 0)
    bad = ~torch.isfinite(y).all(1)
    assert bad.nonzero().flatten().tolist() == [37, 120]
