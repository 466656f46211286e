"""csrc/arena.hip + gnnmp/placement.py (opt-in placement-aware output buffers): the arena finds two placement classes with its probe, hands
out memory of either, recognises its own and foreign buffers; a layer that opts in returns bit-identical results from persistent buffers in the
right ranges; the plain path is untouched when the switch is off."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ar():
    import torch
    import gnnmp
    from gnnmp import _lib as L, placement
    free, _ = torch.cuda.mem_get_info()
    if free < (48 << 30):
        pytest.skip("the arena holds up to 32 GiB of blocks while it looks for the placement classes; this device has less free")
    a = placement.Arena(gib_per_class=4, n_classes=3)        # never raises for want of classes: it comes back with what it found
    if a.n_classes < 2:
        pytest.skip(f"this device did not show two placement classes within the probing budget: {a.info()}")
    return a


def test_arena_finds_classes_inside_its_budget_and_tells_them_apart(ar):
    import torch
    info = ar.info()
    assert info["bytes_per_class"] == 4 << 30 and info["ranges"] in (2, 3) and info["ranges"] == ar.n_classes
    assert 3 <= info["chunks_created"] <= 16, info            # the default budget: 32 GiB of 2 GiB blocks held while probing
    assert info["create_ms"] < 1500, info                     # (~0.3 s of probing + the hipMallocs / hipFrees around it)
    assert info["gave_up_on_budget"] == (min(info["blocks_per_range"]) < 2 or info["ranges"] < 3)
    # the probe saw two clusters at least 6 % apart (that is what makes two classes)
    assert info["probe_us_same_class"] > 1.06 * info["probe_us_two_classes"] > 0
    bufs = [ar.alloc((1 << 20, 128), c) for c in range(ar.n_classes)]
    for c, b in enumerate(bufs):
        assert ar.class_of(b) == c
    bufs[0].fill_(1.0); bufs[1].fill_(2.0)
    assert float(bufs[0].sum()) == float(1 << 27) and float(bufs[1].sum()) == float(2 << 27)
    # an address INSIDE a range answers from the range (an offset view is outside the cache of known addresses)
    for c, t in enumerate(bufs[:2]):
        assert ctypes_class(ar, t[4096:]) == c
    with pytest.raises(Exception):
        ar.alloc((16, 16), ar.n_classes)                      # no such range
    ar.reset()
    assert ar.info()["used"] == (0,) * ar.n_classes


def test_a_buffer_larger_than_a_block(ar):
    """gnnmp_arena_alloc: SAGEConv's (N, 256) output on the products shape is 2.5 GB — more than a 2 GiB block.  It is an allocation of its
    own, classified window by window; served in the class asked for or refused (None: the caller allocates as usual)"""
    import torch
    ar.reset()
    served = 0
    for c in range(ar.n_classes):
        big = ar.alloc((5 << 27, 1), c)                       # 2.5 GiB
        if big is None:
            continue
        served += 1
        assert ar.class_of(big) == c
        big[0] = 1.0; big[-1] = 2.0; big[(1 << 29) - 1] = 3.0; big[1 << 29] = 4.0
        torch.cuda.synchronize()
        assert (float(big[0]), float(big[-1]), float(big[(1 << 29) - 1]), float(big[1 << 29])) == (1.0, 2.0, 3.0, 4.0)
        p0 = big.data_ptr()
        del big
        ar.reset()
        again = ar.alloc((5 << 27, 1), c)                     # after a reset the same buffer is handed out again (no second probe)
        assert again is not None and again.data_ptr() == p0
    ar.reset()
    assert ar.info()["used"] == (0,) * ar.n_classes
    print("buffers of 2.5 GiB served in", served, "of", ar.n_classes, "classes")


def test_a_tiny_budget_is_not_an_error():
    """out of budget = the arena comes back with what it found (gnnmp.h): two blocks can show at most two classes"""
    import torch
    from gnnmp import placement
    a = placement.Arena(gib_per_class=2, n_classes=3, max_probe_gib=4)
    info = a.info()
    assert info["chunks_created"] <= 2 and info["ranges"] in (1, 2) and info["gave_up_on_budget"]
    assert a.n_classes == info["ranges"]
    b = a.alloc((1 << 20, 16), 0)
    assert b is not None and a.class_of(b) == 0


def ctypes_class(ar, t):
    """gnnmp_arena_class_of on an address INSIDE a range answers from the range; to exercise the probe, classify through a foreign alias:
    the same physical memory is not reachable under another address here, so the probe path is run on a torch allocation instead and
    only required to return a valid class"""
    import ctypes
    from gnnmp import _lib as L
    out = ctypes.c_int(-1)
    L.check(L.load().gnnmp_arena_class_of(ar.handle, L.ptr(t), t.numel() * 4, ctypes.byref(out), L.stream_ptr()))
    return out.value


def test_foreign_buffers_are_probed(ar):
    import torch
    x = torch.randn((1 << 21, 128), device="cuda")        # 1 GiB of ordinary (torch) memory
    c = ar.class_of(x)
    assert c in range(ar.n_classes + 1)
    assert ar.class_of(x) == c                            # cached per buffer
    assert ar.class_of(torch.zeros(16, device="cuda")) == ar.n_classes      # too small to matter


def test_placed_layers_are_bit_identical(ar, monkeypatch):
    import torch
    import gnnmp
    from gnnmp import placement, synth
    monkeypatch.setattr(placement, "MIN_BYTES", 0)
    monkeypatch.setattr(placement, "_arena", [ar, False])
    N, E, D = 70000, 900000, 100
    s, t = synth.arxiv_like(N=N, E=E, alpha=2.0, seed=3)
    g = gnnmp.GNNGraph(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), num_nodes=N)
    x = torch.from_numpy(synth.features(N, D, seed=4)).cuda()
    gcn = gnnmp.GCNConv((D, D), "relu", seed=1)
    gat = gnnmp.GATConv((D, 16), "relu", heads=8, seed=2)
    gnnmp.tune(14, 16)                      # the fused layer kernel on this small graph (the path that takes an output buffer)
    try:
        ref_gcn, ref_gat = gcn(g, x).clone(), gat(g, x).clone()
        gcn.place_outputs = gat.place_outputs = True
        cx = ar.class_of(x)
        for it in range(12):
            y1, y2 = gcn(g, x), gat(g, x)
            assert torch.equal(y1, ref_gcn) and torch.equal(y2, ref_gat), it
            torch.cuda.synchronize()        # (lets the polled events complete; the layers themselves never wait)
        # where the buffers lie: GCN's output not in x's class; Wx and the attention output in different classes; the trials are over
        assert ar.class_of(y1) in range(ar.n_classes) and ar.class_of(y1) != cx
        (wx,) = [b for k, b in gat._placed.items() if k[0] == "Wx" and b is not None]
        assert ar.class_of(wx) != cx and ar.class_of(y2) != ar.class_of(wx)
        for layer in (gcn, gat):
            (ch,) = [c for k, c in layer._placed.items() if k[0] == "out" and c is not False]
            assert ch.settled      # (x is ordinary torch memory here: its class may be none of the arena's)
            assert ch.times_ms is None or len(ch.times_ms) in (2, 3)      # (None: one candidate only — a two-class arena)
        a, b = gat(g, x), gat(g, x)
        assert a.data_ptr() == b.data_ptr()     # persistent: the documented aliasing of the opt-in
        gcn.place_outputs = gat.place_outputs = False
        c, d = gat(g, x), gat(g, x)
        assert torch.equal(c, ref_gat) and c.data_ptr() != d.data_ptr()
    finally:
        gnnmp.tune(14, 0)
