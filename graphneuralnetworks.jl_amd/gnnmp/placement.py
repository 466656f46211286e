"""Placement-aware output buffers for the gather kernels of the hot path (opt-in) — the host side of csrc/arena.hip.

Measured on MI355X (tools/experiments/placement_probe.py, placement_map*.py, vmm_probe.py; profiles/README.md, round 4): the 288 GB of HBM3E fall into three
placement classes of 96 GiB of physical memory each.  The one-pass attention kernel and the fused GCN layer kernel (~27 random row reads
per row written) run 6 % slower when the gathered matrix and the output lie in the SAME class than when they lie in two — 5.15 vs 4.84 ms
and 4.94 vs 4.66 ms on the products shape; same binary, same data, same predecessors on the stream.  hipMalloc (and so torch's allocator)
pairs buffers by luck: that was the unexplained gap between the kernel "alone" and "in the step" of round 3, and much of the box-to-box
spread.

The C ABI leaves allocation to its caller, so the policy lives here: `gnnmp_arena_*` (include/gnnmp.h) builds three (or two) address ranges
out of 2 GiB physical chunks of DIFFERENT classes (each chunk classified by a 0.6 ms probe), and a layer that opts in writes its output to
a range whose class differs from the class of the matrix it gathers from — and, with three ranges, from the class the NEXT kernel will
gather from (the output's dirty lines are still being written back while that kernel runs: 4.76 vs 4.65 ms on the GCN layer in the step):
    GCNConv:  out in a class other than x's           GATConv:  Wx = dense_x(x) not in x's class, the attention output in the third
The returned tensor is a PERSISTENT buffer of the layer, overwritten by its next call (like the static outputs of a captured graph): that
is the opt-in.  Results are bit-identical wherever buffers lie.

    gcn = gnnmp.GCNConv(...); gcn.place_outputs = True        # per layer (the returned tensor aliases the layer's next result!)
    sage = gnnmp.SAGEConv(...); sage.place_outputs = True     # the internal aggregate only; + sage.persistent_out = True: the output too
    gnnmp.placement.enable(True)                              # or for every GCNConv / GATConv call
    GNNMP_ARENA_GIB=8                                         # bytes per class (default 4 GiB), rounded up to 2 GiB chunks
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib as L

_ENABLED = [False]
MIN_BYTES = 256 << 20          # smaller outputs: the effect is not worth a persistent buffer


def enable(on=True):
    """Opt every GCNConv / GATConv / SAGEConv call of the process into placed buffers.

    ALIASING — read before enabling: a placed layer RETURNS a persistent arena buffer of that layer and shape, overwritten by the layer's
    next call (GCNConv / GATConv: the output IS what placement places).  A layer applied twice in one model, or outputs collected across
    iterations (`outs.append(layer(g, x))` in an eval loop), alias and are silently overwritten — `.clone()` what must survive the next
    call, or do not enable placement for that layer.  SAGEConv keeps only its INTERNAL aggregate persistent under this switch; its returned
    tensor is an ordinary allocation unless the layer sets `persistent_out = True`."""
    _ENABLED[0] = bool(on)


def enabled(layer=None):
    return _ENABLED[0] or bool(getattr(layer, "place_outputs", False))


def worth_it(shape):
    n = 4
    for d in shape:
        n *= int(d)
    return n >= MIN_BYTES


class _Raw:
    """a device address as torch sees it (torch.as_tensor over __cuda_array_interface__); keeps the arena alive"""

    def __init__(self, ptr, shape, owner):
        self.__cuda_array_interface__ = {"shape": tuple(int(d) for d in shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
        self._owner = owner


class Arena:
    """gnnmp_arena_t: ranges of device memory in different placement classes"""

    def __init__(self, gib_per_class=None, n_classes=3, max_probe_gib=None):
        L.require_gpu()
        if gib_per_class is None:
            gib_per_class = float(os.environ.get("GNNMP_ARENA_GIB", "4"))
        if max_probe_gib is None:          # 0: the library's default budget (32 GiB held while probing, ~0.3 s)
            max_probe_gib = float(os.environ.get("GNNMP_ARENA_PROBE_GIB", "0"))
        self._lib = L.load()
        self.handle = ctypes.c_void_p()
        L.check(self._lib.gnnmp_arena_create(ctypes.byref(self.handle), int(gib_per_class * (1 << 30)), int(n_classes),
                                             int(max_probe_gib * (1 << 30)), L.stream_ptr()))
        self._classes = {}
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_classes = self.info()["ranges"]      # the classes the device showed within the budget: 0 .. n_classes

    def info(self):
        v = (ctypes.c_int64 * 14)()
        L.check(self._lib.gnnmp_arena_info(self.handle, v))
        return {"bytes_per_class": v[0], "ranges": v[7], "used": (v[1], v[2], v[8])[: v[7]], "chunks_created": v[3], "chunks_released": v[4],
                "probe_us_same_class": v[5], "probe_us_two_classes": v[6], "gave_up_on_budget": bool(v[9]),
                "blocks_per_range": (v[10], v[11], v[12])[: v[7]], "create_ms": v[13] / 1e3}

    def alloc(self, shape, cls):
        """a float32 tensor of `shape` in range `cls` (0 .. n_classes - 1); None when the range is full"""
        n = 4
        for d in shape:
            n *= int(d)
        p = ctypes.c_void_p()
        rc = self._lib.gnnmp_arena_alloc(self.handle, int(cls), n, ctypes.byref(p))
        if rc == L.EALLOC:
            return None
        L.check(rc)
        return torch.as_tensor(_Raw(p.value, shape, self), device=self.device)

    def class_of(self, t):
        """c < n_classes: `t` shares the class of range c; n_classes: none of them / mixed / too small to tell.  Foreign memory is probed
        once per buffer."""
        key = (t.data_ptr(), t.numel() * t.element_size())
        c = self._classes.get(key)
        if c is None:
            out = ctypes.c_int(self.n_classes)
            L.check(self._lib.gnnmp_arena_class_of(self.handle, L.ptr(t), key[1], ctypes.byref(out), L.stream_ptr()))
            c = self._classes[key] = out.value
            if len(self._classes) > 256:
                self._classes.pop(next(iter(self._classes)))
        return c

    def reset(self):
        L.check(self._lib.gnnmp_arena_reset(self.handle))

    def __del__(self):
        try:
            if self.handle:
                self._lib.gnnmp_arena_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


_arena = [None, False]      # the arena, "creation was tried and failed"


def arena():
    """the process-wide arena, created at first use inside the library's probing budget (32 GiB held, ~0.3 s: gnnmp.h); None if the device
    did not show two placement classes within it (the layers then allocate as usual)"""
    if _arena[0] is None and not _arena[1]:
        try:
            a = Arena(n_classes=3)
            if a.n_classes >= 2:
                _arena[0] = a
            else:
                _arena[1] = True
                _arena.append(a.info())           # (kept for reports: why there is no placement)
                del a
        except L.GnnmpError as e:
            _arena[1] = True
            import warnings
            warnings.warn(f"gnnmp.placement: no arena ({e}); outputs are allocated as usual")
    return _arena[0]


def trials(on=True):
    """Choice's self-timing (every candidate buffer used three times, the fastest kept) on or off.  Off: a layer takes its first candidate —
    the arena's own classification — and never times itself (no events, no round-robin over buffers in its first calls)."""
    Choice.TRIALS = 2 if on else 0


class Choice:
    """A layer's persistent output buffer, chosen among candidates in different arena classes by TIMING the layer's own kernel on each
    (HIP events around the launch, polled on later calls: no synchronisation).  The arena's probe classifies a block at two places;
    a buffer of a gigabyte can still straddle physical blocks of different classes, and what counts in the end is the time of the
    real kernel on the real buffers: every candidate is used three times (the first is a warm-up), then the fastest stays."""
    TRIALS = 2

    def __init__(self, bufs):
        self.bufs = bufs
        self.spans = [[] for _ in bufs]
        self.rr = 0
        self.times_ms = None
        if len(bufs) == 1 or self.TRIALS <= 0:      # nothing to choose between / self-timing switched off (placement.trials(False))
            self.bufs = bufs[:1]
            self.spans = None

    def _resolve(self):
        if any(len(s) < self.TRIALS + 1 for s in self.spans):
            return
        if not all(e1.query() for s in self.spans for _, e1 in s):
            return
        self.times_ms = [sorted(a.elapsed_time(b) for a, b in s[1:])[len(s[1:]) // 2] for s in self.spans]
        tmin = min(self.times_ms)
        best = next(i for i, t in enumerate(self.times_ms) if t <= 1.01 * tmin)      # the first (most preferred) within 1 % of the best
        self.bufs = [self.bufs[best]]          # (the others stay allocated in the arena: it is a bump allocator)
        self.spans = None

    def begin(self):
        """-> (buffer, token); call end(token) right after the kernel launch"""
        if self.spans is not None:
            self._resolve()
        if self.spans is None:
            return self.bufs[0], None
        i = self.rr % len(self.bufs)
        self.rr += 1
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return self.bufs[i], (i, e0)

    def end(self, tok):
        if tok is None or self.spans is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.spans[tok[0]].append((tok[1], e1))

    @property
    def settled(self):
        return self.spans is None


def choice_for(layer, tag, shape, avoid, prefer_not=()):
    """the layer's Choice for buffer `tag` of `shape`: one candidate per arena class that is in none of `avoid` (the class of the matrix
    the kernel gathers from); classes in `prefer_not` (the class of the matrix the NEXT kernel gathers from: this output's dirty lines
    are written back while that kernel runs) come last and are taken only if they are more than 1 % faster.  None: allocate as usual."""
    a = arena()
    if a is None:
        return None
    cache = layer.__dict__.setdefault("_placed", {})
    avoid = list(avoid)
    key = (tag, tuple(shape), tuple(avoid))
    ch = cache.get(key)
    if ch is False:                 # no class could serve this buffer before: not asked again on every call
        return None
    if ch is None:
        free = [c for c in range(a.n_classes) if c not in avoid and c not in prefer_not]
        free += [c for c in range(a.n_classes) if c not in avoid and c in prefer_not]
        if not free:
            free = [c for c in range(a.n_classes) if c != avoid[0]] or [0]
        bufs = [b for b in (a.alloc(shape, c) for c in free) if b is not None]
        if not bufs:
            cache[key] = False
            return None
        ch = cache[key] = Choice(bufs)
    return ch


def buffer_for(layer, tag, shape, avoid):
    """a single persistent buffer (no trial): (buffer | None, its class)"""
    a = arena()
    if a is None:
        return None, 0
    cache = layer.__dict__.setdefault("_placed", {})
    avoid = list(avoid)
    free = [c for c in range(a.n_classes) if c not in avoid] or [c for c in range(a.n_classes) if c != avoid[0]] or [0]
    for cls in free:
        key = (tag, tuple(shape), cls)
        if key in cache:
            buf = cache[key]
            if buf is None:         # this class could not serve the buffer before: not asked again (a buffer larger than a block costs the
                continue            # arena three probed allocations to refuse — once, not on every call of the layer)
            return buf, cls
        buf = cache[key] = a.alloc(shape, cls)
        if buf is not None:
            return buf, cls
    return None, a.n_classes
