"""ctypes binding of libgnnmp.so (include/gnnmp.h) — the same symbols the Julia extension `@ccall`s.

torch is used only as plumbing: it owns device memory and the HIP stream.  Every compute call goes through the
C ABI; there is NO CPU fallback — if the library is missing or no GPU is visible, calls raise.
"""
from __future__ import annotations

import ctypes
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("GNNMP_LIB") or os.path.join(_PKG, "lib", "libgnnmp.so")   # GNNMP_LIB: another build of the same ABI

OK, EINVAL, EBOUNDS, EALLOC, ELAUNCH, EUNSUPPORTED = 0, -1, -2, -3, -4, -5
SUM, MEAN, MAX, MIN = 0, 1, 2, 3
COPY_XJ, W_MUL_XJ = 0, 1
ACT_IDENTITY, ACT_RELU, ACT_SOFTPLUS, ACT_TANH, ACT_SWISH = 0, 1, 2, 3, 4
LONG_ROW = 512

# every symbol include/gnnmp.h declares (tests check the library exports exactly these)
SYMBOLS = (
    "gnnmp_version", "gnnmp_last_error",
    "gnnmp_plan_create", "gnnmp_plan_from_csc", "gnnmp_plan_destroy", "gnnmp_plan_info", "gnnmp_plan_export", "gnnmp_plan_export64",
    "gnnmp_plan_concat", "gnnmp_plan_select", "gnnmp_plan_release", "gnnmp_plan_status", "gnnmp_plan_edge_index",
    "gnnmp_chain_jobs_pack", "gnnmp_chain_jobs_release", "gnnmp_chain_jobs_export",
    "gnnmp_arena_create", "gnnmp_arena_destroy", "gnnmp_arena_alloc", "gnnmp_arena_reset", "gnnmp_arena_class_of", "gnnmp_arena_info",
    "gnnmp_add_self_loops", "gnnmp_batch_coo",
    "gnnmp_sort_edge_index", "gnnmp_is_bidirected", "gnnmp_has_self_loops", "gnnmp_sample_neighbors",
    "gnnmp_unique_append", "gnnmp_induced_subgraph",
    "gnnmp_gather_f32", "gnnmp_edge_sub_f32", "gnnmp_scatter_f32", "gnnmp_scatter_atomic_f32",
    "gnnmp_propagate_f32", "gnnmp_propagate_emul_f32", "gnnmp_propagate_gated_f32", "gnnmp_propagate_slots_f32", "gnnmp_propagate_slots_act_f32", "gnnmp_plan_slot_gather_f32",
    "gnnmp_degree_f32", "gnnmp_inv_sqrt_f32",
    "gnnmp_edge_softmax_f32", "gnnmp_segment_softmax_f32", "gnnmp_gat_node_scores_f32", "gnnmp_gat_aggregate_f32", "gnnmp_gat_conv_f32",
    "gnnmp_gat_conv_edge_f32", "gnnmp_gat_conv_stats_f32", "gnnmp_gat_conv_grad_f32", "gnnmp_attn_conv_f32",
    "gnnmp_gat_conv_train_f32", "gnnmp_gat_conv_grad2_f32",
    "gnnmp_gat_conv_drop_f32", "gnnmp_dropout_keep_u8", "gnnmp_gat_conv_grad_drop_f32", "gnnmp_attn_conv_drop_f32", "gnnmp_attn_conv_grad_drop_f32",
    "gnnmp_attn_conv_grad_f32",
    "gnnmp_bias_act_f32",
    "gnnmp_segment_pool_f32", "gnnmp_segment_bounds", "gnnmp_segment_pool_ptr_f32", "gnnmp_dense_f32", "gnnmp_fused_conv_f32",
    "gnnmp_graphconv_chain_scratch_floats", "gnnmp_graphconv_chain_f32", "gnnmp_chain_jobs_create", "gnnmp_chain_jobs_destroy",
    "gnnmp_chain_jobs_info", "gnnmp_shard_by_size", "gnnmp_allgather_f32",
    "gnnmp_edge_dot_f32", "gnnmp_edge_dot_plan_f32", "gnnmp_propagate_maxmin_grad_f32",
    "gnnmp_head_mean_f32", "gnnmp_head_mean_grad_f32", "gnnmp_add_f32", "gnnmp_axpy_f32", "gnnmp_mul_rows_f32", "gnnmp_is_sorted",
    "gnnmp_act_grad_f32", "gnnmp_dense_grad_workspace", "gnnmp_dense_grad_w_f32",
    "gnnmp_dense_grad_w2_workspace", "gnnmp_dense_grad_w2_f32", "gnnmp_pool_grad_act_f32", "gnnmp_propagate_add_mask_f32",
    "gnnmp_row_normalize_f32", "gnnmp_row_normalize_grad_f32", "gnnmp_propagate_cg_f32", "gnnmp_gru_pointwise_f32", "gnnmp_propagate_nn_f32", "gnnmp_gmm_weights_f32", "gnnmp_row_sqnorm_normalize_f32", "gnnmp_lstm_pointwise_f32", "gnnmp_rowdot_f32",
    "gnnmp_plan_reset_counters",
    "gnnmp_propagate_f64", "gnnmp_gather_f64", "gnnmp_scatter_f64",
    # the GNNMP_INTERNAL section of the header: experiment / test hooks, exported but not part of the drop-in surface
    "gnnmp_tune", "gnnmp_debug_mock_device", "gnnmp_debug_device_once", "gnnmp_debug_plan_block", "gnnmp_debug_pool_pick",
)


class GnnmpError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"libgnnmp status {status}: {msg}")
        self.status = status


_lib = None


def load():
    """Load libgnnmp.so; raises (loudly) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  gnnmp has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i, i64, f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    u64 = ctypes.c_uint64
    L.gnnmp_version.restype = i
    L.gnnmp_last_error.restype = ctypes.c_char_p
    sig = {
        "gnnmp_plan_create": [ctypes.POINTER(vp), vp, vp, i, i, i64, i64, i64, i, i, vp],
        "gnnmp_plan_from_csc": [ctypes.POINTER(vp), vp, vp, i, i, i64, i64, i64, i, vp],
        "gnnmp_plan_destroy": [vp],
        "gnnmp_plan_info": [vp, ctypes.POINTER(i64)],
        "gnnmp_plan_export": [vp, vp, vp, vp, vp],
        "gnnmp_plan_export64": [vp, vp, vp, vp, vp],
        "gnnmp_plan_concat": [ctypes.POINTER(vp), ctypes.POINTER(vp), i64, vp, vp, i, i, vp],
        "gnnmp_plan_select": [ctypes.POINTER(vp), vp, vp, i64, vp, i, i, i64, i64, i64, vp, vp, vp, vp],
        "gnnmp_plan_release": [vp, vp],
        "gnnmp_plan_status": [vp, vp],
        "gnnmp_plan_edge_index": [vp, i, i, vp, vp, vp],
        "gnnmp_chain_jobs_pack": [ctypes.POINTER(vp), vp, i64, i64, i64, i, vp],
        "gnnmp_chain_jobs_release": [vp, vp],
        "gnnmp_chain_jobs_export": [vp, vp, i64, vp, vp],
        "gnnmp_arena_create": [ctypes.POINTER(vp), i64, i, i64, vp],
        "gnnmp_arena_destroy": [vp],
        "gnnmp_arena_alloc": [vp, i, i64, ctypes.POINTER(vp)],
        "gnnmp_arena_reset": [vp],
        "gnnmp_arena_class_of": [vp, vp, i64, ctypes.POINTER(i), vp],
        "gnnmp_arena_info": [vp, ctypes.POINTER(i64)],
        "gnnmp_add_self_loops": [vp, vp, i, i, i64, i64, vp, vp, vp, vp, vp],
        "gnnmp_batch_coo": [vp, vp, i, i, vp, vp, i64, vp, vp, vp, vp],
        "gnnmp_sort_edge_index": [vp, vp, i, i, i64, vp, vp, vp],
        "gnnmp_is_bidirected": [vp, vp, i, i, i64, ctypes.POINTER(i), vp],
        "gnnmp_has_self_loops": [vp, vp, i, i64, ctypes.POINTER(i), vp],
        "gnnmp_sample_neighbors": [vp, vp, i, i, i64, i64, i, ctypes.c_uint64, vp, vp, i64, ctypes.POINTER(i64), vp],
        "gnnmp_unique_append": [vp, vp, i64, vp, i, i, i64, i64, vp, ctypes.POINTER(i64), vp],
        "gnnmp_induced_subgraph": [vp, vp, vp, i, i, i64, vp, vp, vp, vp, i64, ctypes.POINTER(i64), vp],
        "gnnmp_gather_f32": [vp, vp, i, i, i64, vp, i64, vp],
        "gnnmp_edge_sub_f32": [vp, vp, vp, vp, i, i, i64, i, vp, i64, vp],
        "gnnmp_scatter_f32": [vp, i, vp, vp, i64, vp],
        "gnnmp_scatter_atomic_f32": [i, vp, vp, i, i, i64, vp, i64, vp],
        "gnnmp_propagate_f32": [vp, i, i, vp, vp, vp, vp, vp, i64, vp],
        "gnnmp_propagate_emul_f32": [vp, i, vp, vp, vp, i64, vp],
        "gnnmp_propagate_gated_f32": [vp, i, vp, vp, vp, i64, vp],
        "gnnmp_propagate_slots_f32": [vp, i, vp, vp, vp, vp, vp, i64, vp],
        "gnnmp_propagate_slots_act_f32": [vp, i, vp, vp, vp, vp, vp, i, vp, i64, vp],
        "gnnmp_plan_slot_gather_f32": [vp, i, vp, vp, vp],
        "gnnmp_gat_conv_f32": [vp, vp, vp, vp, f, vp, i, vp, i64, i64, vp],
        "gnnmp_gat_conv_edge_f32": [vp, vp, vp, vp, vp, f, vp, i, vp, i64, i64, vp],
        "gnnmp_gat_conv_stats_f32": [vp, vp, vp, vp, f, vp, i, vp, vp, i64, i64, vp],
        "gnnmp_gat_conv_train_f32": [vp, vp, vp, vp, f, vp, i, vp, vp, vp, vp, i64, i64, vp],
        "gnnmp_gat_conv_grad2_f32": [vp, vp, vp, vp, vp, f, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, vp],
        "gnnmp_attn_conv_f32": [vp, i, vp, vp, vp, vp, f, f, vp, i, vp, vp, i64, i64, vp],
        "gnnmp_attn_conv_grad_f32": [vp, vp, i, vp, vp, vp, vp, f, f, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, vp],
        "gnnmp_gat_conv_grad_f32": [vp, vp, vp, vp, vp, f, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, vp],
        "gnnmp_gat_conv_drop_f32": [vp, vp, vp, vp, f, f, u64, vp, i, vp, vp, i64, i64, vp],
        "gnnmp_dropout_keep_u8": [u64, f, i64, i64, vp, vp],
        "gnnmp_attn_conv_drop_f32": [vp, i, vp, vp, vp, vp, f, f, f, u64, vp, i, vp, vp, i64, i64, vp],
        "gnnmp_attn_conv_grad_drop_f32": [vp, vp, i, vp, vp, vp, vp, f, f, f, u64, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, vp],
        "gnnmp_gat_conv_grad_drop_f32": [vp, vp, vp, vp, vp, f, f, u64, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, vp],
        "gnnmp_degree_f32": [vp, vp, vp, vp],
        "gnnmp_inv_sqrt_f32": [vp, vp, i64, vp],
        "gnnmp_edge_softmax_f32": [vp, vp, vp, i64, vp],
        "gnnmp_segment_softmax_f32": [vp, vp, vp, i64, f, vp],
        "gnnmp_gat_node_scores_f32": [vp, vp, vp, vp, i64, i64, i64, vp],
        "gnnmp_gat_aggregate_f32": [vp, vp, vp, vp, f, vp, i, vp, vp, i64, i64, vp],
        "gnnmp_bias_act_f32": [vp, vp, i, vp, i64, i64, vp],
        "gnnmp_segment_pool_f32": [i, vp, vp, i, i, vp, i64, i64, i64, vp],
        "gnnmp_segment_bounds": [vp, i, i, i64, i64, vp, vp],
        "gnnmp_segment_pool_ptr_f32": [i, vp, vp, vp, i64, i64, i64, vp],
        "gnnmp_dense_f32": [vp, vp, i64, i64, vp, vp, i64, i64, i, vp, i, vp, i64, i64, vp],
        "gnnmp_fused_conv_f32": [vp, i, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, i64, i, vp, i, vp, i64, vp, vp],
        "gnnmp_edge_dot_f32": [vp, vp, vp, vp, i, i, i64, i64, vp, vp],
        "gnnmp_edge_dot_plan_f32": [vp, vp, vp, vp, i64, vp],
        "gnnmp_head_mean_f32": [vp, vp, i, vp, i64, i64, i64, vp],
        "gnnmp_head_mean_grad_f32": [vp, vp, i64, i64, i64, vp],
        "gnnmp_row_normalize_f32": [vp, vp, vp, i64, i64, vp],
        "gnnmp_propagate_cg_f32": [vp, vp, vp, vp, i, vp, i64, vp],
        "gnnmp_propagate_nn_f32": [vp, i, vp, vp, vp, i64, i64, vp],
        "gnnmp_gmm_weights_f32": [vp, vp, vp, vp, i64, i64, i64, i64, vp],
        "gnnmp_row_sqnorm_normalize_f32": [vp, vp, vp, f, i64, i64, vp],
        "gnnmp_lstm_pointwise_f32": [vp, vp, vp, vp, vp, vp, i64, i64, vp],
        "gnnmp_rowdot_f32": [vp, vp, vp, i64, i64, vp],
        "gnnmp_gru_pointwise_f32": [vp, vp, vp, vp, vp, i64, i64, vp],
        "gnnmp_row_normalize_grad_f32": [vp, vp, vp, vp, vp, vp, vp, f, i64, i64, vp],
        "gnnmp_add_f32": [vp, vp, vp, i64, vp],
        "gnnmp_axpy_f32": [f, vp, vp, vp, i64, vp],
        "gnnmp_mul_rows_f32": [vp, i64, vp, vp, i64, i64, vp],
        "gnnmp_is_sorted": [vp, i, i64, ctypes.POINTER(i), vp],
        "gnnmp_propagate_maxmin_grad_f32": [vp, vp, vp, vp, vp, i64, vp],
        "gnnmp_act_grad_f32": [vp, vp, i, vp, i64, vp],
        "gnnmp_dense_grad_w_f32": [vp, vp, i64, i64, i64, vp, vp, vp, i64, vp],
        "gnnmp_dense_grad_w2_f32": [vp, vp, i64, vp, i64, i64, i64, vp, vp, i64, vp],
        "gnnmp_pool_grad_act_f32": [vp, vp, i, i, vp, vp, i, vp, i64, i64, i64, vp],
        "gnnmp_propagate_add_mask_f32": [vp, i, vp, vp, vp, vp, vp, i64, vp],
        "gnnmp_shard_by_size": [ctypes.POINTER(i64), i64, i, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(i64), ctypes.POINTER(i64)],
        "gnnmp_allgather_f32": [vp, vp, vp, i64, vp],
        "gnnmp_chain_jobs_create": [ctypes.POINTER(vp), vp, i64, vp],
        "gnnmp_chain_jobs_destroy": [vp],
        "gnnmp_chain_jobs_info": [vp, ctypes.POINTER(i64)],
        "gnnmp_graphconv_chain_f32": [vp, vp, vp, i64, vp, i, ctypes.POINTER(i64), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                      ctypes.POINTER(vp), ctypes.POINTER(i), i, i, i, vp, vp, i64, vp, vp, vp],
        "gnnmp_tune": [i, i],
        "gnnmp_plan_reset_counters": [vp, vp],
        "gnnmp_propagate_f64": [vp, i, i, vp, vp, vp, vp, vp, i64, vp],
        "gnnmp_gather_f64": [vp, vp, i, i, i64, vp, i64, vp],
        "gnnmp_scatter_f64": [vp, i, vp, vp, i64, vp],
    }
    for name, args in sig.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            # an OLDER build named by GNNMP_LIB for a same-box A/B run may predate an entry point; the library of this tree must
            # export every one of them (tests/test_abi.py checks that)
            if os.environ.get("GNNMP_LIB"):
                continue
            raise
        fn.argtypes = args
        fn.restype = i
    L.gnnmp_dense_grad_workspace.argtypes = [i64, i64, i64]
    L.gnnmp_dense_grad_workspace.restype = i64
    try:
        L.gnnmp_dense_grad_w2_workspace.argtypes = [i64, i64, i64, i64]
        L.gnnmp_dense_grad_w2_workspace.restype = i64
    except AttributeError:
        if not os.environ.get("GNNMP_LIB"):
            raise
    try:
        L.gnnmp_graphconv_chain_scratch_floats.argtypes = [i64, i, ctypes.POINTER(i64), i64]
        L.gnnmp_graphconv_chain_scratch_floats.restype = i64
    except AttributeError:
        if not os.environ.get("GNNMP_LIB"):
            raise
    # GNNMP_KNOBS="0=1,5=2": tuning knobs applied at load (test runs that force the narrow-vector / other template variants of
    # every kernel: `GNNMP_KNOBS=0=1 python -m pytest tests -m gpu`)
    _lib = L
    for kv in filter(None, os.environ.get("GNNMP_KNOBS", "").split(",")):
        k, v = kv.split("=")
        tune(int(k), int(v))
    return L


def check(rc):
    if rc != OK:
        raise GnnmpError(rc, load().gnnmp_last_error().decode())


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("gnnmp needs an AMD GPU (gfx950); there is no CPU fallback on the product path")


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device pointer of a tensor (None -> NULL)"""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


# ---- kernel probe: HIP events around named launches, on the stream they are launched on -------------------------------------------------
# bench.py times the dominant kernels INSIDE its timed steps with this (the roofline figure of the bench line is the in-step duration, the
# one a rocprofv3 kernel trace of the same command shows); off (None) it costs one global read per call.
_probe = None


class EventProbe:
    """spans[name] = list of (start, end) torch.cuda.Event pairs recorded around every launch of `name` while the probe is set"""

    def __init__(self):
        self.spans = {}

    def begin(self):
        import torch
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, name, e0):
        import torch
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.spans.setdefault(name, []).append((e0, e1))

    def times_ms(self, name):
        """call after a synchronisation"""
        return [a.elapsed_time(b) for a, b in self.spans.get(name, [])]


def set_probe(p):
    global _probe
    _probe = p


KNOB_DEFAULTS = {1: -1, 3: 1, 7: 17}      # every other knob starts at 0 (csrc/plan.cpp g_knobs)
_knobs = {}


def knob(k: int) -> int:
    """the value last set through tune() (or the library's default)"""
    return _knobs.get(k, KNOB_DEFAULTS.get(k, 0))


def tune(knob: int, value: int):
    """perf-experiment hook (csrc/common.h Knob); not part of the drop-in surface"""
    check(load().gnnmp_tune(knob, value))
    _knobs[knob] = int(value)
    if knob == 14:                       # the host mirror of fused_conv's gating follows the knob (gnnmp/layers.py)
        from . import layers
        layers._KNOB14[0] = int(value)
