"""One process, several devices (a Julia host switching AMDGPU.device!): everything the library remembers about "the
device" is keyed on the calling thread's current device — kernel attributes (hipFuncSetAttribute acts on the current device's function
object only), the compute-unit count, the block pool, graph prep's scratch cache.  No GPU here: the device index is mocked through the
library's test hook and the machinery runs with a counting stand-in for the attribute call (csrc/common.h DeviceOnce, csrc/pool.hip)."""
import ctypes

import pytest


@pytest.fixture(scope="module")
def lib():
    from gnnmp import _lib
    L = _lib.load()
    L.gnnmp_debug_mock_device.argtypes = [ctypes.c_int]
    L.gnnmp_debug_device_once.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.gnnmp_debug_pool_pick.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_uint64]
    yield L
    L.gnnmp_debug_mock_device(-1)


def _once(lib, devs, fail_on=-1):
    arr = (ctypes.c_int * len(devs))(*devs)
    failed = ctypes.c_int(0)
    ran = lib.gnnmp_debug_device_once(arr, len(devs), fail_on, ctypes.byref(failed))
    return ran, failed.value


def test_kernel_attribute_opt_in_runs_once_per_device(lib):
    assert _once(lib, [0, 0, 0]) == (1, 0)
    assert _once(lib, [0, 1, 0, 3, 1, 7, 7, 0]) == (4, 0)                 # devices 0, 1, 3, 7: four opt-ins, not one
    assert _once(lib, list(range(8)) * 3) == (8, 0)                       # the 8-GPU node, three rounds of calls


def test_a_failed_opt_in_is_reported_on_every_call_of_that_device_only(lib):
    ran, failed = _once(lib, [2, 5, 2, 5, 5], fail_on=5)
    assert ran == 2 and failed == 3


def test_mock_device_is_clamped_and_thread_local(lib):
    import threading
    assert lib.gnnmp_debug_mock_device(3) == 3
    assert lib.gnnmp_debug_mock_device(1000) == 31                        # GNNMP_MAX_DEVICES - 1
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.gnnmp_debug_mock_device(-1)))
    t.start(); t.join()
    assert seen == [0]                                                    # another thread: no GPU here -> device 0, not 31
    assert lib.gnnmp_debug_mock_device(-1) == 0


def test_pool_slot_choice(lib):
    def pick(caps, nbytes):
        arr = (ctypes.c_uint64 * len(caps))(*caps)
        return lib.gnnmp_debug_pool_pick(arr, len(caps), nbytes)
    mb = 1 << 20
    assert pick([0] * 8, mb) == -1
    assert pick([0, 8 * mb, 4 * mb, 0], 3 * mb) == 2                      # the smallest block that fits
    assert pick([0, 64 * mb, 0, 0], 3 * mb) == -1                         # > 2x the request + 1 MiB: not worth pinning
    assert pick([0, 7 * mb, 0, 0], 3 * mb) == 1
    assert pick([2 * mb, 2 * mb], 3 * mb) == -1


def test_sources_keep_no_process_wide_once_for_device_state():
    """the round-4 bug: std::call_once around hipFuncSetAttribute sets the attribute on ONE device for the whole process"""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "graphneuralnetworks.jl_amd", "csrc")
    for f in sorted(os.listdir(root)):
        if not f.endswith((".hip", ".h")):
            continue
        src = open(os.path.join(root, f)).read()
        for m in re.finditer(r"call_once", src):
            window = src[m.start(): m.start() + 600]
            assert "hipFuncSetAttribute" not in window and "hipDeviceGetAttribute" not in window, f"{f}: process-wide once around device state"
