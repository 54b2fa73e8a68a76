"""rt_scene_bounds_static: the entry point that allocates nothing (its scratch is a static device array).

Runs last (the file name sorts after the other tests): its output must equal rt_scene_bounds' bit for bit on two
calls in a row, the second with a smaller input that leaves most of the first call's partial bounds in the scratch.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bounds(device, fn, sc, n):
    from gpuraytracer_amd import capi
    idx, vtx, out = device.upload(sc["indices"]), device.upload(sc["vertices"]), device.alloc(32)
    out.upload(np.full(8, 7.0, dtype=np.float32))
    capi.check(fn(device.handle, n, idx.handle, vtx.handle, out.handle), "scene bounds")
    b = out.download(np.float32, 8)
    for x in (idx, vtx, out):
        x.free()
    return b


def test_static_scratch_entry_point_matches_the_allocating_one(device, oracle):
    from gpuraytracer_amd import capi, scenes
    lib = device.lib
    big, small = scenes.synthetic_scene(300001), scenes.synthetic_scene(700)
    # the smaller scene lies inside the larger one's bounds: stale partials would widen its result
    v = small["vertices"].copy()
    v["position"] *= np.float32(0.01)
    small["vertices"] = v
    for sc in (big, small, big):
        n = len(sc["indices"]) // 3
        got = _bounds(device, lib.rt_scene_bounds_static, sc, n)
        want = _bounds(device, lib.rt_scene_bounds, sc, n)
        assert got.tobytes() == want.tobytes(), "static-scratch bounds differ at %d triangles" % n
        mn, mx = oracle.scene_aabb(sc)
        assert got[:3].tobytes() == mn.tobytes() and got[4:7].tobytes() == mx.tobytes()
        assert got[3] == 0.0 and got[7] == 0.0
    # zero triangles: the host loop's +-FLT_MAX, whatever the scratch holds
    empty = dict(small)
    empty["indices"] = small["indices"][:0]
    got = _bounds(device, lib.rt_scene_bounds_static, empty, 0)
    assert got[0] == np.float32(3.402823466e+38) and got[4] == np.float32(-3.402823466e+38)
    # beyond the limit: refused before anything is launched
    idx, vtx, out = device.upload(small["indices"]), device.upload(small["vertices"]), device.alloc(32)
    rc = lib.rt_scene_bounds_static(device.handle, 0x55555556, idx.handle, vtx.handle, out.handle)
    assert rc == capi.RT_ERR_LIMIT_EXCEEDED and b"limit" in lib.rt_error_string(rc)
    assert lib.rt_scene_bounds_static(device.handle, 700, idx.handle, vtx.handle, None) == capi.RT_ERR_INVALID_ARGUMENT
    for x in (idx, vtx, out):
        x.free()
