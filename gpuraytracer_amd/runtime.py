"""Thin object wrappers over the gpurt C-ABI handles (device, buffer, event, accel).

Everything here forwards to libgpurt.so; numpy is used only to stage host data.
"""
import ctypes as C

import numpy as np

from . import capi


class Device:
    def __init__(self, gpu_index=0):
        self.lib = capi.load()
        h = C.c_void_p()
        capi.check(self.lib.rt_device_create(int(gpu_index), C.byref(h)), "rt_device_create")
        self.handle = h
        self.gpu_index = gpu_index

    def sync(self):
        capi.check(self.lib.rt_device_sync(self.handle), "rt_device_sync")

    def stream(self):
        return self.lib.rt_device_stream(self.handle)

    def set_stream(self, hip_stream):
        capi.check(self.lib.rt_device_set_stream(self.handle, C.c_void_p(hip_stream)), "rt_device_set_stream")

    def alloc(self, nbytes):
        return Buffer(self, nbytes=nbytes)

    def wrap(self, device_ptr, nbytes, keepalive=None):
        return Buffer(self, nbytes=nbytes, device_ptr=device_ptr, keepalive=keepalive)

    def upload(self, array):
        a = np.ascontiguousarray(array)
        b = Buffer(self, nbytes=max(a.nbytes, 16))
        b.upload(a)
        return b

    def event(self):
        return Event(self)

    def close(self):
        if self.handle:
            self.lib.rt_device_destroy(self.handle)
            self.handle = None


class Buffer:
    def __init__(self, dev, nbytes, device_ptr=None, keepalive=None):
        self.dev = dev
        self.lib = dev.lib
        self.nbytes = int(nbytes)
        self.keepalive = keepalive
        h = C.c_void_p()
        if device_ptr is None:
            capi.check(self.lib.rt_buffer_alloc(dev.handle, self.nbytes, C.byref(h)), "rt_buffer_alloc")
        else:
            capi.check(self.lib.rt_buffer_wrap(dev.handle, C.c_void_p(device_ptr), self.nbytes, C.byref(h)), "rt_buffer_wrap")
        self.handle = h

    def upload(self, array, offset=0):
        a = np.ascontiguousarray(array)
        capi.check(self.lib.rt_buffer_upload(self.handle, a.ctypes.data_as(C.c_void_p), a.nbytes, offset), "rt_buffer_upload")

    def download(self, dtype, count=None, offset=0):
        dtype = np.dtype(dtype)
        if count is None:
            count = (self.nbytes - offset) // dtype.itemsize
        out = np.empty(count, dtype=dtype)
        capi.check(self.lib.rt_buffer_readback(self.handle, out.ctypes.data_as(C.c_void_p), out.nbytes, offset),
                   "rt_buffer_readback")
        return out

    def clear(self):
        capi.check(self.lib.rt_buffer_clear(self.handle), "rt_buffer_clear")

    def ptr(self):
        return self.lib.rt_buffer_ptr(self.handle)

    def free(self):
        if self.handle:
            self.lib.rt_buffer_free(self.handle)
            self.handle = None


class Event:
    def __init__(self, dev):
        self.lib = dev.lib
        h = C.c_void_p()
        capi.check(self.lib.rt_event_create(dev.handle, C.byref(h)), "rt_event_create")
        self.handle = h

    def record(self):
        capi.check(self.lib.rt_event_record(self.handle), "rt_event_record")

    def elapsed_ms(self, stop):
        ms = C.c_float()
        capi.check(self.lib.rt_event_elapsed_ms(self.handle, stop.handle, C.byref(ms)), "rt_event_elapsed_ms")
        return ms.value


class Accel:
    def __init__(self, dev, max_primitives):
        self.dev = dev
        self.lib = dev.lib
        h = C.c_void_p()
        capi.check(self.lib.rt_accel_create(dev.handle, int(max_primitives), C.byref(h)), "rt_accel_create")
        self.handle = h

    def pack(self, num_primitives, indices, vertices):
        capi.check(self.lib.rt_accel_pack(self.handle, int(num_primitives), indices.handle, vertices.handle), "rt_accel_pack")

    def pack_bounds(self, num_primitives, indices, vertices, bounds):
        """pack() that also leaves the mesh's centroid bounds (8 floats) in the device buffer `bounds`."""
        capi.check(self.lib.rt_accel_pack_bounds(self.handle, int(num_primitives), indices.handle, vertices.handle,
                                                 bounds.handle), "rt_accel_pack_bounds")

    def build_device_bounds(self, bounds, bvh, morton=None):
        capi.check(self.lib.rt_accel_build_device_bounds(self.handle, bounds.handle, bvh.handle,
                                                         morton.handle if morton else None), "rt_accel_build_device_bounds")

    def refit(self, bvh):
        """Leaves and levels again over the triangles packed since, in the order of the last build (rt_accel_refit)."""
        capi.check(self.lib.rt_accel_refit(self.handle, bvh.handle), "rt_accel_refit")

    def build(self, scene_min, scene_max, bvh, morton=None):
        mn = (C.c_float * 3)(*[float(x) for x in scene_min])
        mx = (C.c_float * 3)(*[float(x) for x in scene_max])
        capi.check(self.lib.rt_accel_build(self.handle, mn, mx, bvh.handle, morton.handle if morton else None), "rt_accel_build")

    def destroy(self):
        if self.handle:
            self.lib.rt_accel_destroy(self.handle)
            self.handle = None
