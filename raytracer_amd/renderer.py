"""HIPShader — host-side mirror of the reference's render-backend interface
(``class ComputeShader : IFrameRender``, src/ComputeShader.h:7-18): constructed
from a filled scene, ``Render(pixels)`` renders one frame with the scene's
current camera and frame counter into the caller's float RGBA array.

Everything goes through the C-ABI in include/rtx.h (ctypes). The HIP library is
required: there is no CPU implementation behind this class.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import Camera, Config, Counters, Info, SceneDesc


class HIPShader:
    def __init__(self, scene, device=0, rank=0, world_size=1, tile=(64, 32), traversal=_capi.TRAVERSE_WIDE,
                 count_stats=False, frames_in_flight=0, devices=None, look_ahead=0, frames_per_batch=0):
        """`scene` is a raytracer_amd.scene.HostScene (or anything with .desc() and
        .camera); its arrays are read once, here (as src/CLShader.cpp:20-49 does)."""
        self._L = _capi.hip_lib()
        self.scene = scene
        d = scene.desc()
        self.width, self.height = d.width, d.height
        self.frame_counter = 0                     # context->frameCounter
        cfg = Config(device=device, rank=rank, worldSize=world_size, tileW=tile[0], tileH=tile[1],
                     traversal=traversal, countStats=1 if count_stats else 0, framesInFlight=frames_in_flight,
                     lookAhead=look_ahead, framesPerBatch=frames_per_batch)
        if devices is not None:            # a device group in this process (RtxConfig.numDevices / devices)
            cfg.numDevices = len(devices)
            for i, dev in enumerate(devices):
                cfg.devices[i] = int(dev)
        h = C.c_void_p()
        rc = self._L.rtx_create(C.byref(d), C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RuntimeError("rtx_create: " + self._L.rtx_last_error().decode())
        self._h = h
        self.rank, self.world_size = rank, world_size

    def close(self):
        if getattr(self, "_h", None):
            self._L.rtx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(what + ": " + self._L.rtx_last_error().decode())

    # ---- the reference interface ------------------------------------------------
    def Render(self, pixels=None, camera=None):
        """IFrameRender::Render(Color* _pixels) (src/IFrameRender.h:8). The caller
        increments frame_counter afterwards, like src/WindowManager.cpp:156-157."""
        cam = camera if camera is not None else self.scene.camera
        ptr = None
        if pixels is not None:
            assert pixels.dtype == np.float32 and pixels.flags.c_contiguous and pixels.size == self.width * self.height * 4
            ptr = pixels.ctypes.data
        self._chk(self._L.rtx_render(self._h, C.byref(cam), self.frame_counter, ptr), "rtx_render")

    def render_frames(self, n, pixels=None, camera=None):
        """The headless -F loop (src/Main.cpp:48-51): n x {Render; frameCounter++}."""
        for i in range(n):
            self.Render(pixels if i == n - 1 else None, camera)
            self.frame_counter += 1

    # ---- probes -------------------------------------------------------------------
    def primary_hits(self, frame_counter=0, camera=None):
        cam = camera if camera is not None else self.scene.camera
        ids = np.empty((self.height, self.width), dtype=np.int32)
        self._chk(self._L.rtx_primary_hits(self._h, C.byref(cam), frame_counter, ids.ctypes.data), "rtx_primary_hits")
        return ids

    def trace_rays(self, origins, directions):
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(directions, dtype=np.float32)
        assert o.shape == d.shape and o.shape[1] == 4
        n = len(o)
        ids = np.empty(n, dtype=np.int32)
        pts = np.empty((n, 4), dtype=np.float32)
        nrm = np.empty((n, 4), dtype=np.float32)
        self._chk(self._L.rtx_trace_rays(self._h, o.ctypes.data, d.ctypes.data, n, ids.ctypes.data, pts.ctypes.data,
                                         nrm.ctypes.data), "rtx_trace_rays")
        return ids, pts, nrm

    def download(self):
        px = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._chk(self._L.rtx_download_pixels(self._h, px.ctypes.data), "rtx_download_pixels")
        return px

    def download_into(self, pixels):
        """rtx_download_pixels into a caller-owned array (no allocation on the way)."""
        assert pixels.dtype == np.float32 and pixels.flags.c_contiguous and pixels.size == self.width * self.height * 4
        self._chk(self._L.rtx_download_pixels(self._h, pixels.ctypes.data), "rtx_download_pixels")

    def present(self, gamma=0.0):
        """rtx_present_rgba8: the running mean as an (h, w, 4) uint8 image (clamp, optional gamma, round to 8 bits on the device)."""
        img = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self._chk(self._L.rtx_present_rgba8(self._h, C.c_float(gamma), img.ctypes.data), "rtx_present_rgba8")
        return img

    def upload(self, pixels):
        px = np.ascontiguousarray(pixels, dtype=np.float32)
        self._chk(self._L.rtx_upload_pixels(self._h, px.ctypes.data), "rtx_upload_pixels")

    def clear(self):
        self._chk(self._L.rtx_clear_pixels(self._h), "rtx_clear_pixels")

    def counters(self):
        c = Counters()
        self._chk(self._L.rtx_get_counters(self._h, C.byref(c)), "rtx_get_counters")
        return c

    def info(self):
        """What rtx_create decided: traversal in effect, culling slack, frames in flight, tree sizes (RtxInfo)."""
        i = Info()
        self._chk(self._L.rtx_get_info(self._h, C.byref(i)), "rtx_get_info")
        return i

    def reset_counters(self):
        self._chk(self._L.rtx_reset_counters(self._h), "rtx_reset_counters")

    # ---- multi-GPU plumbing -------------------------------------------------------
    @property
    def handle(self):
        return self._h

    def local_pixel_count(self):
        return self._L.rtx_local_pixel_count(self._h)

    def packed_floats(self, rank):
        return self._L.rtx_packed_floats(self._h, rank)

    def pack_tiles(self, device_ptr):
        self._chk(self._L.rtx_pack_tiles(self._h, C.c_void_p(device_ptr)), "rtx_pack_tiles")

    def unpack_tiles(self, device_ptr, src_rank):
        self._chk(self._L.rtx_unpack_tiles(self._h, C.c_void_p(device_ptr), src_rank), "rtx_unpack_tiles")
