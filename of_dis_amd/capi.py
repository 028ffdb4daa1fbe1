"""ctypes binding of the C ABI (include/ofdis.h) -- the only way Python reaches the product.

There is no CPU fallback: `lib()` raises if of_dis_amd/lib/libofdis_hip.so is missing, and every
call raises OfdisError on a non-zero status.  numpy helpers move data through the library's own
device-memory helpers so that the tests need nothing but the shared library; the benchmark passes
torch device pointers straight through.
"""
import ctypes as C
import os

import numpy as np

from .params import OfdisParams

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libofdis_hip.so")
LIB_PATH = os.environ.get("OFDIS_LIB", LIB_PATH)  # developer A/B builds (tools/ab_build.py); the product path is the default
_f32 = np.float32
FP = C.POINTER(C.c_float)
VP = C.c_void_p

K_WARP, K_DERIV, K_SYSTEM, K_SOR, K_PATCH, K_DENSIFY, K_UPDATE, K_FUSED, K_COUNT = range(9)
K_NAMES = ["warp", "derivatives", "tv_system", "sor", "patch_optimize", "densify", "tv_finish", "tv_fused"]

# every symbol include/ofdis.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "ofdis_params_oppoint", "ofdis_last_error", "ofdis_version", "ofdis_device_count", "ofdis_set_device",
    "ofdis_flow", "ofdis_batch_create", "ofdis_batch_destroy", "ofdis_batch_input", "ofdis_batch_input_elems",
    "ofdis_batch_upload", "ofdis_batch_upload_b_gradients", "ofdis_batch_initflow_elems", "ofdis_batch_set_initflow", "ofdis_batch_upload_initflow",
    "ofdis_batch_build_pyramids_u8", "ofdis_batch_run", "ofdis_batch_set_pipeline", "ofdis_batch_join", "ofdis_batch_flow",
    "ofdis_batch_level_flow", "ofdis_batch_download", "ofdis_batch_upsample", "ofdis_batch_timing", "ofdis_batch_kernel_time",
    "ofdis_image_warp", "ofdis_get_derivatives", "ofdis_tv_system", "ofdis_sor_coupled", "ofdis_patchgrid_level",
    "ofdis_varref_level", "ofdis_dev_alloc", "ofdis_dev_free", "ofdis_memcpy_h2d", "ofdis_memcpy_d2h", "ofdis_memcpy_d2d", "ofdis_sync",
    "ofdis_batch_set_graph", "ofdis_batch_status", "ofdis_flow_cache_clear", "ofdis_get_tuning", "ofdis_set_tuning", "ofdis_batch_kernel_times", "ofdis_device_pci_bus_id",
    "ofdis_batch_upsample_frames",
    "ofdis_build_id", "ofdis_stream_create", "ofdis_stream_destroy", "ofdis_host_alloc", "ofdis_host_free", "ofdis_memcpy_h2d_async", "ofdis_memcpy_d2h_async",
    "ofdis_event_create", "ofdis_event_destroy", "ofdis_event_record", "ofdis_stream_wait_event", "ofdis_event_sync",
]
OFDIS_VERSION = 3  # include/ofdis.h: the struct layouts below (OfdisTuning: 20 ints) belong to this ABI version


class OfdisTuning(C.Structure):
    """include/ofdis.h: ofdis_tuning -- kernel-selection knobs, every setting bit-identical except `contract`
    (0 = exact arithmetic, 1 = the FMA / fast-reciprocal tolerance contract)."""
    _fields_ = [(n, C.c_int) for n in ("gray8", "rgb12", "rgb12_lpp", "fused_tv", "fused_mw_max", "fused_split",
                                       "finish_fusion", "fused_strip", "prep_band_rows", "graph", "flow_dma", "flow_whole",
                                       "fused_xcu_max", "fused_tp_pipe", "fused_xcu_spin", "contract", "fused_xcu_drop", "prep_densify", "fused_tall_group", "fused_rgb_min")]


class OfdisError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libofdis_hip.so (built by of_dis_amd.build).  Fails loudly when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OfdisError(f"{LIB_PATH} is missing: run `python -m of_dis_amd.build` (no CPU fallback exists)")
        L = C.CDLL(LIB_PATH)
        if L.ofdis_version() != OFDIS_VERSION:  # a stale build: ofdis_get_tuning would write past / read garbage from our struct
            raise OfdisError(f"{LIB_PATH} has ABI version {L.ofdis_version()}, this binding expects {OFDIS_VERSION}: "
                             "rebuild with `python -m of_dis_amd.build`")
        L.ofdis_last_error.restype = C.c_char_p
        L.ofdis_build_id.restype = C.c_char_p
        L.ofdis_dev_alloc.restype = VP
        L.ofdis_dev_alloc.argtypes = [C.c_size_t]
        L.ofdis_dev_free.argtypes = [VP]
        L.ofdis_memcpy_h2d.argtypes = [VP, VP, C.c_size_t]
        L.ofdis_memcpy_d2h.argtypes = [VP, VP, C.c_size_t]
        L.ofdis_sync.argtypes = [VP]
        L.ofdis_stream_create.restype = VP
        L.ofdis_stream_create.argtypes = []
        L.ofdis_stream_destroy.restype = None
        L.ofdis_stream_destroy.argtypes = [VP]
        L.ofdis_host_alloc.restype = VP
        L.ofdis_host_alloc.argtypes = [C.c_size_t]
        L.ofdis_host_free.restype = None
        L.ofdis_host_free.argtypes = [VP]
        L.ofdis_memcpy_h2d_async.argtypes = [VP, VP, C.c_size_t, VP]
        L.ofdis_memcpy_d2h_async.argtypes = [VP, VP, C.c_size_t, VP]
        L.ofdis_event_create.restype = VP
        L.ofdis_event_create.argtypes = []
        L.ofdis_event_destroy.restype = None
        L.ofdis_event_destroy.argtypes = [VP]
        L.ofdis_event_record.argtypes = [VP, VP]
        L.ofdis_stream_wait_event.argtypes = [VP, VP]
        L.ofdis_event_sync.argtypes = [VP]
        L.ofdis_memcpy_d2d.argtypes = [VP, VP, C.c_size_t, VP]
        L.ofdis_batch_create.argtypes = [C.POINTER(VP), C.POINTER(OfdisParams), C.c_int]
        L.ofdis_batch_destroy.argtypes = [VP]
        L.ofdis_batch_input.restype = VP
        L.ofdis_batch_input.argtypes = [VP, C.c_int, C.c_int]
        L.ofdis_batch_input_elems.restype = C.c_size_t
        L.ofdis_batch_input_elems.argtypes = [VP, C.c_int]
        L.ofdis_batch_upload.argtypes = [VP, C.c_int, C.POINTER(FP), C.POINTER(FP), C.POINTER(FP), C.POINTER(FP), VP]
        L.ofdis_batch_build_pyramids_u8.argtypes = [VP, VP, VP, C.c_int, C.c_int, VP]
        L.ofdis_batch_run.argtypes = [VP, VP]
        L.ofdis_batch_status.argtypes = [VP]
        L.ofdis_batch_flow.restype = VP
        L.ofdis_batch_flow.argtypes = [VP]
        L.ofdis_batch_level_flow.restype = VP
        L.ofdis_batch_level_flow.argtypes = [VP, C.c_int]
        L.ofdis_batch_download.argtypes = [VP, C.c_int, FP, VP]
        L.ofdis_batch_timing.argtypes = [VP, C.c_int]
        L.ofdis_batch_kernel_time.argtypes = [VP, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long)]
        L.ofdis_image_warp.argtypes = [VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]
        L.ofdis_get_derivatives.argtypes = [VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]
        L.ofdis_tv_system.argtypes = [VP, VP, VP, VP, VP, VP, VP, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                      C.c_int, C.c_int, VP]
        L.ofdis_sor_coupled.argtypes = [VP, VP, VP, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, VP]
        L.ofdis_patchgrid_level.argtypes = [C.POINTER(OfdisParams), C.c_int, VP, VP, VP, VP, VP, VP, VP, C.c_int, VP]
        L.ofdis_varref_level.argtypes = [C.POINTER(OfdisParams), C.c_int, VP, VP, VP, C.c_int, VP]
        L.ofdis_flow.argtypes = [C.POINTER(OfdisParams)] + [C.POINTER(FP)] * 6 + [FP, FP]
        L.ofdis_params_oppoint.argtypes = [C.POINTER(OfdisParams), C.c_int, C.c_int, C.c_int]
        L.ofdis_batch_upsample.argtypes = [VP, VP, C.c_int, C.c_int, VP]
        L.ofdis_batch_upsample_frames.argtypes = [VP, C.c_int, C.c_int, VP, C.c_int, C.c_int, VP]
        L.ofdis_batch_set_pipeline.argtypes = [VP, C.c_int]
        L.ofdis_batch_join.argtypes = [VP, VP]
        L.ofdis_batch_set_graph.argtypes = [VP, C.c_int]
        L.ofdis_flow_cache_clear.restype = None
        L.ofdis_batch_upload_b_gradients.argtypes = [VP, C.c_int, C.POINTER(FP), C.POINTER(FP), VP]
        L.ofdis_batch_initflow_elems.restype = C.c_size_t
        L.ofdis_batch_initflow_elems.argtypes = [VP]
        L.ofdis_batch_set_initflow.argtypes = [VP, VP]
        L.ofdis_batch_upload_initflow.argtypes = [VP, C.c_int, FP, VP]
        L.ofdis_batch_kernel_times.argtypes = [VP, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
        L.ofdis_get_tuning.argtypes = [C.POINTER(OfdisTuning)]
        L.ofdis_set_tuning.argtypes = [C.POINTER(OfdisTuning)]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise OfdisError(f"ofdis status {rc}: {lib().ofdis_last_error().decode()}")


def build_id():
    """ofdis_build_id(): the hash of the kernel sources + flags the loaded library was built from."""
    return lib().ofdis_build_id().decode()


def device_pci_bus_id(device):
    buf = C.create_string_buffer(32)
    check(lib().ofdis_device_pci_bus_id(device, buf, 32))
    return buf.value.decode()


def get_tuning():
    t = OfdisTuning()
    check(lib().ofdis_get_tuning(C.byref(t)))
    return t


def set_tuning(**kw):
    """Change kernel-selection knobs (ofdis_set_tuning); returns the previous settings (pass them to restore_tuning)."""
    old = get_tuning()
    new = get_tuning()
    for k, v in kw.items():
        setattr(new, k, v)
    check(lib().ofdis_set_tuning(C.byref(new)))
    lib().ofdis_flow_cache_clear()  # cached drop-in contexts were sized under the old settings
    return old


def restore_tuning(old):
    check(lib().ofdis_set_tuning(C.byref(old)))
    lib().ofdis_flow_cache_clear()


class Dev:
    """A device buffer owned through ofdis_dev_alloc / ofdis_dev_free."""

    def __init__(self, arr=None, nbytes=None):
        L = lib()
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            nbytes = arr.nbytes
        self.nbytes = int(nbytes)
        self.ptr = L.ofdis_dev_alloc(self.nbytes)
        if not self.ptr:
            raise OfdisError("device allocation failed")
        if arr is not None:
            check(L.ofdis_memcpy_h2d(self.ptr, arr.ctypes.data, self.nbytes))

    def get(self, shape, dtype=_f32):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(lib().ofdis_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().ofdis_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Stream:
    """A non-blocking HIP stream owned through ofdis_stream_create / ofdis_stream_destroy (`.ptr` goes wherever the ABI takes
    a `stream`)."""

    def __init__(self):
        self.ptr = lib().ofdis_stream_create()
        if not self.ptr:
            raise OfdisError(f"ofdis_stream_create: {lib().ofdis_last_error().decode()}")

    def sync(self):
        check(lib().ofdis_sync(self.ptr))

    def close(self):
        if self.ptr:
            lib().ofdis_stream_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Event:
    """A HIP event owned through ofdis_event_create / ofdis_event_destroy: orders work across streams without the host."""

    def __init__(self):
        self.ptr = lib().ofdis_event_create()
        if not self.ptr:
            raise OfdisError(f"ofdis_event_create: {lib().ofdis_last_error().decode()}")

    def record(self, stream):
        check(lib().ofdis_event_record(self.ptr, stream.ptr if isinstance(stream, Stream) else stream))

    def wait(self, stream):
        """make `stream` wait for the last recorded point"""
        check(lib().ofdis_stream_wait_event(stream.ptr if isinstance(stream, Stream) else stream, self.ptr))

    def sync(self):
        check(lib().ofdis_event_sync(self.ptr))

    def close(self):
        if self.ptr:
            lib().ofdis_event_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostBuf:
    """Page-locked host memory (ofdis_host_alloc) seen as a numpy array: the source / destination of the asynchronous copies."""

    def __init__(self, shape, dtype=_f32):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = lib().ofdis_host_alloc(self.nbytes)
        if not self.ptr:
            raise OfdisError(f"ofdis_host_alloc: {lib().ofdis_last_error().decode()}")
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint8)), shape=(self.nbytes,)).view(dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib().ofdis_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _f(a):
    return np.ascontiguousarray(a, dtype=_f32)


# ------------------------------------------------------------------ per-function entry points (numpy in/out)
def image_warp(src, wx, wy):
    """src [B,noc,h,w], wx/wy [B,h,w] -> dst [B,noc,h,w], mask [B,h,w]"""
    src, wx, wy = _f(src), _f(wx), _f(wy)
    B, noc, h, w = src.shape
    dsrc, dwx, dwy = Dev(src), Dev(wx), Dev(wy)
    ddst, dmask = Dev(nbytes=src.nbytes), Dev(nbytes=wx.nbytes)
    check(lib().ofdis_image_warp(ddst.ptr, dmask.ptr, dsrc.ptr, dwx.ptr, dwy.ptr, w, h, noc, B, None))
    check(lib().ofdis_sync(None))
    return ddst.get(src.shape), dmask.get(wx.shape)


def get_derivatives(im1, im2w):
    """im1, im2w [B,noc,h,w] -> [B,8,noc,h,w]"""
    im1, im2w = _f(im1), _f(im2w)
    B, noc, h, w = im1.shape
    d1, d2 = Dev(im1), Dev(im2w)
    dout = Dev(nbytes=im1.nbytes * 8)
    check(lib().ofdis_get_derivatives(dout.ptr, d1.ptr, d2.ptr, w, h, noc, B, None))
    check(lib().ofdis_sync(None))
    return dout.get((B, 8, noc, h, w))


def tv_system(mask, wx, wy, du, dv, derivs, tv_alpha, tv_gamma, tv_delta):
    """-> [B,7,h,w] = a11,a12,a22,b1,b2,smooth_horiz,smooth_vert"""
    mask, wx, wy, du, dv, derivs = [_f(x) for x in (mask, wx, wy, du, dv, derivs)]
    B, h, w = mask.shape
    noc = derivs.shape[2]
    bufs = [Dev(x) for x in (mask, wx, wy, du, dv, derivs)]
    dout = Dev(nbytes=mask.nbytes * 7)
    check(lib().ofdis_tv_system(dout.ptr, *[b.ptr for b in bufs], tv_alpha, tv_gamma, tv_delta, w, h, noc, B, None))
    check(lib().ofdis_sync(None))
    return dout.get((B, 7, h, w))


def sor_coupled(du, dv, sys, iterations, omega):
    """du, dv [B,h,w]; sys [B,7,h,w] -> new du, dv"""
    du, dv, sys = _f(du), _f(dv), _f(sys)
    B, h, w = du.shape
    ddu, ddv, dsys = Dev(du), Dev(dv), Dev(sys)
    check(lib().ofdis_sor_coupled(ddu.ptr, ddv.ptr, dsys.ptr, iterations, omega, w, h, B, None))
    check(lib().ofdis_sync(None))
    return ddu.get(du.shape), ddv.get(dv.shape)


def patchgrid_level(p, level, im_a, im_a_dx, im_a_dy, im_b, flow_prev=None):
    """Planes [B,tmp_h,tmp_w,noc]; flow_prev [B,h/2,w/2,2] or None -> p [B,nop,2], flow [B,h,w,2]"""
    im_a, im_a_dx, im_a_dy, im_b = [_f(x) for x in (im_a, im_a_dx, im_a_dy, im_b)]
    B = im_a.shape[0]
    w, h = p.level_size(level)
    nw, nh = p.grid(level)
    nop = nw * nh
    bufs = [Dev(x) for x in (im_a, im_a_dx, im_a_dy, im_b)]
    dprev = Dev(_f(flow_prev)) if flow_prev is not None else None
    dp, dflow = Dev(nbytes=B * nop * 2 * 4), Dev(nbytes=B * h * w * p.nop * 4)
    check(lib().ofdis_patchgrid_level(C.byref(p), level, *[b.ptr for b in bufs], dprev.ptr if dprev else None,
                                      dp.ptr, dflow.ptr, B, None))
    return dp.get((B, nop, 2)), dflow.get((B, h, w, p.nop))


def varref_level(p, level, im_a, im_b, flow):
    """im_a, im_b [B,tmp_h,tmp_w,noc]; flow [B,h,w,2] -> refined flow"""
    im_a, im_b, flow = _f(im_a), _f(im_b), _f(flow)
    B = im_a.shape[0]
    da, db, df = Dev(im_a), Dev(im_b), Dev(flow)
    check(lib().ofdis_varref_level(C.byref(p), level, da.ptr, db.ptr, df.ptr, B, None))
    return df.get(flow.shape)


def _ptr_array(planes, n):
    arr = (FP * n)()
    for i in range(n):
        arr[i] = planes[i].ctypes.data_as(FP) if (i < len(planes) and planes[i] is not None) else None
    return arr


def flow(p, pyr_a, pyr_a_dx, pyr_a_dy, pyr_b, initflow=None, pyr_b_dx=None, pyr_b_dy=None):
    """ofdis_flow(): the drop-in for OFC::OFClass::OFClass with host pyramids (lists over levels 0..sc_f)."""
    n = p.sc_f + 1
    keep = [[_f(x) if x is not None else None for x in pl] for pl in (pyr_a, pyr_a_dx, pyr_a_dy, pyr_b)]
    keep_b = [[_f(x) if x is not None else None for x in pl] for pl in (pyr_b_dx, pyr_b_dy) if pl is not None]
    w, h = p.level_size(p.sc_l)
    out = np.zeros((h, w, p.nop), _f32)
    nullarr = C.cast(None, C.POINTER(FP))
    bdx = _ptr_array(keep_b[0], n) if len(keep_b) == 2 else nullarr
    bdy = _ptr_array(keep_b[1], n) if len(keep_b) == 2 else nullarr
    check(lib().ofdis_flow(C.byref(p), _ptr_array(keep[0], n), _ptr_array(keep[1], n), _ptr_array(keep[2], n),
                           _ptr_array(keep[3], n), bdx, bdy, out.ctypes.data_as(FP),
                           _f(initflow).ctypes.data_as(FP) if initflow is not None else None))
    return out


class Batch:
    """ofdis_batch: `nframes` frame pairs of one geometry resident in HBM."""

    def __init__(self, p, nframes):
        self.p = p.copy()
        self.nframes = nframes
        self.h = VP()
        check(lib().ofdis_batch_create(C.byref(self.h), C.byref(self.p), nframes))

    def close(self):
        if self.h:
            lib().ofdis_batch_destroy(self.h)
            self.h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def input_ptr(self, level, kind):
        return lib().ofdis_batch_input(self.h, level, kind)

    def input_elems(self, level):
        return lib().ofdis_batch_input_elems(self.h, level)

    def upload(self, frame, pyr_a, pyr_a_dx, pyr_a_dy, pyr_b, stream=None):
        n = self.p.sc_f + 1
        keep = [[_f(x) if x is not None else None for x in pl] for pl in (pyr_a, pyr_a_dx, pyr_a_dy, pyr_b)]
        check(lib().ofdis_batch_upload(self.h, frame, _ptr_array(keep[0], n), _ptr_array(keep[1], n),
                                       _ptr_array(keep[2], n), _ptr_array(keep[3], n), stream))
        check(lib().ofdis_sync(stream))

    def upload_b_gradients(self, frame, pyr_b_dx, pyr_b_dy, stream=None):
        n = self.p.sc_f + 1
        keep = [[_f(x) if x is not None else None for x in pl] for pl in (pyr_b_dx, pyr_b_dy)]
        check(lib().ofdis_batch_upload_b_gradients(self.h, frame, _ptr_array(keep[0], n), _ptr_array(keep[1], n), stream))
        check(lib().ofdis_sync(stream))

    def set_input(self, level, kind, arr):
        """arr: [nframes, tmp_h, tmp_w, noc] float32 host array"""
        arr = _f(arr)
        assert arr.size == self.input_elems(level) * self.nframes, (arr.shape, self.input_elems(level))
        check(lib().ofdis_memcpy_h2d(self.input_ptr(level, kind), arr.ctypes.data, arr.nbytes))

    def upload_initflow(self, frame, initflow, stream=None):
        a = _f(initflow)
        assert a.size == lib().ofdis_batch_initflow_elems(self.h), (a.shape, lib().ofdis_batch_initflow_elems(self.h))
        check(lib().ofdis_batch_upload_initflow(self.h, frame, a.ctypes.data_as(FP), stream))
        check(lib().ofdis_sync(stream))  # `a` may be a temporary: the copy must have read it before it goes away

    def set_initflow(self, dev_ptr):
        check(lib().ofdis_batch_set_initflow(self.h, dev_ptr))

    def build_pyramids_u8(self, img_a_ptr, img_b_ptr, width_org, height_org, stream=None):
        check(lib().ofdis_batch_build_pyramids_u8(self.h, img_a_ptr, img_b_ptr, width_org, height_org, stream))

    def run(self, stream=None):
        check(lib().ofdis_batch_run(self.h, stream))

    def set_pipeline(self, sub_batches):
        check(lib().ofdis_batch_set_pipeline(self.h, sub_batches))

    def set_graph(self, mode):
        check(lib().ofdis_batch_set_graph(self.h, mode))

    def join(self, stream=None):
        check(lib().ofdis_batch_join(self.h, stream))

    def status(self):
        """ofdis_batch_status: 0, or OFDIS_ERR_DEVICE (-3) when the last pass's results are invalid (call after a sync)."""
        return lib().ofdis_batch_status(self.h)

    def flow_ptr(self):
        return lib().ofdis_batch_flow(self.h)

    def download(self, frame, stream=None):
        w, h = self.p.level_size(self.p.sc_l)
        out = np.zeros((h, w, self.p.nop), _f32)
        check(lib().ofdis_batch_download(self.h, frame, out.ctypes.data_as(FP), stream))
        return out

    def upsample(self, width_org, height_org, out_ptr=None, stream=None):
        """ofdis_batch_upsample.  With out_ptr (device pointer) only enqueues; otherwise returns the host array
        [nframes][height_org][width_org][2]."""
        if out_ptr is not None:
            check(lib().ofdis_batch_upsample(self.h, out_ptr, width_org, height_org, stream))
            return None
        out = np.zeros((self.nframes, height_org, width_org, self.p.nop), _f32)
        d = Dev(nbytes=out.nbytes)
        check(lib().ofdis_batch_upsample(self.h, d.ptr, width_org, height_org, stream))
        check(lib().ofdis_sync(stream))
        check(lib().ofdis_memcpy_d2h(out.ctypes.data, d.ptr, out.nbytes))
        return out

    def upsample_frames(self, first, count, width_org, height_org, stream=None):
        """ofdis_batch_upsample_frames: the full-resolution flow of frames [first, first + count) as a host array."""
        out = np.zeros((count, height_org, width_org, self.p.nop), _f32)
        d = Dev(nbytes=out.nbytes)
        check(lib().ofdis_batch_upsample_frames(self.h, first, count, d.ptr, width_org, height_org, stream))
        check(lib().ofdis_sync(stream))
        check(lib().ofdis_memcpy_d2h(out.ctypes.data, d.ptr, out.nbytes))
        return out

    def download_all(self):
        w, h = self.p.level_size(self.p.sc_l)
        out = np.zeros((self.nframes, h, w, self.p.nop), _f32)
        self.join(None)
        check(lib().ofdis_sync(None))
        check(lib().ofdis_memcpy_d2h(out.ctypes.data, self.flow_ptr(), out.nbytes))
        return out

    def level_flow(self, level):
        w, h = self.p.level_size(level)
        out = np.zeros((self.nframes, h, w, self.p.nop), _f32)
        self.join(None)
        check(lib().ofdis_sync(None))
        check(lib().ofdis_memcpy_d2h(out.ctypes.data, lib().ofdis_batch_level_flow(self.h, level), out.nbytes))
        return out

    def timing(self, enable=True):
        check(lib().ofdis_batch_timing(self.h, int(enable)))

    def kernel_time(self, k):
        ms, n = C.c_double(0), C.c_long(0)
        check(lib().ofdis_batch_kernel_time(self.h, k, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_times(self, k, capacity=4096):
        """Per-launch milliseconds of a kernel class in launch order (a pass launches a class once per level, coarsest first)."""
        buf, n = (C.c_double * capacity)(), C.c_int(0)
        check(lib().ofdis_batch_kernel_times(self.h, k, buf, capacity, C.byref(n)))
        return [buf[i] for i in range(min(n.value, capacity))]
