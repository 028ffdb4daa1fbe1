"""ctypes wrappers of the CHECKERS -- test infrastructure only.

  oracle.C            the plain-C restatement (oracle/ofdis_oracle.c -> liboracle.so)
  oracle.ref(...)     the unmodified reference sources compiled in place (oracle/_ref/*.so)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Both expose the same numpy-level methods so that tests read the same against either.
All plane arguments are float32 numpy arrays; TV planes are packed [c][h][w] (or [h][w]), pyramid
planes are the reference's padded, channel-interleaved [tmp_h][tmp_w][noc].
"""
import ctypes as C
import os
import subprocess

import numpy as np

from of_dis_amd.params import OfdisParams

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32 = np.float32
FP = C.POINTER(C.c_float)


def build(verbose=False):
    """Compile liboracle.so and (when /root/reference exists) oracle/_ref/*.so."""
    r = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stdout)


def _p(a):
    return a.ctypes.data_as(FP)


def _c(a):
    return np.ascontiguousarray(a, dtype=_f32)


def _ptr_array(planes, n):
    arr = (FP * n)()
    for i in range(n):
        arr[i] = _p(planes[i]) if (i < len(planes) and planes[i] is not None) else None
    return arr


class _Base:
    """Common numpy-level API; subclasses provide the raw calls."""
    name = "?"

    def tv_consts(self, p):
        # refine_variational.cpp:40-42
        return (_f32(0.25) * _f32(p.tv_alpha), _f32(p.tv_delta) * _f32(0.5) / _f32(3.0),
                _f32(p.tv_gamma) * _f32(0.5) / _f32(3.0))


class COracle(_Base):
    name = "restatement"

    def __init__(self):
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        self.lib = C.CDLL(path)
        L = self.lib
        L.oracle_plane_elems.restype = C.c_size_t
        for f in ("oracle_varref_level", "oracle_patchgrid_level", "oracle_flow"):
            getattr(L, f).restype = C.c_int

    def set_reduce_order(self, wave64):
        self.lib.oracle_set_reduce_order(int(bool(wave64)))

    def image_warp(self, src, wx, wy):
        src = _c(src)
        h, w = wx.shape
        noc = src.size // (h * w)
        dst = np.zeros_like(src)
        mask = np.zeros((h, w), _f32)
        self.lib.oracle_image_warp(_p(dst), _p(mask), _p(src), _p(_c(wx)), _p(_c(wy)), w, h, noc)
        return dst, mask

    def get_derivatives(self, im1, im2):
        im1, im2 = _c(im1), _c(im2)
        h, w = im1.shape[-2:]
        noc = im1.size // (h * w)
        out = np.zeros((8, noc, h, w), _f32)
        self.lib.oracle_get_derivatives(_p(im1), _p(im2), _p(out), w, h, noc)
        return out

    def compute_smoothness(self, uu, vv, quarter_alpha):
        h, w = uu.shape
        sh, sv = np.zeros((h, w), _f32), np.zeros((h, w), _f32)
        self.lib.oracle_compute_smoothness(_p(sh), _p(sv), _p(_c(uu)), _p(_c(vv)), C.c_float(quarter_alpha), w, h)
        return sh, sv

    def compute_data(self, mask, du, dv, derivs, half_delta_over3, half_gamma_over3):
        h, w = mask.shape
        derivs = _c(derivs)
        noc = derivs.size // (8 * h * w)
        out = np.zeros((5, h, w), _f32)
        self.lib.oracle_compute_data(_p(out), _p(_c(mask)), _p(_c(du)), _p(_c(dv)), _p(derivs),
                                     C.c_float(half_delta_over3), C.c_float(half_gamma_over3), w, h, noc)
        return out

    def sub_laplacian(self, dst, src, wh, wv):
        h, w = dst.shape
        dst = _c(dst).copy()
        self.lib.oracle_sub_laplacian(_p(dst), _p(_c(src)), _p(_c(wh)), _p(_c(wv)), w, h)
        return dst

    def sor_coupled(self, du, dv, a11, a12, a22, b1, b2, sh, sv, iterations, omega, slow=False):
        h, w = du.shape
        du, dv, a11, a12, a22 = [_c(x).copy() for x in (du, dv, a11, a12, a22)]
        if slow:
            self.lib.oracle_sor_coupled_slow(_p(du), _p(dv), _p(a11), _p(a12), _p(a22), _p(_c(b1)), _p(_c(b2)),
                                             _p(_c(sh)), _p(_c(sv)), iterations, C.c_float(omega), w, h)
        else:
            self.lib.oracle_sor_coupled(_p(du), _p(dv), _p(a11), _p(a12), _p(a22), _p(_c(b1)), _p(_c(b2)),
                                        _p(_c(sh)), _p(_c(sv)), iterations, C.c_float(omega), w, h)
        return du, dv, a11, a12, a22

    def varref_level(self, p, level, im_a, im_b, flow):
        flow = _c(flow).copy()
        rc = self.lib.oracle_varref_level(C.byref(p), level, _p(_c(im_a)), _p(_c(im_b)), _p(flow))
        if rc:
            raise RuntimeError(f"oracle_varref_level rc={rc}")
        return flow

    def patchgrid_level(self, p, level, im_a, im_a_dx, im_a_dy, im_b, flow_prev=None, want_pweight=False):
        w, h = p.level_size(level)
        nw, nh = p.grid(level)
        nop = nw * nh
        nv = p.noc * p.p_samp_s ** 2
        pout = np.zeros((nop, 2), _f32)
        pw = np.zeros((nop, nv), _f32)
        flow = np.zeros((h, w, 2), _f32)
        n = C.c_int(0)
        fp = _p(_c(flow_prev)) if flow_prev is not None else None
        rc = self.lib.oracle_patchgrid_level(C.byref(p), level, _p(_c(im_a)), _p(_c(im_a_dx)), _p(_c(im_a_dy)),
                                             _p(_c(im_b)), fp, _p(pout), _p(pw), _p(flow), C.byref(n))
        if rc:
            raise RuntimeError(f"oracle_patchgrid_level rc={rc}")
        assert n.value == nop, (n.value, nop)
        return (pout, flow, pw) if want_pweight else (pout, flow)

    def flow(self, p, pyr_a, pyr_a_dx, pyr_a_dy, pyr_b, initflow=None, want_levels=False):
        n = p.sc_f + 1
        w, h = p.level_size(p.sc_l)
        out = np.zeros((h, w, 2), _f32)
        tot = sum(2 * (p.width >> l) * (p.height >> l) for l in range(p.sc_l, p.sc_f + 1))
        lv = np.zeros(tot, _f32)
        keep = [[_c(x) if x is not None else None for x in pl] for pl in (pyr_a, pyr_a_dx, pyr_a_dy, pyr_b)]
        ini = _p(_c(initflow)) if initflow is not None else None
        rc = self.lib.oracle_flow(C.byref(p), _ptr_array(keep[0], n), _ptr_array(keep[1], n),
                                  _ptr_array(keep[2], n), _ptr_array(keep[3], n), _p(out), ini,
                                  _p(lv) if want_levels else None)
        if rc:
            raise RuntimeError(f"oracle_flow rc={rc}")
        if not want_levels:
            return out
        levels, off = {}, 0
        for l in range(p.sc_f, p.sc_l - 1, -1):
            ww, hh = p.level_size(l)
            levels[l] = lv[off:off + 2 * ww * hh].reshape(hh, ww, 2).copy()
            off += 2 * ww * hh
        return out, levels

    # ---- host pre/post-processing (run_dense.cpp restated)
    def build_pyramid(self, p, img_u8):
        img_u8 = np.ascontiguousarray(img_u8, dtype=np.uint8)
        ho, wo = img_u8.shape[:2]
        n = p.sc_f + 1
        planes = [[np.zeros(p.plane_shape(l), _f32) for l in range(n)] for _ in range(3)]
        self.lib.oracle_build_pyramid(C.byref(p), img_u8.ctypes.data_as(C.POINTER(C.c_uint8)), wo, ho,
                                      _ptr_array(planes[0], n), _ptr_array(planes[1], n), _ptr_array(planes[2], n))
        return planes  # [img, dx, dy] each a list over levels 0..sc_f

    def upsample_crop(self, p, flow, width_org, height_org):
        out = np.zeros((height_org, width_org, 2), _f32)
        self.lib.oracle_upsample_crop(C.byref(p), _p(_c(flow)), width_org, height_org, _p(out))
        return out


class RefLib(_Base):
    """The reference itself (oracle/_ref/libofdis_ref_{int,rgb}[_w64].so; kind "de_int" / "de_rgb" = the stereo-depth
    build, SELECTMODE=2, whose flows have ONE channel)."""

    def __init__(self, kind="int", wave64=False):
        fn = f"libofdis_ref_{kind}{'_w64' if wave64 else ''}.so"
        path = os.path.join(_HERE, "_ref", fn)
        if not os.path.exists(path):
            if os.path.isdir("/root/reference"):
                build()
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} missing (built from /root/reference by oracle/Makefile)")
        self.lib = C.CDLL(path)
        self.noc = self.lib.ofdis_ref_noc()
        self.wave64 = bool(self.lib.ofdis_ref_wave64_order())
        self.name = f"reference[{kind}{',wave64' if wave64 else ''}]"
        self.nop = 1 if kind.startswith("de_") else 2   # oflow.cpp:76-80

    def image_warp(self, src, wx, wy):
        src = _c(src)
        h, w = wx.shape
        assert src.size == self.noc * h * w
        dst = np.zeros_like(src)
        mask = np.zeros((h, w), _f32)
        self.lib.ofdis_ref_image_warp(_p(dst), _p(mask), _p(src), _p(_c(wx)), _p(_c(wy)), w, h)
        return dst, mask

    def get_derivatives(self, im1, im2):
        im1, im2 = _c(im1), _c(im2)
        h, w = im1.shape[-2:]
        out = np.zeros((8, self.noc, h, w), _f32)
        self.lib.ofdis_ref_get_derivatives(_p(im1), _p(im2), _p(out), w, h)
        return out

    def compute_smoothness(self, uu, vv, quarter_alpha):
        h, w = uu.shape
        sh, sv = np.zeros((h, w), _f32), np.zeros((h, w), _f32)
        self.lib.ofdis_ref_compute_smoothness(_p(sh), _p(sv), _p(_c(uu)), _p(_c(vv)), C.c_float(quarter_alpha), w, h)
        return sh, sv

    def compute_data(self, mask, du, dv, derivs, half_delta_over3, half_gamma_over3):
        h, w = mask.shape
        out = np.zeros((5, h, w), _f32)
        self.lib.ofdis_ref_compute_data(_p(out), _p(_c(mask)), _p(_c(du)), _p(_c(dv)), _p(_c(derivs)),
                                        C.c_float(half_delta_over3), C.c_float(0.0), C.c_float(half_gamma_over3), w, h)
        return out

    def sub_laplacian(self, dst, src, wh, wv):
        h, w = dst.shape
        dst = _c(dst).copy()
        self.lib.ofdis_ref_sub_laplacian(_p(dst), _p(_c(src)), _p(_c(wh)), _p(_c(wv)), w, h)
        return dst

    def sor_coupled(self, du, dv, a11, a12, a22, b1, b2, sh, sv, iterations, omega, slow=False):
        h, w = du.shape
        du, dv, a11, a12, a22 = [_c(x).copy() for x in (du, dv, a11, a12, a22)]
        self.lib.ofdis_ref_sor_coupled(_p(du), _p(dv), _p(a11), _p(a12), _p(a22), _p(_c(b1)), _p(_c(b2)),
                                       _p(_c(sh)), _p(_c(sv)), iterations, C.c_float(omega), w, h, int(slow))
        return du, dv, a11, a12, a22

    def varref_level(self, p, level, im_a, im_b, flow):
        w, h = p.level_size(level)
        flow = _c(flow).copy()
        rc = self.lib.ofdis_ref_varref_level(_p(_c(im_a)), _p(_c(im_b)), w, h, level, p.imgpadding, p.p_samp_s, p.noc,
                                             C.c_float(p.tv_alpha), C.c_float(p.tv_gamma), C.c_float(p.tv_delta),
                                             p.tv_innerit, p.tv_solverit, C.c_float(p.tv_sor), _p(flow))
        if rc:
            raise RuntimeError(f"ofdis_ref_varref_level rc={rc}")
        return flow

    def patchgrid_level(self, p, level, im_a, im_a_dx, im_a_dy, im_b, flow_prev=None):
        w, h = p.level_size(level)
        nw, nh = p.grid(level)
        nop = nw * nh
        pout = np.zeros((nop, 2), _f32)
        flow = np.zeros((h, w, self.nop), _f32)
        n = C.c_int(0)
        fp = _p(_c(flow_prev)) if flow_prev is not None else None
        rc = self.lib.ofdis_ref_patchgrid_level(
            _p(_c(im_a)), _p(_c(im_a_dx)), _p(_c(im_a_dy)), _p(_c(im_b)), w, h, level, p.imgpadding, p.max_iter,
            p.min_iter, C.c_float(p.dp_thresh), C.c_float(p.dr_thresh), C.c_float(p.res_thresh), p.p_samp_s,
            C.c_float(p.patove), p.costfct, p.noc, p.patnorm, fp, _p(pout), _p(flow), C.byref(n))
        if rc:
            raise RuntimeError(f"ofdis_ref_patchgrid_level rc={rc}")
        assert n.value == nop, (n.value, nop)
        return pout, flow

    def flow(self, p, pyr_a, pyr_a_dx, pyr_a_dy, pyr_b, initflow=None, pyr_b_dx=None, pyr_b_dy=None):
        n = p.sc_f + 1
        w, h = p.level_size(p.sc_l)
        out = np.zeros((h, w, self.nop), _f32)
        keep = [[_c(x) if x is not None else None for x in pl]
                for pl in (pyr_a, pyr_a_dx, pyr_a_dy, pyr_b, pyr_b_dx or pyr_a_dx, pyr_b_dy or pyr_a_dy)]
        ini = _p(_c(initflow)) if initflow is not None else None
        rc = self.lib.ofdis_ref_flow(
            _ptr_array(keep[0], n), _ptr_array(keep[1], n), _ptr_array(keep[2], n), _ptr_array(keep[3], n),
            _ptr_array(keep[4], n), _ptr_array(keep[5], n), p.imgpadding, _p(out), ini, p.width, p.height, p.sc_f,
            p.sc_l, p.max_iter, p.min_iter, C.c_float(p.dp_thresh), C.c_float(p.dr_thresh), C.c_float(p.res_thresh),
            p.p_samp_s, C.c_float(p.patove), p.usefbcon, p.costfct, p.noc, p.patnorm, p.usetvref,
            C.c_float(p.tv_alpha), C.c_float(p.tv_gamma), C.c_float(p.tv_delta), p.tv_innerit, p.tv_solverit,
            C.c_float(p.tv_sor), p.verbosity)
        if rc:
            raise RuntimeError(f"ofdis_ref_flow rc={rc}")
        return out


_cache = {}


def c_oracle():
    if "c" not in _cache:
        _cache["c"] = COracle()
    return _cache["c"]


def ref(kind="int", wave64=False):
    key = (kind, wave64)
    if key not in _cache:
        _cache[key] = RefLib(kind, wave64)
    return _cache[key]


def have_ref(kind="int", wave64=False):
    fn = f"libofdis_ref_{kind}{'_w64' if wave64 else ''}.so"
    return os.path.exists(os.path.join(_HERE, "_ref", fn)) or os.path.isdir("/root/reference")


def ref_expected():
    """The compiled reference is EXPECTED on this machine: its sources are here (/root/reference) or a build of it was
    shipped with the snapshot (oracle/_ref/ holds libraries).  Only when neither is the case may a test skip its comparison
    against the reference build."""
    d = os.path.join(_HERE, "_ref")
    return os.path.isdir("/root/reference") or (os.path.isdir(d) and any(f.endswith(".so") for f in os.listdir(d)))


def need_ref(kind="int", wave64=False):
    """The compiled reference for a parity test: the library, or -- only on a machine that has neither the reference sources
    nor a shipped build -- None after saying what is being skipped.  A missing library where one is expected is a FAILURE."""
    if have_ref(kind, wave64):
        return ref(kind, wave64)
    name = f"libofdis_ref_{kind}{'_w64' if wave64 else ''}.so"
    if ref_expected():
        raise AssertionError(f"oracle/_ref/{name} is missing although the reference build is expected here (run `make -C oracle`)")
    print(f"[oracle] comparison against the compiled reference ({name}) SKIPPED: neither /root/reference nor oracle/_ref exists")
    return None


def epe_stats(a, b):
    """mean / max end-point error and fraction of pixels above 1e-3 px between two (h,w,2) flows."""
    d = np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum(-1))
    return float(d.mean()), float(d.max()), float((d > 1e-3).mean())
