"""ctypes mirror of `ofdis_params` (include/ofdis.h) and the reference's operating points.

Pure Python, loads nothing native: shared by the product binding (of_dis_amd.capi) and by the
test-only oracle wrapper (oracle/__init__.py).
"""
import ctypes as C
import math


class OfdisParams(C.Structure):
    """Field order == include/ofdis.h == the constructor arguments of OFC::OFClass
    (reference oflow.h:91-111)."""
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("imgpadding", C.c_int),
        ("sc_f", C.c_int), ("sc_l", C.c_int),
        ("max_iter", C.c_int), ("min_iter", C.c_int),
        ("dp_thresh", C.c_float), ("dr_thresh", C.c_float), ("res_thresh", C.c_float),
        ("p_samp_s", C.c_int), ("patove", C.c_float),
        ("usefbcon", C.c_int), ("costfct", C.c_int), ("noc", C.c_int), ("patnorm", C.c_int),
        ("usetvref", C.c_int),
        ("tv_alpha", C.c_float), ("tv_gamma", C.c_float), ("tv_delta", C.c_float),
        ("tv_innerit", C.c_int), ("tv_solverit", C.c_int), ("tv_sor", C.c_float),
        ("verbosity", C.c_int), ("selectmode", C.c_int),
    ]

    def copy(self, **kw):
        q = OfdisParams()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(self))
        for k, v in kw.items():
            setattr(q, k, v)
        return q

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}

    @property
    def nop(self):
        """flow channels: 2 (optical flow) or 1 (stereo depth, selectmode 2) -- the reference's op.nop"""
        return 1 if self.selectmode == 2 else 2

    # ---- derived geometry (reference oflow.cpp:91,138-157; patchgrid.cpp:42-48)
    def level_size(self, level):
        return self.width >> level, self.height >> level

    def plane_shape(self, level):
        w, h = self.level_size(level)
        return (h + 2 * self.imgpadding, w + 2 * self.imgpadding, self.noc)

    @property
    def steps(self):
        import numpy as np
        return max(1, int(math.floor(np.float32(self.p_samp_s) * (np.float32(1) - np.float32(self.patove)))))

    def grid(self, level):
        w, h = self.level_size(level)
        s = self.steps
        return int(math.ceil(w / s)), int(math.ceil(h / s))


def auto_first_scale(width_org, fratio=5, patchsz=8):
    """AutoFirstScaleSelect, reference run_dense.cpp:180-183."""
    return max(0, int(math.floor(math.log2((2.0 * width_org) / (float(fratio) * float(patchsz))))))


def padded_size(width_org, height_org, sc_f):
    """reference run_dense.cpp:298-305"""
    s = 1 << sc_f
    return width_org + (-width_org) % s, height_org + (-height_org) % s


def oppoint(op_point, width_org, height_org, noc=1, usetvref=None, verbosity=0):
    """Operating points 1-4 of the reference CLI (run_dense.cpp:225-265) for an image size."""
    p = OfdisParams()
    p.dp_thresh, p.dr_thresh, p.res_thresh = 0.05, 0.95, 0.0
    p.usefbcon, p.patnorm, p.costfct = 0, 1, 0
    p.tv_alpha, p.tv_gamma, p.tv_delta = 10.0, 10.0, 5.0
    p.tv_innerit, p.tv_solverit, p.tv_sor = 1, 3, 1.6
    p.verbosity = verbosity
    p.noc = noc
    table = {1: (8, 0.3, 2, 16, 0), 2: (8, 0.4, 2, 12, 1), 3: (12, 0.75, 4, 16, 1), 4: (12, 0.75, 5, 128, 1)}
    patchsz, poverl, dl, it, tv = table.get(op_point, table[2])
    p.p_samp_s, p.patove = patchsz, poverl
    p.sc_f = auto_first_scale(width_org, 5, patchsz)
    p.sc_l = max(p.sc_f - dl, 0)
    p.max_iter = p.min_iter = it
    p.usetvref = tv if usetvref is None else int(usetvref)
    p.imgpadding = patchsz
    p.width, p.height = padded_size(width_org, height_org, p.sc_f)
    return p
