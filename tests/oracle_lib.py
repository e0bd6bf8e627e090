"""ctypes access to the CPU oracle (oracle/librt_oracle.so) -- test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

import rtxamd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "librt_oracle.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "rt_ref")
REF_BASE = os.path.join(ORACLE_DIR, "_ref", "rt_base")


class Counts(C.Structure):
    _fields_ = [("trace_calls", C.c_ulonglong), ("shadow_rays", C.c_ulonglong), ("hits", C.c_ulonglong)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "librt_oracle.so"])
        L = C.CDLL(ORACLE_SO)
        SV = C.POINTER(rt.host.SceneView)
        L.orc_render_f32.restype = C.c_int
        L.orc_render_f32.argtypes = [SV, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(Counts)]
        L.orc_render_f32_rect.restype = C.c_int
        L.orc_render_f32_rect.argtypes = [SV, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(Counts)]
        L.orc_render_rgb8.restype = C.c_int
        L.orc_render_rgb8.argtypes = [SV, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(Counts)]
        L.orc_filter_quantise.restype = None
        L.orc_filter_quantise.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_quantise.restype = C.c_ubyte
        L.orc_quantise.argtypes = [C.c_float]
        L.orc_inverse_matrix.restype = rt.host.XMatrix
        L.orc_inverse_matrix.argtypes = [rt.host.XMatrix, C.POINTER(C.c_int)]
        L.orc_intersect_sphere.restype = C.c_int
        L.orc_intersect_sphere.argtypes = [rt.host.Vec3, rt.host.Vec3, C.POINTER(rt.host.Sphere),
                                           C.POINTER(rt.host.Vec3), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_intersect_triangle.restype = C.c_int
        L.orc_intersect_triangle.argtypes = [rt.host.Vec3, rt.host.Vec3, C.POINTER(rt.host.Triangle),
                                             C.POINTER(rt.host.Vec3), C.POINTER(C.c_float)]
        L.orc_set_flags.restype = None
        L.orc_set_flags.argtypes = [C.c_int]
        L.orc_accept_matrix.restype = None
        L.orc_accept_matrix.argtypes = [SV, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


TRACE_PLANES, TRI_MT = 1, 2


class flags:
    """with oracle.flags(oracle.TRACE_PLANES): ...  -- the build's opt-in extensions, off again afterwards."""

    def __init__(self, f):
        self.f = f

    def __enter__(self):
        lib().orc_set_flags(self.f)

    def __exit__(self, *a):
        lib().orc_set_flags(0)


def render_f32(scene, width, height, max_depth=4, row0=0, row1=None, col0=0, col1=None):
    row1 = height if row1 is None else row1
    col1 = width if col1 is None else col1
    out = np.zeros((row1 - row0, col1 - col0, 3), dtype=np.float32)
    cnt = Counts()
    rc = lib().orc_render_f32_rect(scene.ptr, width, height, max_depth, row0, row1, col0, col1, out.ctypes.data, C.byref(cnt))
    assert rc == 0
    return out, cnt


def render_rgb8(scene, width, height, max_depth=4, literal_9x=False):
    out = np.zeros((height, width, 3), dtype=np.uint8)
    cnt = Counts()
    rc = lib().orc_render_rgb8(scene.ptr, width, height, max_depth, int(literal_9x), out.ctypes.data, C.byref(cnt))
    assert rc == 0
    return out, cnt


def filter_quantise(img):
    h, w, _ = img.shape
    img = np.ascontiguousarray(img, dtype=np.float32)
    out = np.zeros((h, w, 3), dtype=np.uint8)
    lib().orc_filter_quantise(img.ctypes.data, w, h, out.ctypes.data)
    return out


def accept_matrix(scene, o, d, shadow=False):
    o = np.ascontiguousarray(o, dtype=np.float32)
    d = np.ascontiguousarray(d, dtype=np.float32)
    n = o.shape[0]
    out = np.zeros((n, scene.view.nspheres + scene.view.ntriangles), dtype=np.uint8)  # spheres and triangles only
    lib().orc_accept_matrix(scene.ptr, n, o.ctypes.data, d.ctypes.data, int(shadow), out.ctypes.data)
    return out


def read_ppm(path):
    with open(path, "rb") as f:
        data = f.read()
    assert data[:3] == b"P6\n"
    nl = data.index(b"\n", 3)
    w, h, mx = (int(t) for t in data[3:nl].split())
    assert mx == 255
    return np.frombuffer(data[nl + 1:], dtype=np.uint8).reshape(h, w, 3).copy()


def have_ref():
    return os.path.exists(REF_BIN)


def run_ref(scene_path, width, height, max_depth=4, raw=False, tmpdir="/tmp"):
    """Run the real reference (oracle/_ref/rt_ref, build container only)."""
    env = dict(os.environ, RT_WIDTH=str(width), RT_HEIGHT=str(height), RT_MAX_DEPTH=str(max_depth))
    out = os.path.join(tmpdir, "ref_%d.ppm" % os.getpid())
    rawp = os.path.join(tmpdir, "ref_%d.f32" % os.getpid())
    if raw:
        env["RT_RAW"] = rawp
    p = subprocess.run([REF_BIN, scene_path, out], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True)
    counts = {}
    for tok in p.stderr.decode().split():
        if "=" in tok:
            k, v = tok.split("=")
            counts[k] = int(v)
    if raw:
        img = np.fromfile(rawp, dtype=np.float32).reshape(height, width, 3)
        os.remove(rawp)
    else:
        img = read_ppm(out)
    os.remove(out)
    return img, counts
