"""ctypes mirror of include/rtx.h.  Struct layouts follow the reference's PODs
(raytracing.cpp:15-68); function names and argument meaning follow the header one to one."""
import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTX_LIB") or os.path.join(_DIR, "librtx.so")   # RTX_LIB: A/B builds of the library (tools/dbg)
RTX_MAX_LEVELS = 32
F_TIME_KERNELS = 1
F_COUNT_TESTS = 2
F_EXACT_ONLY = 4
F_NO_HALO = 8
F_PACKET_STATS = 16
F_SERIAL = 32
F_COUNT_EXEC = 64
F_TRACE_PLANES = 128
F_TRI_MOLLER_TRUMBORE = 256
F_WORST_CASE_QUEUES = 512
F_NO_CACHE = 1024
F_MULTI_ON_ONE_DEVICE = 2048
F_SHADOW_TABLES = 4096
F_SHADOW_PACKETS = 8192
RTX_EXEC_KINDS = 16
EXEC_KERNELS = ("k_closest_f", "k_closest_lanes", "k_shadow_pass")
EXEC_KINDS = ("packet", "cone", "filt", "afilt", "strip", "sph", "sphx", "tri_early", "tri_cull", "tri_full", "poly", "packets_built", "rays", "tri_edge")


class RtxError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = lib().rtx_strerror(code).decode()
        hip = lib().rtx_last_hip_error().decode()
        super().__init__(f"{what}: rtx error {code} ({msg})" + (f" [{hip}]" if hip else ""))


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class XMatrix(C.Structure):
    _fields_ = [("a", Vec3), ("b", Vec3), ("c", Vec3), ("d", Vec3)]


class Sphere(C.Structure):
    _fields_ = [("center", Vec3), ("istransformed", C.c_int32), ("radius", C.c_float),
                ("reflectivity", C.c_float), ("transparency", C.c_float),
                ("refractive_index", C.c_float), ("specexpo", C.c_float),
                ("color", Vec3), ("speccoeff", Vec3), ("diffcolor", Vec3), ("mat", XMatrix)]


class Plane(C.Structure):
    _fields_ = [("normal", Vec3), ("color", Vec3), ("d", C.c_float), ("reflectivity", C.c_float),
                ("transparency", C.c_float), ("refractive_index", C.c_float)]


class Triangle(C.Structure):
    _fields_ = [("u", Vec3), ("v", Vec3), ("w", Vec3), ("color", Vec3), ("speccoeff", Vec3),
                ("diffcolor", Vec3), ("reflectivity", C.c_float), ("transparency", C.c_float),
                ("refractive_index", C.c_float), ("specexpo", C.c_float)]


class Light(C.Structure):
    _fields_ = [("point", Vec3), ("color", Vec3)]


class Polygon(C.Structure):
    _fields_ = [("numvertices", C.c_int32), ("first_vertex", C.c_int32), ("color", Vec3), ("speccoeff", Vec3),
                ("diffcolor", Vec3), ("normal", Vec3), ("centerPoint", Vec3), ("reflectivity", C.c_float),
                ("transparency", C.c_float), ("refractive_index", C.c_float), ("specexpo", C.c_float)]


class SceneView(C.Structure):
    _fields_ = [("spheres", C.POINTER(Sphere)), ("nspheres", C.c_int32),
                ("triangles", C.POINTER(Triangle)), ("ntriangles", C.c_int32),
                ("planes", C.POINTER(Plane)), ("nplanes", C.c_int32),
                ("lights", C.POINTER(Light)), ("nlights", C.c_int32),
                ("camera", Vec3), ("vrp", Vec3), ("vpn", Vec3), ("up", Vec3), ("ambient", Vec3),
                ("polygons", C.POINTER(Polygon)), ("npolygons", C.c_int32),
                ("polygon_vertices", C.POINTER(Vec3)), ("npolygon_vertices", C.c_int32),
                ("npolygons_dropped", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_depth", C.c_int32),
                ("nranks", C.c_int32), ("rank", C.c_int32), ("band_rows", C.c_int32),
                ("device", C.c_int32), ("flags", C.c_uint32), ("workspace_bytes", C.c_uint64)]

    @classmethod
    def make(cls, width=640, height=480, max_depth=4, nranks=1, rank=0, band_rows=0, device=-1,
             workspace_bytes=0, flags=0):
        return cls(width, height, max_depth, nranks, rank, band_rows, device, flags, workspace_bytes)


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_secondary", C.c_uint64),
                ("rays_shadow", C.c_uint64), ("hits", C.c_uint64),
                ("rays_level", C.c_uint64 * RTX_MAX_LEVELS),
                ("tests_closest", C.c_uint64), ("tests_shadow_max", C.c_uint64),
                ("tests_closest_kind", C.c_uint64 * 4), ("tests_shadow_kind", C.c_uint64 * 4),
                ("singular_matrices", C.c_uint64), ("chunks", C.c_int32), ("levels", C.c_int32),
                ("ms_total", C.c_float), ("ms_closest", C.c_float), ("ms_shadow", C.c_float),
                ("packet_stats", C.c_uint64 * 8),
                ("launches_closest", C.c_int32), ("launches_shadow", C.c_int32),
                ("ms_closest_lanes", C.c_float), ("launches_closest_lanes", C.c_int32),
                ("exec_counts", (C.c_uint64 * RTX_EXEC_KINDS) * 3),
                ("shadow_form", C.c_int32), ("reserved0", C.c_int32), ("queue_overflow", C.c_uint64)]

    def exec_dict(self):
        """{kernel: {kind: count}} of the RTX_F_COUNT_EXEC counters."""
        return {kn: {kd: int(self.exec_counts[i][j]) for j, kd in enumerate(EXEC_KINDS)} for i, kn in enumerate(EXEC_KERNELS)}

    @property
    def rays_total(self):
        return self.rays_primary + self.rays_secondary + self.rays_shadow


assert C.sizeof(Vec3) == 12 and C.sizeof(XMatrix) == 48 and C.sizeof(Sphere) == 120
assert C.sizeof(Plane) == 40 and C.sizeof(Triangle) == 88 and C.sizeof(Light) == 24 and C.sizeof(Polygon) == 84

# every symbol include/rtx.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "rtx_abi_version": (C.c_uint32, []),
    "rtx_release_cached_context": (None, []),
    "rtx_light_cell": (C.c_int, [C.POINTER(C.c_float), C.c_int32]),
    "rtx_cell_cone": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_float)]),
    "rtx_strerror": (C.c_char_p, [C.c_int]),
    "rtx_device_count": (C.c_int, []),
    "rtx_last_hip_error": (C.c_char_p, []),
    "rtx_render_rgb8": (C.c_int, [C.POINTER(SceneView), C.POINTER(Params), C.c_int, C.c_void_p, C.POINTER(Stats)]),
    "rtx_render_f32": (C.c_int, [C.POINTER(SceneView), C.POINTER(Params), C.c_void_p, C.POINTER(Stats)]),
    "rtx_context_create": (C.c_int, [C.POINTER(SceneView), C.POINTER(Params), C.POINTER(C.c_void_p)]),
    "rtx_context_destroy": (None, [C.c_void_p]),
    "rtx_context_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "rtx_context_render": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtx_context_read_rgb8": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rtx_context_read_f32": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rtx_context_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "rtx_band_rows": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    "rtx_render_rgb8_noalloc": (C.c_int, [C.POINTER(SceneView), C.POINTER(Params), C.c_void_p, C.POINTER(Stats)]),
    "rtx_filter_records": (C.c_int, [C.POINTER(SceneView), C.c_void_p, C.c_int32, C.POINTER(C.c_float)]),
    "rtx_affine_shadow_records": (C.c_int, [C.POINTER(SceneView), C.c_void_p, C.c_int32]),
    "rtx_triangle_edge_records": (C.c_int, [C.POINTER(SceneView), C.c_void_p, C.c_int32]),
    "rtx_cluster_records": (C.c_int, [C.POINTER(SceneView), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "rtx_cone_tests": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "rtx_cluster_runs": (C.c_int, [C.POINTER(SceneView), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rtx_scene_load": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(SceneView))]),
    "rtx_scene_load_legacy": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(SceneView))]),
    "rtx_scene_free": (None, [C.POINTER(SceneView)]),
    "rtx_write_ppm": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_void_p]),
}

_lib = None


def lib():
    """Load librtx.so.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C ray-tracing_amd/csrc` "
                              "(or __graft_entry__.build()); there is no fallback path")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc, what):
    if rc < 0:
        raise RtxError(rc, what)
    return rc


def device_count():
    return lib().rtx_device_count()


class Scene:
    """A scene parsed by the kept C++ reader (rtx_scene_load)."""

    def __init__(self, path, legacy=False):
        self._p = C.POINTER(SceneView)()
        fn = lib().rtx_scene_load_legacy if legacy else lib().rtx_scene_load
        _check(fn(os.fsencode(path), C.byref(self._p)), f"rtx_scene_load({path})")
        self.path = path

    @property
    def view(self):
        return self._p.contents

    @property
    def ptr(self):
        return self._p

    @property
    def nprims(self):
        return self.view.nspheres + self.view.ntriangles + self.view.npolygons

    def close(self):
        if self._p:
            lib().rtx_scene_free(self._p)
            self._p = C.POINTER(SceneView)()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_scene(path, legacy=False):
    """legacy=True: the opt-in loader for the older grammars of the bundled scenes (rtx_scene_load_legacy)."""
    return Scene(path, legacy)


def band_rows(height, nranks, rank, band):
    n = _check(lib().rtx_band_rows(height, nranks, rank, band, None, 0), "rtx_band_rows")
    rows = np.zeros(n, dtype=np.int32)
    if n:
        _check(lib().rtx_band_rows(height, nranks, rank, band, rows.ctypes.data, n), "rtx_band_rows")
    return rows


def _nrows(params):
    return len(band_rows(params.height, params.nranks, params.rank, params.band_rows))


def render_rgb8(scene, params, ngpus=1):
    rows = params.height if params.nranks <= 1 else _nrows(params)
    out = np.zeros((rows, params.width, 3), dtype=np.uint8)
    st = Stats()
    _check(lib().rtx_render_rgb8(scene.ptr, C.byref(params), ngpus, out.ctypes.data, C.byref(st)), "rtx_render_rgb8")
    return out, st


def release_cached_context():
    """rtx_release_cached_context: free the context the one-shot entry points keep between calls."""
    lib().rtx_release_cached_context()


def render_rgb8_noalloc(scene, params):
    """rtx_render_rgb8_noalloc: the one-shot render out of the library's static device scratch (no hipMalloc)."""
    rows = params.height if params.nranks <= 1 else _nrows(params)
    out = np.zeros((rows, params.width, 3), dtype=np.uint8)
    st = Stats()
    _check(lib().rtx_render_rgb8_noalloc(scene.ptr, C.byref(params), out.ctypes.data, C.byref(st)), "rtx_render_rgb8_noalloc")
    return out, st


def render_f32(scene, params):
    rows = _nrows(params)
    out = np.zeros((rows, params.width, 3), dtype=np.float32)
    st = Stats()
    _check(lib().rtx_render_f32(scene.ptr, C.byref(params), out.ctypes.data, C.byref(st)), "rtx_render_f32")
    return out, st


def filter_records(scene):
    """(records[nprims,4], guard): the kernels' conservative bounding spheres (rtx_filter_records)."""
    n = _check(lib().rtx_filter_records(scene.ptr, None, 0, None), "rtx_filter_records")
    out = np.zeros((n, 4), dtype=np.float32)
    g = C.c_float()
    _check(lib().rtx_filter_records(scene.ptr, out.ctypes.data, n, C.byref(g)), "rtx_filter_records")
    return out, g.value


def triangle_edge_records(scene):
    """records[ntriangles,8] float32: a1.xyz, m, a2.xyz, Q (rtx_triangle_edge_records)."""
    out = np.zeros((scene.view.ntriangles, 8), dtype=np.float32)
    _check(lib().rtx_triangle_edge_records(scene.ptr, out.ctypes.data, scene.view.ntriangles), "rtx_triangle_edge_records")
    return out


def cluster_records(scene, which=0):
    """(cluster[ncl,4], sub[ncl,8,4], member[ncl,64,4], index[ncl,64]): the hierarchy the traversal kernels walk
    (rtx_cluster_records); which = 0 closest hit, 1 shadow rays."""
    n = _check(lib().rtx_cluster_records(scene.ptr, which, None, None, None, None, 0), "rtx_cluster_records")
    cl = np.zeros((n, 4), dtype=np.float32); sub = np.zeros((n, 8, 4), dtype=np.float32)
    mem = np.zeros((n, 64, 4), dtype=np.float32); idx = np.zeros((n, 64), dtype=np.int32)
    if n:
        _check(lib().rtx_cluster_records(scene.ptr, which, cl.ctypes.data, sub.ctypes.data, mem.ctypes.data, idx.ctypes.data, n),
               "rtx_cluster_records")
    return cl, sub, mem, idx


def cone_tests(origin, direction, sphere, lr):
    """(cone_slack, cone_slack_front, cone_slack_cull(lr)) of the per-ray traversal for one ray and one sphere { c, R } (rtx_cone_tests)."""
    o = np.ascontiguousarray(origin, dtype=np.float32); d = np.ascontiguousarray(direction, dtype=np.float32)
    b = np.ascontiguousarray(sphere, dtype=np.float32); out = np.zeros(3, dtype=np.float32)
    _check(lib().rtx_cone_tests(o.ctypes.data, d.ctypes.data, b.ctypes.data, float(np.float32(lr)), out.ctypes.data), "rtx_cone_tests")
    return out


def cluster_runs(scene, which=0):
    """(order[ncl], run[nruns,4], nfirst): the per-ray traversal's cluster order, the { c, R } spheres around its runs of 8
    clusters (affine-first part and the rest counted apart) and the number of affine-sphere clusters (rtx_cluster_runs)."""
    nf, nr = C.c_int32(0), C.c_int32(0)
    n = _check(lib().rtx_cluster_runs(scene.ptr, which, None, None, 0, C.byref(nf), C.byref(nr)), "rtx_cluster_runs")
    order = np.zeros(n, dtype=np.int32); run = np.zeros((nr.value, 4), dtype=np.float32)
    if n:
        _check(lib().rtx_cluster_runs(scene.ptr, which, order.ctypes.data, run.ctypes.data, max(n, nr.value), C.byref(nf), C.byref(nr)), "rtx_cluster_runs")
    return order, run, nf.value


def affine_shadow_records(scene):
    """records[naffine,16] float32 (index field viewed as int32 at column 13)."""
    n = _check(lib().rtx_affine_shadow_records(scene.ptr, None, 0), "rtx_affine_shadow_records")
    out = np.zeros((n, 16), dtype=np.float32)
    if n:
        _check(lib().rtx_affine_shadow_records(scene.ptr, out.ctypes.data, n), "rtx_affine_shadow_records")
    return out


def write_ppm(path, rgb):
    h, w, _ = rgb.shape
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    _check(lib().rtx_write_ppm(os.fsencode(path), w, h, rgb.ctypes.data), "rtx_write_ppm")


class Context:
    """Resident render context (rtx_context_*): scene and every queue stay in HBM."""

    def __init__(self, scene, params):
        self._c = C.c_void_p()
        self.params = params
        _check(lib().rtx_context_create(scene.ptr, C.byref(params), C.byref(self._c)), "rtx_context_create")
        n = _check(lib().rtx_context_rows(self._c, None, 0), "rtx_context_rows")
        self.rows = np.zeros(n, dtype=np.int32)
        if n:
            lib().rtx_context_rows(self._c, self.rows.ctypes.data, n)

    def render(self, stream=None, d_out=None):
        _check(lib().rtx_context_render(self._c, stream, d_out), "rtx_context_render")

    def read_rgb8(self):
        out = np.zeros((len(self.rows), self.params.width, 3), dtype=np.uint8)
        _check(lib().rtx_context_read_rgb8(self._c, out.ctypes.data), "rtx_context_read_rgb8")
        return out

    def read_f32(self):
        out = np.zeros((len(self.rows), self.params.width, 3), dtype=np.float32)
        _check(lib().rtx_context_read_f32(self._c, out.ctypes.data), "rtx_context_read_f32")
        return out

    def stats(self):
        st = Stats()
        _check(lib().rtx_context_stats(self._c, C.byref(st)), "rtx_context_stats")
        return st

    def close(self):
        if self._c:
            lib().rtx_context_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
