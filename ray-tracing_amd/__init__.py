"""MI355X-native render path for the hot loop of ChandanY1011/ray-tracing.

The product is ``librtx.so`` (HIP kernels for gfx950 + C++ host side, built in-tree from
``csrc/`` by ``csrc/Makefile``); its C-ABI is ``include/rtx.h``.  This package is the thin
Python mirror used by tests and ``bench.py``: ctypes bindings (``host``) and the synthetic
scene generator (``scenegen``).  There is no Python or CPU fallback for the render path.

The directory name ``ray-tracing_amd`` is not a Python identifier; import it through the
repo-root shim ``rtxamd`` (``import rtxamd as rt``).
"""
from . import host, scenegen  # noqa: F401  (bands imports torch: import it explicitly)
from .host import (  # noqa: F401
    Context, Params, RtxError, Scene, Stats, band_rows, device_count, lib, load_scene, filter_records, affine_shadow_records, triangle_edge_records, cluster_records, cluster_runs, cone_tests,
    render_f32, render_rgb8, render_rgb8_noalloc, write_ppm,
)
