"""-m gpu: rtx_render_rgb8_noalloc (the entry point that makes no device allocation: scene records, ray queues
and both images live in the library's static 64 MiB device array) against rtx_render_rgb8, through the C-ABI.
Two calls in a row on the same GPU, the second with a smaller input: the array keeps the first call's contents,
so anything the path forgets to clear shows up in the second frame.  Collected last (file name): a failure here
must not hide the other GPU tests from `pytest -x`."""
import os

import numpy as np
import pytest

from conftest import SCENES

pytestmark = pytest.mark.gpu


def _same(rt, sc, p):
    import torch
    want, sw = rt.render_rgb8(sc, p)
    rt.host.release_cached_context()          # (the one-shot call's kept context: free it, so what is compared below is this call alone)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    got, sg = rt.render_rgb8_noalloc(sc, p)
    free1 = torch.cuda.mem_get_info()[0]
    # "allocates nothing": the device's free memory is what it was (a frame's buffers would be tens of MiB; the slack is for
    # the runtime's own kernel-argument pools)
    assert free0 - free1 < (4 << 20), "rtx_render_rgb8_noalloc took %.1f MiB of device memory" % ((free0 - free1) / 2**20)
    assert got.shape == want.shape
    assert np.array_equal(got, want), "%d bytes differ" % int((got != want).sum())
    assert (sg.rays_primary, sg.rays_secondary, sg.rays_shadow, sg.hits) == (sw.rays_primary, sw.rays_secondary, sw.rays_shadow, sw.hits)
    return sg


def test_noalloc_matches_render_rgb8_twice_second_smaller(rt, gpu, tmp_path):
    # first call: 2,000 primitives (affine spheres with their (light, cell) table, reflective and refractive, so
    # fan-out 2 queues, so the capacities come from the calibration pass), 960x540, depth 4: several row chunks of the 64 MiB block
    big = tmp_path / "big.txt"
    rt.scenegen.write_scene(str(big), 1000, 1000, rt.scenegen.lights_c4(), seed=21, frac_affine=0.3,
                            frac_reflective=0.5, frac_refractive=0.2)
    s1 = _same(rt, rt.load_scene(str(big)), rt.Params.make(960, 540, 3))
    assert s1.chunks > 1 and s1.queue_overflow == 0
    # second call, smaller input in every respect: fewer primitives, fewer lights, smaller frame
    s2 = _same(rt, rt.load_scene(os.path.join(SCENES, "probe.txt")), rt.Params.make(96, 64, 4))
    assert s2.chunks == 1
    # and a third with bands (a rank's rows only) and an odd width
    _same(rt, rt.load_scene(os.path.join(SCENES, "room.txt")), rt.Params.make(201, 77, 4, nranks=3, rank=1, band_rows=5))


def test_noalloc_rejects_what_does_not_fit(rt, gpu):
    sc = rt.load_scene(os.path.join(SCENES, "probe.txt"))
    with pytest.raises(rt.RtxError) as e:
        rt.render_rgb8_noalloc(sc, rt.Params.make(3840, 2160, 4))   # 99.5 MB float image alone exceeds the block
    assert e.value.code == -6 and "static scratch" in str(e.value)
    # the refusal leaves the entry point usable
    _same(rt, sc, rt.Params.make(64, 48, 2))
