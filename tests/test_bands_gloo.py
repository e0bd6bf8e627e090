"""The N > 1 path on CPU: two processes, gloo backend.  Each rank owns interleaved scanline
bands, 'renders' them (the oracle stands in for the GPU here: this test is about the band
arithmetic, the halo and the gather/assemble step, which are device-independent), and rank 0
assembles the frame with the same BandGather object bench.py uses over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, SCENES


def _worker(rank, world, port, W, H, band, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib
    import rtxamd as rt
    import oracle_lib as O
    bands = importlib.import_module("rtxamd_pkg.bands")
    sc = rt.load_scene(os.path.join(SCENES, "probe.txt"))
    bg = bands.BandGather(W, H, band, world, rank, torch.device("cpu"))
    rows = bg.my_rows
    # what a rank does on the GPU: trace own rows plus a one-row halo at 1 spp, then filter.
    need = sorted({y for r in rows for y in (r - 1, r, r + 1) if 0 <= y < H})
    img = np.zeros((H, W, 3), dtype=np.float32)
    for y in need:
        img[y] = O.render_f32(sc, W, H, 4, y, y + 1)[0][0]
    full = O.filter_quantise(img)          # rows outside `need` are garbage and never read below
    mine = np.ascontiguousarray(full[rows])
    bg.send[: mine.size] = torch.from_numpy(mine.reshape(-1))
    bg.gather()
    if rank == 0:
        frame = bg.assemble().numpy()
        np.save(out_path, frame)
    # the stream-of-frames path bench.py drives (submit / flush, two buffers in flight): frame i = the band
    # with i added to every byte; each must come back whole and in order
    got = []
    for i in range(3):
        bg.next_send()[: mine.size] = torch.from_numpy(((mine.astype(np.int32) + i) & 255).astype(np.uint8).reshape(-1))
        f = bg.submit()
        if rank == 0 and f is not None:
            got.append(f.numpy().copy())
    f = bg.flush()
    if rank == 0:
        got.append(f.numpy().copy())
        assert bg.flush() is None
        np.save(out_path + ".stream.npy", np.stack(got))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("band", [5, 0])
def test_two_rank_band_gather_equals_single_frame(rt, oracle, tmp_path, band):
    W, H, world = 48, 37, 2
    out = str(tmp_path / "frame.npy")
    port = 29500 + (os.getpid() % 2000) + (7 if band else 0)
    mp.spawn(_worker, args=(world, port, W, H, band, out), nprocs=world, join=True)
    frame = np.load(out)
    sc = rt.load_scene(os.path.join(SCENES, "probe.txt"))
    ref, _ = oracle.render_rgb8(sc, W, H, 4)
    assert np.array_equal(frame, ref)
    stream = np.load(out + ".stream.npy")
    assert stream.shape[0] == 3
    for i in range(3):
        assert np.array_equal(stream[i], ((ref.astype(np.int32) + i) & 255).astype(np.uint8)), i
