#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the render hot path on the synthetic 4K scene
(BASELINE.json configs[3]: 3840x2160, 5,000 spheres + 5,000 triangles, 4 lights, depth 5).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A "step" is one full frame: every rank traces its scanline bands (scene and
all ray queues already resident in HBM), then the 8-bit bands are gathered to rank 0 with an RCCL
gather (torch.distributed, backend nccl).  The frame is FIXED as N grows (strong scaling).
value = rays actually traced (primary + reflection/refraction + shadow) per second, whole job.

The JSON line also carries
  roofline     for the dominant kernel (k_shadow_pass, k_closest_lanes or k_closest_f, whichever took
               longest): ALGORITHMIC FLOPs of the ray x primitive tests the reference's loops run for
               the rays that kernel traces (20 / 65 / 76 / 13 FLOP per sphere / affine
               sphere / triangle / early-exit triangle test, counted by the exact counting kernels in
               a separate untimed frame) / its launch time, measured with HIP events on the kernel's
               own stream during the timed steps.  The bound is FP32 VALU, not HBM or MFMA.
  cpu_baseline the reference renderer itself (oracle/_ref/rt_ref, compiled from the reference's
               source in the build container) or, if that binary is absent, this repo's scalar
               C port (oracle/), timed on ONE host core on a small image of the same scene.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLOP_PER_TEST = (20.0, 65.0, 76.0, 13.0)  # sphere, affine sphere, triangle, triangle early exit
PEAK_FP32_TFLOPS = 157.3                  # MI355X_MICROARCH.md: peak FP32 vector (counts FMA as 2)
PEAK_FP32_NOFMA_TFLOPS = 78.65            # contraction is off for parity: one FLOP per lane-op


def cpu_baseline(scene_path, scene, max_depth, seconds_target=20.0):
    """Time the CPU path on one host core on a small image of the same scene file."""
    ref = os.path.join(ROOT, "oracle", "_ref", "rt_ref")
    w, h = 32, 18
    if os.path.exists(ref):
        # calibrate size: tiny run first
        def run(w, h):
            env = dict(os.environ, RT_WIDTH=str(w), RT_HEIGHT=str(h), RT_MAX_DEPTH=str(max_depth))
            out = os.path.join(tempfile.gettempdir(), "rtx_cpu_%d.ppm" % os.getpid())
            t0 = time.perf_counter()
            p = subprocess.run([ref, scene_path, out], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            os.path.exists(out) and os.remove(out)
            cnt = dict(tok.split("=") for tok in p.stderr.decode().split() if "=" in tok)
            return int(cnt.get("trace", 0)) + int(cnt.get("shadow", 0)), dt
        rays, dt = run(16, 9)
        scale = max(1.0, (seconds_target / max(dt, 1e-3)) ** 0.5)
        w, h = int(16 * scale), int(9 * scale)
        w, h = min(w, 96), min(h, 54)
        rays, dt = run(w, h)
        return dict(value=rays / dt / 1e6, unit="Mrays/s", cores=1, kind="reference",
                    sample="oracle/_ref/rt_ref (the reference's own source, g++ -O2 -ffp-contract=off), same scene file, "
                           "%dx%d image incl. its 9x antialias re-tracing, %d rays in %.1f s" % (w, h, rays, dt))
    import oracle_lib as O
    t0 = time.perf_counter()
    _, cnt = O.render_rgb8(scene, w, h, max_depth, literal_9x=True)
    dt = time.perf_counter() - t0
    rays = cnt.trace_calls + cnt.shadow_rays
    return dict(value=rays / dt / 1e6, unit="Mrays/s", cores=1, kind="port",
                sample="oracle/rt_oracle.c scalar port, same scene file, %dx%d image incl. 9x antialias re-tracing, "
                       "%d rays in %.1f s" % (w, h, rays, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4", choices=["c4", "c5"])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--band-rows", type=int, default=30)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-count", action="store_true", help="skip the untimed test-counting frame")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import rtxamd as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the render path has no CPU fallback")
    # rehearsal of the multi-rank path on a one-GPU box: RTX_BENCH_REHEARSAL=1 puts every rank on device 0 and
    # gathers over gloo (RCCL wants one device per rank); never set by the driver, numbers are meaningless then
    rehearsal = os.environ.get("RTX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    wl = rt.scenegen.WORKLOADS[args.workload]
    W = args.width or wl["width"]
    H = args.height or wl["height"]
    md = wl["max_depth"]
    scene_path = os.path.join(tempfile.gettempdir(), "rtx_bench_%s_%d.txt" % (args.workload, os.getpid()))
    wl["writer"](scene_path)
    scene = rt.load_scene(scene_path)

    def make_ctx(flags):
        p = rt.Params.make(W, H, md, nranks=world, rank=rank, band_rows=args.band_rows if world > 1 else 0,
                           device=local_rank, flags=flags)
        return rt.Context(scene, p)

    ctx = make_ctx(rt.host.F_TIME_KERNELS)
    nrows = len(ctx.rows)
    stream = torch.cuda.current_stream().cuda_stream
    import importlib
    bands = importlib.import_module("rtxamd_pkg.bands")
    bg = bands.BandGather(W, H, args.band_rows if world > 1 else 0, world, rank, dev, via_host=rehearsal and world > 1)
    assert len(bg.my_rows) == nrows

    def step():
        ctx.render(stream, bg.next_send().data_ptr())   # the band lands directly in the gather's send buffer
        if world > 1:
            # RCCL over xGMI, the only cross-device step: frame i's gather is on the wire while frame i+1 is
            # traced; rank 0 assembles frame i-1 here.  Every timed frame is gathered and assembled before sync() returns.
            bg.submit()

    def sync():
        if world > 1:
            bg.flush()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    st = ctx.stats()  # last frame: ray counts and per-kernel HIP-event times
    ust = st
    if world > 1:
        # value counts every ray of the frame ONCE: the halo rows a rank re-traces for the 3x3 filter are
        # real work in the timed region but not extra rays of the image (untimed accounting frame)
        uctx = make_ctx(rt.host.F_NO_HALO)
        uctx.render(stream, None)
        ust = uctx.stats()
        uctx.close()
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    rays = torch.tensor([float(ust.rays_primary), float(ust.rays_secondary), float(ust.rays_shadow)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    rays_frame = float(rays.sum().item())
    value = rays_frame * args.steps / dt / 1e6

    out = None
    if rank == 0:
        out = {
            "metric": "Mrays/sec (primary+secondary+shadow rays traced) at 4K depth 5" if args.workload == "c4"
                      else "Mrays/sec (primary+secondary+shadow rays traced) at 8K depth 8",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["name"] if not (args.width or args.height) else wl["name"] + " [resized %dx%d]" % (W, H),
                       "width": W, "height": H, "max_depth": md, "spheres": scene.view.nspheres,
                       "triangles": scene.view.ntriangles, "lights": scene.view.nlights,
                       "parallelism": "scanline bands of %d rows, round-robin over %d GPU(s), RCCL gather of rgb8" % (args.band_rows, world)
                                      if world > 1 else "1 GPU, whole frame",
                       "rays_per_frame": {"primary": float(rays[0].item()), "secondary": float(rays[1].item()),
                                          "shadow": float(rays[2].item())}},
        }

    # ---- roofline of the dominant kernel (rank 0's share of the frame) ----
    if rank == 0:
        kern = {"k_shadow_pass": (st.ms_shadow, st.launches_shadow),
                "k_closest_lanes": (st.ms_closest_lanes, st.launches_closest_lanes),
                "k_closest_f": (st.ms_closest, st.launches_closest)}
        dom = max(kern, key=lambda k: kern[k][0])
        ms_dom, n_dom = kern[dom]
        roof = {"bound": "valu-fp32", "kernel": dom, "achieved": None,
                "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                "peak_no_fma": PEAK_FP32_NOFMA_TFLOPS, "frac_no_fma": None,
                "launches_per_frame": n_dom, "avg_launch_ms": ms_dom / max(n_dom, 1),
                "ms_per_frame_by_kernel": {k: v[0] for k, v in kern.items()}, "ms_frame": st.ms_total,
                "note": "achieved = the REFERENCE algorithm's test FLOPs for the rays this kernel traces (every primitive in file "
                        "order, shadow rays leaving at their first blocker; counted by the exact kernels in an untimed frame) / "
                        "this kernel's summed launch time (HIP events on its own stream; launches of different streams overlap, "
                        "so the per-kernel times add up to more than ms_frame).  The kernel reaches the same answers "
                        "through a cluster hierarchy and executes a small part of those tests, hence frac > 1: see DESIGN.md 4.3 "
                        "for the executed-instruction counters."}
        if not args.no_count:
            cctx = make_ctx(rt.host.F_COUNT_TESTS)
            cctx.render(stream, None)
            cst = cctx.stats()
            if dom == "k_shadow_pass":
                kinds = [float(x) for x in cst.tests_shadow_kind]
            else:
                # the counting kernel sums over generations; closest-hit tests every primitive for every ray, so a
                # generation's share is its share of the rays
                tot = float(st.rays_primary + st.rays_secondary)
                share = float(st.rays_primary if dom == "k_closest_f" else st.rays_secondary) / max(tot, 1.0)
                kinds = [float(x) * share for x in cst.tests_closest_kind]
            flops = sum(float(kinds[k]) * FLOP_PER_TEST[k] for k in range(4))
            ach = flops / (ms_dom * 1e-3) / 1e12
            roof.update(achieved=ach, frac=ach / PEAK_FP32_TFLOPS, frac_no_fma=ach / PEAK_FP32_NOFMA_TFLOPS,
                        flops_per_frame=flops, tests_per_frame=[float(k) for k in kinds])
            cctx.close()
        # HBM traffic of that kernel: not collectable inside this run (counter passes are separate rocprofv3 runs,
        # tools/pmc_passes.sh); taken from the newest committed passes over the same frame, per launch
        if args.workload == "c4" and not (args.width or args.height) and world == 1:
            import csv, glob
            files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_c4.csv")))
            for row in (csv.DictReader(open(files[-1])) if files else []):
                if row["kernel"].replace("rtx::", "").split("<")[0] == dom and "FETCH_SIZE" in row and "WRITE_SIZE" in row:
                    launches = max(float(row["dispatches_per_pass"]), 1.0)
                    fetch_kb, write_kb = float(row["FETCH_SIZE"]) / launches, float(row["WRITE_SIZE"]) / launches
                    # the guide's gfx950 correction: FETCH_SIZE tallies wide (16 B / lane) reads at half their bytes; the
                    # kernel mixes widths, so the doubled figure is an upper bound
                    roof["traffic"] = (2.0 * fetch_kb + write_kb) * 1024.0
                    roof["traffic_note"] = ("bytes per launch from %s (FETCH_SIZE x 2 + WRITE_SIZE, KB): an upper bound, against "
                                            "%.3g algorithmic FLOPs per launch" % (os.path.basename(files[-1]), roof.get("flops_per_frame", 0.0) / max(n_dom, 1)))
        # executed side, whole frame: wave-level VALU instructions (same counter passes) / the live frame time, against
        # the issue rate the nominal FP32 peak implies (157.3 TFLOP/s = 2 FLOPs x 64 lanes x 1,229 G wave-instructions/s)
        if args.workload == "c4" and not (args.width or args.height) and world == 1 and files:
            tot, frames = 0.0, None
            for row in csv.DictReader(open(files[-1])):
                name = row["kernel"].replace("rtx::", "")
                if name.startswith("k_affine_cells"):
                    continue                      # context creation, not the frame
                if name.startswith("k_raygen"):
                    frames = float(row["dispatches_per_pass"])   # one launch per frame
                tot += float(row.get("SQ_INSTS_VALU", 0.0) or 0.0)
            if frames and tot:
                per_frame = tot / frames
                rate = per_frame / (st.ms_total * 1e-3) / 1e9
                peak_issue = PEAK_FP32_TFLOPS * 1e12 / 128.0 / 1e9
                roof["valu_issue"] = {"wave_instr_per_frame": per_frame, "G_instr_per_s": rate, "peak_G_instr_per_s": peak_issue,
                                      "frac": rate / peak_issue, "source": os.path.basename(files[-1]),
                                      "note": "SQ_INSTS_VALU summed over the frame's kernels (counter passes over the same frame, "
                                              "committed) / this run's frame time; active lanes per instruction: DESIGN.md 4.5"}
        out["roofline"] = roof
    # ---- end to end at the C-ABI: untimed extras, N = 1 only ----
    if rank == 0 and world == 1:
        import numpy as np
        t0 = time.perf_counter()
        ctx.render(stream, None)
        img = ctx.read_rgb8()                      # waits for the frame, copies 3 B / pixel to the host
        t1 = time.perf_counter()
        rgb, _ = rt.render_rgb8(scene, rt.Params.make(W, H, md, device=local_rank))  # create + upload + render + D2H + destroy
        t2 = time.perf_counter()
        assert np.array_equal(img, rgb)
        out["end_to_end"] = {"resident_render_plus_d2h_ms": (t1 - t0) * 1e3, "one_shot_rtx_render_rgb8_ms": (t2 - t1) * 1e3,
                             "note": "value above is the resident render with the frame left in HBM; these two add the "
                                     "24.9 MB device-to-host copy, and context creation (scene upload, queue allocation) + teardown"}
    if world > 1:
        dist.barrier()
    if rank == 0:
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene_path, scene, md)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    scene.close()
    try:
        os.remove(scene_path)
    except OSError:
        pass


if __name__ == "__main__":
    main()
