#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container: it needs oracle/_ref/rt_ref, which oracle/Makefile (`make -C
oracle ref`) compiles from /root/reference/raytracing.cpp where it lies (patches documented
there).  What is committed is data only: the scene files used as inputs (the reference's five
bundled inp*.txt, copied verbatim into tests/golden/bundled/, plus this repo's own HEAD-grammar
scenes under tests/scenes/) and the reference's outputs for them:

  <name>_<W>x<H>_d<D>.ppm   what `rt_ref scene out.ppm` wrote (render2Dantialiasing, 3x3 AA)
  <name>_<W>x<H>_d<D>.f32   raw 1-spp float image of render2d() before quantisation (RT_RAW)
  index.json                md5 of every fixture, md5 of the full 640x480 PPMs (too big to
                            commit), and the reference's ray counters for each run
"""
import glob
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

REFERENCE = "/root/reference"
SMALL = [(64, 48, 4), (64, 48, 1)]
MEDIUM = [(160, 120, 4)]


def md5(b):
    return hashlib.md5(b).hexdigest()


def main():
    assert O.have_ref(), "build oracle/_ref first: make -C oracle ref"
    os.makedirs(os.path.join(HERE, "bundled"), exist_ok=True)
    scenes = []
    for p in sorted(glob.glob(os.path.join(REFERENCE, "inp*.txt"))):
        dst = os.path.join(HERE, "bundled", os.path.basename(p))
        shutil.copyfile(p, dst)
        scenes.append(("bundled/" + os.path.basename(p), dst, SMALL))
    for p in sorted(glob.glob(os.path.join(ROOT, "tests", "scenes", "*.txt"))):
        scenes.append(("scenes/" + os.path.basename(p), p, SMALL + MEDIUM))
    index = {"generator": "tests/golden/make_golden.py", "reference_binary": "oracle/_ref/rt_ref", "runs": []}
    for rel, path, sizes in scenes:
        name = os.path.splitext(os.path.basename(path))[0]
        for (w, h, d) in sizes:
            ppm, counts = O.run_ref(path, w, h, d)
            raw, _ = O.run_ref(path, w, h, d, raw=True)
            stem = "%s_%dx%d_d%d" % (name, w, h, d)
            with open(os.path.join(HERE, stem + ".ppm"), "wb") as f:
                f.write(b"P6\n%d %d 255\n" % (w, h))
                f.write(ppm.tobytes())
            raw.tofile(os.path.join(HERE, stem + ".f32"))
            index["runs"].append(dict(scene=rel, width=w, height=h, max_depth=d, ppm=stem + ".ppm", raw=stem + ".f32",
                                      ppm_md5=md5(open(os.path.join(HERE, stem + ".ppm"), "rb").read()),
                                      raw_md5=md5(raw.tobytes()), trace=counts.get("trace"), shadow=counts.get("shadow")))
        # full-size run: md5 only
        ppm, counts = O.run_ref(path, 640, 480, 4)
        full = b"P6\n640 480 255\n" + ppm.tobytes()
        index["runs"].append(dict(scene=rel, width=640, height=480, max_depth=4, ppm=None, raw=None,
                                  ppm_md5=md5(full), trace=counts.get("trace"), shadow=counts.get("shadow")))
        if rel == "bundled/inpaffine.txt":
            # BASELINE.json configs[2]: the affine scene at 1920x1080 (md5 only)
            ppm, counts = O.run_ref(path, 1920, 1080, 4)
            index["runs"].append(dict(scene=rel, width=1920, height=1080, max_depth=4, ppm=None, raw=None,
                                      ppm_md5=md5(b"P6\n1920 1080 255\n" + ppm.tobytes()), trace=counts.get("trace"), shadow=counts.get("shadow")))
        print(rel, "done")
    # BASELINE.json configs[3] and [4]: the benchmark scenes themselves (ray-tracing_amd/scenegen.py, unmodified:
    # 5,000 + 5,000 primitives / 100,000 triangles) on frames one CPU core finishes in minutes.  The scene files are
    # too big to commit (2 / 20 MB); they are regenerated from the seeded generator and pinned by their md5.
    import tempfile
    import rtxamd as rt
    for wl, (w, h) in (("c4", (48, 27)), ("c5", (24, 13))):
        W = rt.scenegen.WORKLOADS[wl]
        path = os.path.join(tempfile.gettempdir(), "golden_%s.txt" % wl)
        W["writer"](path)
        d = W["max_depth"]
        ppm, counts = O.run_ref(path, w, h, d)
        raw, _ = O.run_ref(path, w, h, d, raw=True)
        stem = "%s_%dx%d_d%d" % (wl, w, h, d)
        with open(os.path.join(HERE, stem + ".ppm"), "wb") as f:
            f.write(b"P6\n%d %d 255\n" % (w, h))
            f.write(ppm.tobytes())
        raw.tofile(os.path.join(HERE, stem + ".f32"))
        index["runs"].append(dict(scene="scenegen:" + wl, scene_md5=md5(open(path, "rb").read()), width=w, height=h, max_depth=d,
                                  ppm=stem + ".ppm", raw=stem + ".f32", ppm_md5=md5(open(os.path.join(HERE, stem + ".ppm"), "rb").read()),
                                  raw_md5=md5(raw.tobytes()), trace=counts.get("trace"), shadow=counts.get("shadow")))
        os.remove(path)
        print(wl, "done")
    with open(os.path.join(HERE, "index.json"), "w") as f:
        json.dump(index, f, indent=1)


if __name__ == "__main__":
    main()
