#!/bin/bash
# A/B of library builds on the C5 workload (8K, 100k triangles, 16 lights, depth 7): frame times and context creation time
cd $GRAFT_REPO_ROOT
for L in "$@"; do
  if [ "$L" = "-" ]; then unset RTX_LIB; echo -n "default   "; else export RTX_LIB=$GRAFT_REPO_ROOT/$L; echo -n "$L  "; fi
  timeout -k 10 280 python3 -c "
import sys, os, tempfile, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import rtxamd as rt
wl = rt.scenegen.WORKLOADS['c5']
p = os.path.join(tempfile.gettempdir(), 'abc5.txt'); wl['writer'](p); sc = rt.load_scene(p)
t0 = time.perf_counter(); ctx = rt.Context(sc, rt.Params.make(wl['width'], wl['height'], wl['max_depth'])); t1 = time.perf_counter()
ms = []
for _ in range(4):
    ctx.render(); ms.append(ctx.stats().ms_total)
print('create %.2f s, frame %.1f ms' % (t1 - t0, min(ms)))
" || exit 1
done
