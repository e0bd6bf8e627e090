#!/bin/bash
# A/B of RTX_BIN_G (cells per axis of the light-space sort grid; unset = the context's own choice)
cd $GRAFT_REPO_ROOT
for g in ${GS:-256 512}; do
  echo -n "binG=$g  "
  RTX_BIN_G=$g timeout -k 10 280 python3 -c "
import sys, os, tempfile, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import rtxamd as rt
wl = rt.scenegen.WORKLOADS['${WL:-c5}']
p = os.path.join(tempfile.gettempdir(), 'abg.txt'); wl['writer'](p); sc = rt.load_scene(p)
ctx = rt.Context(sc, rt.Params.make(wl['width'], wl['height'], wl['max_depth']))
ms = []
for _ in range(4):
    ctx.render(); ms.append(ctx.stats().ms_total)
print('frame %.2f ms' % min(ms))
" || exit 1
done
