#!/bin/bash
# A/B of RTX_BIN_G on one rank's share of the C5 frame (N = 2, 4; bands of 30 rows)
cd $GRAFT_REPO_ROOT
for N in ${NS:-2 4}; do
for g in ${GS:-256 512}; do
  echo -n "N=$N binG=$g  "
  RTX_BIN_G=$g timeout -k 10 280 python3 -c "
import sys, os, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import rtxamd as rt
wl = rt.scenegen.WORKLOADS['c5']
p = os.path.join(tempfile.gettempdir(), 'abg.txt'); wl['writer'](p); sc = rt.load_scene(p)
ctx = rt.Context(sc, rt.Params.make(wl['width'], wl['height'], wl['max_depth'], nranks=$N, rank=1, band_rows=30))
ms = []
for _ in range(4):
    ctx.render(); ms.append(ctx.stats().ms_total)
print('frame %.2f ms' % min(ms))
" || exit 1
done; done
