"""In-kernel timeline of the stick-first voxel chain from the stamp table scripts/cbench dumps (experiment build -DR2_EXP_TS):
python scripts/sticks_timeline.py gpurun_out/ts_sticks.bin"""
import sys

import numpy as np

ts = np.fromfile(sys.argv[1], np.uint64).reshape(16, 2048)
body = ts[:, :2047]
newest = body.max()
live = body >= newest - np.uint64(200000)   # 2 ms at 100 MHz: the last call only
t0 = body[live].min()
us = lambda a: (a.astype(np.int64) - int(t0)) * 0.01
names = {0: "count start", 1: "count end", 2: "scan start", 3: "scan end", 4: "scatter start", 5: "scatter: offsets ready", 6: "scatter end",
         7: "sort start", 8: "sort: entries loaded, range known", 9: "sort: bucket counts", 10: "sort: bucket bases", 11: "sort: placed",
         12: "sort: ranked", 13: "sort end"}
for ph in range(16):
    m = live[ph]
    if not m.any():
        continue
    v = np.sort(us(body[ph][m]))
    print("ph %2d %-36s n %4d  min %7.2f  p10 %7.2f  med %7.2f  p90 %7.2f  max %7.2f" % (ph, names.get(ph, ""), v.size, v[0], v[v.size // 10],
                                                                                     v[v.size // 2], v[(9 * v.size) // 10], v[-1]))
def life(a, b, what, sel=None):
    m = live[a] & live[b]
    if sel is not None:
        m = m & sel
    if not m.any():
        return
    d = np.sort(us(body[b][m]) - us(body[a][m]))
    print("%-44s n %4d  min %6.2f  med %6.2f  p90 %6.2f  max %6.2f us" % (what, d.size, d[0], d[d.size // 2], d[(9 * d.size) // 10], d[-1]))
life(0, 1, "count workgroup life")
life(2, 3, "scan workgroup life")
life(4, 5, "scatter: start -> offsets ready")
life(5, 6, "scatter: instance loop")
blk = np.arange(2047)
# big workgroups come first in the sort kernel's grid: they are the ones with the longest lives; split at the largest jump
for a, b, w in ((7, 8, "sort: start -> loaded"), (8, 9, "sort: bucket counts"), (9, 10, "sort: bases"), (10, 11, "sort: placement"),
                (11, 12, "sort: ranks"), (12, 13, "sort: output"), (7, 13, "sort workgroup life")):
    life(a, b, w)
m = live[7] & live[13]
if m.any():
    st, en = us(body[7]), us(body[13])
    idx = np.nonzero(m)[0]
    print("sort: first start %.2f, last end %.2f; workgroups with stamps %d (blocks %d..%d)" % (st[m].min(), en[m].max(), idx.size, idx[0], idx[-1]))
    # concurrency: how many sort workgroups are alive over time
    ev = np.concatenate([np.stack([st[m], np.ones(idx.size)], 1), np.stack([en[m], -np.ones(idx.size)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    alive = np.cumsum(ev[:, 1])
    for q in (0.1, 0.25, 0.5, 0.75, 0.9):
        k = int(q * (len(ev) - 1))
        print("  t %.2f us: %d workgroups alive" % (ev[k, 0], alive[k]))
