#!/usr/bin/env python
"""Overlap statistics of the last tokenize pass in a compact kernel timeline (tools/kernel_timeline.py output)."""
import collections
import sys

ev = []
for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kt_compact.csv"):
    n, q, s, a, b = l.rstrip("\n").split(";")
    ev.append((n, q, s, int(a), int(b)))
ev.sort(key=lambda e: e[3])
im = [i for i, e in enumerate(ev) if "im2col" in e[0]]
vq = [i for i, e in enumerate(ev) if "argmin" in e[0]]           # vq_argmin_kernel / vq_head_argmin_kernel: the last kernel of a sub-batch
import os
nparts = int(os.environ.get("NPARTS", "2"))
t0 = ev[im[-nparts]][3]
t1 = max(ev[i][4] for i in vq[-nparts:])
seg = [e for e in ev if e[3] >= t0 and e[4] <= t1]
print("pass wall ms %.2f  kernels %d" % ((t1 - t0) / 1e6, len(seg)))
tot, cnt = collections.Counter(), collections.Counter()
for e in seg:
    tot[e[0]] += e[4] - e[3]
    cnt[e[0]] += 1
for k, v in tot.most_common(12):
    print(k.ljust(44), str(cnt[k]).rjust(5), "total ms %7.2f  avg us %7.1f" % (v / 1e6, v / cnt[k] / 1e3))
pts = []
for e in seg:
    pts.append((e[3], 1))
    pts.append((e[4], -1))
pts.sort()
level, last, hist = 0, pts[0][0], collections.Counter()
for t, d in pts:
    hist[level] += t - last
    last = t
    level += d
print("time at concurrency level (ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
# GEMM-only occupancy: time during which at least one gemm kernel is running
g = sorted((e[3], e[4]) for e in seg if "gemm" in e[0])
busy, cs, ce = 0, g[0][0], g[0][1]
for s, e in g[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("some GEMM running: %.2f ms; two GEMMs overlapping: " % (busy / 1e6), end="")
pts = sorted([(s, 1) for s, _ in g] + [(e, -1) for _, e in g])
level, last, two = 0, pts[0][0], 0
for t, d in pts:
    if level >= 2:
        two += t - last
    last = t
    level += d
print("%.2f ms" % (two / 1e6))
print("streams:", collections.Counter((e[1], e[2]) for e in seg))
