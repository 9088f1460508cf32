"""What runs beside what in the pipelined bench run (rocprofv3 --kernel-trace CSV): time-weighted histogram of the number of
kernels in flight, the share of the steady window during which a GEMM-family kernel is on the chip, per queue the busy
fraction and the gaps between consecutive kernels, and which kernels are on the chip while NO GEMM-family kernel is."""
import csv
import gzip
import sys
from collections import defaultdict

rows = list(csv.DictReader(gzip.open(sys.argv[1], 'rt') if sys.argv[1].endswith('.gz') else open(sys.argv[1])))
GEMM = ('linear_kernel', 'mlp_chain', 'mlp_group', 'mlp_rows')
ev = []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    ev.append((s, e, r['Kernel_Name'], r.get('Queue_Id', '?')))
ev.sort()
t0, t1 = ev[0][0], max(e for _, e, _, _ in ev)
# steady window: between the 25th and the 75th percentile of the FULL-SIZE passes of the trace (a pass starts with
# pack_points_kernel; the stream's passes have the largest grid — the one-batch latency legs and the eager self-check passes at
# the end of the run pack 8 scenes; with 80-scene passes those are a third of all launches, so "the middle half of the launches"
# (the first form of this script) reached into them)
packs = [(int(r['Start_Timestamp']), int(r.get('Grid_Size_X') or r.get('Grid_Size') or 0)) for r in rows if 'pack_points_kernel' in r['Kernel_Name']]
if packs:
    full = max(g for _, g in packs)
    starts = sorted(t for t, g in packs if g == full)
    lo, hi = starts[len(starts) // 4], starts[3 * len(starts) // 4]
else:
    lo, hi = ev[len(ev) // 4][0], ev[3 * len(ev) // 4][0]
pts = []
for s, e, name, q in ev:
    if e <= lo or s >= hi:
        continue
    g = any(k in name for k in GEMM)
    big = 'mlp_group_kernel<256, 512, 1024' in name
    pts.append((max(s, lo), 1, g, big, name))
    pts.append((min(e, hi), -1, g, big, name))
pts.sort(key=lambda p: (p[0], p[1]))
hist = defaultdict(float)
gemm_on = big_on = 0.0
n = ng = nb = 0
cur = defaultdict(int)
alone = defaultdict(float)
last = lo
for t, d, g, big, name in pts:
    dt = t - last
    if dt > 0:
        hist[min(n, 12)] += dt
        if ng > 0:
            gemm_on += dt
        else:
            for k, c in cur.items():
                if c > 0:
                    alone[k] += dt
        if nb > 0:
            big_on += dt
    last = t
    n += d; ng += d * g; nb += d * big
    short = name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:48]
    cur[short] += d
wall = hi - lo
print("steady window %.1f ms" % (wall * 1e-6))
print("kernels in flight (share of the window):", " ".join("%d:%.1f%%" % (k, 100 * v / wall) for k, v in sorted(hist.items())))
print("a GEMM-family kernel on the chip %.1f %% of the window; the head's wide group kernel %.1f %%" % (100 * gemm_on / wall, 100 * big_on / wall))
print("on the chip while NO GEMM-family kernel is (share of the window):")
for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:10]:
    print("   %-50s %.2f %%" % (k, 100 * v / wall))
# per queue
byq = defaultdict(list)
for s, e, name, q in ev:
    if s >= lo and e <= hi:
        byq[q].append((s, e))
print("%d queues; per queue: kernels, busy share, median / p90 gap between consecutive kernels (us)" % len(byq))
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    lst.sort()
    busy = sum(e - s for s, e in lst)
    gaps = sorted(max(0, lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1))
    if gaps:
        print("   queue %-6s %6d kernels  busy %.2f  gap median %.1f  p90 %.1f" % (q, len(lst), busy / wall, gaps[len(gaps) // 2] * 1e-3, gaps[int(0.9 * len(gaps))] * 1e-3))
