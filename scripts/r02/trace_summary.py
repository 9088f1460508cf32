"""summary of a rocprofv3 kernel trace of the pipelined bench run: per-kernel durations UNDER CONTENTION, workgroup-time
share (duration x min(workgroups, 256 CUs x occupancy guess) is not knowable from the trace: the share below is plain
duration x workgroups, a proxy for CU-time), concurrency (sum of durations / wall time of the busy window)"""
import csv
import sys
from collections import defaultdict

import gzip
rows = list(csv.DictReader(gzip.open(sys.argv[1], 'rt') if sys.argv[1].endswith('.gz') else open(sys.argv[1])))
def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    depth, out = 0, ''
    for ch in n:                       # drop the argument list, keep template arguments
        if ch == '(' and depth == 0:
            break
        depth += ch == '<'
        depth -= ch == '>'
        out += ch
    return out[:70]
t0 = min(int(r['Start_Timestamp']) for r in rows)
t1 = max(int(r['End_Timestamp']) for r in rows)
# steady window: the stretch of the trace with the most kernels in flight — 50 ms bins whose summed kernel time is at least
# half the maximum, the longest contiguous run of them (the process also runs eager warm-up / self-check / latency passes on
# one stream, with at most one kernel in flight, and idles between its legs)
nb = max(1, int((t1 - t0) / 50e6) + 1)
hist = [0.0] * nb
for r in rows:
    hist[min(nb - 1, int((int(r['Start_Timestamp']) - t0) / 50e6))] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
thr, best, cur = max(hist) * 0.5, (0, 0), None
for i, hcount in enumerate(hist + [0]):
    if hcount >= thr and cur is None:
        cur = i
    elif hcount < thr and cur is not None:
        if i - cur > best[1] - best[0]:
            best = (cur, i)
        cur = None
lo, hi = t0 + best[0] * 50e6 + 0.1 * (best[1] - best[0]) * 50e6, t0 + best[1] * 50e6 - 0.05 * (best[1] - best[0]) * 50e6
agg = defaultdict(lambda: [0, 0.0, 0.0, 1e30, 0.0])
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s < lo or e > hi:
        continue
    d = (e - s) * 1e-3
    wgs = 1
    for ax in ('X', 'Y', 'Z'):
        g, w = int(r.get('Grid_Size_' + ax, r.get('Grid_Size', 1)) or 1), int(r.get('Workgroup_Size_' + ax, r.get('Workgroup_Size', 1)) or 1)
        wgs *= max(1, g // max(w, 1))
    a = agg[short(r['Kernel_Name'])]
    a[0] += 1; a[1] += d; a[2] += d * wgs; a[3] = min(a[3], d); a[4] = max(a[4], d)
wall = (hi - lo) * 1e-3
tot = sum(a[1] for a in agg.values()); totw = sum(a[2] for a in agg.values())
print("window %.1f ms, %d launches, sum of kernel durations %.1f ms -> mean concurrency %.1f kernels in flight" % (wall * 1e-3, sum(a[0] for a in agg.values()), tot * 1e-3, tot / wall))
print("%-72s %7s %9s %9s %9s %7s %7s" % ("kernel", "count", "avg_us", "min_us", "max_us", "dur%", "wg*t%"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print("%-72s %7d %9.1f %9.1f %9.1f %6.1f%% %6.1f%%" % (k, a[0], a[1] / a[0], a[3], a[4], 100 * a[1] / tot, 100 * a[2] / totw))
