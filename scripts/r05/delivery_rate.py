"""round 5: throughput over time from a `bench.py --worker --dump-deliveries FILE` stream: scenes delivered per time bin.
usage: python3 scripts/r05/delivery_rate.py FILE [batch=8] [bin_s=0.25]"""
import sys

path = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
width = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
rows = [(int(a), float(b)) for a, b in (l.split() for l in open(path))]
t_end = rows[-1][1]
bins = [0] * (int(t_end / width) + 1)
for _, t in rows:
    bins[int(t / width)] += 1
print('%d steps delivered in %.3f s; scenes/s per %.2f s bin (last bin partial):' % (len(rows), t_end, width))
print(' '.join('%d' % round(c * batch / width) for c in bins))
