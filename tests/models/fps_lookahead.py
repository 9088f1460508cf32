"""Executable model of the multi-pick decision rule of the farthest point samplers (de6d_amd/csrc/fps_coop.hip, shipped:
the 32768 / 65536-point cooperative sampler; de6d_amd/csrc/fps_seq.hip, experiments build: one workgroup per 16384-point
scene; shared pieces in csrc/fps_multi.h).

TEST INFRASTRUCTURE: a discrete-event simulation of the protocol, checked against a plain sequential restatement of
farthest_point_sampling_kernel (sampling_gpu.cu:101-222: arg-max of the running min-distances, ties to the smallest
(bitrev_{log2 S}(k mod S), k)).  It exists to show that the decision rule is SOUND whatever the schedule of its agents is — a
sequencer that only ever sees published top-K records of the regions and never waits for a rescan unless a bound forces it to:
  * `greedy=True`: the lockstep schedule of the kernels (the sequencer decides until it is blocked, then every owner applies
    the round's picks and republishes: records are fresh at the start of a round);
  * `delay=D`: records reach the sequencer D decisions late (the asynchronous form built and dropped in round 4);
  * default: a random interleaving of owner steps and sequencer steps.

Agents
  owner(region): holds the exact min-distances of its points; consumes the published picks in order; a pick whose distance
    to the region's bounding box is >= the region's current maximum cannot change anything and is skipped; otherwise the
    region is rescanned and its record (tag a = picks applied, top-K candidates under the reference's order) republished.
  sequencer: holds, per region, the last record it accepted plus the CURRENT values of its candidates (updated with every
    pick it makes, by the same distance expression the scan uses).  Region state at decision r:
      X = best current candidate;  exact iff X >= the record's LAST candidate as it was, in the order (every other point of
      the region was ordered after that candidate when the record was made and min-distances only decrease), else unknown
      with that candidate's old value as bound.
    pick r = best exact X, provided every unknown region's bound is strictly below its value; else wait for records.

Round 6: `weights` — the score-weighted sampler (de6d_amd/csrc/fps_seq.hip: fps_seq_w_kernel; furthest_point_sampling_weights_kernel,
sampling_gpu.cu:419-540 with every weight >= 1e-12).  Values become SCORES (fp32(min-distance x weight)) in records, bounds and
comparisons; the owners still hold and update MIN-DISTANCES, their skip test still compares the pick's distance to the box with
the region's maximal min-distance, the sequencer tracks its candidates' min-distances and rebuilds their scores; the first pick
is the arg-max of the weights under the reference's order.
"""
import numpy as np


def tie_key(k, log2s):
    k = int(k)
    low = k & ((1 << log2s) - 1)
    rev = int('{:0{w}b}'.format(low, w=log2s)[::-1], 2) if log2s else 0
    return (rev << 32) | (k >> log2s)


def sqdist(p, s):
    """float32, the op order of d6_sqdist (products rounded one by one is enough for the model: both sides use it)"""
    d = (p - s).astype(np.float32)
    return (d[..., 2] * d[..., 2] + (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32)).astype(np.float32)


def opt_log2s(n):
    s = 1 << int(np.log2(n))
    return int(np.log2(min(max(s, 1), 1024)))


def first_pick(n, log2s, weights):
    """point 0 (D-FPS, sampling_gpu.cu:131-133) or the arg-max of the weights under the reference's order (S-FPS, :452-455)"""
    if weights is None:
        return 0
    w = np.asarray(weights, np.float32)
    cand = np.nonzero(w == w.max())[0]
    return int(min(cand, key=lambda k: tie_key(k, log2s)))


def fps_sequential(xyz, m, weights=None):
    """plain sequential restatement; weights: the score-weighted sampler (arg-max of fp32(min-distance x weight), weights >=
    1e-12: furthest_point_sampling_weights_kernel, sampling_gpu.cu:419-540)"""
    n = xyz.shape[0]
    log2s = opt_log2s(n)
    keys = np.array([tie_key(k, log2s) for k in range(n)], dtype=object)
    t = np.full(n, 1e10, np.float32)
    w = None if weights is None else np.asarray(weights, np.float32)
    picks = [first_pick(n, log2s, weights)]
    for _ in range(1, m):
        t = np.minimum(t, sqdist(xyz, xyz[picks[-1]]))
        sc = t if w is None else (t * w).astype(np.float32)
        v = sc.max()
        cand = np.nonzero(sc == v)[0]
        picks.append(int(min(cand, key=lambda k: keys[k])))
    return picks


class Region:
    def __init__(self, ids, xyz, log2s, hide=None, weights=None):
        self.ids = np.asarray(ids)
        self.p = xyz[self.ids]
        self.w = np.ones(len(ids), np.float32) if weights is None else np.asarray(weights, np.float32)[self.ids]
        self.t = np.full(len(ids), 1e10, np.float32)
        self.keys = [tie_key(k, log2s) for k in self.ids]
        # sq_hide_lane_duplicates (csrc/fps_multi.h): a point with the coordinates of a point of the same region that comes
        # EARLIER in the order may start at min-distance 0 — it is never an arg-max.  `hide` = the random generator that picks
        # which of those points are hidden (the kernels hide the duplicates that share a lane, a subset).
        if hide is not None:
            seen = {}
            for i in sorted(range(len(self.ids)), key=lambda i: self.keys[i]):
                c = tuple(self.p[i].tolist()) + (float(self.w[i]),)      # (weighted form: a duplicate carries the same weight too)
                if c in seen and hide.random() < 0.7:
                    self.t[i] = np.float32(0)
                seen.setdefault(c, i)
        self.lo, self.hi = self.p.min(0), self.p.max(0)
        self.applied = 0          # picks s_0 .. s_{applied-1} are in t
        self.cmax = np.float32(np.inf)

    def lower_bound(self, s):
        g = np.maximum(0, np.maximum(self.lo - s, s - self.hi)).astype(np.float32)
        return sqdist(g[None], np.zeros(3, np.float32))[0]

    def top2(self, k=2):
        """the record: top candidates by SCORE (= min-distance x weight; weight 1 for D-FPS) as (score, index, position,
        min-distance, weight)"""
        sc = (self.t * self.w).astype(np.float32)
        order = sorted(range(len(self.ids)), key=lambda i: (-float(sc[i]), self.keys[i]))[:k]
        return [(np.float32(sc[i]), int(self.ids[i]), self.p[i].copy(), np.float32(self.t[i]), np.float32(self.w[i])) for i in order]


def run(xyz, m, regions, seed=0, max_batch=3, stats=None, greedy=False, delay=None, depth=2, hide_duplicates=False, kernel_keys=False,
        weights=None):
    """regions: list of index arrays partitioning range(n).  Returns the picks; raises on a protocol violation.
    greedy: the sequencer decides until it is blocked, then every owner catches up (counts how often a rescan is on the
    critical path: stats['blocks'])."""
    rng = np.random.default_rng(seed)
    xyz = np.asarray(xyz, np.float32)
    n = xyz.shape[0]
    log2s = opt_log2s(n)
    key = lambda k: tie_key(k, log2s)
    better = lambda a, b: a[0] > b[0] or (a[0] == b[0] and key(a[1]) <= key(b[1]))      # a >= b in the order
    owners = [Region(ids, xyz, log2s, hide=np.random.default_rng(seed + 77) if hide_duplicates else None, weights=weights)
              for ids in regions]
    # weights (round 6, fps_seq.hip: fps_seq_w_kernel): records rank by SCORE, the sequencer keeps the candidates' MIN-DISTANCES
    # exact and rebuilds the scores, the unknown-region bound is the last candidate's old score, the owners' skip test compares
    # the pick's distance to the box with the region's maximal MIN-DISTANCE
    score = lambda cv, c: np.float32(np.float32(cv) * c[4])
    p0 = first_pick(n, log2s, weights)
    hist = [xyz[p0].copy()]         # published picks (positions); hist[0] is point 0 (D-FPS) / the arg-max of the weights
    picks = [p0]
    records = [None] * len(owners)  # published: (tag, [(v1,k1,p1), (v2,k2,p2)])
    seq = [dict(tag=0, rec=None, cv=None) for _ in owners]
    blocked_polls = decisions = 0

    def owner_step(o, w):
        # consume published picks in order; at most max_batch per step
        for _ in range(int(rng.integers(1, max_batch + 1))):
            if o.applied >= len(hist):
                return
            s = hist[o.applied]
            if o.lower_bound(s) >= o.cmax:          # nothing can change
                o.applied += 1
                continue
            o.t = np.minimum(o.t, sqdist(o.p, s))
            o.applied += 1
            o.cmax = o.t.max()
            # (depth 0: every record gets a list length of its own, 1 .. 4 — the kernels shorten their lists after single-pick rounds)
            records[w] = (o.applied, o.top2(depth if depth > 0 else int(rng.integers(1, 5))))

    def sequencer_poll():
        for w, st in enumerate(seq):
            r = records[w]
            if r is not None and r[0] > st['tag']:
                assert r[0] <= len(hist)
                st['tag'], st['rec'] = r[0], r[1]
                cv = [c[3] for c in r[1]]                        # min-distances (== the values for D-FPS)
                for i in range(r[0], len(hist)):                 # picks made since the record
                    for j in range(len(cv)):
                        cv[j] = min(cv[j], sqdist(r[1][j][2][None], hist[i])[0])
                st['cv'] = cv

    def bits(v):
        return int(np.float32(v).view(np.uint32))

    def decide_with_kernel_keys():
        """the key scheme of fps_seq.hip / fps_coop.hip: per candidate thr / alt, ONE global max, no per-region reduction"""
        entries = []          # (key, tie key or None, candidate)
        for st in seq:
            if st['rec'] is None:
                return False
            v_l, k_l = st['rec'][-1][0], st['rec'][-1][1]
            ub = 2 * bits(v_l) + 3
            for j, (cv, c) in enumerate(zip(st['cv'], st['rec'])):
                ekey = 2 * bits(score(cv, c)) + 2
                thr = ub - 1 if key(c[1]) <= key(k_l) else ub
                if ekey >= thr:
                    entries.append((ekey, key(c[1]), (score(cv, c), c[1], c[2])))
                elif j == len(st['rec']) - 1:
                    entries.append((ub, None, None))
        top = max(e[0] for e in entries)
        if top & 1:
            return False
        best = min((e for e in entries if e[0] == top), key=lambda e: e[1])[2]
        picks.append(best[1])
        hist.append(best[2].copy())
        for st in seq:
            for j in range(len(st['cv'])):
                st['cv'][j] = min(st['cv'][j], sqdist(st['rec'][j][2][None], best[2])[0])
        return True

    def sequencer_decide():
        if kernel_keys:
            return decide_with_kernel_keys()
        best, ub = None, np.float32(-np.inf)
        for st in seq:
            if st['rec'] is None:
                return False
            cands = [(score(cv, c), c[1], c[2]) for cv, c in zip(st['cv'], st['rec'])]
            x = cands[0]
            for c in cands[1:]:
                if not better(x, c):
                    x = c
            v2, k2 = st['rec'][-1][0], st['rec'][-1][1]     # every other point of the region was ordered after the LAST candidate
            if better(x, (v2, k2)):
                if best is None or not better(best, x):
                    best = x
            else:
                ub = max(ub, v2)
        if best is None or not (ub < best[0]):
            return False
        picks.append(best[1])
        hist.append(best[2].copy())
        for st in seq:
            for j in range(len(st['cv'])):
                st['cv'][j] = min(st['cv'][j], sqdist(st['rec'][j][2][None], best[2])[0])
        return True

    blocks = 0
    pending = []                     # (visible_at_decision, region, record): the latency model
    while delay is not None and len(picks) < m:
        # every owner applies a pick as soon as it is made; its new record reaches the sequencer `delay` decisions later
        for w, o in enumerate(owners):
            before = records[w]
            while o.applied < len(hist):
                owner_step(o, w)
            if records[w] is not before:
                pending.append((len(picks) + delay, w, records[w]))
                records[w] = before
        def deliver(upto):
            keep = []
            for at, w, rec in pending:
                if at <= upto:
                    if records[w] is None or rec[0] > records[w][0]:
                        records[w] = rec
                else:
                    keep.append((at, w, rec))
            pending[:] = keep
        deliver(len(picks))
        sequencer_poll()
        if sequencer_decide():
            decisions += 1
        else:
            blocks += 1
            deliver(1 << 60)         # wait for everything in flight
            sequencer_poll()
            assert sequencer_decide()
            decisions += 1
    while greedy and len(picks) < m:
        sequencer_poll()
        while len(picks) < m and sequencer_decide():
            decisions += 1
        if len(picks) < m:
            blocks += 1
            for w, o in enumerate(owners):
                while o.applied < len(hist):
                    owner_step(o, w)
    guard = 0
    while len(picks) < m:
        guard += 1
        assert guard < 400 * m + 10000, "no progress: liveness violated"
        a = int(rng.integers(0, len(owners) + 2))
        if a >= len(owners):
            sequencer_poll()
            for _ in range(int(rng.integers(1, 5))):
                if len(picks) >= m:
                    break
                if sequencer_decide():
                    decisions += 1
                else:
                    blocked_polls += 1
                    break
        else:
            owner_step(owners[a], a)
    if stats is not None:
        stats.update(blocked_polls=blocked_polls, decisions=decisions, blocks=blocks)
    return picks
