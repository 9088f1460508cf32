"""how long does stage 1 of a group (pack + hoisted samplers on a sampler stream) take UNDER LOAD, from the moment its stream
reaches it to its `sampled` event?  (idle chip: ~2.4 ms for a 32-scene pass).  usage: stage1_latency.py [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
import torch
import bench
import de6d_amd.runtime as rt
_front = rt.Det6DGroup.launch_front
_rest = rt.GraphedDet6D.launch_rest
stamps, rests = [], []


def front(self, points=None, count=None):
    e0 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(self.hi):
        e0.record()
    out = _front(self, points, count)
    e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(self.hi):
        e1.record()
    stamps.append((e0, e1))
    return out


stalls = []


def rest(self, sampled):
    ready = torch.cuda.Event(enable_timing=True)
    e0 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(self.stream):
        ready.record()                    # the main stream has finished the pass it ran before
        self.stream.wait_event(sampled)
        e0.record()
    stalls.append((ready, e0))
    out = _rest(self, sampled)
    e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(self.stream):
        e1.record()
    rests.append((e0, e1))
    return out


rt.Det6DGroup.launch_front = front
rt.GraphedDet6D.launch_rest = rest
try:
    bench.main()
finally:
    torch.cuda.synchronize()
    for name, lst in (("stage 1 (pack + hoisted samplers)", stamps), ("stage 2 (captured rest of the pass)", rests),
                      ("main stream idle, waiting for its pass's samplers", stalls)):
        ms = sorted(a.elapsed_time(b) for a, b in lst[len(lst) // 4: 3 * len(lst) // 4])
        if ms:
            print("%s under load: n %d  median %.2f ms  p10 %.2f  p90 %.2f  max %.2f" % (name, len(ms), ms[len(ms) // 2], ms[len(ms) // 10], ms[9 * len(ms) // 10], ms[-1]), file=sys.stderr)
