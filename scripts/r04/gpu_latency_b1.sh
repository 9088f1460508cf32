#!/bin/bash
# kernel-by-kernel view of ONE frame (B = 1) replayed on an idle chip: shipped sampler, then DET6D_FPS_SEQ=1 of the experiments build
out=$GRAFT_REPO_ROOT/gpurun_out/r04/lat_b1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/skip -o lat -- python3 $GRAFT_REPO_ROOT/scripts/r04/latency_b1.py > $out/skip.log 2>&1
DET6D_EXPERIMENTS_LIB=1 DET6D_FPS_SEQ=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/seq -o lat -- python3 $GRAFT_REPO_ROOT/scripts/r04/latency_b1.py > $out/seq.log 2>&1
cd $GRAFT_REPO_ROOT
for v in skip seq; do
  f=$(find $out/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep latency $out/$v.log
  python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print('%-70s calls %5s avg %9.1f us  %5.1f %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
done
