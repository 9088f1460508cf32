# round 6: general mlp_rows kernel with the uniform-branch / buffer-store epilogue and next-tile input prefetch: parity, the three
# launches replayed alone (#5 SA2 stack, #10 vote, #15 towers), pipeline A/B against the previous commit's library; one-frame trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r06_t14; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_compact_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 1500 python3 -m pytest tests/test_timed_path_gpu.py -m gpu -x -q -k "coalesced or full_size_vs or other_baseline" 2>&1 | tail -3
python3 scripts/r06/kernel_power.py uniform 0.4 2,5,10,15 2>&1 | grep -v "amdgpu.ids\|^#"
B="--gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline"
one() { tag=$1; shift; extra=""; envs=""
  for a in "$@"; do case $a in --*) extra="$extra ${a/=/ }";; *) envs="$envs $a";; esac; done
  env $envs python3 bench.py $B $extra > $out/$tag.log 2> $out/$tag.err
  grep '^{' $out/$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'])" || tail -3 $out/$tag.err; }
for i in 1 2 3; do
one old_$i DET6D_KNOBS_LIB=$GRAFT_REPO_ROOT/scripts/r06/prev/libdet6d_hip_prev.so
one new_$i X=1
done
# one frame, eager: which kernels make up latency_b1
A="--steps 6 --warmup 2 --batch 1 --streams 1 --no-graph --cpu-scenes 0 --no-roofline --no-legs --preroll 0 --windows 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/b1 -o k -- python3 bench.py $A > $out/b1.log 2>&1
python3 - $out/b1 $out <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = max(i for i, r in enumerate(rows) if 'pack_points_kernel' in r['Kernel_Name'])
t0 = int(rows[last]['Start_Timestamp']); tot = 0.0
with open(sys.argv[2] + '/b1_launches_of_one_frame.txt', 'w') as f:
    for r in rows[last:]:
        n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')
        us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3; tot += us
        f.write("%9.1f us  at %8.1f us  grid %-8s wg %-5s %s\n" % (us, (int(r['Start_Timestamp']) - t0) * 1e-3, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')), n[:110]))
    f.write("sum of kernel durations %.1f us, first start -> last end %.1f us\n" % (tot, (int(rows[-1]['End_Timestamp']) - t0) * 1e-3))
PY
sort -rn $out/b1_launches_of_one_frame.txt | head -14; tail -1 $out/b1_launches_of_one_frame.txt
rm -rf $out/b1
