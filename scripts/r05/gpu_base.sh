# round 5, first call: the GPU suite on the tree as it is (24 hardware queues from conftest), the PMC pass over the PIPELINED worker the
# round-4 review asked for, and a baseline bench line of the same box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_base; mkdir -p $out
( time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1; tail -5 $out/pytest.log
grep -c "alias\|GPU_MAX_HW_QUEUES" $out/pytest.log
export GPU_MAX_HW_QUEUES=24
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs > $out/bench_worker.log 2> $out/bench_worker.err
grep '^{' $out/bench_worker.log | cut -c1-200
p=$out/pipe_pmc; mkdir -p $p
timeout 1200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES --output-format csv -d $p -o pmc -- python3 bench.py --gpus 1 --worker --no-legs --steps 192 --warmup 48 --cpu-scenes 0 --no-roofline > $p/bench_stdout.log 2> $p/bench_stderr.log
grep '^{' $p/bench_stdout.log > $p/bench_under_profiler.json; cut -c1-200 $p/bench_under_profiler.json; tail -3 $p/bench_stderr.log
python3 scripts/r05/pipeline_pmc_summary.py $p $p/bench_under_profiler.json > $out/pipeline_pmc_summary.json; head -50 $out/pipeline_pmc_summary.json
find $out -name "*.csv" -size +3M -delete
