# kernel stats of the 65536-point configuration (BASELINE config 5), eager 8-scene passes on one stream + the pipelined rate
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
BATCH=8 NOPMC=1 bash scripts/r03/gpu_pmc.sh r03_65536 --cfg synthetic_models/det6d_65536.yaml --points 65536
python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "cooperative_sampler_for_large" 2>&1 | tail -2
