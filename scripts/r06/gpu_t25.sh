# round 6: list-depth heuristic of the multi-pick sampler (DET6D_FPS_SEQ_DEPTH_ADD, pick cap) with this round's sequencer, knobs build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export DET6D_KNOBS_LIB=1
for d in 1 0 2 3 5 8; do echo "== depth_add $d"; DET6D_FPS_SEQ_DEPTH_ADD=$d python3 tests/gpu_scripts/fps_seq.py quick 2>&1 | grep "b=8 n=16384 m=4096\|b=8 n=4096 m=512"; done
for p in 16 24; do echo "== picks cap $p"; DET6D_FPS_SEQ_PICKS=$p python3 tests/gpu_scripts/fps_seq.py quick 2>&1 | grep "b=8 n=16384 m=4096"; done
