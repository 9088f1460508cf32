# counters of one eager pass on the three workloads: benchmark scenes, ray-cast scenes, 65536-point scenes (BASELINE configs[4] per-GPU share)
sfx=${1:-z}
bash scripts/r06/gpu_pmc.sh r06${sfx}_z
bash scripts/r06/gpu_pmc.sh r06${sfx}_beam --scene beam
BATCH=8 STEPS=4 bash scripts/r06/gpu_pmc.sh r06${sfx}_65536 --cfg synthetic_models/det6d_65536.yaml --points 65536
# per-kernel stats of the 80-scene pass the timed region issues by default (bench.py's roofline.avg_launch_us is taken on these launches)
NOPMC=1 BATCH=80 STEPS=5 bash scripts/r06/gpu_pmc.sh r06${sfx}_z80
