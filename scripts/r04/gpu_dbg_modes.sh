#!/bin/bash
mkdir -p gpurun_out/r04
for mode in plain zerocoop zerohoist zeros; do
  MODE=$mode timeout 300 python scripts/r04/dbg_graph65536d.py 2>&1 | grep "differs" | grep "trial 1 \|trial 2 " | cut -c1-300
done
