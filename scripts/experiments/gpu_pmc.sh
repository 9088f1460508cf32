cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc_r01z
export TMPDIR=/tmp
STEPS=6; WARM=2; TOT=$((STEPS+WARM+1))
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_r01z/$tag -o pmc -- python3 bench.py --steps $STEPS --warmup $WARM --streams 1 --no-graph --cpu-scenes 0 --no-roofline > gpurun_out/pmc_r01z/$tag.log 2>&1
  tail -2 gpurun_out/pmc_r01z/$tag.log | cut -c1-200
done
python3 scripts/pmc_summarise.py gpurun_out/pmc_r01z $TOT | tee gpurun_out/pmc_r01z/summary.json
find gpurun_out/pmc_r01z -name "*.csv" -size +2M -delete
