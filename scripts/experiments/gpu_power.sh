cd $GRAFT_REPO_ROOT
(python bench.py --steps 15360 --warmup 96 --cpu-scenes 0 --no-roofline 2>/dev/null | tail -1 | cut -c1-200) &
BP=$!
sleep 9
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -i -E "power|sclk|mclk|GPU use" | head -6; echo ---; sleep 1.5; done
wait $BP
echo idle; sleep 3; rocm-smi --showpower --showclocks 2>/dev/null | grep -i -E "power|sclk" | head -3
