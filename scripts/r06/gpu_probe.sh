# round 6, first call: what the box lets an ordinary user read about clocks / power (sysfs, rocm-smi, amd-smi), and a baseline worker line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r06_probe; mkdir -p $out
{
echo "== id"; id
echo "== which"; which rocm-smi amd-smi rocminfo
echo "== drm"; ls /sys/class/drm/ 2>&1 | head -30
for c in /sys/class/drm/card*/device; do
  echo "-- $c"; cat $c/pp_dpm_sclk 2>&1 | head -12; cat $c/pp_dpm_mclk 2>&1 | head -6
  ls $c/hwmon 2>&1; for h in $c/hwmon/hwmon*; do echo "  $h"; ls $h | tr '\n' ' '; echo; for f in power1_average power1_input freq1_input freq2_input temp1_input; do echo -n "  $f: "; cat $h/$f 2>&1; done; done
  ls $c | tr '\n' ' ' | head -c 3000; echo
  echo -n "gpu_metrics bytes: "; wc -c < $c/gpu_metrics 2>&1
done
echo "== rocm-smi"; timeout 60 rocm-smi --showclocks --showpower 2>&1 | head -40
echo "== amd-smi metric"; timeout 60 amd-smi metric -c -p 2>&1 | head -60
echo "== amd-smi python"; python3 -c "
import amdsmi
amdsmi.amdsmi_init()
hs = amdsmi.amdsmi_get_processor_handles()
print(len(hs))
h = hs[0]
print(amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX))
print(amdsmi.amdsmi_get_power_info(h))
try:
    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
    print({k: m[k] for k in m if 'gfxclk' in k or 'power' in k or 'clock' in k})
except Exception as e: print('metrics', e)
" 2>&1 | head -40
} > $out/probe.txt 2>&1
tail -c 6000 $out/probe.txt
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline ) > $out/base.log 2> $out/base.err
grep '^{' $out/base.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', d['value'], d['selfcheck'], d['crosscheck'])"; tail -3 $out/base.err
