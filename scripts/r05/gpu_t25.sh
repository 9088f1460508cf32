# round 5: the round-4 build (git archive of the round-4 commit, built in scratch_r04/) on the long measurement span, beside
# this round's build: what the round-4 headline was worth without the span bias, and what this round gained
# scratch_r04/ is not tracked; recreate it HERE (not on the GPU box, which has no .git) before the gpurun call:
#   mkdir scratch_r04 && git archive 258dc59 | tar -x -C scratch_r04 && (cd scratch_r04 && python -m de6d_amd._build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=$GRAFT_REPO_ROOT/gpurun_out/r05_t25; mkdir -p $out
summ() { grep '^{' $out/b_$1.log | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; k=c['batches_per_pass']
total=c['preroll_steps']+d['warmup']-1+(c['windows']-1)*k+d['steps']+1+c['tail_steps']
print('$1', d['value'], d['selfcheck'], 'whole stream', round(total*c['scenes_per_step_per_gpu']/d['stream_total_s']), 'fit', d.get('crosscheck',{}).get('fit_scenes_per_s'), 'windows', c['windows'])" || tail -3 $out/b_$1.err; }
r4() { tag=$1; shift; (cd scratch_r04 && timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err); summ $tag; }
r5() { tag=$1; shift; timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err; summ $tag; }
for rep in 1 2; do
r4 r4_short_$rep --merge 4
r4 r4_long_$rep --merge 4 --windows 320
r5 r5_short_$rep --merge 4 --windows 192
r5 r5_long_$rep --merge 4
r5 r5_m10_$rep
done
r4 r4_beam_long --merge 4 --windows 320 --scene beam
r5 r5_beam_long --merge 4 --scene beam
r5 r5_beam_m10 --scene beam
