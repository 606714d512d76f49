# one steady-state frame under rocprofv3 --kernel-trace (eager bench run; MODE=graph: the captured frame), summarised by frame_timeline.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/doc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tl -o t --output-format csv -- python3 $R/bench.py --mode ${MODE:-eager} --steps 60 --warmup 5 --min-seconds 0 --no-cpu-baseline --no-dense --multi 0 --no-pipelined --no-variants --no-last-frame > $O/tl_${MODE:-eager}.json 2> $O/tl.err || exit 1
python3 $R/tools/frame_timeline.py $O/tl > $O/timeline_${MODE:-eager}.txt; rm -rf $O/tl
cat $O/timeline_${MODE:-eager}.txt
