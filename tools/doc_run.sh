R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/doc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tl -o t --output-format csv -- python3 $R/bench.py --mode eager --steps 60 --warmup 5 --min-seconds 0 --no-cpu-baseline --no-dense --multi 0 --no-pipelined > $O/tl.json 2> $O/tl.err || exit 1
python3 $R/tools/frame_timeline.py $O/tl > $O/timeline.txt; rm -rf $O/tl
cat $O/timeline.txt
cd $R && timeout -k 10 900 python3 tools/sweep.py --steps 100 > $O/sweep.md 2> $O/sweep.err; cat $O/sweep.md
