# one steady-state frame of tools/pose_target.py (OpenPose, config 4) under rocprofv3 --kernel-trace: the launches in
# order with their durations, totals by kernel (tools/pose_frame_table.py).  usage (through gpurun): bash tools/pose_trace.sh [feedback]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pose_prof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O -o pose --output-format csv -- python3 $R/tools/pose_target.py $1 > $R/gpurun_out/pose_prof.log 2>&1 || exit 1
python3 $R/tools/pose_frame_table.py $O > $R/gpurun_out/pose_frame.txt; rm -f $O/*trace.csv
cat $R/gpurun_out/pose_frame.txt
