# kernel trace of the OpenPose (config 4) part of tools/sweep.py, summarised per kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pose; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/tr -o t --output-format csv -- python3 $R/tools/sweep.py --skip-sweep --steps 20 > $O/pose.txt 2> $O/pose.err || exit 1
f=$(find $O/tr -name "*kernel_stats.csv" | head -1); python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:14]:
    print("%-100s n=%6s avg %8.2f us  %5.1f%%" % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
rm -rf $O/tr
