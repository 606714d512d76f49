#!/bin/bash
# Run ON THE GPU BOX (through gpurun): produces under gpurun_out/prof/ everything profiles/rNN_* is made of.
#   bench.json                 the default bench line
#   stats/                     rocprofv3 --kernel-trace --stats of the same command (+ its bench line)
#   pmc_fetch/, pmc_write/     FETCH_SIZE / WRITE_SIZE passes (counters only, eager launches so that every
#                              kernel is its own dispatch), summarised by tools/pmc_traffic.py
#   fg_stats/, fg_pmc_*        the same for the fine-grained experiment 7 (in-place form)
# usage: collect_profiles.sh <commit>        Every step is bounded; a step that is killed stops the script.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
COMMIT=${1:-unknown}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: $*"; exit 1; fi; return $rc; }
run timeout -k 10 420 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"
run timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- \
    python3 $R/bench.py --no-cpu-baseline --multi 0 --no-secondary > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  d=$O/pmc_$(echo $c | tr A-Z a-z | sed 's/_size//')
  run timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $d -o p --output-format csv -- \
      python3 $R/bench.py --mode eager --steps 40 --warmup 5 --min-seconds 0 --no-cpu-baseline --no-dense --multi 0 --no-variants --no-last-frame --no-pipelined --no-secondary --no-isolated \
      > $d.json 2> $d.err || exit 1
  echo "$c done"
done
python3 $R/tools/pmc_traffic.py $O --json $O/pmc_traffic.json --commit $COMMIT > $O/pmc_traffic.txt
cat $O/pmc_traffic.txt | cut -c1-160
# fine-grained experiment 7, in-place form, 10 % change
run timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/fg_stats -o s --output-format csv -- \
    python3 $R/tools/fg_target.py > $O/fg_target.txt 2> $O/fg_stats.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  d=$O/fg_pmc_$(echo $c | tr A-Z a-z | sed 's/_size//')
  run timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $d -o p --output-format csv -- \
      python3 $R/tools/fg_target.py > $d.txt 2> $d.err || exit 1
done
python3 - <<PY
import csv, glob, collections, statistics
for sub, c in (("fg_pmc_fetch", "FETCH_SIZE"), ("fg_pmc_write", "WRITE_SIZE")):
    by = collections.defaultdict(list)
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "cb_" in r["Kernel_Name"]:
                by[r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:70]].append(float(r["Counter_Value"]))
    for k, v in sorted(by.items()):
        print("fg %-72s n=%4d %s median %10.2f KB" % (k, len(v), c, statistics.median(v[3:] or v)))
PY
echo "fg done"
# OpenPose T=2 fp16 (config 4)
run timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/pose_stats -o s --output-format csv -- \
    python3 $R/tools/pose_target.py > $O/pose_target.txt 2> $O/pose_stats.err || exit 1
# (one steady-state frame of the same run, launch by launch)
python3 $R/tools/pose_frame_table.py $O/pose_stats > $O/pose_frame.txt 2>/dev/null
# the same network with every layer's own detection launch (round 5's form), and in the library-only form (change-based pools
# folded, library concatenation) replayed as a recorded launch program -- un-profiled, for the frames/s
export POSE_NOFOLD=1 POSE_GROUPED=0; run timeout -k 10 200 python3 $R/tools/pose_target.py 2>/dev/null | grep "change-based" | sed 's/^/own detection launches, branches one after the other: /' >> $O/pose_target.txt; unset POSE_NOFOLD POSE_GROUPED
export POSE_POOLS=1 POSE_PROGRAM=1; run timeout -k 10 200 python3 $R/tools/pose_target.py 2>/dev/null | grep "change-based\|program\|producer" | sed 's/^/change-based pools folded + library concat: /' >> $O/pose_target.txt; unset POSE_POOLS POSE_PROGRAM
echo "pose done"
# keep the summaries, drop the bulky raw traces (gpurun merges at most 64 MiB back)
mkdir -p $O/keep
cp $O/bench.json $O/bench_under_rocprof.json $O/pmc_traffic.json $O/pmc_traffic.txt $O/fg_target.txt $O/keep/ 2>/dev/null
cp $O/pose_target.txt $O/pose_frame.txt $O/keep/ 2>/dev/null
for d in stats fg_stats pose_stats; do
  f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/keep/${d}_kernel_stats.csv
done
f=$(find $O/stats -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 $R/tools/trace_summary.py $f > $O/keep/kernel_trace_summary.txt 2>/dev/null
rm -rf $O/stats $O/fg_stats $O/pose_stats $O/pmc_fetch $O/pmc_write $O/fg_pmc_fetch $O/fg_pmc_write
ls -la $O/keep
