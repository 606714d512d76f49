#!/bin/bash
# Run ON THE GPU BOX (through gpurun): produces under gpurun_out/prof/ everything profiles/rNN_* is made of.
#   bench.json                 the default bench line
#   stats/                     rocprofv3 --kernel-trace --stats of the same command (+ its bench line)
#   pmc_fetch/, pmc_write/     FETCH_SIZE / WRITE_SIZE passes (counters only, eager launches so that every
#                              kernel is its own dispatch), summarised by tools/pmc_traffic.py
# Every step is bounded; a step that is killed stops the script.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: $*"; exit 1; fi; return $rc; }
run timeout -k 10 420 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"
run timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- \
    python3 $R/bench.py --no-cpu-baseline --multi 0 > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  d=$O/pmc_$(echo $c | tr A-Z a-z | sed 's/_size//')
  run timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $d -o p --output-format csv -- \
      python3 $R/bench.py --mode eager --steps 40 --warmup 5 --no-cpu-baseline --no-dense --multi 0 \
      > $d.json 2> $d.err || exit 1
  echo "$c done"
done
python3 $R/tools/pmc_traffic.py $O > $O/pmc_traffic.txt
cat $O/pmc_traffic.txt
