# copies what tools/collect_profiles.sh + tools/timeline_run.sh (eager and MODE=graph) left under gpurun_out/ into
# profiles/r06_* (run in the build container after the gpurun call; the bench line itself -- profiles/r06_bench.json --
# is made by a bench run AFTER this, so that its roofline.traffic quotes these passes)
set -e
cd "$(git rev-parse --show-toplevel)"
K=gpurun_out/prof/keep
cp $K/pmc_traffic.txt profiles/r06_pmc_traffic_raw.txt
cp $K/stats_kernel_stats.csv profiles/r06_bench_kernel_stats.csv
cp $K/bench_under_rocprof.json profiles/r06_bench_under_rocprof.json
cp $K/kernel_trace_summary.txt profiles/r06_bench_kernel_clusters.txt
cp $K/fg_target.txt profiles/r06_fg_exp7_target.txt
cp $K/fg_stats_kernel_stats.csv profiles/r06_fg_exp7_kernel_stats.csv
cp $K/pose_target.txt profiles/r06_openpose_target.txt
cp $K/pose_stats_kernel_stats.csv profiles/r06_openpose_kernel_stats.csv
cp $K/pose_frame.txt profiles/r06_openpose_frame.txt
python3 - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
d = json.load(open('gpurun_out/prof/keep/pmc_traffic.json'))
assert d['kernel_source_sha256'] == bench.kernel_source_hash(), "the passes ran on other kernel sources"
json.dump(d, open('profiles/r06_pmc_traffic.json', 'w'), indent=1)
print("pmc traffic of commit", d['commit'], "hash", d['kernel_source_sha256'])
PY
{ echo "# eager stream (the launch form the bench runs): tools/timeline_run.sh, MI355X, commit $(git rev-parse --short HEAD)"
  echo "# the first group is the timed loop (4 launches per frame: DESIGN 5.8, 5.9; under the profiler the host is the bottleneck, so the GAPS are the host's -- the un-profiled frame period equals the sum of the durations); the other groups are the bench's in-frame measurement passes (event records, on-demand compaction)"
  cat gpurun_out/doc/timeline_eager.txt
  echo
  echo "# the same network replayed from a hipGraph (MODE=graph tools/timeline_run.sh): idle time between the kernels of different library calls that the eager stream does not have"
  cat gpurun_out/doc/timeline_graph.txt
  echo
  echo "# the frame as a recorded launch program (MODE=program: pycbinfer.FrameProgram replays the library calls of one eager frame): the eager stream's kernels, three library calls per frame on the host"
  cat gpurun_out/doc/timeline_program.txt; } > profiles/r06_frame_timeline.txt
