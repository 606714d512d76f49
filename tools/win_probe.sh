# diagnostic (round 6): durations of the 16->64 contraction in pixel order (AR = 2) and window order (AR = 3) in the direct
# test of tests/test_gpu_split.py, under rocprofv3 --kernel-trace.  usage: win_probe.sh [make EXTRA flags]
R=$GRAFT_REPO_ROOT
if [ -n "$1" ]; then (cd $R/cbinfer_amd/csrc && make -j8 EXTRA="$1" >/dev/null 2>&1) || exit 1; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tlh -o t --output-format csv -- python3 -m pytest $R/tests/test_gpu_split.py -x -q -k "window_order and 160" > /dev/null 2>&1
python3 $R/tools/trace_hist.py $R/gpurun_out/tlh cbs_conv | cut -c1-200
rm -rf $R/gpurun_out/tlh
