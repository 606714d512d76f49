set -e
cd cbinfer_amd/csrc && touch cb_conv.hip && make EXTRA=-DCB_STAMP >/dev/null 2>&1 && cd ../..
timeout -k 10 120 python tools/stamp_phases.py 2 0.04 0.27 1.0 > gpurun_out/st_wide.log 2>&1
CBINFER_X3_WIDE=0 timeout -k 10 120 python tools/stamp_phases.py 2 0.04 0.27 1.0 > gpurun_out/st_narrow.log 2>&1
echo WIDE; cat gpurun_out/st_wide.log; echo NARROW; cat gpurun_out/st_narrow.log
