# diagnostic: phase stamps of the list contraction (stamp build); STAMP_FLAGS adds build flags, e.g. -DCB_X3_BHALF
set -e
cd cbinfer_amd/csrc && touch cb_conv.hip && make EXTRA="-DCB_STAMP $STAMP_FLAGS" >/dev/null 2>&1 && cd ../..
timeout -k 10 120 python tools/stamp_phases.py 2 ${STAMP_RATIOS:-0.04 0.27 1.0} 2>&1 | grep -v amdgpu.ids
