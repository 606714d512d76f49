# diagnostic: per-workgroup phase stamps of the row-segment kernel (stamp build)
set -e
cd cbinfer_amd/csrc && touch cb_rowconv.hip && make EXTRA=-DCB_ROW_STAMP >/dev/null 2>&1 && cd ../..
timeout -k 10 120 python tools/row_stamps.py "$@" 2>&1 | grep -v amdgpu.ids
