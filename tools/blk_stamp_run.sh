# diagnostic: per-workgroup phase stamps of the patch-staged kernel (stamp build), then the normal build back
set -e
cd cbinfer_amd/csrc && touch cb_blockconv.hip && make EXTRA=-DCB_BLK_STAMP >/dev/null 2>&1 && cd ../..
timeout -k 10 120 python tools/blk_stamps.py "$@" 2>&1 | grep -v amdgpu.ids
