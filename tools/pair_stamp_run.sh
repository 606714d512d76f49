# diagnostic: per-workgroup phase stamps of the row-pair kernel inside the bench frame (stamp build)
set -e
trap 'make -s -C "$(git rev-parse --show-toplevel 2>/dev/null || pwd)/cbinfer_amd/csrc" >/dev/null 2>&1 || echo "WARNING: could not restore the normal build" >&2' EXIT
cd cbinfer_amd/csrc && make EXTRA=-DCBP_STAMP >/dev/null 2>&1 && cd ../..
timeout -k 10 120 python tools/pair_stamps.py "$@" 2>&1 | grep -v amdgpu.ids
