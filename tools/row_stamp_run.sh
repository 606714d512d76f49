# diagnostic: per-workgroup phase stamps of the row-segment kernel (stamp build)
set -e
# whatever happens below, leave the NORMAL library behind (the Makefile's flag stamp makes the plain
# make rebuild the instrumented objects)
trap 'make -s -C "$(git rev-parse --show-toplevel 2>/dev/null || pwd)/cbinfer_amd/csrc" >/dev/null 2>&1 || echo "WARNING: could not restore the normal build" >&2' EXIT
cd cbinfer_amd/csrc && make EXTRA=-DCB_ROW_STAMP >/dev/null 2>&1 && cd ../..
timeout -k 10 120 python tools/row_stamps.py "$@" 2>&1 | grep -v amdgpu.ids
