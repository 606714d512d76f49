# diagnostic: phase stamps of the list contraction (stamp build); STAMP_FLAGS adds build flags, e.g. -DCB_X3_BHALF
set -e
# whatever happens below, leave the NORMAL library behind (the Makefile's flag stamp makes the plain
# make rebuild the instrumented objects)
trap 'make -s -C "$(git rev-parse --show-toplevel 2>/dev/null || pwd)/cbinfer_amd/csrc" >/dev/null 2>&1 || echo "WARNING: could not restore the normal build" >&2' EXIT
cd cbinfer_amd/csrc && make EXTRA="-DCB_STAMP $STAMP_FLAGS" >/dev/null 2>&1 && cd ../..
timeout -k 10 120 python tools/stamp_phases.py 2 ${STAMP_RATIOS:-0.04 0.27 1.0} 2>&1 | grep -v amdgpu.ids
