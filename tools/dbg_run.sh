# diagnostic: X3 list contraction with the gather loads (1), the weight loads (2) ablated; 128x128 form also:
# no LDS stage store (4), no fragment reads + MFMAs (8), no stage loads at all (16); bits add
set -e
# whatever happens below, leave the NORMAL library behind (the Makefile's flag stamp makes the plain
# make rebuild the instrumented objects)
trap 'make -s -C "$(git rev-parse --show-toplevel 2>/dev/null || pwd)/cbinfer_amd/csrc" >/dev/null 2>&1 || echo "WARNING: could not restore the normal build" >&2' EXIT
cd cbinfer_amd/csrc && make EXTRA=-DCB_CONV_DBG >/dev/null 2>&1 && cd ../..
for d in ${DBG_LIST:-0 1 2 3}; do CBINFER_CONV_DBG=$d timeout -k 10 120 python tools/bench_x3.py 2>&1 | grep -v amdgpu.ids; done
