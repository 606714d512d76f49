# diagnostic: split-state contraction with ablations (CBINFER_SPLIT_DBG bits, see cb_split.hip); DBG_LIST picks them
set -e
# whatever happens below, leave the NORMAL library behind (the Makefile's flag stamp makes the plain
# make rebuild the instrumented objects)
trap 'make -s -j8 -C "$(git rev-parse --show-toplevel 2>/dev/null || pwd)/cbinfer_amd/csrc" >/dev/null 2>&1 || echo "WARNING: could not restore the normal build" >&2' EXIT
cd cbinfer_amd/csrc && make -j8 EXTRA=-DCBS_DBG >/dev/null 2>&1 && cd ../..
for d in ${DBG_LIST:-0 1 2 4 8 16 32}; do CBINFER_SPLIT_DBG=$d timeout -k 10 120 python tools/bench_split.py "$@" 2>&1 | grep -v amdgpu.ids; done
