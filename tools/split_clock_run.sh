# diagnostic: shader clock and stage-loop time of the split-state contraction under the CBINFER_SPLIT_DBG ablations (stamp + dbg build)
set -e
trap 'make -s -j8 -C "$(git rev-parse --show-toplevel 2>/dev/null || pwd)/cbinfer_amd/csrc" >/dev/null 2>&1 || echo "WARNING: could not restore the normal build" >&2' EXIT
cd cbinfer_amd/csrc && make -j8 EXTRA="-DCBS_STAMP -DCBS_DBG" >/dev/null 2>&1 && cd ../..
for d in ${DBG_LIST:-0 4 2 24}; do echo "dbg=$d"; CBINFER_SPLIT_DBG=$d timeout -k 10 120 python tools/split_stamps.py "$@" 2>&1 | grep "stage loop of\|128-row\|64-row"; done
