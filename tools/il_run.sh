# diagnostic: the 128x128 list contraction with the interleaved stage body (make EXTRA=-DCB_WIDE_IL=1)
set -e
# whatever happens below, leave the NORMAL library behind (the Makefile's flag stamp makes the plain
# make rebuild the instrumented objects)
trap 'make -s -C "$(git rev-parse --show-toplevel 2>/dev/null || pwd)/cbinfer_amd/csrc" >/dev/null 2>&1 || echo "WARNING: could not restore the normal build" >&2' EXIT
cd cbinfer_amd/csrc && make EXTRA=-DCB_WIDE_IL=1 >/dev/null 2>&1 && cd ../..
timeout -k 10 120 python tools/bench_x3.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "split_contraction or fullsize" 2>&1 | tail -2
