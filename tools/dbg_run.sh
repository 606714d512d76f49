# diagnostic: X3 list contraction with the gather loads (1), the weight loads (2) or both (3) ablated
set -e
cd cbinfer_amd/csrc && touch cb_conv.hip && make EXTRA=-DCB_CONV_DBG >/dev/null 2>&1 && cd ../..
for d in 0 1 2 3; do CBINFER_CONV_DBG=$d timeout -k 10 120 python tools/bench_x3.py 2>&1 | grep -v amdgpu.ids; done
