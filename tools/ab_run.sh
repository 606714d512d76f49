# diagnostic: the bench with the three X3 tile forms of the list contraction on one box (CBINFER_X3_WIDE = 2, 1, 0)
one() { CBINFER_X3_WIDE=$1 timeout -k 10 400 python bench.py --no-cpu-baseline --multi 0 --no-pipelined > gpurun_out/ab.json 2>gpurun_out/ab.err; python - <<PY
import json
d=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("$2 wide=$1", round(d["value"]), [l.get("conv_ms") for l in d["layers"] if "conv_ms" in l], round(d["roofline"]["frac"],3))
PY
}
one 2 base; one 1 base; one 0 base
timeout -k 10 120 python tools/bench_blocks.py 2>&1 | grep "64->256"
