# Same-box A/B of two builds of the library (run through gpurun): tmp_ab/libcbinfer_hip_old.so against the tree's
# cbinfer_amd/libcbinfer_hip.so, alternating, the same short bench each time -- box-to-box variance is +-3 %, this
# resolves 0.3 %.  Prepare: build the baseline, cp cbinfer_amd/libcbinfer_hip.so tmp_ab/libcbinfer_hip_old.so; edit; make.
# usage: bash tools/ab_libs.sh [rounds] [extra bench.py flags]      (AB_ENV_NEW / AB_ENV_OLD: "VAR=1 VAR2=0" per side)
set -e
cd ${GRAFT_REPO_ROOT:-.}
R=${1:-3}; shift || true
mkdir -p tmp_ab; cp cbinfer_amd/libcbinfer_hip.so tmp_ab/new.so
trap 'cp tmp_ab/new.so cbinfer_amd/libcbinfer_hip.so' EXIT
for i in $(seq $R); do
  for v in old new; do
    if [ $v = old ]; then cp tmp_ab/libcbinfer_hip_old.so cbinfer_amd/libcbinfer_hip.so; E="$AB_ENV_OLD"; else cp tmp_ab/new.so cbinfer_amd/libcbinfer_hip.so; E="$AB_ENV_NEW"; fi
    env $E python bench.py --no-variants --no-isolated --multi 0 --no-pipelined --no-cpu-baseline --no-secondary --no-last-frame --no-dense "$@" 2>/dev/null | tail -1 > /tmp/l.json
    python - <<PY
import json
d=json.load(open("/tmp/l.json"))
k=json.load(open("gpurun_out/bench_details.json"))["kernel_trace"]["kernels_as_traced"]
dd=json.load(open("gpurun_out/bench_details.json"))
ms=dd.get("multi_sequence") or {}
if ms: print("$v multi", ms.get("sequences_per_gpu"), round(ms.get("value",0)), {k[:40]: v for k,v in (ms.get("step_of_8_kernels_us") or {}).items()}, {k: round(v["value"]) for k,v in (ms.get("batched") or {}).items()})
print("$v", round(d["value"]), {n.split("(")[0].split("::")[-1][:28]: v["avg_us"] for n,v in k.items()})
PY
  done
done
