#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the row-segment kernel on the bench_rows shapes (counters only).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_rows
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH"
P3="GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $P -d $O/p$i -o p --output-format csv -- python3 $R/tools/bench_rows.py > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; }
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
for i in (1,2,3):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/p%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if "rowconv_f32" in r["Kernel_Name"]:
                by[(r["Grid_Size"] if "Grid_Size" in r else "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for g, d in sorted(by.items()):
        print("pass", i, "grid", g, {k: round(sum(v)/len(v)) for k, v in sorted(d.items())}, "n=%d" % len(next(iter(d.values()))))
PY
