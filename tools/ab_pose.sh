# Same-box A/B of two library builds on the OpenPose target (config 4): tmp_ab/libcbinfer_hip_old.so against the tree's library,
# alternating; prints the change-based frames/s of tools/pose_target.py (POSE_* switches pass through).  usage: ab_pose.sh [rounds]
set -e
cd ${GRAFT_REPO_ROOT:-.}
R=${1:-2}
cp cbinfer_amd/libcbinfer_hip.so tmp_ab/new.so
trap 'cp tmp_ab/new.so cbinfer_amd/libcbinfer_hip.so' EXIT
for i in $(seq $R); do
  for v in old new; do
    if [ $v = old ]; then cp tmp_ab/libcbinfer_hip_old.so cbinfer_amd/libcbinfer_hip.so; else cp tmp_ab/new.so cbinfer_amd/libcbinfer_hip.so; fi
    echo "$v $(timeout -k 10 200 python tools/pose_target.py 2>/dev/null | grep "change-based\|program" | cut -c1-120 | tr "\n" " ")"
  done
done
