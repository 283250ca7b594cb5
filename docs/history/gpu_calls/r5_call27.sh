#!/bin/bash
# round 5, call 27: issue-side counters of the intrinsics sweep next to the pose sweep of the same run (passes a-c of scripts/stall_profile.sh)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=8 PASS_TIMEOUT=80
sed -i 's/^pass d /#pass d /; s/^pass e /#pass e /; s/^pass f /#pass f /' scripts/stall_profile.sh
bash scripts/stall_profile.sh r5_intr --intrinsics 2>&1 | tail -80 > gpurun_out/stall_r5_intr_summary.txt
cat gpurun_out/stall_r5_intr_summary.txt | cut -c1-120
