#!/bin/bash
# round 5, call 37: a creation batch fills the supporting planes once (the flag pass of every keyframe but the last leaves them empty):
# lifecycle / e2e / host-class tests, then the drop-in figure
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call37; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_lifecycle_stages.py tests/test_gpu_e2e_vga.py tests/test_gpu_directba_vs_oracle.py tests/test_gpu_edge_cases.py tests/test_gpu_directba_cpp.py tests/test_gpu_tum_pipeline.py tests/test_gpu_golden_reference.py tests/test_gpu_sharded_loopback.py -q -m gpu -x 2>&1 | tail -4 | tee $O/gpu_tests.log | cut -c1-300
for r in 1 2; do python scripts/drop_in_profile.py 2>&1 | grep "ms per call"; done
