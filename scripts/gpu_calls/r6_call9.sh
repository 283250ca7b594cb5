#!/bin/bash
# round 6, call 9: pipelined merge batches (apply of keyframe j beside insert of keyframe j + 1): parity with the oracle at the stage level,
# through DirectBA (e2e VGA golden, configs[1]-size lifecycle), and the drop-in call's time
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call9; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_lifecycle_stages.py tests/test_gpu_e2e_vga.py tests/test_gpu_directba_vs_oracle.py tests/test_gpu_tum_pipeline.py tests/test_gpu_sharded_loopback.py tests/test_gpu_fast_flavour.py -q -m gpu 2>&1 | tail -12
timeout -k 5 600 python -m pytest tests/test_gpu_scale_parity.py -q -m gpu -k "c2_size" 2>&1 | tail -4
for i in 1 2; do python scripts/drop_in_profile.py 2>&1 | grep "ms per call"; done
