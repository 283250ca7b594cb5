#!/bin/bash
# round 4, call 11: per-(tile, item) log of the LDS sink's add calls on the commit where the immediate form gives wrong sums (short timeouts)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call11; mkdir -p $O
(cd _bisect/pC && timeout 120 python diag_adds.py 2>&1 | tail -12) | tee $O/diag_adds.log
echo "== PCG LDS form parity"; timeout 300 python -m pytest tests/test_gpu_intrinsics_pcg_vs_oracle.py -q -m gpu -k "pcg_iteration" 2>&1 | tail -4 | tee $O/pcg_lds.log
