#!/bin/bash
# The round's counter evidence in one go (run through gpurun): the default bench workload (all passes), then the same scene with the
# intrinsics step and with the PCG scheme (kernel statistics + FETCH_SIZE / WRITE_SIZE: what roofline_intrinsics / roofline_pcg quote).
#   scripts/profile_all.sh <tag prefix, e.g. r6_f>     -> gpurun_out/prof_<tag>, prof_<tag>_intr, prof_<tag>_pcg; copy lines printed
# bench.py quotes a summary only while badslam_amd/buildinfo.csrc_digest() is the one stamped into it: re-run after every kernel change.
set -u
TAG=${1:-r6}
scripts/profile_round.sh $TAG
CAL_FROM=gpurun_out/prof_$TAG PASSES=basic scripts/profile_round.sh ${TAG}_intr --intrinsics
CAL_FROM=gpurun_out/prof_$TAG PASSES=basic scripts/profile_round.sh ${TAG}_pcg --pcg
