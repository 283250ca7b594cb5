"""What a measurement was taken on: a digest of the kernel sources (badslam_amd/csrc: every .hip / .h and the Makefile with its flags).
scripts/summarize_profile.py stamps it into profiles/*_pmc_per_kernel.json; bench.py quotes counter evidence only from a profile whose
stamp equals the digest of the sources it runs on (round 5's line quoted round-4 counters against round-5 kernels).
Works without git (the GPU box receives a snapshot)."""
import glob
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def csrc_digest():
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(CSRC, "Makefile")]):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]
