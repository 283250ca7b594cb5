#!/usr/bin/env python3
"""Launches the known-size read patterns of bahip_debug_read_pattern (run under `rocprofv3 --pmc FETCH_SIZE`, see
scripts/profile_round.sh): 2 GiB read with 4-byte loads, coalesced (pattern 0) and as a one-dword-per-128-byte-line gather
(pattern 1), 3 launches each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from badslam_amd import capi, lowlevel   # noqa: E402

BYTES = 2 << 30
ctx = lowlevel.Context()
for pattern in (0, 1):
    capi.check(ctx.lib.bahip_debug_read_pattern(ctx.handle, BYTES, pattern, 3))
print(BYTES)
