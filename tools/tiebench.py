#!/usr/bin/env python3
"""Development aid: what do reads made of exact tandem repeats (telomeres, microsatellites: every window minimum ties) cost kernel B?
Prints the syncmer kernel's time for a batch of ordinary reads, and for the same batch with a few repeat reads mixed in."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oatk_amd import HipSyncasm, pack_reads  # noqa: E402

rng = np.random.default_rng(1)
normal = [bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 15000)].tolist()) for _ in range(20000)]
rep = [(u * (20000 // len(u) + 1))[:20000] for u in (b"TTAGGG", b"AC", b"GAA", b"ACGTTGCAAGT", b"ACGTTGCAAGTCCATGACTGATCGATCGGATC" * 3)]
hip = HipSyncasm(0)
hip.set_timing(True)
for name, reads in (("ordinary", normal), ("ordinary + 5 repeat reads", normal + rep), ("ordinary + 50 repeat reads", normal + rep * 10),
                    ("ordinary + 1 % repeat reads (200)", normal + rep * 40), ("ordinary + 10 % repeat reads (2000)", normal + rep * 400)):
    seq, off, lens = pack_reads(reads)
    t = []
    for it in range(4):
        hip.scan_host(seq, off, lens, 1001, 31)
        if it:
            t.append(hip.timing()["syncmer"])
    print("%-36s syncmer %.3f ms" % (name, sum(t) / len(t)), flush=True)
