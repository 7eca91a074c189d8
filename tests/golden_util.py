"""Load tests/golden/*.npz (reference outputs committed as data; generator: tests/make_golden.py)."""
import glob
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCAN_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*_k*_s*.npz")))
SCAN_FIELDS = ["hoco_l", "n_scm", "hoco_s", "ho_rl", "ho_l_rl", "n_nucl", "m_pos", "s_mer", "k_mer"]


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)
    return {k: z[k] for k in z.files}


def reads_of(g):
    seq, off = g["seq"], g["off"]
    return [seq[int(off[i]):int(off[i + 1])].tobytes() for i in range(len(off) - 1)]
