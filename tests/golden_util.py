"""Loader for tests/golden/*.npz (written by oracle/gen_golden.py from the real reference)."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Case(dict):
    """dict of torch tensors + .meta"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def load_cases(name):
    data = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(data["meta"]).decode())
    cases = []
    for i, m in enumerate(meta):
        c = Case()
        for k in m["keys"]:
            c[k] = torch.from_numpy(np.array(data["c%d_%s" % (i, k)]))
        c["meta"] = m
        cases.append(c)
    return cases


def case_ids(name):
    return ["%s%d" % (name, i) for i in range(len(load_cases(name)))]
