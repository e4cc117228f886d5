import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = Path(__file__).resolve().parent / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


INDEX_FIXTURES = {
    "randn128_R16_m32": "randn128", "randn128_R64_m16": "randn128",
    "sift128_R64_m32": "sift128", "sift128_R16_m32": "sift128",
    "unit1536_R16_m32": "unit1536", "unit1536_R16_m64": "unit1536",
    "faq32_R16_nopq": "faq32", "deep96_R32_m16": "deep96",
    # round 4: the edges of the reference's legal PQ shapes (adaptive_pq.py:29,80-91): m = 96 (sub_dim 8), m = 128 (sub_dim 2), m = 4 (sub_dim 64)
    "unit768_R16_m96": "unit768", "unit256_R16_m128": "unit256", "unit256_R16_m4": "unit256",
}


class Golden:
    """One golden index fixture: inputs + the reference's expected outputs (tests/golden/gen_golden.py)."""

    def __init__(self, name):
        self.name = name
        data = np.load(GOLDEN / f"data_{INDEX_FIXTURES[name]}.npz")
        self.z = np.load(GOLDEN / f"idx_{name}.npz")
        self.vectors = data["vectors"]
        self.queries = data["queries"]
        self.adj = self.z["adj"]            # on-disk slot order, 0-padded (index.dat)
        self.mem_adj = self.z["mem_adj"]    # in-memory order, 0xFFFFFFFF-padded
        self.medoid = int(self.z["medoid"])
        self.R = int(self.z["R"])
        self.m = int(self.z["m"])
        self.codes = self.z["codes"] if self.m else None
        self.codebook = self.z["codebook"] if self.m else None
        self.cases = json.loads(str(self.z["cases"]))

    def case(self, i):
        c = dict(self.cases[i])
        c["ids"] = self.z[f"c{i}_ids"]
        c["dist"] = self.z[f"c{i}_dist"]
        c["count"] = self.z[f"c{i}_count"]
        c["stats"] = self.z[f"c{i}_stats"]
        if c.get("f64"):
            c["dist64"] = self.z[f"c{i}_dist64"]
        nq = c.get("nq", len(self.queries))
        c["queries"] = self.queries[:nq].astype(np.float64) if c.get("f64") else self.queries[:nq]
        return c


_cache = {}


def load_golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


def all_cases(modes=None, names=None, pred=None):
    out = []
    for name in (names or INDEX_FIXTURES):
        g = load_golden(name)
        for i, c in enumerate(g.cases):
            if (modes is None or c["mode"] in modes) and (pred is None or pred(c)):
                tag = "-".join(f"{k}{v}" for k, v in c.items() if k != "mode")
                out.append(pytest.param(name, i, id=f"{name}-{c['mode']}-{tag}"))
    return out
