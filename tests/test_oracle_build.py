"""The builder restatement (oracle/pybuild.py) pinned on the reference's own output: build_vamana_index_cython
(pydiskann/cython_utils.pyx:269-369) run by tests/golden/gen_golden_build.py on integer-valued points with Python's RNG
seeded. Adjacency lists must match row for row -- including the FIFO result list of the builder's greedy search and the
stale vector reads of its robust prune (both documented in oracle/pybuild.py)."""
import numpy as np
import pytest

from tests.conftest import GOLDEN


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_builder_restatement_reproduces_the_reference_graph(tag):
    from oracle import pybuild
    z = np.load(GOLDEN / "build_int32.npz")
    pts = z["points"]
    R, L, medoid, seed = (int(v) for v in z[f"params_{tag}"])
    alpha = float(z[f"alpha_{tag}"])
    adj = pybuild.build_vamana_index(pts, R, L, alpha, medoid, seed)
    want, deg = z[f"adj_{tag}"], z[f"deg_{tag}"]
    for i, row in enumerate(adj):
        assert row == want[i, :deg[i]].tolist(), (tag, i)


def test_the_stale_read_quirk_is_live_on_the_fixture():
    """How often the reference's robust prune differs from the textbook one on these inputs (documentation of the quirk:
    rows pruned with stale_reads on/off, same candidates)."""
    from oracle import pybuild
    z = np.load(GOLDEN / "build_int32.npz")
    pts = z["points"]
    rs = np.random.RandomState(3)
    differ = 0
    for _ in range(200):
        p = int(rs.randint(len(pts)))
        cands = rs.choice(len(pts), size=40, replace=False)
        a = pybuild.robust_prune_fast(pts, p, cands, 1.2, 8, stale_reads=True)
        b = pybuild.robust_prune_fast(pts, p, cands, 1.2, 8, stale_reads=False)
        assert set(b) <= set(a)            # the textbook row is always a subset: stale reads only ADD neighbours
        differ += a != b
    assert differ > 0


def test_pq_prune_restatement_properties():
    """oracle/pybuild.robust_prune_pq (the checker of the PQ-only builder's prune kernel): picks are distinct candidates other
    than the point, in ascending code-word distance to the point; no pick is occluded by an earlier one; every candidate that was
    left out is occluded by a pick that precedes it in the sorted order (or the row was full); duplicate code words (distance 0
    to each other) never both survive."""
    from oracle import pybuild, pyoracle as orc
    rs = np.random.RandomState(3)
    m, sd, n = 8, 6, 400
    cb = rs.randn(m, 256, sd).astype(np.float32)
    codes = rs.randint(0, 8, size=(n, m)).astype(np.uint8)        # few code values: repeated code words, many ties
    codes[7] = codes[9]

    def dist(a, ids):
        dec = np.concatenate([cb[j, codes[a, j]] for j in range(m)]).astype(np.float32)
        return orc.adc(orc.build_lut(cb, dec), codes[list(ids)])[0]

    for trial in range(25):
        p = int(rs.randint(n))
        cand = rs.choice(n, size=int(rs.choice([5, 60, 200])), replace=False).astype(np.uint32)
        if trial % 3 == 0:
            cand = np.concatenate([cand, [7, 9, p, cand[0]]]).astype(np.uint32)
        alpha, R = float(rs.choice([1.0, 1.2])), int(rs.choice([4, 32]))
        picks = pybuild.robust_prune_pq(cb, codes, p, cand, alpha, R).tolist()
        pool = [int(c) for c in dict.fromkeys(cand.tolist()) if c != p]
        assert len(picks) == len(set(picks)) <= R and set(picks) <= set(pool) and p not in picks
        dp = dict(zip(pool, dist(p, pool)))
        keys = [(dp[c].view(np.uint32), c) for c in picks]
        assert keys == sorted(keys)
        a32 = np.float32(alpha)
        for i, s in enumerate(picks):
            later = picks[i + 1:]
            if later:
                ds = dist(s, later)
                assert all(not (np.float32(a32 * ds[t]) <= dp[c]) for t, c in enumerate(later))
        if len(picks) < R:
            for c in pool:
                if c in picks:
                    continue
                before = [s for s in picks if (dp[s].view(np.uint32), s) < (dp[c].view(np.uint32), c)]
                assert before and any(np.float32(a32 * d) <= dp[c] for d in dist(c, before)), (trial, c)
