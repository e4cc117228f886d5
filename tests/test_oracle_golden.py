"""The CPU oracle (oracle/) against golden vectors produced by the reference itself.

This is what pins the oracle: every later parity test compares the HIP path with the oracle, so the
oracle must first reproduce the reference's own outputs (tests/golden/gen_golden.py).
"""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tests.conftest import GOLDEN, INDEX_FIXTURES, all_cases, load_golden

MODE = {"M1": orc.M1, "M2": orc.M2, "M3": orc.M3, "M4": orc.M4}
PQ_FIXTURES = [n for n in INDEX_FIXTURES if "nopq" not in n]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ------------------------------------------------------------------------------------ kernel-level (K1)

@pytest.mark.parametrize("d", [7, 64, 96, 128, 130, 960, 1536])
def test_exact_distance_matches_numpy_sum_bits(d):
    """A1: np.sum(diff*diff) (search_engine.py:378-379) is reproduced bit for bit."""
    z = np.load(GOLDEN / "k_scalar.npz")
    a, b, want = z[f"a{d}"], z[f"b{d}"], z[f"npsum_{d}"]
    got = np.array([orc.sqdist(a[i], b[i]) for i in range(len(a))], dtype=np.float32)
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("d", [7, 64, 96, 128, 130, 960, 1536])
def test_scalar_kernels_within_reference_tolerance(d):
    """C8: l2_distance_fast_cython / cosine_similarity_cython, at the reference's own tolerance
    (scripts/test_pydiskann_cython.sh:50-54: rtol 1e-5, atol 1e-6)."""
    z = np.load(GOLDEN / "k_scalar.npz")
    a, b = z[f"a{d}"], z[f"b{d}"]
    l2 = np.array([orc.l2_seq(a[i], b[i]) for i in range(len(a))])
    cs = np.array([orc.cosine_dist(a[i], b[i]) for i in range(len(a))])
    np.testing.assert_allclose(l2, z[f"l2_{d}"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(cs, z[f"cos_{d}"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", list(INDEX_FIXTURES))
def test_k1_exact_bits(name):
    g = load_golden(name)
    nodes, want = g.z["k1_nodes"], g.z["k1_exact"]
    for qi in range(want.shape[0]):
        got = np.array([orc.sqdist(g.vectors[n], g.queries[qi]) for n in nodes], dtype=np.float32)
        assert np.array_equal(bits(got), bits(want[qi]))
        # norm (M2/M4 distance) goes through BLAS in the reference: tolerance only
        np.testing.assert_allclose(np.sqrt(got), g.z["k1_norm"][qi], rtol=1e-6)


@pytest.mark.parametrize("name", PQ_FIXTURES)
def test_k1_lut_and_adc_bits(name):
    """A2 + A3: distance table and ADC sums/sqrt, bit for bit (fast_pq.py:294-333)."""
    g = load_golden(name)
    nodes = g.z["k1_nodes"]
    for qi in range(g.z["k1_lut"].shape[0]):
        lut = orc.build_lut(g.codebook, g.queries[qi])
        assert np.array_equal(bits(lut), bits(g.z["k1_lut"][qi]))
        sq, rt = orc.adc(lut, g.codes[nodes])
        assert np.array_equal(bits(sq), bits(g.z["k1_adc_sq"][qi]))
        assert np.array_equal(bits(rt), bits(g.z["k1_adc"][qi]))


def test_lut_f64_query():
    """Q8: a float64 query makes numpy compute the table in f64 and round on store."""
    g = load_golden("randn128_R16_m32")
    q = g.queries[0].astype(np.float64)
    lut = orc.build_lut(g.codebook, q)
    want = np.empty_like(lut)
    sd = g.codebook.shape[2]
    for j in range(g.m):
        diff = g.codebook[j] - q[j * sd:(j + 1) * sd][np.newaxis, :]
        want[j] = np.sum(diff * diff, axis=1)
    assert np.array_equal(bits(lut), bits(want))


# ------------------------------------------------------------------------------------ search parity

def run_case(g, c):
    mode = MODE[c["mode"]]
    flags = 0
    if c["mode"] == "M3" and c["use_pq"]:
        flags |= orc.F_USE_PQ
    if c["mode"] == "M4" and c.get("cython"):
        flags |= orc.F_CYTHON
    adj = g.adj if c["mode"] in ("M1", "M2") else g.mem_adj
    return orc.search_batch(g.vectors, adj, c["queries"], g.medoid, mode, c["k"], L=c.get("L", 100),
                            bw=c.get("bw", 0) or 0, policy=c.get("policy", 0), flags=flags,
                            codes=g.codes, codebook=g.codebook)


@pytest.mark.parametrize("name,ci", all_cases(modes=("M1",)))
def test_m1_bit_exact(name, ci):
    """M1 (_pq_accelerated_graph_search): ids, float bits of distances, hit counts and all four counters."""
    g = load_golden(name)
    c = g.case(ci)
    ids, dist, cnt, stats = run_case(g, c)
    assert np.array_equal(cnt, c["count"])
    assert np.array_equal(ids, c["ids"])
    if c.get("f64"):
        assert np.array_equal(dist.view(np.uint64), c["dist64"].view(np.uint64))
    else:
        assert np.array_equal(bits(dist), bits(c["dist"]))
    assert np.array_equal(stats, c["stats"])


@pytest.mark.parametrize("name,ci", all_cases(modes=("M3",), pred=lambda c: c["use_pq"]))
def test_m3_pq_bit_exact(name, ci):
    """M3 with use_pq=True: ADC-only traversal, bit exact including the heap-layout tie order."""
    g = load_golden(name)
    c = g.case(ci)
    ids, dist, cnt, _ = run_case(g, c)
    assert np.array_equal(cnt, c["count"])
    assert np.array_equal(ids, c["ids"])
    assert np.array_equal(bits(dist), bits(c["dist"]))


def near_tie_mismatches(ids, want_ids, want_dist, cnt, rtol):
    """Count id mismatches that are NOT explained by a near-tie in the reference's own distances
    (library-specific summation order may flip pairs whose distances agree to rtol)."""
    bad = 0
    for qi in range(ids.shape[0]):
        n = cnt[qi]
        if np.array_equal(ids[qi, :n], want_ids[qi, :n]):
            continue
        for j in range(n):
            if ids[qi, j] != want_ids[qi, j]:
                wd = want_dist[qi, j]
                close = np.abs(want_dist[qi, :n] - wd) <= rtol * max(abs(wd), 1e-30)
                if ids[qi, j] not in want_ids[qi, :n][close]:
                    bad += 1
    return bad


@pytest.mark.parametrize("name,ci", all_cases(modes=("M2",)))
def test_m2_ids_and_distances(name, ci):
    """M2 (beam_search_from_disk): np.linalg.norm goes through BLAS sdot, so distances are held to 1e-4
    relative and ids must agree except across near-ties."""
    g = load_golden(name)
    c = g.case(ci)
    ids, dist, cnt, _ = run_case(g, c)
    assert np.array_equal(cnt, c["count"])
    assert near_tie_mismatches(ids, c["ids"], c["dist"], cnt, 1e-5) == 0
    for qi in range(len(cnt)):
        n = cnt[qi]
        np.testing.assert_allclose(np.sort(dist[qi, :n]), np.sort(c["dist"][qi, :n]), rtol=1e-4)


@pytest.mark.parametrize("name,ci", all_cases(modes=("M3",), pred=lambda c: not c["use_pq"]))
def test_m3_exact_ids(name, ci):
    g = load_golden(name)
    c = g.case(ci)
    ids, dist, cnt, _ = run_case(g, c)
    assert np.array_equal(cnt, c["count"])
    assert near_tie_mismatches(ids, c["ids"], c["dist"], cnt, 1e-5) == 0
    np.testing.assert_allclose(dist[:, :], c["dist"], rtol=1e-4)


@pytest.mark.parametrize("name,ci", all_cases(modes=("M4",)))
def test_m4_ids(name, ci):
    """M4 returns ids only; distances are BLAS norms (or -ffast-math squared L2 for the Cython twin)."""
    g = load_golden(name)
    c = g.case(ci)
    ids, dist, cnt, _ = run_case(g, c)
    assert np.array_equal(cnt, c["count"])
    # an id mismatch is only acceptable when the oracle's own distances at the two ranks are a near-tie
    for qi, j in zip(*np.nonzero(ids != c["ids"])):
        row = list(ids[qi])
        assert c["ids"][qi, j] in row, (name, ci, qi, j)
        pos = row.index(c["ids"][qi, j])
        assert abs(dist[qi, pos] - dist[qi, j]) <= 1e-5 * abs(dist[qi, j]), (name, ci, qi, j)


COSINE_FIXTURES = ["randn128_R16_m32", "unit1536_R16_m32", "deep96_R32_m16"]


@pytest.mark.parametrize("name", COSINE_FIXTURES)
def test_m3_cosine_against_the_reference(name):
    """M3 with distance_metric='cosine' (compute_query_distance -> cosine_similarity_cython, vamana_graph.py:324-329,
    cython_utils.pyx:53-70), goldens from tests/golden/gen_golden_cosine.py: the reference sums in float32 under
    -ffast-math (order unpinned), so ids are held equal (no near-ties on these fixtures) and distances to 1e-5."""
    import json
    from tests.conftest import GOLDEN
    g = load_golden(name)
    z = np.load(GOLDEN / f"cos_{name}.npz")
    for ci, c in enumerate(json.loads(str(z["cases"]))):
        ids, dist, cnt, _ = orc.search_batch(g.vectors, g.mem_adj, g.queries, g.medoid, orc.M3, c["k"], L=c["k"], bw=c["bw"], flags=orc.F_COSINE)
        assert np.array_equal(ids, z[f"c{ci}_ids"]) and np.array_equal(cnt, z[f"c{ci}_count"])
        valid = ids != 0xFFFFFFFF
        assert np.allclose(dist[valid], z[f"c{ci}_dist"][valid], rtol=0, atol=1e-5)


@pytest.mark.parametrize("name", PQ_FIXTURES)
def test_pq_mode_without_a_visited_set_equals_the_visited_set_statement(name):
    """Round 4: DR_F_NO_VISITED_SET -- the engine's PQ traversal (mode 5, no reference counterpart) without a visited set: every
    neighbour is scored and one that would enter the list is looked up in the list. Same ids, distances, hit counts and
    expansions as the statement with an explicit visited set, on every PQ fixture, trimmed and not, short and long lists, with
    the exact rerank; only the evaluation counters grow (a node met again after it left the list is scored again)."""
    from oracle import pyoracle as orc
    g = load_golden(name)
    for (L, bw, k) in ((100, 8, 10), (40, 0, 10), (10, 3, 10), (200, 16, 25), (64, 8, 64), (5, 0, 5), (1, 1, 1)):
        for fl in (0, orc.F_RERANK):
            a = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.PQ, k, L=L, bw=bw, flags=fl | orc.F_NO_VISITED_SET, codes=g.codes, codebook=g.codebook)
            b = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.PQ, k, L=L, bw=bw, flags=fl,
                                 codes=g.codes, codebook=g.codebook)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]), (name, L, bw, k, fl)
            va = a[0] != 0xFFFFFFFF
            assert np.array_equal(a[1][va].view(np.uint64), b[1][va].view(np.uint64))
            assert np.array_equal(a[3][:, 0], b[3][:, 0])                                  # expansions
            assert (a[3][:, 3] >= b[3][:, 3]).all()                                        # evaluations >= distinct nodes scored
        # the same on the in-memory adjacency (no 0-padding: no repeated ids in a row)
        a = orc.search_batch(g.vectors, g.mem_adj, g.queries, g.medoid, orc.PQ, k, L=L, bw=bw, flags=orc.F_NO_VISITED_SET, codes=g.codes, codebook=g.codebook)
        b = orc.search_batch(g.vectors, g.mem_adj, g.queries, g.medoid, orc.PQ, k, L=L, bw=bw, codes=g.codes, codebook=g.codebook)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[3][:, 0], b[3][:, 0])


def test_mt19937_restatement_is_numpys_legacy_generator():
    """np.random.seed(int) + np.random.random(): what the reference's coin flip draws from (search_engine.py:393-395)."""
    for seed in (0, 1, 12345, 2 ** 31 + 5, 2 ** 32 - 1):
        np.random.seed(seed)
        want = np.array([np.random.random() for _ in range(1500)])
        assert np.array_equal(orc.mt_doubles(seed, 1500), want), seed


@pytest.mark.parametrize("name", ["unit1536_R16_m32", "unit1536_R16_m64"])
def test_literal_coin_flip_against_the_unpatched_reference(name):
    """band policy 2: the rerank policy's coin flip itself. The goldens are the reference run UNPATCHED with np.random.seed(seed0 + qi) before
    query qi (tests/golden/gen_golden_coinflip.py): ids, distance bits (float32 and float64 queries) and the four counters."""
    import json
    g = load_golden(name)
    z = np.load(GOLDEN / f"coinflip_{name}.npz")
    for ci, c in enumerate(json.loads(str(z["cases"]))):
        q = g.queries.astype(np.float64) if c["f64"] else g.queries
        w = orc.search_batch(g.vectors, g.adj, q, g.medoid, orc.M1, c["k"], L=c["L"], bw=c["bw"], policy=orc.POLICY_COIN(c["seed0"]),
                             codes=g.codes, codebook=g.codebook)
        assert np.array_equal(w[0], z[f"c{ci}_ids"]) and np.array_equal(w[2], z[f"c{ci}_count"]) and np.array_equal(w[3], z[f"c{ci}_stats"]), (name, c)
        valid = z[f"c{ci}_ids"] != 0xFFFFFFFF
        if c["f64"]:
            assert np.array_equal(w[1][valid], z[f"c{ci}_dist"][valid])
        else:
            assert np.array_equal(w[1][valid].astype(np.float32).view(np.uint32), z[f"c{ci}_dist"][valid].astype(np.float32).view(np.uint32))
        assert z[f"c{ci}_draws"].sum() > 0          # the band was live: coins were flipped
