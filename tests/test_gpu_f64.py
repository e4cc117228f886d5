"""float64 queries (the `diskrag search` CLI path, quirk Q8) on the device: dr_search_batch_f64 against the
reference's own float64 outputs (golden) and against the oracle's float64 instantiation on more cases. Needs a GPU."""
import numpy as np
import pytest

from tests.conftest import all_cases, load_golden
from tests.test_gpu_parity import get_index

pytestmark = pytest.mark.gpu


def b64(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("name,ci", all_cases(modes=("M1",), pred=lambda c: c.get("f64")))
def test_m1_f64_bit_exact_vs_reference(name, ci):
    from diskrag_amd import _ffi
    g = load_golden(name)
    c = g.case(ci)
    assert c["queries"].dtype == np.float64
    ix = get_index(name)
    ids, dist, cnt, st = ix.search_batch_f64(c["queries"], c["k"], L=c["L"], beam_width=c["bw"], mode=_ffi.MODE_M1,
                                             band_policy=c.get("policy", 0))
    assert (st["status"] == 0).all()
    assert np.array_equal(cnt, c["count"]) and np.array_equal(ids, c["ids"])
    valid = c["ids"] != 0xFFFFFFFF
    assert np.array_equal(b64(dist[valid]), b64(c["dist64"][valid]))          # float64 distances, bit for bit
    assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), c["stats"])


@pytest.mark.parametrize("name", ["sift128_R64_m32", "sift128_R16_m32", "randn128_R64_m16", "unit1536_R16_m32",
                                  "unit1536_R16_m64", "deep96_R32_m16"])
def test_m1_f64_matches_the_f64_oracle(name):
    """More shapes than the reference goldens cover: ties (sift), the live 0.8/1.2 band under both policies (unit),
    D=96, every size class; the oracle's float64 instantiation is pinned on the goldens above."""
    from diskrag_amd import _ffi
    from oracle import pyoracle as orc
    g = load_golden(name)
    ix = get_index(name)
    q = g.queries[:12].astype(np.float64)
    # perturb below float32 resolution: these queries are NOT representable in float32, so a float32 engine
    # cannot reproduce the float64 distances
    q = q * (1.0 + 1e-9) + 1e-11
    for (L, bw, k, pol) in ((100, 0, 10, 0), (100, 8, 10, 0), (20, 8, 10, 1), (7, 3, 5, 0), (200, 0, 20, 1), (300, 16, 10, 0)):
        ids, dist, cnt, st = ix.search_batch_f64(q, k, L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=pol)
        oi, od, oc, ost = orc.search_batch(g.vectors, g.adj, q, g.medoid, orc.M1, k, L=L, bw=bw, policy=pol,
                                           codes=g.codes, codebook=g.codebook)
        assert (st["status"] == 0).all()
        assert np.array_equal(ids, oi) and np.array_equal(cnt, oc), (L, bw, pol)
        valid = oi != 0xFFFFFFFF
        assert np.array_equal(b64(dist[valid]), b64(od[valid]))
        assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), ost)


def test_m2_f64_on_the_faq_shape():
    """c1's shape (32 x 1536, no PQ, R=16) through M2 with float64 queries: np.linalg.norm's BLAS order is unpinned,
    so 1e-4 like the float32 twin; ids must agree with the oracle wherever its distances are not tied to 1e-9."""
    from diskrag_amd import _ffi
    from oracle import pyoracle as orc
    g = load_golden("faq32_R16_nopq")
    ix = get_index("faq32_R16_nopq")
    q = g.queries.astype(np.float64) * (1.0 + 1e-9)
    ids, dist, cnt, st = ix.search_batch_f64(q, 8, L=100, beam_width=8, mode=_ffi.MODE_M2)
    oi, od, oc, ost = orc.search_batch(g.vectors, g.adj, q, g.medoid, orc.M2, 8, L=100, bw=8)
    assert (st["status"] == 0).all() and np.array_equal(cnt, oc)
    assert np.array_equal(ids, oi)
    valid = oi != 0xFFFFFFFF
    assert np.allclose(dist[valid], od[valid], rtol=1e-4, atol=0)
    assert (np.diff(np.where(valid, dist, np.inf), axis=1) >= 0).all()


def test_f64_entry_rejects_what_it_does_not_serve():
    from diskrag_amd import DiskragHipError, _ffi
    g = load_golden("faq32_R16_nopq")
    ix = get_index("faq32_R16_nopq")
    q = g.queries[:2].astype(np.float64)
    with pytest.raises(DiskragHipError):
        ix.search_batch_f64(q, 5, L=20, beam_width=8, mode=_ffi.MODE_M1)      # no PQ data on this index
    with pytest.raises(DiskragHipError):
        ix.search_batch_f64(q, 5, L=20, beam_width=8, mode=_ffi.MODE_M3)      # CLI modes only
    with pytest.raises(ValueError):
        ix.search_batch_f64(q[:, :100], 5)


@pytest.mark.parametrize("name", ["unit1536_R16_m32", "unit1536_R16_m64"])
def test_literal_coin_flip_on_the_device(name):
    """band_policy = DR_POLICY_COIN(seed0): the reference's coin flip itself, drawn on the device from numpy's MT19937 stream seeded per query --
    float32 queries (dr_search_batch) and float64 queries (dr_search_batch_f64) against the reference run UNPATCHED: ids, distance bits, counters."""
    import json
    from diskrag_amd import _ffi
    from tests.conftest import GOLDEN
    g = load_golden(name)
    ix = get_index(name)
    z = np.load(GOLDEN / f"coinflip_{name}.npz")
    for ci, c in enumerate(json.loads(str(z["cases"]))):
        pol = _ffi.POLICY_COIN(c["seed0"])
        if c["f64"]:
            ids, dist, cnt, st = ix.search_batch_f64(g.queries.astype(np.float64), c["k"], L=c["L"], beam_width=c["bw"], mode=_ffi.MODE_M1, band_policy=pol)
        else:
            ids, dist, cnt, st = ix.search_batch(g.queries, c["k"], L=c["L"], beam_width=c["bw"], mode=_ffi.MODE_M1, band_policy=pol)
        assert int(st["status"].max()) == 0
        assert np.array_equal(ids, z[f"c{ci}_ids"]) and np.array_equal(cnt, z[f"c{ci}_count"]), (name, c)
        valid = z[f"c{ci}_ids"] != 0xFFFFFFFF
        if c["f64"]:
            assert np.array_equal(dist[valid], z[f"c{ci}_dist"][valid])
        else:
            assert np.array_equal(dist[valid].view(np.uint32), z[f"c{ci}_dist"][valid].astype(np.float32).view(np.uint32))
        assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), z[f"c{ci}_stats"]), (name, c)
    # a chunk boundary (more than 64 queries per call): the seed follows the query's index in the CALL
    q = np.concatenate([g.queries] * 4)[:100]
    a = ix.search_batch(q, 10, L=100, beam_width=8, mode=_ffi.MODE_M1, band_policy=_ffi.POLICY_COIN(5))
    from oracle import pyoracle as orc
    w = orc.search_batch(g.vectors, g.adj, q, g.medoid, orc.M1, 10, L=100, bw=8, policy=orc.POLICY_COIN(5), codes=g.codes, codebook=g.codebook)
    assert np.array_equal(a[0], w[0]) and np.array_equal(np.stack([a[3]["steps"], a[3]["visited"], a[3]["exact"], a[3]["pq"]], axis=1), w[3])
    with pytest.raises(_ffi.DiskragHipError):            # the batched paths take the two deterministic branches only
        ix.search_submit(g.queries, 10, L=100, beam_width=8, mode=_ffi.MODE_M1, band_policy=_ffi.POLICY_COIN(5)).wait()
