"""Index directory writer/reader (diskrag_amd.persist) against what the reference itself wrote: the goldens hold
both the in-memory neighbour lists (`mem_adj`, `deg`) and the slots read back from the reference's index.dat (`adj`).
Layout checks follow the reference's own file test (test_disk_write_verify.py:74-83, 150-176): record i at byte
i*4*(D+R), D float32 then R uint32. CPU only."""
import json
import pickle

import numpy as np
import pytest

from diskrag_amd import persist
from tests.conftest import load_golden

NAMES = ["sift128_R16_m32", "randn128_R64_m16", "randn128_R16_m32", "unit1536_R16_m32", "deep96_R32_m16", "faq32_R16_nopq"]


@pytest.mark.parametrize("name", NAMES)
def test_packed_slots_equal_the_reference_file(name):
    g = load_golden(name)
    slots = persist.pack_neighbor_lists(g.z["mem_adj"], g.z["deg"], g.R)
    assert slots.dtype == np.uint32 and np.array_equal(slots, g.adj)
    # the same lists without a degree array (0xFFFFFFFF marks the unused tail)
    assert np.array_equal(persist.pack_neighbor_lists(g.z["mem_adj"], None, g.R), g.adj)


def test_record_layout_and_round_trip(tmp_path):
    g = load_golden("sift128_R16_m32")
    meta = persist.write_index(tmp_path / "index", g.vectors, g.z["mem_adj"], g.medoid, R=g.R, degrees=g.z["deg"],
                               codes=g.codes, codebook=g.codebook, build_params={"L": 40, "alpha": 1.2, "seed": 7})
    n, d = g.vectors.shape
    raw = (tmp_path / "index" / "index.dat").read_bytes()
    rs = 4 * (d + g.R)
    assert len(raw) == n * rs
    for i in (0, 1, 777, n - 1):   # byte-level reader, as MMapNodeReader.get_node does (diskann_persist.py:219-230)
        vec = np.frombuffer(raw, dtype=np.float32, count=d, offset=i * rs)
        nbr = np.frombuffer(raw, dtype=np.uint32, count=g.R, offset=i * rs + 4 * d)
        assert np.array_equal(vec.view(np.uint32), g.vectors[i].view(np.uint32)) and np.array_equal(nbr, g.adj[i])
    assert (tmp_path / "index" / "pq_codes.bin").read_bytes() == g.codes.tobytes()
    on_disk = json.loads((tmp_path / "index" / "meta.json").read_text())
    for key in ("N", "D", "R", "medoid_idx", "use_pq", "n_subvectors"):      # what search reads (T4)
        assert key in on_disk
    assert (on_disk["N"], on_disk["D"], on_disk["R"], on_disk["medoid_idx"], on_disk["use_pq"], on_disk["n_subvectors"]) == \
           (n, d, g.R, g.medoid, True, g.m)
    assert on_disk["seed"] == 7 and on_disk["L"] == 40 and meta["pq_centroids"] == 256
    back = persist.read_index(tmp_path / "index")
    assert np.array_equal(back.vectors, g.vectors) and np.array_equal(back.adjacency, g.adj)
    assert np.array_equal(back.codes, g.codes)
    assert np.array_equal(back.codebook.view(np.uint32), g.codebook.astype(np.float32).view(np.uint32))


def test_long_lists_are_cut_and_short_ones_zero_padded(tmp_path):
    vec = np.arange(5 * 8, dtype=np.float32).reshape(5, 8)
    lists = np.array([[1, 2, 3, 4], [0, 2, 0xFFFFFFFF, 0xFFFFFFFF], [4, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF],
                      [0xFFFFFFFF] * 4, [3, 2, 1, 0]], dtype=np.uint32)
    slots = persist.pack_neighbor_lists(lists, None, 3)
    assert slots.tolist() == [[1, 2, 3], [0, 2, 0], [4, 0, 0], [0, 0, 0], [3, 2, 1]]
    wide = persist.pack_neighbor_lists(lists, np.array([4, 2, 1, 0, 4]), 6)
    assert wide.tolist() == [[1, 2, 3, 4, 0, 0], [0, 2, 0, 0, 0, 0], [4, 0, 0, 0, 0, 0], [0] * 6, [3, 2, 1, 0, 0, 0]]
    with pytest.raises(ValueError):   # a hole inside a list is not a list
        persist.pack_neighbor_lists(np.array([[1, 0xFFFFFFFF, 2]], dtype=np.uint32), None, 3)
    persist.write_index(tmp_path / "i", vec, lists, medoid=2, R=3)
    back = persist.read_index(tmp_path / "i")
    assert back.meta["use_pq"] is False and back.codes is None and back.adjacency.tolist() == slots.tolist()


def test_reader_rejects_a_sheared_file(tmp_path):
    g = load_golden("faq32_R16_nopq")
    persist.write_index(tmp_path / "i", g.vectors, g.adj, g.medoid)
    n, d = g.vectors.shape
    with pytest.raises(ValueError):
        persist.read_records(tmp_path / "i" / "index.dat", n, d, g.R + 1)
    with pytest.raises(ValueError):
        persist.write_index(tmp_path / "j", g.vectors, g.adj, medoid=n)           # medoid out of range
    bad = g.adj.copy(); bad[0, 0] = n
    with pytest.raises(ValueError):
        persist.write_index(tmp_path / "k", g.vectors, bad, g.medoid)             # neighbour id out of range


def test_pq_model_pickle_has_the_reference_layout(tmp_path):
    pytest.importorskip("sklearn")
    g = load_golden("randn128_R64_m16")
    persist.write_index(tmp_path / "i", g.vectors, g.adj, g.medoid, codes=g.codes, codebook=g.codebook)
    with open(tmp_path / "i" / "pq_model.pkl", "rb") as f:
        model = pickle.load(f)
    # keys and checks of _load_new_format_pq (diskann_persist.py:127-185)
    assert model["model_type"] == "DiskANNPQ" and model["is_fitted"] is True
    assert (model["n_subvectors"], model["n_centroids"], model["sub_dim"]) == g.codebook.shape
    assert model["means_"] is None and model["stds_"] is None and len(model["kmeans_list"]) == g.m
    sd = g.codebook.shape[2]
    for j, km in enumerate(model["kmeans_list"]):
        assert km.cluster_centers_.shape == (256, sd) and km.cluster_centers_.dtype == np.float32
        assert np.array_equal(km.cluster_centers_, g.codebook[j])
    # the reference's encode is kmeans.predict per sub-space (fast_pq.py:252-267): the hand-built objects must serve it
    sub = g.vectors[:50, :sd]
    lab = model["kmeans_list"][0].predict(sub)
    d2 = ((sub[:, None, :].astype(np.float64) - g.codebook[0][None].astype(np.float64)) ** 2).sum(-1)
    assert np.array_equal(d2[np.arange(50), lab], d2.min(axis=1))
    # read_index falls back to the pickle when the raw codebook is absent
    (tmp_path / "i" / "pq_codebook.f32").unlink()
    back = persist.read_index(tmp_path / "i")
    assert np.array_equal(back.codebook, g.codebook.astype(np.float32))
