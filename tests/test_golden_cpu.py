"""The committed fixtures of tests/golden/ against the CPU oracle (and the hand KATs against both the oracle and the
numpy twin).  Generator: tests/golden/make_golden.py.  No reference file is read; no GPU is needed."""
import json
import os

import numpy as np
import pytest

import np_twin as tw

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


@pytest.mark.parametrize("name", ["ivfpq_small.npz", "ivfpq_perm.npz", "ivfpq_ties.npz"])
def test_oracle_reproduces_ivfpq_fixture(oracle, name):
    o, z = oracle, load(name)
    tr = int(z["transform"])
    perm = z["perm"] if tr == 2 else None
    ref = o.OracleIndex(o.KIND_IVFPQ, int(z["D"]), int(z["m"]), int(z["ks"]), int(z["C"]), transform=tr, perm=perm)
    ref.set_coarse(z["coarse"])
    ref.set_pq(z["pq"])
    ref.set_w(int(z["w"]))
    cells, codes = ref.encode_batch(z["base"])
    assert np.array_equal(cells, z["cells"]) and np.array_equal(codes, z["codes"])
    ref.add_vectors(z["base"])
    ids, ds, cnt = ref.search_batch(z["queries"], int(z["k"]))
    assert np.array_equal(ids, z["ids"]) and np.array_equal(ds, z["dists"]) and np.array_equal(cnt, z["counts"])
    if tr == 2:
        assert np.array_equal(o.random_permutation(1, int(z["D"])), z["perm"])


def test_oracle_reproduces_pq_fixture(oracle):
    o, z = oracle, load("pq_small.npz")
    ref = o.OracleIndex(o.KIND_PQ, int(z["D"]), int(z["m"]), int(z["ks"]))
    ref.set_pq(z["pq"])
    _, codes = ref.encode_batch(z["base"])
    assert np.array_equal(codes, z["codes"])
    ref.add_vectors(z["base"])
    ids, ds, cnt = ref.search_batch(z["queries"], int(z["k"]))
    assert np.array_equal(ids, z["ids"]) and np.array_equal(ds, z["dists"]) and np.array_equal(cnt, z["counts"])


def test_hand_kats_from_the_fixture_file(oracle):
    o = oracle
    kat = json.load(open(os.path.join(GOLD, "kat_hand.json")))
    k1 = kat["kat1_pq_adc"]
    ix = o.OracleIndex(o.KIND_PQ, D=k1["D"], m=k1["m"], ks=k1["ks"])
    ix.set_pq(np.array(k1["pq"], np.float64))
    for v, code, stored in zip(k1["vectors"], k1["codes"], k1["stored_bytes"]):
        got = ix.encode(np.array(v))
        assert list(got[1]) == code
        assert [o.lib().mmo_transform_to_byte(int(x)) for x in got[1]] == stored
        ix.add_vector(np.array(v))
    q = np.array(k1["query"], np.float64)
    assert ix.lookup_adc(q).tolist() == k1["lut"] == tw.lookup_adc(np.array(k1["pq"], np.float64), q).tolist()
    ids, ds = ix.search(q, k1["k"])
    assert ids.tolist() == k1["ids"] and ds.tolist() == k1["dists"]
    k2 = kat["kat2_ivfpq_residual_sign"]
    iv = o.OracleIndex(o.KIND_IVFPQ, D=k2["D"], m=k2["m"], ks=k2["ks"], C_=k2["C"])
    iv.set_coarse(np.array(k2["coarse"], np.float64))
    iv.set_pq(np.array(k2["pq"], np.float64))
    iv.set_w(k2["w"])
    cell, code = iv.encode(np.array(k2["vector"], np.float64))
    assert cell == k2["cell"] and code.tolist() == k2["code"]
    iv.add_vector(np.array(k2["vector"], np.float64))
    ids, ds = iv.search(np.array(k2["query"], np.float64), k2["k"])
    assert ids.tolist() == k2["ids"] and ds.tolist() == k2["dists"]
    k3 = kat["kat3_jdk"]
    for seed, val in k3["first_next_int"].items():
        assert o.jdk_first_next_int(int(seed)) == val == tw.JRandom(int(seed)).next_int()
    assert o.random_permutation(1, 3).tolist() == k3["perm_seed1_dim3"]
    assert o.random_permutation(1, 8).tolist() == k3["perm_seed1_dim8"] == tw.random_permutation(1, 8).tolist()
    p128 = o.random_permutation(1, 128)
    assert p128[:16].tolist() == k3["perm_seed1_dim128_first16"]
    assert int((np.arange(128) * p128).sum()) == k3["perm_seed1_dim128_sum_i_times_p"]
    k4 = kat["kat4_normalization"]
    assert tw.normalize_l2(np.zeros(5)).tolist() == [1.0] * 5  # Normalization.java:21-37: zero norm -> all ones
    pw = k4["power_0.5"]
    assert (np.sign(pw["in"]) * np.abs(pw["in"]) ** 0.5).tolist() == pw["out"]
