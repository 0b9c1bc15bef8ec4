"""CPU: oracle/sampling_ref.py against tests/golden/sampling.npz -- what transformers' own TemperatureLogitsWarper ->
TopKLogitsWarper -> TopPLogitsWarper keep on committed logits rows (generator: tests/golden/gen_golden_sampling.py; the
reference's default decoding mode, cli/SparkTTS.py:166-168,197-204).  Equality is exact: same torch ops, same order."""
import os

import numpy as np
import pytest
import torch

from oracle.sampling_ref import sampling_probs

CASES = [("tiny", 5), ("big", 3), ("tie", 2), ("edge", 5)]


def load_row(g, name):
    if name != "big":
        return g[f"{name}.logits"]
    seed, v = (int(x) for x in g["big.seed"])
    row = (np.random.Generator(np.random.PCG64(seed)).standard_normal(v) * float(g["big.std"])).astype(np.float32)
    chk = np.array([float(row.astype(np.float64).sum()), float(np.abs(row).astype(np.float64).sum())])
    np.testing.assert_allclose(chk, g["big.checksum"], rtol=0, atol=0)   # the row the fixture was made from
    return row


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "sampling.npz"))


@pytest.mark.parametrize("name,n", CASES)
def test_restated_warper_chain_equals_transformers(g, name, n):
    row = load_row(g, name)
    assert len(g[f"{name}.params"]) == n
    for i, (t, k, p) in enumerate(g[f"{name}.params"]):
        got = sampling_probs(torch.from_numpy(row), float(t), int(k), float(p)).numpy()
        ids = np.nonzero(got)[0]
        np.testing.assert_array_equal(ids, g[f"{name}.{i}.ids"])
        np.testing.assert_array_equal(got[ids], g[f"{name}.{i}.probs"])


def test_fixture_holds_the_cases_it_claims(g):
    # a tie at the k-th value keeps MORE than k tokens
    assert len(g["tie.1.ids"]) == 51 and int(g["tie.params"][1][1]) == 50
    # the nucleus cut falls on either side of the cumulative probabilities 0.0300 and 0.0512
    assert [len(g[f"edge.{i}.ids"]) for i in range(4)] == [7, 6, 6, 5]
    # top_k = 1 leaves one certain token
    assert len(g["tiny.4.ids"]) == 1 and float(g["tiny.4.probs"][0]) == 1.0
