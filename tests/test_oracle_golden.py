"""CPU: pins the oracle (oracle/agpu_oracle.c + oracle/model.py) on EVERY vector the reference's own tests hold for
the hot path — 202 macro invocations + the hand-written tests, extracted by tools/extract_golden.py."""
import numpy as np
import pytest

import golden_runner as G

VECTORS = G.load("reference_vectors.json")
HAND = G.load("reference_handwritten.json")


@pytest.fixture(scope="module")
def ns(oracle_lib):
    import oracle.model as model

    return model


@pytest.mark.parametrize("rec", VECTORS, ids=[r["name"] + "@" + r["ref"].split("/")[1] for r in VECTORS])
def test_reference_vector(ns, rec):
    G.run_vector(ns, rec)


def test_vector_inventory():
    kinds = {}
    for r in VECTORS:
        kinds[r["kind"]] = kinds.get(r["kind"], 0) + 1
    assert len(VECTORS) == 202
    assert kinds == {"scalar_op": 19, "array_op": 93, "unary_op": 32, "sum": 6, "broadcast": 9, "cast": 22,
                     "bitcast": 1, "merge": 9, "put": 6, "take": 5}


@pytest.mark.parametrize("rec", [r for r in HAND if r["kind"] not in ("builder_set_bit", "builder_new_set")],
                         ids=lambda r: r["name"])
def test_reference_handwritten(ns, oracle_lib, rec):
    out = G.run_handwritten(ns, rec)
    if rec["kind"] == "from_optional_and_merge":
        a, b = out
        O = oracle_lib
        assert list(a.validity[:1]) == rec["a_validity"]
        assert list(b.validity[:1]) == rec["b_validity"]
        merged = O.validity_and(b.validity, a.validity, a.len)
        assert list(merged[:1]) == rec["merged_validity"]
