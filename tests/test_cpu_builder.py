"""CPU: host-only pieces of the array model (BooleanBufferBuilder) against the reference's hand-written tests
(crates/array/src/array/null_bit_buffer.rs:68-87)."""
import golden_runner as G

HAND = {r["name"]: r for r in G.load("reference_handwritten.json")}


def test_set_bit():
    from arrow_gpu_amd.array import BooleanBufferBuilder

    r = HAND["test_set_bit"]
    b = BooleanBufferBuilder.new_with_capacity(r["capacity"])
    assert len(b.data) == r["bytes"]
    b.set_bit(0)
    assert b.data[0] == r["expected"][0]
    b.set_bit(9)
    assert b.data[1] == r["expected"][1]
    for pos, want in r["is_set"].items():
        assert b.is_set(int(pos)) == want
    b.unset_bit(9)
    assert not b.is_set(9)
    assert BooleanBufferBuilder.is_set_in_slice(b.data, 0)


def test_new_set_with_capacity():
    from arrow_gpu_amd.array import BooleanBufferBuilder

    r = HAND["test_new_set_with_capacity"]
    b = BooleanBufferBuilder.new_set_with_capacity(r["capacity"])
    assert list(b.data) == r["expected"]
    assert not b.contains_nulls and BooleanBufferBuilder.new_with_capacity(3).contains_nulls
