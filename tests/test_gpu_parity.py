"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import oracle
import zpaqsharp_amd as z
from tests import util

pytestmark = pytest.mark.gpu


def test_device_tables_match_oracle(ctx):
    for dev, ora in zip(ctx.device_tables(), oracle.tables()):
        assert np.array_equal(dev, ora)


@pytest.mark.parametrize("model", ["l1", "min", "mid", "max", "max+e8e9"])
def test_single_block_64k(ctx, model):
    data = util.x86ish(65536) if "e8e9" in model else util.text(65536)
    stream = util.block(model, data)
    assert oracle.decompress(stream) == data
    got = ctx.decompress(stream, verify_sha1=True)
    assert got.tobytes() == data


def test_multi_block_mixed_models(ctx):
    parts, plain = [], []
    for i, model in enumerate(["l1", "mid", "min", "l1", "max", "l1"]):
        d = util.text(3000 + 777 * i, seed=10 + i)
        parts.append(util.block(model, d))
        plain.append(d)
    stream = b"junk before" + b"".join(parts) + b"trailing junk"
    want = b"".join(plain)
    assert oracle.decompress(stream) == want
    assert ctx.decompress(stream, verify_sha1=True).tobytes() == want


@pytest.mark.parametrize("n", [0, 1, 2, 7, 255, 256, 4097])
def test_ragged_sizes(ctx, n):
    d = util.text(n, seed=n + 3)
    for model in ("l1", "mid"):
        s = util.block(model, d)
        assert ctx.decompress(s, verify_sha1=True).tobytes() == d
