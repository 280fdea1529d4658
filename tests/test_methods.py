"""The reference's method strings (LibZPAQ.makeConfig, LibZPAQ.cs:388-1044): generated configs, the LZ77 / BWT / E8E9
pre-processors (LZBuffer.cs formats) and their PCOMP programs, checked on the oracle (CPU) and on the GPU."""
import hashlib

import numpy as np
import pytest

import oracle
import zpaqsharp_amd as z
from tests import util
from tools import methods
from zpaqsharp_amd import zpaql

METHODS = [
    "x0,1,4,0,3,16",                 # level 1: lazy2, bit-packed LZ77, no model (n = 0: stored)      LibZPAQ.cs:427-572
    "x0,5,4,0,3,16",                 # ... + E8E9 at end of segment
    "x6,1,4,0,3,24",                 # ... 64 MiB block: low offset bits travel separately (rb = 2)
    "x0,2,12,0,7,16,1c0,0,511i2",    # level 2: lzpre, byte-aligned LZ77 + ICM/ISSE over the parse state  :575-639
    "x0,6,5,0,3,16c0,0,511",         # ... + E8E9, CM
    "x0,3ci1",                       # level 3: bwtrle, inverse BWT at end of segment                 :642-795
    "x0,7ci1",                       # ... + E8E9
    "x5,3ci1",                       # ... blocks > 16 MiB: the slower list traversal
    "x5,7ci1",
    "x0,4ci1,1,1,1,2am",             # E8E9 alone in front of an ICM-ISSE chain + MATCH + MIX         :802-826
    "x0,0c0,0,255w1i1c256ci1,1,1,1,1,1,2ac0,2,0,255i1c0,3,0,0,255i1c0,4,0,0,0,255i1mm16ts19t0",   # the level-5 recipe
]


def _data(n=9000):
    return util.text(n * 2 // 3, 3) + util.x86ish(n // 4, 4) + b"ab" * (n // 24)


@pytest.mark.parametrize("method", METHODS)
def test_method_configs_on_the_oracle(method):
    data = _data()
    text, args = methods.make_config(method)
    m = zpaql.assemble(text)
    pre = methods.preprocess(data, args)
    if m.pcomp:                                                  # PCOMP inverts the pre-processor
        assert oracle.run_pcomp(m.pcomp, pre, m.header[4], m.header[5], cap=1 << 20) == data
    s = methods.compress_block(method, data)
    assert oracle.decompress(s, cap=1 << 20) == data
    sc = z.scan(s)
    assert sc.n_blocks == 1 and sc.blocks[0].usize_hint == len(data) and bytes(sc.segments[0].sha1) == hashlib.sha1(data).digest()


def test_preprocessor_formats_on_edge_inputs():
    for data in (b"", b"a", b"abcd" * 2, bytes(300), bytes(range(256)) * 2):
        for method in ("x0,1,4,0,3,16", "x0,2,4,0,3,16c0,0,511", "x0,3ci1", "x0,7ci1"):
            text, args = methods.make_config(method)
            m = zpaql.assemble(text)
            pre = methods.preprocess(data, args)
            assert oracle.run_pcomp(m.pcomp, pre, m.header[4], m.header[5], cap=1 << 16) == data, (method, len(data))


@pytest.mark.gpu
@pytest.mark.parametrize("method", METHODS)
def test_method_streams_on_the_gpu(ctx, method):
    data = _data(30000)
    s = methods.compress_block(method, data)
    for kernel in (0, 1):
        assert ctx.decompress(s, verify_sha1=True, kernel=kernel).tobytes() == data, kernel
    m, _ = methods.model_of(method)
    if m.pcomp:
        assert ctx.block_pcomp(s, 0)[2:] == m.pcomp


@pytest.mark.gpu
def test_method_streams_multi_block(ctx):
    parts = [(mt, _data(4000 + 700 * i)) for i, mt in enumerate(METHODS[:7])]
    s = b"".join(methods.compress_block(mt, d) for mt, d in parts)
    assert ctx.decompress(s, verify_sha1=True).tobytes() == b"".join(d for _, d in parts)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["x0,1,4,0,3,16", "x6,1,4,0,3,24", "x0,5,4,0,3,16", "x6,5,4,0,3,24", "x0,2,12,0,7,16",
                                    "x0,6,5,0,3,16c0,0,511", "x0,4ci1,1,1,1,2am", "x0,3ci1", "x0,7ci1"])
def test_translated_pcomps_on_arbitrary_input(ctx, method):
    """The ahead-of-time translations (zh_zpaql_pcomp.h) must do what the interpreter does on ANY input, not only on
    well-formed LZ77 code: the post-processor is fed bytes no encoder would write, and the device's output (or its
    error) is compared with the oracle's interpreter.  Both sides run with an instruction budget: on a block whose BWT
    index is garbage bwtrle's list traversal need not terminate.  The two budgets do not count alike (the interpreter
    counts instructions, a translation its backward jumps), so when the ORACLE runs out only termination is asked of
    the device."""
    budget = 20_000_000
    oracle.set_zpaql_budget(budget)
    try:
        _arbitrary_input_cases(ctx, method, budget)
    finally:
        oracle.set_zpaql_budget(0)


def _arbitrary_input_cases(ctx, method, budget):
    rng = np.random.default_rng(len(method) * 131 + 7)
    feeds = [rng.integers(0, 256, 6000, dtype=np.uint8).tobytes(),
             rng.integers(0, 4, 6000, dtype=np.uint8).tobytes(),
             bytes(6000), rng.integers(0, 256, 40, dtype=np.uint8).tobytes(),
             methods.preprocess(_data(5000), methods.parse_args(method)[1])[:-7] if methods.parse_args(method)[1][1] & 3 else _data(3000)]
    for i, feed in enumerate(feeds):
        s = methods.compress_block(method, b"x" * 16, pre=feed)        # size comment / SHA-1 describe something else: not verified
        try:
            want, werr = oracle.decompress(s, cap=1 << 22), None
        except oracle.OracleError as e:
            want, werr = None, str(e)
        for kernel in (0, 1):
            try:
                got, gerr = ctx.decompress(s, out_cap=1 << 22, kernel=kernel, zpaql_budget=budget).tobytes(), None
            except z.ZpaqError as e:
                got, gerr = None, str(e)
            if werr is not None and "budget" in werr:
                continue                                  # the device came back, with output or an error: all that is asked
            assert (got is None) == (want is None), (method, i, kernel, werr, gerr)
            if want is not None:
                assert got == want, (method, i, kernel)
            else:
                assert werr.split(":")[-1].strip() in gerr or gerr.split(":")[-1].strip() in werr, (method, i, werr, gerr)


def test_generated_zpaql_translations_are_current():
    """zh_zpaql_native.h / zh_zpaql_pcomp.h are what tools/gen_zpaql_native.py makes of models.py / methods.py today,
    and every post-processor the reference's method strings generate has a structural match among the translations."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_zpaql_native", os.path.join(root, "tools", "gen_zpaql_native.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    for name, fn in gen.OUTPUTS.items():
        with open(os.path.join(root, "zpaqsharp_amd", "csrc", name)) as f:
            assert f.read() == fn(), f"{name} is stale: run python tools/gen_zpaql_native.py"
    skeletons = set()
    for mt in gen.PCOMP_SAMPLES:
        code = methods.model_of(mt)[0].pcomp
        free = set(pc + 1 for pc in gen.free_immediates(code))
        skeletons.add(tuple(None if i in free else b for i, b in enumerate(code)))
    for mt in METHODS + ["x4,1,4,0,3,24", "x4,2,5,0,7,24", "x3,1,6,0,4,20", "x2,6,4,0,3,18", "x7,1,4,0,3,24"]:
        code = methods.model_of(mt)[0].pcomp
        if not code:
            continue
        free = set(pc + 1 for pc in gen.free_immediates(code))
        assert tuple(None if i in free else b for i, b in enumerate(code)) in skeletons, mt
