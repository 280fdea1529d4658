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
    # (the short periodic and mirrored inputs: the BWT tool's first doubling step once merged distinct byte pairs for blocks
    # under 254 bytes — found by tools/fuzz_methods.py as streams that every decoder, the oracle included, agreed to be something else)
    per = b"rdk tyzkalbcr"
    for data in (b"", b"a", b"abcd" * 2, bytes(300), bytes(range(256)) * 2, per * 3, (per + per[::-1] + per)[:39], b"w\xce\xc6\xe9\xd2" * 8, bytes(39)):
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


# The models makeConfig writes for levels 3 (BWT) and 4 have min's / mid's component lists with other sizes and another
# HCOMP: zh_nibble.hip decodes them (round 5), their post-processor fed from the chunks its assembly loop parks.
NIBBLE_METHODS = ["x0,0ci1,1,1,1,2am", "x0,4ci1,1,1,1,2am", "x4,0ci1,1,1,1,2awm", "x0,4ci1,1,1,1,2awm", "x0,3ci1", "x0,7ci1", "x4,3ci1",
                  "x0,2,12,0,7,21,1c0,0,511i2", "x4,6,12,0,7,25,1c0,0,511i2", "x0,2,5,0,7,21,1c0,0,511", "x4,6,5,0,7,25,1c0,0,511"]


def test_method_models_join_the_families_of_min_and_mid():
    """zh_framing.cpp files the level-4 model (`ci1,1,1,1,2am`) under mid's kernel family, its text variant (`...2awm`: a
    word-model ICM as eighth mixer input) under a family of its own, the BWT model (`ci1`) under min's — whatever the
    block-size argument makes of the table sizes; a chain of another length stays with the lane-per-component kernel.
    Seen from outside through zpaqhip_block_costs: plaintext bytes x the family's cycles per byte (+ 1500 with PCOMP memory)."""
    data = _data(5000)
    per_byte = {"x0,0ci1,1,1,1,2am": 6800, "x6,4ci1,1,1,1,2am": 6800, "x0,0ci1,1,1,1,2awm": 7000, "x4,4ci1,1,1,1,2awm": 7000,
                "x0,3ci1": 3800 + 1500, "x4,7ci1": 3800 + 1500, "x0,2,12,0,7,21,1c0,0,511i2": 3800 + 1500, "x4,6,12,0,7,25,1c0,0,511i2": 3800 + 1500,
                "x0,2,5,0,7,21,1c0,0,511": 3800 + 1500, "x4,6,5,0,7,25,1c0,0,511": 3800 + 1500, "x0,2,5,0,7,21,1c0,0,255": 4000 + 2200 * 1 + 1500, "x0,0ci1,1,1,2am": 4000 + 2200 * 7, "x0,0ci1,1,1,1,2a": 4000 + 2200 * 7}
    for method, w in per_byte.items():
        s = methods.compress_block(method, data)
        sc = z.scan(s)
        assert list(z.block_costs(s, sc)) == [len(data) * w], method
    for name, w in (("mid", 6800), ("min", 3800)):
        s = util.block(name, data)
        assert list(z.block_costs(s, z.scan(s))) == [len(data) * w], name


@pytest.mark.gpu
def test_method_models_on_the_nibble_kernels(ctx):
    """Every model of NIBBLE_METHODS on inputs that reach the assembly loop, its 256-byte chunks handed to the post-processor
    (E8E9: translated; bwtrle: structurally matched; the rest of a chunk when the loop is left), short and empty blocks that
    never reach it, runs and incompressible data — against the plaintext, the oracle and the lane-per-component kernel
    (opts.kernel 4), which interprets the same HCOMP.  Then blocks of all of them and of the built-in min / mid in ONE stream:
    the launches of mid's and min's families then hold blocks whose helper wave runs different programs."""
    rng = np.random.default_rng(77)
    cases = {"mixed": _data(150000), "text": util.text(70001, seed=8), "x86": util.x86ish(60000, 2), "tiny": b"Az", "empty": b"",
             "runs": np.repeat(rng.integers(0, 256, 900, dtype=np.uint8), 41).tobytes(),
             "random": rng.integers(0, 256, 30000, dtype=np.uint8).tobytes(), "words": b"the quick brown fox " * 2500}
    for method in NIBBLE_METHODS:
        for name, data in cases.items():
            s = methods.compress_block(method, data)
            if len(data) <= 70001:
                assert oracle.decompress(s, cap=len(data) + 16) == data, (method, name)
            for kernel in (0, 4):
                assert ctx.decompress(s, verify_sha1=True, kernel=kernel, out_cap=len(data) + 16).tobytes() == data, (method, name, kernel)
    parts = []
    for i in range(3):
        for method in NIBBLE_METHODS:
            parts.append((methods.compress_block(method, _data(3000 + 977 * i + 13 * len(method))), _data(3000 + 977 * i + 13 * len(method))))
        for name in ("mid", "min"):
            d = util.text(5000 + 333 * i, seed=40 + i)
            parts.append((util.block(name, d), d))
    s = b"".join(p for p, _ in parts)
    assert ctx.decompress(s, verify_sha1=True).tobytes() == b"".join(d for _, d in parts)
    # a damaged stream: whatever the oracle makes of it (garbage into the post-processor, or an error)
    good = methods.compress_block("x0,4ci1,1,1,1,2awm", cases["text"])
    g = z.scan(good).segments[0]
    for trial in range(6):
        dmg = bytearray(good)
        pos = int(g.data_off + rng.integers(40, g.data_len - 8))
        dmg[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            want = ("ok", oracle.decompress(bytes(dmg), cap=1 << 20))
        except oracle.OracleError as e:
            want = ("err", str(e))
        try:
            got = ("ok", ctx.decompress(bytes(dmg)).tobytes())
        except z.ZpaqError as e:
            got = ("err", str(e))
        assert got == want, (trial, pos)


@pytest.mark.gpu
def test_method_streams_multi_block(ctx):
    parts = [(mt, _data(4000 + 700 * i)) for i, mt in enumerate(METHODS[:7])]
    s = b"".join(methods.compress_block(mt, d) for mt, d in parts)
    assert ctx.decompress(s, verify_sha1=True).tobytes() == b"".join(d for _, d in parts)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["x0,1,4,0,3,16", "x6,1,4,0,3,24", "x0,5,4,0,3,16", "x6,5,4,0,3,24", "x0,2,12,0,7,16",
                                    "x0,6,5,0,3,16c0,0,511", "x0,4ci1,1,1,1,2am", "x0,3ci1", "x0,7ci1",
                                    "x0,2,12,0,7,21,1c0,0,511i2", "x0,2,5,0,7,21,1c0,0,511"])
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


# ---- the wave-wide store / LZ77 kernel (zh_store.hip) ------------------------------------------------------------------
def test_cpp_preprocessors_write_the_reference_formats():
    """zpaqgen's fast LZ77 / BWT pre-processors (benchmark streams) against the oracle's run of the reference's
    post-processor programs: the formats of LZBuffer.cs:96-115."""
    from zpaqsharp_amd import synth
    rng = np.random.default_rng(11)
    datas = [util.text(70000, seed=3), bytes(rng.integers(0, 256, 5000, dtype=np.uint8)), b"ab" * 40000, b"", b"x"]
    for mt in ("x0,1,4,0,3,16", "x6,1,4,0,3,24", "x0,2,12,0,7,16", "x0,2,3,0,7,16", "x0,3"):
        model, args = methods.model_of(mt)
        for d in datas:
            pre = synth.preprocess(args, d)
            assert oracle.run_pcomp(model.pcomp, pre, model.header[4], model.header[5], cap=len(d) + 64) == d, (mt, len(d))


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["x2,1,4,0,3,22", "x6,1,4,0,3,24", "x2,2,12,0,7,22", "x2,2,3,0,7,22", "x2,3"])
def test_store_kernel_on_big_distinct_blocks(ctx, method):
    """Blocks of several stored chunks (a chunk header in the middle of a bit-packed code), output past the LDS ring,
    matches from further back than the ring (a block that repeats itself after 1.5 MiB), every block distinct."""
    from zpaqsharp_amd import synth
    model, args = methods.model_of(method)
    s, offs = synth.method_stream(model, args, "T", 6, 700_000, threads=4)
    want = b"".join(synth.plain("T", b, 700_000).tobytes() for b in range(6))
    assert ctx.decompress(s, out_cap=len(want), verify_sha1=True).tobytes() == want
    assert ctx.stats().launches == 1                                   # (nothing was handed back to the generic kernel)
    base = util.text(1_500_000, seed=77)
    far = base + base[:900_000] + bytes(reversed(base[:300_000])) + base[200_000:1_400_000]     # distances of 1.5 and 2.7 MiB
    pre = synth.preprocess(args, far)
    blk = methods.compress_block(method, far, pre=pre)
    assert oracle.decompress(blk, cap=len(far) + 16) == far
    assert ctx.decompress(blk, verify_sha1=True).tobytes() == far
    assert ctx.stats().launches == 1


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["x0,1,4,0,3,16", "x0,2,12,0,7,16"])
def test_store_kernel_when_the_window_wraps(ctx, method):
    """A block longer than the program's M array (2^20 bytes here; no encoder of the reference writes one): distances
    are taken modulo the array like the program does — the result is NOT the plaintext, it is what the oracle gives."""
    from zpaqsharp_amd import synth
    model, args = methods.model_of(method)
    base = util.text(1_500_000, seed=78)
    data = base + base[:900_000] + base[300_000:1_200_000]
    blk = methods.compress_block(method, data, pre=synth.preprocess(args, data))
    want = oracle.decompress(blk, cap=len(data) + 16)
    assert len(want) == len(data) and want != data
    assert ctx.decompress(blk).tobytes() == want
    assert ctx.stats().launches == 1


@pytest.mark.gpu
def test_store_kernel_hands_back_what_it_does_not_take(ctx):
    """Programs that are not the reference's LZ77 ones (E8E9 variants, BWT, an operand changed), damaged chunks: the
    block runs on the generic kernel in the same call (zpaqhip_stats.launches counts the second launch)."""
    d = util.text(30000, seed=5)
    for method, launches in (("x0,5,4,0,3,16", 2), ("x0,7", 2), ("x0,3", 1), ("x5,3", 1)):      # E8E9 variants go back; the plain BWTs are taken
        s = methods.compress_block(method, d)
        assert ctx.decompress(s, verify_sha1=True).tobytes() == d
        assert ctx.stats().launches == launches, method
    # lzpre with one operand changed (a> 254 instead of a> 255 at the top): same structure, another program
    model, args = methods.model_of("x0,2,12,0,7,16")
    pc = bytearray(model.pcomp)
    assert pc[1] == 255
    pc[1] = 254
    pre = methods.preprocess(d, args)
    dec = bytes([1, len(pc) & 255, len(pc) >> 8]) + bytes(pc) + pre
    body = len(dec).to_bytes(4, "big") + dec + b"\0\0\0\0"
    tag = bytes([0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3])
    s = tag + b"zPQ" + bytes([2, 1]) + model.header + b"\x01\0" + str(len(d)).encode() + b"\0\0" + body + b"\xfe\xff"
    want = oracle.decompress(s, cap=len(d) + 16)
    assert ctx.decompress(s).tobytes() == want
    assert ctx.stats().launches == 2


@pytest.mark.gpu
def test_a_bwt_block_over_16_mib(ctx):
    """Blocks over 16 MiB take the other list traversal of bwtrle (LibZPAQ.cs:741-795: plain positions in H instead of
    position << 8 | byte): one real block of 17 MiB, written with the method's own configuration, decoded by the wave-wide
    inverse BWT and checked against the plaintext and the stored SHA-1."""
    from zpaqsharp_amd import synth
    method = "x5,3"
    model, args = methods.model_of(method)
    n = 17 << 20
    s, offs = synth.method_stream(model, args, "T", 1, n, threads=1)
    got = ctx.decompress(s, out_cap=n, verify_sha1=True)
    assert ctx.stats().launches == 1
    assert np.array_equal(got, synth.plain("T", 0, n))


def test_scalar_e8e9_is_the_reference_program(tmp_path):
    """zh_e8e9.h states the reference's E8E9 post-processor (LibZPAQ.cs:802-826) as the few scalar operations it amounts to per
    byte (B, C are all its state); zh_nibble.hip's drain runs that instead of the translated program.  Here the header is compiled
    for the host and fed what the oracle's interpreter is fed: random bytes, bytes dense in E8 / E9 opcodes with 00 / FF top
    bytes (every branch of the program), short inputs (fewer bytes than the program holds back) — same output byte for byte,
    including what the program's own end-of-segment run flushes from the (B, C) the scalar form leaves."""
    import ctypes
    import os
    import shutil
    import subprocess
    from zpaqsharp_amd import models
    if not shutil.which("g++"):
        pytest.skip("needs g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "e8.cpp"
    src.write_text('#include <stdint.h>\n#define __device__\n#define __forceinline__ inline\n'
                   f'#include "{root}/zpaqsharp_amd/csrc/zh_e8e9.h"\n'
                   'extern "C" uint32_t e8_run(const uint8_t *in, uint32_t n, uint8_t *out, uint32_t *B, uint32_t *C) {\n'
                   '  uint32_t k = 0, ob;\n'
                   '  for (uint32_t i = 0; i < n; ++i) if (zh_e8e9_step(*B, *C, in[i], ob)) out[k++] = (uint8_t)ob;\n'
                   '  return k;\n}\n')
    so = tmp_path / "e8.so"
    subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    lib = ctypes.CDLL(str(so))
    lib.e8_run.restype = ctypes.c_uint32
    pcomp = models.E8E9 if hasattr(models, "E8E9") else models.get("max+e8e9").pcomp
    rng = np.random.default_rng(88)
    cases = [b"", b"\xe8", b"\xe8\x01\x02\x03", b"\xe8\x01\x02\x03\x00", rng.integers(0, 256, 5000, dtype=np.uint8).tobytes(),
             util.x86ish(20000, 3)]
    dense = bytearray(rng.integers(0, 256, 6000, dtype=np.uint8))
    for i in range(0, len(dense) - 5, 7):
        dense[i] = 0xE8 + int(rng.integers(0, 2))
        dense[i + 4] = int(rng.choice([0, 255, 1, 254]))
    cases.append(bytes(dense))
    for data in cases:
        want = oracle.run_pcomp(pcomp, data, 0, 0, cap=len(data) + 64)           # whole segment: every byte, then the end-of-segment run
        out = (ctypes.c_uint8 * (len(data) + 8))()
        B, Cc = ctypes.c_uint32(0), ctypes.c_uint32(0)
        k = lib.e8_run(data, len(data), out, ctypes.byref(B), ctypes.byref(Cc))
        got = bytes(out[:k])
        # the program's end-of-segment run (`a> 255` branch) from (B, C): C > 4 -> 4 bytes pending, else C bytes after dropping 5 - C
        b, c = B.value, Cc.value
        if c > 4:
            c = 4
        else:                                                                     # a! a+= 5 a<<= 3 d=a a=b a>>=d b=a
            b >>= ((((c ^ 0xFFFFFFFF) + 5) & 0xFFFFFFFF) << 3) & 31
        tail = bytes((b >> (8 * i)) & 255 for i in range(c))
        assert got + tail == want, (len(data), got[-8:], tail, want[-12:])
