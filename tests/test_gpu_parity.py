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


# ---------------------------------------------------------------------------------------
import hashlib
import json
import os

from zpaqsharp_amd import models, synth, zpaql

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TAG = bytes([0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3])


def _manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", sorted(_manifest()))
@pytest.mark.parametrize("kernel", [0, 1])
def test_golden_fixtures(ctx, name, kernel):
    e = _manifest()[name]
    stream = open(os.path.join(GOLD, name + ".zpaq"), "rb").read()
    got = ctx.decompress(stream, verify_sha1=True, kernel=kernel).tobytes()
    assert len(got) == e["plain_len"] and hashlib.sha1(got).hexdigest() == e["plain_sha1"]


@pytest.mark.parametrize("kind", ["T", "R"])
def test_generic_and_lane_kernels_agree_with_oracle(ctx, kind):
    s, _ = synth.stream("l1", kind, nblocks=6, block_size=50000, threads=4)
    want = oracle.decompress(s.tobytes(), cap=300016)
    for kernel in (1, 0):
        got = ctx.decompress(s, verify_sha1=True, kernel=kernel)
        assert got.tobytes() == want
        assert ctx.stats().kernel_kind == (1 if kernel == 1 else 2)


def test_cm_models_with_other_shapes(ctx):
    # wider/narrower CM tables, an HCOMP that is not the recognised shift form, and a PCOMP on a CM model
    cfgs = [
        "comp 1 0 0 0 1 0 cm 9 3 hcomp a<<= 3 *d=a halt end",
        "comp 2 2 0 0 1 0 cm 20 255 hcomp *b=a a=0 hash b-- hash *d=a b++ b++ halt end",     # order-2 hash
        "comp 0 0 0 0 1 0 cm 22 40 hcomp a*= 77 a+= 5 *d=a halt end",
        "comp 0 0 0 0 1 0 cm 8 255 hcomp a<<= 9 *d=a halt end",                              # < 9 bits: generic kernel
    ]
    data = util.text(30000, seed=4)
    for cfg in cfgs:
        m = zpaql.assemble(cfg)
        s = synth.compress_block(m, data)
        assert oracle.decompress(s) == data
        assert ctx.decompress(s, verify_sha1=True).tobytes() == data
    m = zpaql.assemble("comp 0 0 0 0 1 0 cm 17 255 hcomp a<<= 9 *d=a halt " + models.E8E9_PCOMP.strip())
    x = util.x86ish(20000, seed=5)
    s = synth.compress_block(m, x)
    assert oracle.decompress(s) == x
    assert ctx.decompress(s, verify_sha1=True).tobytes() == x


@pytest.mark.parametrize("kernel", [0, 6])
def test_two_wave_cm_kernel_stress(ctx, kernel):
    """(kernel 6: the two-blocks-per-workgroup form, 32 LDS windows per block, that a launch of more than 256 blocks takes.)
    zh_cm.hip's decoder/helper split: shift forms that use the assembly loop (K >= 9) and the C++ body (K < 9),
    wide tables (window ids beyond one byte), data that thrashes the 36-window LDS cache (evictions, write-backs,
    reloads), runs of one byte (every byte waits for the helper wave's update of the same window), and several
    segments per block (the section is left and re-entered)."""
    rng = np.random.default_rng(77)
    n = 150000
    runs = np.repeat(rng.integers(0, 256, n // 50, dtype=np.uint8), 50).tobytes()
    rand = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    few = rng.integers(0, 40, n, dtype=np.uint8).tobytes()          # just over the cache size: steady misses
    text = util.text(n, seed=12)
    cfgs = ["comp 0 0 0 0 1 0 cm 17 255 hcomp a<<= 9 *d=a halt end",
            "comp 0 0 0 0 1 0 cm 16 255 hcomp a<<= 8 *d=a halt end",      # low context bits not zero: C++ body
            "comp 0 0 0 0 1 0 cm 22 20 hcomp a<<= 14 *d=a halt end",      # window id = byte << 5
            "comp 0 0 0 0 1 0 cm 12 255 hcomp a<<= 10 *d=a halt end"]     # table smaller than the cache
    for cfg in cfgs:
        m = zpaql.assemble(cfg)
        for data in (runs, rand, few, text):
            s = synth.compress_block(m, data)
            assert ctx.decompress(s, verify_sha1=True, kernel=kernel).tobytes() == data, cfg
    # the oracle agrees on one of each (keeps the CPU time of this test small)
    m = zpaql.assemble(cfgs[1])
    assert oracle.decompress(synth.compress_block(m, few[:40000])) == few[:40000]
    # multi-segment block through the Compressor mirror: the window cache survives the segment boundary
    m = models.get("l1")
    parts = [rand[:30000], b"", runs[:20000], text[:50000]]
    c = oracle.Compressor(400000)
    c.write_tag(); c.start_block(m.header)
    for i, part in enumerate(parts):
        c.start_segment(b"f%d" % i, str(len(part)).encode())
        if i == 0:
            c.post_process(m.pcomp)
        c.compress(part)
        c.end_segment(oracle.sha1(part))
    c.end_block()
    s = c.getvalue()
    assert ctx.decompress(s, verify_sha1=True, kernel=kernel).tobytes() == b"".join(parts)


def test_more_single_cm_blocks_than_cus_run_two_per_workgroup(ctx):
    """A launch with more than 256 single-CM blocks puts two blocks on every workgroup (zh_decode_cm_x2: four wavefronts,
    the read-only tables shared, 32 windows per block): 700 small blocks of text-like, x86-like and random plaintext —
    evictions in every block —, each against the generator's plaintext and the stored SHA-1; 512 blocks in flight."""
    nb = 700
    plains = [synth.plain("TXR"[b % 3], b, 2000 + (b * 37) % 3000).tobytes() for b in range(nb)]
    s = b"".join(synth.compress_block("l1", d) for d in plains)
    got = ctx.decompress(s, verify_sha1=True).tobytes()
    assert ctx.stats().concurrent == 512 and ctx.stats().kernel_kind == 2
    assert got == b"".join(plains)
    assert ctx.decompress(s, kernel=2).tobytes() == got and ctx.stats().concurrent == 256      # the one-block form agrees
    small = b"".join(synth.compress_block("l1", d) for d in plains[:3])                        # forced on a launch of three blocks
    assert ctx.decompress(small, verify_sha1=True, kernel=6).tobytes() == b"".join(plains[:3])


@pytest.mark.parametrize("model", ["l1+lz77", "mid+lz77", "min+lz77"])
def test_lz77_postprocessor_with_memory_in_hbm(ctx, model):
    """A PCOMP program whose memory does not fit on chip: the LZ77 post-processor of models.LZ77_PCOMP keeps a
    64 KiB history in M (arena slot in HBM) and runs on the ZPAQL interpreter of every kernel."""
    rng = np.random.default_rng(5)
    parts = [util.text(120000, seed=21), b"ab" * 40000 + util.x86ish(30000, 7), b"",
             bytes(rng.integers(0, 256, 5000, dtype=np.uint8)) * 3]
    m = models.get(model)
    s = b"".join(synth.compress_block(m, d) for d in parts)
    assert len(s) < sum(map(len, parts)) // 2                   # the matches really are used
    want = b"".join(parts)
    assert oracle.decompress(s[:len(synth.compress_block(m, parts[0]))]) == parts[0]
    for kernel in (0, 1):
        assert ctx.decompress(s, verify_sha1=True, kernel=kernel).tobytes() == want
    assert ctx.block_pcomp(s, 0)[2:] == m.pcomp


def test_many_small_segments_in_one_block(ctx):
    """Archives pack many small files into one block: every segment end leaves and re-enters the two-wave section of
    the single-CM kernel (and restarts the coder), the model carries over."""
    rng = np.random.default_rng(9)
    for model in ("l1", "min"):
        m = models.get(model)
        text = util.text(200000, seed=31)
        parts, pos = [], 0
        for i in range(120):
            n = int(rng.integers(0, 2500)) if i % 7 else 0
            parts.append(text[pos:pos + n]); pos += n
        c = oracle.Compressor(400000)
        c.write_tag(); c.start_block(m.header)
        for i, part in enumerate(parts):
            c.start_segment(b"f%d" % i, str(len(part)).encode())
            if i == 0:
                c.post_process(m.pcomp)
            c.compress(part)
            c.end_segment(oracle.sha1(part) if i % 3 else None)
        c.end_block()
        s = c.getvalue()
        out, res = ctx.decompress_segments(s, verify_sha1=True)
        assert out.tobytes() == b"".join(parts)
        assert [int(r.out_len) for r in res[:len(parts)]] == [len(p) for p in parts]
        assert all(r.status == 0 for r in res[:len(parts)])


@pytest.mark.parametrize("bits,shift", [(9, 9), (12, 31), (17, 16), (24, 13), (20, 9), (11, 12), (26, 18)])
def test_single_cm_shift_forms(ctx, bits, shift):
    """Every (table size, shift) the two-wave kernel takes — the window id is (byte << shift & mask) >> 9 — against the oracle."""
    m = zpaql.assemble(f"comp 0 0 0 0 1 0 cm {bits} 255 hcomp a<<= {shift} *d=a halt end")
    rng = np.random.default_rng(bits * 100 + shift)
    data = util.text(20000, seed=shift) + bytes(rng.integers(0, 256, 6000, dtype=np.uint8)) + b"z" * 3000
    s = synth.compress_block(m, data)
    assert oracle.decompress(s) == data
    assert ctx.decompress(s, verify_sha1=True).tobytes() == data
    assert ctx.stats().kernel_kind == 2


def test_two_contexts_in_two_threads():
    """A context is single-threaded, distinct contexts are independent (LICENSE:44-46; include/zpaqhip.h)."""
    import threading
    streams = [synth.stream(m, "T", nblocks=24, block_size=60000, first_block=100 * i, threads=2)[0] for i, m in enumerate(("l1", "mid"))]
    want = [oracle.decompress(s.tobytes(), cap=24 * 60000 + 16) for s in streams]
    got, errs = [None, None], []

    def work(i):
        try:
            c = z.Context(0)
            for _ in range(3):
                got[i] = c.decompress(streams[i], verify_sha1=True).tobytes()
            c.close()
        except BaseException as e:                       # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert got == want


def test_multi_segment_blocks(ctx):
    for model in ("l1", "mid", "max+e8e9"):
        m = models.get(model)
        parts = [util.x86ish(n, n) if "e8e9" in model else util.text(n, n) for n in (3000, 0, 1, 2000)]
        c = oracle.Compressor(60000)
        c.write_tag(); c.start_block(m.header)
        for i, d in enumerate(parts):
            c.start_segment(b"seg%d" % i, str(len(d)).encode())
            if i == 0:
                c.post_process(m.pcomp)
            c.compress(oracle.e8e9(d) if "e8e9" in model else d)
            c.end_segment(oracle.sha1(d) if i % 2 == 0 else None)
        c.end_block()
        s = c.getvalue()
        want = oracle.decompress(s)
        if "e8e9" not in model:
            assert want == b"".join(parts)
        assert ctx.decompress(s, verify_sha1="e8e9" not in model).tobytes() == want


def test_unmodelled_store_block(ctx):
    hdr = zpaql.assemble("comp 0 0 0 0 0 hcomp halt end").header
    payload = b"\0" + bytes(range(256)) * 3
    chunks = payload[:100], payload[100:]
    body = b"".join(len(c).to_bytes(4, "big") + c for c in chunks) + b"\0\0\0\0"
    s = TAG + b"zPQ" + bytes([2, 1]) + hdr + b"\x01name\0comment\0\0" + body + bytes([254, 255])
    want = oracle.decompress(s)
    assert want == payload[1:]
    assert ctx.decompress(s).tobytes() == want


def test_no_size_hint_in_comment(ctx):
    d1, d2 = util.text(7000, 1), util.text(100, 2)
    s = util.block("l1", d1, comment=b"no size here") + util.block("mid", d2, comment=b"jDC\x01")
    assert ctx.decompress(s, verify_sha1=True).tobytes() == d1 + d2
    # a wrong size hint must not corrupt the layout either
    s = util.block("l1", d1, comment=b"5") + util.block("mid", d2, comment=b"99999")
    assert ctx.decompress(s, verify_sha1=True).tobytes() == d1 + d2


@pytest.mark.parametrize("model", ["l1", "mid"])
def test_corrupt_streams_report_the_oracle_error(ctx, model):
    data = util.text(20000, seed=9)
    good = util.block(model, data)
    sc = z.scan(good)
    g = sc.segments[0]
    rng = np.random.default_rng(1)
    outcomes = set()
    for trial in range(24):
        s = bytearray(good)
        pos = int(g.data_off + rng.integers(4, g.data_len - 8))
        s[pos] ^= 1 << int(rng.integers(0, 8))
        if bytes(s[pos - 3:pos + 4]).count(0) >= 4:
            continue
        s = bytes(s)
        try:
            want = ("ok", oracle.decompress(s, cap=1 << 20))
        except oracle.OracleError as e:
            want = ("err", str(e))
        try:
            got = ("ok", ctx.decompress(s).tobytes())
        except z.ZpaqError as e:
            got = ("err", str(e))
        assert got == want, (trial, pos)
        outcomes.add(want[0] if want[0] == "ok" else want[1])
    assert len(outcomes) >= 1
    # SHA-1 verification catches silent corruption
    s = bytearray(good)
    s[-10] ^= 0x55                                       # inside the stored checksum
    with pytest.raises(z.ZpaqError) as e:
        ctx.decompress(bytes(s), verify_sha1=True)
    assert e.value.code == -21


def test_truncated_coded_data(ctx):
    good = util.block("mid", util.text(5000))
    g = z.scan(good).segments[0]
    cut = good[:g.data_off + g.data_len // 2] + b"\0\0\0\0" + bytes([254, 255])
    try:
        want = ("ok", oracle.decompress(cut, cap=1 << 20))
    except oracle.OracleError as e:
        want = ("err", str(e))
    try:
        got = ("ok", ctx.decompress(cut).tobytes())
    except z.ZpaqError as e:
        got = ("err", str(e))
    assert got == want


def test_zpaql_error_and_budget(ctx):
    data = util.text(100)
    bad = zpaql.assemble("comp 0 0 0 0 1 0 cm 9 255 hcomp error halt end")
    s = synth_block_with_header(bad.header, data)
    with pytest.raises(z.ZpaqError) as e:
        ctx.decompress(s)
    assert e.value.code == -4 and "ZPAQL execution error" in str(e.value)
    with pytest.raises(oracle.OracleError, match="ZPAQL execution error"):
        oracle.decompress(s)
    loop = zpaql.assemble("comp 0 0 0 0 1 0 cm 9 255 hcomp do forever halt end")
    s = synth_block_with_header(loop.header, data)
    with pytest.raises(z.ZpaqError) as e:
        ctx.decompress(s, zpaql_budget=100000)
    assert e.value.code == -26


def synth_block_with_header(header, data):
    """A block whose HCOMP misbehaves cannot be ENCODED; borrow the coded bytes of a well-formed twin."""
    twin = zpaql.assemble("comp 0 0 0 0 1 0 cm 9 255 hcomp halt end")
    assert len(twin.header) <= len(header)
    s = synth.compress_block(twin, data)
    i = s.index(twin.header)
    return s[:i] + header + s[i + len(twin.header):]


def test_block_table_api_subset_and_count_mode(ctx):
    import torch
    s, offs = synth.stream("l1", "T", nblocks=5, block_size=30000, threads=4)
    sc = z.scan(s)
    d_in = torch.from_numpy(s).cuda()
    d_out = torch.zeros(5 * 30000, dtype=torch.uint8, device="cuda")
    ids = [4, 1]
    rc, res = ctx.decode_blocks_device(d_in.data_ptr(), s.size, sc, d_out.data_ptr(), [0, 30000], [30000, 30000], ids=ids)
    got = d_out.cpu().numpy()
    assert np.array_equal(got[:30000], synth.plain("T", 4, 30000)) and np.array_equal(got[30000:60000], synth.plain("T", 1, 30000))
    assert res[4].status == 0 and res[4].out_len == 30000 and res[4].out_off == 0 and res[1].out_off == 30000
    assert res[0].status == 0 and res[0].out_len == 0                        # entries of other blocks are left untouched
    assert res[4].in_used == sc.segments[4].data_len
    # count-only mode: capacity 0 -> nothing written, true length reported
    d_out.zero_()
    rc, res = ctx.decode_blocks_device(d_in.data_ptr(), s.size, sc, d_out.data_ptr(), [0] * 5, [0] * 5, raise_on_error=False)
    assert all(r.status == -20 and r.out_len == 30000 for r in res) and int(d_out.sum()) == 0
    # partial capacity: prefix written, rest counted
    rc, res = ctx.decode_blocks_device(d_in.data_ptr(), s.size, sc, d_out.data_ptr(), [0], [1000], ids=[2], raise_on_error=False)
    assert res[2].status == -20 and res[2].out_len == 30000
    got = d_out.cpu().numpy()
    assert np.array_equal(got[:1000], synth.plain("T", 2, 30000)[:1000]) and int(got[1000:].sum()) == 0


def test_unaligned_output_offsets(ctx):
    import torch
    s, _ = synth.stream("l1", "T", nblocks=4, block_size=5003, threads=2)
    sc = z.scan(s)
    d_in = torch.from_numpy(s).cuda()
    d_out = torch.zeros(4 * 5003 + 64, dtype=torch.uint8, device="cuda")
    off = [1 + i * 5003 for i in range(4)]
    ctx.decode_blocks_device(d_in.data_ptr(), s.size, sc, d_out.data_ptr(), off, [5003] * 4)
    got = d_out.cpu().numpy()
    assert got[0] == 0 and int(got[1 + 4 * 5003:].sum()) == 0
    for i in range(4):
        assert np.array_equal(got[off[i]:off[i] + 5003], synth.plain("T", i, 5003))


def test_reader_writer_callback_form(ctx):
    data = util.text(70000, seed=11)
    s = util.block("l1", data[:40000]) + util.block("min", data[40000:])
    pos, out = [0], bytearray()

    def rd(n):
        chunk = s[pos[0]:pos[0] + min(n, 777)]
        pos[0] += len(chunk)
        return chunk

    ctx.decompress_cb(rd, out.extend, verify_sha1=True)
    assert bytes(out) == data


def test_full_size_l1_blocks_round_trip(ctx):
    """BASELINE configs[1] shape at reduced block count: 4 MiB blocks, in-stream SHA-1 + generator plaintext."""
    nb, bs = 48, 4 << 20
    s, _ = synth.stream("l1", "T", nblocks=nb, block_size=bs)
    got = ctx.decompress(s, verify_sha1=True)
    assert got.size == nb * bs
    for b in range(nb):
        assert np.array_equal(got[b * bs:(b + 1) * bs], synth.plain("T", b, bs)), b


def test_more_blocks_than_slots(ctx):
    s, _ = synth.stream("l1", "T", nblocks=40, block_size=3000, threads=4)
    want = np.concatenate([synth.plain("T", b, 3000) for b in range(40)])
    for kernel in (0, 1):
        got = ctx.decompress(s, verify_sha1=True, max_concurrent=7, kernel=kernel)
        assert np.array_equal(got, want)


def test_python_mirror_of_the_reference_api(ctx):
    from zpaqsharp_amd import decompresser as D
    a, b, c = util.text(5000, 1), util.text(0, 2), util.x86ish(3000, 3)
    s = util.block("l1", a, filename=b"a.txt") + util.block("mid", b) + util.block("max+e8e9", c, filename=b"c.bin")
    # LibZPAQ.decompress(Reader, Writer)
    w = D.BytesWriter()
    D.decompress(D.BytesReader(s), w, context=ctx)
    assert bytes(w.buf) == a + b + c == oracle.decompress(s)
    # documented step-wise order, decompress(n) resumption, sha1 string, hcomp(), pcomp()
    d, od = D.Decompresser(ctx), oracle.Decompresser(s)
    d.setInput(D.BytesReader(s))
    names, total = [], bytearray()
    while d.findBlock():
        assert d.memory() == od.find_block()
        h = D.BytesWriter(); d.hcomp(h)
        assert bytes(h.buf) == od.hcomp()
        while True:
            fn = D.BytesWriter()
            more = d.findFilename(fn)
            ofn = od.find_filename()
            assert more == (ofn is not None)
            if not more:
                break
            assert bytes(fn.buf) == ofn
            cm = D.BytesWriter(); d.readComment(cm)
            assert bytes(cm.buf) == od.read_comment()
            out = D.BytesWriter(); d.setOutput(out)
            sha = hashlib.sha1(); d.setSHA1(sha)
            calls = 0
            while d.decompress(1000):
                calls += 1
            want, _ = od.decompress()
            assert bytes(out.buf) == want and calls == len(want) // 1000
            pw = D.BytesWriter()                            # pcomp(): false / nothing for PASS blocks, else len + program
            assert d.pcomp(pw) == bool(od.pcomp()) and bytes(pw.buf) == od.pcomp()
            stored = d.readSegmentEnd()
            assert stored == od.read_segment_end() == sha.digest()
            names.append(bytes(fn.buf)); total += out.buf
    assert od.find_block() is None and names == [b"a.txt", b"", b"c.bin"] and bytes(total) == a + b + c


def test_python_mirror_raises_at_the_failing_segment(ctx):
    from zpaqsharp_amd import decompresser as D
    good1, bad, good2 = util.block("l1", util.text(3000, 1)), bytearray(util.block("l1", util.text(3000, 2))), util.block("l1", util.text(500, 3))
    g = z.scan(bytes(bad)).segments[0]
    bad[g.data_off + 50] ^= 0x10
    s = good1 + bytes(bad) + good2
    d = D.Decompresser(ctx)
    d.setInput(D.BytesReader(s))
    out = D.BytesWriter(); d.setOutput(out)
    assert d.findBlock() and d.findFilename()
    d.readComment(); assert d.decompress() is False; d.readSegmentEnd()
    assert bytes(out.buf) == util.text(3000, 1) and not d.findFilename()
    assert d.findBlock() and d.findFilename()
    d.readComment()
    with pytest.raises(z.ZpaqError):
        d.decompress()


_CPP_TEST = r'''
#include <stdio.h>
#include <string.h>
#include <string>
#include "Decompresser.hpp"
struct In : zpaq::Reader { FILE *f; int get() override { return getc(f); } };
struct Out : zpaq::Writer { std::string s; void put(int c) override { s.push_back((char)c); } };
int main(int argc, char **argv) {
  In in; in.f = fopen(argv[1], "rb");
  Out out;
  try { zpaq::decompress(&in, &out); }                       // LibZPAQ.decompress(Reader, Writer)
  catch (const zpaq::Error &e) { printf("ERR %d %s\n", e.code, e.what()); return 3; }
  fwrite(out.s.data(), 1, out.s.size(), fopen(argv[2], "wb"));
  // step-wise with decompress(n) and sha1string
  rewind(in.f);
  zpaq::Decompresser d;
  d.setInput(&in);
  Out o2; d.setOutput(&o2);
  int segs = 0, with_sha = 0;
  size_t pcomp_bytes = 0;
  while (d.findBlock()) {
    Out name;
    while (d.findFilename(&name)) {
      d.readComment();
      while (d.decompress(777)) {}
      char sha[21]; d.readSegmentEnd(sha);
      ++segs; with_sha += sha[0];
    }
    Out pc;
    if (d.pcomp(&pc)) pcomp_bytes += pc.s.size();
  }
  printf("OK %zu %d %d %d %zu\n", out.s.size(), o2.s == out.s, segs, with_sha, pcomp_bytes);
  return 0;
}
'''


def test_cpp_mirror_of_the_reference_api(ctx, tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a, b = util.text(40000, 1), util.x86ish(9000, 2)
    s = util.block("l1", a) + util.block("max+e8e9", b)
    (tmp_path / "in.zpaq").write_bytes(s)
    (tmp_path / "t.cpp").write_text(_CPP_TEST)
    exe = tmp_path / "t"
    lib = os.path.join(root, "zpaqsharp_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"), "-I", os.path.join(lib, "host"),
                           str(tmp_path / "t.cpp"), "-o", str(exe), "-L", lib, "-lzpaqhip", f"-Wl,-rpath,{lib}",
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe), str(tmp_path / "in.zpaq"), str(tmp_path / "out.bin")]).decode()
    assert out.split() == ["OK", str(len(a) + len(b)), "1", "2", "2", str(len(models.get("max+e8e9").pcomp) + 2)], out
    assert (tmp_path / "out.bin").read_bytes() == a + b


@pytest.mark.parametrize("model,kind", [("min", 3), ("mid", 3), ("max", 3), ("max+e8e9", 3), ("l1", 2)])
def test_kernel_selection_and_cross_kernel_agreement(ctx, model, kind):
    data = util.x86ish(40000, 3) if "e8e9" in model else util.text(40000, 3)
    s = synth.compress_block(model, data)
    auto = ctx.decompress(s, verify_sha1=True).tobytes()
    assert ctx.stats().kernel_kind == kind                      # the specialised kernel really ran
    generic = ctx.decompress(s, verify_sha1=True, kernel=1).tobytes()
    assert ctx.stats().kernel_kind == 1
    assert auto == generic == data == oracle.decompress(s)
    if model == "l1":                                           # a single CM also runs on the lane-per-component kernel
        chain = ctx.decompress(s, verify_sha1=True, kernel=3).tobytes()
        assert ctx.stats().kernel_kind == 3 and chain == data
    else:                                                       # ... and without the model-specialised level code
        plain = ctx.decompress(s, verify_sha1=True, kernel=4).tobytes()
        assert ctx.stats().kernel_kind == 3 and plain == data


def test_random_component_chains(ctx):
    """Random COMP lists exercising every component type and arbitrary input wiring on the
    lane-per-component kernel, against the oracle (encoder: libzpaqgen, decoder: oracle + GPU)."""
    rng = np.random.default_rng(int(os.environ.get("ZPAQ_FUZZ_SEED", "12345")))
    data = util.text(6000, seed=21) + util.x86ish(2000, seed=22)
    hcomp = "c++ *c=a b=c a=0 d= 0 hash *d=a b-- d++ hash *d=a b-- d++ hash *d=a d++ a=*c a<<= 8 *d=a d++ a=*c a*= 200 *d=a halt"
    for trial in range(int(os.environ.get("ZPAQ_FUZZ_TRIALS", "12"))):
        n = int(rng.integers(2, 12))
        comps = []
        for i in range(n):
            choices = ["cm", "icm", "match", "const"] if i == 0 else ["cm", "icm", "isse", "match", "avg", "mix2", "mix", "sse", "const"]
            t = str(rng.choice(choices))
            j, k = (int(rng.integers(0, i)), int(rng.integers(0, i))) if i else (0, 0)
            if t == "cm":
                comps.append(f"{i} cm {int(rng.integers(4, 18))} {int(rng.integers(1, 256))}")
            elif t == "icm":
                comps.append(f"{i} icm {int(rng.integers(0, 12))}")
            elif t == "isse":
                comps.append(f"{i} isse {int(rng.integers(0, 12))} {j}")
            elif t == "match":
                comps.append(f"{i} match {int(rng.integers(2, 16))} {int(rng.integers(8, 17))}")
            elif t == "avg":
                comps.append(f"{i} avg {j} {k} {int(rng.integers(0, 256))}")
            elif t == "mix2":
                comps.append(f"{i} mix2 {int(rng.integers(0, 10))} {j} {k} {int(rng.integers(1, 64))} {int(rng.choice([0, 255, 15]))}")
            elif t == "mix":
                m = int(rng.integers(1, i - j + 1))
                comps.append(f"{i} mix {int(rng.integers(0, 10))} {j} {m} {int(rng.integers(1, 64))} {int(rng.choice([0, 255, 240]))}")
            elif t == "sse":
                lim = int(rng.integers(1, 256))
                comps.append(f"{i} sse {int(rng.integers(0, 10))} {j} {int(rng.integers(0, min(255, lim * 4) + 1))} {lim}")
            else:
                comps.append(f"{i} const {int(rng.integers(0, 256))}")
        if sum(c.split()[1] == "mix" for c in comps) > 4:
            continue
        cfg = f"comp 3 3 0 0 {n}\n" + "\n".join(comps) + f"\nhcomp {hcomp}\nend"
        m = zpaql.assemble(cfg)
        s = synth.compress_block(m, data)
        assert oracle.decompress(s) == data, cfg
        got = ctx.decompress(s, verify_sha1=True).tobytes()
        assert got == data, cfg
        assert ctx.stats().kernel_kind == 3, cfg
