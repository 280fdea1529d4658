"""ZPAQL assembler/disassembler (host tooling; Compiler.cs:319-478, :535-569)."""
import pytest

import oracle
from zpaqsharp_amd import models, zpaql


def test_opcode_table_shape():
    named = [i for i, m in enumerate(zpaql.OPCODES) if m is not None and i != 0]
    assert all(not zpaql.is_error_op(i) for i in named)
    assert all(zpaql.is_error_op(i) for i in range(256) if zpaql.OPCODES[i] is None)
    assert zpaql.OPCODES[59] == "hash" and zpaql.OPCODES[60] == "hashd" and zpaql.OPCODES[57] == "out"
    assert zpaql.OPCODES[207] == "a<<=" and zpaql.OPCODES[112] == "*d=a" and zpaql.OPCODES[255] == "lj"


@pytest.mark.parametrize("name", ["l1", "min", "mid", "max", "max+e8e9"])
def test_assemble_disassemble_round_trip(name):
    m = models.get(name)
    again = zpaql.assemble(zpaql.disassemble(m.header, m.pcomp))
    assert again.header == m.header and again.pcomp == m.pcomp


def test_control_macros_execute_correctly():
    # count down from the input byte, emitting each value: do/while + if/else
    prog = zpaql.assemble("""
        comp 0 0 0 0 0 hcomp halt
        pcomp x ;
          a> 255 if halt endif
          b=a
          do a=b a> 0 if a=b out b-- a=b a> 0 while endif
          a= 7 out
          halt
        end""").pcomp
    assert oracle.run_pcomp(prog, bytes([3, 0, 2])) == bytes([3, 2, 1, 7, 7, 2, 1, 7])


def test_long_jumps():
    body = " ".join(["a++"] * 200)
    prog = zpaql.assemble(f"comp 0 0 0 0 0 hcomp halt pcomp x ; a> 255 ifnotl {body} out endif halt end").pcomp
    assert oracle.run_pcomp(prog, bytes([1])) == bytes([201])


def test_lz77_model_round_trips_on_the_oracle():
    import numpy as np
    import oracle
    from tests import util
    from zpaqsharp_amd import models, synth
    m = models.get("l1+lz77")
    assert zpaql.parse_header(m.header)[3] == 16 and m.pcomp_cmd.startswith("lz77")
    rng = np.random.default_rng(2)
    for d in (util.text(30000, 4), b"a" * 4000, b"", b"xy", bytes(rng.integers(0, 256, 2000, dtype=np.uint8)), util.x86ish(9000, 6)):
        enc = synth.lz77_encode(d)
        s = synth.compress_block(m, d)
        assert oracle.decompress(s) == d
        assert len(enc) <= len(d) + len(d) // 8 + 2          # never much worse than stored
