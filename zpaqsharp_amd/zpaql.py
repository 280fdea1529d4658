"""ZPAQL assembler / disassembler and block-header builder (host tooling).

The reference's `Compiler` (Compiler.cs:319-478, opcode table :535-569,
component names :516-517) turns ZPAQL *source text* into the COMP/HCOMP/PCOMP
byte strings that go into a block header.  It is compress-time only and out of
the GPU hot path; this module is the small,
independent equivalent used to author models, fixtures and synthetic streams.

Config grammar accepted (a subset of the reference's, same token names):

    comp HH HM PH PM N
      <i> <component> <args...>          (N lines, index is checked)
    hcomp
      <instructions> halt
    [pcomp <cmd...> ;
      <instructions> halt]
    end

Control macros: if ifnot else endif do while until forever (short jumps) and
ifl ifnotl elsel (long jumps), encoded as Compiler.cs:331-445 does.
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import List, Optional, Tuple

# Component.cs:27-43 / LibZPAQ.cs:51-63
COMP_NAMES = ["", "const", "cm", "icm", "match", "avg", "mix2", "mix", "isse", "sse"]
COMP_SIZE = [0, 2, 3, 2, 3, 4, 6, 6, 3, 5]

_DST = ["a", "b", "c", "d", "*b", "*c", "*d"]
_UNARY = ["<>a", "++", "--", "!", "=0"]
_ALU = ["+=", "-=", "*=", "/=", "%=", "&=", "&~", "|=", "^=", "<<=", ">>=", "==", "<", ">"]


def _build_opcodes() -> List[Optional[str]]:
    """Opcode → mnemonic, generated from the ISA description (ZPAQL.cs:238-321)."""
    t: List[Optional[str]] = [None] * 256
    t[0] = "error"
    for d, dn in enumerate(_DST):
        for x, xn in enumerate(_UNARY):
            if d == 0 and x == 0:
                continue
            t[d * 8 + x] = dn + xn
    for d, dn in enumerate(_DST[:4]):
        t[d * 8 + 7] = dn + "=r"
    t[39], t[47], t[55], t[63] = "jt", "jf", "r=a", "jmp"
    t[56], t[57], t[59], t[60] = "halt", "out", "hash", "hashd"
    for d, dn in enumerate(_DST):
        for s, sn in enumerate(_DST + [""]):
            t[64 + d * 8 + s] = dn + "=" + sn
    for x, xn in enumerate(_ALU):
        for s, sn in enumerate(_DST + [""]):
            t[128 + x * 8 + s] = "a" + xn + sn
    t[255] = "lj"
    return t


OPCODES = _build_opcodes()
_MNEMONIC = {m: i for i, m in enumerate(OPCODES) if m is not None}
JT, JF, JMP, LJ = 39, 47, 63, 255


def is_error_op(op: int) -> bool:
    """ZPAQL.cs:1338-1342 iserr()."""
    return (op == 0 or 120 <= op <= 127 or 240 <= op <= 254 or op == 58
            or (op < 64 and op % 8 in (5, 6)))


@dataclass
class Model:
    """A compiled block model: `header` is exactly what goes into the stream after
    `zPQ level 1` (hsize[2] hh hm ph pm n COMP 0 HCOMP 0); `pcomp` is the
    post-processor bytecode incl. its trailing 0 (empty = PASS)."""
    header: bytes
    pcomp: bytes = b""
    pcomp_cmd: str = ""

    @property
    def n(self) -> int:
        return self.header[6]


def _tokens(src: str) -> List[str]:
    # comments are (...) possibly spanning tokens, as in the reference's Compiler.
    out, depth = [], 0
    for tok in re.findall(r"\(|\)|[^\s()]+", src):
        if tok == "(":
            depth += 1
        elif tok == ")":
            depth -= 1
        elif depth == 0:
            out.append(tok.lower())
    return out


def _assemble_code(toks: List[str], pos: int, terminators: Tuple[str, ...]) -> Tuple[bytes, int, str]:
    """Assemble instructions until one of `terminators`.  Returns (code+END, new_pos, terminator)."""
    code = bytearray()
    if_stack: List[int] = []
    do_stack: List[int] = []

    def need_num(lo: int, hi: int) -> int:
        nonlocal pos
        v = int(toks[pos], 0)
        pos += 1
        if not lo <= v <= hi:
            raise ValueError(f"operand {v} out of range {lo}..{hi}")
        return v

    while True:
        if pos >= len(toks):
            raise ValueError("unexpected end of ZPAQL source")
        t = toks[pos]
        pos += 1
        if t in terminators:
            break
        if t in ("if", "ifnot"):
            code += bytes([JF if t == "if" else JT, 0])
            if_stack.append(len(code) - 1)
        elif t in ("ifl", "ifnotl"):
            code += bytes([JT if t == "ifl" else JF, 3, LJ, 0, 0])
            if_stack.append(len(code) - 2)
        elif t in ("else", "elsel"):
            a = if_stack.pop()
            long_else = t == "elsel"
            if code[a - 1] != LJ:
                j = len(code) - a + 1 + (1 if long_else else 0)
                if j > 127:
                    raise ValueError("IF too big, try IFL, IFNOTL")
                code[a] = j
            else:
                j = len(code) + 2 + (1 if long_else else 0)
                code[a], code[a + 1] = j & 255, j >> 8
            code += bytes([LJ, 0, 0] if long_else else [JMP, 0])
            if_stack.append(len(code) - (2 if long_else else 1))
        elif t == "endif":
            a = if_stack.pop()
            if code[a - 1] != LJ:
                j = len(code) - a - 1
                if j > 127:
                    raise ValueError("IF too big, try IFL, IFNOTL, ELSEL")
                code[a] = j
            else:
                j = len(code)
                code[a], code[a + 1] = j & 255, j >> 8
        elif t == "do":
            do_stack.append(len(code))
        elif t in ("while", "until", "forever"):
            a = do_stack.pop()
            j = a - len(code) - 2
            if j >= -127:
                code += bytes([{"while": JT, "until": JF, "forever": JMP}[t], j & 255])
            else:
                if t == "while":
                    code += bytes([JF, 3])
                if t == "until":
                    code += bytes([JT, 3])
                code += bytes([LJ, a & 255, a >> 8])
        else:
            if t not in _MNEMONIC:
                raise ValueError(f"unknown ZPAQL token {t!r}")
            op = _MNEMONIC[t]
            code.append(op)
            if op == LJ:
                v = need_num(0, 65535)
                code += bytes([v & 255, v >> 8])
            elif op in (JT, JF, JMP):
                code.append(need_num(-128, 127) & 255)
            elif op & 7 == 7:
                code.append(need_num(0, 255))
    if if_stack or do_stack:
        raise ValueError("unmatched IF or DO")
    code.append(0)
    return bytes(code), pos, t


def assemble(src: str) -> Model:
    """Compile a config text into a Model (Compiler.cs:13-111 equivalent)."""
    toks = _tokens(src)
    pos = 0
    if toks[pos] != "comp":
        raise ValueError("expected 'comp'")
    hh, hm, ph, pm, n = (int(x) for x in toks[pos + 1:pos + 6])
    pos += 6
    comp = bytearray()
    for i in range(n):
        if int(toks[pos]) != i:
            raise ValueError(f"expected component index {i}")
        name = toks[pos + 1]
        typ = COMP_NAMES.index(name)
        nargs = COMP_SIZE[typ] - 1
        args = [int(x) for x in toks[pos + 2:pos + 2 + nargs]]
        if any(not 0 <= a <= 255 for a in args):
            raise ValueError("component argument out of range")
        comp += bytes([typ] + args)
        pos += 2 + nargs
    if toks[pos] != "hcomp":
        raise ValueError("expected 'hcomp'")
    hcomp, pos, term = _assemble_code(toks, pos + 1, ("pcomp", "post", "end"))
    pcomp, cmd = b"", ""
    if term == "pcomp":
        j = toks.index(";", pos)
        cmd = " ".join(toks[pos:j])
        pcomp, pos, term = _assemble_code(toks, j + 1, ("end",))
    elif term == "post":
        # "post 0 end": no post-processing
        if toks[pos] != "0" or toks[pos + 1] != "end":
            raise ValueError("expected 'post 0 end'")
    body = bytes([hh, hm, ph, pm, n]) + bytes(comp) + b"\0" + hcomp
    hsize = len(body)
    if hsize > 65535:
        raise ValueError("program too big")
    return Model(bytes([hsize & 255, hsize >> 8]) + body, pcomp, cmd)


def parse_header(header: bytes):
    """Split a stream-form header into (hh, hm, ph, pm, comps, hcomp_bytes).
    comps is a list of (type, args...) tuples.  Mirrors ZPAQL.cs:112-156."""
    hsize = header[0] + 256 * header[1]
    if len(header) != hsize + 2:
        raise ValueError("header length does not match hsize")
    hh, hm, ph, pm, n = header[2:7]
    p, comps = 7, []
    for _ in range(n):
        typ = header[p]
        if typ >= len(COMP_SIZE) or COMP_SIZE[typ] < 1:
            raise ValueError("Invalid component type")
        comps.append(tuple(header[p:p + COMP_SIZE[typ]]))
        p += COMP_SIZE[typ]
    if header[p] != 0:
        raise ValueError("missing COMP END")
    hcomp = header[p + 1:]
    if not hcomp or hcomp[-1] != 0:
        raise ValueError("missing HCOMP END")
    return hh, hm, ph, pm, comps, bytes(hcomp)


def disassemble_code(code: bytes) -> List[str]:
    out, pc = [], 0
    while pc < len(code):
        op = code[pc]
        name = OPCODES[op] or f"<bad {op}>"
        if op == LJ:
            out.append(f"lj {code[pc + 1] + 256 * code[pc + 2]}")
            pc += 3
        elif op & 7 == 7:
            v = code[pc + 1]
            if op in (JT, JF, JMP):
                v = ((v + 128) & 255) - 128
            out.append(f"{name} {v}")
            pc += 2
        else:
            out.append(name)
            pc += 1
    return out


def disassemble(header: bytes, pcomp: bytes = b"") -> str:
    hh, hm, ph, pm, comps, hcomp = parse_header(header)
    lines = [f"comp {hh} {hm} {ph} {pm} {len(comps)}"]
    for i, c in enumerate(comps):
        lines.append(f"  {i} {COMP_NAMES[c[0]]} " + " ".join(str(x) for x in c[1:]))
    lines.append("hcomp")
    lines.append("  " + " ".join(disassemble_code(hcomp[:-1])))
    if pcomp:
        lines.append("pcomp ;")
        lines.append("  " + " ".join(disassemble_code(pcomp[:-1])))
    lines.append("end")
    return "\n".join(lines)
