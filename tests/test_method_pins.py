"""Pins tools/methods.py (the reference's method configs and post-processor programs, host tooling for fixtures)
to the reference TEXT: LibZPAQ.makeConfig, LibZPAQ.cs:388-1044.

The post-processor part of makeConfig (LibZPAQ.cs:423-830: `pcomp = "..." ...; if (doe8) pcomp += ...`) is loop-free
string building; a small evaluator for exactly that statement subset runs the reference's own text for every
(level, doe8, rb) variant and the result must equal what tools/methods.make_config emits, token for token after ZPAQL
comments `( ... )` and white space are dropped (the assembler drops them too).  The context-model part
(LibZPAQ.cs:835-1041: loops over std::vector, pointer walks) is checked by literals: every string literal of that
region appears, in source order, in the config of a method that exercises its branch.

These tests read /root/reference as text and are skipped where it is absent (marker `reference`)."""
import os
import re

import pytest

from tests.conftest import REFERENCE
from tools import methods

pytestmark = pytest.mark.reference


def _src():
    with open(os.path.join(REFERENCE, "LibZPAQ.cs"), encoding="utf-8", errors="replace") as f:
        return f.read().split("\n")


# ---- tokens of the C-like source ---------------------------------------------------------------------------------
_TOK = re.compile(r'\s+|//[^\n]*|("(?:[^"\\]|\\.)*")|([A-Za-z_][A-Za-z_0-9]*)|(\d+)|(\+=|==|!=|<<|>>|<=|>=|&&|\|\||[-+*/%&|!<>=?:;,(){}\[\]])')


def _tokens(text):
    out, pos = [], 0
    while pos < len(text):
        m = _TOK.match(text, pos)
        assert m, text[pos:pos + 40]
        pos = m.end()
        if m.group(1) is not None:
            out.append(("str", bytes(m.group(1)[1:-1], "utf-8").decode("unicode_escape")))
        elif m.group(2) is not None:
            out.append(("id", m.group(2)))
        elif m.group(3) is not None:
            out.append(("num", int(m.group(3))))
        elif m.group(4) is not None:
            out.append(("op", m.group(4)))
    return out


class _Eval:
    """if / else, blocks, `x = e;`, `x += e;`, declarations; expressions over ints, bools and strings."""

    def __init__(self, toks, env):
        self.t, self.i, self.env, self.live = toks, 0, env, True

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", None)

    def take(self, kind=None, val=None):
        tk = self.peek()
        assert (kind is None or tk[0] == kind) and (val is None or tk[1] == val), (tk, kind, val, self.t[max(0, self.i - 5):self.i + 5])
        self.i += 1
        return tk

    def at(self, kind, val=None):
        tk = self.peek()
        return tk[0] == kind and (val is None or tk[1] == val)

    # statements -------------------------------------------------------------------------------------------------
    def run(self):
        while not self.at("eof"):
            self.stmt(True)

    def stmt(self, live):
        self.live = live                                   # (names of a branch not taken may be undefined)
        if self.at("op", "{"):
            self.take()
            while not self.at("op", "}"):
                self.stmt(live)
            self.take()
        elif self.at("id", "if"):
            self.take()
            self.take("op", "(")
            c = self.expr()
            self.take("op", ")")
            self.stmt(live and bool(c))
            if self.at("id", "else"):
                self.take()
                self.stmt(live and not c)
            self.live = live
        else:
            while self.at("id") and self.peek()[1] in ("const", "int", "bool", "string"):
                self.take()
            name = self.take("id")[1]
            if self.at("op", "("):                        # error("...");
                self.take()
                self.expr()
                self.take("op", ")")
                self.take("op", ";")
                assert not live, f"the reference calls {name}() for this variant"
                return
            if self.at("op", ","):                        # string hdr, pcomp;
                while not self.at("op", ";"):
                    self.take()
                self.take()
                return
            op = self.take("op")[1]
            assert op in ("=", "+="), op
            v = self.expr()
            self.take("op", ";")
            if live:
                self.env[name] = v if op == "=" else self.env.get(name, "") + v

    # expressions (C precedence, the operators the region uses) ---------------------------------------------------
    def expr(self):
        c = self.lor()
        if self.at("op", "?"):
            self.take()
            a = self.expr()
            self.take("op", ":")
            b = self.expr()
            return a if c else b
        return c

    def lor(self):
        v = self.land()
        while self.at("op", "||"):
            self.take()
            r = self.land()
            v = bool(v) or bool(r)
        return v

    def land(self):
        v = self.band()
        while self.at("op", "&&"):
            self.take()
            r = self.band()
            v = bool(v) and bool(r)
        return v

    def band(self):
        v = self.eq()
        while self.at("op", "&"):
            self.take()
            v = int(v) & int(self.eq())
        return v

    def eq(self):
        v = self.rel()
        while self.at("op", "==") or self.at("op", "!="):
            op = self.take()[1]
            r = self.rel()
            v = (v == r) if op == "==" else (v != r)
        return v

    def rel(self):
        v = self.shift()
        while self.at("op") and self.peek()[1] in ("<", "<=", ">", ">="):
            op = self.take()[1]
            r = self.shift()
            v = {"<": v < r, "<=": v <= r, ">": v > r, ">=": v >= r}[op]
        return v

    def shift(self):
        v = self.add()
        while self.at("op") and self.peek()[1] in ("<<", ">>"):
            op = self.take()[1]
            r = self.add()
            v = int(v) << int(r) if op == "<<" else int(v) >> int(r)
        return v

    def add(self):
        v = self.mul()
        while self.at("op") and self.peek()[1] in ("+", "-"):
            op = self.take()[1]
            r = self.mul()
            if isinstance(v, str) or isinstance(r, str):
                assert op == "+"
                v = str(v) + str(r)
            else:
                v = int(v) + int(r) if op == "+" else int(v) - int(r)
        return v

    def mul(self):
        v = self.unary()
        while self.at("op") and self.peek()[1] in ("*", "/", "%"):
            op = self.take()[1]
            r = self.unary()
            v = int(v) * int(r) if op == "*" else int(v) // int(r) if op == "/" else int(v) % int(r)
        return v

    def unary(self):
        if self.at("op", "!"):
            self.take()
            return not self.unary()
        if self.at("op", "-"):
            self.take()
            return -int(self.unary())
        return self.primary()

    def primary(self):
        tk = self.take()
        if tk[0] == "num":
            return tk[1]
        if tk[0] == "str":
            s = tk[1]
            while self.at("str"):                          # adjacent literals concatenate
                s += self.take()[1]
            return s
        if tk[0] == "op" and tk[1] == "(":
            v = self.expr()
            self.take("op", ")")
            return v
        assert tk[0] == "id", tk
        if self.at("op", "("):
            self.take()
            a = self.expr()
            self.take("op", ")")
            assert tk[1] == "itos", tk
            return str(int(a))
        if self.at("op", "["):
            self.take()
            k = self.expr()
            self.take("op", "]")
            return self.env[tk[1]][int(k)] if self.live else 0
        return self.env[tk[1]] if self.live else self.env.get(tk[1], 0)


def _norm(text):
    """ZPAQL tokens: comments in (nested) parentheses and white space dropped."""
    out, depth = [], 0
    for ch in text:
        if ch == "(":
            depth += 1
        elif ch == ")" and depth:
            depth -= 1
        elif not depth:
            out.append(ch)
    return "".join(out).split()


def _ref_postprocessor(args):
    """What LibZPAQ.cs:423-830 leaves in (hdr, pcomp) for these arguments, by evaluating the reference's text."""
    lines = _src()
    a = next(i for i, l in enumerate(lines) if "// Generate the postprocessor" in l)
    b = next(i for i, l in enumerate(lines) if "// Build context model" in l)
    assert 415 < a < 430 and 825 < b < 840, (a, b)           # the region SURVEY.md / DESIGN.md cite
    env = {"args": list(args)}
    _Eval(_tokens("\n".join(lines[a + 1:b])), env).run()
    return env["hdr"], env["pcomp"]


_VARIANTS = [(n1, n2) for n2 in range(0, 8) for n1 in (0, 4, 5, 7)]


@pytest.mark.parametrize("n1,n2", _VARIANTS)
def test_postprocessor_programs_equal_the_reference_text(n1, n2):
    """Every (level, E8E9, block-size) variant of the generated PCOMP: tools/methods == LibZPAQ.cs:423-830."""
    method = f"x{n1},{n2},4,0,3,{n1 + 20 - 4 if n2 & 3 == 1 else 16}"
    _, args, _ = methods.parse_args(method)
    hdr, pcomp = _ref_postprocessor(args)
    text, _ = methods.make_config(method)
    def subst(t):                                                    # $N and $N+M: the Compiler's argument substitution (Compiler.cs:248-262)
        return re.sub(r"\$(\d)(?:\+(\d+))?", lambda m: str(args[int(m.group(1)) - 1] + int(m.group(2) or 0)), t)

    want_hdr = _norm(subst(hdr))
    pcomp = subst(pcomp)
    got = _norm(text)
    assert got[:len(want_hdr)] == want_hdr                          # "comp 9 16 ph pm" (the component count follows)
    k = got.index("halt") + 1                                        # end of HCOMP ("hcomp ... halt")
    assert got[k:] == _norm(pcomp), (method, got[k:k + 12])
    if n2 & 3 == 1:
        assert ("r=a 5" in " ".join(got)) == (n1 > 4)               # the rb > 0 variant really differs


def _literals(lo, hi):
    """String literals of LibZPAQ.cs lines lo..hi (1-based) that end up in the config text, in source order, as ZPAQL
    token lists.  Source comments, char literals and the argument of strchr() are skipped; a ZPAQL comment may run
    over several literals, so the parenthesis depth carries from one literal to the next."""
    out, depth = [], 0
    text = "\n".join(_src()[lo - 1:hi])
    for m in re.finditer(r"""//[^\n]*|'(?:[^'\\]|\\.)'|("(?:[^"\\]|\\.)*")""", text):
        if m.group(1) is None or text[:m.start()].rstrip().endswith("strchr("):
            continue
        kept = []
        for ch in bytes(m.group(1)[1:-1], "utf-8").decode("unicode_escape"):
            if ch == "(":
                depth += 1
            elif ch == ")" and depth:
                depth -= 1
            elif not depth:
                kept.append(ch)
        toks = "".join(kept).split()
        if toks:
            out.append(toks)
    return out


# methods that together take every branch of the context-model generator (LibZPAQ.cs:835-1041): each component letter,
# periodic / distance / masked / lz77-state / skip contexts, wide mixers, ISSE chains, MATCH, word models
_MODEL_METHODS = [
    "x4,0ci1,1,2a24,1,1m16,24t8,24s8,32,255",          # c i a m t s
    "x4,0c256,12,255,127,300,1300,1000w2,65,26,223,20,1m",   # cm with limit, periodic (not a power of 2), masks, skips, word
    "x4,0c0,16,255c0,1010,255m20",                      # periodic power of two, distance context, 20-bit mixer context
    "x4,2,12,0,7,16c0,0,511,300i2",                     # level 2: lz77 parse state in R1/R2, lz77-state contexts
    "x4,6,12,0,7,16c0,0,256",                           # level 2 + E8E9 (the other skip constant)
    "x4,0c0,0,255,2000m",                               # skip of >= 256 bytes
]


def test_context_model_generator_emits_the_reference_literals_in_order():
    """LibZPAQ.cs:835-1041: every string literal of the comp / hcomp generator appears, in source order, in the config
    of a method that takes its branch; and every token tools/methods emits there is the reference's."""
    lines = _src()
    a = next(i for i, l in enumerate(lines) if "// Build context model" in l) + 1
    b = next(i for i, l in enumerate(lines) if "return hdr+itos(ncomp)" in l) + 1
    assert 830 < a < 845 and 1035 < b < 1045, (a, b)
    lits = _literals(a, b)
    outs = []
    for mt in _MODEL_METHODS:
        text, _ = methods.make_config(mt)
        toks = _norm(text)
        outs.append(toks[:toks.index("halt") + 1])
    ref_vocab = {t for x in lits for t in x} | {"halt"}
    seen = [False] * len(lits)
    for toks in outs:
        # (1) nothing foreign: every token is a reference token or a number (itos(...))
        head = toks[:6]                                              # comp hh hm ph pm n
        assert head[0] == "comp" and all(t.isdigit() for t in head[1:]), head
        for t in toks[6:]:
            assert t in ref_vocab or re.fullmatch(r"-?\d+", t), t
        # (2) literals in source order: greedy scan; loops of the generator may repeat a literal, so the scan restarts
        # from the last match of an earlier literal when needed
        for k, lit in enumerate(lits):
            for p in range(len(toks) - len(lit) + 1):
                if toks[p:p + len(lit)] == lit:
                    seen[k] = True
                    break
    missing = [" ".join(lits[k]) for k in range(len(lits)) if not seen[k]]
    assert not missing, missing
    # order inside one pass of the generator's loop body: for the single-component methods the literals found must
    # appear in increasing source order
    for mt in ("x4,0c0,0,255", "x4,0ci1", "x4,0ca24", "x4,0cw2"):
        toks = _norm(methods.make_config(mt)[0])
        toks = toks[:toks.index("halt") + 1]
        pos = 0
        for lit in lits:
            if len(lit) < 3:
                continue                                            # (short fragments recur all over the program)
            for p in range(pos, len(toks) - len(lit) + 1):
                if toks[p:p + len(lit)] == lit:
                    pos = p + len(lit)
                    break
