"""The reference's method strings: block configs with its LZ77 / BWT / E8E9 post-processors, and the matching
pre-processors (host tooling for fixtures; the decode side is the GPU's job).

`make_config(method)` is the equivalent of `LibZPAQ.makeConfig` (LibZPAQ.cs:388-1044): it turns an expanded method
string `{x|s|0}N1,N2,...[{c|i|a|m|t|s|w}N...]...` into ZPAQL config text — `comp`/`hcomp` generated from the component
letters, and the PCOMP program of the chosen pre-processing level:

    level = N2 & 3:  0 none, 1 `lazy2` (bit-packed LZ77, LibZPAQ.cs:427-572), 2 `lzpre` (byte-aligned LZ77, :575-639),
                     3 `bwtrle` (inverse BWT, :642-795);   N2 in 4..7 adds E8E9 (the stand-alone E8E9 program :802-826)

The PCOMP source texts below are the reference's programs (they are data the decoder must run, like the built-in model
bytecodes in models.py); the generator around them is this module's own code.

`preprocess(data, args)` is the equivalent of `LZBuffer` (LZBuffer.cs:96-115 formats, :225-486): it produces the byte
stream those PCOMP programs invert.  Match finding here is a plain greedy hash search — only the CODE FORMAT has to
agree with the reference, not its parse.
"""
from __future__ import annotations

import re
from typing import List, Tuple

import numpy as np

from zpaqsharp_amd import zpaql


def _lg(x: int) -> int:
    """floor(log2(x)) + 1 (LZBuffer.cs:116-126)."""
    return int(x).bit_length()


def _nbits(x: int) -> int:
    return bin(x).count("1")


def parse_args(method: str) -> Tuple[str, List[int], str]:
    """'x4,1,4,0,3,24ci1' -> ('x', [4,1,4,0,3,24,0,0,0], 'ci1')   (LibZPAQ.cs:394-416)."""
    typ = method[0]
    if typ not in "xs0i":
        raise ValueError("method must start with x, s, 0 or i")
    args = [0] * 9
    i, k = 1, 0
    while i < len(method) and k < 9 and (method[i].isdigit() or method[i] in ",."):
        if method[i].isdigit():
            args[k] = args[k] * 10 + int(method[i])
        else:
            k += 1
            if k < 9:
                args[k] = 0
        i += 1
    return typ, args, method[i:]


_E8E9_TAIL = """
    d=b b=0 do
      a=b a==d ifnot
        a+= 4 a<d if
          a=*b a&= 254 a== 232 if
            c=b b++ b++ b++ b++ a=*b a++ a&= 254 a== 0 if
              b-- a=*b
              b-- a<<= 8 a+=*b
              b-- a<<= 8 a+=*b
              a-=b a++
              *b=a a>>= 8 b++
              *b=a a>>= 8 b++
              *b=a b++
            endif
            b=c
          endif
        endif
        a=*b out b++
      forever
    endif
"""


def _pcomp_lazy2(args: List[int], doe8: bool) -> str:
    """LibZPAQ.cs:427-572."""
    rb = args[0] - 4 if args[0] > 4 else 0
    p = """pcomp lazy2 3 ;

  a> 255 if
"""
    if doe8:
        p += _E8E9_TAIL.replace("    d=b b=0 do", "    b=0 d=r 4 do")
    p += """
    a=0 b=0 c=0 d=0 r=a 1 r=a 2 r=a 3 r=a 4
    halt
  endif

  a<<=d a+=c c=a
  a= 8 a+=d d=a

  a=r 1 a== 0 if
    a= 1 r=a 2
    a=c a&= 3 a> 0 if
      a-- a<<= 3 r=a 3
      a=c a>>= 2 c=a
      b=r 3 a&= 7 a+=b r=a 3
      a=c a>>= 3 c=a
      a=d a-= 5 d=a
      a= 1 r=a 1
    else
      a=c a>>= 2 c=a
      d-- d--
      a= 3 r=a 1
    endif
  endif

  do a=r 1 a== 1 if a=d a> 2 if
    a=c a&= 1 a== 1 if
      a=c a>>= 1 c=a
      b=r 2 a=c a&= 1 a+=b a+=b r=a 2
      a=c a>>= 1 c=a
      d-- d--
    else
      a=c a>>= 1 c=a
      a=r 2 a<<= 2 b=a
      a=c a&= 3 a+=b r=a 2
      a=c a>>= 2 c=a
      d-- d-- d--
"""
    p += f"      a= {5 if rb else 2} r=a 1\n"
    p += """    endif
  forever endif endif
"""
    if rb:
        p += f"""
  a=r 1 a== 5 if a=d a> {rb - 1} if
    a=c a&= {(1 << rb) - 1} r=a 5
    a=c a>>= {rb} c=a
    a=d a-= {rb} d=a
    a= 2 r=a 1
  endif endif
"""
    p += """
  a=r 1 a== 2 if a=r 3 a>d ifnot
    a=c r=a 6 a=d r=a 7
    b=r 3 a= 1 a<<=b d=a
    a-- a&=c a+=d
"""
    if rb:
        p += f"    a<<= {rb} d=r 5 a+=d a-= {(1 << rb) - 1}\n"
    p += """    d=a b=r 4 a=b a-=d c=a

    d=r 2 do a=d a> 0 if d--
      a=*c *b=a c++ b++
"""
    if not doe8:
        p += " out\n"
    p += """    forever endif
    a=b r=a 4

    a=r 6 b=r 3 a>>=b c=a
    a=r 7 a-=b d=a
    a=0 r=a 1
  endif endif

  do a=r 1 a== 3 if a=d a> 1 if
    a=c a&= 1 a== 1 if
      a=c a>>= 1 c=a
      b=r 2 a&= 1 a+=b a+=b r=a 2
      a=c a>>= 1 c=a
      d-- d--
    else
      a=c a>>= 1 c=a
      d--
      a= 4 r=a 1
    endif
  forever endif endif

  a=r 1 a== 4 if a=d a> 7 if
    b=r 4 a=c *b=a
"""
    if not doe8:
        p += " out\n"
    p += """    b++ a=b r=a 4
    a=c a>>= 8 c=a
    a=d a-= 8 d=a
    a=r 2 a-- r=a 2 a== 0 if
      a=0 r=a 1
    endif
  endif endif
  halt
end
"""
    return p


def _pcomp_lzpre(args: List[int], doe8: bool) -> str:
    """LibZPAQ.cs:575-639."""
    p = """pcomp lzpre c ;

  a> 255 if
"""
    if doe8:
        p += _E8E9_TAIL
    p += f"""    b=0 c=0 d=0 a=0 r=a 1 r=a 2
  halt
  endif

  c=a a=d a== 0 if
    a=c a>>= 6 a++ d=a
    a== 1 if
      a+=c r=a 1 a=0 r=a 2
    else
      d++ a=c a&= 63 a+= {args[2]} r=a 1 a=0 r=a 2
    endif
  else
    a== 1 if
      a=c *b=a b++
"""
    if not doe8:
        p += " out\n"
    p += """      a=r 1 a-- a== 0 if d=0 endif r=a 1
    else
      a> 2 if
        a=r 2 a<<= 8 a|=c r=a 2 d--
      else
        a=r 2 a<<= 8 a|=c c=a a=b a-=c a-- c=a
        d=r 1
        do
          a=*c *b=a c++ b++
"""
    if not doe8:
        p += " out\n"
    p += """        d-- a=d a> 0 while

      endif
    endif
  endif
  halt
end
"""
    return p


def _pcomp_bwtrle(args: List[int], doe8: bool) -> str:
    """LibZPAQ.cs:642-795."""
    p = """pcomp bwtrle c ;

  a> 255 ifnot
    *b=a b++

  elsel

    b-- a=*b
    b-- a<<= 8 a+=*b
    b-- a<<= 8 a+=*b
    b-- a<<= 8 a+=*b c=a r=a 1

    a=b r=a 2

    do
      a=b a> 0 if
        b-- a=*b a++ a&= 255 d=a d! *d++
      forever
    endif

    d=0 d! *d= 1 a=0
    do
      a+=*d *d=a d--
    d<>a a! a> 255 a! d<>a until

    b=0 do
      a=c a>b if
        d=*b d! *d++ d=*d d-- *d=b
      b++ forever
    endif

    b=c b++ c=r 2 do
      a=c a>b if
        d=*b d! *d++ d=*d d-- *d=b
      b++ forever
    endif
"""
    if args[0] <= 4:
        p += """
    b=0 do
      a=c a>b if
        d=b a=*d a<<= 8 a+=*b *d=a
      b++ forever
    endif

    d=r 1 b=0 do
      a=d a== 0 ifnot
        a=*d a>>= 8 d=a
"""
        p += " *b=*d b++\n" if doe8 else " a=*d out\n"
        p += """      forever
    endif
"""
        if doe8:
            p += "\n" + _E8E9_TAIL
        p += """  endif
  halt
end
"""
    elif doe8:
        p += """
    a=r 2 a-- r=a 2

    c=0 d=r 1 do
      a=d a== 0 ifnot
        d=*d

        b=d a=*b a<<= 24 b=a
        a=r 4 r=a 5 a>>= 8 a|=b r=a 4

        a=c a> 3 if
          a=r 5 a&= 254 a== 232 if
            a=r 4 a>>= 24 b=a a++ a&= 254 a< 2 if
              a=r 4 a-=c a+= 4 a<<= 8 a>>= 8
              b<>a a<<= 24 a+=b r=a 4
            endif
          endif
        endif

        a=c a> 3 if a=r 5 out endif c++

      forever
    endif

    b=r 4
    a=c a> 3 a=b if out endif a>>= 8 b=a
    a=c a> 2 a=b if out endif a>>= 8 b=a
    a=c a> 1 a=b if out endif a>>= 8 b=a
    a=c a> 0 a=b if out endif

  endif
  halt
end
"""
    else:
        p += """
    d=r 1 do
      a=d a== 0 ifnot
        d=*d
        b=d a=*b out
      forever
    endif
  endif
  halt
end
"""
    return p


def make_config(method: str) -> Tuple[str, List[int]]:
    """Config text (with the $-arguments already substituted) and the nine numeric arguments of a method string."""
    from zpaqsharp_amd.models import E8E9_PCOMP
    typ, args, rest = parse_args(method)
    if typ == "0":
        return "comp 0 0 0 0 0 hcomp end\n", args
    level, doe8 = args[1] & 3, 4 <= args[1] <= 7
    membits = args[0] + 20
    if level == 1:
        hdr, pcomp = f"comp 9 16 0 {membits} ", _pcomp_lazy2(args, doe8)
    elif level == 2:
        hdr, pcomp = f"comp 9 16 0 {membits} ", _pcomp_lzpre(args, doe8)
    elif level == 3:
        hdr, pcomp = f"comp 9 16 {membits} {membits} ", _pcomp_bwtrle(args, doe8)
    else:
        hdr, pcomp = "comp 9 16 0 0 ", (E8E9_PCOMP.strip() + "\n" if doe8 else "end\n")

    # ---- context model (LibZPAQ.cs:835-1041): H[0..254] contexts, H[255..511] position of the last byte i-255,
    # M = last 64K bytes filling backward, C = pointer to the most recent byte; level 2 keeps its parse state in R1, R2
    ncomp, sb = 0, 5
    comp: List[str] = []
    hc: List[str] = ["hcomp", "c-- *c=a a+= 255 d=a *d=c"]
    if level == 2:
        hc.append(f"""  a=r 1 a== 0 if
    a= {111 + 57 * int(doe8)}
  else a== 1 if
    a=*c r=a 2
    a> 63 if a>>= 6 a++ a++
    else a++ a++ endif
  else
    a--
  endif endif
  r=a 1""")
    for m in re.finditer(r"([a-z])([0-9,.]*)", rest):
        if ncomp >= 254:
            break
        letter = m.group(1)
        v = [int(x) if x else 0 for x in re.split(r"[,.]", m.group(2))] if m.group(2) else []
        if letter == "c":                                  # context model: N1 limit / memory, N2 offset, N3.. masks
            v += [0] * (2 - len(v)) if len(v) < 2 else []
            sb = 11
            sb += _lg(v[1]) if v[1] < 256 else 6
            for x in v[2:]:
                if x < 512:
                    sb += _nbits(x) * 3 // 4
            sb = min(sb, membits)
            if v[0] % 1000 == 0:
                comp.append(f"{ncomp} icm {sb - 6 - v[0] // 1000}")
            else:
                comp.append(f"{ncomp} cm {sb - 2 - v[0] // 1000} {v[0] % 1000 - 1}")
            hc.append(f"d= {ncomp} *d=0")
            if 1 < v[1] <= 255:
                hc.append(f"a=c a&= {v[1] - 1} hashd" if _lg(v[1]) != _lg(v[1] - 1) else f"a=c a%= {v[1]} hashd")
            elif 1000 <= v[1] <= 1255:
                hc.append(f"a= 255 a+= {v[1] - 1000} d=a a=*d a-=c a> 255 if a= 255 endif d= {ncomp} hashd")
            for k, x in enumerate(v[2:]):
                line = "b=c " if k == 0 else ""
                if x == 255:
                    line += "a=*b hashd"
                elif 0 < x < 255:
                    line += f"a=*b a&= {x} hashd"
                elif 256 <= x < 512:
                    line += ("a=r 1 a> 1 if\n  a=r 2 a< 64 if\n    a=*b " + (f"a&= {x - 256}" if x < 511 else "") +
                             " hashd\n  else\n    a>>= 6 hashd a=r 1 hashd\n  endif\nelse\n  a= 255 hashd a=r 2 hashd\nendif")
                elif x >= 1256:
                    line += f"a= {((x - 1000) >> 8) & 255} a<<= 8 a+= {(x - 1000) & 255} a+=b b=a"
                elif x > 1000:
                    line += f"a= {x - 1000} a+=b b=a"
                if x < 512 and k < len(v[2:]) - 1:
                    line += "\nb++ "
                hc.append(line)
            ncomp += 1
        elif letter in "mts" and ncomp > int(letter == "t"):
            if len(v) < 1:
                v.append(8)
            if len(v) < 2:
                v.append(24 + 8 * int(letter == "s"))
            if letter == "s" and len(v) < 3:
                v.append(255)
            sb = 5 + v[0] * 3 // 4
            if letter == "m":
                comp.append(f"{ncomp} mix {v[0]} 0 {ncomp} {v[1]} 255")
            elif letter == "t":
                comp.append(f"{ncomp} mix2 {v[0]} {ncomp - 1} {ncomp - 2} {v[1]} 255")
            else:
                comp.append(f"{ncomp} sse {v[0]} {ncomp - 1} {v[1]} {v[2]}")
            if v[0] > 8:
                hc.append(f"d= {ncomp} *d=0 b=c a=0")
                w = v[0]
                while w >= 16:
                    hc.append("a<<= 8 a+=*b" + (" b++" if w > 16 else ""))
                    w -= 8
                if w > 8:
                    hc.append(f"a<<= 8 a+=*b a>>= {16 - w}")
                hc.append("a<<= 8 *d=a")
            ncomp += 1
        elif letter == "i" and ncomp > 0:                  # ISSE chain, context order growing by N1, N2, ...
            hc.append(f"d= {ncomp - 1} b=c a=*d d++")
            for k, x in enumerate(v):
                if ncomp >= 254:
                    break
                line = ""
                for j in range(x % 10):
                    line += "hash "
                    if k < len(v) - 1 or j < x % 10 - 1:
                        line += "b++ "
                    sb += 6
                line += "*d=a" + (" d++" if k < len(v) - 1 else "")
                hc.append(line)
                sb = min(sb, membits)
                comp.append(f"{ncomp} isse {sb - 6 - x // 10} {ncomp - 1}")
                ncomp += 1
        elif letter == "a":                                # MATCH
            if len(v) < 1:
                v.append(24)
            v += [0] * (3 - len(v))
            comp.append(f"{ncomp} match {membits - v[2] - 2} {membits - v[1]}")
            hc.append(f"d= {ncomp} a=*d a*= {v[0]} a+=*c a++ *d=a")
            sb = 5 + (membits - v[1]) * 3 // 4
            ncomp += 1
        elif letter == "w":                                # ICM-ISSE chain over word contexts
            dflt = [1, 65, 26, 223, 20, 0]
            v += dflt[len(v):]
            comp.append(f"{ncomp} icm {membits - 6 - v[5]}")
            for i in range(1, v[0]):
                comp.append(f"{ncomp + i} isse {membits - 6 - v[5]} {ncomp + i - 1}")
            hc.append(f"a=*c a&= {v[3]} a-= {v[1]} a&= 255 a< {v[2]} if")
            for i in range(v[0]):
                hc.append(("  d= %d" % ncomp if i == 0 else "  d++") + f" a=*d a*= {v[4]} a+=*c a++ *d=a")
            hc.append("else")
            for i in range(v[0] - 1, 0, -1):
                hc.append(f"  d= {ncomp + i - 1} a=*d d++ *d=a")
            hc.append(f"  d= {ncomp} *d=0\nendif")
            ncomp += v[0] - 1
            sb = membits - v[5]
            ncomp += 1
    text = hdr + str(ncomp) + "\n" + "\n".join(comp) + "\n" + "\n".join(hc) + "\nhalt\n" + pcomp
    return text, args


def model_of(method: str) -> Tuple[zpaql.Model, List[int]]:
    text, args = make_config(method)
    return zpaql.assemble(text), args


# ---------------------------------------------------------------------------------------------------------------------
# pre-processors (LZBuffer equivalents)
# ---------------------------------------------------------------------------------------------------------------------
def e8e9_forward(data: bytes) -> bytes:
    """LibZPAQ.cs:372-384."""
    b = bytearray(data)
    for i in range(len(b) - 5, -1, -1):
        if (b[i] & 254) == 0xE8 and ((b[i + 4] + 1) & 254) == 0:
            a = ((b[i + 1] | b[i + 2] << 8 | b[i + 3] << 16) + i) & 0xFFFFFF
            b[i + 1], b[i + 2], b[i + 3] = a & 255, (a >> 8) & 255, (a >> 16) & 255
    return bytes(b)


class _BitWriter:
    def __init__(self):
        self.out, self.bits, self.n = bytearray(), 0, 0

    def putb(self, x: int, k: int):                        # LSB first (LZBuffer.cs:50-63)
        x &= (1 << k) - 1
        self.bits |= x << self.n
        self.n += k
        while self.n > 7:
            self.out.append(self.bits & 255)
            self.bits >>= 8
            self.n -= 8

    def flush(self):
        if self.n > 0:
            self.out.append(self.bits & 255)
        self.bits = self.n = 0


def _matches(d: bytes, min_match: int, max_match: int, max_off: int):
    """Greedy parse: yields ('lit', start, end) / ('match', length, offset)."""
    n, i, lit0 = len(d), 0, 0
    last = {}
    while i < n:
        best = 0
        if i + min_match <= n:
            key = d[i:i + min_match]
            j = last.get(key, -1)
            if j >= 0 and 0 < i - j <= max_off:
                m = min_match
                while m < max_match and i + m < n and d[j + m] == d[i + m]:
                    m += 1
                best, off = m, i - j
            last[key] = i
        if best:
            if i > lit0:
                yield ("lit", lit0, i)
            yield ("match", best, off)
            for k in range(i + 1, min(i + best, n - min_match + 1)):
                last[d[k:k + min_match]] = k
            i += best
            lit0 = i
        else:
            i += 1
    if n > lit0:
        yield ("lit", lit0, n)


def lz77_level1(data: bytes, args: List[int]) -> bytes:
    """Bit-packed codes of LZBuffer level 1 (LZBuffer.cs:96-107, write_literal :387-405, write_match :422-446)."""
    rb = args[0] - 4 if args[0] > 4 else 0
    min_match = max(4, args[2])
    w = _BitWriter()
    for item in _matches(data, min_match, 1 << 16, (1 << 23) - 1):
        if item[0] == "lit":
            _, a, b = item
            lit = b - a
            ll = _lg(lit)
            w.putb(0, 2)
            ll -= 1
            while ll > 0:
                ll -= 1
                w.putb(1, 1)
                w.putb((lit >> ll) & 1, 1)
            w.putb(0, 1)
            for c in data[a:b]:
                w.putb(c, 8)
        else:
            _, ln, off = item
            ll = _lg(ln) - 1
            off += (1 << rb) - 1
            lo = _lg(off) - 1 - rb
            assert 0 <= lo <= 23 and ll >= 2
            w.putb((lo + 8) >> 3, 2)
            w.putb(lo & 7, 3)
            while ll > 2:
                ll -= 1
                w.putb(1, 1)
                w.putb((ln >> ll) & 1, 1)
            w.putb(0, 1)
            w.putb(ln & 3, 2)
            w.putb(off, rb)
            w.putb(off >> rb, lo)
    w.flush()
    return bytes(w.out)


def lz77_level2(data: bytes, args: List[int]) -> bytes:
    """Byte-aligned codes of LZBuffer level 2 (LZBuffer.cs:109-112, write_literal :406-418, write_match :449-485)."""
    m = args[2]
    assert 1 <= m <= 64
    out = bytearray()
    for item in _matches(data, max(m, 3), m + 63 + 4 * 64, (1 << 24) - 1):
        if item[0] == "lit":
            _, a, b = item
            while a < b:
                k = min(64, b - a)
                out.append(k - 1)
                out += data[a:a + k]
                a += k
        else:
            _, ln, off = item
            off -= 1
            while ln > 0:
                len1 = m + 63 if ln > m * 2 + 63 else ln - m if ln > m + 63 else ln
                assert m <= len1 < m + 64
                if off < (1 << 16):
                    out += bytes([64 + len1 - m, off >> 8, off & 255])
                else:
                    out += bytes([128 + len1 - m, off >> 16, (off >> 8) & 255, off & 255])
                ln -= len1
    return bytes(out)


def bwt_level3(data: bytes) -> bytes:
    """LZBuffer.cs:228-240: BWT with the end-of-string byte coded as 255 and its position in the last 4 bytes."""
    n = len(data)
    if n == 0:
        return bytes([255, 0, 0, 0, 0])
    a = np.frombuffer(data, np.uint8)
    # suffix array by prefix doubling (no suffix-sorting library here; fixtures are small)
    rank = a.astype(np.int64)
    sa = np.argsort(rank, kind="stable")
    k = 1
    while True:
        r2 = np.full(n, -1, np.int64)
        r2[:n - k] = rank[k:]
        order = np.lexsort((r2, rank))
        key = rank[order] * (max(n, 256) + 2) + (r2[order] + 1)      # (ranks start as byte values: the radix must exceed 256 for blocks shorter than that)
        nr = np.zeros(n, np.int64)
        nr[order] = np.concatenate([[0], np.cumsum(key[1:] != key[:-1])])
        rank, sa = nr, order
        if nr.max() == n - 1:
            break
        k *= 2
    out = bytearray([data[n - 1]])
    idx = 0
    for i in range(1, n + 1):
        s = int(sa[i - 1])
        if s == 0:
            idx = i
            out.append(255)
        else:
            out.append(data[s - 1])
    out += idx.to_bytes(4, "little")
    return bytes(out)


def preprocess(data: bytes, args: List[int]) -> bytes:
    """What compressBlock feeds the coder (LibZPAQ.cs:296-311): LZBuffer output for levels 1-3, E8E9 for 4-7."""
    level, doe8 = args[1] & 3, 4 <= args[1] <= 7
    d = e8e9_forward(data) if doe8 else data
    if level == 1:
        return lz77_level1(d, args)
    if level == 2:
        return lz77_level2(d, args)
    if level == 3:
        return bwt_level3(d)
    return d


def compress_block(method: str, data: bytes, filename: bytes = b"", pre: bytes = None) -> bytes:
    """One block the way LibZPAQ.compressBlock frames it (tag, header, segment with the size as comment, SHA-1), coded by
    this repo's CPU stream writer; n = 0 models (methods like "x0,1,4,0,3,24") use the unmodelled store layout.
    `pre`: bytes to feed the post-processor instead of preprocess(data) (tests of the PCOMP programs on input no
    encoder writes; the size comment and SHA-1 still describe `data`)."""
    import hashlib

    from zpaqsharp_amd import synth
    model, args = model_of(method)
    if pre is None:
        pre = preprocess(data, args)
    if model.n:
        return synth.compress_block(model, np.frombuffer(data, np.uint8) if data else np.zeros(0, np.uint8), filename=filename,
                                    pre=np.frombuffer(pre, np.uint8) if pre else np.zeros(0, np.uint8))
    # store path (Encoder.cs:39-73 with n == 0): the decoded stream is selector [+ PCOMP] + data in length-prefixed chunks
    dec = (bytes([1, len(model.pcomp) & 255, len(model.pcomp) >> 8]) + model.pcomp if model.pcomp else b"\0") + pre
    body = b"".join(len(dec[i:i + 65536]).to_bytes(4, "big") + dec[i:i + 65536] for i in range(0, len(dec), 65536)) + b"\0\0\0\0"
    tag = bytes([0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3])
    return (tag + b"zPQ" + bytes([2, 1]) + model.header + b"\x01" + filename + b"\0" + str(len(data)).encode() + b"\0\0"
            + body + b"\xfd" + hashlib.sha1(data).digest() + b"\xff")
