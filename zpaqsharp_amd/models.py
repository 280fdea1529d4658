"""Block models used by the tests, fixtures and benchmarks, as ZPAQL config text.

`min`, `mid`, `max` are the three built-in models of the reference
(`Compressor.startBlock(1..3)`, bytecodes at Compressor.cs:48-74; the ZPAQ
distribution calls them min.cfg / mid.cfg / max.cfg).  They are kept here as
*source text*; tests/test_oracle_pins.py checks that assembling them gives the
reference's hsize (26 / 69 / 196) and header CRCs.

`l1` is the build-defined smallest modelled stream of BASELINE.json configs 1-2
(SURVEY.md §8d): one direct order-1 CM.  `e8e9` is the reference's E8E9
post-processor program (LibZPAQ.cs:802-826), the inverse of its forward
transform (LibZPAQ.cs:372-384); BASELINE.json config 5 = `max` model + this PCOMP.
"""
from __future__ import annotations

from functools import lru_cache

from .zpaql import Model, assemble

L1_CFG = """
comp 0 0 0 0 1
  0 cm 17 255
hcomp
  a<<= 9 *d=a halt
end
"""

MIN_CFG = """
comp 1 2 0 0 2
  0 icm 16
  1 isse 19 0
hcomp
  *b=a a=0 d=0 hash b-- hash *d=a
  d++ b-- hash b-- hash *d=a
  halt
end
"""

MID_CFG = """
comp 3 3 0 0 8
  0 icm 5
  1 isse 13 0
  2 isse 17 1
  3 isse 18 2
  4 isse 18 3
  5 isse 19 4
  6 match 22 24
  7 mix 16 0 7 24 255
hcomp
  c++ *c=a b=c a=0 (save in rotating buffer M)
  d= 1 hash *d=a   (orders 1..5 for the ISSE chain)
  b-- d++ hash *d=a
  b-- d++ hash *d=a
  b-- d++ hash *d=a
  b-- d++ hash *d=a
  b-- d++ hash b-- hash *d=a (order 7 for MATCH)
  d++ a=*c a<<= 8 *d=a       (order 1 for MIX)
  halt
end
"""

MAX_CFG = """
comp 5 9 0 0 22
  0 const 160
  1 icm 5
  2 isse 13 1
  3 isse 16 2
  4 isse 18 3
  5 isse 19 4
  6 isse 19 5
  7 isse 20 6
  8 match 22 24
  9 icm 17
  10 isse 19 9
  11 icm 13
  12 icm 13
  13 icm 13
  14 icm 14
  15 mix 16 0 15 24 255
  16 mix 8 0 16 10 255
  17 mix2 0 15 16 24 0
  18 sse 8 17 32 255
  19 mix2 8 17 18 16 255
  20 sse 16 19 32 255
  21 mix2 0 19 20 16 0
hcomp
  c++ *c=a b=c a=0 (save in rotating buffer)
  d= 2 hash *d=a b-- (orders 1,2,3,4,5,7)
  d++ hash *d=a b--
  d++ hash *d=a b--
  d++ hash *d=a b--
  d++ hash *d=a b--
  d++ hash b-- hash *d=a b--
  d++ hash *d=a b-- (match, order 8)
  d++ a=*c a&~ 32 (case-insensitive letters -> word orders 0,1 in H[9..10])
  a> 64 jf 14 a< 91 jf 10
    d++ hashd d-- *d<>a a+=*d a*= 20 *d=a
  jmp 9
    a=*d a== 0 jt 3
      d++ *d=a d--
    *d=0
  d++ d++ b=c b-- a=0 hash *d=a (sparse order-2 contexts)
  d++ b-- a=0 hash *d=a
  d++ b-- a=0 hash *d=a
  d++ a=b a-= 212 b=a a=0 hash *d=a (2-D contexts for tables)
    b<>a a-= 216 b<>a a=*b a&= 60 hashd
  d++ a=*c a<<= 9 *d=a (mixer contexts)
  d++ d++ d++ d++ d++ *d=a
  halt
end
"""

# The reference's E8E9 post-processor, as its config text (makeConfig, LibZPAQ.cs:802-826, emitted for methods whose
# args[1] is 4: "pcomp e8e9 d ;").  It inverts the forward transform e8e9() of LibZPAQ.cs:372-384.  B is a shift
# register of the last 4 input bytes, the byte shifted out of it is parked in M[0] (ph = pm = 0: M has one byte), C
# counts the bytes held back (<= 4 pending at any time; flushed at EOF, a > 255).
E8E9_PCOMP = """
pcomp e8e9 d ;
  a> 255 if
    a=c a> 4 if
      c= 4
    else
      a! a+= 5 a<<= 3 d=a a=b a>>=d b=a
    endif
    do a=c a> 0 if
      a=b out a>>= 8 b=a c--
    forever endif
  else
    *b=b a<<= 24 d=a a=b a>>= 8 a+=d b=a c++
    a=c a> 4 if
      a=*b out
      a&= 254 a== 232 if
        a=b a>>= 24 a++ a&= 254 a== 0 if
          a=b a>>= 24 a<<= 24 d=a
          a=b a-=c a+= 5
          a<<= 8 a>>= 8 a|=d b=a
        endif
      endif
    endif
  endif
  halt
end
"""


# A byte-oriented LZ77 post-processor with a 64 KiB history in M (the reference builds its LZ77 programs per method
# string, LibZPAQ.cs:427-639; this one is this repo's own, written for the tests of PCOMP programs whose memory lives
# in HBM).  Coded form (synth.lz77_encode): token t < 128: t + 1 literal bytes follow;  t >= 128: a match of
# (t & 127) + 3 bytes at distance lo + 256 * hi (two bytes follow, distance >= 1; source and destination may overlap).
# R0 = mode (0 token, 1 literals, 2 distance low byte, 3 distance high byte), R1 = bytes left, R2 = distance low byte,
# C = write position in the history ring M.
LZ77_PCOMP = """
pcomp lz77inv ;
  a> 255 if halt endif           (end of segment: nothing is buffered)
  b=a                            (the coded byte)
  a=r 0
  a== 0 if                       (token)
    a=b a> 127 if
      a&= 127 a+= 3 r=a 1  a= 2 r=a 0
    else
      a++ r=a 1  a= 1 r=a 0
    endif
    halt
  endif
  a== 1 if                       (literal)
    a=b *c=a c++ out
    a=r 1 a-- r=a 1
    a== 0 if r=a 0 endif
    halt
  endif
  a== 2 if
    a=b r=a 2  a= 3 r=a 0 halt
  endif
  a=b a<<= 8 d=a a=r 2 a+=d d=a  (distance)
  a=c a-=d b=a                   (source position)
  do
    a=*b *c=a out b++ c++
    a=r 1 a-- r=a 1 a> 0
  while
  a=0 r=a 0
  halt
end
"""


def _with_pcomp(cfg: str, pcomp: str, pm: int) -> str:
    """Attach a PCOMP section to a config and set its M size (pm)."""
    lines = cfg.strip().splitlines()
    head = lines[0].split()
    head[4] = str(pm)
    body = "\n".join(lines[1:])
    assert body.rstrip().endswith("end")
    body = body.rstrip()[:-3]
    return " ".join(head) + "\n" + body + pcomp.strip() + "\n"


@lru_cache(maxsize=None)
def get(name: str) -> Model:
    """Models by name: l1, min, mid, max, and `<name>+e8e9` / `<name>+lz77` for any of them."""
    base, _, post = name.partition("+")
    cfg = {"l1": L1_CFG, "min": MIN_CFG, "mid": MID_CFG, "max": MAX_CFG}[base]
    if post == "":
        return assemble(cfg)
    if post == "e8e9":
        return assemble(_with_pcomp(cfg, E8E9_PCOMP, pm=0))
    if post == "lz77":
        return assemble(_with_pcomp(cfg, LZ77_PCOMP, pm=16))
    raise KeyError(name)


NAMES = ("l1", "min", "mid", "max")
