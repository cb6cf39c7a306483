"""R-compatible random numbers for the host-side front end (numpy only).

The reference seeds each chain from R's generator (reference R/stan4bart_fit.R:35-38:
``set.seed(seed); control.stan$seed <- sample.int(.Machine$integer.max, 1L)``) and hands the
live ``.Random.seed`` to the C layer through GetRNGstate (reference src/init.cpp:259).  Without
R, the host mirror has to reproduce ``set.seed`` / ``runif`` / ``rnorm`` / ``sample.int`` itself so
that the C-ABI receives exactly the state an R caller would pass.

Generators restated: Mersenne-Twister + Inversion + Rejection (R >= 3.6 defaults), i.e. R's
src/main/RNG.c and src/nmath/{snorm,qnorm}.c.  numpy's MT19937 bit generator produces the same
raw 32-bit stream once its key is set to R's scrambled seed vector.
"""
from __future__ import annotations

import math
import numpy as np

R_RNG_WORDS = 625
_I2_32M1 = 2.328306437080797e-10


def seed_state(seed: int) -> np.ndarray:
    """``set.seed(seed)`` -> 625 uint32 words {mti, mt[624]} (== .Random.seed[2:626])."""
    s = np.uint32(seed & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(50):
            s = np.uint32(69069) * s + np.uint32(1)
        out = np.empty(R_RNG_WORDS, dtype=np.uint32)
        for j in range(R_RNG_WORDS):
            s = np.uint32(69069) * s + np.uint32(1)
            out[j] = s
    out[0] = 624  # FixupSeeds: dummy[0] = mti = N
    return out


class RRng:
    """Stateful R generator; ``state`` is the 625-word vector handed to the C-ABI."""

    def __init__(self, seed: int | None = None, state: np.ndarray | None = None):
        self._bg = np.random.MT19937()
        if state is None:
            state = seed_state(0 if seed is None else seed)
        self.state = state

    @property
    def state(self) -> np.ndarray:
        st = self._bg.state["state"]
        out = np.empty(R_RNG_WORDS, dtype=np.uint32)
        out[0] = st["pos"]
        out[1:] = st["key"]
        return out

    @state.setter
    def state(self, words: np.ndarray) -> None:
        words = np.asarray(words, dtype=np.uint32)
        assert words.shape == (R_RNG_WORDS,)
        self._bg.state = {"bit_generator": "MT19937",
                          "state": {"key": words[1:].copy(), "pos": int(words[0])}}

    def set_seed(self, seed: int) -> None:
        self.state = seed_state(seed)

    def runif(self, n: int) -> np.ndarray:
        raw = self._bg.random_raw(int(n)).astype(np.float64)
        v = raw * 2.3283064365386963e-10
        v = np.where(v <= 0.0, 0.5 * _I2_32M1, v)
        v = np.where(1.0 - v <= 0.0, 1.0 - 0.5 * _I2_32M1, v)
        return v

    def rnorm(self, n: int, mean: float = 0.0, sd: float = 1.0) -> np.ndarray:
        u = self.runif(2 * int(n)).reshape(-1, 2)
        big = 134217728.0
        p = (np.floor(big * u[:, 0]) + u[:, 1]) / big
        return mean + sd * qnorm(p)

    def unif_index(self, dn: int) -> int:
        """R_unif_index(dn) with sample.kind = "Rejection"."""
        if dn <= 0:
            return 0
        bits = int(math.ceil(math.log2(dn)))
        while True:
            v = 0
            for _ in range(0, bits + 1, 16):
                v1 = int(math.floor(float(self.runif(1)[0]) * 65536))
                v = 65536 * v + v1
            if bits < 64:
                v &= (1 << bits) - 1
            if v < dn:
                return v

    def sample_int(self, n: int, size: int = 1) -> np.ndarray:
        """``sample.int(n, size, replace = TRUE)`` (1-based, as R)."""
        return np.array([self.unif_index(n) + 1 for _ in range(size)], dtype=np.int64)

    def rbinom1(self, n: int, prob: float) -> np.ndarray:
        """``rbinom(n, 1, prob)`` for size = 1: R's inversion branch draws one uniform per value.

        (Synthetic-data convenience; exact R stream compatibility of rbinom is not required by any
        test — the generated data only has to be self-consistent.)
        """
        return (self.runif(n) < prob).astype(np.float64)


def qnorm(p: np.ndarray) -> np.ndarray:
    """Vectorised qnorm5(p, 0, 1, TRUE, FALSE): Wichura AS241 (R src/nmath/qnorm.c)."""
    p = np.asarray(p, dtype=np.float64)
    q = p - 0.5
    out = np.empty_like(p)
    cen = np.abs(q) <= 0.425
    r = 0.180625 - q[cen] * q[cen]
    num = (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r
               + 45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r
            + 133.14166789178437745) * r + 3.387132872796366608)
    den = (((((((r * 5226.495278852545925 + 28729.085735721942674) * r + 39307.89580009271061) * r
               + 21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r
            + 42.313330701600911252) * r + 1.0)
    out[cen] = q[cen] * num / den
    tail = ~cen
    if tail.any():
        pt, qt = p[tail], q[tail]
        r = np.sqrt(-np.log(np.where(qt < 0, pt, 1.0 - pt)))
        val = np.empty_like(r)
        a = r <= 5.0
        ra = r[a] - 1.6
        val[a] = ((((((((ra * 7.7454501427834140764e-4 + .0227238449892691845833) * ra + .24178072517745061177) * ra
                       + 1.27045825245236838258) * ra + 3.64784832476320460504) * ra + 5.7694972214606914055) * ra
                    + 4.6303378461565452959) * ra + 1.42343711074968357734)
                  / (((((((ra * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * ra + .0151986665636164571966) * ra
                         + .14810397642748007459) * ra + .68976733498510000455) * ra + 1.6763848301838038494) * ra
                      + 2.05319162663775882187) * ra + 1.0))
        rb = r[~a] - 5.0
        val[~a] = ((((((((rb * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * rb + .0012426609473880784386) * rb
                        + .026532189526576123093) * rb + .29656057182850489123) * rb + 1.7848265399172913358) * rb
                     + 5.4637849111641143699) * rb + 6.6579046435011037772)
                   / (((((((rb * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * rb + 1.8463183175100546818e-5) * rb
                          + 7.868691311456132591e-4) * rb + .0148753612908506148525) * rb + .13692988092273580531) * rb
                       + .59983220655588793769) * rb + 1.0))
        out[tail] = np.where(qt < 0, -val, val)
    return out
