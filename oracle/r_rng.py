"""R's default random number stream, restated (TEST INFRASTRUCTURE -- only tests/ may import this).

The reference's only end-to-end golden numbers (`examples/numeric_convergence.md:9-48`: the six
average marginal effects of a bigKRLS fit, printed to 7 significant figures) are produced from
`set.seed(2018); rnorm(); runif(); rnorm()` in R 3.4.2.  R is a third-party dependency that is
absent from /root/reference and from this image, so its published algorithms are restated here:

  * `set.seed(seed)`  -- R `src/main/RNG.c` (`RNG_Init`, `FixupSeeds`): the integer seed is
    scrambled by 50 rounds of the LCG  s <- 69069 s + 1 (mod 2^32), then 625 further rounds fill
    the state vector; for the default generator (Mersenne-Twister) word 0 is the position `mti`
    and is reset to 624 (`FixupSeeds`), words 1..624 are the MT19937 state.
  * `unif_rand()`     -- `MT_genrand` (Matsumoto & Nishimura's MT19937 `genrand`, 32-bit output
    times 2.3283064365386963e-10) followed by `fixup` into the open interval (0,1).
  * `norm_rand()`     -- default `INVERSION` branch of `src/nmath/snorm.c`:
    u = unif_rand(); u = (int)(2^27 u) + unif_rand(); qnorm(u / 2^27).
    R's `qnorm` is Wichura's AS 241 (PPND16, relative error ~1e-16); `scipy.special.ndtri` is used
    here as the inverse normal CDF. The two agree to a few ulp, far below the 7 significant
    figures the golden vector carries (the uniform stream itself is reproduced exactly).

Checked against R's well-known outputs in tests/test_oracle.py (set.seed(1); runif(3);
set.seed(1); rnorm(3); set.seed(123); runif(1); set.seed(42); runif(1)).
"""
import numpy as np
from scipy.special import ndtri

_N, _M = 624, 397
_MATRIX_A = 0x9908B0DF
_UPPER, _LOWER = 0x80000000, 0x7FFFFFFF
_I2_32M1 = 2.328306437080797e-10     # 1 / (2^32 - 1): R's fixup() bound
_BIG = 134217728                     # 2^27


class RStream:
    """`set.seed(seed)` followed by calls of runif / rnorm, default kinds (R >= 1.7, < 3.6 identical
    for these two functions; `sample()` changed in 3.6 and is not restated)."""

    def __init__(self, seed: int):
        s = int(seed) & 0xFFFFFFFF
        for _ in range(50):
            s = (69069 * s + 1) & 0xFFFFFFFF
        words = []
        for _ in range(_N + 1):
            s = (69069 * s + 1) & 0xFFFFFFFF
            words.append(s)
        self.mt = words[1:]          # words[0] is `mti`, which FixupSeeds resets to N
        self.mti = _N

    def _genrand(self) -> int:
        mt = self.mt
        if self.mti >= _N:
            for kk in range(_N - _M):
                y = (mt[kk] & _UPPER) | (mt[kk + 1] & _LOWER)
                mt[kk] = mt[kk + _M] ^ (y >> 1) ^ (_MATRIX_A if (y & 1) else 0)
            for kk in range(_N - _M, _N - 1):
                y = (mt[kk] & _UPPER) | (mt[kk + 1] & _LOWER)
                mt[kk] = mt[kk + (_M - _N)] ^ (y >> 1) ^ (_MATRIX_A if (y & 1) else 0)
            y = (mt[_N - 1] & _UPPER) | (mt[0] & _LOWER)
            mt[_N - 1] = mt[_M - 1] ^ (y >> 1) ^ (_MATRIX_A if (y & 1) else 0)
            self.mti = 0
        y = mt[self.mti]
        self.mti += 1
        y ^= (y >> 11)
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= (y >> 18)
        return y & 0xFFFFFFFF

    def unif_rand(self) -> float:
        x = self._genrand() * 2.3283064365386963e-10      # [0,1)
        if x <= 0.0:
            return 0.5 * _I2_32M1
        if 1.0 - x <= 0.0:
            return 1.0 - 0.5 * _I2_32M1
        return x

    def norm_rand(self) -> float:
        u = self.unif_rand()
        u = int(_BIG * u) + self.unif_rand()
        return float(ndtri(u / _BIG))

    def runif(self, n: int) -> np.ndarray:
        return np.array([self.unif_rand() for _ in range(n)])

    def rnorm(self, n: int) -> np.ndarray:
        return np.array([self.norm_rand() for _ in range(n)])


def numeric_convergence_inputs():
    """The data of `examples/numeric_convergence.md:9-15`:
        set.seed(2018); N <- 500; P <- 6
        X <- matrix(rnorm(N*P), ncol=P); X[,P] <- ifelse(X[,P] > 0.12345, 1, 0)
        b <- runif(ncol(X)); y <- X %*% b + rnorm(nrow(X))"""
    r = RStream(2018)
    n, p = 500, 6
    X = r.rnorm(n * p).reshape((n, p), order="F")      # matrix() fills column by column
    X[:, p - 1] = np.where(X[:, p - 1] > 0.12345, 1.0, 0.0)
    b = r.runif(p)
    y = X @ b + r.rnorm(n)
    return X, y


# `bigKRLS.out$avgderivatives`, examples/numeric_convergence.md:40-46 (7 significant figures)
NUMERIC_CONVERGENCE_AVGDERIV = np.array([0.2286663, 0.1150259, 0.006574909, 0.09488611, 0.3828897, 0.7653918])
