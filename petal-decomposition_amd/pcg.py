"""``rand_pcg::Mcg128Xsl64`` (the crate's ``Pcg``, src/pca.rs:13 / src/ica.rs:13) with ``rand_distr::StandardNormal``
(256-layer Ziggurat), restated from the published algorithms -- the third-party crates are not vendored in the
reference, so the STREAM is unpinned except through the crate's ``n_iter == 1`` test (src/ica.rs:412-417), which the
C++ twin of this class (include/petal_decomposition.hpp) reproduces.  Pure Python: meant for drawing Omega / w_init
(d x (k + 10), nc x nc) and for model (de)serialisation, not for bulk sampling."""
import math

import numpy as np

_MASK128 = (1 << 128) - 1
_MASK64 = (1 << 64) - 1
_MUL = (0x2360ED051FC65DA4 << 64) | 0x4385DF649FCCF645
_R = 3.654152885361009
_V = 0.00492867323399


def _tables():
    x = [0.0] * 257
    x[0] = _V / math.exp(-0.5 * _R * _R)
    x[1] = _R
    for i in range(2, 256):
        x[i] = math.sqrt(-2.0 * math.log(_V / x[i - 1] + math.exp(-0.5 * x[i - 1] * x[i - 1])))
    x[256] = 0.0
    return x, [math.exp(-0.5 * v * v) for v in x]


_X, _F = _tables()


class Pcg:
    """``Mcg128Xsl64``: 128-bit multiplicative congruential state (always odd), XSL-RR output."""

    def __init__(self, state: int):
        self.state = (int(state) | 1) & _MASK128

    @classmethod
    def from_seed_be_bytes(cls, seed: int) -> "Pcg":
        """``Pcg::from_seed(seed.to_be_bytes())`` (src/pca.rs:357, src/ica.rs:76): from_seed reads the 16 bytes as a
        little-endian u128, so the state is the byte-swapped seed."""
        return cls(int.from_bytes(int(seed & _MASK128).to_bytes(16, "big"), "little"))

    def next_u64(self) -> int:
        self.state = (self.state * _MUL) & _MASK128
        rot = self.state >> 122
        x = ((self.state >> 64) ^ self.state) & _MASK64
        return ((x >> rot) | (x << ((64 - rot) & 63))) & _MASK64

    def _f64(self) -> float:  # rand Standard: [0, 1)
        return (self.next_u64() >> 11) * (1.0 / 9007199254740992.0)

    def _open01(self) -> float:
        return (self.next_u64() >> 12) * (1.0 / 4503599627370496.0) + (1.0 / 9007199254740992.0)

    def sample_standard_normal(self) -> float:
        while True:
            bits = self.next_u64()
            i = bits & 0xFF
            u = (bits >> 12) * (2.0 / 4503599627370496.0) - 1.0
            x = u * _X[i]
            if abs(x) < _X[i + 1]:
                return x
            if i == 0:  # tail beyond R
                xx, yy = 1.0, 0.0
                while -2.0 * yy < xx * xx:
                    xx = math.log(self._open01()) / _R
                    yy = math.log(self._open01())
                return xx - _R if u < 0 else _R - xx
            if _F[i + 1] + (_F[i] - _F[i + 1]) * self._f64() < math.exp(-0.5 * x * x):
                return x

    def standard_normal(self, shape) -> np.ndarray:
        """Row-major fill, one f64 draw per entry: the order of ``Array2::from_shape_fn`` in src/pca.rs:701-705."""
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        n = int(np.prod(shape)) if shape else 1
        return np.array([self.sample_standard_normal() for _ in range(n)], dtype=np.float64).reshape(shape)

    # serde of rand_pcg::Mcg128Xsl64 (feature "serde"): a struct with the single field `state`
    def to_serde(self) -> dict:
        return {"state": self.state}

    @classmethod
    def from_serde(cls, obj: dict) -> "Pcg":
        return cls(int(obj["state"]))
