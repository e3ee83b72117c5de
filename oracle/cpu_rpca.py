"""Builds and binds oracle/cpu_rpca.cpp (TEST INFRASTRUCTURE): the dependency-free threaded C++ restatement of the
reference's RandomizedPca fit.  Only tests/ and bench.py's cpu_baseline leg may use it."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "cpu_rpca.cpp")
OUT = os.path.join(ROOT, "tests", "_build", "libcpu_rpca.so")
_lib = None


def build() -> str:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        tmp = OUT + f".{os.getpid()}.tmp"
        subprocess.check_call(["g++", "-O3", "-mavx2", "-mfma", "-fopenmp", "-std=c++17", "-fPIC", "-shared", "-o", tmp, SRC])
        os.replace(tmp, OUT)
    return OUT


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        for name in ("oracle_rpca_fit_f32", "oracle_rpca_fit_f64"):
            fn = getattr(_lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_rpca_max_threads.restype = C.c_int
    return _lib


def max_threads() -> int:
    return int(_load().oracle_rpca_max_threads())


class RandomizedPcaCpp:
    """RandomizedPca<A, R> (src/pca.rs:317-551) in the data's precision (float32 / float64), Omega passed in (f64 draw)."""

    def __init__(self, n_components, centering=True, n_oversample=10, n_iter=7, threads=0):
        self.k, self.centering, self.n_oversample, self.n_iter, self.threads = n_components, centering, n_oversample, n_iter, threads

    def fit(self, x, omega):
        x = np.ascontiguousarray(x)
        assert x.dtype in (np.float32, np.float64) and x.ndim == 2
        n, d = x.shape
        omega = np.ascontiguousarray(omega, dtype=np.float64)
        assert omega.shape == (d, self.k + self.n_oversample)
        self.components = np.zeros((self.k, d))
        self.singular = np.zeros(self.k)
        self.means = np.zeros(d)
        tv = np.zeros(1)
        fn = _load().oracle_rpca_fit_f32 if x.dtype == np.float32 else _load().oracle_rpca_fit_f64
        rc = fn(x.ctypes.data, n, d, self.k, self.n_oversample, self.n_iter, int(self.centering), omega.ctypes.data, int(self.threads),
                self.components.ctypes.data, self.singular.ctypes.data, self.means.ctypes.data, tv.ctypes.data)
        if rc != 0:
            raise ValueError(f"every dimension should be at least {self.k}")
        self.total_variance = float(tv[0])
        return self

    def explained_variance_ratio(self):
        return self.singular ** 2 / self.total_variance
