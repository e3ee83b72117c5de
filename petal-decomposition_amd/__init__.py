"""petal-decomposition_amd -- host-side mirror of the reference crate's public interface
(``Pca``/``PcaBuilder``/``RandomizedPca``/``RandomizedPcaBuilder``/``FastIca``/``FastIcaBuilder``,
``src/lib.rs:17-18``) over the C ABI of ``include/petal_hip.h``.

The arithmetic lives in ``libpetal_hip.so`` (hand-written HIP for gfx950, ``csrc/``).  This module
is plumbing only: it describes caller arrays (numpy on the host, torch / ``__cuda_array_interface__``
on the device) as ``petal_matrix`` and forwards.  There is no CPU fallback: if the library is
missing or no MI355X is visible, constructing a context raises.

The directory name carries a hyphen (it is the name the build contract asks for); import it through
``petal_decomposition_amd`` at the repo root.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.path.join(_HERE, "libpetal_hip.so")

PETAL_OK, PETAL_INVALID_INPUT, PETAL_LINALG_ERROR, PETAL_DEVICE_ERROR = 0, 1, 2, 3
PETAL_F32, PETAL_F64 = 0, 1
PETAL_HOST, PETAL_DEVICE = 0, 1
PETAL_SUM, PETAL_MAX, PETAL_MIN = 0, 1, 2
GEMM_SPLIT_BF16X3, GEMM_FP32_MFMA, GEMM_SPLIT_BF16X3_EXACT = 0, 1, 2
# petal_ctx_set_option (petal_hip.h PETAL_OPT_*): name -> option number
OPTIONS = {"two_plane_operands": 0, "two_plane_omega": 1, "two_plane_iterate": 2, "steering_passes": 3, "fused_pass": 4,
           "fused_pass_min_rows": 5, "verdict_threshold": 6, "means_fold_rows": 7, "gram_split": 8, "gram_split_hook": 9,
           "d2h_kernel": 10, "row_pad": 11, "eigh_jacobi": 12, "poison": 13, "force_collective": 14}
ICA_TEXTBOOK, ICA_REFERENCE_LITERAL = 0, 1


from .pcg import Pcg  # noqa: E402  (rand_pcg::Mcg128Xsl64 + Ziggurat StandardNormal)


class DecompositionError(Exception):
    """``DecompositionError`` (src/lib.rs:22-28)."""


class InvalidInput(DecompositionError):
    def __str__(self):  # "invalid matrix: {0}" (src/lib.rs:24)
        return "invalid matrix: " + super().__str__()


class LinalgError(DecompositionError):
    def __str__(self):  # sic, src/lib.rs:26
        return "linear algerba operation failed: " + super().__str__()


class DeviceError(DecompositionError):
    pass


class petal_matrix(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int64),
                ("row_stride", C.c_int64), ("col_stride", C.c_int64),
                ("dtype", C.c_int32), ("space", C.c_int32)]


class petal_stats(C.Structure):
    _fields_ = [("fit_ms", C.c_double),
                ("xp_ms", C.c_double), ("xp_launches", C.c_int64),
                ("atb_ms", C.c_double), ("atb_launches", C.c_int64),
                ("pass_flops", C.c_double), ("pass_bytes", C.c_double),
                ("ica_step_ms", C.c_double), ("ica_step_launches", C.c_int64),
                ("ica_step_flops", C.c_double), ("ica_step_bytes", C.c_double),
                ("n_iter", C.c_int64),
                ("allreduce_calls", C.c_int64), ("allreduce_bytes", C.c_double),
                ("allreduce_ms", C.c_double), ("allreduce_timed", C.c_int64),
                ("x_row_pitch_bytes", C.c_int64), ("x_zero_copy", C.c_int64),
                ("rpca_redo", C.c_int64), ("pow_ms", C.c_double), ("pow_launches", C.c_int64),
                ("stream_ms", C.c_double), ("stream_launches", C.c_int64), ("ica_redo", C.c_int64), ("ica_gram_split", C.c_int64),
                ("means_folded", C.c_int64), ("eigh_redo", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p)

# every symbol include/petal_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_M = C.POINTER(petal_matrix)
ABI = [
    ("petal_ctx_create", C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    ("petal_ctx_destroy", None, [_P]),
    ("petal_last_error", C.c_char_p, [_P]),
    ("petal_version", C.c_char_p, []),
    ("petal_ctx_set_collective", C.c_int, [_P, ALLREDUCE_FN, _P, C.c_int, C.c_int]),
    ("petal_ctx_set_profiling", C.c_int, [_P, C.c_int]),
    ("petal_ctx_set_gemm_mode", C.c_int, [_P, C.c_int]),
    ("petal_ctx_set_option", C.c_int, [_P, C.c_int, C.c_double]),
    ("petal_ctx_get_option", C.c_int, [_P, C.c_int, C.POINTER(C.c_double)]),
    ("petal_rccl_unique_id", C.c_int, [_P]),
    ("petal_ctx_init_rccl", C.c_int, [_P, _P, C.c_int, C.c_int]),
    ("petal_get_stats", C.c_int, [_P, C.POINTER(petal_stats)]),
    ("petal_ctx_collective_info", C.c_int, [_P] + [C.POINTER(C.c_int)] * 6),
    ("petal_power_pass", C.c_int, [_P, _M, _P, _P, C.c_int64, C.POINTER(C.c_double), _M, C.POINTER(C.c_int)]),
    ("petal_pca_fit", C.c_int, [_P, _M, C.c_int64, C.c_int, _P, _P, _P, _P, _M]),
    ("petal_rpca_fit", C.c_int, [_P, _M, C.c_int64, C.c_int64, C.c_int64, C.c_int, _P, _P, _P, _P, _P, _M]),
    ("petal_transform", C.c_int, [_P, _M, _P, _P, C.c_int64, C.c_int64, C.c_int, _M]),
    ("petal_inverse_transform", C.c_int, [_P, _M, _P, _P, C.c_int64, C.c_int64, C.c_int, _M]),
    ("petal_fastica_fit", C.c_int, [_P, _M, C.c_int64, C.c_double, C.c_int64, C.c_int, _P, _P, _P, C.POINTER(C.c_int64), _M]),
    ("petal_ica_par", C.c_int, [_P, _M, C.c_double, C.c_int64, C.c_int, _P, _P, C.POINTER(C.c_int64)]),
    ("petal_symmetric_decorrelation", C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int, _P]),
    ("petal_logcosh", C.c_int, [_P, _M, _M, _P]),
    ("petal_svd_flip", C.c_int, [_P, _M, _M]),
    ("petal_gemm_xp", C.c_int, [_P, _M, _P, _P, C.c_int64, _P, _M]),
    ("petal_gemm_atb", C.c_int, [_P, _M, _P, _M, _P, C.POINTER(C.c_double)]),
]


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 and load them by the unversioned
    file name, so a process that first loads the system HIP runtime (through this library) and then imports
    torch ends up with TWO HSA runtimes and torch sees no GPU.  Importing torch first makes the dynamic
    loader resolve this library's ``libamdhip64.so.7`` to the copy torch already loaded: one runtime, shared
    streams and device pointers.  Skipped when torch is absent or PETAL_NO_TORCH=1."""
    import sys
    if "torch" in sys.modules or os.environ.get("PETAL_NO_TORCH") == "1":
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def load_library(path: Optional[str] = None, preload_torch: bool = True) -> C.CDLL:
    """dlopen a library implementing include/petal_hip.h and type its entry points."""
    path = path or os.environ.get("PETAL_HIP_LIBRARY") or DEFAULT_LIBRARY
    if not os.path.exists(path):
        raise RuntimeError(
            f"petal-decomposition_amd: native library not found at {path}. Build it with "
            f"`python __graft_entry__.py build` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    if preload_torch:
        _preload_torch_hip_runtime()
    lib = C.CDLL(path)
    for name, res, args in ABI:
        fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    lib._petal_path = path
    return lib


_default_lib = None


def default_library() -> C.CDLL:
    global _default_lib
    if _default_lib is None:
        _default_lib = load_library()
    return _default_lib


# ------------------------------------------------------------------------------------------------
def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _np_dtype(code):
    return np.float32 if code == PETAL_F32 else np.float64


def describe(x, keep: list) -> petal_matrix:
    """ndarray-style (ptr, shape, strides) descriptor of a 2-D numpy array / torch tensor."""
    if _is_torch(x):
        import torch
        if x.dim() != 2:
            raise InvalidInput("expected a 2-D array")
        if x.dtype == torch.float32:
            dt = PETAL_F32
        elif x.dtype == torch.float64:
            dt = PETAL_F64
        else:
            raise InvalidInput(f"unsupported dtype {x.dtype}")
        keep.append(x)
        space = PETAL_DEVICE if x.is_cuda else PETAL_HOST
        if x.is_cuda:
            # the ctx launches on its own stream: whatever torch still has queued for this tensor on ITS current stream
            # must be finished first (the calls are synchronous anyway; a no-op when the ctx shares torch's stream)
            torch.cuda.current_stream(x.device).synchronize()
        return petal_matrix(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), x.stride(1), dt, space)
    if hasattr(x, "__cuda_array_interface__") and not isinstance(x, np.ndarray):
        cai = x.__cuda_array_interface__
        shape, typestr = cai["shape"], cai["typestr"]
        if len(shape) != 2:
            raise InvalidInput("expected a 2-D array")
        dt = {"<f4": PETAL_F32, "<f8": PETAL_F64}[typestr]
        isz = 4 if dt == PETAL_F32 else 8
        strides = cai.get("strides") or (shape[1] * isz, isz)
        keep.append(x)
        return petal_matrix(cai["data"][0], shape[0], shape[1], strides[0] // isz, strides[1] // isz, dt, PETAL_DEVICE)
    a = np.asarray(x)
    if a.ndim != 2:
        raise InvalidInput("expected a 2-D array")
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    keep.append(a)
    dt = PETAL_F32 if a.dtype == np.float32 else PETAL_F64
    isz = a.dtype.itemsize
    return petal_matrix(a.ctypes.data, a.shape[0], a.shape[1], a.strides[0] // isz, a.strides[1] // isz, dt, PETAL_HOST)


def _alloc_like(x, rows, cols, dtype_code):
    """Output array in the memory space of x (torch device tensor or numpy)."""
    if _is_torch(x) and x.is_cuda:
        import torch
        return torch.empty((rows, cols), dtype=x.dtype, device=x.device)
    return np.empty((rows, cols), dtype=_np_dtype(dtype_code))


def _host(a, dtype_code, shape=None):
    out = np.ascontiguousarray(np.asarray(a, dtype=_np_dtype(dtype_code)))
    if shape is not None and out.shape != tuple(shape):
        raise InvalidInput(f"expected shape {tuple(shape)}, got {out.shape}")
    return out


class Context:
    """One GPU, one stream, one caching workspace (``petal_ctx``)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None, lib: Optional[C.CDLL] = None):
        self.lib = lib or default_library()
        self._h = C.c_void_p()
        self._cb = None
        rc = self.lib.petal_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self._h))
        if rc != PETAL_OK or not self._h:
            raise DeviceError(f"petal_ctx_create(device={device}) failed with code {rc}: no usable gfx950 device? "
                              f"(library {getattr(self.lib, '_petal_path', '?')})")
        self.rank, self.world_size = 0, 1

    def close(self):
        if getattr(self, "_h", None):
            self.lib.petal_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        if rc == PETAL_OK:
            return
        msg = (self.lib.petal_last_error(self._h) or b"").decode()
        raise {PETAL_INVALID_INPUT: InvalidInput, PETAL_LINALG_ERROR: LinalgError}.get(rc, DeviceError)(msg)

    def set_profiling(self, level):
        """0/False off, 1/True one sampled launch of each hot kernel per fit, 2 every launch (petal_hip.h)."""
        self.check(self.lib.petal_ctx_set_profiling(self._h, int(level)))

    def set_gemm_mode(self, mode):
        """GEMM_SPLIT_BF16X3 (default), GEMM_FP32_MFMA or GEMM_SPLIT_BF16X3_EXACT; also accepts "bf16x3" / "fp32" / "bf16x3-exact"
        (petal_hip.h)."""
        if isinstance(mode, str):
            mode = {"bf16x3": GEMM_SPLIT_BF16X3, "fp32": GEMM_FP32_MFMA, "bf16x3-exact": GEMM_SPLIT_BF16X3_EXACT}[mode]
        self.check(self.lib.petal_ctx_set_gemm_mode(self._h, int(mode)))

    def set_option(self, option, value) -> None:
        """petal_ctx_set_option: `option` a PETAL_OPT_* number or its name in OPTIONS ("steering_passes", "means_fold_rows", ...).
        The defaults were read from the environment once, when the ctx was created; nothing in the library reads it afterwards."""
        opt = OPTIONS[option] if isinstance(option, str) else int(option)
        self.check(self.lib.petal_ctx_set_option(self._h, opt, float(value)))

    def get_option(self, option) -> float:
        opt = OPTIONS[option] if isinstance(option, str) else int(option)
        v = C.c_double(0.0)
        self.check(self.lib.petal_ctx_get_option(self._h, opt, C.byref(v)))
        return v.value

    def collective_info(self) -> dict:
        """kind ("none" / "hook" / "rccl"), rank / world_size as the ctx was told them, and -- for the built-in communicator -- what
        RCCL itself reports: ncclCommCount, ncclCommCuDevice, ncclCommUserRank (-1 where unavailable)."""
        v = [C.c_int(-1) for _ in range(6)]
        self.check(self.lib.petal_ctx_collective_info(self._h, *[C.byref(x) for x in v]))
        return {"kind": {0: "none", 1: "hook", 2: "rccl"}.get(v[0].value, "?"), "rank": v[1].value, "world_size": v[2].value,
                "ncclCommCount": v[3].value, "ncclCommCuDevice": v[4].value, "ncclCommUserRank": v[5].value}

    def stats(self) -> dict:
        s = petal_stats()
        self.lib.petal_get_stats(self._h, C.byref(s))
        return s.as_dict()

    def set_collective(self, fn, rank: int, world_size: int):
        """fn(ptr:int, count:int, dtype:int, op:int, stream:int) -> int (0 = ok)."""
        def tramp(_user, buf, count, dtype, op, stream):
            try:
                return int(fn(int(buf or 0), int(count), int(dtype), int(op), int(stream or 0)) or 0)
            except Exception as e:  # never unwind through C
                import sys
                print(f"petal all-reduce hook failed: {e!r}", file=sys.stderr)
                return 1
        self._cb = ALLREDUCE_FN(tramp)
        self.check(self.lib.petal_ctx_set_collective(self._h, self._cb, None, int(rank), int(world_size)))
        self.rank, self.world_size = rank, world_size

    @staticmethod
    def torch_allreduce_hook(group=None, host_buffers=False):
        """The all-reduce hook as a Python callable ``hook(ptr, count, dtype, op, stream) -> 0``: wraps the raw
        buffer zero-copy as a torch tensor and calls ``torch.distributed.all_reduce`` on ``group``.  Backend "nccl" ==
        RCCL over xGMI on ROCm, enqueued on the ctx stream.  Backend "gloo" with DEVICE buffers (several ranks sharing one
        GPU, where RCCL refuses to form a communicator; or a node without xGMI): the buffer is staged through the host in
        stream order -- D2H on the ctx stream, gloo all-reduce, H2D on the ctx stream.  ``host_buffers=True`` is the
        host-memory simulation of the CPU tests, whose "device" pointers are plain host pointers."""
        import torch
        import torch.distributed as dist
        ops = {PETAL_SUM: dist.ReduceOp.SUM, PETAL_MAX: dist.ReduceOp.MAX, PETAL_MIN: dist.ReduceOp.MIN}
        backend = dist.get_backend(group)

        class _Cai:  # zero-copy view of a device buffer for torch.as_tensor
            def __init__(self, ptr, count, typestr):
                self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 3}

        def hook(ptr, count, dtype, op, stream):
            npdt = _np_dtype(dtype)
            if host_buffers:
                ctype = C.c_float if dtype == PETAL_F32 else C.c_double
                view = np.ctypeslib.as_array((ctype * count).from_address(ptr))
                t = torch.from_numpy(view)
                dist.all_reduce(t, op=ops[op], group=group)
                return 0
            t = torch.as_tensor(_Cai(ptr, count, np.dtype(npdt).str), device="cuda")
            if backend == "gloo":
                ext = torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream()
                with torch.cuda.stream(ext):
                    h = t.cpu()                      # ordered behind the kernels queued on the ctx stream; blocks the host
                    dist.all_reduce(h, op=ops[op], group=group)
                    t.copy_(h)                       # pageable source: returns once staged, ordered on the ctx stream
                return 0
            if stream:
                with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                    dist.all_reduce(t, op=ops[op], group=group)
            else:
                dist.all_reduce(t, op=ops[op], group=group)
            return 0

        return hook

    def use_rccl(self, group=None):
        """Sample-sharded multi-GPU with the library's BUILT-IN collective: ncclAllReduce issued by the library on the ctx
        stream (petal_ctx_init_rccl).  torch.distributed is only used once, to hand rank 0's ncclUniqueId to the others.
        Collective: every rank of ``group`` must call it."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        uid = C.create_string_buffer(128)
        if rank == 0:
            rc = self.lib.petal_rccl_unique_id(uid)
            if rc != 0:
                raise DeviceError("petal_rccl_unique_id failed: RCCL could not be loaded")
        box = [uid.raw if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        buf = C.create_string_buffer(box[0], 128)
        self.check(self.lib.petal_ctx_init_rccl(self._h, buf, rank, world))
        self.rank, self.world_size = rank, world

    def use_torch_distributed(self, group=None):
        """Sample-sharded multi-GPU: sum the small replicated buffers with torch.distributed."""
        import torch.distributed as dist
        host_buffers = bool(getattr(self.lib, "_petal_host_buffers", False))
        self.set_collective(self.torch_allreduce_hook(group, host_buffers), dist.get_rank(group), dist.get_world_size(group))


_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx



# ---- serde interchange (the crate's `serde` feature, Cargo.toml:41-47): models as the JSON serde_json writes for the
# reference structs -- same field names and order (src/pca.rs:41-51, 317-329; src/ica.rs:41-50), ndarray's
# {"v": 1, "dim": [...], "data": [...]} arrays, rand_pcg's {"state": u128} generator.
def _nd_to_serde(a) -> dict:
    a = np.asarray(a)
    return {"v": 1, "dim": list(a.shape), "data": [float(v) for v in a.reshape(-1)]}


def _nd_from_serde(obj: dict, dtype) -> np.ndarray:
    if obj.get("v") != 1:
        raise InvalidInput("unknown ndarray serde version")
    return np.asarray(obj["data"], dtype=dtype).reshape([int(v) for v in obj["dim"]])


def _rng_to_serde(rng):
    if isinstance(rng, Pcg):
        return rng.to_serde()
    raise InvalidInput("only the crate's Pcg generator has a serde form: build the model with seed()/with_seed() or a Pcg")


# ------------------------------------------------------------------------------------------------
class _PcaModel:
    """State shared by Pca and RandomizedPca (src/pca.rs:41-51, 317-329)."""

    def __init__(self, n_components: int, centering: bool = True, ctx: Optional[Context] = None):
        self._k = int(n_components)
        self.centering = bool(centering)
        self.ctx = ctx
        self._components = np.zeros((self._k, 0))
        self._means = np.zeros(0)
        self._singular = np.zeros(0)
        self._total_variance = 0.0
        self.n_samples = 0
        self._dt = PETAL_F64

    def _ctx(self) -> Context:
        if self.ctx is None:
            self.ctx = default_context()
        return self.ctx

    # accessors (src/pca.rs:78-105, 392-419)
    def components(self):
        return self._components

    def mean(self):
        return self._means

    def n_components(self):
        return self._k

    def singular_values(self):
        return self._singular

    def explained_variance_ratio(self):
        return self._singular * self._singular / self._total_variance

    def _serde_fields(self) -> dict:
        dt = self._components.dtype if self._components.size or self._means.size else np.dtype(_np_dtype(self._dt))
        return {"components": _nd_to_serde(self._components), "n_samples": int(self.n_samples),
                "means": _nd_to_serde(self._means), "total_variance": float(np.asarray(self._total_variance, dtype=dt)),
                "singular": _nd_to_serde(self._singular), "centering": bool(self.centering)}

    def _load_serde_fields(self, obj: dict, dtype):
        self._components = _nd_from_serde(obj["components"], dtype)
        self.n_samples = int(obj["n_samples"])
        self._means = _nd_from_serde(obj["means"], dtype)
        self._total_variance = np.asarray(obj["total_variance"], dtype=dtype)[()]
        self._singular = _nd_from_serde(obj["singular"], dtype)
        self.centering = bool(obj["centering"])
        self._k = int(self._components.shape[0])
        self._dt = PETAL_F32 if np.dtype(dtype) == np.float32 else PETAL_F64

    def to_json(self) -> str:
        """The JSON ``serde_json::to_string(&model)`` writes for the reference struct (src/pca.rs:941, 1035)."""
        import json
        return json.dumps(self._serde_fields())

    def _store(self, comp, means, sing, tv, n):
        self._components, self._means, self._singular = comp, means, sing
        self._total_variance = tv[0]
        self.n_samples = n

    def transform(self, x):
        """src/pca.rs:130-135 / 444-449."""
        keep = []
        mx = describe(x, keep)
        ctx = self._ctx()
        d = self._means.shape[0]
        if mx.cols != d:
            raise InvalidInput(f"# of columns should be {d}")
        y = _alloc_like(x, mx.rows, self._k, mx.dtype)
        my = describe(y, keep)
        comp = _host(self._components, mx.dtype)
        mu = _host(self._means, mx.dtype)
        ctx.check(ctx.lib.petal_transform(ctx._h, C.byref(mx), comp.ctypes.data, mu.ctypes.data, self._k, d,
                                          int(self.centering), C.byref(my)))
        return y

    def inverse_transform(self, y):
        """src/pca.rs:176-184 / 490-498."""
        keep = []
        my = describe(y, keep)
        ctx = self._ctx()
        d = self._means.shape[0]
        if my.cols != self._k:
            raise InvalidInput(f"# of columns should be {self._k}")
        x = _alloc_like(y, my.rows, d, my.dtype)
        mx = describe(x, keep)
        comp = _host(self._components, my.dtype)
        mu = _host(self._means, my.dtype)
        ctx.check(ctx.lib.petal_inverse_transform(ctx._h, C.byref(my), comp.ctypes.data, mu.ctypes.data, self._k, d,
                                                  int(self.centering), C.byref(mx)))
        return x


class Pca(_PcaModel):
    """``Pca<A>`` (src/pca.rs:41-232)."""

    @classmethod
    def new(cls, n_components: int, ctx: Optional[Context] = None):
        return cls(n_components, True, ctx)

    @classmethod
    def from_json(cls, text: str, dtype=np.float32, ctx: Optional[Context] = None):
        """``serde_json::from_str::<Pca<A>>`` (A = ``dtype``)."""
        import json
        obj = json.loads(text)
        m = cls(int(obj["components"]["dim"][0]), ctx=ctx)
        m._load_serde_fields(obj, dtype)
        return m

    def _inner_fit(self, x, want_y: bool):
        keep = []
        mx = describe(x, keep)
        ctx = self._ctx()
        npdt = _np_dtype(mx.dtype)
        comp = np.zeros((self._k, mx.cols), dtype=npdt)
        means = np.zeros(mx.cols, dtype=npdt)
        sing = np.zeros(self._k, dtype=npdt)
        tv = np.zeros(1, dtype=npdt)
        y, my = None, None
        if want_y:
            y = _alloc_like(x, mx.rows, self._k, mx.dtype)
            my = describe(y, keep)
        ctx.check(ctx.lib.petal_pca_fit(ctx._h, C.byref(mx), self._k, int(self.centering), comp.ctypes.data,
                                        means.ctypes.data, sing.ctypes.data, tv.ctypes.data,
                                        C.byref(my) if my is not None else None))
        if not (self.centering and mx.rows == 0 and ctx.world_size == 1):
            self._store(comp, means, sing, tv, mx.rows)
        if want_y and mx.rows == 0 and self.centering:
            y = _alloc_like(x, 0, mx.cols if self._k else 0, mx.dtype)[:, : self._k]  # src/pca.rs:210
        return y

    def fit(self, x):
        self._inner_fit(x, False)
        return self

    def fit_transform(self, x):
        return self._inner_fit(x, True)


class PcaBuilder:
    """``PcaBuilder`` (src/pca.rs:246-283)."""

    def __init__(self, n_components: int):
        self._k, self._centering, self._ctx = n_components, True, None

    @classmethod
    def new(cls, n_components: int):
        return cls(n_components)

    def centering(self, centering: bool):
        self._centering = centering
        return self

    def context(self, ctx: Context):
        self._ctx = ctx
        return self

    def build(self) -> Pca:
        return Pca(self._k, self._centering, self._ctx)


class RandomizedPca(_PcaModel):
    """``RandomizedPca<A, R>`` (src/pca.rs:317-551).  The model owns an RNG that advances on every
    fit (src/pca.rs:532); ``rng`` is anything with ``standard_normal(shape)`` filling in row-major order
    like src/pca.rs:701-705: the crate's ``Pcg`` (``with_seed`` / builder ``seed()``; restated PCG + Ziggurat,
    "stream parity unpinned") or a ``numpy.random.Generator`` (the default)."""

    N_OVERSAMPLE = 10  # src/pca.rs:679
    N_ITER = 7         # src/pca.rs:680

    def __init__(self, n_components, centering=True, rng=None, ctx=None, n_oversample=None, n_iter=None):
        super().__init__(n_components, centering, ctx)
        self.rng = rng if rng is not None else np.random.default_rng()
        self.n_oversample = self.N_OVERSAMPLE if n_oversample is None else int(n_oversample)
        self.n_iter = self.N_ITER if n_iter is None else int(n_iter)

    @classmethod
    def new(cls, n_components: int, ctx=None):
        return cls(n_components, ctx=ctx)

    @classmethod
    def with_seed(cls, n_components: int, seed: int, ctx=None):
        """``Pcg::from_seed(seed.to_be_bytes())`` (src/pca.rs:356-359)."""
        return cls(n_components, rng=Pcg.from_seed_be_bytes(seed), ctx=ctx)

    def _serde_fields(self) -> dict:
        return {"rng": _rng_to_serde(self.rng), **super()._serde_fields()}

    @classmethod
    def from_json(cls, text: str, dtype=np.float32, ctx=None):
        """``serde_json::from_str::<RandomizedPca<A, Pcg>>``."""
        import json
        obj = json.loads(text)
        m = cls(int(obj["components"]["dim"][0]), rng=Pcg.from_serde(obj["rng"]), ctx=ctx)
        m._load_serde_fields(obj, dtype)
        return m

    @classmethod
    def with_rng(cls, n_components: int, rng, ctx=None):
        return cls(n_components, rng=rng, ctx=ctx)

    def draw_omega(self, d: int, dtype_code: int) -> np.ndarray:
        size = self._k + self.n_oversample
        return self.rng.standard_normal((d, size)).astype(_np_dtype(dtype_code))  # f64 draw cast to A::Real

    def _inner_fit(self, x, want_y: bool, omega=None):
        keep = []
        mx = describe(x, keep)
        ctx = self._ctx()
        npdt = _np_dtype(mx.dtype)
        if omega is None:
            omega = self.draw_omega(mx.cols, mx.dtype)
        omega = _host(omega, mx.dtype, (mx.cols, self._k + self.n_oversample))
        comp = np.empty((self._k, mx.cols), dtype=npdt)   # (every element is written by a successful fit; a failed one raises)
        means = np.zeros(mx.cols, dtype=npdt)            # (an empty input without centering returns before the means are written)
        sing = np.empty(self._k, dtype=npdt)
        tv = np.zeros(1, dtype=npdt)
        y, my = None, None
        if want_y:
            y = _alloc_like(x, mx.rows, self._k, mx.dtype)
            my = describe(y, keep)
        ctx.check(ctx.lib.petal_rpca_fit(ctx._h, C.byref(mx), self._k, self.n_oversample, self.n_iter,
                                         int(self.centering), omega.ctypes.data, comp.ctypes.data, means.ctypes.data,
                                         sing.ctypes.data, tv.ctypes.data, C.byref(my) if my is not None else None))
        if not (self.centering and mx.rows == 0 and ctx.world_size == 1):
            self._store(comp, means, sing, tv, mx.rows)
        return y

    def fit(self, x, omega=None):
        self._inner_fit(x, False, omega)
        return self

    def fit_transform(self, x, omega=None):
        return self._inner_fit(x, True, omega)


class RandomizedPcaBuilder:
    """``RandomizedPcaBuilder<R>`` (src/pca.rs:564-663)."""

    def __init__(self, n_components: int, rng=None):
        self._k, self._rng, self._centering, self._ctx = n_components, rng, True, None
        self._n_iter, self._n_oversample = None, None

    @classmethod
    def new(cls, n_components: int):
        return cls(n_components)

    @classmethod
    def with_rng(cls, rng, n_components: int):
        return cls(n_components, rng)

    def seed(self, seed: int):
        """``Pcg::from_seed(seed.to_be_bytes())`` like the crate's builders (src/pca.rs:599-602, src/ica.rs:274-277)."""
        self._rng = Pcg.from_seed_be_bytes(seed)
        return self

    def centering(self, centering: bool):
        self._centering = centering
        return self

    def context(self, ctx: Context):
        self._ctx = ctx
        return self

    def n_iter(self, n_iter: int):  # extension: the crate hard-codes 7
        self._n_iter = n_iter
        return self

    def n_oversample(self, n: int):  # extension: the crate hard-codes 10
        self._n_oversample = n
        return self

    def build(self) -> RandomizedPca:
        return RandomizedPca(self._k, self._centering, self._rng, self._ctx, self._n_oversample, self._n_iter)


class FastIca:
    """``FastIca<A, R>`` (src/ica.rs:41-221)."""

    TOL, MAX_ITER = 1e-4, 200  # src/ica.rs:216

    def __init__(self, rng=None, ctx=None, n_components: int = 0, mode: int = ICA_TEXTBOOK, tol=None, max_iter=None):
        self.rng = rng if rng is not None else np.random.default_rng()
        self.ctx = ctx
        self.n_components = int(n_components)  # extension: the crate always uses min(n, d) (src/ica.rs:173)
        self.mode = mode
        self.tol = self.TOL if tol is None else tol
        self.max_iter = self.MAX_ITER if max_iter is None else max_iter
        self.components = np.zeros((0, 0))
        self.means = np.zeros(0)
        self.n_iter = 0

    @classmethod
    def new(cls, ctx=None):
        return cls(ctx=ctx)

    @classmethod
    def with_seed(cls, seed: int, ctx=None):
        """``Pcg::from_seed(seed.to_be_bytes())`` (src/ica.rs:75-78)."""
        return cls(Pcg.from_seed_be_bytes(seed), ctx)

    def to_json(self) -> str:
        """The JSON ``serde_json::to_string(&ica)`` writes for the reference struct (src/ica.rs:41-50, 428)."""
        import json
        return json.dumps({"rng": _rng_to_serde(self.rng), "components": _nd_to_serde(self.components),
                           "means": _nd_to_serde(self.means), "n_iter": int(self.n_iter)})

    @classmethod
    def from_json(cls, text: str, dtype=np.float64, ctx=None):
        """``serde_json::from_str::<FastIca<A>>``."""
        import json
        obj = json.loads(text)
        m = cls(Pcg.from_serde(obj["rng"]), ctx)
        m.components = _nd_from_serde(obj["components"], dtype)
        m.means = _nd_from_serde(obj["means"], dtype)
        m.n_iter = int(obj["n_iter"])
        return m

    @classmethod
    def with_rng(cls, rng, ctx=None):
        return cls(rng, ctx)

    def _ctx(self) -> Context:
        if self.ctx is None:
            self.ctx = default_context()
        return self.ctx

    def _inner_fit(self, x, want_y: bool, w_init=None):
        keep = []
        mx = describe(x, keep)
        ctx = self._ctx()
        npdt = _np_dtype(mx.dtype)
        if mx.rows == 0 and ctx.world_size == 1:  # src/ica.rs:174-176
            return _alloc_like(x, 0, mx.cols, mx.dtype) if want_y else None
        nc = self.n_components or min(mx.rows if ctx.world_size == 1 else 1 << 62, mx.cols)
        if w_init is None:
            w_init = self.rng.standard_normal((nc, nc))  # src/ica.rs:210-214
        w_init = _host(w_init, mx.dtype, (nc, nc))
        comp = np.zeros((nc, mx.cols), dtype=npdt)
        means = np.zeros(mx.cols, dtype=npdt)
        n_iter = C.c_int64(0)
        y, my = None, None
        if want_y:
            y = _alloc_like(x, mx.rows, nc, mx.dtype)
            my = describe(y, keep)
        ctx.check(ctx.lib.petal_fastica_fit(ctx._h, C.byref(mx), self.n_components, float(self.tol),
                                            int(self.max_iter), int(self.mode), w_init.ctypes.data, comp.ctypes.data,
                                            means.ctypes.data, C.byref(n_iter), C.byref(my) if my is not None else None))
        self.components, self.means, self.n_iter = comp, means, int(n_iter.value)
        return y

    def fit(self, x, w_init=None):
        self._inner_fit(x, False, w_init)
        return self

    def fit_transform(self, x, w_init=None):
        return self._inner_fit(x, True, w_init)

    def transform(self, x):
        """src/ica.rs:120-131 (always centres)."""
        keep = []
        mx = describe(x, keep)
        ctx = self._ctx()
        d = self.means.shape[0]
        if mx.cols != d:
            raise InvalidInput("too many columns")  # src/ica.rs:124-128
        nc = self.components.shape[0]
        y = _alloc_like(x, mx.rows, nc, mx.dtype)
        my = describe(y, keep)
        comp = _host(self.components, mx.dtype)
        mu = _host(self.means, mx.dtype)
        ctx.check(ctx.lib.petal_transform(ctx._h, C.byref(mx), comp.ctypes.data, mu.ctypes.data, nc, d, 1, C.byref(my)))
        return y


class FastIcaBuilder:
    """``FastIcaBuilder<R>`` (src/ica.rs:244-308)."""

    def __init__(self, rng=None):
        self._rng, self._ctx, self._nc, self._mode = rng, None, 0, ICA_TEXTBOOK

    @classmethod
    def new(cls):
        return cls()

    @classmethod
    def with_rng(cls, rng):
        return cls(rng)

    def seed(self, seed: int):
        """``Pcg::from_seed(seed.to_be_bytes())`` like the crate's builders (src/pca.rs:599-602, src/ica.rs:274-277)."""
        self._rng = Pcg.from_seed_be_bytes(seed)
        return self

    def context(self, ctx: Context):
        self._ctx = ctx
        return self

    def n_components(self, nc: int):
        self._nc = nc
        return self

    def mode(self, mode: int):
        self._mode = mode
        return self

    def build(self) -> FastIca:
        return FastIca(self._rng, self._ctx, self._nc, self._mode)


# ---- crate-private kernels that carry known-answer tests -------------------------------------------
def ica_par(x1, tol, max_iter, w_init, mode=ICA_TEXTBOOK, ctx: Optional[Context] = None):
    """``ica_par`` (src/ica.rs:319-361): x1 is nc x n.  Returns (W, n_iter)."""
    ctx = ctx or default_context()
    keep = []
    mx = describe(x1, keep)
    w0 = _host(w_init, mx.dtype, (mx.rows, mx.rows))
    w = np.zeros_like(w0)
    n_iter = C.c_int64(0)
    ctx.check(ctx.lib.petal_ica_par(ctx._h, C.byref(mx), float(tol), int(max_iter), int(mode), w0.ctypes.data,
                                    w.ctypes.data, C.byref(n_iter)))
    return w, int(n_iter.value)


def symmetric_decorrelation(w, mode=ICA_TEXTBOOK, ctx: Optional[Context] = None):
    """``symmetric_decorrelation`` (src/ica.rs:363-381)."""
    ctx = ctx or default_context()
    w = np.ascontiguousarray(np.asarray(w))
    if w.dtype not in (np.float32, np.float64):
        w = w.astype(np.float64)
    dt = PETAL_F32 if w.dtype == np.float32 else PETAL_F64
    out = np.zeros_like(w)
    ctx.check(ctx.lib.petal_symmetric_decorrelation(ctx._h, w.ctypes.data, w.shape[0], dt, int(mode), out.ctypes.data))
    return out


def logcosh(x, ctx: Optional[Context] = None):
    """``logcosh`` (src/ica.rs:383-398): returns (tanh(x), mean_j(1 - tanh(x)_ij^2))."""
    ctx = ctx or default_context()
    keep = []
    mx = describe(x, keep)
    g = _alloc_like(x, mx.rows, mx.cols, mx.dtype)
    mg = describe(g, keep)
    gp = np.zeros(mx.rows, dtype=_np_dtype(mx.dtype))
    ctx.check(ctx.lib.petal_logcosh(ctx._h, C.byref(mx), C.byref(mg), gp.ctypes.data))
    return g, gp


def svd_flip(u, vt, ctx: Optional[Context] = None):
    """``svd_flip`` (src/pca.rs:815-850): in place on u (n x m) and vt (m' x d)."""
    ctx = ctx or default_context()
    keep = []
    mu, mv = describe(u, keep), describe(vt, keep)
    ctx.check(ctx.lib.petal_svd_flip(ctx._h, C.byref(mu), C.byref(mv)))
    return u, vt


def gemm_xp(x, p, mu=None, bias=None, ctx: Optional[Context] = None):
    """z = (x - mu) . p + bias -- the K1 power-iteration GEMM kernel on its own (src/pca.rs:707, 714)."""
    ctx = ctx or default_context()
    keep = []
    mx = describe(x, keep)
    ph = _host(p, mx.dtype)
    if ph.shape[0] != mx.cols:
        raise InvalidInput(f"p should have {mx.cols} rows")
    N = ph.shape[1]
    muh = _host(mu, mx.dtype, (mx.cols,)) if mu is not None else None
    bh = _host(bias, mx.dtype, (N,)) if bias is not None else None
    z = _alloc_like(x, mx.rows, N, mx.dtype)
    mz = describe(z, keep)
    ctx.check(ctx.lib.petal_gemm_xp(ctx._h, C.byref(mx), muh.ctypes.data if muh is not None else None, ph.ctypes.data, N,
                                    bh.ctypes.data if bh is not None else None, C.byref(mz)))
    return z


def power_pass(x, p, mu=None, want_z=False, ctx: Optional[Context] = None):
    """(y, z, fused): y (fp64, host) = (x - mu)^T ((x - mu) p) -- one power iteration of the range finder as ONE pass over x
    (src/pca.rs:711 + 714) where the fused kernel exists (fused = True), the two GEMM kernels otherwise; z = (x - mu) p if wanted."""
    ctx = ctx or default_context()
    keep = []
    mx = describe(x, keep)
    ph = _host(p, mx.dtype)
    if ph.shape[0] != mx.cols:
        raise InvalidInput(f"p should have {mx.cols} rows")
    N = ph.shape[1]
    muh = _host(mu, mx.dtype, (mx.cols,)) if mu is not None else None
    y = np.zeros((mx.cols, N), dtype=np.float64)
    z = _alloc_like(x, mx.rows, N, mx.dtype) if want_z else None
    mz = describe(z, keep) if want_z else None
    fused = C.c_int(0)
    ctx.check(ctx.lib.petal_power_pass(ctx._h, C.byref(mx), muh.ctypes.data if muh is not None else None, ph.ctypes.data, N,
                                       y.ctypes.data_as(C.POINTER(C.c_double)), C.byref(mz) if want_z else None, C.byref(fused)))
    return y, z, bool(fused.value)


def gemm_atb(a, b=None, mu_a=None, mu_b=None, ctx: Optional[Context] = None):
    """c (fp64, host) = (a - mu_a)^T . (b - mu_b) -- the K2 power-iteration GEMM kernel on its own
    (src/pca.rs:711, 681); b=None means b = a."""
    ctx = ctx or default_context()
    keep = []
    ma = describe(a, keep)
    mb = describe(b, keep) if b is not None else None
    M, N = ma.cols, (mb.cols if mb is not None else ma.cols)
    mah = _host(mu_a, ma.dtype, (M,)) if mu_a is not None else None
    mbh = _host(mu_b, ma.dtype, (N,)) if mu_b is not None else None
    out = np.zeros((M, N), dtype=np.float64)
    ctx.check(ctx.lib.petal_gemm_atb(ctx._h, C.byref(ma), mah.ctypes.data if mah is not None else None,
                                     C.byref(mb) if mb is not None else None,
                                     mbh.ctypes.data if mbh is not None else None,
                                     out.ctypes.data_as(C.POINTER(C.c_double))))
    return out
