"""ctypes binding of ``libxsdba_hip.so`` (C ABI: ``include/xsdba_hip.h``).

The library is the only compute backend of this package: if it is missing, was not built for the GPU
that is present, or no MI355X is visible, the calls below raise — there is no CPU fallback.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import subprocess
import threading
import weakref
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "libxsdba_hip.so"
CSRC = _PKG / "csrc"

F32, F64 = 0, 1
INTERP = {"nearest": 0, "linear": 1, "cubic": 2}
EXTRAP = {"constant": 0, "nan": 1}

# every symbol include/xsdba_hip.h declares (checked by tests/test_cabi_symbols.py against the header)
SYMBOLS = [
    "xsdba_hip_abi_version", "xsdba_hip_device_count", "xsdba_hip_create", "xsdba_hip_destroy",
    "xsdba_hip_last_error", "xsdba_hip_lane_ordered_atomics", "xsdba_hip_synchronize", "xsdba_hip_get_stream", "xsdba_hip_malloc", "xsdba_hip_free",
    "xsdba_hip_memcpy_h2d", "xsdba_hip_memcpy_d2h", "xsdba_hip_memset", "xsdba_hip_host_register",
    "xsdba_hip_host_unregister", "xsdba_hip_memcpy_async", "xsdba_hip_event_record", "xsdba_hip_stream_wait_event", "xsdba_hip_event_synchronize", "xsdba_hip_shard_plan", "xsdba_hip_comm_unique_id", "xsdba_hip_comm_init", "xsdba_hip_comm_info", "xsdba_hip_gather", "xsdba_hip_comm_destroy", "xsdba_hip_profile_enable",
    "xsdba_hip_profile_read", "xsdba_hip_timer_start", "xsdba_hip_timer_stop", "xsdba_hip_quantile_groups",
    "xsdba_hip_eqm_train", "xsdba_hip_dqm_train", "xsdba_hip_qm_adjust", "xsdba_hip_qm_adjust_linear2d", "xsdba_hip_qm_adjust_cubic2d", "xsdba_hip_dqm_adjust",
    "xsdba_hip_qdm_adjust", "xsdba_hip_tail_mask", "xsdba_hip_jitter", "xsdba_hip_adapt_freq", "xsdba_hip_group_means", "xsdba_hip_scaling_adjust", "xsdba_hip_loci_train", "xsdba_hip_loci_adjust", "xsdba_hip_rank_groups", "xsdba_hip_polydetrend_trend", "xsdba_hip_polydetrend_fit", "xsdba_hip_series_correction", "xsdba_hip_rolling_mean", "xsdba_hip_synth_series", "xsdba_hip_vecquantiles",
]


class XsdbaHipError(RuntimeError):
    """A call into libxsdba_hip.so returned a non-zero status."""

    def __init__(self, code: int, message: str, what: str):
        super().__init__(f"{what}: [{code}] {message}")
        self.code = code


def build(force: bool = False) -> Path:
    """Compile the library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [CSRC / "api.hip", *sorted(CSRC.glob("*.hpp")), _PKG.parent / "include" / "xsdba_hip.h"]
    if force or not LIB_PATH.exists() or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in srcs):
        env = dict(os.environ)
        proc = subprocess.run(["make", "-C", str(CSRC)], env=env, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"building libxsdba_hip.so failed:\n{proc.stdout}\n{proc.stderr}")
    return LIB_PATH


_lib = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """dlopen the in-tree library (no implicit build, no fallback)."""
    global _lib
    with _lock:
        if _lib is None:
            if not LIB_PATH.exists():
                raise ImportError(
                    f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `make -C xsdba_amd/csrc`). xsdba_amd has no CPU fallback."
                )
            lib = C.CDLL(str(LIB_PATH))
            for s in SYMBOLS:
                getattr(lib, s).restype = C.c_int
            _lib = lib
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    load().xsdba_hip_last_error(buf, C.c_size_t(1024))
    return buf.value.decode(errors="replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise XsdbaHipError(rc, last_error(), what)


def device_count() -> int:
    n = C.c_int(0)
    check(load().xsdba_hip_device_count(C.byref(n)), "device_count")
    return n.value


def dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt == np.float32:
        return F32
    if dt == np.float64:
        return F64
    raise TypeError(f"xsdba_amd supports float32/float64 data, got {dt}")


def _release_device_buffer(ctx_ref, ptr: int) -> None:
    """Finalizer of an owning DeviceArray: give the buffer back unless its context is already gone (destroying a context
    frees everything it handed out).  Never raises: it may run during interpreter shutdown."""
    try:
        ctx = ctx_ref()
        if ctx is not None and ctx.handle:
            load().xsdba_hip_free(ctx.handle, C.c_void_p(ptr))
    except Exception:  # pragma: no cover
        pass


class DeviceArray:
    """A dense device buffer with numpy-like shape/dtype/strides metadata (element strides).

    Ownership: an array made by the constructor (or returned by a ``hip.*`` call that was not given ``out=``) owns its
    buffer and returns it to the context when it is garbage-collected or ``free()``d, whichever comes first; ``from_ptr``
    views own nothing.  Buffers still alive when their ``Context`` is closed are released with it."""

    def __init__(self, ctx: "Context", shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        check(load().xsdba_hip_malloc(ctx.handle, C.c_size_t(max(self.nbytes, 1)), C.byref(p)), "malloc")
        self.ptr = p.value
        self._finalizer = weakref.finalize(self, _release_device_buffer, weakref.ref(ctx), self.ptr)

    @classmethod
    def from_ptr(cls, ctx: "Context", ptr: int, shape, dtype) -> "DeviceArray":
        """Non-owning view of device memory allocated elsewhere (e.g. a torch tensor's data_ptr())."""
        self = cls.__new__(cls)
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.ptr = int(ptr)
        self.foreign = True
        return self

    # ---- numpy-like metadata (what the labelled layer needs to carry a device buffer around without touching it) ----
    @property
    def ndim(self) -> int:
        return len(self.shape)

    @property
    def size(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64))

    def reshape(self, *shape) -> "DeviceArray":
        """A view of the same (dense, C-ordered) buffer under another shape; the view keeps its base alive."""
        if len(shape) == 1 and not np.isscalar(shape[0]):
            shape = tuple(shape[0])
        if getattr(self, "cell_time_strides", None) is not None:
            raise ValueError("a time-major device series cannot be reshaped")
        known = int(np.prod([s for s in shape if s != -1], dtype=np.int64))
        shape = tuple(int(self.size // max(known, 1)) if s == -1 else int(s) for s in shape)
        if int(np.prod(shape, dtype=np.int64)) != self.size:
            raise ValueError(f"cannot reshape device array of shape {self.shape} into {shape}")
        v = DeviceArray.from_ptr(self.ctx, self.ptr, shape, self.dtype)
        v._base = self
        return v

    def astype(self, dtype, copy: bool = True) -> "DeviceArray":
        if np.dtype(dtype) != self.dtype:
            raise TypeError(f"device arrays are not converted implicitly ({self.dtype} -> {np.dtype(dtype)}): convert on the host")
        return self

    def __array__(self, dtype=None, copy=None):
        host = self.download()
        return host if dtype is None else host.astype(dtype)

    # ---- transfers ------------------------------------------------------------------------------------------------
    # Big pageable host arrays go through blocks that are page-locked just for their own copy (a few blocks in flight):
    # DMA at the PCIe rate without pinning the whole array, so arrays larger than the lockable memory (memmaps) still work.
    _PIN_BLOCK = 128 << 20
    _PIN_MIN = 32 << 20
    _PIN_GROUP = 4

    def _blocked_copy(self, host: np.ndarray, to_device: bool) -> bool:
        lib, h = load(), self.ctx.handle
        base, n = host.ctypes.data, self.nbytes
        blocks = [(o, min(self._PIN_BLOCK, n - o)) for o in range(0, n, self._PIN_BLOCK)]
        pinned: list[int] = []     # offsets registered and not yet released
        ok = True
        try:
            for gi in range(0, len(blocks), self._PIN_GROUP):
                grp = blocks[gi:gi + self._PIN_GROUP]
                new = []
                for o, m in grp:   # page-lock the next group while the previous group's copies are still running
                    if lib.xsdba_hip_host_register(h, C.c_void_p(base + o), C.c_size_t(m)) != 0:
                        ok = False
                        break
                    new.append(o)
                if gi:
                    check(lib.xsdba_hip_synchronize(h), "synchronize")
                    for o in pinned:
                        lib.xsdba_hip_host_unregister(h, C.c_void_p(base + o))
                    pinned = []
                pinned = new
                if not ok:
                    break
                for o, m in grp:
                    d, s_ = C.c_void_p(self.ptr + o), C.c_void_p(base + o)
                    check(lib.xsdba_hip_memcpy_async(h, d if to_device else s_, s_ if to_device else d, C.c_size_t(m)), "memcpy_async")
        finally:
            lib.xsdba_hip_synchronize(h)
            for o in pinned:
                lib.xsdba_hip_host_unregister(h, C.c_void_p(base + o))
        return ok

    def upload(self, host: np.ndarray) -> "DeviceArray":
        host = np.ascontiguousarray(host, self.dtype)
        assert host.shape == self.shape
        if self.nbytes >= self._PIN_MIN and self._blocked_copy(host, True):
            return self
        check(load().xsdba_hip_memcpy_h2d(self.ctx.handle, C.c_void_p(self.ptr), host.ctypes.data_as(C.c_void_p),
                                          C.c_size_t(self.nbytes)), "memcpy_h2d")
        return self

    def download(self, out: np.ndarray | None = None) -> np.ndarray:
        """the buffer on the host; into `out` (C-contiguous, same shape and dtype: e.g. a slice of a larger array) when given"""
        if out is None:
            out = np.empty(self.shape, self.dtype)
        elif out.shape != tuple(self.shape) or out.dtype != self.dtype or not out.flags.c_contiguous:
            raise ValueError("download(out=...): a C-contiguous array of the buffer's shape and dtype")
        if not (self.nbytes >= self._PIN_MIN and self._blocked_copy(out, False)):
            check(load().xsdba_hip_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr),
                                              C.c_size_t(self.nbytes)), "memcpy_d2h")
        return out

    def free(self) -> None:
        """Release the buffer now (idempotent; a no-op for views and once the context has been closed)."""
        if getattr(self, "foreign", False):
            self.ptr = None
            return
        fin = getattr(self, "_finalizer", None)
        if fin is not None:
            fin.detach()
        if self.ptr and self.ctx.handle:
            check(load().xsdba_hip_free(self.ctx.handle, C.c_void_p(self.ptr)), "free")
        self.ptr = None


class Context:
    """One per (process, device): owns the stream, scratch and cached group plans."""

    def __init__(self, device: int = 0):
        self.handle = C.c_void_p()
        check(load().xsdba_hip_create(C.c_int(device), C.byref(self.handle)), "create")
        self.device = device

    def close(self):
        if self.handle:
            load().xsdba_hip_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(load().xsdba_hip_synchronize(self.handle), "synchronize")

    @property
    def lane_ordered_atomics(self) -> bool:
        """False when the probe of ``xsdba_hip_create`` found returning LDS atomics out of lane order on this device: the
        per-cell sort kernels are then bypassed (include/xsdba_hip.h)."""
        v = C.c_int(0)
        check(load().xsdba_hip_lane_ordered_atomics(self.handle, C.byref(v)), "lane_ordered_atomics")
        return bool(v.value)

    def stream(self) -> int:
        p = C.c_void_p()
        check(load().xsdba_hip_get_stream(self.handle, C.byref(p)), "get_stream")
        return p.value or 0

    def empty(self, shape, dtype) -> DeviceArray:
        return DeviceArray(self, shape, dtype)

    def to_device(self, host: np.ndarray) -> DeviceArray:
        host = np.ascontiguousarray(host)
        return DeviceArray(self, host.shape, host.dtype).upload(host)

    def synth(self, shape_ct, dtype, generator: int, array_id: int, seed: int, cell_offset: int = 0,
              nan_cell_frac: float = 0.0, nan_iid_frac: float = 0.0, time_major: bool = False) -> DeviceArray:
        """Device-generated synthetic [cell, time] block (bench / tests); see xsdba_hip_synth_series."""
        Cn, T = shape_ct
        arr = DeviceArray(self, (T, Cn) if time_major else (Cn, T), dtype)
        if time_major:
            arr.logical_shape = (Cn, T)
            arr.cell_time_strides = (1, Cn)
        st, sc = (Cn, 1) if time_major else (1, T)
        check(load().xsdba_hip_synth_series(self.handle, C.c_void_p(arr.ptr), C.c_int(dtype_code(dtype)), C.c_int64(T),
                                            C.c_int64(Cn), C.c_int64(st), C.c_int64(sc), C.c_int(generator),
                                            C.c_int(array_id), C.c_uint64(seed), C.c_int64(cell_offset),
                                            C.c_float(nan_cell_frac), C.c_float(nan_iid_frac)), "synth_series")
        return arr

    def profile_enable(self, on: bool = True):
        check(load().xsdba_hip_profile_enable(self.handle, C.c_int(int(on))), "profile_enable")

    def profile_read(self, reset: bool = True) -> dict:
        cap = 32
        names = (C.c_char_p * cap)()
        counts = (C.c_int64 * cap)()
        ms = (C.c_double * cap)()
        n = C.c_int(0)
        check(load().xsdba_hip_profile_read(self.handle, names, counts, ms, C.c_int(cap), C.byref(n), C.c_int(int(reset))),
              "profile_read")
        return {names[i].decode(): {"count": int(counts[i]), "total_ms": float(ms[i])} for i in range(n.value)}

    def timer_start(self):
        check(load().xsdba_hip_timer_start(self.handle), "timer_start")

    def timer_stop(self) -> float:
        ms = C.c_double(0)
        check(load().xsdba_hip_timer_stop(self.handle, C.byref(ms)), "timer_stop")
        return ms.value


_default_ctx: dict[int, Context] = {}
_tls = threading.local()


def default_context(device: int | None = None) -> Context:
    """The context a call without an explicit one runs on: the one bound to this thread by :func:`use_context` (the block
    workers of ``map_blocks``), else the process's context of ``LOCAL_RANK`` / ``XSDBA_AMD_DEVICE`` (default 0)."""
    if device is None:
        bound = getattr(_tls, "ctx", None)
        if bound is not None:
            return bound
        device = _default_device()
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def _default_device() -> int:
    device = int(os.environ.get("LOCAL_RANK", os.environ.get("XSDBA_AMD_DEVICE", "0")))
    return device % max(device_count(), 1)


@contextlib.contextmanager
def use_context(ctx: Context):
    """Bind `ctx` to the calling thread: every library call of the thread that is not given a context runs on it."""
    prev = getattr(_tls, "ctx", None)
    _tls.ctx = ctx
    try:
        yield ctx
    finally:
        _tls.ctx = prev


# ---- the node behind the API: one context per device for the cell blocks of map_blocks ---------------------------------------
_worker_ctx: dict[str, list[Context]] = {}


def worker_devices() -> list[int]:
    """Device ordinals the blocks of a host-backed call are dealt over (the role the dask workers play for the reference,
    base.py:640-650).  ``XSDBA_AMD_DEVICES`` = ``all`` | ``0,1,...`` (an ordinal may repeat: several contexts on one device);
    unset: every visible device — unless the process was given ONE device (``LOCAL_RANK`` of a one-process-per-GPU launch,
    ``XSDBA_AMD_DEVICE``), which it then keeps to."""
    spec = os.environ.get("XSDBA_AMD_DEVICES", "").strip().lower()
    n = max(device_count(), 1)
    if not spec:
        if "LOCAL_RANK" in os.environ or "XSDBA_AMD_DEVICE" in os.environ:
            return [_default_device()]
        spec = "all"
    if spec == "all":
        return list(range(n))
    devs = [int(tok) for tok in spec.split(",") if tok.strip() != ""]
    for d in devs:
        if d < 0 or d >= n:
            raise ValueError(f"XSDBA_AMD_DEVICES names device {d}; {n} visible")
    return devs or [_default_device()]


def worker_contexts() -> list[Context]:
    """One context per entry of :func:`worker_devices` (created on first use, kept for the process).  The first entry of the
    process's own device is its default context, so that a one-device run is exactly the plain path."""
    devs = worker_devices()
    key = ",".join(map(str, devs))
    if key not in _worker_ctx:
        out, seen = [], set()
        for d in devs:
            if d not in seen and d == _default_device():
                out.append(default_context())
            else:
                out.append(Context(d))
            seen.add(d)
        _worker_ctx[key] = out
    return _worker_ctx[key]
