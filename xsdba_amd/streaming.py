"""Out-of-core driver: blocks of grid cells stream through the GPU.

The reference leaves problems larger than memory to dask: every chunk of cells is an independent ``map_blocks`` call
(base.py:640-702).  Here the host cubes (numpy arrays or memmaps, ``[cell, time]`` C-contiguous) are cut into blocks of
cells whose slices are page-locked only while their block is in flight; three HIP streams (three library contexts on one device) run as a pipeline ordered by events:

    stream "in"   host -> device copy of block i+1          (DMA at PCIe rate from the pinned cubes)
    stream "k"    train + adjust kernels of block i
    stream "out"  device -> host copy of the results of block i-1

with two sets of device buffers, so the copies hide behind the kernels (or the other way round: for EQM on day-of-year
groups the GPU computes ~2.3 M cells/s while PCIe delivers the three input series of ~0.4 M cells/s — the pipeline
then runs at the PCIe rate instead of the sum of the three stages).
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import hip
from ._lib import Context, DeviceArray, check, load


def _ptr(a: np.ndarray, cell0: int = 0) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data + cell0 * a.strides[0])


class _BlockPins:
    """Page-locks host slices block by block: a slice is pinned just before its block is queued and released once the
    block's read-back has completed, so at most a few blocks are locked at any time — cubes larger than the lockable
    memory (numpy memmaps of a dataset larger than RAM) stream through like dask chunks would.  Slices that cannot be
    locked (exotic mappings) are simply copied unpinned."""

    def __init__(self, ctx: Context):
        self.ctx, self.live = ctx, {}

    def pin(self, key, arrays, c0, n):
        done = []
        for a in arrays:
            p = _ptr(a, c0)
            if load().xsdba_hip_host_register(self.ctx.handle, p, C.c_size_t(n * a.strides[0])) == 0:
                done.append(p)
        self.live[key] = done

    def release(self, key):
        for p in self.live.pop(key, []):
            load().xsdba_hip_host_unregister(self.ctx.handle, p)

    def release_all(self):
        for key in list(self.live):
            self.release(key)


def _rec(ctx: Context, slot: int):
    check(load().xsdba_hip_event_record(ctx.handle, C.c_int(slot)), "event_record")


def _wait(waiter: Context, src: Context, slot: int):
    check(load().xsdba_hip_stream_wait_event(waiter.handle, src.handle, C.c_int(slot)), "stream_wait_event")


def _copy(ctx: Context, dst, src, nbytes: int):
    check(load().xsdba_hip_memcpy_async(ctx.handle, dst, src, C.c_size_t(nbytes)), "memcpy_async")


def train_adjust_streamed(method, ref, hist, sim, gidx, G, W, q, kind="+", *, block_cells=8192, scen=None, time_num=None,
                          interp="nearest", extrap="constant", rank_window=False, device=0, keep_tables=False, stats=None):
    """EQM / DQM / QDM train + adjust of host cubes ``ref, hist, sim`` ([cell, time], C-contiguous float32), streamed
    through the GPU in blocks of ``block_cells`` cells.  Returns ``scen`` ([cell, time], filled in place when given) and,
    with ``keep_tables``, the trained tables ``af, hist_q`` ([cell, group, quantile]) as well.

    method: "eqm" | "dqm" | "qdm".  ``stats`` (a dict) receives wall time, bytes moved and the achieved PCIe rate."""
    if method not in ("eqm", "dqm", "qdm"):
        raise ValueError("method must be 'eqm', 'dqm' or 'qdm'")
    arrs = [np.asarray(a) for a in (ref, hist, sim)]
    for a in arrs:
        if a.ndim != 2 or a.dtype != np.float32 or not a.flags.c_contiguous or a.shape != arrs[0].shape:
            raise ValueError("ref, hist, sim must be C-contiguous float32 [cell, time] arrays of one shape")
    ref, hist, sim = arrs
    Cn, T = ref.shape
    Q = len(q)
    q = np.ascontiguousarray(q, np.float32)
    gidx = np.ascontiguousarray(gidx, np.int32) if gidx is not None else None
    if method == "dqm" and time_num is None:
        raise ValueError("dqm needs time_num (ns since 1970 of every time step)")
    if scen is None:
        scen = np.empty((Cn, T), np.float32)
    elif scen.shape != (Cn, T) or scen.dtype != np.float32 or not scen.flags.c_contiguous:
        raise ValueError("scen must be a C-contiguous float32 [cell, time] array")
    af_h = np.empty((Cn, G, Q), np.float32) if keep_tables else None
    hq_h = np.empty((Cn, G, Q), np.float32) if keep_tables else None
    B = int(min(block_cells, Cn))
    nblocks = (Cn + B - 1) // B
    c_in, c_k, c_out = Context(device), Context(device), Context(device)
    NS = 2  # device buffer sets
    f32 = np.float32
    d_ref = [c_k.empty((B, T), f32) for _ in range(NS)]
    d_hist = [c_k.empty((B, T), f32) for _ in range(NS)]
    d_sim = [c_k.empty((B, T), f32) for _ in range(NS)]
    d_scen = [c_k.empty((B, T), f32) for _ in range(NS)]
    d_af = [c_k.empty((B, G, Q), f32) for _ in range(NS)]
    d_hq = [c_k.empty((B, G, Q), f32) for _ in range(NS)]
    d_sc = [c_k.empty((B, G), f32) for _ in range(NS)] if method == "dqm" else None
    d_tr = [c_k.empty((B, T), f32) for _ in range(NS)] if method == "dqm" else None
    if method == "dqm":  # resident once: a host array would make every block's call wait for its own upload
        time_num = c_k.to_device(np.ascontiguousarray(time_num, np.float64))
    # event slots (per context): s = buffer set; "in": s -> block uploaded; "k": s -> block computed; "out": s -> read back
    pinned = [ref, hist, sim, scen] + ([af_h, hq_h] if keep_tables else [])
    pins = _BlockPins(c_k)
    t_pin = 0.0
    t1 = time.perf_counter()
    try:
        for i in range(nblocks):
            s = i % NS
            c0, n = i * B, min(B, Cn - i * B)
            rows = n * T * 4
            # ---- host slices: block i - NS has been read back (the host waits on its event only), its slices are unpinned;
            #      (the slices of block i were pinned while block i - 1 was being queued / computed)
            if i >= NS:
                check(load().xsdba_hip_event_synchronize(c_out.handle, C.c_int(s)), "event_synchronize")
                pins.release(i - NS)
            if i == 0:
                pins.pin(0, pinned, c0, n)
            # ---- upload: the set's inputs were last read by the kernels of block i - NS
            _wait(c_in, c_k, s)
            for dst, src in ((d_ref[s], ref), (d_hist[s], hist), (d_sim[s], sim)):
                _copy(c_in, C.c_void_p(dst.ptr), _ptr(src, c0), rows)
            _rec(c_in, s)
            # ---- kernels: need the upload of this block, and the read-back of block i - NS (it reads scen / tables of the set)
            _wait(c_k, c_in, s)
            _wait(c_k, c_out, s)
            view = lambda d, shape: DeviceArray.from_ptr(c_k, d.ptr, shape, f32)  # noqa: E731  (the last block may be short)
            r_, h_, s_ = view(d_ref[s], (n, T)), view(d_hist[s], (n, T)), view(d_sim[s], (n, T))
            af_, hq_, sc_o = view(d_af[s], (n, G, Q)), view(d_hq[s], (n, G, Q)), view(d_scen[s], (n, T))
            if method == "dqm":
                scl, trd = view(d_sc[s], (n, G)), view(d_tr[s], (n, T))
                hip.dqm_train(r_, h_, gidx, G, W, q, kind, ctx=c_k, out=(af_, hq_, scl))
                hip.dqm_adjust(s_, af_, hq_, scl, gidx, W, time_num, 1, interp, extrap, kind, ctx=c_k, out=(sc_o, trd))
            else:
                hip.eqm_train(r_, h_, gidx, G, W, q, kind, ctx=c_k, out=(af_, hq_))
                if method == "qdm":
                    hip.qdm_adjust(s_, af_, q, gidx, W, rank_window, interp, extrap, kind, want_sim_q=False, ctx=c_k, out=sc_o)
                else:
                    hip.qm_adjust(s_, af_, hq_, gidx, interp, extrap, kind, ctx=c_k, out=sc_o)
            _rec(c_k, s)
            # ---- read-back
            _wait(c_out, c_k, s)
            _copy(c_out, _ptr(scen, c0), C.c_void_p(d_scen[s].ptr), rows)
            if keep_tables:
                _copy(c_out, _ptr(af_h, c0), C.c_void_p(d_af[s].ptr), n * G * Q * 4)
                _copy(c_out, _ptr(hq_h, c0), C.c_void_p(d_hq[s].ptr), n * G * Q * 4)
            _rec(c_out, s)
            # ---- the next block's slices are page-locked now, while the GPU works on this one
            if i + 1 < nblocks:
                tp = time.perf_counter()
                pins.pin(i + 1, pinned, (i + 1) * B, min(B, Cn - (i + 1) * B))
                t_pin += time.perf_counter() - tp
        c_in.synchronize(); c_k.synchronize(); c_out.synchronize()
        t_pipe = time.perf_counter() - t1
    finally:
        # whatever happened: nothing may still be copying from / into a slice when it is unpinned, and every context and
        # device buffer of the pipeline is given back here, not when the garbage collector gets to them
        for ctx in (c_in, c_k, c_out):
            try:
                ctx.synchronize()
            except Exception:  # noqa: BLE001
                pass
        pins.release_all()
        for lst in (d_ref, d_hist, d_sim, d_scen, d_af, d_hq, d_sc or [], d_tr or []):
            for d in lst:
                try:
                    d.free()
                except Exception:  # noqa: BLE001
                    pass
            lst.clear()
        if isinstance(time_num, DeviceArray):
            time_num.free()
        for ctx in (c_in, c_out, c_k):
            ctx.close()
    if stats is not None:
        h2d = 3 * Cn * T * 4
        d2h = Cn * T * 4 + (2 * Cn * G * Q * 4 if keep_tables else 0)
        stats.update(cells=Cn, blocks=nblocks, block_cells=B, pin_s=t_pin, pipeline_s=t_pipe, h2d_bytes=h2d, d2h_bytes=d2h,
                     h2d_GBps=h2d / t_pipe / 1e9, cells_per_s=Cn / t_pipe, cells_per_s_incl_pinning=Cn / t_pipe)  # (pinning happens inside the pipeline now)
    return (scen, af_h, hq_h) if keep_tables else scen
