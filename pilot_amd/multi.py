"""Multi-GPU pair grid through ``libpilot_ot.so`` -- no torch, no launcher needed.

The N^2 pair problems of ``pilotpy/tools/Trajectory.py:505-515`` are independent given the replicated N x K
proportions and K x K cost, so the grid is partitioned into round-robin row shards (row r -> shard r mod G) and the
only exchange step is one RCCL all-gather of the row blocks.  Both forms below run the same kernels and produce the
bits of the single-device matrix:

* :class:`MultiPlan` -- ONE process drives G devices (``pilot_ot_multi_*``: ``ncclCommInitAll``, a plan and a stream
  per device, grouped ``ncclAllGather``).  ``tl.wasserstein_distance(engine_options={"n_devices": G})`` ends up here.
* :class:`Comm` -- one process PER device (``pilot_ot_comm_*``: ``ncclCommInitRank``); ``bench.py`` uses it when a
  launcher (``python -m torch.distributed.run``, ``mpirun`` ...) has started G ranks.  The 128-byte RCCL unique id
  travels from rank 0 to the others through a file in the temp directory (:func:`exchange_unique_id`).
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import sys
import tempfile
import time

import numpy as np

from . import _lib
from . import engine
from .engine import CHECK_PERIOD, NUM_ITER_MAX, STOP_THR, TAU, _as_f64


def _devices(devices=None, n_devices=None):
    if devices is None:
        if n_devices is None:
            raise ValueError("give devices=[...] or n_devices=G")
        devices = list(range(int(n_devices)))
    devices = [int(d) for d in devices]
    if not devices:
        raise ValueError("empty device list")
    return np.ascontiguousarray(devices, dtype=np.int32)


@contextlib.contextmanager
def stdout_to_stderr():
    """While the block runs, file descriptor 1 points at stderr: RCCL prints a version banner with a plain printf when a
    communicator is created, and a program whose stdout is a data channel (bench.py: one JSON line) wraps communicator
    creation in this.  The APPLICATION's choice -- the library itself never touches file descriptors."""
    sys.stdout.flush()
    libc = ctypes.CDLL(None)
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


class MultiPlan:
    """Device-resident multi-GPU pair-grid problem (one process, G devices)."""

    def __init__(self, P, M, devices=None, n_devices=None, gather="auto"):
        P = _as_f64(P, "P")
        M = _as_f64(M, "M")
        if P.ndim != 2 or M.shape != (P.shape[1], P.shape[1]):
            raise ValueError("shape mismatch: P %s, M %s" % (P.shape, M.shape))
        self.N, self.K = P.shape
        self.sym = int(np.array_equal(M, M.T))
        self.equal_masses = engine.equal_masses(P)          # exact mode: the upper triangle may only be mirrored then
        self.max_cost = float(M.max()) if M.size else 0.0
        self.devices = _devices(devices, n_devices)
        self.G = len(self.devices)
        self.L = _lib.load()
        self.h = ctypes.c_void_p()
        _lib.check(self.L.pilot_ot_multi_create(self.N, self.K, _lib.iptr(self.devices), self.G, _lib.GATHER[gather],
                                                ctypes.byref(self.h)))
        _lib.check(self.L.pilot_ot_multi_set_inputs(self.h, _lib.dptr(P), _lib.dptr(M)))

    def sinkhorn(self, reg, precision="auto", num_iter_max=NUM_ITER_MAX, stop_thr=STOP_THR, tau=TAU,
                 check_period=CHECK_PERIOD, f32_floor_ulps=0.0):
        """Enqueue shard grids + the gather on every device (asynchronous)."""
        # (the library resolves "auto" -- and the max(M)/reg > 600 fallback -- once for all shards from the max(M) it recorded
        # in pilot_ot_multi_set_inputs, exactly like the single-device host entry point)
        prec = _lib.PREC[precision]
        _lib.check(self.L.pilot_ot_multi_sinkhorn(self.h, float(reg), int(num_iter_max), float(stop_thr), float(tau),
                                                  int(check_period), prec, float(f32_floor_ulps), self.sym))

    def emd(self):
        _lib.check(self.L.pilot_ot_multi_emd(self.h, int(self.sym and self.equal_masses)))

    def sync(self):
        _lib.check(self.L.pilot_ot_multi_sync(self.h))

    def fetch(self, info=False):
        E = np.empty((self.N, self.N), dtype=np.float64)
        if not info:
            _lib.check(self.L.pilot_ot_multi_fetch(self.h, _lib.dptr(E), None, None, None))
            return E
        iters = np.zeros((self.N, self.N), dtype=np.int32)
        err = np.zeros((self.N, self.N), dtype=np.float64)
        flags = np.zeros((self.N, self.N), dtype=np.int32)
        _lib.check(self.L.pilot_ot_multi_fetch(self.h, _lib.dptr(E), _lib.iptr(iters), _lib.dptr(err), _lib.iptr(flags)))
        return E, dict(iters=iters, err=err, flags=flags)

    def device_matrix(self, shard=0):
        """The assembled N x N matrix as it sits in the HBM of ``shard``'s device (every shard with the RCCL gather, shard 0
        with peer copies); sync first.  Input of the device-side consumers (``engine.silhouette_of_rows`` ...)."""
        p = ctypes.c_void_p()
        _lib.check(self.L.pilot_ot_multi_device_matrix(self.h, int(shard), ctypes.byref(p)))
        return engine.DeviceMatrix(p, self.N, owner=self)

    def rccl_info(self):
        """(n_ranks, rank) per shard as RCCL reports them (ncclCommCount / ncclCommUserRank); (0, -1) with the peer-copy gather."""
        n = np.zeros(self.G, dtype=np.int32)
        r = np.zeros(self.G, dtype=np.int32)
        _lib.check(self.L.pilot_ot_multi_rccl_info(self.h, _lib.iptr(n), _lib.iptr(r)))
        return [int(x) for x in n], [int(x) for x in r]

    def times_ms(self):
        """(per-shard kernel ms, gather ms) of the last call (HIP events on the shards' streams)."""
        g = (ctypes.c_float * self.G)()
        ga = ctypes.c_float(0.0)
        _lib.check(self.L.pilot_ot_multi_times(self.h, g, ctypes.byref(ga)))
        return np.array(g[:]), float(ga.value)

    def close(self):
        if self.h:
            self.L.pilot_ot_multi_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sinkhorn_grid_multi(P, M, reg, devices=None, n_devices=None, gather="auto", precision="auto",
                        num_iter_max=NUM_ITER_MAX, stop_thr=STOP_THR, tau=TAU, check_period=CHECK_PERIOD,
                        f32_floor_ulps=0.0, return_info=False):
    """Full N x N entropic-OT matrix over several devices (host buffers in / out; context cached in the library)."""
    P = _as_f64(P, "P")
    M = _as_f64(M, "M")
    if P.ndim != 2 or M.shape != (P.shape[1], P.shape[1]):
        raise ValueError("shape mismatch: P %s, M %s" % (P.shape, M.shape))
    N, K = P.shape
    dev = _devices(devices, n_devices)
    E = np.empty((N, N), dtype=np.float64)
    if return_info:
        iters = np.zeros((N, N), dtype=np.int32)
        err = np.zeros((N, N), dtype=np.float64)
        flags = np.zeros((N, N), dtype=np.int32)
        pi, pe, pf = _lib.iptr(iters), _lib.dptr(err), _lib.iptr(flags)
    else:
        pi = pe = pf = None
    _lib.check(_lib.load().pilot_ot_sinkhorn_grid_multi(
        _lib.dptr(P), N, K, _lib.dptr(M), float(reg), int(num_iter_max), float(stop_thr), float(tau), int(check_period),
        _lib.PREC[precision], float(f32_floor_ulps), int(np.array_equal(M, M.T)), _lib.iptr(dev), len(dev),
        _lib.GATHER[gather], _lib.dptr(E), pi, pe, pf))
    if return_info:
        return E, dict(iters=iters, err=err, flags=flags)
    return E


def emd_grid_multi(P, M, devices=None, n_devices=None, gather="auto", return_info=False):
    """Full N x N exact-OT matrix over several devices."""
    P = _as_f64(P, "P")
    M = _as_f64(M, "M")
    if P.ndim != 2 or M.shape != (P.shape[1], P.shape[1]):
        raise ValueError("shape mismatch: P %s, M %s" % (P.shape, M.shape))
    N, K = P.shape
    dev = _devices(devices, n_devices)
    E = np.empty((N, N), dtype=np.float64)
    n_aug = np.zeros((N, N), dtype=np.int32)
    _lib.check(_lib.load().pilot_ot_emd_grid_multi(_lib.dptr(P), N, K, _lib.dptr(M), int(np.array_equal(M, M.T) and engine.equal_masses(P)),
                                                   _lib.iptr(dev), len(dev), _lib.GATHER[gather], _lib.dptr(E),
                                                   _lib.iptr(n_aug)))
    if (n_aug < 0).any():
        raise _lib.PilotOTError("exact-EMD kernel: augmentation guard tripped on %d pairs" % int((n_aug < 0).sum()))
    if return_info:
        return E, dict(n_aug=n_aug)
    return E


# ---- one process per device ---------------------------------------------------------------------------------------------
def _process_start_ticks(pid):
    """Start time of a process in clock ticks since boot (/proc/<pid>/stat field 22); 0 when unavailable."""
    try:
        with open("/proc/%d/stat" % pid, "rb") as fh:
            return int(fh.read().rsplit(b")", 1)[1].split()[19])
    except (OSError, ValueError, IndexError):
        return 0


def rendezvous_key():
    """Name shared by the ranks of ONE launch on this node: the launcher is the common parent of all ranks.  Its pid AND its
    start time are part of the name, so a file left behind by a crashed launch whose pid was reused can never match."""
    ppid = os.getppid()
    return "%s_%s_%d_%d" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"), ppid,
                            _process_start_ticks(ppid))


def rendezvous_dir():
    """A directory only this user can write to (0700, owned by us) under the temp directory."""
    d = os.path.join(tempfile.gettempdir(), "pilot_ot_%d" % os.getuid())
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError("rendezvous directory %s is not private to uid %d" % (d, os.getuid()))
    return d


def exchange_unique_id(rank, world, make_id, key=None, timeout=300.0, directory=None):
    """Rank 0 calls ``make_id()`` (-> bytes) and publishes it; every other rank returns the same bytes.

    Single-node rendezvous through an atomically renamed file in the temp directory -- the launcher only has to start
    the processes (RANK / WORLD_SIZE in the environment); no torch, no MPI, no TCP store."""
    directory = directory or rendezvous_dir()
    path = os.path.join(directory, "pilot_ot_uid_" + (key or rendezvous_key()))
    if rank == 0:
        uid = bytes(make_id())
        if len(uid) != _lib.UNIQUE_ID_BYTES:
            raise ValueError("unique id of %d bytes (expected %d)" % (len(uid), _lib.UNIQUE_ID_BYTES))
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "wb") as fh:
            fh.write(uid)
        os.replace(tmp, path)
        return uid, path
    t0 = time.monotonic()
    while True:
        try:
            with open(path, "rb") as fh:
                uid = fh.read()
            if len(uid) == _lib.UNIQUE_ID_BYTES:       # (written atomically by rename: anything else is not ours)
                return uid, path
        except OSError:
            pass
        if time.monotonic() - t0 > timeout:
            raise TimeoutError("rank %d/%d: no unique id at %s after %.0f s" % (rank, world, path, timeout))
        time.sleep(0.01)


class Comm:
    """RCCL communicator of one rank (one process per GPU); the calling thread's current device is the rank's GPU."""

    def __init__(self, rank, world, key=None):
        self.rank, self.world = int(rank), int(world)
        self.L = _lib.load()

        def make_id():
            buf = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
            _lib.check(self.L.pilot_ot_comm_unique_id(buf))
            return buf.raw

        uid, path = exchange_unique_id(self.rank, self.world, make_id, key=key)
        self.h = ctypes.c_void_p()
        _lib.check(self.L.pilot_ot_comm_init_rank(uid, self.world, self.rank, ctypes.byref(self.h)))
        self._scratch = ctypes.c_void_p()
        _lib.check(self.L.pilot_ot_dev_alloc(ctypes.byref(self._scratch), 64))
        self.barrier()                          # every rank has read the file once the first collective returns
        if self.rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass

    def info(self):
        """(n_ranks, rank) as RCCL itself reports them for this communicator."""
        n, r = ctypes.c_int(0), ctypes.c_int(-1)
        _lib.check(self.L.pilot_ot_comm_info(self.h, ctypes.byref(n), ctypes.byref(r)))
        return n.value, r.value

    def all_gather_rows(self, d_local, n_pad, N, d_stage, d_full, stream=None):
        _lib.check(self.L.pilot_ot_comm_all_gather_rows(self.h, d_local, int(n_pad), int(N), d_stage, d_full,
                                                        ctypes.c_void_p(stream) if stream else None))

    def all_reduce_max(self, value, stream=None):
        """max over ranks of a host float (through an 8-byte device scratch); synchronises the stream."""
        x = np.array([float(value)], dtype=np.float64)
        s = ctypes.c_void_p(stream) if stream else None
        _lib.check(self.L.pilot_ot_memcpy_h2d(self._scratch, x.ctypes.data, 8))
        _lib.check(self.L.pilot_ot_comm_all_reduce_max(self.h, self._scratch, 1, s))
        _lib.check(self.L.pilot_ot_stream_sync(s))
        _lib.check(self.L.pilot_ot_memcpy_d2h(x.ctypes.data, self._scratch, 8))
        return float(x[0])

    def barrier(self, stream=None):
        self.all_reduce_max(0.0, stream)

    def close(self):
        if self.h:
            self.L.pilot_ot_comm_destroy(self.h)
            self.h = ctypes.c_void_p()
        if self._scratch:
            self.L.pilot_ot_dev_free(self._scratch)
            self._scratch = ctypes.c_void_p()
