"""Rank-aware placement of the 5-D vectors in MI355X device memory.

Measured (profiles/r03_placement_pair_map_*.txt, r03_placement_counters.txt, r03_arena_probe.txt): the device memory
of an MI355X falls into THREE classes, each one contiguous third of the physical address space (a fresh process is
handed 47 + 47 + 47 chunks of 2 GiB in that order) - consistent with the three ranks (stack IDs) of its eight 12-high
HBM3E stacks.  Two vectors swept in lock-step (x and p; x and exp_avg_sq) stream at 6.4 - 6.5 TB/s when, at equal
offsets, they lie in different classes, and at 5.4 - 5.5 TB/s when they share one: same request counts, same TLB
behaviour, uniformly over all 128 L2 channels, but 6x the cycles stalled for DRAM read credits - the two streams open
different rows of the same banks.  hipMalloc decides where an allocation lands from its own history, so the product
does not guess: it takes physical chunks (dfx_arena_*, the HIP virtual-memory API), measures each chunk's class with
the real sweep kernel, and maps the chunks behind contiguous virtual ranges so that vector j's i-th chunk and vector
k's i-th chunk never share a class.  No copy, no extra resident memory: chunks the layout does not use go back to the
driver at once.

A process keeps one pool per device: chunks of freed vectors stay classified, so a second solve of the same size
needs no new measurement.  The price of the first: the probes (1.4 ms per chunk and known class, 0.1 - 0.4 s in all)
plus the driver's own price for memory that was used before (~21 ms per GiB, the same for hipMalloc).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import time

import torch

from . import _cabi as cabi

GIB = 1 << 30
CHUNK_BYTES = int(os.environ.get("DFX_ARENA_CHUNK_MIB", 2048)) << 20        # physical chunk
RESERVE_BYTES = int(os.environ.get("DFX_ARENA_RESERVE_MIB", 3072)) << 20    # device memory left to everybody else
MIN_VECTOR_BYTES = int(os.environ.get("DFX_ARENA_MIN_VECTOR_MIB", 8192)) << 20
# a pair of chunks is "the same class" when the probe runs below SAME_CLASS x the rate of pairs known to share a
# class (the two halves of one chunk, median over a few chunks).  Measured (profiles/r03_arena_purity_*.txt): halves of
# one chunk 5.3 - 5.7 TB/s, two chunks of one class 5.45 - 5.9, chunks of different classes 6.1 - 6.45
SAME_CLASS = 1.08
CALIBRATION_CHUNKS = 5
MAX_CHUNKS = 192            # an arena never holds more (288 GiB / 2 GiB = 144 on an MI355X)


class _View:
    """One contiguous virtual range over an ordered list of chunks; exposes __cuda_array_interface__ so that torch
    wraps it without copying and keeps this object (and through it the pool's bookkeeping) alive."""

    def __init__(self, pool, chunks, ptr, n_float):
        self.pool, self.chunks, self.ptr = pool, list(chunks), ptr
        self.__cuda_array_interface__ = {"shape": (n_float,), "typestr": "<f4", "data": (ptr, False), "version": 2}

    def __del__(self):
        try:
            self.pool._give_back(self)
        except Exception as e:       # interpreter shutdown: the driver reclaims everything anyway
            if os is not None and os.environ.get("DFX_ARENA_DEBUG"):
                print(f"datafusion_amd.placement: giving back a vector failed: {e!r}")


class _Raw:
    """A bare device range for torch.as_tensor (no ownership)."""

    def __init__(self, ptr, n_float):
        self.__cuda_array_interface__ = {"shape": (n_float,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class DevicePool:
    """Classified physical chunks of one GPU: one arena; every chunk of it is either mapped in a live vector or on
    the free list of its class."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.arena = None                # dfx_arena handle
        self.n_chunks = 0                # chunks ever created in this arena (ids are 0 .. n_chunks - 1)
        self.cls = {}                    # chunk id -> class id (chunks still owned)
        self.free = {}                   # class id -> [chunk ids]
        self.same_rate = None            # GB/s of the probe on two halves of one chunk (a pair sharing a class)
        self.live_views = 0
        self.exhausted = False           # the last growth took all it may and a layout still had conflicts
        self.lock = threading.RLock()
        self.log = []                    # what happened, for bench / tests
        self._probe_k = None

    # ---- arena plumbing (thin wrappers: the CPU tests replace them) -----------------------
    def _create(self, n_more: int):
        """n_more new chunks -> their ids."""
        lib = cabi.lib()
        with torch.cuda.device(self.index):
            if self.arena is None:
                h = C.c_void_p()
                cabi.check(lib.dfx_arena_create(CHUNK_BYTES, n_more, C.byref(h)), "dfx_arena_create")
                cb = C.c_int64()
                cabi.check(lib.dfx_arena_info(h, C.byref(cb), None, None), "dfx_arena_info")
                if cb.value != CHUNK_BYTES:
                    lib.dfx_arena_destroy(h)
                    raise cabi.DfxError(f"device granularity does not divide the chunk size ({cb.value} != {CHUNK_BYTES})")
                self.arena = h
            else:
                cabi.check(lib.dfx_arena_grow(self.arena, n_more), "dfx_arena_grow")
        ids = list(range(self.n_chunks, self.n_chunks + n_more))
        self.n_chunks += n_more
        return ids

    def _destroy_arena(self):
        if self.arena is not None:
            assert self.live_views == 0
            with torch.cuda.device(self.index):
                cabi.check(cabi.lib().dfx_arena_destroy(self.arena), "dfx_arena_destroy")
        self.arena, self.n_chunks, self.cls, self.free, self.exhausted = None, 0, {}, {}, False

    def _map(self, ids):
        arr = (C.c_int32 * len(ids))(*ids)
        ptr = C.c_void_p()
        with torch.cuda.device(self.index):
            cabi.check(cabi.lib().dfx_arena_map(self.arena, arr, len(ids), C.byref(ptr)), "dfx_arena_map")
        return ptr.value

    def _unmap(self, ptr):
        with torch.cuda.device(self.index):
            cabi.check(cabi.lib().dfx_arena_unmap(self.arena, C.c_void_p(ptr)), "dfx_arena_unmap")

    def _release(self, ids):
        if ids:
            arr = (C.c_int32 * len(ids))(*ids)
            cabi.check(cabi.lib().dfx_arena_release_chunks(self.arena, arr, len(ids)), "dfx_arena_release_chunks")
            for g in ids:
                self.cls.pop(g, None)

    def _allocatable_chunks(self) -> int:
        torch.cuda.empty_cache()
        room = int(max(0, (torch.cuda.mem_get_info(self.index)[0] - RESERVE_BYTES) // CHUNK_BYTES))
        return max(0, min(room, MAX_CHUNKS - len(self.cls)))

    def _device_ctx(self):
        return torch.cuda.device(self.index)

    def _wrap(self, view, n_float):
        return torch.as_tensor(view, device=self.device)

    def _give_back(self, view: _View):
        with self.lock:
            self._unmap(view.ptr)
            self.live_views -= 1
            for g in view.chunks:
                self.free.setdefault(self.cls[g], []).append(g)

    def trim(self):
        """Return every free chunk to the driver (their classification is forgotten)."""
        with self.lock:
            if self.arena is None:
                return
            if self.live_views == 0:
                self._destroy_arena()
            else:
                self._release([g for lst in self.free.values() for g in lst])
                self.free = {}

    # ---- measurement --------------------------------------------------------------------
    def _probe_kernels(self):
        """The probe is the real pass: the factored XP sweep with alpha = beta = 0 over zeroed factors, on a geometry
        known to show the effect (1 MiB slabs of z4 x 256 x 256), streaming HALF a chunk of each operand."""
        if self._probe_k is None:
            from .kernels import HipKernels
            n_t = (CHUNK_BYTES // 2) // (32 << 20)
            k = HipKernels(n_t, 32, 4, 256, 256, 4, self.device)
            k.new_factored_residual()
            k.ra.zero_(), k.rb.zero_()
            self._probe_k = k
        return self._probe_k

    def _rate(self, xptr, pptr, reps=2):
        """GB/s of the probe with x at xptr and p at pptr (half a chunk each)."""
        k = self._probe_kernels()
        x = torch.as_tensor(_Raw(xptr, k.N), device=self.device).view(k.shape)
        p = torch.as_tensor(_Raw(pptr, k.N), device=self.device).view(k.shape)
        st = torch.cuda.current_stream(self.device)
        best = 1e30
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            k.sweep(cabi.SWEEP_XP, x, None, p, factored=True)
            e1.record(st)
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return 16.0 * k.N / best / 1e6

    def _pair_rate(self, a, b, at, off, reps=2):
        """The probe on chunks a and b mapped at at[a], at[b]: x in a at byte offset `off` (0 or half a chunk), p in b
        at `off` - or, for a == b, x in the first half and p in the second (a pair that shares a class for sure)."""
        if a == b:
            return self._rate(at[a], at[a] + off, reps)
        return self._rate(at[a] + off, at[b] + off, reps)

    def _zero(self, ptr, n_float):
        torch.as_tensor(_Raw(ptr, n_float), device=self.device).zero_()

    def _classify(self, new_ids):
        """Class of every chunk of new_ids: a chunk shares the class of a reference chunk when the probe on the pair runs
        as slowly as on the two halves of ONE chunk (which share a class by construction; measured once per arena).
        References: one free chunk per class already known, tried most-recently-matched first (neighbouring chunks
        mostly share a class); a chunk that is fast against every reference founds a new class.  -> probes made."""
        refs = {c: lst[0] for c, lst in self.free.items() if lst}
        order = sorted(refs)                              # classes, most recently matched first
        ids = list(refs.values()) + list(new_ids)
        base = self._map(ids)
        at = {g: base + i * CHUNK_BYTES for i, g in enumerate(ids)}
        half = CHUNK_BYTES // 2
        probes = 0
        rates = []
        try:
            for g in new_ids:                             # fresh physical memory: make what the probe reads finite
                self._zero(at[g], CHUNK_BYTES // 4)
            if self.same_rate is None:
                own = sorted(self._pair_rate(g, g, at, half, reps=3) for g in new_ids[:CALIBRATION_CHUNKS])
                self.same_rate = own[len(own) // 2]
                probes += len(own)
            thr = SAME_CLASS * self.same_rate
            n_cls = (max(self.cls.values()) + 1) if self.cls else 0
            for g in new_ids:
                found = None
                for c in order:
                    r = self._pair_rate(refs[c], g, at, 0)
                    probes += 1
                    if thr * 0.97 < r < thr * 1.04:       # too close to call: the best of more runs, on the chunks' other
                        r = max(r, self._pair_rate(refs[c], g, at, half, reps=4))   # halves (noise only lowers a rate)
                        probes += 1
                    rates.append(r)
                    if r < thr:
                        found = c
                        break
                if found is None:
                    found, n_cls = n_cls, n_cls + 1
                    refs[found] = g
                else:
                    order.remove(found)
                order.insert(0, found)
                self.cls[g] = found
        finally:
            self._unmap(base)
        self.log.append({"new_chunks": len(new_ids), "probes": probes, "same_class_rate_GBps": self.same_rate,
                         "probe_rates_GBps": [min(rates), max(rates)] if rates else None,
                         "class_sequence": "".join(chr(65 + min(self.cls[g], 25)) for g in new_ids)})
        return probes

    # ---- layout -------------------------------------------------------------------------
    @staticmethod
    def layout(counts: dict, n_vectors: int, per_vector: int):
        """Choose, for every chunk position i < per_vector, the classes of the n_vectors chunks mapped there, all
        different when the counts allow it: greedily the classes with the most chunks left (optimal for two vectors).
        -> list over positions of tuples of class ids, or None when there are not enough chunks at all."""
        left = {c: n for c, n in counts.items() if n > 0}
        if sum(left.values()) < n_vectors * per_vector:
            return None
        out = []
        for _ in range(per_vector):
            pick = []
            for _j in range(n_vectors):
                # the fullest class not yet used at this position; if every class is used, the fullest one
                cand = [c for c in left if left[c] > 0 and c not in pick] or [c for c in left if left[c] > 0]
                c = max(cand, key=lambda c: left[c])
                left[c] -= 1
                pick.append(c)
            out.append(tuple(pick))
        return out

    @staticmethod
    def conflicts(lay) -> int:
        """Positions at which two vectors share a class."""
        return sum(1 for t in lay if len(set(t)) < len(t))

    # ---- the public call ----------------------------------------------------------------
    def alloc_vectors(self, n_vectors: int, n_float: int):
        """n_vectors fp32 vectors of n_float elements whose chunks differ in class position by position.
        -> (list of 1-D torch tensors, info dict), or None when the pool cannot serve the request (vectors of an
        earlier request are alive and the free chunks do not suffice, or the device has too little memory left): the
        caller then allocates through torch."""
        nbytes = 4 * n_float
        per = -(-nbytes // CHUNK_BYTES)
        need = n_vectors * per
        t0 = time.perf_counter()
        with self.lock, self._device_ctx():
            def plan():
                counts = {c: len(v) for c, v in self.free.items() if v}
                return counts, (self.layout(counts, n_vectors, per) if counts else None)

            counts, lay = plan()
            probes = created = 0
            if lay is None or (self.conflicts(lay) and not self.exhausted):
                if self.live_views:
                    return None
                if sum(counts.values()) + self._allocatable_chunks() < need:
                    return None
                # grow in batches until the layout has no conflicts (in a fresh process the driver hands out one class
                # after the other, 47 chunks each; later in its life in a scattered order) or the device is full
                while lay is None or self.conflicts(lay):
                    room = self._allocatable_chunks()
                    if room <= 0:
                        break
                    have = sum(len(v) for v in self.free.values())
                    batch = min(room, max(need - have, per, 8, have // 2))
                    ids = self._create(batch)
                    created += batch
                    probes += self._classify(ids)
                    for g in ids:
                        self.free.setdefault(self.cls[g], []).append(g)
                    counts, lay = plan()
                if lay is None:
                    self.trim()
                    return None
                # everything that may be taken has been looked at and two vectors still meet in one class somewhere:
                # later requests that the free chunks can serve take them as they are (until trim())
                self.exhausted = bool(self.conflicts(lay))
            vec_chunks = [[] for _ in range(n_vectors)]
            for pos in lay:
                for j, c in enumerate(pos):
                    vec_chunks[j].append(self.free[c].pop())
            out = []
            for ch in vec_chunks:
                view = _View(self, ch, self._map(ch), n_float)
                self.live_views += 1
                out.append(self._wrap(view, n_float))
                del view
            # chunks the layout does not use go back to the driver now: the rest of the program needs the memory
            if not int(os.environ.get("DFX_ARENA_KEEP_SPARE", "0")):
                self._release([g for lst in self.free.values() for g in lst])
                self.free = {}
            info = {"vectors": n_vectors, "chunks_per_vector": per, "chunk_GiB": CHUNK_BYTES / GIB,
                    "positions_sharing_a_class": self.conflicts(lay), "chunks_created": created, "probes": probes,
                    "class_counts": {str(c): n for c, n in counts.items()}, "seconds": time.perf_counter() - t0}
            self.log.append(info)
            return out, info


_pools = {}
_pools_lock = threading.Lock()


def pool(device) -> DevicePool:
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    with _pools_lock:
        if idx not in _pools:
            _pools[idx] = DevicePool(torch.device("cuda", idx))
        return _pools[idx]


def available_bytes(device) -> int:
    """Device memory a solve could still get: what the driver reports free plus the classified chunks this process's
    pool holds on its free lists."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    free = torch.cuda.mem_get_info(idx)[0]
    with _pools_lock:
        p = _pools.get(idx)
    if p is not None:
        with p.lock:
            free += CHUNK_BYTES * sum(len(v) for v in p.free.values())
    return int(free)


def trim(device=None) -> None:
    """Give every pooled free chunk back to the driver (all devices when device is None)."""
    with _pools_lock:
        items = list(_pools.items())
    for idx, p in items:
        if device is None or torch.device(device).index in (None, idx):
            p.trim()
