"""`MultiDeviceClient`: one host process, G MI355X -- one `Client` (= one kzg_ctx, its own lanes and streams) per GPU,
requests routed by worker index, host threads only, NO collective.

This is the in-process form of the reference's only distribution scheme: Pianist rows are independent, the validator
sends row i to miner i (reference neurons/validator.py:194-222) and each miner process builds ONE prover client
(base/miner.py:73-84).  A host with several GPUs can instead serve

  * a miner's rows:      worker index i  ->  device devices[i mod G]   (`worker_commit` / `worker_open` /
                         `worker_commit_and_open`: same signatures and responses as `Client`, so `Miner(client=...)`
                         takes it unchanged), and `commit_and_open_rows` fans a batch of rows out over the devices;
  * a validator's step:  the 2^machines_scale `eval(fft(poly[i], inverse), alpha)` rows of `generate_challenge`
                         (neurons/validator.py:106-120) spread round-robin over the devices (`fft_eval_rows`).

Every device holds exactly the setup it serves: from a setup file each context reads, checks and tabulates only the slices
of its own worker indices (`kzg_load_srs_file_slices`: mainnet 24 / 8 on G GPUs = 34 / G GB of tables and ~1 / G of the
start time each); with a synthetic seed each context generates only the slices routed to it.

`SegmentedMsm` (below) is the other way to use the GPUs of one process: ONE multi-scalar multiplication over a flat SRS cut
into G contiguous segments, one per GPU (BASELINE.json configs[3] from behind the one-client seam; `kzg_multi_msm`).  SURVEY 8b proposed
`kzg_create(device_count, device_ids)`: the C-ABI has that router too (`kzg_multi_*`, csrc/multi_host.cpp, for native
callers working on bytes); this class is the same routing one level up, where the text codec and the Client surface live."""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence

from .client import Client, Response


class MultiDeviceClient:
    def __init__(self, devices: Sequence[int], port: int = 1337, bin: str = "", uncompressed: bool = True,
                 setup_path: str = "", precompute_path: str = "", seed: Optional[int] = None,
                 synthetic: Optional[bool] = None, engines: Optional[Sequence[object]] = None):
        if not devices:
            raise ValueError("MultiDeviceClient needs at least one device")
        if engines is not None and len(engines) != len(devices):
            raise ValueError("one injected engine per device")
        self.devices = list(devices)
        self._kw = dict(port=port, bin=bin, uncompressed=uncompressed, setup_path=setup_path,
                        precompute_path=precompute_path, seed=seed, synthetic=synthetic)
        self._engines = list(engines) if engines is not None else None
        self.clients: List[Client] = []
        self.scale = self.machines_scale = 0
        self._rr = 0
        self._pool: Optional[ThreadPoolExecutor] = None

    # ------------------------------------------------------------------ lifecycle
    def start(self, scale: int = 18, machines_scale: int = 8) -> None:
        G = len(self.devices)
        M = 1 << machines_scale
        self.scale, self.machines_scale = scale, machines_scale
        started: List[Client] = []
        try:
            for g, dev in enumerate(self.devices):
                kw = dict(self._kw)
                # only the slices this device serves: generated (synthetic) or read from the setup file (kzg_load_srs_file_slices)
                workers = [i for i in range(M) if i % G == g]
                c = Client(device=dev, workers=workers, engine=self._engines[g] if self._engines else None, **kw)
                c.start(scale, machines_scale)
                started.append(c)
        except BaseException:
            for c in started:
                c.stop()
            raise
        self.clients = started
        self._pool = ThreadPoolExecutor(max_workers=4 * G, thread_name_prefix="kzg-multi")   # four lanes per context

    def stop(self) -> None:
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        for c in self.clients:
            c.stop()
        self.clients = []

    # ------------------------------------------------------------------ routing
    def device_of(self, i: int) -> int:
        return self.devices[int(i) % len(self.devices)]

    def _for(self, i: int) -> Client:
        if not self.clients:
            return _NOT_STARTED
        return self.clients[int(i) % len(self.clients)]

    def _any(self) -> Client:
        if not self.clients:
            return _NOT_STARTED
        self._rr = (self._rr + 1) % len(self.clients)        # benign race: any client will do
        return self.clients[self._rr]

    # miner side (reference neurons/miner.py:38-61): same calls, routed by worker index
    def worker_commit(self, i: int, poly: Sequence[str]):
        return self._for(i).worker_commit(i, poly)

    def worker_open(self, i: int, poly: Sequence[str], x: str):
        return self._for(i).worker_open(i, poly, x)          # same device as the commit: its row cache serves the pair

    def worker_commit_and_open(self, i: int, poly: Sequence[str], x: str):
        return self._for(i).worker_commit_and_open(i, poly, x)

    def commit_and_open_rows(self, indices: Sequence[int], polys: Sequence[Sequence[str]], x: str) -> List[Response]:
        """Pianist rows of one challenge, all devices at once: row k runs on the device of indices[k]; responses in input
        order.  Host threads only (ctypes releases the GIL inside every call); nothing is exchanged between devices."""
        if len(indices) != len(polys):
            raise ValueError("one polynomial per index")
        if self._pool is None:
            return [_NOT_STARTED.worker_commit_and_open(i, p, x) for i, p in zip(indices, polys)]
        return list(self._pool.map(lambda t: self.worker_commit_and_open(t[0], t[1], x), zip(indices, polys)))

    # validator side (reference neurons/validator.py:58-120)
    def worker_verify(self, i: int, proof: str, alpha: str, eval: str, commitment: str):
        return self._for(i).worker_verify(i, proof, alpha, eval, commitment)      # host-side pairing: any would do

    def worker_verify_batch(self, indices, proofs, alpha, evals, commitments, threads: int = 16):
        """One batched check PER DEVICE, verdicts AND-ed: with a synthetic setup each context holds the slices and
        verifier-key factors of its own workers only (i mod G == g), so row k goes to the client that serves indices[k].
        Any non-200 answer of a group is returned as it is (the caller then falls back to row-by-row checks)."""
        if not self.clients:
            return _NOT_STARTED.worker_verify_batch(indices, proofs, alpha, evals, commitments, threads)
        if not (len(indices) == len(proofs) == len(evals) == len(commitments)):
            return Response(400, {"error": "worker_verify_batch: ragged input"})
        G = len(self.clients)
        groups = {}
        for k, i in enumerate(indices):
            groups.setdefault(int(i) % G, []).append(k)
        valid = True
        for g, ks in sorted(groups.items()):
            r = self.clients[g].worker_verify_batch([indices[k] for k in ks], [proofs[k] for k in ks], alpha,
                                                    [evals[k] for k in ks], [commitments[k] for k in ks], threads)
            if r.status_code != 200:
                return r
            valid = valid and r.json().get("valid") is True
            if not valid:
                break
        return Response(200, {"valid": valid})

    def fft(self, poly: Sequence[str], left: bool = True, inverse: bool = False):
        return self._any().fft(poly, left, inverse)

    def eval(self, poly: Sequence[str], x: str):
        return self._any().eval(poly, x)

    def fft_eval(self, poly: Sequence[str], x: str, left: bool = True, inverse: bool = True):
        return self._any().fft_eval(poly, x, left, inverse)

    def fft_eval_rows(self, polys: Sequence[Sequence[str]], x: str, left: bool = True, inverse: bool = True) -> List[Response]:
        """The per-row challenge step eval(fft(poly[i], left, inverse), x) for many rows, row k on device k mod G."""
        if self._pool is None:
            return [_NOT_STARTED.fft_eval(p, x, left, inverse) for p in polys]
        G = len(self.clients)
        return list(self._pool.map(lambda t: self.clients[t[0] % G].fft_eval(t[1], x, left, inverse), enumerate(polys)))

    def random_poly(self):
        return self._any().random_poly()

    def random_point(self):
        return self._any().random_point()

    def aggregate_commitments(self, commitments: Sequence[str]):
        return self._any().aggregate_commitments(commitments)     # a sum of points: no slice involved, any device


class SegmentedMsm:
    """ONE MSM over the G GPUs of this process (C-ABI `kzg_multi_*`, SEGMENTS layout): a flat SRS of `n_points` is cut into G
    contiguous segments, segment g resident on devices[g]; `msm` runs the G partial MSMs concurrently, each on its own
    device and lane, and sums the G 192-byte partials once.  No collective: the whole exchange is G x 192 bytes through the
    host.  Results are bit-identical to `HipEngine.msm` over the same points on one device.  (The reference has no
    device-level distribution at all -- one prover client per process, base/miner.py:73-84 -- this is BASELINE.json
    configs[3] reachable from one process; with one process per GPU the same step is `kzg_msm_sharded` over RCCL.)"""

    def __init__(self, devices: Sequence[int]):
        import ctypes

        from . import _native

        self._ct, self._lib = ctypes, _native.load()
        self.devices = list(devices)
        ids = (ctypes.c_int * len(self.devices))(*self.devices)
        h = ctypes.c_void_p()
        rc = self._lib.kzg_multi_create(len(self.devices), ids, ctypes.byref(h))
        if rc != 0:
            raise _native.KzgError(rc, self._lib.kzg_multi_last_error(None).decode(errors="replace"))
        self._h = h
        self.n_points = 0

    def _chk(self, rc: int) -> None:
        if rc != 0:
            from ._native import KzgError

            raise KzgError(rc, self._lib.kzg_multi_last_error(self._h).decode(errors="replace"))

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.kzg_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shard(self, n_points: int, g: int):
        from .distributed import shard_range

        return shard_range(n_points, g, len(self.devices))

    def gen_srs(self, tau: int, n_points: int) -> None:
        """Synthetic flat SRS, point j = [tau^j] G (tests / benches: public trapdoor)."""
        from .engine import R_MODULUS

        s0 = b"".join(pow(tau, self.shard(n_points, g)[0], R_MODULUS).to_bytes(32, "big") for g in range(len(self.devices)))
        self._chk(self._lib.kzg_multi_gen_srs_segments(self._h, (tau % R_MODULUS).to_bytes(32, "big"), s0, n_points))
        self.n_points = n_points

    def load_srs_file(self, path: str, n_points: int, compressed: bool = False) -> None:
        """Device g reads only file points [lo_g, lo_g + n_g) (pread of that byte range)."""
        import os

        self._chk(self._lib.kzg_multi_load_srs_file_segments(self._h, os.fsencode(path), int(compressed), n_points))
        self.n_points = n_points

    def segment(self, g: int):
        arr = (self._ct.c_uint64 * 2)()
        self._chk(self._lib.kzg_multi_segment(self._h, g, arr))
        return int(arr[0]), int(arr[1])

    def msm(self, scalars_be32: bytes, srs_offset: int = 0) -> bytes:
        out = self._ct.create_string_buffer(48)
        self._chk(self._lib.kzg_multi_msm(self._h, scalars_be32, len(scalars_be32) // 32, srs_offset, out))
        return out.raw

    def upload(self, slot: int, scalars_be32: bytes, srs_offset: int = 0) -> None:
        self._chk(self._lib.kzg_multi_upload_fr(self._h, slot, scalars_be32, len(scalars_be32) // 32, srs_offset))

    def msm_resident(self, slot: int) -> bytes:
        out = self._ct.create_string_buffer(48)
        self._chk(self._lib.kzg_multi_msm_resident(self._h, slot, out))
        return out.raw


_NOT_STARTED = Client(engine=None)      # engine None -> every call answers 503 "prover not started", as Client does
_NOT_STARTED._own_engine = False
