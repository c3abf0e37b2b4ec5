"""Thin object wrapper over the C-ABI: one HipEngine = one kzg_ctx = one MI355X.

All arguments and results are bytes in the wire layouts of include/kzg_mi355x.h (Fr 32 B big-endian, G1 affine
96 B / compressed 48 B).  This is the object `Client` drives; tests inject an oracle-backed stand-in with the same
methods to exercise the host logic on machines without a GPU."""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Sequence, Tuple

from . import _native
from ._native import KzgError

R_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def _root_of_unity(n: int) -> int:
    return pow(7, (R_MODULUS - 1) // n, R_MODULUS)


def lagrange_factor(i: int, machines_scale: int, tau_y: int) -> int:
    """L_i(tau_y) over the 2^machines_scale-point Y domain: the per-worker factor of the Pianist SRS slice
    U_{i,j} = tau_x^j L_i(tau_y) G.  Setup-time host arithmetic only."""
    m = 1 << machines_scale
    if m == 1:
        return 1
    wi = pow(_root_of_unity(m), i, R_MODULUS)
    if (tau_y - wi) % R_MODULUS == 0:
        return 1
    num = (pow(tau_y, m, R_MODULUS) - 1) % R_MODULUS
    inv = lambda v: pow(v % R_MODULUS, R_MODULUS - 2, R_MODULUS)  # noqa: E731
    return wi * inv(m) % R_MODULUS * num % R_MODULUS * inv(tau_y - wi) % R_MODULUS


class HipEngine:
    # rows from this length up are decoded and uploaded tile by tile (see _Staged); KZG_STREAM_MIN_LOG moves the threshold
    STREAM_MIN = 1 << int(os.environ.get("KZG_STREAM_MIN_LOG", "19"))
    STREAM_TILE = 1 << int(os.environ.get("KZG_STREAM_TILE_LOG", "18"))   # 2^18 elements: the decode pool's full thread count

    def __init__(self, device: int = 0, window: int = 0):
        self._lib = _native.load()
        h = ctypes.c_void_p()
        rc = self._lib.kzg_create(device, ctypes.byref(h))
        if rc != 0:
            raise KzgError(rc, "kzg_create failed: no usable gfx950 device (this build has no CPU fallback)")
        self._h = h
        self.device = device
        self.scale = self.machines_scale = 0
        self.verifier = None          # host-side pairing verifier (zkp_subnet_amd.verifier.Verifier)
        if window:
            self._chk(self._lib.kzg_set_window(self._h, window))

    # ------------------------------------------------------------------ plumbing
    def _chk(self, rc: int) -> None:
        if rc != 0:
            raise KzgError(rc, self._lib.kzg_last_error(self._h).decode(errors="replace"))

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.kzg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def runtime_info(self) -> Dict[str, int]:
        """What kzg_create measured: lanes of the context, how many of them really run concurrently (hardware queues),
        whether a HIP runtime was already live when the library was loaded, GPU_MAX_HW_QUEUES as seen (0 = unset)."""
        arr = (ctypes.c_int32 * 4)()
        self._chk(self._lib.kzg_runtime_info(self._h, arr))
        return {"lanes": arr[0], "lanes_concurrent": arr[1], "hip_live_at_load": bool(arr[2]), "hw_queues_env": arr[3]}

    @property
    def window(self) -> int:
        return self._lib.kzg_get_window(self._h)

    @property
    def window_offsets(self) -> List[int]:
        """Bit offset of every Pippenger window plus the closing 256 (table w holds 2^offset[w] * P)."""
        arr = (ctypes.c_int32 * 70)()
        nwin = self._lib.kzg_get_window_layout(self._h, arr, 70)
        if nwin < 0:
            raise KzgError(nwin, "no window layout yet (load an SRS first)")
        return list(arr[: nwin + 1])

    @property
    def srs_points(self) -> int:
        return self._lib.kzg_srs_points(self._h)

    @property
    def slice_len(self) -> int:
        return 1 << (self.scale - self.machines_scale)

    # ------------------------------------------------------------------ SRS
    def load_srs(self, points: bytes, scale: int, machines_scale: int, compressed: bool = False) -> None:
        """`points`: 96-byte uncompressed affine records, or 48-byte ZCash-compressed ones (`compressed=True`)."""
        if compressed:
            self._chk(self._lib.kzg_load_srs_compressed(self._h, points, len(points) // 48, scale, machines_scale))
        else:
            self._chk(self._lib.kzg_load_srs(self._h, points, len(points) // 96, scale, machines_scale))
        self.scale, self.machines_scale = scale, machines_scale
        self.verifier = None

    def load_srs_file(self, path: str, scale: int, machines_scale: int, compressed: bool = False) -> None:
        """The setup FILE the reference prover is started with (base/miner.py:75-84): read by the library with
        pread(2) straight into two pinned tiles (host read, upload and GPU decode overlap), never into Python memory."""
        self._chk(self._lib.kzg_load_srs_file(self._h, os.fsencode(path), int(compressed), scale, machines_scale))
        self.scale, self.machines_scale = scale, machines_scale
        self.verifier = None

    def load_srs_file_slices(self, path: str, scale: int, machines_scale: int, first_slice: int, slice_stride: int,
                             compressed: bool = False) -> None:
        """Only the slices one device of a multi-GPU host serves: resident slice k = file slice first_slice + k * stride
        (worker i = g mod G on device g).  pread touches just those byte ranges (kzg_load_srs_file_slices)."""
        self._chk(self._lib.kzg_load_srs_file_slices(self._h, os.fsencode(path), int(compressed), scale, machines_scale,
                                                     first_slice, slice_stride))
        self.scale, self.machines_scale = scale, machines_scale
        self.verifier = None

    def load_srs_file_range(self, path: str, first_point: int, n_points: int, compressed: bool = False) -> None:
        """One contiguous segment of a flat SRS file as a single resident slice (kzg_load_srs_file_range)."""
        scale = max(0, (n_points - 1).bit_length())
        self._chk(self._lib.kzg_load_srs_file_range(self._h, os.fsencode(path), int(compressed), first_point, n_points, scale))
        self.scale, self.machines_scale = scale, 0
        self.verifier = None

    def set_srs_subgroup_check(self, enable: bool) -> None:
        """Loaders test every SRS point for membership in G1 (default on).  False skips it for the NEXT load only (a file
        of established provenance); the library arms the check again after that load."""
        self._chk(self._lib.kzg_set_srs_subgroup_check(self._h, int(enable)))

    def load_stats(self) -> Dict[str, float]:
        """Seconds of the last successful SRS load: host copies, waits for upload + decode, window tables, total."""
        arr = (ctypes.c_double * 4)()
        self._chk(self._lib.kzg_get_load_stats(self._h, arr))
        return {"host_copy_s": arr[0], "gpu_wait_s": arr[1], "tables_s": arr[2], "total_s": arr[3]}

    def set_verifier_key(self, tau_g2_be192: bytes, li_g1_be96: bytes) -> None:
        """Verification key of a loaded SRS: [tau_x]_2 (uncompressed G2, 192 B) and [L_i(tau_y)]_1 per resident slice."""
        from .verifier import Verifier

        self.verifier = Verifier.from_points(tau_g2_be192, li_g1_be96)

    def gen_srs(self, tau_x: int, tau_y: int, scale: int, machines_scale: int,
                workers: Optional[Sequence[int]] = None, factors: Optional[Sequence[int]] = None) -> None:
        """Synthetic tau-derived SRS for the listed worker indices (default: all 2^machines_scale), slice k of
        the resident SRS = worker workers[k].  NOTE: resident slice index == position in `workers`.
        `factors` overrides the per-slice factor s0_k (slice k, point j = [s0_k tau_x^j] G): bench.py uses it to
        generate one SRS *segment* per rank (s0 = tau^(rank * n_local))."""
        if factors is not None:
            s0 = b"".join((f % R_MODULUS).to_bytes(32, "big") for f in factors)
        else:
            if workers is None:
                workers = range(1 << machines_scale)
            s0 = b"".join(lagrange_factor(i, machines_scale, tau_y).to_bytes(32, "big") for i in workers)
        self._chk(self._lib.kzg_gen_srs(self._h, (tau_x % R_MODULUS).to_bytes(32, "big"), s0, len(s0) // 32, scale,
                                        machines_scale))
        self.scale, self.machines_scale = scale, machines_scale
        from .verifier import Verifier

        self.verifier = Verifier.synthetic(tau_x % R_MODULUS, [int.from_bytes(s0[i:i + 32], "big")
                                                               for i in range(0, len(s0), 32)])

    def verify(self, i: int, proof48: bytes, alpha32: bytes, eval32: bytes, commitment48: bytes) -> bool:
        """Pairing check of one opening against resident slice i (host-side; reference neurons/validator.py:77-86)."""
        if self.verifier is None:
            raise NotImplementedError("no verifier key: call set_verifier_key() after load_srs()")
        return self.verifier.verify(i, proof48, alpha32, eval32, commitment48)

    def verify_batch(self, indices: Sequence[int], proofs48: Sequence[bytes], alpha32: bytes, evals32: Sequence[bytes],
                     commitments48: Sequence[bytes], threads: int = 16) -> bool:
        """All rows of a validator step (common alpha) in one pairing check; True only if every row is valid."""
        if self.verifier is None:
            raise NotImplementedError("no verifier key: call set_verifier_key() after load_srs()")
        return self.verifier.verify_batch(indices, proofs48, alpha32, evals32, commitments48, threads)

    def srs_read(self, first: int, count: int, window: int = 0, compressed: bool = False) -> bytes:
        out = ctypes.create_string_buffer((48 if compressed else 96) * count)
        fn = self._lib.kzg_srs_read_compressed if compressed else self._lib.kzg_srs_read
        self._chk(fn(self._h, window, first, count, out))
        return out.raw

    # ------------------------------------------------------------------ hot path
    def commit(self, i: int, row_be32: bytes, evaluation_form: bool = True) -> bytes:
        out = ctypes.create_string_buffer(48)
        self._chk(self._lib.kzg_commit(self._h, i, row_be32, len(row_be32) // 32, int(evaluation_form), out))
        return out.raw

    def open(self, i: int, row_be32: bytes, alpha_be32: bytes, evaluation_form: bool = True) -> Tuple[bytes, bytes]:
        ev, pf = ctypes.create_string_buffer(32), ctypes.create_string_buffer(48)
        self._chk(self._lib.kzg_open(self._h, i, row_be32, len(row_be32) // 32, int(evaluation_form), alpha_be32, ev, pf))
        return ev.raw, pf.raw

    def commit_open(self, i: int, row_be32: bytes, alpha_be32: bytes,
                    evaluation_form: bool = True) -> Tuple[bytes, bytes, bytes]:
        c, ev, pf = ctypes.create_string_buffer(48), ctypes.create_string_buffer(32), ctypes.create_string_buffer(48)
        self._chk(self._lib.kzg_commit_open(self._h, i, row_be32, len(row_be32) // 32, int(evaluation_form),
                                            alpha_be32, c, ev, pf))
        return c.raw, ev.raw, pf.raw

    # ---- the same three calls fed from the synapse's List[str] (reference neurons/miner.py:38-61): the text is decoded
    # by csrc/wire_py.c straight into the library's pinned staging buffer (no bytes object, no pageable bounce)
    class _Staged:
        """One pinned staging buffer of the library's pool holding the decoded polynomial; released on exit.  Concurrent
        requests (the axon's worker threads) each hold their own buffer, so their decodes and GPU calls overlap."""

        def __init__(self, eng: "HipEngine", poly: Sequence[str], tagged: bool = False):
            from . import codec

            if codec._wire is None:
                raise RuntimeError("zkp_subnet_amd._wire is not built: run `python -m zkp_subnet_amd.build`")
            self.eng, self.n = eng, len(poly)
            ptr, tok = ctypes.c_void_p(), ctypes.c_int(-1)
            eng._chk(eng._lib.kzg_staging_acquire(eng._h, 32 * max(self.n, 1), ctypes.byref(ptr), ctypes.byref(tok)))
            self.token = tok.value
            cap = 32 * max(self.n, 1)
            try:
                tile = max(HipEngine.STREAM_TILE, self.n >> 2)
                if not tagged and self.n >= HipEngine.STREAM_MIN and self.n % tile == 0:
                    # long rows of the fused call: decode in (at most four) tiles and start each tile's upload at once
                    # (kzg_staging_flush): the copy engine moves tile k while the pool decodes tile k + 1, and the compute
                    # call finds the row on the device.  Short rows: the per-tile calls would cost what the overlap gives.
                    # Tagged calls (the two-call route) decode in one shot: the tag -- hit or miss -- is only known at
                    # the end, a hit's upload is off the critical path anyway, and tiles cost the decode ~0.1 ms each
                    # (measured: profiles/r04_ab_streamed_upload.log).
                    got, self.tag = 0, None
                    for first in range(0, self.n, tile):
                        got += codec._wire.decode_fr_list_into(poly, ptr.value, cap, 0, first, tile)
                        eng._chk(eng._lib.kzg_staging_flush(eng._h, self.token, 32 * first, 32 * tile))
                elif tagged:   # one pass: base64 -> bytes in the pinned buffer + the 128-bit content tag of those bytes
                    got, self.tag = codec._wire.decode_fr_list_into_tagged(poly, ptr.value, cap)
                else:        # fused / one-shot calls have no use for the tag (it costs ~0.7 ns per element and thread)
                    got, self.tag = codec._wire.decode_fr_list_into(poly, ptr.value, cap), None
            except ValueError as e:
                eng._lib.kzg_staging_release(eng._h, self.token)
                raise codec.CodecError(str(e)) from e
            except BaseException:
                eng._lib.kzg_staging_release(eng._h, self.token)
                raise
            assert got == self.n
            self.row = ctypes.cast(ptr, ctypes.c_char_p)

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            self.eng._lib.kzg_staging_release(self.eng._h, self.token)
            return False

    def commit_list(self, i: int, poly: Sequence[str], evaluation_form: bool = True) -> bytes:
        out = ctypes.create_string_buffer(48)
        # tagged: the coefficient vector stays on the device for the worker_open that follows with the same row
        # (the unchanged reference miner's two-call route, neurons/miner.py:56-61)
        with HipEngine._Staged(self, poly, tagged=True) as st:
            self._chk(self._lib.kzg_commit_cached(self._h, i, st.row, st.n, int(evaluation_form), st.tag, out))
        return out.raw

    def open_list(self, i: int, poly: Sequence[str], alpha_be32: bytes, evaluation_form: bool = True) -> Tuple[bytes, bytes]:
        ev, pf = ctypes.create_string_buffer(32), ctypes.create_string_buffer(48)
        with HipEngine._Staged(self, poly, tagged=True) as st:
            self._chk(self._lib.kzg_open_cached(self._h, i, st.row, st.n, int(evaluation_form), st.tag, alpha_be32, ev, pf))
        return ev.raw, pf.raw

    def commit_open_list(self, i: int, poly: Sequence[str], alpha_be32: bytes,
                         evaluation_form: bool = True) -> Tuple[bytes, bytes, bytes]:
        c, ev, pf = ctypes.create_string_buffer(48), ctypes.create_string_buffer(32), ctypes.create_string_buffer(48)
        with HipEngine._Staged(self, poly) as st:
            self._chk(self._lib.kzg_commit_open(self._h, i, st.row, st.n, int(evaluation_form), alpha_be32, c, ev, pf))
        return c.raw, ev.raw, pf.raw

    def row_cache_stats(self) -> Tuple[int, int]:
        """(hits, misses) of the coefficient cache behind commit_list / open_list."""
        arr = (ctypes.c_uint64 * 2)()
        self._chk(self._lib.kzg_row_cache_stats(self._h, arr))
        return int(arr[0]), int(arr[1])

    def msm(self, scalars_be32: bytes, srs_offset: int = 0) -> bytes:
        out = ctypes.create_string_buffer(48)
        self._chk(self._lib.kzg_msm(self._h, scalars_be32, len(scalars_be32) // 32, srs_offset, out))
        return out.raw

    def msm_partial(self, scalars_be32: bytes, srs_offset: int = 0) -> bytes:
        out = ctypes.create_string_buffer(192)
        self._chk(self._lib.kzg_msm_partial(self._h, scalars_be32, len(scalars_be32) // 32, srs_offset, out))
        return out.raw

    def g1_sum(self, partials_xyzz192: bytes) -> bytes:
        out = ctypes.create_string_buffer(48)
        self._chk(self._lib.kzg_g1_sum(self._h, partials_xyzz192, len(partials_xyzz192) // 192, out))
        return out.raw

    def g1_sum_compressed(self, points_c48: bytes) -> bytes:
        """Sum of 48-byte compressed G1 points (Pianist master aggregation sum_i commit_i) -> 48 bytes."""
        out = ctypes.create_string_buffer(48)
        self._chk(self._lib.kzg_g1_sum_compressed(self._h, points_c48, len(points_c48) // 48, out))
        return out.raw

    def set_host_finish(self, enable: bool) -> None:
        """True (default): the result point's affine conversion + compression run on the host; False: on the GPU."""
        self._chk(self._lib.kzg_set_host_finish(self._h, int(enable)))

    def ntt(self, vals_be32: bytes, inverse: bool) -> bytes:
        buf = ctypes.create_string_buffer(vals_be32, len(vals_be32))
        self._chk(self._lib.kzg_ntt(self._h, buf, len(vals_be32) // 32, int(inverse)))
        return buf.raw

    def eval(self, coeffs_be32: bytes, x_be32: bytes) -> bytes:
        out = ctypes.create_string_buffer(32)
        self._chk(self._lib.kzg_eval(self._h, coeffs_be32, len(coeffs_be32) // 32, x_be32, out))
        return out.raw

    def ntt_eval(self, vals_be32: bytes, inverse: bool, x_be32: bytes) -> bytes:
        """y = (NTT / inverse NTT of vals)(x) in one call: the coefficients never leave the device."""
        out = ctypes.create_string_buffer(32)
        self._chk(self._lib.kzg_ntt_eval(self._h, vals_be32, len(vals_be32) // 32, int(inverse), x_be32, out))
        return out.raw

    def ntt_eval_list(self, poly: Sequence[str], inverse: bool, x_be32: bytes) -> bytes:
        """The same, fed from the wire text (decoded straight into a pinned staging buffer)."""
        out = ctypes.create_string_buffer(32)
        with HipEngine._Staged(self, poly) as st:
            self._chk(self._lib.kzg_ntt_eval(self._h, st.row, st.n, int(inverse), x_be32, out))
        return out.raw

    # ------------------------------------------------------------------ device-resident inputs
    def upload_fr(self, slot: int, be32: bytes, to_mont: bool) -> None:
        self._chk(self._lib.kzg_upload_fr(self._h, slot, be32, len(be32) // 32, int(to_mont)))

    def msm_resident(self, slot: int, n: int, srs_offset: int = 0) -> bytes:
        out = ctypes.create_string_buffer(48)
        self._chk(self._lib.kzg_msm_resident(self._h, slot, n, srs_offset, out))
        return out.raw

    def msm_partial_resident(self, slot: int, n: int, srs_offset: int = 0) -> bytes:
        out = ctypes.create_string_buffer(192)
        self._chk(self._lib.kzg_msm_partial_resident(self._h, slot, n, srs_offset, out))
        return out.raw

    def msm_submit(self, slot: int, n: int, srs_offset: int = 0, partial: bool = False) -> Tuple[int, bool]:
        """Queue an MSM on a free lane and return its ticket; at most two tickets may be outstanding (E_BUSY)."""
        t = ctypes.c_int(-1)
        self._chk(self._lib.kzg_msm_submit(self._h, slot, n, srs_offset, int(partial), ctypes.byref(t)))
        return t.value, partial

    def msm_wait(self, ticket: Tuple[int, bool]) -> bytes:
        out = ctypes.create_string_buffer(192 if ticket[1] else 48)
        self._chk(self._lib.kzg_msm_wait(self._h, ticket[0], out))
        return out.raw

    def msm_cancel(self, ticket) -> None:
        """Give up an outstanding ticket (msm_submit / msm_sharded_begin): drains its lane and frees it."""
        self._chk(self._lib.kzg_msm_cancel(self._h, ticket[0] if isinstance(ticket, tuple) else ticket))

    def msm_partial_resident_dev(self, slot: int, n: int, srs_offset: int, dev_ptr: int) -> None:
        """The 192-byte partial goes to device address `dev_ptr` (e.g. `tensor.data_ptr()`); complete on return."""
        self._chk(self._lib.kzg_msm_partial_resident_dev(self._h, slot, n, srs_offset, ctypes.c_void_p(dev_ptr)))

    def g1_sum_dev(self, dev_ptr: int, count: int) -> bytes:
        """Sum of `count` partials at device address `dev_ptr` (every writer must have completed) -> 48 bytes."""
        out = ctypes.create_string_buffer(48)
        self._chk(self._lib.kzg_g1_sum_dev(self._h, ctypes.c_void_p(dev_ptr), count, out))
        return out.raw

    def msm_sharded_begin(self, slot: int, n: int, srs_offset: int, dev_ptr: int, consumer_stream: int) -> int:
        """Queue this rank's partial MSM; its 192 bytes land at `dev_ptr` and `consumer_stream` (a raw hipStream_t, e.g.
        `torch.cuda.current_stream().cuda_stream`) waits for them on the device.  Returns a ticket; nothing blocks."""
        t = ctypes.c_int(-1)
        self._chk(self._lib.kzg_msm_sharded_begin(self._h, slot, n, srs_offset, ctypes.c_void_p(dev_ptr),
                                                  ctypes.c_void_p(consumer_stream), ctypes.byref(t)))
        return t.value

    def msm_sharded_finish(self, ticket: int, dev_ptr: int, count: int, producer_stream: int) -> bytes:
        """Sum the `count` gathered partials at `dev_ptr` once `producer_stream` has reached this point -> 48 bytes."""
        out = ctypes.create_string_buffer(48)
        self._chk(self._lib.kzg_msm_sharded_finish(self._h, ticket, ctypes.c_void_p(dev_ptr), count,
                                                   ctypes.c_void_p(producer_stream), out))
        return out.raw

    # ------------------------------------------------------------------ the library's own collective (kzg_comm_*)
    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId through the library: ONE rank calls it and hands the 128 bytes to the others (rendezvous is the
        caller's: a store, a file, a socket)."""
        lib = _native.load()
        out = ctypes.create_string_buffer(128)
        rc = lib.kzg_comm_unique_id(out)
        if rc != 0:
            raise KzgError(rc, lib.kzg_last_error(None).decode(errors="replace"))
        return out.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int, timeout_ms: int = 0, init_timeout_ms: int = 0) -> None:
        """Joins the `world`-rank RCCL communicator on this context's GPU (collective: returns when every rank has joined
        and a first checked all_gather has connected them).  `init_timeout_ms` > 0 bounds that rendezvous in the library
        (kzg_comm_init_bounded): peers that never arrive raise KzgError(E_COMM) and the engine stays usable.
        `timeout_ms` is the per-call budget of every later sharded MSM."""
        if len(unique_id) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        self._chk(self._lib.kzg_comm_init_bounded(self._h, unique_id, rank, world, int(init_timeout_ms)))
        if timeout_ms:
            self._chk(self._lib.kzg_comm_set_timeout(self._h, timeout_ms))

    def comm_set_timeout(self, timeout_ms: int) -> None:
        self._chk(self._lib.kzg_comm_set_timeout(self._h, timeout_ms))

    def comm_selftest(self) -> None:
        """One small all_gather with checked content on the communicator (collective): raises KzgError(E_COMM) when the
        ranks cannot actually exchange bytes."""
        self._chk(self._lib.kzg_comm_selftest(self._h))

    def comm_destroy(self) -> None:
        self._chk(self._lib.kzg_comm_destroy(self._h))

    def comm_info(self) -> Dict[str, int]:
        arr = (ctypes.c_int32 * 4)()
        self._chk(self._lib.kzg_comm_info(self._h, arr))
        v = int(arr[2])
        return {"rank": arr[0], "world": arr[1], "rccl_version_code": v, "broken": bool(arr[3]),
                "rccl_version": f"{v // 10000}.{v // 100 % 100}.{v % 100}" if v else None}

    def msm_sharded(self, slot: int, n: int, srs_offset: int = 0) -> bytes:
        """This rank's SRS segment -> partial -> ncclAllGather on the lane's own stream -> sum: 48 bytes, the same on
        every rank.  One host wait; nothing but the library between the partial and the sum."""
        out = ctypes.create_string_buffer(48)
        self._chk(self._lib.kzg_msm_sharded(self._h, slot, n, srs_offset, out))
        return out.raw

    def commit_open_resident(self, i: int, slot: int, T: int, alpha_be32: bytes,
                             evaluation_form: bool = True) -> Tuple[bytes, bytes, bytes]:
        c, ev, pf = ctypes.create_string_buffer(48), ctypes.create_string_buffer(32), ctypes.create_string_buffer(48)
        self._chk(self._lib.kzg_commit_open_resident(self._h, i, slot, T, int(evaluation_form), alpha_be32, c, ev, pf))
        return c.raw, ev.raw, pf.raw

    def ntt_resident(self, slot: int, n: int, inverse: bool) -> None:
        self._chk(self._lib.kzg_ntt_resident(self._h, slot, n, int(inverse)))

    # ------------------------------------------------------------------ measurement
    def set_profiling(self, level) -> None:
        """0 / False: off.  1 / True: HIP events around every stage (calls serialise on one lane).  2: around the
        accumulate kernel only (two events per launch, no serialisation)."""
        self._chk(self._lib.kzg_set_profiling(self._h, int(level)))

    def timings(self) -> Dict[str, float]:
        arr = (ctypes.c_float * len(_native.TIMING_NAMES))()
        self._chk(self._lib.kzg_get_timings(self._h, arr, len(arr)))
        return {k: float(v) for k, v in zip(_native.TIMING_NAMES, arr)}

    def msm_plan(self, n: int) -> Dict[str, int]:
        arr = (ctypes.c_int32 * 4)()
        self._chk(self._lib.kzg_msm_plan(self._h, n, arr))
        return {"chunk": arr[0], "lanes": arr[1], "buckets": arr[2], "windows": arr[3]}

    def calibrate(self, waves_per_simd: int = 2) -> Dict[str, float]:
        """The v_mad_u64_u32 issue rate of THIS GPU right now (kzg_calibrate): what bounds k_msm_accumulate."""
        arr = (ctypes.c_double * 6)()
        self._chk(self._lib.kzg_calibrate(self._h, waves_per_simd, arr))
        return {"ns_per_mad_per_simd": arr[0], "gmad_per_s": arr[1], "memtime_ticks_per_ns": arr[2], "kernel_ms": arr[3],
                "simds": int(arr[4]), "ticks_per_mad_of_one_wave": arr[5], "waves_per_simd": waves_per_simd}

    # ------------------------------------------------------------------ unit-op hooks (parity tests)
    def test_field(self, field: int, op: int, a_be: bytes, b_be: bytes) -> bytes:
        w = 48 if field == 0 else 32
        out = ctypes.create_string_buffer(len(a_be))
        self._chk(self._lib.kzg_test_field(self._h, field, op, a_be, b_be, out, len(a_be) // w))
        return out.raw

    def test_g1(self, op: int, a_be96: bytes, b_be96: bytes) -> bytes:
        out = ctypes.create_string_buffer(len(a_be96))
        self._chk(self._lib.kzg_test_g1(self._h, op, a_be96, b_be96, out, len(a_be96) // 96))
        return out.raw
