"""`Client`: same method surface as the `fourier.Client` the reference miner / validator drive
(construction: reference base/miner.py:73-84, base/validator.py:80-91; calls: neurons/miner.py:39,48 and
neurons/validator.py:59-104), backed by the in-process HIP library instead of a spawned Rust binary over
localhost HTTP.  Every method returns a `Response` usable exactly like the reference uses the HTTP one:

    with client.worker_commit(i, poly) as response:
        if response.status_code != 200: ...
        response.json().get("commitment")

Errors never escape as exceptions (the reference checks status_code, neurons/miner.py:40-44): bad input -> 400,
prover failure -> 500, not implemented -> 501.
"""
from __future__ import annotations

import logging
import os
import secrets
from typing import Any, Dict, List, Optional, Sequence

from . import codec
from ._native import KZG_E_ARG, KZG_E_POINT, KZG_E_SCALAR, KzgError

R_MODULUS = codec.R_MODULUS
log = logging.getLogger("zkp_subnet_amd.client")


class Response:
    """Minimal stand-in for the HTTP response object of the reference client."""

    def __init__(self, status_code: int, body: Optional[Dict[str, Any]] = None):
        self.status_code = status_code
        self._body = body or {}

    def json(self) -> Dict[str, Any]:
        return self._body

    def __enter__(self) -> "Response":
        return self

    def __exit__(self, *exc) -> bool:
        return False

    def close(self) -> None:
        pass


def _guard(fn):
    def wrapped(self, *a, **kw):
        try:
            if self.engine is None:
                return Response(503, {"error": "prover not started"})
            return Response(200, fn(self, *a, **kw))
        except codec.CodecError as e:
            return Response(400, {"error": str(e)})
        except KzgError as e:
            bad_input = e.code in (KZG_E_ARG, KZG_E_SCALAR, KZG_E_POINT)
            return Response(400 if bad_input else 500, {"error": str(e)})
        except NotImplementedError as e:
            return Response(501, {"error": str(e)})
        except Exception as e:  # never let the axon thread die (reference neurons/miner.py:133-135)
            return Response(500, {"error": f"{type(e).__name__}: {e}"})

    wrapped.__name__ = fn.__name__
    wrapped.__doc__ = fn.__doc__
    return wrapped


class Client:
    """Drop-in for fourier.Client.  `port` and `bin` are accepted and ignored (there is no child process);
    `setup_path` names a file holding the 2^scale-point SRS as uncompressed affine G1 points (x||y, 96 B each,
    big-endian) or, with `uncompressed=False`, as 48-byte ZCash-compressed points (decompressed on the GPU).
    A missing setup file is an error, as it is for the reference prover.  A synthetic tau-derived SRS (generated on
    the GPU; its trapdoor is a public function of `seed`, so openings against it can be forged by anyone) is only
    built on explicit request -- `synthetic=True` or an explicit `seed` -- for tests and benches (mirrors
    `fourier setup --generate-setup`, reference tests/conftest.py:50-65), and is logged loudly."""

    def __init__(self, port: int = 1337, bin: str = "", uncompressed: bool = True, setup_path: str = "",
                 precompute_path: str = "", engine: Any = None, device: int = 0, seed: Optional[int] = None,
                 workers: Optional[Sequence[int]] = None, synthetic: Optional[bool] = None):
        self.port, self.bin, self.uncompressed = port, bin, uncompressed
        self.setup_path, self.precompute_path = setup_path, precompute_path
        self.engine = engine
        self._own_engine = engine is None
        self.device = device
        self.seed = seed
        self.synthetic = (seed is not None) if synthetic is None else bool(synthetic)
        self.workers = list(workers) if workers is not None else None
        self.scale = self.machines_scale = 0

    # ------------------------------------------------------------------ lifecycle (base/miner.py:82-84,155,181)
    def start(self, scale: int = 18, machines_scale: int = 8) -> None:
        if scale < machines_scale:
            raise ValueError("scale must be >= machines_scale")
        self.scale, self.machines_scale = scale, machines_scale
        if self.engine is None:
            from .engine import HipEngine  # raises loudly without the HIP library / a gfx950 device

            self.engine = HipEngine(self.device)
            info = self.engine.runtime_info()
            if 0 < info["lanes_concurrent"] < info["lanes"]:
                log.warning("only %d of the context's %d lanes run concurrently on this GPU (GPU_MAX_HW_QUEUES=%s%s): results "
                            "are unaffected, but concurrent requests overlap less.  Export GPU_MAX_HW_QUEUES=8 before the "
                            "process's first HIP call.", info["lanes_concurrent"], info["lanes"],
                            info["hw_queues_env"] or "unset", ", HIP was already initialised when the library was loaded"
                            if info["hip_live_at_load"] else "")
        if self.setup_path and os.path.exists(self.setup_path):
            rec = 96 if self.uncompressed else 48      # the reference's `uncompressed` flag (base/miner.py:77)
            if os.path.getsize(self.setup_path) % rec:
                raise ValueError(f"setup file must be a whole number of {rec}-byte G1 points "
                                 f"(uncompressed={self.uncompressed})")
            rec_points = os.path.getsize(self.setup_path) // rec
            T = 1 << (scale - machines_scale)
            file_slices = rec_points // T if rec_points % T == 0 else 0
            # a client that serves only SOME worker indices (one device of a MultiDeviceClient: i = g mod G) loads only their
            # slices when they form the progression first, first + stride, ... over the file's slices
            if self.workers is not None and not self.workers:      # more devices than worker rows: this one serves none
                self._slice_of = {}
                return
            prog = self._progression(file_slices) if self.workers is not None else None
            load_slices = getattr(self.engine, "load_srs_file_slices", None)
            load_file = getattr(self.engine, "load_srs_file", None)
            if prog is not None and load_slices is not None:
                load_slices(self.setup_path, scale, machines_scale, prog[0], prog[1], compressed=not self.uncompressed)
                self._slice_of = {w: k for k, w in enumerate(self.workers)}
            elif load_file is not None:                # HipEngine: the library reads the file and streams it itself
                load_file(self.setup_path, scale, machines_scale, compressed=not self.uncompressed)
                self._slice_of = None
            else:                                      # an injected engine without a file loader (tests)
                with open(self.setup_path, "rb") as f:
                    self.engine.load_srs(f.read(), scale, machines_scale, compressed=not self.uncompressed)
                self._slice_of = None
            vk_path = self.setup_path + ".vk"   # 192 B [tau_x]_2 (uncompressed) + one 96 B [L_i(tau_y)]_1 per slice
            if os.path.exists(vk_path) and hasattr(self.engine, "set_verifier_key"):
                with open(vk_path, "rb") as f:
                    vk = f.read()
                li = vk[192:]
                if self._slice_of is not None:         # the key's factors in RESIDENT slice order
                    li = b"".join(li[96 * w:96 * w + 96] for w in self.workers)
                self.engine.set_verifier_key(vk[:192], li)
        else:
            if not self.synthetic:
                if self._own_engine:
                    self.engine.close()
                    self.engine = None
                raise FileNotFoundError(
                    f"setup file {self.setup_path!r} not found: generate one with `python -m zkp_subnet_amd.setup_cli "
                    "setup --generate-setup ...`.  A synthetic SRS (public trapdoor: forgeable openings) is only built "
                    "when asked for with Client(synthetic=True) or an explicit seed (tests / benches).")
            seed = self.seed if self.seed is not None else 0
            log.warning("SYNTHETIC SRS from public seed %d: its trapdoor is computable by anyone -- openings can be "
                        "forged.  Tests and benches only; production needs a setup file (setup_path).", seed)
            tau_x, tau_y = derive_taus(seed)
            self.tau_x, self.tau_y = tau_x, tau_y
            if self.workers is not None and not self.workers:
                # a device of a MultiDeviceClient with MORE devices than worker rows serves no row: nothing to generate, and
                # every worker index answers "no resident slice" (400) here
                self._slice_of = {}
                return
            self.engine.gen_srs(tau_x, tau_y, scale, machines_scale, self.workers)
            self._slice_of = {w: k for k, w in enumerate(self.workers)} if self.workers is not None else None

    def _progression(self, file_slices: int):
        """(first, stride) when self.workers is exactly first, first + stride, ... below file_slices; else None."""
        w = self.workers
        if not w or file_slices <= 0:
            return None
        first = w[0]
        stride = (w[1] - w[0]) if len(w) > 1 else max(1, file_slices)
        if stride <= 0 or list(w) != list(range(first, file_slices, stride)):
            return None
        return first, stride

    def stop(self) -> None:
        if self.engine is not None and self._own_engine:
            self.engine.close()
        if self._own_engine:
            self.engine = None

    def _slice(self, i: int) -> int:
        i = int(i)
        if i < 0 or i >= (1 << self.machines_scale):
            raise codec.CodecError(f"worker index {i} outside [0, 2^{self.machines_scale})")
        if self._slice_of is not None:
            if i not in self._slice_of:
                raise codec.CodecError(f"worker index {i} has no resident SRS slice")
            return self._slice_of[i]
        return i

    # ------------------------------------------------------------------ miner side (neurons/miner.py:38-61)
    @_guard
    def worker_commit(self, i: int, poly: Sequence[str]):
        fast = getattr(self.engine, "commit_list", None)      # HipEngine: text decoded straight into pinned staging
        c = fast(self._slice(i), poly, True) if fast and codec._wire else \
            self.engine.commit(self._slice(i), codec.fr_list_to_be32(poly), True)
        return {"commitment": codec.g1_to_b64(c)}

    @_guard
    def worker_open(self, i: int, poly: Sequence[str], x: str):
        fast = getattr(self.engine, "open_list", None)
        ev, pf = fast(self._slice(i), poly, codec.fr_to_be32(x), True) if fast and codec._wire else \
            self.engine.open(self._slice(i), codec.fr_list_to_be32(poly), codec.fr_to_be32(x), True)
        return {"eval": codec.be32_to_fr(ev), "proof": codec.g1_to_b64(pf)}

    @_guard
    def worker_commit_and_open(self, i: int, poly: Sequence[str], x: str):
        """Fused extension (one upload, one IFFT): what Miner.rpc_commit_and_open needs (neurons/miner.py:56-61)."""
        fast = getattr(self.engine, "commit_open_list", None)
        c, ev, pf = fast(self._slice(i), poly, codec.fr_to_be32(x), True) if fast and codec._wire else \
            self.engine.commit_open(self._slice(i), codec.fr_list_to_be32(poly), codec.fr_to_be32(x), True)
        return {"commitment": codec.g1_to_b64(c), "eval": codec.be32_to_fr(ev), "proof": codec.g1_to_b64(pf)}

    @_guard
    def aggregate_commitments(self, commitments: Sequence[str]):
        """Pianist master aggregation: sum_i commit_i of the worker rows' commitments = the commitment of the whole
        bivariate polynomial (reference neurons/validator.py:196-198 distributes the rows; README.md:38 names the
        aggregation as the next milestone).  Points are decompressed and summed on the GPU."""
        raw = b"".join(codec.g1_from_b64(c) for c in commitments)
        return {"commitment": codec.g1_to_b64(self.engine.g1_sum_compressed(raw))}

    # ------------------------------------------------------------------ validator side (neurons/validator.py:58-104)
    @_guard
    def worker_verify(self, i: int, proof: str, alpha: str, eval: str, commitment: str):
        verify = getattr(self.engine, "verify", None)
        if verify is None:
            raise NotImplementedError("this engine has no verifier")
        ok = verify(self._slice(i), codec.g1_from_b64(proof), codec.fr_to_be32(alpha), codec.fr_to_be32(eval),
                    codec.g1_from_b64(commitment))
        return {"valid": bool(ok)}

    @_guard
    def worker_verify_batch(self, indices: Sequence[int], proofs: Sequence[str], alpha: str, evals: Sequence[str],
                            commitments: Sequence[str], threads: int = 16):
        """Extension: every row of a validator step (they share alpha, neurons/validator.py:106-120) in ONE pairing check
        on a random linear combination of the rows.  {"valid": True} only when every row verifies; False does not say
        which row failed -- `validator.verify_all` then falls back to `worker_verify` row by row."""
        vb = getattr(self.engine, "verify_batch", None)
        if vb is None:
            raise NotImplementedError("this engine has no batch verifier")
        ok = vb([self._slice(i) for i in indices], [codec.g1_from_b64(p) for p in proofs], codec.fr_to_be32(alpha),
                [codec.fr_to_be32(e) for e in evals], [codec.g1_from_b64(c) for c in commitments], threads)
        return {"valid": bool(ok)}

    @_guard
    def fft(self, poly: Sequence[str], left: bool = True, inverse: bool = False):
        n = len(poly)
        want = 1 << (self.scale - self.machines_scale) if left else 1 << self.machines_scale
        if self.scale and n != want:
            raise codec.CodecError(f"fft(left={left}) expects {want} elements, got {n}")
        out = self.engine.ntt(codec.fr_list_to_be32(poly), bool(inverse))
        return {"poly": codec.be32_to_fr_list(out)}

    @_guard
    def eval(self, poly: Sequence[str], x: str):
        y = self.engine.eval(codec.fr_list_to_be32(poly), codec.fr_to_be32(x))
        return {"y": codec.be32_to_fr(y)}

    @_guard
    def fft_eval(self, poly: Sequence[str], x: str, left: bool = True, inverse: bool = True):
        """Extension: eval(fft(poly, left, inverse), x) in ONE call -- the validator's per-row challenge step (reference
        neurons/validator.py:115-118 makes the two calls; the 2^16 coefficients then cross the text codec twice)."""
        n = len(poly)
        want = 1 << (self.scale - self.machines_scale) if left else 1 << self.machines_scale
        if self.scale and n != want:
            raise codec.CodecError(f"fft(left={left}) expects {want} elements, got {n}")
        fast = getattr(self.engine, "ntt_eval_list", None)
        xb = codec.fr_to_be32(x)
        if fast and codec._wire:
            y = fast(poly, bool(inverse), xb)
        else:
            y = self.engine.eval(self.engine.ntt(codec.fr_list_to_be32(poly), bool(inverse)), xb)
        return {"y": codec.be32_to_fr(y)}

    @_guard
    def random_poly(self):
        """Bivariate polynomial as 2^machines_scale rows of 2^(scale-machines_scale) Fr (neurons/validator.py:67-75).
        Uniform on [0, r): getrandom + rejection, generated and encoded natively (csrc/wire_py.c) -- 2^24 strings at
        mainnet scale, which a Python loop needs ~40 s for, longer than the 30 s challenge deadline."""
        rows, T = 1 << self.machines_scale, 1 << (self.scale - self.machines_scale)
        if codec._wire is not None:
            return {"poly": codec._wire.random_fr_rows(rows, T)}
        return {"poly": [[codec.be32_to_fr(_random_fr()) for _ in range(T)] for _ in range(rows)]}

    @_guard
    def random_point(self):
        if codec._wire is not None:
            return {"point": codec._wire.random_fr_rows(1, 1)[0][0]}
        return {"point": codec.be32_to_fr(_random_fr())}


def _random_fr() -> bytes:
    return (secrets.randbelow(R_MODULUS)).to_bytes(32, "big")


def derive_taus(seed: int):
    """Deterministic (tau_x, tau_y) for synthetic SRS generation: SHA-256 counter stream reduced mod r."""
    import hashlib

    def h(tag: bytes) -> int:
        sb = seed.to_bytes(max(8, (seed.bit_length() + 7) // 8), "big")   # 8 bytes for seeds < 2^64 (fixture-stable)
        v = int.from_bytes(hashlib.sha256(b"kzg-mi355x-srs" + tag + sb).digest(), "big")
        return v % (R_MODULUS - 2) + 2

    return h(b"x"), h(b"y")
