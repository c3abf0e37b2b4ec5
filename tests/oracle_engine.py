"""TEST-ONLY engine with the HipEngine method surface, backed by the CPU oracle.  Lets the host logic
(Client, Miner, validator mirror, distributed sharding) run on machines without a GPU, and plays the role of the
"reference CPU prover" in the config-1 plumbing test.  Never imported by the product package."""
from __future__ import annotations

from oracle import bls12_381 as o
from oracle import cpu as oc


class OracleEngine:
    def __init__(self):
        self.srs = b""
        self.scale = self.machines_scale = 0
        self.tau_x = self.tau_y = None
        self.workers = None

    def close(self):
        pass

    @property
    def slice_len(self):
        return 1 << (self.scale - self.machines_scale)

    def load_srs(self, points, scale, machines_scale, compressed=False):
        if compressed:
            points = b"".join(o.g1_to_be96(o.g1_decompress(points[48 * k:48 * k + 48])) for k in range(len(points) // 48))
        self.srs, self.scale, self.machines_scale = points, scale, machines_scale

    def gen_srs(self, tau_x, tau_y, scale, machines_scale, workers=None):
        if workers is None:
            workers = range(1 << machines_scale)
        self.workers = list(workers)
        self.tau_x, self.tau_y = tau_x % o.R, tau_y % o.R
        self.srs = b"".join(oc.srs_gen(self.tau_x.to_bytes(32, "big"), self.tau_y.to_bytes(32, "big"), scale,
                                       machines_scale, i) for i in self.workers)
        self.scale, self.machines_scale = scale, machines_scale

    def _slice(self, i, n):
        T = self.slice_len
        if n > T or (i + 1) * T * 96 > len(self.srs):
            raise ValueError("bad slice")
        return self.srs[96 * i * T: 96 * (i * T + n)]

    def srs_read(self, first, count, window=0):
        assert window == 0
        return self.srs[96 * first: 96 * (first + count)]

    def commit(self, i, row, evaluation_form=True):
        return oc.commit(self._slice(i, len(row) // 32), row, evaluation_form, threads=4)

    def open(self, i, row, alpha, evaluation_form=True):
        return oc.open_(self._slice(i, len(row) // 32), row, alpha, evaluation_form, threads=4)

    def commit_open(self, i, row, alpha, evaluation_form=True):
        ev, pf = self.open(i, row, alpha, evaluation_form)
        return self.commit(i, row, evaluation_form), ev, pf

    def msm(self, scalars, srs_offset=0):
        n = len(scalars) // 32
        return oc.msm(self.srs[96 * srs_offset: 96 * (srs_offset + n)], scalars, threads=4)

    # partial sums travel as 192 opaque bytes; here: affine be96 padded (only this engine reads them back)
    def msm_partial(self, scalars, srs_offset=0):
        pt = o.g1_decompress(self.msm(scalars, srs_offset))
        return o.g1_to_be96(pt) + bytes(96)

    def g1_sum(self, partials):
        return oc.g1_sum(b"".join(partials[i:i + 96] for i in range(0, len(partials), 192)))

    def g1_sum_compressed(self, points_c48):
        pts = [o.g1_decompress(points_c48[k:k + 48]) for k in range(0, len(points_c48), 48)]
        return oc.g1_sum(b"".join(o.g1_to_be96(p) for p in pts))

    def ntt(self, vals, inverse):
        return oc.fr_ntt(vals, inverse)

    def eval(self, coeffs, x):
        return oc.fr_eval(coeffs, x)

    def verify(self, i, proof48, alpha32, eval32, commitment48):
        """Algebraic pairing-equation check with the known trapdoor."""
        w = self.workers[i]
        try:
            c, pi = o.g1_decompress(commitment48), o.g1_decompress(proof48)
        except AssertionError:
            return False
        return o.verify_trapdoor(self.tau_x, self.tau_y, self.machines_scale, w, c, pi,
                                 int.from_bytes(alpha32, "big"), int.from_bytes(eval32, "big"))
