"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- pure-Python big-int oracle.

Slow but unambiguous restatement of the arithmetic the reference delegates to the
external ``fourier`` prover (reference call sites: ``neurons/miner.py:39,48``
worker_commit / worker_open; ``neurons/validator.py:59-104`` fft / random_poly /
worker_verify / random_point / eval).  PARITY UNPINNED w.r.t. the real prover
except for the Fr wire encoding + eval KAT (reference ``tests/test_miner.py:33-55``).

Conventions (documented as residual risk in DESIGN.md):
  * Fr on the wire: unpadded std-alphabet base64 of 32 bytes big-endian (KAT-pinned).
  * G1 on the wire: base64 of the 48-byte ZCash compressed encoding.
  * domain of size n: w_n = 7^((r-1)/n); natural (not bit-reversed) order;
    forward transform  eval_i = sum_j c_j w^(ij); inverse has the 1/n factor.
  * worker_commit / worker_open take the row in *evaluation form* on the left (X)
    domain and commit to / open IFFT_left(row) (reference
    ``neurons/validator.py:115-118`` + ``tests/test_validator.py:124-163``).
  * Pianist worker slice U_{i,j} = tau_x^j * L_i(tau_y) * G   (SURVEY.md 3.5).
"""
from __future__ import annotations

import base64
from typing import List, Optional, Sequence, Tuple

# ----------------------------------------------------------------------------- constants
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
BLS_X = -0xD201000000010000
G1_X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
CURVE_B = 4
FR_GENERATOR = 7
FR_TWO_ADICITY = 32

Affine = Optional[Tuple[int, int]]  # None = point at infinity
Jac = Tuple[int, int, int]  # Z == 0 -> infinity


# ----------------------------------------------------------------------------- Fr
def fr_inv(a: int) -> int:
    return pow(a % R, R - 2, R)


def root_of_unity(n: int) -> int:
    """Primitive n-th root of unity in Fr, n a power of two <= 2^32."""
    assert n >= 1 and n & (n - 1) == 0 and n <= 1 << FR_TWO_ADICITY
    return pow(FR_GENERATOR, (R - 1) // n, R)


def dft_naive(a: Sequence[int], inverse: bool = False) -> List[int]:
    n = len(a)
    w = root_of_unity(n)
    if inverse:
        w = fr_inv(w)
    out = []
    for i in range(n):
        wi = pow(w, i, R)
        acc, x = 0, 1
        for j in range(n):
            acc = (acc + a[j] * x) % R
            x = x * wi % R
        out.append(acc)
    if inverse:
        ninv = fr_inv(n)
        out = [v * ninv % R for v in out]
    return out


def ntt(a: Sequence[int], inverse: bool = False) -> List[int]:
    """Iterative radix-2 Cooley-Tukey, natural order in and out."""
    n = len(a)
    assert n & (n - 1) == 0
    a = list(a)
    if n == 1:
        return a
    logn = n.bit_length() - 1
    for i in range(n):
        j = int(format(i, f"0{logn}b")[::-1], 2)
        if i < j:
            a[i], a[j] = a[j], a[i]
    w_n = root_of_unity(n)
    if inverse:
        w_n = fr_inv(w_n)
    length = 2
    while length <= n:
        w_len = pow(w_n, n // length, R)
        half = length // 2
        tw = [1] * half
        for k in range(1, half):
            tw[k] = tw[k - 1] * w_len % R
        for s in range(0, n, length):
            for k in range(half):
                u = a[s + k]
                v = a[s + k + half] * tw[k] % R
                a[s + k] = (u + v) % R
                a[s + k + half] = (u - v) % R
        length <<= 1
    if inverse:
        ninv = fr_inv(n)
        a = [v * ninv % R for v in a]
    return a


def poly_eval(coeffs: Sequence[int], x: int) -> int:
    """Coefficient-form Horner (what the reference KAT pins)."""
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


def poly_quotient(coeffs: Sequence[int], alpha: int) -> Tuple[int, List[int]]:
    """(y, q) with f(X) - y = (X - alpha) q(X); len(q) = len(f) - 1."""
    n = len(coeffs)
    if n == 0:
        return 0, []
    q = [0] * (n - 1)
    acc = coeffs[n - 1]
    for j in range(n - 1, 0, -1):
        q[j - 1] = acc
        acc = (acc * alpha + coeffs[j - 1]) % R
    return acc, q


# ----------------------------------------------------------------------------- Fp / G1
def fp_inv(a: int) -> int:
    return pow(a % P, P - 2, P)


def fp_sqrt(a: int) -> Optional[int]:
    s = pow(a, (P + 1) // 4, P)  # p = 3 mod 4
    return s if s * s % P == a % P else None


def is_on_curve(pt: Affine) -> bool:
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - CURVE_B) % P == 0


G1: Affine = (G1_X, G1_Y)
JAC_INF: Jac = (1, 1, 0)


def to_jac(pt: Affine) -> Jac:
    return JAC_INF if pt is None else (pt[0], pt[1], 1)


def to_affine(pt: Jac) -> Affine:
    X, Y, Z = pt
    if Z == 0:
        return None
    zi = fp_inv(Z)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def jac_double(pt: Jac) -> Jac:
    X, Y, Z = pt
    if Z == 0 or Y == 0:
        return JAC_INF
    A = X * X % P
    B = Y * Y % P
    C = B * B % P
    D = 2 * ((X + B) * (X + B) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def jac_add(p1: Jac, p2: Jac) -> Jac:
    X1, Y1, Z1 = p1
    X2, Y2, Z2 = p2
    if Z1 == 0:
        return p2
    if Z2 == 0:
        return p1
    Z1Z1 = Z1 * Z1 % P
    Z2Z2 = Z2 * Z2 % P
    U1 = X1 * Z2Z2 % P
    U2 = X2 * Z1Z1 % P
    S1 = Y1 * Z2 * Z2Z2 % P
    S2 = Y2 * Z1 * Z1Z1 % P
    if U1 == U2:
        if S1 == S2:
            return jac_double(p1)
        return JAC_INF
    H = (U2 - U1) % P
    Rr = (S2 - S1) % P
    HH = H * H % P
    HHH = H * HH % P
    V = U1 * HH % P
    X3 = (Rr * Rr - HHH - 2 * V) % P
    Y3 = (Rr * (V - X3) - S1 * HHH) % P
    Z3 = Z1 * Z2 * H % P
    return (X3, Y3, Z3)


def jac_neg(pt: Jac) -> Jac:
    return (pt[0], (-pt[1]) % P, pt[2])


def g1_add(a: Affine, b: Affine) -> Affine:
    return to_affine(jac_add(to_jac(a), to_jac(b)))


def g1_neg(a: Affine) -> Affine:
    return None if a is None else (a[0], (-a[1]) % P)


def g1_mul(pt: Affine, k: int) -> Affine:
    k %= R
    acc = JAC_INF
    base = to_jac(pt)
    while k:
        if k & 1:
            acc = jac_add(acc, base)
        base = jac_double(base)
        k >>= 1
    return to_affine(acc)


def batch_to_affine(pts: Sequence[Jac]) -> List[Affine]:
    """Montgomery's trick: one inversion for the whole batch."""
    n = len(pts)
    prefix = [1] * (n + 1)
    for i, (_, _, Z) in enumerate(pts):
        prefix[i + 1] = prefix[i] * (Z if Z else 1) % P
    inv = fp_inv(prefix[n])
    out: List[Affine] = [None] * n
    for i in range(n - 1, -1, -1):
        X, Y, Z = pts[i]
        if Z == 0:
            continue
        zi = inv * prefix[i] % P
        inv = inv * Z % P
        zi2 = zi * zi % P
        out[i] = (X * zi2 % P, Y * zi2 * zi % P)
    return out


class FixedBase:
    """8-bit windowed fixed-base table for many multiples of one point."""

    def __init__(self, base: Affine = G1, window: int = 8):
        self.w = window
        self.nwin = (255 + window - 1) // window
        self.table: List[List[Affine]] = []
        cur = to_jac(base)
        for _ in range(self.nwin):
            row_j = [JAC_INF]
            acc = JAC_INF
            for _d in range(1, 1 << window):
                acc = jac_add(acc, cur)
                row_j.append(acc)
            self.table.append(batch_to_affine(row_j))
            for _s in range(window):
                cur = jac_double(cur)

    def mul_jac(self, k: int) -> Jac:
        k %= R
        acc = JAC_INF
        mask = (1 << self.w) - 1
        for i in range(self.nwin):
            d = (k >> (i * self.w)) & mask
            if d:
                acc = jac_add(acc, to_jac(self.table[i][d]))
        return acc

    def mul(self, k: int) -> Affine:
        return to_affine(self.mul_jac(k))

    def mul_many(self, ks: Sequence[int]) -> List[Affine]:
        return batch_to_affine([self.mul_jac(k) for k in ks])


_G1_TABLE: Optional[FixedBase] = None


def g1_table() -> FixedBase:
    global _G1_TABLE
    if _G1_TABLE is None:
        _G1_TABLE = FixedBase(G1)
    return _G1_TABLE


# ----------------------------------------------------------------------------- serialisation
def g1_compress(pt: Affine) -> bytes:
    """ZCash 48-byte compressed G1: flags = compressed(0x80) | infinity(0x40) | y-sign(0x20)."""
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80
    if y > (P - 1) // 2:
        b[0] |= 0x20
    return bytes(b)


def g1_decompress(b: bytes) -> Affine:
    assert len(b) == 48 and b[0] & 0x80, "not a compressed G1 encoding"
    if b[0] & 0x40:
        assert b[0] == 0xC0 and not any(b[1:]), "bad infinity encoding"
        return None
    sign = bool(b[0] & 0x20)
    x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big")
    assert x < P
    y = fp_sqrt((x * x * x + CURVE_B) % P)
    assert y is not None, "x not on curve"
    if (y > (P - 1) // 2) != sign:
        y = P - y
    return (x, y)


def g1_to_be96(pt: Affine) -> bytes:
    """Uncompressed affine x||y, 2 x 48 B big-endian; infinity = 96 zero bytes."""
    if pt is None:
        return bytes(96)
    return pt[0].to_bytes(48, "big") + pt[1].to_bytes(48, "big")


def g1_from_be96(b: bytes) -> Affine:
    assert len(b) == 96
    if not any(b):
        return None
    return (int.from_bytes(b[:48], "big"), int.from_bytes(b[48:], "big"))


def fr_to_b64(v: int) -> str:
    return base64.b64encode((v % R).to_bytes(32, "big")).decode().rstrip("=")


def fr_from_b64(s: str) -> int:
    raw = base64.b64decode(s + "=" * (-len(s) % 4))
    if len(raw) != 32:
        raise ValueError("Fr must decode to 32 bytes")
    v = int.from_bytes(raw, "big")
    if v >= R:
        raise ValueError("non-canonical Fr")
    return v


def g1_to_b64(pt: Affine) -> str:
    return base64.b64encode(g1_compress(pt)).decode()


def g1_from_b64(s: str) -> Affine:
    return g1_decompress(base64.b64decode(s))


def fr_to_be32(vals: Sequence[int]) -> bytes:
    return b"".join((v % R).to_bytes(32, "big") for v in vals)


def fr_from_be32(b: bytes) -> List[int]:
    return [int.from_bytes(b[i : i + 32], "big") for i in range(0, len(b), 32)]


# ----------------------------------------------------------------------------- MSM
def msm_naive(points: Sequence[Affine], scalars: Sequence[int]) -> Affine:
    acc = JAC_INF
    for pt, k in zip(points, scalars):
        if pt is None or k % R == 0:
            continue
        acc = jac_add(acc, to_jac(g1_mul(pt, k)))
    return to_affine(acc)


def msm_pippenger(points: Sequence[Affine], scalars: Sequence[int], c: int = 8) -> Affine:
    """Unsigned-window bucket method; independent of the GPU's signed/precomputed variant."""
    nwin = (255 + c - 1) // c
    total = JAC_INF
    for w in range(nwin - 1, -1, -1):
        for _ in range(c):
            total = jac_double(total)
        buckets = [JAC_INF] * (1 << c)
        for pt, k in zip(points, scalars):
            d = ((k % R) >> (w * c)) & ((1 << c) - 1)
            if d and pt is not None:
                buckets[d] = jac_add(buckets[d], to_jac(pt))
        run = JAC_INF
        acc = JAC_INF
        for d in range((1 << c) - 1, 0, -1):
            run = jac_add(run, buckets[d])
            acc = jac_add(acc, run)
        total = jac_add(total, acc)
    return to_affine(total)


# ----------------------------------------------------------------------------- SRS (Pianist slices)
def lagrange_at(i: int, m: int, tau_y: int) -> int:
    """L_i(tau_y) over the size-m domain {w_m^k}."""
    if m == 1:
        return 1
    w = root_of_unity(m)
    wi = pow(w, i, R)
    if (tau_y - wi) % R == 0:
        return 1
    num = (pow(tau_y, m, R) - 1) % R
    if num == 0:
        return 0
    return wi * fr_inv(m) % R * num % R * fr_inv(tau_y - wi) % R


def srs_scalars(tau_x: int, tau_y: int, scale: int, machines_scale: int, i: int) -> List[int]:
    """Discrete logs of worker i's slice: tau_x^j * L_i(tau_y), j < T = 2^(scale-machines_scale)."""
    T = 1 << (scale - machines_scale)
    li = lagrange_at(i, 1 << machines_scale, tau_y)
    out, cur = [], li
    for _ in range(T):
        out.append(cur)
        cur = cur * tau_x % R
    return out


def srs_slice(tau_x: int, tau_y: int, scale: int, machines_scale: int, i: int) -> List[Affine]:
    return g1_table().mul_many(srs_scalars(tau_x, tau_y, scale, machines_scale, i))


# ----------------------------------------------------------------------------- KZG worker ops
def worker_commit(srs_i: Sequence[Affine], row: Sequence[int], evaluation_form: bool = True) -> Affine:
    coeffs = ntt(row, inverse=True) if evaluation_form else list(row)
    return msm_pippenger(srs_i[: len(coeffs)], coeffs)


def worker_open(
    srs_i: Sequence[Affine], row: Sequence[int], alpha: int, evaluation_form: bool = True
) -> Tuple[int, Affine]:
    coeffs = ntt(row, inverse=True) if evaluation_form else list(row)
    y, q = poly_quotient(coeffs, alpha)
    return y, msm_pippenger(srs_i[: len(q)], q)


def trapdoor_commit(tau_x: int, tau_y: int, m_scale: int, i: int, coeffs: Sequence[int]) -> Affine:
    """[f(tau_x) * L_i(tau_y)] G -- size-independent check of an MSM against a tau-derived SRS."""
    li = lagrange_at(i, 1 << m_scale, tau_y)
    return g1_table().mul(poly_eval(coeffs, tau_x) * li % R)


def trapdoor_open(
    tau_x: int, tau_y: int, m_scale: int, i: int, coeffs: Sequence[int], alpha: int
) -> Tuple[int, Affine]:
    """(y, [q(tau_x) L_i(tau_y)] G) with q(tau) = (f(tau) - y)/(tau - alpha); requires tau != alpha."""
    li = lagrange_at(i, 1 << m_scale, tau_y)
    y = poly_eval(coeffs, alpha)
    ft = poly_eval(coeffs, tau_x)
    qt = (ft - y) * fr_inv(tau_x - alpha) % R
    return y, g1_table().mul(qt * li % R)


def verify_trapdoor(
    tau_x: int, tau_y: int, m_scale: int, i: int, commitment: Affine, proof: Affine, alpha: int, y: int
) -> bool:
    """Algebraic form of e(C - y[L_i]_1, [1]_2) == e(pi, [tau_x - alpha]_2) using the known trapdoor."""
    li = lagrange_at(i, 1 << m_scale, tau_y)
    lhs = g1_add(commitment, g1_neg(g1_table().mul(y * li % R)))
    rhs = g1_mul(proof, (tau_x - alpha) % R)
    return lhs == rhs
