"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- pure-Python BLS12-381 optimal-ate pairing.

Checker for the product's host-side verifier (zkp_subnet_amd/csrc/pairing_host.cpp), deliberately built differently:
Fp12 is the direct degree-12 extension Fp[w]/(w^12 - 2 w^6 + 2) with schoolbook polynomial arithmetic (the product
uses the 2-3-2 tower), G2 points are untwisted into E(Fp12) and the Miller loop uses plain affine chord / tangent
lines there; the final exponentiation is one big-integer power.  Slow (seconds per pairing) but transparent.

Restates what the reference reaches through client.worker_verify (reference neurons/validator.py:77-86,168-170):
    e(C - y [L_i(tau_y)]_1, [1]_2) == e(pi, [tau_x - alpha]_2).          PARITY UNPINNED vs the real prover.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

from .bls12_381 import G1, P, R, Affine, g1_add, g1_mul, g1_neg, fp_inv

ATE_LOOP = 0xD201000000010000  # |x|, the BLS parameter (x is negative)

# ----------------------------------------------------------------------------- Fp2 (for G2 arithmetic)
Fp2 = Tuple[int, int]  # c0 + c1 u, u^2 = -1


def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)


def f2_inv(a):
    d = fp_inv(a[0] * a[0] + a[1] * a[1])
    return (a[0] * d % P, (-a[1]) * d % P)


G2_X: Fp2 = (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
             0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E)
G2_Y: Fp2 = (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
             0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE)
G2 = (G2_X, G2_Y)
B2: Fp2 = (4, 4)  # twist curve y^2 = x^3 + 4(u + 1)

G2Affine = Optional[Tuple[Fp2, Fp2]]


def g2_is_on_curve(pt: G2Affine) -> bool:
    if pt is None:
        return True
    x, y = pt
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), B2)) == (0, 0)


def g2_add(a: G2Affine, b: G2Affine) -> G2Affine:
    if a is None:
        return b
    if b is None:
        return a
    (x1, y1), (x2, y2) = a, b
    if x1 == x2:
        if y1 != y2 or y1 == (0, 0):
            return None
        lam = f2_mul(f2_mul((3, 0), f2_mul(x1, x1)), f2_inv(f2_mul((2, 0), y1)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))


def g2_neg(a: G2Affine) -> G2Affine:
    return None if a is None else (a[0], f2_neg(a[1]))


def g2_mul(pt: G2Affine, k: int) -> G2Affine:
    k %= R
    acc, base = None, pt
    while k:
        if k & 1:
            acc = g2_add(acc, base)
        base = g2_add(base, base)
        k >>= 1
    return acc


def g2_to_be192(pt: G2Affine) -> bytes:
    """Uncompressed G2 for the C-ABI: x.c1 || x.c0 || y.c1 || y.c0, 4 x 48 B big-endian (ZCash order); zeros = infinity."""
    if pt is None:
        return bytes(192)
    (x0, x1), (y0, y1) = pt
    return b"".join(v.to_bytes(48, "big") for v in (x1, x0, y1, y0))


# ----------------------------------------------------------------------------- Fp12 = Fp[w]/(w^12 - 2 w^6 + 2)
F12 = List[int]
F12_ONE: F12 = [1] + [0] * 11


def f12_mul(a: F12, b: F12) -> F12:
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for k in range(22, 11, -1):  # w^12 = 2 w^6 - 2
        c = t[k]
        if c:
            t[k - 6] += 2 * c
            t[k - 12] -= 2 * c
    return [v % P for v in t[:12]]


def _poly_deg(p):
    d = len(p) - 1
    while d and p[d] == 0:
        d -= 1
    return d


def f12_inv(a: F12) -> F12:
    """Extended Euclid in Fp[w] against the modulus polynomial."""
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [2, 0, 0, 0, 0, 0, P - 2, 0, 0, 0, 0, 0, 1]
    while _poly_deg(low):
        # r = high / low (polynomial division)
        dl, dh = _poly_deg(low), _poly_deg(high)
        temp = list(high)
        q = [0] * 13
        inv_lead = fp_inv(low[dl])
        for i in range(dh - dl, -1, -1):
            q[i] = temp[dl + i] * inv_lead % P
            for c in range(dl + 1):
                temp[c + i] = (temp[c + i] - q[i] * low[c]) % P
        nm, new = list(hm), temp
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] = (nm[i + j] - lm[i] * q[j]) % P
        lm, low, hm, high = nm, new, lm, low
    inv0 = fp_inv(low[0])
    return [v * inv0 % P for v in lm[:12]]


def f12_pow(a: F12, e: int) -> F12:
    acc, base = F12_ONE, a
    while e:
        if e & 1:
            acc = f12_mul(acc, base)
        base = f12_mul(base, base)
        e >>= 1
    return acc


def f12_from_fp2(c: Fp2, shift: int = 0) -> F12:
    """(c0 + c1 u) * w^shift with u = w^6 - 1."""
    out = [0] * 12
    out[0] = (c[0] - c[1]) % P
    out[6] = c[1] % P
    res = out
    for _ in range(shift):
        res = f12_mul(res, [0, 1] + [0] * 10)
    return res


_W2_INV = None
_W3_INV = None


def untwist(q: G2Affine):
    """E'(Fp2) -> E(Fp12) for the M-type twist: (x, y) -> (x / w^2, y / w^3)."""
    global _W2_INV, _W3_INV
    if _W2_INV is None:
        w = [0, 1] + [0] * 10
        w2 = f12_mul(w, w)
        _W2_INV = f12_inv(w2)
        _W3_INV = f12_inv(f12_mul(w2, w))
    x, y = q
    return f12_mul(f12_from_fp2(x), _W2_INV), f12_mul(f12_from_fp2(y), _W3_INV)


def _f12_sub(a, b): return [(x - y) % P for x, y in zip(a, b)]
def _f12_scal(a, k): return [x * k % P for x in a]


def _line(t, q, px: int, py: int):
    """Line through t and q on E(Fp12) evaluated at P = (px, py) in E(Fp); returns (value, t + q)."""
    (x1, y1), (x2, y2) = t, q
    if x1 == x2 and y1 == y2:
        lam = f12_mul(_f12_scal(f12_mul(x1, x1), 3), f12_inv(_f12_scal(y1, 2)))
    elif x1 == x2:
        val = _f12_sub([px] + [0] * 11, x1)  # vertical line
        return val, None
    else:
        lam = f12_mul(_f12_sub(y2, y1), f12_inv(_f12_sub(x2, x1)))
    val = _f12_sub(_f12_sub([py] + [0] * 11, y1), f12_mul(lam, _f12_sub([px] + [0] * 11, x1)))
    x3 = _f12_sub(_f12_sub(f12_mul(lam, lam), x1), x2)
    y3 = _f12_sub(f12_mul(lam, _f12_sub(x1, x3)), y1)
    return val, (x3, y3)


def miller_loop(p: Affine, q: G2Affine) -> F12:
    if p is None or q is None:
        return F12_ONE
    qq = untwist(q)
    t = qq
    f = F12_ONE
    for bit in bin(ATE_LOOP)[3:]:
        val, t = _line(t, t, p[0], p[1])
        f = f12_mul(f12_mul(f, f), val)
        if bit == "1":
            val, t = _line(t, qq, p[0], p[1])
            f = f12_mul(f, val)
    return f  # x < 0: the inversion f -> 1/f is absorbed by comparing products consistently (see pairing_check)


FINAL_EXP = (P**12 - 1) // R


def pairing(p: Affine, q: G2Affine) -> F12:
    """e(P, Q); the sign of the BLS parameter is accounted for by inverting the Miller value."""
    return f12_pow(f12_inv(miller_loop(p, q)), FINAL_EXP)


def pairing_check(pairs) -> bool:
    """prod e(P_i, Q_i) == 1."""
    f = F12_ONE
    for p, q in pairs:
        f = f12_mul(f, miller_loop(p, q))
    return f12_pow(f, FINAL_EXP) == F12_ONE


def kzg_verify(commitment: Affine, proof: Affine, alpha: int, y: int, li_g1: Affine, tau_g2: G2Affine) -> bool:
    """e(C - y [L_i]_1, -[1]_2) * e(pi, [tau]_2 - alpha [1]_2) == 1."""
    lhs = g1_add(commitment, g1_neg(g1_mul(li_g1, y)))
    rhs_q = g2_add(tau_g2, g2_neg(g2_mul(G2, alpha)))
    return pairing_check([(lhs, g2_neg(G2)), (proof, rhs_q)])
