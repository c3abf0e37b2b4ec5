"""Limb-exact model of csrc/fr29.hip.h (Fr on 9 unsaturated 29-bit limbs, Montgomery radix 2^261) and generator of its
constants.  Checks the product / reduction / canonicalisation formulas and their accumulator bounds on random and
extreme inputs before transcription to HIP.      python scripts/models/fr29_model.py [--emit]"""
import random
import sys

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
W, N = 29, 9
MASK = (1 << W) - 1
RR = 1 << (W * N)                       # 2^261
RL = [(R_MOD >> (W * i)) & MASK for i in range(N)]
assert RL[0] == 1 and (-pow(R_MOD, -1, 1 << W)) % (1 << W) == MASK      # q = -acc mod 2^29, r_0 = 1


def limbs(v):
    return [(v >> (W * i)) & MASK for i in range(N - 1)] + [v >> (W * (N - 1))]


def value(l):
    return sum(x << (W * i) for i, x in enumerate(l))


def dominating(k):
    """k*r with limbs i < 8 >= 2^29 - 1 (so that limb-wise K - b never borrows for normalised b)."""
    e = k * R_MOD - ((1 << (W * (N - 1))) - 1)
    el = limbs(e)
    out = [MASK + el[i] for i in range(N - 1)] + [el[N - 1]]
    assert value(out) == k * R_MOD and all(x < (1 << 30) for x in out[:-1])
    return out


ONE = limbs(RR % R_MOD)
R2 = limbs(RR * RR % R_MOD)
M4, M8 = dominating(4), dominating(8)
# M4 dominates every normalised b < 2r, M8 every normalised b < 6r (top limb comparison)
assert M4[8] >= (2 * R_MOD - 1) >> 232 and M8[8] >= (6 * R_MOD - 1) >> 232


def mul(a, b, stats=None):
    """a limbs < 2^31, b limbs < 2^29 (top limb may exceed), a*b < 2^261 r  ->  normalised limbs, value < 2r."""
    acc = 0
    q = [0] * N
    out = [0] * N
    for k in range(2 * N - 1):
        for i in range(N):
            j = k - i
            if 0 <= j < N:
                acc += a[i] * b[j]
        for i in range(N):
            j = k - i
            if i < k and i < N and 1 <= j < N:
                acc += q[i] * RL[j]
        if k < N:
            q[k] = (-acc) & MASK
            acc += q[k]                  # r_0 = 1
            assert acc & MASK == 0
        else:
            out[k - N] = acc & MASK
        if stats is not None:
            stats["max_acc"] = max(stats["max_acc"], acc)
        assert acc < (1 << 64)
        acc >>= W
    out[N - 1] = acc
    return out


def norm(a):
    c = 0
    out = []
    for i in range(N - 1):
        v = a[i] + c
        out.append(v & MASK)
        c = v >> W
    out.append(a[N - 1] + c)
    return out


def canon(a):          # normalised, < 2r  ->  [0, r)
    v = value(a)
    assert v < 2 * R_MOD
    return limbs(v - R_MOD if v >= R_MOD else v)


def main():
    rnd = random.Random(7)
    Rinv = pow(RR, -1, R_MOD)
    stats = {"max_acc": 0}
    for _ in range(4000):
        # a: lazy (a few additions of normalised values): limbs < 2^31, value < 64 r;  b: canonical
        k = rnd.randrange(1, 4)
        parts = [limbs(rnd.randrange(16 * R_MOD)) for _ in range(k)]
        a = [sum(p[i] for p in parts) for i in range(N)]
        assert max(a) < (1 << 31)
        b = limbs(rnd.randrange(R_MOD))
        r = mul(a, b, stats)
        assert value(r) % R_MOD == value(a) * value(b) * Rinv % R_MOD and value(r) < 2 * R_MOD
        assert all(x <= MASK for x in r[:-1])
        assert value(canon(r)) == value(a) * value(b) * Rinv % R_MOD
    # extremes: every limb maximal
    a = [(1 << 31) - 1] * (N - 1) + [(1 << 28)]
    b = [MASK] * (N - 1) + [R_MOD >> 232]
    r = mul(a, b, stats)
    assert value(r) % R_MOD == value(a) * value(b) * Rinv % R_MOD
    # to / from Montgomery, subtraction constants
    for _ in range(200):
        x, y = rnd.randrange(R_MOD), rnd.randrange(R_MOD)
        xm, ym = canon(mul(limbs(x), R2)), canon(mul(limbs(y), R2))
        assert value(xm) == x * RR % R_MOD
        p = mul(xm, ym)
        assert value(canon(mul(p, limbs(1)))) == x * y % R_MOD
        d = [xm[i] + (M4[i] - p[i]) for i in range(N)]          # xm - p, p < 2r normalised
        assert all(0 <= t < (1 << 32) for t in d) and value(d) % R_MOD == (value(xm) - value(p)) % R_MOD
        assert value(norm(d)) == value(d)
    print("fr29 model ok; max accumulator bits", stats["max_acc"].bit_length())
    if "--emit" in sys.argv:
        def arr(name, l):
            print(f"FR9_TABLE({name}, " + ", ".join("0x%08xu" % x for x in l) + ")")
        arr("fr9_r", RL); arr("fr9_one", ONE); arr("fr9_r2", R2); arr("fr9_m4", M4); arr("fr9_m8", M8)
        arr("fr9_2r", limbs(2 * R_MOD))


if __name__ == "__main__":
    main()
