"""Limb-exact model of the lane-parallel Montgomery product (csrc/fp_lp.hip.h): 14 limbs of 28 bits spread over the
16 lanes of a DPP row, one limb per lane.  Checks the value, the accumulator bounds (64-bit) and the number of
normalisation rounds on random and extreme inputs before the algorithm is transcribed to HIP.

    python scripts/models/lp_mul_model.py"""
import random

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
MASK = (1 << 28) - 1
PINV = (-pow(P, -1, 1 << 28)) % (1 << 28)
PL = [(P >> (28 * i)) & MASK for i in range(14)] + [0, 0]
R = 1 << 392
assert PINV == 0x0FFCFFFD


def limbs(v, loose_bits=0, rnd=None):
    l = [(v >> (28 * i)) & MASK for i in range(13)] + [v >> (28 * 13)]
    if rnd is not None and loose_bits:      # denormalise: push up to 2^loose_bits units of limb i+1 down into limb i
        for i in range(13):
            k = min(l[i + 1], rnd.randrange(1 << loose_bits))
            l[i + 1] -= k
            l[i] += k << 28
    return l + [0, 0]


def value(l):
    return sum(x << (28 * i) for i, x in enumerate(l))


def lp_mul(a, b, stats):
    """a, b: 16 lanes (limb j in lane j, lanes 14/15 zero), limbs < 2^30 (loose).  Returns 16 lanes, limbs < 2^28 except
    the top one, value = a*b/R + (something) * p < 2p."""
    t = [0] * 16
    for i in range(14):
        bi = b[i]                                        # row_newbcast:i
        t = [t[j] + a[j] * bi for j in range(16)]        # v_mad_u64_u32
        q = ((t[0] & 0xFFFFFFFF) * PINV) & MASK          # v_mul_lo_u32, v_and     (lane 0's value is the one broadcast)
        t = [t[j] + q * PL[j] for j in range(16)]        # row_newbcast:0 ; v_mad_u64_u32
        stats["max_acc"] = max(stats["max_acc"], max(t))
        assert max(t) < (1 << 64)
        assert t[0] & MASK == 0
        hi = [x >> 28 for x in t]                        # v_lshrrev_b64
        lo = [x & MASK for x in t]                       # v_and
        lo_up = lo[1:] + [0]                             # row_shl:1 (lane j <- lane j+1, zero into lane 15)
        t = [hi[j] + lo_up[j] for j in range(16)]        # 64-bit add
    # lanes 0..13 now hold columns 14..27 (each < 2^37); lanes 14, 15 hold zero
    assert t[14] == 0 and t[15] == 0
    rounds = 0
    while any(x > MASK for x in t[:13]):
        hi = [x >> 28 for x in t]
        lo = [x & MASK for x in t]
        hi_dn = [0] + hi[:-1]                            # row_shr:1 (lane j <- lane j-1)
        t = [(lo[j] if j < 13 else t[j]) + (hi_dn[j] if j <= 13 else 0) for j in range(16)]
        # lane 13 keeps its whole value (top limb holds the excess) and adds the carry of lane 12
        rounds += 1
    stats["max_rounds"] = max(stats["max_rounds"], rounds)
    stats["rounds_hist"][rounds] = stats["rounds_hist"].get(rounds, 0) + 1
    return t


def main():
    rnd = random.Random(1)
    stats = {"max_acc": 0, "max_rounds": 0, "rounds_hist": {}}
    Rinv = pow(R, -1, P)
    cases = []
    for _ in range(3000):
        cases.append((rnd.randrange(32 * P), rnd.randrange(32 * P), 2))
    edge = [0, 1, P - 1, P, P + 1, 2 * P - 1, 32 * P - 1, (1 << 386) - 1, R % P, (R * R) % P]
    for x in edge:
        for y in edge:
            cases.append((x, y, 0))
    # all-ones limb patterns (carry ripple worst cases) and maximal loose limbs
    cases.append((value([MASK] * 13 + [0x1A011]), value([MASK] * 13 + [0x1A011]), 0))
    for x, y, lb in cases:
        a, b = limbs(x, lb, rnd), limbs(y, lb, rnd)
        assert value(a) == x and value(b) == y and max(a) < (1 << 30) + (1 << 28) and max(b) < (1 << 30) + (1 << 28)
        r = lp_mul(a, b, stats)
        v = value(r)
        assert v % P == x * y * Rinv % P, (hex(x), hex(y))
        assert v < 2 * P, "not N class"
        assert all(l <= MASK for l in r[:13]) and r[14] == 0 and r[15] == 0
    # explicit maximal-limb operands (every limb 2^30 - 1 ... beyond the loose class, to see the accumulator margin)
    a = [(1 << 30) - 1] * 14 + [0, 0]
    r = lp_mul(a, a, stats)
    assert value(r) % P == value(a) * value(a) * Rinv % P
    print("lp_mul model ok:", len(cases), "cases; max accumulator bits", stats["max_acc"].bit_length(),
          "; normalisation rounds", stats["rounds_hist"])


if __name__ == "__main__":
    main()
