"""Reads the CHECK lines of scripts/ubench/fp30_mul.hip from stdin and verifies the signed-30-bit-limb Montgomery product
against big-integer arithmetic: r == a b 2^-390 and rn == (2a + b) b 2^-390 (mod p), digits within [-2^29, 2^29]."""
import sys

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
v = {}
for line in sys.stdin:
    if line.startswith("CHECK "):
        f = line.split()
        v[f[1]] = [int(x) for x in f[2:]]
val = lambda d: sum(x << (30 * i) for i, x in enumerate(d))   # noqa: E731
a, b, r, rn = (val(v[k]) for k in ("a", "b", "r", "rn"))
inv = pow(1 << 390, -1, P)
ok1 = (r - a * b * inv) % P == 0 and all(abs(x) <= 1 << 29 for x in v["r"][:12])
ok2 = (rn - (2 * a + b) * b * inv) % P == 0 and all(abs(x) <= 1 << 29 for x in v["rn"][:12])
print({"f30s_product_correct": ok1, "f30s_norm_then_product_correct": ok2, "abs_result_bits": [abs(r).bit_length(), abs(rn).bit_length()]})
sys.exit(0 if ok1 and ok2 else 1)
