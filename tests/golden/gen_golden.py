#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the pure-Python big-int oracle (oracle/bls12_381.py).

The reference prover (`fourier`, Rust) cannot run here, so these vectors are *oracle-generated*
(SURVEY.md 8c G2-G4); the only reference-held data is fr_kat.json, whose strings are the test data
of reference tests/test_miner.py:33-55 (TEST_POLY / TEST_POINT / TEST_EVAL).

    python tests/golden/gen_golden.py        # deterministic; rewrites the JSON files in place
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import bls12_381 as o  # noqa: E402


def hx(v, n=32):
    return v.to_bytes(n, "big").hex()


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=0, separators=(",", ":"))
        f.write("\n")


def main():
    rnd = random.Random(0xB15381)
    # ---- constants
    dump("constants.json", {
        "p": hex(o.P), "r": hex(o.R),
        "g1_compressed": o.g1_compress(o.G1).hex(),
        "g1_be96": o.g1_to_be96(o.G1).hex(),
        "identity_compressed": o.g1_compress(None).hex(),
        "two_g1_compressed": o.g1_compress(o.g1_mul(o.G1, 2)).hex(),
        "neg_g1_compressed": o.g1_compress(o.g1_neg(o.G1)).hex(),
        "roots_of_unity": {str(k): hx(o.root_of_unity(1 << k)) for k in (1, 4, 10, 12, 20, 22, 32)},
    })
    # ---- NTT vectors
    ntt_cases = []
    for n in (1, 2, 16, 1024):
        a = [rnd.randrange(o.R) for _ in range(n)]
        fwd = o.ntt(a)
        if n <= 16:
            assert fwd == o.dft_naive(a)
        ntt_cases.append({"n": n, "input": [hx(v) for v in a], "forward": [hx(v) for v in fwd],
                          "inverse": [hx(v) for v in o.ntt(a, inverse=True)]})
    dump("ntt.json", ntt_cases)
    # ---- KZG worker commit / open vectors (tau-derived Pianist slices)
    tau_x, tau_y = rnd.randrange(1, o.R), rnd.randrange(1, o.R)
    kzg = {"tau_x": hx(tau_x), "tau_y": hx(tau_y), "cases": []}

    def add_case(name, scale, ms, i, row, alpha, evaluation_form=True):
        srs = o.srs_slice(tau_x, tau_y, scale, ms, i)
        coeffs = o.ntt(row, inverse=True) if evaluation_form else list(row)
        c = o.worker_commit(srs, row, evaluation_form)
        y, pi = o.worker_open(srs, row, alpha, evaluation_form)
        assert c == o.trapdoor_commit(tau_x, tau_y, ms, i, coeffs)
        if (tau_x - alpha) % o.R:
            assert (y, pi) == o.trapdoor_open(tau_x, tau_y, ms, i, coeffs, alpha)
            assert o.verify_trapdoor(tau_x, tau_y, ms, i, c, pi, alpha, y)
        kzg["cases"].append({
            "name": name, "scale": scale, "machines_scale": ms, "i": i,
            "evaluation_form": evaluation_form,
            "row": [hx(v) for v in row], "alpha": hx(alpha),
            "commitment": o.g1_compress(c).hex(), "eval": hx(y), "proof": o.g1_compress(pi).hex(),
            "srs_first": o.g1_to_be96(srs[0]).hex(), "srs_last": o.g1_to_be96(srs[-1]).hex(),
        })

    rr = lambda n: [rnd.randrange(o.R) for _ in range(n)]  # noqa: E731
    add_case("T1", 2, 2, 3, rr(1), rnd.randrange(o.R))
    add_case("T2", 3, 2, 0, rr(2), rnd.randrange(o.R))
    add_case("T16_tests_shape", 6, 2, 1, rr(16), rnd.randrange(o.R))       # reference tests: scale 6 / 2
    add_case("T16_coeff_form", 6, 2, 2, rr(16), rnd.randrange(o.R), evaluation_form=False)
    add_case("T16_zero_poly", 6, 2, 0, [0] * 16, rnd.randrange(o.R))
    add_case("T16_all_ones", 6, 2, 0, [1] * 16, rnd.randrange(o.R))
    add_case("T16_all_r_minus_1", 6, 2, 3, [o.R - 1] * 16, rnd.randrange(o.R))
    add_case("T16_alpha_zero", 6, 2, 1, rr(16), 0)
    add_case("T16_alpha_root_of_unity", 6, 2, 1, rr(16), o.root_of_unity(16))
    add_case("T16_small_scalars", 6, 2, 1, [rnd.randrange(1 << 32) for _ in range(16)], rnd.randrange(o.R),
             evaluation_form=False)
    add_case("T256", 10, 2, 2, rr(256), rnd.randrange(o.R))
    add_case("T1024_defaults_shape", 18 - 8 + 3, 3, 5, rr(1024), rnd.randrange(o.R))  # 2^10 row = default flags
    dump("kzg.json", kzg)
    # ---- plain MSM edge cases over arbitrary (non-SRS) points
    msm = []
    base = [o.g1_table().mul(rnd.randrange(1, o.R)) for _ in range(8)]

    def add_msm(name, pts, sc):
        msm.append({"name": name, "points": [o.g1_to_be96(p).hex() for p in pts], "scalars": [hx(s) for s in sc],
                    "result": o.g1_compress(o.msm_naive(pts, sc)).hex()})

    add_msm("random8", base, rr(8))
    add_msm("empty", [], [])
    add_msm("single", base[:1], rr(1))
    add_msm("repeated_point_same_scalar", [base[0]] * 8, [5] * 8)          # forces P+P in a bucket
    add_msm("point_and_negation", [base[0], o.g1_neg(base[0]), base[1]], [7, 7, 3])  # P + (-P)
    add_msm("infinity_inputs", [None, base[2], None], [3, 4, 5])
    add_msm("cancels_to_infinity", [base[3], base[3]], [9, o.R - 9])
    add_msm("max_scalar", base[:4], [o.R - 1, o.R - 2, 1, 0])
    add_msm("high_bit_scalars", base[:4], [(1 << 254) + 3, (1 << 254), (1 << 253) - 1, 1 << 128])
    dump("msm.json", msm)
    # ---- the reference's Fr known-answer vector (data copied from reference tests/test_miner.py:33-55)
    dump("fr_kat.json", {
        "source": "reference tests/test_miner.py:33-55 (TEST_POLY, TEST_POINT, TEST_EVAL)",
        "poly": """aUXcXE/02sinJ4ybjw1GEzIM+H/5R/Iayb9CMn7BlEg aOQMCI2Ce8zgLO80vcjBK7Al++oEe8bADAyMXJJbf68
ZygfrBZOk0i4BpO6MNXU4xHeWHjrPSDjSlhQe0hLJDw X3w3fa5rnZq6113BXk//n+dSDR+FIkyV9IX0SXgVTFo
LYXDdqRAtuJcP3wRVZtqJ2hAI/NsPXoKzX59AZ3jmcc Sm+5XwJBs1g3ceeZEgyHquPIQ+zbUKOCVKkuGYloki8
EAUHn5bsQSpxn+Lp+mfUIdmPtN7EGBRZ5ZQw9dUCvSo ZJYLhpIGLcsBwP+6xWlHiomtiA7Tyd9xC+1c519IRpM
A8KIIVWkR2Qr0h+xzyVT+AlVcT8Ju7vZck4sv9ixnUE CrB/7LWe40NfYSn81gLLUZ5W17QmlBYz43o7Z2okgw8
EvpYYUWe/7rmVIJ9mL/f6lVF3fi7lihXlGPaIfF0YrU amKWoDdtgHUw2wnci7Bp/97D11QUl7gscioZnWt8WwY
FT0sgbVNfhw+g+phx/Zv2IFV8XE+5YHivoQ4yp/uGgI IWvMxK6X/j4dSyHDdcRhQPoVPnhoIBpDSAiJBHrNDC0
OBvU/pJOsQ4I8qIn09sgg6oOWh9mHNPHAsS4qTheeDk cjp2QP1+ZUcxMVY6tVFJFqyGHCaVzmUT5QYeWX5eGoE""".split(),
        "point": "RWAG//VkEtMp1SeQHQKHelgaic+md8qWPrnWgHZiNMw",
        "eval": "KXMqHg4HSrBe5qnld5TFrRlluYtsjG7N6WrHduoG/1s",
    })
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
