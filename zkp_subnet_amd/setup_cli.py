"""`python -m zkp_subnet_amd.setup_cli setup ...` -- the analogue of the reference's
`fourier setup --setup-path P --precompute-path Q --scale S --machines-scale M --generate-setup
--generate-precompute --overwrite` (reference tests/conftest.py:50-65, Makefile:30-48).

Generates a tau-derived SRS **on the GPU** and writes
  <setup-path>      2^scale affine G1 points, x||y, 2 x 48 B big-endian each (worker slice i at points [i*T, (i+1)*T));
                    with --compressed: 48-byte ZCash-compressed points instead
  <setup-path>.vk   192 B [tau_x]_2 (uncompressed G2, x.c1||x.c0||y.c1||y.c0) + 96 B [L_i(tau_y)]_1 per worker
which is what `Client(setup_path=...)` loads.  The window tables the reference keeps in --precompute-path are rebuilt on
the GPU at Client.start(); --precompute-path is accepted for command-line compatibility and only receives a small
note.  Without --seed, tau_x and tau_y are drawn independently and uniformly from [2, r) with `secrets` (full 255-bit
entropy each) and discarded after use; --seed derives them from a public hash and is for tests only.  Either way this
is a SINGLE-PARTY trapdoor: whoever ran the command could have kept it.  Fine for staging; a production SRS must come
from a multi-party ceremony and be loaded through --setup-path like any other file.
"""
from __future__ import annotations

import argparse
import os
import secrets
import sys

from .client import R_MODULUS, derive_taus
from .engine import HipEngine, lagrange_factor
from .verifier import Verifier


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="zkp_subnet_amd.setup_cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    sp = sub.add_parser("setup")
    sp.add_argument("--setup-path", required=True)
    sp.add_argument("--precompute-path", default="")
    sp.add_argument("--scale", type=int, default=18)
    sp.add_argument("--machines-scale", type=int, default=8)
    sp.add_argument("--generate-setup", action="store_true")
    sp.add_argument("--generate-precompute", action="store_true")
    sp.add_argument("--overwrite", action="store_true")
    sp.add_argument("--compressed", action="store_true",
                    help="write 48-byte ZCash-compressed points (load with Client(uncompressed=False))")
    sp.add_argument("--seed", type=int, default=None)
    sp.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    if not a.generate_setup:
        print("nothing to do (pass --generate-setup)", file=sys.stderr)
        return 0
    if os.path.exists(a.setup_path) and not a.overwrite:
        print(f"{a.setup_path} exists (use --overwrite)", file=sys.stderr)
        return 1
    if a.seed is not None:
        print(f"WARNING: --seed {a.seed}: the trapdoor is a public function of the seed (tests only)", file=sys.stderr)
        tau_x, tau_y = derive_taus(a.seed)
    else:
        tau_x, tau_y = secrets.randbelow(R_MODULUS - 2) + 2, secrets.randbelow(R_MODULUS - 2) + 2
    m = 1 << a.machines_scale
    T = 1 << (a.scale - a.machines_scale)
    eng = HipEngine(a.device)
    eng.gen_srs(tau_x, tau_y, a.scale, a.machines_scale)
    with open(a.setup_path, "wb") as f:
        step = max(1, (1 << 20) // T)
        for i in range(0, m, step):
            f.write(eng.srs_read(i * T, min(step, m - i) * T, compressed=a.compressed))
    vk = Verifier.synthetic(tau_x, [lagrange_factor(i, a.machines_scale, tau_y) for i in range(m)])
    with open(a.setup_path + ".vk", "wb") as f:
        f.write(vk.export(m))
    if a.precompute_path and a.generate_precompute:
        with open(a.precompute_path, "w") as f:
            f.write(f"window tables are rebuilt on the GPU at Client.start(); window bits = {eng.window}\n")
    eng.close()
    print(f"wrote {a.setup_path} ({m * T} points, {m} worker slices of {T}) and {a.setup_path}.vk")
    return 0


if __name__ == "__main__":
    sys.exit(main())
