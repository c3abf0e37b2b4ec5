"""CPU-only tests of the host side above the C-ABI: the Client method surface / status mapping, the miner and
validator mirrors, and the config-1 plumbing loop (degree-2^12 commit through the CPU prover under a mock
miner/validator loop, BASELINE.json configs[0]).  The engine is the oracle-backed stand-in (tests/oracle_engine.py);
the same tests run against the HIP engine in tests/test_gpu_*.py."""
import base64
import os
import random

import pytest

from oracle import bls12_381 as o
from oracle import cpu as oc
from tests.oracle_engine import OracleEngine
from zkp_subnet_amd import codec
from zkp_subnet_amd.client import Client, derive_taus
from zkp_subnet_amd.miner import Miner, default_config
from zkp_subnet_amd.protocol import Prove
from zkp_subnet_amd.validator import Challenge, generate_challenge, reward

H = bytes.fromhex


def make_client(scale, ms, workers=None, seed=3):
    c = Client(port=1337, bin="./prover", uncompressed=False, setup_path="", precompute_path="",
               engine=OracleEngine(), seed=seed, workers=workers)
    c.start(scale=scale, machines_scale=ms)
    return c


def test_client_surface_matches_reference_call_sites():
    # keyword names used by reference tests/test_miner.py:88-107
    c = make_client(6, 2)
    rnd = random.Random(1)
    poly = [o.fr_to_b64(rnd.randrange(o.R)) for _ in range(16)]
    x = o.fr_to_b64(rnd.randrange(o.R))
    with c.worker_commit(i=0, poly=poly) as resp:
        assert resp.status_code == 200
        commitment = resp.json().get("commitment")
    with c.worker_open(i=0, poly=poly, x=x) as resp:
        assert resp.status_code == 200
        ev, proof = resp.json().get("eval"), resp.json().get("proof")
    with c.worker_verify(i=0, proof=proof, alpha=x, eval=ev, commitment=commitment) as resp:
        assert resp.status_code == 200 and resp.json().get("valid") is True
    assert len(commitment) == 64 and len(proof) == 64 and len(ev) == 43
    # fused extension agrees with the two-call path
    with c.worker_commit_and_open(0, poly, x) as resp:
        assert resp.json() == {"commitment": commitment, "eval": ev, "proof": proof}
    # P3: eval == eval(fft(poly, left=True, inverse=True), x)   (reference neurons/validator.py:115-118)
    with c.fft(poly, left=True, inverse=True) as resp:
        coeffs = resp.json().get("poly")
    with c.eval(coeffs, x) as resp:
        assert resp.json().get("y") == ev
    with c.random_poly() as resp:
        rows = resp.json().get("poly")
        assert len(rows) == 4 and all(len(r) == 16 for r in rows)
    with c.random_point() as resp:
        assert len(resp.json().get("point")) == 43
    c.stop()


def test_client_golden_vectors(golden_kzg):
    tx, ty = int(golden_kzg["tau_x"], 16), int(golden_kzg["tau_y"], 16)
    for case in golden_kzg["cases"]:
        if not case["evaluation_form"]:
            continue
        eng = OracleEngine()
        eng.gen_srs(tx, ty, case["scale"], case["machines_scale"], [case["i"]])
        c = Client(engine=eng)
        c.scale, c.machines_scale, c._slice_of = case["scale"], case["machines_scale"], {case["i"]: 0}
        poly = [o.fr_to_b64(int(v, 16)) for v in case["row"]]
        with c.worker_commit(case["i"], poly) as r:
            assert base64.b64decode(r.json()["commitment"]).hex() == case["commitment"], case["name"]
        with c.worker_open(case["i"], poly, o.fr_to_b64(int(case["alpha"], 16))) as r:
            assert o.fr_from_b64(r.json()["eval"]) == int(case["eval"], 16)
            assert base64.b64decode(r.json()["proof"]).hex() == case["proof"]


def test_client_error_statuses_never_raise():
    c = make_client(6, 2, workers=[1])
    good = [o.fr_to_b64(5)] * 16
    assert c.worker_commit(1, good).status_code == 200
    assert c.worker_commit(0, good).status_code == 400          # no resident slice for worker 0
    assert c.worker_commit(7, good).status_code == 400          # index >= 2^machines_scale
    assert c.worker_commit(1, ["@@@"]).status_code == 400       # not base64
    assert c.worker_open(1, good, "short").status_code == 400
    not_started = Client(engine=None)
    not_started.engine = None
    assert not_started.worker_commit(0, good).status_code == 503


def test_miner_forward_matches_client_and_swallows_errors():
    """Reference tests/test_miner.py:62-121: forward() returns exactly the client's commitment / proof; any failure
    echoes the request (neurons/miner.py:133-135)."""
    cfg = default_config(scale=6, machines_scale=2, seed=11)
    client = Client(engine=OracleEngine(), seed=11)
    rnd = random.Random(2)
    poly = [o.fr_to_b64(rnd.randrange(o.R)) for _ in range(16)]
    alpha = o.fr_to_b64(rnd.randrange(o.R))
    for fused in (False, True):
        cfg.fused = fused
        miner = Miner(cfg, client=client)
        syn = Prove(index=2, poly=poly, alpha=alpha)
        with miner.client.worker_commit(i=2, poly=poly) as r:
            commitment = r.json()["commitment"]
        with miner.client.worker_open(i=2, poly=poly, x=alpha) as r:
            ev, proof = r.json()["eval"], r.json()["proof"]
        ret = miner.forward(syn)
        assert (ret.commitment, ret.eval, ret.proof) == (commitment, ev, proof)
        assert ret.poly == [] and ret.alpha is None and ret.index == 2
        bad = Prove(index=2, poly=poly, alpha=None)   # include_point=False branch of the reference test
        assert miner.forward(bad) is bad
        bad2 = Prove(index=9, poly=poly, alpha=alpha)
        assert miner.forward(bad2) is bad2


def test_validator_reward_table():
    """Reference tests/test_validator.py:60-121: ok / missing / late / bit-flipped proof / half-time."""
    c = make_client(6, 2, seed=5)
    miner = Miner(default_config(scale=6, machines_scale=2), client=c)
    ch = generate_challenge(c, 2)
    assert isinstance(ch, Challenge) and len(ch.evals) == 2
    responses = [miner.forward(ch.to_synapse(i)) for i in range(2)]
    assert [reward(c, ch, responses[i], i, 0.0) for i in range(2)] == [1.0, 1.0]
    assert reward(c, ch, None, 0, 0.0) == 0.0
    assert reward(c, ch, responses[0], 0, 31.0) == 0.0
    assert reward(c, ch, responses[0], 0, 15.0) == 0.5
    raw = base64.b64decode(responses[0].proof)                     # reference test_validator.py:79-86
    bumped = base64.b64encode((int.from_bytes(raw, "big") + 1).to_bytes(len(raw), "big")).decode()
    corrupt = responses[0].model_copy(update={"proof": bumped})
    assert reward(c, ch, corrupt, 0, 0.0) == 0.0
    # the miner's eval equals the validator's independently computed one
    assert responses[1].eval == ch.evals[1]
    # get_rewards (reference neurons/validator.py:178-192): the same five scenarios as one float32 array, row order kept
    from zkp_subnet_amd.validator import get_rewards
    import numpy as np

    got = get_rewards(c, ch, [responses[1], responses[0]], [0.0, 5.0], timeout=10.0)
    assert got.dtype == np.float32 and list(got) == [1.0, 0.5]
    assert list(get_rewards(c, ch, [corrupt, responses[1]], [0.0, 0.0], 10.0)) == [0.0, 1.0]
    assert list(get_rewards(c, ch, [responses[0].model_copy(update={"commitment": None}), responses[1]], [0.0, 0.0], 10.0)) == [0.0, 1.0]
    assert list(get_rewards(c, ch, [responses[0], responses[1]], [11.0, 0.0], 10.0)) == [0.0, 1.0]
    assert list(get_rewards(c, ch, [None, responses[1]], [None, 2.5], 10.0)) == [0.0, 0.75]
    assert [float(x) for x in get_rewards(c, ch, responses, [0.0, 0.0], 30.0)] == [reward(c, ch, responses[i], i, 0.0) for i in range(2)]


def test_get_rewards_scores_each_response_on_its_own_proof_even_when_indices_repeat():
    """Reference neurons/validator.py:160-170 verifies every response's OWN proof and commitment against
    challenge[response.index]: a corrupt answer echoing another miner's index can neither score nor zero the honest one
    (ADVICE r4, high), and an index the challenge holds no eval for scores 0.0 instead of raising (medium)."""
    from zkp_subnet_amd.validator import get_rewards, verify_rows

    c = make_client(6, 2, seed=5)
    miner = Miner(default_config(scale=6, machines_scale=2), client=c)
    ch = generate_challenge(c, 2)                                   # 4 rows of polys, evals for the first 2 only
    assert len(ch.polys) == 4 and len(ch.evals) == 2
    responses = [miner.forward(ch.to_synapse(i)) for i in range(2)]
    raw = base64.b64decode(responses[1].proof)
    bumped = base64.b64encode((int.from_bytes(raw, "big") + 1).to_bytes(len(raw), "big")).decode()
    forged = responses[1].model_copy(update={"proof": bumped})      # index 1, corrupt proof
    honest = responses[1]
    for threads in (1, 4):
        assert list(get_rewards(c, ch, [forged, honest], [0.0, 0.0], 10.0, threads)) == [0.0, 1.0]
        assert list(get_rewards(c, ch, [honest, forged], [0.0, 0.0], 10.0, threads)) == [1.0, 0.0]
        assert list(get_rewards(c, ch, [honest, honest, responses[0]], [0.0, 5.0, 0.0], 10.0, threads)) == [1.0, 0.5, 1.0]
        assert list(get_rewards(c, ch, [forged, responses[0], honest, forged], [0.0] * 4, 10.0, threads)) == [0.0, 1.0, 1.0, 0.0]
    # another row's valid proof under the wrong index is invalid for THAT index
    swapped = responses[0].model_copy(update={"index": 1})
    assert list(get_rewards(c, ch, [swapped, honest], [0.0, 0.0], 10.0)) == [0.0, 1.0]
    # indices in [len(evals), len(polys)) and outside the challenge altogether: 0.0, nobody else loses the step
    for idx in (2, 3, 4, 99, -1):
        stray = responses[0].model_copy(update={"index": idx})
        assert list(get_rewards(c, ch, [stray, honest, responses[0]], [0.0] * 3, 10.0)) == [0.0, 1.0, 1.0], idx
    assert verify_rows(c, ch, [3, 1, 0, 7], [responses[0], honest, None, honest]) == [False, True, False, False]
    with pytest.raises(ValueError):
        verify_rows(c, ch, [0], [responses[0], honest])


def test_config1_plumbing_degree_4096_commit_under_mock_loop():
    """BASELINE.json configs[0]: degree-2^12 random polynomial (scale 20 / machines_scale 8 shape) through the CPU
    prover under the miner/validator loop.  Trapdoor cross-check keeps it independent of the prover's own MSM."""
    seed = 7
    c = make_client(20 - 8 + 1, 1, workers=[1], seed=seed)      # T = 2^12, one resident worker slice
    miner = Miner(default_config(scale=13, machines_scale=1), client=c)
    rnd = random.Random(0)
    row = [rnd.randrange(o.R) for _ in range(1 << 12)]
    alpha = rnd.randrange(o.R)
    syn = Prove(index=1, poly=[o.fr_to_b64(v) for v in row], alpha=o.fr_to_b64(alpha))
    ret = miner.forward(syn)
    assert ret.commitment is not None and ret.proof is not None
    tx, ty = derive_taus(seed)
    coeffs = o.ntt(row, inverse=True)
    assert codec.g1_from_b64(ret.commitment) == o.g1_compress(o.trapdoor_commit(tx, ty, 1, 1, coeffs))
    y, pi = o.trapdoor_open(tx, ty, 1, 1, coeffs, alpha)
    assert o.fr_from_b64(ret.eval) == y and codec.g1_from_b64(ret.proof) == o.g1_compress(pi)
    with c.worker_verify(1, ret.proof, syn.alpha, ret.eval, ret.commitment) as r:
        assert r.json()["valid"] is True


def test_commitment_api_shape(fr_kat):
    """reference api/commit.py:34-100: poly -> Prove -> k random axons -> first str commitment, "" when none answers."""
    import random

    from zkp_subnet_amd.api import CommitmentAPI, CommitOnlyAxon, commit
    from zkp_subnet_amd.miner import Miner, default_config

    client = Client(engine=OracleEngine(), seed=9)
    faithful = Miner(default_config(scale=6, machines_scale=2, seed=9), client=client)
    assert commit(fr_kat["poly"], [faithful], index=2) == ""      # the reference miner rejects an alpha-less synapse
    miner = CommitOnlyAxon(faithful)

    class Dead:
        def forward(self, synapse):
            return synapse                              # what a failing miner returns (neurons/miner.py:133-135)

    with client.worker_commit(2, fr_kat["poly"]) as r:
        want = r.json()["commitment"]
    assert commit(fr_kat["poly"], [miner], index=2) == want
    assert commit(fr_kat["poly"], [Dead(), miner, Dead()], index=2, k=8, rng=random.Random(1)) == want
    assert commit(fr_kat["poly"], [Dead()], index=2) == "" and commit(fr_kat["poly"], []) == ""
    api = CommitmentAPI([miner])
    syn = api.prepare_synapse(fr_kat["poly"], 1)
    assert syn.poly == fr_kat["poly"] and syn.alpha is None and syn.index == 1
    assert api.select_commitment([None, 3, "", "abc", "def"]) == "abc" and api.select_commitment([]) is None


def test_missing_setup_file_is_an_error_unless_synthetic_requested(tmp_path, caplog):
    """The reference prover fails when its setup file is absent (base/miner.py:75-84 hands the path to the binary).  A
    synthetic SRS has a public trapdoor, so it must be asked for explicitly (synthetic=True or an explicit seed)."""
    missing = str(tmp_path / "no_such_setup")
    c = Client(engine=OracleEngine(), setup_path=missing)
    with pytest.raises(FileNotFoundError):
        c.start(scale=6, machines_scale=2)
    with pytest.raises(FileNotFoundError):      # the miner's defaults (setup_path="./setup", no seed) do not fall back
        Miner(default_config(scale=6, machines_scale=2, setup_path=missing), client=None if False else Client(
            engine=OracleEngine(), setup_path=missing))
    import logging
    with caplog.at_level(logging.WARNING, logger="zkp_subnet_amd.client"):
        ok = Client(engine=OracleEngine(), setup_path=missing, synthetic=True)
        ok.start(scale=6, machines_scale=2)
    assert any("SYNTHETIC SRS" in r.message for r in caplog.records)
    assert ok.worker_commit(0, [o.fr_to_b64(5)] * 16).status_code == 200
    # seeds beyond 64 bits derive a trapdoor instead of overflowing
    from zkp_subnet_amd.client import derive_taus
    assert derive_taus(1 << 70) != derive_taus(1 << 71)


def test_challenge_synapse_carries_eval():
    """reference neurons/validator.py:41-42: the synapse sent to the miner carries the expected eval."""
    from zkp_subnet_amd.validator import Challenge
    ch = Challenge(polys=[["a"], ["b"]], alpha="x", evals=["e0", "e1"])
    s = ch.to_synapse(1)
    assert (s.index, s.poly, s.alpha, s.eval) == (1, ["b"], "x", "e1")


def test_aggregate_commitments_is_the_bivariate_commitment():
    """SURVEY 8f-4 / reference README.md:38: sum_i commit_i over ALL worker rows equals [f(tau_x, tau_y)] G for the
    bivariate polynomial whose row i (evaluation form in X) is worker i's polynomial, f = sum_i L_i(Y) f_i(X)."""
    scale, ms, seed = 5, 2, 13
    c = make_client(scale, ms, seed=seed)
    T, m = 1 << (scale - ms), 1 << ms
    rnd = random.Random(8)
    rows = [[rnd.randrange(o.R) for _ in range(T)] for _ in range(m)]
    comms = []
    for i in range(m):
        with c.worker_commit(i, [o.fr_to_b64(v) for v in rows[i]]) as r:
            assert r.status_code == 200
            comms.append(r.json()["commitment"])
    with c.aggregate_commitments(comms) as r:
        assert r.status_code == 200
        total = r.json()["commitment"]
    from zkp_subnet_amd.client import derive_taus
    tx, ty = derive_taus(seed)
    acc = 0
    for i in range(m):
        acc = (acc + o.lagrange_at(i, m, ty) * o.poly_eval(o.ntt(rows[i], inverse=True), tx)) % o.R
    assert base64.b64decode(total) == o.g1_compress(o.g1_mul(o.G1, acc))
    assert c.aggregate_commitments(["AAAA"]).status_code == 400
    with c.aggregate_commitments([]) as r:
        assert base64.b64decode(r.json()["commitment"]) == b"\xc0" + bytes(47)


def test_native_random_rows_and_fused_challenge_step_match_the_two_call_form():
    """Validator side at scale (reference neurons/validator.py:58-120): random_poly comes from the native generator
    (getrandom + rejection below r, text produced by the codec's thread pool), generate_challenge uses the fused
    fft_eval step when the client has one -- same evals as the reference's fft-then-eval pair -- and verify_all agrees
    with reward() row by row."""
    from zkp_subnet_amd.validator import verify_all

    assert codec._wire is not None
    c = make_client(8, 2, seed=9)
    with c.random_poly() as r:
        rows = r.json()["poly"]
    assert len(rows) == 4 and all(len(row) == 64 for row in rows)
    vals = [int.from_bytes(codec.fr_to_be32(s), "big") for row in rows for s in row]
    assert all(v < o.R for v in vals) and len(set(vals)) == len(vals) and max(vals).bit_length() >= 253
    with c.random_point() as r:
        assert int.from_bytes(codec.fr_to_be32(r.json()["point"]), "big") < o.R
    big = codec._wire.random_fr_rows(3, 40000)                   # several generator batches and thread ranges
    assert [len(x) for x in big] == [40000] * 3
    raw = codec.fr_list_to_be32(big[2])
    assert len({raw[i:i + 32] for i in range(0, len(raw), 32)}) == 40000
    top = sorted(raw[i] for i in range(0, len(raw), 32))
    assert top[0] == 0 and top[-1] == 0x73 and 50 < sum(top) / len(top) < 65      # uniform below r: top byte in [0, 0x73]
    ch = generate_challenge(c, 4)                                # fused step
    for i in range(4):
        with c.fft(ch.polys[i], left=True, inverse=True) as r:
            coeffs = r.json()["poly"]
        with c.eval(coeffs, ch.alpha) as r:
            assert r.json()["y"] == ch.evals[i]
    assert c.fft_eval(ch.polys[0][:5], ch.alpha).status_code == 400       # wrong row length, as fft()
    miner = Miner(default_config(scale=8, machines_scale=2), client=c)
    responses = [miner.forward(ch.to_synapse(i)) for i in range(4)]
    responses[3] = responses[3].model_copy(update={"commitment": responses[0].commitment})
    got = verify_all(c, ch, responses, threads=3)
    assert got == [True, True, True, False] == [reward(c, ch, responses[i], i, 0.0) > 0 for i in range(4)]
    assert verify_all(c, ch, [None, responses[1]], threads=1) == [False, True]
    # one MALFORMED miner answer (not base64 / wrong length / not a curve point) is an invalid row, not an exception that
    # loses the whole step's results
    for bad in ("not-base64!!", responses[1].proof[:-4], base64.b64encode(b"\xff" * 48).decode()):
        resp = list(responses[:3]) + [responses[3].model_copy(update={"commitment": responses[3].commitment})]
        resp[1] = resp[1].model_copy(update={"proof": bad})
        assert verify_all(c, ch, resp, threads=3) == [True, False, True, False], bad
        assert verify_all(c, ch, resp, threads=1) == [True, False, True, False], bad


def test_bench_gpus_n_without_a_launcher_self_launches_or_refuses_cleanly():
    """`python bench.py --gpus N` (N > 1, no torch.distributed.run in front) must never end in a bare SystemExit: more
    ranks than visible GPUs -> rc 2 with a message; otherwise the parent starts the launch line as a child and returns
    ITS exit code (here the ranks stop because this container has no GPU)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_ONE_GPU")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--steps", "1"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 2 and "GPU(s) are visible" in out.stderr, (out.returncode, out.stderr[-500:])
    import torch

    if torch.cuda.device_count() == 0:              # the relay path: ranks are started, fail, and the code comes back
        env["BENCH_ONE_GPU"] = "1"
        env["BENCH_BACKEND"] = "gloo"
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"],
                             capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode not in (0, 2) and "needs an MI355X" in out.stderr, (out.returncode, out.stderr[-500:])


def test_multi_device_client_routes_rows_by_worker_index():
    """MultiDeviceClient (SURVEY 8b's multi-device form, reference base/miner.py:73-84 builds one client per process):
    G contexts, worker index i served by context i mod G, the rest of the surface on any of them.  With oracle-backed
    engines standing in for the GPUs: every answer equals a single Client's, each engine only ever sees its own rows,
    the Miner takes the router unchanged, and the challenge step spreads its rows over the devices."""
    from zkp_subnet_amd import MultiDeviceClient
    from zkp_subnet_amd.validator import verify_all

    class Spy(OracleEngine):
        def __init__(self):
            super().__init__()
            self.seen = []
            self.batches = []

        def commit_open(self, i, row, alpha, evaluation_form=True):
            self.seen.append(("commit_open", self.workers[i]))
            return super().commit_open(i, row, alpha, evaluation_form)

        def commit(self, i, row, evaluation_form=True):
            self.seen.append(("commit", self.workers[i]))
            return super().commit(i, row, evaluation_form)

        def open(self, i, row, alpha, evaluation_form=True):
            self.seen.append(("open", self.workers[i]))
            return super().open(i, row, alpha, evaluation_form)

        def verify_batch(self, slices, proofs, alpha, evals, commitments, threads=1):
            self.batches.append([self.workers[i] for i in slices])
            return all(self.verify(i, p, alpha, e, c) for i, p, e, c in zip(slices, proofs, evals, commitments))

    single = make_client(7, 2, seed=14)
    engines = [Spy(), Spy(), Spy()]
    multi = MultiDeviceClient(devices=[0, 1, 2], seed=14, engines=engines)
    assert multi.worker_commit(0, ["x"]).status_code == 503             # not started yet, as Client
    multi.start(scale=7, machines_scale=2)
    assert [e.workers for e in engines] == [[0, 3], [1], [2]] and multi.device_of(3) == 0
    ch = generate_challenge(multi, 4)                                    # fft_eval_rows: rows spread over the devices
    for i in range(4):
        with single.fft_eval(ch.polys[i], ch.alpha) as r:
            assert r.json()["y"] == ch.evals[i]
    answers = multi.commit_and_open_rows(range(4), ch.polys, ch.alpha)
    for i, a in enumerate(answers):
        with single.worker_commit_and_open(i, ch.polys[i], ch.alpha) as b:
            assert a.status_code == 200 and a.json() == b.json()
        with multi.worker_commit(i, ch.polys[i]) as c, multi.worker_open(i, ch.polys[i], ch.alpha) as d:
            assert c.json()["commitment"] == a.json()["commitment"] and d.json()["proof"] == a.json()["proof"]
    for g, e in enumerate(engines):                                      # every engine saw its own worker indices only
        assert e.seen and all(w % 3 == g for _, w in e.seen), (g, e.seen)
    miner = Miner(default_config(scale=7, machines_scale=2), client=multi)
    responses = [miner.forward(ch.to_synapse(i)) for i in range(4)]
    assert [r.commitment for r in responses] == [a.json()["commitment"] for a in answers]
    assert verify_all(multi, ch, responses, threads=2) == [True] * 4
    # the batched check runs once per device on that device's own rows (each context only holds its workers' slices and
    # verifier-key factors) and the verdicts are AND-ed -- it must SUCCEED here, not fall back to row-by-row
    idx = list(range(4))
    with multi.worker_verify_batch(idx, [r.proof for r in responses], ch.alpha, ch.evals, [r.commitment for r in responses]) as r:
        assert r.status_code == 200 and r.json() == {"valid": True}
    assert [e.batches[-1] for e in engines] == [[0, 3], [1], [2]]
    bad = [r.proof for r in responses]
    bad[2] = responses[1].proof
    with multi.worker_verify_batch(idx, bad, ch.alpha, ch.evals, [r.commitment for r in responses]) as r:
        assert r.status_code == 200 and r.json() == {"valid": False}
    assert multi.worker_verify_batch(idx, bad[:3], ch.alpha, ch.evals, [r.commitment for r in responses]).status_code == 400
    assert multi.worker_commit(9, ch.polys[0]).status_code == 400        # index outside 2^machines_scale, as Client
    with multi.aggregate_commitments([r.commitment for r in responses]) as r, \
            single.aggregate_commitments([r.commitment for r in responses]) as s:
        assert r.status_code == s.status_code and r.json() == s.json()
    multi.stop()
    assert multi.worker_commit(0, ch.polys[0]).status_code == 503


def test_multi_device_client_from_a_setup_file_loads_only_the_served_slices(tmp_path):
    """VERDICT r5 task 4: with a setup FILE every device of a MultiDeviceClient loads only the slices of the worker indices
    it serves (first = g, stride = G: kzg_load_srs_file_slices), never the whole file; worker i is then resident slice
    i // G of device i % G.  With oracle-backed engines that record what they were asked to load: the arguments, the
    resident bytes, and every answer against a single whole-file Client.  A client whose workers are NOT a progression over
    the file falls back to the whole file."""
    from zkp_subnet_amd import MultiDeviceClient

    scale, ms = 9, 3                       # 8 slices of 64 points
    T, M, G = 1 << (scale - ms), 1 << ms, 3
    tx, ty = (0xF11E5 % o.R).to_bytes(32, "big"), (0x5EED % o.R).to_bytes(32, "big")
    slices = [oc.srs_gen(tx, ty, scale, ms, i) for i in range(M)]
    path = tmp_path / "setup_9_3.uncompressed"
    path.write_bytes(b"".join(slices))

    class FileEngine(OracleEngine):
        def __init__(self):
            super().__init__()
            self.loads = []

        def load_srs_file(self, p, scale, machines_scale, compressed=False):
            self.loads.append(("whole",))
            with open(p, "rb") as f:
                self.load_srs(f.read(), scale, machines_scale, compressed)

        def load_srs_file_slices(self, p, scale, machines_scale, first_slice, slice_stride, compressed=False):
            self.loads.append(("slices", first_slice, slice_stride))
            t = 96 << (scale - machines_scale)
            with open(p, "rb") as f:
                blob = f.read()
            self.load_srs(b"".join(blob[k * t:(k + 1) * t] for k in range(first_slice, len(blob) // t, slice_stride)), scale,
                          machines_scale, compressed)

    engines = [FileEngine() for _ in range(G)]
    multi = MultiDeviceClient(devices=[0, 1, 2], setup_path=str(path), engines=engines)
    multi.start(scale, ms)
    assert [e.loads for e in engines] == [[("slices", g, G)] for g in range(G)]
    for g, e in enumerate(engines):
        assert e.srs == b"".join(slices[i] for i in range(g, M, G))          # 3 + 3 + 2 slices: 1 / G of the file each
    single = Client(setup_path=str(path), engine=FileEngine())
    single.start(scale, ms)
    assert single.engine.loads == [("whole",)]
    rnd = random.Random(4)
    alpha = codec.be32_to_fr(rnd.randrange(o.R).to_bytes(32, "big"))
    for i in range(M):
        poly = codec.be32_to_fr_list(b"".join(rnd.randrange(o.R).to_bytes(32, "big") for _ in range(T)))
        with multi.worker_commit_and_open(i, poly, alpha) as a, single.worker_commit_and_open(i, poly, alpha) as b:
            assert a.status_code == 200 and a.json() == b.json(), i
    assert multi.worker_commit(M, ["x"]).status_code == 400
    multi.stop()
    # workers that are not first, first + stride, ... over the file's slices: the whole file, slice i = worker i
    odd = Client(setup_path=str(path), engine=FileEngine(), workers=[0, 1, 5])
    odd.start(scale, ms)
    assert odd.engine.loads == [("whole",)] and odd._slice(5) == 5
    prog = Client(setup_path=str(path), engine=FileEngine(), workers=[1, 4, 7])
    prog.start(scale, ms)
    assert prog.engine.loads == [("slices", 1, 3)] and prog._slice(7) == 2
    assert prog.worker_commit(2, ["x"]).status_code == 400                   # worker 2 has no resident slice here
    # MORE devices than worker rows (machines_scale 1: two rows, three devices): the third device serves nothing and says so
    small = [FileEngine() for _ in range(3)]
    m2 = MultiDeviceClient(devices=[0, 1, 2], seed=3, engines=small)
    m2.start(5, 1)
    assert [e.workers for e in small] == [[0], [1], None]                    # (the third engine was never asked to generate)
    assert m2.clients[2].worker_commit(0, ["x"]).status_code == 400
    with m2.worker_commit(1, codec.be32_to_fr_list(bytes(32 * 16))) as r:
        assert r.status_code == 200
    m2.stop()


def test_lane_book_drive_without_a_sanitizer(tmp_path):
    """csrc/lanebook.h (lanes, MSM tickets, staging pool, row-cache slots: the HIP-free half of csrc/lanes.hip) under the
    12-thread fake-back-end drive of tests/lanebook_tsan.cpp, here compiled WITHOUT a sanitizer as a functional check of its
    invariants (no slot ever handed to two holders, tickets claimed exactly once, BUSY instead of waits that only the caller
    could end, hits return the row they were filled with).  The ThreadSanitizer run is scripts/sanitize_cpu.sh tsan-lanes
    (profiles/r05_sanitize_tsan_lanes.log)."""
    import subprocess

    exe = str(tmp_path / "lanebook_drive")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(root, "zkp_subnet_amd", "csrc"),
                          os.path.join(root, "tests", "lanebook_tsan.cpp"), "-o", exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    out = subprocess.run([exe, "1.5", "8"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "invariant failures: 0" in out.stdout, (out.stdout[-500:], out.stderr[-1000:])


def test_evidence_keep_refuses_lines_of_another_tree_and_byte_identical_files(tmp_path):
    """VERDICT r4 task 4, enforced by code: a bench line gets into profiles/ only if its identity.source_sha16 is THIS tree's;
    a line measured on other sources, a file without a JSON line and a byte-identical twin of an existing evidence file are
    refused (exit 1) and nothing is written for them."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import source_sha16

    keep = os.path.join(root, "scripts", "evidence_keep.py")
    good = {"metric": "m", "value": 1.0, "identity": {"source_sha16": source_sha16(root), "git_head": None, "utc": "t"}}
    bad = {"metric": "m", "value": 2.0, "identity": {"source_sha16": "0" * 16, "git_head": "abc"}}
    (tmp_path / "good.json").write_text("noise before the line\n" + json.dumps(good) + "\n")
    (tmp_path / "bad.json").write_text(json.dumps(bad) + "\n")
    (tmp_path / "empty.json").write_text("no json here\n")
    dst = os.path.join(root, "profiles", "_test_evidence_keep.json")
    try:
        r = subprocess.run([sys.executable, keep, str(tmp_path / "good.json"), dst], capture_output=True, text=True)
        assert r.returncode == 0 and "kept" in r.stdout and json.load(open(dst)) == good, r.stdout + r.stderr
        twin = os.path.join(root, "profiles", "_test_evidence_keep_twin.json")
        r = subprocess.run([sys.executable, keep, str(tmp_path / "good.json"), twin], capture_output=True, text=True)
        assert r.returncode == 1 and "byte-identical" in r.stdout and not os.path.exists(twin)
        for name, why in (("bad.json", "measured on source set"), ("empty.json", "no JSON line")):
            out = os.path.join(root, "profiles", "_test_evidence_refused.json")
            r = subprocess.run([sys.executable, keep, str(tmp_path / name), out], capture_output=True, text=True)
            assert r.returncode == 1 and "REFUSED" in r.stdout and why in r.stdout and not os.path.exists(out), r.stdout
    finally:
        for f in ("_test_evidence_keep.json", "_test_evidence_keep_twin.json", "_test_evidence_refused.json"):
            try:
                os.remove(os.path.join(root, "profiles", f))
            except OSError:
                pass
