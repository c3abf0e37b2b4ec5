"""The parts of the reference validator that DEFINE what the miner must compute (reference
neurons/validator.py:35-42,106-120,135-176): challenge generation (IFFT-then-eval convention), and the reward rule
(verify against the challenge's alpha/eval, linear latency discount).  Chain plumbing is out of scope."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

from .protocol import Prove


@dataclass
class Challenge:  # reference neurons/validator.py:35-42
    polys: List[List[str]]
    alpha: str
    evals: List[str]

    def to_synapse(self, i: int) -> Prove:
        return Prove(index=i, poly=self.polys[i], alpha=self.alpha, eval=self.evals[i])


def _ok(resp, key, what):
    with resp as response:
        if response.status_code != 200:
            raise Exception(f"Failed to {what}.")
        return response.json().get(key)


def generate_challenge(client, machines_count: int) -> Challenge:  # reference neurons/validator.py:106-120
    poly = _ok(client.random_poly(), "poly", "generate a random polynomial")
    alpha = _ok(client.random_point(), "point", "generate a random x")
    evals = []
    rows = getattr(client, "fft_eval_rows", None)   # MultiDeviceClient: the rows of the step spread over the host's GPUs
    if rows is not None:
        return Challenge(polys=poly, alpha=alpha,
                         evals=[_ok(r, "y", "evaluate the polynomial") for r in rows(poly[:machines_count], alpha, True, True)])
    fused = getattr(client, "fft_eval", None)       # this package's Client: both steps in one call, nothing through text
    for i in range(machines_count):
        if fused is not None:
            evals.append(_ok(fused(poly[i], alpha, left=True, inverse=True), "y", "evaluate the polynomial"))
            continue
        fft_coeffs = _ok(client.fft(poly[i], left=True, inverse=True), "poly", "fft")
        evals.append(_ok(client.eval(fft_coeffs, alpha), "y", "evaluate the polynomial"))
    return Challenge(polys=poly, alpha=alpha, evals=evals)


def verify_rows(client, challenge: Challenge, indices: Sequence[int], responses: Sequence[Optional[Prove]],
                threads: int = 16) -> List[bool]:
    """One verdict PER RESPONSE: response k is checked as worker row `indices[k]` against the challenge's alpha and
    `challenge.evals[indices[k]]` with ITS OWN proof and commitment (reference neurons/validator.py:160-170).  Two
    responses may name the same worker index -- each still gets its own verdict (they are separate rows of the batched
    check and separate row-by-row checks), so a corrupt answer can neither borrow an honest one's validity nor take
    it away.  An index without a challenge eval, or a response with missing fields, is False without a pairing.

    First ONE batched check of all checkable rows (they share alpha; `Client.worker_verify_batch`: a random linear
    combination, two Miller loops in total); if that passes every such row is valid.  Otherwise -- or with a client
    that has no batch verifier -- the independent host-side checks run row by row on a thread pool (ctypes releases the
    GIL) to tell the valid rows from the invalid ones."""
    from concurrent.futures import ThreadPoolExecutor

    n = len(responses)
    if len(indices) != n:
        raise ValueError("one worker index per response")
    n_rows = len(challenge.evals)
    answered = [k for k in range(n) if responses[k] is not None and responses[k].commitment is not None
                and responses[k].proof is not None and 0 <= int(indices[k]) < n_rows]
    batch = getattr(client, "worker_verify_batch", None)
    if batch is not None and len(answered) > 1:
        with batch([int(indices[k]) for k in answered], [responses[k].proof for k in answered], challenge.alpha,
                   [challenge.evals[int(indices[k])] for k in answered], [responses[k].commitment for k in answered],
                   threads) as r:
            if r.status_code == 200 and r.json().get("valid") is True:
                ok = set(answered)
                return [k in ok for k in range(n)]

    def one(k):
        r = responses[k]
        # 400 = the MINER's bytes are unusable (not base64, wrong length, off-curve / non-canonical): an invalid row, never
        # a reason to lose the other rows.  Only a failing verifier (5xx / 501 / 503) raises, as in the reference.
        i = int(indices[k])
        with client.worker_verify(i, r.proof, challenge.alpha, challenge.evals[i], r.commitment) as resp:
            if resp.status_code == 400:
                return False
            if resp.status_code != 200:
                raise Exception("Failed to verify the proof.")
            return bool(resp.json().get("valid"))

    verdict = [False] * n
    if len(answered) <= 1 or threads <= 1:
        for k in answered:
            verdict[k] = one(k)
        return verdict
    with ThreadPoolExecutor(max_workers=min(threads, len(answered))) as ex:
        for k, v in zip(answered, ex.map(one, answered)):
            verdict[k] = v
    return verdict


def verify_all(client, challenge: Challenge, responses: Sequence[Optional[Prove]], threads: int = 16) -> List[bool]:
    """worker_verify for every row of a step laid out BY WORKER INDEX (responses[i] answers row i; reference
    neurons/validator.py:168-170 verifies inside reward(), one row at a time: 256 pairing checks of ~4-7 ms each at
    mainnet scale).  `verify_rows` with indices 0..n-1."""
    return verify_rows(client, challenge, range(len(responses)), responses, threads)


def reward(client, challenge: Challenge, response: Optional[Prove], index: int, process_time: Optional[float],
           timeout: float = 30.0) -> float:  # reference neurons/validator.py:135-176
    if response is None or response.commitment is None or response.proof is None:
        return 0.0
    if process_time is None or process_time > timeout:
        return 0.0
    valid = _ok(client.worker_verify(index, response.proof, challenge.alpha, challenge.evals[index],
                                     response.commitment), "valid", "verify the proof")
    if not valid:
        return 0.0
    return 1 - process_time / timeout


def get_rewards(client, challenge: Challenge, responses: Sequence[Optional[Prove]], process_times: Sequence[Optional[float]],
                timeout: float = 30.0, threads: int = 16):
    """reference neurons/validator.py:178-192: one reward per response, as a float32 array -- `reward()` row by row.
    The reference reads the latency from `response.dendrite.process_time` (bittensor; not part of the wire type), here it
    comes as `process_times[k]`; the worker index of response k is `responses[k].index`, as in the reference.  Rows that
    can be scored without a pairing (missing fields, too late) are; all the others are verified TOGETHER (`verify_rows`:
    one batched check for a step whose rows share alpha, row by row only to name culprits)."""
    import numpy as np

    n = len(responses)
    if len(process_times) != n:
        raise ValueError("one process time per response")
    scores = [0.0] * n
    todo = []
    for k, (r, t) in enumerate(zip(responses, process_times)):
        if r is None or r.commitment is None or r.proof is None:
            continue                                # incomplete proof: 0.0 (reference :146-148)
        if t is None or t > timeout:
            continue                                # too slow: not even verified (:152-154)
        if not 0 <= r.index < len(challenge.evals):
            continue                                # an index the challenge holds no eval for cannot be verified: 0.0
        todo.append(k)
    if todo:
        # one verdict per RESPONSE (its own proof and commitment against challenge[response.index]), never per index:
        # two responses echoing one index do not share a verdict (reference neurons/validator.py:160-170)
        ok = verify_rows(client, challenge, [responses[k].index for k in todo], [responses[k] for k in todo], threads)
        for k, good in zip(todo, ok):
            if good:
                scores[k] = 1.0 - process_times[k] / timeout
    return np.array(scores, dtype=np.float32)
