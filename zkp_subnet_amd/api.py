"""Client-side commitment query: the shape of the reference's `api/commit.py` (`CommitmentAPI` :34-72, `commit()`
:75-100) without bittensor.  The reference wraps `poly` in a `Prove`, sends it to three randomly chosen axons and
returns the first `str` commitment ("" when none answered).  Here an "axon" is anything with `forward(Prove) -> Prove`
(in-process `zkp_subnet_amd.miner.Miner` objects; a bittensor deployment plugs its dendrite call in instead).

The reference's own `select_commitment` wraps a list in `with`, which cannot run; the intended behaviour (first valid
string) is what is implemented."""
from __future__ import annotations

import random
from typing import Any, List, Optional, Sequence

from .protocol import Prove

COMMITMENT_API_NAME = "commitment"


class CommitOnlyAxon:
    """Serves the alpha-less synapse of this API from a miner's commit handler (reference neurons/miner.py:38-45).
    `Miner.forward` itself keeps the reference's behaviour for such a synapse -- it fails and echoes the request
    (reference tests/test_miner.py, include_point=False) -- which is why `api/commit.py` cannot work against the
    reference miner as written."""

    def __init__(self, miner: Any):
        self.miner = miner

    def forward(self, synapse: Prove) -> Prove:
        try:
            return Prove(index=int(synapse.index), poly=[], commitment=self.miner.rpc_commit(synapse.index, synapse.poly))
        except Exception:
            return synapse


class CommitmentAPI:
    def __init__(self, axons: Sequence[Any]):
        self.axons = list(axons)
        self.name = COMMITMENT_API_NAME

    def prepare_synapse(self, p: List[str], index: int = 0) -> Prove:
        return Prove(index=index, poly=p)

    @staticmethod
    def select_commitment(outputs: List[Any]) -> Optional[str]:
        valid = [o for o in outputs if isinstance(o, str) and o]
        return valid[0] if valid else None

    def process_responses(self, responses: List[Any]) -> str:
        return self.select_commitment([getattr(r, "commitment", None) for r in responses]) or ""

    def __call__(self, axons: Sequence[Any], poly: List[str], index: int = 0) -> str:
        synapse = self.prepare_synapse(poly, index)
        return self.process_responses([a.forward(synapse) for a in axons])


def commit(poly: List[str], axons: Sequence[Any], index: int = 0, k: int = 3, rng: Optional[random.Random] = None) -> str:
    """Commit to `poly` (row `index`) through k randomly chosen axons; returns the commitment string or ""."""
    handler = CommitmentAPI(axons)
    if not handler.axons:
        return ""
    chosen = (rng or random).choices(handler.axons, k=k)
    return handler(chosen, poly, index)
