"""Hot-path half of the reference miner (reference neurons/miner.py:35-61,106-135 and the prover lifecycle of
base/miner.py:73-84,155,181), with `self.client` built from zkp_subnet_amd.Client instead of fourier.Client.
Same method names, arguments and error behaviour; the chain-policy half (axon, blacklist, priority, metagraph) is
out of scope."""
from __future__ import annotations

import logging
import time
import typing
from types import SimpleNamespace

from .client import Client
from .protocol import Prove

log = logging.getLogger("zkp_subnet_amd.miner")


def default_config(**over) -> SimpleNamespace:
    """The six prover flags of the reference (utils/config.py:124-164), same names and defaults.  `seed` / `synthetic`
    are test-only: without them a missing setup file makes the miner fail at start, like the reference prover."""
    cfg = SimpleNamespace(prover_path="./prover", uncompressed=False, setup_path="./setup",
                          precompute_path="./precompute", scale=18, machines_scale=8, device=0, seed=None,
                          synthetic=None, fused=False, workers=None)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


class Miner:
    def __init__(self, config=None, client: typing.Optional[Client] = None):
        self.config = config or default_config()
        # reference base/miner.py:73-84
        PORT = 1337
        self.client = client or Client(
            port=PORT,
            bin=self.config.prover_path,
            uncompressed=self.config.uncompressed,
            setup_path=self.config.setup_path,
            precompute_path=self.config.precompute_path,
            device=getattr(self.config, "device", 0),
            seed=getattr(self.config, "seed", None),
            workers=getattr(self.config, "workers", None),
            synthetic=getattr(self.config, "synthetic", None),
        )
        self.client.start(scale=self.config.scale, machines_scale=self.config.machines_scale)

    def stop(self) -> None:  # reference base/miner.py:155,181
        self.client.stop()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.stop()

    # ---- reference neurons/miner.py:38-45
    def rpc_commit(self, i: int, poly: typing.List[str]) -> str:
        with self.client.worker_commit(i, poly) as response:
            if response.status_code != 200:
                log.error("RPC request failed with status: %s", response.status_code)
                raise Exception("Failed to commit to the polynomial.")
            return response.json().get("commitment")

    # ---- reference neurons/miner.py:47-54
    def rpc_open(self, i: int, poly: typing.List[str], x: str) -> typing.Tuple[str, str]:
        with self.client.worker_open(i, poly, x) as response:
            if response.status_code != 200:
                log.error("RPC request failed with status: %s", response.status_code)
                raise Exception("Failed to verify the proof.")
            return response.json().get("eval"), response.json().get("proof")

    # ---- reference neurons/miner.py:56-61 (two sequential calls); fused=True uses the single-upload extension
    def rpc_commit_and_open(self, i: int, poly: typing.List[str], alpha: str) -> typing.Tuple[str, str, str]:
        if getattr(self.config, "fused", False):
            with self.client.worker_commit_and_open(i, poly, alpha) as response:
                if response.status_code != 200:
                    log.error("RPC request failed with status: %s", response.status_code)
                    raise Exception("Failed to commit to / open the polynomial.")
                body = response.json()
                return body.get("commitment"), body.get("eval"), body.get("proof")
        commitment = self.rpc_commit(i, poly)
        eval, proof = self.rpc_open(i, poly, alpha)
        return commitment, eval, proof

    # ---- reference neurons/miner.py:106-135
    def forward(self, synapse: Prove) -> Prove:
        try:
            before = time.perf_counter()
            commitment, eval, proof = self.rpc_commit_and_open(synapse.index, synapse.poly, synapse.alpha)
            elapsed = time.perf_counter() - before
            log.info("Proof generation completed in %s seconds", elapsed)
            return Prove(index=int(synapse.index), poly=[], alpha=None, eval=eval, commitment=commitment, proof=proof)
        except Exception as e:  # echo the request; the validator scores it 0 (neurons/validator.py:146-148)
            log.error("Failed to forward synapse: %s", e)
            return synapse
