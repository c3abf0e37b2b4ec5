"""`Prove`: field-for-field mirror of the reference wire type (reference base/protocol.py:24-63), minus the
bittensor `Synapse` base class (bittensor is not installable here).  Field names, optionality and encodings are
identical, so a `bt.Synapse` subclass can adopt this class body unchanged."""
from __future__ import annotations

from typing import List, Optional

from pydantic import BaseModel, Field


class Prove(BaseModel):
    index: int = Field(..., title="Worker Index", description="The Index that the miner should use to identify itself.",
                       frozen=True)
    poly: List[str] = Field(..., title="Polynomial", description="The polynomial to prove.", frozen=True)
    alpha: Optional[str] = Field(default=None, title="Input", description="The input to evaluate the polynomial at.")
    eval: Optional[str] = Field(default=None, title="Evaluation",
                                description="The evaluation of the polynomial at the input.")
    commitment: Optional[str] = Field(default=None, title="Commitment", description="The commitment to the polynomial.")
    proof: Optional[str] = Field(default=None, title="Proof", description="The proof of the commitment.")

    def deserialize(self) -> "Prove":
        return self
