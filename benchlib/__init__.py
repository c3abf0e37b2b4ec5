"""Parts of the repo-root bench.py (the driver's measurement contract): common helpers, the multi-rank control plane, the extra
objects of a bench line.  bench.py itself keeps the contract: arguments, the timed region, the JSON line."""
