"""genstark_amd._mirror — CHECKER, not product: the reference's *callers* restated line by line in Python.

`stark.py` is lib/Stark.ts (prove / verify / serialize / parse), `components/` are lib/components/*.ts, in the reference's own order
and with its identifiers, issuing one C-ABI call per galois / merkle member call.  They exist so that the parity tests read like the
reference's code and so that the native drivers have something independent to be compared with byte for byte: tests/, bench.py's
cross-check and __graft_entry__.smoke() import them; the product's prove() never does (csrc/prover.cc and its distributed mode
csrc/prover_dist.h through genstark_amd/prover.py, and the N-API path under the reference's own Stark.js).  `distributed.py` /
`sharded.py` are the first (Python, SPMD) forms of the multi-GPU prover, kept as a second opinion on the native one.
The verifier side (Stark.verify, lib/Stark.ts:167-248) is this restatement too: genstark_amd/prover.py's verify() delegates here.
"""
