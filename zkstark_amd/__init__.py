"""zkstark_amd -- MI355X-native STARK-101 prover path (gfx950 HIP kernels behind a C ABI).

Host-side mirror of the reference crate's interface for this path
(Crocodoctopus/zkstark: prover.rs / proof.rs / channel.rs / merkle.rs / field.rs):

    from zkstark_amd import Channel, generate_proof
    proof = generate_proof(Channel())     # prover.rs:9
    proof.verify()                        # proof.rs:15
    proof.size()                          # proof.rs:151

All device work goes through libzkstark_amd.so; there is no CPU fallback.
"""
from ._lib import ZkError, load  # noqa: F401
from .host import (P, BatchContext, Channel, Context, Merkle, Proof, compute_root_from_path, field, generate_proof,  # noqa: F401
                   host_hash_mode, lde, ntt, probe_hash_chain, prove_many, shard_plan, shard_unique_id, ShardContext, trace_fibsq, trace_fibsq_batch)
