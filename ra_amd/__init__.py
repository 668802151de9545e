"""ra_amd -- host side of ra_gpu_batch, the MI355X-native batched evaluator of rabbitmq/ra's
per-server Raft transition (ra_server:handle_leader/2, handle_follower/2, ...).

The compute path is the HIP library ra_amd/csrc/libra_gpu_batch.so behind the C ABI declared
in include/ra_gpu_batch.h.  There is no CPU fallback: importing the engine without the built
library, or opening it without a GPU, raises.
"""
from . import abi  # noqa: F401

__all__ = ["abi"]
