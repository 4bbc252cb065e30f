"""gobblet-rl_amd -- MI355X-native batched Gobblet Gobblers environment.

    BatchedGobblet   lockstep vector env over N boards in HBM (the throughput path)
    BatchedBoard     the reference ``Board`` interface over N boards
    gobblet_v1       ``env() / raw_env()``: the reference's single-env AEC surface over the same engine
    GreedyGobbletPolicy  the reference's depth-1/2 lookahead policy, batched

The compute path is the hand-written HIP library ``csrc/libgobblet_hip.so`` (C-ABI in
``include/gobblet_hip.h``); there is no CPU fallback.  Importing this package needs torch;
using it needs an MI355X.
"""
from . import _native  # noqa: F401
from . import gobblet_v1  # noqa: F401
from ._native import GobbletHipError, build  # noqa: F401
from .board import BatchedBoard  # noqa: F401
from .vector_env import BatchedGobblet  # noqa: F401
from .greedy_policy import GreedyGobbletPolicy  # noqa: F401
from .random_policy import RandomAdmissiblePolicy  # noqa: F401
from .sharding import make_shard, reduce_counters, shard_bounds  # noqa: F401

__version__ = "0.1.0"
