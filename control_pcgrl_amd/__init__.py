"""control_pcgrl_amd -- MI355X-native batched PCGRL environment engine (host side).

Drop-in for the reference's env hot path only (control_pcgrl/rl/envs.py:make_env and what it builds):
  make_env(cfg)            single-env adapter with the reference's reset()/step() tuple shapes
  make_vec_env(cfg, n)     batched engine: torch tensors in/out, one HIP launch per step for all envs
  VecPcgrlEnv              the batched env class
  PcgrlVectorEnv           the same batch behind ray.rllib's VectorEnv call shape (vector_step / reset_at ...)
The compute lives in csrc/libpcgrl_amd.so (hand-written HIP for gfx950) behind the C ABI of
include/pcgrl_amd.h; this package fails loudly if that library is missing -- there is no CPU fallback.
"""
from .problems import PROBLEMS, REPRESENTATIONS, ProblemSpec, problem_spec  # noqa: F401
from .vec_env import SubBatchedVecEnv, VecPcgrlEnv, make_vec_env  # noqa: F401
from .envs import make_env, PcgrlGymEnv  # noqa: F401
from .rllib_env import PcgrlVectorEnv  # noqa: F401
from .dist import EpisodeStatsReducer, shard_env_range  # noqa: F401

__version__ = "0.6.0"  # csrc/pcgrl_engine.hip pcgrl_version() carries the same number
