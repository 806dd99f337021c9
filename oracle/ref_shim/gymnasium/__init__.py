"""Minimal stand-in for the `gymnasium` API surface that control-pcgrl's env path touches.

TEST INFRASTRUCTURE ONLY. This lets `oracle/gen_golden.py` import the *reference* package from
/root/reference inside the build container (gymnasium itself is not installed and there is no
network). It is our own code, not a copy of gymnasium; only the behaviour the reference relies on is
provided:
  * Env / Wrapper with attribute forwarding (public attributes forwarded, `_private` refused,
    observation_space / action_space overridable per wrapper),
  * spaces.Box / Discrete / MultiDiscrete / Dict / Tuple as plain shape/bounds holders,
  * envs.registration.register / make,
  * utils.seeding.np_random with gymnasium-0.27.1's definition
    (PCG64(SeedSequence(seed))) -- the only third-party arithmetic that affects grid state.
"""
from . import spaces  # noqa: F401
from .core import Env, Wrapper, ObservationWrapper, ActionWrapper, RewardWrapper  # noqa: F401
from .envs.registration import make, register, registry  # noqa: F401
from . import utils  # noqa: F401
from . import wrappers  # noqa: F401

__version__ = "0.27.1-shim"
