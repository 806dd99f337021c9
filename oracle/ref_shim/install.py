"""Make the *reference* package importable in the build container.

TEST INFRASTRUCTURE ONLY (used by oracle/gen_golden.py and the optional reference cross-check
tests; never by the product path, never on the GPU box where /root/reference does not exist).

control-pcgrl imports gymnasium, ray, hydra, omegaconf, cv2, ... at module import time; none of them
is installed here and there is no network. `install()` puts our tiny `gymnasium` stand-in first on
sys.path and pre-seeds sys.modules with inert stubs for everything else the env path imports but
never executes on the reset()/step() path.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PCGRL_REFERENCE_ROOT", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))


class _Inert(types.ModuleType):
    """Module whose every missing attribute is a harmless callable/class factory."""

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        val = _make_dummy(f"{self.__name__}.{name}")
        setattr(self, name, val)
        return val


def _make_dummy(qualname):
    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            # usable as decorator: @dummy / @dummy(...)
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return self

        def __getattr__(self, n):
            if n.startswith("__") and n.endswith("__"):
                raise AttributeError(n)
            return _make_dummy(f"{qualname}.{n}")

    _Dummy.__name__ = qualname.rsplit(".", 1)[-1]
    _Dummy.__qualname__ = qualname
    return _Dummy


class _InertFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Resolve `import <prefix>.anything` to an inert module for the listed top-level packages."""

    def __init__(self, prefixes):
        self.prefixes = tuple(prefixes)

    def find_spec(self, fullname, path=None, target=None):
        top = fullname.split(".")[0]
        if top in self.prefixes or fullname in self.prefixes:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Inert(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        _specialise(module)


def _override(*a, **k):
    def deco(fn):
        return fn
    return deco


def _specialise(module):
    name = module.__name__
    if name == "ray":
        module.remote = lambda *a, **k: (a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f))
        module.get = lambda x: x
    elif name == "ray.rllib":
        class MultiAgentEnv:  # base class only
            pass
        module.MultiAgentEnv = MultiAgentEnv
    elif name == "ray.rllib.env.apis.task_settable_env":
        class TaskSettableEnv:
            pass
        module.TaskSettableEnv = TaskSettableEnv
    elif name == "ray.rllib.env.env_context":
        class EnvContext(dict):
            pass
        module.EnvContext = EnvContext
    elif name == "ray.rllib.utils.annotations":
        module.override = _override
    elif name == "hydra.core.config_store":
        class ConfigStore:
            @staticmethod
            def instance():
                return ConfigStore()

            def store(self, *a, **k):
                pass
        module.ConfigStore = ConfigStore
    elif name == "omegaconf":
        module.MISSING = "???"
    elif name == "control_pcgrl.envs.probs.minecraft.utils":
        # the real one writes into the (read-only) source tree
        module.patch_grpc_evocraft_imports = lambda *a, **k: None


_INERT_TOP = (
    "ray", "hydra", "omegaconf", "gym", "turtle", "pyscreenshot", "cv2", "imageio", "wandb", "pyglet",
    "neat", "tkinter", "gi", "grpc_tools", "submitit", "qdpy", "ribs", "opensimplex", "matplotlib",
    "seaborn", "pandas_stub_never",
)
_INERT_EXACT = (
    "control_pcgrl.reward_model_wrappers",
    "control_pcgrl.envs.probs.minecraft.mc_render",
    "control_pcgrl.envs.probs.minecraft.minecraft_pb2",
    "control_pcgrl.envs.probs.minecraft.minecraft_pb2_grpc",
    "control_pcgrl.envs.probs.minecraft.gl_render",
    "control_pcgrl.envs.probs.minecraft.utils",
)

_installed = False


def install():
    """Idempotent. Returns True when the reference tree is present and wired up."""
    global _installed
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "control_pcgrl")):
        return False
    if _installed:
        return True
    if _HERE not in sys.path:
        sys.path.insert(0, _HERE)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(1, REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # the reference tree is read-only
    sys.meta_path.insert(0, _InertFinder(_INERT_TOP + _INERT_EXACT))
    _installed = True
    return True
