"""Build *reference* environments (imported from /root/reference through oracle/ref_shim).

TEST INFRASTRUCTURE ONLY: used by oracle/gen_golden.py (fixture generation) and by the optional
`reference`-marked cross-check tests that run in the build container. Never used on the GPU box.

The cfg tree mirrors the fields the reference's env path reads (reference
control_pcgrl/configs/config.py:253-320, rl/envs.py:28-81).
"""
import os
import sys
from types import SimpleNamespace as NS

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, "ref_shim"))
import install as _install  # noqa: E402

# weights exactly as the reference's task configs give them
# (configs/task/binary.yaml:5-7, configs/task/zelda.yaml:8-16, configs/config.py:94-102, :161-167)
TASK_WEIGHTS = {
    "binary": {"path-length": 1, "regions": 1},
    "zelda": {"player": 3, "key": 3, "door": 3, "regions": 5, "enemies": 1, "nearest-enemy": 2, "path-length": 1},
    "sokoban": {"player": 3, "crate": 2, "target": 2, "regions": 5, "ratio": 2, "dist-win": 0, "sol-length": 1},
    "minecraft_3D_maze": {"path-length": 100, "n_jump": 100, "regions": 0},
}


def available():
    return _install.install()


def make_cfg(problem, representation, map_shape, obs_window=None, weights=None, max_board_scans=3,
             change_percentage=None):
    map_shape = tuple(map_shape)
    if obs_window is None:
        # rl/utils.py:302-334 default is 2*map_shape; wide needs obs_window == map_shape
        obs_window = map_shape if representation == "wide" else tuple(2 * s for s in map_shape)
    if weights is None:
        weights = dict(TASK_WEIGHTS[problem])
    return NS(
        render_mode=None, render=False, infer=False, evaluate=False, evaluation_env=False,
        controls=None, change_percentage=change_percentage, max_board_scans=max_board_scans,
        n_aux_tiles=0, static_prob=None, n_static_walls=None, static_tile_wrapper=False,
        act_window=None, show_agents=False, train_reward_model=False,
        representation=representation, env_name=f"{problem}-{representation}-v0",
        task=NS(name=problem, problem=problem, map_shape=map_shape, obs_window=tuple(obs_window),
                weights=weights, controls=None, alp_gmm=False),
        multiagent=NS(n_agents=0, policies="centralized"),
        model=NS(name=None),
    )


def make_reference_env(cfg, seed=None):
    """C1-C4: rl/envs.py:make_env. C5 (3-D maze): ControlWrapper(PcgrlEnv3D) because the reference's
    image wrappers crash on that problem."""
    assert available(), "reference tree not present"
    import control_pcgrl  # noqa: F401  (registers env ids)
    if "3D" in cfg.task.problem:
        import gymnasium as gym
        from control_pcgrl.control_wrappers import ControlWrapper
        env = ControlWrapper(gym.make(cfg.env_name, cfg=cfg), ctrl_metrics=None, cfg=cfg)
    else:
        from control_pcgrl.rl.envs import make_env
        env = make_env(cfg)
    if seed is not None:
        env.unwrapped.seed(int(seed))
    return env
