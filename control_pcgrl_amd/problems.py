"""Per-problem constant tables of the hot path (tile vocabularies, targets, bounds, default weights).

Host-side mirror of the reference's Problem classes, evaluated for a given map shape; file:line
references are relative to the reference's control_pcgrl/ directory.  Only data lives here -- the
statistics themselves are computed by the HIP kernels.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple, Union

PROBLEMS = {"binary": 0, "zelda": 1, "sokoban": 2, "minecraft_3D_maze": 3}
REPRESENTATIONS = {"narrow": 0, "turtle": 1, "wide": 2}

Target = Union[float, Tuple[float, float]]


@dataclass
class ProblemSpec:
    name: str
    tile_types: List[str]
    stat_keys: List[str]            # column order of every `stats` tensor
    static_trgs: Dict[str, Target]  # Problem.static_trgs
    cond_bounds: Dict[str, Tuple[float, float]]
    default_weights: Dict[str, float]  # cfg.task.weights of the reference's task config
    problem_weights: Dict[str, float] = field(default_factory=dict)  # Problem._reward_weights (keys matter)

    @property
    def n_tiles(self):
        return len(self.tile_types)


def _binary(h, w):
    # envs/probs/binary/binary_prob.py:17 tiles; :50 max path (zig-zag); :59-63 targets; :66-84 bounds
    max_path = math.ceil(w / 2) * h + math.floor(h / 2)
    return ProblemSpec(
        "binary", ["empty", "solid"], ["regions", "path-length"],
        {"regions": 1, "path-length": max_path},
        {"regions": (0, w * math.ceil(h / 2)), "path-length": (0, max_path)},
        {"path-length": 1, "regions": 1},  # configs/task/binary.yaml:5-7
        {"regions": 100, "path-length": 100},
    )


def _zelda(h, w):
    # envs/probs/zelda/zelda_prob.py:20 tiles, :30 _max_enemies; zelda_ctrl_prob.py:19-73
    max_nearest = math.ceil(w / 2 + 1) * h
    max_path = (math.ceil(w / 2) * h + math.floor(h / 2)) * 2 - 1
    n = w * h
    return ProblemSpec(
        "zelda", ["empty", "solid", "player", "key", "door", "bat", "scorpion", "spider"],
        ["player", "key", "door", "enemies", "regions", "nearest-enemy", "path-length"],
        {"enemies": (2, 5), "path-length": max_path, "nearest-enemy": (5, max_nearest), "regions": 1, "player": 1,
         "key": 1, "door": 1},
        {"nearest-enemy": (0, max_nearest), "enemies": (0, n - 2), "player": (0, n - 2), "key": (0, n - 2),
         "door": (0, n - 2), "regions": (0, n / 2), "path-length": (0, max_path)},
        {"player": 3, "key": 3, "door": 3, "regions": 5, "enemies": 1, "nearest-enemy": 2, "path-length": 1},
        {"player": 3, "key": 3, "door": 3, "regions": 5, "enemies": 1, "nearest-enemy": 1, "path-length": 1},
    )


def _sokoban(h, w):
    # envs/probs/sokoban/sokoban_prob.py:26 tiles; :30-31 sets _width=_height=5 BEFORE sokoban_ctrl_prob.py:13,
    # :27-49 derive targets and bounds, so those are frozen at 5x5 whatever the map size.
    fw = fh = 5
    max_path = math.ceil(fw / 2 + 1) * fh
    return ProblemSpec(
        "sokoban", ["empty", "solid", "player", "crate", "target"],
        ["player", "crate", "target", "regions", "dist-win", "sol-length", "ratio"],
        {"player": 1, "crate": (2, 3), "regions": 1, "ratio": 0, "dist-win": 0, "sol-length": max_path},
        {"player": (1, fw * fh), "crate": (1, fw * fh / 2 - max(fw, fh)), "target": (1, fw * fh), "ratio": (0, fw * fh),
         "dist-win": (0, fw * fh * (fw + fh)), "sol-length": (0, 2 * max_path), "regions": (0, fw * fh / 2)},
        {"player": 3, "crate": 2, "target": 2, "regions": 5, "ratio": 2, "dist-win": 0, "sol-length": 1},
        {"player": 3, "crate": 1, "regions": 5, "ratio": 2, "dist-win": 0.0, "sol-length": 1},
    )


def _mc3dmaze(shape):
    # envs/probs/minecraft/minecraft_3D_maze_prob.py:26 tiles; :33-35 sizes frozen at 15 (adjust_param only
    # updates _length from kwargs, :108); :41-58 targets and bounds
    length = width = height = 15
    per_floor = math.ceil(width / 2) * length + math.floor(length / 2)
    max_path = 2 * (height // 3) * per_floor
    return ProblemSpec(
        "minecraft_3D_maze", ["AIR", "DIRT"], ["regions", "path-length", "n_jump"],
        {"regions": 1, "path-length": 10 * max_path, "n_jump": 5},
        {"regions": (0, math.ceil(width * length / 2 * height)), "path-length": (0, max_path),
         "n_jump": (0, max_path // 2)},
        {"path-length": 100, "n_jump": 100, "regions": 0},
        {"regions": 0, "path-length": 100, "n_jump": 100},
    )


def problem_spec(problem: str, map_shape) -> ProblemSpec:
    map_shape = tuple(int(s) for s in map_shape)
    if problem == "binary":
        return _binary(*map_shape)
    if problem == "zelda":
        return _zelda(*map_shape)
    if problem == "sokoban":
        return _sokoban(*map_shape)
    if problem == "minecraft_3D_maze":
        return _mc3dmaze(map_shape)
    raise ValueError(f"problem '{problem}' is outside the accelerated hot path (supported: {sorted(PROBLEMS)})")


def target_interval(trg: Target):
    """control_wrappers.py:334-341: a tuple target (lo, hi) means min |arange(lo, hi) - val| -- the upper
    bound is excluded and the grid is integer -- so the zero-loss interval is [lo, last element]."""
    if isinstance(trg, tuple):
        lo, hi = trg
        n = int(math.ceil(hi - lo))
        if n < 1:  # e.g. zelda's nearest-enemy range (5, ceil(w / 2 + 1) * h) on a 1 x 5 map
            raise ValueError(f"empty target range {trg}: the reference's get_loss fails on it too (min of an empty "
                             "arange, control_wrappers.py:339) -- the map is too small for this problem's static targets")
        return float(lo), float(lo + n - 1)
    return float(trg), float(trg)
