"""ctypes binding of csrc/libpcgrl_amd.so (the C ABI of include/pcgrl_amd.h).  No fallback: a missing or
stale library is an error."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libpcgrl_amd.so")
# translation units (compiled in parallel, see csrc/pcgrl_dispatch.h) and the headers they depend on
UNITS = ["pcgrl_engine.hip", "pcgrl_k_binary32.hip", "pcgrl_k_binary64.hip", "pcgrl_k_zelda32.hip", "pcgrl_k_zelda64.hip",
         "pcgrl_k_sokoban32_8.hip", "pcgrl_k_sokoban32_16.hip", "pcgrl_k_sokoban32_32.hip",
         "pcgrl_k_sokoban32_64.hip", "pcgrl_k_sokoban64_32.hip", "pcgrl_k_sokoban64_64.hip", "pcgrl_k_3d.hip"]
HEADERS = ["pcgrl_kernels2d.h", "pcgrl_kernels3d.h", "pcgrl_sokoban.h", "pcgrl_common.h", "pcgrl_dispatch.h"]
SOURCES = UNITS + HEADERS
HEADER = os.path.join(os.path.dirname(_HERE), "include", "pcgrl_amd.h")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-falign-loops=32", "-fPIC"]

PCGRL_MAX_STATS = 8
ERRORS = {1: "EINVAL", 2: "EUNSUPPORTED", 3: "EHIP", 4: "EACTION", 5: "ESTALE"}


class PcgrlConfig(C.Structure):
    _fields_ = [
        ("problem", C.c_int32), ("representation", C.c_int32), ("ndim", C.c_int32),
        ("dims", C.c_int32 * 3), ("obs_window", C.c_int32 * 3),
        ("max_iterations", C.c_int32), ("max_changes", C.c_int32), ("n_stats", C.c_int32),
        ("has_trg", C.c_int32 * PCGRL_MAX_STATS), ("weights", C.c_double * PCGRL_MAX_STATS),
        ("trg_lo", C.c_double * PCGRL_MAX_STATS), ("trg_hi", C.c_double * PCGRL_MAX_STATS),
        ("solver_power", C.c_int32),
        ("n_ctrl", C.c_int32), ("ctrl_idx", C.c_int32 * PCGRL_MAX_STATS), ("ctrl_range", C.c_double * PCGRL_MAX_STATS),
        ("act_window", C.c_int32 * 3), ("static_tiles", C.c_int32), ("n_static_walls", C.c_int32),
        ("static_eval", C.c_int32), ("static_prob", C.c_double),
    ]


SYMBOLS = {
    # name: (restype, argtypes)
    "pcgrl_create": (C.c_int, [C.POINTER(PcgrlConfig), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "pcgrl_destroy": (None, [C.c_void_p]),
    "pcgrl_seed": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pcgrl_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_step_seq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p]),
    "pcgrl_step_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_rollout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_rollout_is_one_launch": (C.c_int32, [C.c_void_p]),
    "pcgrl_set_rollout_form": (C.c_int, [C.c_void_p, C.c_int32]),
    "pcgrl_rollout_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_queue_targets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_ctrl_observe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_set_target_resampling": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p]),
    "pcgrl_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_refresh_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_observe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_get_static": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_set_static": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, C.c_int32]),
    "pcgrl_obs_bytes": (C.c_int64, [C.c_void_p]),
    "pcgrl_obs_shape": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32 * 4), C.POINTER(C.c_int32)]),
    "pcgrl_get_state": (C.c_int, [C.c_void_p] + [C.c_void_p] * 7),
    "pcgrl_get_last_episode": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5),
    "pcgrl_stats_for_grids": (C.c_int, [C.POINTER(PcgrlConfig), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_void_p]),
    "pcgrl_stats_for_grids_h": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_stats_poll_error": (C.c_int, [C.POINTER(PcgrlConfig), C.c_int32]),
    "pcgrl_stats_cache_clear": (None, []),
    "pcgrl_reduce_episodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "pcgrl_set_state": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6),
    "pcgrl_state_bytes": (C.c_int64, [C.c_void_p]),
    "pcgrl_export_state": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "pcgrl_import_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "pcgrl_get_rng_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_set_rng_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_poll_error": (C.c_int, [C.c_void_p]),
    "pcgrl_sample_actions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "pcgrl_num_actions": (C.c_int32, [C.c_void_p]),
    "pcgrl_set_solver_budget": (C.c_int, [C.c_void_p, C.c_int32]),
    "pcgrl_get_solver_budget": (C.c_int32, [C.c_void_p]),
    "pcgrl_step_ready": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "pcgrl_env_busy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcgrl_reserve_solver_pool": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "pcgrl_solver_pool_slots": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pcgrl_copy_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pcgrl_stream_synchronize": (C.c_int, [C.c_void_p]),
    "pcgrl_graph_upload": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pcgrl_debug_counters": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "pcgrl_last_error": (C.c_char_p, []),
    "pcgrl_version": (C.c_char_p, []),
}


def build(force=False, verbose=False, out=None, defines=(), jobs=None):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU), one object per translation unit, compiled in
    parallel, then linked into the shared library.  Rebuilds when a source is newer than the library.
    `out` / `defines`: development builds (tools/phase_timing.py, tools/wave_trace.py)."""
    out = out or LIB_PATH
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [HEADER]
    if (not force and os.path.exists(out)
            and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs if os.path.exists(s))):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tag = "".join(sorted(d.replace("=", "_") for d in defines))
    objdir = os.path.join(CSRC, "_obj" + ("_" + tag if tag else ""))
    os.makedirs(objdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    hdr_time = max(hdr_time, os.path.getmtime(HEADER))
    jobs_todo, objs = [], []
    for u in UNITS:
        obj = os.path.join(objdir, u.replace(".hip", ".o"))
        objs.append(obj)
        if (force or not os.path.exists(obj)
                or os.path.getmtime(obj) < max(hdr_time, os.path.getmtime(os.path.join(CSRC, u)))):
            jobs_todo.append([hipcc] + HIPCC_FLAGS + ["-D" + d for d in defines] + ["-c", os.path.join(CSRC, u), "-o", obj])
    if jobs_todo:
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

        with ThreadPoolExecutor(max_workers=jobs or min(len(jobs_todo), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs_todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
            build()  # in-tree build of the HIP extension (never a CPU fallback)
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or control_pcgrl_amd._lib.build()).  There is no CPU fallback.")
        # development: another build of the same ABI for A/B timing -- only a file of this package's csrc directory
        override = os.environ.get("PCGRL_LIB")
        if override and os.path.dirname(os.path.realpath(override)) != os.path.realpath(CSRC):
            raise RuntimeError(f"PCGRL_LIB={override}: only libraries inside {CSRC} are loaded")
        L = C.CDLL(override or LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            if override and not hasattr(L, name):
                continue
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().pcgrl_last_error().decode()
        exc = ValueError if rc in (1, 4) else (NotImplementedError if rc == 2 else RuntimeError)
        raise exc(f"{what}: PCGRL_{ERRORS.get(rc, rc)}: {msg}")
