"""ctypes binding of ``libg2048_hip.so`` (the C ABI declared in ``include/g2048.h``).

There is exactly one compute path: the HIP library.  If it is missing, cannot be loaded, or finds
no GPU, everything here raises -- nothing falls back to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libg2048_hip.so")

ACT_RANDOM, ACT_U8, ACT_I32, ACT_I64 = 0, 1, 2, 3
OBS_U8, OBS_F16, OBS_F32 = 0, 1, 2
ABI_VERSION = 15


class G2048Error(RuntimeError):
    """A call into libg2048_hip.so failed (message from g2048_last_error())."""


class StepIO(C.Structure):
    """g2048_step_io (include/g2048.h)."""
    _fields_ = [
        ("actions", C.c_void_p),
        ("action_dtype", C.c_int32),
        ("reward", C.c_void_p),
        ("terminated", C.c_void_p),
        ("illegal", C.c_void_p),
        ("highest", C.c_void_p),
        ("terminal_boards", C.c_void_p),
        ("obs", C.c_void_p),
        ("obs_dtype", C.c_int32),
        ("boards_out", C.c_void_p),
    ]


class HostIO(C.Structure):
    """g2048_host_io (include/g2048.h): host addresses of the engine's pinned, device-mapped I/O block."""
    _fields_ = [
        ("actions", C.c_void_p),
        ("reward", C.c_void_p),
        ("terminated", C.c_void_p),
        ("illegal", C.c_void_p),
        ("highest", C.c_void_p),
        ("boards", C.c_void_p),
        ("terminal_boards", C.c_void_p),
        ("scores", C.c_void_p),
    ]


class Stats(C.Structure):
    """g2048_stats (include/g2048.h)."""
    _fields_ = [
        ("episodes", C.c_uint64),
        ("illegal_ends", C.c_uint64),
        ("last_count", C.c_uint64),
        ("last_score_sum", C.c_int64),
        ("last_score_max", C.c_int32),
        ("max_exp", C.c_uint32),
        ("highest_hist", C.c_uint32 * 32),
        ("return_sum", C.c_int64),
    ]


_E = C.c_void_p      # g2048_engine*
_S = C.c_void_p      # hipStream_t
_u64, _u32, _i32 = C.c_uint64, C.c_uint32, C.c_int32

# name -> (restype, argtypes); the single source of truth for tests/test_abi.py as well
SIGNATURES = {
    "g2048_last_error": (C.c_char_p, []),
    "g2048_abi_version": (C.c_int, []),
    "g2048_create": (C.c_int, [_u64, C.c_int, _u64, _u64, C.POINTER(_E)]),
    "g2048_destroy": (C.c_int, [_E]),
    "g2048_seed": (C.c_int, [_E, _u64, _S]),
    "g2048_get_clock": (C.c_int, [_E, C.POINTER(_u64)]),
    "g2048_set_clock": (C.c_int, [_E, _u64]),
    "g2048_num_boards": (_u64, [_E]),
    "g2048_set_illegal_move_reward": (C.c_int, [_E, C.c_float]),
    "g2048_set_max_tile": (C.c_int, [_E, C.c_int]),
    "g2048_set_strict_actions": (C.c_int, [_E, C.c_int]),
    "g2048_get_strict_actions": (C.c_int, [_E]),
    "g2048_reset": (C.c_int, [_E, C.c_int, _u32, C.c_void_p, _S]),
    "g2048_step": (C.c_int, [_E, C.POINTER(StepIO), C.c_int, _S]),
    "g2048_host_io_map": (C.c_int, [_E, C.POINTER(HostIO)]),
    "g2048_step_host": (C.c_int, [_E, C.c_int, _S]),
    "g2048_fetch_host": (C.c_int, [_E, _S]),
    "g2048_stream_signal": (C.c_int, [_E, _S, C.POINTER(_u64)]),
    "g2048_stream_wait": (C.c_int, [_E, _u64, _S]),
    "g2048_rollout": (C.c_int, [_E, _u32, C.POINTER(StepIO), _u64, C.c_int, _S]),
    "g2048_rollout_prepare": (C.c_int, [_E, _u32, C.POINTER(StepIO), _u64, C.c_int]),
    "g2048_set_chains": (C.c_int, [_E, C.c_int]),
    "g2048_get_chains": (C.c_int, [_E]),
    "g2048_get_chains_used": (C.c_int, [_E]),
    "g2048_get_graph_replays": (C.c_uint64, [_E]),
    "g2048_graph_status": (C.c_char_p, [_E]),
    "g2048_rollout_fused": (C.c_int, [_E, _u32, C.POINTER(StepIO), _u64, C.c_int, _S]),
    "g2048_rollout_random": (C.c_int, [_E, _u32, _S]),
    "g2048_move": (C.c_int, [_E, C.c_void_p, _i32, C.c_int, C.c_void_p, C.c_void_p, _S]),
    "g2048_query": (C.c_int, [_E, C.c_void_p, C.c_void_p, _S]),
    "g2048_legal_actions": (C.c_int, [_E, C.c_void_p, _S]),
    "g2048_add_tile": (C.c_int, [_E, _u32, _S]),
    "g2048_fill_random_actions": (C.c_int, [_E, _u64, _u32, C.c_void_p, _S]),
    "g2048_onehot": (C.c_int, [_E, C.c_void_p, _i32, _S]),
    "g2048_get_boards": (C.c_int, [_E, C.c_void_p, _S]),
    "g2048_set_boards": (C.c_int, [_E, C.c_void_p, _S]),
    "g2048_get_scores": (C.c_int, [_E, C.c_void_p, _S]),
    "g2048_set_scores": (C.c_int, [_E, C.c_void_p, _S]),
    "g2048_get_last_scores": (C.c_int, [_E, C.c_void_p, _S]),
    "g2048_set_last_records": (C.c_int, [_E, C.c_int, _S]),
    "g2048_get_last_records": (C.c_int, [_E]),
    "g2048_records_ptr": (C.c_void_p, [_E]),
    "g2048_last_records_ptr": (C.c_void_p, [_E]),
    "g2048_episode_stats": (C.c_int, [_E, C.POINTER(Stats), _S]),
    "g2048_episode_stats_async": (C.c_int, [_E, C.c_void_p, _S]),
    "g2048_returns_summary_async": (C.c_int, [_E, C.c_void_p, _S]),
    "g2048_set_numpy_rng": (C.c_int, [_E, C.c_void_p, _S]),
    "g2048_get_numpy_rng": (C.c_int, [_E, C.c_void_p, _S]),
    "g2048_seed_numpy": (C.c_int, [_E, _u64, _S]),
    "g2048_augment": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _u64, C.c_void_p, C.c_void_p, C.c_void_p, _S]),
    "g2048_state_bytes": (_u64, [_E]),
    "g2048_get_state": (C.c_int, [_E, C.c_void_p, _S]),
    "g2048_set_state": (C.c_int, [_E, C.c_void_p, _u64, _S]),
    "g2048_canonicalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _u64, C.c_void_p, _S]),
    "g2048_comm_unique_id": (C.c_int, [C.c_void_p]),
    "g2048_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "g2048_comm_destroy": (C.c_int, [C.c_void_p]),
    "g2048_allgather_returns": (C.c_int, [_E, C.c_void_p, C.c_void_p, _S]),
    "g2048_allgather_summary": (C.c_int, [_E, C.c_void_p, C.c_void_p, _S]),
    "g2048_comm_world": (C.c_int, [C.c_void_p]),
    "g2048_comm_rank": (C.c_int, [C.c_void_p]),
    "g2048_comm_local_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "g2048_comm_local_destroy": (C.c_int, [C.c_void_p]),
    "g2048_allgather_returns_local": (C.c_int, [C.c_void_p, C.POINTER(_E), C.POINTER(C.c_void_p), C.POINTER(_S)]),
}
COMM_ID_BYTES = 128

_lib = None


def load():
    """Load libg2048_hip.so (once).  Raises G2048Error when it is not built or not loadable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise G2048Error(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the machine
        raise G2048Error(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.g2048_abi_version() != ABI_VERSION:
        raise G2048Error(f"ABI mismatch: library {lib.g2048_abi_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise G2048Error(f"libg2048_hip error {rc}: {load().g2048_last_error().decode(errors='replace')}")
