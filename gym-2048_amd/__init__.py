"""gym-2048_amd -- MI355X-native batched 2048 environment (import name: ``gym2048_amd``).

The hot path of rgal/gym-2048 -- ``Game2048Env.step()/reset()`` and what they call -- as hand-written
HIP kernels for gfx950 behind a C ABI (``include/g2048.h``), with the reference's Gymnasium surface on
top.  See DESIGN.md / INTEGRATION.md.
"""
from ._lib import G2048Error, LIB_PATH  # noqa: F401
from .env import Game2048Env, IllegalMove, stack  # noqa: F401
from .vec_env import Vec2048  # noqa: F401
from .sharding import Shard, LocalShards, shard_range, weak_shard, allgather_returns  # noqa: F401
from .evaluate import evaluate_model, report_evaluation_results  # noqa: F401

__version__ = "0.1.0"
ENV_ID = "2048-v0"  # the reference's registration id (env/__init__.py:3-6)


def __getattr__(name):
    if name == "Batched2048":  # needs torch + a GPU; imported on first use
        from .batched import Batched2048
        return Batched2048
    raise AttributeError(name)


def register(entry_point="gym2048_amd:Game2048Env", force=True):
    """Register ``'2048-v0'`` with gymnasium like the reference's ``env/__init__.py:1-6``.

    ``force=False`` (what the import-time registration uses): when the id already points at ANOTHER entry point -- the
    reference's own ``env`` package imported in the same process registers the same id -- that registration is KEPT and a
    ``UserWarning`` says so (``gym.make("2048-v0")`` would otherwise silently change meaning with the import order);
    call ``gym2048_amd.register()`` explicitly to take the id over.  Returns True when the id now points here."""
    from gymnasium.envs import registration
    existing = getattr(registration, "registry", {}).get(ENV_ID) if hasattr(getattr(registration, "registry", None), "get") else None
    current = getattr(existing, "entry_point", None)
    if existing is not None and current != entry_point and not force:
        import warnings
        warnings.warn(f"gym2048_amd: '{ENV_ID}' is already registered with entry point {current!r}; keeping it.  "
                      f"Call gym2048_amd.register() to point '{ENV_ID}' at the MI355X engine instead.", UserWarning, stacklevel=2)
        return False
    registration.register(id=ENV_ID, entry_point=entry_point)
    return True


# env/__init__.py:1-6 registers '2048-v0' when the package is imported, and its callers rely on that
# (ppo_train.py:102 `gym.make("2048-v0", render_mode="rgb_array")`, :123 `make_vec_env("2048-v0", ...)`): so does this
# package, whenever gymnasium is importable.  Without gymnasium there is nothing to register with.
try:
    import gymnasium as _gymnasium  # noqa: F401
except ImportError:
    _gymnasium = None
if _gymnasium is not None:
    register(force=False)
del _gymnasium
