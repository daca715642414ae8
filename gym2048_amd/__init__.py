"""Import alias: the package directory is ``gym-2048_amd/`` (not a valid Python identifier), so
``import gym2048_amd`` resolves here and loads the real package from that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gym-2048_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
