"""Import alias: the package sources live in ``control-pcgrl_amd/`` (a hyphen is not importable).

``import control_pcgrl_amd`` resolves sub-modules from that directory and executes its ``__init__``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "control-pcgrl_amd")
__path__[:] = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f, _os
