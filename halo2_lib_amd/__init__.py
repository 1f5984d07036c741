"""Import alias: the package directory is `halo2-lib_amd/` (not a valid Python identifier), so
`import halo2_lib_amd` maps onto it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "halo2-lib_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
