"""Import shim: the package directory is named `halo-accumulation_amd` (not a Python identifier).

`import halo_accumulation_amd` loads that directory as a regular package.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "halo-accumulation_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _os, _f
