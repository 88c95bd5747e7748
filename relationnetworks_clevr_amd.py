"""Import shim: exposes the hyphenated package directory `relationnetworks-clevr_amd/` as the
importable package `relationnetworks_clevr_amd` (a module that sets __path__ is a package)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "relationnetworks-clevr_amd")]
__package__ = __name__
_init = _os.path.join(__path__[0], "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
