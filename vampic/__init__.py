"""Importable alias of ``efficient-pic-with-variance-aware-masking_amd/`` (a directory name
with hyphens cannot be imported directly): this package's search path IS that directory."""
import os as _os

# VAMPIC_PKG_DIR: alternative package directory (A/B experiments against an older checkout of the package)
__path__ = [_os.environ.get("VAMPIC_PKG_DIR") or
            _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                          "efficient-pic-with-variance-aware-masking_amd")]
_init = _os.path.join(__path__[0], "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
