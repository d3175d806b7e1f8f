"""Import alias: the real package lives in ../eeg-gnn-ssl_amd/ (hyphens are not importable)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "eeg-gnn-ssl_amd")]
_init = _os.path.join(__path__[0], "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
