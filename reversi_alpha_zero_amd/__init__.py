"""Import alias: the product package lives in the directory `reversi-alpha-zero_amd/` (the name the
build contract fixes), which is not a valid Python identifier.  This shim makes it importable as
`reversi_alpha_zero_amd` by pointing the package search path at that directory and executing its
__init__ in this namespace.  No code lives here."""
import os as _os

_real_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                          "reversi-alpha-zero_amd")
__path__ = [_real_dir]
_init = _os.path.join(_real_dir, "__init__.py")
with open(_init, "rt") as _f:
    exec(compile(_f.read(), _init, "exec"))
del _os, _f, _init
