"""Import alias: the package directory is ``gobblet-rl_amd`` (not an identifier); ``import gobblet_rl_amd``
returns that package."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("gobblet-rl_amd")
sys.modules[__name__] = _pkg
