"""Import alias: the package lives in the directory ``ray-tracer_amd`` (not a
valid Python identifier), so ``import ray_tracer_amd`` resolves to it here."""
import importlib
import sys
from pathlib import Path

_root = str(Path(__file__).resolve().parent)
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("ray-tracer_amd")
sys.modules[__name__] = _pkg
