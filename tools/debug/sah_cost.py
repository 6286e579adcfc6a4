"""Host only: surface-area estimate of the fast tree's cost for the hippo scene under the builder's tuning variables."""
import ctypes as C, os, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import importlib
rta = importlib.import_module("ray-tracer_amd")
bvh = rta.init_environment()
out = (C.c_uint32 * 8)()
d = bvh.desc()
rta._lib.check(rta._lib.lib().rt_fast_tree_check(C.byref(d), 0, 1, out))
print({k: os.environ[k] for k in os.environ if k.startswith("RT_TUNE")}, list(out))
