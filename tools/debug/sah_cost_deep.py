"""Host only: surface-area estimate of the fast tree's cost for the 11.9 M primitive scene."""
import ctypes as C, os, sys, pathlib, importlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
rta = importlib.import_module("ray-tracer_amd")
t0 = time.time()
bvh = rta.scenes.build(rta.scenes.subdivided_scene(4, 100_000))
out = (C.c_uint32 * 8)()
d = bvh.desc()
print("scene", round(time.time() - t0, 1), "s", flush=True)
t0 = time.time()
rta._lib.check(rta._lib.lib().rt_fast_tree_check(C.byref(d), 0, 8, out))
print({k: os.environ[k] for k in os.environ if k.startswith("RT_TUNE")}, list(out), round(time.time() - t0, 1), "s")
