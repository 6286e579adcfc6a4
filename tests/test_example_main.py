"""examples/render_main.cpp: the reference's main() (main.rs:19-106) written against the C++ host
mirror. Built by `make -C ray-tracer_amd/csrc example` (part of __graft_entry__.build())."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "examples" / "render_main"
MESH = ROOT / "ray-tracer_amd" / "assets" / "hippo.rtmesh"


def run(args):
    return subprocess.run([str(EXE)] + [str(a) for a in args], capture_output=True, text=True, timeout=300)


def test_example_is_built_and_checks_its_arguments(tmp_path):
    if not EXE.exists():  # a fresh checkout: the same command __graft_entry__.build() runs
        subprocess.run(["make", "-C", str(ROOT / "ray-tracer_amd" / "csrc"), "-j8", "all", "example"], check=True)
    assert EXE.exists(), "build it with `make -C ray-tracer_amd/csrc example`"
    assert run([]).returncode == 2
    r = run([tmp_path / "missing.rtmesh", tmp_path / "x.ppm", 8, 1])
    assert r.returncode == 1 and "cannot open" in r.stderr


@pytest.mark.gpu
def test_example_renders_the_same_picture_as_the_python_host(rta, hippo_bvh, tmp_path):
    res, spp, seed = 48, 4, 9
    out = tmp_path / "hippo.ppm"
    r = run([MESH, out, res, spp, seed])
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["primitives"] == 46206 and info["nodes"] == 30113 and info["paths"] == res * res * 5 * spp
    raw = out.read_bytes()
    header = f"P6\n{res} {res}\n255\n".encode()
    assert raw.startswith(header)
    rgb = np.frombuffer(raw[len(header):], dtype=np.uint8).reshape(res, res, 3)
    _, rgba, st = rta.render(hippo_bvh, rta.reference_camera(), res, res, spp, seed=seed)
    assert np.array_equal(rgb, rgba[..., :3]) and int(st["rays"]) == info["rays"]
    assert rgb.max() > 0
    # the reference's own output format: image.png, RGBA8 (main.rs:242-244)
    from test_host_flatten import decode_png
    png = tmp_path / "image.png"
    assert run([MESH, png, res, spp, seed]).returncode == 0
    assert np.array_equal(decode_png(png.read_bytes()), rgba)
