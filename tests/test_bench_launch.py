"""bench.py starts its own ranks: `python bench.py --gpus N` (no torchrun around it) launches N child
processes through torch.distributed.run before it touches a GPU."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
SMALL = ["--steps", "1", "--warmup", "0", "--no-cpu", "--no-microbench", "--no-deep", "--width", "320", "--height",
         "200", "--spp-per-step", "2"]


def run_bench(*extra, timeout=600):
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *SMALL, *extra], capture_output=True, text=True,
                          timeout=timeout, cwd=str(ROOT))


def test_plain_start_with_two_ranks_reaches_both_ranks():
    """Without a GPU every rank stops at 'needs a HIP device': that both did shows the launch itself works."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered on the GPU by test_self_launched_ranks_render_the_same_frame")
    r = run_bench("--gpus", "2", "--backend", "gloo")
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a HIP device") >= 2, r.stderr[-2000:]


@pytest.mark.gpu
def test_self_launched_ranks_render_the_same_frame(tmp_path):
    """Two ranks on one GPU (gloo rehearsal backend), started by bench.py itself: the tile-sharded frame is byte
    for byte the single-process frame, and sample-range sharding renders twice the samples of the same image."""
    one, two, two_s = tmp_path / "one.npy", tmp_path / "two.npy", tmp_path / "two_s.npy"
    r1 = run_bench("--gpus", "1", "--save-frame", str(one))
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = run_bench("--gpus", "2", "--backend", "gloo", "--shard", "tiles", "--save-frame", str(two))
    assert r2.returncode == 0, r2.stderr[-2000:]
    l1, l2 = json.loads(r1.stdout.strip().splitlines()[-1]), json.loads(r2.stdout.strip().splitlines()[-1])
    assert l1["n_gpus"] == 1 and l2["n_gpus"] == 2 and l2["scaling"] == "strong"
    assert l1["vs_baseline"] is None and l1["metric"] == l2["metric"]
    a, b = np.load(one), np.load(two)
    assert a.shape == (200, 320, 4) and np.array_equal(a, b)
    # tiles: the same paths in total
    assert abs(l2["paths_per_s"] * l2["ms_per_step"] - l1["paths_per_s"] * l1["ms_per_step"]) < 1e-6 * l1["paths_per_s"] * l1["ms_per_step"]
    r3 = run_bench("--gpus", "2", "--backend", "gloo", "--shard", "samples", "--save-frame", str(two_s))
    assert r3.returncode == 0, r3.stderr[-2000:]
    l3 = json.loads(r3.stdout.strip().splitlines()[-1])
    assert l3["scaling"] == "weak"
    # each rank renders the whole frame for its own sample range: twice the paths, same picture up to noise
    # (2 against 4 samples per pixel under a concave tone curve: only a loose bound on the mean level)
    assert abs(l3["paths_per_s"] * l3["ms_per_step"] / (l1["paths_per_s"] * l1["ms_per_step"]) - 2.0) < 1e-6
    c = np.load(two_s)
    assert c.shape == a.shape and (c[..., 3] == 255).all()
    assert np.abs(c[..., :3].astype(int).mean() - a[..., :3].astype(int).mean()) < 40.0
