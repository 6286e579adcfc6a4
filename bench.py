#!/usr/bin/env python3
"""Benchmark of the hot path: hippo scene, 1920x1080, closest-hit queries per second.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of `--spp-per-step` samples per pixel over the frame (default 16,
so the default 16 steps are the headline workload: hippo, 1920x1080, 256 spp). With N > 1
(launched by torch.distributed.run, one rank per GPU) the scene is replicated and every
rank renders the whole frame for its own sample range (weak scaling, default), or the
frame is split into 32x32 tiles dealt round-robin over the ranks (--shard tiles, strong
scaling); either way the only exchange is one RCCL reduce to rank 0 at the end of the
timed region.

Prints ONE JSON line on rank 0. `value` = BVH closest-hit queries (bvh.intersect calls
of the path tracer while it renders the 1080p frame) per second over all ranks, inputs
resident in HBM. The line also carries samples/s, `closest_hit_mrays_per_s` (the
16,777,216-ray intersection microbenchmark, N = 1 only), the roofline of the dominant
kernel and the CPU baseline (oracle restatement, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E datasheet peak
SIMDS = 256 * 4                # 256 CUs x 4 SIMDs
SIMD_LANES = 16                # a wave64 vector instruction occupies a SIMD's 16 lanes for 4 cycles
PEAK_CLOCK_HZ = 2.4e9
FP64_VECTOR_PEAK_TFLOPS = 78.6
VISIT_COUNTS_FILE = ROOT / "profiles" / "reference_visit_counts.json"
# per-launch hardware counters of the traversal kernels on exactly these workloads (rocprofv3 --pmc passes,
# tools/prof_passes.sh + tools/make_counters_json.py; the summaries they come from sit next to the file).
# The newest round's file that exists is used; it records the commit it was taken at.
COUNTERS_FILES = [ROOT / "profiles" / r / "counters.json" for r in ("r03", "r02")]
COUNTERS_FILE = next((f for f in COUNTERS_FILES if f.exists()), COUNTERS_FILES[0])
# the stored counters describe one workload: they are combined with a live launch time only when the live
# launches carry the same number of rays to within this relative tolerance
COUNTERS_RAYS_TOLERANCE = 0.03


def reference_algorithm_flops_per_ray(c):
    """40 flops per node the REFERENCE order visits (shell test + prune) + 70 per polygon test + 45 per
    sphere test (operation counts of bvh.rs:106-110,152-164 and primitives.rs:45-81,206-248)."""
    return 40.0 * c["nodes_per_ray"] + 70.0 * c["polys_per_ray"] + 45.0 * c["spheres_per_ray"]


def algorithmic_bytes_per_ray(c):
    """64 B per node visit + 176 B per polygon test + 32 B per sphere test + 48 B ray in +
    16 B hit out, with the visit counts of the REFERENCE traversal order (bvh.rs:97-149)."""
    return 64.0 * c["nodes_per_ray"] + 176.0 * c["polys_per_ray"] + 32.0 * c["spheres_per_ray"] + 48.0 + 16.0


def own_walk_bytes_per_ray(c):
    """What the fast walk itself reads: 64 B per node step (one quantised 4-wide node), 128 B per leaf visit
    (the screen records of two primitives side by side), 96 + 128 B per exact test, 48 B ray in, 16 B hit out
    (counts measured on the GPU by the counting build, `make count`)."""
    leaf = 128.0 * c["leaf_visits_per_ray"] if "leaf_visits_per_ray" in c else 64.0 * c["screened_per_ray"]
    return 64.0 * c["node_steps_per_ray"] + leaf + 224.0 * c["exact_tests_per_ray"] + 64.0


def issue_roofline(kernel, counters, avg_launch_ms, launches, rays_per_launch):
    """Roofline of a cache-resident walk. Nothing on this path is a dense contraction and the 23 MB scene is
    served by the caches, so the bound is the vector ALU: frac = useful lane-slots / all lane-slots =
    SQ_THREAD_CYCLES_VALU (lanes that carry a ray, summed over the cycles of every issued vector instruction)
    per launch / (live launch duration x 1024 SIMDs x 16 lanes x 2.4 GHz). Beside it valu_busy_frac, the
    share of SIMD cycles in which some vector instruction issues whatever its lanes do (SQ_ACTIVE_INST_VALU
    counts 4-cycle issue slots), and the memory-side traffic. Returns None when the stored counters were
    taken on another workload than the live launches (rays per launch differ)."""
    c = counters
    stored = c.get("rays_per_launch")
    if stored and abs(rays_per_launch - stored) > COUNTERS_RAYS_TOLERANCE * stored:
        return {"kernel": kernel, "skipped": "counters do not match this workload",
                "counters_workload_rays_per_launch": stored, "rays_per_launch": rays_per_launch,
                "avg_launch_ms": avg_launch_ms, "launches": int(launches)}
    secs = avg_launch_ms * 1e-3
    achieved = c["SQ_THREAD_CYCLES_VALU"] / secs / 1e12
    peak = SIMDS * SIMD_LANES * PEAK_CLOCK_HZ / 1e12
    busy = 4.0 * c["SQ_ACTIVE_INST_VALU"] / secs / (SIMDS * PEAK_CLOCK_HZ)
    traffic = 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0 if "FETCH_SIZE" in c else None
    r = {"kernel": kernel, "bound": "valu", "achieved": achieved, "peak": peak,
         "unit": "T useful lane-cycles/s", "frac": achieved / peak, "traffic": traffic,
         "valu_busy_frac": busy,
         "launches": int(launches), "avg_launch_ms": avg_launch_ms, "rays_per_launch": rays_per_launch,
         "counters_per_launch": {k: c[k] for k in sorted(c) if k.isupper() or k.startswith("SQ_") or k.startswith("TCC_")},
         "counters_workload_rays_per_launch": stored,
         "counters_file": str(COUNTERS_FILE.relative_to(ROOT)),
         "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
         "wave_cycles_waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
         "valu_busy_by_wave_cycles": c["SQ_ACTIVE_INST_VALU"] * c["SQ_WAVES"] / (SIMDS * c["SQ_WAVE_CYCLES"]),
         "note": "frac = SQ_THREAD_CYCLES_VALU / (t x 1024 SIMDs x 16 lanes x 2.4 GHz): the share of the vector "
                 "ALUs' lane-slots that do a ray's work (counters per launch from counters_file, t measured live "
                 "with HIP events on the launch stream); valu_busy_frac = 4 x SQ_ACTIVE_INST_VALU / (t x 1024 x "
                 "2.4 GHz): SIMD cycles in which a vector instruction issues; lane_utilisation = their ratio; "
                 "wave_cycles_waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES; traffic = bytes per launch at the L2's "
                 "memory side (2 x FETCH_SIZE + WRITE_SIZE), hbm = that over t against 8 TB/s"}
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        r["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if traffic is not None:
        gbs = traffic / secs / 1e9
        r["hbm"] = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    return r


def reference_algorithm_figures(counts, rays, secs):
    """The reference-algorithm bytes and flops of the same rays per second of kernel time. The fast walk
    returns the reference's hits with a fraction of the reference's node visits, so neither figure is work
    this kernel executes: they are labelled equivalents, not fractions of a hardware limit."""
    bpr, fpr = algorithmic_bytes_per_ray(counts), reference_algorithm_flops_per_ray(counts)
    tf = fpr * rays / secs / 1e12
    return {"bytes_per_ray": bpr, "gbs": bpr * rays / secs / 1e9,
            "flops_per_ray": fpr, "tflops_equivalent": tf,
            "tflops_equivalent_over_fp64_vector_peak": tf / FP64_VECTOR_PEAK_TFLOPS,
            "visits_per_ray": {k: counts[k] for k in ("nodes_per_ray", "polys_per_ray", "spheres_per_ray")},
            "workload": counts.get("workload"),
            "note": "reference-algorithm bytes (64 B per node the REFERENCE order visits + 176 B per polygon test + "
                    "32 B per sphere test + 48 + 16) and flops (40 per node + 70 per polygon + 45 per sphere test) "
                    "per second of kernel time: equivalent work, NOT executed and not a fraction of anything; the "
                    "fast walk visits a sixteenth of those nodes"}


def cpu_baseline(args, cores):
    """Times the oracle (C restatement of the reference's CPU path, private scene per
    thread, boxed tree, per-query stack allocation) on a bounded sample of the same
    workload, and counts the reference traversal's node/primitive visits on it."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as O
    O.build()
    L = O.lib_timed()  # the -O3 build; the timed calls pass no counter block, so nothing but the path is timed
    v, f = O.load_rtmesh(ROOT / "ray-tracer_amd" / "assets" / "hippo.rtmesh")
    vp, fp = O.dp(v), O.u32p(f)
    cam = O.reference_camera()
    stride = args.cpu_pixel_stride
    rgba = np.zeros(args.width * args.height * 4, dtype=np.uint8)
    n_rays = C.c_uint64(0)
    secs = L.o_time_render(vp, v.shape[0], fp, f.shape[0], C.byref(cam), args.width, args.height,
                           args.spp_per_step, C.c_uint64(args.seed), cores, stride,
                           rgba.ctypes.data_as(O.c_u8_p), C.byref(n_rays), None)
    n_pix = sum(len(range(t, args.width * args.height, stride)) for t in range(cores))
    # visit counts of the reference order: a separate, untimed pass over a sixteenth of that sample
    st, n_counted = O.Stats(), C.c_uint64(0)
    L.o_time_render(vp, v.shape[0], fp, f.shape[0], C.byref(cam), args.width, args.height,
                    args.spp_per_step, C.c_uint64(args.seed), cores, stride * 16,
                    rgba.ctypes.data_as(O.c_u8_p), C.byref(n_counted), C.byref(st))
    counts = dict(nodes_per_ray=st.nodes_popped / n_counted.value, polys_per_ray=st.poly_tests / n_counted.value,
                  spheres_per_ray=st.sphere_tests / n_counted.value, rays=int(n_counted.value),
                  workload=f"hippo {args.width}x{args.height} {args.spp_per_step} spp, pixel index % {stride * 16} "
                           f"< {cores}")
    out = {
        "value": n_rays.value / secs / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"{n_pix} pixels (pixel index % {stride} < {cores}) of the {args.width}x{args.height} hippo frame "
                  f"at {args.spp_per_step} spp = {n_pix * args.spp_per_step * 5} paths, {n_rays.value} rays, "
                  f"{secs:.1f} s; C restatement of the reference CPU path built -O3 (no contraction, no fast-math), "
                  f"one private scene per thread, no counters in the timed loop",
        "samples_per_s": n_pix * args.spp_per_step / secs,
    }
    # the reference's own thread count (NUM_THREADS = 7, main.rs:17) on a smaller sample
    n7 = C.c_uint64(0)
    secs7 = L.o_time_render(vp, v.shape[0], fp, f.shape[0], C.byref(cam), args.width, args.height,
                            args.spp_per_step, C.c_uint64(args.seed), 7, stride * 4,
                            rgba.ctypes.data_as(O.c_u8_p), C.byref(n7), None)
    out["seven_threads"] = {"value": n7.value / secs7 / 1e6, "unit": "Mrays/s", "cores": 7,
                            "sample": f"pixel index % {stride * 4} < 7, {n7.value} rays, {secs7:.1f} s"}
    return out, counts, O


def cpu_microbench(O, rays, cores):
    v, f = O.load_rtmesh(ROOT / "ray-tracer_amd" / "assets" / "hippo.rtmesh")
    r = np.ascontiguousarray(rays)
    ids = np.empty(len(r), dtype=np.uint32)
    dist = np.empty(len(r), dtype=np.float64)
    secs = O.lib_timed().o_time_intersect_batch(O.dp(v), v.shape[0], O.u32p(f), f.shape[0], O.dp(r), len(r), cores,
                                          O.u32p(ids), O.dp(dist))
    return len(r) / secs / 1e6, ids, dist


def deep_leg(rta, torch, dev, stream, args, counters_all):
    """Config 5: the hippo mesh subdivided four times (11.8 M triangles) + 100,000 spheres, 6.3 GB on the device: the
    one workload whose walk is bound by memory. 16,777,216-ray batch, both traversal flavours."""
    t0 = time.perf_counter()
    bvh = rta.scenes.build(rta.scenes.subdivided_scene(args.deep_levels, 100_000), device=dev.index or 0)
    bvh.scene()
    build_s = time.perf_counter() - t0
    info = bvh.scene_info()
    n = args.microbench_rays
    rays = torch.from_numpy(rta.make_ray_batch(bvh, n)).to(dev)
    ids = torch.empty(n, dtype=torch.int32, device=dev)
    dist = torch.empty(n, dtype=torch.float64, device=dev)
    lib = rta._lib.lib()
    out = {"scene": f"hippo subdivided {args.deep_levels}x + 100,000 spheres: {info['n_prims']:,} primitives, "
                    f"{info['device_bytes'] / 1e9:.2f} GB on the device, fast tree {info['fast_nodes']:,} 64-byte nodes, "
                    f"{info['fast_depth']} levels", "rays": n, "scene_build_s": build_s}
    keep = {}
    for name, m, reps in (("fast", rta.TRAVERSE_FAST, 3), ("reference_order", rta.TRAVERSE_REFERENCE_ORDER, 1)):
        call = lambda: rta._lib.check(lib.rt_intersect_batch_device(bvh.scene(), rays.data_ptr(), n, m, ids.data_ptr(),
                                                                    dist.data_ptr(), None, None, stream))
        if name == "fast":
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name] = {"mrays_per_s": n / ms / 1e3, "ms_per_launch": ms}
        keep[name] = (ids.clone(), dist.clone())
    hit = keep["fast"][0] != -1
    out["hit_fraction"] = float(hit.double().mean().item())
    out["modes_agree"] = bool(torch.equal(keep["fast"][0], keep["reference_order"][0]) and
                              torch.equal(keep["fast"][1][hit], keep["reference_order"][1][hit]))
    c = counters_all.get("deep_batch_16m", {})
    ck, own = c.get("k_intersect_fast"), c.get("fast_walk_visits_per_ray")
    if ck is not None and ck.get("rays_per_launch") and abs(n - ck["rays_per_launch"]) > COUNTERS_RAYS_TOLERANCE * n:
        out["roofline"] = {"skipped": "counters do not match this workload",
                           "counters_workload_rays_per_launch": ck["rays_per_launch"], "rays_per_launch": n}
        ck = None
    if ck is not None and own is not None:
        # the two kernels of a fast-mode launch: the walk (counters) and the small second pass
        walk_ms = ck.get("kernel_ms")
        secs = (walk_ms if walk_ms else out["fast"]["ms_per_launch"]) * 1e-3
        bpr = own_walk_bytes_per_ray(own)
        traffic = 2.0 * ck["FETCH_SIZE"] * 1024.0 + ck["WRITE_SIZE"] * 1024.0
        out["roofline"] = {
            "kernel": "k_intersect_fast", "bound": "hbm", "achieved": bpr * n / secs / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": bpr * n / secs / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_gbs": traffic / secs / 1e9, "traffic_frac": traffic / secs / 1e9 / HBM_PEAK_GBS,
            "bytes_per_ray": bpr, "visits_per_ray": own, "kernel_ms": secs * 1e3,
            "l2_hit_rate": ck["TCC_HIT_sum"] / (ck["TCC_HIT_sum"] + ck["TCC_MISS_sum"]) if "TCC_HIT_sum" in ck else None,
            "bytes_per_ray_split": {"node_steps": 64.0 * own["node_steps_per_ray"],
                                    "leaf_visits": 128.0 * own.get("leaf_visits_per_ray", 0.0),
                                    "exact_tests": 224.0 * own["exact_tests_per_ray"], "ray_in_hit_out": 64.0},
            "counters_file": str(COUNTERS_FILE.relative_to(ROOT)),
            "note": "achieved = bytes of records the walk requests (64 B per node step, 128 B per leaf visit, 224 B "
                    "per exact test, 64 B ray in/hit out; counts from the counting build) per second of the walk "
                    "kernel's time (rocprofv3 kernel trace, counters_file); traffic = memory-side bytes per launch "
                    "(2 x FETCH_SIZE + WRITE_SIZE) of the same kernel"}
        if "SQ_THREAD_CYCLES_VALU" in ck:
            out["roofline"]["valu_useful_lane_frac"] = ck["SQ_THREAD_CYCLES_VALU"] / (secs * SIMDS * SIMD_LANES * PEAK_CLOCK_HZ)
            out["roofline"]["lane_utilisation"] = ck["SQ_THREAD_CYCLES_VALU"] / (64.0 * ck["SQ_ACTIVE_INST_VALU"])
    if VISIT_COUNTS_FILE.exists():
        mini = json.loads(VISIT_COUNTS_FILE.read_text()).get("deep_miniature")
        if mini is not None:
            # the oracle cannot hold 11.9 M primitives: the reference order's visit counts are those of the
            # 99,004-primitive miniature of the same construction (a lower bound for the full scene's)
            out["reference_algorithm"] = reference_algorithm_figures(mini, n, out["fast"]["ms_per_launch"] * 1e-3)
            out["reference_algorithm"]["note"] += ("; visit counts of the REFERENCE order measured by the oracle on "
                                                   "the scene's 99,004-primitive miniature (the full scene's tree is "
                                                   "deeper: these are lower bounds)")
    del rays, ids, dist, keep
    bvh.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-per-step", type=int, default=16)
    ap.add_argument("--seed", type=int, default=20261111)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--shard", choices=("samples", "tiles"), default="samples",
                    help="N > 1: 'samples' = each rank renders the frame for its own sample range (weak scaling, "
                         "default); 'tiles' = the frame's tiles are split over the ranks (strong scaling)")
    ap.add_argument("--microbench-rays", type=int, default=1 << 24)
    ap.add_argument("--cpu-pixel-stride", type=int, default=61)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-microbench", action="store_true")
    ap.add_argument("--reference-order", action="store_true", help="use RT_TRAVERSE_REFERENCE_ORDER")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="nccl = RCCL (one rank per GPU). gloo is a rehearsal mode for boxes with fewer GPUs than "
                         "ranks: ranks share devices round-robin and the one collective goes through host memory")
    ap.add_argument("--scene", choices=("hippo", "sphere", "hippo_spheres", "deep"), default="hippo",
                    help="hippo = the reference scene (the judged workload); sphere = one sphere on the ground; "
                         "hippo_spheres = hippo + --scene-spheres spheres; deep = hippo subdivided --deep-levels "
                         "times + --scene-spheres spheres. The CPU leg and the roofline exist for hippo only.")
    ap.add_argument("--scene-spheres", type=int, default=None)
    ap.add_argument("--deep-levels", type=int, default=4)
    ap.add_argument("--no-deep", action="store_true", help="skip the config-5 leg (deep scene, 16.7 M-ray batch)")
    ap.add_argument("--save-frame", default=None, help="rank 0 writes the combined RGBA8 frame to this .npy file")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # Started as a plain `python bench.py --gpus N`: launch the N ranks as fresh child processes (one
        # per GPU, torch.distributed.run) BEFORE anything here touches the GPU, and hand their output on.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    import torch
    import torch.distributed as dist
    import ray_tracer_amd as rta

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the library has no CPU fallback")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    mode = rta.TRAVERSE_REFERENCE_ORDER if args.reference_order else rta.TRAVERSE_FAST
    if args.scene == "hippo":
        bvh = rta.init_environment(device=local_rank)
        scene_name = "hippo mesh + room (46,206 polygons)"
    else:
        if args.scene == "sphere":
            objs = rta.scenes.sphere_on_ground()
        elif args.scene == "hippo_spheres":
            objs = rta.scenes.hippo_with_spheres(12 if args.scene_spheres is None else args.scene_spheres)
        else:
            objs = rta.scenes.subdivided_scene(args.deep_levels,
                                               100_000 if args.scene_spheres is None else args.scene_spheres)
        bvh = rta.scenes.build(objs, device=local_rank)
        del objs
        info = bvh.info()
        scene_name = f"{args.scene} ({info['n_prims']:,} primitives, {info['n_nodes']:,} nodes, depth {info['depth']})"
        args.no_cpu = True
    cam = rta.reference_camera()
    by_tiles = args.shard == "tiles"
    if by_tiles:   # strong scaling: the frame's tiles are dealt over the ranks
        film = rta.Film(bvh, args.width, args.height, shard_index=rank, shard_count=world, tile_size=args.tile)
        sample0 = 0
    else:          # weak scaling: every rank renders the whole frame for its own sample range
        film = rta.Film(bvh, args.width, args.height)
        sample0 = rank * args.steps * args.spp_per_step
    film.set_camera(cam, args.seed)
    npix = args.width * args.height
    rgba = torch.zeros((args.height, args.width, 4), dtype=torch.uint8, device=dev)
    radiance = torch.zeros((args.height, args.width, 3), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    lib = rta._lib.lib()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def exchange(fn, t):
        if args.backend == "gloo" and world > 1:  # rehearsal: through host memory
            torch.cuda.synchronize()
            h = t.cpu()
            fn(h, dist)
            t.copy_(h)
        else:
            fn(t, dist)

    def combine():
        """The one exchange per frame (the analogue of main.rs:87-100)."""
        if by_tiles:   # disjoint tiles: summing the RGBA8 frames merges them
            film.resolve_device(None, rgba.data_ptr(), stream)
            exchange(rta.sharding.combine_tiles, rgba)
        else:          # sample ranges: average the mean-radiance frames, then tone-map once on rank 0
            film.resolve_device(radiance.data_ptr(), None, stream)
            exchange(rta.sharding.combine_samples, radiance)
            if rank == 0:
                rta._lib.check(lib.rt_tone_map_device(radiance.data_ptr(), npix, rgba.data_ptr(), stream))

    for w in range(args.warmup):
        film.render(sample0 + w * args.spp_per_step, args.spp_per_step, seed=args.seed, mode=mode, stream=stream)
    combine()  # warms the communicator up as well
    torch.cuda.synchronize()
    film.clear()

    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        film.render(sample0 + k * args.spp_per_step, args.spp_per_step, seed=args.seed, mode=mode, stream=stream)
    combine()
    barrier()
    elapsed = time.perf_counter() - t0

    st = film.stats()
    tot = torch.tensor([float(st["rays"]), float(st["paths"]), st["traverse_ms"], float(st["traverse_launches"])],
                       dtype=torch.float64, device=dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        if args.backend == "gloo":
            tot, tmax = tot.cpu(), tmax.cpu()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    rays_total, paths_total, trav_ms_sum, launches = [float(x) for x in tot.tolist()]

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    frame = rgba.cpu().numpy()
    assert (frame[..., 3] == 255).all(), "combined frame has holes"
    if args.save_frame:
        np.save(args.save_frame, frame)
    spp_rank = args.steps * args.spp_per_step
    spp_total = spp_rank if by_tiles else spp_rank * world
    if by_tiles:
        sharding = f"{args.tile}x{args.tile} tiles round-robin over {world} GPU(s), one RGBA8 reduce to rank 0"
    else:
        sharding = (f"sample ranges: {spp_rank} spp per GPU x {world} GPU(s) = {spp_total} spp, scene replicated, "
                    f"one f64 radiance reduce to rank 0")
    result = {
        "metric": "Mrays/s closest-hit (hippo BVH): value = BVH closest-hit queries/s of the 1080p path-traced "
                  "render (samples/s in samples_per_s); the 16,777,216-ray intersection microbenchmark is "
                  "closest_hit_mrays_per_s",
        "value": rays_total / elapsed / 1e6,
        "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong" if by_tiles else "weak",
        "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{scene_name}, {args.width}x{args.height}, "
                               f"{spp_rank} spp per GPU = {args.steps} steps x {args.spp_per_step} spp, 5 sub-rays per "
                               f"sample, max depth 5, camera of main.rs:22-30",
                   "traversal": "reference_order" if args.reference_order else "fast",
                   "sharding": sharding},
        "samples_per_s": args.width * args.height * spp_total / elapsed,
        "paths_per_s": paths_total / elapsed,
        "rays_per_path": rays_total / max(paths_total, 1.0),
        "frame_mean_rgb": [float(x) for x in frame[..., :3].reshape(-1, 3).mean(axis=0)],
    }

    cores = max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or (os.cpu_count() or 1))))
    counts, O = None, None
    if world == 1 and not args.no_cpu:
        result["cpu_baseline"], counts, O = cpu_baseline(args, cores)
    if counts is None and args.scene == "hippo" and VISIT_COUNTS_FILE.exists():
        counts = json.loads(VISIT_COUNTS_FILE.read_text())["render"]
    kname = "k_traverse_paths_ref" if args.reference_order else "k_traverse_paths_fast"
    counters_all = json.loads(COUNTERS_FILE.read_text()) if COUNTERS_FILE.exists() else {}
    ck = counters_all.get("hippo_render_1080p", {}).get(kname)
    if trav_ms_sum > 0 and ck is not None and args.scene == "hippo" and not args.reference_order:
        roof = issue_roofline(kname, ck, trav_ms_sum / max(launches, 1.0), launches, rays_total / max(launches, 1.0))
        own = counters_all.get("hippo_render_1080p", {}).get("fast_walk_visits_per_ray")
        if counters_all.get("commit"):
            roof["counters_commit"] = counters_all["commit"]
        if own is not None and "skipped" not in roof:
            obpr = own_walk_bytes_per_ray(own)
            roof["own_walk"] = {"bytes_per_ray": obpr, "visits_per_ray": own,
                                "requested_gbs": obpr * rays_total / (trav_ms_sum * 1e-3) / 1e9,
                                "note": "bytes of records the walk requests per second of kernel time; the 23 MB scene "
                                        "is served by L1/L2, so this is cache bandwidth, not HBM"}
        if counts is not None:
            roof["reference_algorithm"] = reference_algorithm_figures(counts, rays_total, trav_ms_sum * 1e-3)
        result["roofline"] = roof
    if result.get("cpu_baseline"):
        result["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]

    if world == 1 and not args.no_microbench:
        n = args.microbench_rays
        rays_h = rta.make_ray_batch(bvh, n)
        rays = torch.from_numpy(rays_h).to(dev)
        ids = torch.empty(n, dtype=torch.int32, device=dev)
        d = torch.empty(n, dtype=torch.float64, device=dev)
        lib = rta._lib.lib()
        micro = {"rays": n, "seed": "0x5EED0001", "generator": "ray-tracer_amd/batches.py"}
        for name, m, reps in (("fast", rta.TRAVERSE_FAST, 5), ("reference_order", rta.TRAVERSE_REFERENCE_ORDER, 2)):
            rta._lib.check(lib.rt_intersect_batch_device(bvh.scene(), rays.data_ptr(), n, m, ids.data_ptr(),
                                                         d.data_ptr(), None, None, stream))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                rta._lib.check(lib.rt_intersect_batch_device(bvh.scene(), rays.data_ptr(), n, m, ids.data_ptr(),
                                                             d.data_ptr(), None, None, stream))
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            micro[name] = {"mrays_per_s": n / ms / 1e3, "ms_per_launch": ms}
            if name == "fast":
                fast_ids = ids.cpu().numpy().view(np.uint32).copy()
                fast_d = d.cpu().numpy().copy()
        micro["hit_fraction"] = float((fast_ids != rta.MISS).mean())
        ref_ids = ids.cpu().numpy().view(np.uint32)
        micro["modes_agree"] = bool(np.array_equal(fast_ids, ref_ids))
        if not micro["modes_agree"]:
            # not expected: the fast walk repeats order-dependent rays in reference order
            differ = fast_ids != ref_ids
            ref_d = d.cpu().numpy()
            both = differ & (fast_ids != rta.MISS) & (ref_ids != rta.MISS)
            micro["modes_differ"] = {"rays": int(differ.sum()), "fraction": float(differ.mean()),
                                     "fast_is_closer_or_equal": int((fast_d[both] <= ref_d[both]).sum()),
                                     "both_hit": int(both.sum())}
        if O is not None:
            m_cpu = min(n, 1 << 20)
            stt = O.Stats()
            cpu_rate, cpu_ids, cpu_d = cpu_microbench(O, rays_h[:m_cpu], cores)
            hit = cpu_ids != rta.MISS
            micro["cpu_mrays_per_s"] = cpu_rate
            micro["cpu_sample"] = f"first {m_cpu} rays, {cores} threads"
            micro["ids_match_cpu"] = bool(np.array_equal(cpu_ids, fast_ids[:m_cpu]))
            micro["dist_bits_match_cpu"] = bool(np.array_equal(cpu_d[hit].view(np.uint64),
                                                               fast_d[:m_cpu][hit].view(np.uint64)))
            ref = O.hippo_scene()
            k = min(m_cpu, 1 << 17)
            ref.intersect_batch(rays_h[:k], stats=stt)
            mc = dict(nodes_per_ray=stt.nodes_popped / k, polys_per_ray=stt.poly_tests / k,
                      spheres_per_ray=stt.sphere_tests / k)
        elif args.scene == "hippo" and VISIT_COUNTS_FILE.exists():
            mc = json.loads(VISIT_COUNTS_FILE.read_text())["microbench"]
        else:
            mc = None
        if mc is not None:
            micro["reference_algorithm"] = reference_algorithm_figures(mc, n, micro["fast"]["ms_per_launch"] * 1e-3)
        hb = counters_all.get("hippo_batch_16m", {})
        bk = hb.get("kernel", "k_intersect_fast")  # k_intersect_fast2 (two rays per lane) for a small scene
        cb = hb.get(bk)
        if cb is not None:
            micro["roofline"] = issue_roofline(bk, cb, micro["fast"]["ms_per_launch"], 5, n)
        result["intersect_microbench"] = micro
        result["closest_hit_mrays_per_s"] = micro["fast"]["mrays_per_s"]
        if counts is not None and mc is not None and os.environ.get("RT_WRITE_VISIT_COUNTS"):
            VISIT_COUNTS_FILE.parent.mkdir(exist_ok=True)
            VISIT_COUNTS_FILE.write_text(json.dumps({"render": counts, "microbench": mc}, indent=1))
        del rays, ids, d

    if world == 1 and not args.no_deep and args.scene == "hippo":
        result["deep"] = deep_leg(rta, torch, dev, stream, args, counters_all)

    print(json.dumps(result))


if __name__ == "__main__":
    main()
