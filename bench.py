#!/usr/bin/env python3
"""Benchmark of the MI355X render path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A step is one frame: every rank renders its round-robin row bands of the frame through the
C ABI (scene already resident in HBM), the 8-bit framebuffer is gathered to rank 0 over RCCL
(N > 1) and re-ordered into the frame.  Rays = closest-hit rays + shadow rays, counted on
the device.  Rank 0 prints one JSON line.

Workloads (the configurations the headline metric and the parity cases are quoted on):
    teapot3_8k16  teapot-3.nff geometry at 7680x4320, -s 4 (16 spp)   [default; the config the
                  1/2/4/8-GPU metric is quoted on; strong scaling over row bands]
    teapot3       teapot-3.nff at its own 512x512, 1 spp
    balls         balls.nff 512x512, 1 spp
    refract       refract.nff 512x512, 1 spp
    glass         glass.nff 900x900, 1 spp (light-camera caustics on; the reference's hide.ppm frame)
    glass_setup   the same, with the scene uploaded again in every step: the frame then includes what the
                  reference also does once per run -- hierarchies, the four light cameras' photon passes
                  (231,361 light rays each) and their point octrees
    gears         gears.nff 512x512, 1 spp (36 k triangles, transparent materials, 5 lights)
    balls2        balls2.nff 512x512, 1 spp (every ray reflects to the depth limit)

For N > 1 launch as: python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    #  name: (scene, resx, resy, N, default steps, default warmup)
    "teapot3_8k16": ("teapot-3", 7680, 4320, 4, 10, 2),
    "teapot3": ("teapot-3", 0, 0, 1, 50, 5),
    "balls": ("balls", 0, 0, 1, 50, 5),
    "refract": ("refract", 0, 0, 1, 50, 5),
    "glass": ("glass", 0, 0, 1, 20, 3),
    "glass_setup": ("glass", 0, 0, 1, 5, 1),
    "gears": ("gears", 0, 0, 1, 20, 3),
    "balls2": ("balls2", 0, 0, 1, 20, 3),
    "tetra3": ("tetra-3", 0, 0, 1, 50, 5),
}
HBM_PEAK_GBPS = 8000.0
# Vector issue.  Cycles one 64-wide instruction occupies a SIMD, measured on MI355X with every SIMD running four waves
# of independent instructions (tools/micro/valu_rates.hip, profiles/r02a/valu_rates.jsonl): FP64 add, mul, fma and
# compare 4.3; FP64 rcp, sqrt (the TRANS counter) 16.2; FP32 fma/mul, integer add, moves 2.4.  (max3/min3 and selects
# with an explicit mask also take 4.3; they are priced at 2.4 here with everything that is not counted separately,
# which makes the peak below an upper bound and the fraction a lower one.)
SIMDS, CLOCK_GHZ = 256 * 4, 2.4
CYC_FP64, CYC_TRANS64, CYC_OTHER = 4.3, 16.2, 2.4
BAND_ROWS = 16
# Bytes this build's own traversal touches per step: one 4-wide node (six 16-byte plane rows and the child codes, 112 of its
# 128 bytes), the 80 bytes of a leaf triangle record a test reads, a 32-byte sphere record plus key and id.
BVH_NODE_B, LEAF_TRI_B, LEAF_SPH_B = 128, 80, 40

# Algorithmic bytes per ray (fixed by the reference's traversal, not by what this build fetches):
#   node tests x 64 B node record + triangle tests x 96 B + sphere tests x 32 B + 192 B of queue
#   traffic (a 96-byte ray or hit record written once and read once).
NODE_B, TRI_B, SPH_B, QUEUE_B = 64, 96, 32, 192


# Per-ray test counts of the reference's traversal on the default workload, as the CPU sample of a
# one-GPU run measures them (32 rows spread over the frame); used for the roofline object of runs that
# do not take the CPU sample themselves (N > 1, --no-cpu-baseline).
RECORDED_BYTES_PER_RAY = {
    "teapot3_8k16": {"closest": 41.75074814500611 * NODE_B + 217.95551503359582 * TRI_B + QUEUE_B,
                     "shadow": 36.538628432664986 * NODE_B + 199.73884823610032 * TRI_B + QUEUE_B},
}

CPU_SAMPLE_TARGET = 4_000_000      # primary samples in the CPU sample (about 12 M rays, 10-30 s of CPU work)


def cpu_reference_sample(scene_name, resx, resy, N, threads, lightcam=True):
    """Times the CPU oracle on a bounded sample of the workload: the whole frame when it is small, else
    single rows spread evenly over the frame.  Returns rates and the per-ray test counts of the sample."""
    import raytracer_amd as rt
    from oracle import binding as ob
    o = ob.OracleScene(rt.scene_path(scene_name))
    if resx > 0:
        o.set_resolution(resx, resy)
    per_row = o.resx * N * N
    if per_row * o.resy <= CPU_SAMPLE_TARGET:
        rows = list(range(o.resy))
    else:
        n = max(2 * threads, min(CPU_SAMPLE_TARGET // per_row, o.resy))
        rows = sorted({int((k + 0.5) * o.resy / n) for k in range(n)})
    out = {}
    for label, th in (("all", threads), ("two", 2)):
        if label == "two":          # a fresh scene, so that this leg also builds its light cameras inside the timed call
            o.close()
            o = ob.OracleScene(rt.scene_path(scene_name))
            if resx > 0:
                o.set_resolution(resx, resy)
        t0 = time.perf_counter()
        fr = o.render(num_samples=N, threads=th, rows=rows, flags=0 if lightcam else ob.FLAG_NO_LIGHTCAM,
                      want_hits=False, want_mask=False, want_u8=False)
        dt = time.perf_counter() - t0
        tot = fr["counters"]
        out[label] = {"seconds": dt, "threads": th, "mrays_s": (tot["closest_rays"] + tot["shadow_rays"]) / dt / 1e6}
    c = tot
    tri_share = o.n_triangles / max(o.n_solids, 1)
    out["per_ray"] = {
        "closest_nodes": c["node_tests_closest"] / max(c["closest_rays"], 1),
        "closest_prims": c["prim_tests_closest"] / max(c["closest_rays"], 1),
        "shadow_nodes": c["node_tests_shadow"] / max(c["shadow_rays"], 1),
        "shadow_prims": c["prim_tests_shadow"] / max(c["shadow_rays"], 1),
    }
    # scenes here are (almost) purely triangles or purely spheres; price a primitive test by the majority type
    prim_b = TRI_B if tri_share >= 0.5 else SPH_B
    out["bytes_per_ray"] = {
        "closest": out["per_ray"]["closest_nodes"] * NODE_B + out["per_ray"]["closest_prims"] * prim_b + QUEUE_B,
        "shadow": out["per_ray"]["shadow_nodes"] * NODE_B + out["per_ray"]["shadow_prims"] * prim_b + QUEUE_B,
    }
    out["sample"] = "%s.nff %dx%d %d spp, %s (%d rays)" % (
        scene_name, o.resx, o.resy, N * N,
        "full frame" if len(rows) == o.resy else "%d rows spread evenly over the frame" % len(rows),
        c["closest_rays"] + c["shadow_rays"])
    o.close()
    return out


def _newest_profile(workload, stem):
    """Newest committed profiles/<tag>/<stem>_<workload>.json, with the build it was measured on."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "%s_%s.json" % (stem, workload))))
    if not files:
        return None, None, None
    with open(files[-1]) as f:
        js = json.load(f)
    meta = js.get("_meta", {})
    return js, meta.get("build_id"), os.path.relpath(files[-1], ROOT)


def measured_counters(workload, kernel, build_id):
    """Counter totals of `kernel` from the newest committed counter passes of this workload (separate rocprofv3 --pmc
    runs, tools/pmc_passes.sh; counters cannot be collected inside a timed run).  Counters of another build of the
    library are refused: the dict then only says so."""
    js, their_build, path = _newest_profile(workload, "pmc")
    if js is None:
        return None
    if their_build != build_id:
        return {"stale": "counters in %s were measured on build %s, this library is build %s: not used" % (path, their_build, build_id)}
    tot = {}
    n = 0
    rd = wr = 0.0
    prefixes = kernel if isinstance(kernel, (tuple, list)) else (kernel,)
    for name, e in js.items():
        if name == "_meta" or not name.startswith(tuple(prefixes)) or "hbm_read_bytes_per_launch" not in e:
            continue
        n += e["dispatches"]
        rd += e["hbm_read_bytes_per_launch"] * e["dispatches"]
        wr += e["hbm_write_bytes_per_launch"] * e["dispatches"]
        for k, v in e["counters"].items():
            tot[k] = tot.get(k, 0.0) + v
    if n == 0:
        return None
    g = lambda k: tot.get(k, 0.0) / n
    fp64 = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64")
    trans = g("SQ_INSTS_VALU_TRANS_F64")
    valu = g("SQ_INSTS_VALU")
    return {"source": path, "build_id": their_build, "launches": int(n),
            "hbm_read_bytes_per_launch": rd / n, "hbm_write_bytes_per_launch": wr / n,
            "valu_per_launch": valu, "valu_fp64_per_launch": fp64, "valu_trans64_per_launch": trans,
            "valu_int32_per_launch": g("SQ_INSTS_VALU_INT32"), "salu_per_launch": g("SQ_INSTS_SALU"),
            "issue_cycles_per_launch": fp64 * CYC_FP64 + trans * CYC_TRANS64 + max(valu - fp64 - trans, 0.0) * CYC_OTHER,
            "l2_hit_rate": tot.get("TCC_HIT_sum", 0.0) / tot["TCC_REQ_sum"] if tot.get("TCC_REQ_sum") else None,
            "lane_utilisation": tot.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * tot["SQ_ACTIVE_INST_VALU"])
            if tot.get("SQ_ACTIVE_INST_VALU") else None,
            "waiting_share_of_wave_cycles": tot.get("SQ_WAIT_ANY", 0.0) / tot["SQ_WAVE_CYCLES"] if tot.get("SQ_WAVE_CYCLES") else None}


def measured_walk(workload, build_id):
    """Per-ray traversal work of this build (node steps, leaf tests) from the newest committed run of the counting
    build (tools/walk_stats.py --json), refused like the counters when it belongs to another build."""
    js, their_build, path = _newest_profile(workload, "walk")
    if js is None or their_build != build_id:
        return None
    js["source"] = path
    return js


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="teapot3_8k16", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="extra RT_FLAG_* bits for the render calls")
    ap.add_argument("--budget-gib", type=float, default=0.0, help="queue memory budget of the context (0: the library's default)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N>1; gloo (frame staged through host memory) is only for rehearsing "
                         "several ranks on one GPU, where RCCL refuses duplicate devices")
    args = ap.parse_args()

    import torch
    import raytracer_amd as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus %d needs the torch.distributed.run launcher (see the module docstring)" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the render path has no CPU fallback")
    # one process per GPU; if the launcher already narrowed the visible devices to one, use that one
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    scene_name, resx, resy, N, dsteps, dwarm = WORKLOADS[args.workload]
    steps = args.steps if args.steps is not None else dsteps
    warmup = args.warmup if args.warmup is not None else dwarm

    scene = rt.Scene.load(scene_name)
    cam = scene.camera(resx, resy)
    resx, resy = cam.resx, cam.resy
    gpu = rt.Renderer(dev_index)
    if args.budget_gib > 0:
        gpu.set_memory_budget(int(args.budget_gib * (1 << 30)))
    gpu.upload(scene)

    from raytracer_amd.multi import FrameGather
    band = BAND_ROWS if world > 1 else 0
    dev = torch.device("cuda", dev_index)
    staged = world > 1 and args.backend == "gloo"
    # Two frame buffers in turn: the gather of one frame (RCCL, on torch's stream) may still be reading its
    # buffer while the next frame is rendered (on the context's stream) into the other one.
    n_buf = 2 if world > 1 else 1
    fgs = [FrameGather(resx, resy, band, world, rank, torch.device("cpu") if staged else dev, dist) for _ in range(n_buf)]
    assert fgs[0].local_rows == rt.part_rows(cam, band, world, rank)
    render_bufs = [torch.zeros_like(fg.local, device=dev) if staged else fg.local for fg in fgs]
    gathered = [None] * n_buf          # event after the last gather that read buffer k
    frame_no = [0]

    def step(flags):
        k = frame_no[0] % n_buf
        frame_no[0] += 1
        if gathered[k] is not None:
            gathered[k].synchronize()
        if args.workload.endswith("_setup"):
            gpu.upload(scene)
        st = gpu.render_device(cam, rgb_u8_ptr=render_bufs[k].data_ptr(), num_samples=N, flags=flags, band_rows=band,
                               n_parts=world, part=rank)
        if world > 1:
            if staged:
                fgs[k].local.copy_(render_bufs[k])
            fgs[k].gather()
            if not staged:
                gathered[k] = torch.cuda.Event()
                gathered[k].record()
        return st

    def last_frame():
        k = (frame_no[0] - 1) % n_buf
        return fgs[k].frame if world > 1 else render_bufs[k]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step(args.flags)
    fence()
    t0 = time.perf_counter()
    rays = 0
    for _ in range(steps):
        st = step(args.flags)
        rays += st["closest_rays"] + st["shadow_rays"]
    fence()
    elapsed = time.perf_counter() - t0

    # per-kernel times: a few extra frames with HIP events around every launch on the render stream
    prof_steps = max(1, min(steps, 5))
    kt = {"closest_ms": 0.0, "shadow_ms": 0.0, "primary_ms": 0.0, "closest_launches": 0, "shadow_launches": 0,
          "primary_launches": 0, "closest_rays": 0, "shadow_rays": 0, "gpu_ms": 0.0, "primary_rays": 0, "primary_hits": 0}
    for _ in range(prof_steps):
        st = step(args.flags | rt.FLAG_TIMING)
        for k in kt:
            kt[k] += st["rays_per_generation"][0] if k == "primary_rays" else (st["hits_per_generation"][0] if k == "primary_hits" else st[k])
    fence()
    # A primary pass that deals with the lights at its hits itself (no shadow pass for generation 0: one shadow launch fewer
    # than closest-hit launches per batch) traces those shadow rays too: they count with the pass that traces them.
    n_lights = len(scene.lights())
    fused_lights = 0 < kt["shadow_launches"] < kt["closest_launches"]
    own_shadow_rays = kt["primary_hits"] * n_lights if fused_lights else 0

    if world > 1:
        t = torch.tensor([elapsed, float(rays)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays = float(t[1])

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            cpu = cpu_reference_sample(scene_name, resx, resy, N, threads,
                                       lightcam=not (args.flags & rt.FLAG_NO_LIGHTCAM))
        # ---- roofline of the dominant kernel: the one with the largest total time among the launches of a frame
        build_id = rt.build_id()
        # (kernel names as the profiler prints them, by prefix: the passes that read the view grids are kernels of their own)
        prefixes = {"k_local / k_fused (the frame is one launch)": ("k_local", "k_fused"),
                    "primary pass: k_closest<true, true, *>": ("k_closest<true, true",),
                    "queued closest-hit passes: k_closest<false, true, *>": ("k_closest<false, true",),
                    "shadow passes: k_shadow<true> / k_shadow_grid": ("k_shadow<true>", "k_shadow_grid")}
        if kt["shadow_launches"] == 0 and kt["closest_launches"] == kt["primary_launches"]:
            classes = {"k_local / k_fused (the frame is one launch)": ("fused", kt["primary_ms"], kt["primary_launches"], kt["closest_rays"] + kt["shadow_rays"])}
        else:
            classes = {
                "primary pass: k_closest<true, true, *>": ("primary", kt["primary_ms"], kt["primary_launches"], kt["primary_rays"] + own_shadow_rays),
                "queued closest-hit passes: k_closest<false, true, *>": ("queued", kt["closest_ms"] - kt["primary_ms"],
                                                                         kt["closest_launches"] - kt["primary_launches"],
                                                                         kt["closest_rays"] - kt["primary_rays"]),
                "shadow passes: k_shadow<true> / k_shadow_grid": ("shadow", kt["shadow_ms"], kt["shadow_launches"], kt["shadow_rays"] - own_shadow_rays),
            }
        kname = max(classes, key=lambda k: classes[k][1])
        dom, ms_total, launches, rays_k = classes[kname]
        launches = max(launches, 1)
        avg_ms = ms_total / launches
        # (for N > 1 the launches, their rays and their times are rank 0's own; the counter passes are one-GPU runs)
        ctr = measured_counters(args.workload, prefixes[kname], build_id) if world == 1 else None
        stale = ctr.get("stale") if ctr else None
        if stale:
            ctr = None
        walk = measured_walk(args.workload, build_id) if world == 1 else None
        roofline = {"bound": "valu_issue", "kernel": kname, "avg_launch_ms": avg_ms, "launches": launches,
                    "rays_per_launch": rays_k / launches, "build_id": build_id,
                    "kernel_ms_per_frame": {k: v[1] / prof_steps for k, v in classes.items()},
                    "frame_gpu_ms": kt["gpu_ms"] / prof_steps,
                    "primary_pass_shades_its_hits": bool(fused_lights),
                    "achieved": None, "peak": None, "unit": "G wave-instructions/s", "frac": None, "traffic": None}
        if stale:
            roofline["counters"] = stale
        if ctr:
            # Vector issue: wave-level vector instructions per launch (counter pass) over the live launch time, against what
            # the SIMDs can issue of THIS kernel's instruction mix (cycles per instruction by kind, see CYC_* above).
            rate = ctr["valu_per_launch"] / (avg_ms * 1e-3) / 1e9
            peak = SIMDS * CLOCK_GHZ / (ctr["issue_cycles_per_launch"] / ctr["valu_per_launch"])
            hbm_bytes = ctr["hbm_read_bytes_per_launch"] + ctr["hbm_write_bytes_per_launch"]
            hbm_rate = hbm_bytes / (avg_ms * 1e-3) / 1e9
            roofline.update({
                "achieved": rate, "peak": peak, "frac": rate / peak, "traffic": hbm_bytes,
                "issue_cycles_per_instruction": ctr["issue_cycles_per_launch"] / ctr["valu_per_launch"],
                "valu_wave_instructions_per_64_rays": ctr["valu_per_launch"] / max(rays_k / launches / 64.0, 1),
                "hbm": {"achieved": hbm_rate, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm_rate / HBM_PEAK_GBPS,
                        "read_bytes_per_launch": ctr["hbm_read_bytes_per_launch"],
                        "write_bytes_per_launch": ctr["hbm_write_bytes_per_launch"], "l2_hit_rate": ctr["l2_hit_rate"]},
                "counters": {k: ctr[k] for k in ("source", "build_id", "launches", "valu_per_launch", "valu_fp64_per_launch",
                                                 "valu_trans64_per_launch", "valu_int32_per_launch", "salu_per_launch",
                                                 "lane_utilisation", "waiting_share_of_wave_cycles")}})
        if walk and dom in walk:
            # what this build's own traversal needs per ray: node steps and leaf tests counted by the -DRT_COUNT build,
            # plus the queue records the kernel reads and writes
            w = walk[dom]
            queue_b = {"primary": 80 if fused_lights else 64 + 80, "queued": 80 + 64 + 80, "shadow": 64 / max(walk.get("lights", 1), 1) + 24,
                       "fused": 24}[dom]
            alg = w["nodes_per_ray"] * BVH_NODE_B + w["tri_tests_per_ray"] * LEAF_TRI_B + w["sph_tests_per_ray"] * LEAF_SPH_B + queue_b
            roofline["algorithmic"] = {"bytes_per_ray": alg, "bytes_per_launch": alg * rays_k / launches,
                                       "node_steps_per_ray": w["nodes_per_ray"], "leaf_tests_per_ray": w["tri_tests_per_ray"] + w["sph_tests_per_ray"],
                                       "queue_bytes_per_ray": queue_b, "source": walk["source"],
                                       "note": "most of it is served from LDS and L2 (the scene is smaller than either); only the "
                                               "queue records have to come from HBM"}
        bpr = cpu["bytes_per_ray"] if cpu is not None else RECORDED_BYTES_PER_RAY.get(args.workload)
        if bpr is not None and dom in ("primary", "queued", "shadow"):
            ref_dom = "shadow" if dom == "shadow" else "closest"
            roofline["reference_algorithm_bytes"] = {
                "bytes_per_ray": bpr[ref_dom], "bytes_per_launch": bpr[ref_dom] * rays_k / launches,
                "note": "node and primitive tests of the REFERENCE's octree walk per ray x record sizes + 192 B of queue "
                        "traffic; this build answers the same queries with a fraction of those tests, so this is not a "
                        "measure of what the kernel moves and no fraction is formed from it"}
        # the timed frame itself: checksum of its 8-bit image, and whether it is the frame the GPU tests pin
        frame_sha = hashlib.sha256(last_frame().cpu().numpy().tobytes()).hexdigest()
        pinned = None
        try:
            with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
                pinned = json.load(f).get("frame_sha256", {}).get(args.workload)
        except OSError:
            pass
        line = {
            "metric": "Mrays/sec (primary+secondary)",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": args.workload, "scene": scene_name + ".nff", "resolution": [resx, resy],
                       "spp": N * N, "rays_per_frame": rays / steps,
                       "partition": ("the whole frame on one GPU (no gather)" if world == 1 else
                                     "%d-row bands round-robin over %d GPUs, %s gather of RGB8"
                                     % (band, world, "RCCL" if args.backend == "nccl" else "gloo (rehearsal)"))},
        }
        line["frame_sha256"] = frame_sha
        line["frame_matches_pinned"] = (frame_sha == pinned) if pinned else None
        line["output"] = "RGB8 frame in device memory (the double frame of the boundary is not written in the timed steps)"
        line["roofline"] = roofline
        if cpu is not None:
            line["cpu_baseline"] = {"value": cpu["all"]["mrays_s"], "unit": "Mrays/s", "cores": cpu["all"]["threads"],
                                    "kind": "port", "sample": cpu["sample"],
                                    "two_thread_value": cpu["two"]["mrays_s"],
                                    "seconds": cpu["all"]["seconds"] + cpu["two"]["seconds"]}
        print(json.dumps(line), flush=True)

    gpu.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
