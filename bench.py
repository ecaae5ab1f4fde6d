#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path: formula-1 scene, 2000x1200, depth 8 (the 8192-spp frame in slices).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step is one pass of the path kernel over every pixel of the frame for `--spp-per-step` further
samples of the 8192-spp frame (the frame is 8192/spp-per-step such passes; nothing is cached
between steps: each step traces new sample indices).  With N > 1 (launched by torch.distributed.run,
one process per GPU) the scene is replicated, every rank traces its own sample range in each step
(weak scaling) and the accumulation buffers are summed onto rank 0 with one RCCL reduce per step,
inside the timed region.  Rank 0 prints ONE JSON line.

value      = whole-job Mrays/s: closest-hit queries (device counter) of all ranks / max-over-ranks time.
roofline   = the dominant kernel (wavefront: wf_extend_kernel, the closest-hit walk; megakernel: path_kernel) against
             8 TB/s of HBM.  `achieved` = algorithmic bytes per launch / mean launch time, where the algorithmic bytes
             are what the launch MUST move through HBM: every ray's pool record in (16 bytes for a new camera ray, 32 for
             a continued path) and its hit record out (16 bytes), plus the walked part of the scene (node and triangle
             records) once.  Launch times come from HIP events the library records on the launch stream around every
             launch of that kernel in the timed region (rt_get_timing).  `traffic` = HBM bytes per launch from the
             FETCH_SIZE (x2 on gfx950) and WRITE_SIZE counter passes of this same command, taken from
             profiles/<round>/traffic.json -- only when that file was made from the kernel sources being run (source
             hash), else null.  `demand_bytes_per_ray` = 32*box_tests + 36*triangle_tests: what the walk asks of the
             memory hierarchy per ray; LDS, L1, L2 and Infinity Cache serve nearly all of it, so it is not an HBM figure
             (it would be 15 TB/s).  `binding` names the roof the counters say the kernel sits under.
cpu_baseline = the CPU oracle (a port of upstream's logic, OpenMP) on a bounded sample of the same
             workload, rank 0, N = 1 only; `cfg1` inside it = the luyu scene at 400x240, 16 spp, depth 8 in full.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x5EED0001
FRAME_SPP = 8192
HBM_PEAK_GBPS = 8000.0


HASHED_SOURCES = ("raytrace_amd/csrc/rt_kernels.hip", "raytrace_amd/csrc/rt_device.h", "raytrace_amd/csrc/rt_packed.h",
                  "raytrace_amd/csrc/rt_host.cpp", "raytrace_amd/csrc/rt_host.h", "Makefile")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def source_hash():
    """Identifies the kernels and the scene packing a counter pass was taken on."""
    h = hashlib.sha256()
    for f in HASHED_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def workload_key(args, cfg):
    return f"{args.scene}_{cfg.width}x{cfg.height}_detail{args.detail:g}_depth{cfg.max_depth}" + ("_gpubvh" if args.gpu_bvh else "") + ("_sharedwalk" if args.share_primary == 1 else "")


def counter_passes(key):
    """The committed rocprofv3 --pmc summary for this workload, if one was taken on the sources being run."""
    sh = source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
        try:
            doc = json.load(open(path))
        except Exception:
            continue
        if doc.get("source_hash") == sh and key in doc.get("workloads", {}):
            return doc["workloads"][key], os.path.relpath(path, ROOT)
    return None, f"none: no profiles/r*/traffic.json was taken on these kernel sources (hash {sh}) for {key}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=16, help="semantic batch (upstream's <batch_size>)")
    ap.add_argument("--scene", default="formula1", choices=["formula1", "formula1_glass", "shuttle", "luyu", "stress"])
    ap.add_argument("--width", type=int, default=2000)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--cpu-spp", type=int, default=16, help="samples per pixel of the bounded CPU-baseline run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-device-tree", action="store_true", help="skip the extra timing on the device-built tree")
    ap.add_argument("--save-image", default="")
    ap.add_argument("--chunk", type=int, default=0, help="samples per lane (0 = library default)")
    ap.add_argument("--pass-spp", type=int, default=0, help="samples per buffered pass (0 = library default)")
    ap.add_argument("--lds-slots", type=int, default=0, help="traversal stack levels in LDS (0 = library default)")
    ap.add_argument("--mode", default="", choices=["", "megakernel", "wavefront"], help="path-state design (default: library default)")
    ap.add_argument("--wf-log2", type=int, default=0, help="wavefront pool capacity, log2 paths (0 = library default)")
    ap.add_argument("--wf-run", type=int, default=0)
    ap.add_argument("--wf-burst", type=int, default=0)
    ap.add_argument("--hot-nodes", type=int, default=-1)
    ap.add_argument("--wf-sort", type=int, default=-1)
    ap.add_argument("--wf-vote", type=int, default=-1)
    ap.add_argument("--wf-refill", type=int, default=0)
    ap.add_argument("--wf-block", type=int, default=0)
    ap.add_argument("--wf-lanes", type=int, default=0)
    ap.add_argument("--slot-ring-mb", type=int, default=-1, help="MiB the ring of tile slots may take (default: library default)")
    ap.add_argument("--share-equal", type=int, default=-1, help="1/0: samples that leave a mirror or glass along one ray share its walk / every continued path is walked (default: 0 for the headline, library default beside it)")
    ap.add_argument("--share-primary", type=int, default=-1, help="1/0: one closest-hit walk per pixel and semantic batch / one per sample (default: library default)")
    ap.add_argument("--hot-tune", type=int, default=-1, help="2/1/0: number the LDS-resident node records by measured visits per band of the picture / of the whole view / not at all (default: library default)")
    ap.add_argument("--gpu-bvh", action="store_true", help="build the scene tree on the device (RT_OPT_GPU_BVH) instead of re-packing upstream's trees")
    ap.add_argument("--depth", type=int, default=0, help="override the scene's max depth (experiments only: not the headline workload)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytrace_amd import capi, scene
    from raytrace_amd.dist import make_group, reduce_to_root

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            log(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes; running 1")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # RT_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks
    # (ranks share devices, the reduce goes through host memory); the measured runs use nccl (= RCCL).
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    stat_device = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- scene (replicated on every rank) ----
    frame_spp = FRAME_SPP
    t0 = time.time()
    if args.scene == "formula1":
        asm = scene.formula1(args.width, args.height, FRAME_SPP, detail=args.detail)
    elif args.scene == "formula1_glass":
        asm = scene.formula1(args.width, args.height, FRAME_SPP, glass=True, detail=args.detail)
    elif args.scene == "shuttle":
        asm = scene.shuttle(args.width, args.height, FRAME_SPP, detail=args.detail)
    elif args.scene == "stress":
        # tens of millions of triangles: 64 instances of ~470k-triangle meshes at 3840x2160 unless --width/--height/--detail say otherwise
        w, h = (3840, 2160) if (args.width, args.height) == (2000, 1200) else (args.width, args.height)
        asm = scene.stress(w, h, 1024, tris_per_instance=int(470_000 * args.detail))
        frame_spp = 1024
    else:
        asm = scene.luyu(args.width, args.height, FRAME_SPP)
    cfg = asm.config
    if args.depth:
        cfg.max_depth = args.depth
    t_assemble = time.time() - t0
    if args.gpu_bvh:
        # only the transform runs on the host; the tree is built on the device from the triangle arrays
        t_tr = time.time()
        objs = []
        for tri, (sc_, pos_, rot_) in zip(asm.triangles, asm.transforms):
            t = tri.copy()
            capi.apply_transform(t, sc_, pos_, rot_)
            objs.append((t, None))
        log(f"  transformed {len(objs)} objects in {time.time() - t_tr:.1f}s; the tree is built on the device")
    else:
        objs = scene.finalize(asm, progress=(lambda i, n: log(f"  object {i}/{n} transformed, BVH built") if (i % 8 == 0 or i == n) and rank == 0 else None))
    t_build = time.time() - t0 - t_assemble
    ctx = capi.Context(device_index)
    if args.gpu_bvh:
        ctx.set_option(capi.OPT_GPU_BVH, 1)
    t1 = time.time()
    ctx.upload_scene(objs, asm.materials, asm.textures)
    t_upload = time.time() - t1
    stats = ctx.scene_stats()
    if args.chunk:
        ctx.set_option(capi.OPT_CHUNK, args.chunk)
    if args.pass_spp:
        ctx.set_option(capi.OPT_PASS, args.pass_spp)
    if args.lds_slots:
        ctx.set_option(capi.OPT_LDS_SLOTS, args.lds_slots)
    if args.mode:
        ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT if args.mode == "wavefront" else capi.MODE_MEGAKERNEL)
    if args.wf_log2:
        ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, args.wf_log2)
    if args.wf_run:
        ctx.set_option(capi.OPT_WF_RUN, args.wf_run)
    if args.wf_burst:
        ctx.set_option(capi.OPT_WF_BURST, args.wf_burst)
    if args.wf_sort >= 0:
        ctx.set_option(capi.OPT_WF_SORT, args.wf_sort)
    if args.wf_vote >= 0:
        ctx.set_option(capi.OPT_WF_VOTE, args.wf_vote)
    if args.wf_refill:
        ctx.set_option(capi.OPT_WF_REFILL, args.wf_refill)
    if args.hot_nodes >= 0:
        ctx.set_option(capi.OPT_HOT_NODES, args.hot_nodes)
    if args.wf_block:
        ctx.set_option(capi.OPT_WF_BLOCK, args.wf_block)
    if args.wf_lanes:
        ctx.set_option(capi.OPT_WF_LANES, args.wf_lanes)
    if args.slot_ring_mb >= 0:
        ctx.set_option(capi.OPT_SLOT_RING_MB, args.slot_ring_mb)
    if args.hot_tune >= 0:
        ctx.set_option(capi.OPT_HOT_TUNE, args.hot_tune)
    # the headline counts closest-hit queries as upstream executes them, so it walks every one (see `shared_primary` on the line)
    ctx.set_option(capi.OPT_SHARE_PRIMARY, args.share_primary if args.share_primary >= 0 else 0)
    # likewise the walks of continued paths whose rays coincide (RT_OPT_SHARE_EQUAL): shared only where the first walk is
    share_equal = args.share_equal if args.share_equal >= 0 else (1 if args.share_primary == 1 else 0)
    ctx.set_option(capi.OPT_SHARE_EQUAL, share_equal)
    wavefront_mode = args.mode in (None, "", "wavefront")
    ctx.init_randstate(SEED, cfg.width, cfg.height)
    ctx.setup_raytrace(cfg.width, cfg.height, cfg.camera_pos, cfg.camera_dir, cfg.camera_up, cfg.fov)
    npix = cfg.width * cfg.height
    if rank == 0:
        log(f"scene {args.scene}: {stats['triangles']} triangles, {stats['objects']} objects, "
            f"{stats['device_bytes'] / 1e6:.1f} MB packed, models generated in {t_assemble:.1f}s, "
            f"transform + BVH build {t_build:.1f}s ({os.environ.get('RT_HOST_THREADS', 'up to 16')} threads), re-pack + upload {t_upload:.1f}s"
            + (f"; device build {stats['gpu_build_ms']:.0f} ms + triangle copy {stats['gpu_copy_ms']:.0f} ms, stack need {stats['stack_slots']}" if args.gpu_bvh else ""))

    # the reduce goes through the library (rt_group_create / rt_reduce_to_root); the gloo rehearsal, whose ranks share a
    # device (RCCL takes a device once per communicator), sums through host memory instead
    group, reduce_via = None, "none (single GPU)"
    if world > 1 and backend == "nccl":
        try:
            group = make_group(ctx, rank, world)
        except Exception as e:                      # reported on the bench line; the ranks then agree on torch's reduce below
            log(f"rank {rank}: rt_group_create failed: {e}")
        ok = torch.tensor([1 if group is not None else 0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if group is not None:
                group.close()
            group = None
        reduce_via = "rt_reduce_to_root (one ncclReduce through the C ABI)" if group is not None else "torch.distributed.reduce (rt_group_create failed on some rank)"
    elif world > 1:
        reduce_via = f"torch.distributed.reduce over {backend} through host memory (rehearsal)"
    accum = torch.zeros(npix, 3, dtype=torch.float32, device="cuda")
    staging = torch.zeros_like(accum) if world > 1 else None
    reduced = None
    stream = torch.cuda.current_stream().cuda_stream
    spp = args.spp_per_step
    assert spp % args.batch == 0, "--spp-per-step must be a multiple of --batch"

    def first_sample(step):
        # step s covers global samples [s*world*spp, (s+1)*world*spp); rank r takes the r-th slice of it
        return (step * world + rank) * spp

    def run_step(step, ev=None):
        nonlocal reduced
        if ev:
            ev[0].record()
        ctx.raytrace(npix, first_sample(step), spp, args.batch, cfg.max_depth, cfg.background, cfg.russian_roulette,
                     accum.data_ptr(), stream)
        if ev:
            ev[1].record()
        if world > 1:
            if group is not None:
                # the library's own collective: one ncclReduce (RCCL) of a copy of the buffer onto rank 0, on the render stream
                staging.copy_(accum)
                reduce_to_root(group, staging, stream)
                reduced = staging
            elif backend == "nccl":
                staging.copy_(accum)
                dist.reduce(staging, dst=0, op=dist.ReduceOp.SUM)
                reduced = staging
            else:
                host = accum.cpu()
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                reduced = host.cuda()

    # ---- instrumented pass: per-ray work mix of this kernel on this workload (not timed) ----
    ctx.set_option(capi.OPT_COUNTERS, 1)
    ctx.set_option(capi.OPT_TIMING, 1)
    ctx.reset_counters()
    scratch = torch.zeros_like(accum)
    inst_spp = min(spp, int(os.environ.get("RT_BENCH_INST_SPP", "256")))
    ctx.raytrace(npix, first_sample(0), inst_spp, args.batch, cfg.max_depth, cfg.background, cfg.russian_roulette,
                 scratch.data_ptr(), stream)
    c = ctx.counters()
    tim_i = ctx.timing()
    ctx.set_option(capi.OPT_TIMING, 0)
    del scratch
    rays_i = max(1, c[capi.CNT_RAYS])
    mix = {"box_tests_per_ray": c[capi.CNT_BOX_TESTS] / rays_i, "tri_tests_per_ray": c[capi.CNT_TRI_TESTS] / rays_i,
           "shaded_per_ray": c[capi.CNT_SHADED] / rays_i, "tex_per_ray": c[capi.CNT_TEX_FETCHES] / rays_i,
           "rays_per_path": c[capi.CNT_RAYS] / max(1, c[capi.CNT_PATHS]), "max_stack": c[capi.CNT_MAX_STACK],
           "stack_slots_needed": stats["stack_slots"],
           "node_visits_per_ray": c[capi.CNT_NODE_VISITS] / rays_i,
           # lanes doing useful work per executed trip of the node loop (1.0 = all 64 lanes busy)
           "node_loop_lane_util": c[capi.CNT_NODE_VISITS] / max(1, 64 * c[capi.CNT_WAVE_NODE_TRIPS]),
           # triangle tests per executed leaf phase and lane (2.0 = every lane tests a full leaf every time)
           "leaf_phase_tests_per_lane": c[capi.CNT_TRI_TESTS] / max(1, 64 * c[capi.CNT_WAVE_LEAF_TRIPS]),
           "boxes_per_node_visit": 0.0,
           "wave_node_trips_per_ray": c[capi.CNT_WAVE_NODE_TRIPS] * 64 / rays_i,
           "wave_leaf_trips_per_ray": c[capi.CNT_WAVE_LEAF_TRIPS] * 64 / rays_i}
    if c[capi.CNT_CYC_WAVE]:
        mix["wave_time_split"] = {"refill": round(c[capi.CNT_CYC_REFILL] / c[capi.CNT_CYC_WAVE], 4), "node": round(c[capi.CNT_CYC_NODE] / c[capi.CNT_CYC_WAVE], 4),
                                  "leaf": round(c[capi.CNT_CYC_LEAF] / c[capi.CNT_CYC_WAVE], 4),
                                  "claim_wait": round(c[capi.CNT_CYC_CLAIM] / c[capi.CNT_CYC_WAVE], 4),
                                  "cycles_per_claim": round(c[capi.CNT_CYC_CLAIM] / max(1, c[capi.CNT_CLAIMS]), 1),
                                  "node_cycles_per_trip": round(c[capi.CNT_CYC_NODE] / max(1, c[capi.CNT_WAVE_NODE_TRIPS]), 1),
                                  "leaf_cycles_per_trip": round(c[capi.CNT_CYC_LEAF] / max(1, c[capi.CNT_WAVE_LEAF_TRIPS]), 1)}
    if c[capi.CNT_WAVES] and tim_i["extend_ms"] > 0:
        # waves of the closest-hit kernel in flight, averaged over its launches (wave life spans / kernel time): the chip holds
        # 6 x 4 x 256 = 6144 of them, so the shortfall is ramp-up and drain inside the launches
        mix["extend_mean_waves_in_flight"] = round(c[capi.CNT_WAVE_SPAN_TICKS] * 1e-8 / (tim_i["extend_ms"] * 1e-3), 1)
        if c[capi.CNT_LAUNCH_SLOT_TICKS]:
            mix["extend_tail_idle_fraction"] = round(c[capi.CNT_TAIL_IDLE_TICKS] / c[capi.CNT_LAUNCH_SLOT_TICKS], 4)
            mix["extend_launch_span_ms"] = round(c[capi.CNT_LAUNCH_SLOT_TICKS] * 1e-5 / max(1, c[capi.CNT_WAVES]) * (c[capi.CNT_WAVES] / max(1, tim_i["extend_launches"])) , 3)
        mix["extend_waves_per_launch"] = round(c[capi.CNT_WAVES] / max(1, tim_i["extend_launches"]), 1)
    mix["max_visits_of_one_ray"] = c[capi.CNT_MAX_RAY_VISITS]
    mix["node_visits_from_lds"] = round(c[capi.CNT_HOT_VISITS] / max(1, c[capi.CNT_NODE_VISITS]), 4)
    mix["boxes_per_node_visit"] = c[capi.CNT_BOX_TESTS] / max(1, c[capi.CNT_NODE_VISITS])
    bytes_per_ray = 32.0 * mix["box_tests_per_ray"] + 36.0 * mix["tri_tests_per_ray"] + 100.0 * mix["shaded_per_ray"] + 16.0 * mix["tex_per_ray"]
    ctx.set_option(capi.OPT_COUNTERS, 0)

    def timed_region(first_step, n_steps):
        """n_steps steps bracketed by barrier + synchronize on both sides; max-over-ranks time, whole-job counts."""
        ctx.set_option(capi.OPT_TIMING, 1)
        ctx.reset_counters()
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_steps)]
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n_steps):
            run_step(first_step + k, events[k])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        cnt = ctx.counters()
        tim = ctx.timing()
        rays, paths, walks = float(cnt[capi.CNT_RAYS]), float(cnt[capi.CNT_PATHS]), float(cnt[capi.CNT_WALKS])
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=stat_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            r = torch.tensor([rays, paths, walks], dtype=torch.float64, device=stat_device)
            dist.all_reduce(r, op=dist.ReduceOp.SUM)
            rays, paths, walks = (float(x) for x in r.tolist())
        return {"elapsed": elapsed, "cnt": cnt, "tim": tim, "rays": rays, "paths": paths, "walks": walks,
                "step_ms": [a.elapsed_time(b) for a, b in events]}

    # ---- the same frame with the library's shared first walk (RT_OPT_SHARE_PRIMARY), timed the same way ----
    # `value` below is measured with every one of upstream's closest-hit queries walked on its own.  The samples of a semantic
    # batch start with one and the same primary ray, and the library by default walks it once for all of them: the image is
    # the same bit for bit and the frame is faster, but fewer walks are executed than rays are counted, so that rate is
    # reported beside the headline, not as it.
    shared_line = None
    if wavefront_mode and args.share_primary < 0:
        ctx.set_option(capi.OPT_SHARE_PRIMARY, 1)
        ctx.set_option(capi.OPT_SHARE_EQUAL, 1 if args.share_equal < 0 else args.share_equal)
        run_step(0)
        sh = timed_region(1, args.steps)
        ctx.set_option(capi.OPT_SHARE_PRIMARY, 0)
        ctx.set_option(capi.OPT_SHARE_EQUAL, share_equal)
        accum.zero_()
        sh_ms = sh["elapsed"] * 1e3 / max(1, args.steps)
        shared_line = {"option": "RT_OPT_SHARE_PRIMARY = 1, RT_OPT_SHARE_EQUAL = 1 (library defaults): rays that coincide bit for bit are walked once -- the first walk of a semantic batch, continued paths that leave a mirror or glass the same way; same image bit for bit",
                       "ms_per_step": round(sh_ms, 3),
                       "seconds_per_frame": round(sh_ms * 1e-3 * frame_spp / (spp * world), 3),
                       "upstream_rays_mrays_per_s": round(sh["rays"] / sh["elapsed"] / 1e6, 2),
                       "walks_executed_mrays_per_s": round(sh["walks"] / sh["elapsed"] / 1e6, 2),
                       "walks_per_upstream_ray": round(sh["walks"] / max(1.0, sh["rays"]), 4),
                       "kernel_share": {"extend_ms": round(sh["tim"]["extend_ms"], 2), "extend_launches": sh["tim"]["extend_launches"],
                                        "shade_ms": round(sh["tim"]["shade_ms"], 2), "shade_launches": sh["tim"]["shade_launches"],
                                        "wall_ms": round(sh["elapsed"] * 1e3, 2)}}

    # ---- warmup ----
    for s in range(args.warmup):
        run_step(s)
    torch.cuda.synchronize()

    # ---- timed region ----
    main = timed_region(args.warmup, args.steps)
    elapsed, cnt, tim, rays, paths, step_ms = main["elapsed"], main["cnt"], main["tim"], main["rays"], main["paths"], main["step_ms"]

    if rank == 0:
        ms_per_step = elapsed * 1e3 / max(1, args.steps)
        mrays = rays / elapsed / 1e6
        my_rays = float(cnt[capi.CNT_RAYS])
        my_paths = float(cnt[capi.CNT_PATHS])
        my_walks = float(cnt[capi.CNT_WALKS])        # closest-hit walks run = rays, unless the first walk of a batch is shared
        demand_bytes_per_ray = 32.0 * mix["box_tests_per_ray"] + 36.0 * mix["tri_tests_per_ray"]
        walked_scene_bytes = 128.0 * stats["inner_nodes"] + 48.0 * stats["triangles"]
        if tim["extend_launches"]:
            kname, k_ms_total, k_launches = "wf_extend_kernel", tim["extend_ms"], tim["extend_launches"]
            # per walk: the pool record in (one 16-byte piece for a new entry, two for a continued path) and the hit out;
            # continued paths = rays - paths whether or not new entries share their walk
            algo_bytes_per_ray = 16.0 + 16.0 * (my_rays - my_paths) / max(1.0, my_walks) + 16.0
            algo_scene_bytes = walked_scene_bytes
        else:
            kname, k_ms_total, k_launches = "path_kernel", tim["path_ms"], tim["path_launches"]
            # rays never leave the registers; per path one 12-byte sample slot out; the shading records are walked as well
            algo_bytes_per_ray = 12.0 * my_paths / max(1.0, my_rays)
            algo_scene_bytes = float(stats["device_bytes"])
        k_ms = k_ms_total / max(1, k_launches)
        rays_per_launch = my_rays / max(1, k_launches)
        walks_per_launch = (my_walks if tim["extend_launches"] else my_rays) / max(1, k_launches)
        algo_bytes_per_launch = walks_per_launch * algo_bytes_per_ray + algo_scene_bytes
        achieved = algo_bytes_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        pmc, pmc_source = counter_passes(workload_key(args, cfg))
        kp = (pmc or {}).get(kname, {})
        traffic = kp.get("hbm_bytes_per_launch")
        # the counter passes ran the same command; their launch time (profiled clocks run ~3 % lower) prices their bytes
        traffic_gbps = traffic / (kp["kernel_ms_mean"] * 1e-3) / 1e9 if traffic and kp.get("kernel_ms_mean") else None
        # the per-ray demand figure in full (box and triangle reads of the walk, shading records and texels of the hits): what the
        # path asks of the memory hierarchy, priced against HBM although LDS, L1, L2 and Infinity Cache serve nearly all of it
        survey_gbps = walks_per_launch * bytes_per_ray / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        other_workloads = {}
        if pmc_source and not pmc_source.startswith("none"):
            try:
                doc = json.load(open(os.path.join(ROOT, pmc_source)))
                for wk, wv in doc.get("workloads", {}).items():
                    if wk != workload_key(args, cfg) and "three_fractions" in wv:
                        other_workloads[wk] = wv["three_fractions"]
            except Exception:
                pass
        binding = None
        if kp.get("valu_issue_frac") is not None:
            roofs = {"vector-instruction issue (SQ_ACTIVE_INST_VALU x 4 / SIMD cycles)": kp.get("valu_issue_frac"),
                     "vector L1 accesses per clock per CU (TCP_TOTAL_CACHE_ACCESSES / CUs / cycles; ceiling 1)": kp.get("l1_accesses_per_clk_per_cu"),
                     "HBM traffic / 8 TB/s (FETCH_SIZE x 2 + WRITE_SIZE)": round(traffic_gbps / HBM_PEAK_GBPS, 4) if traffic_gbps else None}
            roofs = {k: v for k, v in roofs.items() if v is not None}
            busiest = max(roofs, key=roofs.get)
            l1_key = "vector L1 accesses per clock per CU (TCP_TOTAL_CACHE_ACCESSES / CUs / cycles; ceiling 1)"
            # The busiest unit is not by itself the one the kernel waits for.  For the closest-hit kernel two measurements decide
            # (profiles/experiments/r03_extend.md): 8-10 vector instructions fewer per node visit moved it by 0.3 %, and the L1
            # serves one lane access per clock per CU however few lanes a load has active -- the latency-bound waves of four
            # SIMDs queue for that port.
            top = l1_key if (kname == "wf_extend_kernel" and l1_key in roofs) else busiest
            binding = {"roof": top, "frac": roofs[top], "busiest_unit": busiest, "all": roofs,
                       "why": ("vector issue is the busier unit, but removing 8-10 of ~85 vector instructions per node visit changed the kernel's time by "
                               "0.3 % and adding 16 changed it by nothing; the vector L1 charges every ACTIVE lane one clock (microbenchmark tools/microbench/l1_lanes.hip: 19/34/49/64 "
                               "clocks per wave-level load with 16/32/48/64 lanes), hits included, and is what the waves queue for -- four neighbouring "
                               "lanes that ask for one address are served in one clock, which is why tracing the samples of a batch (identical "
                               "primary rays) in neighbouring lanes took 17 % of the accesses per ray away and added 10.5 % to the rate")
                               if top != busiest else None,
                       "wave_cycle_split": kp.get("wave_cycle_split"),
                       "resident_waves_mean": kp.get("mean_waves_in_flight"), "resident_wave_limit": kp.get("resident_wave_limit")}
        what = "closest-hit queries as upstream counts them; coinciding rays are walked once" if args.share_primary == 1 else "closest-hit queries"
        out = {
            "metric": f"Mrays/s ({what}), formula1 2000x1200 8192spp depth8" if args.scene == "formula1" and (cfg.width, cfg.height) == (2000, 1200)
                      else f"Mrays/s ({what}), {args.scene} {cfg.width}x{cfg.height} {frame_spp}spp depth{cfg.max_depth}",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (procedural stand-in for the formula-1 OBJ, which upstream does not ship)",
            "config": {"workload": f"scene_{args.scene}, {cfg.width}x{cfg.height}, depth {cfg.max_depth}, rr {cfg.russian_roulette}, "
                                   f"{frame_spp}-spp frame traced in steps of {spp} spp per GPU, semantic batch {args.batch}",
                       "triangles": int(stats["triangles"]), "objects": int(stats["objects"]), "spp_per_step_per_gpu": spp,
                       "mode": args.mode or "library default (wavefront)",
                       "tree": "built on the device by agglomeration (RT_OPT_GPU_BVH)" if args.gpu_bvh else "upstream's median-split trees, re-packed",
                       "partition": f"spp-split x{world}, scene replicated, RCCL reduce per step" if world > 1 else "single GPU",
                       "reduce": reduce_via,
                       "first_walk": "shared by the samples of a semantic batch (RT_OPT_SHARE_PRIMARY = 1)" if args.share_primary == 1
                                     else "one per sample, as upstream executes it (RT_OPT_SHARE_PRIMARY = 0)"},
            "seconds_per_frame": round(ms_per_step * 1e-3 * frame_spp / (spp * world), 3),
            "mpaths_per_s": round(paths / elapsed / 1e6, 2),
            "gpu_ms_per_step_events": round(float(np.mean(step_ms)), 3) if step_ms else None,
            "work_mix": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in mix.items()},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "traffic_GBps": round(traffic_gbps, 1) if traffic_gbps else None,
                         "traffic_frac": round(traffic_gbps / HBM_PEAK_GBPS, 4) if traffic_gbps else None,
                         "kernel": kname, "kernel_ms_mean": round(k_ms, 4), "launches": int(k_launches),
                         "algorithmic_bytes_per_walk": round(algo_bytes_per_ray, 2), "algorithmic_scene_bytes_per_launch": int(algo_scene_bytes),
                         "algorithmic_bytes_per_launch": int(algo_bytes_per_launch), "rays_per_launch": round(rays_per_launch, 1), "walks_per_launch": round(walks_per_launch, 1),
                         "demand_bytes_per_ray": round(demand_bytes_per_ray, 1),
                         "demand_note": "32*box_tests + 36*triangle_tests per ray: requested of LDS/L1/L2/Infinity Cache, not of HBM",
                         "survey_bytes_per_ray": round(bytes_per_ray, 1),
                         "survey_formula": "32*box_tests + 36*triangle_tests + 100*shaded + 16*texture_fetches per ray",
                         "survey_demand_GBps": round(survey_gbps, 1),
                         "survey_demand_frac": round(survey_gbps / HBM_PEAK_GBPS, 4),
                         # one glance: the same launch priced three ways against 8 TB/s
                         "three_fractions": {
                             "demand_cache_served (survey bytes per ray x walks / launch time; above 1 = not HBM's bytes)": round(survey_gbps / HBM_PEAK_GBPS, 4),
                             "counter_traffic (FETCH_SIZE x2 + WRITE_SIZE of the committed passes)": round(traffic_gbps / HBM_PEAK_GBPS, 4) if traffic_gbps else None,
                             "compulsory (frac: pool record in, hit out, walked scene once)": round(achieved / HBM_PEAK_GBPS, 5)},
                         "l2_hit_rate": kp.get("l2_hit_rate"),
                         "other_workloads": other_workloads or None,
                         "counter_passes": pmc_source},
            "binding": binding,
            "shared_primary": shared_line,
            "walks_per_upstream_ray": round(main["walks"] / max(1.0, rays), 4),
            "kernel_share": {"extend_ms": round(tim["extend_ms"], 2), "extend_launches": tim["extend_launches"],
                             "shade_ms": round(tim["shade_ms"], 2), "shade_launches": tim["shade_launches"],
                             "path_ms": round(tim["path_ms"], 2), "wall_ms": round(elapsed * 1e3, 2)},
            "source_hash": source_hash(), "workload_key": workload_key(args, cfg),
            "published_reference": {"seconds_per_frame": 70, "hardware": "RTX 4090", "note": "upstream README.md:21, different geometry"},
        }
        if world == 1 and not args.no_cpu_baseline and not args.gpu_bvh:
            out["cpu_baseline"] = cpu_baseline(asm, objs, args)
            out["cpu_baseline"]["cfg1"] = cpu_cfg1(args)
        if world == 1 and wavefront_mode and args.share_primary < 0 and not args.gpu_bvh and not args.no_device_tree:
            try:
                out["device_tree"] = device_tree_line(args, asm, cfg, objs, device_index, spp, frame_spp)
            except Exception as e:               # an extra; the line stands without it
                out["device_tree"] = {"error": str(e)}
        if args.save_image:
            # with several ranks the picture is the reduced buffer of the last step, which holds every rank's samples so far
            save_image(ctx, reduced if world > 1 else accum, cfg, (args.warmup + args.steps) * spp * world, args.save_image)
        print(json.dumps(out), flush=True)
    if group is not None:
        group.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def device_tree_line(args, asm, cfg, objs, device_index, spp, frame_spp):
    """The same steps on the library's fastest configuration: the scene tree built on the device (RT_OPT_GPU_BVH: another
    tree than upstream's, so exact ties may go to another triangle -- why it is not the headline) and the first walk of a
    batch shared.  A context of its own; timed like the headline (synchronize on both sides)."""
    import torch
    from raytrace_amd import capi
    ctx = capi.Context(device_index)
    try:
        ctx.set_option(capi.OPT_GPU_BVH, 1)
        t0 = time.time()
        ctx.upload_scene([(t, None) for t, _ in objs], asm.materials, asm.textures)
        t_load = time.time() - t0
        st = ctx.scene_stats()
        ctx.init_randstate(SEED, cfg.width, cfg.height)
        ctx.setup_raytrace(cfg.width, cfg.height, cfg.camera_pos, cfg.camera_dir, cfg.camera_up, cfg.fov)
        npix = cfg.width * cfg.height
        accum = torch.zeros(npix, 3, dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        ctx.raytrace(npix, 0, spp, args.batch, cfg.max_depth, cfg.background, cfg.russian_roulette, accum.data_ptr(), stream)
        torch.cuda.synchronize()
        ctx.reset_counters()
        t0 = time.perf_counter()
        for k in range(args.steps):
            ctx.raytrace(npix, (1 + k) * spp, spp, args.batch, cfg.max_depth, cfg.background, cfg.russian_roulette, accum.data_ptr(), stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        cnt = ctx.counters()
        ms = dt * 1e3 / max(1, args.steps)
        return {"option": "RT_OPT_GPU_BVH = 1 with the library's defaults otherwise: tree built on the device by agglomeration, coinciding rays walked once",
                "ms_per_step": round(ms, 3), "seconds_per_frame": round(ms * 1e-3 * frame_spp / spp, 3),
                "upstream_rays_mrays_per_s": round(cnt[capi.CNT_RAYS] / dt / 1e6, 2),
                "walks_executed_mrays_per_s": round(cnt[capi.CNT_WALKS] / dt / 1e6, 2),
                "device_build_ms": round(st["gpu_build_ms"], 1), "triangle_copy_ms": round(st["gpu_copy_ms"], 1), "upload_call_s": round(t_load, 3)}
    finally:
        ctx.close()


def cpu_cfg1(args):
    """BASELINE config 1 in full on the CPU path: the luyu scene (assets/scene/scene_luyu.txt:6-58, stand-in statue) at
    400x240, 16 spp, depth 8."""
    from oracle import oracle as orc
    from raytrace_amd import scene
    asm = scene.luyu(400, 240, 16)
    t0 = time.perf_counter()
    objs = scene.finalize(asm, backend=orc.OracleBackend())
    t_build = time.perf_counter() - t0
    sc = orc.Scene(objs, asm.materials, asm.textures)
    t0 = time.perf_counter()
    _, cnt = sc.render(orc.make_params(asm.config, SEED, args.batch), 0, 16)
    dt = time.perf_counter() - t0
    return {"workload": f"scene_luyu 400x240, 16 spp, depth 8, {asm.n_triangles} triangles, whole frame", "seconds_per_frame": round(dt, 3),
            "value": round(float(cnt[0]) / dt / 1e6, 3), "unit": "Mrays/s", "cores": orc.num_threads(), "kind": "port",
            "transform_and_bvh_build_s": round(t_build, 3)}


def cpu_baseline(asm, objs, args):
    """The oracle (a CPU port of upstream's path logic; upstream's own cpu-rt is a different renderer and cannot be
    built here) on the same scene and frame size for --cpu-spp samples per pixel, all host threads OpenMP gives us."""
    from oracle import oracle as orc
    sc = orc.Scene(objs, asm.materials, asm.textures)
    p = orc.make_params(asm.config, SEED, args.batch)
    t0 = time.perf_counter()
    _, cnt = sc.render(p, 0, args.cpu_spp)
    dt = time.perf_counter() - t0
    return {"value": round(float(cnt[0]) / dt / 1e6, 3), "unit": "Mrays/s", "cores": orc.num_threads(), "kind": "port",
            "sample": f"same scene, {asm.config.width}x{asm.config.height}, {args.cpu_spp} spp ({int(cnt[0])} rays in {dt:.1f}s)",
            "seconds_per_frame_extrapolated": round(dt * FRAME_SPP / args.cpu_spp, 1)}


def save_image(ctx, accum, cfg, spp_done, path):
    from raytrace_amd import scene_files
    img = accum.clone()
    ctx.post_process(cfg.width * cfg.height, spp_done, cfg.gamma, img.data_ptr())
    ctx.synchronize()
    a = img.cpu().numpy().reshape(cfg.height, cfg.width, 3)[::-1]
    scene_files.write_png_rgb(path, np.clip(np.rint(a * 255.0), 0, 255).astype(np.uint8))


if __name__ == "__main__":
    main()
