#!/usr/bin/env python3
"""Times rt_raytrace on the many-instance stress scene at several mesh sizes, with and without the counting kernels.
Usage: python tools/stress_probe.py [spp] [detail ...]"""
import sys, time
sys.path.insert(0, ".")
import torch
from raytrace_amd import capi, scene

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
details = [float(x) for x in sys.argv[2:]] or [0.02, 0.1, 0.5, 1.0]
for detail in details:
    asm = scene.stress(3840, 2160, 1024, tris_per_instance=int(470_000 * detail))
    cfg = asm.config
    objs = scene.finalize(asm)
    ctx = capi.Context(0)
    ctx.upload_scene(objs, asm.materials, asm.textures)
    ctx.init_randstate(1, cfg.width, cfg.height)
    ctx.setup_raytrace(cfg.width, cfg.height, cfg.camera_pos, cfg.camera_dir, cfg.camera_up, cfg.fov)
    st = ctx.scene_stats()
    npix = cfg.width * cfg.height
    buf = torch.zeros(npix, 3, device="cuda")
    torch.cuda.synchronize()
    for counters, first, seed in ((1, 0, 1), (0, 0, 1), (0, 16, 1), (0, 16, 0x5EED0001), (0, 32, 0x5EED0001)):
        ctx.init_randstate(seed, cfg.width, cfg.height)
        ctx.set_option(capi.OPT_COUNTERS, counters)
        ctx.set_option(capi.OPT_TIMING, 1)
        ctx.reset_counters()
        t0 = time.perf_counter()
        ctx.raytrace(npix, first, spp, 16, cfg.max_depth, cfg.background, cfg.russian_roulette, buf.data_ptr())
        c = ctx.counters()
        dt = time.perf_counter() - t0
        tim = ctx.timing()
        print(f"detail {detail} tris {st['triangles']} MB {st['device_bytes'] / 1e6:.0f} counters {counters} first {first} seed {seed:#x}: {dt * 1e3:.1f} ms, "
              f"{c[capi.CNT_RAYS] / dt / 1e6:.1f} Mrays/s, extend {tim['extend_ms']:.1f} ms / {tim['extend_launches']} launches, shade {tim['shade_ms']:.1f} ms, max visits of one ray {c[capi.CNT_MAX_RAY_VISITS]}, non-finite rays {c[capi.CNT_NONFINITE_RAYS]}, "
              f"visits/ray {(c[capi.CNT_NODE_VISITS]) / max(1, c[capi.CNT_RAYS]):.1f}",
              flush=True)
    ctx.close()
    del buf
