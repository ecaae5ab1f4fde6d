#!/usr/bin/env python3
"""The benchmark configurations as scene files for `bin/main <scene> <outdir> <batch>`.

    tools/make_bench_scenes.py                 rewrite the five tracked text files under benchmarks/scenes/
    tools/make_bench_scenes.py --assets NAME…  also write NAME's model files (OBJ, MTL, PNG) under benchmarks/assets/NAME/

The text files are small and tracked.  The model files are procedural stand-ins for the models upstream's scene files
download (raytrace_amd/procgen.py), deterministic, 10 MB to 1.2 GB of text per scene: they are generated where they are
needed and not tracked.  Paths inside the scene files are relative to the repository root, where bin/main is started,
as in upstream's assets/scene/*.txt.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raytrace_amd import scene, scene_files  # noqa: E402

# name -> (assembler, what BASELINE.json calls it)
CONFIGS = {
    "luyu": (lambda: scene.luyu(400, 240, 16), "config 1: scene_luyu at 400x240, 16 spp, depth 8"),
    "formula1": (lambda: scene.formula1(2000, 1200, 8192), "config 2: scene_formula1 textured, 2000x1200, 8192 spp, depth 8"),
    "formula1_glass": (lambda: scene.formula1(2000, 1200, 8192, glass=True), "config 3: scene_formula1, every car triangle refractive"),
    "shuttle": (lambda: scene.shuttle(2000, 1200, 8192), "config 4: scene_shuttle, 2000x1200, 8192 spp (spp-split across GPUs with RT_GPUS)"),
    "stress": (lambda: scene.stress(3840, 2160, 1024), "config 5: 64 instances of seven 470k-triangle meshes, 3840x2160, 1024 spp"),
}


def write(name, assets, root=ROOT, assembler=None):
    asm = (assembler or CONFIGS[name][0])()
    directory = os.path.join(root, "benchmarks", "assets", name)
    os.makedirs(os.path.join(root, "benchmarks", "scenes"), exist_ok=True)
    return scene_files.write_scene(asm, directory, name=name, relative_to=root, write_models=assets,
                                   scene_path=os.path.join(root, "benchmarks", "scenes", f"scene_{name}.txt")), asm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--assets", nargs="*", default=None, help="scenes whose model files to write as well (no name = all)")
    args = ap.parse_args()
    with_assets = set(CONFIGS) if args.assets == [] else set(args.assets or [])
    for name in CONFIGS:
        if args.assets is not None and name not in with_assets:
            continue
        path, asm = write(name, name in with_assets)
        print(f"{path}: {asm.n_triangles} triangles" + (", model files written" if name in with_assets else ""), flush=True)


if __name__ == "__main__":
    main()
