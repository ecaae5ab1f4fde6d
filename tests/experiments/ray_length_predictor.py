"""How well does a path segment's traversal length predict the next segment's?  (Analysis behind the note on the
closest-hit launch's tail under profiles/experiments/; not a test.  Uses the CPU oracle, one ray per call, for its per-ray box-test count.)

    python tests/experiments/ray_length_predictor.py [n_paths=15000]
"""
import os
import sys
import time

import numpy as np
from scipy.stats import spearmanr

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc          # noqa: E402
from raytrace_amd import scene           # noqa: E402


def main():
    n_paths = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
    asm = scene.formula1(width=160, height=96, n_samples=1, detail=0.5, tex_size=64)
    objs = scene.finalize(asm)
    oracle = orc.Scene(objs, asm.materials, asm.textures)
    rays = orc.primary_rays(orc.make_params(asm.config, 0x5EED0001, 16), 0)
    rng = np.random.default_rng(1)

    def trace(rays):
        n = len(rays)
        boxes, obj, tri, t = np.zeros(n), np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.float32)
        for i in range(n):
            o, k, tt, _, _, c = oracle.closest_hit(rays[i:i + 1])
            boxes[i], obj[i], tri[i], t[i] = c[1], o[0], k[0], tt[0]
        return boxes, obj, tri, t

    def bounce(rays, obj, tri, t):
        """Diffuse continuation off the geometric normal of every hit; (new rays, index of the ray each one continues)."""
        out, keep = [], []
        for i in range(len(rays)):
            if obj[i] < 0:
                continue
            p = objs[obj[i]][0][tri[i]]["v"]["position"]
            n = np.cross(p[1] - p[0], p[2] - p[0])
            length = np.linalg.norm(n)
            if length == 0:
                continue
            n /= length
            d = rays[i, 3:6]
            if np.dot(d, n) > 0:
                n = -n
            while True:
                r = rng.uniform(-1, 1, 3)
                if r @ r < 1:
                    break
            nd = n + r
            nd /= np.linalg.norm(nd)
            out.append(np.concatenate([rays[i, 0:3] + d * t[i] + nd * 1e-4, nd]))
            keep.append(i)
        return np.array(out, dtype=np.float32), np.array(keep)

    t0 = time.time()
    seg = [rays[rng.choice(len(rays), min(n_paths, len(rays)), replace=False)]]
    counts, parents = [], []
    for depth in range(4):
        c, o, k, t = trace(seg[-1])
        counts.append(c)
        nxt, keep = bounce(seg[-1], o, k, t)
        seg.append(nxt)
        parents.append(keep)
    print(f"{[len(s) for s in seg[:4]]} rays per segment, {time.time() - t0:.1f} s; mean box tests {[round(c.mean(), 1) for c in counts]}, "
          f"max {[int(c.max()) for c in counts]}")
    for d in range(3):
        prev, cur = counts[d][parents[d]], counts[d + 1]
        line = f"segment {d} -> {d + 1}: Spearman {spearmanr(prev, cur).correlation:.2f}"
        for q in (0.9, 0.98):
            longest = cur >= np.quantile(cur, q)
            for pq in (0.5, 0.75):
                share = ((prev > np.quantile(prev, pq)) & longest).sum() / longest.sum()
                line += f"; top {100 * (1 - q):.0f} % with previous above its {pq:.2f} quantile: {share:.2f}"
        print(line)


if __name__ == "__main__":
    main()
