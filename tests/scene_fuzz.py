"""Random .rts scenes for parity fuzzing (shared by CPU and GPU tests)."""
import numpy as np


def fmt(x):
    return "%f" % x


def random_scene(rng, n, path, W=96, H=64, textures=(), spheres=True, duplicates=True, degenerate=True, scale=1.0):
    """Writes a scene with n objects: triangles with random column counts (16..38), spheres, every material,
    exact duplicates (equal-t ties), degenerate and axis-aligned triangles, shared vertices.
    scale: every position, radius and the focus distance times this (the reference pads leaf boxes by an ABSOLUTE 0.01: a scene a fiftieth of the size has
    triangles as small as the padding, and the wide tree enters them with their own bounds -- wide_tree = 2)."""
    lines = []
    cam = (rng.uniform(-1, 1, 3) * 0.5 + np.array([0, -1.0, 6.0])) * scale
    depth = int(rng.integers(1, 7))
    spp = int(rng.integers(1, 3))
    env = "no"
    if textures and rng.random() < 0.5:
        env = textures[int(rng.integers(0, len(textures)))]
    lines.append("*," + ",".join(fmt(v) for v in cam) + ",%f,0,0,0,%f,%d,%d,%d,%f,%s,%d,%d" % (
        rng.choice([0.0, 0.01, 0.3]) * scale, rng.uniform(3, 8) * scale, int(rng.integers(30, 70)), depth, spp, rng.uniform(0.3, 1.2), env, W, H))
    objs = []
    while len(objs) < n:
        r = rng.random()
        if spheres and r < 0.15:
            c = rng.uniform(-2.5, 2.5, 3) * scale
            mat = int(rng.choice([0, 1, 2, 3, 4, 5]))
            cols = rng.uniform(0.1, 1.0, 3) if mat != 1 else rng.uniform(1, 5, 3)
            objs.append(",".join([fmt(c[0]), fmt(c[1]), fmt(c[2]), "0", fmt(cols[0]), fmt(cols[1]), fmt(cols[2]),
                                  fmt(rng.choice([0.0, 0.2, 1.5])), "0", fmt(rng.uniform(0.2, 0.9) * scale), "0", "0", str(mat)]))
            continue
        v0 = rng.uniform(-3, 3, 3)
        if degenerate and r < 0.2:
            v1, v2 = v0 + np.array([1.0, 0, 0]), v0 + np.array([2.0, 0, 0])          # zero-area
        elif r < 0.35:
            v0 = np.round(v0)                                                         # axis-aligned, integer coordinates
            v1, v2 = v0 + np.array([2.0, 0, 0]), v0 + np.array([0, 0, 2.0])
        else:
            v1, v2 = v0 + rng.uniform(-1.5, 1.5, 3), v0 + rng.uniform(-1.5, 1.5, 3)
        v0, v1, v2 = v0 * scale, v1 * scale, v2 * scale
        mat = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5]))
        col = rng.uniform(0.1, 1.0, 3) if mat != 1 else rng.uniform(1, 4, 3)
        add_y = 1.5 if mat == 4 else float(rng.choice([0.0, 0.1, 0.5]))
        add_x = float(rng.choice([0, 0, 1]))
        nrm = np.cross(v1 - v0, v2 - v0)
        nl = np.linalg.norm(nrm)
        nrm = nrm / nl if nl > 0 else np.array([0, -1.0, 0])
        cols = [fmt(v0[0]), fmt(v0[1]), fmt(v0[2]), "2", fmt(col[0]), fmt(col[1]), fmt(col[2]), fmt(add_y), fmt(add_x),
                fmt(v1[0]), fmt(v1[1]), fmt(v1[2]), fmt(mat), fmt(v2[0]), fmt(v2[1]), fmt(v2[2])]
        ncols = int(rng.choice([16, 19, 28, 34, 36, 37, 38]))
        if ncols >= 19:
            cols += [fmt(v) for v in nrm]
        if ncols >= 28:
            for _ in range(3):
                vn = nrm + rng.normal(scale=0.2, size=3)
                vn /= np.linalg.norm(vn)
                cols += [fmt(v) for v in vn]
        if ncols >= 34:
            cols += [fmt(v) for v in rng.uniform(-1, 2, 6)]
        if ncols >= 36:
            cols += [fmt(float(rng.integers(0, 2))), fmt(float(rng.integers(0, 2)))]
        if ncols >= 37:
            cols.append(textures[int(rng.integers(0, len(textures)))] if textures and rng.random() < 0.5 else "no")
        if ncols >= 38:
            cols.append(textures[int(rng.integers(0, len(textures)))] if textures and rng.random() < 0.3 else "no")
        line = ",".join(cols)
        objs.append(line)
        if duplicates and rng.random() < 0.1 and len(objs) < n:
            # the same triangle again with another colour: equal t, the reference keeps the first it reaches
            c2 = cols[:]
            c2[4:7] = [fmt(v) for v in rng.uniform(0.1, 1.0, 3)]
            objs.append(",".join(c2))
    lines += objs[:n]
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path


def stadium_scene(path, n=64, ratio=1.5, W=96, H=64):
    """n triangles whose size and distance grow geometrically: the surface-area heuristic peels them off one or two at a time, so the wide tree reaches the
    deepest shape the builder allows (wide_depth == WIDE_MAX_DEPTH = 17 for n = 64, ratio = 1.5 and for n = 200, ratio = 1.1: every word of the
    kernel's per-lane stack is used).  The camera looks along the row, through all of them."""
    with open(path, "w") as f:
        f.write("*,-3.0,-1.0,0.5,0.01,%f,0,0,10,45,4,1,1,no,%d,%d\n" % (ratio ** (n - 1), W, H))
        for i in range(n):
            x = ratio ** i
            s = 0.5 * x
            mat = (0, 3, 5, 2)[i % 4]
            f.write("%f,%f,%f,2,0.8,0.7,0.6,0.3,0,%f,%f,%f,%d,%f,%f,%f\n" % (x, -s, -s, x, s, -s, mat, x, 0, s))
    return path
