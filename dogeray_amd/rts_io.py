"""Writing .rts scenes without Blender (the reference's only writer is plugin/rtsexport.py, which needs bpy).

The format, as the reference's reader consumes it (kernel.cu read() K:1186-1530) and its exporter emits it
(rtsexport.py P:205-315): text, one record per line, comma separated.
  /...   comment                                  K:1218
  *,camx,camy,camz,aperture,lookx,looky,lookz,focus,fov,max_depth,spp,background,envmap|no,width,height      K:1230-1293
  one object per line, 38 columns (trailing columns may be omitted; they then keep the struct defaults K:55-71):
     0-2 pos (vertex 0 / sphere centre) | 3 type (0 sphere, 2 triangle) | 4-6 colour | 7 roughness or IOR | 8 diffuse mode
     9-11 dim (vertex 1, or radius in column 9) | 12 material (0 diffuse, 1 emissive, 2 mirror, 3 metal, 4 glass, 5 glossy)
     13-15 rot (vertex 2) | 16-18 face normal | 19-27 vertex normals | 28-33 uv pairs | 34 smooth | 35 checker
     36 albedo texture name|no | 37 roughness texture name|no
Numbers are written with "%f" (six decimals) like the exporter (P:207); the reader's stoi() takes the integer part
of "45.000000".  write_rts() -> Scene.load() reproduces every field that six decimals can carry.
"""
import numpy as np

from . import OBJECT_DTYPE


def _f(v):
    return "%f" % float(v)


def write_rts(path, objects, settings=None, texture_names=(), ncols=38, comment="written by dogeray_amd.rts_io"):
    """objects: OBJECT_DTYPE array of the N objects to write (Scene.objects() returns N+1 slots, the last one is the
    reference's never-written slot: pass objects()[:-1]); settings: DrSettings-like or None; texture_names[i]: name written for texnum == i."""
    objects = np.asarray(objects, dtype=OBJECT_DTYPE)
    if ncols < 13 or ncols > 38:
        raise ValueError("ncols must be between 13 and 38")

    def tex(i):
        return texture_names[i] if 0 <= i < len(texture_names) else "no"

    with open(path, "w") as f:
        if comment:
            f.write("/" + comment + "\n")
        if settings is not None:
            s = settings
            f.write("*," + ",".join(_f(v) for v in (s.campos[0], s.campos[1], s.campos[2], s.aperture, s.look[0], s.look[1], s.look[2],
                                                    s.focus_dist, s.fov, s.max_depth, s.spp, s.background)) +
                    ",%s,%d,%d\n" % (tex(s.backtex), s.width, s.height))
        for o in objects:
            cols = [_f(o["pos"][0]), _f(o["pos"][1]), _f(o["pos"][2]), "%d" % o["type"], _f(o["col"][0]), _f(o["col"][1]), _f(o["col"][2]),
                    _f(o["addional"][1]), _f(o["addional"][0]), _f(o["dim"][0]), _f(o["dim"][1]), _f(o["dim"][2]), _f(o["mat"]),
                    _f(o["rot"][0]), _f(o["rot"][1]), _f(o["rot"][2])]
            cols += [_f(v) for v in o["norm"]]
            for k in ("n1", "n2", "n3"):
                cols += [_f(v) for v in o[k]]
            for k in ("t1", "t2", "t3"):
                cols += [_f(o[k][0]), _f(o[k][1])]
            cols += [_f(1 if o["smooth"] else 0), _f(1 if o["tex"] else 0), tex(int(o["texnum"])), tex(int(o["rtexnum"]))]
            f.write(",".join(cols[:ncols]) + "\n")
    return path


def validate_rts(path):
    """Checks a file the way the reader would fail on it; returns a list of (line number, problem)."""
    problems = []
    n_objects = 0
    with open(path, "r", errors="replace") as f:
        for ln, line in enumerate(f.read().split("\n")[:-1], 1):       # the reader takes lines up to the last newline
            if line.startswith("/"):
                continue
            fields = line.split(",")
            if line.startswith("*"):
                numeric = [(i, v) for i, v in enumerate(fields) if 1 <= i <= 12 or i in (14, 15)]
            else:
                n_objects += 1
                numeric = [(i, v) for i, v in enumerate(fields) if i <= 35]
                if len(fields) > 3 and fields[3].strip()[:1] not in ("0", "2"):
                    problems.append((ln, "type %r is neither 0 (sphere) nor 2 (triangle): never hit" % fields[3]))
            for i, v in numeric:
                if v == "r" and not line.startswith("*"):
                    continue
                try:
                    float(v.strip().split()[0]) if v.strip() else float("")
                except (ValueError, IndexError):
                    problems.append((ln, "column %d: %r is not a number (stof/stoi would throw)" % (i, v)))
    if n_objects < 2:
        problems.append((0, "fewer than 2 objects: the reference's BVH build does not terminate"))
    return problems
