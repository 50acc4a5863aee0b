#!/bin/bash
# the native present loop (dogeray --group 1: one displayed image per frame through the pipeline, display divide + download of every frame) timed from outside:
# the difference between a 2404-frame and a 404-frame run takes the scene load out
mkdir -p gpurun_out
export TMPDIR=/tmp
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4t_native_loop.txt
import os, subprocess, time
import bench
S = bench.ensure_scene('/tmp/dogeray_bench', 709, 1920, 1080)
exe = os.path.join(os.getcwd(), "dogeray_amd", "bin", "dogeray")
def run(opts, group, frames):
    env = dict(os.environ, DOGERAY_OPTIONS=opts)
    t0 = time.perf_counter()
    subprocess.run([exe, S, "--textures", "", "--frames", str(frames), "--group", str(group), "--cache", "--quiet", "--out", "/tmp/o.bmp"], env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return time.perf_counter() - t0
run("", 8, 50)
for opts, group in (("pipe_group=8", 1), ("pipe_group=1", 1), ("pipe_group=16", 1), ("pipe_group=8", 8), ("pipe_group=8", 32)):
    a = min(run(opts, group, 404) for _ in range(2)); b = min(run(opts, group, 2404) for _ in range(2))
    print("DOGERAY_OPTIONS=%s dogeray --group %d: 404 frames %.3f s, 2404 frames %.3f s -> %.3f ms per frame%s" % (opts, group, a, b, (b - a) / 2000 * 1e3,
          " (every frame displayed: divide + 6 MB download)" if group == 1 else " (one present per %d frames)" % group), flush=True)
PY
