#!/usr/bin/env python3
"""bench.py -- Mrays/s + ms/frame of the DOGERAY render path on MI355X.

Workload (BASELINE.json metric; config "highpoly.rts (~1M tris) 1920x1080"): the reference scene is a
missing blob, so the stand-in SURVEY.md 8(d) specifies is generated: a 709x709-vertex heightfield
(1 002 528 triangles, 38-column .rts, smooth normals, materials {0,3,5}), 1920x1080, 1 spp per frame,
depth 10.  A "step" is one full frame (every pixel once).  A "ray" is one closest-hit query (one hit()
call, kernel.cu K:800).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R]

What is timed: after W warm-up frames, the K-frame region (inputs resident, barrier + synchronize on
both sides) is run R times (default 10); `ms_per_step` is the MEDIAN region time / K (min and max are
reported beside it) and `value` = rays of one region / median time.  The library renders the K frames
of a region in launches of up to --batch frames (one work queue over all their tiles: BATCHED
throughput); `single_frame_ms` is the same scene with ONE frame per launch, what the reference's
present loop does (one CudaStarter call per displayed frame, K:2154-2224).

N > 1: the framebuffer is tiled by interleaved 8-pixel block columns, one rank per GPU, and gathered to
rank 0 over RCCL every --gather-every frames (scaling: strong, the frame is fixed).  Started by
torch.distributed.run (WORLD_SIZE set) it is one process per GPU over torch.distributed; started plainly
(`python bench.py --gpus N`) it is ONE process whose N ranks are the library's own dr_group (a context and a
host thread per GPU, ncclCommInitAll + ncclSend / grouped ncclRecv; csrc/group.cpp).  DOGERAY_GROUP_DEVICES=0,0
rehearses that path on one GPU (peer copies instead of RCCL).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_LANE_OPS = 78.65e12   # 256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz = the 157.3 TFLOP/s fp32 vector peak / 2 flops per fma (same guide)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def under_profiler():
    """True when this process was started by rocprofv3 & co: their preloaded library may already have initialised
    the GPU, and a process in that state must not start other programs."""
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def usable_cpus():
    """CPUs this process may actually use: affinity mask, capped by the cgroup's CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def scenegen(*argv):
    """tools/scenegen.cpp, called in-process through tools/libscenegen.so (built by __graft_entry__.build())."""
    import ctypes
    so = os.path.join(ROOT, "tools", "libscenegen.so")
    src = os.path.join(ROOT, "tools", "scenegen.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    lib = ctypes.CDLL(so)
    args = [b"scenegen"] + [os.fsencode(str(a)) for a in argv]
    arr = (ctypes.c_char_p * len(args))(*args)
    rc = lib.scenegen_run(len(args), arr)
    if rc != 0:
        raise RuntimeError("scenegen %s failed (%d)" % (" ".join(map(str, argv)), rc))


def ensure_scene(cache_dir, verts, W, H):
    path = os.path.join(cache_dir, "heightfield_%d_%dx%d.rts" % (verts, W, H))
    if not os.path.exists(path):
        os.makedirs(cache_dir, exist_ok=True)
        tmp = path + ".tmp%d" % os.getpid()
        scenegen("heightfield", tmp, verts, W, H)
        os.replace(tmp, path)
    return path


def select_workload(args):
    """(scene path, texture directory, W, H, label) of --config.  C4 (default) is the configuration BASELINE.json's metric is quoted on;
    C2 / C3 / C5 are the other configs' stand-ins, so that their rates come from this same harness."""
    os.makedirs(args.cache, exist_ok=True)
    if args.config == "C4":
        return ensure_scene(args.cache, args.verts, args.width, args.height), "", args.width, args.height, \
            "C4 stand-in for samples/highpoly.rts: heightfield"
    if args.config == "C2":
        path = os.path.join(args.cache, "bunnyish_6_1280x720.rts")
        if not os.path.exists(path):
            scenegen("bunnyish", path + ".tmp", 6, 1280, 720); os.replace(path + ".tmp", path)
        return path, "", 1280, 720, "C2 stand-in for samples/sanford.blend.rts: displaced icosphere"
    if args.config == "C5":
        path = os.path.join(args.cache, "city_200_3840x2160.rts")
        if not os.path.exists(path):
            scenegen("city", path + ".tmp", 200, 3840, 2160); os.replace(path + ".tmp", path)
        return path, "", 3840, 2160, "C5 stand-in for samples/city.blend.rts: box city"
    if args.config == "C3":        # the reference's own bolter2.blend.rts with its two textures (fixtures of tests/golden/reference_image)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import reference_image as ri
        path, texdir = ri.bolter_scene(args.cache)
        return path, texdir, 1920, 1080, "C3 (sponza absent): samples/bolter2.blend.rts + boltersmall.ppm + env.ppm"
    raise SystemExit("unknown --config %r" % (args.config,))


def measure_counters(args):
    """PMC counters of the timed render kernel, per launch, from rocprofv3 child runs of this same command -- each group
    in its own pass, kernel-trace/stats never combined with --pmc (MI355X_MICROARCH.md, HBM / rocprofv3 sections):
    FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of 16-byte-per-lane reads, so it
    is doubled (calibrated for streaming reads; taken over for this kernel's 16-byte gathers).  The SQ pass gives the
    wave-level VALU instruction count and the active-lane count behind `roofline` (bound: VALU issue).
    Returns None on any problem."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    out = {}
    tmp = tempfile.mkdtemp(prefix="dogeray_pmc_")
    try:
        groups = (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_WAVES"],
                  # the instruction classes gfx950's SQ counts separately (round 4): what the issue slots are spent on
                  ["SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32"],
                  ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_INT64", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"])
        for gi, group in enumerate(groups):
            d = os.path.join(tmp, group[0])
            cmd = [prof, "--pmc"] + group + ["-d", d, "-o", "pmc", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                   "--config", args.config, "--steps", str(args.steps), "--warmup", str(max(args.warmup, args.steps)), "--batch", str(args.batch), "--traversal", args.traversal,
                   "--verts", str(args.verts), "--width", str(args.width), "--height", str(args.height), "--cache", args.cache,
                   "--repeats", "1", "--no-cpu-baseline", "--no-traffic", "--no-extras"]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, env=env, cwd="/tmp")
            if r.returncode != 0:
                if gi >= 3:                               # the class counters are an extra: without them the roofline stands as before
                    out.setdefault("unavailable", []).extend(group)
                    continue
                return None
            vals = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    # the timed launches only: the non-counting build of the render kernel
                    if row["Counter_Name"] in group and "render_" in row["Kernel_Name"] and "<false" in row["Kernel_Name"]:
                        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for name in group:
                v = vals.get(name)
                if not v:
                    if gi >= 3:
                        out.setdefault("unavailable", []).append(name)
                        continue
                    return None
                full = max(v)                         # launches that cover a full batch report the largest value
                v = [x for x in v if x > 0.8 * full]
                out[name] = sum(v) / len(v)
        out["FETCH_SIZE"] *= 1024.0
        out["WRITE_SIZE"] *= 1024.0
        out["bytes_per_launch"] = 2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def algorithmic_bytes(c, frames, W, H):
    """32 B per node visit, 36 B per triangle test, 128 B per shaded hit, 4 B per texel,
    12 B per pixel written per frame."""
    return 32 * c["node_visits"] + 36 * c["prim_tests"] + 128 * c["shades"] + 4 * c["texels"] + 12 * W * H * frames


def metric_label(args, W, H):
    """BASELINE.json's metric, named for the configuration that is run (--config)."""
    if args.config == "C4":
        return "Mrays/sec + ms/frame, 1M-tri .rts at 1920x1080"
    return "Mrays/sec + ms/frame, config %s at %dx%d (BASELINE.json's headline is C4: 1M-tri .rts at 1920x1080)" % (args.config, W, H)


def main_group(args):
    """`python bench.py --gpus N` without a launcher: the N ranks are the library's dr_group -- one context and one host thread
    per GPU in THIS process, stripes gathered to rank 0 by ncclSend / grouped ncclRecv on a second stream per rank (peer copies
    when ranks share a device: DOGERAY_GROUP_DEVICES=0,0).  Same frames, same seeds, same JSON line as the one-GPU run."""
    import numpy as np
    import dogeray_amd as dr
    N = args.gpus
    env_dev = [int(x) for x in os.environ.get("DOGERAY_GROUP_DEVICES", "").split(",") if x.strip() != ""]
    devices = [env_dev[r] if r < len(env_dev) else r for r in range(N)]
    if args.batch <= 0:
        args.batch = min(256, 32 * N)
    if args.gather_every <= 0:
        args.gather_every = args.batch
    scene_path, texdir, W, H, label = select_workload(args)
    t0 = time.time(); scene = dr.Scene.load(scene_path, texdir); t_parse = time.time() - t0
    t0 = time.time(); scene.build_bvh(); t_bvh = time.time() - t0
    s = scene.settings()
    ntris = scene.num_objects
    grp = dr.Group(devices)
    t0 = time.time(); grp.upload(scene); t_upload = time.time() - t0
    mode = {"wide": dr.TRAVERSAL_WIDE, "ordered": dr.TRAVERSAL_ORDERED, "threaded": dr.TRAVERSAL_THREADED}[args.traversal]
    ctxs = [grp.context(r) for r in range(N)]
    for c in ctxs:
        c.set_traversal(mode)
        c.set_option("batch_frames", min(args.batch, 256))
        if os.environ.get("DOGERAY_RESERVE_CUS"):      # experiment: room for the gather beside the rendering
            c.set_option("reserve_cus", int(os.environ["DOGERAY_RESERVE_CUS"]))
    st = dr.pack_settings13(s, 1, spp=1)
    log("scene %s: %d triangles, parse %.1fs, BVH %.1fs, upload on %d ranks %.2fs; transport %s" % (
        os.path.basename(scene_path), ntris, t_parse, t_bvh, N, t_upload, "rccl" if grp.uses_rccl else "copy"))
    seed_base, seed_stride = 1, 1000003
    grp.accum_reset(W, H)

    def run_frames(first, count):     # returns when every rank's stream and the gather have drained (dr_group joins its threads)
        grp.render_accumulate(st, W, H, s.background, seed_base + first * seed_stride, seed_stride, count, args.gather_every)

    run_frames(0, args.warmup)
    region_s, reps = [], []
    for rep in range(max(1, args.repeats)):
        for c in ctxs:
            c.stats_reset()
        t0 = time.perf_counter()
        run_frames(args.warmup, args.steps)
        dt = time.perf_counter() - t0
        region_s.append(dt)
        reps.append((dt, [c.stats() for c in ctxs]))
    srt = sorted(region_s)
    elapsed = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    timed = min(reps, key=lambda r: abs(r[0] - elapsed))      # the statistics of the MEDIAN repeat, the one `value` is quoted on (as the one-GPU path does)
    # the assembled accumulator against one context rendering the whole frame: the same frames, bit for bit
    grp.accum_reset(W, H)
    run_frames(args.warmup, min(args.steps, 4))
    assembled = grp.accum_read()
    # rays: count what the timed frames traced, every rank its stripe
    for c in ctxs:
        c.enable_counters(True); c.stats_reset()
    grp.accum_reset(W, H)
    run_frames(args.warmup, args.steps)
    counted = [c.stats() for c in ctxs]
    for c in ctxs:
        c.enable_counters(False)
    rays = sum(x["rays"] for x in counted)
    ranks_transport = grp.rccl_ranks
    uses_rccl = grp.uses_rccl
    grp.close()
    whole = dr.Context(devices[0]).upload(scene)
    whole.set_traversal(mode)
    whole.accum_reset(W, H)
    whole.render_accumulate(st, W, H, s.background, seed_base + args.warmup * seed_stride, seed_stride, min(args.steps, 4))
    identical = bool(np.array_equal(whole.accum_read(), assembled))
    whole.close()
    frames = args.steps
    per_rank_ms = [x["kernel_ms"] for x in timed[1]]
    result = {
        "metric": metric_label(args, W, H),
        "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / frames * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "%s, %d triangles, %dx%d, 1 spp/frame, depth %d, %s traversal" % (label, ntris, W, H, s.max_depth, args.traversal),
            "triangles": ntris, "width": W, "height": H, "spp_per_frame": 1, "max_depth": int(s.max_depth), "frames": frames,
            "parallelism": "framebuffer block-column stripes x%d, one process (dr_group: a context and a host thread per GPU)" % N,
            "gather_every": args.gather_every, "frames_per_launch": min(args.batch, 256, args.gather_every),
        },
        "launcher": "dr_group (in-process)",
        "transport": "rccl" if uses_rccl else "copy",
        "rccl_ranks": ranks_transport,
        "devices": devices,
        "assembled_frame_identical_to_one_context": identical,
        "repeats": len(region_s), "ms_per_step_min": min(region_s) / frames * 1e3, "ms_per_step_max": max(region_s) / frames * 1e3,
        "rays_per_frame": rays / frames, "primary_samples_per_s": (W * H * frames) / elapsed,
        "kernel_ms_per_rank": per_rank_ms,
        # what the timed region spends outside the slowest rank's kernels (host threads, gather of the last batch): region wall - max rank kernel time
        "region_ms": timed[0] * 1e3, "region_minus_slowest_rank_ms": timed[0] * 1e3 - max(per_rank_ms),
        "region_overhead_frac": (timed[0] * 1e3 - max(per_rank_ms)) / (timed[0] * 1e3),
        "setup_s": {"parse": t_parse, "bvh_build": t_bvh, "upload": t_upload},
        "roofline": None, "cpu_baseline": None,
        "note": "roofline and cpu_baseline are reported by the one-GPU run (same kernels; N = 1 is the configuration the metric is quoted on)",
    }
    print(json.dumps(result), flush=True)
    return 0 if identical else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", choices=["C2", "C3", "C4", "C5"], default="C4",
                    help="C4 (default) = the configuration the metric is quoted on; C2 / C3 / C5 = the other configs' stand-ins from the same harness")
    ap.add_argument("--verts", type=int, default=709, help="heightfield vertices per side (709 -> 1 002 528 triangles)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--traversal", choices=["wide", "threaded", "ordered"], default="wide")
    ap.add_argument("--batch", type=int, default=0, help="frames per kernel launch (default 32 x GPUs, at most 256: a launch has to outlast its longest pixel, ~3 ms)")
    ap.add_argument("--gather-every", type=int, default=0, help="frames between two gathers to rank 0 (default: --batch)")
    ap.add_argument("--repeats", type=int, default=10, help="how often the K-step timed region is run; the median is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes that fill roofline (traffic, VALU counters)")
    ap.add_argument("--no-extras", action="store_true", help="skip single_frame_ms and the gather-ceiling probe")
    ap.add_argument("--cpu-col-mod", type=int, default=1, help="cpu_baseline renders every n-th block column")
    ap.add_argument("--cache", default=os.environ.get("DOGERAY_BENCH_CACHE", "/tmp/dogeray_bench"))
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return main_group(args)          # one process, N ranks inside the library (dr_group)
    if world != args.gpus:
        args.gpus = world

    if args.batch <= 0:
        args.batch = min(256, 32 * world)     # constant work per launch and GPU: a launch has to outlast its longest pixel (~3 ms)
    # HBM traffic of the timed kernel: rocprofv3 PMC passes over child runs of this same command.  Done first,
    # before this process touches the GPU (a process that has initialised HIP should not spawn programs).
    traffic_probe = None
    if world == 1 and not args.no_traffic and not under_profiler():
        select_workload(args)
        t0 = time.time()
        traffic_probe = measure_counters(args)
        log("counter probe (3 rocprofv3 passes): %.1f s -> %s" % (time.time() - t0, "ok" if traffic_probe else "unavailable"))

    import torch
    import dogeray_amd as dr
    from dogeray_amd import multigpu

    dist = None
    backend = os.environ.get("DOGERAY_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N > 1 path on a box with fewer GPUs than ranks
    device_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(device_index)
        # a collective that is never matched must end the run, not hang it: torch's watchdog aborts the process after this long (default 10 min)
        import datetime
        limit = datetime.timedelta(seconds=int(os.environ.get("DOGERAY_BENCH_TIMEOUT_S", "240")))
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index), timeout=limit)
        else:
            dist.init_process_group(backend=backend, timeout=limit)

    t0 = time.time()
    if rank == 0:
        select_workload(args)
    if dist is not None:
        dist.barrier()
    scene_path, texdir, W, H, label = select_workload(args)
    t0 = time.time()
    scene = dr.Scene.load(scene_path, texdir)
    t_parse = time.time() - t0
    t0 = time.time()
    scene.build_bvh()
    t_bvh = time.time() - t0
    s = scene.settings()
    ntris = scene.num_objects
    ctx = dr.Context(device_index)
    t0 = time.time()
    ctx.upload(scene)
    t_upload = time.time() - t0
    mode = {"wide": dr.TRAVERSAL_WIDE, "ordered": dr.TRAVERSAL_ORDERED, "threaded": dr.TRAVERSAL_THREADED}[args.traversal]
    ctx.set_traversal(mode)
    ctx.set_stripe(world, rank)
    if args.gather_every <= 0:
        args.gather_every = args.batch
    ctx.set_option("batch_frames", min(args.batch, 256))
    st = dr.pack_settings13(s, 1, spp=1)
    if rank == 0:
        log("scene %s: %d triangles, parse %.1fs, BVH %.1fs, upload %.2fs" % (os.path.basename(scene_path), ntris, t_parse, t_bvh, t_upload))

    seed_base, seed_stride = 1, 1000003
    ctx.accum_reset(W, H)
    # N > 1: backend nccl (= RCCL) gathers on the device with the library's pack / unpack kernels, stream-ordered (no host
    # wait between a batch and its gather); backend gloo (rehearsal on a box with fewer GPUs than ranks) goes through host copies
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    gatherer = acc = None
    if world > 1:
        if backend == "nccl":
            gatherer = multigpu.StripeGatherer(ctx, W, H, world, rank, torch.device("cuda", device_index))
        else:
            acc = multigpu.accumulator_tensor(ctx, torch.device("cuda", device_index))
            gatherer = multigpu.FrameGatherer(acc.cpu(), W, H, world, rank)

    def run_frames(first, count):
        """Render frames [first, first+count) into the accumulator; gather every --gather-every frames."""
        if world == 1:
            ctx.render_accumulate(st, W, H, s.background, seed_base + first * seed_stride, seed_stride, count)
            return
        k = 0
        while k < count:
            n = min(args.gather_every, count - k)
            if backend == "nccl":
                ctx.render_accumulate_async(st, W, H, s.background, seed_base + (first + k) * seed_stride, seed_stride, n)
                gatherer.gather_async()
            else:
                ctx.render_accumulate(st, W, H, s.background, seed_base + (first + k) * seed_stride, seed_stride, n)
                gatherer.gather(acc.cpu())
            k += n
        if backend == "nccl":
            gatherer.finish()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup, then the timed region, `repeats` times (same frames, same seeds: same work every time)
    run_frames(0, args.warmup)
    region_s, rep_stats = [], []
    for rep in range(max(1, args.repeats)):
        ctx.stats_reset()
        fence()
        t0 = time.perf_counter()
        run_frames(args.warmup, args.steps)
        fence()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        region_s.append(dt)
        rep_stats.append(ctx.stats())
    srt = sorted(region_s)
    elapsed = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    # launch statistics (HIP events on the render stream, clock stamps) of the MEDIAN repeat: the one `value` is quoted on
    timed = rep_stats[min(range(len(region_s)), key=lambda i: (abs(region_s[i] - elapsed), i))]

    # ---- one frame per launch (the reference's call pattern), same frames
    single = None
    if world == 1 and not args.no_extras:
        ctx.set_option("batch_frames", 1)
        n1 = min(args.steps, 16)
        ctx.accum_reset(W, H)
        ctx.render_accumulate(st, W, H, s.background, seed_base + args.warmup * seed_stride, seed_stride, n1)     # establishes the tile order
        ctx.stats_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.render_accumulate(st, W, H, s.background, seed_base + args.warmup * seed_stride, seed_stride, n1)
        wall1 = time.perf_counter() - t0
        st1 = ctx.stats()
        single = {"kernel_ms": st1["kernel_ms"] / max(1, st1["frames"]), "wall_ms": wall1 / n1 * 1e3, "frames": n1,
                  "launches": st1["launches"]}
        # the same one-frame launches through the pipeline (frame k + 1 starts while frame k drains; per-frame buffers added in order),
        # without and with the display divide + download of every frame (what an interactive viewer pays)
        try:
            n2 = max(n1, 64)
            ctx._acc_shape = (W, H, 3)
            ctx.render_accumulate_pipelined(st, W, H, s.background, seed_base + args.warmup * seed_stride, seed_stride, 8)
            walls = []
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.render_accumulate_pipelined(st, W, H, s.background, seed_base + args.warmup * seed_stride, seed_stride, n2)
                walls.append((time.perf_counter() - t0) / n2 * 1e3)
            single["pipelined_wall_ms"] = sorted(walls)[1]
            single["pipe_group"] = ctx.get_option("pipe_group")
            # ... and with a launch per frame (pipe_group 1: round 3's pipeline, two launches side by side)
            ctx.set_option("pipe_group", 1)
            walls = []
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.render_accumulate_pipelined(st, W, H, s.background, seed_base + args.warmup * seed_stride, seed_stride, n2)
                walls.append((time.perf_counter() - t0) / n2 * 1e3)
            single["pipelined_group1_wall_ms"] = sorted(walls)[1]
            ctx.set_option("pipe_group", single["pipe_group"])
            t0 = time.perf_counter()
            pend = []
            window = 2 * max(1, single["pipe_group"])
            for k in range(n2):
                pend.append(ctx.pipeline_submit(st, W, H, s.background, seed_base + (args.warmup + k) * seed_stride, k + 1))
                if len(pend) == window:
                    ctx.pipeline_wait(pend.pop(0), want_image=True, in_place=True)
            for t in pend:
                ctx.pipeline_wait(t, want_image=True, in_place=True)
            single["pipelined_present_wall_ms"] = (time.perf_counter() - t0) / n2 * 1e3      # display divide + 6 MB download per frame, image read in place
            single["pipelined_frames"] = n2
        except Exception as e:
            log("pipelined single frames failed: %r" % (e,))
        ctx.set_option("batch_frames", min(args.batch, 256))

    # ---- gather ceiling of this GPU on the resident walk array (roofline.gather)
    gather = None
    if world == 1 and not args.no_extras and ctx.get_option("traversal") == dr.TRAVERSAL_WIDE:
        try:
            gather = {"l2_resident_3MB": ctx.probe_gather(3 * 1024 * 1024 // 64, 4000), "whole_array": ctx.probe_gather(0, 2000),
                      "array_bytes": 64 * (ctx.get_option("wide_nodes") + ntris)}
        except Exception as e:
            log("gather probe failed: %r" % (e,))

    # ---- untimed: count what the timed frames traced (same seeds -> same paths)
    ctx.enable_counters(True)
    ctx.stats_reset()
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, seed_base + args.warmup * seed_stride, seed_stride, args.steps)
    own = ctx.stats()
    ref_order = own
    if mode != dr.TRAVERSAL_THREADED:
        # algorithmic bytes are defined on the reference's traversal order: count that too
        ctx.set_traversal(dr.TRAVERSAL_THREADED)
        ctx.stats_reset()
        ctx.accum_reset(W, H)
        ctx.render_accumulate(st, W, H, s.background, seed_base + args.warmup * seed_stride, seed_stride, args.steps)
        ref_order = ctx.stats()
        ctx.set_traversal(mode)
    ctx.enable_counters(False)
    rays_local = own["rays"]
    rays = rays_local
    if dist is not None:
        rt = torch.tensor([rays_local], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(rt)
        rays = int(rt.item())

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    frames = args.steps
    kernel_ms = timed["kernel_ms"] / max(1, timed["frames"])          # HIP events on the render stream, rank 0; per FRAME
    launch_ms = timed["kernel_ms"] / max(1, timed["launches"])        # per kernel launch (a launch may cover a batch of frames)
    frames_per_launch = timed["frames"] / max(1, timed["launches"])
    abytes = algorithmic_bytes(ref_order, frames, W if world == 1 else W // world, H) / frames
    kbytes = algorithmic_bytes(own, frames, W if world == 1 else W // world, H) / frames
    algorithmic_gbs = abytes / (kernel_ms * 1e-3) / 1e9
    records_per_s = own["node_visits"] / frames / (kernel_ms * 1e-3)          # records the kernel's own walk fetched (nodes + leaves)
    result = {
        "metric": "Mrays/sec + ms/frame, 1M-tri .rts at 1920x1080" if args.config == "C4" else "Mrays/sec + ms/frame, config %s" % args.config,
        "value": rays / elapsed / 1e6,
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / frames * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "%s, %d triangles, %dx%d, 1 spp/frame, depth %d, %s traversal" % (label, ntris, W, H, s.max_depth, args.traversal),
            "triangles": ntris, "width": W, "height": H, "spp_per_frame": 1, "max_depth": int(s.max_depth),
            "frames": frames, "parallelism": "framebuffer block-column stripes x%d" % world,
            "gather_every": args.gather_every if world > 1 else None, "frames_per_launch": frames_per_launch,
            "mode": "batched: %d frames per launch share one work queue (single_frame_ms: one frame per launch)" % int(round(frames_per_launch)),
        },
        "repeats": len(region_s),
        "ms_per_step_min": min(region_s) / frames * 1e3,
        "ms_per_step_max": max(region_s) / frames * 1e3,
        "single_frame_ms": single["kernel_ms"] if single else None,
        "single_frame_pipelined_ms": single.get("pipelined_wall_ms") if single else None,
        "single_frame_pipelined_group1_ms": single.get("pipelined_group1_wall_ms") if single else None,
        "single_frame": single,
        "rays_per_frame": rays / frames,
        "primary_samples_per_s": (W * H * frames) / elapsed,
        "kernel_ms_per_frame": kernel_ms,
        "per_ray": {
            "reference_order": {"V": ref_order["node_visits"] / ref_order["rays"], "L": ref_order["prim_tests"] / ref_order["rays"],
                                "S": ref_order["shades"] / ref_order["rays"], "T": ref_order["texels"] / ref_order["rays"],
                                "bytes": abytes * frames / ref_order["rays"]},
            "kernel": {"V": own["node_visits"] / own["rays"], "L": own["prim_tests"] / own["rays"],
                       "bytes": kbytes * frames / own["rays"]},
        },
        # What binds this kernel is VALU issue -- one more VALU instruction per node step costs its full 2.4-cycle issue time -- with the CUs' texture
        # addressers as a second nearly full resource (busy 59 % of the launch: one cache access per active lane and load, profiles/r4_o_node_three_units.txt;
        # one more load per lane and step costs 1-5 %).  The headline is the VALU
        # roofline on USEFUL work: active-lane VALU operations per second against 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz.
        # The HBM figures SURVEY 8(d) defines are kept beside it: `hbm` (bytes that reach the fabric, PMC) and `hbm_algorithmic`
        # (the reference traversal's bytes, mostly served by L1/L2/Infinity Cache -- it may exceed the HBM peak).
        "roofline": {
            "bound": "valu",
            "achieved": None, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "Tlaneop/s", "frac": None,
            "traffic": None,
            "valu": None,
            "hbm": None,
            "hbm_algorithmic": {"bytes_per_launch": abytes * frames_per_launch, "GBs": algorithmic_gbs, "frac_of_hbm_peak": algorithmic_gbs / HBM_PEAK_GBS,
                                "note": "SURVEY 8(d): 32 V + 36 L + 128 S + 4 T + 12 W H with the REFERENCE traversal's visit counts; these bytes are what the "
                                        "reference algorithm asks for, not what reaches HBM"},
            "gather": None,
            "kernel_own_visit_bytes_per_launch": 64 * own["node_visits"] / frames * frames_per_launch,      # records visited x their 64-byte stride (a node's fourth 16 bytes are not fetched)
            "launch_ms": launch_ms,
            "frames_per_launch": frames_per_launch,
            "note": "launch_ms: HIP events on the library's render stream in the MEDIAN repeat of the timed region (the repeat `value` is quoted on); "
                    "counters (traffic, valu): rocprofv3 --pmc child runs of this same command with --repeats 1, per launch",
        },
        "timed_waves": {"shader_clock_mhz": 100.0 * timed["diag"][0] / max(1, timed["diag"][7]),
                        "wave_cycles_per_frame": timed["diag"][0] / max(1, timed["frames"])},
        "setup_s": {"parse": t_parse, "bvh_build": t_bvh, "upload": t_upload},
        "diag": own.get("diag"),

        "simd_efficiency": {"node_loop": ref_order["node_visits"] / max(1, ref_order["trav_slots"]),
                            "bounce_loop": ref_order["rays"] / max(1, ref_order["ray_slots"])},
    }

    rf = result["roofline"]
    if gather is not None:
        rf["gather"] = {"records_per_s": records_per_s, "ceiling_l2_resident": gather["l2_resident_3MB"], "ceiling_whole_array_random": gather["whole_array"],
                        "frac_of_l2_resident_ceiling": records_per_s / gather["l2_resident_3MB"], "record_bytes": 64, "array_bytes": gather["array_bytes"],
                        "note": "64-byte records fetched by the kernel's own walk per second, against dr_context_probe_gather on the same resident array: "
                                "divergent dependent fetches within a 3 MB set (L2-resident) and over the whole array (random: below what coherent rays see)"}
    if traffic_probe is not None:
        t = traffic_probe
        lane_ops = t["SQ_THREAD_CYCLES_VALU"] / (launch_ms * 1e-3)
        rf["achieved"] = lane_ops / 1e12
        rf["frac"] = lane_ops / VALU_PEAK_LANE_OPS
        rf["traffic"] = t["bytes_per_launch"]
        rf["valu"] = {"wave_instructions_per_launch": t["SQ_INSTS_VALU"], "active_lane_ops_per_launch": t["SQ_THREAD_CYCLES_VALU"],
                      "lane_use": t["SQ_THREAD_CYCLES_VALU"] / (64.0 * t["SQ_INSTS_VALU"]),
                      "issue_busy_frac_at_2_cycles_per_instruction": t["SQ_INSTS_VALU"] * 2.0 / (1024 * 2.4e9 * launch_ms * 1e-3),
                      "issue_cost_note": "measured issue cost per wave-level instruction at six waves per SIMD (tools/valu_rate.hip, profiles/r3_l_*): 2.4 cycles "
                                         "(fma / mul / add f32, and / or / xor / bitop3, add u32, mov), 4.2 (min / max, compares, cndmask, conversions, shifts, "
                                         "fma_mix, perm, every f64 operation), 8.2 (rcp, sqrt); the classes overlap partly",
                      "wave_instructions_per_ray": t["SQ_INSTS_VALU"] / (rays / frames * frames_per_launch),
                      "salu_per_launch": t["SQ_INSTS_SALU"], "vmem_rd_per_launch": t["SQ_INSTS_VMEM_RD"]}
        # How close to the issue limit: the SQ's per-class instruction counts priced with the measured issue cost of each class (tools/valu_rate.hip: 2.4 SIMD
        # cycles for fma / add / mul f32, 8.2 for rcp / sqrt, 4.2 for conversions and every f64 operation), over the SIMD cycles of the launch.  The SQ does not
        # split the integer / logic / compare / select / move instructions (INT32 and the unnamed rest) into the 2.4- and the 4.2-cycle members of that group, so
        # the fraction comes as a pair: all of them at 2.4 (low) and all at 4.2 (high); the static ISA has them about half and half (tools/isa_cost.py).
        cls = {k: t[k] for k in t if k.startswith("SQ_INSTS_VALU_")}
        if cls and "SQ_INSTS_VALU_INT32" in cls:
            simd_cycles = 1024 * 2.4e9 * launch_ms * 1e-3
            fast = cls.get("SQ_INSTS_VALU_FMA_F32", 0) + cls.get("SQ_INSTS_VALU_ADD_F32", 0) + cls.get("SQ_INSTS_VALU_MUL_F32", 0)
            trans = cls.get("SQ_INSTS_VALU_TRANS_F32", 0)
            slow = cls.get("SQ_INSTS_VALU_CVT", 0) + cls.get("SQ_INSTS_VALU_FMA_F64", 0) + cls.get("SQ_INSTS_VALU_ADD_F64", 0) + cls.get("SQ_INSTS_VALU_MUL_F64", 0)
            mixed = max(0.0, t["SQ_INSTS_VALU"] - fast - trans - slow)      # INT32, INT64, compares, selects, moves, min / max ...
            rf["valu"]["classes_per_launch"] = dict(cls, other=mixed - cls.get("SQ_INSTS_VALU_INT32", 0) - cls.get("SQ_INSTS_VALU_INT64", 0))
            rf["valu"]["issue_priced_frac_low"] = (2.4 * fast + 8.2 * trans + 4.2 * slow + 2.4 * mixed) / simd_cycles
            rf["valu"]["issue_priced_frac_high"] = (2.4 * fast + 8.2 * trans + 4.2 * slow + 4.2 * mixed) / simd_cycles
            rf["valu"]["issue_priced_frac"] = (2.4 * fast + 8.2 * trans + 4.2 * slow + 3.3 * mixed) / simd_cycles
            for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"):
                if k in t:
                    rf["valu"][k.lower() + "_per_launch"] = t[k]
        if t.get("unavailable"):
            rf["valu"]["counters_unavailable"] = t["unavailable"]
        hbm_gbs = t["bytes_per_launch"] / (launch_ms * 1e-3) / 1e9
        rf["hbm"] = {"achieved_GBs": hbm_gbs, "peak_GBs": HBM_PEAK_GBS, "frac": hbm_gbs / HBM_PEAK_GBS,
                     "note": "rocprofv3 PMC, separate passes, per launch of the timed kernel: 2 x FETCH_SIZE (%.3g B raw; gfx950 reports half of 16-B/lane reads) "
                             "+ WRITE_SIZE (%.3g B)" % (t["FETCH_SIZE"], t["WRITE_SIZE"])}

    if world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import orc
            ncpu = usable_cpus()
            t0 = time.time()
            osc = orc.Scene(scene_path, texdir or None)
            osc.build_bvh()
            t_setup = time.time() - t0
            # frames of the timed region (same seeds) until about 2 s of wall time = 2 s x ncpu of CPU work
            cpu_rays, dt, nfr, img = 0, 0.0, 0, None
            while nfr < args.steps and dt < 2.0:
                t0 = time.perf_counter()
                fr, c = osc.render(st, W, H, s.background, seed_base + (args.warmup + nfr) * seed_stride, nthreads=ncpu,
                                   col_mod=args.cpu_col_mod, col_rem=0)
                dt += time.perf_counter() - t0
                cpu_rays += c["rays"]
                if img is None:
                    img = fr
                nfr += 1
            # parity spot check on the sampled block columns against the GPU's first timed frame
            ctx.accum_reset(W, H)
            ctx.render_accumulate(st, W, H, s.background, seed_base + args.warmup * seed_stride, 0, 1)
            gpu = ctx.accum_read()
            import numpy as np
            cols = (np.arange(W) // 8) % args.cpu_col_mod == 0
            same = float(np.all(gpu[cols] == img[cols], axis=2).mean())
            # one thread, every 2nd block column of the first frame (about 1.5 s of work)
            t0 = time.perf_counter()
            _, c1 = osc.render(st, W, H, s.background, seed_base + args.warmup * seed_stride, nthreads=1, col_mod=2, col_rem=0)
            dt1 = time.perf_counter() - t0
            result["cpu_baseline"] = {
                "value": cpu_rays / dt / 1e6, "unit": "Mrays/s", "cores": ncpu, "kind": "port",
                "single_thread": {"value": c1["rays"] / dt1 / 1e6, "unit": "Mrays/s", "cores": 1,
                                  "sample": "every 2nd block column of the first timed frame: %d rays in %.2f s" % (c1["rays"], dt1)},
                "sample": "oracle (oracle/dogeray_oracle.cpp), %d std::threads (the CPUs this process may use), every %d-th 8-pixel block "
                          "column of the first %d %dx%d frames of the timed region (same scene, same seeds): %d rays in %.2f s "
                          "(+%.1f s oracle parse+BVH, not timed)" % (ncpu, args.cpu_col_mod, nfr, W, H, cpu_rays, dt, t_setup),
                "pixels_identical_to_gpu_on_sample": same,
            }
            # north_star's named baseline: the build's OWN kernel source compiled for the host (tools/host_kernel.cpp = device_core.hpp with
            # -DDR_HOST_BUILD, wide walk, std::threads over block columns), same sample of the same frames
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import host_kernel as hk
                hsc = hk.Scene(scene_path, texdir or "")
                hk_rays, hdt, hfr, himg = 0, 0.0, 0, None
                while hfr < args.steps and hdt < 2.0:
                    t0 = time.perf_counter()
                    fr, c = hsc.render(st, W, H, s.background, seed_base + (args.warmup + hfr) * seed_stride, traversal=2, nthreads=ncpu,
                                       col_mod=args.cpu_col_mod, col_rem=0)
                    hdt += time.perf_counter() - t0
                    hk_rays += c["rays"]
                    if himg is None:
                        himg = fr
                    hfr += 1
                t0 = time.perf_counter()
                _, h1 = hsc.render(st, W, H, s.background, seed_base + args.warmup * seed_stride, traversal=2, nthreads=1, col_mod=2, col_rem=0)
                hdt1 = time.perf_counter() - t0
                result["cpu_baseline"]["same_source"] = {
                    "value": hk_rays / hdt / 1e6, "unit": "Mrays/s", "cores": ncpu, "kind": "same-source",
                    "single_thread": {"value": h1["rays"] / hdt1 / 1e6, "unit": "Mrays/s", "cores": 1},
                    "sample": "tools/host_kernel.cpp (the library's device_core.hpp compiled for the host, wide walk), %d std::threads, the same block columns of "
                              "the first %d frames: %d rays in %.2f s" % (ncpu, hfr, hk_rays, hdt),
                    "pixels_identical_to_gpu_on_sample": float(np.all(gpu[cols] == himg[cols], axis=2).mean()),
                }
            except Exception as e:
                result["cpu_baseline"]["same_source"] = {"value": None, "kind": "same-source", "sample": "failed: %r" % (e,)}
        except Exception as e:   # the baseline is a reported extra; never lose the GPU line over it
            result["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}

    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
