"""Builds libdogeray_amd.so (HIP kernels for gfx950 + host C++ + the C ABI) in-tree with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdogeray_amd.so")
HOST_SOURCES = ["rts_reader.cpp", "bvh_builder.cpp", "linearise.cpp", "wide_builder.cpp", "capi_host.cpp", "group.cpp", "context.cpp"]
# one translation unit per family of kernels (kernels.hpp)
DEVICE_SOURCES = ["kernels_render.hip", "kernels_aux.hip"]
# -ffp-contract=off: no FMA contraction on host or device -- the BVH build and the kernel's
# arithmetic are specified operation by operation (DESIGN.md "arithmetic contract").
COMMON = ["-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wextra",
          "-Wno-unused-parameter", "-pthread"]
ARCH = "gfx950"


def _hipcc():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    headers.append(os.path.join(HERE, "..", "include", "dogeray_amd.h"))
    objs = []
    hipcc = _hipcc()
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    cmds = []
    for src in HOST_SOURCES + DEVICE_SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(HERE, "_build", src + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + headers + [os.path.abspath(__file__)]):
            # -fno-slp-vectorize: hipcc otherwise packs adjacent scalar f32 ops into v_pk_* instructions, which
            # issue slower than the scalars they replace on gfx950 (measured: +4.6 % rays/s without them)
            slp = [] if os.environ.get("DOGERAY_SLP") == "1" else ["-fno-slp-vectorize"]      # DOGERAY_SLP=1: experiment knob
            # -enable-post-misched=0: without the post-register-allocation machine scheduler the persistent kernel is 1.1 % faster
            # (0.6176 against 0.6244 ms/frame, three interleaved runs on two boxes; same registers, no spills); max-ilp / iterative-ilp /
            # max-memory-clause scheduling strategies: -0.4 % / +2.6 % / 0
            # -amdgpu-use-amdgpu-trackers=1 (the scheduler tracks register pressure with the target's own trackers): another 1.0 % on round 3's
            # kernel (0.5670 against 0.5733 ms/frame, profiles/r3_p_*; same 80 VGPRs, no spills); relaxed occupancy / no high-pressure reschedule: +-0
            sched = ["-mllvm", "-enable-post-misched=0", "-mllvm", "-amdgpu-use-amdgpu-trackers=1"]
            cmd = [hipcc] + COMMON + slp + sched + ["--offload-arch=" + ARCH, "-c", sp, "-o", obj]
            if src.endswith(".cpp"):
                cmd = [hipcc] + COMMON + ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-c", sp, "-o", obj]
            cmds.append(cmd)
    if cmds:      # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(cmds), max(1, (os.cpu_count() or 2) // 2))) as ex:
            list(ex.map(run, cmds))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs + ["-pthread", "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # the headless host application: plain C++ over the C ABI only
    exe = os.path.join(HERE, "bin", "dogeray")
    main_src = os.path.join(CSRC, "dogeray_main.cpp")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    if force or _stale(exe, [main_src, LIB] + headers):
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", main_src, "-o", exe, "-L" + HERE, "-ldogeray_amd", "-Wl,-rpath,$ORIGIN/.."]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
