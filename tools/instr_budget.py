"""Instruction budget of the persistent kernel: ISA-level instruction counts of one node step, one leaf step, one shade /
refill phase and the loop's bookkeeping in the lean build `render_persistent_kernel<false, 6, 32, 16, 2, true, false>`, from a
-DDR_ISA_MARKS=1 -save-temps build (the marks are assembler comments: they change no instruction), times the per-frame counts and mean
active lanes of the bench scene (counting build, bench.py `diag`).  No GPU needed:   python tools/instr_budget.py > profiles/r3_instr_budget.txt
Regions: a region runs from a mark to the next mark in program order; code the scheduler moved across a mark is counted where it
landed (marks are volatile asm, the compiler keeps them in order but may move ordinary instructions past them), so single numbers
are good to a few instructions.  The kernel loop holds TWO merged steps (unroll 2): both copies are listed."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "render_persistent_kernelILb0ELi6ELi32ELi16ELi2ELb1ELb0EE"      # <false, 6, 32, 16, 2, true, false>
# per frame, bench scene 1920x1080 (profiles/r2_b; reproduced on this round's boxes: gpurun_out/r3a_ab.txt)
PER_FRAME = {"rays": 3.284e6, "node_steps": 1.70e6, "leaf_steps": 1.01e6, "phases": 8.83e4, "iterations": 8.54e5}
LANES = {"node": 30, "leaf": 23, "phase": 37}

def main():
    d = tempfile.mkdtemp(prefix="dr_isa_")
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm", "-enable-post-misched=0", "-mllvm", "-amdgpu-use-amdgpu-trackers=1",
           "-DDR_ISA_MARKS=1", "--offload-arch=gfx950", "-c", os.path.join(ROOT, "dogeray_amd", "csrc", "kernels_render.hip"), "-o", os.path.join(d, "r.o"), "-save-temps"]
    subprocess.check_call(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(os.path.join(d, "kernels_render-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    m = re.search(r"^_ZN2dr24%s\w*:\n(.*?)^\s*\.end_amdhsa_kernel|^_ZN2dr24%s\w*:\n(.*?)\.Lfunc_end" % (KERNEL, KERNEL), text, re.S | re.M)
    start = text.index("_ZN2dr24" + KERNEL)
    start = text.index(":\n", start)
    end = text.index(".Lfunc_end", start)
    body = text[start:end].split("\n")
    regions, cur, order = {}, "prologue", []
    def bump(kind):
        r = regions.setdefault(cur, {"valu": 0, "valu_f64": 0, "salu": 0, "vmem": 0, "lds": 0, "branch": 0, "other": 0})
        r[kind] += 1
    seen = {}
    for line in body:
        t = line.strip()
        mm = re.match(r";\s*DRMARK (\w+)", t)
        if mm:
            name = mm.group(1)
            seen[name] = seen.get(name, 0) + 1
            cur = "%s#%d" % (name, seen[name])
            order.append(cur)
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            bump("valu")
            if "_f64" in op: bump("valu_f64")
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")): bump("vmem")
        elif op.startswith("ds_"): bump("lds")
        elif op.startswith(("s_cbranch", "s_branch")): bump("branch")
        elif op.startswith("s_"): bump("salu")
        else: bump("other")
    print("# Instruction budget of the persistent kernel's lean build (tools/instr_budget.py; ISA of this tree, -DDR_ISA_MARKS=1 changes no instruction)")
    print("# region = from this mark to the next one in program order; '#k' = k-th copy of the mark in the kernel (the loop holds two merged steps)")
    print("%-18s %6s %6s %6s %6s %6s %6s" % ("region", "VALU", "(f64)", "SALU", "VMEM", "LDS", "branch"))
    tot = {}
    for name in ["prologue"] + order:
        r = regions.get(name)
        if not r: continue
        print("%-18s %6d %6d %6d %6d %6d %6d" % (name, r["valu"], r["valu_f64"], r["salu"], r["vmem"], r["lds"], r["branch"]))
        base = name.split("#")[0]
        t = tot.setdefault(base, {"valu": 0, "salu": 0, "n": 0})
        t["valu"] += r["valu"]; t["salu"] += r["salu"]; t["n"] += 1
    def mean(base):
        t = tot.get(base, {"valu": 0, "n": 1}); return t["valu"] / max(1, t["n"])
    node, leaf, phase = mean("node_begin"), mean("leaf_begin"), mean("phase_begin")
    # what lies between a step's end mark and the next begin (fetch, ballots, parking decision) and the loop top
    glue = mean("node_end") + mean("leaf_end") + mean("loop_top") + mean("phase_end")
    pf = PER_FRAME
    print()
    print("# per frame of the bench scene (counting build): %.3g rays, %.3g wave-level node steps at %d lanes, %.3g leaf steps at %d, %.3g shade/refill phases at %d, %.3g loop iterations"
          % (pf["rays"], pf["node_steps"], LANES["node"], pf["leaf_steps"], LANES["leaf"], pf["phases"], LANES["phase"], pf["iterations"]))
    rows = [("node step", node, pf["node_steps"], LANES["node"]), ("leaf step", leaf, pf["leaf_steps"], LANES["leaf"]), ("shade/refill phase (one pass; its rejection loop turns ~6 times)", phase, pf["phases"], LANES["phase"])]
    total = 0
    print("%-72s %10s %14s %12s %10s" % ("part", "VALU/step", "steps/frame", "wave-instr", "per ray"))
    for name, v, n, l in rows:
        total += v * n
        print("%-72s %10.0f %14.3g %12.3g %10.1f" % (name, v, n, v * n, v * n / pf["rays"]))
    print("%-72s %10.0f %14.3g %12.3g %10.1f" % ("between steps (fetch addresses, ballots, parking, loop top), per iteration", glue, pf["iterations"], glue * pf["iterations"], glue * pf["iterations"] / pf["rays"]))
    total += glue * pf["iterations"]
    print("%-72s %10s %14s %12.3g %10.1f" % ("sum of STATIC passes", "", "", total, total / pf["rays"]))
    print("# measured (SQ_INSTS_VALU / rays, profiles/r2_b and r3): 158-160 per ray.  The static counts include code a wave skips when none of its lanes needs it (a leaf step's")
    print("# sphere test: ~70 of its ~224; a phase's texture, checker, glass and emissive branches) and count loops once (a phase's rejection loop turns ~6 times at 106 per turn).")
    print()
    print("# ceilings of the levers, in wave-level VALU instructions per ray (of ~160):")
    n_r, l_r = node * pf["node_steps"] / pf["rays"], leaf * pf["leaf_steps"] / pf["rays"]
    print("#  (a) node and leaf steps at 60 of 64 lanes instead of %d / %d: node %.1f -> %.1f, leaf %.1f -> %.1f per ray: the pool kernel reached 61 / 58 lanes and 104 per ray in all, and lost"
          % (LANES["node"], LANES["leaf"], n_r, n_r * LANES["node"] / 60, l_r, l_r * LANES["leaf"] / 60))
    print("#      to latency (profiles/r3_c_pool_kernel_more_waves.txt): the CU's registers + LDS hold ~1 500 paths however they are arranged")
    print("#  (b) optimistic leaves (primitive test first, exact box only to confirm): saves the box test (~25 of a leaf step's %.0f) only for the WAVE, i.e. when none of a step's ~%d lanes has an accepting candidate:" % (leaf, LANES["leaf"]))
    print("#      with 0.15-0.2 candidates per tested leaf that is 0.85^23 = 2 % of the steps -> < 0.1 instructions per ray; not built")
    print("#  (c) per-step overhead (stack, key select, addresses: ~%.0f of a node step's %.0f): 8-wide nodes would halve the node steps but test every child of both levels (+30-50 %% box tests);" % (node - 4 * 22 - 12, node))
    print("#      the kernel answers extra VALU work at its full issue cost (DESIGN 4.5), so 8-wide is a loss on the VALU side and was not built")

if __name__ == "__main__":
    main()
