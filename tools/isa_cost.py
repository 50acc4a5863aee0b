"""Weighted issue cost of the persistent kernel's regions: every VALU instruction of the ISA between two DR_MARKs priced with the SIMD cycles per
wave-level instruction measured by tools/valu_rate.hip on gfx950 at six waves per SIMD (profiles/r3_l_valu_issue_cost.txt): 2.4 cycles for the
"fast" class (fma / mul / add / sub f32, and / or / xor / bitop3, add / sub u32, mov, lshrrev), 8.2 for rcp / sqrt / rsq, 4.2 for everything else
(min / max / med3, compares, cndmask, conversions, shifts, bfe, perm, and_or, lshl_add, mad24, mul_lo, fma_mix, pk_fma, every f64 operation).
No GPU needed:   python tools/isa_cost.py [extra -D flags] [--list REGION]"""
import os, re, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_or_b32", "v_and_b32", "v_xor_b32", "v_bitop3_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_mov_b32", "v_lshrrev_b32", "v_not_b32", "v_mac_f32", "v_madak_f32", "v_madmk_f32", "v_fmaak_f32", "v_fmamk_f32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_add_co_u32", "v_addc_co_u32"}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_rcp_f64", "v_sqrt_f64", "v_rsq_f64", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32"}
KERNELS = (("_ZN2dr24render_persistent_kernelILb0ELi6ELi32ELi20ELi2ELb1ELb0ELb0EE", "lean (6 waves/SIMD)"), ("_ZN2dr24render_persistent_kernelILb0ELi5ELi32ELi12ELi4ELb1ELb1ELb0EE", "work-sharing (5 waves/SIMD)"))

def cost(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in FAST: return 2.4
    if base in TRANS: return 8.2
    return 4.2

def main():
    args = sys.argv[1:]
    listing = None
    if "--list" in args:
        i = args.index("--list"); listing = args[i + 1]; del args[i:i + 2]
    d = tempfile.mkdtemp(prefix="dr_isa_")
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm", "-enable-post-misched=0", "-mllvm", "-amdgpu-use-amdgpu-trackers=1",
           "-DDR_ISA_MARKS=1"] + args + ["--offload-arch=gfx950", "-c", os.path.join(ROOT, "dogeray_amd", "csrc", "kernels_render.hip"), "-o", os.path.join(d, "r.o"), "-save-temps"]
    subprocess.check_call(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(os.path.join(d, "kernels_render-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    for K, title in KERNELS:
        s = text.index(K); s = text.index(":\n", s); e = text.index(".Lfunc_end", s)
        cur, order, reg, seen = "prologue", [], {}, collections.Counter()
        for line in text[s:e].split("\n"):
            t = line.strip()
            m = re.match(r";\s*DRMARK (\w+)", t)
            if m:
                seen[m.group(1)] += 1
                cur = "%s#%d" % (m.group(1), seen[m.group(1)]); order.append(cur); continue
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
            op = t.split()[0]
            r = reg.setdefault(cur, {"n": 0, "cyc": 0.0, "fast": 0, "slow": 0, "trans": 0, "ops": collections.Counter(), "scratch": 0})
            if op.startswith("v_"):
                c = cost(op); r["n"] += 1; r["cyc"] += c; r["ops"][re.sub(r"_(e32|e64)$", "", op)] += 1
                r["fast" if c < 3 else ("trans" if c > 5 else "slow")] += 1
            elif op.startswith("scratch_"): r["scratch"] += 1
        m = re.search(r"\.name:\s*" + re.escape(K) + r"\S*(?:.*\n)*?.*\.vgpr_count:\s*(\d+)\n.*\.vgpr_spill_count:\s*(\d+)", text)
        print("# %s: %s VGPRs, %s spilled" % (title, m.group(1), m.group(2)))
        print("%-16s %5s %6s %6s %6s %9s" % ("region", "VALU", "fast", "slow", "trans", "cycles"))
        for k in ["prologue"] + order:
            if k in reg:
                r = reg[k]
                print("%-16s %5d %6d %6d %6d %9.0f%s" % (k, r["n"], r["fast"], r["slow"], r["trans"], r["cyc"], "   (%d scratch accesses)" % r["scratch"] if r["scratch"] else ""))
                if listing and k.startswith(listing):
                    print("     " + "  ".join("%s x%d" % (o, c) for o, c in r["ops"].most_common()))

if __name__ == "__main__":
    main()
