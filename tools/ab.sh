#!/bin/bash
# A/B option sets on the GPU box: tools/ab.sh "kernel=0 kernel=1,park_min=0" [bench args]
VARS=$1; shift
mkdir -p gpurun_out
for v in $VARS; do
  DOGERAY_OPTIONS=$v python3 bench.py --steps ${STEPS:-16} --warmup ${WARM:-2} --no-cpu-baseline --no-traffic --no-extras --repeats ${REPEATS:-3} "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -5 gpurun_out/ab_$v.err; exit 1; }
  python3 - <<PY
import json
j=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
print("options $v: %.1f Mrays/s  kernel %.4f ms/frame (wall %.4f, min %.4f max %.4f)  clock %.0f MHz" % (j["value"], j["kernel_ms_per_frame"], j["ms_per_step"], j["ms_per_step_min"], j["ms_per_step_max"], j["timed_waves"]["shader_clock_mhz"]))
d=j.get("diag")
if d and d[2]:
    fr=j["steps"]
    print("   diag/frame: wave-cycles %.3g  phase-cycles %.3g (%.1f%%)  iters %.3g  phases %.3g  cycles/iter %.0f  cycles/phase %.0f" % (d[0]/fr, d[1]/fr, 100.0*d[1]/max(1,d[0]), d[2]/fr, d[3]/fr, (d[0]-d[1])/max(1,d[2]), d[1]/max(1,d[3])))
    if len(d) > 6 and d[4]:
        if len(d) > 7 and d[7]: print("   shader clock while the counting build ran: %.0f MHz" % (100.0 * d[0] / d[7]))
        print("   wave-level steps/frame: node %.3g  leaf %.3g;  lanes shaded per phase %.1f;  records/ray %.2f  prim tests/ray %.2f" % (d[4]/fr, d[5]/fr, d[6]/max(1,d[3]), j["per_ray"]["kernel"]["V"], j["per_ray"]["kernel"]["L"]))
PY
done
