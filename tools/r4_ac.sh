#!/bin/bash
# wide_tree 2 against 1 on the other configurations, then a fuzz campaign (wide_tree drawn from 2 / 2 / 2 / 1 / 0 per scene)
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r4ac_configs.txt
for c in C2 C3 C5; do
  for o in "wide_tree=2" "wide_tree=1"; do
    DOGERAY_OPTIONS=$o timeout -k 10 400 python3 bench.py --config $c --steps 16 --warmup 4 --repeats 5 --no-cpu-baseline --no-traffic --no-extras 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$c $o', round(j['kernel_ms_per_frame'],4), round(j['value'],1), 'records/ray %.2f' % j['per_ray']['kernel']['V'])" >> gpurun_out/r4ac_configs.txt || exit 1
  done
done
cat gpurun_out/r4ac_configs.txt
timeout -k 10 900 python3 tools/fuzz_campaign.py ${FUZZ_SCENES:-1500} 70007 > gpurun_out/r4ac_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r4ac_fuzz.txt
