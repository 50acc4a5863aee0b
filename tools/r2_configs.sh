#!/bin/bash
# Rates of the other BASELINE configs (stand-ins per SURVEY 8(d)) on the GPU box: C2 bunnyish, C3-like matball, C5 city
mkdir -p /tmp/cfg/tex
tools/scenegen bunnyish /tmp/cfg/bunny.rts 6 1280 720
tools/scenegen city /tmp/cfg/city.rts 200 3840 2160
tools/scenegen matball /tmp/cfg/matball.rts 1920 1080
tools/scenegen ppm /tmp/cfg/tex/synth_albedo.ppm 128 128 0; tools/scenegen ppm /tmp/cfg/tex/synth_rough.ppm 64 64 1; tools/scenegen ppm /tmp/cfg/tex/synth_env.ppm 256 128 2
for t in 2 0; do
  echo "== traversal $t"
  DOGERAY_TRAV=$t python3 tools/exp_scene_rate.py /tmp/cfg/bunny.rts "" 32
  DOGERAY_TRAV=$t python3 tools/exp_scene_rate.py /tmp/cfg/matball.rts /tmp/cfg/tex 32
  DOGERAY_TRAV=$t python3 tools/exp_scene_rate.py /tmp/cfg/city.rts "" 8
done 2>&1 | grep -v amdgpu.ids
