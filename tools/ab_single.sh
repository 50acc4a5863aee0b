for r in 1 2 3; do for v in default exclcoop; do
  if [ "$v" = default ]; then unset DOGERAY_AMD_LIB; else export DOGERAY_AMD_LIB=$PWD/tools/_exp/lib_$v.so; fi
  echo "== $v"; python3 tools/exp_single.py "" 2>&1 | grep -v amdgpu; FRAMES=48 python3 tools/exp_pipeline.py "" 2>&1 | grep -v amdgpu
done; done
