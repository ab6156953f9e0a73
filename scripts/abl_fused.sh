# ablation builds of the fused backward kernel (scripts/build_variant.sh ablN conv_bwd_fused.hip -DFUSED_ABL=N), timing only
for v in default $@; do
  if [ "$v" = "default" ]; then unset SGNN_LIB; else export SGNN_LIB=$(pwd)/sgnn_amd/lib/variants/libsgnn_hip_$v.so; fi
  echo "== $v"; timeout 120 python scripts/bench_bwd_fused.py --no-parity --iters 30 2>&1 | tail -1
done
