# same-box A/B of library variants on the working tree: scripts/ab_lib.sh name1 name2 ... ("default" = the regular build); two rounds
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = "default" ]; then unset SGNN_LIB; else export SGNN_LIB=$(pwd)/sgnn_amd/lib/variants/libsgnn_hip_$v.so; fi
  timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-traffic --no-other-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d.get('launches_per_step'))"
done
done
