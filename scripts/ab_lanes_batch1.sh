for v in "" "SGNN_SIDE_LANE=0 SGNN_SIDE_PYRAMID=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "SGNN_SIDE_PYRAMID=0" "SGNN_SIDE_LANE=0"; do
  env $v timeout -k 10 400 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['other_legs']['batch1']; s=b['stats']
print('[$v]', 'bs32', d['ms_per_step'], 'host', d.get('host_graph_launch_ms'), '| batch1', b['ms_per_step'], 'host', round(s['replay_host_ms']/max(s['replays'],1),3))"
done
