mkdir -p gpurun_out/dot && cd gpurun_out/dot
DEBUG_HIP_GRAPH_DOT_PRINT=1 AMD_LOG_LEVEL=3 timeout 300 python ../../bench.py --steps 2 --warmup 1 --settle 20 --no-cpu-baseline --no-traffic --no-other-mode > out.txt 2> log.txt
ls -la . /tmp/*.dot 2>/dev/null | head -20
grep -c "" log.txt
grep -i "hipGraph\]" log.txt | grep -v "capture node\|Add \|Root node" | sort | uniq -c | sort -rn | head -20
grep -i "hipGraph\]" log.txt | grep -v "capture node" | head -5
