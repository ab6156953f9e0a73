# the +0.9 ms state (one stamp between the last fork and the final join) under runtime switches: which one dissolves it?
for v in "" "DEBUG_HIP_GRAPH_BATCH_SIZE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=1024" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=3" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "GPU_MAX_HW_QUEUES=8" "DEBUG_HIP_DYNAMIC_QUEUES=0"; do
  echo -n "[$v] "; env $v SGNN_STAMP_ONLY="join<" timeout 200 python scripts/lane_stamps.py --steps 6 --group 8 2>/dev/null | grep "whole step" | sed 's/.*; //'
done
