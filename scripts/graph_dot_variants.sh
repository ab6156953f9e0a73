# graph dumps (DEBUG_HIP_GRAPH_DOT_PRINT) of the captured step with different stamp subsets: gpurun_out/dot_<tag>/
i=0
for only in "$@"; do
  i=$((i+1)); d=gpurun_out/dotv_$i; mkdir -p $d; ( cd $d; echo "$only" > only.txt
  SGNN_STAMP_ONLY="$only" DEBUG_HIP_GRAPH_DOT_PRINT=1 timeout 300 python ../../scripts/lane_stamps.py --settle 20 --steps 4 --group 4 2>/dev/null | grep "whole step" )
done
