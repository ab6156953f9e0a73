# Which stamp disturbs the replayed step?  One run of scripts/lane_stamps.py per label subset (SGNN_STAMP_ONLY), printing the
# stamps and ms per step.  usage: scripts/stamp_subsets.sh "join<" "join>" "bwd<,join>" ...   (no arguments: the round-6 list)
if [ $# -eq 0 ]; then
  set -- "join<" "join>" "bwd<" "bwd>" "bwd<,bwd>" "lane-end" "fork" "lane<,lane>" "prog<,prog>" "down-wait<,down-wait>" \
         "bwd<,join>" "prog<,bwd<,join>" "bwd<,lane-end,join<"
fi
for only in "$@"; do
  echo "== $only"
  SGNN_STAMP_ONLY="$only" timeout 300 python scripts/lane_stamps.py --steps 10 --group 8 2>/dev/null | grep -E "^ +[0-9]+ |whole step|join costs"
done
