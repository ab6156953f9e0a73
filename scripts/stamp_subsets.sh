for only in "join<,join>" "lane-end" "bwd<,bwd>" "bwd<" "fork,lane<,lane>,prog<,prog>,down-wait<,down-wait>" "prog<,prog>" "fork" "lane<,lane>" "down-wait<,down-wait>" "none"; do
  SGNN_STAMP_ONLY="$only" timeout 300 python scripts/lane_stamps.py --steps 10 --group 8 2>/dev/null | grep -E "^# (whole step|[0-9]+ stamps)" | tr '\n' ' '; echo " [$only]"
done
