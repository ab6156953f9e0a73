# when do the lane's weight-gradient kernels finish, relative to the training stream's programs?  bad state (a stamp in front
# of the final join) and, as far as stamps on the lane leave it alone, the good one
for only in "join<,bwd<,dw>" "bwd<,dw>,join>"; do echo "== $only"; SGNN_STAMP_ONLY="$only" timeout 300 python scripts/lane_stamps.py --steps 8 --group 8 2>/dev/null | grep -E "^ +[0-9]+ |whole step"; done
