for cfg in "111 21 corner" "111 0 xyz" "110 0 xy" "100 0 x"; do
  set -- $cfg
  for k in 16 24 32; do
    FPR_DROP_FACES=$2 timeout -k 10 120 python3 tools/exp_overlap.py 512 $1 40 diff3_comm_units=$k 2>&1 | grep "fused pair" | sed "s/^/$3 k=$k: /"
  done
done
