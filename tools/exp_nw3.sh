for L in 44110 44125 44157; do
  echo "== check L=$L: $(timeout -k 10 300 python tools/check_fin.py 8 2 $L 2>&1 | tail -1)"
  for v in "PAL_FIN=1" "PAL_FIN=0"; do
    echo "L=$L $v $(env $v timeout -k 10 120 python tools/length_sweep.py $L 1 4 2>/dev/null | tail -1)"
  done
done
