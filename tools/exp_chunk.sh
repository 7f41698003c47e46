set -e
mkdir -p gpurun_out
for L in 44110 44111 44112 44120 44130; do
for v in "PAL_FIN=1" "PAL_FIN=0" ; do
  echo "L=$L $v $(env $v timeout -k 10 120 python tools/length_sweep.py $L 1 4 2>/dev/null | tail -1)"
done; done | tee gpurun_out/chunk_exp.txt
