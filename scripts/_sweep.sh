for k in 1 2 3 4 2 4; do
  MTSAMD_STREAMS=$k timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/b.log 2>&1
  echo "$k $(tail -n 1 gpurun_out/b.log | cut -c100-150)"
done
