# usage (GPU box, repo root): bash tools/pmc_conv.sh <tag>  ->  gpurun_out/<tag>_pmc.txt : per-kernel averages of FETCH_SIZE, WRITE_SIZE, L2 hit / miss
# counters (separate rocprofv3 --pmc passes, kernel-trace only)
R=$PWD; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_pmc.txt; : > $OUT
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"; do
  rm -rf /tmp/pp
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pp -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-also "$@" > /dev/null 2>&1
  echo "## $C" >> $OUT
  python $R/tools/pmc_summary.py $(find /tmp/pp -name "*.db" | head -1) wgrad >> $OUT 2>&1
done
cd $R
