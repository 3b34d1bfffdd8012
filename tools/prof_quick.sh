# usage (on the GPU box, from the repo root): bash tools/prof_quick.sh <tag> [bench args...]  ->  gpurun_out/<tag>_kernel_stats.txt
set -x
R=$PWD; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq
rocprofv3 --kernel-trace --stats -d /tmp/pq -o p -- python $R/bench.py --steps 10 --warmup 3 --settle 0 --regions 1 --no-cpu-baseline --no-roofline --no-also "$@" > /dev/null 2>&1
python $R/tools/prof_summary.py $(find /tmp/pq -name "*.db" | head -1) 13 > $R/gpurun_out/${TAG}_kernel_stats.txt
cd $R
