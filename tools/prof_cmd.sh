# usage (on the GPU box, from the repo root): bash tools/prof_cmd.sh <tag> <steps> <python script + args...>  ->  gpurun_out/<tag>_kernel_stats.txt
R=$PWD; TAG=$1; STEPS=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq
rocprofv3 --kernel-trace --stats -d /tmp/pq -o p -- python "$@" > /tmp/pq_out.txt 2>&1
mkdir -p $R/gpurun_out
python $R/tools/prof_summary.py $(find /tmp/pq -name "*.db" | head -1) $STEPS > $R/gpurun_out/${TAG}_kernel_stats.txt
cd $R
