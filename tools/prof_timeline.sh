# usage (GPU box, repo root): bash tools/prof_timeline.sh <tag> <launches per step> [bench args...]  ->  gpurun_out/<tag>_timeline.txt
R=$PWD; TAG=$1; N=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt
rocprofv3 --kernel-trace -d /tmp/pt -o p -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-also "$@" > /dev/null 2>&1
python $R/tools/step_timeline.py $(find /tmp/pt -name "*.db" | head -1) $N > $R/gpurun_out/${TAG}_timeline.txt
cd $R
