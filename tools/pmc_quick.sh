R=$PWD; cd /tmp && export TMPDIR=/tmp
python -m pytest $R/tests/test_gpu_ops.py -m gpu -q -k "conv or wgrad" 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p -- python $R/bench.py --steps 10 --warmup 3 --settle 0 --regions 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) 13 | grep -i "summary\|wgrad"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p3 -o p -- python $R/bench.py --steps 4 --warmup 3 --settle 0 --regions 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find /tmp/p3 -name "*.db" | head -1) stem | grep -A1 "wgrad"
