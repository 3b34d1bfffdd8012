# usage (GPU box, repo root): bash tools/refresh_profiles.sh <tag e.g. r2_v16>  ->  gpurun_out/<tag>/ : the default bench line, kernel stats of
# cfg2 / cfg3 and the FETCH_SIZE / WRITE_SIZE passes of the same bench command (separate rocprofv3 --pmc runs, kernel-trace only).  Copy
# what should be judged into profiles/ and run tools/pmc_to_traffic.py on the pmc file.
set -x
R=$PWD; TAG=${1:-r2}; O=$R/gpurun_out/$TAG; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3 /tmp/p4
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p -- python $R/bench.py --steps 10 --warmup 3 --settle 0 --regions 1 --no-cpu-baseline --no-roofline --no-also > /dev/null 2>&1
python $R/tools/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) 15 > $O/kernel_stats.txt
rocprofv3 --kernel-trace --stats -d /tmp/p2 -o p -- python $R/bench.py --workload cfg3 --steps 10 --warmup 3 --settle 0 --regions 1 --no-cpu-baseline --no-roofline --no-also > /dev/null 2>&1
python $R/tools/prof_summary.py $(find /tmp/p2 -name "*.db" | head -1) 15 > $O/kernel_stats_cfg3.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p3 -o p -- python $R/bench.py --steps 4 --warmup 3 --settle 0 --regions 1 --no-cpu-baseline --no-roofline --no-also > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p4 -o p -- python $R/bench.py --steps 4 --warmup 3 --settle 0 --regions 1 --no-cpu-baseline --no-roofline --no-also > /dev/null 2>&1
( echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only), python bench.py --steps 4 --warmup 3; per-dispatch averages, raw counter units (KB)."
  echo "# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of wide coalesced reads -> traffic = 2*FETCH_SIZE + WRITE_SIZE."
  echo "## FETCH_SIZE"; python $R/tools/pmc_summary.py $(find /tmp/p3 -name "*.db" | head -1) stem
  echo "## WRITE_SIZE"; python $R/tools/pmc_summary.py $(find /tmp/p4 -name "*.db" | head -1) stem ) > $O/pmc_hbm.txt
cd $R
head -c 700 $O/bench.json; echo; tail -2 $O/bench.err; head -5 $O/kernel_stats.txt; wc -l $O/pmc_hbm.txt
