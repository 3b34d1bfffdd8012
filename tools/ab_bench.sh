# usage (GPU box, repo root): bash tools/ab_bench.sh <rounds> <lib1.so> <lib2.so> ...  -- interleaved bench runs of several builds of the library
# (SAR_LIB_PATH) on ONE box: median ms per step of each run (bench.py --regions 5), run to run.  Extra environment passes through.
R=$1; shift
for r in $(seq 1 $R); do
  for L in "$@"; do
    ms=$(SAR_LIB_PATH=$PWD/$L python bench.py --no-also --no-cpu-baseline --no-roofline ${AB_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['ms_per_step_min'], d['config']['ms_per_step_max'])")
    echo "round $r $L $ms"
  done
done
