# usage (GPU box, repo root): bash tools/ab_env.sh <rounds> "<ENV=.. ENV=..>" "<ENV=..>" ...  -- interleaved bench runs of ONE build under several
# environments (A/B of a run-time switch): median / min / max ms per step of each run (bench.py --regions 5).  "-" = no extra environment.
R=$1; shift
for r in $(seq 1 $R); do
  for E in "$@"; do
    [ "$E" = "-" ] && EV="" || EV="$E"
    ms=$(env $EV python bench.py --no-also --no-cpu-baseline --no-roofline ${AB_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['ms_per_step_min'], d['config']['ms_per_step_max'])")
    echo "round $r [$E] $ms"
  done
done
