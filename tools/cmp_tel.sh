cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for t in 2.5 0; do
    ms=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --telemetry-s $t 2>/dev/null | tail -1 | grep -o 'avg_step_ms": [0-9.]*' | head -1 | cut -d' ' -f2)
    echo "rep $rep telemetry $t: $ms"
  done
  ms=$(cd tools/ab/r03 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | grep -o 'avg_step_ms": [0-9.]*' | head -1 | cut -d' ' -f2)
  echo "rep $rep r03: $ms"
done
