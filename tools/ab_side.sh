for c in C2 C3; do
for s in 1 0; do
  for r in 1 2; do
  ms=$(S2T_WGRAD_STREAM=$s python bench.py --config $c --steps 15 --warmup 3 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys;print(round(json.loads(sys.stdin.read())['ms_per_step'],2))")
  echo "$c S2T_WGRAD_STREAM=$s  $ms ms/step"
  done
done
done
