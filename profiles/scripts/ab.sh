for i in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export ADN_NO_GROUPED_BACKWARD=1; else unset ADN_NO_GROUPED_BACKWARD; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile --accurate-precision none 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_grouped_backward=$v', d['ms_per_step'])"
  done
done
