cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  ( sleep 25; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|fclk\|mclk\|Power (W)\|Socket Power\|junction\|hotspot" | tr '\n' ';' ; echo ) &
  python bench.py --no-secondary --no-cpu-baseline --no-traffic --steps 400 --warmup 5 --detail '' 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('run', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms_per_step'))"
  wait
done
