# usage: scripts/bench_ab.sh <lib.so>...   (same-box A/B of bench.py: ms/step and the MSDA event times, twice each, interleaved)
for rep in 1 2; do for lib in "$@"; do
  ZIRA_MSDA_LIB=$PWD/$lib python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);k=d['roofline']['kernels'];print('$lib','%.2f ms/step'%d['ms_per_step'],' '.join('%s %.1f'%(n,v['avg_us']) for n,v in k.items()))"
done; done
