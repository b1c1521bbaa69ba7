#!/bin/bash
# bounce_groups (cap of the bounce kernels' grid; 0 = one workgroup per tile) sweep per configuration: tools/exp/r03_grid_sweep.sh
for C in C3 C4 C5; do
  for g in 0 1024 2048 4096 8192 16384; do python bench.py --config $C --no-cpu-baseline --no-parity --steps 60 --warmup 8 --option bounce_groups=$g 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$C', j['config'].get('options'), j['ms_per_step'], {k.split('(')[0]: round(v['ms'],4) for k,v in j['roofline']['kernels'].items()})"; done
done
python bench.py --config C3 --no-cpu-baseline --no-parity --steps 60 --warmup 8 --option bounce_split=1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('C3 split', j['ms_per_step'], {k.split('(')[0]: round(v['ms'],4) for k,v in j['roofline']['kernels'].items()})"
