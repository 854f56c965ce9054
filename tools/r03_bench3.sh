#!/bin/bash
mkdir -p gpurun_out/r03z
for i in 1 2 3; do timeout 600 python bench.py --no_cpu_baseline --no_configs45 > gpurun_out/r03z/b$i.json 2> gpurun_out/r03z/b$i.err; done
timeout 600 python bench.py --full --no_cpu_baseline --no_configs45 > gpurun_out/r03z/f1.json 2> gpurun_out/r03z/f1.err
for f in b1 b2 b3 f1; do python3 - $f <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/r03z/'+sys.argv[1]+'.json') if l.startswith('{')][-1])
print(sys.argv[1], d['ms_per_step'], {k:v for k,v in d['config']['stream'].items() if k.startswith('ms_')})
PY
done
