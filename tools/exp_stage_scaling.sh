#!/bin/bash
# how the stage times of ONE batch scale with the batch size (throughput-bound stages scale, latency-bound ones do not)
for b in 16 32 64 128 256; do
  echo -n "batch $b: "; python bench.py --no-cpu-baseline --batch $b --in-flight 1 --steps 30 --windows 5 2>/dev/null | python -c "import sys,json; json.loads(sys.stdin.read().strip().splitlines()[-1]); d=json.load(open('gpurun_out/bench_detail.json')); print(round(d['value']), d['stage_ms'])"
done
