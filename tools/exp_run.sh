#!/bin/bash
# timing experiments: run bench with each tools/exp/libfotg_*.so variant swapped in (scratch copy on the GPU box only)
cp flowonthego_amd/libfotg.so /tmp/libfotg_base.so
for lib in /tmp/libfotg_base.so tools/exp/libfotg_*.so; do
  cp $lib flowonthego_amd/libfotg.so
  for v in "$@"; do
    echo -n "$(basename $lib) $v: "
    env $v python bench.py --no-cpu-baseline --steps 10 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d[\"value\"]), [d[\"stage_ms\"][k] for k in (\"varref[6]\",\"varref[5]\",\"varref[4]\")])"
  done
done
cp /tmp/libfotg_base.so flowonthego_amd/libfotg.so
