cp flowonthego_amd/libfotg.so /tmp/libfotg_base.so
for lib in /tmp/libfotg_base.so tools/exp/libfotg_fastdata.so /tmp/libfotg_base.so; do
  cp $lib flowonthego_amd/libfotg.so
  echo "== $(basename $lib)"
  timeout 600 python bench.py --no-cpu-baseline --windows 9 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=r['fast_math']; print('parity in flight', round(r['value']), 'one at a time', round(r['one_batch_at_a_time']['value']), '| fast one at a time', round(f['value']), 'in flight', round(f.get('in_flight',{}).get('value',0)), 'epe', f.get('epe_vs_parity_mode_px'))
c=r.get('config_4k_op4_fast_math',{}); print('4k fast', c.get('ms_per_pair'), c.get('epe_vs_parity_mode_px'), ' 4k parity', r.get('config_4k_op4',{}).get('ms_per_pair'))"
done
cp /tmp/libfotg_base.so flowonthego_amd/libfotg.so
