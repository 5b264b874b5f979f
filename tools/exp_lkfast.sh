# fast_math LK variants: staged window radius x lanes per patch (run on the GPU box)
for cfg in "2 0" "0 0" "2 16" "2 4"; do set -- $cfg
  echo "== FOTG_LK_FAST_R=$1 FOTG_LK_LPP=$2"
  FOTG_LK_FAST_R=$1 FOTG_LK_LPP=$2 python tools/lk_fast_probe.py 4k 1080p 2>/dev/null | python -c "
import sys,json,re
t=sys.stdin.read()
for name,blk in re.findall(r'^(\w+) (\{.*?^\})', t, re.S|re.M):
    r=json.loads(blk); print(name, 'fast lk', r['fast']['lk_ms'], 'step', round(r['fast']['ms_per_step'],4), 'epe', '%.2g' % r['epe_fast_vs_exact_fullres']['mean'])
"
done
