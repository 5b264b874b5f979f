# in-flight rate against the pyramid's share of the chip (persistent launch of N workgroups): tools/exp_persist.sh
for cfg in "16 0" "1 1" "1 128" "1 192" "1 384"; do set -- $cfg
  v=$(FOTG_PIPE_PYR_SPLIT=$1 FOTG_PIPE_PYR_PERSIST=$2 python bench.py --no-cpu-baseline --no-breakdown --windows 9 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value']), round(r['ms_per_step'],4), round(r['one_batch_at_a_time']['value']))")
  echo "split $1 persist $2: $v"
  v=$(FOTG_PYR_PERSIST=$2 python bench.py --no-cpu-baseline --no-breakdown --windows 3 --in-flight 1 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value']))")
  echo "   one at a time with the same pyramid: $v"
done
