#!/bin/bash
# experiment: base pyramid as a fixed number of walking workgroups (FOTG_PIPE_PYR_WALK / FOTG_PYR_WALK)
export GPU_MAX_HW_QUEUES=6
python - <<'PY'
import numpy as np, torch, flowonthego_amd as F
# parity of the walking kernel against the per-tile launch (both gfx950): identical bits
import os
res = {}
for walk in (0, 768):
    os.environ["FOTG_PYR_WALK"] = str(walk)
    op = F.operating_point(2, 1920, 1)
    eng = F.OFClass(op, F.img_params(width=1920, height=1080, padding=op.patch_size), max_batch=8, device=0)
    g = torch.Generator(device="cpu").manual_seed(1)
    a = torch.rand((8, 1080, 1920), generator=g).mul(255).cuda(); b = torch.rand((8, 1080, 1920), generator=g).mul(255).cuda()
    res[walk] = eng.calc_batch(a, b).cpu().numpy()
print("walk parity:", np.array_equal(res[0], res[768]))
PY
for w in 0 512 768 1024 1536 2048; do
  echo -n "pipe walk $w: "
  FOTG_PIPE_PYR_WALK=$w python bench.py --no-cpu-baseline --windows 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['one_batch_at_a_time']['value']), d['stage_ms']['pyramid(I0,I1)'])"
done
for w in 512 768 1024 2048; do
  echo -n "alone walk $w: "
  FOTG_PYR_WALK=$w FOTG_PIPE_PYR_WALK=$w python bench.py --no-cpu-baseline --windows 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['one_batch_at_a_time']['value']), d['stage_ms']['pyramid(I0,I1)'])"
done
