#!/bin/bash
# per-kernel register / scratch / occupancy table of libfotg.so (hipcc -Rpass-analysis=kernel-resource-usage)
cd "$(dirname "$0")/../flowonthego_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -c fotg_capi.hip -o /tmp/fotg_res.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur={"name":subprocess.run(["c++filt",m.group(1)],capture_output=True,text=True).stdout.strip()[:60]}; rows.append(cur); continue
    for k in ("VGPRs","AGPRs","ScratchSize [bytes/lane]","Occupancy [waves/SIMD]","LDS Size [bytes/block]","TotalSGPRs"):
        m=re.search(re.escape(k)+r": (\d+)",l)
        if m and cur is not None: cur[k]=m.group(1)
print("%-60s %5s %5s %7s %4s %6s"%("kernel","VGPR","SGPR","scratch","occ","LDS"))
for r in rows: print("%-60s %5s %5s %7s %4s %6s"%(r["name"],r.get("VGPRs"),r.get("TotalSGPRs"),r.get("ScratchSize [bytes/lane]"),r.get("Occupancy [waves/SIMD]"),r.get("LDS Size [bytes/block]")))
'
