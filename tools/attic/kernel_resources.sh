#!/bin/bash
# Dev tool: per-kernel VGPR / AGPR / spill / scratch / occupancy of the HIP sources (hipcc -Rpass-analysis=kernel-resource-usage).
#   tools/kernel_resources.sh [file.hip ...]     (default: the conv kernels)
cd "$(dirname "$0")/../retrieval-fuse_amd/csrc"
FILES=${@:-conv3d_mfma.hip conv3d_up.hip conv3d_small.hip conv_valid_mfma.hip retrieval.hip}
for f in $FILES; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys, subprocess
cur = {}
def flush():
    if cur:
        name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        print("%-70s VGPR %3s AGPR %3s spill %3s scratch %4s occ %s LDS %s" % (name[:70], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("VGPRs Spill"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
for line in sys.stdin:
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?):\s*(\S+) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?):\s*(\S+) \[-Rpass", line) or re.search(r":\d+:\d+:\s+(.*?):\s*(\S+) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k in ("Function Name", "Name"):
        flush(); cur = {"name": v}
    else:
        cur[k] = v
flush()
'
done
