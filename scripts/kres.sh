#!/bin/bash
# Per-kernel register / scratch / LDS report of one HIP source (compile-time, no GPU): scripts/kres.sh csrc/kernels_conv.hip
cd "$(dirname "$0")/../fast-artistic-videos_amd"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt ${KRES_FLAGS} -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 | python3 -c '
import re, sys
for l in sys.stdin:
    m = re.search(r"remark: +(.*?) \[-Rpass", l)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        print(); print(t.split(":", 1)[1].strip()[:120], end=" | ")
    elif any(t.startswith(k) for k in ("TotalSGPRs", "VGPRs:", "AGPRs", "ScratchSize", "Occupancy", "LDS Size")):
        print(t.replace(" [bytes/lane]", "").replace(" [bytes/block]", "").replace(" [waves/SIMD]", ""), end="; ")
print()' | c++filt
