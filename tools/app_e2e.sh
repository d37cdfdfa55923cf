#!/bin/bash
# End-to-end run of apps/Diffusion on a 1024^3 16-bit RAW volume (the bench structure as grey values), direction = All:
# what a user of the reference's `Diffusion inputs` sees -- read + threshold, upload, three masks, three solves, results file.
# usage (GPU box): bash tools/app_e2e.sh [size] [flow_through|homogenization]
set -e
n=${1:-1024}
method=${2:-flow_through}
cd "$(dirname "$0")/.."
python3 - <<PY
import numpy as np, sys, time
sys.path.insert(0, ".")
import openimpala_amd as oia
n = $n
buf = oia.gen_spheres_device((n, n, n), 20251031, 24, 0.4)
ph = buf.download(np.uint8, (n, n, n)); buf.free()
rng = np.random.default_rng(1)
grey = (ph.astype(np.uint16) * 30000 + rng.integers(0, 20000, size=ph.shape, dtype=np.uint16))   # pore = bright, threshold 25000
grey.tofile("/tmp/vol_u16.raw")
print("wrote /tmp/vol_u16.raw", grey.nbytes / 1e9, "GB, porosity", ph.mean())
PY
for rep in 1 2; do
  ./apps/Diffusion apps/inputs filename=vol_u16.raw data_path=/tmp raw.width=$n raw.height=$n raw.depth=$n \
      raw.data_type=UINT16_LE threshold_val=25000 phase_id=1 direction=All calculation_method=$method hypre.eps=1e-9 results_path=/tmp/results_e2e/ verbose=1 2>&1 || true
done
rm -f /tmp/vol_u16.raw
