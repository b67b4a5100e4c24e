#!/bin/bash
# fp64 matrix pipe: what it sustains (VGPR accumulators; -DACC... see the source) and whether vector work hides under it
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 mfma_f64_peak.hip -o /tmp/mfma_f64_peak 2>/dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -DAGPR_FORM mfma_f64_peak.hip -o /tmp/mfma_f64_peak_agpr 2>/dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o /tmp/mfma_valu_overlap 2>/dev/null
echo "== accumulators in architectural VGPRs (launch bounds 256, 2)"; /tmp/mfma_f64_peak
echo "== accumulators in AGPRs (launch bounds 256: the compiler's choice with 512 registers available)"; /tmp/mfma_f64_peak_agpr
echo "== matrix / vector overlap"; /tmp/mfma_valu_overlap
