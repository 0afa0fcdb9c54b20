#!/bin/bash
# Round 3 profile collection on the GPU box (inside gpurun): mesh PMC passes (RGB + spectral), the kernel trace of `bench.py --config mesh`,
# the headline kernel trace + PMC passes.  Summaries: scripts/summarize_mesh_pmc.py r03 [spectral], scripts/summarize_profiles.py r03.
set -o pipefail
bash scripts/collect_mesh_profile.sh r03 pmc bench || exit 1
VARIANT=spectral bash scripts/collect_mesh_profile.sh r03 pmc || exit 1
bash scripts/collect_profiles.sh r03 || exit 1
echo r03 profiles collected
