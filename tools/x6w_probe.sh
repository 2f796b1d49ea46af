#!/bin/bash
# the timing probes of vel_x6w.hip, one build each (GPU box): tools/x6w_probe.sh > gpurun_out/x6w_probe.txt     (VARIANTS overrides the list)
cd "$(dirname "$0")/.."
if [ -z "$VARIANTS" ]; then VARIANTS='|-DX6W_PROBE_NO_SILU|-DX6W_PROBE_NO_SPLIT|-DX6W_PROBE_NO_SILU -DX6W_PROBE_NO_SPLIT|-DX6W_PROBE_NO_ENC|-DX6W_PROBE_NO_OUT|-DX6W_PROBE_SAME_ENTRIES|-DX6W_PROBE_QUARTER_LOADS|-DX6W_PROBE_LDS_A|-DX6W_PROBE_LDS_A -DX6W_PROBE_NO_SILU -DX6W_PROBE_NO_SPLIT -DX6W_PROBE_NO_OUT -DX6W_PROBE_NO_ENC'; fi
IFS='|' read -ra LIST <<< "$VARIANTS"
for fl in "${LIST[@]}"; do
  touch nvfi_amd/csrc/vel_x6w.hip
  NVFI_EXTRA_FLAGS="$fl" python -m nvfi_amd.build > /dev/null 2>gpurun_out/x6w_probe_build.err || { echo "build failed: $fl"; tail -5 gpurun_out/x6w_probe_build.err; continue; }
  echo "== flags: [$fl]"
  python tools/x6w_time.py 2>&1 | tail -1
done
touch nvfi_amd/csrc/vel_x6w.hip; python -m nvfi_amd.build > /dev/null 2>&1
