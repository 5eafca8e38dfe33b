#!/bin/bash
# round 6: what bounds an update tile?  The single-precision tile (wg_tile_dma_f32) of a -DTMPC_ABLATE build with one ingredient removed at a time,
# isolated factorisation bench with every update in single precision; kernel times from rocprofv3 --kernel-trace.
# build first: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DTMPC_ABLATE tunempc_amd/csrc/tmpc_api.hip -o tunempc_amd/lib/libtunempc_hip_ablate.so
mkdir -p gpurun_out /tmp/prof; export TMPDIR=/tmp
export TMPC_LIB=$PWD/tunempc_amd/lib/libtunempc_hip_ablate.so
for V in 0 2 3 4 5 6; do
  rm -rf /tmp/prof/abl
  FB_LOWP_VALUE=$V timeout 200 rocprofv3 --kernel-trace -d /tmp/prof/abl -o kt -- python3 scripts/factor_bench.py 433,64,300 > /tmp/prof/abl.out 2> /tmp/prof/abl.err < /dev/null
  case $V in 0) N="fp64 tile (reference)";; 2) N="fp32 tile";; 3) N="fp32 tile, no C load/store";; 4) N="fp32 tile, no MFMA";; 5) N="fp32 tile, no slab DMA";; 6) N="fp32 tile, no barrier";; esac
  echo "## $N"; tail -1 /tmp/prof/abl.out
  python3 scripts/rocpd_stats.py $(find /tmp/prof/abl -name '*.db' | head -1) 2>&1 < /dev/null | grep -E "k_cr_update_dma|k_cr_trsm_dma" | cut -c1-130
done
