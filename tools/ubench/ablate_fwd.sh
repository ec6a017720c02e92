set -e
cd $GRAFT_REPO_ROOT
for A in 0 1 2; do
  cp coivo_amd/csrc/warp_loss_ablate.hip.txt /tmp/wl_abl.hip
  sed -i 's#"common.h"#"'$GRAFT_REPO_ROOT'/coivo_amd/csrc/common.h"#' /tmp/wl_abl.hip
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -w -DABLATE=$A -c /tmp/wl_abl.hip -o coivo_amd/lib/obj/warp_loss.o
  hipcc -shared -fPIC --offload-arch=gfx950 -o coivo_amd/lib/libcolvo.so coivo_amd/lib/obj/*.o
  echo "ABLATE=$A"; python tools/bench_loss.py 32 512 640
done
