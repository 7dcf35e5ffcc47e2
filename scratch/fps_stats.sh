set -e
mkdir -p /tmp/statslib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -ffp-contract=off -DFPS_STATS -c butd_detr_amd/csrc/fps_pruned.hip -o /tmp/statslib/fps_pruned.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/statslib/lib.so /tmp/statslib/fps_pruned.o butd_detr_amd/lib/obj/pointnet2_ops.o
cp butd_detr_amd/lib/libbutd_detr_hip.so /tmp/keep.so
cp /tmp/statslib/lib.so butd_detr_amd/lib/libbutd_detr_hip.so
timeout 120 python - <<'PY'
import sys; sys.path.insert(0,'.')
import numpy as np, torch
from butd_detr_amd import pointnet2_ext as ext
from butd_detr_amd.synthetic_scenes import scene_batch
pcs = np.ascontiguousarray(scene_batch(2, 1184, 50000)[..., :3])
d = torch.from_numpy(pcs).cuda()
ext.furthest_point_sampling(d, 2048); torch.cuda.synchronize()
PY
cp /tmp/keep.so butd_detr_amd/lib/libbutd_detr_hip.so
