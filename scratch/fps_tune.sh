timeout 200 python scratch/fps_time.py 8
for cfg in 256,8 512,4 1024,2; do echo "cfg $cfg n=2048"; BUTD_FPS_CFG=$cfg timeout 100 python scratch/fps_time.py 8 2>&1 | grep "FPS 2048"; done
for cfg in 256,4 512,2 1024,1; do echo "cfg $cfg n=1024"; BUTD_FPS_CFG=$cfg timeout 100 python scratch/fps_time.py 8 2>&1 | grep "FPS 1024"; done
for cfg in 256,2 512,1; do echo "cfg $cfg n=512"; BUTD_FPS_CFG=$cfg timeout 100 python scratch/fps_time.py 8 2>&1 | grep "FPS 512"; done
# pruned with 4 / 16 waves
for W in 4 16; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -ffp-contract=off -DFPS_LOOP_WAVES=$W -c butd_detr_amd/csrc/fps_pruned.hip -o /tmp/fps_pruned_$W.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$W.so /tmp/fps_pruned_$W.o butd_detr_amd/lib/obj/pointnet2_ops.o
cp butd_detr_amd/lib/libbutd_detr_hip.so /tmp/keep.so; cp /tmp/lib_$W.so butd_detr_amd/lib/libbutd_detr_hip.so
echo "pruned waves=$W"; timeout 100 python scratch/fps_time.py 8 2>&1 | grep pruned
cp /tmp/keep.so butd_detr_amd/lib/libbutd_detr_hip.so
done
timeout 300 python -m pytest tests/test_gpu_pointnet2_parity.py -x -q -m gpu 2>&1 | tail -3
