for BK in 16 32 64; do
sed "s/kBK = 64, kLd/kBK = $BK, kLd/" butd_detr_amd/csrc/attention_ops.hip > /tmp/attention_ops_bk.hip
cp /tmp/attention_ops_bk.hip butd_detr_amd/csrc/_tmp_bk.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -c butd_detr_amd/csrc/_tmp_bk.hip -o /tmp/attn_bk.o
rm butd_detr_amd/csrc/_tmp_bk.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_bk.so /tmp/attn_bk.o butd_detr_amd/lib/obj/pointnet2_ops.o butd_detr_amd/lib/obj/fps_pruned.o
cp butd_detr_amd/lib/libbutd_detr_hip.so /tmp/keep.so; cp /tmp/lib_bk.so butd_detr_amd/lib/libbutd_detr_hip.so
echo "BK=$BK"; timeout 100 python scratch/gemm_bench.py
cp /tmp/keep.so butd_detr_amd/lib/libbutd_detr_hip.so
done
