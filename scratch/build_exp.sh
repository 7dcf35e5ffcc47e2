# build ablation variants of the library into scratch/exp/
set -e
cd /root/repo
OBJ=butd_detr_amd/lib/obj
for v in BASE NOMFMA NOLOAD NOCOMMIT NOBAR; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DGEMM_EXP_$v -c butd_detr_amd/csrc/attention_ops.hip -o scratch/exp/attn_$v.o &
done
wait
for v in BASE NOMFMA NOLOAD NOCOMMIT NOBAR; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/exp/lib_$v.so scratch/exp/attn_$v.o $OBJ/pointnet2_ops.o $OBJ/fps_pruned.o $OBJ/sa_ops.o $OBJ/optim_ops.o $OBJ/mlp_ops.o
done
ls -la scratch/exp/*.so
