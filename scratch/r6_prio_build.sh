# builds scratch/exp/libprio<N><tag>.so: the main-queue units with -DBUTD_MAIN_PRIO=<N>; UNITS="a b c" selects them
set -e
cd /root/repo; mkdir -p scratch/exp
N=${1:-3}; TAG=${2:-}
UNITS=${UNITS:-"gemm_ops attention_ops mlp_ops sa_ops sa_last_bwd sa_fused sa_first_linear criterion_ops lsap_ops optim_ops"}
EXCL=$(echo $UNITS | sed 's/ /.o\\|\//g'); OBJS=$(ls butd_detr_amd/lib/obj/*.o | grep -v "/$EXCL.o")
NEW=""
for U in $UNITS; do
  X=""; case $U in gemm_ops|attention_ops|sa_last_bwd|sa_fused) X="-mllvm -amdgpu-mfma-vgpr-form=1";; criterion_ops) X="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $X -DBUTD_MAIN_PRIO=$N -c butd_detr_amd/csrc/$U.hip -o scratch/exp/${U}_prio$N.o &
  NEW="$NEW scratch/exp/${U}_prio$N.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/exp/libprio$N$TAG.so $NEW $OBJS
ls -la scratch/exp/libprio$N$TAG.so
