# usage: build_abl.sh tag1 "-Dflags1" tag2 "-Dflags2" ...  -> scratch/exp/libabl_<tag>.so
set -e
cd /root/repo
mkdir -p scratch/exp
OBJS=$(ls butd_detr_amd/lib/obj/*.o | grep -v gemm_ops)
args=("$@")
for ((i=0; i<${#args[@]}; i+=2)); do
  t=${args[i]}; f=${args[i+1]}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $f -c butd_detr_amd/csrc/gemm_ops.hip -o scratch/exp/gemm_abl_$t.o &
done
wait
for ((i=0; i<${#args[@]}; i+=2)); do
  t=${args[i]}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/exp/libabl_$t.so scratch/exp/gemm_abl_$t.o $OBJS
done
ls scratch/exp/*.so
