# usage: build_abl.sh <unit> tag1 "-Dflags1" tag2 "-Dflags2" ...  -> scratch/exp/libabl_<tag>.so
# (<unit> = gemm_ops | attention_ops | ...: the translation unit rebuilt with the flags; the others are reused)
set -e
cd /root/repo
mkdir -p scratch/exp
U=$1; shift
OBJS=$(ls butd_detr_amd/lib/obj/*.o | grep -v /$U.o)
args=("$@")
for ((i=0; i<${#args[@]}; i+=2)); do
  t=${args[i]}; f=${args[i+1]}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -mllvm -amdgpu-mfma-vgpr-form=1 $f -c butd_detr_amd/csrc/$U.hip -o scratch/exp/${U}_abl_$t.o &
done
wait
for ((i=0; i<${#args[@]}; i+=2)); do
  t=${args[i]}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/exp/libabl_$t.so scratch/exp/${U}_abl_$t.o $OBJS
done
ls scratch/exp/*.so
