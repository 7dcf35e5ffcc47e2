mkdir -p gpurun_out/r2c11
BUTD_HIP_LIB=$PWD/scratch/exp/libr1.so timeout 600 python scratch/gemm_shapes.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c11/shapes_r1.txt
timeout 600 python scratch/gemm_shapes.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c11/shapes_r2.txt
head -3 gpurun_out/r2c11/shapes_r1.txt; head -3 gpurun_out/r2c11/shapes_r2.txt
