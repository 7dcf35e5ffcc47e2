set -e
mkdir -p gpurun_out
timeout 600 python scratch/gemm_shapes.py --no-cpu-baseline > gpurun_out/gemm_shapes.txt 2>&1 || tail -30 gpurun_out/gemm_shapes.txt
tail -80 gpurun_out/gemm_shapes.txt
