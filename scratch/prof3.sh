cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 3 --warmup 2 --backend hip --no-cpu-baseline > /tmp/bench_prof.log 2>&1
head -1 /tmp/prof/bench_kernel_trace.csv
for p in attn_fwd attn_bwd_dq attn_bwd_dkv gemm_kernel ln_bwd; do echo "== $p"; python scratch/trace_detail.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel $p | head -14; done
echo "== binary"; python scratch/trace_detail.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel "elementwise_kernel_manual_unroll<128, 4, at::native::gpu_kernel_impl_nocast<at::native::BinaryFunctor" | head -5
grep -o "elementwise_kernel_manual_unroll<128, 4, at::native::gpu_kernel_impl_nocast<at::native::BinaryFunctor[^\"]*" /tmp/prof/bench_kernel_trace.csv | sort | uniq -c | sort -rn | head -3 | cut -c1-400
