timeout 1200 python -m pytest tests/test_gpu_graph_step.py tests/test_gpu_two_ranks.py tests/test_gpu_flat_adamw.py -q -m gpu 2>&1 | tail -15
