nproc; python -c "import os;print(os.cpu_count(), len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; free -g | head -2
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 300 python bench.py --steps 3 --warmup 1 --backend torch --no-cpu-baseline 2>&1 | tail -3
