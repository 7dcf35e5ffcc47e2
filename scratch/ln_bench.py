import torch, os
from butd_detr_amd import _hiplib
lib = _hiplib.load()
s = torch.cuda.current_stream().cuda_stream
for rows in (8192, 2048, 640):
    cols = 288
    dy, x, res = (torch.randn(rows, cols, device="cuda") for _ in range(3))
    g = torch.ones(cols, device="cuda"); mean = torch.zeros(rows, device="cuda"); rstd = torch.ones(rows, device="cuda")
    dx, dres = torch.empty_like(x), torch.empty_like(x)
    dg, db = torch.zeros(cols, device="cuda"), torch.zeros(cols, device="cuda")
    ctr = torch.zeros(1, dtype=torch.int64, device="cuda")
    slots = torch.zeros(16 * 2 * cols + 1, device="cuda")
    use = os.environ.get('LN_SLOTS', '1') == '1'
    def f(): lib.butd_add_dropout_layernorm_bwd(rows, cols, dy.data_ptr(), x.data_ptr(), res.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dres.data_ptr(), dg.data_ptr(), db.data_ptr(), slots.data_ptr() if use else None, 0.1, 5, ctr.data_ptr(), s)
    def f2():
        slots.zero_(); f()
    f0 = f
    for _ in range(5): f2()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(50): f2()
    b.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('BUTD_HIP_LIB','default')[-12:]} rpw={os.environ.get('BUTD_LN_RPW','-')} rows={rows}: {a.elapsed_time(b)/50*1e3:.1f} us")
