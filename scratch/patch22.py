p='butd_detr_amd/_hiplib.py'
s=open(p).read()
s=s.replace('''ALL_SYMBOLS = dict(POINTNET2_SYMBOLS)
ALL_SYMBOLS.update(ATTENTION_SYMBOLS)
ALL_SYMBOLS.update(SA_SYMBOLS)''','''OPTIM_SYMBOLS = {
    "butd_adamw_flat": (_c_int, [_P] * 4 + [_c_long, _c_long] + [_c_float] * 5 + [_P, _P, _P]),
}

ALL_SYMBOLS = dict(POINTNET2_SYMBOLS)
ALL_SYMBOLS.update(ATTENTION_SYMBOLS)
ALL_SYMBOLS.update(SA_SYMBOLS)
ALL_SYMBOLS.update(OPTIM_SYMBOLS)''')
open(p,'w').write(s)

p='butd_detr_amd/train_step.py'
s=open(p).read()
a=s.index('class FlatGradients:')
new='''class FlatAdamW:
    """AdamW over PACKED parameters (include/butd_optim.h): the parameters of every group are moved
    into one contiguous fp32 buffer (``p.data`` become views of it, group by group, each segment padded
    to 4 floats), with matching flat gradient / moment buffers; ``step()`` is one streaming kernel per
    group.  Same update rule and groups as ``make_optimizer`` (main_utils.py:258-283); the clip
    coefficient of ``clip_grad_norm_`` is applied inside the kernel."""

    def __init__(self, model, lr=1e-4, lr_backbone=1e-3, text_encoder_lr=1e-5, weight_decay=5e-4,
                 betas=(0.9, 0.999), eps=1e-8):
        from . import _hiplib
        self._lib = _hiplib.load()
        self._check = _hiplib.check
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        groups = [
            ([p for n, p in named if "backbone_net" not in n and "text_encoder" not in n], lr),
            ([p for n, p in named if "backbone_net" in n], lr_backbone),
            ([p for n, p in named if "text_encoder" in n], text_encoder_lr),
        ]
        groups = [(ps, l) for ps, l in groups if ps]
        dev = named[0][1].device
        pad4 = lambda n: (n + 3) // 4 * 4
        total = sum(pad4(p.numel()) for ps, _ in groups for p in ps)
        self.flat_p = torch.zeros(total, device=dev)
        self.flat_g = torch.zeros(total, device=dev)
        self.flat_m = torch.zeros(total, device=dev)
        self.flat_v = torch.zeros(total, device=dev)
        self.param_groups, self.grad_views, self.segments = [], [], []
        off = 0
        for ps, l in groups:
            begin = off
            for p in ps:
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_p[off:off + n].view_as(p)
                self.grad_views.append(self.flat_g[off:off + n].view_as(p))
                off += pad4(n)
            self.segments.append((begin, off, l))
            self.param_groups.append({"params": ps, "lr": l, "weight_decay": weight_decay})
        self.params = [p for g in self.param_groups for p in g["params"]]
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.step_count = torch.zeros(1, device=dev)
        self.grad_scale = torch.ones(1, device=dev)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def clip_(self, max_norm):
        """clip_grad_norm_ on the flat gradient buffer: only the coefficient is computed here."""
        norm = torch.linalg.vector_norm(self.flat_g)
        torch.clamp(max_norm / (norm + 1e-6), max=1.0, out=self.grad_scale[0])
        return norm

    def step(self):
        self.step_count.add_(1.0)
        stream = torch.cuda.current_stream(self.flat_p.device).cuda_stream
        for begin, end, lr in self.segments:
            err = self._lib.butd_adamw_flat(self.flat_p.data_ptr(), self.flat_g.data_ptr(),
                                            self.flat_m.data_ptr(), self.flat_v.data_ptr(), begin, end, lr,
                                            self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                            self.step_count.data_ptr(), self.grad_scale.data_ptr(), stream)
            self._check(err, "butd_adamw_flat")


'''
s=s[:a]+new+s[a:]
# GraphedTrainStep: support FlatAdamW
s=s.replace('''        self.flat = FlatGradients([p for g in optimizer.param_groups for p in g["params"]])
        self._sig = None''','''        self.flat_opt = isinstance(optimizer, FlatAdamW)
        self.flat = (_OptimizerGradients(optimizer) if self.flat_opt else
                     FlatGradients([p for g in optimizer.param_groups for p in g["params"]]))
        self._sig = None''')
s=s.replace('''    def _update(self):
        self.flat.attach()
        if self.clip_norm:
            torch.nn.utils.clip_grad_norm_(self.flat.views, self.clip_norm, foreach=True)
        self.optimizer.step()''','''    def _update(self):
        if self.flat_opt:
            if self.clip_norm:
                self.optimizer.clip_(self.clip_norm)
            self.optimizer.step()
            return
        self.flat.attach()
        if self.clip_norm:
            torch.nn.utils.clip_grad_norm_(self.flat.views, self.clip_norm, foreach=True)
        self.optimizer.step()''')
s=s.replace('''class GraphedTrainStep:''','''class _OptimizerGradients(FlatGradients):
    """FlatGradients over a FlatAdamW's own gradient buffer (no second copy)."""

    def __init__(self, opt):
        self.params, self.flat, self.views = opt.params, opt.flat_g, opt.grad_views


class GraphedTrainStep:''')
open(p,'w').write(s)

p='bench.py'
s=open(p).read()
s=s.replace('''        opt = make_optimizer(model, capturable=True)
        graphed = GraphedTrainStep(model, opt)''','''        from butd_detr_amd.train_step import FlatAdamW
        opt = FlatAdamW(model)
        graphed = GraphedTrainStep(model, opt)''')
s=s.replace('"launch": "eager+DDP" if args.eager else "hipGraph replay + flat-gradient all-reduce"','"launch": "eager+DDP+torch AdamW" if args.eager else "hipGraph replay + flat-gradient all-reduce + flat AdamW"')
open(p,'w').write(s)
