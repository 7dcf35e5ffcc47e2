p='butd_detr_amd/train_step.py'
s=open(p).read()
a=s.index('class FlatGradients:')
new = '''class FlatGradients:
    """One contiguous fp32 buffer holding every trainable gradient, so the data-parallel exchange is a
    single large all-reduce (85.7 MB for the full model) instead of 601 small ones -- the message size
    RCCL's ring over the 7 xGMI links is efficient at (SURVEY.md section 5).  ``views[i]`` aliases the
    slice of parameter i; ``gather`` fills the buffer from freshly produced ``.grad`` tensors with
    multi-tensor copies (a handful of launches), after which the parameters' ``.grad`` point at the
    views so clip/AdamW read the reduced values in place."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=torch.float32, device=ref.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def gather(self, grads):
        torch._foreach_copy_(self.views, grads)

    def attach(self):
        for p, v in zip(self.params, self.views):
            p.grad = v

    def detach(self):
        for p in self.params:
            p.grad = None

    def all_reduce_mean(self, group=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))


class GraphedTrainStep:
    """The whole iteration as hipGraph replays: tokenise on the host, copy into static buffers, replay
    ``forward_tokenized -> surrogate loss -> backward -> gather grads into the flat buffer`` (graph 1),
    all-reduce the flat buffer across ranks (outside the graph; skipped at world size 1), replay
    ``clip -> AdamW`` on the flat views (graph 2).

    Eager PyTorch launches ~4 900 kernels per step here and is host-bound (SURVEY.md: "HIP streams and
    graphs instead of a tracing compiler"); a graph replay removes the launch overhead without
    changing a single kernel.  Shapes are static: a new (batch, points, tokens) signature re-captures.
    """

    def __init__(self, model, optimizer, clip_norm=0.1, warmup=3, group=None):
        self.model, self.optimizer, self.clip_norm, self.group = model, optimizer, clip_norm, group
        self.warmup = warmup
        self.flat = FlatGradients([p for g in optimizer.param_groups for p in g["params"]])
        self._sig = None

    # -- pieces shared by the eager warm-up and the captured region
    def _fwd_bwd(self):
        end_points = self.model.forward_tokenized(self.s_inputs, self.s_tok)
        loss = surrogate_loss(end_points, self.s_targets)
        self.flat.detach()                      # fresh .grad tensors: no per-parameter accumulate
        loss.backward()
        self.flat.gather([p.grad for p in self.flat.params])
        return loss.detach()

    def _update(self):
        self.flat.attach()
        if self.clip_norm:
            torch.nn.utils.clip_grad_norm_(self.flat.views, self.clip_norm, foreach=True)
        self.optimizer.step()

    def _copy_in(self, inputs, targets, tok):
        for k, v in inputs.items():
            if torch.is_tensor(v):
                self.s_inputs[k].copy_(v, non_blocking=True)
        for k, v in targets.items():
            self.s_targets[k].copy_(v, non_blocking=True)
        for k in self.s_tok.keys():
            self.s_tok[k].copy_(tok[k], non_blocking=True)

    def _capture(self, inputs, targets, tok):
        from transformers import BatchEncoding
        self.s_inputs = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in inputs.items()}
        self.s_targets = {k: v.clone() for k, v in targets.items()}
        self.s_tok = BatchEncoding({k: v.clone() for k, v in tok.items()})
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self._fwd_bwd()
                self.flat.all_reduce_mean(self.group)
                self._update()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.g_fwd_bwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fwd_bwd):
            self.s_loss = self._fwd_bwd()
        self.g_update = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_update, pool=self.g_fwd_bwd.pool()):
            self._update()

    def __call__(self, inputs, targets):
        tok = self.model.tokenize(inputs)                      # host work stays in the step
        sig = (tuple(inputs["point_clouds"].shape), tuple(tok["input_ids"].shape))
        if sig != self._sig:
            self._capture(inputs, targets, tok)
            self._sig = sig
        self._copy_in(inputs, targets, tok)
        self.g_fwd_bwd.replay()
        self.flat.all_reduce_mean(self.group)
        self.g_update.replay()
        return self.s_loss
'''
s=s[:a]+new
open(p,'w').write(s)

# ---- LDS-accumulating scatter-add for small source sets
p='butd_detr_amd/csrc/pointnet2_ops.hip'
s=open(p).read()
old=s[s.index('__global__ __launch_bounds__(256) void index_scatter_add_kernel'):s.index('// three_nn: one thread per unknown point')]
new=old+'''// Same scatter-add when the source set is small (SA2..SA4 grouping grads: n <= 4096): a workgroup
// owns kLdsChan channels of one scene, accumulates ALL positions into an LDS image [chan][n] with
// ds_add_f32 (no global atomics, no contention across CUs) and writes the image out coalesced.
// grad_points needs no pre-zeroing on this path.
constexpr int kLdsScatterMaxN = 4096;
constexpr int kLdsChan = 8;
__global__ __launch_bounds__(1024) void index_scatter_add_lds_kernel(int c, int n, int P,
                                                                     const float *__restrict__ grad_out,
                                                                     const int *__restrict__ idx,
                                                                     float *__restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [kLdsChan][n]
  const int b = blockIdx.y;
  const int l0 = blockIdx.x * kLdsChan;
  const int nl = min(kLdsChan, c - l0);
  for (int i = threadIdx.x; i < kLdsChan * n; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  const float *src = grad_out + ((size_t)b * c + l0) * P;
  const int *ix = idx + (size_t)b * P;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    const int a = ix[p];
    for (int l = 0; l < nl; ++l) atomicAdd(&acc[l * n + a], src[(size_t)l * P + p]);
  }
  __syncthreads();
  float *dst = grad_points + ((size_t)b * c + l0) * n;
  for (int i = threadIdx.x; i < nl * n; i += blockDim.x) dst[i] = acc[i];
}

'''
s=s.replace(old,new)
old='''  const int P = npoints * nsample;
  hipLaunchKernelGGL(index_scatter_add_kernel, chan_grid(P, c, b), dim3(256), 0,
                     (hipStream_t)stream, c, n, P, grad_out, idx, grad_points);
  return launch_status();'''
assert old in s
s=s.replace(old,'''  const int P = npoints * nsample;
  if (n <= kLdsScatterMaxN && P >= 4 * n) {
    hipLaunchKernelGGL(index_scatter_add_lds_kernel, dim3((c + kLdsChan - 1) / kLdsChan, b),
                       dim3(1024), sizeof(float) * kLdsChan * n, (hipStream_t)stream, c, n, P,
                       grad_out, idx, grad_points);
    return launch_status();
  }
  hipLaunchKernelGGL(index_scatter_add_kernel, chan_grid(P, c, b), dim3(256), 0,
                     (hipStream_t)stream, c, n, P, grad_out, idx, grad_points);
  return launch_status();''')
open(p,'w').write(s)
