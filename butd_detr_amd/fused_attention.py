"""Fused attention / FFN blocks on the gfx950 kernels of include/butd_attention.h.

One ``autograd.Function`` per *block* (not per op): the forward of
``LayerNorm(residual + Dropout(MHA(query, key, value)))`` is 4 launches -- grouped Q/K/V projection
(fp32 MFMA GEMM, ``src + pos`` folded into the operand load, ``1/sqrt(head_dim)`` and bias in the
epilogue), flash-style attention core, output projection, residual+dropout+LayerNorm -- and the
hand-written backward is 7; ``LayerNorm(x + FFN(x))`` is 3 + 3.  The stock-torch chain behind the
same reference code (encoder_decoder_layers.py:87-122,149-155,179-185,356-404) is ~25 + ~50 launches
per block, which is what made the 45 attention blocks of a step launch-bound.

Dropout masks come from a counter-based hash (csrc/rng.h): nothing is stored, backward regenerates
them from (step counter, site id, element index).  ``new_step()`` bumps the device-resident counter
once per model forward, so a captured hipGraph draws fresh masks on every replay.
"""
import ctypes
import math
import os

import torch

from . import _hiplib, switches
from ._hiplib import GemmProblem

_lib = _hiplib.load()

_rng_counters = {}
_site = [0]


def rng_counter(device):
    t = _rng_counters.get(device)
    if t is None:
        t = torch.zeros(1, dtype=torch.int64, device=device)
        _rng_counters[device] = t
    return t


def new_step(device):
    """Advance the dropout step counter (device-side add: capturable) and restart site numbering."""
    rng_counter(device).add_(1)
    _site[0] = 0


def _next_site():
    _site[0] += 1
    return _site[0]


class ZeroArena:
    """One zero-filled device buffer per training step for everything the fused backward passes
    accumulate into with atomics (weight-gradient slabs, BatchNorm sums, scatter targets): ``reset()``
    is ONE memset at the start of a step and ``zeros()`` hands out consecutive 16-byte aligned views,
    instead of ~130 ``torch.zeros`` fills per step.  Views are only valid until the next ``reset()``,
    so the arena is opt-in (``with arena:``) for loops that consume the gradients inside the step --
    ``GraphedTrainStep`` copies them into its flat buffer in the same graph.  The capacity is found on
    the first (eager) steps: overflow goes to extra chunks that ``reset()`` later merges."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.chunks, self.index, self.offset = [], 0, 0
        self.captured = False

    def reset(self):
        capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        # once a captured graph has the chunks' addresses baked in they are never released or replaced: a later
        # signature's eager warm-up merging them would leave the earlier graph replaying into freed memory
        self.captured = self.captured or capturing
        if len(self.chunks) > 1 and not capturing and not self.captured:
            total = sum(c.numel() for c in self.chunks)
            self.chunks = [torch.zeros(total, dtype=torch.uint8, device=self.device)]
        else:
            for c in self.chunks:
                c.zero_()
        self.index, self.offset = 0, 0

    def zeros(self, shape, dtype=torch.float32):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        n = 1
        for d in shape:
            n *= d
        nbytes = (n * torch.empty((), dtype=dtype).element_size() + 15) // 16 * 16
        while True:
            if self.index < len(self.chunks):
                c = self.chunks[self.index]
                if self.offset + nbytes <= c.numel():
                    raw = c[self.offset:self.offset + nbytes]
                    self.offset += nbytes
                    return raw.view(dtype)[:n].view(shape)
                self.index, self.offset = self.index + 1, 0
                continue
            self.chunks.append(torch.zeros(max(nbytes, 1 << 20), dtype=torch.uint8, device=self.device))

    def __enter__(self):
        global _arena
        self._prev, _arena = _arena, self
        return self

    def __exit__(self, *exc):
        global _arena
        _arena = self._prev


_arena = None


def zeros(shape, dtype=torch.float32, device=None):
    """``torch.zeros`` for accumulation targets of the fused blocks, served from the active arena."""
    if _arena is not None and device is not None and torch.device(device) == _arena.device:
        return _arena.zeros(shape, dtype)
    return torch.zeros(shape, dtype=dtype, device=device)


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


_compute_bf16 = [False]
# measurement hook (bench.py / step_regions.py): called as hook(kind, flops) for every attention-core launch
# ("attn_fwd": 4 B H Lq Lk D; "attn_bwd": 2.5 x that); None on the product path
_work_hook = [None]


def _count_attention(kind, B, H, Lq, Lk, D):
    if _work_hook[0] is not None:
        _work_hook[0](kind, (4.0 if kind == "attn_fwd" else 10.0) * B * H * Lq * Lk * D)


def set_compute_dtype(name):
    """'f32' (the reference's precision: exact fp32 MFMA) or 'bf16' (BASELINE configs[3]: operands of every
    grouped product -- projections, FFN, 1x1-conv chains and their gradient products -- and of the attention
    core's matrix steps rounded to bf16 on the matrix cores, fp32 accumulation; LayerNorm / softmax / BatchNorm
    statistics and all tensors in memory stay fp32).  Returns the previous setting."""
    if name not in ("f32", "bf16"):
        raise ValueError(name)
    prev = "bf16" if _compute_bf16[0] else "f32"
    _compute_bf16[0] = name == "bf16"
    return prev


_split_fwd = [False]     # round 6 microbenchmark hook (scratch/r6_split_bf16.py): the forward core on split-bf16 operands


def _attn_fwd():
    if _split_fwd[0]:
        return _lib.butd_attention_fwd_split_bf16
    return _lib.butd_attention_fwd_bf16 if _compute_bf16[0] else _lib.butd_attention_fwd


def _attn_bwd():
    return _lib.butd_attention_bwd_bf16 if _compute_bf16[0] else _lib.butd_attention_bwd


_short_keys = [True]
_SHORT_KEYS_MAX = int(_lib.butd_attention_bwd_short_keys_max())


def set_short_keys(flag):
    """Test switch: the round-4 one-kernel backward for key sets of <= 144 rows (the fallback where the one-pass kernel does not serve)."""
    prev, _short_keys[0] = _short_keys[0], bool(flag)
    return prev


_long_keys = [switches.flag("attn_long_keys", True)]


def _short_key_bwd(Lq, Lk, B=0, H=0, D=0):
    """The round-4 single-pass backward for short key sets (``butd_attention_bwd_short_keys``: dk / dv accumulated with
    atomics, caller zero-fills) serves this call: only where the round-5 one-pass kernel (no atomics; 1024 x 80: 64 us
    against 89) does not -- other head dimensions, or BUTD_AB=attn_long_keys=0.  Measured for that case
    (profiles/r04_attention_short_keys.txt, 8 x 8 heads): it wins where a workgroup walks several query tiles over one
    staged key set -- 1024 x 80: 88 us vs 115 us for the two kernels -- and loses where its dK / dV atomics are not
    amortised: 256 x 80: 58 vs 40 us, 256 x 132: 91 vs 47, 1024 x 132: 135 vs 123."""
    if not _short_keys[0] or _compute_bf16[0] or Lk > _SHORT_KEYS_MAX:
        return False
    if _long_keys[0] and B and int(_lib.butd_attention_bwd_long_keys_scratch(B, H, Lq, Lk, D, 0)) >= 0:
        return False
    return Lq >= 512 and Lk <= 80


def set_long_keys(flag):
    """A/B switch (BUTD_AB=attn_long_keys=0): the one-pass backward (butd_attention_bwd_long_keys)."""
    prev, _long_keys[0] = _long_keys[0], bool(flag)
    return prev


def _attention_backward(B, H, Lq, Lk, D, q, k, v, mask, att, d_att, lse, dq_ptr, dk_ptr, dv_ptr, ld_dq, ld_dkv, scale,
                        p_attn, site_attn, short, ref):
    """dq, dk, dv of one attention call (include/butd_attention.h): the one-pass kernel for long key sets (every score
    tile once; dQ through per-chunk slabs), the short-key kernel where the caller chose it (dk / dv zero-filled), else
    the two-kernel walk."""
    dev = ref.device
    ctr = rng_counter(dev).data_ptr()
    with torch.cuda.device(dev):
        need = -1
        # (which shapes it serves, and with how many keys per workgroup: the library's measured rule, attention_ops.hip)
        if _long_keys[0] and not short and not _compute_bf16[0]:
            need = int(_lib.butd_attention_bwd_long_keys_scratch(B, H, Lq, Lk, D, ld_dq))
        bf16_need = -1
        if _long_keys[0] and not short and _compute_bf16[0]:
            bf16_need = int(_lib.butd_attention_bwd_long_keys_bf16_scratch(B, H, Lq, Lk, D, ld_dq))
        if bf16_need >= 0:      # the bf16 operating point: the long key sets in one pass as well
            ws = torch.empty(max(bf16_need, 1), device=dev)
            err = _lib.butd_attention_bwd_long_keys_bf16(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                                         _ptr(mask), att.data_ptr(), d_att.data_ptr(), lse.data_ptr(),
                                                         dq_ptr, dk_ptr, dv_ptr, ld_dq, ld_dkv, scale, p_attn, site_attn,
                                                         ctr, ws.data_ptr(), bf16_need, _stream(ref))
        elif need >= 0:
            ws = torch.empty(max(need, 1), device=dev)
            err = _lib.butd_attention_bwd_long_keys(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), _ptr(mask),
                                                    att.data_ptr(), d_att.data_ptr(), lse.data_ptr(), dq_ptr, dk_ptr,
                                                    dv_ptr, ld_dq, ld_dkv, scale, p_attn, site_attn, ctr, ws.data_ptr(),
                                                    need, _stream(ref))
        else:
            delta = torch.empty((B, H, Lq), device=dev)
            err = (_lib.butd_attention_bwd_short_keys if short else _attn_bwd())(
                B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), _ptr(mask), att.data_ptr(), d_att.data_ptr(),
                lse.data_ptr(), delta.data_ptr(), dq_ptr, dk_ptr, dv_ptr, ld_dq, ld_dkv, scale, p_attn, site_attn, ctr,
                _stream(ref))
    _hiplib.check(err, "butd_attention_bwd")
    _count_attention("attn_bwd", B, H, Lq, Lk, D)


def get_compute_dtype():
    return "bf16" if _compute_bf16[0] else "f32"


def _problem(a, b, c, M, N, K, lda, ldb, ldc, *, a2=None, a2_mode=0, a2_scale=1.0, bias=None,
             bias_grad=None, scale=1.0, relu=False, accumulate=False, ones_col=False, split_k=1,
             dropout_p=0.0, site=0, a_affine=None, b_affine=None, a_drop=(0.0, 0), b_drop=(0.0, 0),
             col_stats=None, c_add=False, c2=None, col_slots=(0, 0), gate=None, gate_scale=1.0, a_bn=None, c_bn=None):
    """a_bn = (sum, sumsq, gamma, beta, running_mean, running_var, nbt, out (4 rows), ld_out, count, eps, momentum): the A
    operand's BatchNorm + ReLU computed by the kernel from the producer's column sums (butd_gemm_problem.a_bn_*)."""
    asc, ash = a_affine if a_affine is not None else (None, None)
    bsc, bsh = b_affine if b_affine is not None else (None, None)
    return GemmProblem(_ptr(a), _ptr(a2), _ptr(b), _ptr(bias), _ptr(c), _ptr(bias_grad), M, N, K,
                       lda[0], lda[1], ldb[0], ldb[1], ldc, scale, a2_mode, a2_scale, int(relu),
                       int(accumulate), int(ones_col), split_k, dropout_p, site,
                       _ptr(asc), _ptr(ash), _ptr(bsc), _ptr(bsh),
                       float(a_drop[0]), int(a_drop[1]), float(b_drop[0]), int(b_drop[1]),
                       _ptr(col_stats[0]) if col_stats is not None else None,
                       _ptr(col_stats[1]) if col_stats is not None else None,
                       int(c_add), _ptr(c2), int(col_slots[0]), int(col_slots[1]), int(_compute_bf16[0]),
                       _ptr(gate), float(gate_scale),
                       *((None,) * 8 + (0, 0, 0.0, 0.0) if a_bn is None else
                         (_ptr(a_bn[0]), _ptr(a_bn[1]), _ptr(a_bn[2]), _ptr(a_bn[3]), _ptr(a_bn[4]),
                          _ptr(a_bn[5]), _ptr(a_bn[6]), _ptr(a_bn[7]), int(a_bn[8]), int(a_bn[9]),
                          float(a_bn[10]), float(a_bn[11]))),
                       # c_bn = (z, aff (4 rows: mean, rstd, scale, shift), ld of aff, dropout p, site): butd_gemm_problem.c_bn_*
                       *(() if c_bn is None else (_ptr(c_bn[0]), _ptr(c_bn[1]), int(c_bn[2]), float(c_bn[3]), int(c_bn[4]))))


_MAX_GROUP = 32        # butd_gemm_grouped's limit (kMaxProblems)
# Deterministic split-K (include/butd_attention.h: c_partial / fold_src).  A weight-gradient problem writes per-slice slabs;
# its fold is an element-wise rider that joins the NEXT grouped launch of the same backward function (no launch of its
# own), and what is still pending when that function returns goes out as one small launch (``fold_scope``).  The gradient
# tensors a backward function hands to autograd are therefore complete in stream order when it returns -- AccumulateGrad
# may clone or add them right away (it steals the reference only when nothing else holds one), and
# ``fused_sa._finish`` reshapes one; deferring the folds across functions (one flush per backward pass from an engine
# callback) was tried first and left ffn.0.weight's gradient a clone of the untouched slab.  Outside a scope the fold is
# launched at once.
# OFF by default -- MEASURED (profiles/r05_wgrad_slabs.txt, 3 x 60 steps alternating): 24.15 ms per step with the slabs,
# 23.35 with round 4's float atomics.  The atomics are executed by the memory side next to the matrix work of the other
# workgroups (that is the 2x "wasted" traffic the counters show), the slabs cost ~90 small flush launches per step plus
# their own write + read.  set_wgrad_slabs(True) / BUTD_AB=wgrad_slabs=1 is the bit-reproducible training mode.
_wgrad_slabs = [switches.flag("wgrad_slabs", False)]
_pending_folds = []        # fold problems whose producing launch is enqueued (current scope, current stream)
_scope_depth = [0]


def set_wgrad_slabs(flag):
    prev, _wgrad_slabs[0] = _wgrad_slabs[0], bool(flag)
    return prev


def _launch(problems, device, stream_handle):
    arr = (GemmProblem * len(problems))(*problems)
    with torch.cuda.device(device):
        err = _lib.butd_gemm_grouped(arr, len(problems), rng_counter(device).data_ptr(), stream_handle)
    _hiplib.check(err, "butd_gemm_grouped")


def _flush_folds(ref):
    """Launch the pending folds on the stream of ``ref`` (the end of a backward function, or an eager call)."""
    folds = list(_pending_folds)
    del _pending_folds[:]
    for i in range(0, len(folds), _MAX_GROUP):
        _launch(folds[i:i + _MAX_GROUP], ref.device, _stream(ref))


def fold_scope(backward):
    """Decorator of a ``Function.backward``: the split-K folds of its launches ride in its later launches; the rest is
    flushed before it returns (a backward function runs on ONE stream: the stream of its forward)."""
    import functools

    @functools.wraps(backward)
    def wrapped(ctx, *grads):
        _scope_depth[0] += 1
        try:
            out = backward(ctx, *grads)
        except BaseException:
            if _scope_depth[0] == 1:
                del _pending_folds[:]       # their slabs die with this frame: a later launch must not carry them
            raise
        finally:
            _scope_depth[0] -= 1
        if _pending_folds and _scope_depth[0] == 0:
            ref = next((g for g in grads if g is not None), None)
            if ref is None:                 # (a node whose incoming gradients are all None: the current device's stream)
                ref = torch.empty(0, device=torch.device("cuda", torch.cuda.current_device()))
            _flush_folds(ref)
        return out
    return wrapped


def _gemm(problems, ref):
    if _pending_folds and len(problems) < _MAX_GROUP:    # riders: folds of earlier launches of this backward function
        room = _MAX_GROUP - len(problems)
        problems = list(problems) + _pending_folds[:room]
        del _pending_folds[:room]
    _launch(problems, ref.device, _stream(ref))
    new = [p._fold for p in problems if getattr(p, "_fold", None) is not None]
    if new:
        _pending_folds.extend(new)
        if _scope_depth[0] == 0:
            _flush_folds(ref)


# operand descriptors ------------------------------------------------------------------------------
def _fwd(x, w, y, M, N, K, **kw):
    """y[M,N] = x[M,K] @ w[N,K]^T (both contraction-contiguous)."""
    return _problem(x, w, y, M, N, K, (K, 1), (K, 1), N, **kw)


def _dgrad(dy, w, dx, M, N, K, **kw):
    """dx[M,K] = dy[M,N] @ w[N,K]: contraction over N; B(col=k, n) = w[n*K + k]."""
    return _problem(dy, w, dx, M, K, N, (N, 1), (1, K), K, **kw)


def _wgrad(dy, x, dw, db, M, N, K, **kw):
    """dw[N,K] += dy[M,N]^T @ x[M,K], db[N] += column sums of dy: contraction over M (split-K)."""
    # measured (288x288 products, graph replay): slices of ~256 rows, but never fewer than 8 slices once
    # there are 512 rows to share out -- a 640-row product takes 9.6 us with 8 slices, 15.9 us with 2
    # (a thin 10^6-row product such as SA1's first layer, 64 x 8: 155 us with 256 slices, 105 with 512)
    cap = 512 if N * K <= 1024 else 256
    # (re-measured in the step, round 5: 128 / 512 rows per slice, caps of 16 / 64 slices: within noise or slower)
    split = max(1, min(cap, M // 512)) if M >= 16384 else min(32, max(M // 256, min(8, M // 64), 1))
    return _slabbed(_problem(dy, x, dw, N, K, M, (1, N), (1, K), K, bias_grad=db, ones_col=db is not None,
                             accumulate=True, split_k=split, **kw), dw, db)


def _slabbed(prob, dw, db):
    """The accumulate (split-K, atomics) problem ``prob`` with target dw (dense rows) / db in its deterministic form:
    per-slice slabs + a fold problem (``prob._fold``) that ``_gemm`` schedules.  The fold STORES: dw / db need no zero fill --
    and must have this ONE producer (every call site hands a fresh, disjoint slice of its gradient slab; a target that
    already holds another product's share has to stay on the atomic form)."""
    if not _wgrad_slabs[0] or (prob.M * prob.N) % 4 or prob.ldc != prob.N or (dw.data_ptr() & 15):
        return prob
    kslab = (prob.K + 31) // 32
    split = max(1, min(prob.split_k, kslab))
    per = -(-kslab // split)
    slices = -(-kslab // per)                  # every slice owns at least one 32-deep slab
    len1, len2 = prob.M * prob.N, (prob.M if db is not None else 0)
    stride = (len1 + len2 + 3) // 4 * 4
    ws = torch.empty(slices * stride, device=dw.device)
    prob.split_k, prob.c_partial, prob.c_partial_stride = slices, ws.data_ptr(), stride
    fold = GemmProblem()
    fold.c, fold.bias_grad = dw.data_ptr(), _ptr(db)
    fold.M, fold.N, fold.K = 1, 1, 1
    fold.fold_src, fold.fold_count, fold.fold_stride = ws.data_ptr(), slices, stride
    fold.fold_len, fold.fold_len2 = len1, len2
    fold._keep = (ws, dw, db)
    prob._fold = fold
    return prob


def _check(*tensors):
    for t in tensors:
        if t is not None:
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), \
                "fused attention kernels take contiguous fp32 CUDA tensors"


# -------------------------------------------------------------------------------------------------
# LayerNorm backward without same-address atomics: the kernel writes per-workgroup column sums, the grouped launch that
# follows it in every block folds them (one more problem: dgamma | dbeta = ones(1, blocks) . partials).  Measured alone
# (profiles/r04_small_kernels.txt): the 128 .. 512 atomics per column are 2.9 us of 8.5 us at 2048 rows, 5.0 of 18.4 at
# 8192.  set_ln_fold(False) / BUTD_AB=ln_fold=0: the atomic kernel (round 3's).
# -------------------------------------------------------------------------------------------------
_ln_fold = [switches.flag("ln_fold", True)]
_ones_rows = {}
_ones_retired = []      # never released: a captured graph may have an older row's address baked in


def set_ln_fold(flag):
    prev, _ln_fold[0] = _ln_fold[0], bool(flag)
    return prev


def _ones_row(dev, n):
    t = _ones_rows.get(dev)
    if t is None or t.numel() < n:
        if torch.cuda.is_current_stream_capturing():      # (a fill captured into a graph has not run yet: no fold now)
            return None
        if t is not None:
            _ones_retired.append(t)
        t = torch.ones(max(n, 4096), device=dev)
        _ones_rows[dev] = t
    return t


def _ln_bwd(M, E, dy, x, res, gamma, mean, rstd, dx, d_res, d_gamma, d_beta, p, site, ref):
    """butd_add_dropout_layernorm_bwd; returns the problems the NEXT grouped launch has to carry ([] or the fold).
    The fold problem owns the partial-sum table (``_keep``): the caller's list keeps it alive until the grouped launch
    that reads it has been enqueued -- an output of that very launch allocated in between must not land on its memory."""
    dev = ref.device
    nb = _lib.butd_layernorm_bwd_blocks(M)
    ones = None
    if (_ln_fold[0] and not _compute_bf16[0] and nb >= 4 and nb % 4 == 0 and E % 4 == 0
            and d_beta.data_ptr() == d_gamma.data_ptr() + 4 * E):
        ones = _ones_row(dev, nb)
    with torch.cuda.device(dev):
        if ones is not None:
            part = torch.empty((nb, 2 * E), device=dev)
            err = _lib.butd_add_dropout_layernorm_bwd_partial(
                M, E, dy.data_ptr(), x.data_ptr(), _ptr(res), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                dx.data_ptr(), d_res.data_ptr(), part.data_ptr(), p, site, rng_counter(dev).data_ptr(), _stream(ref))
        else:
            err = _lib.butd_add_dropout_layernorm_bwd(
                M, E, dy.data_ptr(), x.data_ptr(), _ptr(res), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                dx.data_ptr(), d_res.data_ptr(), d_gamma.data_ptr(), d_beta.data_ptr(), p, site,
                rng_counter(dev).data_ptr(), _stream(ref))
    _hiplib.check(err, "butd_add_dropout_layernorm_bwd")
    if ones is None:
        return []
    fold = _problem(ones, part, d_gamma, 1, 2 * E, nb, (nb, 1), (1, 2 * E), 2 * E)
    fold._keep = (part, ones)
    return [fold]


class _AttentionBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, residual, xq, xk, xv, mask, w_in, b_in, w_o, b_o, gamma, beta,
                num_heads, eps, p_attn, p_out, site_attn, site_out):
        B, Lq, E = xq.shape
        Lk = xk.shape[1]
        H, D = num_heads, E // num_heads
        Mq, Mk = B * Lq, B * Lk
        dev = xq.device
        q = torch.empty((B, Lq, E), device=dev)
        k = torch.empty((B, Lk, E), device=dev)
        v = torch.empty((B, Lk, E), device=dev)
        scale = math.sqrt(1.0 / float(D))
        _gemm([_fwd(xq, w_in[:E], q, Mq, E, E, bias=b_in[:E], scale=scale),
               _fwd(xk, w_in[E:2 * E], k, Mk, E, E, bias=b_in[E:2 * E]),
               _fwd(xv, w_in[2 * E:], v, Mk, E, E, bias=b_in[2 * E:])], xq)
        att = torch.empty((B, Lq, E), device=dev)
        lse = torch.empty((B, H, Lq), device=dev)
        with torch.cuda.device(dev):
            err = _attn_fwd()(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                          _ptr(mask), att.data_ptr(), lse.data_ptr(), p_attn, site_attn,
                                          rng_counter(dev).data_ptr(), _stream(xq))
        _hiplib.check(err, "butd_attention_fwd")
        _count_attention("attn_fwd", B, H, Lq, Lk, D)
        proj = torch.empty((B, Lq, E), device=dev)
        _gemm([_fwd(att, w_o, proj, Mq, E, E, bias=b_o)], xq)
        y = torch.empty((B, Lq, E), device=dev)
        mean = torch.empty((Mq,), device=dev)
        rstd = torch.empty((Mq,), device=dev)
        with torch.cuda.device(dev):
            err = _lib.butd_add_dropout_layernorm_fwd(
                Mq, E, proj.data_ptr(), residual.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps,
                y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), p_out, site_out,
                rng_counter(dev).data_ptr(), _stream(xq))
        _hiplib.check(err, "butd_add_dropout_layernorm_fwd")
        ctx.save_for_backward(residual, xq, xk, xv, mask, w_in, w_o, gamma, q, k, v, att, lse, proj,
                              mean, rstd)
        ctx.cfg = (H, p_attn, p_out, site_attn, site_out)
        return y

    @staticmethod
    @fold_scope
    def backward(ctx, dy):
        (residual, xq, xk, xv, mask, w_in, w_o, gamma, q, k, v, att, lse, proj, mean,
         rstd) = ctx.saved_tensors
        H, p_attn, p_out, site_attn, site_out = ctx.cfg
        B, Lq, E = xq.shape
        Lk = xk.shape[1]
        D = E // H
        Mq, Mk = B * Lq, B * Lk
        dev = xq.device
        dy = dy.contiguous()
        # one zero-filled slab for everything that is accumulated with atomics
        slab = zeros(3 * E * E + 3 * E + E * E + E + 2 * E, device=dev)
        o = 0
        d_w_in = slab[o:o + 3 * E * E].view(3 * E, E); o += 3 * E * E
        d_b_in = slab[o:o + 3 * E]; o += 3 * E
        d_w_o = slab[o:o + E * E].view(E, E); o += E * E
        d_b_o = slab[o:o + E]; o += E
        d_gamma = slab[o:o + E]; o += E
        d_beta = slab[o:o + E]
        d_res = torch.empty((B, Lq, E), device=dev)
        d_proj = torch.empty((B, Lq, E), device=dev) if p_out > 0 else d_res
        fold = _ln_bwd(Mq, E, dy, proj, residual, gamma, mean, rstd, d_proj, d_res, d_gamma, d_beta, p_out, site_out, xq)
        d_att = torch.empty((B, Lq, E), device=dev)
        _gemm([_dgrad(d_proj, w_o, d_att, Mq, E, E),
               _wgrad(d_proj, att, d_w_o, d_b_o, Mq, E, E)] + fold, xq)
        short = _short_key_bwd(Lq, Lk, B, H, D)
        dq = torch.empty((B, Lq, E), device=dev)
        dk = zeros((B, Lk, E), device=dev) if short else torch.empty((B, Lk, E), device=dev)
        dv = zeros((B, Lk, E), device=dev) if short else torch.empty((B, Lk, E), device=dev)
        _attention_backward(B, H, Lq, Lk, D, q, k, v, mask, att, d_att, lse, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                            0, 0, 1.0, p_attn, site_attn, short, xq)
        scale = math.sqrt(1.0 / float(D))
        d_xq = torch.empty((B, Lq, E), device=dev)
        d_xk = torch.empty((B, Lk, E), device=dev)
        d_xv = torch.empty((B, Lk, E), device=dev)
        _gemm([_dgrad(dq, w_in[:E], d_xq, Mq, E, E, scale=scale),
               _dgrad(dk, w_in[E:2 * E], d_xk, Mk, E, E),
               _dgrad(dv, w_in[2 * E:], d_xv, Mk, E, E)], xq)
        _gemm([_wgrad(dq, xq, d_w_in[:E], d_b_in[:E], Mq, E, E, scale=scale),
               _wgrad(dk, xk, d_w_in[E:2 * E], d_b_in[E:2 * E], Mk, E, E),
               _wgrad(dv, xv, d_w_in[2 * E:], d_b_in[2 * E:], Mk, E, E)], xq)
        return (d_res, d_xq, d_xk, d_xv, None, d_w_in, d_b_in, d_w_o, d_b_o, d_gamma, d_beta,
                None, None, None, None, None, None)


class _FfnBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, eps, p1, p2, site1, site2, next_pos=None):
        """``next_pos``: also return y + next_pos (values only), written by the LayerNorm kernel: the `src + pos` of the
        NEXT layer's first block (encoder_decoder_layers.py:179-185)."""
        B, L, E = x.shape
        Fh = w1.shape[0]
        M = B * L
        dev = x.device
        h = torch.empty((B, L, Fh), device=dev)
        o = torch.empty((B, L, E), device=dev)
        y = torch.empty((B, L, E), device=dev)
        mean = torch.empty((M,), device=dev)
        rstd = torch.empty((M,), device=dev)
        _gemm([_fwd(x, w1, h, M, Fh, E, bias=b1, relu=True, dropout_p=p1, site=site1)], x)
        _gemm([_fwd(h, w2, o, M, E, Fh, bias=b2)], x)
        y_pos = torch.empty((B, L, E), device=dev) if next_pos is not None else None
        with torch.cuda.device(dev):
            err = _lib.butd_add_dropout_layernorm_fwd_pos(
                M, E, o.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, y.data_ptr(),
                mean.data_ptr(), rstd.data_ptr(), p2, site2, rng_counter(dev).data_ptr(), _ptr(next_pos),
                _ptr(y_pos), _stream(x))
        _hiplib.check(err, "butd_add_dropout_layernorm_fwd_pos")
        ctx.save_for_backward(x, w1, w2, gamma, h, o, mean, rstd)
        ctx.cfg = (p1, p2, site1, site2)
        if y_pos is None:
            return y
        ctx.mark_non_differentiable(y_pos)
        ctx.set_materialize_grads(False)
        return y, y_pos

    @staticmethod
    @fold_scope
    def backward(ctx, dy, _d_y_pos=None):
        if dy is None:
            return (None,) * 13
        x, w1, w2, gamma, h, o, mean, rstd = ctx.saved_tensors
        p1, p2, site1, site2 = ctx.cfg
        B, L, E = x.shape
        Fh = w1.shape[0]
        M = B * L
        dev = x.device
        dy = dy.contiguous()
        slab = zeros(Fh * E + Fh + E * Fh + E + 2 * E, device=dev)
        off = 0
        d_w1 = slab[off:off + Fh * E].view(Fh, E); off += Fh * E
        d_b1 = slab[off:off + Fh]; off += Fh
        d_w2 = slab[off:off + E * Fh].view(E, Fh); off += E * Fh
        d_b2 = slab[off:off + E]; off += E
        d_gamma = slab[off:off + E]; off += E
        d_beta = slab[off:off + E]
        d_x = torch.empty((B, L, E), device=dev)          # residual path first, FFN path accumulates
        d_o = torch.empty((B, L, E), device=dev)
        fold = _ln_bwd(M, E, dy, o, x, gamma, mean, rstd, d_o, d_x, d_gamma, d_beta, p2, site2, x)
        d_h = torch.empty((B, L, Fh), device=dev)
        gate = 1.0 / (1.0 - p1) if p1 > 0 else 1.0        # h = relu(z) * keep / (1-p): h > 0 <=> live
        # the product that creates d_h applies the ReLU / dropout gate in its epilogue (c_gate): the two products that
        # read d_h then stay on the float4 staging path (they carried h as a companion operand before)
        _gemm([_dgrad(d_o, w2, d_h, M, E, Fh, gate=h, gate_scale=gate),
               _wgrad(d_o, h, d_w2, d_b2, M, E, Fh)] + fold, x)
        _gemm([_dgrad(d_h, w1, d_x, M, Fh, E, c_add=True),
               _wgrad(d_h, x, d_w1, d_b1, M, Fh, E)], x)
        return d_x, d_w1, d_b1, d_w2, d_b2, d_gamma, d_beta, None, None, None, None, None, None


class _LinearReluChain(torch.autograd.Function):
    """y = W_n(relu(... relu(W_1 x + b_1) ...)) + b_n over the rows of x: the contrastive-alignment projections
    (bdetr.py:104-121: Linear-ReLU-Linear-ReLU-Linear).  n products forward (ReLU in the epilogue); backward n
    launches, each the input-gradient product of a layer -- with the previous layer's ReLU gate applied in its
    epilogue -- together with that layer's weight / bias gradient."""

    @staticmethod
    def forward(ctx, x, *params):
        n = len(params) // 2
        ws, bs = params[0::2], params[1::2]
        lead, dev = x.shape[:-1], x.device
        a = x.reshape(-1, x.shape[-1])
        M = a.shape[0]
        acts = [a]
        for i, (w, b) in enumerate(zip(ws, bs)):
            N, K = w.shape
            out = torch.empty((M, N), device=dev)
            _gemm([_fwd(acts[-1], w, out, M, N, K, bias=b, relu=i < n - 1)], x)
            acts.append(out)
        ctx.save_for_backward(*acts[:-1], *ws)
        ctx.n, ctx.has_bias = n, [b is not None for b in bs]
        return acts[-1].view(*lead, ws[-1].shape[0])

    @staticmethod
    @fold_scope
    def backward(ctx, dy):
        n = ctx.n
        acts, ws = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        dev = dy.device
        g = dy.reshape(-1, dy.shape[-1]).contiguous()
        M = g.shape[0]
        sizes = [w.numel() + w.shape[0] for w in ws]
        slab = zeros(sum(sizes), device=dev)
        grads, off = [], 0
        for w, s in zip(ws, sizes):
            grads.append((slab[off:off + w.numel()].view_as(w), slab[off + w.numel():off + s]))
            off += s
        need_dx = ctx.needs_input_grad[0]
        dx = None
        for i in reversed(range(n)):
            N, K = ws[i].shape
            probs = [_wgrad(g, acts[i], grads[i][0], grads[i][1] if ctx.has_bias[i] else None, M, N, K)]
            if i > 0:       # the gradient of relu(z_{i-1}): acts[i] = relu(z_{i-1}) > 0 exactly where it passes
                d_in = torch.empty((M, K), device=dev)
                probs.append(_dgrad(g, ws[i], d_in, M, N, K, gate=acts[i], gate_scale=1.0))
            elif need_dx:
                d_in = torch.empty((M, K), device=dev)
                probs.append(_dgrad(g, ws[i], d_in, M, N, K))
            _gemm(probs, dy)
            if i > 0:
                g = d_in
            elif need_dx:
                dx = d_in.view(*dy.shape[:-1], K)
        out = [dx]
        for (dw, db), hb in zip(grads, ctx.has_bias):
            out += [dw, db if hb else None]
        return tuple(out)


def linear_relu_chain(seq, x):
    """``seq`` = nn.Sequential(Linear, ReLU, Linear, ReLU, ..., Linear) applied to the last dimension of ``x``."""
    lins = [m for m in seq if isinstance(m, torch.nn.Linear)]
    ok = (len(seq) == 2 * len(lins) - 1 and all(isinstance(m, torch.nn.ReLU) for m in list(seq)[1::2])
          and all(l.in_features % 4 == 0 and l.out_features % 4 == 0 for l in lins))
    if not ok or not x.is_cuda or x.dtype != torch.float32:
        return seq(x)
    x = x.contiguous()
    _check(x)
    params = []
    for l in lins:
        params += [l.weight, l.bias]
    return _LinearReluChain.apply(x, *params)


class _LinearF32(_LinearReluChain):
    """The same products with fp32 operands whatever ``set_compute_dtype`` says: the bf16 operating point is
    "bf16 attention / FFN" (BASELINE configs[3]); the two input embeddings that ``linear`` serves stay fp32 as
    their stock ``nn.Linear`` was."""

    @staticmethod
    def forward(ctx, x, *params):
        prev, _compute_bf16[0] = _compute_bf16[0], False
        try:
            return _LinearReluChain.forward(ctx, x, *params)
        finally:
            _compute_bf16[0] = prev

    @staticmethod
    @fold_scope
    def backward(ctx, dy):
        prev, _compute_bf16[0] = _compute_bf16[0], False
        try:
            return _LinearReluChain.backward(ctx, dy)
        finally:
            _compute_bf16[0] = prev


def linear(lin, x):
    """``nn.Linear`` on the grouped GEMM (bias in the epilogue; weight AND bias gradient from one split-K product
    with a ones column).  Besides the launch count this keeps torch's ``sum`` for the bias gradient out of the
    captured step: a multi-block torch reduction zeroes its semaphore with hipMemsetAsync, i.e. a MEMSET node
    (graph_audit.py)."""
    if (not x.is_cuda or x.dtype != torch.float32 or lin.in_features % 4 or lin.out_features % 4):
        return lin(x)
    x = x.contiguous()
    _check(x)
    return _LinearF32.apply(x, lin.weight, lin.bias)


def conv1x1(conv, x_pm):
    """``nn.Conv1d(kernel_size=1)`` applied to position-major rows ``x_pm`` (..., C_in) -> (..., C_out): the same
    product as ``conv(x.transpose(1, 2)).transpose(1, 2)`` on the grouped GEMM (weight and bias gradient from one
    split-K product), fp32 operands."""
    w = conv.weight.view(conv.out_channels, conv.in_channels)
    x_pm = x_pm.contiguous()
    _check(x_pm)
    return _LinearF32.apply(x_pm, w, conv.bias)


def _as_mask(mask):
    if mask is None:
        return None
    m = mask if mask.dtype == torch.bool else mask.bool()
    return m.contiguous()


def attention_block(attn, dropout, norm, residual, query, key, value, key_padding_mask=None):
    """LayerNorm(residual + Dropout(MHA(query, key, value)))."""
    training = attn.training
    p_attn = float(attn.dropout) if training else 0.0
    p_out = float(dropout.p) if (dropout is not None and dropout.training) else 0.0
    residual, query, key, value = (t.contiguous() for t in (residual, query, key, value))
    _check(residual, query, key, value)
    return _AttentionBlock.apply(
        residual, query, key, value, _as_mask(key_padding_mask),
        attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.weight, attn.out_proj.bias,
        norm.weight, norm.bias, attn.num_heads, float(norm.eps), p_attn, p_out,
        _next_site(), _next_site())


def ffn_block(ffn, norm, x):
    """LayerNorm(x + FFN(x)); ffn = Sequential(Linear, ReLU, Dropout, Linear, Dropout)."""
    lin1, _, drop1, lin2, drop2 = ffn
    p1 = float(drop1.p) if drop1.training else 0.0
    p2 = float(drop2.p) if drop2.training else 0.0
    x = x.contiguous()
    _check(x)
    return _FfnBlock.apply(x, lin1.weight, lin1.bias, lin2.weight, lin2.bias, norm.weight, norm.bias,
                           float(norm.eps), p1, p2, _next_site(), _next_site())


def multi_head_attention(attn, query, key, value, key_padding_mask=None):
    """Bare MHA (no residual / norm) -- not on the model's path; routed through the torch maths."""
    from . import attention_blocks
    return attention_blocks._mha_torch(attn, query, key, value, key_padding_mask)


# -------------------------------------------------------------------------------------------------
# The block as the model uses it: every call site of the encoder / decoder layers
# (encoder_decoder_layers.py:87-122,149-155,179-185,356-404) is one of
#     self-attention :  query = key = x (+ pos),  value = x,        residual = x
#     cross-attention:  query = x (+ pos),        key = value = memory, residual = x
# Told that structure, the backward returns the SUMMED gradients of x / pos / memory itself: the
# gradients of the attention core are written side by side (dq|dk|dv or dk|dv) so the input-projection
# gradients are single products over the concatenated contraction, and the residual-path gradient is
# accumulated by the GEMM epilogue (c_add / c2) -- autograd used to launch 2-4 tensor adds per block.
# -------------------------------------------------------------------------------------------------
def _xwgrad(dy, ldy, x, dw, db, M, N, K):
    """dw[N,K] += dy[M,N]^T @ x[M,K] (dy rows ldy floats apart), db[N] += column sums of dy."""
    split = max(1, min(256, M // 512)) if M >= 16384 else min(32, max(M // 256, min(8, M // 64), 1))
    return _slabbed(_problem(dy, x, dw, N, K, M, (1, ldy), (1, K), K, bias_grad=db, ones_col=True,
                             accumulate=True, split_k=split), dw, db)


class _XpmBlock(torch.autograd.Function):
    """LayerNorm(x + Dropout(MHA(x + pos, key, value))) with (key, value) = (x + pos, x) when
    ``mem is None`` and (mem, mem) otherwise."""

    @staticmethod
    def forward(ctx, x, pos, mem, mask, w_in, b_in, w_o, b_o, gamma, beta,
                num_heads, eps, p_attn, p_out, site_attn, site_out, xq_pre=None, next_pos=None):
        """-> y, or (y, y + next_pos) when ``next_pos`` is given (values only, written by the LayerNorm kernel).
        ``xq_pre``: x + pos as the previous block's LayerNorm kernel already wrote it."""
        B, Lq, E = x.shape
        H, D = num_heads, E // num_heads
        dev = x.device
        self_attn = mem is None
        scale = math.sqrt(1.0 / float(D))
        Mq = B * Lq
        xq = x if pos is None else xq_pre
        if xq is None:
            xq = x + pos
        xk, xv = (xq, x) if self_attn else (mem, mem)
        Lk = xk.shape[1]
        Mk = B * Lk
        q = torch.empty((B, Lq, E), device=dev)
        k = torch.empty((B, Lk, E), device=dev)
        v = torch.empty((B, Lk, E), device=dev)
        _gemm([_fwd(xq, w_in[:E], q, Mq, E, E, bias=b_in[:E], scale=scale),
               _fwd(xk, w_in[E:2 * E], k, Mk, E, E, bias=b_in[E:2 * E]),
               _fwd(xv, w_in[2 * E:], v, Mk, E, E, bias=b_in[2 * E:])], x)
        att = torch.empty((B, Lq, E), device=dev)
        lse = torch.empty((B, H, Lq), device=dev)
        with torch.cuda.device(dev):
            err = _attn_fwd()(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                          _ptr(mask), att.data_ptr(), lse.data_ptr(), p_attn, site_attn,
                                          rng_counter(dev).data_ptr(), _stream(x))
        _hiplib.check(err, "butd_attention_fwd")
        _count_attention("attn_fwd", B, H, Lq, Lk, D)
        proj = torch.empty((B, Lq, E), device=dev)
        y = torch.empty((B, Lq, E), device=dev)
        mean = torch.empty((Mq,), device=dev)
        rstd = torch.empty((Mq,), device=dev)
        y_pos = torch.empty((B, Lq, E), device=dev) if next_pos is not None else None
        _gemm([_fwd(att, w_o, proj, Mq, E, E, bias=b_o)], x)
        with torch.cuda.device(dev):
            err = _lib.butd_add_dropout_layernorm_fwd_pos(
                Mq, E, proj.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps,
                y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), p_out, site_out,
                rng_counter(dev).data_ptr(), _ptr(next_pos), _ptr(y_pos), _stream(x))
        _hiplib.check(err, "butd_add_dropout_layernorm_fwd_pos")
        ctx.save_for_backward(x, xq if pos is not None else None, mem, mask, w_in, w_o, gamma, q, k, v,
                              att, lse, proj, mean, rstd)
        ctx.cfg = (H, p_attn, p_out, site_attn, site_out)
        if y_pos is None:
            return y
        ctx.mark_non_differentiable(y_pos)
        ctx.set_materialize_grads(False)             # no zero tensor for the (non-existent) gradient of y + next_pos
        return y, y_pos

    @staticmethod
    @fold_scope
    def backward(ctx, dy, *_d_extras):
        if dy is None:                                 # (gradients are not materialised: the block's output unused)
            return (None,) * 18
        x, xq_saved, mem, mask, w_in, w_o, gamma, q, k, v, att, lse, proj, mean, rstd = ctx.saved_tensors
        H, p_attn, p_out, site_attn, site_out = ctx.cfg
        has_pos = xq_saved is not None
        xq = xq_saved if has_pos else x
        self_attn = mem is None
        B, Lq, E = x.shape
        Lk = Lq if self_attn else mem.shape[1]
        D = E // H
        Mq, Mk = B * Lq, B * Lk
        dev = x.device
        dy = dy.contiguous()
        slab = zeros(3 * E * E + 3 * E + E * E + E + 2 * E, device=dev)
        o = 0
        d_w_in = slab[o:o + 3 * E * E].view(3 * E, E); o += 3 * E * E
        d_b_in = slab[o:o + 3 * E]; o += 3 * E
        d_w_o = slab[o:o + E * E].view(E, E); o += E * E
        d_b_o = slab[o:o + E]; o += E
        d_gamma = slab[o:o + E]; o += E
        d_beta = slab[o:o + E]
        R = torch.empty((B, Lq, E), device=dev)              # d(residual path) -> total gradient of x
        d_proj = torch.empty((B, Lq, E), device=dev) if p_out > 0 else R
        fold = _ln_bwd(Mq, E, dy, proj, x, gamma, mean, rstd, d_proj, R, d_gamma, d_beta, p_out, site_out, x)
        d_att = torch.empty((B, Lq, E), device=dev)
        _gemm([_dgrad(d_proj, w_o, d_att, Mq, E, E),
               _wgrad(d_proj, att, d_w_o, d_b_o, Mq, E, E)] + fold, x)
        # gradients of the attention core, packed: self: G = [dq | dk | dv]; cross: dq, G = [dk | dv]
        short = _short_key_bwd(Lq, Lk, B, H, D)      # (one kernel that ACCUMULATES dk / dv: zero-filled from the step's arena)
        new_g = (lambda shape: zeros(shape, device=dev)) if short else (lambda shape: torch.empty(shape, device=dev))
        if self_attn:
            G = new_g((B, Lq, 3 * E))
            ldg = 3 * E
            dq_ptr, dk_ptr, dv_ptr = G.data_ptr(), G.data_ptr() + 4 * E, G.data_ptr() + 8 * E
            ld_dq = ldg
        else:
            dq = torch.empty((B, Lq, E), device=dev)
            G = new_g((B, Lk, 2 * E))
            ldg = 2 * E
            dq_ptr, dk_ptr, dv_ptr = dq.data_ptr(), G.data_ptr(), G.data_ptr() + 4 * E
            ld_dq = E
        scale = math.sqrt(1.0 / float(D))
        _attention_backward(B, H, Lq, Lk, D, q, k, v, mask, att, d_att, lse, dq_ptr, dk_ptr, dv_ptr, ld_dq, ldg, scale,
                            p_attn, site_attn, short, x)
        d_pos = d_mem = None
        if self_attn and has_pos:
            # d_pos = [dq|dk] W_in[:2E]  (also added into R);  R += dv Wv in a second launch (same target)
            d_pos = torch.empty((B, Lq, E), device=dev)
            _gemm([_problem(G, w_in, d_pos, Mq, E, 2 * E, (ldg, 1), (1, E), E, c2=R),
                   _xwgrad(G, ldg, xq, d_w_in[:2 * E], d_b_in[:2 * E], Mq, 2 * E, E)], x)
            Gv = G[:, :, 2 * E:]
            _gemm([_problem(Gv, w_in[2 * E:], R, Mq, E, E, (ldg, 1), (1, E), E, c_add=True),
                   _xwgrad(Gv, ldg, x, d_w_in[2 * E:], d_b_in[2 * E:], Mq, E, E)], x)
        elif self_attn:
            _gemm([_problem(G, w_in, R, Mq, E, 3 * E, (ldg, 1), (1, E), E, c_add=True),
                   _xwgrad(G, ldg, x, d_w_in, d_b_in, Mq, 3 * E, E)], x)
        else:
            d_mem = torch.empty((B, Lk, E), device=dev)
            if has_pos:
                d_pos = torch.empty((B, Lq, E), device=dev)
                qprob = _problem(dq, w_in, d_pos, Mq, E, E, (E, 1), (1, E), E, c2=R)
            else:
                qprob = _problem(dq, w_in, R, Mq, E, E, (E, 1), (1, E), E, c_add=True)
            _gemm([qprob,
                   _problem(G, w_in[E:], d_mem, Mk, E, 2 * E, (ldg, 1), (1, E), E),
                   _xwgrad(dq, E, xq, d_w_in[:E], d_b_in[:E], Mq, E, E),
                   _xwgrad(G, ldg, mem, d_w_in[E:], d_b_in[E:], Mk, 2 * E, E)], x)
        return (R, d_pos, d_mem, None, d_w_in, d_b_in, d_w_o, d_b_o, d_gamma, d_beta,
                None, None, None, None, None, None, None, None)


# -------------------------------------------------------------------------------------------------
# The decoder's memories (round 6).  Every BiDecoderLayer cross-attends to the SAME three tensors -- the encoder's text,
# box and seed features (bdetr.py:277-299, encoder_decoder_layers.py:376-395) -- so the key / value projections of all
# layers depend on the encoder output alone.  ``DecoderMemory.project`` computes the 2 x 3 x n_layers projections in a few
# grouped launches BEFORE the decoder (on a forked stream, next to the launch-bound query generation), and a layer's
# cross-attention block is then  Q projection (2048 rows) -> attention -> output projection -> LayerNorm.
# Backward mirrors it: every block writes its dK | dV side by side into ONE matrix per memory (row stride 2 E n_layers),
# and when autograd has passed the last block the hoisted node computes the memory-side input gradient of ALL layers as one
# product over the concatenated contraction ([dK_0 dV_0 ... dK_5 dV_5] . [W_0; ...; W_5]: no per-layer d_mem, no fan-in
# sum) and the K / V weight gradients of all layers in grouped launches.  The blocks and the hoisted node are tied
# together by a one-element token tensor (the autograd edge that orders them); the gradient matrices travel in the
# shared ``DecoderMemory`` object, not through autograd.  BUTD_AB=decoder_kv_hoist=0 restores the per-block projections.
# -------------------------------------------------------------------------------------------------
_hoist = [switches.flag("decoder_kv_hoist", True)]
_hoist_side = [switches.flag("decoder_kv_side", False)]   # measured, round 6: 21.35 ms forked, 21.17 on the main stream, 21.27 without the hoist (profiles/r06_decoder_hoist.txt)
_hoist_streams = {}
# problems per launch of the hoisted node (the library takes 32): measured in the step, 8 per launch 21.17 ms, everything
# of a kind in one launch (12 x 8192-row, 24 small, 21 backward problems) 21.26 -- a launch that large picks the wide
# tiles and its tail runs alone
_HOIST_GROUP = [int(os.environ.get("BUTD_HOIST_GROUP", "12"))]


def set_decoder_kv_hoist(flag, side=None):
    prev = (_hoist[0], _hoist_side[0])
    _hoist[0] = bool(flag)
    if side is not None:
        _hoist_side[0] = bool(side)
    return prev


class DecoderMemory:
    """What the hoisted node and the cross-attention blocks of one forward pass share."""

    def __init__(self, n_layers, E):
        self.n_layers, self.E = n_layers, E
        self.mem, self.kv, self.wcat = {}, {}, {}
        self.G, self.dw, self.dead = {}, {}, set()
        self.token = None
        self.side, self.joined = None, True

    def join(self, ref):
        """The first consumer on the main stream waits for the forked projections."""
        if not self.joined:
            torch.cuda.current_stream(ref.device).wait_stream(self.side)
            self.joined = True

    def grad_matrix(self, name):
        G = self.G.get(name)
        if G is None:
            B, Lk, _ = self.mem[name].shape
            G = self.G[name] = torch.empty((B, Lk, 2 * self.E * self.n_layers), device=self.mem[name].device)
        return G

    def weight_grads(self, layer, name, device):
        t = self.dw.get((layer, name))
        if t is None:
            E = self.E
            slab = zeros(3 * E * E + 3 * E, device=device)
            t = self.dw[(layer, name)] = (slab[:3 * E * E].view(3 * E, E), slab[3 * E * E:])
        return t

    def release(self):
        self.kv.clear(); self.G.clear(); self.dw.clear(); self.wcat.clear(); self.mem.clear()


class _HoistedKV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, shared, names, *tensors):
        """tensors = the memories (one per name), then per memory and layer (in_proj_weight, in_proj_bias)."""
        ctx.set_materialize_grads(False)
        nl, E = shared.n_layers, shared.E
        mems = tensors[:len(names)]
        params = tensors[len(names):]
        dev = mems[0].device
        main = torch.cuda.current_stream(dev)
        side = None
        if _hoist_side[0]:
            from . import graph_audit
            side = _hoist_streams.get(dev)
            if side is None:
                side = _hoist_streams[dev] = graph_audit.own_stream(dev, role="decoder.memory")
            side.wait_stream(main)
        groups = []
        outs = []
        for j, (name, mem) in enumerate(zip(names, mems)):
            B, Lk, _ = mem.shape
            Mk = B * Lk
            buf = torch.empty((nl, 2, B, Lk, E), device=dev)
            outs.append(buf)
            probs = []
            for i in range(nl):
                w, b = params[2 * (j * nl + i)], params[2 * (j * nl + i) + 1]
                probs.append(_fwd(mem, w[E:2 * E], buf[i, 0], Mk, E, E, bias=b[E:2 * E]))
                probs.append(_fwd(mem, w[2 * E:], buf[i, 1], Mk, E, E, bias=b[2 * E:]))
                shared.kv[(i, name)] = (buf[i, 0], buf[i, 1])
            shared.mem[name] = mem
            groups.append((Mk, probs))
        # the launches: equal-sized problems together (one tile rule per launch), the small memories first (the first
        # cross-attention of layer 0 reads the text projections)
        groups.sort(key=lambda g: g[0])
        queue, launches = [], []
        for Mk, probs in groups:
            if Mk >= 4096:                      # a launch of its own kind
                if queue:
                    launches += [queue[k:k + _HOIST_GROUP[0]] for k in range(0, len(queue), _HOIST_GROUP[0])]
                    queue = []
                launches += [probs[k:k + _HOIST_GROUP[0]] for k in range(0, len(probs), _HOIST_GROUP[0])]
            else:
                queue += probs
        if queue:
            launches += [queue[k:k + _HOIST_GROUP[0]] for k in range(0, len(queue), _HOIST_GROUP[0])]
        with torch.cuda.stream(side) if side is not None else _null():
            for probs in launches:
                _gemm(probs, mems[0])
        if mems[0].requires_grad or any(p.requires_grad for p in params):
            # the stacked K | V weights of a memory, for the one-product input gradient of the backward pass
            with torch.cuda.stream(side) if side is not None else _null():
                for j, name in enumerate(names):
                    shared.wcat[name] = torch.cat([params[2 * (j * nl + i)][E:] for i in range(nl)])
        shared.side, shared.joined = side, side is None
        ctx.shared, ctx.names = shared, names
        ctx.save_for_backward(*tensors)
        return torch.empty(1, device=dev)       # the autograd edge between this node and the blocks (no data)

    @staticmethod
    @fold_scope
    def backward(ctx, _d_token):
        shared, names = ctx.shared, ctx.names
        tensors = ctx.saved_tensors
        nl, E = shared.n_layers, shared.E
        mems, params = tensors[:len(names)], tensors[len(names):]
        dev = mems[0].device
        ld = 2 * E * nl
        d_mems, first, rest = [], [], []
        grads = []
        for j, (name, mem) in enumerate(zip(names, mems)):
            B, Lk, _ = mem.shape
            Mk = B * Lk
            G = shared.G.get(name)
            if G is None:                                   # no block of this memory had a gradient
                d_mems.append(None)
                grads += [None] * (2 * nl)
                continue
            for i in range(nl):
                if (i, name) in shared.dead:
                    G[:, :, 2 * E * i:2 * E * (i + 1)].zero_()
            G2 = G.view(Mk, ld)
            d_mem = torch.empty((B, Lk, E), device=dev)
            d_mems.append(d_mem)
            first.append(_problem(G2, shared.wcat[name], d_mem, Mk, E, ld, (ld, 1), (1, E), E))
            for i in range(nl):
                dw, db = shared.weight_grads(i, name, dev)
                rest.append(_xwgrad(G2[:, 2 * E * i:2 * E * (i + 1)], ld, mem, dw[E:], db[E:], Mk, 2 * E, E))
                grads += [dw, db]
        # the memory-side input gradients first (the encoder's backward waits for them), the weight gradients after them
        probs = first + rest
        for k in range(0, len(probs), _HOIST_GROUP[0]):
            _gemm(probs[k:k + _HOIST_GROUP[0]], mems[0])
        shared.release()
        return (None, None) + tuple(d_mems) + tuple(grads)


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def decoder_memory(layers, memories):
    """``layers``: the BiDecoderLayers; ``memories``: [(name, attention-module attribute, tensor (B, Lk, E) | None)].
    -> DecoderMemory (its token is the autograd edge) or None when the hoist does not apply."""
    mems = [(n, a, t) for n, a, t in memories if t is not None]
    if not _hoist[0] or not mems or not mems[0][2].is_cuda:
        return None
    E = mems[0][2].shape[-1]
    shared = DecoderMemory(len(layers), E)
    tensors = [t.contiguous() for _, _, t in mems]
    _check(*tensors)
    params = []
    for _, attr, _ in mems:
        for layer in layers:
            attn = getattr(layer, attr)
            params += [attn.in_proj_weight, attn.in_proj_bias]
    shared.token = _HoistedKV.apply(shared, tuple(n for n, _, _ in mems), *tensors, *params)
    return shared


class _XhoistBlock(torch.autograd.Function):
    """LayerNorm(x + Dropout(MHA(x + pos, K, V))) with K, V = a layer's hoisted projections of a decoder memory."""

    @staticmethod
    def forward(ctx, x, pos, token, mask, w_in, b_in, w_o, b_o, gamma, beta, num_heads, eps, p_attn, p_out, site_attn,
                site_out, xq_pre, next_pos, shared, layer, name):
        B, Lq, E = x.shape
        H, D = num_heads, E // num_heads
        dev = x.device
        scale = math.sqrt(1.0 / float(D))
        Mq = B * Lq
        xq = x if pos is None else xq_pre
        if xq is None:
            xq = x + pos
        shared.join(x)
        k, v = shared.kv[(layer, name)]
        Lk = k.shape[1]
        q = torch.empty((B, Lq, E), device=dev)
        _gemm([_fwd(xq, w_in[:E], q, Mq, E, E, bias=b_in[:E], scale=scale)], x)
        att = torch.empty((B, Lq, E), device=dev)
        lse = torch.empty((B, H, Lq), device=dev)
        with torch.cuda.device(dev):
            err = _attn_fwd()(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(),
                              _ptr(mask), att.data_ptr(), lse.data_ptr(), p_attn, site_attn,
                              rng_counter(dev).data_ptr(), _stream(x))
        _hiplib.check(err, "butd_attention_fwd")
        _count_attention("attn_fwd", B, H, Lq, Lk, D)
        proj = torch.empty((B, Lq, E), device=dev)
        y = torch.empty((B, Lq, E), device=dev)
        mean = torch.empty((Mq,), device=dev)
        rstd = torch.empty((Mq,), device=dev)
        y_pos = torch.empty((B, Lq, E), device=dev) if next_pos is not None else None
        _gemm([_fwd(att, w_o, proj, Mq, E, E, bias=b_o)], x)
        with torch.cuda.device(dev):
            err = _lib.butd_add_dropout_layernorm_fwd_pos(
                Mq, E, proj.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps,
                y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), p_out, site_out,
                rng_counter(dev).data_ptr(), _ptr(next_pos), _ptr(y_pos), _stream(x))
        _hiplib.check(err, "butd_add_dropout_layernorm_fwd_pos")
        ctx.save_for_backward(x, xq if pos is not None else None, mask, w_in, w_o, gamma, q, k, v, att, lse, proj, mean, rstd)
        ctx.cfg = (H, p_attn, p_out, site_attn, site_out, shared, layer, name)
        if y_pos is None:
            return y
        ctx.mark_non_differentiable(y_pos)
        ctx.set_materialize_grads(False)
        return y, y_pos

    @staticmethod
    @fold_scope
    def backward(ctx, dy, *_d_extras):
        H, p_attn, p_out, site_attn, site_out, shared, layer, name = ctx.cfg
        if dy is None:
            shared.dead.add((layer, name))
            return (None,) * 21
        x, xq_saved, mask, w_in, w_o, gamma, q, k, v, att, lse, proj, mean, rstd = ctx.saved_tensors
        has_pos = xq_saved is not None
        xq = xq_saved if has_pos else x
        B, Lq, E = x.shape
        Lk = k.shape[1]
        D = E // H
        Mq = B * Lq
        dev = x.device
        dy = dy.contiguous()
        slab = zeros(E * E + E + 2 * E, device=dev)
        d_w_o = slab[:E * E].view(E, E)
        d_b_o = slab[E * E:E * E + E]
        d_gamma = slab[E * E + E:E * E + 2 * E]
        d_beta = slab[E * E + 2 * E:]
        R = torch.empty((B, Lq, E), device=dev)
        d_proj = torch.empty((B, Lq, E), device=dev) if p_out > 0 else R
        fold = _ln_bwd(Mq, E, dy, proj, x, gamma, mean, rstd, d_proj, R, d_gamma, d_beta, p_out, site_out, x)
        d_att = torch.empty((B, Lq, E), device=dev)
        _gemm([_dgrad(d_proj, w_o, d_att, Mq, E, E),
               _wgrad(d_proj, att, d_w_o, d_b_o, Mq, E, E)] + fold, x)
        # dK | dV of this layer go into the memory's gradient matrix, at this layer's columns
        G = shared.grad_matrix(name)
        ld = 2 * E * shared.n_layers
        short = _short_key_bwd(Lq, Lk, B, H, D)
        dq = torch.empty((B, Lq, E), device=dev)
        base = G.data_ptr() + 4 * (2 * E * layer)
        if short:           # (the accumulating kernel of other head dimensions: through a zero-filled staging matrix)
            tmp = zeros((B, Lk, 2 * E), device=dev)
            dk_ptr, dv_ptr, ld_dkv = tmp.data_ptr(), tmp.data_ptr() + 4 * E, 2 * E
        else:
            dk_ptr, dv_ptr, ld_dkv = base, base + 4 * E, ld
        scale = math.sqrt(1.0 / float(D))
        _attention_backward(B, H, Lq, Lk, D, q, k, v, mask, att, d_att, lse, dq.data_ptr(), dk_ptr, dv_ptr, E, ld_dkv,
                            scale, p_attn, site_attn, short, x)
        if short:
            G[:, :, 2 * E * layer:2 * E * (layer + 1)].copy_(tmp)
        d_w_in, d_b_in = shared.weight_grads(layer, name, dev)
        d_pos = None
        if has_pos:
            d_pos = torch.empty((B, Lq, E), device=dev)
            qprob = _problem(dq, w_in, d_pos, Mq, E, E, (E, 1), (1, E), E, c2=R)
        else:
            qprob = _problem(dq, w_in, R, Mq, E, E, (E, 1), (1, E), E, c_add=True)
        _gemm([qprob, _xwgrad(dq, E, xq, d_w_in[:E], d_b_in[:E], Mq, E, E)], x)
        return (R, d_pos, None, None, None, None, d_w_o, d_b_o, d_gamma, d_beta) + (None,) * 11


def block(attn, dropout, norm, x, pos=None, memory=None, key_padding_mask=None, xq_pre=None, next_pos=None, ffn=None,
          hoisted=None):
    """LayerNorm(x + Dropout(MHA(x + pos, k, v))), (k, v) = (x + pos, x) or (memory, memory).
    ``next_pos``: also return ``y + next_pos`` (written by the kernel that produced y; no gradient flows through it) for
    the next block, which takes it as ``xq_pre`` in place of computing ``x + pos`` itself.
    ``ffn`` = (ffn Sequential, norm): the FFN block on y behind it (then ``next_pos`` belongs to the FFN's output).
    -> y                       (neither asked for)
    -> (y, y + next_pos | None)   otherwise; with ``ffn`` the first element is LayerNorm(y + FFN(y)).
    (Round 4's row-panel chain kernels -- out-projection + LayerNorm + the next blocks' projections + FFN as one launch --
    measured 0.9 ms slower in the step and are gone: profiles/r04_panel_chain.txt)"""
    training = attn.training
    p_attn = float(attn.dropout) if training else 0.0
    p_out = float(dropout.p) if (dropout is not None and dropout.training) else 0.0
    x = x.contiguous()
    pos = None if pos is None else pos.contiguous()
    memory = None if memory is None else memory.contiguous()
    next_pos = None if next_pos is None else next_pos.detach().contiguous()
    xq_pre = None if (xq_pre is None or pos is None) else xq_pre.detach()
    _check(x, pos, memory)
    extras_asked = next_pos is not None or ffn is not None
    blk_pos = None if ffn is not None else next_pos
    ffn_sites = (_next_site(), _next_site()) if ffn is not None else None     # (numbered before the block's own, as ever)
    if hoisted is not None:         # (DecoderMemory, layer index, memory name): K / V of ``memory`` are already projected
        shared, layer, name = hoisted
        out = _XhoistBlock.apply(x, pos, shared.token, _as_mask(key_padding_mask),
                                 attn.in_proj_weight.detach(), attn.in_proj_bias.detach(), attn.out_proj.weight,
                                 attn.out_proj.bias, norm.weight, norm.bias, attn.num_heads, float(norm.eps), p_attn,
                                 p_out, _next_site(), _next_site(), xq_pre, blk_pos, shared, layer, name)
    else:
        out = _XpmBlock.apply(x, pos, memory, _as_mask(key_padding_mask),
                              attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.weight, attn.out_proj.bias,
                              norm.weight, norm.bias, attn.num_heads, float(norm.eps), p_attn, p_out,
                              _next_site(), _next_site(), xq_pre, blk_pos)
    y, y_pos = out if isinstance(out, tuple) else (out, None)
    if ffn is not None:
        seq, norm2 = ffn
        lin1, _, drop1, lin2, drop2 = seq
        p1 = float(drop1.p) if drop1.training else 0.0
        p2 = float(drop2.p) if drop2.training else 0.0
        r = _FfnBlock.apply(y, lin1.weight, lin1.bias, lin2.weight, lin2.bias, norm2.weight, norm2.bias,
                            float(norm2.eps), p1, p2, *ffn_sites, next_pos)
        y, y_pos = r if isinstance(r, tuple) else (r, None)
    return (y, y_pos) if extras_asked else y
