p='butd_detr_amd/fused_attention.py'
s=open(p).read()
# --- simplify: drop xq2/xk2
s=s.replace('''    def forward(ctx, residual, xq, xq2, xk, xk2, xv, mask, w_in, b_in, w_o, b_o, gamma, beta,
                num_heads, eps, p_attn, p_out, site_attn, site_out):''','''    def forward(ctx, residual, xq, xk, xv, mask, w_in, b_in, w_o, b_o, gamma, beta,
                num_heads, eps, p_attn, p_out, site_attn, site_out):''')
s=s.replace('''        _gemm([_fwd(xq, w_in[:E], q, Mq, E, E, a2=xq2, bias=b_in[:E], scale=scale),
               _fwd(xk, w_in[E:2 * E], k, Mk, E, E, a2=xk2, bias=b_in[E:2 * E]),''','''        _gemm([_fwd(xq, w_in[:E], q, Mq, E, E, bias=b_in[:E], scale=scale),
               _fwd(xk, w_in[E:2 * E], k, Mk, E, E, bias=b_in[E:2 * E]),''')
s=s.replace('''        ctx.save_for_backward(residual, xq, xq2, xk, xk2, xv, mask, w_in, w_o, gamma, q, k, v, att, lse,
                              proj, mean, rstd)''','''        ctx.save_for_backward(residual, xq, xk, xv, mask, w_in, w_o, gamma, q, k, v, att, lse, proj,
                              mean, rstd)''')
s=s.replace('''        (residual, xq, xq2, xk, xk2, xv, mask, w_in, w_o, gamma, q, k, v, att, lse, proj, mean,
         rstd) = ctx.saved_tensors''','''        (residual, xq, xk, xv, mask, w_in, w_o, gamma, q, k, v, att, lse, proj, mean,
         rstd) = ctx.saved_tensors''')
a=s.index('        _gemm([_wgrad(dq, xq, d_w_in[:E], d_b_in[:E], Mq, E, E, a2=None, scale=scale),')
b=s.index('class _FfnBlock')
s=s[:a]+'''        _gemm([_wgrad(dq, xq, d_w_in[:E], d_b_in[:E], Mq, E, E, scale=scale),
               _wgrad(dk, xk, d_w_in[E:2 * E], d_b_in[E:2 * E], Mk, E, E),
               _wgrad(dv, xv, d_w_in[2 * E:], d_b_in[2 * E:], Mk, E, E)], xq)
        return (d_res, d_xq, d_xk, d_xv, None, d_w_in, d_b_in, d_w_o, d_b_o, d_gamma, d_beta,
                None, None, None, None, None, None)


'''+s[b:]
s=s.replace('''        d_x = torch.empty((B, L, E), device=dev)          # residual path first, FFN path accumulates
        d_o = torch.empty((B, L, E), device=dev) if p2 > 0 else d_x''','''        d_x = torch.empty((B, L, E), device=dev)          # residual path first, FFN path accumulates
        d_o = torch.empty((B, L, E), device=dev)''')
s=s.replace('''        if p2 == 0:                                         # d_o aliases d_x: keep a private copy
            d_o = d_x.clone()
''','')
s=s.replace('''def attention_block(attn, dropout, norm, residual, query, key, value, key_padding_mask=None,
                    query_pos=None, key_pos=None):
    """LayerNorm(residual + Dropout(MHA(query (+query_pos), key (+key_pos), value)))."""''','''def attention_block(attn, dropout, norm, residual, query, key, value, key_padding_mask=None):
    """LayerNorm(residual + Dropout(MHA(query, key, value)))."""''')
s=s.replace('''    _check(residual, query, key, value, query_pos, key_pos)
    return _AttentionBlock.apply(
        residual, query, query_pos, key, key_pos, value, _as_mask(key_padding_mask),''','''    _check(residual, query, key, value)
    return _AttentionBlock.apply(
        residual, query, key, value, _as_mask(key_padding_mask),''')
open(p,'w').write(s)
