"""ONE A/B hook for the kernel paths of the product instead of one environment variable per path.

    BUTD_AB="sa_last_bwd=0,ln_fold=0" python bench.py ...        (scratch/ab_env.sh alternates two such settings)

Every name below is a path that is ON in the product and whose predecessor is kept for the parity tests (which flip the
module-level flag directly) and for committed A/B measurements under profiles/.  Paths that measured slower in the step
(round 4: row-panel chains, forked decoder key / value projections, forked prediction heads, the short-key attention
backward at every site) are not switches any more -- they were deleted in round 5; their measurements are in
profiles/r04_panel_chain.txt, r04_side_branches.txt, r04_attention_short_keys.txt.  Paths settled for two rounds keep their
module-level flag for the parity tests but left this hook (the short-key attention backward -- now only the fallback of
other head dimensions --, the Conv+BN prologue fold, the inverted-list gather of SA2-4, the one-kernel inference level).
"""
import os

KNOWN = {
    "attn_long_keys":   "one-pass attention backward, butd_attention_bwd_long_keys (fused_attention)",
    "ln_fold":          "LayerNorm backward: dgamma / dbeta partials folded by the next grouped launch (fused_attention)",
    "wgrad_slabs":      "(default OFF) deterministic split-K weight gradients: partial slabs + ride-along folds instead of float atomics; bit-reproducible, +0.8 ms per step (fused_attention)",
    "mlp_fuse_stats":   "Conv+BN chains backward: gate + BatchNorm sums in the producing product's epilogue (fused_mlp)",
    "sa_last_bwd":      "set abstraction: last layer + max-pool backward by linearity (fused_sa)",
    "sa_last_fwd":      "set abstraction: last layer forward without writing Z3 (fused_sa)",
    "sa_first_bwd":     "SA1: first layer backward without dZ1 (fused_sa)",
    "sa_mid_bwd":       "SA1: layer 2 + layer 1 backward in one pass (fused_sa)",
    "sa_mid_wide":      "SA2-4: layer 2 backward in one pass, only the gated gradient written (fused_sa)",
    "sa_first_linear":  "SA2-4: the first layer by linearity of the grouping -- product over the level's N points, Z1 gathered, backward along the inverted lists; no grouped input (fused_sa)",
    "sa_no_z1":         "SA1: forward without Z1 (fused_sa)",
    "sa_fuse_stats":    "SA2-4: layer 1's gate + BatchNorm sums in the epilogue of the product that writes dH1 (fused_sa)",
    "decoder_kv_hoist": "decoder: the memories' key / value projections of all layers before the decoder, their input / weight gradients after it (fused_attention.DecoderMemory)",
    "decoder_kv_side":  "... with the forward projections on a forked stream next to the query generation (fused_attention)",
    "fan_out":          "gradient fan-in of multiply-used tensors as one launch (fan_out)",
    "encoder_fork":     "language side of an encoder layer on a forked stream (encoder_decoder_layers)",
    "text_overlap":     "frozen language model on a forked stream when it is not prefetched (bdetr)",
}


def _parse():
    out = {}
    for item in os.environ.get("BUTD_AB", "").replace(";", ",").split(","):
        item = item.strip()
        if not item:
            continue
        name, _, value = item.partition("=")
        if name not in KNOWN:
            raise ValueError(f"BUTD_AB: unknown switch {name!r} (known: {', '.join(sorted(KNOWN))})")
        out[name] = value.strip().lower() not in ("0", "false", "off", "no", "")
    return out


_OVERRIDES = _parse()


def flag(name, default=True):
    """The product default unless BUTD_AB overrides it (read once, at import)."""
    assert name in KNOWN, name
    return _OVERRIDES.get(name, default)
