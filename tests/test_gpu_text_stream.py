"""-m gpu: the text stream (butd_detr_amd/text_stream.py; reference call models/bdetr.py:164-173) on the device --
bf16 language-model arithmetic against fp32, and the utterance cache under the hipGraph training step."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_bf16_language_model_tracks_fp32():
    from butd_detr_amd import text_stream
    from butd_detr_amd.offline_text import HashTokenizer, random_roberta_base, synthetic_utterances
    enc = random_roberta_base(0).cuda().eval()
    texts = synthetic_utterances(8, tokens=80, seed=3)
    tok = HashTokenizer().batch_encode_plus(texts).to("cuda")
    with torch.no_grad():
        a = text_stream.run_language_model(enc, tok, "f32")
        b = text_stream.run_language_model(enc, tok, "bf16")
    assert b.dtype == torch.float32
    err = (a - b).abs().max() / a.abs().max()
    cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0)
    assert err < 5e-2 and cos > 0.999, (float(err), float(cos))
    # the cache on the device: other batch compositions equal a fresh fp32 pass
    cache = text_stream.UtteranceCache("cuda", rows=256)
    with torch.no_grad():
        text_stream.encode(enc, tok, texts, cache)
        pick = [texts[5], texts[2], texts[7]]
        t2 = HashTokenizer().batch_encode_plus(pick).to("cuda")
        got = text_stream.encode(enc, t2, pick, cache)
        want = enc(**t2).last_hidden_state
    assert cache.hits == 1
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)


def test_graph_step_with_the_cache_trains_like_the_step_that_runs_the_language_model():
    """cache_in_training=True moves the language model out of the graphs (hit or miss is host control flow): the
    losses over rotating batches must equal those of the step that encodes every batch, the second pass over the
    data must not run RoBERTa at all, and an unannounced batch must still get ITS text."""
    from tests.test_gpu_graph_step import _model
    from butd_detr_amd import attention_blocks, text_stream
    from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, synthetic_batch
    try:
        dev = torch.device("cuda", 0)
        batches = [synthetic_batch(2, dev, seed=500 + 13 * i, n_points=4096, tokens=24) for i in range(3)]
        ref_model = _model()                      # every dropout p = 0: the tower is deterministic in both runs
        cached_model = copy.deepcopy(ref_model)
        cached_model.text_cache = text_stream.UtteranceCache(dev, rows=256, cache_in_training=True)
        calls = []
        cached_model.text_encoder.register_forward_hook(lambda *a: calls.append(1))
        ref = GraphedTrainStep(ref_model, FlatAdamW(ref_model, lr=0.0, lr_backbone=0.0), warmup=1)
        cached = GraphedTrainStep(cached_model, FlatAdamW(cached_model, lr=0.0, lr_backbone=0.0), warmup=1)
        assert cached.text_outside and not ref.text_outside
        order = [0, 1, 2, 0, 1, 2, 1, 0]
        ref_losses, got_losses, calls_at = [], [], []
        for k, i in enumerate(order):
            nxt = batches[order[k + 1]][0] if k + 1 < len(order) and k != 5 else None   # step 6 arrives unannounced
            ref_losses.append(float(ref(*batches[i], next_inputs=nxt)))
            got_losses.append(float(cached(*batches[i], next_inputs=nxt)))
            calls_at.append(len(calls))
        for a, b in zip(got_losses, ref_losses):
            assert abs(a - b) <= 1e-4 * max(abs(b), 1.0), (got_losses, ref_losses)
        assert len({round(l, 3) for l in ref_losses[:3]}) == 3                  # the batches do differ
        assert calls_at[-1] == calls_at[2], calls_at                              # nothing encoded after the first pass
        assert len(cached_model.text_cache) == 6 and cached_model.text_cache.hits >= 5
    finally:
        attention_blocks.set_backend("torch")
