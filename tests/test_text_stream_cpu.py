"""not gpu: the utterance cache of the text stream (butd_detr_amd/text_stream.py) rebuilds, for ANY batch
composition, the hidden states the language model itself returns (reference call: models/bdetr.py:164-167) --
valid rows and padded positions -- and stays out of the way where it may not change the arithmetic."""
import pytest
import torch

from butd_detr_amd import text_stream
from butd_detr_amd.offline_text import HashTokenizer


def _tiny_roberta(seed=0):
    from transformers import RobertaConfig, RobertaModel
    cfg = RobertaConfig(vocab_size=50265, max_position_embeddings=64, type_vocab_size=1, hidden_size=32,
                        num_hidden_layers=2, num_attention_heads=4, intermediate_size=64, pad_token_id=1,
                        bos_token_id=0, eos_token_id=2)
    torch.manual_seed(seed)
    return RobertaModel(cfg, add_pooling_layer=False).eval()


UTTERANCES = ["the chair next to the window", "a lamp", "brown table in the middle of the room near the sofa",
              "door", "the second monitor from the left on the long desk by the wall", "small bin under it"]


def test_cache_rebuilds_any_batch_composition_including_padded_positions():
    enc, tok = _tiny_roberta(), HashTokenizer()
    cache = text_stream.UtteranceCache("cpu", hidden=32, rows=8)           # tiny: has to grow
    first = [UTTERANCES[i] for i in (0, 1, 2)]
    second = [UTTERANCES[i] for i in (3, 4, 5)]
    for texts in (first, second):
        t = tok.batch_encode_plus(texts)
        got = text_stream.encode(enc, t, texts, cache)
        with torch.no_grad():
            want = enc(**t).last_hidden_state
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    assert len(cache) == 6 and cache.misses == 2 and cache.hits == 0
    assert cache.used == sum(len(u.split()) + 2 + 1 for u in UTTERANCES)   # n tokens + one pad row each
    # other groupings / orders / paddings: served from the store, equal to a fresh pass of the language model
    for pick in ((1, 4), (5, 3, 1, 0), (2,), (3, 3, 1)):
        texts = [UTTERANCES[i] for i in pick]
        t = tok.batch_encode_plus(texts)
        before = cache.hits
        got = text_stream.encode(enc, t, texts, cache)
        assert cache.hits == before + 1
        with torch.no_grad():
            want = enc(**t).last_hidden_state
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)       # padded positions too
    # a batch with one unknown utterance runs the model for the batch and admits the newcomer
    texts = [UTTERANCES[0], "something never seen"]
    t = tok.batch_encode_plus(texts)
    got = text_stream.encode(enc, t, texts, cache)
    with torch.no_grad():
        want = enc(**t).last_hidden_state
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    assert len(cache) == 7


def test_cache_is_bypassed_in_training_unless_opted_in_and_rows_are_dropout_free():
    enc, tok = _tiny_roberta(), HashTokenizer()
    texts = UTTERANCES[:3]
    t = tok.batch_encode_plus(texts)
    cache = text_stream.UtteranceCache("cpu", hidden=32)
    enc.train()                                                            # model.train() reaches the frozen tower
    text_stream.encode(enc, t, texts, cache, training=True)
    assert len(cache) == 0                                                 # the reference's dropout noise is kept
    opted = text_stream.UtteranceCache("cpu", hidden=32, cache_in_training=True)
    a = text_stream.encode(enc, t, texts, opted, training=True)
    assert enc.training and len(opted) == 3
    b = text_stream.encode(enc, t, texts, opted, training=True)
    enc.eval()
    with torch.no_grad():
        want = enc(**t).last_hidden_state
    torch.testing.assert_close(a, want, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(b, want, rtol=1e-5, atol=1e-5)


def test_capacity_limit_and_bad_precision():
    enc, tok = _tiny_roberta(), HashTokenizer()
    cache = text_stream.UtteranceCache("cpu", hidden=32, rows=8, max_rows=16)
    for texts in (UTTERANCES[:2], UTTERANCES[2:4]):
        text_stream.encode(enc, tok.batch_encode_plus(texts), texts, cache)
    assert cache.used <= 16 and len(cache) < 4                             # full: newcomers are not admitted ...
    texts = UTTERANCES[2:4]
    got = text_stream.encode(enc, tok.batch_encode_plus(texts), texts, cache)
    with torch.no_grad():
        want = enc(**tok.batch_encode_plus(texts)).last_hidden_state
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)           # ... and results stay right
    with pytest.raises(ValueError):
        text_stream.run_language_model(enc, tok.batch_encode_plus(texts), precision="fp8")


def test_model_eval_forward_uses_the_cache():
    """BeaUTyDETR.encode_text / forward with ``text_cache`` set: second pass over the same utterances does not
    call the language model."""
    import warnings
    from butd_detr_amd.bdetr import BeaUTyDETR
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = BeaUTyDETR(num_queries=8, num_decoder_layers=1, num_encoder_layers=1,
                       text_encoder_factory=lambda: (HashTokenizer(), _tiny_roberta()), d_model=288).eval()
    calls = []
    m.text_encoder.register_forward_hook(lambda *a: calls.append(1))
    m.text_cache = text_stream.UtteranceCache("cpu", hidden=32)
    inputs = {"text": UTTERANCES[:2], "point_clouds": torch.zeros(2, 16, 6)}
    tok = m.tokenize(inputs)
    a = m.encode_text(tok, inputs["text"])
    b = m.encode_text(tok, inputs["text"])
    assert calls == [1]
    torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
