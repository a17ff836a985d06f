"""Prompt-weighting host code vs vectors produced by the reference's own functions (tests/golden)."""
import json

import numpy as np
import torch

from gyre_amd import text as T


def test_parse_prompt_attention_matches_reference(golden):
    prompts = json.loads(str(golden["lpw_prompts"]))
    want = json.loads(str(golden["lpw_parsed"]))
    for ptxt, w in zip(prompts, want):
        got = T.parse_prompt_attention(ptxt)
        assert [g[0] for g in got] == [x[0] for x in w], (ptxt, got, w)
        assert np.allclose([g[1] for g in got], [x[1] for x in w], rtol=1e-12), (ptxt, got, w)


def test_pad_tokens_and_weights_matches_reference(golden):
    toks = [[5, 6, 7], [], list(range(100, 180))]
    wts = [[1.1, 1.0, 0.5], [], [1.0 + 0.01 * i for i in range(80)]]
    for nbm in (True, False):
        t2, w2 = T.pad_tokens_and_weights(toks, wts, 152, 49406, 49407, no_boseos_middle=nbm, chunk_length=77)
        assert np.array_equal(np.array(t2), golden[f"lpw_pad_tokens_nbm{int(nbm)}"])
        assert np.allclose(np.array(w2), golden[f"lpw_pad_weights_nbm{int(nbm)}"], rtol=0, atol=0)


def test_token_weighting_matches_reference(golden):
    out = T.apply_token_weights(torch.from_numpy(golden["lpw_weighting_in"]), torch.from_numpy(golden["lpw_weighting_w"]))
    assert np.allclose(out.numpy(), golden["lpw_weighting_out"], rtol=1e-6, atol=1e-7)


def test_lpw_embedder_long_prompt_chunks():
    calls = []

    def enc(ids):
        calls.append(tuple(ids.shape))
        return torch.nn.functional.one_hot(ids % 8, 8).float() + 0.1

    emb = T.LPWTextEmbedder(enc, lambda frag: [ord(c) for c in frag if c != " "], max_embeddings_multiples=3)
    cond, unc = emb.get_embeddings(["a (b:2) " + "c" * 90], [""])
    assert cond.shape == (1, 152, 8) and unc.shape == (1, 152, 8)   # 2 chunks of 75 + BOS/EOS
    assert calls == [(1, 77), (1, 77), (1, 77), (1, 77)]
    short, _ = emb.get_embeddings(["(x:1.5) y"])
    assert short.shape == (1, 77, 8)


def test_clip_alt_layer():
    """TextEncoderAltLayer semantics (reference text_encoder_alt_layer.py:6-36): 'final' = last_hidden_state,
    'penultimate' / n = final LayerNorm of an earlier hidden state."""
    import torch
    from gyre_amd.text import ClipTextEncoder, synthetic_prompt_ids
    ids = synthetic_prompt_ids(2, seed=3)
    enc = ClipTextEncoder.synthetic("cpu", hidden=64, layers=3, heads=4, seed=1)
    with torch.no_grad():
        full = enc.model(input_ids=ids, output_hidden_states=True, return_dict=True)
    ln = enc._final_layer_norm()
    assert torch.allclose(enc(ids), full.last_hidden_state.float())
    for layer, idx in (("penultimate", -2), (2, -2), (3, -3), (1, -1)):
        e = ClipTextEncoder(enc.model, "cpu", layer=layer)
        with torch.no_grad():
            want = ln(full.hidden_states[idx]).float()
        assert torch.allclose(e(ids), want, atol=1e-6), layer
    assert torch.allclose(ClipTextEncoder(enc.model, "cpu", layer=1)(ids), enc(ids), atol=1e-5)   # last state + LN == final
    import pytest
    with pytest.raises(ValueError):
        ClipTextEncoder(enc.model, "cpu", layer="first")


def _tiny_sdxl_towers():
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    torch.manual_seed(0)
    kw = dict(vocab_size=300, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2, max_position_embeddings=77,
              bos_token_id=298, eos_token_id=299, pad_token_id=299)
    te1 = CLIPTextModel(CLIPTextConfig(hidden_size=32, **kw)).eval()
    te2 = CLIPTextModelWithProjection(CLIPTextConfig(hidden_size=48, projection_dim=40, **kw)).eval()
    return te1, te2


def test_sdxl_conditioner_follows_the_published_scheme():
    """SDXL conditioning (extension, BASELINE configs[3]; parity unpinned - the scheme is restated here from the published
    pipeline): penultimate hidden states of both towers concatenated, pooled projection of the second, zeros for an empty
    negative prompt, time_ids = (orig h, w, crop top, left, target h, w)."""
    te1, te2 = _tiny_sdxl_towers()
    tok = lambda text: [3 + (ord(c) % 200) for c in text if c != " "]
    cnd = T.SDXLTextConditioner(te1, tok, te2, tok, "cpu", max_embeddings_multiples=1, bos=298, eos=299)
    prompts, neg = ["a cat", "dog"], ["", "blurry"]
    cond, pooled, unc, upooled = cnd(prompts, neg, True)

    def ids_of(text, pad=299):
        t = tok(text)
        return torch.tensor([[298] + t + [299] + [pad] * (75 - len(t))])
    with torch.no_grad():
        for i, (ptxt, ntxt) in enumerate(zip(prompts, neg)):
            o1 = te1(input_ids=ids_of(ptxt), output_hidden_states=True, return_dict=True)
            o2 = te2(input_ids=ids_of(ptxt, 0), output_hidden_states=True, return_dict=True)      # tower 2 pads with id 0
            want = torch.cat([o1.hidden_states[-2], o2.hidden_states[-2]], dim=-1)[0]
            assert cond.shape == (2, 77, 80) and pooled.shape == (2, 40)
            assert torch.allclose(cond[i], want, atol=1e-5) and torch.allclose(pooled[i], o2.text_embeds[0], atol=1e-5)
            # an EXPLICIT negative prompt - also the empty string - is encoded
            u2 = te2(input_ids=ids_of(ntxt, 0), output_hidden_states=True, return_dict=True)
            assert torch.allclose(upooled[i], u2.text_embeds[0], atol=1e-5) and float(unc[i].abs().max()) > 0
    # no negative prompt at all: zeros for the whole batch (force_zeros_for_empty_prompt)
    _, _, unc0, up0 = cnd(prompts, None, True)
    assert unc0.shape == cond.shape and float(unc0.abs().max()) == 0 and float(up0.abs().max()) == 0
    ids = T.sdxl_time_ids(3, 1024, 768, original_size=(512, 512), crops_coords_top_left=(8, 16))
    assert ids.tolist() == [[512.0, 512.0, 8.0, 16.0, 1024.0, 768.0]] * 3
    assert T.sdxl_time_ids(1, 1024, 1024).tolist() == [[1024.0, 1024.0, 0.0, 0.0, 1024.0, 1024.0]]


def test_sdxl_conditioner_never_truncates_a_long_prompt():
    """A 100-token prompt beside an empty negative prompt, CFG on, three chunks allowed: both towers and both sides are encoded at
    the chunk count the LONGEST of them needs (2 here) - the positive conditioning keeps every chunk and equals what it is without
    any negative prompt; the negative side is padded up, not the positive cut down."""
    te1, te2 = _tiny_sdxl_towers()
    tok = lambda text: [3 + (ord(c) % 200) for c in text if c != " "]
    cnd = T.SDXLTextConditioner(te1, tok, te2, tok, "cpu", max_embeddings_multiples=3, bos=298, eos=299)
    long_prompt = "x" * 100
    cond, pooled, unc, upooled = cnd([long_prompt], [""], True)
    alone, pooled_alone, _, _ = cnd([long_prompt], None, False)
    assert cond.shape == (1, 2 * 75 + 2, 80) and unc.shape == cond.shape
    assert torch.equal(cond, alone) and torch.equal(pooled, pooled_alone)
    assert float(unc.abs().max()) > 0
    # the second chunk really carries the prompt's tail: it differs from the encoding of the first 75 tokens alone
    short, _, _, _ = cnd([long_prompt[:75]], None, False)
    assert short.shape[1] == 77 and not torch.allclose(cond[:, :76], short[:, :76], atol=1e-4) or cond.shape[1] > short.shape[1]
