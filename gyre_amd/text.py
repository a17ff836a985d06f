"""CLIP text conditioning - stays thin PyTorch-ROCm host code (north_star); one call per request.

The reference reaches transformers.CLIPTextModel through LPWTextEmbedding
(gyre/pipeline/text_embedding/lpw_text_embedding.py:217-234) and TextEncoderAltLayer
(text_encoder_alt_layer.py:6-36).  There is no tokenizer vocabulary or checkpoint offline, so the
bench / tests feed synthetic token ids (BOS 49406 ... EOS 49407 padding) to a randomly initialised
encoder of the SD1.x shape (CLIP ViT-L/14 text tower: 12 layers, width 768, 12 heads, 77 positions).
"""
from __future__ import annotations

from typing import Optional

import torch

BOS, EOS, VOCAB, MAX_LEN = 49406, 49407, 49408, 77


def synthetic_prompt_ids(batch: int, seed: int = 1234, min_len: int = 5, max_len: int = 40) -> torch.Tensor:
    """[B,77] int64: BOS, random tokens, EOS, EOS-padding (how the CLIP tokenizer pads for SD1.x)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.full((batch, MAX_LEN), EOS, dtype=torch.int64)
    ids[:, 0] = BOS
    for b in range(batch):
        n = int(torch.randint(min_len, max_len + 1, (1,), generator=g))
        ids[b, 1:1 + n] = torch.randint(0, BOS, (n,), generator=g)
    return ids


def empty_prompt_ids(batch: int) -> torch.Tensor:
    ids = torch.full((batch, MAX_LEN), EOS, dtype=torch.int64)
    ids[:, 0] = BOS
    return ids


class ClipTextEncoder:
    """ids [B,77] -> last_hidden_state [B,77,768] (final LayerNorm applied, as SD1.x uses)."""

    def __init__(self, model, device, dtype=torch.float32, layer="final"):
        """layer: "final" | "penultimate" | n (n-th hidden state from the end) - the reference's TextEncoderAltLayer
        (text_encoder_alt_layer.py:6-36, "clip skip"): earlier layers are passed through the final LayerNorm."""
        if not (layer in ("final", "penultimate") or (isinstance(layer, int) and layer >= 1)):
            raise ValueError(f"layer must be 'final', 'penultimate' or a positive int, got {layer!r}")
        self.model = model.to(device=device, dtype=dtype).eval()
        self.device = device
        self.layer = layer

    @classmethod
    def synthetic(cls, device="cuda:0", dtype=torch.float32, seed: int = 0, hidden: int = 768, layers: int = 12,
                  heads: int = 12, layer="final"):
        from transformers import CLIPTextConfig, CLIPTextModel
        cfg = CLIPTextConfig(vocab_size=VOCAB, hidden_size=hidden, intermediate_size=4 * hidden,
                             num_hidden_layers=layers, num_attention_heads=heads, max_position_embeddings=MAX_LEN,
                             hidden_act="quick_gelu", bos_token_id=BOS, eos_token_id=EOS, pad_token_id=EOS)
        torch.manual_seed(seed)
        with torch.device(device):  # initialise straight on the target device (CPU init of 123 M params takes ~15 s)
            model = CLIPTextModel(cfg)
        return cls(model, device, dtype, layer)

    def _final_layer_norm(self):
        # transformers <= 4.x nests the tower under .text_model (what the reference uses); 5.x flattens it
        tower = getattr(self.model, "text_model", self.model)
        return tower.final_layer_norm

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor) -> torch.Tensor:
        final = self.layer == "final"
        out = self.model(input_ids=input_ids.to(self.device), output_hidden_states=not final, return_dict=True)
        if final:
            return out.last_hidden_state.float()
        back = 2 if self.layer == "penultimate" else int(self.layer)
        return self._final_layer_norm()(out.hidden_states[-back]).float()


# ------------------------------------------------------------------------------------------------
# Long-prompt weighting (host logic of reference text_embedding/lpw_text_embedding.py:35-405):
# "(text)" x1.1, "(text:w)" x w, "[text]" /1.1, backslash escapes; tokens are padded to 75k+2, encoded in
# 77-token chunks, multiplied by their weights and rescaled to keep the embedding mean.
# Pinned by tests/golden (parser outputs, padding, weighting produced by the reference functions).
# ------------------------------------------------------------------------------------------------
def parse_prompt_attention(text: str):
    """-> [[fragment, weight], ...] with adjacent equal-weight fragments merged."""
    up, down = 1.1, 1.0 / 1.1
    res, rounds, squares = [], [], []

    def scale_from(start, m):
        for item in res[start:]:
            item[1] *= m

    i, n, buf = 0, len(text), ""

    def flush():
        nonlocal buf
        if buf:
            res.append([buf, 1.0])
            buf = ""

    while i < n:
        ch = text[i]
        if ch == "\\" and i + 1 < n and text[i + 1] in "()[]\\":
            flush()
            res.append([text[i + 1], 1.0])
            i += 2
            continue
        if ch == "\\":  # a lone backslash is dropped (the reference grammar consumes it as an escape lead-in)
            flush()
            res.append(["", 1.0])
            i += 1
            continue
        if ch == "(":
            flush(); rounds.append(len(res)); i += 1; continue
        if ch == "[":
            flush(); squares.append(len(res)); i += 1; continue
        if ch == ":":
            # ":<number>)" closes a round bracket with an explicit weight
            j = i + 1
            while j < n and text[j] == " ":
                j += 1
            k = j
            if k < n and text[k] in "+-":
                k += 1
            d0 = k
            while k < n and (text[k].isdigit() or text[k] == "."):
                k += 1
            e = k
            while e < n and text[e] == " ":
                e += 1
            if k > d0 and e < n and text[e] == ")":
                try:
                    w = float(text[j:k])
                    if rounds:
                        flush(); scale_from(rounds.pop(), w)
                    else:
                        buf += text[i:e + 1]; flush()
                    i = e + 1
                    continue
                except ValueError:
                    pass
            buf += ch; i += 1; continue
        if ch == ")":
            flush()
            if rounds:
                scale_from(rounds.pop(), up)
            else:
                res.append([")", 1.0])
            i += 1; continue
        if ch == "]":
            flush()
            if squares:
                scale_from(squares.pop(), down)
            else:
                res.append(["]", 1.0])
            i += 1; continue
        buf += ch
        i += 1
    flush()
    for pos in rounds:
        scale_from(pos, up)
    for pos in squares:
        scale_from(pos, down)
    if not res:
        res = [["", 1.0]]
    merged = [res[0]]
    for frag, w in res[1:]:
        if merged[-1][1] == w:
            merged[-1][0] += frag
        else:
            merged.append([frag, w])
    return merged


def pad_tokens_and_weights(tokens, weights, max_length: int, bos: int, eos: int, no_boseos_middle: bool = True,
                           chunk_length: int = MAX_LEN, pad: Optional[int] = None):
    """BOS + tokens + EOS padding to max_length; weights padded with 1.0 (per chunk when BOS/EOS are kept).  `pad`: id behind the
    closing EOS (default: EOS itself, the reference's long-prompt convention; SDXL's second tokenizer pads with 0)."""
    mult = (max_length - 2) // (chunk_length - 2)
    out_t, out_w = [], []
    for tk, wt in zip(tokens, weights):
        tail = max_length - 1 - len(tk)
        out_t.append([bos] + list(tk) + ([eos] * tail if pad is None or tail <= 0 else [eos] + [pad] * (tail - 1)))
        if no_boseos_middle:
            out_w.append([1.0] + list(wt) + [1.0] * (max_length - 1 - len(wt)))
        else:
            total = mult * chunk_length
            if not wt:
                w = [1.0] * total
            else:
                w = []
                body = chunk_length - 2
                for j in range(mult):
                    w.append(1.0)
                    w += list(wt[j * body:min(len(wt), (j + 1) * body)])
                    w.append(1.0)
                w += [1.0] * (total - len(w))
            out_w.append(w)
    return out_t, out_w


def chunked_encode(encoder, input_ids: torch.Tensor, chunk_length: int = MAX_LEN, no_boseos_middle: bool = True):
    """Encode [B, 75k+2] ids in k chunks of 77 (BOS/EOS re-attached per chunk), concatenated along the token axis."""
    mult = (input_ids.shape[1] - 2) // (chunk_length - 2)
    if mult <= 1:
        return encoder(input_ids)
    body = chunk_length - 2
    parts = []
    for i in range(mult):
        chunk = input_ids[:, i * body:(i + 1) * body + 2].clone()
        chunk[:, 0] = input_ids[0, 0]
        chunk[:, -1] = input_ids[0, -1]
        e = encoder(chunk)
        if no_boseos_middle:
            e = e[:, :-1] if i == 0 else (e[:, 1:] if i == mult - 1 else e[:, 1:-1])
        parts.append(e)
    return torch.cat(parts, dim=1)


def apply_token_weights(emb: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """emb * w, then rescaled so that the per-prompt mean of the embedding is unchanged."""
    prev = emb.mean(dim=(-2, -1))
    out = emb * weights.unsqueeze(-1)
    return out * (prev / out.mean(dim=(-2, -1))).unsqueeze(-1).unsqueeze(-1)


class LPWTextEmbedder:
    """get_embeddings(prompts, negative_prompts) -> (cond [B,75k+2... ,D], uncond).  `tokenize(fragment) -> [ids]`
    is supplied by the caller (a CLIP tokenizer's ids without BOS/EOS); no vocabulary ships offline."""

    def __init__(self, encoder, tokenize, max_embeddings_multiples: int = 3, bos: int = BOS, eos: int = EOS, pad: Optional[int] = None):
        self.encoder, self.tokenize, self.mult, self.bos, self.eos, self.pad = encoder, tokenize, max_embeddings_multiples, bos, eos, pad

    def chunks_needed(self, prompts) -> int:
        """77-token chunks the longest of `prompts` needs under this tokenizer (1 ... max_embeddings_multiples)."""
        body = MAX_LEN - 2
        toks, _ = self._tokens(prompts, body * self.mult)
        longest = max([len(t) for t in toks] + [1])
        return max(1, min(self.mult, (longest - 1) // body + 1))

    def _tokens(self, prompts, max_len):
        toks, wts = [], []
        for ptxt in prompts:
            t, w = [], []
            for frag, weight in (parse_prompt_attention(ptxt) if isinstance(ptxt, str) else ptxt):
                ids = list(self.tokenize(frag))
                t += ids
                w += [weight] * len(ids)
                if len(t) > max_len:
                    break
            toks.append(t[:max_len]); wts.append(w[:max_len])
        return toks, wts

    def get_embeddings(self, prompts, uncond_prompts=None, chunks: Optional[int] = None):
        """chunks: encode at exactly this many 77-token chunks (callers that must agree on one length across several calls);
        default: what the longest of prompts / uncond_prompts needs."""
        body = MAX_LEN - 2
        lim = body * self.mult
        pt, pw = self._tokens(prompts, lim)
        ut, uw = self._tokens(uncond_prompts, lim) if uncond_prompts is not None else ([], [])
        longest = max([len(t) for t in pt + ut] + [1])
        mult = max(1, min(self.mult, (longest - 1) // body + 1))
        if chunks is not None:
            mult = max(mult, int(chunks))
        max_length = body * mult + 2
        dev = getattr(self.encoder, "device", "cpu")

        def run(tok, wt):
            tok, wt = pad_tokens_and_weights(tok, wt, max_length, self.bos, self.eos, True, MAX_LEN, self.pad)
            ids = torch.tensor(tok, dtype=torch.long, device=dev)
            emb = chunked_encode(self.encoder, ids, MAX_LEN, True)
            return apply_token_weights(emb, torch.tensor(wt, dtype=emb.dtype, device=emb.device))

        cond = run(pt, pw)
        return cond, (run(ut, uw) if uncond_prompts is not None else None)


# ------------------------------------------------------------------------------------------------
# SDXL conditioning (BASELINE.json configs[3]; the reference has no SDXL - an extension on the same engine surface).
# Published scheme of the SDXL-base pipeline (Podell et al. 2023, diffusers StableDiffusionXLPipeline.encode_prompt /
# _get_add_time_ids; PARITY UNPINNED: neither is in /root/reference):
#   * two text towers see the same prompt: CLIP ViT-L (width 768) and OpenCLIP ViT-bigG (width 1280); each contributes its
#     PENULTIMATE hidden state WITHOUT the final LayerNorm; concatenated along channels -> context [B, 77, 2048];
#   * the second tower's pooled, projected EOS feature -> text_embeds [B, 1280];
#   * time_ids [B, 6] = (original_h, original_w, crop_top, crop_left, target_h, target_w);
#   * an empty negative prompt conditions on ZEROS (force_zeros_for_empty_prompt, the SDXL-base default).
# UNet side: gyre_amd.modules.GyreHipUNet._aug_embedding (text_time MLP on the host) -> temb_add of the native call.
# ------------------------------------------------------------------------------------------------
class SDXLTextConditioner:
    """(prompts, negative prompts) -> context / pooled embeddings of both SDXL text towers, with the long-prompt weighting of
    LPWTextEmbedder applied per tower (same parser, chunking and mean renormalisation as the SD1.x path)."""

    def __init__(self, text_encoder, tokenize, text_encoder_2, tokenize_2, device, max_embeddings_multiples: int = 3,
                 force_zeros_for_empty_prompt: bool = True, bos: int = BOS, eos: int = EOS, pad_1: Optional[int] = None,
                 pad_2: Optional[int] = 0):
        self.te1, self.te2, self.tok1, self.tok2 = text_encoder, text_encoder_2, tokenize, tokenize_2
        self.device, self.mult, self.force_zeros = device, max_embeddings_multiples, force_zeros_for_empty_prompt
        # pad ids as the published pipeline's tokenizers have them: tower 1 pads with EOS, tower 2 (tokenizer_2.pad_token_id) with 0
        self.bos, self.eos, self.pad_1, self.pad_2 = bos, eos, pad_1, pad_2

    def _tower(self, model, pooled_sink: Optional[list]):
        dev = self.device

        def encode(ids):
            out = model(input_ids=ids.to(dev), output_hidden_states=True, return_dict=True)
            if pooled_sink is not None and not pooled_sink:           # pooled feature of the FIRST 77-token chunk
                pooled = getattr(out, "text_embeds", None)
                if pooled is None:
                    raise ValueError("the second SDXL text encoder must return text_embeds (CLIPTextModelWithProjection)")
                pooled_sink.append(pooled.float())
            return out.hidden_states[-2].float()
        encode.device = dev
        return encode

    def _embedders(self, sink):
        return (LPWTextEmbedder(self._tower(self.te1, None), self.tok1, self.mult, self.bos, self.eos, self.pad_1),
                LPWTextEmbedder(self._tower(self.te2, sink), self.tok2, self.mult, self.bos, self.eos, self.pad_2))

    def _run(self, prompts, chunks):
        sink = []
        e1, e2 = self._embedders(sink)
        c1, _ = e1.get_embeddings(prompts, None, chunks)
        c2, _ = e2.get_embeddings(prompts, None, chunks)
        assert c1.shape[1] == c2.shape[1], "both towers are encoded at the same chunk count"
        return torch.cat([c1, c2], dim=-1), sink[0]

    def __call__(self, prompts, negative_prompts=None, do_cfg: bool = True):
        """-> (context [B,77k,2048], pooled [B,1280], uncond context | None, uncond pooled | None).
        ONE chunk count for both towers and both sides: the largest any (prompt, tokenizer) pair needs - the conditioning is never
        truncated to a shorter partner (a 100-token prompt beside an empty negative keeps all its chunks and its closing EOS).
        negative_prompts None: zeros for the whole batch when force_zeros_for_empty_prompt (the published pipeline's rule); an
        explicit "" is ENCODED like any other prompt."""
        prompts = list(prompts)
        no_negative = negative_prompts is None
        neg = [""] * len(prompts) if no_negative else list(negative_prompts)
        if len(neg) == 1 and len(prompts) > 1:
            neg = neg * len(prompts)
        e1, e2 = self._embedders([])
        chunks = max(e1.chunks_needed(prompts), e2.chunks_needed(prompts))
        if do_cfg:
            chunks = max(chunks, e1.chunks_needed(neg), e2.chunks_needed(neg))
        cond, pooled = self._run(prompts, chunks)
        if not do_cfg:
            return cond, pooled, None, None
        if no_negative and self.force_zeros:
            return cond, pooled, torch.zeros_like(cond), torch.zeros_like(pooled)
        unc, upooled = self._run(neg, chunks)
        return cond, pooled, unc, upooled


def sdxl_time_ids(batch: int, height: int, width: int, original_size=None, crops_coords_top_left=(0, 0), target_size=None,
                  device="cpu") -> torch.Tensor:
    """[B, 6] float32: (original_h, original_w, crop_top, crop_left, target_h, target_w); the sizes default to the request's."""
    oh, ow = original_size or (height, width)
    th, tw = target_size or (height, width)
    row = torch.tensor([[float(oh), float(ow), float(crops_coords_top_left[0]), float(crops_coords_top_left[1]), float(th), float(tw)]],
                       device=device)
    return row.expand(batch, -1).contiguous()
