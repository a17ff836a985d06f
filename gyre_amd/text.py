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

    def __init__(self, model, device, dtype=torch.float32):
        self.model = model.to(device=device, dtype=dtype).eval()
        self.device = device

    @classmethod
    def synthetic(cls, device="cuda:0", dtype=torch.float32, seed: int = 0, hidden: int = 768, layers: int = 12,
                  heads: int = 12):
        from transformers import CLIPTextConfig, CLIPTextModel
        cfg = CLIPTextConfig(vocab_size=VOCAB, hidden_size=hidden, intermediate_size=4 * hidden,
                             num_hidden_layers=layers, num_attention_heads=heads, max_position_embeddings=MAX_LEN,
                             hidden_act="quick_gelu", bos_token_id=BOS, eos_token_id=EOS, pad_token_id=EOS)
        torch.manual_seed(seed)
        with torch.device(device):  # initialise straight on the target device (CPU init of 123 M params takes ~15 s)
            model = CLIPTextModel(cfg)
        return cls(model, device, dtype)

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor) -> torch.Tensor:
        out = self.model(input_ids=input_ids.to(self.device), return_dict=True)
        return out.last_hidden_state.float()
