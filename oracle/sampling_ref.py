"""ORACLE (test infrastructure, never the product path): the token-selection chain of the reference's default
decoding mode, ``model.generate(do_sample=True, top_k=50, top_p=0.95, temperature=0.8)`` at
``cli/SparkTTS.py:166-168,197-204``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.  The
arithmetic is third-party (``transformers==4.46.2`` pinned at ``requirements.txt:12``; 5.15.0 installed here):
``generate`` builds a ``LogitsProcessorList`` of ``TemperatureLogitsWarper`` -> ``TopKLogitsWarper`` ->
``TopPLogitsWarper`` (``generation/logits_process.py``, abbreviated ``LP``; line numbers of 5.15.0), takes
``softmax`` of the surviving scores and draws one ``torch.multinomial`` sample.  Restated below with the same
torch ops in the same order, so that the probabilities are bit-equal.

Pinning: ``tests/golden/gen_golden_sampling.py`` runs transformers' own three warper classes on committed
logits rows (the tiny model's, a 166 000-entry row, a row with a tie at the k-th value and a row whose nucleus
cut falls next to a cumulative probability) and stores the surviving ids and probabilities in
``tests/golden/sampling.npz``; ``tests/test_oracle_sampling.py`` requires equality.  The draw itself is not
restated: the product uses its own counter-based stream (Philox keyed per sequence), so the GPU tests compare
empirical frequencies with these probabilities.
"""
from __future__ import annotations

import torch


def temperature_warp(scores: torch.Tensor, temperature: float) -> torch.Tensor:
    """LP:300-303  scores / temperature."""
    return scores / temperature


def top_k_warp(scores: torch.Tensor, top_k: int, filter_value: float = -float("inf")) -> torch.Tensor:
    """LP:589-595.  Everything strictly below the k-th largest value is removed, so values that TIE with the k-th
    all stay (more than k survivors)."""
    top_k = min(top_k, scores.size(-1))
    kth = torch.topk(scores, top_k)[0][..., -1, None]
    return scores.masked_fill(scores < kth, filter_value)


def top_p_warp(scores: torch.Tensor, top_p: float, min_tokens_to_keep: int = 1,
               filter_value: float = -float("inf")) -> torch.Tensor:
    """LP:526-540.  Ascending sort, softmax, cumulative sum; the low tail whose cumulative probability is
    <= 1 - top_p goes, the last ``min_tokens_to_keep`` sorted entries always stay."""
    sorted_logits, sorted_indices = torch.sort(scores, descending=False)
    cumulative = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
    remove_sorted = cumulative <= (1 - top_p)
    remove_sorted[..., -min_tokens_to_keep:] = 0
    remove = remove_sorted.scatter(-1, sorted_indices, remove_sorted)
    return scores.masked_fill(remove, filter_value)


def sampling_probs(logits: torch.Tensor, temperature: float, top_k: int, top_p: float) -> torch.Tensor:
    """Full-vocabulary probability vector ``generate`` draws from (``GenerationMixin._sample``:
    ``probs = softmax(next_token_scores)`` after the processor list); logits [V] or [B][V], fp32."""
    z = logits.to(torch.float32)
    squeeze = z.dim() == 1
    if squeeze:
        z = z[None]
    z = temperature_warp(z, temperature)
    z = top_k_warp(z, top_k)
    if top_p < 1.0:     # generate() only adds the warper when top_p < 1 (GenerationMixin._get_logits_processor)
        z = top_p_warp(z, top_p)
    p = torch.nn.functional.softmax(z, dim=-1)
    return p[0] if squeeze else p
