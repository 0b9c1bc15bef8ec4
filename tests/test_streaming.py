"""Chunked streaming: the host logic on CPU, and on the GPU that every streamed chunk
is the vocoder's output for exactly the tokens the reference's schedule assigns to it."""
import math
import os

import numpy as np
import pytest

from sparkmi.streaming import ChunkScheduler, crossfade, stream_chunks


def _reference_schedule(n_tokens, dur=1.0, max_dur=30.0, scale=8.0, ov=0.1, rate=50):
    """The loop of runtime/triton_trtllm/model_repo/spark_tts/1/model.py:347-385, restated on indices."""
    max_chunk, chunk, overlap = math.ceil(max_dur * rate), math.ceil(dur * rate), math.ceil(ov * rate)
    arr, out = [], []
    for t in range(n_tokens):
        arr.append(t)
        if len(arr) >= chunk:
            out.append(arr[:chunk])
            arr = arr[chunk - overlap:]
            chunk = min(max_chunk, int(chunk * scale))
    if arr:
        out.append(arr)
    return out


@pytest.mark.parametrize("n", [0, 1, 49, 50, 51, 449, 450, 2000, 3000])
@pytest.mark.parametrize("stride", [1, 7, 64])
def test_scheduler_matches_the_reference_loop(n, stride):
    s = ChunkScheduler()
    incs = [list(range(i, min(i + stride, n))) for i in range(0, n, stride)]
    got = list(stream_chunks(iter(incs), s))
    assert got == _reference_schedule(n)


def test_scheduler_growth_and_overlap():
    s = ChunkScheduler(0.5, 2.0, 2.0, 0.2, 50)    # 25, 50, 100 (cap), overlap 10
    chunks = s.push(range(400)) + s.flush()
    assert [len(c) for c in chunks[:4]] == [25, 50, 100, 100]
    for a, b in zip(chunks, chunks[1:]):
        assert a[-10:] == b[:10]
    with pytest.raises(AssertionError):
        ChunkScheduler(audio_chunk_duration=0.2)
    with pytest.raises(AssertionError):
        ChunkScheduler(audio_chunk_size_scale_factor=0.5)


def test_crossfade_reconstruction():
    rng = np.random.default_rng(0)
    n = 160
    a, b, c = (rng.standard_normal(1000).astype(np.float32) for _ in range(3))
    out = crossfade([a, b, c], n)
    fo, fi = np.linspace(1, 0, n), np.linspace(0, 1, n)
    want = np.concatenate([a[:-n], b[:n] * fi + a[-n:] * fo, b[n:-n], c[:n] * fi + b[-n:] * fo, c[n:-n], c[-n:]])
    np.testing.assert_array_equal(out, want)
    assert crossfade([a], n) is not None and len(crossfade([a], n)) == 1000
    assert crossfade([], n).size == 0
    # a signal that is already continuous across the overlap is reproduced exactly in the overlap
    x = np.sin(np.arange(3000) * 0.01)
    ch = [x[:1000], x[1000 - n:2000], x[2000 - n:]]
    y = crossfade(ch, n)
    np.testing.assert_allclose(y[: 3000 - n], x[: 3000 - n], atol=1e-12)


@pytest.mark.gpu
def test_stream_chunks_are_the_vocoder_output_of_their_tokens(tmp_path):
    import torch
    from sparkmi.pipeline import SparkTTS
    from sparkmi.synthetic import make_model_dir
    lcfg, vcfg = make_model_dir(tmp_path / "m")
    tts = SparkTTS(tmp_path / "m", torch.device("cuda:0"), max_positions=4096, max_frames=512)
    tts._eos = None      # random weights: generate the full budget instead of stopping at a chance eos
    rng = np.random.default_rng(5)
    glob = torch.tensor(rng.integers(0, 4096, size=(1, 1, vcfg.spk_token_num)))
    prompt_tokens = (glob, torch.tensor(rng.integers(0, vcfg.codebook_size, size=(1, 40))))
    kw = dict(text="hello there", prompt_tokens=prompt_tokens, prompt_text="hi", do_sample=False, max_new_tokens=2400)
    hop = tts.audio_tokenizer.model.hop
    rate = tts.sample_rate // hop
    ids = tts.tokenizer([tts.process_prompt(kw["text"], None, kw["prompt_text"], prompt_tokens)[0]], return_tensors="pt").input_ids[0].tolist()
    sem, _ = tts._parse(tts.model.generate_ids([ids], 2400, tts._eos)[0])
    assert len(sem) > 60, "the synthetic model should emit enough semantic tokens for several chunks"
    sp = dict(audio_chunk_duration=0.5, audio_chunk_size_scale_factor=2.0, max_audio_chunk_duration=1.5,
              audio_chunk_overlap_duration=0.1)
    chunks = list(tts.inference_stream(**kw, **sp, decode_stride=7))
    sched = _reference_schedule(len(sem), 0.5, 1.5, 2.0, 0.1, rate)
    assert len(sched) >= 2
    assert [len(c) for c in chunks] == [len(s) * hop for s in sched]
    # every chunk equals the vocoder run on that chunk's tokens alone
    for c, idx in zip(chunks, sched):
        toks = torch.tensor([[sem[i] for i in idx]])
        ref = tts.audio_tokenizer.model.detokenize(toks, glob.long(), lengths=[len(idx)])
        np.testing.assert_array_equal(c, ref.reshape(-1)[: len(idx) * hop].cpu().numpy())
    # a different host polling stride changes nothing
    again = list(tts.inference_stream(**kw, **sp, decode_stride=32))
    assert len(again) == len(chunks) and all(np.array_equal(a, b) for a, b in zip(again, chunks))
    n = math.ceil(0.1 * rate) * hop
    if len(chunks[-1]) >= 2 * n:
        assert len(crossfade(chunks, n)) == len(sem) * hop
