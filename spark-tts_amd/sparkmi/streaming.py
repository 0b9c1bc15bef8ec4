"""Streaming (chunked) vocoding: audio is emitted while the LLM is still generating.

Mirrors the reference's decoupled Triton mode: the server side cuts the growing semantic-token
stream into chunks that overlap by ``audio_chunk_overlap_duration`` and grow by
``audio_chunk_size_scale_factor`` up to ``max_audio_chunk_duration``
(``runtime/triton_trtllm/model_repo/spark_tts/1/model.py:347-385``; defaults
``runtime/triton_trtllm/run.sh:53-56``), each chunk is vocoded on its own, and the client
cross-fades consecutive chunks over the overlap (``runtime/triton_trtllm/client_grpc.py:390-415``).

``ChunkScheduler`` is the pure host logic (no GPU), ``crossfade`` the client-side reconstruction;
``SparkTTS.inference_stream`` (pipeline.py) drives the HIP LLM and vocoder with them.
"""
from __future__ import annotations

import math
from typing import Iterable, Iterator, List, Sequence

import numpy as np


class ChunkScheduler:
    """Feed semantic tokens as they are generated; ``push`` returns the chunks that became ready
    and ``flush`` the final (shorter) one.  Token bookkeeping is exactly the reference loop's:
    a ready chunk is the first ``chunk_size`` buffered tokens, the buffer then keeps the last
    ``overlap`` of them, and ``chunk_size`` grows (model.py:358-375)."""

    def __init__(self, audio_chunk_duration: float = 1.0, max_audio_chunk_duration: float = 30.0,
                 audio_chunk_size_scale_factor: float = 8.0, audio_chunk_overlap_duration: float = 0.1,
                 frame_rate: int = 50):
        # same argument checks as model.py:121-129
        assert float(audio_chunk_duration) >= 0.5, "audio_chunk_duration at least 0.5 seconds"
        assert float(audio_chunk_size_scale_factor) >= 1.0, \
            "audio_chunk_size_scale_factor should be greater than 1, change it according to your actual rtf"
        self.max_chunk_size = math.ceil(max_audio_chunk_duration * frame_rate)
        self.chunk_size = math.ceil(audio_chunk_duration * frame_rate)
        self.overlap = math.ceil(audio_chunk_overlap_duration * frame_rate)
        self.scale = float(audio_chunk_size_scale_factor)
        self.buf: List[int] = []

    def push(self, tokens: Iterable[int]) -> List[List[int]]:
        out = []
        for t in tokens:
            self.buf.append(int(t))
            if len(self.buf) >= self.chunk_size:
                out.append(self.buf[: self.chunk_size])
                self.buf = self.buf[self.chunk_size - self.overlap:]
                self.chunk_size = min(self.max_chunk_size, int(self.chunk_size * self.scale))
        return out

    def flush(self) -> List[List[int]]:
        out = [self.buf] if self.buf else []
        self.buf = []
        return out


def crossfade(chunks: Sequence[np.ndarray], overlap_samples: int) -> np.ndarray:
    """Client-side reconstruction (client_grpc.py:390-415): linear fade over the overlap; the
    first chunk loses its tail, middle chunks lose both ends, the last chunk's tail is kept."""
    chunks = [np.asarray(c).reshape(-1) for c in chunks if np.asarray(c).size > 0]
    if not chunks:
        return np.zeros(0, dtype=np.float32)
    if len(chunks) == 1:
        return chunks[0]
    n = int(overlap_samples)
    fade_out = np.linspace(1, 0, n)
    fade_in = np.linspace(0, 1, n)
    out = [chunks[0][:-n]]
    for i in range(1, len(chunks)):
        out.append(chunks[i][:n] * fade_in + chunks[i - 1][-n:] * fade_out)
        out.append(chunks[i][n:-n])
    out.append(chunks[-1][-n:])
    return np.concatenate(out)


def stream_chunks(token_iter: Iterator[Sequence[int]], scheduler: ChunkScheduler) -> Iterator[List[int]]:
    """Token increments in, ready chunks out (the generator form of the reference loop)."""
    for inc in token_iter:
        if inc is None or len(inc) == 0:
            break
        for c in scheduler.push(inc):
            yield c
    for c in scheduler.flush():
        yield c
