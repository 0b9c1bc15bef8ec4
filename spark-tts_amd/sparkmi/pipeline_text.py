"""Prompt-string construction and generated-token parsing for the Spark-TTS pipeline.

Host-side string logic only (no tensors).  Token tables restate the *values* of
``sparktts/utils/token_parser.py:1-33``; the two prompt layouts follow
``cli/SparkTTS.py:83-104`` (voice clone) and ``cli/SparkTTS.py:128-155`` (controllable TTS).
"""
from __future__ import annotations

import re
from typing import Iterable, List, Optional, Sequence

TASK_TOKEN_MAP = {
    "vc": "<|task_vc|>", "tts": "<|task_tts|>", "asr": "<|task_asr|>", "s2s": "<|task_s2s|>",
    "t2s": "<|task_t2s|>", "understand": "<|task_understand|>", "caption": "<|task_cap|>",
    "controllable_tts": "<|task_controllable_tts|>", "prompt_tts": "<|task_prompt_tts|>",
    "speech_edit": "<|task_edit|>",
}
LEVELS_MAP = {"very_low": 0, "low": 1, "moderate": 2, "high": 3, "very_high": 4}
GENDER_MAP = {"female": 0, "male": 1}

_SEM_RE = re.compile(r"bicodec_semantic_(\d+)")
_GLB_RE = re.compile(r"bicodec_global_(\d+)")


def _tok(kind: str, ids: Iterable[int]) -> str:
    return "".join(f"<|bicodec_{kind}_{int(i)}|>" for i in ids)


def build_clone_prompt(text: str, global_ids: Sequence[int], semantic_ids: Sequence[int],
                       prompt_text: Optional[str]) -> str:
    """Voice-clone prompt.  With a transcript the prompt also carries the prompt audio's
    semantic tokens and leaves ``<|start_semantic_token|>`` open for continuation."""
    head = TASK_TOKEN_MAP["tts"] + "<|start_content|>"
    glob = "<|start_global_token|>" + _tok("global", global_ids) + "<|end_global_token|>"
    if prompt_text is None:
        return head + text + "<|end_content|>" + glob
    return (head + prompt_text + text + "<|end_content|>" + glob
            + "<|start_semantic_token|>" + _tok("semantic", semantic_ids))


def build_control_prompt(gender: str, pitch: str, speed: str, text: str) -> str:
    """Controllable-TTS prompt; invalid attribute names raise AssertionError like the
    reference's asserts at ``cli/SparkTTS.py:129-131``."""
    assert gender in GENDER_MAP
    assert pitch in LEVELS_MAP
    assert speed in LEVELS_MAP
    style = (f"<|gender_{GENDER_MAP[gender]}|>" f"<|pitch_label_{LEVELS_MAP[pitch]}|>"
             f"<|speed_label_{LEVELS_MAP[speed]}|>")
    return (TASK_TOKEN_MAP["controllable_tts"] + "<|start_content|>" + text + "<|end_content|>"
            + "<|start_style_label|>" + style + "<|end_style_label|>")


def parse_semantic(decoded: str) -> List[int]:
    """``re.findall(r"bicodec_semantic_(\\d+)")`` of ``cli/SparkTTS.py:216-220``."""
    return [int(t) for t in _SEM_RE.findall(decoded)]


def parse_global(decoded: str) -> List[int]:
    """``cli/SparkTTS.py:222-228`` (control mode: global tokens come from the LM)."""
    return [int(t) for t in _GLB_RE.findall(decoded)]
