"""Import-path shim: with ``spark-tts_amd/`` on ``sys.path``, ``from cli.SparkTTS import SparkTTS``
resolves to the MI355X-native drop-in exactly where the reference's callers look for it
(reference callers: cli/inference.py:22, webui.py)."""
from sparkmi.pipeline import SparkTTS  # noqa: F401
from sparkmi.pipeline_text import GENDER_MAP, LEVELS_MAP, TASK_TOKEN_MAP  # noqa: F401
