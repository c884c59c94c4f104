"""Drop-in name for maua/prompt.py:12-60: re-exports the prompt classes of maua_amd.grad (+ EmbeddingPrompt: a text prompt whose
embedding was computed by a text tower outside this build)."""
from maua_amd.grad import ContentPrompt, EmbeddingPrompt, ImagePrompt, StylePrompt, TextPrompt  # noqa: F401
