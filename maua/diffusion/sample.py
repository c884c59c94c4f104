"""BASELINE configs[3] names `maua.diffusion.sample` (the reference has no such module: its diffusion entry points are
maua/diffusion/image.py / video.py and carry no audio coupling): the audio-onset-switched DDIM sampler of maua_amd.diffusion."""
from maua_amd.diffusion import ImageTarget, MSEGuide, onset_prompt_schedule, sample  # noqa: F401
