"""CPU oracle for the audio-reactive StyleGAN2 render path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``maua_amd/`` may import this package.
Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — always as the *checker*, never as the
thing that is shipped or measured as the product.

It is a plain restatement (PyTorch-CPU fp32 / numpy / one small C file) of the
reference's in-tree arithmetic for this path:

* ``oracle.ops``        <- maua/GAN/wrappers/inference/ops.py
* ``oracle.stylegan2``  <- maua/GAN/wrappers/inference/stylegan2.py
* ``oracle.audio``      <- maua/audiovisual/audioreactive/selfsupervised/features/{audio,processing}.py
                           and .../features/rosa/{spectral,beat,convert,helpers}.py
* ``oracle.signal``     <- maua/audiovisual/audioreactive/signal.py
* ``oracle.latent``     <- maua/audiovisual/audioreactive/latent.py and
                           .../selfsupervised/latent.py
* ``oracle.noise``      <- .../selfsupervised/noise.py
* ``oracle.quantile``   <- .../features/efficient_quantile/efficient_quantile.cpp (C restatement in quantile.c)
* ``oracle.io``         <- maua/ops/io.py:47-70 (tensor2bytes), maua/GAN/wrappers/stylegan.py:58-69 (seeds)

Pinning: every function here is checked in ``tests/test_oracle_golden.py``
against fixtures under ``tests/golden/`` that were produced by importing the
reference itself in the authoring container (``tests/golden/make_golden.py``).
Where the reference code cannot execute as written (the up=2
branch of conv2d_resample), the golden is composed from the reference pieces
that do run, exactly as ops.py:211-225 composes them.
"""
