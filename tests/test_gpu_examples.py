"""The examples (device counterparts of the reference's examples/modem.rs, plotting.rs waterfall,
pipeline.rs) run end to end and report sane results."""
import os
import runpy
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EX = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples")


def _load(name):
    return runpy.run_path(os.path.join(EX, name))


def test_modem_example(ctx):
    errors = _load("modem.py")["main"](nbits=1 << 16)
    assert errors == 0            # noise amplitude 0.01 per the double scaling (noise.rs:41-42,58): far inside the decision regions


def test_waterfall_example(ctx):
    db = _load("waterfall.py")["main"](fft_len=2048, frames=64)
    # N(0,1) on each component (noise.rs:39-43) = power 2 = 3.01 dB; Scale::SN keeps it per bin
    assert db.shape == (2048,) and abs(db.mean() - 3.01) < 0.5


def test_pipeline_example(ctx):
    assert _load("pipeline.py")["main"](n=1 << 20)
