#!/usr/bin/env python3
"""Writes tests/golden/generator_v3.json: the build's OWN complex-normal generator (csrc/aeth_rng.h, version 3 of its
floating-point stage) pinned as data -- samples of a few streams at a few positions, and the Box-Muller stage on chosen
word pairs, as the 32-bit patterns of their f32 values.  Produced by the CPU oracle (oracle/awgn_restatement.inc); the
reference's StdRng stream cannot be reproduced, so this fixture does not pin the reference: it pins the DEFINITION both
the oracle and the kernel implement, so that a change that slips into both at once still shows.

Run:  python tests/golden/make_generator_fixture.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as orc                                          # noqa: E402


def hexes(z):
    return [f"{v:08x}" for v in np.ascontiguousarray(z, np.complex64).view(np.uint32)]


streams = []
for seed, offset, power in ((815, 0, 1.0), (815, 1, 1.0), (7, 1 << 33, 0.25), (0xFEDCBA9876543210, (1 << 40) + 3, 0.01)):
    streams.append({"seed": seed, "offset": offset, "power": power, "n": 24, "fill": hexes(orc.awgn_fill(24, power, seed, offset))})
a = np.array([0, 0x100, 0xffffffff, 0x80000000, 0x12345678, 0x00000123, 0xfffffe00, 0x3504f300], np.uint32)
b = np.array([0, 0x40000000, 0x80000000, 0xc0000000, 0xffffffff, 0x12345678, 0x7fffffff, 0x00000001], np.uint32)
out = {"what": __doc__.split("\n\nRun:")[0],
       "generator": "Philox4x32-7, key = seed, counter = (position >> 1, 0); sample 2k from words 0,1 and 2k+1 from words 2,3",
       "streams": streams,
       "pairs": {"a": [f"{v:08x}" for v in a], "b": [f"{v:08x}" for v in b], "normal": hexes(orc.rng_normal_pairs(a, b))}}
json.dump(out, open(os.path.join(HERE, "generator_v3.json"), "w"), indent=1)
print("wrote", os.path.join(HERE, "generator_v3.json"))
