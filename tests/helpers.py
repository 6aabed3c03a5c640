"""Shared helpers for the parity tests (fixture expansion, seeded inputs)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SCALE_KIND = {"None": 0, "SN": 1, "N": 2, "X": 3}


def load_kat():
    with open(os.path.join(HERE, "golden", "reference_kat.json")) as f:
        return {c["id"]: c for c in json.load(f)["cases"]}


def expand(v):
    """fixture vector -> complex64 array"""
    if "rep" in v:
        return np.full(v["n"], complex(*v["rep"]), np.complex64)
    if "seq" in v:
        return np.array([complex(a, b) for a, b in v["seq"]], np.complex64)
    if "dc" in v:
        out = np.zeros(v["n"], np.complex64)
        out[0] = complex(*v["dc"])
        return out
    raise KeyError(v)


def rand_c64(seed, n, scale=1.0):
    rng = np.random.default_rng(seed)
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * (scale / np.sqrt(2))).astype(np.complex64)


def bits_equal(a, b):
    """bit-exact comparison of complex64 arrays (NaN-safe, distinguishes -0.0)"""
    a = np.ascontiguousarray(a, np.complex64).view(np.uint32)
    b = np.ascontiguousarray(b, np.complex64).view(np.uint32)
    return a.shape == b.shape and bool((a == b).all())
