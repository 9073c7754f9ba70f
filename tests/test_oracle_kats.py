"""CPU tests: the oracle (and the tables it shares with the kernels) against known-answer vectors that were derived
from the written definitions by tests/golden/make_kats.py, which imports nothing from oracle/ or renderer_amd/ and
reads no generated table.  A common-mode error in the hash, the sample positions, the occlusion-ray directions or
the (cos, sin) table would show here."""
import json
import os
import re

import numpy as np

import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kats():
    with open(os.path.join(common.GOLDEN, "kats.json")) as f:
        return json.load(f)


def test_hash32_known_answers(oracle):
    for x, want in _kats()["hash32"]:
        assert oracle.hash32(x) == want, hex(x)


def test_sample_offsets_known_answers(oracle):
    for pixel, s, seed, sx, sy in _kats()["sample_offset"]:
        got = oracle.sample_offset(pixel, s, seed)
        assert (float(got[0]).hex(), float(got[1]).hex()) == (sx, sy), (pixel, s, seed)


def test_angle_table_known_answers():
    """Both copies of the generated table (the oracle's and the kernels') against float32-rounded cos / sin."""
    want = np.array([[float.fromhex(c), float.fromhex(s)] for c, s in _kats()["angle_table_hex"]], np.float32)
    assert want.shape == (256, 2)
    for rel in ("oracle/ao_dirs.inc", "renderer_amd/csrc/ao_dirs.inc"):
        text = open(os.path.join(ROOT, rel)).read()
        vals = re.findall(r"\{(-?0x[0-9a-f.]+p[-+]?\d+)f, (-?0x[0-9a-f.]+p[-+]?\d+)f\}", text)
        got = np.array([[float.fromhex(c), float.fromhex(s)] for c, s in vals], np.float32)
        assert got.shape == (256, 2), rel
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), rel


def test_occlusion_ray_directions_known_answers(oracle):
    for case in _kats()["ao_directions"]:
        e1 = [float.fromhex(v) for v in case["e1"]]
        e2 = [float.fromhex(v) for v in case["e2"]]
        got = oracle.ao_directions(e1, e2, case["pixel"], case["seed"], case["rays"])
        want = np.array([[float.fromhex(c) for c in row] for row in case["w_hex"]], np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), case["pixel"]
        # cosine-weighted about the normal: unit length, in the upper hemisphere
        n = np.cross(np.array(e1, np.float64), np.array(e2, np.float64))
        n /= np.linalg.norm(n)
        assert np.allclose(np.linalg.norm(got.astype(np.float64), axis=1), 1.0, atol=1e-5) and (got @ n > -1e-6).all()
