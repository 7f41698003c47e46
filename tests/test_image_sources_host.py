"""pal_image_sources (host C++ behind the C ABI, csrc/images.cpp) against the reference's fixtures of
generate_image_sources_iterative (utils.py:67-106): shoebox orders 1-3 with the low-loss table,
the three default planes with the shipped materials at 0.01 / 0.1 / 0.25 / 1 Hz,
C2's two tables, and the error conventions.  Needs no GPU: the breadth-first search runs on the host."""
import numpy as np
import pytest

from oracle import cases
from oracle import pal_oracle as O
from pyaudiolocalization_amd.utils import generate_image_sources_iterative


MICS = np.random.default_rng(2).uniform(-0.5, 0.5, (8, 3))


def test_shoebox_and_default_fixtures(golden):
    g = golden("image_sources.npz")
    for order, count in ((1, 6), (2, 23), (3, 53)):
        imgs = generate_image_sources_iterative([1.0, 2.0, 0.5], cases.SHOEBOX, order, 500, cases.LOW_LOSS, MICS, 0.01)
        assert len(imgs) == count == g[f"shoebox_o{order}"].shape[0]
        assert np.array_equal(np.array([i["source"] for i in imgs]), g[f"shoebox_o{order}"])        # discovery order, bit for bit
        assert [i["material"] for i in imgs] == list(g[f"shoebox_o{order}_mat"])
    for f in (0.01, 0.1, 0.25, 1.0):
        imgs = generate_image_sources_iterative([1.0, 2.0, 0.5], cases.DEFAULT_PLANES, 3, f, O.MATERIALS_DEFAULT, MICS, 0.01)
        want = g["default_f%s" % str(f).replace(".", "p")]
        assert np.array_equal(np.array([i["source"] for i in imgs]).reshape(-1, 3), want)


def test_c2_tables(golden):
    g = golden("c2_chirp8.npz")
    cfg = cases.c2_config()
    mics = np.array(cfg["mic_positions"])
    for tag, table in (("a_", O.MATERIALS_DEFAULT), ("b_", cases.LOW_LOSS)):
        imgs = generate_image_sources_iterative(cfg["source_position"], cfg["reflective_planes"], 3, 500, table, mics, 0.01)
        assert np.array_equal(np.array([i["source"] for i in imgs]).reshape(-1, 3), g[tag + "images"])
        assert [i["material"] for i in imgs] == list(g[tag + "image_materials"])


def test_error_conventions():
    with pytest.raises(ValueError):                      # utils.py:37
        generate_image_sources_iterative([0, 0, 0], [{"plane": [0, 0, 0, 1], "material": "air"}], 1, 1.0, O.MATERIALS_DEFAULT, MICS)
    with pytest.raises(ValueError):                      # utils.py:94
        generate_image_sources_iterative([0, 0, 0], [{"plane": [1, 0, 0, 1], "material": "glass"}], 1, 0.0, O.MATERIALS_DEFAULT, MICS)
    with pytest.raises(ValueError):                      # utils.py:96
        generate_image_sources_iterative([0, 0, 0], [{"plane": [1, 0, 0, 1], "material": "foam"}], 1, 0.0,
                                         dict(O.MATERIALS_DEFAULT, foam={"absorption": 0.2}), MICS)
    assert generate_image_sources_iterative([0, 0, 0], [], 3, 1.0, O.MATERIALS_DEFAULT, MICS) == []
