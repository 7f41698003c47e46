"""Acoustic material table (values of the reference's materials.py:3-15; data, not code)."""

material_properties = {
    "air": {"absorption": 0.01, "freq": 0.1},
    "wood": {"absorption": 0.05, "freq": 0.8},
    "metal": {"absorption": 0.1, "freq": 0.6},
}
