#!/usr/bin/env python3
"""tests/golden/make_natural.py -- cut the NATURAL-CONTENT fixtures (round 6) and pin the oracle on them.

Until round 6 every image that went through the strict, product and audit paths was synthetic and builder-generated (tests/synth.py and the
families derived from it).  The guard's listed-pixel rate, EASU's 1/32768 zero-direction branch, its dering clamp on real edges
(ffx_fsr1.h:388-437) and NVScaler's edge classification (NIS_Scaler.h:176-293) all depend on content statistics, so three 256x256 crops of
images that exist offline in this container are committed as test data (inputs only -- category "fixture": data, not source):

  natural_cube.npz     /root/reference/samples/bin/cube_texture.png (the reference's own sample texture), window (1100, 1500): rendered game art
  natural_portal.npz   /root/reference/samples/unity_keyboard_sample/Assets/SteamVR/Textures/portalworkshop.png, window (300, 230):
                       a rendered UI -- anti-aliased text, a logo, flat panels with 1-px rules (what a VR overlay looks like)
  natural_hopper.npz   matplotlib's sample_data/grace_hopper.jpg (a photograph; JPEG noise, skin, fabric), window (128, 32)

Each file holds `rgba8` [256, 256, 4] (alpha 255) and its provenance.  natural_golden.json holds, per fixture, the SHA-256 of what the CPU
oracle (oracle/, pinned to the reference compiled through the shim) returns for it at 256 -> 341 (the C2 ratio): the EASU pass in UNORM8, the
EASU -> UNORM8 -> RCAS pipeline at sharpness 0.9, NVScaler in UNORM8 -- tests/test_oracle.py checks the oracle against them on every CPU run, and
where /root/reference exists also the reference build itself (oracle/_ref).

    python tests/golden/make_natural.py          (needs /root/reference and matplotlib's sample data: this container)"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

SOURCES = {
    "cube": ("/root/reference/samples/bin/cube_texture.png", 1100, 1500),
    "portal": ("/root/reference/samples/unity_keyboard_sample/Assets/SteamVR/Textures/portalworkshop.png", 300, 230),
    "hopper": (None, 128, 32),  # matplotlib sample data, resolved below
}


def golden_hashes(rgba8):
    """what tests/test_oracle.py recomputes: see natural_digests() there (kept in one place: tests/natural.py)"""
    from tests import natural
    return natural.oracle_digests(rgba8)


def main():
    from PIL import Image
    import matplotlib
    out = {}
    for name, (path, x0, y0) in SOURCES.items():
        if path is None:
            path = os.path.join(os.path.dirname(matplotlib.__file__), "mpl-data", "sample_data", "grace_hopper.jpg")
        rgb = np.asarray(Image.open(path).convert("RGB"))
        crop = rgb[y0:y0 + 256, x0:x0 + 256]
        assert crop.shape == (256, 256, 3), crop.shape
        rgba8 = np.concatenate([crop, np.full((256, 256, 1), 255, np.uint8)], axis=2)
        np.savez_compressed(os.path.join(HERE, "natural_%s.npz" % name), rgba8=rgba8, source=np.array(os.path.basename(path)), window=np.array([x0, y0, 256, 256]))
        out[name] = golden_hashes(rgba8)
        print(name, rgba8.shape, "std %.1f" % rgba8[..., :3].std(), out[name])
    json.dump(out, open(os.path.join(HERE, "natural_golden.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
