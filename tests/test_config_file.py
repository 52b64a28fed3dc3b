"""openvr_mod.cfg parsing with the defaulting rules of Config::Load (Config.h:30-63).  No GPU."""
import os

import numpy as np
import pytest

import openvr_fsr_amd as A

CFG = """
{
  // comment line
  "fsr": {
    "enabled": true,   // trailing comment
    "useNIS": false,
    /* block comment */
    "renderScale": 0.77,
    "sharpness": 0.9,
    "radius": 0.5,
    "applyMIPBias": true,
    "debugMode": false,
    "hotkeys": { "enabled": true, "requireCtrl": false, "toggleUseNIS": 112, "captureOutput": 118 }
  }
}
"""


def test_parse_shipped_style_config():
    rc, cfg = A.config_from_json(CFG)
    assert rc == 0
    assert cfg.fsr_enabled == 1 and cfg.use_nis == 0 and cfg.debug_mode == 0
    assert abs(cfg.render_scale - 0.77) < 1e-7 and abs(cfg.sharpness - 0.9) < 1e-7 and cfg.radius == 0.5
    assert A.output_size(cfg, 1728, 1920) == (int(np.float32(1728) / np.float32(0.77)), int(np.float32(1920) / np.float32(0.77)))


def test_defaults_follow_config_load_not_the_struct():
    rc, cfg = A.config_from_json('{"fsr": {"enabled": true}}')
    assert rc == 0 and cfg.fsr_enabled == 1
    assert cfg.sharpness == 1.0          # fsr.get("sharpness", 1.0), Config.h:39 -- NOT the struct default 0.75
    assert cfg.render_scale == 1.0 and cfg.radius == 0.5 and cfg.use_nis == 0
    rc, cfg = A.config_from_json('{"fsr": {"sharpness": -3, "useNIS": true, "debugMode": 1}}')
    assert rc == 0 and cfg.sharpness == 0.0 and cfg.use_nis == 1 and cfg.debug_mode == 1 and cfg.fsr_enabled == 0
    rc, cfg = A.config_from_json("{}")
    assert rc == 0 and cfg.fsr_enabled == 0 and cfg.sharpness == 1.0


def test_unreadable_config_keeps_struct_defaults():
    rc, cfg = A.config_from_json('{"fsr": {"enabled": tru')
    assert rc == 1
    assert cfg.fsr_enabled == 0 and cfg.sharpness == 0.75 and cfg.render_scale == 1.0   # Config.h:11-15


@pytest.mark.skipif(not os.path.exists("/root/reference/src/openvr_mod.cfg"), reason="reference tree not present")
def test_reference_shipped_file():
    rc, cfg = A.config_from_json(open("/root/reference/src/openvr_mod.cfg").read())
    assert rc == 0 and cfg.fsr_enabled == 1 and cfg.use_nis == 0
    assert abs(cfg.render_scale - 0.77) < 1e-7 and abs(cfg.sharpness - 0.9) < 1e-7 and cfg.radius == 0.5 and cfg.debug_mode == 0
