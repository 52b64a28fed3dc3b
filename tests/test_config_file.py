"""openvr_mod.cfg parsing with the defaulting rules of Config::Load (Config.h:30-63).  No GPU."""
import os

import numpy as np
import pytest

import openvr_fsr_amd as A

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


def test_config_parses_the_same_under_a_comma_decimal_locale(tmp_path):
    """The library lives inside a host process, and hosts call setlocale(LC_ALL, "").  Where the decimal separator is a comma, atof("0.77") is 0:
    renderScale, sharpness and radius all read 0 (round 6, found by probing; jsoncpp, the reference's parser, is locale-independent).  The parser now
    converts numbers in an explicit "C" locale.  No such locale is installed here, so the test compiles one (LC_NUMERIC with decimal_point ",") with
    localedef and loads it through LOCPATH in a subprocess; first it shows that the locale really does that to atof."""
    import shutil
    import subprocess
    import sys
    if not shutil.which("localedef"):
        pytest.skip("no localedef on this box")
    (tmp_path / "comma.src").write_text('LC_NUMERIC\ndecimal_point "<U002C>"\nthousands_sep "<U002E>"\ngrouping 3;3\nEND LC_NUMERIC\n')
    (tmp_path / "ascii.cm").write_text("<code_set_name> ANSI_X3.4-1968\n<comment_char> %\n<escape_char> /\n<mb_cur_max> 1\nCHARMAP\n" +
                                       "".join("<U%04X>     /x%02x         CHAR%d\n" % (i, i, i) for i in range(128)) + "END CHARMAP\n")
    subprocess.run(["localedef", "-c", "-i", str(tmp_path / "comma.src"), "-f", str(tmp_path / "ascii.cm"), str(tmp_path / "xx_XX")], capture_output=True)
    if not (tmp_path / "xx_XX" / "LC_NUMERIC").exists():
        pytest.skip("localedef could not build the test locale")
    code = ("import ctypes, json, locale, sys\n"
            "sys.path.insert(0, %r)\n"
            "locale.setlocale(locale.LC_ALL, 'xx_XX')\n"
            "libc = ctypes.CDLL(None); libc.atof.restype = ctypes.c_double\n"
            "from openvr_fsr_amd import _capi as K\n"
            "rc, c = K.config_from_json('{\"fsr\": {\"enabled\": true, \"renderScale\": 0.77, \"sharpness\": 0.9, \"radius\": 0.5, \"useNIS\": 1.5}}')\n"
            "print(json.dumps(dict(point=locale.localeconv()['decimal_point'], atof=libc.atof(b'0.77'), rc=rc, scale=c.render_scale, sharp=c.sharpness, radius=c.radius, nis=c.use_nis)))\n"
            % ROOT_DIR)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, LOCPATH=str(tmp_path)), timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    import json
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["point"] == "," and d["atof"] == 0.0                      # the hazard is live in that process
    assert d["rc"] == 0 and abs(d["scale"] - 0.77) < 1e-6 and abs(d["sharp"] - 0.9) < 1e-6 and d["radius"] == 0.5 and d["nis"] == 1
