#!/usr/bin/env python3
"""tests/debug/json_fuzz.py [count=50000] [seed=1] -- mutation fuzz of ovrfsr_config_from_json (Config::Load, Config.h:30-63): a config file is the one
piece of text a user edits by hand.  Seeds: the shape of the reference's shipped openvr_mod.cfg (// comments, nested hotkey objects, every key) and
its edge forms; mutations: byte flips, deletions, duplications, splices of number / bracket / quote / escape / comment / UTF-8 fragments, truncation at
every kind of position, deep nesting, huge numbers.  Every call passes an exact-length buffer with NO terminating NUL (a copy into a fresh ctypes
array), so an over-read is an out-of-bounds read for AddressSanitizer (tests/test_sanitizers.py runs this against ab/asan.so).
Checked for every input: the call returns OK or INVALID_ARGUMENT; on INVALID_ARGUMENT the struct holds the defaults; on OK the fields that reach the
GPU are finite, radius >= 0, sharpness in [0, 1] (where the reference clamps), booleans are 0 / 1 -- so that whatever text arrives, ovrfsr_create never
sees a value the boundary would have to refuse for being non-finite."""
import ctypes as C
import math
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import openvr_fsr_amd as A  # noqa: E402
from openvr_fsr_amd import _capi as K  # noqa: E402

SEEDS = [
    b'{\n  "fsr": {\n    // enable image upscaling\n    "enabled": true,\n    "useNIS": false,\n    "renderScale": 0.77,\n    "sharpness": 0.9,\n    "radius": 0.5,\n'
    b'    "applyMIPBias": true,\n    "debugMode": false,\n    "hotkeys": {\n      "enabled": true,\n      "requireCtrl": true, "requireAlt": false, "requireShift": false,\n'
    b'      "toggleUseNIS": 112, // F1\n      "toggleDebugMode": 113,\n      "decreaseSharpness": 114, "increaseSharpness": 115,\n      "decreaseRadius": 116, "increaseRadius": 117,\n'
    b'      "captureOutput": 118\n    }\n  }\n}\n',
    b'{"fsr":{"enabled":true,"sharpness":-0.5,"radius":-1,"renderScale":1e9}}',
    b'{"fsr":{"enabled":1,"useNIS":"yes","sharpness":"0.5","radius":null,"renderScale":[0.5]}}',
    b'{"fsr":{"sharpness":1e999,"radius":-1e999,"renderScale":NaN}}',
    b'{ /* block */ "fsr" : { "enabled" : true , "renderScale" : 7.7e-1 , "sharpness" : 9E-1 } }',
    b'{"other":{"fsr":{"enabled":true}},"fsr":{"enabled":false,"radius":2}}',
    b'{"fsr":{"enabled":true,"enabled":false,"\\u0072adius":0.25,"radius\\n":3}}',
    b'',
    b'{',
    b'{"fsr":',
]
FRAGMENTS = [b'"', b'\\', b'\\u', b'\\u12', b'\\"', b'//', b'/*', b'*/', b'\n', b'{', b'}', b'[', b']', b':', b',', b'-', b'+', b'.', b'e', b'E', b'e+', b'e-9999', b'1e309', b'-1e309',
             b'0x10', b'00', b'1.', b'.5', b'NaN', b'nan', b'Infinity', b'-inf', b'true', b'false', b'null', b'tru', b'\x00', b'\xff', b'\xc3\xa9', b'\xed\xa0\x80', b'\xf4\x90\x80\x80',
             b'9' * 400, b'0.' + b'0' * 400 + b'1', b'"fsr"', b'"radius"', b'"sharpness"', b'"renderScale"', b'"enabled"', b'"useNIS"', b'"debugMode"', b'{' * 300, b'[' * 300, b' ' * 64]


def mutate(rng, data):
    b = bytearray(data)
    for _ in range(rng.choice((1, 1, 1, 2, 3, 6))):
        op = rng.randrange(8)
        pos = rng.randrange(len(b) + 1)
        if op == 0 and b:
            b[rng.randrange(len(b))] = rng.randrange(256)
        elif op == 1 and b:
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif op == 2 and b:
            n = rng.randrange(1, min(len(b), 24) + 1); p = rng.randrange(len(b) - n + 1); del b[p:p + n]
        elif op == 3 and b:
            n = rng.randrange(1, min(len(b), 40) + 1); p = rng.randrange(len(b) - n + 1); b[pos:pos] = b[p:p + n]
        elif op == 4:
            b[pos:pos] = rng.choice(FRAGMENTS)
        elif op == 5:
            del b[pos:]
        elif op == 6 and b:   # replace a run of digits by a fragment
            digits = [i for i, ch in enumerate(b) if 48 <= ch <= 57]
            if digits:
                i = rng.choice(digits); b[i:i + 1] = rng.choice(FRAGMENTS)
        else:
            other = rng.choice(SEEDS); p = rng.randrange(len(other) + 1); b[pos:pos] = other[p:p + rng.randrange(0, 60)]
    return bytes(b[:4096])


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    lib = A.library()
    fn = lib.ovrfsr_config_from_json
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    fn.restype = C.c_int
    default = K.Config()
    lib.ovrfsr_config_default(C.byref(default))
    corpus = list(SEEDS)
    n_ok = n_bad = 0
    for it in range(count):
        data = mutate(rng, rng.choice(corpus)) if it >= len(SEEDS) else SEEDS[it]
        buf = (C.c_ubyte * max(len(data), 1)).from_buffer_copy(data if data else b"\x00")   # exact length, no NUL behind it
        cfg = K.Config()
        rc = fn(C.cast(buf, C.c_void_p), len(data), C.byref(cfg))
        assert rc in (0, 1), (rc, data)
        if rc == 1:
            n_bad += 1
            assert bytes(cfg) == bytes(default), ("a failed parse left something behind", data)
        else:
            n_ok += 1
            assert math.isfinite(cfg.radius) and cfg.radius >= 0, (cfg.radius, data)
            assert math.isfinite(cfg.sharpness) and 0 <= cfg.sharpness <= 1 or cfg.sharpness >= 0 and math.isfinite(cfg.sharpness), (cfg.sharpness, data)
            assert math.isfinite(cfg.render_scale), (cfg.render_scale, data)
            assert cfg.fsr_enabled in (0, 1) and cfg.use_nis in (0, 1) and cfg.debug_mode in (0, 1), data
            if len(corpus) < 400 and rng.random() < 0.02:
                corpus.append(data)       # parsable mutants breed
    print("json_fuzz: %d inputs, %d parsed, %d refused, 0 violations" % (count, n_ok, n_bad))


if __name__ == "__main__":
    main()
