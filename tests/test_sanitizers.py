"""The SANITIZER build ("-fsanitize=address on host tests"; VERDICT r5 next #1b).

tools/build_asan.sh compiles the library's host translation units and the two plain-C drivers with AddressSanitizer +
UndefinedBehaviorSanitizer (ab/asan.so, ab/headless_asan, ab/bench_node_asan; -fno-sanitize-recover: the first report aborts).
  * CPU: the host-side tests (constants, C ABI error paths, config file / capture parsers, incl. the out-of-range float probes of
    test_capi.py) run in a subprocess against ab/asan.so;
  * GPU: the two C drivers run against it on the device; a slice of the GPU suite (launch manager: back-to-back batches, pair_submit,
    masked lists, formats, fuzz) runs against ab/asan_gcc.so, the same translation units under GCC's sanitizers (see build_asan.sh for why two).
Device code is not instrumented (GPU ASan needs xnack+ code objects): tests/test_gpu_bounds.py covers that side."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_TESTS = ["tests/test_capi.py", "tests/test_constants.py", "tests/test_config_file.py"]


SAN = "-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined"   # tools/build_asan.sh


def _build():
    # (a fresh stamp = built from these sources: the object directories do not travel to the GPU box, so `make` there would start from nothing)
    products = [os.path.join(ROOT, "ab", f) for f in ("asan.so", "asan_gcc.so", "headless_asan", "bench_node_asan", "thread_stress_asan")]
    fresh = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "variant_fresh.py"), "asan", SAN]).returncode == 0
    if not (fresh and all(os.path.exists(f) for f in products)):
        r = subprocess.run([os.path.join(ROOT, "tools", "build_asan.sh")], capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            pytest.skip("sanitizer build unavailable here: " + (r.stderr or r.stdout)[-300:])
    rt = subprocess.run([os.path.join(ROOT, "tools", "build_asan.sh"), "--runtime"], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(rt), rt
    return rt


def _env(rt):
    return dict(os.environ, OVRFSR_LIB=os.path.join(ROOT, "ab", "asan.so"), LD_PRELOAD=rt, PYTHONPATH=ROOT,
                ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=86", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=87")


def _clean(r):
    out = r.stdout + r.stderr
    assert "runtime error:" not in out and "ERROR: AddressSanitizer" not in out and "UndefinedBehaviorSanitizer" not in out, out[-3000:]
    assert r.returncode == 0, out[-3000:]


def test_host_tests_clean_under_asan_ubsan():
    rt = _build()
    if os.environ.get("OVRFSR_LIB", "").endswith("asan.so"):
        pytest.skip("already inside the sanitizer run")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + HOST_TESTS, capture_output=True, text=True, timeout=900, env=_env(rt), cwd=ROOT)
    _clean(r)
    assert " passed" in r.stdout, r.stdout[-500:]


def test_config_parser_survives_mutated_files_under_asan_ubsan():
    """tests/debug/json_fuzz.py: 40 000 mutants of the shipped openvr_mod.cfg's shape (and of its broken forms) through ovrfsr_config_from_json in
    exact-length buffers without a terminating NUL -- status OK or INVALID_ARGUMENT only, defaults after a refusal, finite and clamped fields
    after a parse, no report from either sanitizer (900 000 mutants over three seeds in round 6: profiles/r06_bounds.txt section 2)."""
    rt = _build()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "debug", "json_fuzz.py"), "40000", "7"], capture_output=True, text=True, timeout=900, env=_env(rt), cwd=ROOT)
    _clean(r)
    assert "0 violations" in r.stdout, r.stdout[-500:]


def test_sanitizer_build_is_live(tmp_path):
    """the instrumented library does report: the un-fixed conversion of round 5 (a negative float cast to uint32) is re-created in a probe
    translation unit compiled with the same flags; UBSan must name it -- otherwise a clean run above says nothing"""
    rt = _build()
    clang = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "clang")
    src = str(tmp_path / "ubsan_probe.c")
    open(src, "w").write("#include <stdint.h>\n#include <stdio.h>\nint main(int c, char **v) { volatile float r = -1.0f * c; uint32_t u = (uint32_t)(0.5f * r * 2492); printf(\"%u\\n\", u); return 0; }\n")
    exe = str(tmp_path / "ubsan_probe")
    subprocess.check_call([clang, "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "runtime error" in r.stderr and "outside the range of representable values" in r.stderr, r.stderr


@pytest.mark.gpu
def test_c_drivers_and_launch_manager_clean_under_asan_ubsan(gpu, tmp_path):
    rt = _build()
    env = _env(rt)
    env.pop("LD_PRELOAD")  # the C drivers link the runtime themselves
    for args in (["headless_asan", "-", str(tmp_path / "eye.ppm")], ["headless_asan", "-", str(tmp_path / "eye.dds"), "--pair"],
                 ["bench_node_asan", "--gpus", "2", "--oversubscribe", "--pairs", "2", "--steps", "3", "--warmup", "1"],
                 ["bench_node_asan", "--gpus", "2", "--oversubscribe", "--pairs", "2", "--steps", "2", "--warmup", "1", "--fused", "--radius", "0.5"],
                 # every launch form from four threads at once, pair_submit in R,L order among them: UBSan found the batch of two's modular offsets
                 # added to a POINTER there (undefined beyond the object; now integer arithmetic on the address, postprocessor.cpp at_offset)
                 ["thread_stress_asan", "--threads", "4", "--rounds", "2"]):
        r = subprocess.run([os.path.join(ROOT, "ab", args[0])] + args[1:], capture_output=True, text=True, timeout=600, env=env)
        _clean(r)
    # LeakSanitizer over 400 ctx life cycles: the HIP runtime keeps a few kilobytes until exit (reported, not ours); nothing allocated under an
    # ovrfsr:: frame may be among the leaks
    r = subprocess.run([os.path.join(ROOT, "ab", "thread_stress_asan"), "--threads", "4", "--rounds", "10"], capture_output=True, text=True, timeout=900,
                       env=dict(env, ASAN_OPTIONS="detect_leaks=1:exitcode=0", LSAN_OPTIONS="exitcode=0"))
    assert "runtime error:" not in r.stderr and "ERROR: AddressSanitizer: heap" not in r.stderr, r.stderr[-3000:]
    assert "ovrfsr" not in r.stderr, r.stderr[-4000:]
    # the launch manager from Python: ctx life cycle, lazy rebuilds, tile lists, pair_submit, every format, capture writers -- against the
    # GCC-sanitized build (ab/asan_gcc.so: ROCm's clang ASan runtime cannot live in a process that holds torch's HIP runtime, build_asan.sh)
    rt_gcc = subprocess.run([os.path.join(ROOT, "tools", "build_asan.sh"), "--runtime-gcc"], capture_output=True, text=True).stdout.strip()
    env = dict(_env(rt), OVRFSR_LIB=os.path.join(ROOT, "ab", "asan_gcc.so"), LD_PRELOAD=rt_gcc)
    # (ASan's dlopen interceptor makes libasan the caller of every dlopen: torch's RUNPATH-relative loads need the directory spelt out)
    import importlib.util
    tlib = os.path.join(os.path.dirname(importlib.util.find_spec("torch").origin), "lib")
    env["LD_LIBRARY_PATH"] = tlib + (":" + env["LD_LIBRARY_PATH"] if env.get("LD_LIBRARY_PATH") else "")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "gpu", "tests/test_gpu_back_to_back.py", "tests/test_gpu_formats.py",
                        "tests/test_gpu_fuzz.py", "-k", "not real_shards and not half_intermediate_is_the_strict"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    _clean(r)
    assert " passed" in r.stdout, r.stdout[-500:]
