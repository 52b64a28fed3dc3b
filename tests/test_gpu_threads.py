"""Race detection (the reference has none -- one render thread, one D3D11 immediate context).

include/openvr_fsr_amd.h promises "one ctx per device; not thread-safe; distinct ctxs are independent".  tests/debug/thread_stress.c turns the
second half into a test: T host threads on one device, each creating, driving and destroying its own ctxs through every launch form (two
kernels, sorted tiles + the concurrent outside kernel, NVScaler, fused, pair_submit, debug-mode timing, size changes, set_config / reset, the
strict build) at the same time, every downloaded result compared with the checksum the same job gave when it ran alone.
  * the product library runs it;
  * the ThreadSanitizer build of the host translation units (tools/build_tsan.sh -> ab/tsan.so, ab/thread_stress_tsan) runs it and must stay
    silent -- reports that lie inside the uninstrumented HIP / HSA runtimes are dropped by tools/tsan.supp;
  * the same instrumented binary, told to break the first half of the promise (--misuse: two threads on ONE ctx), must make the detector
    report races inside ovrfsr::PostProcessor -- otherwise the silence above proves nothing."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = "-fsanitize=thread -fno-omit-frame-pointer -gline-tables-only"   # tools/build_tsan.sh


def _product_driver():
    exe = os.path.join(ROOT, "tests", "debug", "thread_stress")
    if not os.path.exists(exe):
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-pthread", "-D_POSIX_C_SOURCE=200809L", "-D__HIP_PLATFORM_AMD__", exe + ".c",
                               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(rocm, "include"), "-L" + os.path.join(ROOT, "openvr_fsr_amd"),
                               "-lopenvr_fsr_amd", "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lm", "-Wl,-rpath,$ORIGIN/../../openvr_fsr_amd",
                               "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe])
    return exe


def _c_driver(name, extra=()):
    exe = os.path.join(ROOT, "tests", "debug", name)
    if not os.path.exists(exe):
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-D_POSIX_C_SOURCE=200809L", "-D__HIP_PLATFORM_AMD__", *extra, exe + ".c", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(rocm, "include"), "-L" + os.path.join(ROOT, "openvr_fsr_amd"), "-lopenvr_fsr_amd", "-L" + os.path.join(rocm, "lib"),
                               "-lamdhip64", "-lm", "-Wl,-rpath,$ORIGIN/../../openvr_fsr_amd", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe])
    return exe


def _tsan_driver():
    exe = os.path.join(ROOT, "ab", "thread_stress_tsan")
    fresh = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "variant_fresh.py"), "tsan", SAN]).returncode == 0
    if not (fresh and os.path.exists(exe)):
        r = subprocess.run([os.path.join(ROOT, "tools", "build_tsan.sh")], capture_output=True, text=True, timeout=1500)
        if r.returncode != 0 or not os.path.exists(exe):
            pytest.skip("ThreadSanitizer build unavailable here: " + (r.stderr or r.stdout)[-300:])
    return exe


def _tsan_env():
    return dict(os.environ, TSAN_OPTIONS="suppressions=%s:exitcode=66:halt_on_error=0:second_deadlock_stack=1" % os.path.join(ROOT, "tools", "tsan.supp"))


@pytest.mark.gpu
def test_distinct_ctxs_on_distinct_threads_are_independent(gpu):
    r = subprocess.run([_product_driver(), "--threads", "6", "--rounds", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all checksums equal the serial run" in r.stdout, (r.stdout + r.stderr)[-2000:]


@pytest.mark.gpu
def test_thread_stress_is_silent_under_thread_sanitizer(gpu):
    r = subprocess.run([_tsan_driver(), "--threads", "4", "--rounds", "2"], capture_output=True, text=True, timeout=900, env=_tsan_env())
    out = r.stdout + r.stderr
    assert "ThreadSanitizer" not in out, out[-4000:]
    assert r.returncode == 0 and "all checksums equal the serial run" in r.stdout, out[-2000:]


@pytest.mark.gpu
def test_thread_sanitizer_build_is_live(gpu):
    """two threads on ONE ctx (forbidden by the header): the instrumented library must say so, through the same suppression file"""
    r = subprocess.run([_tsan_driver(), "--misuse"], capture_output=True, text=True, timeout=600, env=_tsan_env())
    out = r.stdout + r.stderr
    assert "WARNING: ThreadSanitizer: data race" in out and "ovrfsr::PostProcessor" in out, out[-3000:]
    assert r.returncode == 66, r.returncode


@pytest.mark.gpu
def test_a_pending_hip_error_of_the_callers_does_not_fail_the_next_apply(gpu):
    """HIP keeps a thread's last error until it is read, and a launch reports failure only there: the host's own failed hipMalloc (handled
    through its return value) used to fail the next frame as "EASU launch: out of memory".  Every launch_* now clears the state first
    (csrc/fsr_launch.h launch_fresh); tests/debug/stale_error.c: apply, provoke the error, apply and reset + rebuild -- same pixels."""
    r = subprocess.run([_c_driver("stale_error")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "neither failed nor changed" in r.stdout, (r.stdout + r.stderr)[-2000:]


def test_thread_stress_driver_compiles_warning_free(tmp_path):
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    r = subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-Wno-unused-result", "-pthread", "-D_POSIX_C_SOURCE=200809L", "-D__HIP_PLATFORM_AMD__", "-c",
                        os.path.join(ROOT, "tests", "debug", "thread_stress.c"), "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(rocm, "include"),
                        "-o", str(tmp_path / "thread_stress.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
