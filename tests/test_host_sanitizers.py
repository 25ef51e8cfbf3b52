"""AddressSanitizer + UBSan over the host half of the drop-in library (CPU
build only: GPU sanitizers are not available on the pool).  The control-plane
parsers see attacker-controlled bytes, and the reference's off-by-one array
bounds (src/h264_syntax.h:43,545,566,688) must not come along."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "libh264_amd", "lib")


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_control_plane_fuzz_under_asan_ubsan(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libh264_amd_engine.so")):
        from libh264_amd import build
        build.build_all()
    lib = tmp_path / "libh264.so"
    common = ["-std=gnu99", "-D_GNU_SOURCE", "-O1", "-g",
              "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
              "-I" + os.path.join(ROOT, "include")]
    subprocess.check_call(
        ["gcc"] + common + ["-DH264_API_EXPORTS", "-fvisibility=hidden",
                            "-fPIC", "-shared",
                            "-I" + os.path.join(ROOT, "libh264_amd", "csrc")] +
        sorted(glob.glob(os.path.join(ROOT, "libh264_amd", "csrc", "host",
                                      "*.c"))) +
        ["-L" + LIBDIR, "-lh264_amd_engine", "-lpthread",
         "-Wl,-rpath," + LIBDIR, "-o", str(lib)])
    exe = tmp_path / "fuzz_ctl"
    subprocess.check_call(
        ["gcc"] + common + [os.path.join(ROOT, "tests", "helpers",
                                         "fuzz_ctl.c"),
                            "-L" + str(tmp_path), "-lh264",
                            "-Wl,-rpath," + str(tmp_path), "-o", str(exe)])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([str(exe), "60000"], env=env, capture_output=True,
                         text=True, timeout=600)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-3000:]
    assert "runtime error" not in text, text[-3000:]
    assert "AddressSanitizer" not in text, text[-3000:]
    assert "ok " in out.stdout
