"""Host-side sanitizer job: `make asan-host` builds the library with its
HOST code under AddressSanitizer + UndefinedBehaviorSanitizer and runs
tests/asan_host_driver.cpp against it (plan building, arena carving, descriptor walks, argument
checks for the ViT-B shapes; no GPU needed).  Any sanitizer report fails the make target."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_and_ubsan():
  if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
    pytest.skip("no hipcc")
  r = subprocess.run(["make", "-C", os.path.join(ROOT, "precondition_amd", "csrc"), "-f", "Makefile.asan", "-j",
                      str(min(8, os.cpu_count() or 4)), "asan-host"],
                     capture_output=True, text=True, timeout=900)
  tail = (r.stdout + r.stderr)[-3000:]
  assert r.returncode == 0, tail
  assert "asan host driver: ok" in r.stdout, tail
  assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
