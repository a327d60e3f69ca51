"""Sanitizer runs of everything in the library that needs no GPU (GPU AddressSanitizer is not available
on this pool, so this is the CPU side only):
  * tests/sanitize/host_harness.cpp = hmc.jl_amd/csrc/host_util.hpp (ScatterPool, partition_windows, plan_chunks: the very
    header the library compiles) + csrc/hmcg_csv.cpp (format_float, the table writers and their thread pool) under
    AddressSanitizer + UBSan and under ThreadSanitizer;
  * the oracle's C restatement under AddressSanitizer + UBSan (tests/sanitize/oracle_harness.c)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "sanitize")


def build_and_run(tmp_path, name, compiler, flags, sources, args=(), libs=()):
    exe = str(tmp_path / name)
    cmd = [compiler, "-O1", "-g", "-fno-omit-frame-pointer"] + flags + ["-o", exe] + sources + list(libs)
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe] + list(args), capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "ok" in r.stdout and "ERROR" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-3000:]


HOST_SRC = [os.path.join(SAN, "host_harness.cpp"), os.path.join(ROOT, "hmc.jl_amd", "csrc", "hmcg_csv.cpp")]


@pytest.mark.timeout(900)
def test_host_code_under_asan_ubsan(tmp_path):
    out = tmp_path / "csv"
    out.mkdir()
    build_and_run(tmp_path, "host_asan", "g++", ["-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                                                 "-I" + os.path.join(ROOT, "include")], HOST_SRC, [str(out)], ["-lpthread"])


@pytest.mark.timeout(900)
def test_host_code_under_tsan(tmp_path):
    out = tmp_path / "csv"
    out.mkdir()
    build_and_run(tmp_path, "host_tsan", "g++", ["-std=c++17", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include")], HOST_SRC,
                  [str(out)], ["-lpthread"])


@pytest.mark.timeout(900)
def test_oracle_under_asan_ubsan(tmp_path):
    build_and_run(tmp_path, "oracle_asan", "gcc", ["-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
                  [os.path.join(SAN, "oracle_harness.c"), os.path.join(ROOT, "oracle", "hmc_oracle.c")], libs=["-lm"])
