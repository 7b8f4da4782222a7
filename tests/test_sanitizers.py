"""CPU sanitizer builds ("race detection / sanitizers"): the library's host-only logic (slab allocator and block
cache, bracket state machine, multi-device orchestration: storm_amd/csrc/hostlogic/), the host mirror and the oracle, compiled
by g++ with -fsanitize=address,undefined and with -fsanitize=thread and run here on the CPU (tests/sanitize/).  The binaries
are rebuilt only when their sources changed (about a minute from scratch)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_logic_host_mirror_and_oracle_are_clean_under_asan_ubsan_and_tsan():
    lib = os.path.join(ROOT, "storm_amd", "lib", "libstorm_hip.so")
    assert os.path.exists(lib), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    env = dict(os.environ)
    env.pop("STORM_SANITIZE_EXPECT_GPU", None)
    env["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0"
    env["TSAN_OPTIONS"] = "halt_on_error=1"
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "sanitize"), "-j4", "run"], capture_output=True, text=True,
                       env=env, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "sanitizer builds: all clean" in r.stdout
    for marker in ("hostlogic stress: all clean", "oracle driver: all clean", "host driver: all clean"):
        assert marker in r.stdout
