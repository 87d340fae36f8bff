"""The host-only sources (audio files, XML configurations, match selector) built with g++ under AddressSanitizer and
UndefinedBehaviorSanitizer and fed damaged files, damaged documents and malformed candidate lists
(tools/host_sanitize.cpp); the threads that read database files ahead of their use under ThreadSanitizer
(tools/readahead_sanitize.cpp).  CPU only; skipped where the compiler has no sanitizer runtime."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_sources_are_clean_under_the_sanitizers(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    probe = tmp_path / "p.cpp"
    probe.write_text("int main() { return 0; }\n")
    if subprocess.run(["g++", "-fsanitize=address,undefined", str(probe), "-o", str(tmp_path / "p")],
                      capture_output=True).returncode != 0:
        pytest.skip("g++ without sanitizer runtimes")
    if subprocess.run(["g++", "-fsanitize=thread", str(probe), "-o", str(tmp_path / "p")],
                      capture_output=True).returncode != 0:
        pytest.skip("g++ without the thread sanitizer's runtime")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "host_sanitize.sh")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "host code clean under the sanitizers" in r.stdout
    assert "read-ahead clean under the thread sanitizer" in r.stdout
