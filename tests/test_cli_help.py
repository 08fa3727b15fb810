"""`msk144hipdecoder --help` prints the reference's help text verbatim (main.cu:58-67, snapshot in tests/golden/ref_constants.json),
then its own additions.  Needs no GPU: --help returns before any HIP call."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "msk144cudecoder_amd", "msk144hipdecoder")


def test_help_text_is_the_references_verbatim():
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_constants.json")))["help_lines"]
    p = subprocess.run([EXE, "--help"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0
    lines = p.stdout.split("\n")
    want = [l.replace("{prog}", EXE) for l in ref]
    assert lines[:len(want)] == want
    extra = "\n".join(lines[len(want):])
    for flag in ("--inputs=", "--inputs-file=", "--interleaved=", "--connect-timeout-ms=", "--timing", "--hop-timeout-ms=", "--skip-wav-header", "--reference-decode-cache", "--strict-decode", "--print-bits", "--device=", "--every-slot", "--max-results="):
        assert flag in extra
