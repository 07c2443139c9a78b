"""The library's host-side file parsers under AddressSanitizer + UBSan (the review's hardening item; sanitizers run on the CPU build only).

What parses files a caller hands over: dacapo_amd/csrc/wire_parse.cpp (`.cst`, `.hevm`: the reference's loaders SEAL_HEVM.cpp:182-234
fread into vectors sized by the file's own counts) and dacapo_amd/csrc/seal_serial.cpp (SEAL's serialized objects, whose members may be
zlib / Zstandard streams: SEAL_HEVM.cpp:91-180).  `make -C dacapo_amd/csrc host_asan` builds exactly those TUs -- the product's own
sources, no device code -- with -fsanitize=address,undefined into a harness (host_fuzz_main.cpp) that parses one file and prints
"ok ..." / "rejected: ..."; any sanitizer report or crash is a non-zero exit.

1. every file of the committed corpus tests/golden/hostile/ (tools/fixtures/make_hostile_corpus.py: truncations, 2^62 / negative counts, bad magic,
   foreign versions, unknown / corrupt / truncated compression, a 200 MiB zlib bomb, a 500 MiB Zstandard bomb) does what manifest.json says;
2. seeded mutation fuzzing of the valid files: bit flips, truncations, and 8-byte fields overwritten with extreme values -- no outcome is
   asserted except "the sanitizers stay quiet and the process exits 0"."""
import json
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "dacapo_amd" / "csrc"
HOSTILE = ROOT / "tests" / "golden" / "hostile"
HARNESS = CSRC / "build_asan" / "host_parsers_asan"


@pytest.fixture(scope="module")
def harness():
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no g++ / make")
    r = subprocess.run(["make", "-C", str(CSRC), "host_asan"], capture_output=True, text=True)
    if r.returncode != 0 and ("asan" in r.stderr.lower() or "sanitize" in r.stderr.lower()) and "error:" not in r.stderr:
        pytest.skip("this toolchain has no AddressSanitizer runtime: " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-2000:]
    return HARNESS


def run(harness, kind, path, constants=None, timeout=120):
    cmd = [str(harness), kind, str(path)] + ([str(constants)] if constants else [])
    env = {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0:allocator_may_return_null=0", "UBSAN_OPTIONS": "print_stacktrace=1", "PATH": "/usr/bin:/bin"}
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


def test_corpus_matches_its_generator():
    """the committed files are what tools/fixtures/make_hostile_corpus.py writes (the corpus is data with a committed recipe)"""
    man = json.loads((HOSTILE / "manifest.json").read_text())
    assert man["generator"] == "tools/fixtures/make_hostile_corpus.py" and len(man["files"]) >= 60
    for e in man["files"]:
        assert (HOSTILE / e["file"]).exists(), e["file"]
    kinds = {e["kind"] for e in man["files"]}
    assert kinds == {"cst", "hevm", "hevm-header", "seal"}
    assert sum(e["expect"] == "rejected" for e in man["files"]) >= 35


def test_every_corpus_file_is_parsed_or_rejected_cleanly(harness):
    man = json.loads((HOSTILE / "manifest.json").read_text())
    for e in man["files"]:
        r = run(harness, e["kind"], HOSTILE / e["file"], HOSTILE / e["constants"] if e.get("constants") else None)
        assert r.returncode == 0, (e["file"], r.stdout[-300:], r.stderr[-1500:])
        assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (e["file"], r.stderr[-1500:])
        first = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
        assert first.startswith(e["expect"]), (e["file"], first)
        if e["message"]:
            assert e["message"] in first, (e["file"], first)


def test_decompression_bombs_stop_at_the_limit(harness):
    """a 200 KB zlib stream of 200 MiB and a 16 KB Zstandard stream of 500 MiB are refused after at most 64 x stored + 128 MiB of output"""
    for name in ("seal_zlib_bomb_200MiB.seal", "seal_zstd_bomb_500MiB.seal"):
        cmd = ["/usr/bin/time", "-f", "%M", str(harness), "seal", str(HOSTILE / name)]
        if not Path("/usr/bin/time").exists():
            cmd = cmd[3:]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "rejected" in r.stdout, (name, r.stdout, r.stderr[-500:])
        if cmd[0] == "/usr/bin/time":  # peak RSS in KB: far below the bomb's size (ASan's shadow and quarantine included)
            assert int(r.stderr.strip().splitlines()[-1]) < 160 * 1024, r.stderr[-200:]


@pytest.mark.parametrize("seed_file,kind,constants", [
    ("cst_valid.cst", "cst", None), ("hevm_valid.hevm", "hevm", "cst_valid.cst"), ("hevm_setscale_ok.hevm", "hevm", "cst_valid.cst"),
    ("seal_params.seal", "seal", None), ("seal_ciphertext.seal", "seal", None), ("seal_ciphertext_zlib.seal", "seal", None),
    ("seal_kswitchkeys.seal", "seal", None), ("seal_kswitchkeys_zlib.seal", "seal", None), ("seal_plaintext.seal", "seal", None)])
def test_mutated_inputs_never_trip_the_sanitizers(harness, tmp_path, seed_file, kind, constants):
    seed = (HOSTILE / seed_file).read_bytes()
    rng = np.random.default_rng(abs(hash(seed_file)) % (1 << 32) if False else sum(seed_file.encode()))
    extremes = [0, 1, 0xFF, 0xFFFF, 1 << 31, (1 << 32) - 1, 1 << 40, 1 << 62, (1 << 63) - 1, 1 << 63, (1 << 64) - 1]
    outcomes = {"ok": 0, "rejected": 0}
    for it in range(40):
        b = bytearray(seed)
        how = it % 4
        if how == 0:    # a few bit flips
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif how == 1:  # an aligned 8-byte field takes an extreme value
            off = 8 * int(rng.integers(0, max(1, len(b) // 8)))
            b[off:off + 8] = int(extremes[int(rng.integers(0, len(extremes)))]).to_bytes(8, "little")
        elif how == 2:  # truncation
            b = b[: int(rng.integers(0, len(b)))]
        else:           # an unaligned 4-byte field + a bit flip
            off = int(rng.integers(0, max(1, len(b) - 4)))
            b[off:off + 4] = int(rng.integers(0, 1 << 32)).to_bytes(4, "little")
            b[int(rng.integers(0, len(b)))] ^= 0x80
        f = tmp_path / f"m{it}"
        f.write_bytes(bytes(b))
        r = run(harness, kind, f, HOSTILE / constants if constants else None)
        assert r.returncode == 0 and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (seed_file, it, how, r.stdout[-200:], r.stderr[-1500:])
        outcomes[r.stdout.split()[0] if r.stdout.split() else "rejected"] = outcomes.get(r.stdout.split()[0] if r.stdout.split() else "rejected", 0) + 1
    assert outcomes.get("rejected:", 0) + outcomes.get("rejected", 0) > 0  # (the mutations are not all no-ops)
