#!/bin/bash
# ONE command that turns `parity` from "unpinned" to pinned on the first machine that has Microsoft SEAL 4.x (the library the reference links,
# README.md:65-73; it is neither vendored in the reference nor installed in the build image or on the GPU box):
#
#     SEAL_ROOT=/path/to/seal/install/prefix bash tools/fixtures/make_seal_vectors.sh [logN=12] [primes=4]
#
# builds tools/fixtures/seal_diff_gen.cpp against that SEAL, runs it, and writes tests/golden/seal_vectors/ -- the key directory exactly as
# SEAL_HEVM::create_context writes it (SEAL_HEVM.cpp:44-89), two encrypted inputs, and SEAL's own result ciphertexts for the evaluator calls of
# the opcode handlers (SEAL_HEVM.cpp:268-323: rotate 1 / 37 / -100, negate, add, modswitch, multiply + relinearize, rescale, multiply_plain,
# add_plain), all in SEAL's serialization, plus MANIFEST.json (SEAL version, parameters, sha256 of every file).  Commit the directory
# (~19 MB at the defaults: 24 Galois keys of [3][2][4][4096] words do not compress): from then on tests/test_seal_diff.py needs no SEAL --
# `pytest -m "not gpu"` holds the oracle to SEAL's limbs and `pytest -m gpu` holds the MI355X runtime to them (keys through initFullVM,
# ciphertexts through hevm_load_ctxt), bit for bit, in every later round.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
LOGN=${1:-12}; PRIMES=${2:-4}
: "${SEAL_ROOT:?set SEAL_ROOT to the install prefix of Microsoft SEAL 4.x (the directory holding include/SEAL-4.*/seal/seal.h)}"
INC=$(ls -d "$SEAL_ROOT"/include/SEAL-4.* 2>/dev/null | head -1)
[ -f "$INC/seal/seal.h" ] || { echo "no include/SEAL-4.*/seal/seal.h under $SEAL_ROOT" >&2; exit 2; }
LIB=$(ls "$SEAL_ROOT"/lib*/libseal-4.*.a "$SEAL_ROOT"/lib*/libseal*.so 2>/dev/null | head -1)
[ -n "$LIB" ] || { echo "no libseal under $SEAL_ROOT/lib*" >&2; exit 2; }
TMP=$(mktemp -d); trap 'rm -rf "$TMP"' EXIT
for EXTRA in "" "-lzstd -lz" "-lz"; do
  if g++ -std=c++17 -O2 "$ROOT/tools/fixtures/seal_diff_gen.cpp" -I"$INC" "$LIB" $EXTRA -lpthread -o "$TMP/seal_diff_gen" 2>"$TMP/link.err"; then break; fi
done
[ -x "$TMP/seal_diff_gen" ] || { cat "$TMP/link.err" >&2; echo "the generator does not link against $LIB" >&2; exit 2; }
OUT=$ROOT/tests/golden/seal_vectors
rm -rf "$OUT"; mkdir -p "$OUT"
LD_LIBRARY_PATH="$(dirname "$LIB"):${LD_LIBRARY_PATH:-}" "$TMP/seal_diff_gen" "$OUT" "$LOGN" "$PRIMES" | tee "$TMP/gen.log"
python3 - "$OUT" "$LOGN" "$PRIMES" "$(tail -1 "$TMP/gen.log")" <<'PY'
import hashlib, json, sys
from pathlib import Path
out, logn, primes, banner = Path(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
files = {p.name: {"bytes": p.stat().st_size, "sha256": hashlib.sha256(p.read_bytes()).hexdigest()} for p in sorted(out.iterdir()) if p.is_file()}
(out / "MANIFEST.json").write_text(json.dumps({"generator": "tools/fixtures/seal_diff_gen.cpp (tools/fixtures/make_seal_vectors.sh)", "seal": banner,
                                               "logN": logn, "primes": primes, "files": files}, indent=1))
print(f"{len(files)} files, {sum(f['bytes'] for f in files.values()) >> 20} MiB -> {out}")
PY
cd "$ROOT" && python3 -m pytest tests/test_seal_diff.py -q -m "not gpu"
echo "now: git add tests/golden/seal_vectors && git commit   (then, on a GPU box: python -m pytest tests/test_seal_diff.py -m gpu)"
