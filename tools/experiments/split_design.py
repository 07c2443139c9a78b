#!/usr/bin/env python3
"""One-off (round 5): move DESIGN.md's deep dives to docs/design/*.md and its round-by-round tables / measured negatives to docs/rounds/r0N.md,
verbatim, re-wrapped at 120 columns (tables and code untouched).  DESIGN.md itself is then rewritten by hand as the current design."""
import re
import textwrap
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
src = (ROOT / "DESIGN.md").read_text().splitlines()


def wrap_line(ln):
    if not ln.strip() or ln.lstrip().startswith("|") or ln.startswith("#") or ln.startswith("```") or len(ln) <= 120:
        return [ln]
    m = re.match(r"^(\s*)(\* |\d+\. |- )?", ln)
    ind, bullet = m.group(1), m.group(2) or ""
    body = ln[len(ind) + len(bullet):]
    return textwrap.wrap(body, width=120, initial_indent=ind + bullet, subsequent_indent=ind + " " * len(bullet), break_long_words=False, break_on_hyphens=False)


def rng(a, b):  # 1-based inclusive line range of the ORIGINAL file
    out = []
    for ln in src[a - 1:b]:
        out.extend(wrap_line(ln))
    return out


def emit(path, title, intro, parts):
    lines = [f"# {title}", ""] + textwrap.wrap(intro, 120) + [""]
    for head, (a, b) in parts:
        if head:
            lines += [f"## {head}", ""]
        lines += rng(a, b) + [""]
    (ROOT / path).parent.mkdir(parents=True, exist_ok=True)
    (ROOT / path).write_text("\n".join(lines).rstrip() + "\n")
    print(path, len(lines), "lines")


CITE = "All `path:line` citations are relative to `/root/reference/`; \"[SEAL-upstream]\" marks behaviour of Microsoft SEAL 4.0.0, which the reference links but does not vendor."
emit("docs/design/boundary_and_runtime.md", "Path, boundary, options, randomness (detail)",
     "Moved verbatim from DESIGN.md sections 1-2 as of round 4 (re-wrapped). The current summary is DESIGN.md. " + CITE,
     [("The path and its boundary", (8, 69)), ("Data layout in HBM", (72, 86))])
emit("docs/design/arithmetic.md", "Arithmetic on gfx950 (detail)",
     "Moved verbatim from DESIGN.md section 3 as of round 4 (re-wrapped). " + CITE,
     [("", (89, 142))])
emit("docs/design/kernels.md", "Kernels and launch sequences (detail)",
     "Moved verbatim from DESIGN.md section 4 as of round 4 (re-wrapped): the kernel table, NTT tiling, the single-crossing NTT, the fused key "
     "switch / rescale, operand expressions, the batched plan, chain fusion, opcode 10, real bootstrapping, grouped-digit key switching and "
     "its fused sequence, launch order and tile placement, bootstrapping precision. Round-5 changes are in docs/rounds/r05.md and DESIGN.md. " + CITE,
     [("", (145, 472))])
emit("docs/design/oracle.md", "Oracle (detail)",
     "Moved verbatim from DESIGN.md section 5 as of round 4 (re-wrapped). " + CITE,
     [("", (511, 593))])
emit("docs/design/measurement.md", "Measurement: workload, end-to-end check, lowerings (detail)",
     "Moved verbatim from DESIGN.md section 6 as of round 4 (re-wrapped): what the bench line measures and how; the per-round result tables "
     "are in docs/rounds/. " + CITE,
     [("", (596, 624)), ("CPU baseline and sanity anchor", (696, 704)), ("The timed step on the byte roofline; real bootstrapping; throughput mode; config 3; set-up", (728, 736)),
      ("", (750, 770))])
emit("docs/design/multi_gpu.md", "Multi-GPU: replicas (detail)",
     "Moved verbatim from DESIGN.md section 7 as of round 4 (re-wrapped). " + CITE,
     [("", (773, 806))])
emit("docs/design/scope_and_extensions.md", "Out of scope; on-line encode; VM lifetime; config 4 (detail)",
     "Moved verbatim from DESIGN.md section 8 as of round 4 (re-wrapped). " + CITE,
     [("", (809, 860))])
emit("docs/rounds/r02.md", "Round 2: results, measured negatives, review items",
     "Moved verbatim from DESIGN.md as of round 4 (re-wrapped). Round 1's history is profiles/README.history.md. " + CITE,
     [("Round 2 against the round-1 review, item by item", (892, 904)), ("Round-2 result table and how it got there", (678, 695)),
      ("The roofline leg in round 2 (two-launch tiles): what bounds it, what was tried", (705, 727)),
      ("Where the idle time sits; chain latency", (737, 749)),
      ("Why launch chains and not one persistent kernel; scheduling dead ends", (473, 487))])
emit("docs/rounds/r03.md", "Round 3: results, review items, what came next",
     "Moved verbatim from DESIGN.md as of round 4 (re-wrapped). " + CITE,
     [("Round 3 against the round-2 review, item by item", (907, 920)), ("Round-3 results", (644, 677)), ("Round 3's list of next steps", (875, 889))])
emit("docs/rounds/r04.md", "Round 4: results, measured negatives, review items, what came next",
     "Moved verbatim from DESIGN.md as of round 4 (re-wrapped). Experiment log: profiles/r04_experiments.txt. " + CITE,
     [("Round 4 in one paragraph", (22, 30)), ("Round 4 against the round-3 review, item by item", (923, 934)), ("Round-4 results", (625, 643)),
      ("The plan's graph built explicitly: a measured negative", (488, 508)), ("After round 4, what comes next", (861, 874))])
