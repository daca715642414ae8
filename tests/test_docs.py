"""The documents cite their evidence by file name: every `rNN_*` name DESIGN.md, INTEGRATION.md, README.md and
profiles/README.md mention must exist under profiles/ (or profiles/archive/, where the intermediate states live), and
DESIGN.md stays within the size it was cut to."""
import fnmatch
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _profile_names():
    names = set(os.listdir(os.path.join(ROOT, "profiles")))
    names |= set(os.listdir(os.path.join(ROOT, "profiles", "archive")))
    return names


def test_cited_profile_files_exist():
    have = _profile_names()
    missing = []
    for doc in ("DESIGN.md", "INTEGRATION.md", "README.md", os.path.join("profiles", "README.md")):
        text = open(os.path.join(ROOT, doc), encoding="utf-8").read()
        for m in re.finditer(r"`(?:profiles/)?((?:r0[1-9]|traffic)_[A-Za-z0-9_*.…]+)`", text):
            name = m.group(1).rstrip(".")
            if "…" in name:
                continue
            pattern = name if ("*" in name or "." in name) else name + "*"
            if not any(fnmatch.fnmatch(f, pattern) for f in have):
                missing.append((doc, name))
    assert not missing, missing


def test_design_md_is_the_short_current_state_document():
    size = os.path.getsize(os.path.join(ROOT, "DESIGN.md"))
    assert size <= 35 * 1024, f"DESIGN.md is {size} bytes: move history to DESIGN_HISTORY.md"
