"""The documents cite files (profiles, tests, scripts, headers) as evidence: every cited path exists."""
import glob
import os
import re

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
DOCS = ("DESIGN.md", "README.md", "INTEGRATION.md", "scripts/README.md", "profiles/r05_abort_hunt.md")


def test_cited_paths_exist():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for m in re.finditer(r"`((?:profiles|tests|scripts|oracle|include|quickrank_amd)/[A-Za-z0-9_./*\-]+)`", text):
            p = m.group(1).rstrip(".").split("::")[0]
            if "NN" in p or ".." in p:          # a naming scheme (rNN_...), a range (g1..g4)
                continue
            if p.startswith(("oracle/_ref", "quickrank_amd/lib", "quickrank_amd/bin")):
                continue                        # built, not tracked
            found = glob.glob(os.path.join(ROOT, p)) if "*" in p else os.path.exists(os.path.join(ROOT, p))
            if not found:
                missing.append((doc, p))
    assert not missing, missing
