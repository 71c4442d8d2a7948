"""DESIGN.md's headline figures against the committed profiles they cite (tools/check_design_numbers.py: the table of
profiles/FIGURES.md, every quoted value of which must occur in DESIGN.md's text): a figure edited in the text without its file, or a
refreshed file without the text, fails here."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_quotes_what_the_profiles_say():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_design_numbers.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "0 not reproduced" in r.stdout
