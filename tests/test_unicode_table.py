"""The generated Unicode table of the device-side EN normalisation (map_neo_amd/csrc/mdd_unicode_tab.h) against THIS
interpreter: the committed header is what tools/gen_unicode_tables.py produces here, and its rules hold -- a string of
unflagged code points is NFC-normalised and lower() acts on it character by character."""
import os
import random
import subprocess
import sys
import unicodedata as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_committed_table_matches_this_interpreter():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_unicode_tables.py"), "--check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr


def test_unflagged_code_points_are_nfc_inert_and_lower_per_character():
    import gen_unicode_tables as G
    maybe = G.maybe_set()
    ok = [c for c in range(0x110000) if not G.flagged(c, maybe)]
    assert len(ok) > 250_000
    for c in (0xe9, 0x2019, 0x201c, 0x2014, 0xa0, 0x3000, 0x4e2d, 0x1f600, 0xdf, 0x1e9e):
        assert not G.flagged(c, maybe), hex(c)
    for c in (0x301, 0x130, 0x3a3, 0x212b, 0x2000, 0x1161, 0x11a8, 0x378, 0xd800, 0x9be, 0x2f800):
        assert G.flagged(c, maybe), hex(c)
    rnd = random.Random(7)
    cased = [c for c in ok if chr(c).lower() != chr(c)]
    for _ in range(30000):
        pool = cased if rnd.random() < 0.3 else ok
        s = "".join(chr(rnd.choice(pool)) for _ in range(rnd.randint(1, 10)))
        assert U.normalize("NFC", s) == s
        assert s.lower() == "".join(ch.lower() for ch in s)
        assert all(len(ch.lower()) == 1 for ch in s)
    # whitespace as strip() / re's \s see it = str.isspace, per character
    import re
    for c in ok:
        assert bool(re.fullmatch(r"\s", chr(c))) == chr(c).isspace()
