"""Independent pin of the CPU oracle above N=8 (SURVEY Appendix A.10 re-expressed, CPU).

tests/oracle_invariants.cc plays random games on the oracle at N in {5,6,7,8,12,24} (a third
with a forced swap) and after EVERY move recomputes flags, result, legal lists and the link
scan from first principles -- BFS over link bits, region & ~occupied, integer segment
geometry (twixt_geom.h::crosses, which is derived without the reference's 72-entry table).
A shared misreading of a rule by the oracle and the HIP engine would have to survive this."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    import oracle_binding as ob
    ob.build()
    exe = str(tmp_path / "oracle_invariants")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", exe,
                           os.path.join(ROOT, "tests", "oracle_invariants.cc"),
                           "-L", os.path.join(ROOT, "oracle"), "-ltwixt_oracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-fopenmp"])
    return exe


def test_oracle_invariants_on_every_move(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"OK games=(\d+) moves=(\d+) peg_flag_checks=(\d+) result_checks=(\d+) legal_checks=(\d+) "
                  r"link_scan_checks=(\d+) final_link_checks=(\d+) swapped_games=(\d+)", r.stdout)
    assert m, r.stdout
    games, moves, flags, results, legal, scans, final, swapped = map(int, m.groups())
    assert games >= 4000 and results == moves == legal and flags > 1e7 and scans > 1e5 and swapped > 500
    for n in (5, 6, 7, 8, 12, 24):                      # every size played, wins and draws seen
        row = re.search(r"N=\s*%d:\s+(\d+) games,\s+(\d+) moves, results red/blue/draw = (\d+)/(\d+)/(\d+)" % n,
                        r.stdout)
        assert row and all(int(v) > 0 for v in row.groups()), r.stdout


def test_the_checker_notices_corruptions(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "--selftest"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert re.search(r"selftest: \d{4,} corruptions tried, 0 unnoticed", r.stdout), r.stdout
