"""Batched-leaf MCTS (twixt_for_open_spiel_amd/mcts.py): the search driven
by the device engine must be the SAME search as the one driven by the CPU oracle -- same
visit counts, same moves -- because leaf states, legal lists and play-out results are
bit-exact."""
import numpy as np
import pytest


class OracleEvaluator:
    """Test double with DeviceEvaluator's interface, computed on the CPU oracle."""

    def __init__(self, board_size, max_leaves, first_board_id=0):
        self.n, self.L, self.first = board_size, max_leaves, first_board_id
        self.history = []

    def _board(self, extra=()):
        import oracle_binding as ob
        b = ob.Board(self.n)
        for a in list(self.history) + list(extra):
            b.apply(a)
        return b

    def reset_root(self):
        self.history = []

    def advance_root(self, action):
        self.history.append(action)
        return self._board().result

    def root_info(self):
        b = self._board()
        return b.result, b.current_player(), b.legal_actions()

    def evaluate(self, paths, seed, base=0):
        out = []
        for i, p in enumerate(paths):
            b = self._board(p)
            res, cp, legal = b.result, b.current_player(), b.legal_actions()
            # the leaf slot (base + i) is a copy of the root slot: episode 0, policy keyed by the slot's global id
            b.rollout(seed, self.first + base + i, self.n * self.n - 3, False)
            out.append((res, cp, legal if res == 0 else [], b.result))
        return out


def test_tree_logic_on_oracle():
    """CPU: the driver's bookkeeping on the oracle evaluator (position of twixt_test.cc:163-177,
    red to move with a winning reply 48 available)."""
    from twixt_for_open_spiel_amd.mcts import BatchedMCTS
    ev = OracleEvaluator(8, 64)
    for a in (21, 38, 15, 11, 27, 17, 42, 45):
        ev.advance_root(a)
    res, player, legal = ev.root_info()
    assert res == 0 and player == 0 and 48 in legal
    best, visits = BatchedMCTS(ev, uct_c=0.5, leaves_per_batch=64, seed=5).search(player, legal, 1500)
    assert sum(visits.values()) == 1500 and set(visits) <= set(legal)
    assert visits[best] == max(visits.values())
    assert visits[48] > 1500 // len(legal)          # the immediate win is visited more than uniformly
    again = BatchedMCTS(ev, uct_c=0.5, leaves_per_batch=64, seed=5).search(player, legal, 1500)
    assert again == (best, visits)                  # deterministic for a given seed


@pytest.mark.gpu
def test_device_search_equals_oracle_search():
    from twixt_for_open_spiel_amd.mcts import BatchedMCTS, DeviceEvaluator, self_play
    n, L = 8, 128
    dev, orc = DeviceEvaluator(n, L, first_board_id=11), OracleEvaluator(n, L, first_board_id=11)
    hd, rd, _ = self_play(n, max_simulations=640, leaves_per_batch=L, max_moves=6, seed=3, evaluator=dev)
    ho, ro, _ = self_play(n, max_simulations=640, leaves_per_batch=L, max_moves=6, seed=3, evaluator=orc)
    assert hd == ho and rd == ro
    # and the visit counts of one more search from that position
    _, pl, legal = dev.root_info()
    assert (pl, legal) == orc.root_info()[1:]
    bd, vd = BatchedMCTS(dev, leaves_per_batch=L, seed=9).search(pl, legal, 1024)
    bo, vo = BatchedMCTS(orc, leaves_per_batch=L, seed=9).search(pl, legal, 1024)
    assert vd == vo and bd == bo


def _build_cpp_driver(out):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "twixt_for_open_spiel_amd")
    from twixt_for_open_spiel_amd import build
    build.build()
    subprocess.check_call(["g++", "-O2", "-fopenmp", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(root, "include"),
                           "-I/opt/rocm/include", os.path.join(pkg, "spiel_adapter", "mcts_driver.cc"), "-o", out,
                           "-L" + pkg, "-ltwixt_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + pkg,
                           "-Wl,-rpath,/opt/rocm/lib"])
    return out


def _build_cpu_search(out):
    """The same driver source with the CPU oracle behind the evaluator interface
    (tests/mcts_cpu_evaluator.h): 'the CPU engine doing the same search' of SURVEY 8(f3)."""
    import os
    import subprocess
    import oracle_binding as ob
    ob.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["g++", "-O2", "-fopenmp", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
                           "-DMCTS_EVALUATOR_HEADER=\"mcts_cpu_evaluator.h\"", "-I" + os.path.join(root, "include"),
                           "-I/opt/rocm/include", "-I" + os.path.join(root, "tests"), "-I" + os.path.join(root, "oracle"),
                           os.path.join(root, "twixt_for_open_spiel_amd", "spiel_adapter", "mcts_driver.cc"), "-o", out,
                           "-L" + os.path.join(root, "oracle"), "-ltwixt_oracle",
                           "-Wl,-rpath," + os.path.join(root, "oracle")])
    return out


MOVE_PAT = r"move (\d+): player (\d) plays (\d+) \((\d+) visits\)"


def _run_driver(exe, args, timeout=900):
    import re
    import subprocess
    out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, timeout=timeout)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    text = out.stdout.decode()
    rate = float(re.search(r"= (\d+) simulations/s", text).group(1))
    return re.findall(MOVE_PAT, text), rate, text


def test_cpp_driver_compiles(tmp_path):
    _build_cpp_driver(str(tmp_path / "mcts_driver"))


def test_cpu_search_equals_python_oracle_search(tmp_path, capsys):
    """CPU: mcts_driver.cc over the oracle evaluator header plays the game mcts.py plays over
    the Python OracleEvaluator (same moves, same visit counts) -- this pins the header the
    full-size GPU test and tools/mcts_bench.py compare the device search with."""
    import re
    from twixt_for_open_spiel_amd.mcts import self_play
    exe = _build_cpu_search(str(tmp_path / "mcts_cpu"))
    cpp_lines, _, _ = _run_driver(exe, [8, 640, 6, 128, 3])
    capsys.readouterr()
    hist, _, _ = self_play(8, max_simulations=640, leaves_per_batch=128, max_moves=6, seed=3,
                           evaluator=OracleEvaluator(8, 128), verbose=True)
    py_lines = re.findall(MOVE_PAT, capsys.readouterr().out)
    assert cpp_lines == py_lines and [int(m[2]) for m in cpp_lines] == hist and len(hist) == 6


def test_lanes_never_reach_the_root_slots(tmp_path, capsys):
    """CPU: the two lanes split the leaf slots only while each half holds a leaf of every game.  With
    leaves == games, or a single leaf slot, a split second lane would start at or run past slot L --
    the games' roots (the driver exits on it).  One leaf per round trip must still be the search
    mcts.py does with leaves_per_batch=1, move for move and visit for visit."""
    import re
    from twixt_for_open_spiel_amd.mcts import self_play
    exe = _build_cpu_search(str(tmp_path / "mcts_cpu"))
    cpp_lines, _, _ = _run_driver(exe, [5, 48, 3, 1, 2])          # leaves = 1, one game
    capsys.readouterr()
    hist, _, _ = self_play(5, max_simulations=48, leaves_per_batch=1, max_moves=3, seed=2,
                           evaluator=OracleEvaluator(5, 1), verbose=True)
    py_lines = re.findall(MOVE_PAT, capsys.readouterr().out)
    assert cpp_lines == py_lines and len(hist) == 3
    for leaves, games in ((2, 2), (3, 2), (5, 3), (4, 2)):         # below, at and above leaves = 2 * games
        _, _, text = _run_driver(exe, [5, 24, 2, leaves, 7, games])
        assert len(re.findall(r"game \d move \d+: player \d plays \d+", text)) == 2 * games
    # the two-lane order with a split (leaves = 2): still the Python search
    cpp_lines, _, _ = _run_driver(exe, [5, 48, 3, 2, 2])
    capsys.readouterr()
    self_play(5, max_simulations=48, leaves_per_batch=2, max_moves=3, seed=2, evaluator=OracleEvaluator(5, 2), verbose=True)
    assert cpp_lines == re.findall(MOVE_PAT, capsys.readouterr().out)


@pytest.mark.gpu
def test_lanes_never_reach_the_root_slots_on_the_device(tmp_path):
    """The same corner on the engine: leaves == games and a single leaf slot (a lane that reached into
    the root slots would overwrite live roots: 'illegal root move' or a silently different search)."""
    dev_exe = _build_cpp_driver(str(tmp_path / "mcts_driver"))
    cpu_exe = _build_cpu_search(str(tmp_path / "mcts_cpu"))
    for args in ([5, 48, 3, 1, 2], [5, 24, 2, 2, 7, 2], [5, 24, 2, 3, 7, 2]):
        _, _, dev_text = _run_driver(dev_exe, args)
        _, _, cpu_text = _run_driver(cpu_exe, args)
        pick = lambda t: [ln for ln in t.splitlines() if " plays " in ln or "result " in ln and "time:" not in ln]
        assert pick(dev_text) == pick(cpu_text) and pick(dev_text)


@pytest.mark.gpu
def test_config5_full_size(tmp_path, capsys):
    """BASELINE config 5 at its stated size: board_size=12, max_simulations=20000 per move, 2,048
    leaves per device round trip, two moves of self-play (README.md:36-42's mcts_example run).
    The device-driven C++ search, the same C++ search on the CPU oracle and mcts.py over the
    Python OracleEvaluator choose the same moves with the same visit counts; both C++ rates are
    printed (pytest -s) and recorded by tools/mcts_bench.py."""
    import re
    from twixt_for_open_spiel_amd.mcts import self_play
    dev_exe = _build_cpp_driver(str(tmp_path / "mcts_driver"))
    cpu_exe = _build_cpu_search(str(tmp_path / "mcts_cpu"))
    args = [12, 20000, 2, 2048, 1]
    dev_lines, dev_rate, _ = _run_driver(dev_exe, args)
    cpu_lines, cpu_rate, _ = _run_driver(cpu_exe, args)
    assert len(dev_lines) == 2 and dev_lines == cpu_lines
    capsys.readouterr()
    hist, _, total = self_play(12, max_simulations=20000, leaves_per_batch=2048, max_moves=2, seed=1,
                               evaluator=OracleEvaluator(12, 2048), verbose=True)
    py_lines = re.findall(MOVE_PAT, capsys.readouterr().out)
    assert total == 40000 and py_lines == dev_lines
    print("config 5 (N=12, 20000 simulations/move, 2048 leaves/batch): device search %.0f simulations/s, "
          "CPU search (oracle, all host cores) %.0f simulations/s" % (dev_rate, cpu_rate))


@pytest.mark.gpu
def test_cpp_driver_plays_the_same_game(tmp_path, capsys):
    """spiel_adapter/mcts_driver.cc (C++ host over the C ABI; children's statistics in
    per-node arrays, AVX2 UCT values) and mcts.py make the same moves with the same visit
    counts."""
    import re
    import subprocess
    from twixt_for_open_spiel_amd.mcts import self_play
    exe = _build_cpp_driver(str(tmp_path / "mcts_driver"))
    out = subprocess.run([exe, "8", "640", "6", "128", "3"], capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    pat = r"move (\d+): player (\d) plays (\d+) \((\d+) visits\)"
    cpp_lines = re.findall(pat, out.stdout.decode())
    capsys.readouterr()
    hist, _, _ = self_play(8, max_simulations=640, leaves_per_batch=128, max_moves=6, seed=3, verbose=True)
    py_lines = re.findall(pat, capsys.readouterr().out)
    assert [int(m[2]) for m in cpp_lines] == hist and len(hist) == 6
    assert cpp_lines == py_lines


@pytest.mark.gpu
def test_cpp_driver_many_games(tmp_path):
    """games > 1: independent trees share the device round trips; every game makes its six
    moves (the driver aborts on a move the engine rejects) and game results are consistent."""
    import re
    import subprocess
    exe = _build_cpp_driver(str(tmp_path / "mcts_driver"))
    out = subprocess.run([exe, "8", "512", "6", "512", "5", "4"], capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    text = out.stdout.decode()
    moves = re.findall(r"game (\d) move (\d+): player (\d) plays (\d+) \((\d+) visits\)", text)
    assert len(moves) == 4 * 6
    for g in range(4):
        mine = [m for m in moves if int(m[0]) == g]
        assert [int(m[1]) for m in mine] == list(range(6))
        assert [int(m[2]) for m in mine] == [0, 1, 0, 1, 0, 1]       # nobody wins an 8x8 game in 6 moves
        assert all(0 <= int(m[3]) < 64 and 1 <= int(m[4]) <= 512 for m in mine)
    assert "%d simulations" % (4 * 6 * 512) in text
