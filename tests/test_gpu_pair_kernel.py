"""GPU parity of the step kernels of the big boards (split layout: N >= 17; pair kernel: N >= 19) against the CPU oracle: the default path --
split layout, step_window1_kernel (windowed journal launches, one lane per board) + step_window_full_kernel
(compactions, flushes) -- and the two older ones on the chunk layout, step_pair_kernel (TWX_WINDOW=0: two lanes
per board, whole boards streamed) and step_kernel64 (TWX_PAIR=0: one lane per board) -- every board, every
step: actions, status bytes,
twx_status in the middle of a journal period (the scalars chunk is then stale and the journal
records carry counter / result / swapped), the full state at the end.  Also every journal
period and batch order on the pair kernel, external actions with swaps and illegal moves, and
launches that alternate between the two kernels on one handle."""
import numpy as np
import pytest

from test_gpu_parity import compare_planes, make_env, u32  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a HIP device")
    return torch


KERNEL_PATHS = {"window": {}, "pair": {"TWX_WINDOW": "0"}, "lane": {"TWX_PAIR": "0"}}


# (N = 17, 18: split layout and windowed step as well -- TWX_SPLIT_CHT; their other two paths are the small-board
# kernel on the chunk layout)
@pytest.mark.parametrize("n,batch,steps", [(17, 900, 330), (18, 640, 380), (19, 1000, 420), (21, 700, 150), (22, 330, 200),
                                           (24, 1500, 640)])
@pytest.mark.parametrize("path", ["window", "pair", "lane"])
def test_every_step_against_the_oracle(torch_mod, n, batch, steps, path, monkeypatch):
    import oracle_binding as ob
    torch = torch_mod
    for k, v in KERNEL_PATHS[path].items():
        monkeypatch.setenv(k, v)
    seed, first = 77, 4000
    env = make_env(n, batch, first=first)
    acts = np.zeros((batch, steps), dtype=np.int32)
    stat = np.zeros((batch, steps), dtype=np.uint8)
    probe = {}  # twx_status at odd places of the journal period (does not materialise the journal)
    for t in range(steps):
        a, s = env.step_random(seed, True)
        acts[:, t] = a.cpu().numpy()
        stat[:, t] = s.cpu().numpy()
        if t % 37 in (0, 3, 5) or t == steps - 1:
            res, mc, cp = env.status()
            probe[t] = (res.cpu().numpy().copy(), mc.cpu().numpy().copy(), cp.cpu().numpy().copy())
    ref = ob.rollout_batch(n, seed, first, batch, steps, True, want_planes=True, want_cells=True, want_trace=True)
    bad = np.argwhere(ref["actions"] != acts)
    assert bad.size == 0, "first action mismatch at (board, step) %s" % (bad[0],)
    bad = np.argwhere(ref["status"] != stat)
    assert bad.size == 0, "first status mismatch at (board, step) %s: ref %x got %x" % (
        bad[0], ref["status"][tuple(bad[0])], stat[tuple(bad[0])])
    compare_planes(env, ref["planes"], what="final")
    assert np.array_equal(env.export_cells().cpu().numpy(), ref["cells"])
    assert (stat & 0x10).any() and (stat & 3).any()      # games ended and were reset on the way
    # twx_status against the status bytes of the same step: result, and the move counter from
    # the number of moves since the last reset
    for t, (res, mc, cp) in probe.items():
        want_res = stat[:, t] & 3
        assert np.array_equal(res, want_res), "twx_status result at step %d" % t
        moves = np.zeros(batch, dtype=np.int64)
        for b in range(batch):
            resets = np.nonzero(stat[b, :t + 1] & 0x10)[0]
            start = resets[-1] if resets.size else 0
            moves[b] = t + 1 - start
        assert np.array_equal(mc, moves), "twx_status move counter at step %d" % t
        assert np.array_equal(cp, np.where(want_res != 0, -4, moves & 1)), "current player at step %d" % t
    env.close()


def test_small_batches_rewrite_in_place(torch_mod, monkeypatch):
    """The default routing of a small batch of a big board (a search's leaves): every step rewrites its boards
    through step_window_full_kernel, the journal stays empty; same results, status queries in between."""
    import oracle_binding as ob
    monkeypatch.setenv("TWX_SMALL_BATCH_REWRITE", "1")
    n, batch, seed, first, steps = 24, 3000, 5, 99, 130
    env = make_env(n, batch, first=first)
    stat = None
    for t in range(steps):
        _, stat = env.step_random(seed, True, want_actions=False)
        if t % 17 == 3:
            res, mc, _ = env.status()
            assert np.array_equal(res.cpu().numpy(), stat.cpu().numpy() & 3)
    ref = ob.rollout_batch(n, seed, first, batch, steps, True, want_trace=True)
    assert np.array_equal(stat.cpu().numpy(), ref["status"][:, -1])
    compare_planes(env, ref["planes"], what="small batch, rewritten in place")
    env.close()


@pytest.mark.parametrize("period,pingpong", [(1, 0), (2, 1), (3, 0), (4, 1), (5, 1), (6, 0), (7, 1), (8, 0), (8, 1)])
def test_pair_kernel_journal_periods(torch_mod, period, pingpong, monkeypatch):
    """The log-structured state must be invisible on the pair kernel too: any compaction period
    and batch order, queries (which materialise the journal) interleaved at odd places."""
    import oracle_binding as ob
    monkeypatch.setenv("TWX_JOURNAL_PERIOD", str(period))
    monkeypatch.setenv("TWX_PINGPONG", str(pingpong))
    n, batch, seed = 20, 333, 21
    env = make_env(n, batch, first=9)
    done = 0
    for upto in (1, 2, 5, 9, 10, 17, 30, 31, 64, 100, 163, 260, 395):
        for _ in range(upto - done):
            env.step_random(seed, True, want_actions=False, want_status=False)
        done = upto
        ref = ob.rollout_batch(n, seed, 9, batch, upto, True, want_cells=True)
        compare_planes(env, ref["planes"], what="period %d after %d steps" % (period, upto))
        if upto in (9, 31, 395):
            assert np.array_equal(env.export_cells().cpu().numpy(), ref["cells"])
    env.close()


@pytest.mark.parametrize("period", [3, 8])
def test_rolling_compaction_on_a_big_board(torch_mod, period, monkeypatch):
    """TWX_ROLLING=1 (a tested option of the one-lane kernel) keeps working on the 32-board tiles
    of N >= 19: whole-batch steps then run step_kernel64, not the pair kernel."""
    import oracle_binding as ob
    monkeypatch.setenv("TWX_ROLLING", "1")
    monkeypatch.setenv("TWX_JOURNAL_PERIOD", str(period))
    n, batch, seed = 22, 5000, 8
    env = make_env(n, batch, first=3)
    done = 0
    for upto in (1, 4, 9, 30, 77, 200, 470):
        for _ in range(upto - done):
            env.step_random(seed, True, want_actions=False, want_status=False)
        done = upto
        res, mc, _ = env.status()                      # mid-period: scalars from the journal
        ref = ob.rollout_batch(n, seed, 3, batch, upto, True)
        sw = ref["planes"].shape[1]
        assert np.array_equal(res.cpu().numpy(), ref["planes"][:, sw - 3] & 0xff)
        assert np.array_equal(mc.cpu().numpy(), ref["planes"][:, sw - 4] & 0xffff)
        compare_planes(env, ref["planes"], what="rolling period %d after %d steps" % (period, upto))
    env.close()


@pytest.mark.parametrize("dynamic,nstatic", [(0, 0), (1, 0), (1, 1), (1, 2), (1, 1000)])
@pytest.mark.parametrize("path", ["window", "pair"])
def test_pair_kernel_batch_order(torch_mod, dynamic, nstatic, path, monkeypatch):
    """Ticket-drawn rounds on the window kernels (64- and 32-board batches) and on the pair kernel: one wave
    per CU over ~50k boards gives static rounds, drawn rounds and a ragged last round; every board is stepped
    exactly once."""
    import oracle_binding as ob
    for k, v in KERNEL_PATHS[path].items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("TWX_PAIR_WAVES_PER_CU", "1")
    monkeypatch.setenv("TWX_WIN_WAVES_PER_CU", "1")
    monkeypatch.setenv("TWX_WIN_FULL_WAVES_PER_CU", "1")
    monkeypatch.setenv("TWX_DYNAMIC", str(dynamic))
    monkeypatch.setenv("TWX_DYNAMIC_STATIC", str(nstatic))
    n, batch, seed, first, steps = 19, 50021, 31, 12345, 40
    env = make_env(n, batch, first=first)
    stat = None
    for _ in range(steps):
        _, stat = env.step_random(seed, True, want_actions=False)
    ref = ob.rollout_batch(n, seed, first, batch, steps, True, want_trace=True)
    assert np.array_equal(stat.cpu().numpy(), ref["status"][:, -1])
    compare_planes(env, ref["planes"], what="dynamic=%d nstatic=%d" % (dynamic, nstatic))
    env.close()


@pytest.mark.parametrize("batch", [1, 31, 33, 64, 65, 97])
@pytest.mark.parametrize("pair", [1, 0])
def test_ragged_batches(torch_mod, batch, pair, monkeypatch):
    """Batches that do not fill a wave or a tile (the allocation is padded to 64 boards; padding
    boards never move and are never reported): a few boards, whole-batch steps, N=24 and N=19."""
    import oracle_binding as ob
    monkeypatch.setenv("TWX_PAIR", str(pair))
    for n, steps in ((24, 60), (19, 40), (18, 40)):
        env = make_env(n, batch, first=7)
        acts = []
        for _ in range(steps):
            a, st = env.step_random(3, True)
            acts.append(a.cpu().numpy())
        ref = ob.rollout_batch(n, 3, 7, batch, steps, True, want_trace=True)
        assert np.array_equal(np.stack(acts, axis=1), ref["actions"])
        assert np.array_equal(st.cpu().numpy(), ref["status"][:, -1])
        compare_planes(env, ref["planes"], what="batch %d N=%d" % (batch, n))
        res, mc, cp = env.status()
        assert mc.shape[0] == batch and int(mc.min()) == steps
        env.close()


def test_kernels_alternate_on_one_state(torch_mod, monkeypatch):
    """Same state layout and journal: two handles, one per kernel, exchange their boards through
    export / import in the middle of a journal period and keep playing the same games."""
    import oracle_binding as ob
    n, batch, seed, first = 24, 777, 5, 31
    monkeypatch.setenv("TWX_PAIR", "1")
    a = make_env(n, batch, first=first)
    monkeypatch.setenv("TWX_PAIR", "0")
    b = make_env(n, batch, first=first)
    total = 0
    for k in (3, 5, 11, 2, 120, 7, 260):
        for _ in range(k):
            a.step_random(seed, True, want_actions=False, want_status=False)
        total += k
        sa = a.export_state()
        b.import_state(sa)                      # the lane-per-board kernel continues from the pair kernel's state
        for _ in range(4):
            a.step_random(seed, True, want_actions=False, want_status=False)
            b.step_random(seed, True, want_actions=False, want_status=False)
        total += 4
        assert np.array_equal(u32(a.export_state()), u32(b.export_state()))
        ref = ob.rollout_batch(n, seed, first, batch, total, True)
        compare_planes(a, ref["planes"], what="after %d steps" % total)
    a.close()
    b.close()


@pytest.mark.parametrize("pair", [1, 0])
def test_external_actions_swaps_and_illegal_moves(torch_mod, pair, monkeypatch):
    """twx_step on N=24 (whole batch): numpy-chosen legal actions, forced swaps on move 1, an
    illegal move (occupied / out of range / opponent's border) for a third of the boards at
    some steps: untouched boards, TWX_ST_ILLEGAL, everything else as the oracle plays it."""
    import oracle_binding as ob
    torch = torch_mod
    monkeypatch.setenv("TWX_PAIR", str(pair))
    n, batch, steps = 24, 192, 90
    rng = np.random.default_rng(11)
    env = make_env(n, batch)
    boards = [ob.Board(n) for _ in range(batch)]
    first_action = np.zeros(batch, dtype=np.int64)
    for t in range(steps):
        acts = np.zeros(batch, dtype=np.int32)
        want_illegal = np.zeros(batch, dtype=bool)
        for b, bd in enumerate(boards):
            legal = bd.legal_actions()
            if not legal:                                   # finished game: anything is illegal
                acts[b], want_illegal[b] = 5, True
                continue
            a = legal[rng.integers(len(legal))]
            if t == 0:
                first_action[b] = a
            if t == 1 and b % 2 == 0 and first_action[b] in legal:
                a = int(first_action[b])                    # the swap (tb.cc:462-473)
            if t in (7, 30) and b % 3 == 0:
                a = [int(first_action[b]) if t == 7 else -3, n * n + 4, 0][b % 9 // 3]  # occupied / range / corner
                if a in legal:
                    a = -1
                want_illegal[b] = True
            acts[b] = a
            if not want_illegal[b]:
                bd.apply(a)
        st = env.step(torch.from_numpy(acts)).cpu().numpy()
        assert np.array_equal((st & 4) != 0, want_illegal), "illegal flags at step %d" % t
        assert np.array_equal(st & 3, np.array([bd.result for bd in boards], dtype=np.uint8)), "results at step %d" % t
        if t in (1, 2, 8, 31, steps - 1):
            want = np.stack([bd.planes() for bd in boards])
            got = u32(env.export_state())
            assert np.array_equal(got[:, :-2], want[:, :-2]), "state after step %d" % t
            res, mc, _ = env.status()
            assert np.array_equal(mc.cpu().numpy(), [bd.move_counter for bd in boards])
    assert any(bd.s.swapped for bd in boards)
    env.close()


@pytest.mark.parametrize("n,batch", [(22, 330), (17, 410), (12, 450)])
def test_whole_batch_and_listed_boards_interleaved(torch_mod, n, batch):
    """Whole-batch steps (which journal), listed-boards steps, play-outs of a few boards and status
    queries on one handle, in an order that leaves journal rows stale and boards reset in the middle
    of a period: every board follows the oracle board that made the same moves."""
    import oracle_binding as ob
    torch = torch_mod
    seed, first = 9, 77
    rng = np.random.default_rng(n)
    env = make_env(n, batch, first=first)
    boards = [ob.Board(n) for _ in range(batch)]

    def oracle_step(ids, auto_reset):
        acts, stat = np.zeros(len(ids), dtype=np.int32), np.zeros(len(ids), dtype=np.uint8)
        for j, b in enumerate(ids):
            a, st, _ = boards[b].rollout(seed, first + int(b), 1, auto_reset)
            acts[j], stat[j] = a[0], st[0]
        return acts, stat

    def oracle_planes():
        ref = np.stack([bd.planes() for bd in boards])
        ref[:, -2] = [bd.episode.value for bd in boards]   # (Board.planes leaves the episode word to the caller)
        return ref

    everyone = np.arange(batch)
    for t in range(150):
        kind = rng.integers(0, 10)
        if kind < 6:                                  # whole batch, auto-reset: appends a journal record
            a, st = env.step_random(seed, True)
            ra, rs = oracle_step(everyone, True)
            assert np.array_equal(a.cpu().numpy(), ra) and np.array_equal(st.cpu().numpy(), rs), "whole-batch step %d" % t
        elif kind < 8:                                # a few listed boards: flushes the journal first
            ids = np.sort(rng.choice(batch, size=int(rng.integers(1, 90)), replace=False)).astype(np.int32)
            a, st = env.step_random(seed, True, ids=torch.from_numpy(ids))
            ra, rs = oracle_step(ids, True)
            assert np.array_equal(a.cpu().numpy(), ra) and np.array_equal(st.cpu().numpy(), rs), "listed step %d" % t
        elif kind == 8:                               # status in the middle of a period: no flush
            res, mc, cp = env.status()
            assert np.array_equal(res.cpu().numpy(), [bd.result for bd in boards]), "status at %d" % t
            assert np.array_equal(mc.cpu().numpy(), [bd.move_counter for bd in boards])
        else:                                         # play three boards out (a compaction of listed boards)
            ids = np.sort(rng.choice(batch, size=3, replace=False)).astype(np.int32)
            st = env.random_rollout(seed, ids=torch.from_numpy(ids)).cpu().numpy()
            for j, b in enumerate(ids):
                boards[b].rollout(seed, first + int(b), n * n, False)
                assert (st[j] & 3) == boards[b].result and boards[b].result != 0
        if t % 25 == 24:
            compare_planes(env, oracle_planes(), what="after %d calls" % (t + 1))
    compare_planes(env, oracle_planes(), what="final")
    env.close()
