"""Batched-leaf UCT search over the device engine (BASELINE config 5).

What open_spiel's `mcts_example` does with one State at a time (Clone, descend with
LegalActions / ApplyAction, random play-out to the end, Returns) is done here for a whole
batch of leaves per round trip:

    tree (host)  --select L leaves with virtual loss-->  paths of actions
    device       twx_copy_boards(root -> L slots), twx_step_boards level by level,
                 twx_legal_mask_boards (expansion), twx_random_rollout (play-outs),
                 twx_status_boards (returns)
    tree (host)  <--back-propagate--

The tree never stores boards, only actions; a leaf's board is rebuilt from the root slot.
The evaluator is pluggable so that tests can run the identical search on the CPU oracle.
"""
import math

import numpy as np


class DeviceEvaluator:
    """Leaf batches on a TwixtBatchEnv: slots 0..L-1 are scratch leaves, slot L is the root."""

    def __init__(self, board_size, max_leaves, device=0, first_board_id=0):
        import torch
        from .batch_env import TwixtBatchEnv
        self.torch = torch
        self.n = board_size
        self.L = max_leaves
        self.env = TwixtBatchEnv(board_size, max_leaves + 1, device, first_board_id)
        self.root = max_leaves
        self.first = first_board_id

    def reset_root(self):
        self.env.reset([self.root])

    def advance_root(self, action):
        st = self.env.step([action], ids=[self.root])
        assert not int(st[0]) & 4, "illegal root move %d" % action
        return int(st[0]) & 3

    def root_info(self):
        res, mc, cp = self.env.status([self.root])
        mask, _ = self.env.legal_mask([self.root])
        return int(res[0]), int(cp[0]), _mask_to_actions(mask.cpu().numpy().view(np.uint32)[0], self.n)

    def evaluate(self, paths, seed, base=0):
        """paths: list of action lists from the root; leaf i is rebuilt in slot base + i (its
        policy key).  Returns per leaf (result after the path, player to move, legal actions,
        play-out result)."""
        torch = self.torch
        k = len(paths)
        assert base + k <= self.L
        ids = torch.arange(base, base + k, dtype=torch.int32, device=self.env.device)
        self.env.copy_boards(ids, self.env, torch.full((k,), self.root, dtype=torch.int32))
        depth = max((len(p) for p in paths), default=0)
        for t in range(depth):
            rows = [i for i, p in enumerate(paths) if len(p) > t]
            st = self.env.step([paths[i][t] for i in rows], ids=[base + i for i in rows])
            assert not (st & 4).any(), "tree path contains an illegal move"
        res, _, cp = self.env.status(ids)
        mask, _ = self.env.legal_mask(ids)
        res, cp = res.cpu().numpy(), cp.cpu().numpy()
        mask = mask.cpu().numpy().view(np.uint32)
        final = self.env.random_rollout(seed, ids=ids).cpu().numpy() & 3  # terminal leaves are left alone
        return [(int(res[i]), int(cp[i]), _mask_to_actions(mask[i], self.n) if res[i] == 0 else [], int(final[i]))
                for i in range(k)]


def _mask_to_actions(words, n):
    return [a for a in range(n * n) if (int(words[a >> 5]) >> (a & 31)) & 1]


def _returns(result):  # twixt.h:50-63
    return (1.0, -1.0) if result == 1 else (-1.0, 1.0) if result == 2 else (0.0, 0.0)


class Node:
    __slots__ = ("parent", "action", "mover", "children", "legal", "N", "W", "terminal", "player")

    def __init__(self, parent, action, mover):
        self.parent, self.action, self.mover = parent, action, mover
        self.children = None   # dict action -> Node once expanded
        self.legal = None
        self.N = 0
        self.W = 0.0           # summed return for `mover`, the player who played `action`
        self.terminal = None   # result if the game is over here
        self.player = None     # player to move here (known once expanded)


class BatchedMCTS:
    def __init__(self, evaluator, uct_c=2.0, leaves_per_batch=256, seed=1):
        self.ev = evaluator
        self.c = uct_c
        self.L = leaves_per_batch
        self.seed = seed
        self.batches = 0

    def _select(self, root):
        """One descent with virtual loss; returns (leaf node, path of actions)."""
        node, path = root, []
        while True:
            node.N += 1
            node.W -= 1.0  # virtual loss, undone in _backup
            if node.terminal is not None or node.children is None:
                return node, path
            best, best_v = None, -1e30
            logn = math.log(node.N)
            for a in node.legal:  # ascending action id: deterministic tie-break
                ch = node.children.get(a)
                if ch is None or ch.N == 0:
                    v = 1e9
                else:
                    v = ch.W / ch.N + self.c * math.sqrt(logn / ch.N)
                if v > best_v:
                    best, best_v = a, v
            ch = node.children.get(best)
            if ch is None:
                ch = node.children[best] = Node(node, best, node.player)
            node = ch
            path.append(best)

    def _backup(self, leaf, result):
        r = _returns(result)
        node = leaf
        while node is not None:
            node.W += 1.0  # undo the virtual loss
            if node.mover is not None:
                node.W += r[node.mover]
            node = node.parent

    def search(self, root_player, root_legal, max_simulations):
        root = Node(None, None, None)
        root.children, root.legal = {}, list(root_legal)
        root.player = root_player
        # The leaves of a round trip go out in two half-batches ("lanes"), and a lane's results are
        # applied only after the next lane has been selected -- launch(0), launch(1), finish(0),
        # launch(2), finish(1), ...: the order in which spiel_adapter/mcts_driver.cc overlaps the
        # device with the tree work.  Here a lane is simply evaluated when it is finished.
        # (with a single leaf slot there is nothing to split: both lanes use slot 0, as the driver does
        # whenever a half would not hold a leaf of every game)
        split = self.L >= 2
        lane_leaves = self.L // 2 if split else self.L
        flying, launched, lane = [], 0, 0

        def launch():
            nonlocal launched, lane
            k = min(lane_leaves, max_simulations - launched)
            flying.append(([self._select(root) for _ in range(k)], self.seed + self.batches,
                           lane * lane_leaves if split else 0))
            self.batches += 1
            launched += k
            lane ^= 1

        launch()
        while flying:
            if launched < max_simulations:
                launch()
            picks, seed, base = flying.pop(0)
            out = self.ev.evaluate([p for _, p in picks], seed, base)
            for (leaf, _), (res, cp, legal, final) in zip(picks, out):
                if res != 0:
                    leaf.terminal = res
                elif leaf.children is None:
                    leaf.children, leaf.legal, leaf.player = {}, legal, cp
                self._backup(leaf, res if res != 0 else final)
        visits = {a: ch.N for a, ch in root.children.items()}
        best = max(sorted(visits), key=lambda a: visits[a])
        return best, visits


def self_play(board_size=12, max_simulations=20000, leaves_per_batch=1024, max_moves=None, seed=1,
              evaluator=None, verbose=False):
    """mcts_example-style self-play: both sides search with the same budget.  Returns
    (history, result, total simulations)."""
    ev = evaluator or DeviceEvaluator(board_size, leaves_per_batch)
    ev.reset_root()
    mcts = BatchedMCTS(ev, leaves_per_batch=leaves_per_batch, seed=seed)
    history, sims = [], 0
    while True:
        res, player, legal = ev.root_info()
        if res != 0 or (max_moves is not None and len(history) >= max_moves):
            return history, res, sims
        a, visits = mcts.search(player, legal, max_simulations)
        sims += max_simulations
        if verbose:
            print("move %d: player %d plays %d (%d visits)" % (len(history), player, a, visits[a]))
        ev.advance_root(a)
        history.append(a)
