"""TwixtBatchEnv: B independent TwixT boards resident on one MI355X.

Host-side mirror of the reference's TwixTState surface (twixt.h:31-112), batched:
every method is one call through the C ABI of libtwixt_hip.so.  torch is used
for device buffers and streams only.
"""
import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class TwixtBatchEnv:
    def __init__(self, board_size=8, batch=1, device=0, first_board_id=0):
        if not torch.cuda.is_available():
            raise _lib.TwixtHipError("TwixtBatchEnv needs a HIP device; there is no CPU fallback")
        self.lib = _lib.load()
        self.board_size = int(board_size)
        self.batch = int(batch)
        self.device = torch.device("cuda", int(device))
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.twx_create(self.board_size, self.batch, int(device), int(first_board_id),
                                           C.byref(self._h)))
        self.state_words = self.lib.twx_state_words(self.board_size)
        self.mask_words = self.lib.twx_mask_words(self.board_size)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.twx_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _ids(self, ids):
        if ids is None:
            return None
        return torch.as_tensor(ids, dtype=torch.int32, device=self.device).contiguous()

    # TwixTGame::NewInitialState (twixt.h:118-120)
    def reset(self, ids=None):
        if ids is None:
            _lib.check(self.lib.twx_reset(self._h, self._stream()))
        else:
            t = self._ids(ids)
            _lib.check(self.lib.twx_reset_boards(self._h, _ptr(t), t.numel(), self._stream()))

    # State::ApplyAction -> TwixTState::DoApplyAction (twixt.h:93-104); with `ids` only
    # those (distinct) boards move: actions[i] / status[i] belong to board ids[i]
    def step(self, actions, status=None, ids=None):
        a = torch.as_tensor(actions, dtype=torch.int32, device=self.device).contiguous()
        t = self._ids(ids)
        n = self.batch if t is None else t.numel()
        assert a.numel() == n
        if status is None:
            status = torch.empty(n, dtype=torch.uint8, device=self.device)
        if t is None:
            _lib.check(self.lib.twx_step(self._h, _ptr(a), _ptr(status), self._stream()))
        else:
            _lib.check(self.lib.twx_step_boards(self._h, _ptr(t), n, _ptr(a), _ptr(status), self._stream()))
        return status

    def step_random(self, seed, auto_reset=True, actions_out=None, status=None, want_actions=True,
                    want_status=True, ids=None):
        t = self._ids(ids)
        n = self.batch if t is None else t.numel()
        if actions_out is None and want_actions:
            actions_out = torch.empty(n, dtype=torch.int32, device=self.device)
        if status is None and want_status:
            status = torch.empty(n, dtype=torch.uint8, device=self.device)
        if t is None:
            _lib.check(self.lib.twx_step_random(self._h, int(seed) & 0xFFFFFFFF, int(bool(auto_reset)),
                                                _ptr(actions_out), _ptr(status), self._stream()))
        else:
            _lib.check(self.lib.twx_step_random_boards(self._h, _ptr(t), n, int(seed) & 0xFFFFFFFF,
                                                       int(bool(auto_reset)), _ptr(actions_out), _ptr(status),
                                                       self._stream()))
        return actions_out, status

    # whole random play-outs in one launch (RandomRolloutEvaluator's inner loop)
    def random_rollout(self, seed, max_steps=None, ids=None, want_status=True):
        t = self._ids(ids)
        n = self.batch if t is None else t.numel()
        if max_steps is None:
            max_steps = self.board_size * self.board_size - 3  # MaxGameLength (twixt.h:136-139)
        status = torch.empty(n, dtype=torch.uint8, device=self.device) if want_status else None
        _lib.check(self.lib.twx_random_rollout(self._h, _ptr(t), 0 if t is None else n, int(seed) & 0xFFFFFFFF,
                                               int(max_steps), _ptr(status), self._stream()))
        return status

    # TwixTState::LegalActions (twixt.h:86-90) as masks
    def legal_mask(self, ids=None):
        t = self._ids(ids)
        n = self.batch if t is None else t.numel()
        mask = torch.empty((n, self.mask_words), dtype=torch.int32, device=self.device)
        count = torch.empty(n, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.twx_legal_mask_boards(self._h, _ptr(t), n, _ptr(mask), _ptr(count), self._stream()))
        return mask, count

    # result / move_counter / CurrentPlayer (twixtboard.h:60-61, twixt.h:38)
    def status(self, ids=None):
        t = self._ids(ids)
        n = self.batch if t is None else t.numel()
        res = torch.empty(n, dtype=torch.uint8, device=self.device)
        mc = torch.empty(n, dtype=torch.int32, device=self.device)
        cp = torch.empty(n, dtype=torch.int8, device=self.device)
        _lib.check(self.lib.twx_status_boards(self._h, _ptr(t), n, _ptr(res), _ptr(mc), _ptr(cp), self._stream()))
        return res, mc, cp

    # TwixTState::ObservationTensor (twixt.cc:101-132); `count`: the first `count` boards
    def observation(self, ids=None, count=None, out=None):
        t = self._ids(ids)
        m = (self.batch if count is None else int(count)) if t is None else t.numel()
        n = self.board_size
        if out is None:
            out = torch.empty((m, 12, n, n - 2), dtype=torch.float32, device=self.device)
        assert out.numel() == m * 12 * n * (n - 2) and out.dtype == torch.float32 and out.is_contiguous()
        _lib.check(self.lib.twx_observation_boards(self._h, _ptr(t), m, _ptr(out), self._stream()))
        return out

    # TwixTState::Clone (twixt.h:80-82)
    def copy_boards(self, dst_ids, src, src_ids):
        d, s = self._ids(dst_ids), src._ids(src_ids)
        assert d.numel() == s.numel()
        _lib.check(self.lib.twx_copy_boards(self._h, _ptr(d), src._h, _ptr(s), d.numel(), self._stream()))

    def export_state(self, ids=None):
        t = self._ids(ids)
        n = self.batch if t is None else t.numel()
        out = torch.empty((n, self.state_words), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.twx_export_state(self._h, _ptr(t), n, _ptr(out), self._stream()))
        return out

    def import_state(self, words, ids=None):
        t = self._ids(ids)
        w = torch.as_tensor(words, dtype=torch.int32, device=self.device).contiguous()
        n = w.shape[0]
        assert w.shape[1] == self.state_words
        _lib.check(self.lib.twx_import_state(self._h, _ptr(t), n, _ptr(w), self._stream()))

    def export_cells(self, ids=None):
        t = self._ids(ids)
        n = self.batch if t is None else t.numel()
        nn = self.board_size * self.board_size
        out = torch.empty((n, nn, 4), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.twx_export_cells(self._h, _ptr(t), n, _ptr(out), self._stream()))
        return out
