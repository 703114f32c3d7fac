"""tools/debug_steps.py -- diagnostic: where does the engine first leave the oracle?  (GPU box only.)

Steps a batch through twx_step_random and compares actions / status bytes of every step with the oracle's
trace; then re-runs fresh handles for k steps (k around the first mismatch) and compares the exported state
with the oracle plane by plane, to show which plane and word went wrong first.

  python tools/debug_steps.py N BATCH STEPS [FIRST=4000] [SEED=77]
Knobs (TWX_JOURNAL_PERIOD, TWX_WINDOW, ...) come from the environment.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PLANES = ["red", "blue", "NNE", "ENE", "ESE", "SSE", "START", "END"]


def main():
    import oracle_binding as ob
    from twixt_for_open_spiel_amd import TwixtBatchEnv
    n, batch, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    first = int(sys.argv[4]) if len(sys.argv) > 4 else 4000
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 77
    W = (n * n + 31) // 32
    env = TwixtBatchEnv(n, batch, 0, first)
    acts = np.zeros((batch, steps), dtype=np.int32)
    stat = np.zeros((batch, steps), dtype=np.uint8)
    for t in range(steps):
        a, s = env.step_random(seed, True)
        acts[:, t] = a.cpu().numpy()
        stat[:, t] = s.cpu().numpy()
    ref = ob.rollout_batch(n, seed, first, batch, steps, True, want_planes=True, want_trace=True)
    bad_a = np.argwhere(ref["actions"] != acts)
    bad_s = np.argwhere(ref["status"] != stat)
    print("N=%d batch=%d steps=%d: %d action mismatches, %d status mismatches" % (n, batch, steps, len(bad_a), len(bad_s)))
    if not len(bad_a) and not len(bad_s):
        got = env.export_state().cpu().numpy().view(np.uint32)
        print("final planes equal:", np.array_equal(got[:, :-1], ref["planes"][:, :-1]))
        return 0
    tmin = min([x[1] for x in bad_a] + [x[1] for x in bad_s])
    boards = sorted(set([int(x[0]) for x in bad_a if x[1] <= tmin + 2] + [int(x[0]) for x in bad_s if x[1] <= tmin + 2]))
    print("first mismatch at step", tmin, "boards", boards[:10])
    for b in boards[:3]:
        lo = max(0, tmin - 3)
        print(" board %d: ref actions %s status %s" % (b, ref["actions"][b, lo:tmin + 2], [hex(v) for v in ref["status"][b, lo:tmin + 2]]))
        print("           got actions %s status %s" % (acts[b, lo:tmin + 2], [hex(v) for v in stat[b, lo:tmin + 2]]))
    # which plane went wrong, and when: fresh handles stepped k times, exported (that materialises the journal)
    for k in range(max(1, tmin - 12), tmin + 2):
        e2 = TwixtBatchEnv(n, batch, 0, first)
        for _ in range(k):
            e2.step_random(seed, True, want_actions=False, want_status=False)
        got = e2.export_state().cpu().numpy().view(np.uint32)
        r2 = ob.rollout_batch(n, seed, first, batch, k, True, want_planes=True)["planes"]
        diff = np.argwhere(got[:, :-1] != r2[:, :-1])
        e2.close()
        if len(diff):
            print("after %d steps: %d words differ; first:" % (k, len(diff)))
            for b, w in diff[:8]:
                what = "%s word %d" % (PLANES[w // W], w % W) if w < 8 * W else "scalar %d" % (w - 8 * W)
                print("   board %d %s: ref %08x got %08x (xor %08x)  last ref actions %s" % (
                    b, what, r2[b, w], got[b, w], r2[b, w] ^ got[b, w], ref["actions"][b, max(0, k - 3):k]))
            break
    else:
        print("exported planes equal up to step", tmin + 1)
    return 1


if __name__ == "__main__":
    sys.exit(main())
