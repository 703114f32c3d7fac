#!/usr/bin/env python3
"""Extract golden vectors from the reference's playthrough fixture.

Reads  /root/reference/open_spiel/integration_tests/playthroughs/playthrough.txt
(a DATA file held by the reference's tests: one recorded N=8 game) and writes
tests/golden/playthrough_n8.json: the game facts, the 35 (action, action
string) pairs, and for every fully printed state its ObservationString,
ObservationTensor(0/1) bits, LegalActions, CurrentPlayer, IsTerminal, Returns.

Run only where /root/reference exists; the JSON is committed.
"""
import json
import os
import re
import sys

SRC = "/root/reference/open_spiel/integration_tests/playthroughs/playthrough.txt"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "playthrough_n8.json")


def unquote(s):
    s = s.strip()
    assert s[0] == '"' and s[-1] == '"', s[:40]
    body, out, i = s[1:-1], [], 0
    while i < len(body):  # C-style escapes written by the playthrough tool
        if body[i] == "\\" and i + 1 < len(body):
            out.append({"n": "\n", "\\": "\\", '"': '"'}.get(body[i + 1], body[i:i + 2]))
            i += 2
        else:
            out.append(body[i])
            i += 1
    return "".join(out)


def parse_tensor(lines, planes, rows, cols):
    """`rows` lines; each has `planes` groups of `cols` glyphs."""
    bits = [[[0] * cols for _ in range(rows)] for _ in range(planes)]
    for r, ln in enumerate(lines):
        groups = ln.rstrip("\n").split("  ")
        assert len(groups) == planes, (len(groups), ln)
        for p, g in enumerate(groups):
            assert len(g) == cols, g
            for c, ch in enumerate(g):
                assert ch in "◯◉", repr(ch)
                bits[p][r][c] = 1 if ch == "◉" else 0
    return bits


def main():
    if not os.path.exists(SRC):
        print("reference not present; nothing to do")
        return 0
    with open(SRC, encoding="utf-8") as f:
        lines = f.read().split("\n")
    facts = {}
    for ln in lines[:32]:
        m = re.match(r"^([A-Za-z_.]+(?:\(\))?) = (.*)$", ln)
        if m:
            facts[m.group(1)] = m.group(2)
    shape = json.loads(facts["ObservationTensorShape()"])
    planes, rows, cols = shape
    actions, states = [], []
    i = 0
    cur = None
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^# State (\d+)$", ln)
        if m:
            cur = {"index": int(m.group(1))}
            i += 1
            continue
        m = re.match(r'^# Apply action "(.*)"$', ln)
        if m:
            a = int(re.match(r"^action: (\d+)$", lines[i + 1]).group(1))
            actions.append({"action": a, "string": m.group(1)})
            i += 2
            continue
        if cur is not None:
            if ln.startswith("IsTerminal() = "):
                cur["is_terminal"] = ln.endswith("True")
                states.append(cur)
            elif ln.startswith("History() = "):
                cur["history"] = json.loads(ln.split(" = ", 1)[1])
            elif ln.startswith("CurrentPlayer() = "):
                cur["current_player"] = int(ln.split(" = ", 1)[1])
            elif ln.startswith("ObservationString(0) = "):
                cur["observation_string"] = unquote(ln.split(" = ", 1)[1])
            elif ln.startswith("ObservationString(1) = "):
                assert unquote(ln.split(" = ", 1)[1]) == cur["observation_string"]
            elif ln.startswith("InformationStateString("):
                cur.setdefault("information_state_string", unquote(ln.split(" = ", 1)[1]))
            elif re.match(r"^ObservationTensor\((\d)\):$", ln):
                p = int(ln[len("ObservationTensor(")])
                cur["tensor%d" % p] = parse_tensor(lines[i + 1:i + 1 + rows], planes, rows, cols)
                i += rows
            elif ln.startswith("Returns() = "):
                cur["returns"] = json.loads(ln.split(" = ", 1)[1])
            elif ln.startswith("LegalActions() = "):
                cur["legal_actions"] = json.loads(ln.split(" = ", 1)[1])
            elif ln.startswith("StringLegalActions() = "):
                cur["string_legal_actions"] = json.loads(ln.split(" = ", 1)[1])
        i += 1
    out = {
        "source": "open_spiel/integration_tests/playthroughs/playthrough.txt (reference fixture)",
        "facts": facts,
        "actions": actions,
        "states": states,
    }
    with open(DST, "w", encoding="utf-8") as f:
        json.dump(out, f, indent=0, ensure_ascii=True)
    print("wrote", DST, "states:", [s["index"] for s in states], "actions:", len(actions))
    return 0


if __name__ == "__main__":
    sys.exit(main())
