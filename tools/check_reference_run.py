#!/usr/bin/env python3
"""Diff the output of erlang/ra_server_oracle.escript (the REFERENCE executing the transcribed vectors on a machine
with Erlang/OTP) against the expectations held in tests/golden/ra_server_suite_vectors.json -- the same `expect`
blocks the CPU checker and the HIP engine are tested against (tests/vector_runner.py).  Turns "pinned by
transcription" into "pinned by execution":

    python tools/check_reference_run.py observed.jsonl [vectors.json]

Exit status 0 when every driven step matches; skipped steps (the harness says why) are listed, never counted as
matches.  tests/test_reference_run_checker.py runs this checker on a synthetic observation file (no OTP here)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STATE_KEYS = ("current_term", "commit_index", "last_applied", "leader_id", "voted_for", "votes", "last_index",
              "last_term", "last_written", "query_index", "pre_vote_token")
REPLY_KEYS = ("to", "term", "next_index", "last_index", "last_term", "success", "query_index", "token")


def check_step(exp, obs):
    """-> list of mismatch strings"""
    bad = []
    if "invariant" in exp:
        if obs.get("status") not in ("exit", "error"):
            bad.append(f"expected the reference to exit (invariant {exp['invariant']}), it returned {obs.get('role')}")
        return bad
    if obs.get("status") != "ok":
        return [f"reference {obs.get('status')}: {obs.get('reason')}"]
    if "role" in exp and obs.get("role") != exp["role"]:
        bad.append(f"role {obs.get('role')} expected {exp['role']}")
    for k, v in (exp.get("state") or {}).items():
        if k in STATE_KEYS:
            got = obs["state"].get(k)
            if isinstance(v, list):
                got = list(got) if got is not None else None
            if got != v:
                bad.append(f"state.{k}={got} expected {v}")
    for name, p in (exp.get("peers") or {}).items():
        for k, v in p.items():
            # the vectors use the engine's field names; the reference's peer map calls it query_index
            got = (obs.get("peers") or {}).get(name, {}).get("query_index" if k == "peer_query_index" else k)
            if got != v:
                bad.append(f"peer {name}.{k}={got} expected {v}")
    if exp.get("no_reply") and obs.get("reply") is not None:
        bad.append(f"unexpected reply {obs['reply']}")
    if "reply" in exp:
        r, got = exp["reply"], obs.get("reply")
        if got is None:
            bad.append("no reply")
        else:
            for flag in ("vote", "pre_vote", "heartbeat"):
                if bool(r.get(flag, False)) != bool(got.get(flag, False)):
                    bad.append(f"reply kind {flag}: {got}")
            for k in REPLY_KEYS:
                if k in r and got.get(k) != r[k]:
                    bad.append(f"reply.{k}={got.get(k)} expected {r[k]}")
    if "rpcs" in exp:
        got = {r["peer"]: r for r in obs.get("rpcs") or []}
        if exp.get("rpcs_exact") and sorted(got) != sorted(e["peer"] for e in exp["rpcs"]):
            bad.append(f"rpc peers {sorted(got)} expected {sorted(e['peer'] for e in exp['rpcs'])}")
        for e in exp["rpcs"]:
            g = got.get(e["peer"])
            if g is None:
                bad.append(f"no rpc for {e['peer']}")
                continue
            if "term" in e and g.get("term") != e["term"]:
                bad.append(f"rpc {e['peer']}.term={g.get('term')} expected {e['term']}")
            if "prev" in e and [g.get("prev_log_index"), g.get("prev_log_term")] != list(e["prev"]):
                bad.append(f"rpc {e['peer']}.prev={g.get('prev_log_index')}:{g.get('prev_log_term')} expected {e['prev']}")
            if "commit" in e and g.get("leader_commit") != e["commit"]:
                bad.append(f"rpc {e['peer']}.leader_commit={g.get('leader_commit')} expected {e['commit']}")
            if "entries" in e:
                lo = g.get("prev_log_index", 0) + 1
                hi = g.get("prev_log_index", 0) + g.get("n_entries", 0)
                if [lo, hi] != list(e["entries"]):
                    bad.append(f"rpc {e['peer']}.entries={lo}..{hi} expected {e['entries']}")
    return bad


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 2
    vec_path = argv[2] if len(argv) > 2 else os.path.join(ROOT, "tests", "golden", "ra_server_suite_vectors.json")
    vectors = {v["id"]: v for v in json.load(open(vec_path))["vectors"]}
    obs = {}
    for line in open(argv[1]):
        line = line.strip()
        if line:
            o = json.loads(line)
            obs[(o["id"], o["step"])] = o
    n_ok = n_bad = n_skip = n_missing = 0
    for vid, v in vectors.items():
        if (vid, -1) in obs:
            print(f"{vid}: harness error before the first step: {obs[(vid, -1)].get('reason')}")
            n_bad += 1
            continue
        for sn, s in enumerate(v["steps"]):
            o = obs.get((vid, sn))
            if o is None:
                n_missing += 1
                print(f"{vid} step {sn}: not in the observation file")
                continue
            if o.get("status") == "skipped":
                n_skip += 1
                print(f"{vid} step {sn}: skipped by the harness ({o.get('reason')})")
                continue
            bad = check_step(s["expect"], o)
            if bad:
                n_bad += 1
                print(f"{vid} step {sn} ({v['source']}): " + "; ".join(bad))
            else:
                n_ok += 1
    print(f"{n_ok} steps match the reference, {n_bad} differ, {n_skip} skipped, {n_missing} missing")
    return 0 if n_bad == 0 and n_missing == 0 else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv))
