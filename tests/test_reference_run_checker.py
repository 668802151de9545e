"""tools/check_reference_run.py (the diff between erlang/ra_server_oracle.escript's output and the transcribed
expectations) exercised without OTP: observations are synthesised from what the CPU checker itself returns for every
vector step, in the escript's output format -- they must all match; a perturbed observation must be reported."""
import json
import os
import subprocess
import sys

import numpy as np

import vector_runner as VR
from ra_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROLE_NAMES = {v: k for k, v in VR.ROLE.items()}


def name(slot):
    return None if slot == abi.NONE else f"n{slot + 1}"


def observations(oracle_lib):
    """One escript-style line per vector step, produced by the CPU checker."""
    out = []
    for v in VR.load()["vectors"]:
        n, i = v["n_members"], VR.slot(v["self"])
        eng = oracle_lib.Oracle(1, n)
        init = VR.initial_state(v)
        eng.set_state(0, init)
        for sn, s in enumerate(v["steps"]):
            if s.get("reset"):
                eng.set_state(0, init)
            before = eng.get_state(0, n).copy()
            cur = before.copy()
            cur["role"][i] = VR.ROLE[s["as"]]
            if v.get("log_model") == "mem" and s["msg"].get("kind") == "written":
                cur["pending_first"][i] = int(cur["last_index"][i]) + 1
            eng.set_state(0, cur)
            dec, rpcs = eng.step(VR.make_msg(v, s["msg"]))
            d, row = dec[0], eng.get_state(0, n)[i]
            fl = int(d["flags"])
            if s.get("fork"):
                eng.set_state(0, before)                   # a fork step does not carry its state forward
            if fl & abi.F_INVARIANT:
                out.append({"id": v["id"], "step": sn, "status": "exit", "reason": str(int(d["invariant"]))})
                continue
            reply = None
            if fl & abi.F_REPLY:
                reply = {"to": name(int(d["reply_to"])), "term": int(d["reply_term"]),
                         "success": bool(fl & abi.F_REPLY_SUCCESS)}
                if fl & abi.F_REPLY_VOTE:
                    reply["vote"] = True
                elif fl & abi.F_REPLY_PRE_VOTE:
                    reply.update(pre_vote=True, token=int(d["reply_next_index"]))
                elif fl & abi.F_REPLY_HEARTBEAT:
                    reply.update(heartbeat=True, query_index=int(d["reply_next_index"]))
                else:
                    reply.update(next_index=int(d["reply_next_index"]), last_index=int(d["reply_last_index"]),
                                 last_term=int(d["reply_last_term"]))
            st = {"current_term": int(row["current_term"]), "commit_index": int(row["commit_index"]),
                  "last_applied": int(row["last_applied"]), "leader_id": name(int(row["leader_id"])),
                  "voted_for": name(int(row["voted_for"])), "votes": int(row["votes"]),
                  "last_index": int(row["last_index"]), "last_term": int(row["last_term"]),
                  "last_written": [int(row["last_written_index"]), int(row["last_written_term"])],
                  "query_index": int(row["query_index"]), "pre_vote_token": int(row["pre_vote_token"])}
            peers = {f"n{j + 1}": {"match_index": int(row["match_index"][j]), "next_index": int(row["next_index"][j]),
                                   "commit_index_sent": int(row["commit_index_sent"][j]),
                                   "query_index": int(row["peer_query_index"][j])} for j in range(n)}
            out.append({"id": v["id"], "step": sn, "status": "ok", "role": ROLE_NAMES[int(d["role"])], "state": st,
                        "peers": peers, "reply": reply,
                        "rpcs": [{"peer": name(int(r["peer"])), "term": int(r["term"]),
                                  "prev_log_index": int(r["prev_log_index"]), "prev_log_term": int(r["prev_log_term"]),
                                  "leader_commit": int(r["leader_commit"]), "n_entries": int(r["n_entries"])}
                                 for r in rpcs]})
        eng.close()
    return out


def run_checker(path):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_reference_run.py"), path],
                          capture_output=True, text=True)


def test_checker_accepts_the_checkers_own_observations_and_reports_a_perturbed_one(oracle_lib, tmp_path):
    obs = observations(oracle_lib)
    assert len(obs) == sum(len(v["steps"]) for v in VR.load()["vectors"])
    p = tmp_path / "observed.jsonl"
    p.write_text("\n".join(json.dumps(o) for o in obs) + "\n")
    r = run_checker(str(p))
    assert r.returncode == 0, r.stdout[-3000:]
    assert f"{len(obs)} steps match the reference, 0 differ, 0 skipped, 0 missing" in r.stdout
    # a wrong commit index, a missing step and a skipped one are all reported
    k = next(j for j, o in enumerate(obs) if o["status"] == "ok" and o["id"] == "F1" and o["step"] == 1)
    obs[k]["state"]["commit_index"] += 1
    obs[k + 1] = {"id": "F1", "step": 2, "status": "skipped", "reason": "test"}
    del obs[k + 2]
    p.write_text("\n".join(json.dumps(o) for o in obs) + "\n")
    r = run_checker(str(p))
    assert r.returncode == 1
    assert "F1 step 1" in r.stdout and "state.commit_index=2 expected 1" in r.stdout
    assert "1 differ, 1 skipped, 1 missing" in r.stdout


def test_the_escript_is_shipped_and_names_its_inputs():
    src = open(os.path.join(ROOT, "erlang", "ra_server_oracle.escript")).read()
    for needle in ("ra_log_memory", "meck:expect", "ra_server:Handler(Msg, S)", "json:decode", "base_state",
                   "empty_state", "NOT RUN IN THIS REPOSITORY"):
        assert needle in src
