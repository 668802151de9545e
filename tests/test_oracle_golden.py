"""The CPU checker (oracle/) against every known-answer vector transcribed from the reference's
own unit tests.  This is what pins the oracle (the reference itself cannot run here)."""
import pytest

import vector_runner as VR

DATA = VR.load()


def test_agreed_commit_known_answers(oracle_lib):
    # src/ra_server.erl:4225-4238 agreed_commit_test
    for kat in DATA["agreed_commit"]:
        assert oracle_lib.agreed_commit(kat["indexes"]) == kat["expected"], kat


@pytest.mark.parametrize("v", DATA["vectors"], ids=[v["id"] for v in DATA["vectors"]])
def test_oracle_matches_reference_vector(oracle_lib, v):
    VR.run_vector(lambda g, n: oracle_lib.Oracle(g, n), v)
