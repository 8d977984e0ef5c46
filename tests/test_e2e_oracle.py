"""not gpu: the reference's END-TO-END known answers pin the oracle's L1-L3 chain (KKT assembly,
LDL', refinement, Zero/NN/SOC scalings and step operations, DefaultKKTSystem RHS algebra,
DefaultResiduals) through the IPM loop of tests/ipm_driver.py."""
import numpy as np
import pytest

from tests import e2e_problems as E
from tests import ipm_driver as ipm


def _run(oracle, pr, trace=None):
    be = ipm.OracleBackend(oracle, pr["n"], pr["m"], pr["P"], pr["A"], pr["q"], pr["b"], pr["cones"])
    return ipm.solve(be, pr["cones"], pr["q"], pr["b"], trace=trace)


@pytest.mark.parametrize("name", ["basic_qp", "basic_lp", "basic_socp", "basic_expcone", "basic_powcone", "basic_sdp", "basic_genpowcone", "basic_unconstrained",
                                  "basic_eq_constrained"])
def test_reference_end_to_end_answers(oracle, name):
    pr = getattr(E, name)()
    out = _run(oracle, pr)
    assert out["status"] == "Solved"
    if pr["x"] is not None:
        assert np.linalg.norm(out["x"] - np.array(pr["x"])) <= pr["tol"]   # basic_*.rs: x.dist(refsol)
    assert abs(out["obj_val"] - pr["obj"]) <= pr["tol"]
    assert out["iterations"] <= 30


def test_socp_sparse_soc_variant_solves(oracle):
    # basic_socp.rs:72-84: the SOC(6) variant exercises the sparse u/v expansion end to end
    out = _run(oracle, E.basic_socp(sparse_soc=True))
    assert out["status"] == "Solved"


def test_kktsystem_identities(oracle):
    """DefaultKKTSystem::solve output satisfies the homogeneous-embedding step equations it was
    derived from (kktsystem.rs:127-209): P dx + A' dz + q dtau = rhs.x,  A dx + ds - b dtau = -rhs.z ...
    checked through the residual definitions of residuals.rs:69-111 on the stepped point."""
    pr = E.basic_socp()
    tr = []
    out = _run(oracle, pr, trace=tr)
    assert out["status"] == "Solved"
    mus = [t[0] for t in tr]
    assert all(b < a for a, b in zip(mus, mus[1:]))  # mu decreases monotonically on this problem
    assert tr[-1][3] < 1e-8 and tr[-1][4] < 1e-8


def test_json_fixture_hs35(oracle):
    """the reference's on-disk problem format (default/json.rs:13-21; it ships HS35 as
    examples/data/hs35.json): Hock-Schittkowski 35, optimum x = (4/3, 7/9, 4/9), f = 1/9"""
    import os
    from tests import json_problem
    gold = os.path.join(os.path.dirname(__file__), "golden")
    pr = json_problem.load(os.path.join(gold, "hs35_reference.json"))  # byte-identical copy of the reference's file
    assert pr["settings"]["direct_solve_method"] == "qdldl"  # (what the reference saved it with)
    hand = json_problem.load(os.path.join(gold, "hs35.json"))  # the same problem entered by hand from its definition
    for key in ("P", "A"):  # (same pattern; the reference's file carries its values with a last-bit rounding of its own)
        assert np.array_equal(pr[key][0], hand[key][0]) and np.array_equal(pr[key][1], hand[key][1])
        assert np.allclose(pr[key][2], hand[key][2], rtol=1e-14, atol=0.0)
    assert np.allclose(pr["q"], hand["q"], rtol=1e-14) and np.allclose(pr["b"], hand["b"], rtol=1e-14) and pr["cones"] == hand["cones"]
    out = _run(oracle, pr)
    assert out["status"] == "Solved"
    assert np.linalg.norm(out["x"] - np.array([4.0 / 3.0, 7.0 / 9.0, 4.0 / 9.0])) <= 1e-6
    assert abs(out["obj_val"] + 9.0 - 1.0 / 9.0) <= 1e-6


@pytest.mark.parametrize("min_switch", [0.1, 0.999])
def test_mixed_conic_both_scaling_strategies(oracle, min_switch):
    """tests/mixed_conic.rs:4-45: Zero + NN + SOC + Power + Exponential in one problem; the second
    variant (min_switch_step_length = 0.999) forces the dual scaling and the barrier backtracking
    (solver.rs:571-584) through compute_barrier of every cone type"""
    pr = E.mixed_conic()
    be = ipm.OracleBackend(oracle, pr["n"], pr["m"], pr["P"], pr["A"], pr["q"], pr["b"], pr["cones"])
    out = ipm.solve(be, pr["cones"], pr["q"], pr["b"], min_switch_step_length=min_switch)
    assert out["status"] == "Solved"
    assert abs(out["obj_val"]) <= 1e-8 and np.linalg.norm(out["x"]) <= 1e-6
