"""The numpy model of the stage-2 bulge chasing (tools/experiments/bc_pair_dependency.py; DESIGN.md section 7, the row
on two columns per visit): the textbook order of the reflectors gives the tridiagonal matrix with the band matrix's
eigenvalues; every reflector H(s+1,t) reads entries H(s,t+1) wrote and every H(s,t+1) entries H(s,t) wrote, so the
chain of dependent reflectors grows by about three per sweep; a pair of sweeps applied together at one band location
keeps the eigenvalues and loses the tridiagonal form."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    spec = importlib.util.spec_from_file_location(
        "bc_pair_dependency", os.path.join(ROOT, "tools", "experiments", "bc_pair_dependency.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_chase_has_two_messages_per_sweep_on_its_critical_chain():
    r = _model().main(96, 6)
    assert r["legal_off_tridiagonal"] < 1e-13 and r["legal_eig_err"] < 1e-13
    assert r["frac_needing_same_sweep_previous_location"] == 1.0
    assert r["frac_needing_previous_sweep_next_location"] == 1.0
    assert 2.5 < r["chain_per_sweep"] <= 3.0
    assert r["chain_head"][:7] == [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2), (2, 0)]


def test_a_pair_of_sweeps_at_one_location_is_not_a_tridiagonalisation():
    r = _model().main(64, 4)
    assert r["dense_model_diff"] < 1e-11          # the dense model follows the band model in the legal order
    assert r["paired_eig_err"] < 1e-12            # still an orthogonal similarity ...
    assert r["paired_off_tridiagonal"] > 1e-2     # ... but not tridiagonal
