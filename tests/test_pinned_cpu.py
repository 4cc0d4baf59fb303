"""CPU tests against the fixtures that pin what used to be checked only by this repository's own
restatements (all produced by running the reference's files, oracle/make_golden.py):

* ``constraints_seed*.npz``: outputs of the reference ``constraints.py:20-48`` -> the host class
  ``ilqr_amd.constraints.BubbleConstraint``, the h / dh_dx inside the oracle's barrier terms and the
  constraint rows the sequential-QP oracle builds;
* ``ltv12x4_*.npz``: the reference ``ilqr.py`` on a step-counting System over the n = 12, m = 4
  tables -> the oracle's table-driven model;
* ``car_N200_warmstart.npz``: the reference solver object called a second time, plainly and after a
  shifted warm start -> the oracle's continuation semantics (``ilqr.py:107-111``).
"""
import numpy as np
import pytest

from conftest import golden_names, load_golden, noise_floor_mask, system_from_golden


def nrm(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("name", golden_names(prefix="constraints_"))
def test_bubble_constraint_class_matches_reference_outputs(name):
    from ilqr_amd.constraints import BubbleConstraint
    g = load_golden(name)
    st, c, r, vb = g["states"], g["centers"], g["radius"], g["vel_bounds"]
    N = st.shape[0]
    con = BubbleConstraint(N)
    con.setup(c, r, vb)
    assert con.state_size == 4 and con.constraint_size == 2 and con.horizon == N
    for i in range(N):
        assert con.get_h(st[i], c[i], r[i]) == g["h"][i]
        assert np.array_equal(con.dh_dx(st[i], c[i]), g["dh_dx"][i])
        assert con.get_uieq(st[i], c[i], r[i]) == g["uieq"][i]
    assert np.array_equal(con.get_linear_constraint(st), g["Cu"])
    xmin, xmax = con.get_bounds(st)
    assert np.array_equal(xmin, g["xmin"]) and np.array_equal(xmax, g["xmax"])


@pytest.mark.parametrize("name", golden_names(prefix="constraints_"))
def test_sqp_oracle_rows_match_reference_outputs(name):
    """The rows ``lo <= G x <= hi`` the sequential-QP oracle hands to its QP are the blocks of the
    reference's ``get_linear_constraint`` / ``get_bounds``."""
    from conftest import demo_car_system
    from ilqr_amd.constraints import BubbleConstraint
    from oracle import sqp_oracle
    g = load_golden(name)
    st = g["states"]
    N = st.shape[0]
    con = BubbleConstraint(N)
    con.setup(g["centers"], g["radius"], g["vel_bounds"])
    _, _, _, G, lo, hi = sqp_oracle.qp_data(demo_car_system(), con, st, st, np.zeros((N - 1, 2)))
    for i in range(N):
        assert np.array_equal(G[i], g["Cu"][2 * i:2 * i + 2, 4 * i:4 * i + 4])
    assert np.array_equal(lo.ravel(), g["xmin"]) and np.array_equal(hi.ravel(), g["xmax"])


@pytest.mark.parametrize("name", golden_names(prefix="constraints_"))
def test_oracle_barrier_is_built_on_the_reference_h(oracle, name):
    """beta(-h) with the reference's own h and dh/dx: above the relaxation width the barrier value is
    -mu [log(-h) + log(v - vlo) + log(vhi - v)] and its position gradient mu * dh/dx / (-h); below
    it (outside the bubble: h > 0) the quadratic extension of the same z = -h."""
    from conftest import demo_car_system
    g = load_golden(name)
    mu, delta = 2.5, 1e-9
    vlo, vhi = map(float, g["inside_vel_bounds"])
    d = oracle.make_desc(demo_car_system(), 10, barrier=True, mu=mu, delta=delta, vel_bounds=(vlo, vhi))
    st, tg, rad = g["inside_states"], g["inside_targets"], g["inside_radius"]
    for i in range(st.shape[0]):
        v, grad, _ = oracle.barrier(d, st[i], tg[i, :2], float(rad[i]))
        h = float(g["inside_h"][i])
        want = -mu * ((np.log(-h) + np.log(st[i, 2] - vlo)) + np.log(vhi - st[i, 2]))
        assert abs(v - want) <= 1e-13 * abs(want) + 1e-13
        assert np.allclose(grad[:2], mu * g["inside_dh_dx"][i][0, :2] / (-h), rtol=1e-13, atol=0)
    # outside points of the general set: relaxed branch, z = -h <= delta
    delta = 0.4
    d = oracle.make_desc(demo_car_system(), 10, barrier=True, mu=mu, delta=delta, vel_bounds=(-1e9, 1e9))
    base = -2.0 * np.log(1e9)       # the two speed terms, far from their bounds
    for i in range(g["states"].shape[0]):
        h = float(g["h"][i])
        v, grad, _ = oracle.barrier(d, g["states"][i], g["centers"][i], float(g["radius"][i]))
        z = -h
        beta = -np.log(z) if z > delta else 0.5 * (((z - 2 * delta) / delta) ** 2 - 1) - np.log(delta)
        d1 = -1 / z if z > delta else (z - 2 * delta) / delta ** 2
        x2 = g["states"][i][2]
        sp = -np.log(x2 + 1e9) - np.log(1e9 - x2)
        assert abs(v - mu * (beta + sp)) <= 1e-9 * abs(mu * (beta + sp))
        assert np.allclose(grad[:2], mu * d1 * -g["dh_dx"][i][0, :2], rtol=1e-12, atol=1e-12)
        assert base < 0


@pytest.mark.parametrize("name", golden_names(prefix="constraints_"))
def test_oracle_barrier_cost_identity(oracle, name):
    """J(barrier) - J(plain) = -mu * sum_{i >= 1} [log(-h_ref_i) + log(v_i - vlo) + log(vhi - v_i)] on a
    trajectory strictly inside every bubble, h_ref from the reference's get_h."""
    from conftest import demo_car_system
    g = load_golden(name)
    car = demo_car_system()
    st, tg, ui, rad = (np.ascontiguousarray(g[k]) for k in ("inside_states", "inside_targets", "inside_inputs", "inside_radius"))
    N = st.shape[0]
    vlo, vhi = map(float, g["inside_vel_bounds"])
    mu = 7.0
    plain = oracle.cost(oracle.make_desc(car, N), tg, st, ui)
    db = oracle.make_desc(car, N, barrier=True, mu=mu, delta=1e-9, vel_bounds=(vlo, vhi))
    with_b = oracle.cost(db, tg, st, ui, centers=np.ascontiguousarray(tg[:, :2]), radius=rad)
    logs = np.log(-g["inside_h"][1:]) + np.log(st[1:, 2] - vlo) + np.log(vhi - st[1:, 2])
    assert abs((with_b - plain) - (-mu * logs.sum())) <= 1e-10 * abs(mu * logs.sum())


@pytest.mark.parametrize("name", golden_names(prefix="ltv12x4_"))
def test_oracle_ltv_matches_reference_fixture(oracle, name):
    """The n = 12, m = 4 table-driven model against the REFERENCE solver (driven through a
    step-counting System).  A clamped linear-quadratic problem: a handful of real iterations, then
    round-off; trajectories are compared where the fixture holds snapshots."""
    g = load_golden(name)
    s = system_from_golden(g)
    N = g["target_states"].shape[0]
    d = oracle.make_desc(s, N)
    st, ui = np.array(g["init_states"]), np.array(g["init_inputs"])
    r = oracle.solve(d, np.ascontiguousarray(g["target_states"]), st, ui)
    assert abs(r["cost0"] - g["cost0"]) <= 1e-12 * abs(g["cost0"])
    assert nrm(r["K_first"], g["K_first"]) <= 1e-11 and nrm(r["k_first"], g["k_first"]) <= 1e-11
    it = int(g["iters"])
    solid = noise_floor_mask(g["hist_cost"][:it], float(g["cost0"]))
    last_solid = int(np.nonzero(solid)[0].max()) + 1
    n = min(r["iters"], it)
    assert n >= last_solid
    rel = np.abs(r["hist_cost"][:n] - g["hist_cost"][:n]) / np.abs(g["hist_cost"][:n])
    assert rel.max() <= 1e-9
    assert np.array_equal(r["hist_alpha"][:n][solid[:n]], g["hist_alpha"][:n][solid[:n]])
    if "hist_states" in g.files:
        for k in range(last_solid):
            assert nrm(r["hist_states"][k], g["hist_states"][k]) <= 1e-9
            assert nrm(r["hist_inputs"][k], g["hist_inputs"][k]) <= 1e-8
    assert abs(r["min_cost"] - float(g["min_cost"])) <= 1e-9 * float(g["min_cost"])


def test_oracle_second_call_and_shifted_warm_start(oracle):
    """ilqr.py:107-111: a second call recomputes min_cost from the attributes and continues;
    `converge` is whatever it was."""
    g = load_golden("car_N200_warmstart")
    s = system_from_golden(g)
    N = g["target_states"].shape[0]
    tg = np.ascontiguousarray(g["target_states"])
    st, ui = np.array(g["init_states"]), np.array(g["init_inputs"])
    d6 = oracle.make_desc(s, N, max_iter=int(g["first_max_iter"]))
    r1 = oracle.solve(d6, tg, st, ui)
    assert r1["iters"] == int(g["first_iters"]) and r1["converge"] == bool(g["first_converge"])
    assert np.abs(r1["hist_cost"] - g["first_hist_cost"]).max() <= 1e-9 * g["first_hist_cost"].max()
    assert nrm(st, g["first_states"]) <= 1e-9
    # the plain second call, continuing from the oracle's own first result
    r2 = oracle.solve(d6, tg, st, ui, converge=r1["converge"])
    assert abs(r2["cost0"] - float(g["again_cost0"])) <= 1e-9 * float(g["again_cost0"])
    assert r2["iters"] == int(g["again_iters"])
    assert (np.abs(r2["hist_cost"] - g["again_hist_cost"]) / g["again_hist_cost"]).max() <= 1e-9
    assert np.array_equal(r2["hist_alpha"], g["again_hist_alpha"])
    assert nrm(st, g["again_states"]) <= 1e-9
    # the shifted warm start, from the fixture's first result (so the comparison does not inherit
    # the first call's rounding)
    sh = int(g["shift"])
    xs = np.concatenate([g["first_states"][sh:], np.repeat(g["first_states"][-1:], sh, axis=0)])
    us = np.concatenate([g["first_inputs"][sh:], np.repeat(g["first_inputs"][-1:], sh, axis=0)])
    xs[0] = g["x0_new"]
    d8 = oracle.make_desc(s, N, max_iter=int(g["shifted_max_iter"]))
    r3 = oracle.solve(d8, np.ascontiguousarray(g["target_states_shifted"]), xs, us)
    assert abs(r3["cost0"] - float(g["shifted_cost0"])) <= 1e-12 * float(g["shifted_cost0"])
    assert r3["iters"] == int(g["shifted_iters"])
    assert (np.abs(r3["hist_cost"] - g["shifted_hist_cost"]) / g["shifted_hist_cost"]).max() <= 1e-9
    assert np.array_equal(r3["hist_alpha"], g["shifted_hist_alpha"])
    assert nrm(xs, g["shifted_states"]) <= 1e-9 and nrm(us, g["shifted_inputs"]) <= 1e-8
