import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# fixtures that are not a plain "construct, assign inputs, call once" record of a shipped or
# run-time compiled model and have tests of their own
_SPECIAL = ("constraints_", "car_N200_warmstart", "ltv12x4_")


def golden_names(single=True, prefix=None):
    names = []
    for f in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        name = os.path.basename(f)[:-4]
        if prefix is not None:
            if name.startswith(prefix):
                names.append(name)
            continue
        if name.startswith(_SPECIAL):
            continue
        batched = "batch" in name
        if single != batched:
            names.append(name)
    return names


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def system_from_golden(g):
    """Rebuild the model-setup calls the fixture was generated with, on ilqr_amd.systems."""
    from ilqr_amd import systems, user_models
    name = str(g["model"])
    if name == "StepCountingLTV":
        # the reference solver ran on a step-counting System over the tables of the seeded
        # generator (oracle/make_golden.py); the tables themselves are not stored
        from ilqr_amd import problems
        A, Bm, _, _ = problems.ltv_system(N=g["target_states"].shape[0])
        assert np.array_equal(np.array([A.sum(), Bm.sum()]), g["table_sums"]), "problems.ltv_system changed"
        s = systems.LTVSystem(A, Bm, dt=float(g["dt"]))
    else:
        s = (getattr(systems, name, None) or getattr(user_models, name))()
    s.set_dt(float(g["dt"]))
    s.set_cost(np.array(g["Q"]), np.array(g["R"]))
    s.set_final_cost(np.array(g["Q_f"]))
    if bool(g["control_limited"]):
        s.set_control_limit(list(map(float, g["lo"])), list(map(float, g["hi"])))
    return s


def demo_car_system(limited=True):
    from ilqr_amd import problems, systems
    car = systems.Car()
    car.set_dt(0.2)
    car.set_cost(problems.CAR_Q.copy(), problems.CAR_R.copy())
    if limited:
        car.set_control_limit(*problems.CAR_LIMITS)
    return car


def noise_floor_mask(hist_cost, cost0, rel=1e-12):
    """Iterations whose accepted decrease exceeds rel * cost: before the round-off-driven tail
    of the line search, where accept/reject decisions are reproducible."""
    prev = np.concatenate([[cost0], hist_cost[:-1]])
    return (prev - hist_cost) > rel * np.abs(hist_cost)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc
