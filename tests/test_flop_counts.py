"""Operation counts of the CPU restatement ("count them from the oracle's operation counter (instrument the
templated scalar) rather than guessing").  liboracle_count.so is the oracle compiled with the counter; profiles/flop_counts.json holds
the numbers bench.py reports in `roofline_fp64` (bench.py itself must not call into oracle/ outside its cpu_baseline leg).

    python tests/test_flop_counts.py        regenerates profiles/flop_counts.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PATH = os.path.join(ROOT, "profiles", "flop_counts.json")


def dense_algebra_counts(nx, nu, nc, nut):
    """Multiply-add counts (x 2 = flops) of the dense steps of the restatement behind the lineariser, per node / stage:
    FullPivLU of D (nc x nu) with the right-hand sides [C | e] carried along, the change of variables, one Riccati stage."""
    r = nu - nut                                   # rank of D
    lu = sum((nc - k - 1) * (nu - k - 1 + nx + 1) for k in range(r)) + r * r / 2.0 * (nx + 1 + nut)
    w = nx + 1 + nut                               # packed width [Px | Pe | Pu]
    cov = nx * nu * w + nu * nu * w + w * nu * w   # B X, R X, X' (R X)
    ric = nx * nx * w + nut * nx * w + nx * nx * (nx + 1) + nut ** 3 / 3.0 + nut * nut * (nx + 1) + 3 * nx * nut * (nx + 1)
    return dict(lu_projection=2 * lu, change_of_variables=2 * cov, riccati_stage=2 * ric)


def compute():
    from oracle import ingest, oracle_py, reference_py as rp
    from tests import oracle_bridge as ob
    out = {"unit": "double-precision operations of the CPU restatement (forward-mode AD over all nx + nu directions; one add / mul / div / "
                   "sqrt / sin / cos = 1)", "robots": {}}
    for robot, (nc, nut) in (("h1", (14, 9)), ("g1", (14, 11))):
        m = ob.model(robot)
        c = oracle_py.flop_counts(ingest.model_blob(m), m["initial_state"], rp.weight_compensating_input(m, 3), mode=1)
        c.update(dense_algebra_counts(m["nx"], m["nu"], nc, nut))
        out["robots"][robot] = {k: int(v) for k, v in c.items()}
    return out


def test_counts_match_the_committed_file():
    got = compute()
    ref = json.load(open(PATH))
    assert got["robots"] == ref["robots"]
    h1 = got["robots"]["h1"]
    # a value-only flow map is ~8 k operations; differentiating it in forward mode over 44 directions costs ~85 x that; a node
    # linearisation = two such Jacobians + the end-effector Jacobians + the RK2 products
    assert 5e3 < h1["flow_map"] < 2e4 and 60 < h1["flow_map_ad"] / h1["flow_map"] < 120
    assert h1["node_linearization"] > 2 * h1["flow_map_ad"] + h1["ee_kinematics_ad"]
    assert 5e4 < h1["riccati_stage"] < 3e5 and 5e4 < h1["change_of_variables"] < 3e5


if __name__ == "__main__":
    json.dump(compute(), open(PATH, "w"), indent=1)
    print(open(PATH).read())
