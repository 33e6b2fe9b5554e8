"""BASELINE.json configs[3]: Unitree G1 (nx = nu = 24, hip pitch / roll / yaw, knee, ankle pitch / roll - a different joint order and
tilted hip / knee / ankle joint frames compared with H1 and OpenLoong) walking, i.e. following `standing_trot` (SURVEY.md section 8d
"Config 4").  The reference ships no OCS2 configuration for G1: sole frames and INFO files are authored by tools/make_assets.py, so
these results are SELF-DEFINED, NOT REFERENCE PARITY - they check the HIP path against the oracle on that configuration.
Tolerances as for H1: LQ model 1e-11, QP step 1e-9, solve outputs 1e-8 (relative to max(1, |oracle|_max))."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.tolerances import rel_K, rel_u, rel_x  # noqa: E402  (per physical block: forces vs joint velocities, ...)
ROBOT = "g1"
WALK = "standing_trot"


def _rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.fixture(scope="module")
def ctx():
    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios
    from tests import oracle_bridge as ob
    itf = scenarios.interface(ROBOT)
    assert itf.stateDim == 24 and itf.jointNames()[0] == "left_hip_pitch_joint"
    return dict(bp=bp, sc=scenarios, ob=ob, itf=itf)


@pytest.mark.parametrize("gait", [WALK, "trot", "flying_trot"])
def test_g1_linearize_matches_oracle(ctx, gait):
    bp, sc, ob, itf = ctx["bp"], ctx["sc"], ctx["ob"], ctx["itf"]
    B, NN = 3, 56
    prob = sc.trot_problem(itf, batch=B, n_intervals=36, gait=gait)
    mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=NN, materialize_lq=True)
    lay = mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    mpc.enqueue(); mpc.synchronize()          # one accepted step: a generic iterate
    mpc.stage("linearize"); mpc.synchronize()
    nx = nu = 24
    x = mpc.read("x").reshape(B, NN + 1, nx); u = mpc.read("u").reshape(B, NN, nu)
    shapes = dict(A=(nx, nx), B=(nx, nu), b=(nx,), Q=(nx, nx), R=(nu, nu), P=(nu, nx), q=(nx,), r=(nu,), c=(), C=(16, nx), D=(16, nu), e=(16,), perf=(3,))
    dev = {k: mpc.read(k).reshape(B, NN, *s) for k, s in shapes.items()}
    nc = mpc.read("nc").reshape(B, NN)
    om = ob.oracle(ROBOT)
    worst = {}
    modes = set()
    for b in range(B):
        nodes = ob.oracle_nodes(prob, b, robot=ROBOT)
        assert nodes["N"] == lay["n_nodes_max"]
        for k in range(nodes["N"]):
            o = om.node_lq(nodes["kind"][k], nodes["dt"][k], x[b, k], u[b, k], x[b, k + 1], nodes["xref"][k], nodes["mode"][k], nodes["zref"][k], nodes["zdref"][k])
            assert int(nc[b, k]) == o["nc"]
            modes.add(int(nodes["mode"][k]))
            for name in shapes:
                worst[name] = max(worst.get(name, 0.0), _rel(dev[name][b, k], o[name]))
    assert len(modes) >= (3 if gait == WALK else 2)      # standing_trot: LF, RF and double stance; trot: LF, RF; flying trot: + flight
    assert max(worst.values()) < 1e-11, worst


def test_g1_qp_step_matches_oracle(ctx):
    bp, sc, ob, itf = ctx["bp"], ctx["sc"], ctx["ob"], ctx["itf"]
    B, NN = 2, 64
    prob = sc.trot_problem(itf, batch=B, n_intervals=40, gait=WALK)
    mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=NN, return_gains=True)
    mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    mpc.enqueue(); mpc.synchronize()
    for st in ("linearize", "project", "riccati"):
        mpc.stage(st)
    mpc.synchronize()
    nx = nu = 24
    x = mpc.read("x").reshape(B, NN + 1, nx); u = mpc.read("u").reshape(B, NN, nu)
    dx = mpc.read("dx").reshape(B, NN + 1, nx); du = mpc.read("du").reshape(B, NN, nu); K = mpc.read("K").reshape(B, NN, nu, nx)
    om = ob.oracle(ROBOT)
    for b in range(B):
        nodes = ob.oracle_nodes(prob, b, robot=ROBOT)
        N = nodes["N"]
        odx, odu, oK = om.qp_step(nodes, prob["x0"][b], x[b, :N + 1], u[b, :N])
        assert rel_x(dx[b, :N + 1], odx) < 1e-9 and rel_u(du[b, :N], odu) < 1e-9 and rel_K(K[b, :N], oK) < 1e-9


@pytest.mark.parametrize("gait,iterations", [(WALK, 1), (WALK, 3), ("trot", 2)])
def test_g1_solve_matches_oracle(ctx, gait, iterations):
    bp, sc, ob, itf = ctx["bp"], ctx["sc"], ctx["ob"], ctx["itf"]
    B = 4
    prob = sc.trot_problem(itf, batch=B, n_intervals=50, gait=gait)
    mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=72, sqp_iterations=iterations, return_gains=True)
    t, x, u, K, stats = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)
    for b in range(B):
        xo, uo, Ko, st = ob.oracle_solve_like(prob, b, iterations=iterations, robot=ROBOT)
        n = stats[b].n_nodes
        its = int(sum(1 for r in st if r[10] > 0))
        assert stats[b].iterations == its and stats[b].step_size == st[its - 1][3]
        assert rel_x(x[b, :n + 1], xo) < 1e-11 and rel_u(u[b, :n], uo) < 1e-11 and rel_K(K[b, :n], Ko) < 1e-10
    # reference kernel bodies (lane-emulation verified on the CPU tier) agree with the fast ones
    ref = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=72, sqp_iterations=iterations, reference_kernels=True)
    t2, x2, u2, _, _ = ref.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    assert rel_x(x2, x) < 1e-9 and rel_u(u2, u) < 1e-9


def test_g1_full_size_properties(ctx):
    """configs[3] at its full size (batch 1024, horizon 100, walk), through size-independent properties: the QP step satisfies the
    linearised dynamics and the eliminated equality rows, three SQP iterations bring the violation down for every problem, a
    problem's solution is bitwise independent of its batch neighbours, and a sample of problems matches the oracle."""
    bp, sc, ob, itf = ctx["bp"], ctx["sc"], ctx["ob"], ctx["itf"]
    B, N, NN = 1024, 100, 120
    prob = sc.trot_problem(itf, batch=B, n_intervals=N, gait=WALK)
    mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=NN, sqp_iterations=3, materialize_lq=True)
    mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    for st in ("linearize", "project", "riccati"):
        mpc.stage(st)
    mpc.synchronize()
    nx = nu = 24
    n = mpc.layout()["n_nodes_max"]
    S = slice(0, 256)                        # the dense checks read a quarter of the batch back (the LQ model of all 1024 is 4 GB)
    dx = mpc.read("dx").reshape(B, NN + 1, nx)[S]; du = mpc.read("du").reshape(B, NN, nu)[S]
    A = mpc.read("A").reshape(B, NN, nx, nx)[S]; Bm = mpc.read("B").reshape(B, NN, nx, nu)[S]; bv = mpc.read("b").reshape(B, NN, nx)[S]
    res_dyn = np.einsum("bkij,bkj->bki", A[:, :n], dx[:, :n]) + np.einsum("bkij,bkj->bki", Bm[:, :n], du[:, :n]) + bv[:, :n] - dx[:, 1:n + 1]
    assert np.abs(res_dyn).max() < 1e-9
    del A, Bm
    C = mpc.read("C").reshape(B, NN, 16, nx)[S]; D = mpc.read("D").reshape(B, NN, 16, nu)[S]; e = mpc.read("e").reshape(B, NN, 16)[S]
    kind = mpc.read("g_kind")[:n]
    inter = kind == 0
    res_eq = np.einsum("bkij,bkj->bki", C[:, :n], dx[:, :n]) + np.einsum("bkij,bkj->bki", D[:, :n], du[:, :n]) + e[:, :n]
    nc = mpc.read("nc").reshape(B, NN)[S, :n]
    nut = mpc.read("nut").reshape(B, NN)[S, :n]
    violated = (np.abs(res_eq) > 1e-8).sum(axis=2)
    dropped = nc - (nu - nut)
    assert np.all(violated[:, inter] <= dropped[:, inter]) and dropped[:, inter].max() <= 2
    mpc.reset(); mpc.enqueue(); mpc.synchronize()
    t, x, u, K, stats = mpc.fetch()
    assert all(s.status == 0 for s in stats)
    viol0 = np.array([np.sqrt(s.dynamics_sse_before + s.equality_sse_before) for s in stats])
    viol1 = np.array([np.sqrt(s.dynamics_sse_after + s.equality_sse_after) for s in stats])
    # cold-start violation is O(1); the G1 walk converges more slowly than the H1 trot (oracle: 1.0 -> 0.27 -> 0.05 -> 0.03 typical)
    assert np.all(viol1 < 0.15) and np.median(viol1) < 0.06 and np.all(viol1 <= viol0 + 1e-12)
    sub = [1000, 3, 517]
    prob2 = dict(prob, x0=prob["x0"][sub], targets=[prob["targets"][i] for i in sub])
    mpc2 = bp.BatchedSqpMpc(itf, max_batch=3, max_nodes=NN, sqp_iterations=3)
    t2, x2, u2, _, st2 = mpc2.run(prob2["t0"], prob2["x0"], prob2["schedule"], prob2["targets"], horizon=prob2["horizon"])
    # the batch of 1024 runs the four-wave Riccati sweep (two problems per CU), a batch that fits one problem per CU the eight-wave
    # one (riccati_mfma8.h): the same mathematics in another elimination order, so the two agree to rounding, not bitwise
    for j, i in enumerate(sub):
        assert rel_x(x2[j], x[i]) < 1e-10 and rel_u(u2[j], u[i]) < 1e-10
    # ... and bitwise between two batches on the same sweep: another size, another order
    sub3 = [517, 1000, 3, 42]
    prob3 = dict(prob, x0=prob["x0"][sub3], targets=[prob["targets"][i] for i in sub3])
    mpc3 = bp.BatchedSqpMpc(itf, max_batch=4, max_nodes=NN, sqp_iterations=3)
    _, x3, u3, _, _ = mpc3.run(prob3["t0"], prob3["x0"], prob3["schedule"], prob3["targets"], horizon=prob3["horizon"])
    for j, i in enumerate(sub):
        assert np.array_equal(x2[j], x3[sub3.index(i)]) and np.array_equal(u2[j], u3[sub3.index(i)])
    xo, uo, _, _ = ob.oracle_solve_like(prob2, 1, iterations=3, robot=ROBOT)
    nn = st2[1].n_nodes
    assert rel_x(x2[1, :nn + 1], xo) < 1e-11 and rel_u(u2[1, :nn], uo) < 1e-11


@pytest.mark.parametrize("batch", [260, 530, 1100])
def test_g1_batches_larger_than_the_chip_match_oracle(ctx, batch):
    """More problems than CUs: the sweep changes kernel (four-wave workgroups with Gauss-Jordan up to two problems per CU, riccati_wave.h - a
    wavefront per problem, forward elimination and back substitution - up to four, riccati_wave2.h - two such waves per SIMD - beyond).  The 24-state robot is the hard case for both: its reduced
    Hessian has a condition number of 2.6e5 and gains of 6e3.  First and last problem of the batch against the oracle."""
    bp, sc, ob, itf = ctx["bp"], ctx["sc"], ctx["ob"], ctx["itf"]
    prob = sc.trot_problem(itf, batch=batch, n_intervals=45, gait=WALK)
    mpc = bp.BatchedSqpMpc(itf, max_batch=batch, max_nodes=72, sqp_iterations=2, return_gains=True)
    t, x, u, K, stats = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)
    n = stats[0].n_nodes
    assert all(s.status == 0 for s in stats)
    for b in (0, batch - 1):
        xo, uo, Ko, _ = ob.oracle_solve_like(prob, b, iterations=2, robot=ROBOT)
        assert rel_x(x[b, :n + 1], xo) < 1e-11 and rel_u(u[b, :n], uo) < 1e-11 and rel_K(K[b, :n], Ko) < 1e-10
