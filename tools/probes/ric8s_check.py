# bitwise comparison of the two eight-wave sweeps (riccati_mfma8.h / riccati_mfma8s.h): dx, du, K, Acl, summary after one sweep of the bench workload
import os, sys, subprocess, numpy as np
def run(tag, gait_start):
    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios
    itf = scenarios.h1_interface()
    out = {}
    for name, kw in (("trot", dict(gait="trot")), ("stand", dict(gait="stance")), ("strot", dict(gait="standing_trot"))):
        try:
            prob = scenarios.trot_problem(itf, batch=64, n_intervals=100, **kw)
        except TypeError:
            prob = scenarios.trot_problem(itf, batch=64, n_intervals=100)
        mpc = bp.BatchedSqpMpc(itf, 64, 116)
        mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
        for st in ("linearize", "project", "riccati"): mpc.stage(st)
        mpc.synchronize()
        for q in ("dx", "du", "K", "summary"):
            out[name + "_" + q] = mpc.read(q).copy()
        for chunks in (3,):
            pass
    np.savez("/tmp/ric8s_%s.npz" % tag, **out)
if len(sys.argv) > 1:
    run(sys.argv[1], 0); sys.exit(0)
for tag, env in (("old", "0"), ("new", "1")):
    e = dict(os.environ, BPMPC_RICCATI8_S=env, PYTHONPATH=".")
    subprocess.check_call([sys.executable, __file__, tag], env=e)
a, b = np.load("/tmp/ric8s_old.npz"), np.load("/tmp/ric8s_new.npz")
for k in a.files:
    d = np.abs(a[k] - b[k]); print(k, a[k].shape, "max abs diff", np.nanmax(d), "bitwise", np.array_equal(a[k], b[k]), "nan", np.isnan(b[k]).sum(), "absmax", np.nanmax(np.abs(a[k])))
