# Experiment: does replaying one solve as a hipGraph shorten the step (launch gaps)?  Captures reset + solve + export on the solver's stream.
import ctypes as C, time, torch, bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
hip = C.CDLL("libamdhip64.so")
itf = sc.h1_interface()
B, NI = 256, 100
prob = sc.trot_problem(itf, batch=B, n_intervals=NI)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=NI + 16, profile=False, stream=s.cuda_stream)
    mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    def step():
        mpc.reset(); mpc.enqueue()
    for _ in range(5): step()
    mpc.synchronize()
    t = time.perf_counter()
    for _ in range(50): step()
    mpc.synchronize(); plain = (time.perf_counter() - t) / 50
    stream = C.c_void_p(s.cuda_stream)
    graph, gexec = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamBeginCapture(stream, 0) == 0          # hipStreamCaptureModeGlobal
    step()
    rc = hip.hipStreamEndCapture(stream, C.byref(graph)); assert rc == 0, rc
    rc = hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0); assert rc == 0, rc
    for _ in range(5): assert hip.hipGraphLaunch(gexec, stream) == 0
    mpc.synchronize()
    t = time.perf_counter()
    for _ in range(50): hip.hipGraphLaunch(gexec, stream)
    mpc.synchronize(); graphed = (time.perf_counter() - t) / 50
    _, x, u, _, st = mpc.fetch()
    print("plain %.4f ms, graph %.4f ms per step; accepted %d" % (1e3 * plain, 1e3 * graphed, sum(1 for q in st if q.status == 0)))
