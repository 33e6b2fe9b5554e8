"""ORACLE (test infrastructure, not product code): ctypes binding of oracle/liboracle.so (see oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.cpp")
    cnt = os.path.join(_HERE, "liboracle_count.so")
    if force or not os.path.exists(so) or not os.path.exists(cnt) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def flop_counts(blob, x, u, mode=3):
    """Operation counts of the restatement at (x, u) (liboracle_count.so, a separate build with the counter compiled in):
    dict(flow_map, ee_kinematics, flow_map_ad, ee_kinematics_ad, node_linearization)."""
    so = os.path.join(_HERE, "liboracle_count.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle_count.so"])
    cl = C.CDLL(so)
    cl.oracle_model_create.restype = C.c_void_p
    cl.oracle_model_create.argtypes = [_dp, C.c_int]
    cl.oracle_model_destroy.argtypes = [C.c_void_p]
    blob, x, u = _arr(blob), _arr(x), _arr(u)
    h = cl.oracle_model_create(_d(blob), len(blob))
    out = np.zeros(5)
    rc = cl.oracle_flop_counts(C.c_void_p(h), _d(x), _d(u), int(mode), _d(out))
    cl.oracle_model_destroy(C.c_void_p(h))
    if rc != 0:
        raise RuntimeError("liboracle_count.so was built without ORACLE_COUNT_FLOPS")
    return dict(zip(("flow_map", "ee_kinematics", "flow_map_ad", "ee_kinematics_ad", "node_linearization"), out.tolist()))


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.oracle_model_create.restype = C.c_void_p
        _LIB.oracle_model_create.argtypes = [_dp, C.c_int]
        _LIB.oracle_model_destroy.argtypes = [C.c_void_p]
        _LIB.oracle_model_nx.argtypes = [C.c_void_p]
    return _LIB


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _arr(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleModel:
    def __init__(self, blob):
        blob = _arr(blob)
        self._h = lib().oracle_model_create(_d(blob), len(blob))
        if not self._h:
            raise ValueError("bad model blob")
        self.nx = lib().oracle_model_nx(C.c_void_p(self._h))
        self.nu = self.nx
        self.nj = self.nx - 12

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_model_destroy(C.c_void_p(self._h))
            self._h = None

    @property
    def h(self):
        return C.c_void_p(self._h)

    def flow_map(self, x, u, lin=False):
        x, u = _arr(x), _arr(u)
        f = np.zeros(self.nx)
        if not lin:
            lib().oracle_flow_map(self.h, _d(x), _d(u), _d(f), None, None)
            return f
        A = np.zeros((self.nx, self.nx))
        B = np.zeros((self.nx, self.nu))
        lib().oracle_flow_map(self.h, _d(x), _d(u), _d(f), _d(A), _d(B))
        return f, A, B

    def ee_kinematics(self, x, u, lin=False):
        x, u = _arr(x), _arr(u)
        pos, vel = np.zeros((4, 3)), np.zeros((4, 3))
        if not lin:
            lib().oracle_ee_kinematics(self.h, _d(x), _d(u), _d(pos), _d(vel), None, None, None)
            return pos, vel
        dpdx, dvdx, dvdu = np.zeros((12, self.nx)), np.zeros((12, self.nx)), np.zeros((12, self.nu))
        lib().oracle_ee_kinematics(self.h, _d(x), _d(u), _d(pos), _d(vel), _d(dpdx), _d(dvdx), _d(dvdu))
        return pos, vel, dpdx, dvdx, dvdu

    def cmm(self, q):
        q = _arr(q)
        A = np.zeros((6, 6 + self.nj))
        com = np.zeros(3)
        lib().oracle_cmm(self.h, _d(q), _d(A), _d(com))
        return A, com

    def node_lq(self, kind, dt, x, u, xnext, xref, mode, zref, zdref):
        nx, nu = self.nx, self.nu
        x, u, xnext, xref, zref, zdref = map(_arr, (x, u, xnext, xref, zref, zdref))
        o = dict(A=np.zeros((nx, nx)), B=np.zeros((nx, nu)), b=np.zeros(nx), Q=np.zeros((nx, nx)), R=np.zeros((nu, nu)),
                 P=np.zeros((nu, nx)), q=np.zeros(nx), r=np.zeros(nu), c=np.zeros(1), C=np.zeros((16, nx)), D=np.zeros((16, nu)),
                 e=np.zeros(16), perf=np.zeros(3))
        nc = C.c_int(0)
        lib().oracle_node_lq(self.h, int(kind), C.c_double(dt), _d(x), _d(u), _d(xnext), _d(xref), int(mode), _d(zref), _d(zdref),
                             _d(o["A"]), _d(o["B"]), _d(o["b"]), _d(o["Q"]), _d(o["R"]), _d(o["P"]), _d(o["q"]), _d(o["r"]), _d(o["c"]),
                             _d(o["C"]), _d(o["D"]), _d(o["e"]), C.byref(nc), _d(o["perf"]))
        o["nc"] = nc.value
        o["c"] = float(o["c"][0])
        return o

    def node_perf(self, kind, dt, x, u, xnext, xref, mode, zref, zdref):
        x, u, xnext, xref, zref, zdref = map(_arr, (x, u, xnext, xref, zref, zdref))
        perf = np.zeros(3)
        lib().oracle_node_perf(self.h, int(kind), C.c_double(dt), _d(x), _d(u), _d(xnext), _d(xref), int(mode), _d(zref), _d(zdref), _d(perf))
        return perf

    def _node_args(self, nodes):
        kind = np.ascontiguousarray(nodes["kind"], np.int32)
        mode = np.ascontiguousarray(nodes["mode"], np.int32)
        dt, zref, zdref, xref = map(_arr, (nodes["dt"], nodes["zref"], nodes["zdref"], nodes["xref"]))
        keep = (kind, mode, dt, zref, zdref, xref)
        return keep, (int(nodes["N"]), _i(kind), _d(dt), _i(mode), _d(zref), _d(zdref), _d(xref))

    def qp_step(self, nodes, x0, x, u, reg_prim=0.0):
        keep, args = self._node_args(nodes)
        N = int(nodes["N"])
        x0, x, u = _arr(x0), _arr(x), _arr(u)
        dx, du, K = np.zeros((N + 1, self.nx)), np.zeros((N, self.nu)), np.zeros((N, self.nu, self.nx))
        rc = lib().oracle_qp_step_reg(self.h, *args, _d(x0), _d(x), _d(u), C.c_double(reg_prim), _d(dx), _d(du), _d(K))
        if rc != 0:
            raise RuntimeError("oracle_qp_step failed")
        return dx, du, K

    def solve(self, nodes, x0, x_init, u_init, iterations=1, g_max=1e-2, g_min=1e-6, alpha_decay=0.5, alpha_min=1e-4, gamma_c=1e-6,
              armijo_factor=1e-4, delta_tol=1e-4, reg_prim=0.0):
        keep, args = self._node_args(nodes)
        N = int(nodes["N"])
        x0, x_init, u_init = _arr(x0), _arr(x_init), _arr(u_init)
        opts = np.array([iterations, g_max, g_min, alpha_decay, alpha_min, gamma_c, armijo_factor, delta_tol, reg_prim], float)
        xo, uo, K = np.zeros((N + 1, self.nx)), np.zeros((N, self.nu)), np.zeros((N, self.nu, self.nx))
        stats = np.zeros((iterations, 16))
        rc = lib().oracle_solve(self.h, *args, _d(x0), _d(x_init), _d(u_init), _d(opts), _d(xo), _d(uo), _d(K), _d(stats))
        if rc != 0:
            raise RuntimeError("oracle_solve failed")
        return xo, uo, K, stats


def lu_projection(Cm, D, e):
    Cm, D, e = _arr(Cm), _arr(D), _arr(e)
    nc, nx = Cm.shape
    nu = D.shape[1]
    Px, Pu, Pe = np.zeros((nu, nx)), np.zeros((nu, nu)), np.zeros(nu)
    rank = C.c_int(0)
    lib().oracle_lu_projection(nc, nx, nu, _d(Cm), _d(D), _d(e), _d(Px), _d(Pu), _d(Pe), C.byref(rank))
    return Px, Pu[:, :nu - rank.value].copy(), Pe, rank.value
