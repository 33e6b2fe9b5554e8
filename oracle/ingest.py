"""ORACLE (test infrastructure, not product code): independent Python ingest of the robot description.

Restates, in numpy, how the reference turns its data files into model constants:

  * INFO files: boost::property_tree INFO semantics as consumed through OCS2 `loadData::*`
    [OCS2-upstream; call sites ocs2_bipedal_robot/src/BipedalRobotInterface.cpp:92-108,239-291,
    src/common/ModelSettings.cpp:40-67, src/gait/ModeSequenceTemplate.cpp:50-111].
  * URDF -> kinematic tree with the conventions of `centroidal_model::createPinocchioInterface`
    (call site BipedalRobotInterface.cpp:117) [OCS2-upstream]: joints not in `jointNames` are welded
    at zero and their link inertias merged into the parent body; floating base = translation + ZYX Euler;
    actuated joints ordered depth-first with children visited in joint-name order.
  * Input-cost matrix R (BipedalRobotInterface.cpp:239-271) and Q (:276-278).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module.
Parity status: UNPINNED (the reference holds no golden vectors for this path); the
values are pinned by hand-derived known answers in tests/ instead (total mass 51.641 kg, ...).
"""
import math
import xml.etree.ElementTree as ET

import numpy as np

# ----------------------------------------------------------------------------------------------
# INFO parser (boost property_tree info_parser semantics: per line alternate key / data tokens,
# ';' starts a comment, braces nest, "quoted strings" may hold spaces; text after the data token on a
# line, e.g. '// remark', becomes further harmless key/data pairs exactly as in boost).
# ----------------------------------------------------------------------------------------------


def _tokenize(line):
    toks = []
    i, n = 0, len(line)
    while i < n:
        ch = line[i]
        if ch in " \t\r\n":
            i += 1
        elif ch == ";":
            break
        elif ch == '"':
            j = i + 1
            buf = []
            while j < n and line[j] != '"':
                if line[j] == "\\" and j + 1 < n:
                    j += 1
                buf.append(line[j])
                j += 1
            toks.append(("s", "".join(buf)))
            i = j + 1
        elif ch in "{}":
            toks.append((ch, ch))
            i += 1
        else:
            j = i
            while j < n and line[j] not in " \t\r\n;":
                j += 1
            toks.append(("w", line[i:j]))
            i = j
    return toks


def parse_info(path, raw=False):
    """Returns the tree as a list of (key, (data, children)) pairs, order and duplicates preserved."""
    root = []
    stack = [root]
    last = None  # the [data, children] cell of the most recent key
    with open(path) as f:
        for line in f:
            expect_data = False
            for kind, tok in _tokenize(line):
                if kind == "{":
                    if last is None:
                        raise ValueError("unexpected { in " + path)
                    stack.append(last[1])
                    last = None
                    expect_data = False
                elif kind == "}":
                    if len(stack) == 1:
                        raise ValueError("unbalanced } in " + path)
                    stack.pop()
                    last = None
                    expect_data = False
                elif expect_data:
                    last[0] = tok
                    expect_data = False
                else:
                    last = ["", []]
                    stack[-1].append((tok, last))
                    expect_data = True
    if len(stack) != 1:
        raise ValueError("unbalanced { in " + path)

    def freeze(node):
        return [(k, (cell[0], freeze(cell[1]))) for k, cell in node]

    return freeze(root)


def info_get(tree, dotted, default=None):
    """ptree.get<T>(path): first match at every level; '.' separates levels."""
    node = tree
    val = None
    for part in dotted.split("."):
        for k, (v, children) in node:
            if k == part:
                val, node = v, children
                break
        else:
            return default
    return val


def info_child(tree, dotted):
    node = tree
    for part in dotted.split("."):
        for k, (v, children) in node:
            if k == part:
                node = children
                break
        else:
            return None
    return node


def load_matrix(tree, name, rows, cols):
    """loadData::loadEigenMatrix [OCS2-upstream]: entries '(i,j)', optional 'scaling' and 'default'."""
    scaling = float(info_get(tree, name + ".scaling", 1.0))
    default = float(info_get(tree, name + ".default", 0.0))
    M = np.zeros((rows, cols))
    failed = 0
    for i in range(rows):
        for j in range(cols):
            v = info_get(tree, "%s.(%d,%d)" % (name, i, j))
            if v is None:
                failed += 1
                aij = default
            else:
                aij = float(v)
            M[i, j] = scaling * aij
    if failed == rows * cols:
        raise ValueError("could not load matrix " + name)
    return M


def load_std_vector(tree, name, conv=float):
    """loadData::loadStdVector [OCS2-upstream]: keys '[0]', '[1]', ... until the first missing one."""
    out = []
    while True:
        v = info_get(tree, "%s.[%d]" % (name, len(out)))
        if v is None:
            break
        out.append(conv(v))
    return out


# ----------------------------------------------------------------------------------------------
# small rotation helpers
# ----------------------------------------------------------------------------------------------


def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], float)


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], float)


def rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], float)


def rpy_to_rot(rpy):
    """URDF fixed-axis roll-pitch-yaw: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    return rot_z(rpy[2]) @ rot_y(rpy[1]) @ rot_x(rpy[0])


def axis_angle(a, th):
    a = np.asarray(a, float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) * math.cos(th) + (1 - math.cos(th)) * np.outer(a, a) + math.sin(th) * K


def _vec(s, n=3):
    v = [float(t) for t in s.split()]
    assert len(v) == n
    return np.array(v)


# ----------------------------------------------------------------------------------------------
# URDF -> tree
# ----------------------------------------------------------------------------------------------


def parse_urdf(path):
    robot = ET.parse(path).getroot()
    links, joints = {}, {}
    for el in robot:
        if el.tag == "link":
            m, c, I = 0.0, np.zeros(3), np.zeros((3, 3))
            ine = el.find("inertial")
            if ine is not None:
                m = float(ine.find("mass").get("value"))
                org = ine.find("origin")
                Rin = np.eye(3)
                if org is not None:
                    c = _vec(org.get("xyz", "0 0 0"))
                    Rin = rpy_to_rot(_vec(org.get("rpy", "0 0 0")))
                it = ine.find("inertia")
                g = lambda k: float(it.get(k, "0"))  # noqa: E731
                I0 = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
                I = Rin @ I0 @ Rin.T
            links[el.get("name")] = (m, c, I)
        elif el.tag == "joint":
            org = el.find("origin")
            xyz = _vec(org.get("xyz", "0 0 0")) if org is not None else np.zeros(3)
            rpy = _vec(org.get("rpy", "0 0 0")) if org is not None else np.zeros(3)
            ax = el.find("axis")
            axis = _vec(ax.get("xyz")) if ax is not None else np.array([1.0, 0, 0])
            joints[el.get("name")] = dict(type=el.get("type"), parent=el.find("parent").get("link"),
                                          child=el.find("child").get("link"), xyz=xyz, rpy=rpy, axis=axis)
    return links, joints


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def build_tree(links, joints, joint_names, contact_names):
    children = {}
    is_child = set()
    for jn, j in joints.items():
        children.setdefault(j["parent"], []).append(jn)
        is_child.add(j["child"])
    roots = [l for l in links if l not in is_child]
    assert len(roots) == 1, roots
    parent, Rfix, pfix, axis, names = [], [], [], [], []
    acc = [dict(m=0.0, mc=np.zeros(3), Io=np.zeros((3, 3)))]
    frames = {}

    def add_inertia(body, R, p, link):
        m, c, I = links[link]
        cw = p + R @ c
        acc[body]["m"] += m
        acc[body]["mc"] += m * cw
        acc[body]["Io"] += R @ I @ R.T + m * (cw @ cw * np.eye(3) - np.outer(cw, cw))

    def visit(link, body, R, p):
        add_inertia(body, R, p, link)
        frames[link] = (body, R.copy(), p.copy())
        for jn in sorted(children.get(link, [])):
            j = joints[jn]
            Rj = R @ rpy_to_rot(j["rpy"])
            pj = p + R @ j["xyz"]
            if jn in joint_names and j["type"] in ("revolute", "continuous"):
                a = j["axis"] / np.linalg.norm(j["axis"])
                parent.append(body)
                Rfix.append(Rj)
                pfix.append(pj)
                axis.append(a)
                names.append(jn)
                acc.append(dict(m=0.0, mc=np.zeros(3), Io=np.zeros((3, 3))))
                visit(j["child"], len(acc) - 1, np.eye(3), np.zeros(3))
            else:
                if jn in joint_names:
                    raise ValueError("actuated joint %s has unsupported type %s" % (jn, j["type"]))
                visit(j["child"], body, Rj, pj)

    visit(roots[0], 0, np.eye(3), np.zeros(3))
    nj = len(parent)
    mass = np.array([a["m"] for a in acc])
    com = np.zeros((nj + 1, 3))
    inertia = np.zeros((nj + 1, 3, 3))
    for b, a in enumerate(acc):
        if a["m"] > 0:
            c = a["mc"] / a["m"]
            com[b] = c
            inertia[b] = a["Io"] - a["m"] * (c @ c * np.eye(3) - np.outer(c, c))
    cbody = [frames[n][0] for n in contact_names]
    coff = np.array([frames[n][2] for n in contact_names])
    return dict(nj=nj, parent=np.array(parent, int), Rfix=np.array(Rfix), pfix=np.array(pfix), axis=np.array(axis),
                joint_names=names, mass=mass, com=com, inertia=inertia, contact_body=np.array(cbody, int), contact_off=coff)


# ----------------------------------------------------------------------------------------------
# numpy kinematics (used for the R matrix and as a third, slow implementation in tests)
# ----------------------------------------------------------------------------------------------


def fk(model, q):
    """q = [p(3), yaw, pitch, roll, joints]. Returns per-body (R, o) world placements (body 0 = base)."""
    nj = model["nj"]
    R = [rot_z(q[3]) @ rot_y(q[4]) @ rot_x(q[5])]
    o = [np.array(q[0:3], float)]
    for j in range(nj):
        lam = model["parent"][j]
        R.append(R[lam] @ model["Rfix"][j] @ axis_angle(model["axis"][j], q[6 + j]))
        o.append(o[lam] + R[lam] @ model["pfix"][j])
    return R, o


def contact_positions(model, q):
    R, o = fk(model, q)
    return np.array([o[b] + R[b] @ model["contact_off"][i] for i, b in enumerate(model["contact_body"])])


def is_ancestor_or_self(model, j_body, body):
    """True if movable body j_body (1..nj) lies on the path from the base to `body`."""
    b = body
    while b != 0:
        if b == j_body:
            return True
        b = model["parent"][b - 1]
    return False


def contact_jacobian_joints(model, q):
    """3 x nj blocks: d(contact position)/d(leg joints), world aligned (LOCAL_WORLD_ALIGNED linear part)."""
    nj = model["nj"]
    R, o = fk(model, q)
    J = np.zeros((len(model["contact_body"]) * 3, nj))
    for i, b in enumerate(model["contact_body"]):
        p = o[b] + R[b] @ model["contact_off"][i]
        for j in range(nj):
            if is_ancestor_or_self(model, j + 1, b):
                a = R[j + 1] @ model["axis"][j]
                J[3 * i:3 * i + 3, j] = np.cross(a, p - o[j + 1])
    return J


# ----------------------------------------------------------------------------------------------
# full model
# ----------------------------------------------------------------------------------------------


def build_model(urdf_path, task_path, reference_path, use_hard_friction_cone=False):
    task = parse_info(task_path)
    ref = parse_info(reference_path)
    joint_names = load_std_vector(task, "model_settings.jointNames", str)
    contact_names = load_std_vector(task, "model_settings.contactNames3DoF", str)
    assert len(contact_names) == 4 and not load_std_vector(task, "model_settings.contactNames6DoF", str)
    links, joints = parse_urdf(urdf_path)
    m = build_tree(links, joints, set(joint_names), contact_names)
    nj = m["nj"]
    assert nj == len(joint_names)
    nx = nu = 12 + nj
    m["nx"], m["nu"] = nx, nu
    m["robot_mass"] = float(sum(l[0] for l in links.values()))
    m["contact_names"] = contact_names
    m["initial_state"] = load_matrix(task, "initialState", nx, 1)[:, 0]
    m["default_joint_state"] = load_matrix(ref, "defaultJointState", nj, 1)[:, 0]
    m["com_height"] = float(info_get(ref, "comHeight"))
    m["target_displacement_velocity"] = float(info_get(ref, "targetDisplacementVelocity"))
    m["target_rotation_velocity"] = float(info_get(ref, "targetRotationVelocity"))
    m["Q"] = load_matrix(task, "Q", nx, nx)
    # R: BipedalRobotInterface.cpp:239-271
    Rt = load_matrix(task, "R", 24, 24)
    J = contact_jacobian_joints(m, m["initial_state"][6:])
    R = np.zeros((nu, nu))
    R[:12, :12] = Rt[:12, :12]
    R[12:, 12:] = J.T @ Rt[12:, 12:] @ J
    m["R"] = R
    m["friction_coefficient"] = float(info_get(task, "frictionConeSoftConstraint.frictionCoefficient"))
    m["barrier_mu"] = float(info_get(task, "frictionConeSoftConstraint.mu"))
    m["barrier_delta"] = float(info_get(task, "frictionConeSoftConstraint.delta"))
    # FrictionConeConstraint::Config defaults (include/.../constraint/FrictionConeConstraint.h:66-67)
    m["cone_regularization"] = 25.0
    m["cone_gripper_force"] = 0.0
    m["cone_hessian_shift"] = 1e-6
    # BipedalRobotInterface.cpp:68-69,181-182: with useHardFrictionConeConstraint the cone is an inequality constraint of the problem and the
    # SQP solver penalises it with sqp.inequalityConstraintMu / Delta (task.info:74-75; [OCS2-upstream] defaults 0 / 1e-6)
    m["hard_friction_cone"] = bool(use_hard_friction_cone)
    m["ineq_mu"] = float(info_get(task, "sqp.inequalityConstraintMu", 0.0))
    m["ineq_delta"] = float(info_get(task, "sqp.inequalityConstraintDelta", 1e-6))
    m["position_error_gain"] = float(info_get(task, "model_settings.positionErrorGain"))
    m["phase_transition_stance_time"] = float(info_get(task, "model_settings.phaseTransitionStanceTime"))
    m["swing"] = {k: float(info_get(task, "swing_trajectory_config." + k))
                  for k in ("liftOffVelocity", "touchDownVelocity", "swingHeight", "swingTimeScale")}
    # ddp block (task.info:115-156): what the ILQR restatement of oracle/ddp_py.py reads
    m["ddp"] = dict(algorithm=str(info_get(task, "ddp.algorithm", "SLQ")), maxNumIterations=int(info_get(task, "ddp.maxNumIterations", 15)),
                    timeStep=float(info_get(task, "ddp.timeStep", 0.01)), useFeedbackPolicy=str(info_get(task, "ddp.useFeedbackPolicy", "true")).lower() in ("true", "1"),
                    strategy=str(info_get(task, "ddp.strategy", "LINE_SEARCH")),
                    minStepLength=float(info_get(task, "ddp.lineSearch.minStepLength", 0.05)), maxStepLength=float(info_get(task, "ddp.lineSearch.maxStepLength", 1.0)),
                    hessianCorrectionStrategy=str(info_get(task, "ddp.lineSearch.hessianCorrectionStrategy", "DIAGONAL_SHIFT")),
                    hessianCorrectionMultiple=float(info_get(task, "ddp.lineSearch.hessianCorrectionMultiple", 1e-6)))
    m["rollout"] = dict(AbsTolODE=float(info_get(task, "rollout.AbsTolODE", 1e-5)), RelTolODE=float(info_get(task, "rollout.RelTolODE", 1e-3)),
                        timeStep=float(info_get(task, "rollout.timeStep", 0.015)), maxNumStepsPerSecond=int(info_get(task, "rollout.maxNumStepsPerSecond", 10000)))
    m["sqp"] = dict(dt=float(info_get(task, "sqp.dt")), sqpIteration=int(info_get(task, "sqp.sqpIteration")),
                    deltaTol=float(info_get(task, "sqp.deltaTol")), g_max=float(info_get(task, "sqp.g_max")),
                    g_min=float(info_get(task, "sqp.g_min")))
    m["mpc"] = dict(timeHorizon=float(info_get(task, "mpc.timeHorizon")))
    m["initial_mode_schedule"] = (load_std_vector(ref, "initialModeSchedule.eventTimes"),
                                  [MODE_NUMBER[s] for s in load_std_vector(ref, "initialModeSchedule.modeSequence", str)])
    m["default_template"] = (load_std_vector(ref, "defaultModeSequenceTemplate.switchingTimes"),
                             [MODE_NUMBER[s] for s in load_std_vector(ref, "defaultModeSequenceTemplate.modeSequence", str)])
    return m


# include/ocs2_bipedal_robot/gait/MotionPhaseDefinition.h:47-52,101-110
MODE_NUMBER = {"FLY": 0, "LF": 1, "RF": 2, "STANCE": 3}


def load_gait_template(gait_path, name):
    """loadModeSequenceTemplate (src/gait/ModeSequenceTemplate.cpp:50-71)."""
    g = parse_info(gait_path)
    times = load_std_vector(g, name + ".switchingTimes")
    modes = [MODE_NUMBER[s] for s in load_std_vector(g, name + ".modeSequence", str)]
    if not times or not modes:
        raise ValueError("failed to load gait " + name)
    return times, modes


def model_blob(m):
    """Flat double array handed to the C++ oracle (layout documented in oracle/oracle.h)."""
    nj = m["nj"]
    parts = [[nj], m["parent"], m["Rfix"].reshape(-1), m["pfix"].reshape(-1), m["axis"].reshape(-1), m["mass"], m["com"].reshape(-1),
             m["inertia"].reshape(-1), m["contact_body"], m["contact_off"].reshape(-1), m["Q"].reshape(-1), m["R"].reshape(-1),
             [m["friction_coefficient"], m["cone_regularization"], m["cone_gripper_force"], m["cone_hessian_shift"],
              m["barrier_mu"], m["barrier_delta"], m["position_error_gain"], m["robot_mass"]]]
    if m.get("hard_friction_cone"):
        parts.append([1.0, m["ineq_mu"], m["ineq_delta"]])
    return np.concatenate([np.asarray(p, float).reshape(-1) for p in parts])
