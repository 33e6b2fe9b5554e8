// Fast node linearisation (HIP only; the lane-emulated reference with identical mathematics is node_lq.h +
// centroidal_eval.h).
//
// Several nodes per wavefront: a node owns LPN = 16 lanes (nx = 22) or 32 lanes (nx = 24), one lane per generalised
// coordinate g (0-2 base translation, 3-5 yaw/pitch/roll, 6.. leg joints; lane g >= 5 also owns body g - 5).  All
// per-column quantities (world axis, composite inertia, momentum-matrix column, twist, d(Av)/dq column, contact Jacobian
// columns, the columns of df/dx and df/du) live in that lane's registers, so one instruction serves every node of the
// wave.  Cross-lane traffic inside a node is limited to: the Euler sin/cos (shuffles), chain walks and subtree sums over
// small LDS tables, three 16-lane DPP all-reduces, and the 9 x 12 block of the second RK2 stage.
// Each lane plays up to four column roles when the dense LQ model is written:
//   x column 6+g (q_g, every lane), x column g (momentum, lanes 0..5), u column g (force, lanes 0..11),
//   u column 12+(g-6) (joint velocity, lanes 6..).
#pragma once
#include <hip/hip_runtime.h>

#include "node_lq.h"
#include "riccati_fast.h"   // lds_wave_sync

namespace bpmpc {

// PACK: when the robot has more than 16 generalised coordinates but no more than 16 NON-TRIVIAL ones (12 leg joints: 3 Euler angles +
// 12 joints = 15), the three base translations give up their lanes: every column they own is a constant (nothing in the dynamics depends
// on the base position; their columns of df/dx are zero, of the contact Jacobians the identity), so lane l carries coordinate l + 3 and
// the lanes 0..2 write those constant columns next to their other roles.  A node then fits 16 lanes and a wavefront serves four nodes
// instead of two.  The roles that are numbered by LANE (momentum column l < 6, force column l < 12, contact l < 4) do not move.
// CHAIN: the robot is two serial legs of NJ / 2 joints whose joints sit in consecutive lanes in chain order (DeviceModel::serial_legs; every
// robot of the reference is).  The tree walks of an evaluation - chain composition, joint twists, subtree sums of composites and momenta -
// then run through DPP row shifts between neighbouring lanes (a joint's parent is the lane before it, its child the lane behind it): no LDS
// table, no memory round trip per level (a wave of this kernel lived mostly in the ~40 dependent LDS round trips of those walks per
// evaluation).  The arithmetic of the joint lanes is the table walk's operation for operation; the whole-robot sums associate differently.
template <int NJ, bool PACK = false, bool CHAIN_ = false>
struct LinFastCfg {
  static constexpr bool CHAIN = CHAIN_;
  static constexpr int LEG = NJ / 2;                 // joints per leg (CHAIN)
  static constexpr int NB = NJ + 1, G = 6 + NJ, NX = 12 + NJ, NU = 12 + NJ;
  static constexpr int G0 = (PACK && G > 16 && G - 3 <= 16) ? 3 : 0;   // coordinate of lane 0
  static constexpr int LPN = (G - G0 <= 16) ? 16 : 32;   // lanes per node
  static constexpr int NPW = kWave / LPN;           // nodes per wavefront
};

// Model constants that the lanes index by their own coordinate, staged once per workgroup.  Reading them from global
// memory inside the node would put vector loads behind the node's output stores (vmcnt retires in order on this
// architecture), i.e. every such load would wait for the stores in flight.
// The model's SCALAR constants (same field names as DeviceModel, so that cone_terms / nominal_input of node_lq.h take either).  Round 6: read
// from the model in global memory inside the node they were VECTOR loads (the compiler cannot prove that the kernel's own stores leave the
// model alone, so no scalar load), each followed by s_waitcnt vmcnt(0): a memory round trip of ~1 .. 2 us on a loaded chip per use - eight in
// a row for the contact points of an evaluation (contact_body -> contact_off, dependent) - and, behind the first output row, a wait for every
// store in flight (vmcnt retires in order).  From LDS they are broadcast reads on a counter of their own.
struct LinFastScalars {
  double contact_off[kNumContacts][3];
  double friction, cone_reg, cone_grip, cone_shift, barrier_mu, barrier_delta, pos_gain, robot_mass;
  unsigned contact_path[kNumContacts];
  int contact_body[kNumContacts];
  int max_depth, cone_gauss_newton;
};
// Layout: the scalars and the kinematic constants first, the cost weights LAST - the block of the value-only kernels (COST = false) is a prefix of the
// full one, so both are staged from one image by a flat copy.
template <int NJ, bool COST = true>   // COST = false: without the cost weights (the value-only kernels read them from global memory)
struct alignas(16) LinFastShared {
  using C = LinFastCfg<NJ>;
  LinFastScalars sc;
  double Rfix[C::NB][9], pfix[C::NB][3], axis[C::NB][3], com[C::NB][3], inertia[C::NB][6], mass[C::NB];
  int depth[C::NB];
  unsigned subtree[C::NB];
  int path[C::NB][NJ];
  alignas(16) double Q[COST ? C::NX * C::NX : 2], R[COST ? C::NU * C::NU : 2];
};
// The image of LinFastShared<NJ, true> lies behind the DeviceModel in the same device allocation (solver.hip uploads both).  Round 6: the block used
// to be gathered field by field from the DeviceModel - seven loops, each compiled to load -> s_waitcnt vmcnt(0) -> LDS store, i.e. seven
// dependent memory round trips (5 .. 6 us of a workgroup's 40 us, tools/lin_timeline.py) before the workgroup's barrier.  Now every thread has
// its 16-byte pieces of the image in flight together with its node's inputs and waits once.
constexpr size_t kLinImageOffset = (sizeof(DeviceModel) + 255) / 256 * 256;
template <int NJ>
inline void fill_shared_image(const DeviceModel& md, LinFastShared<NJ, true>& sh) {      // host
  using C = LinFastCfg<NJ>;
  for (int i = 0; i < C::NX * C::NX; ++i) { sh.Q[i] = md.Q[i]; sh.R[i] = md.R[i]; }
  for (int b = 0; b < C::NB; ++b) {
    for (int i = 0; i < 9; ++i) sh.Rfix[b][i] = md.Rfix[b][i];
    for (int i = 0; i < 6; ++i) sh.inertia[b][i] = md.inertia[b][i];
    for (int i = 0; i < 3; ++i) { sh.pfix[b][i] = md.pfix[b][i]; sh.axis[b][i] = md.axis[b][i]; sh.com[b][i] = md.com[b][i]; }
    sh.mass[b] = md.mass[b]; sh.depth[b] = md.depth[b]; sh.subtree[b] = md.subtree[b];
    for (int i = 0; i < NJ; ++i) sh.path[b][i] = md.path[b][i];
  }
  for (int c = 0; c < kNumContacts; ++c) {
    for (int i = 0; i < 3; ++i) sh.sc.contact_off[c][i] = md.contact_off[c][i];
    sh.sc.contact_path[c] = md.contact_path[c]; sh.sc.contact_body[c] = md.contact_body[c];
  }
  sh.sc.friction = md.friction; sh.sc.cone_reg = md.cone_reg; sh.sc.cone_grip = md.cone_grip; sh.sc.cone_shift = md.cone_shift;
  sh.sc.barrier_mu = md.barrier_mu; sh.sc.barrier_delta = md.barrier_delta; sh.sc.pos_gain = md.pos_gain; sh.sc.robot_mass = md.robot_mass;
  sh.sc.max_depth = md.max_depth; sh.sc.cone_gauss_newton = md.cone_gauss_newton;
}
template <int NT, int NJ, bool COST>    // NT: threads of the workgroup, all of which call
__device__ __forceinline__ void load_shared_model(const DeviceModel& md, LinFastShared<NJ, COST>& sh, int tid) {
  static_assert(sizeof(LinFastShared<NJ, COST>) % 16 == 0 && sizeof(LinFastShared<NJ, COST>) <= sizeof(LinFastShared<NJ, true>), "flat 16-byte copy of a prefix");
  constexpr int N16 = sizeof(LinFastShared<NJ, COST>) / 16, K = (N16 + NT - 1) / NT;
  const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(&md) + kLinImageOffset);
  uint4* dst = reinterpret_cast<uint4*>(&sh);
  uint4 v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { const int i = tid + k * NT; v[k] = src[i < N16 ? i : 0]; }      // (unconditional loads: nothing to wait for in between)
#pragma unroll
  for (int k = 0; k < K; ++k) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y), "+v"(v[k].z), "+v"(v[k].w));   // all of them in flight before the first store
                                                                                                             // (else each load sinks into its store's branch)
#pragma unroll
  for (int k = 0; k < K; ++k) { const int i = tid + k * NT; if (i < N16) dst[i] = v[k]; }
}

// Per-node LDS tables.  FULL = false: value-only evaluation (line search trials, policy rollout): no derivative tables.
// FULL = true (the lineariser): 2.6 KB per node at nx = 22, so that three four-wave workgroups (48 nodes) fit the 160 KB of a CU next to
// their three copies of the model block.  Tables that are never alive together share storage; everything is accessed by one wavefront
// in program order (lds_wave_sync between a write and a cross-lane read):
//   tab:   chain rotations T (eval_lane, dead after the walks) -> per-body composites -> per-body momenta (both inside eval_lane)
//          -> contact velocities (between the first evaluation and the constraint rows) -> ... second evaluation ... -> the stage-two
//          block a2 -> after the rows of A and B the cost vectors dx, du
//   swing references (read by the constraint rows, BEFORE the second evaluation) / stage-two momentum and base position xh2
//   (the cone terms of the cost phase lie beside dx, du in tab)
// Node-level results that every lane of the node computes identically (flow-map rows 0..5, base velocity, Euler sines / cosines) are
// parked here by one lane and read back where they are used instead of occupying registers of all lanes across the derivative phases.
template <int NJ, bool FULL = true, bool CHAIN = false, bool PARK = false>
struct LinFastNodeLds {
  static constexpr bool kPark = PARK;   // the stage-one Jacobian columns wait in LDS (when two workgroups per CU still fit) instead of in HBM scratch
  using C = LinFastCfg<NJ>;
  static constexpr int NT = CHAIN ? 1 : NJ, NBT = CHAIN ? 1 : C::NB, NW = CHAIN ? 3 : C::G - 3;   // CHAIN: no walk tables, only the Euler rates
  static constexpr bool kFull = FULL;
  // node inputs, staged once so that nothing is loaded from global memory after the first output store
  double x[C::NX], u[C::NU];
  union {
    double xh2[9];               // normalised momentum and base position of the second RK2 stage (the first stage reads x[0..8])
    struct { double zref[kNumContacts], zdref[kNumContacts]; };   // swing references of the node record (dead before the second stage starts)
  };
  union {
    double T[NT][9];             // joint-local rotation E of joint g-6 (its fixed offset is a model constant: LinFastShared::pfix)
    double a2[FULL ? 9 : 1][12]; // rows 3..11, x columns 0..11 of the stage-two Jacobian
    double comp[NBT][10];        // per body mass / first moment / inertia about o0
    double hb[NBT][6];           // per body momentum about o0
    double cvel_full[FULL ? kNumContacts : 1][3];   // contact point velocities of the first stage (FULL)
    // cost phase (both evaluations and the combination are done by then): the cost vectors and, beside them, the cone terms
    // (value-only: the barrier value stays in the lane that computes it)
    struct { double dx[C::NX], du[C::NU], cone[FULL ? kNumContacts : 1][FULL ? 13 : 1]; };
  };
  double og[NW][3], wv[NW][3];   // joint origins (coordinates 3..), a_g * v_g (twist walk of an evaluation; CHAIN: the Euler rates only)
  double cpos_v[FULL ? 1 : kNumContacts][3], cvel_v[FULL ? 1 : kNumContacts][3];     // value-only: contact points and velocities of the current evaluation
  // node-level results of the two stages: A_b^{-1} blocks, contact points, com, flow-map rows 0..5, base linear velocity, Euler sin / cos
  double X12[FULL ? 2 : 1][FULL ? 9 : 1], X22[FULL ? 2 : 1][FULL ? 9 : 1], cps[FULL ? 2 : 1][FULL ? kNumContacts : 1][3], com[FULL ? 2 : 1][3];
  double fh[FULL ? 2 : 1][FULL ? 6 : 1], vlin[FULL ? 2 : 1][FULL ? 3 : 1], trig[FULL ? 4 : 1];
  double park[(FULL && PARK) ? 16 : 1][(FULL && PARK) ? 16 : 1];   // [row][lane]: rows 3..11 of x column 6+g, rows 6..11 of the joint-velocity column, f[6+g] of the first stage
  // value-only evaluation at nx = 24: the lane's entries of x_next and x_ref, known at the top and used at the bottom, wait here - at three waves per SIMD
  // (168 registers) the allocator put exactly these six values into scratch memory
  static constexpr bool kLate = !FULL && NJ > 10;
  double late[kLate ? 6 : 1][kLate ? 16 : 1];
  // contact points / velocities of an evaluation (FULL: the stage's own copy, kept for the RK2 combination)
  __device__ __forceinline__ double (*cpos(int stage))[3] { if constexpr (FULL) return cps[stage]; else return cpos_v; }
  __device__ __forceinline__ double (*cvel())[3] { if constexpr (FULL) return cvel_full; else return cvel_v; }
};

// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ double row16_allreduce_add(double x) {
#define BP_ROR_ADD(n)                                                                                           \
  {                                                                                                             \
    const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x120 + n, 0xf, 0xf, false);              \
    const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x120 + n, 0xf, 0xf, false);              \
    x += __hiloint2double(hi_, lo_);                                                                            \
  }
  BP_ROR_ADD(1) BP_ROR_ADD(2) BP_ROR_ADD(4) BP_ROR_ADD(8)
#undef BP_ROR_ADD
  return x;
}
template <int LPN>
__device__ __forceinline__ double node_allreduce_add(double x) {
  x = row16_allreduce_add(x);
  if (LPN == 32) x += __shfl_xor(x, 16);
  return x;
}

// x of the lane CTRL points at inside this lane's 16-lane DPP row (0.0 where that lane lies outside the row): row_shr:n = lane - n, row_shl:n = lane + n.
// Must run with every lane of the row active (a disabled source lane does not deliver).
constexpr int kDppRowShl = 0x100, kDppRowShr = 0x110;
template <int CTRL>
__device__ __forceinline__ double dpp_row_f64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// v[g] for g < 6 (0 otherwise) without dynamic register indexing (which would push the array into scratch memory)
__device__ __forceinline__ double lane_pick6(const double (&v)[6], int g) {
  double out = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double vi = v[i];
    asm volatile("" : "+v"(vi));
    out = (g == i) ? vi : out;
  }
  return out;
}

// per-lane tree bookkeeping of the body this lane owns (the inertial constants are read from the model at their
// single point of use to keep register live ranges short)
struct LaneBody {
  int depth, body;
  unsigned subtree;              // members of the subtree this lane's coordinate moves (bodies)
};

// result of one evaluation of the centroidal dynamics in this lane (the momentum and force columns are rebuilt from the
// node-level quantities kept in LDS: X12/X22 and the contact points)
struct LaneEval {
  double ar_q[9];   // rows 3..11 of column 6+g of df/dx
  double br_j[6];   // rows 6..11 of column 12+(g-6) of df/du (lanes 6..); its rows 3..5 are zero
  double fh[6];     // rows 0..5 of f (identical in all lanes)
  double vg;        // f[6+g]
};

// rows 3..11 of the momentum column g (lanes 0..5) of df/dx: [0; m A_b^{-1}]
__device__ __forceinline__ double momentum_col(const double* X12, const double* X22, double im, double mass_total, int g, int rr) {
  if (rr < 3) return 0.0;
  const int i = rr - 3;
  double e;
  if (i < 3) e = g < 3 ? (g == i ? im : 0.0) : (g < 6 ? X12[3 * i + (g - 3)] : 0.0);
  else e = (g >= 3 && g < 6) ? X22[3 * (i - 3) + (g - 3)] : 0.0;
  return mass_total * e;
}
// rows 3..11 of the force column g (lanes 0..11) of df/du: [ [p_i - com]_x / m ; 0 ]
__device__ __forceinline__ double force_col(const double (*cp)[3], const double* com, double imt, int g, int rr) {
  if (rr >= 3 || g >= 12) return 0.0;
  const int i = g / 3, k = g % 3;
  if (k == rr) return 0.0;
  const int other = 3 - rr - k;
  const double d = (cp[i][other] - com[other]) * imt;
  return (k == (rr + 2) % 3) ? d : -d;
}

template <int NJ>
struct LaneKin {    // what the contact part needs from the evaluation
  double ah[3], og[3], omg[3], vog[3], vb[6];
  double sy, cy, sp, cp;   // Euler sines / cosines (world axes of the Euler joints)
};

// CHAIN: v[] holds this lane's own body entry (zeros in the lanes without a body) and becomes the sum over the subtree its coordinate
// moves: joints add their child's finished sum level by level from the leaf up (the additions of the table loop, in its order); the
// base body (coordinate 5) adds the two leg heads, the other base coordinates (whole robot) copy the base body's lane.
// lane SRC of this lane's 16-lane row to every lane of the row (DPP row_newbcast: no LDS round trip, unlike __shfl's ds_bpermute)
template <int SRC>
__device__ __forceinline__ double row_bcast_f64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x150 + SRC, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x150 + SRC, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <class C, int SRC_COORD>      // the value of the lane that carries coordinate SRC_COORD, in every lane of the node
__device__ __forceinline__ double node_bcast(double x) {
  if constexpr (C::LPN == 16) return row_bcast_f64<SRC_COORD - C::G0>(x);
  else
  return __shfl(x, SRC_COORD - C::G0, C::LPN);
}
template <class C, int N>
__device__ __forceinline__ void chain_subtree_sum(double (&v)[N], int g, bool is_joint, int depth) {
#pragma unroll
  for (int d = C::LEG - 1; d >= 1; --d) {
    const bool on = is_joint && depth == d;
    const double onf = on ? 1.0 : 0.0;       // one multiply-add instead of an addition and two selects (1.0 * child is exact; 0.0 * child adds a zero)
    for (int c = 0; c < N; ++c) { const double child = dpp_row_f64<kDppRowShl + 1>(v[c]); v[c] = fma(onf, child, v[c]); }
  }
  for (int c = 0; c < N; ++c) {
    const double h1 = dpp_row_f64<kDppRowShl + 1>(v[c]), h2 = dpp_row_f64<kDppRowShl + 1 + C::LEG>(v[c]);
    v[c] = g == 5 ? (v[c] + h1) + h2 : v[c];
  }
  for (int c = 0; c < N; ++c) { const double whole = node_bcast<C, 5>(v[c]); v[c] = g < 5 ? whole : v[c]; }
}

// One evaluation of the centroidal dynamics for the node owned by this lane group.  `stage` selects where the node-level
// results (A_b^{-1} blocks, contact points, com) are kept in LDS.
#ifdef BPMPC_EVAL_PROFILE
#define EVPROF(slot) do { if (evp) { const long long tn_ = clock64(); evp[slot] += tn_ - evp[9]; evp[9] = tn_; } } while (0)
#else
#define EVPROF(slot) ((void)0)
#endif
template <int NJ, bool DERIV = true, bool TWIST = true, class NodeLds = LinFastNodeLds<NJ>, class Shared = LinFastShared<NJ>, class Cfg = LinFastCfg<NJ>>
__device__ __forceinline__ void eval_lane(const DeviceModel& md, const Shared& sh, NodeLds& nl, int stage, const LaneBody& lb, const int* path, int g,
                                          const double* xh /*LDS: momentum [6], base position [3]*/, double qg, double ujg, LaneEval& ev, LaneKin<NJ>& kin,
                                          long long* evp = nullptr) {
#ifdef BPMPC_EVAL_PROFILE
  if (evp) evp[9] = clock64();
#endif
  using C = Cfg;
  constexpr int NB = C::NB, G = C::G, LPN = C::LPN, G0 = C::G0;    // g = lane in node + G0: coordinate c lives in lane c - G0
  const bool is_joint = g >= 6 && g < G, is_body = g >= 5 && g < G;
  const LinFastScalars& sc = sh.sc;        // the model's scalar constants, from LDS (LinFastScalars)
  const double mass_total = sc.robot_mass;
  const double* pb = xh + 6;           // base position, read from LDS at every use (registers are the scarce resource here)
  double (*cpos)[3] = nl.cpos(stage);   // contact points of this evaluation
  // ---- sin/cos of the own angle; Euler sin/cos to everybody
  double sg = 0.0, cg = 1.0;
  if (g >= 3 && g < G) sincos(qg, &sg, &cg);
  const double sy = node_bcast<C, 3>(sg), cy = node_bcast<C, 3>(cg), sp = node_bcast<C, 4>(sg), cp = node_bcast<C, 4>(cg),
               sr = node_bcast<C, 5>(sg), cr = node_bcast<C, 5>(cg);
  kin.sy = sy; kin.cy = cy; kin.sp = sp; kin.cp = cp;
  if constexpr (NodeLds::kFull) { if (g == G0) { nl.trig[0] = sy; nl.trig[1] = cy; nl.trig[2] = sp; nl.trig[3] = cp; } }
  EVPROF(0);
  // ---- joint-local transforms (to LDS for the table walk), chain walk
  double E[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
  if (is_joint) {
    const double* a = sh.axis[lb.body];
    const double v = 1.0 - cg;
    const double rot[9] = {cg + v * a[0] * a[0],        v * a[0] * a[1] - sg * a[2], v * a[0] * a[2] + sg * a[1],
                           v * a[1] * a[0] + sg * a[2], cg + v * a[1] * a[1],        v * a[1] * a[2] - sg * a[0],
                           v * a[2] * a[0] - sg * a[1], v * a[2] * a[1] + sg * a[0], cg + v * a[2] * a[2]};
    mat3_mul(sh.Rfix[lb.body], rot, E);
    if constexpr (!C::CHAIN) for (int i = 0; i < 9; ++i) nl.T[g - 6][i] = E[i];
  }
  if constexpr (!C::CHAIN) lds_wave_sync();
  double R[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                 -sp, cp * sr, cp * cr};
  double o[3] = {pb[0], pb[1], pb[2]};
  const int maxdepth = sc.max_depth;
  if constexpr (C::CHAIN) {
    // level by level: a joint of depth d composes the frame of its parent - the base (d = 1) or the lane before it, which finished at
    // level d - 1 - with its own local transform: the same two products as a step of the table walk, in the same order
    const double* pfx = sh.pfix[lb.body];
    const double pj[3] = {pfx[0], pfx[1], pfx[2]};
#pragma unroll
    for (int d = 1; d <= C::LEG; ++d) {
      double Rp[9], op[3];
      if (d == 1) { for (int i = 0; i < 9; ++i) Rp[i] = R[i]; for (int i = 0; i < 3; ++i) op[i] = o[i]; }
      else { for (int i = 0; i < 9; ++i) Rp[i] = dpp_row_f64<kDppRowShr + 1>(R[i]); for (int i = 0; i < 3; ++i) op[i] = dpp_row_f64<kDppRowShr + 1>(o[i]); }
      double t[3], Rn[9];
      mat3_vec(Rp, pj, t);
      mat3_mul(Rp, E, Rn);
      const bool on = is_joint && lb.depth == d;
      for (int i = 0; i < 3; ++i) o[i] = on ? op[i] + t[i] : o[i];
      for (int i = 0; i < 9; ++i) R[i] = on ? Rn[i] : R[i];
    }
  } else
#pragma nounroll
  for (int d = 0; d < maxdepth; ++d) {
    const bool on = is_joint && d < lb.depth;
    const int gj = on ? path[d] - 1 : 0;   // joint index (0-based) of the d-th body on the chain
    double Ej[9], pj[3], t[3], Rn[9];
    for (int i = 0; i < 9; ++i) Ej[i] = nl.T[gj][i];
    for (int i = 0; i < 3; ++i) pj[i] = sh.pfix[gj + 1][i];
    mat3_vec(R, pj, t);
    mat3_mul(R, Ej, Rn);
    for (int i = 0; i < 3; ++i) o[i] = on ? o[i] + t[i] : o[i];
    for (int i = 0; i < 9; ++i) R[i] = on ? Rn[i] : R[i];
  }
  // world axis and origin of the own coordinate (Euler ZYX = three successive revolute joints: z, rotated y, rotated x)
  double ah[3] = {0.0, 0.0, 0.0};
  if (g < 3) { ah[0] = g == 0 ? 1.0 : 0.0; ah[1] = g == 1 ? 1.0 : 0.0; ah[2] = g == 2 ? 1.0 : 0.0; }
  else if (g == 3) { ah[2] = 1.0; }
  else if (g == 4) { ah[0] = -sy; ah[1] = cy; }
  else if (g == 5) { ah[0] = cy * cp; ah[1] = sy * cp; ah[2] = -sp; }
  else if (g < G) mat3_vec(R, sh.axis[lb.body], ah);
  for (int i = 0; i < 3; ++i) { kin.ah[i] = ah[i]; kin.og[i] = o[i]; }
  EVPROF(1);
  // ---- body quantities, contact positions
  double s[10];                      // the lane's composite (CHAIN: starts as its own body's entry)
  for (int c = 0; c < 10; ++c) s[c] = 0.0;
  double cw[3] = {0.0, 0.0, 0.0}, Iwb[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (is_body) {
    double cb[3], d[3];
    mat3_vec(R, sh.com[lb.body], cb);
    for (int i = 0; i < 3; ++i) { cw[i] = cb[i] + o[i]; d[i] = cw[i] - pb[i]; }
    const double* I = sh.inertia[lb.body];
    const double Ib[9] = {I[0], I[1], I[2], I[1], I[3], I[4], I[2], I[4], I[5]};
    double T[9];
    mat3_mul(R, Ib, T);
    Iwb[0] = T[0] * R[0] + T[1] * R[1] + T[2] * R[2];
    Iwb[1] = T[0] * R[3] + T[1] * R[4] + T[2] * R[5];
    Iwb[2] = T[0] * R[6] + T[1] * R[7] + T[2] * R[8];
    Iwb[3] = T[3] * R[3] + T[4] * R[4] + T[5] * R[5];
    Iwb[4] = T[3] * R[6] + T[4] * R[7] + T[5] * R[8];
    Iwb[5] = T[6] * R[6] + T[7] * R[7] + T[8] * R[8];
    const double m = sh.mass[lb.body];
    const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    double* cm = C::CHAIN ? s : nl.comp[lb.body];
    cm[0] = m;
    cm[1] = m * d[0]; cm[2] = m * d[1]; cm[3] = m * d[2];
    cm[4] = Iwb[0] + m * (dd - d[0] * d[0]);
    cm[5] = Iwb[1] - m * d[0] * d[1];
    cm[6] = Iwb[2] - m * d[0] * d[2];
    cm[7] = Iwb[3] + m * (dd - d[1] * d[1]);
    cm[8] = Iwb[4] - m * d[1] * d[2];
    cm[9] = Iwb[5] + m * (dd - d[2] * d[2]);
    for (int i = 0; i < kNumContacts; ++i)
      if (sc.contact_body[i] == lb.body) {
        double t[3];
        mat3_vec(R, sc.contact_off[i], t);
        for (int a = 0; a < 3; ++a) cpos[i][a] = o[a] + t[a];
      }
  }
  lds_wave_sync();
  EVPROF(2);
  // ---- subtree sums -> composite mass, com, inertia about the composite com (lanes below 5 see the whole robot)
  if constexpr (C::CHAIN) chain_subtree_sum<C, 10>(s, g, is_joint, lb.depth);
  else {
#pragma nounroll
    for (int m = NB - 1; m >= 0; --m) {
      const double sel = ((lb.subtree >> m) & 1u) ? 1.0 : 0.0;
      for (int c = 0; c < 10; ++c) s[c] += sel * nl.comp[m][c];
    }
  }
  const double Mc = s[0];
  const double invM = Mc > 0.0 ? 1.0 / Mc : 0.0;
  const double Dv[3] = {s[1] * invM, s[2] * invM, s[3] * invM};
  const double DD = Dv[0] * Dv[0] + Dv[1] * Dv[1] + Dv[2] * Dv[2];
  const double Cc[3] = {pb[0] + Dv[0], pb[1] + Dv[1], pb[2] + Dv[2]};
  const double Ic[6] = {s[4] - Mc * (DD - Dv[0] * Dv[0]), s[5] + Mc * Dv[0] * Dv[1], s[6] + Mc * Dv[0] * Dv[2],
                        s[7] - Mc * (DD - Dv[1] * Dv[1]), s[8] + Mc * Dv[1] * Dv[2], s[9] - Mc * (DD - Dv[2] * Dv[2])};
  // whole-robot mass and com from lane 5 (the base body lane)
  const double Mtot = node_bcast<C, 5>(Mc);
  const double com[3] = {node_bcast<C, 5>(Cc[0]), node_bcast<C, 5>(Cc[1]), node_bcast<C, 5>(Cc[2])};
  if constexpr (DERIV) { if (g == G0) for (int i = 0; i < 3; ++i) nl.com[stage][i] = com[i]; }
  EVPROF(3);
  // ---- centroidal momentum matrix column
  double Ac[6];
  if (g < 3) {
    for (int i = 0; i < 3; ++i) { Ac[i] = (i == g) ? Mtot : 0.0; Ac[3 + i] = 0.0; }
  } else {
    const double rC[3] = {Cc[0] - o[0], Cc[1] - o[1], Cc[2] - o[2]};
    double vC[3], t[3], Iw[3];
    cross3(ah, rC, vC);
    const double dC[3] = {Cc[0] - com[0], Cc[1] - com[1], Cc[2] - com[2]};
    cross3(dC, vC, t);
    sym3_mul(Ic, ah, Iw);
    for (int i = 0; i < 3; ++i) { Ac[i] = Mc * vC[i]; Ac[3 + i] = Iw[i] + Mc * t[i]; }
  }
  if (g >= G) for (int i = 0; i < 6; ++i) Ac[i] = 0.0;
  EVPROF(3);
  // ---- base velocity: rhs = m hbar - A_j v_j, A_b^{-1} blocks (kept in LDS, every later use re-reads them)
  const double im = 1.0 / Mtot;
  double th[3], pd[3];
  {
    double rhs[6];
    for (int i = 0; i < 6; ++i) {
      const double part = is_joint ? Ac[i] * ujg : 0.0;
      rhs[i] = mass_total * xh[i] - node_allreduce_add<LPN>(part);
    }
    double A12[9], A22[9], X12[9], X22[9];
    for (int i = 0; i < 3; ++i) {
      A12[3 * i] = node_bcast<C, 3>(Ac[i]); A12[3 * i + 1] = node_bcast<C, 4>(Ac[i]); A12[3 * i + 2] = node_bcast<C, 5>(Ac[i]);
      A22[3 * i] = node_bcast<C, 3>(Ac[3 + i]); A22[3 * i + 1] = node_bcast<C, 4>(Ac[3 + i]); A22[3 * i + 2] = node_bcast<C, 5>(Ac[3 + i]);
    }
    const double* M = A22;
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double idet = 1.0 / (M[0] * c00 + M[1] * c01 + M[2] * c02);
    X22[0] = c00 * idet; X22[1] = (M[2] * M[7] - M[1] * M[8]) * idet; X22[2] = (M[1] * M[5] - M[2] * M[4]) * idet;
    X22[3] = c01 * idet; X22[4] = (M[0] * M[8] - M[2] * M[6]) * idet; X22[5] = (M[2] * M[3] - M[0] * M[5]) * idet;
    X22[6] = c02 * idet; X22[7] = (M[1] * M[6] - M[0] * M[7]) * idet; X22[8] = (M[0] * M[4] - M[1] * M[3]) * idet;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) X12[3 * i + j] = -im * (A12[3 * i] * X22[j] + A12[3 * i + 1] * X22[3 + j] + A12[3 * i + 2] * X22[6 + j]);
    if constexpr (DERIV) { if (g == G0) for (int i = 0; i < 9; ++i) { nl.X12[stage][i] = X12[i]; nl.X22[stage][i] = X22[i]; } }
    mat3_vec(X22, &rhs[3], th);
    mat3_vec(X12, &rhs[3], pd);
    for (int i = 0; i < 3; ++i) pd[i] += im * rhs[i];
  }
  for (int i = 0; i < 3; ++i) { kin.vb[i] = pd[i]; kin.vb[3 + i] = th[i]; }
  if constexpr (NodeLds::kFull) { if (g == G0) for (int i = 0; i < 3; ++i) nl.vlin[stage][i] = pd[i]; }
  const double vg = g == 0 ? pd[0] : g == 1 ? pd[1] : g == 2 ? pd[2] : g == 3 ? th[0] : g == 4 ? th[1] : g == 5 ? th[2] : (g < G ? ujg : 0.0);
  ev.vg = vg;
  EVPROF(4);
  // ---- flow map rows 0..5 (the derivative kernel re-reads the node's com from LDS from here on: three values less in every lane's registers)
  if constexpr (DERIV) lds_wave_sync();
  const double* comn = com;
  if constexpr (DERIV) comn = nl.com[stage];
  {
    double lin[3] = {0.0, 0.0, -9.81 * mass_total}, ang[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < kNumContacts; ++i) {
      const double r[3] = {cpos[i][0] - comn[0], cpos[i][1] - comn[1], cpos[i][2] - comn[2]};
      const double Fi[3] = {nl.u[3 * i], nl.u[3 * i + 1], nl.u[3 * i + 2]};
      lin[0] += Fi[0]; lin[1] += Fi[1]; lin[2] += Fi[2];
      ang[0] += r[1] * Fi[2] - r[2] * Fi[1];
      ang[1] += r[2] * Fi[0] - r[0] * Fi[2];
      ang[2] += r[0] * Fi[1] - r[1] * Fi[0];
    }
    const double imt_c = 1.0 / mass_total;             // one division instead of six (an fp64 division is ~25 instructions)
    if constexpr (NodeLds::kFull) { if (g == G0) for (int i = 0; i < 3; ++i) { nl.fh[stage][i] = lin[i] * imt_c; nl.fh[stage][3 + i] = ang[i] * imt_c; } }
    else for (int i = 0; i < 3; ++i) { ev.fh[i] = lin[i] * imt_c; ev.fh[3 + i] = ang[i] * imt_c; }
  }
  double om[3] = {0.0, 0.0, 0.0}, vo[3] = {pd[0], pd[1], pd[2]};
  if constexpr (TWIST) {
  EVPROF(4);
  // ---- twists: omega_g = sum over the revolute ancestors (self included) of a v, v_og = velocity of the joint origin
  const double wvo[3] = {ah[0] * vg, ah[1] * vg, ah[2] * vg};
  if constexpr (C::CHAIN) { if (g >= 3 && g < 6) for (int i = 0; i < 3; ++i) nl.wv[g - 3][i] = wvo[i]; }
  else if (g >= 3 && g < G) {
    for (int i = 0; i < 3; ++i) { nl.wv[g - 3][i] = wvo[i]; nl.og[g - 3][i] = o[i]; }
  }
  lds_wave_sync();
  {
    const int top = g < 3 ? 2 : (g < 6 ? g : 5);        // Euler joints up to the own one (joints and bodies: all three)
    for (int k = 3; k <= 5; ++k) {
      const double sel = k <= top ? 1.0 : 0.0;
      for (int i = 0; i < 3; ++i) om[i] += sel * nl.wv[k - 3][i];
    }
    double prev[3] = {pb[0], pb[1], pb[2]};
    if constexpr (C::CHAIN) {
      // level by level as the composition: the parent's twist (angular velocity with its own joint rate, velocity of its origin) and origin
      // come from the base (depth 1) or from the lane before
#pragma unroll
      for (int d = 1; d <= C::LEG; ++d) {
        double omp[3], vop[3], orp[3];
        if (d == 1) { for (int i = 0; i < 3; ++i) { omp[i] = om[i]; vop[i] = vo[i]; orp[i] = prev[i]; } }
        else { for (int i = 0; i < 3; ++i) { omp[i] = dpp_row_f64<kDppRowShr + 1>(om[i]); vop[i] = dpp_row_f64<kDppRowShr + 1>(vo[i]); orp[i] = dpp_row_f64<kDppRowShr + 1>(o[i]); } }
        const double r[3] = {o[0] - orp[0], o[1] - orp[1], o[2] - orp[2]};
        double t[3];
        cross3(omp, r, t);
        const bool on = is_joint && lb.depth == d;
        for (int i = 0; i < 3; ++i) {
          vo[i] = on ? vop[i] + t[i] : vo[i];
          om[i] = on ? omp[i] + wvo[i] : om[i];
        }
      }
    } else
#pragma nounroll
    for (int d = 0; d < maxdepth; ++d) {
      const bool on = is_joint && d < lb.depth;
      const int gj = on ? path[d] - 1 : 0;   // joint index (0-based) of the d-th body on the chain
      const double oj[3] = {nl.og[gj + 3][0], nl.og[gj + 3][1], nl.og[gj + 3][2]};
      const double wj[3] = {nl.wv[gj + 3][0], nl.wv[gj + 3][1], nl.wv[gj + 3][2]};
      const double r[3] = {oj[0] - prev[0], oj[1] - prev[1], oj[2] - prev[2]};
      double t[3];
      cross3(om, r, t);
      for (int i = 0; i < 3; ++i) {
        vo[i] = on ? vo[i] + t[i] : vo[i];
        om[i] = on ? om[i] + wj[i] : om[i];
        prev[i] = on ? oj[i] : prev[i];
      }
    }
  }
  for (int i = 0; i < 3; ++i) { kin.omg[i] = om[i]; kin.vog[i] = vo[i]; }
  }
  if constexpr (DERIV) {
  EVPROF(5);
  // ---- body momenta about o0 (hb shares LDS with comp, which is dead now), subtree momenta
  double hs[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (is_body) {
    const double rc[3] = {cw[0] - o[0], cw[1] - o[1], cw[2] - o[2]};
    double t[3], l[3], Iw[3], L[3];
    cross3(om, rc, t);
    const double mb = sh.mass[lb.body];
    for (int i = 0; i < 3; ++i) l[i] = mb * (vo[i] + t[i]);
    sym3_mul(Iwb, om, Iw);
    const double d0[3] = {cw[0] - pb[0], cw[1] - pb[1], cw[2] - pb[2]};
    cross3(d0, l, L);
    if constexpr (C::CHAIN) for (int i = 0; i < 3; ++i) { hs[i] = l[i]; hs[3 + i] = Iw[i] + L[i]; }
    else for (int i = 0; i < 3; ++i) { nl.hb[lb.body][i] = l[i]; nl.hb[lb.body][3 + i] = Iw[i] + L[i]; }
  }
  if constexpr (C::CHAIN) chain_subtree_sum<C, 6>(hs, g, is_joint, lb.depth);
  else {
    lds_wave_sync();
#pragma nounroll
    for (int m = NB - 1; m >= 0; --m) {
      const double sel = ((lb.subtree >> m) & 1u) ? 1.0 : 0.0;
      for (int c = 0; c < 6; ++c) hs[c] += sel * nl.hb[m][c];
    }
  }
  const double ltot[3] = {node_bcast<C, 5>(hs[0]), node_bcast<C, 5>(hs[1]), node_bcast<C, 5>(hs[2])};
  EVPROF(6);
  // ---- column 6+g: d(A v)/dq_g -> d v_base/dq_g, angular-momentum-rate row; joint-velocity column
  {
    double dl[3] = {0.0, 0.0, 0.0}, dL[3] = {0.0, 0.0, 0.0};
    if (g >= 3 && g < G) {
      const double* l = hs;
      const double sv[3] = {pb[0] - o[0], pb[1] - o[1], pb[2] - o[2]};
      double t[3], Lk[3];
      cross3(sv, l, t);
      for (int i = 0; i < 3; ++i) Lk[i] = hs[3 + i] + t[i];
      double wp[3], up[3], vC[3], rC[3], linp[3], angp[3], Iw[3];
      cross3(ah, om, wp);
      cross3(ah, vo, up);
      for (int i = 0; i < 3; ++i) rC[i] = Cc[i] - o[i];
      cross3(wp, rC, t);
      for (int i = 0; i < 3; ++i) vC[i] = up[i] + t[i];
      for (int i = 0; i < 3; ++i) linp[i] = Mc * vC[i];
      sym3_mul(Ic, wp, Iw);
      cross3(rC, vC, t);
      for (int i = 0; i < 3; ++i) angp[i] = Iw[i] + Mc * t[i];
      double al[3], aL[3], dLk[3];
      cross3(ah, l, al);
      cross3(ah, Lk, aL);
      for (int i = 0; i < 3; ++i) { dl[i] = al[i] - linp[i]; dLk[i] = aL[i] - angp[i]; }
      const double sc[3] = {o[0] - comn[0], o[1] - comn[1], o[2] - comn[2]};
      const double jc[3] = {Ac[0] * im, Ac[1] * im, Ac[2] * im};
      double t2[3];
      cross3(sc, dl, t);
      cross3(jc, ltot, t2);
      for (int i = 0; i < 3; ++i) dL[i] = dLk[i] + t[i] - t2[i];
    }
    const double* X12 = nl.X12[stage];
    const double* X22 = nl.X22[stage];
    for (int i = 0; i < 3; ++i) {
      ev.ar_q[3 + i] = -(im * dl[i] + X12[3 * i] * dL[0] + X12[3 * i + 1] * dL[1] + X12[3 * i + 2] * dL[2]);
      ev.ar_q[6 + i] = -(X22[3 * i] * dL[0] + X22[3 * i + 1] * dL[1] + X22[3 * i + 2] * dL[2]);
      ev.br_j[i] = -(im * Ac[i] + X12[3 * i] * Ac[3] + X12[3 * i + 1] * Ac[4] + X12[3 * i + 2] * Ac[5]);
      ev.br_j[3 + i] = -(X22[3 * i] * Ac[3] + X22[3 * i + 1] * Ac[4] + X22[3 * i + 2] * Ac[5]);
    }
    if (!is_joint) for (int i = 0; i < 6; ++i) ev.br_j[i] = 0.0;
    const double jcm[3] = {Ac[0] * im, Ac[1] * im, Ac[2] * im};
    double acc[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < kNumContacts; ++i) {
      double jcol[3] = {0.0, 0.0, 0.0};
      if (g < 3) { jcol[0] = g == 0 ? 1.0 : 0.0; jcol[1] = g == 1 ? 1.0 : 0.0; jcol[2] = g == 2 ? 1.0 : 0.0; }
      else if (g < G && (g < 6 || ((sc.contact_path[i] >> (g - 5)) & 1u))) {
        const double r[3] = {cpos[i][0] - o[0], cpos[i][1] - o[1], cpos[i][2] - o[2]};
        cross3(ah, r, jcol);
      }
      const double d[3] = {jcol[0] - jcm[0], jcol[1] - jcm[1], jcol[2] - jcm[2]};
      const double Fi[3] = {nl.u[3 * i], nl.u[3 * i + 1], nl.u[3 * i + 2]};
      acc[0] += d[1] * Fi[2] - d[2] * Fi[1];
      acc[1] += d[2] * Fi[0] - d[0] * Fi[2];
      acc[2] += d[0] * Fi[1] - d[1] * Fi[0];
    }
    const double imt = 1.0 / mass_total;
    for (int r = 0; r < 3; ++r) ev.ar_q[r] = acc[r] * imt;
  }
  }
  EVPROF(7);
}

// Output views of the fast linearisation: wave-uniform base pointers of the whole batch plus the node slot of this lane
// group.  Per-node pointers (14 x 64 bit per lane) would stay live across the whole kernel; the addresses are formed at
// the point of use instead.
// Stage-one columns parked in HBM scratch (not in LDS: 1.6 KB per node would cost a quarter of the occupancy): rows 3..11 of
// column 6+g for 16 lanes, rows 6..11 of the joint-velocity column for the joints; layout [row][lane] (coalesced).  They are
// written before the second evaluation and read after it, by when the stores have long retired.
constexpr int kLinParkDoublesPerLane = 15;   // times the lanes per node
// Role stores (RoleSlots below) are issued by every lane of a node; a lane without the role writes into this scratch region instead of
// being masked out: kLinDumpNodes lines of 16 doubles (the node slot picks the line) plus the largest row offset a store adds.
constexpr int kLinDumpNodes = 1024, kLinDumpSlack = 1024;
constexpr size_t kLinDumpDoubles = (size_t)kLinDumpNodes * 16 + kLinDumpSlack;
// Q and R of a node are dt x (constant weight) except for the Hessian shift on their diagonals and the four 3x3 force blocks of R
// (cone Hessians).  Besides the full matrices the lineariser leaves exactly that part in a compact record: [shift, block entries
// (column-major inside the block row: 3 * column + row % 3)], so that the change of variables need not read 7.7 KB of mostly
// constant numbers per node.
constexpr int kQrdStride = 40;
constexpr int kQrdRShift = 37;     // ILQR lineariser only: the shift of the diagonal of R (the Hessian shift + DIAGONAL_SHIFT / dt), read by project_mfma.h
// The node record of the lane-per-coordinate kernels (Buffers::n_aux): dt, zref[4], zdref[4], padded to a 128-byte line
constexpr int kNodeAux = 16;
// step lengths of the DDP line search incl. the baseline 0 (task.info:147-155: 1, 0.5, .. >= minStepLength 1e-2 - eight), rolled out in one launch (rollout.h)
constexpr int kMaxDdpSteps = 16;

// What a lane needs of the node's iterate, loaded by the kernel wrapper BEFORE the model block is staged (and before the node is known to
// exist: the addresses only depend on the slot), so that the memory round trips of a workgroup's start overlap instead of following
// each other: x / u elements ln and ln + 16, and the entries of x_next and x_ref of the lane's rows.
struct LinFastPre {
  double x0, x1, u0, u1, xn_q, xn_h, xn_t, xr_q, xr_h, xr_t;
  double aux;       // lane ln: entry ln of the node record (kNodeAux: dt, zref[4], zdref[4])
};
template <class Cfg>
__device__ __forceinline__ LinFastPre linearize_preload(const double* x, const double* xnext, const double* u, const double* xref, int ln, const double* aux = nullptr) {
  constexpr int G = Cfg::G, NX = Cfg::NX, G0 = Cfg::G0;
  const int g = ln + G0;
  const bool tr = G0 > 0 && ln < 3;
  LinFastPre p;
  p.aux = (aux && ln < kNodeAux) ? aux[ln] : 0.0;
  p.x0 = x[ln]; p.u0 = u[ln];
  p.x1 = ln + 16 < NX ? x[ln + 16] : 0.0; p.u1 = ln + 16 < NX ? u[ln + 16] : 0.0;
  p.xn_q = g < G ? xnext[6 + g] : 0.0; p.xn_h = ln < 6 ? xnext[ln] : 0.0; p.xn_t = tr ? xnext[6 + ln] : 0.0;
  p.xr_q = g < G ? xref[6 + g] : 0.0; p.xr_h = ln < 6 ? xref[ln] : 0.0; p.xr_t = tr ? xref[6 + ln] : 0.0;
  return p;
}

struct LinFastOut {
  double *A, *B, *b, *Q, *R, *q, *r, *c, *C, *D, *e, *perf;
  int* nc;
  double* park;      // scratch, 15 * LPN doubles per node: the stage-one Jacobian columns wait here for the RK2 combination
  double* dump;      // kLinDumpDoubles: where the lanes without a role write (never read)
  double* qrd;       // kQrdStride doubles per node: the node-dependent part of Q and R in compact form (read by project_mfma.h)
  double* prof;      // this node's debug slot or nullptr
  size_t s;          // node slot (problem * max_nodes + node)
  double ilqr_shift; // ILQR only: ddp.lineSearch.hessianCorrectionMultiple, added to every diagonal entry of the dt-scaled R (DIAGONAL_SHIFT)
};

// A lane owns up to four columns of the node's matrices ("roles"): x column 6+g (its coordinate), x column 6+ln (base translation: packed
// layout, lanes 0..2), x column ln (momentum, lanes 0..5), u column ln (force, lanes 0..11), u column 12+(g-6) (joint velocity).  A row of
// a matrix pair [X-type | U-type] ([A | B], [C | D], [Q | R]; NX = NU doubles per row each) leaves in NS store instructions that ALL lanes
// issue: roles whose lane sets are disjoint share an instruction (the lane picks pointer and value), and a lane without a role in a slot
// points into the dump.  No execution mask is touched (a predicated store costs s_and_saveexec / branch / s_or around it - the row loops
// of this kernel were two thirds bookkeeping), the pointers are formed once per matrix pair and the row offset is an immediate.
//   16 coordinates (G0 = 0):  slot 0 = coordinate | slot 1 = momentum (lanes 0..5) + joint velocity (6..15) | slot 2 = force (0..11)
//   packed (G0 = 3):          slot 0 = coordinate (0..14) | slot 1 = translation (0..2) + joint velocity (3..14) | slot 2 = force | slot 3 = momentum
// The row loops are fully unrolled (compile-time row offsets); without a fence between the rows the scheduler hoists the LDS reads of
// all of them to the top (hundreds of registers, spills).
#define LIN_ROW_FENCE() __builtin_amdgcn_sched_barrier(0)
template <class Cfg>
struct RoleSlots {
  static constexpr bool PACKED = Cfg::G0 > 0;
  static constexpr int NS = PACKED ? 4 : 3;
  static_assert(Cfg::LPN == 16 && (PACKED || Cfg::G == 16), "role slots are laid out for sixteen lanes per node");
  int ln, g;
  double* dump;      // this lane's word of the node's dump line
  __device__ __forceinline__ void pointers(double* Xn, double* Un, double* (&p)[NS]) const {
    if constexpr (!PACKED) {
      p[0] = Xn + (6 + ln);
      p[1] = ln < 6 ? Xn + ln : Un + (12 + ln - 6);
      p[2] = ln < 12 ? Un + ln : dump;
    } else {
      p[0] = g < Cfg::G ? Xn + (6 + g) : dump;
      p[1] = ln < 3 ? Xn + (6 + ln) : (g < Cfg::G ? Un + (12 + g - 6) : dump);
      p[2] = ln < 12 ? Un + ln : dump;
      p[3] = ln < 6 ? Xn + ln : dump;
    }
  }
  // values of the roles of this lane (what a lane passes for a role it does not have is never stored where it matters)
  __device__ __forceinline__ void store(double* const (&p)[NS], int off, double v_coord, double v_trans, double v_mom, double v_force, double v_joint) const {
    p[0][off] = v_coord;
    if constexpr (!PACKED) {
      p[1][off] = ln < 6 ? v_mom : v_joint;
      p[2][off] = v_force;
    } else {
      p[1][off] = ln < 3 ? v_trans : v_joint;
      p[2][off] = v_force;
      p[3][off] = v_mom;
    }
  }
};

// MAT = true: the complete per-node LQ model of the reference (A, B, b, Q, R, q, r, c, C, D, e zero padded to 16 rows) is written -
// the materialised formulation the parity stages read and the roofline unit is defined on.  MAT = false ("fused" solve mode): only
// what the rest of the solve reads leaves the kernel - rows 3..11 of A and B (the others are structural and regenerated by the change
// of variables), b, q, r, the nc rows of C, D, e, the 320-byte record of the node-dependent part of Q and R; 9.6 instead of 21.8 KB per
// node.  The numbers that are written are the same bits in both modes.
// `ln`: lane inside the node's lane group; it carries coordinate g = ln + G0 (LinFastCfg) and the lane-numbered roles.
// ILQR (the DDP solver, solver.hip run_ddp; reference body: node_lq.h linearize_node with `ilqr`): the continuous-time model at the node discretised
// by ONE Euler step - A = I + dt A_c, B = dt B_c: no second evaluation -, no dynamics bias (b = 0: the nominal trajectory of a DDP is a roll-out),
// o.ilqr_shift on the diagonal of the dt-scaled R; cost and constraints as in the multiple-shooting transcription.
template <int NJ, bool MAT = true, class Cfg = LinFastCfg<NJ, true>, class NL = LinFastNodeLds<NJ, true, Cfg::CHAIN, false>, bool ILQR = false>
__device__ __forceinline__ void linearize_fast(const DeviceModel& md, const LinFastShared<NJ>& sh, NL& nl, bool valid,
                                               const NodeInputs& in, const LinFastPre& pre, const LinFastOut& o, int ln) {
#ifdef BPMPC_LINFAST_PROFILE
  long long lf_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long lf_prev = clock64();
  const long long lf_wall0 = wall_clock64(), lf_c0 = lf_prev;      // slots 6, 7: the wave's cycles and its wall time in 10 ns ticks (-> shader clock)
#define LFPROF(slot) do { const long long tn_ = clock64(); lf_t[slot] += tn_ - lf_prev; lf_prev = tn_; } while (0)
#else
#define LFPROF(slot) ((void)0)
#endif
  using C = Cfg;
  constexpr int G = C::G, NX = C::NX, NU = C::NU, LPN = C::LPN, G0 = C::G0;
  constexpr bool PACKED = G0 > 0;
  const int g = ln + G0;
  const bool tr = PACKED && ln < 3;    // this lane also writes the constant columns of base translation ln (x column 6 + ln)
  if (!valid) return;
  if (in.kind == 1) {  // event node: identity jump map, no input, no cost (LPN lanes write the node)
    double d2 = 0.0;
    if constexpr (MAT) {      // fused mode: the change of variables generates the identity jump map and the zero cost of an event node itself
      for (int idx = ln; idx < NX * NX; idx += LPN) { (o.A + o.s * (NX * NX))[idx] = (idx / NX == idx % NX) ? 1.0 : 0.0; (o.Q + o.s * (NX * NX))[idx] = 0.0; }
      for (int idx = ln; idx < NX * NU; idx += LPN) { (o.B + o.s * (NX * NU))[idx] = 0.0; }
      for (int idx = ln; idx < NU * NU; idx += LPN) (o.R + o.s * (NU * NU))[idx] = 0.0;
      for (int idx = ln; idx < kMaxEqRows * NX; idx += LPN) (o.C + o.s * (kMaxEqRows * NX))[idx] = 0.0;
      for (int idx = ln; idx < kMaxEqRows * NU; idx += LPN) (o.D + o.s * (kMaxEqRows * NU))[idx] = 0.0;
      for (int idx = ln; idx < kMaxEqRows; idx += LPN) (o.e + o.s * (kMaxEqRows))[idx] = 0.0;
    }
    {   // b = x - x_next from the lane's preloaded entries (rows 6 + g, ln and - packed lanes 0..2 - 6 + ln): no load behind the node's own stores
      static_assert(LPN == 16 && NX <= 32, "two elements of x per lane");
      const double xq_lo = __shfl(pre.x0, (6 + g) & 15, LPN), xq_hi = __shfl(pre.x1, (6 + g - 16) & 15, LPN);      // (both by every lane of the node)
      const double xq = 6 + g < 16 ? xq_lo : xq_hi;
      const double xt = __shfl(pre.x0, (6 + ln) & 15, LPN);       // (used by the packed lanes 0..2 only: 6 + ln < 16)
      if (g < G) { const double d = ILQR ? 0.0 : xq - pre.xn_q; (o.b + o.s * (NX))[6 + g] = d; d2 += d * d; }
      if (ln < 6) { const double d = ILQR ? 0.0 : pre.x0 - pre.xn_h; (o.b + o.s * (NX))[ln] = d; d2 += d * d; }
      if (tr) { const double d = ILQR ? 0.0 : xt - pre.xn_t; (o.b + o.s * (NX))[6 + ln] = d; d2 += d * d; }
      if constexpr (MAT) for (int idx = ln; idx < NX; idx += LPN) { (o.q + o.s * (NX))[idx] = 0.0; (o.r + o.s * (NU))[idx] = 0.0; }
    }
    d2 = node_allreduce_add<LPN>(d2);
    if (ln == 0) { if constexpr (MAT) (o.c + o.s * (1))[0] = 0.0; (o.nc + o.s * (1))[0] = 0; (o.perf + o.s * (3))[0] = 0.0; (o.perf + o.s * (3))[1] = d2; (o.perf + o.s * (3))[2] = 0.0; }
    return;
  }
  const bool is_joint = g >= 6 && g < G;
  static_assert(LPN == 16, "the node record is spread over sixteen lanes");
  const double dt = row_bcast_f64<0>(pre.aux), hdt = 0.5 * dt;      // entry 0 of the node record (kNodeAux)
  const int mode = in.mode;
  const LinFastScalars& sc = sh.sc;
  const double mass_total = sc.robot_mass, imt = 1.0 / sc.robot_mass;
  using Slots = RoleSlots<C>;
  constexpr int NS = Slots::NS;
  const Slots slots{ln, g, o.dump + ((o.s & (size_t)(kLinDumpNodes - 1)) * 16 + ln)};
  // ---- stage the node inputs in LDS: after this block nothing is read from global memory except model constants
  static_assert(LPN == 16 && NX <= 32 && NX == NU, "two elements of x and of u per lane");
  nl.x[ln] = pre.x0; nl.u[ln] = pre.u0;
  if (ln + 16 < NX) { nl.x[ln + 16] = pre.x1; nl.u[ln + 16] = pre.u1; }
  // the entries of x_next and x_ref this lane needs later (rows 6+g, ln and - packed lanes 0..2 - 6+ln)
  const double xn_q = pre.xn_q, xn_h = pre.xn_h, xr_q = pre.xr_q, xr_h = pre.xr_h, xn_t = pre.xn_t, xr_t = pre.xr_t;
  if (ln >= 1 && ln < 1 + kNumContacts) nl.zref[ln - 1] = pre.aux;
  if (ln >= 1 + kNumContacts && ln < 1 + 2 * kNumContacts) nl.zdref[ln - 1 - kNumContacts] = pre.aux;
  LaneBody lb;
  {
    const int body = (g >= 5 && g < G) ? g - 5 : 0;
    lb.body = body;
    lb.depth = sh.depth[body];
    lb.subtree = sh.subtree[body];     // lanes below 5 move the whole robot (body 0's subtree)
  }
  const int* path = sh.path[lb.body];
  lds_wave_sync();
  const double* xh = nl.x;             // normalised momentum and base position, read from LDS where they are used
  const double* pb = nl.x + 6;
  const double qg = g < G ? nl.x[6 + g] : 0.0;
  const double ujg = is_joint ? nl.u[12 + g - 6] : 0.0;

  LFPROF(0);
  LaneEval e1;
  LaneKin<NJ> kin;
#ifdef BPMPC_EVAL_PROFILE
  long long evacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  eval_lane<NJ, true, true, NL, LinFastShared<NJ>, C>(md, sh, nl, 0, lb, path, g, xh, qg, ujg, e1, kin, evacc);
#else
  eval_lane<NJ, true, true, NL, LinFastShared<NJ>, C>(md, sh, nl, 0, lb, path, g, xh, qg, ujg, e1, kin);
#endif
  LFPROF(1);
  // park the stage-one columns (HBM scratch) for the RK2 combination
  if constexpr (NL::kPark) {
    for (int rr = 0; rr < 9; ++rr) nl.park[rr][ln] = e1.ar_q[rr];
    for (int rr = 0; rr < 6; ++rr) nl.park[9 + rr][ln] = e1.br_j[rr];
  } else {
    double* pk = o.park + o.s * (15 * LPN);
    for (int rr = 0; rr < 9; ++rr) pk[rr * LPN + ln] = e1.ar_q[rr];
    for (int rr = 0; rr < 6; ++rr) pk[9 * LPN + rr * LPN + ln] = is_joint ? e1.br_j[rr] : 0.0;   // whole 128-byte lines
  }
  if constexpr (NL::kPark) nl.park[15][ln] = e1.vg;      // (the lane's own slot: no barrier) - waits there for b instead of in a register across the second evaluation

  // =========================== contact part (first stage only) ===========================
  double (*cpos1)[3] = nl.cps[0];
  double (*cvel1)[3] = nl.cvel();       // shares storage with the tables of the evaluation, which are dead now
  lds_wave_sync();
  if (g >= 5 && g < G)
    for (int i = 0; i < kNumContacts; ++i)
      if (sc.contact_body[i] == lb.body) {
        const double r[3] = {cpos1[i][0] - kin.og[0], cpos1[i][1] - kin.og[1], cpos1[i][2] - kin.og[2]};
        double t[3];
        cross3(kin.omg, r, t);
        for (int a = 0; a < 3; ++a) cvel1[i][a] = kin.vog[a] + t[a];
      }
  lds_wave_sync();
  const double tsy = nl.trig[0], tcy = nl.trig[1], tsp = nl.trig[2], tcp = nl.trig[3];   // Euler sines / cosines of the first stage
  // rows in registration order zeroForce_i, zeroVelocity_i, normalVelocity_i (src/BipedalRobotInterface.cpp:187-191).  Per contact the
  // three rows of a stance contact (zero velocity) or of a swing contact (zero force) and, swing only, the normal-velocity row; every
  // store is issued by all lanes (RoleSlots), the fourth row of a stance contact goes to the dump.
  double eq_sse = 0.0;
  double e_mine = 0.0;                  // lane r keeps e[r]: one coalesced store at the end instead of one word per row
  int nc;
  {
    double* pcd[NS];
    slots.pointers(o.C + o.s * (kMaxEqRows * NX), o.D + o.s * (kMaxEqRows * NU), pcd);
    if constexpr (MAT) {                // zero padding of rows 12..15 first (the rows that exist overwrite it: same lane, same address, program order);
      for (int r = 12; r < kMaxEqRows; ++r) slots.store(pcd, r * NX, 0.0, 0.0, 0.0, 0.0, 0.0);   // the LU kernel reads nc rows only
    }
    double hcol[6];                     // own momentum column (rows 6..11 of df/dx, lanes 0..5): the same for every contact
    for (int l = 0; l < 6; ++l) hcol[l] = momentum_col(nl.X12[0], nl.X22[0], imt, mass_total, ln, 3 + l);
    // J_i,base (d v_base / d column) = d(pdot) + d(omega_base) x (p_i - o0),  d(omega_base) = W d(thetadot): the angular part W d(thetadot) of the lane's three
    // columns does not depend on the contact - formed once here, not once per contact
    auto omega_of = [&](const double* col6, double* w) {
      const double th0 = col6[3], th1 = col6[4], th2 = col6[5];
      w[0] = -tsy * th1 + tcy * tcp * th2; w[1] = tcy * th1 + tsy * tcp * th2; w[2] = th0 - tsp * th2;
    };
    double wq[3], wh[3], wj[3];
    omega_of(&e1.ar_q[3], wq);
    omega_of(hcol, wh);
    omega_of(e1.br_j, wj);
    const double pos_gain = sc.pos_gain;
    int row = 0;
#pragma nounroll
    for (int i = 0; i < kNumContacts; ++i) {
      const bool stance = stance_flag(mode, i);
      const double cp_i[3] = {cpos1[i][0], cpos1[i][1], cpos1[i][2]};
      const double cv_i[3] = {cvel1[i][0], cvel1[i][1], cvel1[i][2]};
      // own columns of J_i and d(J_i v)/dq
      const bool on_path = g >= 3 && g < G && (g < 6 || ((sc.contact_path[i] >> (g - 5)) & 1u));
      double Jc[3], DJ[3];
      {
        const double r[3] = {cp_i[0] - kin.og[0], cp_i[1] - kin.og[1], cp_i[2] - kin.og[2]};
        double jr[3];
        cross3(kin.ah, r, jr);
        const double dv[3] = {cv_i[0] - kin.vog[0], cv_i[1] - kin.vog[1], cv_i[2] - kin.vog[2]};
        double t1[3], wa[3], t2[3];
        cross3(kin.ah, dv, t1);
        cross3(kin.omg, kin.ah, wa);
        cross3(wa, r, t2);
        for (int k = 0; k < 3; ++k) {
          Jc[k] = on_path ? jr[k] : ((g == k) ? 1.0 : 0.0);        // base translation g < 3: unit column (no such lane in the packed layout)
          DJ[k] = on_path ? t1[k] + t2[k] : 0.0;
        }
      }
      const double rb[3] = {cp_i[0] - pb[0], cp_i[1] - pb[1], cp_i[2] - pb[2]};
      auto base_part = [&](const double* col6, const double* w, double* outv) {
        double t[3];
        cross3(w, rb, t);
        for (int a = 0; a < 3; ++a) outv[a] = col6[a] + t[a];
      };
      double bq[3], bh[3], bj[3];
      base_part(&e1.ar_q[3], wq, bq);
      base_part(hcol, wh, bh);
      base_part(e1.br_j, wj, bj);
      // velocity row of axis a: zero velocity of a stance contact (a = 0..2), normal velocity of a swing contact (a = 2)
      double vqa[3], vja[3];
      for (int a = 0; a < 3; ++a) { vqa[a] = bq[a] + DJ[a]; vja[a] = bj[a] + Jc[a]; }
      double vt2 = 0.0;
      if (pos_gain != 0.0) { vqa[2] += pos_gain * Jc[2]; vt2 = ln == 2 ? pos_gain : 0.0; }      // d p_z / d (base z) = 1 (packed layout: translation role)
      double* pc[NS];
      for (int k = 0; k < NS; ++k) pc[k] = pcd[k] + row * NX;
      for (int a = 0; a < 3; ++a) {
        // stance: zero velocity (ZeroVelocityConstraintCppAd); swing: zero force (ZeroForceConstraint.cpp:64-72)
        double ev = cv_i[a];
        if (a == 2 && pos_gain != 0.0) ev += pos_gain * cp_i[2];
        ev = stance ? ev : nl.u[3 * i + a];
        slots.store(pc, a * NX, stance ? vqa[a] : 0.0, (a == 2 && stance) ? vt2 : 0.0, stance ? bh[a] : 0.0,
                    (!stance && ln == 3 * i + a) ? 1.0 : 0.0, stance ? vja[a] : 0.0);
        e_mine = (ln == row + a) ? ev : e_mine;
        eq_sse += ev * ev;
      }
      {                                 // swing: normal velocity (NormalVelocityConstraintCppAd); stance: no fourth row (stores go to the dump)
        double ev = cv_i[2] - nl.zdref[i];
        if (pos_gain != 0.0) ev += pos_gain * (cp_i[2] - nl.zref[i]);
        double* p3[NS];
        for (int k = 0; k < NS; ++k) p3[k] = stance ? slots.dump : pc[k];
        slots.store(p3, 3 * NX, vqa[2], vt2, bh[2], 0.0, vja[2]);
        e_mine = (!stance && ln == row + 3) ? ev : e_mine;
        eq_sse = stance ? eq_sse : eq_sse + ev * ev;
      }
      row += stance ? 3 : 4;
      LIN_ROW_FENCE();
    }
    nc = row;
    (o.e + o.s * (kMaxEqRows))[ln] = e_mine;      // all kMaxEqRows = LPN entries: zeros beyond nc
  }

  LFPROF(2);
  // =========================== second RK2 stage ===========================
  LaneEval e2;
  double v2t = 0.0;                  // base linear velocity component ln of the second stage (translation role)
  if constexpr (!ILQR) {
    lds_wave_sync();                 // the constraint rows have read the swing references, whose storage becomes xh2
    if (ln < 6) nl.xh2[ln] = xh[ln] + dt * nl.fh[0][ln];
    if (ln < 3) nl.xh2[6 + ln] = pb[ln] + dt * nl.vlin[0][ln];
    const double* xh2 = nl.xh2;      // published by the lds_wave_sync at the top of eval_lane
    const double qg2 = qg + dt * e1.vg;
    LaneKin<NJ> kin2;
    eval_lane<NJ, true, true, NL, LinFastShared<NJ>, C>(md, sh, nl, 1, lb, path, g, xh2, qg2, ujg, e2, kin2);
    LIN_ROW_FENCE();           // keeps the loads of the combination phase (parked columns, stage-one blocks) out of the evaluation's registers
    lds_wave_sync();
    v2t = tr ? nl.vlin[1][ln] : 0.0;
  }
  LFPROF(3);
  // A2[rows 3..11][x columns 0..11] to LDS (a2 shares storage with the chain tables, dead now): columns 0..5 are the
  // momentum columns of lanes 0..5, columns 6..11 the q columns of lanes 0..5
  lds_wave_sync();
  if constexpr (!ILQR) {
    if (ln < 6)
      for (int r = 0; r < 9; ++r) nl.a2[r][ln] = momentum_col(nl.X12[1], nl.X22[1], imt, mass_total, ln, r);
    if (g < 6)
      for (int r = 0; r < 9; ++r) nl.a2[r][6 + g] = e2.ar_q[r];
    if (tr)
      for (int r = 0; r < 9; ++r) nl.a2[r][6 + ln] = 0.0;      // nothing depends on the base position
  }
  lds_wave_sync();
  // rows of A and B;  A = I + dt/2 (A1 + A2 + dt A2 A1),  B = dt/2 (B1 + B2 + dt A2 B1)
  double c1q[9], c1h[9], c1f[9], c1j[9];
  if constexpr (NL::kPark) {
    for (int rr = 0; rr < 9; ++rr) c1q[rr] = nl.park[rr][ln];
    for (int rr = 0; rr < 3; ++rr) c1j[rr] = 0.0;
    for (int rr = 3; rr < 9; ++rr) c1j[rr] = is_joint ? nl.park[9 + rr - 3][ln] : 0.0;
  } else {
    const double* pk = o.park + o.s * (15 * LPN);
    for (int rr = 0; rr < 9; ++rr) c1q[rr] = pk[rr * LPN + ln];
    for (int rr = 0; rr < 9; ++rr) c1j[rr] = 0.0;
    if (is_joint) for (int rr = 3; rr < 9; ++rr) c1j[rr] = pk[9 * LPN + (rr - 3) * LPN + ln];
  }
  for (int rr = 0; rr < 9; ++rr) {
    c1h[rr] = momentum_col(nl.X12[0], nl.X22[0], imt, mass_total, ln, rr);
    c1f[rr] = force_col(nl.cps[0], nl.com[0], imt, ln, rr);
  }
  {
    double* pab[NS];
    slots.pointers(o.A + o.s * (NX * NX), o.B + o.s * (NX * NU), pab);
#pragma unroll
    for (int r = MAT ? 0 : 3; r < (MAT ? NX : 12); ++r) {      // fused mode: the structural rows (0..2, 12..) are regenerated downstream
      double aq, ah_, bf, bj;
      if (r < 3 || r >= 12) {
        aq = (r == 6 + g) ? 1.0 : 0.0;
        ah_ = (r == ln) ? 1.0 : 0.0;
        bf = (r < 3 && (ln % 3) == r) ? dt * imt : 0.0;
        bj = (r >= 12 && r == 12 + g - 6) ? dt : 0.0;
      } else if constexpr (ILQR) {       // A = I + dt A_c, B = dt B_c: the stage-one columns are the model
        const int rr = r - 3;
        aq = ((r == 6 + g) ? 1.0 : 0.0) + dt * c1q[rr];
        ah_ = ((r == ln) ? 1.0 : 0.0) + dt * c1h[rr];
        bf = dt * c1f[rr];
        bj = dt * c1j[rr];
      } else {
        const int rr = r - 3;
        double sq = 0.0, sh = 0.0, sf = 0.0, sj = 0.0;
        for (int l = 0; l < 9; ++l) {
          const double a = nl.a2[rr][3 + l];
          sq += a * c1q[l]; sh += a * c1h[l]; sf += a * c1f[l]; sj += a * c1j[l];
        }
        // B1 rows 0..2 are (1/m) on the force components, rows 12.. are the identity on the joint velocities
        sf += nl.a2[rr][ln % 3] * imt;
        sj += e2.ar_q[rr];
        const double e2h = momentum_col(nl.X12[1], nl.X22[1], imt, mass_total, ln, rr);
        const double e2f = force_col(nl.cps[1], nl.com[1], imt, ln, rr);
        const double e2j = rr < 3 ? 0.0 : e2.br_j[rr - 3];
        aq = ((r == 6 + g) ? 1.0 : 0.0) + hdt * (c1q[rr] + e2.ar_q[rr] + dt * sq);
        ah_ = ((r == ln) ? 1.0 : 0.0) + hdt * (c1h[rr] + e2h + dt * sh);
        bf = hdt * (c1f[rr] + e2f + dt * sf);
        bj = hdt * (c1j[rr] + e2j + dt * sj);
      }
      // base translation (packed layout, lanes 0..2): identity column
      slots.store(pab, r * NX, aq, (r == 6 + ln) ? 1.0 : 0.0, ah_, bf, bj);
      LIN_ROW_FENCE();
    }
  }
  // b = x + dt/2 (f1 + f2) - x_next
  double dyn_sse = 0.0;
  if (g < G) {
    // the lane's own q and first-stage rate come back from LDS (nl.x, the park): two values less alive across the second evaluation
    const double v1g = NL::kPark ? nl.park[15][ln] : e1.vg;
    const double bb = ILQR ? 0.0 : nl.x[6 + g] + hdt * v1g + hdt * e2.vg - xn_q;
    (o.b + o.s * (NX))[6 + g] = bb;
    dyn_sse += bb * bb;
  }
  if (ln < 6) {
    const double bb = ILQR ? 0.0 : (ln < 6 ? xh[ln] : 0.0) + hdt * nl.fh[0][ln] + hdt * nl.fh[1][ln] - xn_h;
    (o.b + o.s * (NX))[ln] = bb;
    dyn_sse += bb * bb;
  }
  if (tr) {                          // base position: the same expression as a coordinate lane (q + dt/2 v1 + dt/2 v2 - x_next)
    const double bb = ILQR ? 0.0 : pb[ln] + hdt * nl.vlin[0][ln] + hdt * v2t - xn_t;
    (o.b + o.s * (NX))[6 + ln] = bb;
    dyn_sse += bb * bb;
  }
  LFPROF(4);
  // =========================== cost ===========================
  lds_wave_sync();   // a2 is dead: its storage becomes dx / du; the twist tables are dead: their storage becomes the cone terms
  if (ln < kNumContacts && stance_flag(mode, ln)) cone_terms(sc, &nl.u[3 * ln], true, nl.cone[ln]);
  const double pt = tr ? pb[ln] : 0.0;       // read before dx / du overwrite nothing of x (x lives in its own array) - kept for symmetry
  if (g < G) nl.dx[6 + g] = nl.x[6 + g] - xr_q;
  if (tr) nl.dx[6 + ln] = pt - xr_t;
  if (ln < 6) nl.dx[ln] = (ln < 6 ? xh[ln] : 0.0) - xr_h;
  if (ln < 12) nl.du[ln] = nl.u[ln] - nominal_input(sc, mode, ln);
  if (is_joint) nl.du[12 + g - 6] = ujg;
  lds_wave_sync();
  double shift = 0.0;
  for (int i = 0; i < kNumContacts; ++i)
    if (stance_flag(mode, i)) shift += -nl.cone[i][2] * sc.cone_shift;
  double cost = 0.0;
  {
    const int cq = 6 + g, ch = ln, cf = ln, cj = 12 + g - 6, ct = 6 + ln;
    // lanes without a role read row / column 0 of the weights (any valid address) and their sums are dropped
    const int cqs = g < G ? cq : 0, chs = ln < 6 ? ch : 0, cfs = ln < 12 ? cf : 0, cjs = is_joint ? cj : 0, cts = tr ? ct : 0;
    const bool cf_stance = ln < 12 && stance_flag(mode, cfs / 3);
    double accq = 0.0, acch = 0.0, accf = 0.0, accj = 0.0, acct = 0.0;
    double* pqr[NS];
    if constexpr (MAT) slots.pointers(o.Q + o.s * (NX * NX), o.R + o.s * (NU * NU), pqr);
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const double dxr = nl.dx[r], dur = nl.du[r];
      // gradient entries use row c of the weight (as the reference kernel), the written element is (r, c)
      accq += sh.Q[cqs * NX + r] * dxr;
      if constexpr (PACKED) acct += sh.Q[cts * NX + r] * dxr;
      acch += sh.Q[chs * NX + r] * dxr;
      accf += sh.R[cfs * NU + r] * dur;
      accj += sh.R[cjs * NU + r] * dur;
      if constexpr (MAT) {
        double w = sh.R[r * NU + cfs];
        if (r == cf) w += shift;
        if (r < 12) {                    // the 3 x 3 block of a stance contact carries the cone Hessian
          const double* cn = nl.cone[r / 3];
          const int a = r % 3, b2 = cfs % 3;
          const int lo = a < b2 ? a : b2, hi = a < b2 ? b2 : a;
          const int sidx = lo == 0 ? hi : (lo == 1 ? 2 + hi : 5);
          const double wc = w + (cn[3] * cn[4 + a] * cn[4 + b2] + cn[2] * cn[7 + sidx]);
          w = (cf_stance && r / 3 == cfs / 3) ? wc : w;
        }
        const double isf = (ILQR && r == cf) ? o.ilqr_shift : 0.0, isj = (ILQR && r == cj) ? o.ilqr_shift : 0.0;
        slots.store(pqr, r * NX, dt * (sh.Q[r * NX + cqs] + (r == cq ? shift : 0.0)), dt * (sh.Q[r * NX + cts] + (r == ct ? shift : 0.0)),
                    dt * (sh.Q[r * NX + chs] + (r == ch ? shift : 0.0)), dt * w + isf, dt * (sh.R[r * NU + cjs] + (r == cj ? shift : 0.0)) + isj);
      }
      // the sums are needed behind the loop only: left alone, their multiply-adds sink there and the 88 weights they read wait in registers
      asm volatile("" : "+v"(accq), "+v"(acch), "+v"(accf), "+v"(accj));
      if constexpr (PACKED) asm volatile("" : "+v"(acct));
      LIN_ROW_FENCE();
    }
    // the node-dependent part of R in compact form: the three rows of the own 3 x 3 force block (lanes 0..11)
    if (ln < 12) {
      const int blk = cf / 3, b2 = cf % 3;
      const double* cn = nl.cone[blk];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int r = 3 * blk + a;
        double w = sh.R[r * NU + cf];
        if (r == cf) w += shift;
        if (cf_stance) {
          const int lo = a < b2 ? a : b2, hi = a < b2 ? b2 : a;
          const int sidx = lo == 0 ? hi : (lo == 1 ? 2 + hi : 5);
          w += cn[3] * cn[4 + a] * cn[4 + b2] + cn[2] * cn[7 + sidx];
        }
        (o.qrd + o.s * kQrdStride)[1 + 3 * cf + a] = dt * w + ((ILQR && r == cf) ? o.ilqr_shift : 0.0);
      }
    }
    if (g < G) { (o.q + o.s * (NX))[cq] = dt * accq; cost += 0.5 * nl.dx[cq] * accq; }
    if (tr) { (o.q + o.s * (NX))[ct] = dt * acct; cost += 0.5 * nl.dx[ct] * acct; }
    if (ln < 6) { (o.q + o.s * (NX))[ch] = dt * acch; cost += 0.5 * nl.dx[ch] * acch; }
    if (ln < 12) {
      cost += 0.5 * nl.du[cf] * accf;
      if (stance_flag(mode, cf / 3)) accf += nl.cone[cf / 3][2] * nl.cone[cf / 3][4 + cf % 3];
      (o.r + o.s * (NU))[cf] = dt * accf;
    }
    if (is_joint) { (o.r + o.s * (NU))[cj] = dt * accj; cost += 0.5 * nl.du[cj] * accj; }
    if (ln < kNumContacts && stance_flag(mode, ln)) cost += nl.cone[ln][1];
  }
  // P (cost cross term) is structurally zero for this problem: the buffer is zero-filled once at allocation and never written
  cost = node_allreduce_add<LPN>(cost);
  dyn_sse = node_allreduce_add<LPN>(dyn_sse);
  if (ln == 0) {
    (o.qrd + o.s * kQrdStride)[0] = shift;
    if constexpr (ILQR) (o.qrd + o.s * kQrdStride)[kQrdRShift] = shift + o.ilqr_shift / dt;      // dt (R_ii + this) = dt (R_ii + shift) + DIAGONAL_SHIFT up to rounding
    if constexpr (MAT) (o.c + o.s * (1))[0] = dt * cost;
    (o.nc + o.s * (1))[0] = nc;
    (o.perf + o.s * (3))[0] = dt * cost; (o.perf + o.s * (3))[1] = dt * dyn_sse; (o.perf + o.s * (3))[2] = dt * eq_sse;
  }
  LFPROF(5);
#ifdef BPMPC_LINFAST_PROFILE
  if (o.prof && ln == 0) {
    lf_t[6] = clock64() - lf_c0; lf_t[7] = wall_clock64() - lf_wall0;
    for (int i = 0; i < 8; ++i) o.prof[i] = (double)lf_t[i];
  }
#endif
#ifdef BPMPC_EVAL_PROFILE
  if (o.prof && ln == 0)
    for (int i = 0; i < 8; ++i) o.prof[i] = (double)evacc[i];
#endif
}

// Value-only metrics of x + alpha dx, u + alpha du at one node for the filter line search (same lane layout as
// linearize_fast; reference version: trial_node in linesearch.h).
// EQV: also store the values of the active equality rows (registration order zeroForce_i, zeroVelocity_i, normalVelocity_i per contact,
// BipedalRobotInterface.cpp:187-191) to eqv[0..nc): the solution metrics of the solver observers (bpmpc_solver_constraint_values).
// The lane's entries of the iterate and of the step, loaded by the kernel wrapper before the model block is staged (as LinFastPre)
struct TrialPre {
  LinFastPre x, d;      // of (x, x_next, u, x_ref) and of (dx, dx_next, du, -)
};
template <class Cfg>
__device__ __forceinline__ TrialPre trial_preload(const double* x, const double* xnext, const double* u, const double* xref, const double* dx,
                                                  const double* dxn, const double* du, int ln, const double* aux = nullptr) {
  TrialPre p;
  p.x = linearize_preload<Cfg>(x, xnext, u, xref, ln, aux);
  p.d = linearize_preload<Cfg>(dx, dxn, du, dx /* unused */, ln);
  return p;
}
template <int NJ, class Cfg = LinFastCfg<NJ, true>, bool EQV = false, class NL = LinFastNodeLds<NJ, false, Cfg::CHAIN>>
__device__ __forceinline__ void trial_fast(const DeviceModel& md, const LinFastShared<NJ, false>& sh, NL& nl, bool valid,
                                           const NodeInputs& in, double alpha, const double* dx, const double* du, const double* dxn,
                                           double* perf, int ln, double* eqv = nullptr, const TrialPre* pre = nullptr) {
  using C = Cfg;
  constexpr int G = C::G, NX = C::NX, NU = C::NU, LPN = C::LPN, G0 = C::G0;
  const int g = ln + G0;
  const bool tr = G0 > 0 && ln < 3;    // see linearize_fast
  if (!valid) return;
  if (in.kind == 1) {
    double d2 = 0.0;
    for (int idx = ln; idx < NX; idx += LPN) {
      const double d = (in.x[idx] + alpha * dx[idx]) - (in.xnext[idx] + alpha * dxn[idx]);
      d2 += d * d;
    }
    d2 = node_allreduce_add<LPN>(d2);
    if (ln == 0) { perf[0] = 0.0; perf[1] = d2; perf[2] = 0.0; }
    return;
  }
  const bool is_joint = g >= 6 && g < G;
  double dt = in.dt;
  const int mode = in.mode;
  const LinFastScalars& sc = sh.sc;
  double xn_q, xn_h, xr_q, xr_h, xn_t, xr_t;
  if (pre) {                            // (compile-time after inlining) the same expressions on values that arrived before the model block
    static_assert(LPN == 16 && NX <= 32, "two elements of x and of u per lane");
    dt = row_bcast_f64<0>(pre->x.aux);                                  // the node record (kNodeAux): dt, zref[4], zdref[4]
    if (ln >= 1 && ln < 1 + kNumContacts) nl.zref[ln - 1] = pre->x.aux;
    if (ln >= 1 + kNumContacts && ln < 1 + 2 * kNumContacts) nl.zdref[ln - 1 - kNumContacts] = pre->x.aux;
    nl.x[ln] = pre->x.x0 + alpha * pre->d.x0; nl.u[ln] = pre->x.u0 + alpha * pre->d.u0;
    if (ln + 16 < NX) { nl.x[ln + 16] = pre->x.x1 + alpha * pre->d.x1; nl.u[ln + 16] = pre->x.u1 + alpha * pre->d.u1; }
    xn_q = g < G ? pre->x.xn_q + alpha * pre->d.xn_q : 0.0; xn_h = ln < 6 ? pre->x.xn_h + alpha * pre->d.xn_h : 0.0;
    xn_t = tr ? pre->x.xn_t + alpha * pre->d.xn_t : 0.0;
    xr_q = pre->x.xr_q; xr_h = pre->x.xr_h; xr_t = pre->x.xr_t;
  } else {
    for (int idx = ln; idx < NX; idx += LPN) {
      nl.x[idx] = in.x[idx] + alpha * dx[idx];
      nl.u[idx] = in.u[idx] + alpha * du[idx];
    }
    xn_q = g < G ? in.xnext[6 + g] + alpha * dxn[6 + g] : 0.0; xn_h = ln < 6 ? in.xnext[ln] + alpha * dxn[ln] : 0.0;
    xr_q = g < G ? in.xref[6 + g] : 0.0; xr_h = ln < 6 ? in.xref[ln] : 0.0;
    xn_t = tr ? in.xnext[6 + ln] + alpha * dxn[6 + ln] : 0.0; xr_t = tr ? in.xref[6 + ln] : 0.0;
  }
  const double hdt = 0.5 * dt;
  if constexpr (NL::kLate) {
    static_assert(LPN == 16, "one slot per lane of the node");
    nl.late[0][ln] = xn_q; nl.late[1][ln] = xn_h; nl.late[2][ln] = xn_t; nl.late[3][ln] = xr_q; nl.late[4][ln] = xr_h; nl.late[5][ln] = xr_t;
  }
  LaneBody lb;
  {
    const int body = (g >= 5 && g < G) ? g - 5 : 0;
    lb.body = body;
    lb.depth = sh.depth[body];
    lb.subtree = sh.subtree[body];
  }
  const int* path = sh.path[lb.body];
  lds_wave_sync();
  const double* xh = nl.x;             // normalised momentum and base position, read from LDS where they are used
  const double* pb = nl.x + 6;
  const double qg = g < G ? nl.x[6 + g] : 0.0;
  const double ujg = is_joint ? nl.u[12 + g - 6] : 0.0;
  LaneEval e1;
  LaneKin<NJ> kin;
  eval_lane<NJ, false, true, NL, LinFastShared<NJ, false>, C>(md, sh, nl, 0, lb, path, g, xh, qg, ujg, e1, kin);
  const double v1t = ln == 0 ? kin.vb[0] : (ln == 1 ? kin.vb[1] : kin.vb[2]);
  if (g >= 5 && g < G)
    for (int i = 0; i < kNumContacts; ++i)
      if (sc.contact_body[i] == lb.body) {
        const double r[3] = {nl.cpos_v[i][0] - kin.og[0], nl.cpos_v[i][1] - kin.og[1], nl.cpos_v[i][2] - kin.og[2]};
        double t[3];
        cross3(kin.omg, r, t);
        for (int a = 0; a < 3; ++a) nl.cvel_v[i][a] = kin.vog[a] + t[a];
      }
  double cone_own[4] = {0.0, 0.0, 0.0, 0.0};          // h, barrier value, first and second derivative of this lane's contact
  if (ln < kNumContacts && stance_flag(mode, ln)) cone_terms(sc, &nl.u[3 * ln], false, cone_own);
  lds_wave_sync();
  double eq_sse = 0.0;
  int row = 0;
  for (int i = 0; i < kNumContacts; ++i) {
    const double cz = nl.cpos_v[i][2];
    if (stance_flag(mode, i)) {
      for (int a = 0; a < 3; ++a) {
        double ev = nl.cvel_v[i][a];
        if (sc.pos_gain != 0.0 && a == 2) ev += sc.pos_gain * cz;
        eq_sse += ev * ev;
        if constexpr (EQV) { if (ln == 0) eqv[row] = ev; ++row; }
      }
    } else {
      for (int a = 0; a < 3; ++a) { const double ev = nl.u[3 * i + a]; eq_sse += ev * ev; if constexpr (EQV) { if (ln == 0) eqv[row] = ev; ++row; } }
      double ev = nl.cvel_v[i][2] - (pre ? nl.zdref[i] : in.zdref[i]);      // (without the preload - back-tracking tail, observers - straight from the grid tables)
      if (sc.pos_gain != 0.0) ev += sc.pos_gain * (cz - (pre ? nl.zref[i] : in.zref[i]));
      eq_sse += ev * ev;
      if constexpr (EQV) { if (ln == 0) eqv[row] = ev; ++row; }
    }
  }
  (void)row;
  double cone_pen = 0.0;
  if (ln < kNumContacts && stance_flag(mode, ln)) cone_pen = cone_own[1];
  LaneEval e2;
  double v2t;
  const double f1 = lane_pick6(e1.fh, ln);          // the lane's own row of the first stage: the other five need not live through the second evaluation
  {
    if (ln < 6) nl.xh2[ln] = xh[ln] + dt * f1;
    if (ln < 3) nl.xh2[6 + ln] = pb[ln] + dt * v1t;
    const double* xh2 = nl.xh2;      // published by the lds_wave_sync at the top of eval_lane
    const double qg2 = qg + dt * e1.vg;
    LaneKin<NJ> kin2;
    eval_lane<NJ, false, false, NL, LinFastShared<NJ, false>, C>(md, sh, nl, 1, lb, path, g, xh2, qg2, ujg, e2, kin2);
    v2t = ln == 0 ? kin2.vb[0] : (ln == 1 ? kin2.vb[1] : kin2.vb[2]);
  }
  if constexpr (NL::kLate) {           // (the lane's own slots: no barrier)
    typedef const volatile __attribute__((address_space(3))) double* late_p;
    xn_q = *(late_p)&nl.late[0][ln]; xn_h = *(late_p)&nl.late[1][ln]; xn_t = *(late_p)&nl.late[2][ln];
    xr_q = *(late_p)&nl.late[3][ln]; xr_h = *(late_p)&nl.late[4][ln]; xr_t = *(late_p)&nl.late[5][ln];
  }
  double dyn_sse = 0.0;
  if (g < G) {
    const double bb = qg + hdt * e1.vg + hdt * e2.vg - xn_q;
    dyn_sse += bb * bb;
  }
  if (ln < 6) {
    const double f2 = lane_pick6(e2.fh, ln);
    const double bb = (ln < 6 ? xh[ln] : 0.0) + hdt * f1 + hdt * f2 - xn_h;
    dyn_sse += bb * bb;
  }
  const double pt = tr ? pb[ln] : 0.0;
  if (tr) {
    const double bb = pt + hdt * v1t + hdt * v2t - xn_t;
    dyn_sse += bb * bb;
  }
  lds_wave_sync();   // the chain tables are dead: their storage becomes dx / du
  if (g < G) nl.dx[6 + g] = qg - xr_q;
  if (tr) nl.dx[6 + ln] = pt - xr_t;
  if (ln < 6) nl.dx[ln] = (ln < 6 ? xh[ln] : 0.0) - xr_h;
  if (ln < 12) nl.du[ln] = nl.u[ln] - nominal_input(sc, mode, ln);
  if (is_joint) nl.du[12 + g - 6] = ujg;
  lds_wave_sync();
  double cost = cone_pen;
  {
    const int cq = 6 + g, ch = ln, cf = ln, cj = 12 + g - 6, ct = 6 + ln;
    double accq = 0.0, acch = 0.0, accf = 0.0, accj = 0.0, acct = 0.0;
    for (int r = 0; r < NX; ++r) {
      const double dxr = nl.dx[r], dur = nl.du[r];
      if (g < G) accq += md.Q[cq * NX + r] * dxr;       // cost weights from global memory (cache resident): this kernel stores next to nothing,
      if (tr) acct += md.Q[ct * NX + r] * dxr;
      if (ln < 6) acch += md.Q[ch * NX + r] * dxr;      // so the loads never queue behind stores, and the LDS they would take buys a third
      if (ln < 12) accf += md.R[cf * NU + r] * dur;     // workgroup per CU
      if (is_joint) accj += md.R[cj * NU + r] * dur;
    }
    if (g < G) cost += 0.5 * nl.dx[cq] * accq;
    if (tr) cost += 0.5 * nl.dx[ct] * acct;
    if (ln < 6) cost += 0.5 * nl.dx[ch] * acch;
    if (ln < 12) cost += 0.5 * nl.du[cf] * accf;
    if (is_joint) cost += 0.5 * nl.du[cj] * accj;
  }
  cost = node_allreduce_add<LPN>(cost);
  dyn_sse = node_allreduce_add<LPN>(dyn_sse);
  if (ln == 0) { perf[0] = dt * cost; perf[1] = dt * dyn_sse; perf[2] = dt * eq_sse; }
}


}  // namespace bpmpc
