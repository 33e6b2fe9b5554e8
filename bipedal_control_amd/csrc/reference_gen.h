// Host pre-pass of one MPC solve: mode schedule (gait tiling), swing-height splines, shooting grid with
// event nodes and target trajectories.  Restates, in this engine's own data layout, the reference's
//   GaitSchedule                     ocs2_bipedal_robot/src/gait/GaitSchedule.cpp:40-137
//   SwingTrajectoryPlanner/SplineCpg/CubicSpline   src/foot_planner/*.cpp
//   SwitchedModelReferenceManager    src/reference_manager/SwitchedModelReferenceManager.cpp:55-69
//   cmdVel/goal -> TargetTrajectories bipedal_controllers/src/TargetTrajectoriesPublisher.cpp:30-99
// and, [OCS2-upstream], timeDiscretizationWithEvents / getIntervalStart / getIntervalEnd / ModeSchedule::modeAtTime.
#pragma once
#include <vector>

#include "robot_model.h"

namespace bpmpc {

// lower_bound lookup: a time exactly on an event belongs to the phase before it
int phase_index(const std::vector<double>& event_times, double t);
inline int mode_at(const ModeSchedule& s, double t) { return s.modes[phase_index(s.event_times, t)]; }
inline bool contact_flag(int mode, int contact) { return contact < 2 ? (mode == LF || mode == STANCE) : (mode == RF || mode == STANCE); }

class GaitSchedule {
 public:
  GaitSchedule(ModeSchedule initial, ModeTemplate tmpl, double phase_transition_stance_time);
  void insert_template(const ModeTemplate& tmpl, double start_time, double final_time);
  // mutates the stored schedule exactly like the reference (old events dropped, template tiled up to `upper`)
  const ModeSchedule& mode_schedule(double lower, double upper);
  const ModeSchedule& current() const { return schedule_; }

 private:
  void tile(double start_time, double final_time);
  ModeSchedule schedule_;
  ModeTemplate template_;
  double stance_time_;
};

// One cubic height segment z(t) on [t0, t1]
struct HeightSegment {
  double t0 = 0, dt = 1, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  static HeightSegment through(double ta, double za, double va, double tb, double zb, double vb);
  double position(double t) const;
  double velocity(double t) const;
};
struct SwingProfile {  // lift-off -> apex -> touch-down, two cubic segments
  double mid_time = 0;
  HeightSegment up, down;
  double position(double t) const { return t < mid_time ? up.position(t) : down.position(t); }
  double velocity(double t) const { return t < mid_time ? up.velocity(t) : down.velocity(t); }
};

class SwingPlanner {
 public:
  explicit SwingPlanner(SwingConfig cfg) : cfg_(cfg) {}
  // throws std::runtime_error when a swing phase has no lift-off / touch-down inside the schedule
  void update(const ModeSchedule& schedule, double terrain_height = 0.0);
  double z_position(int contact, double t) const;
  double z_velocity(int contact, double t) const;

 private:
  SwingConfig cfg_;
  std::vector<double> events_;
  std::vector<SwingProfile> profiles_[kNumContacts];
};

enum NodeEvent { kNone = 0, kPreEvent = 1, kPostEvent = 2 };
struct GridNode { double time; int event; };
std::vector<GridNode> shooting_grid(double t0, double tf, double dt, const std::vector<double>& event_times, double dt_min = 1e-8);
double interval_start(const GridNode& n);
double interval_end(const GridNode& n);

// Per-interval data the device consumes (k = 0..N-1), shared by every problem that uses the same (t0, tf, schedule).
struct NodeTable {
  int N = 0;
  std::vector<double> node_time;   // N+1 grid times
  std::vector<int> kind;           // 0 intermediate, 1 event (pre-event node: identity jump, no input)
  std::vector<double> start, dt;   // interval start (mode / reference look-up time) and duration
  std::vector<int> mode;
  std::vector<double> zref, zdref; // N x 4 swing-height references at `start`
};
NodeTable build_node_table(const RobotModel& m, double t0, double tf, double dt, const ModeSchedule& schedule, const SwingPlanner& planner);

// two-point target trajectories (times[2], states[2*nx])
void cmd_vel_to_targets(const RobotModel& m, const double cmd_vel[4], double t_now, const double* x_now, double time_to_target,
                        double times[2], double* states);
void goal_to_targets(const RobotModel& m, const double goal[4], double t_now, const double* x_now, double times[2], double* states);
// clamped piecewise-linear interpolation of a target trajectory
void interpolate_targets(int n_pts, const double* times, const double* states, int nx, double t, double* out);

}  // namespace bpmpc
