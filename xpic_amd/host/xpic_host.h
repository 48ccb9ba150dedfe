// xpic_host.h -- C++ mirror of xpic's Simulation / Particles / Command / Diagnostic interfaces on top of the
// C ABI (include/xpic_hip.h).  Same names, argument meaning and error behaviour as the reference so that the
// backends select like any other `"Simulation"` of the JSON config:
//   interfaces::Simulation   src/interfaces/simulation.h:14-83, simulation.cpp:16-182
//   interfaces::Particles    src/interfaces/particles.h:11-78
//   interfaces::Command      src/interfaces/command.h:14-31
//   interfaces::Diagnostic   src/interfaces/diagnostic.h
//   Configuration            src/utils/configuration.h, configuration.cpp:19-130
//   Builder::parse_value     src/interfaces/builder.cpp:54-81
// PetscErrorCode is `int` here (0 = success); PetscCall is the early-return macro XCALL.
#pragma once

#include <fstream>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/xpic_hip.h"
#include "json.h"

using PetscErrorCode = int;
using PetscInt = int;
using PetscReal = double;

#define XCALL(expr)            \
  do {                         \
    int xrc_ = (expr);         \
    if (xrc_ != 0) return xrc_; \
  } while (0)

// ---- global geometry, as src/constants.h:10-28
extern PetscReal dx, dy, dz, dt;
extern PetscReal geom_x, geom_y, geom_z, geom_t;
extern PetscInt geom_nx, geom_ny, geom_nz, geom_nt;
extern PetscInt diagnose_period;
constexpr PetscReal mec2 = 511.0;

struct Vector3R {
  PetscReal data[3] = {0, 0, 0};
  PetscReal& operator[](int i) { return data[i]; }
  const PetscReal& operator[](int i) const { return data[i]; }
  PetscReal squared() const { return data[0] * data[0] + data[1] * data[1] + data[2] * data[2]; }
};

struct Point { // src/interfaces/point.h:7-35
  Vector3R r, p;
};

struct SortParameters { // src/interfaces/sort_parameters.h:7-19
  std::string sort_name;
  PetscInt Np = 1;
  PetscReal n = 0, q = 0, m = 1;
  PetscReal px = 0, py = 0, pz = 0;
  PetscReal Tx = 0, Ty = 0, Tz = 0;
};

class Configuration { // src/utils/configuration.h
public:
  using json_t = xjson::Value;
  static const Configuration& get();
  static void init(const std::string& config_path);
  static void overwrite(json_t&& json);
  static void set_out_dir(const std::string& dir);
  static bool is_loaded_from_backup(); // configuration.cpp:76-86
  std::string config_path;
  json_t json;
  std::string out_dir;

private:
  static Configuration config;
};
#define CONFIG() Configuration::get()

struct World { // src/utils/world.h:11-61 (the DMDA is the device-side grid of the context)
  PetscErrorCode initialize();
  xpic_geometry geom{};
  static void set_geometry(PetscReal gx, PetscReal gy, PetscReal gz, PetscReal gt, PetscReal dx_, PetscReal dy_,
    PetscReal dz_, PetscReal dt_, PetscReal dtp);
};

namespace interfaces {

class Simulation;

struct Builder { // src/interfaces/builder.cpp:22-113
  static Vector3R parse_vector(const Configuration::json_t& info, const std::string& name);
  static PetscReal parse_value(const Configuration::json_t& value);
};

class Particles { // src/interfaces/particles.h:11-78
public:
  Particles(Simulation& simulation, const SortParameters& parameters);
  virtual ~Particles() = default;

  const SortParameters parameters;
  int sort_id = -1; // handle of the species inside the device context

  /// Points pass the local-box test on the device; `is_added` answers immediately from the same FLOOR_STEP test.
  PetscErrorCode add_particle(const Point& point, bool* is_added = nullptr);
  PetscErrorCode flush();          // ships buffered points (xpic_sort_add_particles)
  PetscErrorCode update_cells();   // update_cells_seq / update_cells_mpi
  PetscErrorCode storage(std::vector<Point>& points, std::vector<int>& cell_of); // host mirror on demand
  PetscInt count();

  PetscReal q_m() const { return parameters.q / parameters.m; }
  PetscReal n_Np() const { return parameters.n / parameters.Np; }
  PetscReal qn_Np() const { return parameters.q * parameters.n / parameters.Np; }

protected:
  Simulation& simulation_;
  std::vector<double> pending_;
};

class Command { // src/interfaces/command.h:14-31
public:
  virtual ~Command() = default;
  virtual PetscErrorCode finalize() { return 0; }
  virtual PetscErrorCode execute(PetscInt timestep) = 0;
};

class Diagnostic {
public:
  virtual ~Diagnostic() = default;
  virtual PetscErrorCode finalize() { return 0; }
  virtual PetscErrorCode diagnose(PetscInt timestep) = 0;
};

class Simulation { // src/interfaces/simulation.h:14-83
public:
  Simulation() = default;
  virtual ~Simulation();

  PetscInt start = 0;
  World world;
  xpic_ctx* ctx = nullptr; // plays the role of `DM da` + the owned Vec/Mat/KSP objects

  std::vector<std::shared_ptr<Particles>> particles_;

  PetscErrorCode initialize();
  PetscErrorCode calculate();
  virtual PetscErrorCode finalize();

  /// "E" | "B" | "B0" -> field id of the C ABI (get_named_vector, simulation.cpp:135-143)
  int get_named_vector(const std::string& name) const;
  Particles& get_named_particles(const std::string& name);

  virtual int scheme() const = 0;

protected:
  PetscErrorCode init_particles(); // src/interfaces/simulation.tpp:7-79
  virtual PetscErrorCode initialize_implementation();
  virtual PetscErrorCode timestep_implementation(PetscInt timestep);

  std::vector<std::unique_ptr<Command>> step_presets_;
  std::vector<std::unique_ptr<Diagnostic>> diagnostics_;
  friend class ::Configuration;
};

}  // namespace interfaces

namespace basic {
class Simulation final : public interfaces::Simulation { // src/impls/basic/simulation.h
public:
  int scheme() const override { return XPIC_BASIC; }
};
}  // namespace basic

namespace ecsim {
class Simulation : public interfaces::Simulation { // src/impls/ecsim/simulation.h
public:
  int scheme() const override { return XPIC_ECSIM; }
  PetscInt last_ksp_iterations = 0;

protected:
  PetscErrorCode timestep_implementation(PetscInt timestep) override;
};
}  // namespace ecsim

namespace ecsimcorr {
class Simulation final : public ecsim::Simulation { // src/impls/ecsimcorr/simulation.h
public:
  int scheme() const override { return XPIC_ECSIMCORR; }
};
}  // namespace ecsimcorr

/// @returns Concrete simulation using `config` specification (src/interfaces/simulation.cpp:160-182).
std::unique_ptr<interfaces::Simulation> build_simulation();

// ---- commands (src/commands/set_particles.cpp, set_magnetic_field.cpp, builders/*)
using CoordinateGenerator = std::function<Vector3R()>;
using MomentumGenerator = std::function<Vector3R(const Vector3R&)>;

class SetParticles : public interfaces::Command {
public:
  SetParticles(interfaces::Particles& particles, PetscInt number_of_particles, CoordinateGenerator gc,
    MomentumGenerator gm);
  PetscErrorCode execute(PetscInt t) override;
  PetscInt added_particles = 0;
  PetscReal added_energy = 0;

private:
  interfaces::Particles& particles_;
  PetscInt number_of_particles_;
  CoordinateGenerator generate_coordinate_;
  MomentumGenerator generate_momentum_;
};

class SetMagneticField : public interfaces::Command {
public:
  SetMagneticField(interfaces::Simulation& sim, int field, int field_axpy, const Vector3R& uniform_value);
  PetscErrorCode execute(PetscInt t) override;

private:
  interfaces::Simulation& sim_;
  int field_, field_axpy_;
  Vector3R value_;
};

PetscErrorCode build_commands(interfaces::Simulation& simulation, const std::string& name,
  std::vector<std::unique_ptr<interfaces::Command>>& result);

// ---- diagnostics (src/diagnostics/utils/table_diagnostic.cpp, src/diagnostics/energy.cpp)
class TableDiagnostic : public interfaces::Diagnostic {
public:
  explicit TableDiagnostic(const std::string& filename);
  PetscErrorCode diagnose(PetscInt t) override;
  PetscErrorCode finalize() override { file_.close(); return 0; }
  virtual PetscErrorCode initialize() { return 0; }
  virtual PetscErrorCode add_columns(PetscInt /* t */) { return 0; }
  void add(PetscInt w, std::string title, const char* printf_fmt, double value, PetscInt pos = -1);
  void add_int(PetscInt w, std::string title, long value);

protected:
  void write_formatted(const std::vector<std::string>& container);
  std::string filename_;
  std::ofstream file_;
  std::vector<std::string> titles_, values_;
  bool initialized_ = false;
};

class Energy : public interfaces::Diagnostic {
public:
  explicit Energy(interfaces::Simulation& simulation);
  PetscErrorCode diagnose(PetscInt t) override;

protected:
  PetscErrorCode calculate();
  interfaces::Simulation& simulation;
  TableDiagnostic energy, energy_cons;
  PetscReal E = 0, E0 = 0, B = 0, B0 = 0, std_E = 0, std_B = 0;
  std::vector<PetscReal> K, K0, std_K;
};

class ChargeConservation : public TableDiagnostic { // src/diagnostics/charge_conservation.cpp:100-171
public:
  explicit ChargeConservation(interfaces::Simulation& simulation);
  PetscErrorCode initialize() override;
  PetscErrorCode add_columns(PetscInt t) override;

private:
  interfaces::Simulation& simulation;
};

class MomentumConservation : public TableDiagnostic { // src/diagnostics/momentum_conservation.cpp:7-131
public:
  explicit MomentumConservation(interfaces::Simulation& simulation);
  PetscErrorCode initialize() override;
  PetscErrorCode add_columns(PetscInt t) override;

private:
  PetscErrorCode calculate(); // P1, QE of every sort: the sums stay on the device (xpic_momentum)
  interfaces::Simulation& simulation;
  std::vector<Vector3R> P0, P1, QE;
};

PetscErrorCode build_diagnostics(interfaces::Simulation& simulation,
  std::vector<std::unique_ptr<interfaces::Diagnostic>>& result); // diagnostic_builder.cpp:16-75

// ---- float32 dumps (src/diagnostics/field_view.cpp, src/utils/mpi_binary_file.cpp:95-108): one file
// <out_dir>/<format_time(t)> per diagnose period, the region's values in C order [z][y][x][c] as float32
class FieldView : public interfaces::Diagnostic {
public:
  struct Region { // field_view.h:15-20, axes in x, y, z, component order
    PetscInt dim = 4, dof = 3;
    PetscInt start[4] = {0, 0, 0, 0};
    PetscInt size[4] = {0, 0, 0, 3};
  };
  FieldView(const std::string& out_dir, interfaces::Simulation& simulation, int field, const Region& region);
  PetscErrorCode diagnose(PetscInt t) override;
  static std::string format_time(PetscInt t); // src/interfaces/diagnostic.cpp:21-25

protected:
  virtual PetscErrorCode fetch(std::vector<double>& data); // the whole local array, [z][y][x][dof]
  interfaces::Simulation& simulation;
  std::string out_dir_;
  int field_;
  Region region_;
};

class DistributionMoment final : public FieldView { // src/diagnostics/distribution_moment.cpp, moment "density"
public:
  DistributionMoment(const std::string& out_dir, interfaces::Simulation& simulation, interfaces::Particles& particles,
    const Region& region);

protected:
  PetscErrorCode fetch(std::vector<double>& data) override;
  interfaces::Particles& particles_;
};

// ---- restart files (src/diagnostics/simulation_backup.cpp:30-160): <out_dir>/<t>/{E,B,B0} as PETSc binary Vecs
// (big endian: int32 VEC_FILE_CLASSID = 1211214, int32 n, n doubles in natural [z][y][x][c] order), particles as
// <sort_name> = raw big-endian doubles {r, p} per particle and <sort_name>.numparts = one big-endian int32
class SimulationBackup final : public interfaces::Diagnostic, public interfaces::Command {
public:
  SimulationBackup(const std::string& out_dir, PetscInt diagnose_period, interfaces::Simulation& simulation);
  PetscErrorCode diagnose(PetscInt t) override { return save(t); }
  PetscErrorCode execute(PetscInt t) override { return load(t); }
  PetscErrorCode finalize() override { return 0; }
  PetscErrorCode save(PetscInt t);
  PetscErrorCode load(PetscInt t);
  static constexpr PetscInt num_periods_being_kept = 2;

private:
  std::string out_dir_;
  PetscInt diagnose_period_;
  interfaces::Simulation& simulation;
};
