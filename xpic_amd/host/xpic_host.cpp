// xpic_host.cpp -- see xpic_host.h.  Host logic only; every grid/particle operation goes through the C ABI.
#include "xpic_host.h"

#include <sys/stat.h>

#include <cstdint>
#include <cstring>
#include <filesystem>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <iostream>
#include <random>
#include <sstream>
#include <stdexcept>

PetscReal dx = 0, dy = 0, dz = 0, dt = 0;
PetscReal geom_x = 0, geom_y = 0, geom_z = 0, geom_t = 0;
PetscInt geom_nx = 0, geom_ny = 0, geom_nz = 0, geom_nt = 0;
PetscInt diagnose_period = 0;

#define LOG(msg) (std::cout << msg << "\n")
#define ROUND_STEP(s, ds) static_cast<PetscInt>(std::round((s) / (ds))) /* src/utils/utils.h:77 */
#define FLOOR_STEP(s, ds) static_cast<PetscInt>(std::floor((s) / (ds))) /* :78 */

static int check(interfaces::Simulation* sim, int rc, const char* what)
{
  (void)sim;
  if (rc != 0) std::cerr << "xpic_hip error in " << what << ": " << xpic_last_error() << std::endl;
  return rc;
}
#define HIPCALL(expr) XCALL(check(nullptr, (expr), #expr))

static void make_dirs(const std::string& path)
{
  std::string cur;
  for (size_t i = 0; i <= path.size(); ++i) {
    if (i == path.size() || path[i] == '/') {
      if (!cur.empty()) mkdir(cur.c_str(), 0777);
    }
    if (i < path.size()) cur += path[i];
  }
}

// ---- Configuration (src/utils/configuration.cpp:7-24)
Configuration Configuration::config;
const Configuration& Configuration::get() { return config; }
void Configuration::overwrite(json_t&& json)
{
  config.json = std::move(json);
  config.out_dir = config.json.at("OutputDirectory").as_string();
}
void Configuration::init(const std::string& config_path)
{
  std::ifstream file(config_path);
  if (!file) throw std::runtime_error("Cannot open configuration file " + config_path);
  std::stringstream ss;
  ss << file.rdbuf();
  overwrite(xjson::parse(ss.str()));
  config.config_path = config_path;
}
bool Configuration::is_loaded_from_backup()
{
  const json_t* it = config.json.find("SimulationBackup");
  if (!it) return false;
  const json_t* load_it = it->find("load_from");
  return load_it && load_it->type == xjson::Value::Number && load_it->as_double() == std::floor(load_it->as_double());
}
void Configuration::set_out_dir(const std::string& dir) { config.out_dir = dir; }

// ---- World (src/utils/world.cpp:11-112)
/* static */ void World::set_geometry(PetscReal gx, PetscReal gy, PetscReal gz, PetscReal gt, PetscReal dx_,
  PetscReal dy_, PetscReal dz_, PetscReal dt_, PetscReal dtp)
{
  dx = dx_; dy = dy_; dz = dz_; dt = dt_;
  geom_x = gx; geom_y = gy; geom_z = gz; geom_t = gt;
  geom_nx = ROUND_STEP(geom_x, dx);
  geom_ny = ROUND_STEP(geom_y, dy);
  geom_nz = ROUND_STEP(geom_z, dz);
  geom_nt = ROUND_STEP(geom_t, dt);
  diagnose_period = ROUND_STEP(dtp, dt);
}

PetscErrorCode World::initialize()
{
  const Configuration::json_t& geometry = CONFIG().json.at("Geometry");
  // cell sizes first, parse_value() needs them (world.cpp:17-21)
  dx = geometry.at("dx").as_double();
  dy = geometry.at("dy").as_double();
  dz = geometry.at("dz").as_double();
  dt = geometry.at("dt").as_double();
  using interfaces::Builder;
  set_geometry(Builder::parse_value(geometry.at("x")), Builder::parse_value(geometry.at("y")),
    Builder::parse_value(geometry.at("z")), Builder::parse_value(geometry.at("t")), dx, dy, dz, dt,
    Builder::parse_value(geometry.at("diagnose_period")));
  for (const char* b : {"da_boundary_x", "da_boundary_y", "da_boundary_z"})
    if (geometry.at(b).as_string() != "DM_BOUNDARY_PERIODIC")
      throw std::runtime_error("the HIP backends support DM_BOUNDARY_PERIODIC only");
  geom.n[0] = geom_nx; geom.n[1] = geom_ny; geom.n[2] = geom_nz;
  geom.d[0] = dx; geom.d[1] = dy; geom.d[2] = dz;
  geom.dt = dt;
  geom.periodic[0] = geom.periodic[1] = geom.periodic[2] = 1;
  geom.rank = 0; geom.nranks = 1; geom.device = 0;
  return 0;
}

namespace interfaces {

// ---- Builder (src/interfaces/builder.cpp:22-81)
static bool ends_with(const std::string& s, const std::string& suf)
{
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

/* static */ PetscReal Builder::parse_value(const Configuration::json_t& value)
{
  if (!value.is_string()) return value.as_double();
  const std::string& str = value.as_string();
  if (str == "geom_x" || str == "geom_nx") return geom_x;
  if (str == "geom_y" || str == "geom_ny") return geom_y;
  if (str == "geom_z" || str == "geom_nz") return geom_z;
  if (ends_with(str, " [dx]")) return std::stod(str.substr(0, str.size() - 5)) * dx;
  if (ends_with(str, " [dy]")) return std::stod(str.substr(0, str.size() - 5)) * dy;
  if (ends_with(str, " [dz]")) return std::stod(str.substr(0, str.size() - 5)) * dz;
  if (ends_with(str, " [dt]")) return std::stod(str.substr(0, str.size() - 5)) * dt;
  if (ends_with(str, " [c/w_pe]") || ends_with(str, " [1/w_pe]")) return std::stod(str.substr(0, str.size() - 9));
  throw std::runtime_error("Unknown string format to convert: " + str);
}

/* static */ Vector3R Builder::parse_vector(const Configuration::json_t& info, const std::string& name)
{
  const Configuration::json_t& value = info.at(name);
  Vector3R result;
  if (value.is_array()) {
    if (value.arr.size() != 3) throw std::runtime_error(name + " vector should be of size 3.");
    for (int i = 0; i < 3; ++i) result[i] = parse_value(value.arr[i]);
    return result;
  }
  if (value.is_string()) {
    const std::string& str = value.as_string();
    if (str == "Geom") { result[0] = geom_x; result[1] = geom_y; result[2] = geom_z; }
    else if (str == "Geom / 2") { result[0] = geom_x / 2; result[1] = geom_y / 2; result[2] = geom_z / 2; }
    return result;
  }
  const PetscReal v = parse_value(value);
  result[0] = result[1] = result[2] = v;
  return result;
}

// ---- Particles
Particles::Particles(Simulation& simulation, const SortParameters& parameters)
  : parameters(parameters), simulation_(simulation)
{
}

PetscErrorCode Particles::add_particle(const Point& point, bool* is_added)
{
  // the same FLOOR_STEP bounds test the device applies (src/interfaces/particles.cpp:50-56)
  const PetscInt vx = FLOOR_STEP(point.r[0], dx), vy = FLOOR_STEP(point.r[1], dy), vz = FLOOR_STEP(point.r[2], dz);
  const bool inside = 0 <= vx && vx < geom_nx && 0 <= vy && vy < geom_ny && 0 <= vz && vz < geom_nz;
  if (!inside) return 0;
  for (int c = 0; c < 3; ++c) pending_.push_back(point.r[c]);
  for (int c = 0; c < 3; ++c) pending_.push_back(point.p[c]);
  if (is_added) *is_added = true;
  return 0;
}

PetscErrorCode Particles::flush()
{
  if (pending_.empty()) return 0;
  int64_t added = 0;
  HIPCALL(xpic_sort_add_particles(simulation_.ctx, sort_id, (int64_t)(pending_.size() / 6), pending_.data(), &added));
  pending_.clear();
  pending_.shrink_to_fit();
  return 0;
}

PetscErrorCode Particles::update_cells()
{
  int64_t n;
  HIPCALL(xpic_update_cells(simulation_.ctx, sort_id, &n));
  return 0;
}

PetscInt Particles::count()
{
  int64_t n = 0;
  xpic_sort_count(simulation_.ctx, sort_id, &n);
  return (PetscInt)n;
}

PetscErrorCode Particles::storage(std::vector<Point>& points, std::vector<int>& cell_of)
{
  const PetscInt n = count();
  std::vector<double> raw((size_t)n * 6);
  cell_of.resize(n);
  HIPCALL(xpic_sort_get_particles(simulation_.ctx, sort_id, raw.data(), cell_of.data()));
  points.resize(n);
  for (PetscInt i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c) { points[i].r[c] = raw[6 * i + c]; points[i].p[c] = raw[6 * i + 3 + c]; }
  return 0;
}

// ---- Simulation (src/interfaces/simulation.cpp:16-155)
Simulation::~Simulation()
{
  if (ctx) xpic_destroy(ctx);
}

PetscErrorCode Simulation::initialize()
{
  XCALL(world.initialize());
  LOG("Geometric constants for the current setup:");
  LOG("  Nx = " << geom_nx << ", Ny = " << geom_ny << ", Nz = " << geom_nz << ", Nt = " << geom_nt << ", dt = " << dt);
  LOG("Running initialize implementation");
  XCALL(initialize_implementation());

  // the always-on conservation diagnostic (simulation.cpp:39-50)
  diagnostics_.emplace_back(std::make_unique<Energy>(*this));
  diagnostics_.emplace_back(std::make_unique<ChargeConservation>(*this)); // simulation.cpp:52-53
  diagnostics_.emplace_back(std::make_unique<MomentumConservation>(*this)); // simulation.cpp:55-56

  std::vector<std::unique_ptr<Command>> presets;
  XCALL(build_commands(*this, "Presets", presets));
  XCALL(build_commands(*this, "StepPresets", step_presets_));
  XCALL(build_diagnostics(*this, diagnostics_));

  LOG("Executing presets");
  for (auto&& preset : presets) XCALL(preset->execute(start));
  for (auto& diagnostic : diagnostics_) XCALL(diagnostic->diagnose(start));
  return 0;
}

PetscErrorCode Simulation::calculate()
{
  LOG("Running the main simulation cycle");
  for (PetscInt t = start + 1; t <= geom_nt; ++t) {
    for (auto& command : step_presets_) XCALL(command->execute(t));
    XCALL(timestep_implementation(t));
    for (auto& diagnostic : diagnostics_) XCALL(diagnostic->diagnose(t));
  }
  return 0;
}

PetscErrorCode Simulation::finalize()
{
  for (auto& command : step_presets_) XCALL(command->finalize());
  for (auto& diagnostic : diagnostics_) XCALL(diagnostic->finalize());
  if (ctx) {
    HIPCALL(xpic_destroy(ctx));
    ctx = nullptr;
  }
  return 0;
}

int Simulation::get_named_vector(const std::string& name) const
{
  if (name == "E") return XPIC_E;
  if (name == "B") return XPIC_B;
  if (name == "B0") return XPIC_B0;
  throw std::out_of_range("no vector named " + name);
}

Particles& Simulation::get_named_particles(const std::string& name)
{
  for (auto& sort : particles_)
    if (sort->parameters.sort_name == name) return *sort;
  throw std::runtime_error("No particles with name " + name);
}

PetscErrorCode Simulation::initialize_implementation()
{
  HIPCALL(xpic_create(&world.geom, scheme(), &ctx));
  return init_particles();
}

PetscErrorCode Simulation::init_particles()
{
  const Configuration::json_t& json = CONFIG().json;
  const Configuration::json_t* it = json.find("Particles");
  if (!it || it->arr.empty()) return 0;
  // capacity: the SetParticles presets tell how many points each sort will get; leave head-room for migration
  for (auto&& info : it->arr) {
    if (!info.contains("sort_name")) continue;
    SortParameters p;
    p.sort_name = info.at("sort_name").as_string();
    p.Np = info.at("Np").as_int();
    p.n = info.at("n").as_double();
    p.q = info.at("q").as_double();
    p.m = info.at("m").as_double();
    if (info.contains("T")) p.Tx = p.Ty = p.Tz = info.at("T").as_double();
    else {
      p.Tx = info.at("Tx").as_double();
      p.Ty = info.at("Ty").as_double();
      p.Tz = info.at("Tz").as_double();
    }
    // EXTENSION (not reference behaviour): the reference's init_particles never reads the initial momentum
    // (src/interfaces/simulation.tpp:24-41), although SortParameters has it (sort_parameters.h:13-15) and MaxwellianMomentum
    // adds it (src/utils/particles_load.cpp:59-70): without it the JSON surface cannot express the two counter-streaming
    // beams of a two-stream set-up.  Absent keys leave the reference's behaviour (zero drift) untouched.
    if (info.contains("px")) p.px = info.at("px").as_double();
    if (info.contains("py")) p.py = info.at("py").as_double();
    if (info.contains("pz")) p.pz = info.at("pz").as_double();
    auto sort = std::make_shared<Particles>(*this, p);
    xpic_sort_params sp{p.Np, p.n, p.q, p.m};
    const int64_t cells = (int64_t)geom_nx * geom_ny * geom_nz;
    const int64_t capacity = std::max<int64_t>(cells * p.Np * 5 / 4 + 4096, 1 << 16);
    HIPCALL(xpic_add_sort(ctx, &sp, capacity, &sort->sort_id));
    particles_.emplace_back(sort);
    LOG("  " << p.sort_name << " are added");
  }
  return 0;
}

PetscErrorCode Simulation::timestep_implementation(PetscInt /* t */)
{
  int its = 0;
  HIPCALL(xpic_step(ctx, &its));
  return 0;
}

}  // namespace interfaces

PetscErrorCode ecsim::Simulation::timestep_implementation(PetscInt /* t */)
{
  int its = 0;
  HIPCALL(xpic_step(ctx, &its));
  last_ksp_iterations = its;
  LOG("  KSPSolve() has finished, iterations: " << its);
  return 0;
}

std::unique_ptr<interfaces::Simulation> build_simulation()
{
  const std::string simulation_str = CONFIG().json.at("Simulation").as_string();
  std::unique_ptr<interfaces::Simulation> simulation;
  if (simulation_str == "basic") simulation = std::make_unique<basic::Simulation>();
  else if (simulation_str == "ecsim") simulation = std::make_unique<ecsim::Simulation>();
  else if (simulation_str == "ecsimcorr") simulation = std::make_unique<ecsimcorr::Simulation>();
  else throw std::runtime_error("Unkown simulation is used: " + simulation_str);
  LOG("Simulation is built, scheme " << simulation_str);
  return simulation;
}

// ---- particle loaders (src/utils/random_generator.h:8-35, src/utils/particles_load.cpp:11-76)
static std::mt19937& generator()
{
  static std::mt19937 gen; // RANDOM_SEED false (src/constants.h:5): default seed
  return gen;
}
static PetscReal random_01()
{
  static std::uniform_real_distribution<double> distribution(0.0, 1.0);
  return distribution(generator());
}
static PetscReal temperature_momentum(PetscReal temperature, PetscReal mass)
{
  return std::sqrt(-2.0 * (temperature * mass / mec2) * std::log(random_01()));
}

SetParticles::SetParticles(interfaces::Particles& particles, PetscInt number_of_particles, CoordinateGenerator gc,
  MomentumGenerator gm)
  : particles_(particles), number_of_particles_(number_of_particles), generate_coordinate_(std::move(gc)),
    generate_momentum_(std::move(gm))
{
}

PetscErrorCode SetParticles::execute(PetscInt /* t */) // src/commands/set_particles.cpp:19-43
{
  added_energy = 0.0;
  added_particles = 0;
  const PetscReal m = particles_.parameters.m;
  const PetscReal mpw = particles_.parameters.n / particles_.parameters.Np;
  for (PetscInt p = 0; p < number_of_particles_; ++p) {
    Vector3R coordinate = generate_coordinate_();
    Vector3R momentum = generate_momentum_(coordinate);
    bool is_added = false;
    Point point;
    point.r = coordinate;
    point.p = momentum;
    XCALL(particles_.add_particle(point, &is_added));
    if (is_added) {
      added_energy += 0.5 * (m * momentum.squared()) * mpw; // Energy::get_kinetic
      added_particles++;
    }
  }
  XCALL(particles_.flush());
  LOG("  Particles have been added into \"" << particles_.parameters.sort_name << "\": " << added_particles);
  return 0;
}

SetMagneticField::SetMagneticField(interfaces::Simulation& sim, int field, int field_axpy, const Vector3R& v)
  : sim_(sim), field_(field), field_axpy_(field_axpy), value_(v)
{
}

PetscErrorCode SetMagneticField::execute(PetscInt /* t */) // src/commands/set_magnetic_field.cpp:12-19, 27-35
{
  const size_t n = (size_t)geom_nx * geom_ny * geom_nz;
  std::vector<double> v(3 * n);
  for (size_t i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c) v[3 * i + c] = value_[c]; // VecStrideSet
  HIPCALL(xpic_field_set(sim_.ctx, field_, v.data()));
  if (field_axpy_ >= 0) HIPCALL(xpic_vec_axpy(sim_.ctx, field_axpy_, 1.0, field_));
  return 0;
}

// build_commands (src/commands/builders/command_builder.cpp:16-62) with ParticlesBuilder::load_coordinate /
// load_momentum (particles_builder.cpp:9-68) and SetMagneticFieldBuilder (set_magnetic_field_builder.cpp:11-63)
PetscErrorCode build_commands(interfaces::Simulation& simulation, const std::string& name,
  std::vector<std::unique_ptr<interfaces::Command>>& result)
{
  using interfaces::Builder;
  if (name == "Presets" && Configuration::is_loaded_from_backup()) { // command_builder.cpp:24-27
    const auto& info = CONFIG().json.at("SimulationBackup");
    const PetscInt load_from = (PetscInt)info.at("load_from").as_double();
    LOG("Restoring simulation from backup at " << load_from * dt << " [1/w_pe], " << load_from << " [dt]");
    simulation.start = load_from; // simulation_backup_builder.cpp:75-79
    result.emplace_back(std::make_unique<SimulationBackup>(CONFIG().out_dir + "/simulation_backup/", -1, simulation));
    return 0;
  }
  const Configuration::json_t* it = CONFIG().json.find(name);
  if (!it || it->arr.empty()) return 0;
  for (auto&& info : it->arr) {
    if (!info.contains("command")) continue;
    const std::string command = info.at("command").as_string();
    if (command == "SetParticles") {
      auto& particles = simulation.get_named_particles(info.at("particles").as_string());
      const PetscInt Np = particles.parameters.Np;
      const PetscReal frac = Np / (dx * dy * dz);
      PetscInt number_of_particles = 0;
      CoordinateGenerator gc;
      const auto& ci = info.at("coordinate");
      const std::string cname = ci.at("name").as_string();
      if (cname == "PreciseCoordinate") {
        number_of_particles = Np;
        Vector3R dot = Builder::parse_vector(ci, "value");
        gc = [dot]() { return dot; };
      }
      else if (cname == "CoordinateInBox") {
        Vector3R mn, mx;
        mx[0] = geom_x; mx[1] = geom_y; mx[2] = geom_z; // Builder::load_geometry defaults (builder.cpp:83-94)
        if (ci.contains("min")) mn = Builder::parse_vector(ci, "min");
        if (ci.contains("max")) mx = Builder::parse_vector(ci, "max");
        number_of_particles = ((mx[0] - mn[0]) * (mx[1] - mn[1]) * (mx[2] - mn[2])) * frac; // truncation, :26
        gc = [mn, mx]() {
          Vector3R r;
          r[0] = mn[0] + random_01() * (mx[0] - mn[0]);
          r[1] = mn[1] + random_01() * (mx[1] - mn[1]);
          r[2] = mn[2] + random_01() * (mx[2] - mn[2]);
          return r;
        };
      }
      else throw std::runtime_error("Unknown coordinate generator name " + cname);
      MomentumGenerator gm;
      const auto& mi = info.at("momentum");
      const std::string mname = mi.at("name").as_string();
      if (mname == "PreciseMomentum") {
        Vector3R value = Builder::parse_vector(mi, "value");
        gm = [value](const Vector3R&) { return value; };
      }
      else if (mname == "MaxwellianMomentum") {
        bool tov = mi.contains("tov") ? mi.at("tov").as_bool() : false;
        SortParameters params = particles.parameters;
        gm = [params, tov](const Vector3R&) { // particles_load.cpp:57-76; sin(2 pi u) is drawn before sqrt(-2..ln u)
          Vector3R result;
          const PetscReal sx = std::sin(2.0 * M_PI * random_01());
          result[0] = params.px + sx * temperature_momentum(params.Tx, params.m);
          const PetscReal sy = std::sin(2.0 * M_PI * random_01());
          result[1] = params.py + sy * temperature_momentum(params.Ty, params.m);
          const PetscReal sz = std::sin(2.0 * M_PI * random_01());
          result[2] = params.pz + sz * temperature_momentum(params.Tz, params.m);
          if (tov) {
            const PetscReal g = std::sqrt(params.m * params.m + result.squared());
            for (int c = 0; c < 3; ++c) result[c] /= g;
          }
          return result;
        };
      }
      else throw std::runtime_error("Unknown coordinate generator name " + mname);
      result.emplace_back(std::make_unique<SetParticles>(particles, number_of_particles, gc, gm));
      LOG("  SetParticles command is added for \"" << particles.parameters.sort_name << "\"");
    }
    else if (command == "SetMagneticField") {
      const int field = simulation.get_named_vector(info.at("field").as_string());
      int axpy = -1;
      if (info.contains("field_axpy")) axpy = simulation.get_named_vector(info.at("field_axpy").as_string());
      const auto& setter = info.at("setter");
      const std::string sname = setter.at("name").as_string();
      if (sname != "SetUniformField") throw std::runtime_error("Unknown setter name " + sname);
      result.emplace_back(std::make_unique<SetMagneticField>(simulation, field, axpy, Builder::parse_vector(setter, "value")));
    }
    else throw std::runtime_error("Unknown command name " + command);
  }
  return 0;
}

// ---- FieldView / DistributionMoment / SimulationBackup and their builders
/* static */ std::string FieldView::format_time(PetscInt t)
{
  const size_t width = std::to_string(geom_nt).size();
  std::string s = std::to_string(t);
  return s.size() < width ? std::string(width - s.size(), '0') + s : s;
}

FieldView::FieldView(const std::string& out_dir, interfaces::Simulation& simulation, int field, const Region& region)
  : simulation(simulation), out_dir_(out_dir), field_(field), region_(region)
{
}

PetscErrorCode FieldView::fetch(std::vector<double>& data)
{
  data.resize((size_t)geom_nx * geom_ny * geom_nz * 3);
  HIPCALL(xpic_field_get(simulation.ctx, field_, data.data()));
  return 0;
}

PetscErrorCode FieldView::diagnose(PetscInt t)
{
  if (diagnose_period > 0 && t % diagnose_period != 0) return 0;
  std::vector<double> data;
  XCALL(fetch(data));
  const PetscInt dof = region_.dof;
  const PetscInt c0 = dof > 1 ? region_.start[3] : 0, nc = dof > 1 ? region_.size[3] : 1;
  std::vector<float> out;
  out.reserve((size_t)region_.size[0] * region_.size[1] * region_.size[2] * nc);
  for (PetscInt z = region_.start[2]; z < region_.start[2] + region_.size[2]; ++z)
    for (PetscInt y = region_.start[1]; y < region_.start[1] + region_.size[1]; ++y)
      for (PetscInt x = region_.start[0]; x < region_.start[0] + region_.size[0]; ++x)
        for (PetscInt cc = c0; cc < c0 + nc; ++cc)
          out.push_back((float)data[(((size_t)z * geom_ny + y) * geom_nx + x) * dof + cc]); // write_floats: double -> float
  make_dirs(out_dir_);
  std::ofstream f(out_dir_ + "/" + format_time(t), std::ios::binary);
  if (!f) throw std::runtime_error("Cannot open " + out_dir_ + "/" + format_time(t));
  f.write(reinterpret_cast<const char*>(out.data()), (std::streamsize)(out.size() * sizeof(float)));
  return 0;
}

DistributionMoment::DistributionMoment(const std::string& out_dir, interfaces::Simulation& simulation,
  interfaces::Particles& particles, const Region& region)
  : FieldView(out_dir, simulation, -1, region), particles_(particles)
{
}

PetscErrorCode DistributionMoment::fetch(std::vector<double>& data)
{
  data.resize((size_t)geom_nx * geom_ny * geom_nz);
  HIPCALL(xpic_moment_density(simulation.ctx, particles_.sort_id, data.data()));
  return 0;
}

namespace {

// FieldViewBuilder::parse_region_start_size / parse_res_dir_suffix / check_region (field_view_builder.cpp:60-147)
void parse_plane_position(const Configuration::json_t& info, std::string& plane, PetscReal& position)
{
  plane = info.at("plane").as_string();
  if (plane == "X") position = 0.5 * geom_x;
  else if (plane == "Y") position = 0.5 * geom_y;
  else if (plane == "Z") position = 0.5 * geom_z;
  else throw std::runtime_error("Unknown plane " + plane);
  if (info.contains("position")) position = info.at("position").as_double();
}

void parse_region(const Configuration::json_t& info, FieldView::Region& region, std::string& suffix,
  const std::string& name)
{
  using interfaces::Builder;
  Vector3R start, size;
  size[0] = geom_x; size[1] = geom_y; size[2] = geom_z;
  std::string type = info.contains("type") ? info.at("type").as_string() : "3D";
  if (type != "3D" && type != "2D") throw std::runtime_error("Incorrect type is used for " + name + " .");
  if (info.contains("start")) start = Builder::parse_vector(info, "start");
  if (info.contains("size")) size = Builder::parse_vector(info, "size");
  const PetscReal d[3] = {dx, dy, dz};
  if (type == "2D") {
    std::string plane;
    PetscReal position;
    parse_plane_position(info, plane, position);
    const int dir = plane == "X" ? 0 : (plane == "Y" ? 1 : 2);
    start[dir] = position;
    size[dir] = d[dir];
    char buf[32];
    std::snprintf(buf, sizeof(buf), "plane%s_%04d", plane.c_str(), (int)FLOOR_STEP(position, d[dir]));
    suffix += buf;
  }
  for (int i = 0; i < 3; ++i) {
    region.start[i] = FLOOR_STEP(start[i], d[i]);
    region.size[i] = FLOOR_STEP(size[i], d[i]);
  }
}

void check_region(const FieldView::Region& region, const std::string& name)
{
  const PetscInt n[3] = {geom_nx, geom_ny, geom_nz};
  for (int i = 0; i < 3; ++i)
    if (region.start[i] < 0 || region.start[i] + region.size[i] > n[i])
      throw std::runtime_error("Region is not in global boundaries for " + name + " diagnostic.");
  if (!(region.size[0] > 0 && region.size[1] > 0 && region.size[2] > 0))
    throw std::runtime_error("Sizes are invalid for " + name + " diagnostic.");
}

}  // namespace

PetscErrorCode build_diagnostics(interfaces::Simulation& simulation,
  std::vector<std::unique_ptr<interfaces::Diagnostic>>& result)
{
  using interfaces::Builder;
  LOG("Building diagnostics");
  if (const Configuration::json_t* it = CONFIG().json.find("SimulationBackup"); it && !it->obj.empty()) {
    const PetscReal dp_wp = Builder::parse_value(it->at("diagnose_period"));
    const PetscInt dp = ROUND_STEP(dp_wp, dt);
    LOG("  Simulation backup diagnostic is added, diagnose period: " << dp_wp << " [1/w_pe], " << dp << " [dt]");
    const std::string res_dir = CONFIG().out_dir + "/simulation_backup";
    make_dirs(res_dir);
    if (!CONFIG().config_path.empty()) { // Configuration::save (configuration.cpp:26-36)
      std::error_code ec;
      std::filesystem::copy(CONFIG().config_path, res_dir, std::filesystem::copy_options::overwrite_existing, ec);
    }
    result.emplace_back(std::make_unique<SimulationBackup>(res_dir + "/", dp, simulation));
  }
  const Configuration::json_t* list = CONFIG().json.find("Diagnostics");
  if (!list || list->arr.empty()) return 0;
  for (auto&& info : list->arr) {
    if (!info.contains("diagnostic")) continue;
    const std::string name = info.at("diagnostic").as_string();
    FieldView::Region region;
    region.size[0] = geom_nx; region.size[1] = geom_ny; region.size[2] = geom_nz;
    std::string suffix;
    if (name == "FieldView") {
      const std::string field = info.at("field").as_string();
      region.dim = 4; region.dof = 3; region.start[3] = 0; region.size[3] = 3;
      if (info.contains("region")) parse_region(info.at("region"), region, suffix, field);
      check_region(region, field);
      LOG("  field view diagnostic is added for " << field << ", suffix: " << (suffix.empty() ? "<empty>" : suffix));
      if (!suffix.empty()) suffix = "_" + suffix;
      result.emplace_back(std::make_unique<FieldView>(CONFIG().out_dir + "/" + field + suffix + "/", simulation,
        simulation.get_named_vector(field), region));
    }
    else if (name == "DistributionMoment") {
      const std::string particles = info.at("particles").as_string(), moment = info.at("moment").as_string();
      static const char* known[] = {"density", "current", "momentum_flux", "momentum_flux_cyl", "momentum_flux_diag",
        "momentum_flux_diag_cyl"};
      if (std::find_if(std::begin(known), std::end(known), [&](const char* k) { return moment == k; }) == std::end(known))
        throw std::runtime_error("Unknown moment name " + moment + " for particles " + particles);
      if (moment != "density")
        throw std::runtime_error("moment " + moment + " is not offered by the HIP backends (density only)");
      region.dim = 3; region.dof = 1; region.size[3] = 1;
      if (info.contains("region")) parse_region(info.at("region"), region, suffix, particles + " " + moment);
      check_region(region, particles + " " + moment);
      LOG("  " << moment << " diagnostic is added for " << particles << ", suffix: " << (suffix.empty() ? "<empty>" : suffix));
      if (!suffix.empty()) suffix = "_" + suffix;
      result.emplace_back(std::make_unique<DistributionMoment>(CONFIG().out_dir + "/" + particles + "/" + moment + suffix,
        simulation, simulation.get_named_particles(particles), region));
    }
    else throw std::runtime_error("Unknown diagnostic name " + name + " (HIP backends: FieldView, DistributionMoment)");
  }
  return 0;
}

namespace {

// PETSc binary viewers write big endian (PetscViewerBinaryWrite -> PetscByteSwap on little-endian hosts)
void put_be(std::ofstream& f, const void* src, size_t width, size_t count)
{
  const unsigned char* p = static_cast<const unsigned char*>(src);
  std::vector<unsigned char> buf(width * count);
  for (size_t i = 0; i < count; ++i)
    for (size_t b = 0; b < width; ++b) buf[i * width + b] = p[i * width + (width - 1 - b)];
  f.write(reinterpret_cast<const char*>(buf.data()), (std::streamsize)buf.size());
}

bool get_be(std::ifstream& f, void* dst, size_t width, size_t count)
{
  std::vector<unsigned char> buf(width * count);
  f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)buf.size());
  if ((size_t)f.gcount() != buf.size()) return false;
  unsigned char* p = static_cast<unsigned char*>(dst);
  for (size_t i = 0; i < count; ++i)
    for (size_t b = 0; b < width; ++b) p[i * width + b] = buf[i * width + (width - 1 - b)];
  return true;
}

constexpr int32_t kVecFileClassId = 1211214; // petscvec.h VEC_FILE_CLASSID
const char* const kBackupFields[3] = {"E", "B", "B0"};

}  // namespace

SimulationBackup::SimulationBackup(const std::string& out_dir, PetscInt diagnose_period, interfaces::Simulation& simulation)
  : out_dir_(out_dir), diagnose_period_(diagnose_period), simulation(simulation)
{
}

PetscErrorCode SimulationBackup::save(PetscInt t)
{
  if (diagnose_period_ <= 0 || t % diagnose_period_ != 0) return 0;
  const std::string dir = out_dir_ + "/" + std::to_string(t);
  make_dirs(dir);
  const size_t n3 = (size_t)geom_nx * geom_ny * geom_nz * 3;
  // the PETSc binary Vec header and the .numparts file carry 32-bit counts (PetscInt without --with-64-bit-indices)
  if (n3 > (size_t)INT32_MAX) throw std::runtime_error("SimulationBackup: the Vec length exceeds the 32-bit PetscInt of the file format");
  std::vector<double> data(n3);
  for (const char* name : kBackupFields) { // save_fields :48-63, VecView of a DMDA vector = natural ordering
    HIPCALL(xpic_field_get(simulation.ctx, simulation.get_named_vector(name), data.data()));
    std::ofstream f(dir + "/" + name, std::ios::binary);
    const int32_t header[2] = {kVecFileClassId, (int32_t)n3};
    put_be(f, header, 4, 2);
    put_be(f, data.data(), 8, n3);
  }
  for (auto& sort : simulation.particles_) { // save_particles :65-91
    std::vector<Point> points;
    std::vector<int> cells;
    XCALL(sort->storage(points, cells));
    if (points.size() > (size_t)INT32_MAX)
      throw std::runtime_error("SimulationBackup: more particles than the 32-bit .numparts file can hold");
    const int32_t numparts = (int32_t)points.size();
    std::ofstream fn(dir + "/" + sort->parameters.sort_name + ".numparts", std::ios::binary);
    put_be(fn, &numparts, 4, 1);
    std::ofstream fp(dir + "/" + sort->parameters.sort_name, std::ios::binary);
    static_assert(sizeof(Point) == 6 * sizeof(double), "Point is six doubles");
    put_be(fp, points.data(), 8, 6 * points.size());
  }
  std::error_code ec; // save_temporal_diagnostics :94-101
  if (std::filesystem::exists(CONFIG().out_dir + "/temporal"))
    std::filesystem::copy(CONFIG().out_dir + "/temporal", dir + "/temporal",
      std::filesystem::copy_options::overwrite_existing | std::filesystem::copy_options::recursive, ec);
  std::filesystem::remove_all(out_dir_ + "/" + std::to_string(t - num_periods_being_kept * diagnose_period_), ec);
  return 0;
}

PetscErrorCode SimulationBackup::load(PetscInt t)
{
  if (!std::filesystem::exists(out_dir_)) throw std::runtime_error("Cannot load the timestep, no backup directory");
  const std::string dir = out_dir_ + "/" + std::to_string(t);
  const size_t n3 = (size_t)geom_nx * geom_ny * geom_nz * 3;
  std::vector<double> data(n3);
  for (const char* name : kBackupFields) { // load_fields :117-131
    std::ifstream f(dir + "/" + name, std::ios::binary);
    int32_t header[2] = {0, 0};
    if (!f || !get_be(f, header, 4, 2) || header[0] != kVecFileClassId || (size_t)header[1] != n3 ||
      !get_be(f, data.data(), 8, n3))
      throw std::runtime_error(std::string("Incorrect field data in backup file ") + dir + "/" + name);
    HIPCALL(xpic_field_set(simulation.ctx, simulation.get_named_vector(name), data.data()));
  }
  for (auto& sort : simulation.particles_) { // load_particles :133-160
    const std::string fname = dir + "/" + sort->parameters.sort_name;
    std::ifstream fn(fname + ".numparts", std::ios::binary);
    int32_t numparts = 0;
    if (!fn || !get_be(fn, &numparts, 4, 1)) throw std::runtime_error("Incorrect number of particles to read is specified");
    std::ifstream fp(fname, std::ios::binary);
    Point point;
    for (int32_t i = 0; i < numparts; ++i) {
      if (!fp || !get_be(fp, &point, 8, 6)) throw std::runtime_error("Point structure consists of 6 PetscReal values");
      XCALL(sort->add_particle(point));
    }
    XCALL(sort->flush());
  }
  std::error_code ec; // load_temporal_diagnostics :162-169
  if (std::filesystem::exists(dir + "/temporal"))
    std::filesystem::copy(dir + "/temporal", CONFIG().out_dir + "/temporal",
      std::filesystem::copy_options::overwrite_existing | std::filesystem::copy_options::recursive, ec);
  LOG("  Simulation is successfully loaded from " << t * dt << " [1/w_pe], " << t << " [dt]");
  return 0;
}

// ---- TableDiagnostic (src/diagnostics/utils/table_diagnostic.cpp:8-59, table_diagnostic.h:19-38)
static std::string pad_left(const std::string& s, int w) // std::format("{:<{}.{}s}")
{
  std::string t = s.substr(0, w);
  t.append(w - t.size(), ' ');
  return t;
}
static std::string pad_center(const std::string& s, int w) // std::format("{:^{}.{}s}")
{
  std::string t = s.substr(0, w);
  const int pad = w - (int)t.size();
  const int left = pad / 2;
  return std::string(left, ' ') + t + std::string(pad - left, ' ');
}

TableDiagnostic::TableDiagnostic(const std::string& filename) : filename_(filename) {}

void TableDiagnostic::add(PetscInt w, std::string title, const char* printf_fmt, double value, PetscInt pos)
{
  char buf[64];
  std::snprintf(buf, sizeof(buf), printf_fmt, value);
  title = pad_left(title, w);
  std::string fvalue = pad_center(buf, w);
  if (pos >= 0) {
    titles_.insert(titles_.begin() + pos, title);
    values_.insert(values_.begin() + pos, fvalue);
  }
  else {
    titles_.push_back(title);
    values_.push_back(fvalue);
  }
}

void TableDiagnostic::add_int(PetscInt w, std::string title, long value)
{
  titles_.push_back(pad_left(title, w));
  values_.push_back(pad_center(std::to_string(value), w));
}

void TableDiagnostic::write_formatted(const std::vector<std::string>& container)
{
  for (size_t i = 0; i + 1 < container.size(); ++i) file_ << container[i] << "  ";
  std::string last = container.back();
  while (!last.empty() && last.back() == ' ') last.pop_back();
  file_ << last << "\n";
}

PetscErrorCode TableDiagnostic::diagnose(PetscInt t)
{
  if (!file_.is_open()) {
    const size_t slash = filename_.rfind('/');
    if (slash != std::string::npos) make_dirs(filename_.substr(0, slash));
    file_.open(filename_, t == 0 ? std::ios::out : std::ios::app); // a run restored from a backup appends
    if (!file_) throw std::runtime_error("Cannot open " + filename_);
  }
  if (!initialized_) { // t == 0, or the first step of a run restored from a backup
    XCALL(initialize());
    initialized_ = true;
  }
  XCALL(add_columns(t));
  if (!values_.empty()) {
    if (t == 0) write_formatted(titles_);
    write_formatted(values_);
    titles_.clear();
    values_.clear();
  }
  file_.flush(); // every row (the reference flushes per diagnose period): a backup taken this step copies whole tables
  return 0;
}

// ---- ChargeConservation: the densities, the divergence and both norms stay on the device
ChargeConservation::ChargeConservation(interfaces::Simulation& simulation)
  : TableDiagnostic(CONFIG().out_dir + "/temporal/charge_conservation.txt"), simulation(simulation)
{
}

PetscErrorCode ChargeConservation::initialize()
{
  HIPCALL(xpic_charge_collect(simulation.ctx));
  return 0;
}

PetscErrorCode ChargeConservation::add_columns(PetscInt t)
{
  auto& particles = simulation.particles_;
  std::vector<double> norm(2 * particles.size() + 2);
  HIPCALL(xpic_charge_columns(simulation.ctx, norm.data()));
  add_int(6, "Time", t);
  for (size_t i = 0; i < particles.size(); ++i) {
    add(13, "N1dQ_" + particles[i]->parameters.sort_name, "% .6e", norm[2 * i]);
    add(13, "N2dQ_" + particles[i]->parameters.sort_name, "% .6e", norm[2 * i + 1]);
  }
  add(13, "N1dQ_tot", "% .6e", norm[2 * particles.size()]);
  add(13, "N2dQ_tot", "% .6e", norm[2 * particles.size() + 1]);
  return 0;
}

// ---- MomentumConservation (src/diagnostics/momentum_conservation.cpp): dP/dt against the force q E on the particles
MomentumConservation::MomentumConservation(interfaces::Simulation& simulation)
  : TableDiagnostic(CONFIG().out_dir + "/temporal/momentum_conservation.txt"), simulation(simulation)
{
}

PetscErrorCode MomentumConservation::calculate()
{
  const size_t n = simulation.particles_.size();
  std::vector<double> sums(6 * n);
  HIPCALL(xpic_momentum(simulation.ctx, sums.data()));
  P1.resize(n);
  QE.resize(n);
  P0.resize(n);
  for (size_t i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c) {
      P1[i][c] = sums[6 * i + c];
      QE[i][c] = sums[6 * i + 3 + c];
    }
  return 0;
}

PetscErrorCode MomentumConservation::initialize()
{
  XCALL(calculate());
  P0 = P1;
  return 0;
}

PetscErrorCode MomentumConservation::add_columns(PetscInt t)
{
  XCALL(calculate());
  add_int(6, "Time", t);
  auto length = [](const Vector3R& v) { return std::sqrt(v.squared()); };
  Vector3R total;
  for (size_t i = 0; i < P1.size(); ++i) {
    const std::string& name = simulation.particles_[i]->parameters.sort_name;
    static const char* axis[3] = {"x_", "y_", "z_"};
    for (int c = 0; c < 3; ++c) add(13, std::string("P") + axis[c] + name, "% .6e", P1[i][c]);
    for (int c = 0; c < 3; ++c) add(13, std::string("QE") + axis[c] + name, "% .6e", QE[i][c]);
    Vector3R defect, diff, mean;
    for (int c = 0; c < 3; ++c) {
      diff[c] = P1[i][c] - P0[i][c];
      mean[c] = P1[i][c] + P0[i][c];
      defect[c] = diff[c] / dt - QE[i][c]; // (p1 - p0) / dt - qe   :56
      total[c] += defect[c];
    }
    PetscReal freq = 0;
    if (const PetscReal denom = length(mean); std::abs(denom) > 1e-10 /* PETSC_SMALL */)
      freq = (length(diff) / denom) / (0.5 * dt);
    add(13, "N2dP_" + name, "% .6e", length(defect));
    add(13, "fP_" + name, "% .6e", freq);
    P0[i] = P1[i];
  }
  add(13, "N2dP", "% .6e", length(total));
  return 0;
}

// ---- Energy (src/diagnostics/energy.cpp:9-185; ecsimcorr::Energy src/impls/ecsimcorr/simulation.cpp:157-197)
Energy::Energy(interfaces::Simulation& simulation)
  : simulation(simulation), energy(CONFIG().out_dir + "/temporal/energy.txt"),
    energy_cons(CONFIG().out_dir + "/temporal/energy_conservation.txt")
{
  const size_t n = simulation.particles_.size();
  K.assign(n, 0);
  K0.assign(n, 0);
  std_K.assign(n, 0);
}

PetscErrorCode Energy::calculate()
{
  std::vector<double> out(4 + 2 * K.size());
  HIPCALL(xpic_energy(simulation.ctx, out.data()));
  E = out[0]; B = out[1]; std_E = out[2]; std_B = out[3];
  for (size_t i = 0; i < K.size(); ++i) { K[i] = out[4 + 2 * i]; std_K[i] = out[5 + 2 * i]; }
  return 0;
}

PetscErrorCode Energy::diagnose(PetscInt t)
{
  if (t == simulation.start) XCALL(calculate());
  E0 = E; B0 = B; K0 = K;
  XCALL(calculate());

  auto& particles = simulation.particles_;
  // fill_energy :111-135
  energy.add_int(6, "Time", t);
  energy.add(13, "wE", "% .6e", E);
  energy.add(13, "wB", "% .6e", B);
  for (size_t i = 0; i < K.size(); ++i) energy.add(13, "wK_" + particles[i]->parameters.sort_name, "% .6e", K[i]);
  energy.add(13, "sE", "% .6e", std_E);
  energy.add(13, "sB", "% .6e", std_B);
  for (size_t i = 0; i < K.size(); ++i) energy.add(13, "sK_" + particles[i]->parameters.sort_name, "% .6e", std_K[i]);

  // fill_energy_cons :137-185
  energy_cons.add_int(6, "Time", t);
  const PetscReal dE = E - E0, dB = B - B0;
  PetscReal dF = dE + dB, dK = 0;
  energy_cons.add(13, "dE", "% .6e", dE);
  energy_cons.add(13, "dB", "% .6e", dB);
  for (size_t i = 0; i < K.size(); ++i) {
    energy_cons.add(13, "dK_" + particles[i]->parameters.sort_name, "% .6e", K[i] - K0[i]);
    dK += K[i] - K0[i];
  }
  energy_cons.add(13, "dE+dB+dK", "% .6e", dF + dK);
  if (simulation.scheme() == XPIC_ECSIMCORR) {
    PetscInt off = 3;
    PetscReal corr_w = 0.0;
    for (size_t i = 0; i < K.size(); ++i) {
      double s[6];
      HIPCALL(xpic_ecsimcorr_scalars(simulation.ctx, particles[i]->sort_id, s));
      const std::string& name = particles[i]->parameters.sort_name;
      energy_cons.add(13, "CWD_" + name, "% .6e", s[2], ++off);            // lambda_dK
      energy_cons.add(13, "PWD_" + name, "% .6e", s[3] - dt * s[0], ++off); // pred_dK - dt pred_w
      energy_cons.add(13, "LdK_" + name, "% .6e", s[4] - dt * s[1], ++off); // corr_dK - dt corr_w
      ++off;
      corr_w += s[1];
    }
    energy_cons.add(13, "WD", "% .6e", dK - dt * corr_w);
  }
  XCALL(energy.diagnose(t));
  XCALL(energy_cons.diagnose(t));
  return 0;
}
