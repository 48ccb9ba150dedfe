// main.cpp -- mirror of the reference's src/main.cpp:9-40: configuration -> build_simulation -> initialize ->
// calculate -> finalize, exceptions reported and turned into a non-zero exit.
//   usage: xpic_hip.out <config.json> [output directory override]
#include <fstream>
#include <iostream>

#include "xpic_host.h"

// Host-logic self test (no GPU needed): the JSON reader, Builder::parse_value and the table text format,
// checked against literal lines of the reference's golden files.
static int selftest()
{
  int bad = 0;
  auto expect = [&](bool ok, const char* what) {
    if (!ok) { std::cerr << "selftest FAILED: " << what << "\n"; ++bad; }
  };
  Configuration::overwrite(xjson::parse(R"({"OutputDirectory": "/tmp/xpic_selftest", "Simulation": "ecsim",
    "Geometry": {"x": "2 [dx]", "y": 5.0, "z": "32 [dz]", "t": "30 [1/w_pe]", "dx": 0.5, "dy": 0.5, "dz": 0.25,
      "dt": 0.1, "diagnose_period": "100 [dt]", "da_boundary_x": "DM_BOUNDARY_PERIODIC",
      "da_boundary_y": "DM_BOUNDARY_PERIODIC", "da_boundary_z": "DM_BOUNDARY_PERIODIC"},
    "Particles": [{"sort_name": "electrons", "Np": 1000, "n": 1, "q": -1, "m": 1, "T": 1}],
    "Presets": [{"command": "SetParticles", "particles": "electrons", "flag": true, "nothing": null,
      "momentum": {"name": "MaxwellianMomentum", "tov": true}, "coordinate": {"name": "CoordinateInBox"}}]})"));
  World w;
  w.initialize();
  expect(geom_nx == 2 && geom_ny == 10 && geom_nz == 32 && geom_nt == 300 && diagnose_period == 100, "geometry");
  expect(geom_x == 1.0 && geom_z == 8.0 && geom_t == 30.0, "parse_value units");
  const auto& pre = CONFIG().json.at("Presets").arr[0];
  expect(pre.at("flag").as_bool() && pre.at("nothing").type == xjson::Value::Null, "bool / null");
  expect(CONFIG().json.obj[0].first == "OutputDirectory" && CONFIG().json.obj[1].first == "Simulation", "key order kept");
  bool threw = false;
  try { interfaces::Builder::parse_value(xjson::parse("\"3 [parsec]\"")); } catch (const std::runtime_error&) { threw = true; }
  expect(threw, "unknown unit throws");
  threw = false;
  try { xjson::parse("{\"a\": [1, 2,, 3]}"); } catch (const std::runtime_error&) { threw = true; }
  expect(threw, "malformed json throws");
  // table format: header and first rows of tests/ecsim/expected/ecsim_ex1/temporal/energy_conservation.txt
  {
    struct T : TableDiagnostic {
      using TableDiagnostic::TableDiagnostic;
      int row = 0;
      PetscErrorCode add_columns(PetscInt t) override
      {
        add_int(6, "Time", t);
        const double v[2][4] = {{0, 0, 0, 0}, {4.682143e-04, 1.136926e-04, -5.819068e-04, 1.816351e-13}};
        const char* names[4] = {"dE", "dB", "dK_electrons", "dE+dB+dK"};
        for (int i = 0; i < 4; ++i) add(13, names[i], "% .6e", v[row][i]);
        ++row;
        return 0;
      }
    } tab("/tmp/xpic_selftest/temporal/t.txt");
    tab.diagnose(0);
    tab.diagnose(1);
    tab.finalize();
  }
  std::ifstream f("/tmp/xpic_selftest/temporal/t.txt");
  std::string l0, l1, l2;
  std::getline(f, l0); std::getline(f, l1); std::getline(f, l2);
  expect(l0 == "Time    dE             dB             dK_electrons   dE+dB+dK", "table header");
  expect(l1 == "  0      0.000000e+00   0.000000e+00   0.000000e+00   0.000000e+00", "table row 0");
  expect(l2 == "  1      4.682143e-04   1.136926e-04  -5.819068e-04   1.816351e-13", "table row 1");
  if (!bad) std::cout << "selftest ok\n";
  return bad ? 1 : 0;
}

int main(int argc, char** argv)
{
  if (argc == 2 && std::string(argv[1]) == "--selftest") return selftest();
  if (argc < 2) {
    std::cerr << "usage: " << argv[0] << " <config.json> [output directory]\n";
    return 2;
  }
  try {
    Configuration::init(argv[1]);
    if (argc > 2) Configuration::set_out_dir(argv[2]);
    std::unique_ptr<interfaces::Simulation> simulation = build_simulation();
    int rc = simulation->initialize();
    if (rc == 0) rc = simulation->calculate();
    if (rc == 0) rc = simulation->finalize();
    if (rc != 0) {
      std::cerr << "xpic_hip: error code " << rc << ": " << xpic_last_error() << std::endl;
      return 1;
    }
  }
  catch (const std::exception& e) {
    std::cerr << "what(): " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
