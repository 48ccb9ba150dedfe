// main.cpp -- mirror of the reference's src/main.cpp:9-40: configuration -> build_simulation -> initialize ->
// calculate -> finalize, exceptions reported and turned into a non-zero exit.
//   usage: xpic_hip.out <config.json> [output directory override]
#include <iostream>

#include "xpic_host.h"

int main(int argc, char** argv)
{
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
