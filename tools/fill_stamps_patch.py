#!/usr/bin/env python3
"""Temporarily instruments k_ecsim_fill with clock64() stamps (XPIC_FILL_STAMPS=<file>); summarise with tools/stamps.py.
Apply, build, run bench.py, then `git checkout xpic_amd/csrc/ecsim.hip`."""
import re
p = 'xpic_amd/csrc/ecsim.hip'
s = open(p).read()
def rep(a, b, count=1):
    global s
    assert a in s, a
    s = s.replace(a, b, count)
rep("int per_y, int per_z, int first_sort)\n{", "int per_y, int per_z, int first_sort, long long* stamps)\n{\n#define STAMP(k) do { if (stamps && blockIdx.x % 97 == 0 && blockIdx.x / 97 < 32 && lane == 0 && j >= 8 && j < 40) stamps[(((blockIdx.x / 97) * 32 + (j - 8)) * 4 + wave) * 8 + (k)] = clock64(); } while (0)\n")
rep("    mfma_acc acc[3][2];\n", "    STAMP(0);\n    mfma_acc acc[3][2];\n")
rep("    // next chunk's cell: particle data and B neighbourhood travel", "    STAMP(1);\n    // next chunk's cell: particle data and B neighbourhood travel")
rep("    lds_barrier();\n    double* win = sh;", "    STAMP(2);\n    lds_barrier();\n    STAMP(3);\n    double* win = sh;")
rep("    lds_barrier();\n    if (active) {\n      // row node x", "    lds_barrier();\n    STAMP(4);\n    if (active) {\n      // row node x")
rep("    lds_barrier();\n    // ---- stream out the finished columns", "    lds_barrier();\n    STAMP(5);\n    // ---- stream out the finished columns")
rep("    lds_barrier();\n  }\n\n  // ---- the two columns still carried", "    STAMP(6);\n    lds_barrier();\n    STAMP(7);\n  }\n\n  // ---- the two columns still carried")
rep("per_y, per_z, first_sort ? 1 : 0);", "per_y, per_z, first_sort ? 1 : 0, (a == 0 && b == 0) ? dbg_stamps : nullptr);")
rep('  Timed t(c, "fill_current");', '''  static long long* dbg_stamps = nullptr;
  static int dbg_calls = 0;
  if (getenv("XPIC_FILL_STAMPS") && !dbg_stamps) { (void)hipMalloc(&dbg_stamps, 32 * 32 * 4 * 8 * 8); (void)hipMemset(dbg_stamps, 0, 32 * 32 * 4 * 8 * 8); }
  if (dbg_stamps && ++dbg_calls == 3) {
    std::vector<long long> h(32 * 32 * 4 * 8);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), dbg_stamps, h.size() * 8, hipMemcpyDeviceToHost);
    FILE* f = fopen(getenv("XPIC_FILL_STAMPS"), "w");
    for (size_t i = 0; i < h.size(); i += 8) { for (int k = 0; k < 8; ++k) fprintf(f, "%lld ", h[i + k]); fprintf(f, "\\n"); }
    fclose(f);
  }
  Timed t(c, "fill_current");''')
rep("#include <cstdint>\n", "#include <cstdint>\n#include <cstdio>\n#include <cstdlib>\n#include <vector>\n")
open(p, 'w').write(s)
