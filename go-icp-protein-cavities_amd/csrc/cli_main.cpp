// cli_main.cpp -- command line with the reference's argument order (reference jly_main.cpp:181-229, README.md:17):
//   goicp_cli <TARGET mol2> <SOURCE mol2> <Nd downsampled> <CONFIG> <OUTPUT> <PAIR#> [--device N] [--quiet]
// run from a directory that holds cavities/ cfpfh/ cavitiesN/ (as the reference requires), or
//   goicp_cli --txt <MODEL txt> <DATA txt> <Nd> <CONFIG> <OUTPUT>      (upstream Go-ICP demo clouds)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "goicp_c.h"

int main(int argc, char** argv) {
    std::vector<std::string> a;
    int device = 0, verbose = 1; bool txt = false;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--quiet")) verbose = 0;
        else if (!strcmp(argv[i], "--txt")) txt = true;
        else a.push_back(argv[i]);
    }
    // defaults of parseInput, jly_main.cpp:183-188
    std::string model = a.size() > 0 ? a[0] : "model.txt", data = a.size() > 1 ? a[1] : "data.txt";
    int nd = a.size() > 2 ? atoi(a[2].c_str()) : 0;
    std::string config = a.size() > 3 ? a[3] : "config.txt", output = a.size() > 4 ? a[4] : "output.txt";
    int pair = a.size() > 5 ? atoi(a[5].c_str()) : 1;
    double r[24];
    int rc = txt ? goicp_run_txt_files(model.c_str(), data.c_str(), nd, config.c_str(), output.c_str(), device, verbose, r)
                 : goicp_run_pair_files("", model.c_str(), data.c_str(), nd, config.c_str(), output.c_str(), pair, device, verbose, r);
    if (rc != GOICP_OK) { fprintf(stderr, "goicp_cli: failed with status %d (2 = no usable MI355X / HIP device)\n", rc); return rc; }
    printf("Optimal Rotation Matrix:\n%12.7f %12.7f %12.7f \n%12.7f %12.7f %12.7f \n%12.7f %12.7f %12.7f \n", r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]);
    printf("Optimal Translation Vector:\n%12.7f \n%12.7f \n%12.7f \n", r[9], r[10], r[11]);
    printf("Error: %g  Compatibilities: %d  Finished in %g s (DT %g s)  cube-evals: %.0f\n", r[12], (int)r[13], r[15], r[16], r[20]);
    return 0;
}
