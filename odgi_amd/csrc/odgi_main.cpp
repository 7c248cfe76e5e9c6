// odgi_main.cpp — `odgi layout ...` entry point: the one subcommand this build provides
// (the reference dispatches on argv[1] through its registry, src/main.cpp:36-54).
#include <cstdio>
#include <cstring>

#include "../../include/pgsgd.h"

int main(int argc, char** argv) {
    if (argc < 2 || strcmp(argv[1], "layout") != 0) {
        fprintf(stderr, "odgi (MI355X layout build): only the 'layout' subcommand is available.\nusage: odgi layout -i graph.gfa -o graph.lay [-T graph.tsv] [options]\n");
        return 1;
    }
    return pgsgd_main_layout(argc, argv);
}
