// Reads every file named on the command line with one of the host readers (gfa | og | lay); built with
// -fsanitize=address,undefined by tests/test_host_logic.py::test_readers_under_sanitizers and fed damaged files.
#include "pgsgd.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
int main(int argc, char** argv) {
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[1], "gfa")) { pgsgd_graph* g = nullptr; if (pgsgd_graph_from_gfa(argv[i], 2, &g) == 0) pgsgd_graph_free(g); }
        else if (!strcmp(argv[1], "og")) { pgsgd_graph* g = nullptr; if (pgsgd_graph_from_og(argv[i], 2, &g) == 0) pgsgd_graph_free(g); }
        else { uint64_t n; double *X, *Y; if (pgsgd_read_lay(argv[i], &n, &X, &Y) == 0) { pgsgd_free(X); pgsgd_free(Y); } }
    }
    return 0;
}
