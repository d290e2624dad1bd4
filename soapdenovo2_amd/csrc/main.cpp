// main.cpp -- `${bin} pregraph -s config -K k -o prefix ...` (standardPregraph/main.c:59-104 dispatches the same
// way).  Built twice: SOAPdenovo-63mer (call_pregraph) and SOAPdenovo-127mer (-DPG_MER127, call_pregraph_127mer).
// Only the pregraph sub-command lives here; contig / map / scaff are the reference's unchanged stages and
// consume the files this one writes.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/soapdenovo2_amd.h"
#include "env.hpp"

static void usage(void) {
    fprintf(stderr, "\n%s\n\nUsage: SOAPdenovo <command> [option]\n", pg_version());
    fprintf(stderr, "    pregraph        construct kmer-graph (MI355X)\n");
    fprintf(stderr, "  (sparse_pregraph, contig, map, scaff, all: run the reference binary on the files written here)\n");
}

int main(int argc, char** argv) {
    if (argc < 2) { usage(); return 1; }
    if (strcmp(argv[1], "pregraph") == 0) {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        pg_process_exits_after_this(1);              // nothing follows in this process: big tables are left to the exit path
#ifdef PG_MER127
        const int rc = call_pregraph_127mer(argc - 1, argv + 1);
#else
        const int rc = call_pregraph(argc - 1, argv + 1);
#endif
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (pg::env_user("PG_HOST_VERBOSE")) fprintf(stderr, "[cli] call_pregraph returned after %.2fs\n", (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
        return rc;
    }
    fprintf(stderr, "Command '%s' is not part of this build.\n", argv[1]);
    usage();
    return 1;
}
