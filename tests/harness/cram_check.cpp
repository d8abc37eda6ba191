// tests/harness/cram_check.cpp — TEST INFRASTRUCTURE: prints (tid, pos, flag, mapq, reference-consuming shape) of every
// record of a SAM / BAM (AlnReader) or CRAM (CramReader) file, one line each, so that the CRAM reader can be compared
// with the SAM text the CRAM was made from.
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include "../../pandepth_amd/host/bam.h"
#include "../../pandepth_amd/host/cram.h"
using namespace pdh;
static void dump(const AlnRec &r)
{
    printf("%d\t%d\t%u\t%u\t", r.tid, r.pos, r.flag, r.mapq);
    char last = 0; long len = 0;
    for (uint32_t i = 0; i < r.n_cigar; ++i) {
        const uint32_t op = r.cigar[i] & 0xf; const long n = r.cigar[i] >> 4;
        char c = (op == 0 || op == 7 || op == 8) ? 'M' : op == 2 ? 'D' : op == 3 ? 'N' : 0;
        if (!c || !n) continue;
        if (c == last) len += n; else { if (last) printf("%ld%c", len, last); last = c; len = n; }
    }
    if (last) printf("%ld%c", len, last);
    printf("\n");
}
int main(int argc, char **argv)
{
    std::string err; AlnRec r;
    if (CramReader::is_cram(argv[1])) {
        CramReader c; AlnHeader h;
        if (!c.open(argv[1], &h, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        if (argc > 2) c.set_threads(atoi(argv[2]));            // containers decoded ahead on helper threads
        for (size_t i = 0; i < h.names.size(); ++i) printf("@%s\t%u\n", h.names[i].c_str(), h.lens[i]);
        int rc; while ((rc = c.next(&r)) == 1) dump(r);
        if (rc < 0) { fprintf(stderr, "error: %s\n", c.error().c_str()); return 1; }
        return 0;
    }
    AlnReader a;
    if (!a.open(argv[1], &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    for (size_t i = 0; i < a.header().names.size(); ++i) printf("@%s\t%u\n", a.header().names[i].c_str(), a.header().lens[i]);
    int rc; while ((rc = a.next(&r)) == 1) dump(r);
    return rc < 0;
}
