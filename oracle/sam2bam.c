/* oracle/sam2bam.c — TEST INFRASTRUCTURE (fixture generation only).
 *
 * Converts a text SAM into BAM (+BAI when the input is coordinate sorted) using the
 * htslib 1.19 archive that ships with the reference checkout (/root/reference/lib/libhts.a).
 * It exists only so that fixtures can be produced in the dev container, where no samtools
 * is installed; the binary lands in oracle/_ref/ and is never used by the product path.
 *
 *   usage: sam2bam in.sam out.bam [noindex]
 */
#include <stdio.h>
#include <string.h>
#include "sam.h"

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.sam out.bam [noindex]\n", argv[0]); return 2; }
    samFile *in = sam_open(argv[1], "r");
    if (!in) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
    sam_hdr_t *hdr = sam_hdr_read(in);
    if (!hdr) { fprintf(stderr, "cannot read header of %s\n", argv[1]); return 1; }
    samFile *out = sam_open(argv[2], "wb");
    if (!out) { fprintf(stderr, "cannot open %s for writing\n", argv[2]); return 1; }
    if (sam_hdr_write(out, hdr) < 0) { fprintf(stderr, "header write failed\n"); return 1; }
    bam1_t *rec = bam_init1();
    long n = 0;
    int r;
    while ((r = sam_read1(in, hdr, rec)) >= 0) {
        if (sam_write1(out, hdr, rec) < 0) { fprintf(stderr, "record write failed\n"); return 1; }
        ++n;
    }
    if (r < -1) { fprintf(stderr, "parse error after %ld records\n", n); return 1; }
    bam_destroy1(rec);
    sam_hdr_destroy(hdr);
    sam_close(in);
    if (sam_close(out) < 0) { fprintf(stderr, "close failed\n"); return 1; }
    if (argc < 4 || strcmp(argv[3], "noindex") != 0) {
        if (sam_index_build(argv[2], 0) < 0) { fprintf(stderr, "index build failed\n"); return 1; }
    }
    fprintf(stderr, "sam2bam: %ld records\n", n);
    return 0;
}
