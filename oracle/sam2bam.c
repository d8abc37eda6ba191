/* oracle/sam2bam.c — TEST INFRASTRUCTURE (fixture generation only).
 *
 * Converts a text SAM into BAM (+BAI when the input is coordinate sorted) using the
 * htslib 1.19 archive that ships with the reference checkout (/root/reference/lib/libhts.a).
 * It exists only so that fixtures can be produced in the dev container, where no samtools
 * is installed; the binary lands in oracle/_ref/ and is never used by the product path.
 *
 *   usage: sam2bam in.sam out.bam [noindex]
 *          sam2bam in.sam|in.bam out.cram [noindex] [ref=genome.fa]     CRAM 3.0; without ref= the file is written
 *                                                                      reference-free (CRAM_OPT_NO_REF)
 *                  [sps=N] records per slice  [spc=N] slices per container  [multiseq] multi-reference slices
 *                  [embedref] (with ref=) the reference bases travel inside the slices
 *                  [fmt=cram,version=3.1,small ...] an htslib format string instead of plain CRAM 3.0
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sam.h"

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.sam out.bam [noindex]\n", argv[0]); return 2; }
    samFile *in = sam_open(argv[1], "r");
    if (!in) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
    sam_hdr_t *hdr = sam_hdr_read(in);
    if (!hdr) { fprintf(stderr, "cannot read header of %s\n", argv[1]); return 1; }
    const size_t ol = strlen(argv[2]);
    const int cram = ol > 5 && strcmp(argv[2] + ol - 5, ".cram") == 0;
    int noindex = 0, sps = 0, spc = 0, multiseq = 0, embedref = 0;
    const char *ref = NULL, *fmtstr = NULL;
    for (int k = 3; k < argc; ++k) {
        if (strcmp(argv[k], "noindex") == 0) noindex = 1;
        else if (strncmp(argv[k], "ref=", 4) == 0) ref = argv[k] + 4;
        else if (strncmp(argv[k], "sps=", 4) == 0) sps = atoi(argv[k] + 4);
        else if (strncmp(argv[k], "spc=", 4) == 0) spc = atoi(argv[k] + 4);
        else if (strcmp(argv[k], "multiseq") == 0) multiseq = 1;
        else if (strcmp(argv[k], "embedref") == 0) embedref = 1;
        else if (strncmp(argv[k], "fmt=", 4) == 0) fmtstr = argv[k] + 4;
    }
    htsFormat fmt;
    memset(&fmt, 0, sizeof fmt);
    if (fmtstr && hts_parse_format(&fmt, fmtstr) < 0) { fprintf(stderr, "bad format string %s\n", fmtstr); return 1; }
    samFile *out = fmtstr ? sam_open_format(argv[2], "wc", &fmt) : sam_open(argv[2], cram ? "wc" : "wb");
    if (!out) { fprintf(stderr, "cannot open %s for writing\n", argv[2]); return 1; }
    if (cram) {
        if (ref) { if (hts_set_fai_filename(out, ref) < 0) { fprintf(stderr, "cannot use reference %s\n", ref); return 1; } }
        else hts_set_opt(out, CRAM_OPT_NO_REF, 1);
        if (sps > 0) hts_set_opt(out, CRAM_OPT_SEQS_PER_SLICE, sps);
        if (spc > 0) hts_set_opt(out, CRAM_OPT_SLICES_PER_CONTAINER, spc);
        if (multiseq) hts_set_opt(out, CRAM_OPT_MULTI_SEQ_PER_SLICE, 1);
        if (embedref && ref) hts_set_opt(out, CRAM_OPT_EMBED_REF, 1);
    }
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
    if (!noindex) {
        if (sam_index_build(argv[2], 0) < 0) { fprintf(stderr, "index build failed\n"); return 1; }
    }
    fprintf(stderr, "sam2bam: %ld records\n", n);
    return 0;
}
