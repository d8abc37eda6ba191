// tools/sitecheck.cpp — BENCH TOOLING (not product code): is a per-site file (<prefix>.SiteDepth.gz, PD:4264-4284: "name\tindex\tdepth"
// for every cell) consistent with the window table written in the same run (<prefix>.win.stat.gz, PD:4366-4389)?  Streams both once:
//   * per-site: line count, and per contig the cells that the table's windows cover (the reference's window loop `for (j = 1; j < len;
//     j += w)` drops a final window of ONE base: when len % w == 1 the last cell belongs to no window), their covered count
//     (depth >= 1) and depth sum;  indices must run 0, 1, 2, ... inside a contig
//   * table: sums of the Length, CoveredSite and TotalDepth columns over all rows, and the window width from the first row
// Prints one line; exit 0 when the three sums agree.  Used by bench.py's full-size `-w 100 -a` leg when no reference hashes are at hand.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <string>

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: sitecheck x.SiteDepth.gz x.win.stat.gz\n"); return 2; }
    // ---- the table ----
    gzFile t = gzopen(argv[2], "rb");
    if (!t) { perror(argv[2]); return 2; }
    gzbuffer(t, 1 << 20);
    static char line[1 << 16];
    unsigned long long rows = 0, tl = 0, tc = 0; long long td = 0; long long w = 0;
    while (gzgets(t, line, sizeof line)) {
        if (line[0] == '#') continue;
        char *p = strchr(line, '\t'); if (!p) continue;
        const long long st = strtoll(p + 1, &p, 10), en = strtoll(p + 1, &p, 10), len = strtoll(p + 1, &p, 10), cov = strtoll(p + 1, &p, 10), dep = strtoll(p + 1, &p, 10);
        if (!w) w = en - st + 1;
        ++rows; tl += (unsigned long long)len; tc += (unsigned long long)cov; td += dep;
    }
    gzclose(t);
    if (w <= 0) { printf("sitecheck: no rows in the table\n"); return 1; }
    // ---- the per-site stream ----
    gzFile s = gzopen(argv[1], "rb");
    if (!s) { perror(argv[1]); return 2; }
    gzbuffer(s, 1 << 22);
    unsigned long long lines = 0, sl = 0, sc = 0, sd = 0, bad_index = 0;
    std::string cur; unsigned long long n_in = 0, last_depth = 0;
    auto close_contig = [&]() {
        if (!n_in) return;
        // cells 0 .. n_in - 1 (len = n_in); the windows cover all of them unless len % w == 1 (then not the last one); len < 2: none
        unsigned long long covered_cells = n_in;
        if (n_in < 2) covered_cells = 0; else if ((long long)(n_in % (unsigned long long)w) == 1 % w && w > 1) covered_cells = n_in - 1;
        if (covered_cells != n_in) {                  // take the last cell (or the only one) back out of the sums
            sl -= n_in - covered_cells;
            if (last_depth >= 1) { sc -= 1; sd -= last_depth; }
        }
    };
    while (gzgets(s, line, sizeof line)) {
        ++lines;
        char *p = strchr(line, '\t'); if (!p) { ++bad_index; continue; }
        const size_t nl = (size_t)(p - line);
        if (cur.size() != nl || memcmp(cur.data(), line, nl) != 0) { close_contig(); cur.assign(line, nl); n_in = 0; }
        const unsigned long long idx = strtoull(p + 1, &p, 10), dep = strtoull(p + 1, &p, 10);
        if (idx != n_in) ++bad_index;
        ++n_in; ++sl; if (dep >= 1) { ++sc; sd += dep; }
        last_depth = dep;
    }
    close_contig();
    gzclose(s);
    const bool ok = bad_index == 0 && sl == tl && sc == tc && (long long)sd == td;
    printf("sitecheck: %llu per-site lines, %llu table rows of %lld; cells under windows %llu vs RegionLength %llu; covered %llu vs %llu; depth %llu vs %lld; "
           "index errors %llu: %s\n", lines, rows, w, sl, tl, sc, tc, sd, td, bad_index, ok ? "CONSISTENT" : "MISMATCH");
    return ok ? 0 : 1;
}
