// options.cpp — see options.h
#include "options.h"
#include <string.h>
#include <stdlib.h>
#include <zlib.h>
#include <iostream>
#include <map>
#include <mutex>
#include <sstream>

namespace pdh {

namespace {
std::mutex g_tune_mu;
std::map<std::string, std::string> g_tune;
bool g_tune_env_read = false;
void tune_parse(const std::string &list)
{
    size_t o = 0;
    while (o <= list.size()) {
        size_t e = list.find(',', o);
        if (e == std::string::npos) e = list.size();
        const std::string kv = list.substr(o, e - o);
        const size_t q = kv.find('=');
        if (!kv.empty()) g_tune[q == std::string::npos ? kv : kv.substr(0, q)] = q == std::string::npos ? std::string("1") : kv.substr(q + 1);
        o = e + 1;
    }
}
void tune_env()
{
    if (g_tune_env_read) return;
    g_tune_env_read = true;
    if (const char *e = getenv("PANDEPTH_TUNE")) tune_parse(e);
    // the development switches these variables used to be (rounds 1-3) are keys of PANDEPTH_TUNE / -X now: a script that still sets one
    // would silently get the default path instead of the one it asked for
    static const char *const legacy[][2] = {
        {"PANDEPTH_DEVICE_DECODE", "device_decode"}, {"PANDEPTH_DEVICE_DEFLATE", "device_deflate"}, {"PANDEPTH_DD_BATCH_MB", "dd_batch_mb"}, {"PANDEPTH_DD_THREADS", "dd_threads"},
        {"PANDEPTH_NO_RCCL", "rccl=0"}, {"PANDEPTH_FORCE_RCCL", "rccl=force"}, {"PANDEPTH_RCCL_VERBOSE", "rccl_verbose"}, {"PANDEPTH_GPUS", "gpus"},
        {"PANDEPTH_SITE_RESIDENT", "site_resident"}, {"PANDEPTH_SITE_OVERLAP", "site_overlap"}, {"PANDEPTH_SITE_IDENTICAL", "site_identical"}, {"PANDEPTH_SITE_PARALLEL_MIN", "site_parallel_min"},
        {"PANDEPTH_TABLE_RESIDENT", "table_resident"}, {"PANDEPTH_TABLE_RESIDENT_MIN", "table_resident_min"}, {"PANDEPTH_DECODE_ONLY", "decode_only"}};
    for (const auto &l : legacy)
        if (getenv(l[0])) fprintf(stderr, "pandepth: note: %s is no longer read; use PANDEPTH_TUNE=%s%s (or -X)\n", l[0], l[1], strchr(l[1], '=') ? "" : "=<value>");
}
} // namespace

const char *tune(const char *key)
{
    std::lock_guard<std::mutex> lk(g_tune_mu);
    tune_env();
    auto it = g_tune.find(key);
    return it == g_tune.end() ? nullptr : it->second.c_str();       // (entries are never removed: the pointer stays valid)
}
long long tune_int(const char *key, long long dflt) { const char *v = tune(key); return v ? strtoll(v, nullptr, 10) : dflt; }
void tune_add(const std::string &kv_list) { std::lock_guard<std::mutex> lk(g_tune_mu); tune_env(); tune_parse(kv_list); }

void print_help()
{
    std::cout <<
        "Usage: pandepth -i in.bam [-g gene.gff | -b region.bed] -o outPrefix\n"
        " Input/Output options:\n"
        "   -i    <str>     input of sam/bam/cram/paf or #.list file\n"
        "   -o    <str>     prefix of output file\n"
        " Target options:\n"
        "   -g    <str>     input gff/gtf file for gene region\n"
        "   -f    <str>     gff/gtf feature type to parse, CDS or exon [CDS]\n"
        "   -b    <str>     input bed file for list of regions\n"
        "   -w    <int>     windows size (bp)\n"
        "   -a              output all the site depth\n"
        " Filter options:\n"
        "   -q    <int>     min mapping quality [0]\n"
        "   -d    <int>     min site depth for statistics [1]\n"
        "   -x    <int>     exclude reads with any of the bits in FLAG set [1796]\n"
        " Other options:\n"
        "   -t    <int>     number of host reader threads [3]\n"
        "   -r    <str>     reference genome file for GC parse (cram is decoded without it)\n"
        "   -c              enable the calculation of GC content (requires -r)\n"
        "   -h              show this help [MI355X engine, PanDepth v2.26 compatible]\n"
        "\n";
}

bool read_lines(const std::string &path, std::vector<std::string> *lines)
{
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    std::string data;
    char buf[1 << 16];
    int n;
    while ((n = gzread(f, buf, sizeof buf)) > 0) data.append(buf, (size_t)n);
    gzclose(f);
    lines->clear();
    size_t o = 0;
    for (;;) {
        const size_t e = data.find('\n', o);
        if (e == std::string::npos) { lines->push_back(data.substr(o)); break; }
        lines->push_back(data.substr(o, e - o));
        o = e + 1;
    }
    return true;
}

static void split_ws(const std::string &s, std::vector<std::string> *tok)
{
    tok->clear();
    size_t i = s.find_first_not_of(" \t");
    while (i != std::string::npos) {
        const size_t e = s.find_first_of(" \t", i);
        tok->push_back(s.substr(i, e == std::string::npos ? std::string::npos : e - i));
        if (e == std::string::npos) break;
        i = s.find_first_not_of(" \t", e);
    }
}

static std::string ext_of(const std::string &p)
{
    const size_t d = p.rfind('.');
    return d == std::string::npos ? std::string() : p.substr(d + 1);
}

int parse_options(int argc, char **argv, Options *o)
{
    if (argc <= 1) { print_help(); return 0; }
    int bed_count = 0, n_inputs = 0;
    std::vector<std::string> bed_list;
    auto lack = [](const std::string &f) { std::cerr << "Error: Lack argument for [ -" << f << " ]" << std::endl; };
    for (int i = 1; i < argc; ++i) {
        if (argv[i][0] != '-') {
            std::cerr << "Error: Command option error! Please check the provided options." << std::endl;
            return 0;
        }
        std::string flag;
        for (const char *p = argv[i]; *p; ++p) if (*p != '-') flag += *p;      // every '-' removed (PD:97)
        auto arg = [&](std::string *dst) -> bool {
            if (i + 1 == argc) { lack(flag); return false; }
            *dst = argv[++i];
            return true;
        };
        std::string v;
        if (flag == "i") {
            if (!arg(&v)) return 0;
            const std::string ext = ext_of(v);
            if (ext == "list" || ext == "List") {
                std::vector<std::string> ls;
                if (!read_lines(v, &ls)) std::cerr << "open List error: " << v << std::endl;
                else for (auto &l : ls) if (!l.empty()) { o->inputs.push_back(l); ++n_inputs; }
            } else {
                ++n_inputs; o->inputs.push_back(v); o->input = v;
            }
        } else if (flag == "o") { if (!arg(&o->out)) return 0; }
        else if (flag == "c") o->gc = true;
        else if (flag == "a") o->site_out = true;
        else if (flag == "r") { if (!arg(&o->reference)) return 0; }
        else if (flag == "f") { if (!arg(&o->feature)) return 0; }
        else if (flag == "x") { if (!arg(&v)) return 0; o->flag_mask = (uint32_t)atoi(v.c_str()); }
        else if (flag == "g") {
            if (!arg(&v)) return 0;
            o->region_file = v;
            std::vector<std::string> ls;
            // a file that cannot be opened does not trip the reference's `INGFF.fail()` check (PD:154-158, its gzstream
            // reports nothing): it reads no line and ends in the format message below
            (void)read_lines(v, &ls);
            // sniff the first 167 lines: the LAST line mentioning Parent / transcript_id decides (PD:162-181)
            for (size_t k = 0; k < ls.size() && k < 167; ++k) {
                const std::string &t = ls[k];
                if (t.length() < 2 || t[0] == '#') continue;
                if (t.find("Parent") != std::string::npos) o->mode = 1;
                else if (t.find("transcript_id") != std::string::npos) o->mode = 2;
            }
            if (o->mode == 0) {
                std::cerr << "Error: The format of the input GFF/GTF file is incorrect. Please check the file format: " << v << std::endl;
                return 0;
            }
        } else if (flag == "b") { if (!arg(&v)) return 0; ++bed_count; bed_list.push_back(v); o->mode = 3; }
        else if (flag == "t") { if (!arg(&v)) return 0; o->threads = atoi(v.c_str()); }
        else if (flag == "w") {
            if (!arg(&v)) return 0;
            o->win = atoi(v.c_str());
            if (o->win < 1) { std::cerr << "Warning: -w should >= 1, set to 1\n"; o->win = 1; }
        } else if (flag == "q") { if (!arg(&v)) return 0; o->min_mapq = atoi(v.c_str()); }
        else if (flag == "s") o->use_index = false;
        else if (flag == "X") { if (!arg(&v)) return 0; tune_add(v); }          // hidden: development switches (options.h)
        else if (flag == "d") { if (!arg(&v)) return 0; o->min_dep = atoi(v.c_str()); if (o->min_dep < 1) o->min_dep = 1; }
        else if (flag == "help" || flag == "h") { print_help(); return 0; }
        else { std::cerr << "Error UnKnow argument -" << flag << std::endl; return 0; }
    }
    if (n_inputs > 0) o->input = o->inputs[0];
    if (o->input.empty() || o->out.empty()) { std::cerr << "Error: lack argument -i or -o " << std::endl; return 0; }
    if (bed_count != 0 && o->region_file.empty()) {
        o->region_file = bed_list[0];
        std::vector<std::string> ls;
        // an unreadable file does not trip the reference's `!LISTTT.good()` (PD:268-273, its gzstream reports nothing):
        // it is an empty BED3 list, and an empty target list falls back to whole-chromosome mode
        (void)read_lines(o->region_file, &ls);
        std::vector<std::string> t1, t2;
        split_ws(ls.size() > 0 ? ls[0] : std::string(), &t1);
        split_ws(ls.size() > 1 ? ls[1] : std::string(), &t2);
        if (t1.size() == 4 || t2.size() == 4) o->mode = 4;
    }
    if (ext_of(o->out) != "gz") o->out += ".gz";
    return n_inputs;
}

} // namespace pdh
