// regions.cpp — see regions.h
#include "regions.h"
#include <stdlib.h>
#include <algorithm>
#include <iostream>
#include <sstream>
#include <thread>

namespace pdh {

namespace {

void split_any(const std::string &s, const char *delims, std::vector<std::string> *tok)
{
    tok->clear();
    size_t i = s.find_first_not_of(delims);
    while (i != std::string::npos) {
        const size_t e = s.find_first_of(delims, i);
        tok->push_back(s.substr(i, e == std::string::npos ? std::string::npos : e - i));
        if (e == std::string::npos) break;
        i = s.find_first_not_of(delims, e);
    }
}

void erase_all(std::string *s, char c)
{
    size_t w = 0;
    for (size_t r = 0; r < s->size(); ++r) if ((*s)[r] != c) (*s)[w++] = (*s)[r];
    s->resize(w);
}

void add_entry(RegionModel *rm, const RefSeqs *ref, int32_t tid, const std::string &id, long long start, long long end)
{
    Gene &g = rm->genes[tid][id];
    const int32_t s = (int32_t)start, e = (int32_t)end;
    if (g.cds.empty()) {
        g.start = s; g.end = e;
        if (ref) g.gc = (int32_t)ref->gc(tid, (int32_t)start, (int32_t)end);      // `for (int ii = Start-1; ii < End; ii++)`
    }
    else { if (g.start > s) g.start = s; if (g.end < e) g.end = e; }
    g.length += (uint64_t)(end - start + 1);
    g.cds.emplace_back(s, e);
}

void unknown_contig(const std::string &line)
{
    std::cerr << line << "Warning: This region may be incorrect.\n" << std::endl;
}

} // namespace

static void merge_spans(RegionModel *rm)
{
    // PD:3912-3972: per contig, spans keyed by start (largest end wins), merged when the next
    // start is <= the current end; a span starting at end+1 stays separate.
    for (auto &kv : rm->genes) {
        std::map<int32_t, int32_t> by_start;
        for (auto &g : kv.second) {
            auto it = by_start.find(g.second.start);
            if (it == by_start.end()) by_start[g.second.start] = g.second.end;
            else if (g.second.end > it->second) it->second = g.second.end;
        }
        std::vector<std::pair<int32_t, int32_t>> out;
        for (auto &se : by_start) {
            if (out.empty() || se.first > out.back().second) out.emplace_back(se.first, se.second);
            else if (se.second > out.back().second) out.back().second = se.second;
        }
        rm->merged[kv.first] = out;
    }
}

bool build_regions(Options *o, const AlnHeader &hdr, RegionModel *rm, RefSeqs *ref, int threads,
                   const std::map<std::string, int32_t> *names)
{
    std::map<std::string, int32_t> chr2tid;
    if (names) chr2tid = *names;
    else {
        for (size_t i = 0; i < hdr.names.size(); ++i) chr2tid.insert({hdr.names[i], (int32_t)i});   // first name wins
        if (ref && !load_reference(o->reference, &chr2tid, ref)) return false;
    }

    if (o->mode != 0) {
        std::vector<std::string> lines;
        // PD:3550-3555 tests `!LIST.good()`, which the reference's gzstream never sets for a file it could not open: the
        // target list is then simply empty (and an empty list falls back to whole-chromosome mode below)
        (void)read_lines(o->region_file, &lines);
        // these live across lines in the reference too: a short line re-uses the previous values
        std::string chr, id, start_s, end_s;
        int bstart = 0, bend = 0;
        std::istringstream is;
        for (std::string &line : lines) {
            if (line.empty()) continue;
            if (line[0] == '#') continue;
            if (o->mode == 1) {                                   // GFF3, PD:3557-3647
                is.clear(); is.str(line);                    // one stream object for all lines (constructing one per line costs ~1 us)
                std::string f2, feat, strand, attr;
                long long s = 0, e = 0;
                is >> chr >> f2 >> feat;
                if (feat != o->feature) continue;
                is >> s >> e >> f2 >> strand >> f2 >> attr;
                std::vector<std::string> inf, kv;
                split_any(attr, ",;", &inf);
                if (inf.empty()) continue;
                split_any(inf[0], "=", &kv);
                std::string gid = kv.empty() ? std::string() : kv.back();
                for (size_t j = 1; j < inf.size(); ++j) {
                    split_any(inf[j], "=", &kv);
                    if (!kv.empty() && kv[0] == "Parent") gid = kv.back();
                }
                auto it = chr2tid.find(chr);
                if (it == chr2tid.end()) unknown_contig(line);
                else add_entry(rm, ref, it->second, gid, s, e);
            } else if (o->mode == 2) {                            // GTF, PD:3649-3740
                erase_all(&line, '"');
                erase_all(&line, ';');
                is.clear(); is.str(line);                    // one stream object for all lines (constructing one per line costs ~1 us)
                std::string f2, feat;
                long long s = 0, e = 0;
                is >> chr >> f2 >> feat;
                if (feat != o->feature) continue;
                is >> s >> e;
                std::vector<std::string> inf;
                split_any(line, "\t ", &inf);
                if (inf.size() < 10) continue;
                auto it = chr2tid.find(chr);
                if (it == chr2tid.end()) unknown_contig(line);
                else add_entry(rm, ref, it->second, inf[9], s, e);
            } else if (o->mode == 3) {                            // BED3, PD:3741-3819
                is.clear(); is.str(line);                    // one stream object for all lines (constructing one per line costs ~1 us)
                is >> chr >> start_s >> end_s;
                id = chr + "_" + start_s + "_" + end_s;
                bstart = atoi(start_s.c_str()); bend = atoi(end_s.c_str());
                if (bstart > bend) { std::cerr << line << "Warning: This region may be incorrect.\n" << std::endl; continue; }
                auto it = chr2tid.find(chr);
                if (it == chr2tid.end()) unknown_contig(line);
                else add_entry(rm, ref, it->second, id, bstart, bend);
            } else if (o->mode == 4) {                            // BED4, PD:3821-3898
                is.clear(); is.str(line);                    // one stream object for all lines (constructing one per line costs ~1 us)
                is >> chr >> bstart >> bend >> id;
                if (bstart > bend) { std::cerr << line << "Warning: This region may be incorrect. \n" << std::endl; continue; }
                auto it = chr2tid.find(chr);
                if (it == chr2tid.end()) unknown_contig(line);
                else add_entry(rm, ref, it->second, id, bstart, bend);
            }
        }
    }
    merge_spans(rm);
    if (rm->merged.empty()) {
        // no targets: whole-contig bins (PD:3974-4051)
        int width = 10000000;
        if (o->win == 0) o->mode = 0;
        else if (o->win < 150) o->mode = 6;
        else { o->mode = 5; width = o->win; }
        for (size_t i = 0; i < hdr.names.size(); ++i) {
            const long long len = hdr.lens[i];
            long long start = 1, end = 2;
            // the loop condition tests the PREVIOUS bin's end + 2: contigs shorter than 2 get no bin,
            // and a final 1-base bin (len = k*width + 1) is never created
            std::vector<Bin> *v = nullptr;
            std::vector<std::pair<int32_t, int32_t>> *m = nullptr;
            for (start = 1; end <= len; start += width) {
                end = start + width - 1;
                if (end > len) end = len;
                if (!v) { v = &rm->bins[(int32_t)i]; m = &rm->merged[(int32_t)i]; }
                Bin b; b.start = (int32_t)start; b.end = (int32_t)end;
                v->push_back(b);
                // merge rule of PD:3959: the next bin starts at end+1 > end, so every bin stays its own span
                m->emplace_back((int32_t)start, (int32_t)end);
                end += 2;
            }
        }
        if (ref) {
            // PD:4017-4023: every bin counts its own bases; whole-genome passes, so spread over the threads
            std::vector<Bin *> all;
            for (auto &kv : rm->bins) for (Bin &b : kv.second) all.push_back(&b);
            std::vector<int32_t> tid_of;
            for (auto &kv : rm->bins) tid_of.insert(tid_of.end(), kv.second.size(), kv.first);
            const size_t T = std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), all.size()));
            auto work = [&](size_t s) {
                for (size_t k = s; k < all.size(); k += T) all[k]->gc = (int32_t)ref->gc(tid_of[k], all[k]->start, all[k]->end);
            };
            std::vector<std::thread> th;
            for (size_t s = 1; s < T; ++s) th.emplace_back(work, s);
            work(0);
            for (auto &x : th) x.join();
        }
    }
    return true;
}

} // namespace pdh
