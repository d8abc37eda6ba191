// tests/harness/chain_check.cpp — TEST INFRASTRUCTURE: pdb2::check_chain (pandepth_amd/csrc/pd_bamwalk.h), the host half of
// the speculative record walk, against a model walker: a unit of back-to-back records of known sizes, cut in segments; the
// first walk of a segment "guesses" (right, or wrong at chosen segments: a false start a few bytes early whose chain stops,
// or one that lands on a later true record); a repeated walk follows the hint it is given — into garbage if the hint is.
// Checks: the loop ends with every segment confirmed at the true chain; with short records an isolated wrong guess costs ONE
// repeat whatever comes behind it and k wrong guesses in a row at most k + 1 (records longer than a segment leave segments
// without any start between two guesses, which then count as neighbours: a few more); a stopped confirmed chain ends the unit.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <random>
#include <set>
#include <vector>
#include "../../pandepth_amd/csrc/pd_bamwalk.h"

struct Model {
    std::vector<uint64_t> rec;                    // true record starts, ascending; rec.back() = end of the last record
    uint64_t stop_at = ~0ull;                     // a true record that cannot be walked (corrupt / hand-over)
    uint32_t stop_flag = 0;
    // walks segment s from `start` (NONE = nothing known): fills used_start / e_last / flags as walk_segment does
    void walk(pdb2::Seg &s, uint64_t start) const
    {
        s.flags = 0; s.n_rec = 0;
        if (start == pdb2::NONE) { s.used_start = pdb2::NONE; s.e_last = 0; return; }
        if (start >= s.end) { s.used_start = pdb2::NONE; s.e_last = start; return; }
        s.used_start = start;
        auto it = std::lower_bound(rec.begin(), rec.end(), start);
        if (it == rec.end() || *it != start) { s.e_last = pdb2::STOPPED; s.flags = pdb2::WF_MORE; return; }     // not a record: its "size" runs off
        uint64_t p = start;
        while (p < s.end) {
            if (p == stop_at) { s.e_last = pdb2::STOPPED; s.flags |= stop_flag; return; }
            ++it; ++s.n_rec;
            if (it == rec.end()) { s.e_last = pdb2::STOPPED; s.flags |= pdb2::WF_MORE; return; }
            p = *it;
        }
        s.e_last = p;
    }
    uint64_t first_in(uint64_t a, uint64_t b) const { auto it = std::lower_bound(rec.begin(), rec.end(), a); return it != rec.end() && *it < b && it + 1 != rec.end() ? *it : pdb2::NONE; }
};

static int run_case(const Model &m, uint64_t seg_bytes, const std::set<size_t> &wrong_early, const std::set<size_t> &wrong_late, int max_rounds, const char *what)
{
    std::vector<pdb2::Seg> segs;
    const uint64_t stop = m.rec[m.rec.size() - 1];
    for (uint64_t b = m.rec[0]; b < stop; b += seg_bytes) {
        pdb2::Seg s{}; s.begin = b; s.end = std::min(b + seg_bytes, stop); s.avail = stop; s.unit_first = b == m.rec[0]; s.hint = s.unit_first ? b : pdb2::NONE;
        segs.push_back(s);
    }
    for (size_t j = 0; j < segs.size(); ++j) {
        uint64_t g = segs[j].unit_first ? segs[j].begin : m.first_in(segs[j].begin, segs[j].end);
        if (wrong_early.count(j) && g != pdb2::NONE && g >= segs[j].begin + 2) g -= 2;               // a false start: not a record
        if (wrong_late.count(j)) { const uint64_t h = g == pdb2::NONE ? pdb2::NONE : m.first_in(g + 1, segs[j].end); if (h != pdb2::NONE) g = h; }   // skips a record
        if (wrong_late.count(j) && g == pdb2::NONE) g = segs[j].begin + 1;                           // a start where none is (inside a long record)
        m.walk(segs[j], g);
    }
    {   // the device form of the same confirmation (pdb2::chain_device, one wave; here its 64 lanes in a loop) on a copy: whenever it
        // does not leave the batch to the host, it must arrive where check_chain's rounds arrive, and give every segment its place
        std::vector<pdb2::Seg> dv = segs, hv = segs;
        for (auto &x : dv) x.n_first = x.n_rec;
        std::vector<int> member_ok(7, 0);
        pdb2::ChainOut co; memset(&co, 0xAB, sizeof co);
        uint32_t walks = 0;
        pdb2::chain_device<pdw::HostWave>(dv.data(), (uint32_t)dv.size(), member_ok.data(), (uint32_t)member_ok.size(), ~0ull, ~0ull, 1000000u,
            [&](uint32_t j, uint64_t start) { ++walks; m.walk(dv[j], start); dv[j].hint = start; dv[j].n_first = dv[j].n_rec;
                                              return pdb2::WalkOut{dv[j].used_start, dv[j].e_last, dv[j].n_first, 0u, 0u, dv[j].n_rec, dv[j].flags, 0u}; }, &co);
        std::vector<uint32_t> r2; int rr = 0;
        while (pdb2::check_chain(hv, &r2) > 0 && ++rr <= 64) for (uint32_t j : r2) m.walk(hv[j], hv[j].hint);
        bool host_flag = false;
        for (auto &x : hv) if (x.flags) host_flag = true;
        if (host_flag != (co.slow != 0)) { fprintf(stderr, "%s: device chain slow = %u, the host's chain %s a flag\n", what, co.slow, host_flag ? "carries" : "carries no"); return 1; }
        if (!co.slow) {
            uint64_t nf = 0, nr = 0;
            for (size_t j = 0; j < dv.size(); ++j) {
                if (dv[j].used_start != hv[j].used_start || dv[j].e_last != hv[j].e_last || dv[j].n_rec != hv[j].n_rec) { fprintf(stderr, "%s: device chain differs from the host's at segment %zu\n", what, j); return 1; }
                if (dv[j].base_first != nf) { fprintf(stderr, "%s: device chain: place of segment %zu is %llu, the sum before it %llu\n", what, j, (unsigned long long)dv[j].base_first, (unsigned long long)nf); return 1; }
                nf += dv[j].n_first; nr += dv[j].n_rec;
            }
            if (co.n_first != nf || co.n_rec != nr || co.n_redo != walks) { fprintf(stderr, "%s: device chain totals\n", what); return 1; }
            if (walks > wrong_early.size() + wrong_late.size() + (size_t)max_rounds * 4 + 8) { fprintf(stderr, "%s: device chain walked %u segments again\n", what, walks); return 1; }
        }
    }
    std::vector<uint32_t> redo;
    int rounds = 0;
    while (pdb2::check_chain(segs, &redo) > 0) {
        if (++rounds > 64) { fprintf(stderr, "%s: no convergence\n", what); return 1; }
        if (getenv("CHAIN_DBG") && !strcmp(getenv("CHAIN_DBG"), what)) { fprintf(stderr, "round %d redo:", rounds); for (uint32_t j : redo) fprintf(stderr, " %u(hint %lld, used %lld)", j, (long long)segs[j].hint, (long long)segs[j].used_start); fprintf(stderr, "\n"); }
        for (uint32_t j : redo) m.walk(segs[j], segs[j].hint);
    }
    if (rounds > max_rounds) { fprintf(stderr, "%s: %d rounds (at most %d expected)\n", what, rounds, max_rounds); return 1; }
    // the confirmed chain is the true one up to the stop
    uint64_t n = 0; bool dead = false;
    for (size_t j = 0; j < segs.size(); ++j) {
        const pdb2::Seg &s = segs[j];
        if (dead) { if (!(s.flags & (pdb2::WF_BAD | pdb2::WF_MORE))) { fprintf(stderr, "%s: segment %zu after the stop carries no flag\n", what, j); return 1; } continue; }
        const uint64_t want = m.first_in(s.begin, s.end);
        const uint64_t lastrec = m.rec[m.rec.size() - 2];
        const uint64_t w2 = (want != pdb2::NONE) ? want : (s.begin <= lastrec && lastrec < s.end ? lastrec : pdb2::NONE);
        if (s.used_start != w2) { fprintf(stderr, "%s: segment %zu starts at %llu, the chain says %llu\n", what, j, (unsigned long long)s.used_start, (unsigned long long)w2); return 1; }
        n += s.n_rec;
        if (s.e_last >= pdb2::STOPPED) dead = true;
    }
    (void)n;
    return 0;
}

// several units in one batch (region fetch: a unit per joined index chunk, many of them a segment or two long): the device form confirms
// every unit's chain by itself — a segmented prefix maximum — and must arrive where check_chain arrives, with the places running through
static int run_units(std::mt19937_64 &rng, int rep)
{
    const int n_units = 2 + (int)(rng() % 90);
    std::vector<Model> ms((size_t)n_units);
    std::vector<pdb2::Seg> segs; std::vector<int> unit_of;
    uint64_t p = 5000;
    for (int u = 0; u < n_units; ++u) {
        const size_t nrec = 1 + rng() % (rep % 2 ? 40 : 900);
        for (size_t i = 0; i < nrec; ++i) { ms[(size_t)u].rec.push_back(p); p += 150 + rng() % 400; }
        ms[(size_t)u].rec.push_back(p);
        const uint64_t first = ms[(size_t)u].rec[0], stop = p;
        for (uint64_t b = first; b < stop; b += 65536) {
            pdb2::Seg sg{}; sg.begin = b; sg.end = std::min<uint64_t>(b + 65536, stop); sg.avail = stop; sg.unit_first = b == first; sg.hint = sg.unit_first ? b : pdb2::NONE;
            segs.push_back(sg); unit_of.push_back(u);
        }
        p += 1000 + rng() % 200000;
    }
    for (size_t j = 0; j < segs.size(); ++j) {
        const Model &m = ms[(size_t)unit_of[j]];
        uint64_t g = segs[j].unit_first ? segs[j].begin : m.first_in(segs[j].begin, segs[j].end);
        if (!segs[j].unit_first && rng() % 5 == 0 && g != pdb2::NONE && g >= segs[j].begin + 2) g -= 2;          // a false start
        else if (!segs[j].unit_first && rng() % 7 == 0) { const uint64_t h = g == pdb2::NONE ? segs[j].begin + 1 : m.first_in(g + 1, segs[j].end); if (h != pdb2::NONE) g = h; }
        m.walk(segs[j], g);
    }
    std::vector<pdb2::Seg> dv = segs, hv = segs;
    for (auto &x : dv) x.n_first = x.n_rec;
    std::vector<int> member_ok(3, 0);
    pdb2::ChainOut co; memset(&co, 0xAB, sizeof co);
    pdb2::chain_device<pdw::HostWave>(dv.data(), (uint32_t)dv.size(), member_ok.data(), (uint32_t)member_ok.size(), ~0ull, ~0ull, 1000000u,
        [&](uint32_t j, uint64_t start) { ms[(size_t)unit_of[j]].walk(dv[j], start); dv[j].hint = start; dv[j].n_first = dv[j].n_rec;
                                          return pdb2::WalkOut{dv[j].used_start, dv[j].e_last, dv[j].n_first, 0u, 0u, dv[j].n_rec, dv[j].flags, 0u}; }, &co);
    std::vector<uint32_t> r2; int rr = 0;
    while (pdb2::check_chain(hv, &r2) > 0 && ++rr <= 64) for (uint32_t j : r2) ms[(size_t)unit_of[j]].walk(hv[j], hv[j].hint);
    bool host_flag = false;
    for (auto &x : hv) if (x.flags) host_flag = true;
    if (host_flag != (co.slow != 0)) { fprintf(stderr, "units rep %d: device chain slow = %u, the host's chain %s a flag\n", rep, co.slow, host_flag ? "carries" : "carries no"); return 1; }
    if (co.slow) return 0;
    uint64_t nf = 0;
    for (size_t j = 0; j < dv.size(); ++j) {
        if (dv[j].used_start != hv[j].used_start || dv[j].e_last != hv[j].e_last || dv[j].n_rec != hv[j].n_rec) { fprintf(stderr, "units rep %d: device chain differs from the host's at segment %zu of %zu (%d units)\n", rep, j, dv.size(), n_units); return 1; }
        if (dv[j].base_first != nf) { fprintf(stderr, "units rep %d: place of segment %zu\n", rep, j); return 1; }
        nf += dv[j].n_first;
    }
    uint64_t fs = ~0ull, E0 = 0;
    for (size_t j = 0; j < hv.size() && unit_of[j] == 0; ++j) { if (fs == ~0ull && hv[j].used_start != pdb2::NONE) fs = hv[j].used_start; if (hv[j].e_last > E0) E0 = hv[j].e_last; }
    if (co.n_first != nf || co.first_start != fs || co.next_start != (E0 ? E0 : ~0ull)) { fprintf(stderr, "units rep %d: totals / unit 0's ends\n", rep); return 1; }
    return 0;
}

int main()
{
    std::mt19937_64 rng(3);
    int bad = 0, cases = 0;
    for (int rep = 0; rep < 300; ++rep) {
        Model m;
        uint64_t p = 1000 + rng() % 5000;
        const bool long_reads = rep % 3 == 2;
        const size_t nrec = 2000 + rng() % 3000;
        for (size_t i = 0; i < nrec; ++i) { m.rec.push_back(p); p += long_reads ? 200 + rng() % 300000 : 200 + rng() % 400; }
        m.rec.push_back(p);
        const uint64_t seg = 65536;
        const size_t nseg = (size_t)((p - m.rec[0] + seg - 1) / seg);
        if (nseg < 12) continue;
        char what[128];
        // every guess right: no repeats
        snprintf(what, sizeof what, "rep %d clean", rep); bad += run_case(m, seg, {}, {}, 0, what); ++cases;
        // isolated wrong guesses (at least two segments apart): one repeat
        { std::set<size_t> e, l; for (size_t j = 2; j + 1 < nseg; j += 3 + rng() % 5) ((rng() & 1) ? e : l).insert(j);
          snprintf(what, sizeof what, "rep %d isolated", rep); bad += run_case(m, seg, e, l, long_reads ? 8 : 1, what); ++cases; }
        // k wrong guesses in a row: at most k + 1 repeats
        { const size_t k = 2 + rng() % 6, j0 = 1 + rng() % (nseg - k - 1); std::set<size_t> e, l; for (size_t j = j0; j < j0 + k; ++j) ((rng() & 1) ? e : l).insert(j);
          snprintf(what, sizeof what, "rep %d run of %zu", rep, k); bad += run_case(m, seg, e, l, long_reads ? (int)k + 8 : (int)k + 1, what); ++cases; }
        // a record that cannot be walked in the middle: the unit ends there, with or without wrong guesses around it
        { Model c = m; c.stop_at = m.rec[nrec / 2]; c.stop_flag = (rep & 1) ? pdb2::WF_BAD : pdb2::WF_HOST;
          std::set<size_t> e; for (size_t j = 3; j + 1 < nseg; j += 4) e.insert(j);
          snprintf(what, sizeof what, "rep %d stop", rep); bad += run_case(c, seg, e, {}, long_reads ? 8 : 1, what); ++cases; }
    }
    for (int rep = 0; rep < 400; ++rep) { bad += run_units(rng, rep); ++cases; }
    printf("chain_check: %d cases, %d failures\n", cases, bad);
    return bad ? 1 : 0;
}
