// report.cpp — see report.h
#include "report.h"
#include <stdio.h>
#include <zlib.h>

namespace pdh {

bool GzWriter::open(const std::string &path)
{
    close();
    f_ = gzopen(path.c_str(), "wb");
    if (f_) gzbuffer((gzFile)f_, 1 << 18);
    return f_ != nullptr;
}

void GzWriter::write(const char *p, size_t n)
{
    while (n) {
        const unsigned k = n > (1u << 30) ? (1u << 30) : (unsigned)n;
        gzwrite((gzFile)f_, p, k);
        p += k; n -= k;
    }
}

bool GzWriter::close()
{
    if (!f_) return true;
    const int r = gzclose((gzFile)f_);
    f_ = nullptr;
    return r == Z_OK;
}

std::string fmt2(double v)
{
    char b[64];
    snprintf(b, sizeof b, "%.2f", v);
    return b;
}

} // namespace pdh
