// report.h — gz text output with the reference's byte stream: one zlib gz member, default level,
// written through gzopen/gzwrite like its gzstream wrapper (include/gzstream.c:46-67,133-142), so
// the .gz bytes depend only on the text.  Number formatting = iostream fixed/setprecision(2),
// i.e. printf("%.2f").
#ifndef PD_REPORT_H_
#define PD_REPORT_H_
#include <stdint.h>
#include <string>

namespace pdh {

class GzWriter {
public:
    ~GzWriter() { close(); }
    bool open(const std::string &path);
    void write(const char *p, size_t n);
    void write(const std::string &s) { write(s.data(), s.size()); }
    bool close();
    bool good() const { return f_ != nullptr; }
private:
    void *f_ = nullptr;
};

std::string fmt2(double v);          // "%.2f"

} // namespace pdh
#endif
