// bgzf.h — BGZF block access for BAM (SAM spec §4.1): independent gzip members of <= 64 KiB with
// the compressed size in a "BC" extra field, so blocks can be located without inflating and
// inflated independently on any thread.  Written from the SAM/BAM specification; the reference
// gets this from htslib (lib/libhts.a), which this project does not link.
#ifndef PD_BGZF_H_
#define PD_BGZF_H_
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>

namespace pdh {

// Raw-DEFLATE inflater: libdeflate when /usr/lib*/libdeflate.so.0 can be dlopen'ed (2-3x faster),
// zlib otherwise.  One instance per thread.
class Inflater {
public:
    Inflater();
    ~Inflater();
    // inflates exactly out_len bytes; returns false on corrupt data
    bool inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);
    bool inflate_member(const uint8_t *payload, size_t in_len, uint8_t *out, size_t out_len);     // + the CRC-32 behind the payload
    static const char *backend();
private:
    void *ld_ = nullptr;      // libdeflate decompressor
    void *zs_ = nullptr;      // z_stream
};

struct BgzfBlockInfo { uint32_t csize; uint32_t usize; };   // whole-block compressed size, ISIZE

// Parses a BGZF block header at p (needs >= 18 bytes available); returns total block size
// (BSIZE+1) or 0 if p does not start a BGZF block.  *data_off = offset of the deflate payload.
uint32_t bgzf_block_size(const uint8_t *p, size_t avail, uint32_t *data_off);

// Sequential reader over a BGZF (or plain / plain-gzip) file with a small read-ahead buffer.
// Gives a byte-stream view (read) plus the virtual offset of the next unread byte (tell).
class BgzfReader {
public:
    BgzfReader();
    ~BgzfReader();
    bool open(const std::string &path, std::string *err);
    void close();
    bool is_bgzf() const { return is_bgzf_; }
    // read up to n bytes of uncompressed data; returns bytes read (0 at EOF), -1 on error
    long read(void *dst, size_t n);
    bool read_exact(void *dst, size_t n) { return read(dst, n) == (long)n; }
    // virtual file offset (coffset << 16 | uoffset) of the next byte read() will return
    uint64_t tell() const;
    bool seek(uint64_t voffset);
    // direct access to the current block's unread bytes (avoids copies in the record loop)
    const uint8_t *peek(size_t *avail);
    void consume(size_t n);
    bool eof();
    const std::string &error() const { return err_; }
    // Sequential streams only (no seek afterwards): inflate blocks on n helper threads ahead of the
    // consumer (a read-ahead ring of ~1 MiB compressed chunks, served strictly in file order).
    void set_threads(int n);
private:
    struct Pipe;
    bool load_block();
    bool load_block_threaded();
    Pipe *pipe_ = nullptr;
    const uint8_t *udata_ = nullptr;     // current block's bytes (ubuf_ or a pipeline chunk)
    int fd_ = -1;
    bool is_bgzf_ = false;
    void *gz_ = nullptr;                 // gzFile for non-BGZF input
    std::vector<uint8_t> cbuf_;          // compressed read-ahead
    size_t cpos_ = 0, cend_ = 0;
    uint64_t cfile_off_ = 0;             // file offset of cbuf_[0]
    uint64_t block_coff_ = 0;            // file offset of the current block
    uint64_t next_coff_ = 0;             // file offset of the next block
    std::vector<uint8_t> ubuf_;
    size_t upos_ = 0, ulen_ = 0;
    bool at_eof_ = false;
    Inflater inf_;
    std::string err_;
};

} // namespace pdh
#endif
