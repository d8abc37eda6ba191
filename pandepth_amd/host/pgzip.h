// pgzip.h — the byte stream of `gzopen(path, "wb")` + gzwrite + gzclose (zlib 1.2.x: deflate level 6,
// default strategy, memLevel 8, gzip wrapper with mtime 0 / OS 3), produced with several threads.
//
// The reference writes every table through one zlib gz stream (include/gzstream.c:46-67), so its
// .gz bytes are a function of the text; a 3 Gb `-w 100` table is 1.2 GB of text and a minute of
// single-threaded deflate.  zlib's output factors into two stages:
//   1. the LZ77 parse (deflate_slow: hash chains, lazy matching) — all of the time, and a function
//      of the preceding 32 KiB only;
//   2. block splitting (every 16383 symbols) + Huffman trees + bit packing — cheap, but sequential.
// Stage 1 runs on worker threads, each calling zlib ITSELF on one chunk of the text primed with the
// previous 32 KiB as dictionary (raw deflate), and reading the symbols back out of zlib's output;
// neighbouring chunks overlap by a tail and are stitched where both parses emit a match ending at
// the same position (from there on zlib's state is a function of the window alone).  Stage 2 is
// re-stated here (RFC 1951 + the choices zlib's trees.c makes: heap order and tie-breaks of the
// Huffman construction, length-limit repair, run-length coding of the code lengths, fixed vs
// dynamic choice), and is checked against zlib byte for byte by tests/test_pgzip.py.
// Whenever a case outside the re-statement shows up (a block zlib would store, chunks that do not
// re-synchronise) the function returns false and the caller uses zlib's own serial stream.
#ifndef PD_PGZIP_H_
#define PD_PGZIP_H_
#include <stddef.h>
#include <stdint.h>
#include <functional>
#include <memory>
#include <utility>
#include <vector>

namespace pgz {

// Stage 1 done elsewhere (the engine: pd_deflate_parse on the GPU): the parse of the chunks (start, end, origin triples: positions
// in text[0, n), origin = where the chunk's <= 32 KiB of history begin) — symbols of chunk k at syms[off[k], off[k + 1]).  It must be
// zlib's parse of each chunk primed with its history, except within 1 KiB of a chunk's end; false = not available, zlib does it.
template <class T> struct NoInit : std::allocator<T> {        // resize() leaves the new elements uninitialised (symbol buffers of hundreds of MB)
    template <class U> struct rebind { typedef NoInit<U> other; };
    template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
typedef std::vector<uint32_t, NoInit<uint32_t>> SymVec;
typedef std::function<bool(const uint8_t *text, size_t n, const uint64_t *chunks, size_t n_chunks, SymVec &syms, std::vector<uint64_t> &off)> ParseFn;

struct Params {
    size_t chunk = (size_t)1 << 20;      // text per zlib call
    size_t tail = (size_t)1 << 16;       // overlap in which neighbouring parses must meet
    size_t batch = 0;                    // text buffered between rounds (0 = 96 MiB)
    unsigned calls = 2;                  // provider calls of a round in flight at once (1, 2 or 4; each on its share of the round's chunks)
    ParseFn parse;                       // optional stage-1 provider; a stream's last chunk always goes to zlib (it sees the end of the input)
    // optional, with a provider that copies the text elsewhere: page-lock / release a text buffer of the stream (two buffers of
    // batch + 3 chunks each, never reallocated while locked; pin returns false if it could not)
    std::function<bool(void *, size_t)> pin;
    std::function<void(void *)> unpin;
    // the geometry that suits a provider with one wave per chunk: many small chunks (16 KiB + 4 KiB of overlap), rounds of 96 MiB
    static Params for_device(ParseFn fn);
};

// A text source that is not host memory (the engine's device-resident stream, pd_text_*): the stream is only told how many bytes
// exist (announce); `parse` is a ParseFn on the stretch [off, off + n) of the source that also returns the CRC-32 of the first
// crc_span bytes of every chunk (its own bytes, without the overlap); `fetch` copies a stretch to the host (the stream's tail, which zlib parses itself); `release`: nothing before `off`
// will be asked for again.
struct Remote {
    std::function<bool(uint64_t off, size_t n, const uint64_t *chunks, size_t n_chunks, SymVec &syms, std::vector<uint64_t> &sym_off, uint32_t *crc, uint64_t crc_span)> parse;
    std::function<bool(uint64_t off, size_t n, uint8_t *dst)> fetch;
    std::function<void(uint64_t off)> release;
};

// the host emulation of the engine's parse (csrc/pd_lz77.h, 64 lanes in a loop) as a provider: tests of the plumbing without a GPU
ParseFn host_emulation_parse();

// Streaming form: text in, the .gz file's bytes out through `sink`, in rounds of `batch` bytes (bounded
// memory: a 3 Gb per-site file is 60 GB of text).  write()/finish() return false when the parallel form
// stops applying (or the sink fails); the bytes already handed to the sink are then a prefix of zlib's
// stream and the caller must start the file over with zlib itself.
class Stream {
public:
    Stream(int threads, std::function<bool(const uint8_t *, size_t)> sink, const Params &p = Params());
    Stream(int threads, std::function<bool(const uint8_t *, size_t)> sink, const Params &p, const Remote &source);   // text at `source`: announce(), not write()
    ~Stream();
    bool write(const void *data, size_t n);
    bool announce(uint64_t n);           // (a Remote source) n more bytes of text exist there
    bool wait_idle();                    // the round in flight, if any, is over when this returns (its failure: false)
    bool finish();
private:
    struct Impl;
    Impl *p_;
    Stream(const Stream &) = delete;
    Stream &operator=(const Stream &) = delete;
};

// Appends the complete .gz file image of `data` to `out`.  Returns false (out untouched) when the
// parallel form does not apply; the result, when produced, equals zlib's.
bool gzip_identical(const uint8_t *data, size_t n, int threads, std::vector<uint8_t> &out);

bool gzip_identical(const uint8_t *data, size_t n, int threads, std::vector<uint8_t> &out, const Params &p);

// zlib's own LZ77 parse (literal = byte, match = len << 16 | dist) of data[dict, dict + n) with data[0, dict) as its dictionary
// (dict <= 32768 is what zlib uses): stage 1 of one chunk, exposed for the tests of the engine's parse.
bool zlib_chunk_symbols(const uint8_t *data, size_t dict, size_t n, std::vector<uint32_t> &syms);

} // namespace pgz
#endif
