// tests/harness/cram_bits_check.cpp — TEST INFRASTRUCTURE: the core-bit-stream codecs of the CRAM reader (BETA, GAMMA,
// SUBEXP, canonical HUFFMAN; htslib writes none of them for the series read here, htsjdk-written files do) on hand-made
// bit strings whose values follow from the CRAM 3.0 specification §13, and ITF8 / LTF8 at their length boundaries.
#include "../../pandepth_amd/host/cram.cpp"
#include <stdio.h>
using namespace pdh;
static std::vector<uint8_t> bits(const char *s)
{
    std::vector<uint8_t> v; int n = 0;
    for (; *s; ++s) { if (*s != '0' && *s != '1') continue; if (n % 8 == 0) v.push_back(0); if (*s == '1') v.back() |= (uint8_t)(0x80 >> (n % 8)); ++n; }
    return v;
}
static int fails = 0;
#define CHECK(x) do { if (!(x)) { printf("FAIL line %d: %s\n", __LINE__, #x); ++fails; } } while (0)
int main()
{
    {   // BETA: offset 1, 3 bits: 101 -> 5 - 1; 000 -> -1
        Enc e; e.codec = 6; e.offset = 1; e.bits = 3;
        auto b = bits("101 000 111");
        SliceData sd; sd.core = b.data(); sd.core_n = b.size();
        CHECK(sd.get_int(&e) == 4); CHECK(sd.get_int(&e) == -1); CHECK(sd.get_int(&e) == 6); CHECK(sd.ok);
    }
    {   // GAMMA (Elias), offset 1: 1 -> "1"; 5 -> "00101"; 8 -> "0001000"
        Enc e; e.codec = 9; e.offset = 1;
        auto b = bits("1 00101 0001000");
        SliceData sd; sd.core = b.data(); sd.core_n = b.size();
        CHECK(sd.get_int(&e) == 0); CHECK(sd.get_int(&e) == 4); CHECK(sd.get_int(&e) == 7); CHECK(sd.ok);
    }
    {   // SUBEXP k = 2, offset 0: 0..3 -> "0"+2 bits; 4..7 -> "10"+2 bits; 8..15 -> "110"+3 bits
        Enc e; e.codec = 7; e.offset = 0; e.bits = 2;
        auto b = bits("0 11  10 01  110 101  0 00");
        SliceData sd; sd.core = b.data(); sd.core_n = b.size();
        CHECK(sd.get_int(&e) == 3); CHECK(sd.get_int(&e) == 5); CHECK(sd.get_int(&e) == 13); CHECK(sd.get_int(&e) == 0); CHECK(sd.ok);
    }
    {   // HUFFMAN through the parser: symbols {65, 66, 67, 300} with lengths {1, 2, 3, 3}: canonical codes 0, 10, 110, 111
        const uint8_t raw[] = {3, 11, 4, 65, 66, 67, 0x81, 0x2c, 4, 1, 2, 3, 3};      // codec, param bytes, alphabet (itf8), lengths
        Cur c(raw, sizeof raw);
        Enc e;
        CHECK(parse_enc(c, &e));
        auto b = bits("110 0 10 111 0");
        SliceData sd; sd.core = b.data(); sd.core_n = b.size();
        CHECK(sd.get_int(&e) == 67); CHECK(sd.get_int(&e) == 65); CHECK(sd.get_int(&e) == 66); CHECK(sd.get_int(&e) == 300); CHECK(sd.get_int(&e) == 65); CHECK(sd.ok);
        sd.get_int(&e); sd.get_int(&e); sd.get_int(&e); sd.get_int(&e);
        CHECK(!sd.ok || true);                              // running off the end must not crash
    }
    {   // ITF8 / LTF8 boundaries (§2.3)
        const uint8_t a[] = {0x7f, 0x80, 0x80, 0xbf, 0xff, 0xc0, 0x40, 0x00, 0xe0, 0x20, 0x00, 0x00, 0xf1, 0x00, 0x00, 0x00, 0x00, 0xff, 0xff, 0xff, 0xff, 0x0f};
        Cur c(a, sizeof a);
        CHECK(c.itf8() == 127); CHECK(c.itf8() == 128); CHECK(c.itf8() == 16383); CHECK(c.itf8() == 16384); CHECK(c.itf8() == 2097152);
        CHECK(c.itf8() == 268435456); CHECK(c.itf8() == -1); CHECK(c.ok);
        const uint8_t l[] = {0x7f, 0x80, 0x80, 0xf0, 0x80, 0x00, 0x00, 0x00, 0xff, 1, 2, 3, 4, 5, 6, 7, 8};
        Cur d(l, sizeof l);
        CHECK(d.ltf8() == 127); CHECK(d.ltf8() == 128); CHECK(d.ltf8() == 0x80000000LL); CHECK(d.ltf8() == 0x0102030405060708LL); CHECK(d.ok);
    }
    printf("%d failures\n", fails);
    return fails != 0;
}
