// tests/harness/nx16_check.cpp — TEST INFRASTRUCTURE (dev container: links the reference's libhts.a, whose bundled
// htscodecs exports the rANS Nx16 and adaptive-arithmetic ENCODERS).  Every transform combination (order 0 / 1, 4- and 32-way interleave, stripe,
// bit packing, run lengths, stored) on random, few-symbol, long-run and periodic inputs of 0 .. 300 001 bytes is encoded by
// htscodecs and must come back byte for byte from the product's decoders (pandepth_amd/host/cram.cpp: nx16_decode, arith_decode).
#include "../../pandepth_amd/host/cram.cpp"
extern "C" unsigned char *rans_compress_to_4x16(unsigned char *in, unsigned int in_size, unsigned char *out, unsigned int *out_size, int order);
extern "C" unsigned char *arith_compress_to(unsigned char *in, unsigned int in_size, unsigned char *out, unsigned int *out_size, int order);
#include <random>
int main() {
    std::mt19937 rng(7);
    int bad = 0, n = 0;
    const int orders[] = {0, 1, 4, 5, 0x40, 0x41, 0x44, 0x45, 0x80, 0x81, 0x84, 0x85, 0xc0, 0xc1, 0xc4, 0xc5, 0x08, 0x09, 0x20};
    for (int o : orders)
        for (int kind = 0; kind < 4; ++kind)
            for (size_t len : {0ul, 1ul, 3ul, 31ul, 32ul, 33ul, 100ul, 1000ul, 4097ul, 70000ul, 300001ul}) {
                std::vector<unsigned char> in(len);
                for (size_t i = 0; i < len; ++i) {
                    if (kind == 0) in[i] = rng() & 0xff;
                    else if (kind == 1) in[i] = "DXSIN"[rng() % 5];
                    else if (kind == 2) in[i] = (rng() % 50) ? 'M' : 'X';               // long runs
                    else in[i] = (i / 97) & 3;
                }
                unsigned int osz = 0;
                unsigned char *c = rans_compress_to_4x16(in.data(), (unsigned)len, nullptr, &osz, o);
                if (!c) continue;
                std::vector<uint8_t> out;
                const bool ok = pdh::nx16_decode(c, osz, &out, 0, false, 0) && out.size() == len && (len == 0 || memcmp(out.data(), in.data(), len) == 0);
                ++n;
                if (!ok) { ++bad; printf("FAIL order 0x%02x kind %d len %zu (stream flags 0x%02x, %u bytes)\n", o, kind, len, c[0], osz); }
                free(c);
            }
    // the adaptive arithmetic coder (block method 6): order 0 / 1, with modelled run lengths, packed, striped, stored
    const int aorders[] = {0, 1, 0x40, 0x41, 0x80, 0x81, 0xc0, 0xc1, 0x08, 0x09, 0x20};
    for (int o : aorders)
        for (int kind = 0; kind < 4; ++kind)
            for (size_t len : {0ul, 1ul, 3ul, 33ul, 100ul, 1000ul, 4097ul, 70000ul, 300001ul}) {
                std::vector<unsigned char> in(len);
                for (size_t i = 0; i < len; ++i) {
                    if (kind == 0) in[i] = rng() & 0xff;
                    else if (kind == 1) in[i] = "DXSIN"[rng() % 5];
                    else if (kind == 2) in[i] = (rng() % 50) ? 'M' : 'X';
                    else in[i] = (i / 97) & 3;
                }
                unsigned int osz = 0;
                unsigned char *c = arith_compress_to(in.data(), (unsigned)len, nullptr, &osz, o);
                if (!c) continue;
                std::vector<uint8_t> out;
                const bool ok = pdh::arith_decode(c, osz, &out, 0, false, 0) && out.size() == len && (len == 0 || memcmp(out.data(), in.data(), len) == 0);
                ++n;
                if (!ok) { ++bad; printf("FAIL arith order 0x%02x kind %d len %zu (stream flags 0x%02x, %u bytes)\n", o, kind, len, c[0], osz); }
                free(c);
            }
    // crafted order-0 streams whose uint7 frequencies wrap a 32-bit sum back to 4096 (F[1] = 0x80000000, F[2] = 0x80001000):
    // the decoder used to memset 2 GiB past its 4096-byte table; every single frequency and the running total are bounded now
    {
        auto u7 = [](std::vector<uint8_t> &v, uint32_t x) { uint8_t t[5]; int k = 0; do { t[k++] = x & 0x7f; x >>= 7; } while (x); while (k--) v.push_back(t[k] | (k ? 0x80 : 0)); };
        for (int variant = 0; variant < 3; ++variant) {
            std::vector<uint8_t> st;
            st.push_back(0x00);                                    // flags: order 0, 4-way, no transforms
            u7(st, 64);                                            // uncompressed size
            st.push_back(1); st.push_back(2); st.push_back(0);     // alphabet {1, 2}
            if (variant == 0) { u7(st, 0x80000000u); u7(st, 0x80001000u); }
            else if (variant == 1) { u7(st, 0xFFFFFFFFu); u7(st, 4097u); }
            else { u7(st, 5000u); u7(st, 0xFFFFF000u - 904u); }
            for (int k = 0; k < 64; ++k) st.push_back((uint8_t)(rng() & 0xff));
            std::vector<uint8_t> out;
            ++n;
            if (pdh::nx16_decode(st.data(), st.size(), &out, 0, false, 0)) { ++bad; printf("FAIL wrapped frequency sum accepted (variant %d)\n", variant); }
        }
    }
    printf("%d cases, %d failures\n", n, bad);
    return bad != 0;
}
