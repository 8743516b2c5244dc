// The reference's bzip2::tests::test_unit (src/bzip2/mod.rs:41-58) and test_long (:150-172) written
// against the C++ mirror.  Build: g++ -std=c++17 example_test_unit.cpp -L.. -lbz2_mi355x -Wl,-rpath,..
#include "compression.hpp"
#include <cstdio>
#include <cstring>
#include <string>

using namespace compression;

int main()
{
    const std::string src = "a\n";
    BZip2Encoder enc(9);
    auto ret = collect(encode(src, enc, Action::Finish));
    if (!ret.ok) {
        std::printf("error: %s\n", description(ret.error));
        return 2;
    }
    const uint8_t expect[] = {0x42, 0x5A, 0x68, 0x39, 0x31, 0x41, 0x59, 0x26, 0x53, 0x59, 0x63, 0x3E, 0xD6,
                              0xE2, 0x00, 0x00, 0x00, 0xC1, 0x00, 0x00, 0x10, 0x20, 0x00, 0x20, 0x00, 0x21,
                              0x00, 0x82, 0xB1, 0x77, 0x24, 0x53, 0x85, 0x09, 0x06, 0x33, 0xED, 0x6E, 0x20};
    if (ret.value.size() != sizeof(expect) || std::memcmp(ret.value.data(), expect, sizeof(expect)) != 0) {
        std::printf("test_unit: MISMATCH (%zu bytes)\n", ret.value.size());
        return 1;
    }
    BZip2Encoder enc2; // Default == level 9
    const std::string longs(1000, 'a');
    auto r2 = collect(encode(longs, enc2, Action::Finish));
    if (!r2.ok || r2.value.size() != 45) {
        std::printf("test_long: unexpected size %zu\n", r2.value.size());
        return 1;
    }
    // the reference's decoder tests (src/bzip2/mod.rs:60-82 style): decode what was just encoded
    {
        BZip2Decoder dec;
        auto back = collect(decode(ret.value, dec));
        if (!back.ok || std::string(back.value.begin(), back.value.end()) != src) {
            std::printf("decode(test_unit): MISMATCH\n");
            return 1;
        }
        BZip2Decoder dec2;
        auto back2 = collect(decode(r2.value, dec2));
        if (!back2.ok || std::string(back2.value.begin(), back2.value.end()) != longs) {
            std::printf("decode(test_long): MISMATCH\n");
            return 1;
        }
        BZip2Decoder dec3;
        const std::string bad = "BZh0";
        auto r3 = collect(decode(bad, dec3));
        if (r3.ok || r3.error != BZip2Error::DataErrorMagicFirst ||
            to_compression_error(r3.error) != CompressionError::DataError) {
            std::printf("decode(bad magic): wrong verdict\n");
            return 1;
        }
    }
    // the reference's Deflate / zlib / gzip test_unit vectors (src/deflate/encoder.rs:681-701,
    // src/zlib/encoder.rs:161-174, src/gzip/encoder.rs:147-164) through the C++ mirrors
    {
        const std::string a = "a";
        Inflater inf;
        auto d = collect(encode(a, inf, Action::Finish));
        const uint8_t e0[] = {0x4B, 0x04, 0x00};
        ZlibEncoder zl;
        auto z = collect(encode(a, zl, Action::Finish));
        const uint8_t e1[] = {0x78, 0xDA, 0x4B, 0x04, 0x00, 0x00, 0x62, 0x00, 0x62};
        GZipEncoder gz;
        auto g = collect(encode(a, gz, Action::Finish));
        const uint8_t e2[] = {0x1f, 0x8b, 0x08, 0, 0, 0, 0, 0, 0, 0xFF, 0x4b, 0x04, 0x00, 0x43, 0xbe, 0xb7, 0xe8, 0x01, 0, 0, 0};
        if (!d.ok || d.value.size() != sizeof(e0) || std::memcmp(d.value.data(), e0, sizeof(e0)) != 0 || !z.ok ||
            z.value.size() != sizeof(e1) || std::memcmp(z.value.data(), e1, sizeof(e1)) != 0 || !g.ok ||
            g.value.size() != sizeof(e2) || std::memcmp(g.value.data(), e2, sizeof(e2)) != 0) {
            std::printf("deflate/zlib/gzip test_unit: MISMATCH\n");
            return 1;
        }
    }
    bool threw = false;
    try {
        BZip2Encoder bad(0);
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    std::printf("test_unit ok, test_long ok (%zu bytes), invalid level %s\n", r2.value.size(),
                threw ? "rejected" : "ACCEPTED");
    return threw ? 0 : 1;
}
