// host_test.cpp -- the reference's own tests, restated against the C++ host mirror
// (include/lz4.hpp) over liblz4hip.so, with the CPU oracle (oracle/liblz4oracle.so) as
// the checker for compressed bytes.  Built by tests/test_gpu_cpp.py; needs a GPU.
//
//   block_test.go:   TestCompressUncompressBlock, TestCompressCornerCase_CopyDstUpperBound,
//                    TestIssue23, TestIssue116
//   decode_test.go:  TestBlockDecode (a few vectors), TestDecodeBlockInvalid, TestDecodeWithDict
//   writer_test.go:  TestWriter (goldens x options), TestWriterLegacy, TestIssue41/Reset
//   reader_test.go:  TestReader (golden frames), TestReaderLegacy, ValidFrameHeader
//   options:         option validation and applicability
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>

#include "../../include/lz4.hpp"
#include "../../oracle/lz4_oracle.h"

using lz4::Bytes;
using lz4::Error;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        g_checks++;                                                        \
        if (!(cond)) {                                                     \
            g_fail++;                                                      \
            fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
        }                                                                  \
    } while (0)

static std::string g_dir;
static Bytes load(const std::string &name) {
    std::ifstream f(g_dir + "/" + name, std::ios::binary);
    if (!f) {
        fprintf(stderr, "cannot open %s/%s\n", g_dir.c_str(), name.c_str());
        exit(2);
    }
    return Bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static Bytes str(const std::string &s) { return Bytes(s.begin(), s.end()); }

// --- block_test.go:31-109 ---
static void TestCompressUncompressBlock() {
    const char *files[] = {"e.txt", "gettysburg.txt", "Mark.Twain-Tom.Sawyer.txt", "pg1661.txt",
                           "pi.txt", "random.data", "repeat.txt", "issue51.data", "pg_control.tar"};
    for (const char *name : files) {
        const Bytes src = load(name);
        for (int hc = 0; hc < 2; hc++) {
            Bytes z((size_t)lz4::CompressBlockBound((int)src.size()));
            const lz4::Result r = hc ? lz4::CompressBlockHC(src, z, lz4::Level3) : lz4::CompressBlock(src, z);
            CHECK(r.err == Error::None && r.n > 0);
            // byte-identical to the oracle's restatement of the same algorithm
            Bytes want(z.size());
            const int64_t k = hc ? lz4o_compress_hc(src.data(), src.size(), want.data(), want.size(), lz4::Level3)
                                 : lz4o_compress_fast(src.data(), src.size(), want.data(), want.size());
            CHECK(k == r.n && memcmp(z.data(), want.data(), (size_t)k) == 0);
            z.resize((size_t)r.n);
            Bytes back(src.size());
            const lz4::Result u = lz4::UncompressBlock(z, back);
            CHECK(u.err == Error::None && (size_t)u.n == src.size() && back == src);
            // a destination one byte short must fail, not overflow
            if (src.size() > 1) {
                Bytes small(src.size() - 1);
                CHECK(lz4::UncompressBlock(z, small).err == Error::InvalidSourceShortBuffer);
            }
        }
    }
}

// --- block_test.go:117-150: a dst of 40 % must give ErrInvalidSourceShortBuffer ---
static void TestCompressCornerCase_CopyDstUpperBound() {
    const Bytes src = load("upperbound.data");
    for (int hc = 0; hc < 2; hc++) {
        Bytes dst((size_t)((double)src.size() * 0.40));
        const lz4::Result r = hc ? lz4::CompressBlockHC(src, dst, (lz4::CompressionLevel)16)
                                 : lz4::CompressBlock(src, dst);
        CHECK(r.err == Error::InvalidSourceShortBuffer && r.n == 0);
    }
}

// --- block_test.go:143-160: incompressible input with a big enough dst compresses ---
static void TestIssue23() {
    Bytes src = str("aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa");
    Bytes z((size_t)lz4::CompressBlockBound((int)src.size()));
    lz4::Compressor c;
    const lz4::Result r = c.CompressBlock(src, z);
    CHECK(r.err == Error::None && r.n > 0 && (size_t)r.n < src.size());
}

// --- decode_test.go ---
static void TestBlockDecode() {
    struct V { Bytes src, want; };
    const V vs[] = {
        {{0x50, 'h', 'e', 'l', 'l', 'o'}, str("hello")},                               // literal_only_short
        {{0x10, 'a', 1, 0}, str("aaaaa")},                                               // repeat_match_len
        {{0x10, 'a', 1, 0, 0x10, 'B', 1, 0}, str("aaaaaBBBBB")},                         // 2 sequences
    };
    for (int i = 0; i < 3; i++) {
        Bytes dst(vs[i].want.size());
        const lz4::Result r = lz4::UncompressBlock(vs[i].src, dst);
        CHECK(r.err == Error::None && (size_t)r.n == vs[i].want.size() && dst == vs[i].want);
    }
    // empty source: (0, nil) at the API (block.go:45-47)
    Bytes dst(8);
    const lz4::Result e = lz4::UncompressBlock(Bytes{}, dst);
    CHECK(e.err == Error::None && e.n == 0);
}
static void TestDecodeBlockInvalid() {
    const Bytes bad[] = {
        {0x10, 'a', 0, 0},                       // offset 0
        {0x10, 'a', 2, 0},                       // offset before the start
        {0xF0, 0xFF},                            // literal length runs off the input
        {0x40, 'a', 'b'},                        // literals cut short
        {0x1F, 'a', 1, 0},                       // match length byte missing
    };
    for (const Bytes &b : bad) {
        Bytes dst(64);
        CHECK(lz4::UncompressBlock(b, dst).err == Error::InvalidSourceShortBuffer);
    }
}
static void TestDecodeWithDict() {  // decode_test.go:262-284
    const Bytes dict = str("0123456789abcdef");
    // 2 literals "xy", then 6 bytes from 8 back: all six lie in the dictionary's tail
    const Bytes src = {0x22, 'x', 'y', 8, 0, 0x10, 'z'};
    Bytes dst(64);
    const lz4::Result r = lz4::UncompressBlockWithDict(src, dst, dict);
    Bytes want(64);
    const int64_t k = lz4o_decode_block(want.data(), want.size(), src.data(), src.size(), dict.data(), dict.size());
    CHECK(r.err == Error::None && r.n == k && memcmp(dst.data(), want.data(), (size_t)k) == 0);
    CHECK(r.n == 9 && memcmp(dst.data(), "xyabcdefz", 9) == 0);
    Bytes nodict(64);
    CHECK(lz4::UncompressBlock(src, nodict).err == Error::InvalidSourceShortBuffer);
}

// --- writer_test.go:19-92 (TestWriter) + reader round trip ---
static Bytes oracle_frame(const Bytes &src, uint32_t bs, uint32_t flags, uint32_t level, uint64_t csize) {
    Bytes out(lz4o_frame_bound(src.size(), bs, flags));
    const int64_t k = lz4o_frame_compress(src.data(), src.size(), bs, flags, level, csize, out.data(), out.size());
    if (k < 0) {
        fprintf(stderr, "oracle frame_compress failed\n");
        exit(2);
    }
    out.resize((size_t)k);
    return out;
}
static void TestWriter() {
    const char *files[] = {"empty.txt", "e.txt", "gettysburg.txt", "Mark.Twain-Tom.Sawyer.txt",
                           "pg1661.txt", "pi.txt", "random.data", "repeat.txt", "issue102.data"};
    struct Opt { const char *name; lz4::Option o; uint32_t flags_on, flags_off, bs, level; uint64_t csize; };
    const Opt opts[] = {
        {"ConcurrencyOption(1)", lz4::ConcurrencyOption(1), 0, 0, 0, 0, 0},
        {"BlockChecksumOption(true)", lz4::BlockChecksumOption(true), LZ4O_F_BLOCK_CHECKSUM, 0, 0, 0, 0},
        {"SizeOption(123)", lz4::SizeOption(123), LZ4O_F_CONTENT_SIZE, 0, 0, 0, 123},
        {"BlockSizeOption(Block64Kb)", lz4::BlockSizeOption(lz4::Block64Kb), 0, 0, lz4::Block64Kb, 0, 0},
        {"ChecksumOption(false)", lz4::ChecksumOption(false), 0, LZ4O_F_CONTENT_CHECKSUM, 0, 0, 0},
        {"CompressionLevelOption(Level3)", lz4::CompressionLevelOption(lz4::Level3), 0, 0, 0, lz4::Level3, 0},
    };
    for (const char *name : files) {
        const Bytes raw = load(name);
        for (const Opt &o : opts) {
            for (int iocopy = 0; iocopy < 2; iocopy++) {
                std::ostringstream zout;
                lz4::Writer zw(zout);
                CHECK(zw.Apply(o.o) == Error::None);
                if (iocopy) {
                    std::istringstream in(std::string(raw.begin(), raw.end()));
                    CHECK(zw.ReadFrom(in).n == (int64_t)raw.size());
                } else {
                    CHECK(zw.Write(raw).n == (int64_t)raw.size());
                }
                CHECK(zw.Close() == Error::None);
                const std::string z = zout.str();
                uint32_t flags = (LZ4O_F_CONTENT_CHECKSUM | o.flags_on) & ~o.flags_off;
                if (iocopy) flags |= LZ4O_F_READFROM;
                const Bytes want = oracle_frame(raw, o.bs ? o.bs : lz4::Block4Mb, flags, o.level, o.csize);
                CHECK(z.size() == want.size() && memcmp(z.data(), want.data(), z.size()) == 0);
                // and back through the Reader
                std::istringstream zin(z);
                lz4::Reader zr(zin);
                std::ostringstream back;
                const lz4::Result r = zr.WriteTo(back);
                CHECK(r.err == Error::None && (size_t)r.n == raw.size());
                CHECK(back.str() == std::string(raw.begin(), raw.end()));
                if (o.csize) CHECK(zr.Size() == (int64_t)o.csize);
            }
        }
    }
}

// --- reader_test.go:19-138 (TestReader: goldens made by the C lz4 tool) ---
static void TestReader() {
    const char *names[] = {"e.txt", "gettysburg.txt", "Mark.Twain-Tom.Sawyer.txt", "pg1661.txt",
                           "pi.txt", "random.data", "repeat.txt", "pg_control.tar"};
    for (const char *name : names) {
        const Bytes raw = load(name), frame = load(std::string(name) + ".lz4");
        std::istringstream zin(std::string(frame.begin(), frame.end()));
        lz4::Reader zr(zin);
        CHECK(zr.Apply(lz4::ConcurrencyOption(4)) == Error::None);
        Bytes out;
        uint8_t buf[4096];  // small reads, like io.Copy's 32 KiB buffer
        for (;;) {
            const lz4::Result r = zr.Read(buf, sizeof buf);
            CHECK(r.err == Error::None);
            if (r.n == 0) break;
            out.insert(out.end(), buf, buf + r.n);
        }
        CHECK(out == raw);
    }
    // linked-block and 4 MiB-block goldens (reader_test.go:40-46): content checksum inside
    for (const char *name : {"Mark.Twain-Tom.Sawyer_long.txt.lz4", "Mark.Twain-Tom.Sawyer_linked.txt.lz4"}) {
        const Bytes frame = load(name);
        std::istringstream zin(std::string(frame.begin(), frame.end()));
        lz4::Reader zr(zin);
        std::ostringstream back;
        const lz4::Result r = zr.WriteTo(back);
        CHECK(r.err == Error::None && r.n == 4266361);
        const std::string s = back.str();
        CHECK(lz4::ChecksumZero(reinterpret_cast<const uint8_t *>(s.data()), s.size()) == 0x788577F8u);
    }
    {  // malformed.block.lz4 => invalid block checksum
        const Bytes frame = load("malformed.block.lz4");
        std::istringstream zin(std::string(frame.begin(), frame.end()));
        lz4::Reader zr(zin);
        std::ostringstream back;
        CHECK(zr.WriteTo(back).err == Error::InvalidBlockChecksum);
        CHECK(zr.WriteTo(back).err == Error::InvalidBlockChecksum);  // stays in error
    }
}

// --- writer_test.go:256-299 / reader_test.go:192-236 ---
static void TestLegacy() {
    const Bytes raw = load("bzImage_lz4_isolated");
    {
        const Bytes frame = load("bzImage_lz4_isolated.lz4");
        std::istringstream zin(std::string(frame.begin(), frame.end()));
        lz4::Reader zr(zin);
        std::ostringstream back;
        CHECK(zr.WriteTo(back).err == Error::None);
        CHECK(back.str() == std::string(raw.begin(), raw.end()));
    }
    std::ostringstream zout;
    lz4::Writer zw(zout);
    CHECK(zw.Apply(lz4::LegacyOption(true), lz4::CompressionLevelOption(lz4::Fast)) == Error::None);
    std::istringstream in(std::string(raw.begin(), raw.end()));
    CHECK(zw.ReadFrom(in).err == Error::None);
    CHECK(zw.Close() == Error::None);
    const std::string z = zout.str();
    const Bytes want = oracle_frame(raw, 0, LZ4O_F_LEGACY | LZ4O_F_READFROM, 0, 0);
    CHECK(z.size() == want.size() && memcmp(z.data(), want.data(), z.size()) == 0);
    std::istringstream zin(z);
    lz4::Reader zr(zin);
    std::ostringstream back;
    CHECK(zr.WriteTo(back).err == Error::None && back.str() == std::string(raw.begin(), raw.end()));
}

// --- compressing_reader_test.go:13-78 ---
static void TestCompressingReader() {
    const char *files[] = {"e.txt", "gettysburg.txt", "Mark.Twain-Tom.Sawyer.txt", "pg1661.txt",
                           "pi.txt", "random.data", "repeat.txt", "issue102.data"};
    for (const char *name : files) {
        const Bytes raw = load(name);
        for (int size_opt = 0; size_opt < 2; size_opt++) {
            std::istringstream in(std::string(raw.begin(), raw.end()));
            lz4::CompressingReader zcomp(in);
            CHECK(zcomp.Apply(size_opt ? lz4::SizeOption(123) : lz4::BlockChecksumOption(true),
                              lz4::CompressionLevelOption(lz4::Level1)) == Error::None);
            Bytes zout;
            uint8_t buf[8192];
            for (;;) {
                const lz4::Result r = zcomp.Read(buf, sizeof buf);
                CHECK(r.err == Error::None);
                if (r.n == 0) break;
                zout.insert(zout.end(), buf, buf + r.n);
            }
            const Bytes want = oracle_frame(raw, lz4::Block4Mb,
                                            LZ4O_F_CONTENT_CHECKSUM | (size_opt ? LZ4O_F_CONTENT_SIZE : LZ4O_F_BLOCK_CHECKSUM),
                                            lz4::Level1, size_opt ? 123 : 0);
            CHECK(zout == want);
            std::istringstream zin(std::string(zout.begin(), zout.end()));
            lz4::Reader zr(zin);
            std::ostringstream back;
            const lz4::Result r = zr.WriteTo(back);
            CHECK(r.err == Error::None && (size_t)r.n == raw.size());
            CHECK(back.str() == std::string(raw.begin(), raw.end()));
            if (size_opt) CHECK(zr.Size() == 123);
            CHECK(zcomp.Apply(lz4::ChecksumOption(false)) == Error::OptionClosedOrError);
        }
    }
    std::istringstream none;
    lz4::CompressingReader zc(none);
    CHECK(zc.Apply(lz4::ConcurrencyOption(2)) == Error::OptionNotApplicable);
    CHECK(zc.Apply(lz4::LegacyOption(true)) == Error::OptionNotApplicable);
}

// --- options.go:197-220 (OnBlockDoneOption) ---
static void TestOnBlockDone() {
    const Bytes raw = load("pg1661.txt");  // 9 x 64 KiB + 5,109 bytes
    const size_t bs = lz4::Block64Kb;
    std::vector<int> want, sizes;
    for (size_t at = 0; at < raw.size(); at += bs) {
        const size_t len = std::min(bs, raw.size() - at);
        Bytes dst(len);
        const int64_t n = lz4o_compress_fast(raw.data() + at, len, dst.data(), len);
        want.push_back(n > 0 ? (int)n : (int)len);  // lz4stream/block.go:224-252
        sizes.push_back((int)len);
    }
    std::vector<int> seen;
    const auto note = [&seen](int n) { seen.push_back(n); };
    std::ostringstream zout;
    lz4::Writer zw(zout);
    CHECK(zw.Apply(lz4::BlockSizeOption(lz4::Block64Kb), lz4::OnBlockDoneOption(note)) == Error::None);
    zw.Write(raw);
    CHECK(zw.Close() == Error::None);
    CHECK(seen == want);

    seen.clear();
    std::ostringstream zout2;
    lz4::Writer zw2(zout2);
    CHECK(zw2.Apply(lz4::BlockSizeOption(lz4::Block64Kb), lz4::OnBlockDoneOption(note)) == Error::None);
    std::istringstream in(std::string(raw.begin(), raw.end()));
    zw2.ReadFrom(in);
    CHECK(zw2.Close() == Error::None);
    std::vector<int> both;
    for (size_t i = 0; i < want.size(); i++) {
        both.push_back(want[i]);
        both.push_back(sizes[i]);
    }
    CHECK(seen == both);

    seen.clear();
    std::istringstream in2(std::string(raw.begin(), raw.end()));
    lz4::CompressingReader zc(in2);
    CHECK(zc.Apply(lz4::BlockSizeOption(lz4::Block64Kb), lz4::OnBlockDoneOption(note)) == Error::None);
    Bytes all(zout.str().size() + 16);
    CHECK((size_t)zc.Read(all.data(), all.size()).n == zout.str().size());
    CHECK(seen == want);

    seen.clear();
    std::istringstream zin(zout.str());
    lz4::Reader zr(zin);
    CHECK(zr.Apply(lz4::OnBlockDoneOption(note)) == Error::None);
    std::ostringstream back;
    CHECK((size_t)zr.WriteTo(back).n == raw.size());
    CHECK(seen == sizes);

    seen.clear();
    std::istringstream zin2(zout.str());
    lz4::Reader zr2(zin2);
    CHECK(zr2.Apply(lz4::OnBlockDoneOption(note)) == Error::None);
    Bytes buf(100000);  // straddles block ends: every portion is reported
    size_t total = 0;
    for (;;) {
        const lz4::Result r = zr2.Read(buf.data(), buf.size());
        if (r.n == 0) break;
        total += (size_t)r.n;
    }
    long sum = 0;
    for (int n : seen) sum += n;
    CHECK(total == raw.size() && (size_t)sum == raw.size());
}

// --- options.go / writer.go:43-59 / reader.go:49-64,265-275 ---
static void TestOptionsAndHeader() {
    std::ostringstream sink;
    lz4::Writer zw(sink);
    CHECK(zw.Apply(lz4::BlockSizeOption(12345)) == Error::OptionInvalidBlockSize);
    lz4::Writer zw2(sink);
    CHECK(zw2.Apply(lz4::CompressionLevelOption(7)) == Error::OptionInvalidCompressionLevel);
    lz4::Writer zw3(sink);
    CHECK(zw3.Write(str("abc")).n == 3);
    CHECK(zw3.Apply(lz4::ChecksumOption(false)) == Error::OptionClosedOrError);  // after the first Write
    std::istringstream empty;
    lz4::Reader zr(empty);
    CHECK(zr.Apply(lz4::BlockSizeOption(lz4::Block64Kb)) == Error::OptionNotApplicable);

    // frame_test.go:19-22: header bytes; writer_test.go:243-246: 7 bytes
    const Bytes hdr = {0x04, 0x22, 0x4D, 0x18, 0x64, 0x70, 0xB9};
    CHECK(lz4::ValidFrameHeader(hdr) == std::make_pair(true, Error::None));
    Bytes badsum = hdr;
    badsum[6] ^= 1;
    CHECK(lz4::ValidFrameHeader(badsum) == std::make_pair(false, Error::InvalidHeaderChecksum));
    const Bytes notlz4 = {'P', 'K', 3, 4, 0, 0, 0};
    CHECK(lz4::ValidFrameHeader(notlz4) == std::make_pair(false, Error::None));
    CHECK(std::string(lz4::ErrorString(Error::InvalidFrame)) == "lz4: bad magic number");

    // Reset keeps the options (writer.go:186-196)
    std::ostringstream a, b;
    lz4::Writer w(a);
    CHECK(w.Apply(lz4::BlockSizeOption(lz4::Block64Kb), lz4::ChecksumOption(false)) == Error::None);
    w.Write(str("hello hello hello hello hello hello"));
    CHECK(w.Close() == Error::None);
    w.Reset(b);
    w.Write(str("hello hello hello hello hello hello"));
    CHECK(w.Close() == Error::None);
    CHECK(a.str() == b.str() && !a.str().empty());
}

int main(int argc, char **argv) {
    g_dir = argc > 1 ? argv[1] : "tests/golden/testdata";
    if (lz4hip_device_count() < 1) {
        fprintf(stderr, "no HIP device\n");
        return 3;
    }
    TestCompressUncompressBlock();
    TestCompressCornerCase_CopyDstUpperBound();
    TestIssue23();
    TestBlockDecode();
    TestDecodeBlockInvalid();
    TestDecodeWithDict();
    TestWriter();
    TestReader();
    TestLegacy();
    TestCompressingReader();
    TestOnBlockDone();
    TestOptionsAndHeader();
    printf("host_test: %d checks, %d failed\n", g_checks, g_fail);
    return g_fail ? 1 : 0;
}
