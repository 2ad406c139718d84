// lz4.hpp -- C++ host-side mirror of pierrec/lz4's public API over the C ABI of
// include/lz4hip.h.  Header-only; link with -llz4hip.
//
// The reference is Go and no Go toolchain exists in the build image, so the host side
// above the boundary is written in C++ with the reference's names, argument meaning
// and error behaviour:
//
//   lz4.go:27-131        CompressBlockBound, CompressBlock, CompressBlockHC,
//                        UncompressBlock, UncompressBlockWithDict, Compressor, CompressorHC
//   options.go:29-242    BlockSizeOption, BlockChecksumOption, ChecksumOption, SizeOption,
//                        ConcurrencyOption, CompressionLevelOption, LegacyOption
//   writer.go:20-242     Writer: Apply, Write, ReadFrom, Flush, Close, Reset
//   reader.go:19-275     Reader: Apply, Read, WriteTo, Size, Reset; ValidFrameHeader
//   internal/lz4errors   the error values and their texts
//
// Go's (n int, err error) pairs become lz4::Result {n, err}; io.Reader / io.Writer
// become std::istream / std::ostream.  The Go Writer compresses block by block (or
// ConcurrencyOption(n) blocks at a time); here the stream is buffered and handed to
// the GPU as ONE batch of independent blocks at Close() -- the bytes written are the
// ones the Go Writer writes.  All compute runs in liblz4hip.so; there is no CPU path.
#ifndef LZ4_HPP
#define LZ4_HPP

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <istream>
#include <iterator>
#include <ostream>
#include <string>
#include <vector>

#include "lz4hip.h"

namespace lz4 {

// internal/lz4errors/errors.go:7-19 (same texts), plus Hip for a failed GPU call.
enum class Error {
    None = 0,
    InvalidSourceShortBuffer,
    InvalidFrame,
    InternalUnhandledState,
    InvalidHeaderChecksum,
    InvalidBlockChecksum,
    InvalidFrameChecksum,
    OptionInvalidCompressionLevel,
    OptionClosedOrError,
    OptionInvalidBlockSize,
    OptionNotApplicable,
    WriterNotClosed,
    Hip,
};

inline const char *ErrorString(Error e) {
    switch (e) {
    case Error::None: return "";
    case Error::InvalidSourceShortBuffer: return "lz4: invalid source or destination buffer too short";
    case Error::InvalidFrame: return "lz4: bad magic number";
    case Error::InternalUnhandledState: return "lz4: unhandled state";
    case Error::InvalidHeaderChecksum: return "lz4: invalid header checksum";
    case Error::InvalidBlockChecksum: return "lz4: invalid block checksum";
    case Error::InvalidFrameChecksum: return "lz4: invalid frame checksum";
    case Error::OptionInvalidCompressionLevel: return "lz4: invalid compression level";
    case Error::OptionClosedOrError: return "lz4: cannot apply options on closed or in error object";
    case Error::OptionInvalidBlockSize: return "lz4: invalid block size";
    case Error::OptionNotApplicable: return "lz4: option not applicable";
    case Error::WriterNotClosed: return "lz4: writer not closed";
    case Error::Hip: return "lz4: GPU call failed";
    }
    return "";
}

struct Result {
    int64_t n;
    Error err;
    explicit operator bool() const { return err == Error::None; }
};

// options.go:38-43,151-165
enum BlockSize : uint32_t { Block64Kb = 1u << 16, Block256Kb = 1u << 18, Block1Mb = 1u << 20, Block4Mb = 1u << 22 };
enum CompressionLevel : uint32_t {
    Fast = 0,
    Level1 = 1u << 9, Level2 = 1u << 10, Level3 = 1u << 11, Level4 = 1u << 12, Level5 = 1u << 13,
    Level6 = 1u << 14, Level7 = 1u << 15, Level8 = 1u << 16, Level9 = 1u << 17,
};

namespace detail {
inline Result block_result(int64_t r) {
    if (r >= 0) return {r, Error::None};
    if (r == LZ4HIP_ERR_SHORT || r == -1) return {0, Error::InvalidSourceShortBuffer};
    return {0, Error::Hip};
}
inline Error frame_error(int64_t r) {
    switch (r) {
    case LZ4HIP_ERR_FRAME: return Error::InvalidFrame;
    case LZ4HIP_ERR_HEADER_CHECKSUM: return Error::InvalidHeaderChecksum;
    case LZ4HIP_ERR_BLOCK_SIZE: return Error::OptionInvalidBlockSize;
    case LZ4HIP_ERR_BLOCK_CHECKSUM: return Error::InvalidBlockChecksum;
    case LZ4HIP_ERR_FRAME_CHECKSUM: return Error::InvalidFrameChecksum;
    case LZ4HIP_ERR_BLOCK: return Error::InvalidSourceShortBuffer;
    }
    return Error::Hip;
}
}  // namespace detail

// ---- block API: lz4.go:27-131 ----
inline int CompressBlockBound(int n) { return (int)lz4hip_compress_bound((size_t)n); }

// (n, nil) | (0, nil) incompressible | (0, ErrInvalidSourceShortBuffer)
inline Result CompressBlock(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len) {
    const int64_t r = lz4hip_compress_block(src, n, dst, dst_len);
    return r == -1 ? Result{0, Error::InvalidSourceShortBuffer} : detail::block_result(r);
}
inline Result CompressBlockHC(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                              CompressionLevel depth) {
    const int64_t r = lz4hip_compress_block_hc(src, n, dst, dst_len, (uint32_t)depth);
    return r == -1 ? Result{0, Error::InvalidSourceShortBuffer} : detail::block_result(r);
}
inline Result UncompressBlockWithDict(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len,
                                      const uint8_t *dict, size_t dict_len) {
    const int64_t r = lz4hip_uncompress_block(src, n, dst, dst_len, dict, dict_len);
    return r == -1 ? Result{0, Error::InvalidSourceShortBuffer} : detail::block_result(r);
}
inline Result UncompressBlock(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len) {
    return UncompressBlockWithDict(src, n, dst, dst_len, nullptr, 0);
}
using Bytes = std::vector<uint8_t>;
inline Result CompressBlock(const Bytes &src, Bytes &dst) {
    return CompressBlock(src.data(), src.size(), dst.data(), dst.size());
}
inline Result CompressBlockHC(const Bytes &src, Bytes &dst, CompressionLevel depth) {
    return CompressBlockHC(src.data(), src.size(), dst.data(), dst.size(), depth);
}
inline Result UncompressBlock(const Bytes &src, Bytes &dst) {
    return UncompressBlock(src.data(), src.size(), dst.data(), dst.size());
}
inline Result UncompressBlockWithDict(const Bytes &src, Bytes &dst, const Bytes &dict) {
    return UncompressBlockWithDict(src.data(), src.size(), dst.data(), dst.size(), dict.data(),
                                   dict.size());
}
// internal/xxh32/xxh32zero.go:142
inline uint32_t ChecksumZero(const uint8_t *p, size_t n) { return lz4hip_xxh32(p, n); }

// lz4.go:57-72 / :103-123.  Stateless here: the tables live on the device per call.
struct Compressor {
    Result CompressBlock(const Bytes &src, Bytes &dst) { return lz4::CompressBlock(src, dst); }
};
struct CompressorHC {
    CompressionLevel Level = Fast;
    Result CompressBlock(const Bytes &src, Bytes &dst) { return lz4::CompressBlockHC(src, dst, Level); }
};

// ---- options: options.go ----
class Writer;
class Reader;
class CompressingReader;
struct Applier {  // the `applier` interface: exactly one of the three is set
    Writer *w = nullptr;
    Reader *r = nullptr;
    CompressingReader *c = nullptr;
};
using Option = std::function<Error(Applier)>;
using BlockDoneHandler = std::function<void(int)>;  // options.go:195-220

namespace detail {
// One frame for `src` plus, when a handler is set, len(block.Data) of every block.
inline int64_t frame_compress(const Bytes &src, uint32_t block_size, uint32_t flags, uint32_t level,
                              uint64_t content_size, Bytes &out, std::vector<uint32_t> *lens) {
    out.resize(lz4hip_frame_bound(src.size(), block_size, flags));
    if (!lens)
        return lz4hip_frame_compress(src.data(), src.size(), block_size, flags, level, content_size,
                                     out.data(), out.size());
    const uint32_t bs = (flags & LZ4HIP_F_LEGACY) ? (8u << 20) : (block_size ? block_size : 1);
    lens->resize(src.size() / bs + 2);
    size_t nb = 0;
    const int64_t k = lz4hip_frame_compress_lens(src.data(), src.size(), block_size, flags, level,
                                                 content_size, out.data(), out.size(), lens->data(),
                                                 lens->size(), &nb);
    lens->resize(k < 0 ? 0 : nb);
    return k;
}
}  // namespace detail

// ---- Writer: writer.go ----
class Writer {
  public:
    explicit Writer(std::ostream &dst) : dst_(&dst) {}

    // writer.go:43-59: only before the first Write
    template <class... O> Error Apply(O... options) {
        if (state_ == State::Error) return err_;
        if (state_ != State::New) return Error::OptionClosedOrError;
        Error e = Error::None;
        for (const Option &o : {Option(options)...})
            if ((e = o(Applier{this, nullptr})) != Error::None) return fail(e);
        return e;
    }
    // writer.go:74-121
    Result Write(const uint8_t *p, size_t n) {
        if (state_ == State::Closed || state_ == State::Error) return {0, err_};
        state_ = State::Write;
        buf_.insert(buf_.end(), p, p + n);
        return {(int64_t)n, Error::None};
    }
    Result Write(const Bytes &b) { return Write(b.data(), b.size()); }
    // io.Copy(zw, r): writer.go:199-242 (a stream that is a block multiple ends in an
    // empty block)
    Result ReadFrom(std::istream &r) {
        if (state_ == State::Closed || state_ == State::Error) return {0, err_};
        state_ = State::Write;
        const size_t before = buf_.size();
        buf_.insert(buf_.end(), std::istreambuf_iterator<char>(r), std::istreambuf_iterator<char>());
        readfrom_ = true;
        return {(int64_t)(buf_.size() - before), Error::None};
    }
    // writer.go:148-169: nothing reaches dst before Close() here (one batch per frame)
    Error Flush() { return state_ == State::Error ? err_ : Error::None; }
    // writer.go:173-184
    Error Close() {
        if (state_ == State::Closed) return Error::None;
        if (state_ == State::Error) return err_;
        uint32_t flags = (block_checksum ? LZ4HIP_F_BLOCK_CHECKSUM : 0u) |
                         (content_checksum ? LZ4HIP_F_CONTENT_CHECKSUM : 0u) |
                         (content_size > 0 ? LZ4HIP_F_CONTENT_SIZE : 0u) |
                         (readfrom_ ? LZ4HIP_F_READFROM : 0u) | (legacy ? LZ4HIP_F_LEGACY : 0u);
        Bytes out;
        std::vector<uint32_t> lens;
        const int64_t k = detail::frame_compress(buf_, block_size, flags, (uint32_t)level, content_size,
                                                 out, handler ? &lens : nullptr);
        if (k < 0) return fail(detail::frame_error(k));
        dst_->write(reinterpret_cast<const char *>(out.data()), k);
        const size_t bs = legacy ? (8u << 20) : block_size;
        for (size_t i = 0; i < lens.size(); i++) {
            handler((int)lens[i]);  // writer.go:127
            if (readfrom_)          // writer.go:234: the bytes read for this block
                handler((int)(i * bs >= buf_.size() ? 0 : std::min(bs, buf_.size() - i * bs)));
        }
        state_ = State::Closed;
        return Error::None;
    }
    // writer.go:192-196: keeps the options
    void Reset(std::ostream &dst) {
        dst_ = &dst;
        buf_.clear();
        readfrom_ = false;
        state_ = State::New;
        err_ = Error::None;
    }

    // set through the options
    uint32_t block_size = Block4Mb;  // DefaultBlockSizeOption
    bool block_checksum = false;
    bool content_checksum = true;  // DefaultChecksumOption
    uint64_t content_size = 0;
    CompressionLevel level = Fast;
    int concurrency = 1;
    bool legacy = false;
    BlockDoneHandler handler;  // OnBlockDoneOption; called in block order once the batch is done

  private:
    enum class State { New, Write, Closed, Error };
    Error fail(Error e) {
        state_ = State::Error;
        err_ = e;
        return e;
    }
    std::ostream *dst_;
    Bytes buf_;
    bool readfrom_ = false;
    State state_ = State::New;
    Error err_ = Error::None;
};

// ---- Reader: reader.go ----
class Reader {
  public:
    explicit Reader(std::istream &src) : src_(&src) {}

    template <class... O> Error Apply(O... options) {  // reader.go:49-64
        if (state_ == State::Error) return err_;
        if (state_ != State::New) return Error::OptionClosedOrError;
        Error e = Error::None;
        for (const Option &o : {Option(options)...})
            if ((e = o(Applier{nullptr, this})) != Error::None) return fail(e);
        return e;
    }
    // reader.go:103-157.  n == 0 with Error::None after the last byte is io.EOF.
    Result Read(uint8_t *buf, size_t len) {
        const Error e = fill();
        if (e != Error::None) return {0, e};
        const size_t k = len < out_.size() - pos_ ? len : out_.size() - pos_;
        memcpy(buf, out_.data() + pos_, k);
        report(pos_, pos_ + k);
        pos_ += k;
        return {(int64_t)k, Error::None};
    }
    // reader.go:213-262
    Result WriteTo(std::ostream &w) {
        const Error e = fill();
        if (e != Error::None) return {0, e};
        const size_t k = out_.size() - pos_;
        report(pos_, out_.size());
        w.write(reinterpret_cast<const char *>(out_.data() + pos_), (std::streamsize)k);
        pos_ = out_.size();
        return {(int64_t)k, Error::None};
    }
    // reader.go:66-76: the content size of the header, once reading has started
    int64_t Size() const { return state_ == State::Read ? (int64_t)size_ : 0; }
    void Reset(std::istream &src) {  // reader.go:200-211
        src_ = &src;
        out_.clear();
        ends_.clear();
        pos_ = 0;
        size_ = 0;
        state_ = State::New;
        err_ = Error::None;
    }
    int concurrency = 1;
    BlockDoneHandler handler;  // OnBlockDoneOption: decoded bytes per block portion handed out

  private:
    enum class State { New, Read, Error };
    // handler(bn) for every block portion inside [lo, hi) of the stream: reader.go:154,255
    void report(size_t lo, size_t hi) {
        if (!handler) return;
        size_t i = std::upper_bound(ends_.begin(), ends_.end(), lo) - ends_.begin();
        for (; lo < hi && i < ends_.size(); i++) {
            const size_t cut = std::min(hi, ends_[i]);
            handler((int)(cut - lo));
            lo = cut;
        }
    }
    Error fail(Error e) {
        state_ = State::Error;
        err_ = e;
        return e;
    }
    // Upper bound of the decoded size from the block index (lz4stream/block.go:279-317).
    static size_t decoded_bound(const Bytes &f, size_t si, uint32_t bs, uint32_t flags) {
        size_t nb = 0;  // bytes
        const bool legacy = flags & LZ4HIP_F_LEGACY, bsum = flags & LZ4HIP_F_BLOCK_CHECKSUM;
        while (si + 4 <= f.size()) {
            uint32_t w;
            memcpy(&w, f.data() + si, 4);
            si += 4;
            if (legacy && w == 0x184C2102u) continue;
            if (!legacy && w == 0) break;
            const size_t size = w & 0x7FFFFFFFu;
            si += size + (bsum ? 4 : 0);
            // a stored block keeps its size, a compressed one grows at most 255-fold
            const size_t grow = (w >> 31) ? size : size * 255;
            nb += grow < bs ? grow : bs;
        }
        return nb;
    }
    Error fill() {
        if (state_ == State::Error) return err_;
        if (state_ == State::Read) return Error::None;
        Bytes frame((std::istreambuf_iterator<char>(*src_)), std::istreambuf_iterator<char>());
        uint64_t csize = 0;
        uint32_t bs = 0, flags = 0;
        size_t hlen = 0;
        const int rc = lz4hip_frame_parse_header(frame.data(), frame.size(), &bs, &flags, &csize, &hlen);
        if (rc == LZ4HIP_ERR_SHORT) return fail(Error::InvalidFrame);
        if (rc != LZ4HIP_OK) return fail(detail::frame_error(rc));
        out_.resize(decoded_bound(frame, hlen, bs, flags) + 16);
        std::vector<uint32_t> lens(handler ? frame.size() / 4 + 1 : 0);  // >= 4 bytes per block
        size_t nb = 0;
        const int64_t k =
            handler ? lz4hip_frame_uncompress_lens(frame.data(), frame.size(), out_.data(), out_.size(),
                                                   lens.data(), lens.size(), &nb)
                    : lz4hip_frame_uncompress(frame.data(), frame.size(), out_.data(), out_.size());
        if (k < 0) return fail(detail::frame_error(k));
        out_.resize((size_t)k);
        ends_.clear();
        for (size_t b = 0, at = 0; b < nb; b++) ends_.push_back(at += lens[b]);
        size_ = csize;
        state_ = State::Read;
        return Error::None;
    }
    std::istream *src_;
    Bytes out_;
    std::vector<size_t> ends_;  // end offset of every block in the decoded stream
    size_t pos_ = 0;
    uint64_t size_ = 0;
    State state_ = State::New;
    Error err_ = Error::None;
};

// ---- CompressingReader: compressing_reader.go ----
// Read() yields the LZ4 frame of everything the source holds (the opposite of Reader).
// The source is drained and compressed as one batch on the first Read; the bytes are those
// of init + Compress().Write per block + CloseW (compressing_reader.go:76-81,116-157).
class CompressingReader {
  public:
    explicit CompressingReader(std::istream &src) : src_(&src) {}

    std::istream &Source() { return *src_; }  // compressing_reader.go:46-49
    Error Close() { return Error::None; }     // :51-56: a std::istream has nothing to close
    // :58-72: only before the first Read
    template <class... O> Error Apply(O... options) {
        if (state_ != State::Initial) return Error::OptionClosedOrError;
        Error e = Error::None;
        for (const Option &o : {Option(options)...})
            if ((e = o(Applier{nullptr, nullptr, this})) != Error::None) return e;
        return e;
    }
    // :83-165.  n == 0 with Error::None is io.EOF.
    Result Read(uint8_t *p, size_t len) {
        if (state_ == State::Done) return {0, err_};
        if (state_ == State::Initial) {
            const Bytes raw((std::istreambuf_iterator<char>(*src_)), std::istreambuf_iterator<char>());
            const uint32_t flags = (block_checksum ? LZ4HIP_F_BLOCK_CHECKSUM : 0u) |
                                   (content_checksum ? LZ4HIP_F_CONTENT_CHECKSUM : 0u) |
                                   (content_size > 0 ? LZ4HIP_F_CONTENT_SIZE : 0u);
            std::vector<uint32_t> lens;
            const int64_t k = detail::frame_compress(raw, block_size, flags, (uint32_t)level,
                                                     content_size, out_, handler ? &lens : nullptr);
            if (k < 0) {
                state_ = State::Done;
                return {0, err_ = detail::frame_error(k)};
            }
            out_.resize((size_t)k);
            for (uint32_t size : lens) handler((int)size);  // :126,142
            state_ = State::Reading;
        }
        const size_t k = std::min(len, out_.size() - pos_);
        memcpy(p, out_.data() + pos_, k);
        pos_ += k;
        return {(int64_t)k, Error::None};
    }
    // :167-174: keeps the options
    void Reset(std::istream &src) {
        src_ = &src;
        out_.clear();
        pos_ = 0;
        state_ = State::Initial;
        err_ = Error::None;
    }

    uint32_t block_size = Block4Mb;  // DefaultBlockSizeOption
    bool block_checksum = false;
    bool content_checksum = true;  // DefaultChecksumOption
    uint64_t content_size = 0;
    CompressionLevel level = Fast;
    BlockDoneHandler handler;

  private:
    enum class State { Initial, Reading, Done };
    std::istream *src_;
    Bytes out_;
    size_t pos_ = 0;
    State state_ = State::Initial;
    Error err_ = Error::None;
};

// reader.go:265-275: (true, nil) | (false, nil) for a bad magic | (false, err).
// A header cut short is Go's io.ErrUnexpectedEOF: reported as InvalidFrame here.
inline std::pair<bool, Error> ValidFrameHeader(const Bytes &in) {
    const int rc = lz4hip_frame_parse_header(in.data(), in.size(), nullptr, nullptr, nullptr, nullptr);
    if (rc == LZ4HIP_OK) return {true, Error::None};
    if (rc == LZ4HIP_ERR_FRAME) return {false, Error::None};
    if (rc == LZ4HIP_ERR_SHORT) return {false, Error::InvalidFrame};
    return {false, detail::frame_error(rc)};
}

// ---- the options themselves: options.go:44-242 ----
namespace detail {
// the options that options.go gives to both *Writer and *CompressingReader
template <class F> Error frame_option(Applier a, F set) {
    if (a.w) return set(*a.w);
    if (a.c) return set(*a.c);
    return Error::OptionNotApplicable;
}
}  // namespace detail
inline Option BlockSizeOption(uint32_t size) {
    return [size](Applier a) {
        return detail::frame_option(a, [size](auto &t) {
            if (size != Block64Kb && size != Block256Kb && size != Block1Mb && size != Block4Mb)
                return Error::OptionInvalidBlockSize;
            t.block_size = size;
            return Error::None;
        });
    };
}
inline Option BlockChecksumOption(bool flag) {
    return [flag](Applier a) {
        return detail::frame_option(a, [flag](auto &t) {
            t.block_checksum = flag;
            return Error::None;
        });
    };
}
inline Option ChecksumOption(bool flag) {
    return [flag](Applier a) {
        return detail::frame_option(a, [flag](auto &t) {
            t.content_checksum = flag;
            return Error::None;
        });
    };
}
inline Option SizeOption(uint64_t size) {
    return [size](Applier a) {
        return detail::frame_option(a, [size](auto &t) {
            t.content_size = size;
            return Error::None;
        });
    };
}
// The GPU path always takes all buffered blocks as one batch, and the Go output does
// not depend on n either; accepted for Writer and Reader like the reference.
inline Option ConcurrencyOption(int n) {
    return [n](Applier a) {
        if (a.w) a.w->concurrency = n;
        else if (a.r) a.r->concurrency = n;
        else return Error::OptionNotApplicable;
        return Error::None;
    };
}
inline Option CompressionLevelOption(uint32_t level) {
    return [level](Applier a) {
        return detail::frame_option(a, [level](auto &t) {
            bool ok = level == Fast;
            for (uint32_t l = Level1; l <= Level9; l <<= 1) ok = ok || level == l;
            if (!ok) return Error::OptionInvalidCompressionLevel;
            t.level = (CompressionLevel)level;
            return Error::None;
        });
    };
}
// options.go:197-220.  Writer / CompressingReader: len(block.Data) per block (and, from
// ReadFrom, the bytes read for it); Reader: decoded bytes per block portion handed out.
inline Option OnBlockDoneOption(BlockDoneHandler handler) {
    return [handler](Applier a) {
        if (a.w) a.w->handler = handler;
        else if (a.r) a.r->handler = handler;
        else if (a.c) a.c->handler = handler;
        else return Error::OptionNotApplicable;
        return Error::None;
    };
}
inline Option LegacyOption(bool legacy) {
    return [legacy](Applier a) {
        if (!a.w) return Error::OptionNotApplicable;
        a.w->legacy = legacy;
        return Error::None;
    };
}

}  // namespace lz4
#endif  // LZ4_HPP
